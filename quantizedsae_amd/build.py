"""Build the gfx950 HIP library (`quantizedsae_amd/lib/libqsae_hip.so`) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the
resulting .so travels with the repo snapshot to the GPU box (it is git-ignored, not
gpurun-ignored).  No CUDA path, no hipify, no multi-backend switch: gfx950 only.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libqsae_hip.so"
DEBUG_LIB = LIBDIR / "libqsae_hip_debug.so"      # same sources + -DQSAE_DEBUG_BUILD: qsae_debug_* switches and ablation kernels
OBJDIR = PKG / "lib" / "obj"

SOURCES = ["encode.hip", "topk.hip", "binary.hip", "dense_dec.hip", "encode_topk.hip", "misc.hip", "analysis.hip"]
HEADERS = sorted(p.name for p in (Path(__file__).resolve().parent / "csrc").glob("*.h"))   # every header: all sources rebuild
ARCH = "gfx950"
FLAGS = ["-O3", f"--offload-arch={ARCH}", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: the HIP library cannot be built (ROCm 7.x required)")
    return exe


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_native(force: bool = False, verbose: bool = False, debug: bool = False) -> Path:
    """Compile every .hip source for gfx950 and link the shared library.  Idempotent.

    debug=False: the product library (no qsae_debug_* symbol, no ablation kernel, every tuning switch a constant).
    debug=True: libqsae_hip_debug.so for tools/ and the tests that force a path on a small shape."""
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    hdrs = [CSRC / h for h in HEADERS] + [PKG.parent / "include" / "qsae.h"]
    objdir = OBJDIR / "debug" if debug else OBJDIR
    lib = DEBUG_LIB if debug else LIB
    flags = FLAGS + (["-DQSAE_DEBUG_BUILD=1"] if debug else [])
    objdir.mkdir(parents=True, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for src in srcs:
        obj = objdir / (src.stem + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + flags + ["-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for msg in ex.map(compile_one, jobs):
                if verbose and msg:
                    print(msg)
    objs = [objdir / (s.stem + ".o") for s in srcs]
    if force or jobs or _stale(lib, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(lib)] + [str(o) for o in objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


def exported_symbols(lib: Path):
    """Dynamic symbols of a built library (nm -D), used by the tests and by __graft_entry__.build()."""
    r = subprocess.run(["nm", "-D", "--defined-only", str(lib)], capture_output=True, text=True, check=True)
    return sorted(line.split()[-1] for line in r.stdout.splitlines() if line.strip())


if __name__ == "__main__":
    import sys
    print(build_native(force="--force" in sys.argv, verbose=True))
    print(build_native(force="--force" in sys.argv, verbose=True, debug=True))
