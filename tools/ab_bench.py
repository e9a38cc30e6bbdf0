#!/usr/bin/env python3
"""Same-box A/B of library builds on the headline step (boxes differ by +-0.04 ms; variants have to meet on one).

  python tools/ab_bench.py build TAG [-DNAME=VALUE ...]     # in the build container: build/ab/TAG/libqsae_hip.so
  python tools/ab_bench.py run TAG [TAG ...] [--rounds 3] [--steps 60]
                                                            # on the GPU box: one child process per (round, tag), interleaved
  (TAG "base" = the product library of the tree, no extra defines)

Switches the sources know (each keeps the earlier form of one round-3 change, for the numbers quoted in DESIGN.md 8):
  -DQSAE_AB_NO_SLICED=1      refinement as one launch at every batch size (no select / slice-major chains / rank)
  -DQSAE_AB_NARROW_DECODE=1  row decode with four dictionary rows in flight
  -DQSAE_AB_NO_DUAL=1        no second chains on the low lanes of the one-launch refinement
  -DQSAE_AB_FULL_BISECT=1    bisection of the approximate k-th value down to the last key bit

Variant libraries live under build/ab/ (git-ignored, they travel with the gpurun snapshot).  The child process points the ctypes
loader at the variant before the package's first call; nothing in the package knows about variants."""
import argparse
import statistics
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
AB = ROOT / "build" / "ab"


def lib_of(tag: str) -> Path:
    from quantizedsae_amd import build as qb
    return qb.LIB if tag == "base" else AB / tag / "libqsae_hip.so"


def build_variant(tag: str, defines):
    from quantizedsae_amd import build as qb
    out = AB / tag
    (out / "obj").mkdir(parents=True, exist_ok=True)
    hipcc = qb._hipcc()
    objs = []
    procs = []
    for s in qb.SOURCES:
        obj = out / "obj" / (Path(s).stem + ".o")
        objs.append(obj)
        procs.append((s, subprocess.Popen([hipcc] + qb.FLAGS + list(defines) + ["-c", str(qb.CSRC / s), "-o", str(obj)],
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit(f"hipcc failed for {s}:\n{log}")
    lib = out / "libqsae_hip.so"
    subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={qb.ARCH}", "-o", str(lib)] + [str(o) for o in objs], check=True)
    print(lib)


def one(libpath: str, steps: int, mode: str):
    import torch
    from quantizedsae_amd import _lib
    _lib.LIB_PATH = Path(libpath)
    import bench
    from quantizedsae_amd import ops
    dev = torch.device("cuda:0")
    model = bench.build_model(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1000)
    xs = [torch.randn((bench.ROWS_PER_GPU, bench.D), device=dev, generator=g) for _ in range(4)]
    model.decoder.packed()
    acc = torch.zeros((), dtype=torch.float64, device=dev)

    def run(n):
        pending, xprev = None, None
        for i in range(n):
            xi = xs[i % 4]
            h = model.forward_submit(xi, slot=i % 2)
            if pending is not None:
                _l, rec, _p = pending.result()
                ops.sq_err_sum(rec, xprev, acc)
            pending, xprev = h, xi
        _l, rec, _p = pending.result()
        ops.sq_err_sum(rec, xprev, acc)

    run(5)
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / steps * 1e3)
    print(f"RESULT {min(best):.4f} {statistics.median(best):.4f} {float(acc):.6e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["build", "run", "one"])
    ap.add_argument("tags", nargs="*")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--mode", default="headline")
    args, extra = ap.parse_known_args()
    if args.cmd == "build":
        build_variant(args.tags[0], [e for e in extra if e.startswith("-D")])
    elif args.cmd == "one":
        one(args.tags[0], args.steps, args.mode)
    else:
        res = {t: [] for t in args.tags}
        for r in range(args.rounds):
            for t in args.tags:
                p = subprocess.run([sys.executable, __file__, "one", str(lib_of(t)), "--steps", str(args.steps), "--mode", args.mode],
                                   capture_output=True, text=True)
                line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")]
                if p.returncode != 0 or not line:
                    raise SystemExit(f"{t}: child failed\n{p.stdout}\n{p.stderr}")
                mn, med, acc = line[0].split()[1:]
                res[t].append(float(mn))
                print(f"round {r} {t:16s} min {mn} median {med} ms/step   sq_err {acc}", flush=True)
        for t in args.tags:
            print(f"{t:16s} best {min(res[t]):.4f}  median of rounds {statistics.median(res[t]):.4f} ms/step")


if __name__ == "__main__":
    main()
