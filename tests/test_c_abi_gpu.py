"""A native client of the C ABI: tests/c_abi/abi_forward.cpp includes include/qsae.h, links libqsae_hip.so and the
CPU oracle (as the checker), runs the BinarySAE forward on its own hipMalloc'd buffers and compares bit for bit."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "tests" / "c_abi" / "abi_forward.cpp"
EXE = ROOT / "tests" / "c_abi" / "abi_forward"


def build_client() -> Path:
    from quantizedsae_amd import build
    build.build_native()
    lib, oracle = ROOT / "quantizedsae_amd" / "lib", ROOT / "oracle"
    if not (oracle / "libqsae_oracle.so").exists():
        subprocess.run(["make", "-C", str(oracle)], check=True)
    deps = [SRC, ROOT / "include" / "qsae.h"]
    if EXE.exists() and all(EXE.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return EXE
    # a plain host program: g++, the HIP runtime API header and three shared libraries (rpaths relative to the binary)
    cmd = ["g++", "-O2", "-std=c++17", str(SRC), "-I", str(ROOT / "include"), "-I", "/opt/rocm/include",
           "-D__HIP_PLATFORM_AMD__", "-o", str(EXE), f"-L{lib}", "-lqsae_hip", f"-L{oracle}", "-lqsae_oracle",
           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,$ORIGIN/../../quantizedsae_amd/lib", "-Wl,-rpath,$ORIGIN/../../oracle",
           "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return EXE


def test_native_client_builds():
    """CPU: the client compiles and links against the header and the library (no GPU call)."""
    assert build_client().exists()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2304, 512, 8192, 16, 4), (300, 64, 1024, 5, 3), (4096, 256, 16384, 32, 8)])
def test_native_client_forward_matches_oracle(shape):
    exe = build_client()
    out = subprocess.run([str(exe)] + [str(v) for v in shape], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("PASS"), out.stdout + out.stderr
