"""Register / scratch use of the kernels of one source file, from the compiler's own resource remarks
(-Rpass-analysis=kernel-resource-usage; nothing is written).  usage: python tools/kernel_regs.py encode_topk.hip [substring ...]"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from quantizedsae_amd.build import CSRC, FLAGS, _hipcc  # noqa: E402


def kernel_table(src: str, debug: bool = False):
    cmd = [_hipcc()] + FLAGS + (["-DQSAE_DEBUG_BUILD=1"] if debug else []) + \
          ["-Rpass-analysis=kernel-resource-usage", "-c", str(CSRC / src), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[bytes/\w+\])?(?: \[waves/SIMD\])?: (\S+) \[-Rpass", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            cur = {"name": subprocess.run(["c++filt", val], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    return rows


if __name__ == "__main__":
    src, pats = sys.argv[1], sys.argv[2:]
    for r in kernel_table(src):
        if not pats or any(p in r["name"] for p in pats):
            print(f"vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>4} sgpr {r.get('TotalSGPRs','?'):>4} scratch "
                  f"{r.get('ScratchSize','?'):>5} spill {r.get('VGPRs Spill','?'):>3} occ {r.get('Occupancy','?'):>2}  {r['name'][:120]}")
