"""Scan every kernel of gesture2vec_amd/csrc for instructions that betray a lost address space or a spill:
flat_load / flat_store / flat_atomic (a pointer the compiler could not prove global: such accesses count in vmcnt AND lgkmcnt and make
every wait conservative -- round 6 found the streaming GRU BPTT's 24 prefetched vectors per thread this way) and scratch_ (spills,
or a by-value struct copied to private memory).  CPU only: hipcc -S --cuda-device-only per file (about a minute in all).
usage: python gpurun_tools/isa_scan.py [file.hip ...]"""
import os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gesture2vec_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
tmp = tempfile.mkdtemp()


def asm(f):
    out = os.path.join(tmp, f + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-o", out,
                    os.path.join(CSRC, f)], check=True, stderr=subprocess.DEVNULL)
    return f, out


bad = 0
with ThreadPoolExecutor(4) as ex:
    for f, path in ex.map(asm, files):
        cur, stats = None, {}
        for line in open(path):
            m = re.match(r"^([A-Za-z_][\w$]*):\s", line)
            if m and not line.startswith(".L"):
                cur = m.group(1)
                stats[cur] = [0, 0, 0, 0]
                continue
            if cur is None:
                continue
            if "flat_load" in line: stats[cur][0] += 1
            elif "flat_store" in line or "flat_atomic" in line: stats[cur][1] += 1
            elif "scratch_" in line: stats[cur][2] += 1
            elif "v_mfma" in line: stats[cur][3] += 1
        for k, v in stats.items():
            if v[0] or v[1] or v[2]:
                bad += 1
                print(f"{f}: {k[:90]}  flat loads {v[0]}, flat stores/atomics {v[1]}, scratch ops {v[2]}, MFMAs {v[3]}")
print(f"{bad} kernel(s) with flat or scratch instructions in {len(files)} file(s)")
