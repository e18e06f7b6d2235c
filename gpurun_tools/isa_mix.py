"""Instruction mix of one kernel of an ISA dump (hipcc -S --cuda-device-only), per region between two s_barrier instructions:
   python gpurun_tools/isa_mix.py file.s kernel_name_prefix
Static counts (every instruction once, loops and both sides of branches included): an upper bound of the executed path."""
import re, sys
src, pref = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(pref) and l.rstrip().endswith(":") or (l.startswith(pref) and ": " in l))
region, regions = {}, []
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    return "other"
for l in lines[start + 1:]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    if op == "s_barrier":
        regions.append(region); region = {}
        continue
    c = cls(op)
    region[c] = region.get(c, 0) + 1
    if op == "s_endpgm":
        break
regions.append(region)
tot = {}
for k, r in enumerate(regions):
    print(k, r)
    for c, n in r.items():
        tot[c] = tot.get(c, 0) + n
print("total", tot)
