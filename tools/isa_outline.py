"""Outline of one kernel's assembly: labels, branches and memory operations, with the number of VALU /
SALU instructions between them."""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 10 ** 9)
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith("E"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
v = s = 0
n = 0
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith((";", ".")) and not t.startswith(".LBB"):
        continue
    n += 1
    if not (lo <= n <= hi):
        continue
    op = t.split()[0]
    show = t.startswith(".LBB") or op.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_", "s_cbranch", "s_branch", "s_waitcnt", "s_load", "s_memtime", "s_barrier", "v_readlane", "v_writelane"))
    if op.startswith(("v_readlane", "v_writelane")):
        v += 1
        continue
    if show:
        if v or s:
            print(f"        ... {v} valu, {s} salu")
            v = s = 0
        print(f"{n:6d}  {t.split(';')[0].rstrip()}")
    elif op.startswith("v_"):
        v += 1
    else:
        s += 1
