"""Instruction histogram of one kernel in an assembly listing (hipcc -S --cuda-device-only)."""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith("E"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
ops = collections.Counter()
for l in lines[start + 1:end + 1]:
    l = l.strip()
    if not l or l.startswith((";", ".")) or l.endswith(":"):
        continue
    ops[l.split()[0]] += 1
tot = sum(ops.values())
cls = collections.Counter()
for k, v in ops.items():
    c = "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else "vmem"
    cls[c] += v
print(f"{key}: lines {start}-{end}, {tot} instructions", dict(cls))
for k, v in ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"  {k:28s} {v}")
