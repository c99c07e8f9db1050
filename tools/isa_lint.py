"""Scan a device assembly listing (hipcc -S --cuda-device-only, or the .s that -save-temps leaves) for vector
instructions placed before the lanes of a divergent region are switched back on (see multigrid_amd/_isa_lint.py).
    python tools/isa_lint.py file.s   -> prints the occurrences, exit status 1 if there are any"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_amd._isa_lint import find_masked_copies  # noqa: E402

hits = find_masked_copies(sys.argv[1])
for k, label, ins in hits:
    print(f"{k}\n  {label}: {len(ins)} vector instruction(s) before the lanes are switched back on: {ins[:3]}")
print(f"{len(hits)} occurrence(s)")
sys.exit(1 if hits else 0)
