"""Runs the dispatch sweeps of tests/test_gpu_instantiations.py (every one checked against the oracle) and writes
which instantiations of the tile kernel the library was built with, which of them the sweeps launched, and the
registers each one needs.    python tools/instantiation_census.py profiles/r03_instantiations.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest  # noqa: E402
import test_gpu_instantiations as ti  # noqa: E402
from multigrid_amd import _lib  # noqa: E402

mp = pytest.MonkeyPatch()
for view in (3, 5, 7):
    for see in (0, 1):
        ti.sweep(view, see, mp)
for view in (9, 11):
    for see in (0, 1):
        ti.sweep_wide(view, see)
mp.undo()
name = lambda t: ",".join(str(x) for x in t)
compiled = {name(t) for t in _lib.compiled_instantiations()}
launched = {name(t): n for t, n in _lib.launch_census().items()}
registers = json.load(open(_lib.REGISTERS_PATH))
out = dict(template_arguments="V,SEE,MODE,WALLED,OH,MULTI,COMP,CROWD,STG (MODE 0 step, 1 gen_obs, 2 reset)",
           compiled=len(compiled), launched_and_checked_against_the_oracle=len(set(launched) & compiled),
           never_launched=sorted(compiled - set(launched)), launched_but_not_in_the_register_report=sorted(set(launched) - compiled),
           launches={k: launched[k] for k in sorted(launched)},
           vector_registers={k.replace("void mg_tile_kernel", "").replace("(KParams)", ""): v["vgpr_count"] for k, v in sorted(registers.items()) if "mg_tile_kernel" in k},
           spilled_vector_registers=sum(v["vgpr_spill_count"] for k, v in registers.items() if "mg_tile_kernel" in k))
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(f"{out['compiled']} compiled, {out['launched_and_checked_against_the_oracle']} launched, never launched: {out['never_launched']}")
sys.exit(1 if out["never_launched"] else 0)
