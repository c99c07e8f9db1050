"""One-round launches: the same batch as 2048 full wavefronts (two per SIMD) or as 4096 wavefronts of half the work (four
per SIMD), with a kernel that fits four wavefronts per SIMD (see-through: 128 registers) - does the shorter chain of a
smaller wavefront shorten the launch?"""
import os, sys
os.environ["MGRID_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import exp_tiles

exp_tiles.CASES[6] = ("MultiGrid-Empty-16x16-v0", 2, 65536)   # see-through kernels, 2 agents: 32 environments fill a wavefront
exp_tiles.CASES[7] = ("MultiGrid-Empty-16x16-v0", 4, 65536)
for cid, tiles in ((6, (None, "32,1", "16,1", "16,2", "8,1")), (7, (None, "16,2", "16,1", "8,1", "8,2"))):
    for t in tiles:
        exp_tiles.run(cid, t)
