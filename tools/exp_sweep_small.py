"""Timing of hand-picked tile shapes (see exp_tiles.py); SWEEP="cfg:NE,G;NE,G ..." selects them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import exp_tiles
for part in os.environ.get("SWEEP", "5:auto;8,1;8,2;8,3;8,4;4,4;4,6;6,3;7,3").split():
    cid, tiles = part.split(":")
    for t in tiles.split(";"):
        exp_tiles.run(int(cid), None if t == "auto" else t)
