#!/bin/sh
# Diagnostic build with phase-ablation switches (MGRID_DBG / MGRID_LDS_PAD); used only by tools/exp_phase.py.
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -Os -std=c++17 --offload-arch=gfx950 -fPIC -shared -DMGRID_EXPERIMENT -Iinclude -Imultigrid_amd/csrc multigrid_amd/csrc/mgrid.hip -o tools/libmgrid_exp.so
