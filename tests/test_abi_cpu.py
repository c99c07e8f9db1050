"""The C-ABI library loads on a machine without a GPU and exports every symbol include/mgrid.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mgrid.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mg_[a-z_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    from multigrid_amd import _lib
    assert set(declared_symbols()) == set(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    from multigrid_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libmgrid_hip.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.mg_abi_version() == _lib.ABI_VERSION


def test_bad_arguments_fail_without_a_gpu():
    from multigrid_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libmgrid_hip.so not built")
    L = _lib.lib()
    h = ctypes.c_void_p()
    cfg = _lib.MgConfig(kind=0, width=8, height=8, num_agents=2, view_size=4, max_steps=10, num_envs=4)
    assert L.mg_create(ctypes.byref(cfg), ctypes.byref(h)) == -1  # MG_EINVAL: view size must be odd, 3 .. 11
    assert b"view_size" in L.mg_last_error(None)
    cfg.view_size = 13
    assert L.mg_create(ctypes.byref(cfg), ctypes.byref(h)) == -1 and b"view_size" in L.mg_last_error(None)
    cfg = _lib.MgConfig(kind=3, width=13, height=9, num_agents=2, view_size=7, max_steps=10, num_envs=4, num_rooms=8)  # 8 rooms need 13 x 17
    assert L.mg_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"LockedHallway" in L.mg_last_error(None)
    cfg = _lib.MgConfig(kind=0, width=8, height=8, num_agents=2, view_size=7, max_steps=10, num_envs=4, batch_envs=-1)
    assert L.mg_create(ctypes.byref(cfg), ctypes.byref(h)) == -1 and b"batch_envs" in L.mg_last_error(None)
    assert L.mg_step_shards(None, 0, None, None, None, None) == -1  # an empty shard list
    assert b"mg_step_shards" in L.mg_last_error(None)
    assert L.mg_episode_stats(4, 2, None, None, None, None, None, None, None, None, None) == -1
    assert b"mg_episode_stats" in L.mg_last_error(None)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "multigrid_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, os.path.join(dirpath, f)


def test_build_check_recognises_a_masked_copy(tmp_path):
    """The scan `_lib.build` runs over the device assembly: a vector instruction between a join label and the
    instruction that switches the lanes back on is reported, the regular order is not."""
    from multigrid_amd._isa_lint import find_masked_copies
    bad = tmp_path / "bad.s"
    bad.write_text("_Zkernel_a: ; @_Zkernel_a\n\ts_cbranch_execz .LBB1_2\n\tv_add_u32_e32 v1, 1, v1\n.LBB1_2:\n"
                   "\tv_mov_b32_e32 v67, v63\n\ts_or_b64 exec, exec, s[2:3]\n\ts_endpgm\n")
    good = tmp_path / "good.s"
    good.write_text("_Zkernel_b: ; @_Zkernel_b\n\ts_cbranch_execz .LBB1_2\n\tv_add_u32_e32 v1, 1, v1\n.LBB1_2:\n"
                    "\ts_or_b64 exec, exec, s[2:3]\n\tv_mov_b32_e32 v67, v63\n\ts_endpgm\n")
    hits = find_masked_copies(str(bad))
    assert len(hits) == 1 and hits[0][0] == "_Zkernel_a" and hits[0][1] == ".LBB1_2"
    assert find_masked_copies(str(good)) == []
