"""ctypes binding of ``libmgrid_hip.so`` (declared in ``include/mgrid.h``).

There is no CPU fallback: if the HIP library has not been built, importing a symbol
from here raises, and every product path stops with it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("MGRID_LIB", os.path.join(PKG_DIR, "libmgrid_hip.so"))
SRC_PATH = os.path.join(PKG_DIR, "csrc", "mgrid.hip")
INCLUDE_DIR = os.path.join(ROOT, "include")

MG_OK = 0
KIND_EMPTY, KIND_REDBLUEDOORS, KIND_BLOCKEDUNLOCKPICKUP, KIND_LOCKEDHALLWAY, KIND_COMPETITIVE = range(5)
AUTORESET_NONE, AUTORESET_SAME_STEP, AUTORESET_NEXT_STEP = range(3)
NUM_STATS = 8
ABI_VERSION = 6
ONE_HOT_DIM = 24  # 11 types + 6 colours + 7 states, multigrid/wrappers.py:95

EXPORTS = ("mg_abi_version", "mg_create", "mg_destroy", "mg_last_error", "mg_upload_layouts", "mg_generate_layouts",
           "mg_set_reward_schemes",
           "mg_reset", "mg_step", "mg_step_shards", "mg_step_n", "mg_gen_obs", "mg_gen_obs_of_state", "mg_get_state", "mg_set_state",
           "mg_stats", "mg_episode_stats", "mg_agents_mask")


class MgConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "kind", "width", "height", "num_agents", "view_size", "max_steps", "see_through_walls",
        "allow_agent_overlap", "joint_reward", "success_any", "failure_any", "num_rooms",
        "num_envs", "device", "autoreset", "env_id_offset", "death_match", "per_agent_outputs")] + [
        ("seed", ctypes.c_uint64), ("batch_envs", ctypes.c_int32), ("reserved0", ctypes.c_int32)]


class MgOut(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "image_dev", "direction_dev", "reward_dev", "terminated_dev", "truncated_dev", "onehot_dev")]


class MgLayoutSpec(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "size", "random_start", "start_x", "start_y", "start_dir", "max_hallway_keys",
        "max_keys_per_room")] + [("agent_color", ctypes.c_uint8 * 64)]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if not force and os.path.exists(LIB_PATH) and \
            os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(SRC_PATH),
                                              os.path.getmtime(os.path.join(INCLUDE_DIR, "mgrid.h"))):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -Os, measured: the step kernel is 7 % faster than at -O3 (smaller code, and 126 instead of 129 vector
    # registers, which lets a fourth wavefront onto every SIMD)
    # The build keeps its intermediate files in a scratch directory so that the device assembly can be checked
    # for a code pattern this compiler was seen to get wrong (see _isa_lint.py) before the library is put in place.
    import glob
    import shutil
    import tempfile
    from ._isa_lint import find_masked_copies
    tmp = tempfile.mkdtemp(prefix="mgrid_build_")
    try:
        out = os.path.join(tmp, "libmgrid_hip.so")
        cmd = [hipcc, "-Os", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-save-temps",
               "-I", INCLUDE_DIR, SRC_PATH, "-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=tmp)
        listings = glob.glob(os.path.join(tmp, "*gfx950*.s"))
        if not listings:
            raise RuntimeError("the build left no device assembly to check")
        for path in listings:
            hits = find_masked_copies(path)
            if hits:
                what = "; ".join(f"{k} {label}: {ins[0]}" for k, label, ins in hits[:3])
                raise RuntimeError(f"device code writes vector registers before a divergent region's lanes are "
                                   f"switched back on ({len(hits)} place(s): {what}); change the source around the "
                                   "named kernel until the compiler stops doing that")
        shutil.move(out, LIB_PATH)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (there is no CPU fallback for the MI355X path)")
        L = ctypes.CDLL(LIB_PATH)
        vp, i32 = ctypes.c_void_p, ctypes.c_int32
        L.mg_abi_version.restype = ctypes.c_int
        L.mg_create.argtypes = [ctypes.POINTER(MgConfig), ctypes.POINTER(vp)]
        L.mg_destroy.argtypes = [vp]
        L.mg_last_error.argtypes = [vp]
        L.mg_last_error.restype = ctypes.c_char_p
        L.mg_upload_layouts.argtypes = [vp, vp, vp, vp, i32]
        if hasattr(L, "mg_set_reward_schemes"):  # (tuning builds of older sources loaded through MGRID_LIB lack it)
            L.mg_set_reward_schemes.argtypes = [vp, vp]
        L.mg_generate_layouts.argtypes = [vp, ctypes.POINTER(MgLayoutSpec), i32, ctypes.c_uint64]
        L.mg_reset.argtypes = [vp, vp, vp, ctypes.POINTER(MgOut), vp]
        L.mg_step.argtypes = [vp, vp, vp, ctypes.POINTER(MgOut), vp]
        if hasattr(L, "mg_step_shards"):
            L.mg_step_shards.argtypes = [ctypes.POINTER(vp), i32, vp, vp, ctypes.POINTER(MgOut), ctypes.POINTER(vp)]
        L.mg_step_n.argtypes = [vp, i32, vp, vp, ctypes.POINTER(MgOut), vp]
        L.mg_gen_obs.argtypes = [vp, ctypes.POINTER(MgOut), vp]
        L.mg_gen_obs_of_state.argtypes = [ctypes.POINTER(MgConfig), vp, vp, ctypes.POINTER(MgOut), vp]
        L.mg_get_state.argtypes = [vp] * 7
        L.mg_set_state.argtypes = [vp] * 7
        L.mg_stats.argtypes = [vp, vp, vp]
        if hasattr(L, "mg_agents_mask"):
            L.mg_agents_mask.argtypes = [vp, vp, vp, vp]
        if hasattr(L, "mg_episode_stats"):
            L.mg_episode_stats.argtypes = [i32, i32] + [vp] * 9
        for name in EXPORTS:
            if name not in ("mg_last_error",) and hasattr(L, name):
                getattr(L, name).restype = ctypes.c_int
        _lib = L
    return _lib


class MgError(RuntimeError):
    pass


def check(rc: int, handle=None):
    if rc != MG_OK:
        msg = lib().mg_last_error(handle)
        raise MgError(f"libmgrid_hip error {rc}: {msg.decode() if msg else '?'}")
