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
ABI_VERSION = 8
ONE_HOT_DIM = 24  # 11 types + 6 colours + 7 states, multigrid/wrappers.py:95

EXPORTS = ("mg_abi_version", "mg_create", "mg_destroy", "mg_last_error", "mg_upload_layouts", "mg_generate_layouts",
           "mg_set_reward_schemes",
           "mg_reset", "mg_step", "mg_step_shards", "mg_step_n", "mg_gen_obs", "mg_gen_obs_of_state", "mg_get_state", "mg_set_state",
           "mg_stats", "mg_episode_stats", "mg_agents_mask", "mg_launch_census")


class MgConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "kind", "width", "height", "num_agents", "view_size", "max_steps", "see_through_walls",
        "allow_agent_overlap", "joint_reward", "success_any", "failure_any", "num_rooms",
        "num_envs", "device", "autoreset", "env_id_offset", "death_match", "per_agent_outputs")] + [
        ("seed", ctypes.c_uint64), ("batch_envs", ctypes.c_int32), ("reserved0", ctypes.c_int32)]


class MgOut(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "image_dev", "direction_dev", "reward_dev", "terminated_dev", "truncated_dev", "onehot_dev",
        "rows_dev", "reward64_dev", "info_dev")]


class MgLayoutSpec(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "size", "random_start", "start_x", "start_y", "start_dir", "max_hallway_keys",
        "max_keys_per_room")] + [("agent_color", ctypes.c_uint8 * 64)]


def kernel_resources(listing: str):
    """Registers and spills of every kernel in a device assembly listing (the `amdhsa.kernels` metadata hipcc
    writes at its end): ``{mangled name: {vgpr_count, agpr_count, sgpr_count, sgpr_spill_count, vgpr_spill_count,
    private_segment_fixed_size, group_segment_fixed_size}}``."""
    import re
    text = open(listing).read()
    at = text.find("amdhsa.kernels:")
    out = {}
    if at < 0:
        return out
    for entry in re.split(r"\n  - ", text[at:])[1:]:
        name = re.search(r"\.name:\s+(\S+)", entry)
        if not name:
            continue
        rec = {}
        for key in ("vgpr_count", "agpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count",
                    "private_segment_fixed_size", "group_segment_fixed_size"):
            m = re.search(r"\.%s:\s+(\d+)" % key, entry)
            rec[key] = int(m.group(1)) if m else 0
        out[name.group(1)] = rec
    return out


def _demangle(names):
    import shutil
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt", path="/opt/rocm/lib/llvm/bin")
    if not tool:
        return list(names)
    r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
    got = r.stdout.split("\n")[:len(names)]
    return got if r.returncode == 0 and len(got) == len(names) else list(names)


REGISTERS_PATH = os.path.join(PKG_DIR, "kernel_registers.json")


def build(force: bool = False, verbose: bool = False, extra_flags=(), out_path: str | None = None) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU).

    The build keeps its intermediate files in a scratch directory and looks at the device assembly before the
    library is put in place: every kernel's register and spill counts are listed (``kernel_registers.json`` beside
    the library; printed with ``verbose``), a step kernel that spills vector registers to scratch memory fails the
    build, and so does a code pattern this compiler was seen to get wrong (see _isa_lint.py)."""
    target = out_path or LIB_PATH
    sources = [SRC_PATH, os.path.join(PKG_DIR, "csrc", "mgrid_experiment.h"), os.path.join(INCLUDE_DIR, "mgrid.h")]
    if not force and os.path.exists(target) and os.path.getmtime(target) >= max(os.path.getmtime(f) for f in sources):
        return target
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -Os, measured: the step kernel is 7 % faster than at -O3 (smaller code, and 126 instead of 129 vector
    # registers, which lets a fourth wavefront onto every SIMD)
    import glob
    import json
    import shutil
    import tempfile
    from ._isa_lint import find_masked_copies
    tmp = tempfile.mkdtemp(prefix="mgrid_build_")
    try:
        out = os.path.join(tmp, "libmgrid_hip.so")
        cmd = [hipcc, "-Os", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-save-temps",
               "-I", INCLUDE_DIR, *extra_flags, SRC_PATH, "-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=tmp)
        listings = glob.glob(os.path.join(tmp, "*gfx950*.s"))
        if not listings:
            raise RuntimeError("the build left no device assembly to check")
        report = {}
        for path in listings:
            hits = find_masked_copies(path)
            if hits:
                what = "; ".join(f"{k} {label}: {ins[0]}" for k, label, ins in hits[:3])
                raise RuntimeError(f"device code writes vector registers before a divergent region's lanes are "
                                   f"switched back on ({len(hits)} place(s): {what}); change the source around the "
                                   "named kernel until the compiler stops doing that")
            res = kernel_resources(path)
            for mangled, plain in zip(res, _demangle(list(res))):
                report[plain.replace("(anonymous namespace)::", "")] = res[mangled]
        # template arguments of mg_tile_kernel: view, see-through, MODE (0 = step), ...
        # (a private segment without spilled vector registers is a frame object the compiler reserved and then never
        # touched - the listing has no scratch instruction for it; it is reported, not refused)
        spilling = [k for k, r in report.items() if "mg_tile_kernel" in k and r["vgpr_spill_count"]]
        if verbose:
            for k in sorted(report):
                r = report[k]
                print(f"  {r['vgpr_count']:3d} vgpr {r['sgpr_count']:3d} sgpr  spills {r['sgpr_spill_count']:3d} s / "
                      f"{r['vgpr_spill_count']:3d} v  scratch {r['private_segment_fixed_size']:4d} B  {k}")
        if spilling and os.environ.get("MGRID_ALLOW_SPILLS") != "1":  # (the switch is for work in progress only)
            raise RuntimeError(f"{len(spilling)} instantiation(s) of the tile kernel spill vector registers to scratch "
                               f"memory, e.g. {spilling[0]}: {report[spilling[0]]}")
        # a new inode, renamed into place: processes that have the old library mapped keep the old file
        staged = f"{target}.tmp.{os.getpid()}"
        shutil.copy2(out, staged)
        os.replace(staged, target)
        if out_path is None:
            with open(REGISTERS_PATH, "w") as f:
                json.dump(report, f, indent=1, sort_keys=True)
                f.write("\n")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return target


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
        if hasattr(L, "mg_launch_census"):
            L.mg_launch_census.argtypes = [ctypes.c_char_p, i32]
        for name in EXPORTS:
            if name not in ("mg_last_error",) and hasattr(L, name):
                getattr(L, name).restype = ctypes.c_int
        _lib = L
    return _lib


def launch_census() -> dict:
    """``{(V, SEE, MODE, WALLED, OH, MULTI, COMP, CROWD, STG): launches}`` of this process so far (``mg_launch_census``)."""
    buf = ctypes.create_string_buffer(1 << 16)
    check(lib().mg_launch_census(buf, len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        what, n = line.split(":")
        out[tuple(int(x) for x in what.split(","))] = int(n)
    return out


def compiled_instantiations() -> set:
    """The instantiations of the tile kernel the library was built with, from the build's register report
    (``kernel_registers.json``), as tuples like the keys of ``launch_census``."""
    import json
    import re
    out = set()
    for name in json.load(open(REGISTERS_PATH)):
        m = re.search(r"mg_tile_kernel<([^>]*)>", name)
        if m:
            out.add(tuple({"true": 1, "false": 0}.get(x.strip(), None) if x.strip() in ("true", "false") else int(x)
                          for x in m.group(1).split(",")))
    return out


class MgError(RuntimeError):
    pass


def check(rc: int, handle=None):
    if rc != MG_OK:
        msg = lib().mg_last_error(handle)
        raise MgError(f"libmgrid_hip error {rc}: {msg.decode() if msg else '?'}")
