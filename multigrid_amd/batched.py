"""``BatchedMultiGrid``: E independent MultiGrid environments resident on one MI355X.

Host-side mirror of ``MultiGridEnv`` (multigrid/base.py:37) for a whole batch: ``reset`` /
``step`` / ``gen_obs`` launch the HIP kernels of ``libmgrid_hip.so`` on the current torch
stream and return PyTorch-ROCm tensors that the kernels wrote in place (zero-copy).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import MgConfig, MgOut, check


@dataclass
class StepOutput:
    image: torch.Tensor       # uint8 [E, A, V, V, 3]
    direction: torch.Tensor   # uint8 [E, A]
    reward: torch.Tensor      # float32 [E, A]
    terminated: torch.Tensor  # uint8 [E, A]
    truncated: torch.Tensor   # uint8 [E, A]
    onehot: Optional[torch.Tensor] = None  # uint8 [E, A, V, V, 24], multigrid/wrappers.py:114-146
    rows: Optional[torch.Tensor] = None      # uint8 [E, A, 8] agent rows of the state the observation shows
    reward64: Optional[torch.Tensor] = None  # float64 [E, A] the rewards as the reference's doubles
    info: Optional[torch.Tensor] = None      # uint8 [E, A] CompetativeRedBlueDoor: the step's info dict, packed per agent


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def gen_obs_of_state(spec: dict, grid: torch.Tensor, agents: torch.Tensor, image: Optional[torch.Tensor] = None,
                     direction: Optional[torch.Tensor] = None, onehot: Optional[torch.Tensor] = None):
    """``gen_obs`` of states the caller holds (``grid`` uint8 ``[E, W, H, 3]``, ``agents`` uint8 ``[E, A, 8]``, on the
    GPU) through ``mg_gen_obs_of_state``: no handle, and the library allocates nothing - the outputs are the
    caller's tensors too.  Limited to what fits the library's fixed 8 MiB staging buffer; calls on one GPU
    must not overlap."""
    dev = grid.device
    for t in (grid, agents, image, direction, onehot):
        if t is not None and (t.dtype != torch.uint8 or t.device != dev or not t.is_contiguous()):
            raise ValueError("tensors must be contiguous uint8 tensors on one GPU")
    e, a, v = int(grid.shape[0]), int(spec["num_agents"]), int(spec["view_size"])
    w, h = int(spec["width"]), int(spec["height"])
    if tuple(grid.shape) != (e, w, h, 3) or tuple(agents.shape) != (e, a, 8):
        raise ValueError(f"bad state shapes {tuple(grid.shape)} {tuple(agents.shape)}")
    for t, shape in ((image, (e, a, v, v, 3)), (direction, (e, a)), (onehot, (e, a, v, v, _lib.ONE_HOT_DIM))):
        if t is not None and tuple(t.shape) != shape:
            raise ValueError(f"expected an output of shape {shape}, got {tuple(t.shape)}")
    cfg = MgConfig(width=w, height=h, num_agents=a, view_size=v, see_through_walls=int(bool(spec["see_through_walls"])),
                   num_envs=e, device=dev.index or 0, max_steps=1)
    p = lambda t: None if t is None else t.data_ptr()
    mo = MgOut(p(image), p(direction), None, None, None, p(onehot))
    check(_lib.lib().mg_gen_obs_of_state(ctypes.byref(cfg), _ptr(grid), _ptr(agents), ctypes.byref(mo),
                                         ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))


class BatchedMultiGrid:
    """``spec`` is a dict with the keys of ``MgConfig`` up to ``num_rooms`` (see include/mgrid.h)."""

    def __init__(self, spec: dict, num_envs: int, device: int | str | torch.device = 0,
                 autoreset: int = _lib.AUTORESET_NONE, env_id_offset: int = 0, seed: int = 0,
                 one_hot: bool = False, compact_image: bool = True, batch_envs: int = 0,
                 out: Optional[StepOutput] = None):
        """``one_hot``: also return the image as ``OneHotObsWrapper`` encodes it (multigrid/wrappers.py:60-146),
        written by the same launch; ``compact_image=False`` then skips the ``(V, V, 3)`` image.  ``batch_envs``:
        the size of the whole batch when this handle holds one slice of it (``sharded.StreamShardedMultiGrid``);
        ``out``: tensors to adopt as the default outputs instead of allocating them."""
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedMultiGrid needs a ROCm GPU (no CPU fallback)")
        dev = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        self.device = dev
        self._dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        self.spec = dict(spec)
        self.E, self.A, self.V = int(num_envs), int(spec["num_agents"]), int(spec["view_size"])
        self.W, self.H = int(spec["width"]), int(spec["height"])
        cfg = MgConfig(
            kind=int(spec["kind"]), width=self.W, height=self.H, num_agents=self.A,
            view_size=self.V, max_steps=int(spec["max_steps"]),
            see_through_walls=int(bool(spec["see_through_walls"])),
            allow_agent_overlap=int(bool(spec["allow_agent_overlap"])),
            joint_reward=int(bool(spec["joint_reward"])), success_any=int(bool(spec["success_any"])),
            failure_any=int(bool(spec["failure_any"])), num_rooms=int(spec.get("num_rooms", 0)),
            num_envs=self.E, device=dev.index or 0, autoreset=int(autoreset),
            env_id_offset=int(env_id_offset), death_match=int(bool(spec.get("death_match", False))),
            per_agent_outputs=int(bool(spec.get("per_agent_outputs", True))), seed=int(seed),
            batch_envs=int(batch_envs))
        self._h = ctypes.c_void_p()
        check(_lib.lib().mg_create(ctypes.byref(cfg), ctypes.byref(self._h)))
        self._competitive = int(spec["kind"]) == _lib.KIND_COMPETITIVE
        if self._competitive:  # reward_schemes of multigrid/envs/__init__.py:100-108
            rs = np.ascontiguousarray(spec["reward_schemes"], np.float64)
            if rs.shape != (self.A, 4):
                raise ValueError(f"reward_schemes must be [{self.A}][4], got {rs.shape}")
            check(_lib.lib().mg_set_reward_schemes(self._h, rs.ctypes.data), self._h)
        e, a, v = self.E, self.A, self.V
        self.out = out if out is not None else StepOutput(
            image=torch.zeros((e, a, v, v, 3), dtype=torch.uint8, device=dev) if compact_image or not one_hot else None,
            direction=torch.zeros((e, a), dtype=torch.uint8, device=dev),
            reward=torch.zeros((e, a), dtype=torch.float32, device=dev),
            terminated=torch.zeros((e, a), dtype=torch.uint8, device=dev),
            truncated=torch.zeros((e, a), dtype=torch.uint8, device=dev),
            onehot=torch.zeros((e, a, v, v, _lib.ONE_HOT_DIM), dtype=torch.uint8, device=dev) if one_hot else None,
            info=torch.zeros((e, a), dtype=torch.uint8, device=dev) if self._competitive else None)
        self._mgout = self._make_out(self.out)
        self._ea = torch.Size((self.E, self.A))
        self._mg_step = _lib.lib().mg_step
        self.num_layouts = 0

    # -- plumbing -----------------------------------------------------------------
    @staticmethod
    def _make_out(o: StepOutput) -> MgOut:
        p = lambda t: None if t is None else t.data_ptr()
        return MgOut(p(o.image), p(o.direction), p(o.reward), p(o.terminated), p(o.truncated), p(o.onehot),
                     p(o.rows), p(o.reward64), p(o.info))

    def _stream(self):
        """The current torch stream of this device as a ``hipStream_t``."""
        if _RAW_STREAM is not None:  # the raw handle without building a Stream object (1 us less per call)
            return ctypes.c_void_p(_RAW_STREAM(self._dev_index))
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev_u8(self, x, shape):
        t = torch.as_tensor(x)
        if t.dtype != torch.uint8 or t.device != self.device:
            t = t.to(device=self.device, dtype=torch.uint8)
        t = t.contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            torch.cuda.synchronize(self.device)
            _lib.lib().mg_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- API ----------------------------------------------------------------------
    def upload_layouts(self, grids: np.ndarray, agents: np.ndarray, meta: Optional[np.ndarray] = None):
        """Start states from the host-side layout generators (``multigrid_amd.layouts``)."""
        grids = np.ascontiguousarray(grids, np.uint8)
        agents = np.ascontiguousarray(agents, np.uint8)
        n = grids.shape[0]
        if grids.shape != (n, self.W, self.H, 3) or agents.shape != (n, self.A, 8):
            raise ValueError(f"bad layout shapes {grids.shape} {agents.shape}")
        if meta is None and int(self.spec["kind"]) == _lib.KIND_REDBLUEDOORS:
            raise ValueError("RedBlueDoors layouts need `meta`: the red and blue door positions the hook looks at")
        meta = np.zeros((n, 4), np.uint8) if meta is None else np.ascontiguousarray(meta, np.uint8)
        check(_lib.lib().mg_upload_layouts(self._h, grids.ctypes.data, agents.ctypes.data,
                                           meta.ctypes.data, n), self._h)
        self.num_layouts = n

    def generate_layouts(self, gen, n: int, seed: int = 0):
        """Draw ``n`` start states on the device with the environment class's own ``_gen_grid`` procedure
        (``mg_generate_layouts``) instead of uploading host-generated ones.  ``gen`` is the layout generator
        returned by ``envs.make_spec`` (a ``functools.partial`` whose keywords carry the family's parameters)."""
        kw = dict(getattr(gen, "keywords", {}) or {})
        ls = _lib.MgLayoutSpec()
        ls.size = int(kw.get("size", kw.get("room_size", 0)))
        start = kw.get("agent_start_pos", (1, 1))
        sdir = kw.get("agent_start_dir", 0)
        ls.random_start = int(start is None or sdir is None)
        if not ls.random_start:
            ls.start_x, ls.start_y, ls.start_dir = int(start[0]), int(start[1]), int(sdir)
        ls.max_hallway_keys = int(kw.get("max_hallway_keys", 1))
        ls.max_keys_per_room = int(kw.get("max_keys_per_room", 2))
        for k, c in enumerate(kw["agent_colors"]):
            ls.agent_color[k] = int(c)
        check(_lib.lib().mg_generate_layouts(self._h, ctypes.byref(ls), int(n), int(seed) & (2 ** 64 - 1)), self._h)
        self.num_layouts = int(n)

    def reset(self, layout_idx=None, mask=None, out: Optional[StepOutput] = None) -> StepOutput:
        """``MultiGridEnv.reset`` (multigrid/base.py:310) for all (or the masked) environments."""
        li = None
        if layout_idx is not None:
            li = torch.as_tensor(layout_idx).to(device=self.device, dtype=torch.int32).contiguous()
            assert li.shape == (self.E,)
        mk = None if mask is None else self._dev_u8(mask, (self.E,))
        o = self.out if out is None else out
        mo = self._mgout if out is None else self._make_out(out)
        check(_lib.lib().mg_reset(self._h, _ptr(li), _ptr(mk), ctypes.byref(mo), self._stream()), self._h)
        return o

    def step(self, actions, order=None, out: Optional[StepOutput] = None) -> StepOutput:
        """``MultiGridEnv.step`` (multigrid/base.py:364) for every environment."""
        if (type(actions) is torch.Tensor and actions.dtype is torch.uint8 and actions.device == self.device and
                actions.shape == self._ea and actions.is_contiguous()):
            act = actions  # the usual case, checked without the conversions below
        else:
            act = self._dev_u8(actions, self._ea)
        od = None if order is None else self._dev_u8(order, self._ea)
        o = self.out if out is None else out
        mo = self._mgout if out is None else self._make_out(out)
        rc = self._mg_step(self._h, ctypes.c_void_p(act.data_ptr()), _ptr(od), ctypes.byref(mo), self._stream())
        if rc != 0:
            check(rc, self._h)
        return o

    def step_n(self, actions, order=None, out: Optional[StepOutput] = None, one_hot: bool = False) -> StepOutput:
        """T consecutive ``step`` calls queued by one library call; ``actions`` is ``[T, E, A]`` and every
        tensor of the returned ``StepOutput`` has a leading ``T`` dimension."""
        t_steps = int(torch.as_tensor(actions).shape[0])
        act = self._dev_u8(actions, (t_steps, self.E, self.A))
        od = None if order is None else self._dev_u8(order, (t_steps, self.E, self.A))
        if out is None:
            e, a, v, dev = self.E, self.A, self.V, self.device
            out = StepOutput(
                image=torch.empty((t_steps, e, a, v, v, 3), dtype=torch.uint8, device=dev),
                direction=torch.empty((t_steps, e, a), dtype=torch.uint8, device=dev),
                reward=torch.empty((t_steps, e, a), dtype=torch.float32, device=dev),
                terminated=torch.empty((t_steps, e, a), dtype=torch.uint8, device=dev),
                truncated=torch.empty((t_steps, e, a), dtype=torch.uint8, device=dev),
                onehot=torch.empty((t_steps, e, a, v, v, _lib.ONE_HOT_DIM), dtype=torch.uint8, device=dev)
                if one_hot else None,
                info=torch.empty((t_steps, e, a), dtype=torch.uint8, device=dev) if self._competitive else None)
        elif any(t is not None and t.shape[0] != t_steps for t in
                 (out.image, out.direction, out.reward, out.terminated, out.truncated, out.onehot, out.info)):
            raise ValueError("out must hold one slice per step")
        mo = self._make_out(out)
        check(_lib.lib().mg_step_n(self._h, t_steps, _ptr(act), _ptr(od), ctypes.byref(mo), self._stream()), self._h)
        return out

    def gen_obs(self, out: Optional[StepOutput] = None) -> StepOutput:
        """``MultiGridEnv.gen_obs`` (multigrid/base.py:410)."""
        o = self.out if out is None else out
        mo = self._mgout if out is None else self._make_out(out)
        check(_lib.lib().mg_gen_obs(self._h, ctypes.byref(mo), self._stream()), self._h)
        return o

    def get_state(self):
        """Grid, agent rows, door metadata, step counts and flags in the reference's layouts."""
        e, dev = self.E, self.device
        grid = torch.empty((e, self.W, self.H, 3), dtype=torch.uint8, device=dev)
        agents = torch.empty((e, self.A, 8), dtype=torch.uint8, device=dev)
        meta = torch.empty((e, 4), dtype=torch.uint8, device=dev)
        steps = torch.empty((e,), dtype=torch.int32, device=dev)
        flags = torch.empty((e,), dtype=torch.uint8, device=dev)
        check(_lib.lib().mg_get_state(self._h, _ptr(grid), _ptr(agents), _ptr(meta), _ptr(steps),
                                      _ptr(flags), self._stream()), self._h)
        return dict(grid=grid, agents=agents, meta=meta, steps=steps, flags=flags)

    def agent_rows(self):
        """Agent rows uint8 ``[E, A, 8]`` (x y dir terminated carry_type carry_color carry_state color) and step
        counts int32 ``[E]`` of the current state, without converting the grids."""
        agents = torch.empty((self.E, self.A, 8), dtype=torch.uint8, device=self.device)
        steps = torch.empty((self.E,), dtype=torch.int32, device=self.device)
        check(_lib.lib().mg_get_state(self._h, None, _ptr(agents), None, _ptr(steps), None, self._stream()), self._h)
        return agents, steps

    def agents_mask(self, mask: Optional[torch.Tensor] = None, done: Optional[torch.Tensor] = None):
        """``PettingZooWrapper.agents`` as a bool ``[E, A]`` tensor and ``MultiGridEnv.is_done`` as a bool ``[E]``
        tensor for every environment, by one launch (``mg_agents_mask``); ``mask`` / ``done``: uint8 tensors to write
        into instead of fresh ones."""
        if mask is None:
            mask = torch.empty((self.E, self.A), dtype=torch.uint8, device=self.device)
        if done is None:
            done = torch.empty((self.E,), dtype=torch.uint8, device=self.device)
        check(_lib.lib().mg_agents_mask(self._h, _ptr(mask), _ptr(done), self._stream()), self._h)
        return mask.view(torch.bool), done.view(torch.bool)

    def set_state(self, grid=None, agents=None, meta=None, steps=None, flags=None):
        e = self.E
        g = None if grid is None else self._dev_u8(grid, (e, self.W, self.H, 3))
        a = None if agents is None else self._dev_u8(agents, (e, self.A, 8))
        m = None if meta is None else self._dev_u8(meta, (e, 4))
        f = None if flags is None else self._dev_u8(flags, (e,))
        s = None
        if steps is not None:
            s = torch.as_tensor(steps).to(device=self.device, dtype=torch.int32).contiguous()
        check(_lib.lib().mg_set_state(self._h, _ptr(g), _ptr(a), _ptr(m), _ptr(s), _ptr(f),
                                      self._stream()), self._h)
        torch.cuda.current_stream(self.device).synchronize()  # inputs may be temporaries

    def stats(self) -> torch.Tensor:
        """int64[8] device tensor of episode statistics (see ``mg_stats`` in include/mgrid.h)."""
        t = torch.zeros((_lib.NUM_STATS,), dtype=torch.int64, device=self.device)
        check(_lib.lib().mg_stats(self._h, _ptr(t), self._stream()), self._h)
        return t
