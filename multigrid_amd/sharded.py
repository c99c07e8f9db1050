"""``StreamShardedMultiGrid``: one batch of environments held as S slices that step on HIP streams of their own.

Environments are independent (multigrid/base.py:364-408 touches one environment's state only), but launches on one
stream are not: step t + 1 of every environment waits until the slowest wavefront of step t has drained.  Held as
S handles on S streams (``mg_step_shards``), a slice waits only for itself, so the start and the drain of one
slice's launch overlap the other slices' work.  The reference runs groups of environments side by side in the same
spirit, as rollout workers (multigrid/utils/training_utilis.py:273) and as a vector of environments
(multigrid/scripts/train_ppo_cleanrl.py:366).

Results are those of one ``BatchedMultiGrid`` of the same size (layout choice and agent order depend on the global
environment index): whole-batch tensors in, whole-batch tensors out, every slice reading and writing its part.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import check
from .batched import BatchedMultiGrid, StepOutput, _ptr

ALIGN_ENVS = 64  # slices start at a multiple of this: their parts of the output arrays stay 16-byte aligned


def slice_bounds(num_envs: int, shards: int):
    """``[(first, end), ...]``: at most ``shards`` consecutive slices of ``num_envs`` environments, equal in size up
    to rounding to ``ALIGN_ENVS`` (the last one takes what is left; fewer slices when the batch is small)."""
    if num_envs < 1 or shards < 1:
        raise ValueError("num_envs and shards must be positive")
    per = -(-num_envs // shards)
    per = -(-per // ALIGN_ENVS) * ALIGN_ENVS
    return [(lo, min(lo + per, num_envs)) for lo in range(0, num_envs, per)]


class StreamShardedMultiGrid:
    def __init__(self, spec: dict, num_envs: int, shards: int = 2, device: int | str | torch.device = 0,
                 autoreset: int = _lib.AUTORESET_NONE, env_id_offset: int = 0, seed: int = 0,
                 one_hot: bool = False, compact_image: bool = True):
        if not torch.cuda.is_available():
            raise RuntimeError("StreamShardedMultiGrid needs a ROCm GPU (no CPU fallback)")
        dev = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        self.device, self.spec = dev, dict(spec)
        self.E, self.A, self.V = int(num_envs), int(spec["num_agents"]), int(spec["view_size"])
        self.W, self.H = int(spec["width"]), int(spec["height"])
        self.bounds = slice_bounds(self.E, shards)
        e, a, v = self.E, self.A, self.V
        self.out = StepOutput(
            image=torch.zeros((e, a, v, v, 3), dtype=torch.uint8, device=dev) if compact_image or not one_hot else None,
            direction=torch.zeros((e, a), dtype=torch.uint8, device=dev),
            reward=torch.zeros((e, a), dtype=torch.float32, device=dev),
            terminated=torch.zeros((e, a), dtype=torch.uint8, device=dev),
            truncated=torch.zeros((e, a), dtype=torch.uint8, device=dev),
            onehot=torch.zeros((e, a, v, v, _lib.ONE_HOT_DIM), dtype=torch.uint8, device=dev) if one_hot else None)
        self._mgout = BatchedMultiGrid._make_out(self.out)
        self.shards: List[BatchedMultiGrid] = []
        self.streams: List[torch.cuda.Stream] = []
        for lo, hi in self.bounds:
            self.shards.append(BatchedMultiGrid(
                spec, hi - lo, device=dev, autoreset=autoreset, env_id_offset=env_id_offset + lo, seed=seed,
                one_hot=one_hot, compact_image=compact_image, batch_envs=self.E, out=self._slice(self.out, lo, hi)))
            self.streams.append(torch.cuda.Stream(device=dev))
        n = len(self.shards)
        self._handles = (ctypes.c_void_p * n)(*[s._h.value for s in self.shards])
        self._stream_ptrs = (ctypes.c_void_p * n)(*[s.cuda_stream for s in self.streams])
        self.num_layouts = 0

    @staticmethod
    def _slice(o: StepOutput, lo: int, hi: int) -> StepOutput:
        cut = lambda t: None if t is None else t[lo:hi]
        return StepOutput(cut(o.image), cut(o.direction), cut(o.reward), cut(o.terminated), cut(o.truncated),
                          cut(o.onehot))

    # -- ordering between the caller's stream and the slices' streams -------------------------
    def fork(self):
        """The slices' streams wait for everything queued so far on the current stream (e.g. the kernel that
        wrote the actions)."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        for s in self.streams:
            s.wait_event(ev)

    def join(self, shard: Optional[int] = None):
        """The current stream waits for everything queued so far on the slices' streams (or on one of them)."""
        cur = torch.cuda.current_stream(self.device)
        for k in (range(len(self.streams)) if shard is None else (shard,)):
            cur.wait_stream(self.streams[k])

    def _each(self, fn):
        """Runs ``fn(k, shard, lo, hi)`` for every slice on its own stream, ordered after and before the current one."""
        self.fork()
        res = []
        for k, (sh, (lo, hi)) in enumerate(zip(self.shards, self.bounds)):
            with torch.cuda.stream(self.streams[k]):
                res.append(fn(k, sh, lo, hi))
        self.join()
        # tensors the slices allocated on their streams are read on the current stream from here on
        cur = torch.cuda.current_stream(self.device)
        for r in res:
            for t in (r.values() if isinstance(r, dict) else r if isinstance(r, (tuple, list)) else (r,)):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(cur)
        return res

    # -- API (as BatchedMultiGrid) -------------------------------------------------------------
    def upload_layouts(self, grids: np.ndarray, agents: np.ndarray, meta: Optional[np.ndarray] = None):
        for sh in self.shards:
            sh.upload_layouts(grids, agents, meta)
        self.num_layouts = self.shards[0].num_layouts

    def generate_layouts(self, gen, n: int, seed: int = 0):
        for sh in self.shards:  # layout l depends on (seed, l) only: every slice draws the same pool
            sh.generate_layouts(gen, n, seed)
        self.num_layouts = int(n)

    def reset(self, layout_idx=None, mask=None) -> StepOutput:
        """``MultiGridEnv.reset`` (multigrid/base.py:310) for all (or the masked) environments; the current stream
        has waited for it on return."""
        li = None if layout_idx is None else torch.as_tensor(layout_idx).to(device=self.device, dtype=torch.int32)
        mk = None if mask is None else torch.as_tensor(mask).to(device=self.device, dtype=torch.uint8)
        self._each(lambda k, sh, lo, hi: sh.reset(None if li is None else li[lo:hi], None if mk is None else mk[lo:hi]))
        return self.out

    def step(self, actions, order=None, fork: bool = True, join: bool = True) -> StepOutput:
        """``MultiGridEnv.step`` (multigrid/base.py:364) for every environment: one launch per slice, each on the
        slice's stream.  ``fork``: the launches wait for what the current stream holds so far (the producer of
        ``actions``); ``join``: the current stream then waits for the launches, so that anything queued on it
        afterwards sees the outputs.  With both (the default) this is ``BatchedMultiGrid.step`` in S pieces, correct
        for any caller and no faster.  The slices run ahead of each other only with ``join=False``: the caller then
        consumes slice k on ``streams[k]`` (or after ``join(k)``), and may pass ``fork=False`` when the actions
        are known to be complete.  ``fork=False`` is honoured only for inputs that are used as they are (contiguous
        uint8 tensors on this device): anything converted or copied here is produced on the current stream, which
        the slices then have to wait for, so such a call forks regardless."""
        sh0 = self.shards[0]
        ready = lambda x: (isinstance(x, torch.Tensor) and x.dtype == torch.uint8 and x.device == self.device and
                           x.is_contiguous() and x.shape == (self.E, self.A))
        act = actions if ready(actions) else sh0._dev_u8(actions, (self.E, self.A))
        od = None if order is None else (order if ready(order) else sh0._dev_u8(order, (self.E, self.A)))
        if act is not actions or (od is not None and od is not order):
            fork = True  # a temporary made on the current stream
        if fork:
            self.fork()
            for s in self.streams:  # the allocator must not hand these out again before the slices have read them
                act.record_stream(s)
                if od is not None:
                    od.record_stream(s)
        check(_lib.lib().mg_step_shards(self._handles, len(self.shards), _ptr(act), _ptr(od),
                                        ctypes.byref(self._mgout), self._stream_ptrs), sh0._h)
        if join:
            self.join()
        return self.out

    def step_n(self, actions, order=None, outs: Optional[List[StepOutput]] = None, fork: bool = True,
               join: bool = True) -> List[StepOutput]:
        """T consecutive steps, every slice taking them in ONE fused launch on its stream (``mg_step_n``).
        ``actions`` (and ``order``) are either one ``[T, E, A]`` tensor of the whole batch - every slice then gets
        a contiguous copy of its part - or a list with one contiguous ``[T, E_k, A]`` tensor per slice.  Returns one
        ``StepOutput`` per slice with ``[T, E_k, ...]`` tensors (``bounds[k]`` names the slice's environments):
        the slices' results are not interleaved in memory, their consumers work per slice anyway."""
        made = []  # anything converted or copied here is produced on the current stream

        def parts(x):
            if x is None:
                return [None] * len(self.shards)
            if isinstance(x, (list, tuple)):
                out = []
                for (lo, hi), t in zip(self.bounds, x):
                    ok = (isinstance(t, torch.Tensor) and t.dtype == torch.uint8 and t.device == self.device and
                          t.is_contiguous())
                    out.append(t if ok else torch.as_tensor(t).to(device=self.device, dtype=torch.uint8).contiguous())
                    if not ok:
                        made.append(out[-1])
                return out
            x = torch.as_tensor(x).to(device=self.device, dtype=torch.uint8)
            cut = [x[:, lo:hi].contiguous() for lo, hi in self.bounds]
            made.extend(cut)
            return cut
        acts, ords = parts(actions), parts(order)
        if made:
            fork = True  # the slices' streams must wait for those copies, whatever the caller knows about its inputs
        if fork:
            self.fork()
        res = []
        for k, sh in enumerate(self.shards):
            for t in (acts[k], ords[k]):
                if fork and isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(self.streams[k])
            with torch.cuda.stream(self.streams[k]):
                res.append(sh.step_n(acts[k], ords[k], out=None if outs is None else outs[k]))
        if join:
            self.join()
            if outs is None:  # allocated on the slices' streams, read on the current one from here on
                cur = torch.cuda.current_stream(self.device)
                for o in res:
                    for t in (o.image, o.direction, o.reward, o.terminated, o.truncated, o.onehot):
                        if t is not None:
                            t.record_stream(cur)
        return res

    def gen_obs(self) -> StepOutput:
        """``MultiGridEnv.gen_obs`` (multigrid/base.py:410)."""
        self._each(lambda k, sh, lo, hi: sh.gen_obs())
        return self.out

    def get_state(self):
        parts = self._each(lambda k, sh, lo, hi: sh.get_state())
        return {key: torch.cat([p[key] for p in parts]) for key in parts[0]}

    def agent_rows(self):
        """Agent rows uint8 ``[E, A, 8]`` and step counts int32 ``[E]`` of the current state (as
        ``BatchedMultiGrid.agent_rows``), so that the batch views of ``views.py`` work over a sharded batch too."""
        parts = self._each(lambda k, sh, lo, hi: sh.agent_rows())
        return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])

    def set_state(self, grid=None, agents=None, meta=None, steps=None, flags=None):
        dev = self.device
        t = lambda x: None if x is None else torch.as_tensor(x).to(device=dev)
        g, a, m, s, f = t(grid), t(agents), t(meta), t(steps), t(flags)
        cut = lambda x, lo, hi: None if x is None else x[lo:hi]
        self._each(lambda k, sh, lo, hi: sh.set_state(cut(g, lo, hi), cut(a, lo, hi), cut(m, lo, hi), cut(s, lo, hi),
                                                      cut(f, lo, hi)))

    def stats(self) -> torch.Tensor:
        """int64[8] episode statistics of the whole batch: sums over the slices; entry 5 (steps taken) is the
        same for every slice and reported once."""
        parts = torch.stack(self._each(lambda k, sh, lo, hi: sh.stats()))
        total = parts.sum(dim=0)
        total[5] = parts[:, 5].max()
        return total

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        for sh in self.shards:
            sh.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
