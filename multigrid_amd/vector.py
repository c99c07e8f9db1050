"""Vector-environment view of one ``BatchedMultiGrid``: N environments behind one ``reset`` / ``step``.

This is the contract of ``gymnasium.vector.SyncVectorEnv`` as the reference uses it
(multigrid/scripts/train_ppo_cleanrl.py:366 builds it, :517-529 steps it and reads
``info["final_info"][i]["episode"]["r" / "l"]`` left there by ``RecordEpisodeStatistics``, :173):

* ``step`` never needs a ``reset`` call in between: an environment whose episode ended
  (``all(terminations)`` or truncation, multigrid/base.py:587-596) restarts inside the same launch,
  the observation returned for it is the first one of its next episode, and reward / terminated /
  truncated are those of the step that ended the episode;
* ``infos["_final_info"]`` marks those environments and ``infos["final_info"][i]`` holds
  ``{"episode": {"r": return, "l": length}}`` for them (``None`` elsewhere).  The same numbers
  are available without leaving the device as ``infos["episode_return"]`` / ``["episode_length"]``.
  With ``numpy=False`` (the default) ``step`` never waits for the GPU: it returns the mask and those two
  tensors every step (zeros where no episode ended) and leaves out the per-environment Python objects.
  Those three tensors are written in place, two sets used in turn: what one ``step`` returned is overwritten by
  the second ``step`` after it (observations, rewards and flags already by the next one), so copy what must last.
* ``final_observation=True`` adds the other half of that contract, ``infos["final_observation"]``: the observation
  of the state the finished episode ended in, which the returned observation (the next episode's first) has
  replaced.  The step is then taken without the restart inside the launch; the terminal observations are copied
  aside and one masked ``mg_reset`` launch restarts the finished environments (same start states, same order
  stream, same results as the one-launch form; two more launches and one copy of the observations per step).
  With ``numpy=True`` it is an object array of per-environment observations (``None`` elsewhere) next to
  ``infos["_final_observation"]``; otherwise the whole batch's tensors, meaningful where ``infos["_final_info"]`` is set.

Every array is a PyTorch-ROCm tensor written in place by the HIP kernels (``numpy=True`` converts
at the boundary like ``SyncVectorEnv`` does).  With ``single_agent=True`` (one agent per
environment, the ``SingleAgentWrapper`` case of multigrid/wrappers.py:149-176) the agent
dimension is dropped, which is the exact ``SyncVectorEnv`` shape; otherwise arrays are ``[N, A, ...]``.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib, envs, layouts
from .batched import BatchedMultiGrid


class MultiGridVectorEnv:
    def __init__(self, name: str, num_envs: int, agents: Optional[int] = None, device: int | str = 0,
                 num_layouts: int = 1024, seed: int = 0, one_hot: bool = False, single_agent: bool = False,
                 numpy: bool = False, env_id_offset: int = 0, device_layouts: bool = False,
                 final_observation: bool = False, **kwargs):
        """``device_layouts``: draw the pool of start states on the GPU (``mg_generate_layouts``) instead of
        generating it on the host; large pools then cost milliseconds."""
        self.spec, self._gen = envs.make_spec(name, num_agents=agents, **kwargs)
        self.num_envs = int(num_envs)
        self.num_agents = self.spec["num_agents"]
        if single_agent and self.num_agents != 1:
            raise ValueError("single_agent=True needs agents=1")
        self.single_agent, self.one_hot, self.numpy = single_agent, one_hot, numpy
        self.num_actions = 7  # multigrid/core/actions.py:4-15
        self.final_observation = bool(final_observation)
        self._sim = BatchedMultiGrid(self.spec, self.num_envs, device=device, env_id_offset=env_id_offset, seed=seed,
                                     one_hot=one_hot, autoreset=_lib.AUTORESET_NONE
                                     if self.final_observation else _lib.AUTORESET_SAME_STEP)
        self._num_layouts, self._seed, self._device_layouts = int(num_layouts), int(seed), bool(device_layouts)
        self._uploaded = None
        dev = self._sim.device
        self._ret = torch.zeros((self.num_envs, self.num_agents), dtype=torch.float64, device=dev)
        self._len = torch.zeros((self.num_envs,), dtype=torch.int64, device=dev)
        # outputs of mg_episode_stats, two sets used in turn
        self._ep_ret = torch.zeros((2, self.num_envs, self.num_agents), dtype=torch.float64, device=dev)
        self._ep_len = torch.zeros((2, self.num_envs), dtype=torch.int64, device=dev)
        self._done = torch.zeros((2, self.num_envs), dtype=torch.uint8, device=dev)
        self._flip = 0
        if self.final_observation:
            o = self._sim.out
            self._final_image = torch.zeros_like(o.onehot if one_hot else o.image)
            self._final_dir = torch.zeros_like(o.direction)
        self._mg_episode_stats = _lib.lib().mg_episode_stats
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        o = self._sim.out  # every tensor involved lives as long as this object: the argument lists are made once
        self._stats_sets = [
            (self._ep_ret[k], self._ep_len[k], self._done[k].view(torch.bool),
             (self.num_envs, self.num_agents, p(o.reward), p(o.terminated), p(o.truncated), p(self._ret), p(self._len),
              p(self._done[k]), p(self._ep_ret[k]), p(self._ep_len[k]))) for k in range(2)]

    # -- helpers ----------------------------------------------------------------------------------
    def _view(self, t: torch.Tensor):
        if self.single_agent:
            t = t[:, 0]
        return t.cpu().numpy() if self.numpy else t

    def _obs(self, out):
        return {"image": self._view(out.onehot if self.one_hot else out.image), "direction": self._view(out.direction)}

    def _layouts(self, seed: int):
        if self._uploaded != seed:
            if self._device_layouts:
                self._sim.generate_layouts(self._gen, self._num_layouts, seed=seed)
            else:
                grids, ags, metas, _ = layouts.make_pool(self._gen, self._num_layouts, seed0=seed)
                self._sim.upload_layouts(grids, ags, metas)
            self._uploaded = seed

    # -- SyncVectorEnv surface -----------------------------------------------------------------------
    def reset(self, seed: Optional[int] = None, options=None):
        """Start every environment on a fresh episode.  ``seed`` selects the pool of start states (layout k of
        the pool is drawn with generator seed ``seed + k``); without it the pool of the last reset is kept."""
        if seed is not None:
            self._layouts(int(seed))
        elif self._uploaded is None:
            self._layouts(self._seed)
        out = self._sim.reset()
        self._ret.zero_()
        self._len.zero_()
        return self._obs(out), {}

    def step(self, actions):
        a = torch.as_tensor(actions)
        if self.single_agent and a.dim() == 1:
            a = a[:, None]
        out = self._sim.step(a.to(torch.uint8))
        # RecordEpisodeStatistics, one set of counters per environment, by ONE launch (mg_episode_stats): running
        # return and length, the mask of the environments whose episode ended, their statistics (zero elsewhere)
        ep_ret, ep_len, done, args = self._stats_sets[self._flip]
        self._flip ^= 1  # what the last step returned stays valid for one more step
        rc = self._mg_episode_stats(*args, self._sim._stream())
        if rc != 0:
            _lib.check(rc)
        infos = {}
        if self.final_observation:
            # the terminal observations aside, then the finished environments restarted by one masked reset launch
            # that writes their first observations over them
            self._final_image.copy_(out.onehot if self.one_hot else out.image)
            self._final_dir.copy_(out.direction)
            self._sim.reset(mask=self._done[self._flip ^ 1])
        if not self.numpy:
            # everything stays on the GPU and nothing here waits for it
            infos["_final_info"] = done
            infos["episode_return"] = self._view(ep_ret)
            infos["episode_length"] = ep_len
            if self.final_observation:
                infos["final_observation"] = {"image": self._view(self._final_image), "direction": self._view(self._final_dir)}
        elif bool(done.any()):
            infos["_final_info"] = self._view_mask(done)
            infos["episode_return"] = self._view(ep_ret)
            infos["episode_length"] = ep_len.cpu().numpy() if self.numpy else ep_len
            final = np.full((self.num_envs,), None, dtype=object)
            idx = done.nonzero()[:, 0].cpu().numpy()
            r, l = ep_ret[idx].cpu().numpy(), ep_len[idx].cpu().numpy()
            for k, i in enumerate(idx):
                final[i] = {"episode": {"r": float(r[k, 0]) if self.single_agent else r[k], "l": int(l[k])}}
            infos["final_info"] = final
            if self.final_observation:
                img, direction = self._view(self._final_image), self._view(self._final_dir)
                last = np.full((self.num_envs,), None, dtype=object)
                for i in idx:
                    last[i] = {"image": img[i], "direction": direction[i]}
                infos["final_observation"] = last
                infos["_final_observation"] = infos["_final_info"]
        # (the flags are bytes of 0 / 1: reinterpreted as booleans, not converted)
        return (self._obs(out), self._view(out.reward), self._view(out.terminated.view(torch.bool)),
                self._view(out.truncated.view(torch.bool)), infos)

    def _view_mask(self, done):
        return done.cpu().numpy() if self.numpy else done

    def stats(self) -> torch.Tensor:
        return self._sim.stats()

    def close(self):
        self._sim.close()
