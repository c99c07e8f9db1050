"""Framework views over a whole batch: the RLlib ``MultiAgentEnv`` and the PettingZoo ``ParallelEnv``
conventions served from the ``[E, A]`` tensors of one ``BatchedMultiGrid``, without leaving the GPU.

The reference wraps ONE environment (multigrid/rllib/__init__.py:47-76, multigrid/pettingzoo/__init__.py:35-70);
a sampler that holds E of them calls E ``step`` methods and reads E dicts.  Here one ``step`` advances all E
environments with one kernel launch and the per-environment parts of those conventions are tensors:

* ``BatchedRLlibView.step`` returns ``terminations`` / ``truncations`` dicts whose ``"agents"`` entry is the
  ``[E, A]`` tensor of per-agent flags and whose ``"__all__"`` entry is the ``[E]`` tensor
  ``all(per-agent flags)`` - the ``"__all__"`` key ``RLlibWrapper.step`` adds (rllib/__init__.py:71-76);
* ``BatchedParallelView.agents_mask`` is the ``[E, A]`` boolean tensor of ``PettingZooWrapper.agents``
  (pettingzoo/__init__.py:48-52): agent k of environment e is listed while it is not terminated
  (``agent.terminated``, the environment's own flag, not the returned dict) and the environment is not done
  (``is_done()``, multigrid/base.py:587-596: truncated or every agent terminated).

No call here waits for the GPU or copies to the host: everything is enqueued on the current stream.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .batched import BatchedMultiGrid


def competitive_v2_observation(image: torch.Tensor, direction: torch.Tensor) -> torch.Tensor:
    """The ``(V, V, 25)`` observation of ``CompetativeRedBlueDoorWrapperV2`` (multigrid/wrappers.py:372-416) for a whole
    batch, on the GPU: ``image`` uint8 ``[..., V, V, 3]``, ``direction`` uint8 ``[...]`` -> uint8 ``[..., V, V, 25]``.
    Plane 0 holds the agent's direction.  The other 24 are what the wrapper's second pass of ``one_hot`` over the
    already one-hot image leaves (:409): that pass reads the first three one-hot channels - "type is unseen", "type is
    empty", "type is wall" - as indices into the type, colour and state groups, so planes 1 + (type == 0),
    12 + (type == 1) and 18 + (type == 2) are set and nothing else."""
    t = image[..., 0]
    out = torch.zeros(image.shape[:-1] + (25,), dtype=torch.uint8, device=image.device)
    out[..., 0] = direction[..., None, None]
    idx = torch.stack((1 + (t == 0).long(), 12 + (t == 1).long(), 18 + (t == 2).long()), dim=-1)
    out.scatter_(-1, idx, 1)
    return out


class BatchedRLlibView:
    """``MultiAgentEnv`` convention over ``sim`` (which keeps its own auto-reset mode)."""

    def __init__(self, sim: BatchedMultiGrid, one_hot: bool = False):
        self.sim = sim
        self.one_hot = one_hot
        self.num_envs, self.num_agents = sim.E, sim.A

    def get_agent_ids(self):
        """Integer agent ids, as the in-scope environments use them (rllib/__init__.py:68-69)."""
        return set(range(self.num_agents))

    def _obs(self, out) -> Dict[str, torch.Tensor]:
        return {"image": out.onehot if self.one_hot else out.image, "direction": out.direction}

    def reset(self, layout_idx=None, mask=None):
        return self._obs(self.sim.reset(layout_idx=layout_idx, mask=mask)), {}

    def step(self, actions: torch.Tensor, order: Optional[torch.Tensor] = None):
        """``actions`` uint8 ``[E, A]`` on the GPU.  Returns ``(obs, rewards, terminations, truncations, infos)``;
        ``terminations["__all__"][e] == all(terminations["agents"][e])`` and likewise for truncations."""
        out = self.sim.step(actions, order)
        # the flags are bytes of 0 / 1: reinterpreted as booleans, not converted
        term, trunc = out.terminated.view(torch.bool), out.truncated.view(torch.bool)
        terminations = {"agents": term, "__all__": term.all(dim=1)}
        # one truncation flag per environment, repeated for every agent (multigrid/base.py:401-402): all() of a
        # row is its first entry
        truncations = {"agents": trunc, "__all__": trunc[:, 0]}
        return self._obs(out), out.reward, terminations, truncations, {}


class BatchedParallelView:
    """``ParallelEnv`` convention over ``sim``.  Meant for ``MG_AUTORESET_NONE`` / ``NEXT_STEP`` handles, where
    the state after ``step`` is still that of the episode the step belonged to (as in the reference, where the
    caller resets); with same-step auto-reset the mask describes the fresh episode instead."""

    def __init__(self, sim: BatchedMultiGrid):
        self.sim = sim
        self.num_envs, self.num_agents = sim.E, sim.A
        self.max_steps = int(sim.spec["max_steps"])
        self.reset = sim.reset
        self.step = sim.step
        self.close = sim.close

    @property
    def possible_agents(self):
        return list(range(self.num_agents))

    def is_done(self) -> torch.Tensor:
        """``MultiGridEnv.is_done`` per environment (multigrid/base.py:587-596), bool ``[E]``."""
        if hasattr(self.sim, "agents_mask"):  # one launch
            return self.sim.agents_mask()[1]
        rows, steps = self.sim.agent_rows()
        return (steps >= self.max_steps) | rows[:, :, 3].bool().all(dim=1)

    @property
    def agents_mask(self) -> torch.Tensor:
        """bool ``[E, A]``: agent k is in ``PettingZooWrapper(env_e).agents`` (pettingzoo/__init__.py:48-52)."""
        if hasattr(self.sim, "agents_mask"):  # one launch
            return self.sim.agents_mask()[0]
        rows, steps = self.sim.agent_rows()
        alive = ~rows[:, :, 3].bool()
        done = (steps >= self.max_steps) | ~alive.any(dim=1)
        return alive & ~done[:, None]
