"""The rule by which the HIP act phase decides which agents of an environment must act in order.

The kernel lets every agent whose action cannot interact with another agent's act at once, from the state at
the start of the step, and replays only the others in the reference's order (multigrid/base.py:457-464).
This file restates that rule on the CPU and checks its consequence against the oracle: stepping with the
unmarked agents first (in index order) and the marked ones afterwards in their drawn order must give
exactly the state, rewards and terminations of the fully sequential step.
"""
import copy

import numpy as np
import pytest

from multigrid_amd import envs
from oracle import pyref
from oracle.pyref import (BALL, BOX, CC, CT, DIR, DIR_VEC, DOOR, EMPTY, FLOOR, GOAL, KEY, LAVA, LOCKED, OPEN, TERM, X, Y)


def action_facts(env, i, act):
    """What eval_action() publishes about one agent: where it stands, the cell it faces, what the action
    would do against the start-of-step state and what it could do against any state."""
    a = env.agents
    alive = a[i, TERM] == 0
    dx, dy = DIR_VEC[int(a[i, DIR])]
    fx, fy = int(a[i, X]) + dx, int(a[i, Y]) + dy
    on = bool(alive and 2 <= act <= 5 and 0 <= fx < env.spec.width and 0 <= fy < env.spec.height)
    f = dict(pos=(int(a[i, X]), int(a[i, Y])), front=(fx, fy), on=on, fwd=False, drop=False, write=False,
             try_fwd=on and act == 2, try_drop=on and act == 4)
    if on:
        t, c, s = (int(v) for v in env.grid[fx, fy])
        f["fwd"] = act == 2 and (t in (EMPTY, GOAL, FLOOR, LAVA) or (t == DOOR and s == OPEN))
        f["drop"] = act == 4 and a[i, CT] != EMPTY and t == EMPTY
        pickup = act == 3 and t in (KEY, BALL, BOX) and a[i, CT] == EMPTY
        unlock = act == 5 and t == DOOR and s == LOCKED and a[i, CT] == KEY and a[i, CC] == c
        flip = act == 5 and t == DOOR and s != LOCKED
        f["write"] = bool(f["drop"] or pickup or unlock or flip or (act == 5 and t == BOX))
    return f


def marked_agents(env, acts, overlap):
    n = env.spec.num_agents
    fact = [action_facts(env, i, int(acts[i])) for i in range(n)]
    asks = [(f["fwd"] and not overlap) or f["drop"] for f in fact]  # wants to know who stands in front
    marked, unstable = [False] * n, [False] * n
    for i in range(n):
        for k in range(n):
            if k == i:
                continue
            me, o = fact[i], fact[k]
            at_front, same_front, at_its_front = o["pos"] == me["front"], o["front"] == me["front"], o["front"] == me["pos"]
            if me["on"] and same_front and o["write"]:
                marked[i] = unstable[i] = True
            if asks[i] and o["fwd"] and (at_front or same_front):
                marked[i] = True
            if o["on"] and same_front and me["write"]:
                marked[i] = True
            if asks[k] and me["fwd"] and (at_its_front or same_front):
                marked[i] = True
    for i in range(n):  # second look: towards agents whose front cell may change
        for k in range(n):
            if k == i or not unstable[k]:
                continue
            me, o = fact[i], fact[k]
            at_front, same_front, at_its_front = o["pos"] == me["front"], o["front"] == me["front"], o["front"] == me["pos"]
            could_ask = (o["try_fwd"] and not overlap) or o["try_drop"]
            if asks[i] and o["try_fwd"] and (at_front or same_front):
                marked[i] = True
            if could_ask and me["fwd"] and (at_its_front or same_front):
                marked[i] = True
    return marked


@pytest.mark.parametrize("name,agents,kw,seeds", [
    ("MultiGrid-LockedHallway-4Rooms-v0", 16, dict(max_steps=40, allow_agent_overlap=False), 100),
    ("MultiGrid-LockedHallway-2Rooms-v0", 6, dict(max_steps=40, allow_agent_overlap=False), 30),
    ("MultiGrid-RedBlueDoors-6x6-v0", 8, dict(max_steps=40, allow_agent_overlap=False), 30),
    ("MultiGrid-BlockedUnlockPickup-v0", 10, dict(max_steps=60, allow_agent_overlap=False), 100),
    ("MultiGrid-BlockedUnlockPickup-v0", 8, dict(max_steps=60), 30),
    ("MultiGrid-Empty-5x5-v0", 8, dict(max_steps=30, allow_agent_overlap=False, success_termination_mode="all"), 30),
])
def test_unmarked_first_equals_sequential(name, agents, kw, seeds):
    spec, gen = envs.make_spec(name, num_agents=agents, **dict(dict(agent_view_size=7), **kw))
    sp = pyref.EnvSpec(**spec)
    replayed = 0
    for seed in range(seeds):
        rng = np.random.default_rng(seed)
        g, a, m, _ = gen(mixin_rng=np.random.default_rng(seed), env_rng=np.random.default_rng(seed + 100))
        e1 = pyref.Env(sp, fast_obs=True)
        e1.load_layout(g, a, m)
        for t in range(40):
            acts = rng.choice(7, size=agents, p=[0.12, 0.12, 0.36, 0.12, 0.08, 0.18, 0.02])
            order = rng.random(agents).argsort()
            mk = marked_agents(e1, acts, spec["allow_agent_overlap"])
            replayed += sum(mk)
            order2 = [i for i in range(agents) if not mk[i]] + [int(i) for i in order if mk[i]]
            e2 = copy.deepcopy(e1)
            d = {i: int(acts[i]) for i in range(agents)}
            r1, r2 = e1.step(d, order), e2.step(d, order2)
            where = f"{name} seed {seed} step {t}"
            np.testing.assert_array_equal(e1.grid, e2.grid, err_msg=where)
            np.testing.assert_array_equal(e1.agents, e2.agents, err_msg=where)
            assert all(r1[1][i] == r2[1][i] and r1[2][i] == r2[2][i] for i in range(agents)), where
            if e1.is_done():
                break
    assert replayed > 0
