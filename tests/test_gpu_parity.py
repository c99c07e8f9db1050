"""HIP path vs the golden trajectories recorded from the reference, through the C ABI."""
import numpy as np
import pytest

from golden_util import INFO_MASK, group_names, load_group, pack_info

pytestmark = pytest.mark.gpu


from multigrid_amd.gym_api import competitive_missions_after_step  # noqa: E402


def _mk(spec, n, **kw):
    from multigrid_amd.batched import BatchedMultiGrid
    return BatchedMultiGrid(spec, n, device=0, **kw)


@pytest.mark.parametrize("name", group_names())
def test_hip_replays_reference(name):
    import torch

    spec, d = load_group(name)
    n = len(d["length"])
    env = _mk(spec, n)
    env.upload_layouts(d["layout_grid"], d["layout_agents"], d["layout_meta"])
    out = env.reset(layout_idx=np.arange(n))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.image.cpu().numpy(), d["reset_image"], err_msg=f"{name} reset obs")
    np.testing.assert_array_equal(out.direction.cpu().numpy(), d["layout_agents"][:, :, 2])
    missions = d["reset_mission"].tolist() if "mission" in d else None
    for t in range(int(d["length"].max())):
        live = d["length"] > t
        out = env.step(d["actions"][:, t], d["order"][:, t])
        st = env.get_state()
        where = f"{name} step {t}"
        np.testing.assert_array_equal(out.image.cpu().numpy()[live], d["image"][live, t], err_msg=where + " image")
        np.testing.assert_array_equal(out.direction.cpu().numpy()[live], d["direction"][live, t],
                                      err_msg=where + " direction")
        np.testing.assert_allclose(out.reward.cpu().numpy()[live], d["reward"][live, t], rtol=0, atol=1e-6,
                                   err_msg=where + " reward")  # reference rewards are float64
        np.testing.assert_array_equal(out.terminated.cpu().numpy()[live].astype(bool), d["terminated"][live, t],
                                      err_msg=where + " terminated")
        np.testing.assert_array_equal(out.truncated.cpu().numpy()[live].astype(bool), d["truncated"][live, t],
                                      err_msg=where + " truncated")
        np.testing.assert_array_equal(st["grid"].cpu().numpy()[live], d["post_grid"][live, t], err_msg=where + " grid")
        np.testing.assert_array_equal(st["agents"].cpu().numpy()[live], d["post_agents"][live, t],
                                      err_msg=where + " agents")
        if "post_flags" in d:  # CompetativeRedBlueDoorEnvV3: key objects used up, doors orphaned, ball discounts
            np.testing.assert_array_equal(st["flags"].cpu().numpy()[live], d["post_flags"][live, t], err_msg=where + " flags")
            np.testing.assert_array_equal(st["meta"].cpu().numpy()[live], d["post_meta"][live, t], err_msg=where + " meta")
        if "info" in d:  # the step's info dict (multigrid/envs/competative_red_blue_door.py:380-388), a byte per agent
            info = out.info.cpu().numpy()
            np.testing.assert_array_equal(info[live] & INFO_MASK, pack_info(d["info"][live, t]), err_msg=where + " info")
            # bits 6 / 7 (rewarded key / ball pickups) drive the mission texts the next observations carry (:438-468)
            np.testing.assert_array_equal(np.asarray(missions)[live], d["mission"][live, t], err_msg=where + " missions")
            rows = st["agents"].cpu().numpy()
            missions = [competitive_missions_after_step(missions[k], info[k], d["order"][k, t], rows[k]) for k in range(n)]
    env.close()
