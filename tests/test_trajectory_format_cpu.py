"""The frame lists built by evacuation_amd.trajectory are the REFERENCE's own rendering memory (SURVEY.md 8(f)
row 4): the reference env is run here with draw=True, so that ITS Pedestrians.save / Agent.save fill
``pedestrians.memory`` / ``agent.memory`` (pedestrians.py:33-35, area.py:32-33, env.py:137,153-155); the same
trajectory, laid out like rollout(capture_envs=K)'s device record, must come out of capture_to_memory as the same
lists -- lengths, element types, dtypes, shapes and values.  (The reference's save_animation itself pins
matplotlib 3.8.0 and does not run under this image's 3.10, whatever the data.)  Needs /root/reference."""
import os

import numpy as np
import pytest

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists in the build container only")
def test_capture_to_memory_equals_the_reference_memory(tmp_path):
    from evacuation_amd.trajectory import capture_to_memory, feed_reference_env
    from tests.golden import _reference_loader as L
    pkg = L.load_full()
    n, T = 17, 25
    cfg = pkg.EnvConfig(number_of_pedestrians=n, wandb_enabled=False, draw=True, giff_freq=10 ** 9,
                        path_logs=str(tmp_path / "logs"), path_giff=str(tmp_path / "giff"), experiment_name="capture")
    ref = pkg.EvacuationEnv(cfg)
    np.random.seed(5)
    ref.reset()
    rng = np.random.default_rng(1)
    initial = {"pos": ref.pedestrians.positions.copy()[None], "status": L.status_codes(ref.pedestrians.statuses)[None]}
    pos, st, ag = [], [], []
    for _ in range(T):
        ref.step(rng.uniform(-1, 1, 2).astype(np.float32))
        pos.append(ref.pedestrians.positions.copy())
        st.append(L.status_codes(ref.pedestrians.statuses))
        ag.append(ref.agent.position.copy())
    # the layout of rollout(capture_envs=1): [T, K, N, 2] / [T, K, N] (status codes as f32) / [T, K, 2]
    ro = {"positions": np.stack(pos)[:, None], "statuses": np.stack(st)[:, None].astype(np.float32), "agent_positions": np.stack(ag)[:, None]}
    ped_mem, agent_mem = capture_to_memory(ro, 0, initial=initial, status_cls=pkg.Status)
    ref_ped, ref_agent = ref.pedestrians.memory, ref.agent.memory
    assert set(ped_mem) == set(ref_ped) and set(agent_mem) == set(ref_agent)
    assert len(ped_mem["positions"]) == len(ref_ped["positions"]) == T + 1            # reset frame + one per step
    assert len(ped_mem["statuses"]) == len(ref_ped["statuses"]) == T + 1
    assert len(agent_mem["position"]) == len(ref_agent["position"]) == T              # the leader has no reset frame
    for a, b in zip(ped_mem["positions"], ref_ped["positions"]):
        assert a.dtype == b.dtype and a.shape == b.shape and (a == b).all()
    for a, b in zip(ped_mem["statuses"], ref_ped["statuses"]):
        assert a.dtype == b.dtype == object and a.shape == b.shape and all(x is y for x, y in zip(a, b))
    for a, b in zip(agent_mem["position"], ref_agent["position"]):
        assert a.dtype == b.dtype and a.shape == b.shape and (a == b).all()
    # and what save_animation reads survives the hand-over
    fresh = pkg.EvacuationEnv(cfg)
    fresh.reset()
    feed_reference_env(fresh, ped_mem, agent_mem)
    sel = fresh.pedestrians.memory["statuses"][T] == pkg.Status.VISCEK                # env.py:303 does exactly this
    assert sel.dtype == bool and sel.shape == (n,) and fresh.time.now == T
    assert fresh.pedestrians.memory["positions"][T][sel, 0].shape == (int(sel.sum()),)
