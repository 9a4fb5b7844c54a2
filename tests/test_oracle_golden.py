"""Pin the oracle (oracle/evac_oracle.py) against the fixtures generated from the real reference.

CPU only.  f64 'ref' precision must match to 1e-12 teacher-forced AND free-running (the oracle is
deterministic given the reset draws, the actions and the per-pedestrian noise)."""
import json
import os

import numpy as np
import pytest

from oracle import evac_oracle as O
from tests import helpers as H

TOL = 1e-12


@pytest.mark.parametrize("path", H.traj_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_reset_matches_reference(path):
    d = np.load(path)
    p = H.load_params(d["params_json"])
    st = O.env_reset(p, d["draw_pos"], d["draw_dir"])
    np.testing.assert_allclose(st.pos, d["pos"][0], rtol=0, atol=0)
    np.testing.assert_allclose(st.dir, d["dir"][0], rtol=TOL, atol=TOL)
    np.testing.assert_array_equal(st.status, d["status"][0])
    np.testing.assert_array_equal(st.agent_pos, d["agent_pos"][0])
    np.testing.assert_array_equal(st.agent_dir, d["agent_dir"][0])
    assert st.pos.dtype == np.float64 and st.agent_pos.dtype == np.float32


def _check_step(out, st, d, k):
    np.testing.assert_allclose(st.pos, d["pos"][k + 1], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(st.dir, d["dir"][k + 1], rtol=TOL, atol=TOL)
    np.testing.assert_array_equal(st.status, d["status"][k + 1])
    np.testing.assert_allclose(st.agent_pos, d["agent_pos"][k + 1], rtol=0, atol=0)
    np.testing.assert_allclose(st.agent_dir, d["agent_dir"][k + 1], rtol=0, atol=0)
    assert st.now == d["now"][k + 1]
    for key in ("reward", "reward_agent", "reward_ped", "intrinsic"):
        np.testing.assert_allclose(out[key], d[key][k], rtol=TOL, atol=TOL, err_msg=key)
    assert out["terminated"] == bool(d["terminated"][k])
    assert out["truncated"] == bool(d["truncated"][k])


@pytest.mark.parametrize("path", H.traj_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_teacher_forced_steps(path):
    d = np.load(path)
    p = H.load_params(d["params_json"])
    for k in range(len(d["action"])):
        st = H.state_at(d, k)
        out = O.env_step(p, st, d["action"][k], d["noise"][k])
        _check_step(out, st, d, k)


@pytest.mark.parametrize("path", H.episode_record_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_episode_record_matches_the_references_log(path):
    """The nine-key dict the reference logs at the reset that follows an episode (env.py:114-127), captured from the reference itself:
    the oracle's bookkeeping (EpisodeLog) over the same steps gives the same sums (1e-9) and exactly the same counts."""
    d = np.load(path)
    p = H.load_params(d["params_json"])
    keys = json.loads(str(d["episode_record_keys"]))
    assert tuple(keys) == O.EpisodeLog.KEYS
    log = O.EpisodeLog()
    for k in range(len(d["action"])):
        st = H.state_at(d, k)                                  # teacher-forced: the reference's own pre-state of every step
        log.after_step(O.env_step(p, st, d["action"][k], d["noise"][k]))
    assert bool(d["terminated"][-1]) or bool(d["truncated"][-1])
    rec = log.record(st)
    for k, v in zip(keys, d["episode_record"]):
        if k.endswith("_pedestrians") or k in ("episode_length", "overall_timesteps"):
            assert rec[k] == int(v), (k, rec[k], v)
        else:
            np.testing.assert_allclose(rec[k], v, rtol=1e-9, atol=1e-9, err_msg=k)
    assert sum(rec[k] for k in keys if k.endswith("_pedestrians")) == p.number_of_pedestrians
    if "all_escaped" in path:
        assert rec["escaped_pedestrians"] == p.number_of_pedestrians and bool(d["terminated"][-1]) and not bool(d["truncated"][-1])


@pytest.mark.parametrize("path", H.traj_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_free_running_episode(path):
    d = np.load(path)
    p = H.load_params(d["params_json"])
    st = O.env_reset(p, d["draw_pos"], d["draw_dir"])
    for k in range(len(d["action"])):
        out = O.env_step(p, st, d["action"][k], d["noise"][k])
        # free-running f64: rounding differences of the mask-based sums may accumulate a little
        np.testing.assert_allclose(st.pos, d["pos"][k + 1], rtol=1e-9, atol=1e-9)
        np.testing.assert_array_equal(st.status, d["status"][k + 1])
        np.testing.assert_allclose(out["reward"], d["reward"][k], rtol=1e-9, atol=1e-9)
        assert out["terminated"] == bool(d["terminated"][k]) and out["truncated"] == bool(d["truncated"][k])


@pytest.mark.parametrize("path", H.traj_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_observation_variants(path):
    d = np.load(path)
    p = H.load_params(d["params_json"])
    for j, k in enumerate(d["obs_index"]):
        st = H.state_at(d, int(k))
        H.check_observations(lambda key: d[key][j], st, p.eps)


@pytest.mark.parametrize("name,c", list(H.crafted_cases()), ids=[n for n, _ in H.crafted_cases()])
def test_crafted_edge_cases(name, c):
    p = H.load_params(c["params_json"])
    st = O.OracleState(c["pre_pos"].copy(), c["pre_dir"].copy(), c["pre_status"].copy(),
                       c["pre_agent_pos"].copy(), c["pre_agent_dir"].copy(), int(c["pre_now"]))
    with np.errstate(all="ignore"):
        out = O.env_step(p, st, c["action"], c["noise"])
    np.testing.assert_allclose(st.pos, c["post_pos"], rtol=TOL, atol=TOL, equal_nan=True)
    np.testing.assert_allclose(st.dir, c["post_dir"], rtol=TOL, atol=TOL, equal_nan=True)
    np.testing.assert_array_equal(st.status, c["post_status"])
    np.testing.assert_array_equal(st.agent_pos, c["post_agent_pos"])
    np.testing.assert_array_equal(st.agent_dir, c["post_agent_dir"])
    for key in ("reward", "reward_agent", "reward_ped", "intrinsic"):
        np.testing.assert_allclose(out[key], c[key], rtol=TOL, atol=TOL, equal_nan=True, err_msg=key)
    assert out["terminated"] == bool(c["terminated"]) and out["truncated"] == bool(c["truncated"])
    with np.errstate(all="ignore"):
        H.check_observations(lambda key: c[key], st, p.eps)


def test_crafted_cases_cover_the_edges():
    """The fixture really contains the situations SURVEY.md 8(c) lists."""
    cases = dict(H.crafted_cases())
    assert cases["leader_wall_hit"]["reward_agent"] == -5.0
    assert bool(cases["leader_wall_hit_terminates"]["terminated"])
    assert bool(cases["all_escape"]["terminated"]) and (cases["all_escape"]["post_status"] == 4).all()
    assert bool(cases["truncation"]["truncated"]) and not bool(cases["truncation"]["terminated"])
    assert np.isnan(cases["nan_poison_zero_heading"]["post_pos"]).any()
    assert (cases["exiting_lands_on_exit"]["post_pos"][0] == [0.0, -1.0]).all()
    c = cases["corner_reflection"]
    assert (c["post_pos"][0] < 1.0).all() and (c["post_dir"][0] < 0).all()
    assert cases["reward_transitions"]["reward"] > 40
    assert np.all(cases["escaped_pinned_no_fv"]["noise"] == 0)
    assert (cases["zero_action"]["post_agent_dir"] == 0).all()


def test_f32_mode_close_to_reference_off_ties():
    """precision='f32' (what the GPU computes in) stays within 1e-5 of the f64 reference for one
    teacher-forced step whenever no threshold comparison is within 1e-6 of a tie."""
    worst = 0.0
    for path in H.traj_files():
        d = np.load(path)
        p = H.load_params(d["params_json"])
        for k in range(len(d["action"])):
            if d["margin"][k] < 1e-6:
                continue
            st = H.state_at(d, k, np.float32)
            ref = H.state_at(d, k, np.float64)
            ref.pos = st.pos.astype(np.float64); ref.dir = st.dir.astype(np.float64)   # identical f32-representable input
            o32 = O.env_step(p, st, d["action"][k], d["noise"][k], precision="f32")
            o64 = O.env_step(p, ref, d["action"][k], d["noise"][k], precision="ref")
            assert st.pos.dtype == np.float32
            if O.threshold_margin(ref.pos, ref.agent_pos, H.state_at(d, k).pos, p.width, p.height, ref.status) < 1e-6:
                continue
            np.testing.assert_array_equal(st.status, ref.status)
            err = max(np.abs(st.pos - ref.pos).max(), np.abs(st.dir - ref.dir).max())
            worst = max(worst, float(err))
            assert err < 1e-5
            assert abs(o32["reward"] - o64["reward"]) < 1e-4 * max(1.0, abs(o64["reward"]))
    assert worst < 1e-6
