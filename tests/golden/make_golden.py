#!/usr/bin/env python3
"""Generate golden fixtures by running the REAL reference (/root/reference) -- build container only.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

The fixtures are data only: inputs (reset draws, actions, the per-step noise the reference drew
from the global NumPy RNG, scattered to pedestrian index) and the reference's outputs
(trajectories, rewards, flags, every observation variant).  No reference source text is stored.
See _reference_loader.py for how the reference is imported without gymnasium/wandb.

Fixture kinds
  traj_*.npz     free-running episodes from ``np.random.seed(seed); env.reset()``
  crafted.npz    single steps from hand-built states exercising the edge cases of SURVEY 8(c)
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _reference_loader as L  # noqa: E402

ALPHAS = (2, 3, 5)


def _params_dict(cfg) -> dict:
    keys = ("number_of_pedestrians", "width", "height", "step_size", "noise_coef", "eps",
            "enslaving_degree", "is_new_exiting_reward", "is_new_followers_reward",
            "intrinsic_reward_coef", "is_termination_agent_wall_collision",
            "init_reward_each_step", "max_timesteps")
    return {k: getattr(cfg, k) for k in keys}


class RefHarness:
    """One reference EvacuationEnv plus every observation-wrapper variant around it."""

    def __init__(self, ref, **cfg_kw):
        self.ref = ref
        self.cfg = ref.EnvConfig(wandb_enabled=False, path_logs=L.log_dir(), giff_freq=10**9, **cfg_kw)
        self.env = ref.EvacuationEnv(self.cfg)
        W = ref.EnvWrappersConfig
        self.wrappers = {}
        for a in ALPHAS:
            self.wrappers[f"grav_a{a}"] = W(positions="grav", alpha=a).wrap_env(self.env)
        for pos in ("abs", "rel"):
            for st in ("no", "ohe", "cat"):
                self.wrappers[f"{pos}_{st}_box"] = W(positions=pos, statuses=st, type="Box").wrap_env(self.env)
                if not (pos == "abs" and st == "no"):
                    self.wrappers[f"{pos}_{st}_dict"] = W(positions=pos, statuses=st, type="Dict").wrap_env(self.env)
        # spy on the reference's own return values (no arithmetic is altered)
        self._last = {}
        area = self.env.area
        orig_ped, orig_agent = area.pedestrians_step, area.agent_step

        def ped_spy(peds, agent, now):
            out = orig_ped(peds, agent, now)
            self._last.update(term_ped=out[1], reward_ped=out[2], intrinsic=out[3])
            return out

        def agent_spy(action, agent):
            out = orig_agent(action, agent)
            self._last.update(term_agent=out[1], reward_agent=out[2])
            return out

        area.pedestrians_step, area.agent_step = ped_spy, agent_spy

    # -- state access -------------------------------------------------------------------
    def snapshot(self) -> dict:
        u = self.env
        return dict(pos=u.pedestrians.positions.copy(), dir=u.pedestrians.directions.copy(),
                    status=L.status_codes(u.pedestrians.statuses),
                    agent_pos=np.array(u.agent.position, dtype=np.float32),
                    agent_dir=np.array(u.agent.direction, dtype=np.float32), now=int(u.time.now))

    def set_state(self, pos, dr, status, agent_pos, agent_dir, now):
        u = self.env
        S = self.ref.Status
        by_code = {s.value: s for s in S}
        u.pedestrians.positions = np.array(pos, dtype=np.float64)
        u.pedestrians.directions = np.array(dr, dtype=np.float64)
        u.pedestrians.statuses = np.array([by_code[int(c)] for c in status])
        u.agent.position = np.array(agent_pos, dtype=np.float32)
        u.agent.direction = np.array(agent_dir, dtype=np.float32)
        u.time.now = int(now)

    def _through_chain(self, w, obs):
        """Apply the wrapper chain inner-to-outer, as gymnasium's ObservationWrapper.step does."""
        if w is self.env:
            return obs
        return w.observation(self._through_chain(w.env, obs))

    def observations(self) -> dict:
        out = {}
        for name, w in self.wrappers.items():
            o = self._through_chain(w, self.env._get_observation())
            if isinstance(o, dict):
                for k, v in o.items():
                    out[f"obs_{name}__{k}"] = np.array(v)
            else:
                out[f"obs_{name}"] = np.array(o)
        return out

    # -- stepping with noise capture ------------------------------------------------------
    def peek_noise(self) -> np.ndarray:
        """The draw the next step will consume (area.py:124), scattered to pedestrian index."""
        codes = L.status_codes(self.env.pedestrians.statuses)
        fv = (codes == 1) | (codes == 2)
        c = self.cfg.noise_coef
        st = np.random.get_state()
        nz = np.random.uniform(low=-c / 2, high=c / 2, size=int(fv.sum()))
        np.random.set_state(st)
        out = np.zeros(codes.shape[0])
        out[fv] = nz
        return out

    def reset(self):
        n = self.cfg.number_of_pedestrians
        st = np.random.get_state()
        a = np.random.uniform(-1.0, 1.0, size=(n, 2))
        b = np.random.uniform(-1.0, 1.0, size=(n, 2))
        np.random.set_state(st)
        self.env.reset()
        return a, b

    def step(self, action):
        noise = self.peek_noise()
        self._last = {}
        obs, reward, terminated, truncated, _ = self.env.step(np.array(action, dtype=np.float32))
        rec = dict(noise=noise, reward=float(reward), terminated=bool(terminated), truncated=bool(truncated),
                   reward_agent=float(self._last["reward_agent"]), reward_ped=float(self._last["reward_ped"]),
                   intrinsic=float(self._last["intrinsic"]))
        return rec


EPISODE_RECORD_KEYS = ("episode_intrinsic_reward", "episode_status_reward", "episode_reward", "episode_length", "escaped_pedestrians",
                       "exiting_pedestrians", "following_pedestrians", "viscek_pedestrians", "overall_timesteps")


def capture_episode_record(h: "RefHarness") -> np.ndarray:
    """The dict the reference logs for the episode that just ended (env.py:114-127): it is built and emitted by the NEXT reset(), so
    that reset is made here, with the reference's own wandb hand-off switched on and pointed at a recorder (the shell module of
    _reference_loader; no arithmetic involved).  The global RNG is put back, so the fixture's recorded draws stay what they were."""
    wandb = sys.modules["wandb"]
    got = []
    old_log, old_flag = wandb.log, h.env.wandb_enabled
    st = np.random.get_state()
    try:
        wandb.log = lambda d, *a, **k: got.append(dict(d))
        h.env.wandb_enabled = True
        h.env.reset()
    finally:
        wandb.log, h.env.wandb_enabled = old_log, old_flag
        np.random.set_state(st)
    assert len(got) == 1 and tuple(got[0]) == EPISODE_RECORD_KEYS, got
    return np.array([float(got[0][k]) for k in EPISODE_RECORD_KEYS], dtype=np.float64)


def _margin(pre, post, width, height):
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle.evac_oracle import threshold_margin
    return threshold_margin(post["pos"], post["agent_pos"], pre["pos"], width, height, post["status"])


def make_trajectory(ref, name, seed, steps, obs_every=1, **cfg_kw):
    h = RefHarness(ref, **cfg_kw)
    np.random.seed(seed)
    draw_pos, draw_dir = h.reset()
    act_rng = np.random.Generator(np.random.PCG64(1000 + seed))
    snaps = [h.snapshot()]
    obs = [h.observations()]
    recs = []
    actions = []
    margins = []
    for k in range(steps):
        a = act_rng.uniform(-1.0, 1.0, size=2).astype(np.float32)
        pre = snaps[-1]
        rec = h.step(a)
        post = h.snapshot()
        actions.append(a)
        recs.append(rec)
        snaps.append(post)
        obs.append(h.observations())
        margins.append(_margin(pre, post, h.cfg.width, h.cfg.height))
        if rec["terminated"] or rec["truncated"]:
            break
    out = dict(params_json=json.dumps(_params_dict(h.cfg)), seed=seed, draw_pos=draw_pos, draw_dir=draw_dir,
               action=np.array(actions, dtype=np.float32), margin=np.array(margins))
    if recs[-1]["terminated"] or recs[-1]["truncated"]:      # the episode ended: what the reference logs for it (env.py:114-127)
        out["episode_record_keys"] = json.dumps(list(EPISODE_RECORD_KEYS))
        out["episode_record"] = capture_episode_record(h)
    for k in ("pos", "dir", "status", "agent_pos", "agent_dir", "now"):
        out[k] = np.array([s[k] for s in snaps])
    for k in ("noise", "reward", "reward_agent", "reward_ped", "intrinsic", "terminated", "truncated"):
        out[k] = np.array([r[k] for r in recs])
    idx = sorted(set(list(range(0, len(snaps), obs_every)) + [len(snaps) - 1]))
    out["obs_index"] = np.array(idx)
    for k in obs[0]:
        out[k] = np.array([obs[i][k] for i in idx])
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    if "episode_record" in out:
        print(f"   episode record: " + ", ".join(f"{k}={v:g}" for k, v in zip(EPISODE_RECORD_KEYS, out["episode_record"])))
    print(f"{name}: steps={len(recs)} N={h.cfg.number_of_pedestrians} "
          f"term={recs[-1]['terminated']} trunc={recs[-1]['truncated']} min_margin={min(margins):.2e} "
          f"size={os.path.getsize(path)/1024:.0f} KiB")


def make_terminating_episode(ref, name="episode_all_escaped", seed=12, n=12):
    """An episode that TERMINATES because every pedestrian has escaped (area.py:175-178), for the counts of the episode record
    (env.py:114-127): after a normal reset the state is set by hand -- a crowd within the exit's radius, two stragglers a little
    outside it and next to the leader -- and the reference runs on RandomAgent-style actions until it says `terminated`.  The fixture
    holds the state it started from, the actions, the noise the reference drew, every step's outputs and the record."""
    h = RefHarness(ref, number_of_pedestrians=n, is_new_exiting_reward=True, is_new_followers_reward=True, intrinsic_reward_coef=1.0,
                   max_timesteps=400)
    np.random.seed(seed)
    h.reset()
    rng = np.random.Generator(np.random.PCG64(77))
    ang = rng.uniform(0.15 * np.pi, 0.85 * np.pi, size=n)
    rad = rng.uniform(0.05, 0.38, size=n)
    rad[:2] = (0.47, 0.52)                                     # two stragglers outside the exit's radius (0.4) ...
    pos = np.stack([rad * np.cos(ang), -1.0 + rad * np.sin(ang)], axis=1)
    agent_pos = np.array([pos[0, 0] + 0.05, pos[0, 1] + 0.05], dtype=np.float32)     # ... with the leader among them
    dr = np.stack([_unit(v) * 0.01 for v in rng.uniform(-1, 1, size=(n, 2))])
    status = np.where(rad < 0.4, 3, 2)                         # EXITING | FOLLOWER (what update_statuses gives for these positions)
    h.set_state(pos, dr, status, agent_pos, np.zeros(2, np.float32), 0)
    snaps, recs, actions = [h.snapshot()], [], []
    for k in range(400):
        a = np.array([0.0, -1.0], dtype=np.float32) + rng.uniform(-0.3, 0.3, size=2).astype(np.float32)    # towards the exit, wobbling
        rec = h.step(a)
        actions.append(a); recs.append(rec); snaps.append(h.snapshot())
        if rec["terminated"] or rec["truncated"]:
            break
    assert recs[-1]["terminated"] and not recs[-1]["truncated"], "the crafted episode did not terminate"
    out = dict(params_json=json.dumps(_params_dict(h.cfg)), seed=seed, action=np.array(actions, dtype=np.float32),
               episode_record_keys=json.dumps(list(EPISODE_RECORD_KEYS)), episode_record=capture_episode_record(h))
    for k in ("pos", "dir", "status", "agent_pos", "agent_dir", "now"):
        out[k] = np.array([s[k] for s in snaps])
    for k in ("noise", "reward", "reward_agent", "reward_ped", "intrinsic", "terminated", "truncated"):
        out[k] = np.array([r[k] for r in recs])
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: steps={len(recs)} N={n} terminated; record: " + ", ".join(f"{k}={v:g}" for k, v in zip(EPISODE_RECORD_KEYS, out["episode_record"])))


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def crafted_cases():
    """Hand-built pre-states (N=8) for the edge cases listed in SURVEY.md 8(c).  Statuses are the
    ones the reference's classifier would assign to these positions unless a case says otherwise."""
    N = 8
    far = np.array([[-0.9 + 0.23 * i, 0.9 - 0.013 * i] for i in range(N)])   # harmless VISCEK filler row, no neighbours, no ties
    base_dir = np.tile(_unit([1.0, 0.3]) * 0.01, (N, 1))
    cases = []

    def case(name, pos, dr, status, agent_pos, agent_dir, action, now=3, **cfg):
        cases.append(dict(name=name, pos=np.array(pos, float), dir=np.array(dr, float),
                          status=np.array(status, np.int8), agent_pos=np.array(agent_pos, np.float32),
                          agent_dir=np.array(agent_dir, np.float32), action=np.array(action, np.float32),
                          now=now, cfg=cfg))

    # 1. leader hits the wall: move rejected, reward -5, but its direction still enslaves followers
    pos = far.copy(); pos[0] = [0.93, 0.05]; pos[1] = [0.90, -0.08]
    st = [2, 2, 1, 1, 1, 1, 1, 1]
    case("leader_wall_hit", pos, base_dir, st, [0.995, 0.0], [0.01, 0.0], [1.0, 0.0])
    # 1b. same with termination-on-collision enabled and partial enslaving
    case("leader_wall_hit_terminates", pos, base_dir, st, [0.995, 0.0], [0.01, 0.0], [1.0, 0.2],
         is_termination_agent_wall_collision=True, enslaving_degree=0.5)
    # 2. pedestrian leaves through a corner: reflection on both axes, both direction signs flip
    pos = far.copy(); pos[0] = [0.996, 0.997]
    dr = base_dir.copy(); dr[0] = _unit([1.0, 1.0]) * 0.01
    case("corner_reflection", pos, dr, [1] * N, [0.0, 0.0], [0.0, 0.01], [0.0, 1.0])
    # 2b. left / bottom walls with a bigger step
    pos = far.copy(); pos[0] = [-0.98, 0.5]; pos[1] = [0.3, -0.97]; pos[1] = [0.95, -0.97]
    dr = base_dir.copy() * 5; dr[0] = _unit([-1.0, 0.1]) * 0.05; dr[1] = _unit([0.2, -1.0]) * 0.05
    case("left_bottom_reflection", pos, dr, [1] * N, [0.0, 0.0], [0.0, 0.05], [0.3, 1.0], step_size=0.05)
    # 3. exiting pedestrian closer to the exit than one step: lands exactly on the exit (step 0.05)
    pos = far.copy(); pos[0] = [0.02, -0.98]; pos[1] = [-0.2, -0.8]; pos[2] = [0.0, -0.985]
    st = [3, 3, 3, 1, 1, 1, 1, 1]
    case("exiting_lands_on_exit", pos, base_dir, st, [0.5, 0.5], [0.0, 0.05], [1.0, 1.0], step_size=0.05,
         is_new_exiting_reward=True)
    # 4. everybody escapes this step -> terminated
    ang = np.linspace(0.3, 2.8, N)
    pos = np.stack([0.0 + 0.015 * np.cos(ang), -1.0 + 0.015 * np.sin(ang)], axis=1)
    case("all_escape", pos, base_dir, [3] * N, [0.5, 0.5], [0.0, 0.01], [1.0, 0.0], is_new_exiting_reward=True)
    # 4b. already-escaped pedestrians stay pinned, nobody consumes noise (empty draw)
    st = [4, 4, 4, 4, 3, 3, 3, 3]
    case("escaped_pinned_no_fv", pos, base_dir, st, [0.5, 0.5], [0.0, 0.01], [-1.0, 0.4])
    # 5. truncation: now+1 reaches max_timesteps
    case("truncation", far, base_dir, [1] * N, [0.0, 0.0], [0.0, 0.01], [0.2, -0.7], now=4, max_timesteps=5)
    # 6. new followers + new exiting rewards, intrinsic reward on, status transitions V->F, V->E, F->E
    pos = far.copy()
    pos[0] = [0.195, 0.0]; pos[1] = [0.0, -0.2095]; pos[2] = [0.35, -0.797]; pos[3] = [0.05, 0.05]
    dr = base_dir.copy(); dr[0] = [-0.01, 0.0]; dr[1] = [0.0, 0.01]; dr[2] = _unit([-0.35, -0.2]) * 0.01
    st = [1, 1, 1, 2, 1, 1, 1, 1]
    case("reward_transitions", pos, dr, st, [0.0, 0.0], [0.01, 0.0], [-0.3, -1.0], now=100,
         is_new_exiting_reward=True, is_new_followers_reward=True, intrinsic_reward_coef=1.0)
    # 7. Vicsek averaging: a tight cluster (all within 0.1) with different headings, plus an exiting neighbour
    pos = far.copy()
    pos[0] = [0.30, -0.62]; pos[1] = [0.33, -0.60]; pos[2] = [0.28, -0.58]; pos[3] = [0.31, -0.68]  # [3] is EXITING
    dr = base_dir.copy(); dr[0] = _unit([1, 0]) * .01; dr[1] = _unit([0, 1]) * .01; dr[2] = _unit([-1, 1]) * .01
    dr[3] = _unit([-0.31, -0.32]) * .01
    st = [1, 1, 1, 3, 1, 1, 1, 1]
    case("cluster_with_exiting_neighbour", pos, dr, st, [-0.5, 0.5], [0.0, 0.01], [0.1, 0.9], noise_coef=0.5)
    # 8. zero action with a fully enslaved follower: 0/0 heading -> NaN poisons every fv pedestrian next step
    pos = far.copy(); pos[0] = [0.05, 0.05]
    st = [2, 1, 1, 1, 1, 1, 1, 1]
    dr = base_dir.copy(); dr[0] = [0.0, 0.0]                     # what a zero action leaves behind
    case("nan_poison_zero_heading", pos, dr, st, [0.0, 0.0], [0.0, 0.0], [0.5, 0.5])
    # 9. zero action itself (direction becomes exactly 0, leader does not move)
    case("zero_action", far, base_dir, [1] * N, [0.1, 0.1], [0.0, 0.01], [0.0, 0.0])
    # 10. exactly antiparallel neighbours: mean heading (0,0) -> arctan2(0,0) = 0
    pos = far.copy(); pos[0] = [0.5, 0.0]; pos[1] = [0.52, 0.0]
    dr = base_dir.copy(); dr[0] = [0.01, 0.0]; dr[1] = [-0.01, 0.0]
    case("antiparallel_zero_mean", pos, dr, [1] * N, [-0.5, 0.5], [0.0, 0.01], [0.1, 0.9])
    return cases


def make_crafted(ref):
    out = {}
    names = []
    for c in crafted_cases():
        n = c["pos"].shape[0]
        h = RefHarness(ref, number_of_pedestrians=n, **c["cfg"])
        np.random.seed(12345)
        h.reset()
        h.set_state(c["pos"], c["dir"], c["status"], c["agent_pos"], c["agent_dir"], c["now"])
        pre = h.snapshot()
        with np.errstate(all="ignore"):
            rec = h.step(c["action"])
        post = h.snapshot()
        with np.errstate(all="ignore"):
            obs = h.observations()
        name = c["name"]
        if np.isfinite(post["pos"]).all():
            mg = _margin(pre, post, h.cfg.width, h.cfg.height)
            assert mg > 1e-5, (name, mg)      # crafted cases must not sit on a threshold
        names.append(name)
        out[f"{name}__params_json"] = json.dumps(_params_dict(h.cfg))
        out[f"{name}__action"] = c["action"]
        for k, v in pre.items():
            out[f"{name}__pre_{k}"] = np.array(v)
        for k, v in post.items():
            out[f"{name}__post_{k}"] = np.array(v)
        for k, v in rec.items():
            out[f"{name}__{k}"] = np.array(v)
        for k, v in obs.items():
            out[f"{name}__{k}"] = v
        print(f"crafted/{name}: reward={rec['reward']:.4f} term={rec['terminated']} trunc={rec['truncated']} "
              f"status {pre['status'].tolist()} -> {post['status'].tolist()}")
    out["names"] = np.array(names)
    path = os.path.join(HERE, "crafted.npz")
    np.savez_compressed(path, **out)
    print(f"crafted: {len(names)} cases, {os.path.getsize(path)/1024:.0f} KiB")


def main(argv=None):
    """`make_golden.py` regenerates every fixture; `make_golden.py NAME [NAME ...]` only the named trajectories / "crafted"."""
    only = set(sys.argv[1:] if argv is None else argv)
    if not L.reference_available():
        raise SystemExit("reference not found at " + L.REFERENCE_ROOT)
    ref = L.load_full()

    def traj(name, *a, **kw):
        if not only or name in only:
            make_trajectory(ref, name, *a, **kw)
    # free-running episodes -- the hyper-parameter values are the ones the reference's sweeps use
    # (run_scripts: n=60, noise .2/.5, enslaving 1/.5/.1, alpha 2..5); sizes per SURVEY.md 8(c): N in {60, 256} x 64 steps x 4
    # seeds, N = 1024 x 4 steps x 2 seeds
    traj("traj_n60_s0", 0, 64, obs_every=2, number_of_pedestrians=60, is_new_exiting_reward=True)
    traj("traj_n60_s1_noise05_ens05", 1, 64, obs_every=4, number_of_pedestrians=60, noise_coef=0.5,
         enslaving_degree=0.5, intrinsic_reward_coef=1.0)
    traj("traj_n60_s2_ens01", 2, 64, obs_every=4, number_of_pedestrians=60, enslaving_degree=0.1,
         is_new_followers_reward=False, init_reward_each_step=0.0)
    traj("traj_n60_s3_step05_trunc", 3, 40, obs_every=4, number_of_pedestrians=60, step_size=0.05,
         max_timesteps=40, is_new_exiting_reward=True, intrinsic_reward_coef=1.0)
    traj("traj_n10_s4_long", 4, 400, obs_every=16, number_of_pedestrians=10, is_new_exiting_reward=True)
    traj("traj_n256_s5", 5, 64, obs_every=8, number_of_pedestrians=256, is_new_exiting_reward=True)
    traj("traj_n256_s6_noise05", 6, 64, obs_every=8, number_of_pedestrians=256, noise_coef=0.5, enslaving_degree=0.5)
    traj("traj_n1024_s7", 7, 4, obs_every=4, number_of_pedestrians=1024, is_new_exiting_reward=True)
    traj("traj_n1024_s8_noise05_ens05", 8, 4, obs_every=4, number_of_pedestrians=1024, noise_coef=0.5,
         enslaving_degree=0.5, intrinsic_reward_coef=1.0)
    traj("traj_n256_s9_ens01_step05", 9, 64, obs_every=16, number_of_pedestrians=256, enslaving_degree=0.1,
         step_size=0.05, is_new_exiting_reward=True, intrinsic_reward_coef=1.0)
    traj("traj_n256_s10_noreward", 10, 64, obs_every=16, number_of_pedestrians=256,
         is_new_followers_reward=False, init_reward_each_step=0.0, noise_coef=0.05)
    if not only or "episode_all_escaped" in only:
        make_terminating_episode(ref)
    if not only or "crafted" in only:
        make_crafted(ref)


if __name__ == "__main__":
    main()
