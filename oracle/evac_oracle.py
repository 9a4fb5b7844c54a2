"""CPU oracle for the evacuation env hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a plain-NumPy restatement of the reference algorithm (cinemere/evacuation,
``/root/reference``): the leader move, the leader-augmented Vicsek pedestrian update, the
positional status classifier, the status / intrinsic rewards, ``EvacuationEnv.step/reset``
orchestration and the observation wrappers.  Every function cites the reference file:line it
follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.  The product
package ``evacuation_amd`` never imports it and has no CPU fallback.

Pinning: the reference holds no tests, golden vectors or fixtures for this
path, and its arithmetic bottoms out in unpinned third-party packages (numpy, scipy
``distance_matrix``).  The oracle is therefore pinned against **outputs of the reference
itself run in the build container**: ``tests/golden/make_golden.py`` imports the reference
from ``/root/reference`` and records per-step inputs/outputs to ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` replays them through this file (f64, tol 1e-12).
``tests/test_oracle_vs_reference.py`` additionally runs oracle and reference side by side
whenever ``/root/reference`` is present.

Formulation differences from the reference (results identical, see tests):
* boolean-mask compaction (``pos[fv]``, ``pos[efv]``) is replaced by full ``[N]``/``[N,N]``
  arrays with 0/1 weights, so the per-step noise is indexed by pedestrian (the reference's
  ``size=n_fv`` draw is scattered to the fv indices in ascending order, which is the order
  the reference consumes it in);
* statuses are int8 codes (VISCEK=1, FOLLOWER=2, EXITING=3, ESCAPED=4; statuses.py:16-27)
  instead of an object array of Enum members;
* ``precision='ref'`` reproduces the reference's dtypes (pedestrians f64, leader/exit f32);
  ``precision='f32'`` runs everything in f32 (what the HIP kernels compute in).
"""
from __future__ import annotations

import dataclasses
from typing import Dict, Optional, Tuple

import numpy as np

# statuses.py:16-27 -- Enum auto() values
VISCEK, FOLLOWER, EXITING, ESCAPED = 1, 2, 3, 4
# constants.py:35-38 via distances.py:17-21
R_LEADER = 0.2
R_PEDESTRIAN = 0.1
R_EXIT = 0.4
R_ESCAPE = 0.01
# area.py:39
EXIT_POSITION = np.array([0, -1], dtype=np.float32)
AGENT_WALL_PENALTY = -5.0  # area.py:198


@dataclasses.dataclass
class OracleParams:
    """The subset of EnvConfig (config.py:11-59) that enters the arithmetic."""
    number_of_pedestrians: int = 10
    width: float = 1.0
    height: float = 1.0
    step_size: float = 0.01
    noise_coef: float = 0.2
    eps: float = 1e-8
    enslaving_degree: float = 1.0
    is_new_exiting_reward: bool = False
    is_new_followers_reward: bool = True
    intrinsic_reward_coef: float = 0.0
    is_termination_agent_wall_collision: bool = False
    init_reward_each_step: float = -1.0
    max_timesteps: int = 2000


@dataclasses.dataclass
class OracleState:
    pos: np.ndarray          # [N,2]  pedestrians.py:17
    dir: np.ndarray          # [N,2]  pedestrians.py:18-19
    status: np.ndarray       # [N] int8 codes
    agent_pos: np.ndarray    # [2] f32  area.py:28
    agent_dir: np.ndarray    # [2] f32  area.py:29 / area.py:192
    now: int = 0             # area.py:44

    def copy(self) -> "OracleState":
        return OracleState(self.pos.copy(), self.dir.copy(), self.status.copy(),
                           self.agent_pos.copy(), self.agent_dir.copy(), int(self.now))


def _ped_dtype(precision: str):
    if precision == "ref":
        return np.float64
    if precision == "f32":
        return np.float32
    raise ValueError(precision)


# ------------------------------------------------------------------------------------------
# distances (scipy.spatial.distance_matrix restated)
# ------------------------------------------------------------------------------------------
def pairwise_distance(x: np.ndarray, y: np.ndarray, dtype) -> np.ndarray:
    """scipy.spatial.distance_matrix(x, y, 2): scipy 1.15.3 ``_kdtree.py:48-60,869-921`` --
    ``minkowski_distance(x[:,None,:], y[None,:,:], 2)`` = ``sum(|y-x|**2, axis=-1) ** (1/2)``
    after promoting both inputs to a common float type (f64 in the reference because the
    pedestrians are f64).  ``dtype`` makes the promotion explicit."""
    x = np.asarray(x, dtype=dtype)
    y = np.asarray(y, dtype=dtype)
    d = np.abs(y[None, :, :] - x[:, None, :])
    return np.sqrt(np.sum(d * d, axis=-1))


def is_distance_low(pos: np.ndarray, dest: np.ndarray, radius: float, dtype) -> np.ndarray:
    """distances.py:24-48: ``distance_matrix(pos, dest[None]) < radius`` squeezed to [N]."""
    return pairwise_distance(pos, np.asarray(dest)[None, :], dtype)[:, 0] < radius


def mean_distance(pos: np.ndarray, dest: np.ndarray, dtype) -> float:
    """distances.py:51-56 (``sum_distance``, despite its name a mean over all N)."""
    d = pairwise_distance(pos, np.asarray(dest)[None, :], dtype)
    return d.sum() / pos.shape[0]


# ------------------------------------------------------------------------------------------
# statuses
# ------------------------------------------------------------------------------------------
def classify_statuses(pos: np.ndarray, agent_pos: np.ndarray, exit_pos: np.ndarray, dtype) -> np.ndarray:
    """statuses.py:29-48.  Every element is overwritten, so the result is a pure function of
    the positions: FOLLOWER (<0.2 of leader), overridden by EXITING (<0.4 of exit), overridden
    by ESCAPED (<0.01 of exit), else VISCEK."""
    following = is_distance_low(pos, agent_pos, R_LEADER, dtype)
    exiting = is_distance_low(pos, exit_pos, R_EXIT, dtype)
    escaped = is_distance_low(pos, exit_pos, R_ESCAPE, dtype)
    st = np.full(pos.shape[0], VISCEK, dtype=np.int8)
    st[following] = FOLLOWER
    st[exiting] = EXITING
    st[escaped] = ESCAPED
    return st


# ------------------------------------------------------------------------------------------
# leader
# ------------------------------------------------------------------------------------------
def agent_step(p: OracleParams, st: OracleState, action) -> Tuple[bool, float]:
    """area.py:182-210.  The action is normalised to (almost) unit length, the leader's
    direction is always overwritten, the move is rejected if the target is strictly outside
    the walls.  Arithmetic stays in the action's dtype (f32 for Box(-1,1,f32) actions:
    NumPy-2 weak scalars keep ``eps``/``step_size`` from promoting it)."""
    a = np.array(action)
    if not np.issubdtype(a.dtype, np.floating):
        raise TypeError("action must be a float array (area.py:190 divides in place)")
    a = a / (np.linalg.norm(a) + p.eps)                    # area.py:190
    st.agent_dir = p.step_size * a                          # area.py:192
    pt = st.agent_pos + st.agent_dir                        # area.py:201
    hit = bool(pt[0] < -p.width or pt[0] > p.width or pt[1] < -p.height or pt[1] > p.height)
    if not hit:
        st.agent_pos = st.agent_pos + st.agent_dir          # area.py:195
        return False, 0.0
    return bool(p.is_termination_agent_wall_collision), AGENT_WALL_PENALTY   # area.py:198


# ------------------------------------------------------------------------------------------
# rewards
# ------------------------------------------------------------------------------------------
def status_reward(p: OracleParams, old: np.ndarray, new: np.ndarray, now: int) -> float:
    """reward.py:23-47 (the code, not its docstring, is normative)."""
    r = p.init_reward_each_step
    tf = 1 - now / (200 * p.number_of_pedestrians)
    if p.is_new_exiting_reward:
        n = int(np.sum(((old == VISCEK) | (old == FOLLOWER)) & (new == EXITING)))
        r += (15 + 10 * tf) * n
    if p.is_new_followers_reward:
        n = int(np.sum((old == VISCEK) & (new == FOLLOWER)))
        r += (10 + 5 * tf) * n
    return r


# ------------------------------------------------------------------------------------------
# pedestrians
# ------------------------------------------------------------------------------------------
def pedestrians_step(p: OracleParams, st: OracleState, noise: np.ndarray, precision: str = "ref"
                     ) -> Tuple[bool, float, float]:
    """area.py:76-180.  ``noise`` is [N], indexed by pedestrian; only the entries of
    FOLLOWER/VISCEK pedestrians are used (area.py:124 draws exactly ``n_fv`` values).
    Mutates ``st.pos/dir/status``; returns (terminated, reward_pedestrians, intrinsic)."""
    dt = _ped_dtype(precision)
    pos, dr, s = st.pos, st.dir, st.status
    n = pos.shape[0]
    exit_pos = EXIT_POSITION

    escaped = s == ESCAPED                                   # area.py:79-81
    dr[escaped] = 0
    pos[escaped] = exit_pos

    exiting = s == EXITING                                   # area.py:84-90
    if exiting.any():
        v = exit_pos - pos[exiting]
        ln = np.linalg.norm(v, axis=1)
        sz = np.minimum(ln, p.step_size)
        dr[exiting] = (v.T / ln * sz).T

    following = s == FOLLOWER                                # area.py:93
    viscek = s == VISCEK                                     # area.py:96
    efv = exiting | following | viscek                       # area.py:99
    fv = following | viscek                                  # area.py:104

    with np.errstate(invalid="ignore", divide="ignore"):
        u = (dr.T / np.linalg.norm(dr, axis=1)).T            # area.py:100-101 (rows of efv only)
    dm = pairwise_distance(pos, pos, dt)                     # area.py:105-106
    inter = np.where(dm < R_PEDESTRIAN, 1, 0)                # area.py:107
    inter = inter * efv[None, :]                             # columns restricted to efv
    cnt = np.maximum(1, inter.sum(axis=1))                   # area.py:108
    # area.py:118-119: (intersection * u).sum(axis=1)/n.  NaN*0 = NaN: an efv pedestrian with a
    # zero direction (0/0 above) poisons every fv row, exactly as in the reference.
    with np.errstate(invalid="ignore"):
        wx = np.where(efv[None, :], inter * u[:, 0][None, :], 0)
        wy = np.where(efv[None, :], inter * u[:, 1][None, :], 0)
        mx = wx.sum(axis=1) / cnt
        my = wy.sum(axis=1) / cnt
        theta = np.arctan2(my, mx)                           # area.py:120
        theta = theta + np.asarray(noise, dtype=dt)          # area.py:124-127
        new_dir = np.stack((np.cos(theta), np.sin(theta)), axis=1) * p.step_size   # area.py:129,136
    dr[fv] = new_dir[fv]

    e = p.enslaving_degree                                   # area.py:139-142
    dr[following] = e * st.agent_dir + (1.0 - e) * dr[following]

    pos[efv] += dr[efv]                                      # area.py:145

    lo = np.array([-p.width, -p.height], dtype=dt)           # area.py:148-152
    hi = np.array([p.width, p.height], dtype=dt)
    clipped = np.clip(pos, lo, hi)
    miss = pos - clipped
    pos -= 2 * miss
    dr *= np.where(miss != 0, -1, 1)

    old = s.copy()                                           # area.py:155-161
    new = classify_statuses(pos, st.agent_pos, exit_pos, dt)
    r_ped = status_reward(p, old, new, st.now)               # area.py:162-167
    intrinsic = 0 - mean_distance(pos, exit_pos, dt)         # area.py:168-171, reward.py:19-21
    st.status = new                                          # area.py:172
    terminated = bool(np.sum(new == ESCAPED) == n)           # area.py:175-178
    return terminated, r_ped, intrinsic


def fv_mask(st: OracleState) -> np.ndarray:
    """Pedestrians that consume a noise draw this step (area.py:104,124)."""
    return (st.status == FOLLOWER) | (st.status == VISCEK)


def draw_step_noise(p: OracleParams, st: OracleState, rng=np.random) -> np.ndarray:
    """Draw the step's noise exactly as the reference would (area.py:124: ONE global-RNG call
    of size n_fv) and scatter it to pedestrian index."""
    m = fv_mask(st)
    out = np.zeros(st.pos.shape[0], dtype=np.float64)
    out[m] = rng.uniform(low=-p.noise_coef / 2, high=p.noise_coef / 2, size=int(m.sum()))
    return out


# ------------------------------------------------------------------------------------------
# env orchestration
# ------------------------------------------------------------------------------------------
def env_reset(p: OracleParams, draw_pos: np.ndarray, draw_dir: np.ndarray, precision: str = "ref"
              ) -> OracleState:
    """env.py:129-137 + pedestrians.py:16-27 + area.py:27-30.  ``draw_pos``/``draw_dir`` are
    the two U(-1,1) [N,2] draws (pedestrians.py:17-18; note the hard-coded +-1)."""
    dt = _ped_dtype(precision)
    pos = np.array(draw_pos, dtype=dt)
    d = np.array(draw_dir, dtype=dt)
    d = (d.T / np.linalg.norm(d, axis=1)).T                  # pedestrians.py:29-31
    agent_pos = np.zeros(2, dtype=np.float32)                # area.py:22,28
    agent_dir = np.zeros(2, dtype=np.float32)                # area.py:29 (copies start_position)
    status = classify_statuses(pos, agent_pos, EXIT_POSITION, dt)
    return OracleState(pos, d, status, agent_pos, agent_dir, 0)


def draw_reset(n: int, rng=np.random) -> Tuple[np.ndarray, np.ndarray]:
    """pedestrians.py:17-18: two global-RNG draws of shape (N,2), positions first."""
    a = rng.uniform(-1.0, 1.0, size=(n, 2))
    b = rng.uniform(-1.0, 1.0, size=(n, 2))
    return a, b


def env_step(p: OracleParams, st: OracleState, action, noise: np.ndarray, precision: str = "ref"
             ) -> Dict[str, object]:
    """env.py:141-171.  Returns the pieces the reference combines into its step tuple."""
    st.now += 1                                              # area.py:53-56
    truncated = st.now >= p.max_timesteps                    # area.py:58-59
    term_agent, r_agent = agent_step(p, st, action)          # env.py:146
    term_ped, r_ped, intrinsic = pedestrians_step(p, st, noise, precision)   # env.py:149-150
    reward = r_agent + r_ped + p.intrinsic_reward_coef * intrinsic           # env.py:158
    return dict(reward=reward, reward_agent=r_agent, reward_ped=r_ped, intrinsic=intrinsic,
                terminated=bool(term_agent or term_ped), truncated=bool(truncated))


class EpisodeLog:
    """The per-episode bookkeeping of EvacuationEnv: the three sums its step() keeps (env.py:65-67, 168-170) and the nine-key dict its
    NEXT reset() logs for the episode that ended (env.py:114-125).  ``overall_timesteps`` is Time.overall_timesteps (area.py:47,55):
    steps of this env since it was created."""

    KEYS = ("episode_intrinsic_reward", "episode_status_reward", "episode_reward", "episode_length", "escaped_pedestrians",
            "exiting_pedestrians", "following_pedestrians", "viscek_pedestrians", "overall_timesteps")

    def __init__(self, overall_timesteps: int = 0):
        self.episode_reward = 0.0             # env.py:129-131 (zeroed by every reset)
        self.episode_intrinsic_reward = 0.0
        self.episode_status_reward = 0.0
        self.overall_timesteps = int(overall_timesteps)

    def after_step(self, out: Dict[str, object]) -> None:
        """env.py:168-170, with the dict env_step returns; Time.step counts the step (area.py:55)."""
        self.episode_reward += out["reward"]
        self.episode_intrinsic_reward += out["intrinsic"]
        self.episode_status_reward += out["reward_agent"] + out["reward_ped"]
        self.overall_timesteps += 1

    def record(self, st: OracleState) -> Dict[str, float]:
        """env.py:114-125: built from the final state of the episode, before the reset that logs it touches anything."""
        return {"episode_intrinsic_reward": self.episode_intrinsic_reward, "episode_status_reward": self.episode_status_reward,
                "episode_reward": self.episode_reward, "episode_length": st.now,
                "escaped_pedestrians": int(np.sum(st.status == ESCAPED)), "exiting_pedestrians": int(np.sum(st.status == EXITING)),
                "following_pedestrians": int(np.sum(st.status == FOLLOWER)), "viscek_pedestrians": int(np.sum(st.status == VISCEK)),
                "overall_timesteps": self.overall_timesteps}


# ------------------------------------------------------------------------------------------
# observations
# ------------------------------------------------------------------------------------------
def obs_abs(st: OracleState) -> Dict[str, np.ndarray]:
    """env.py:98-104."""
    return {"agent_position": st.agent_pos, "pedestrians_positions": st.pos,
            "exit_position": EXIT_POSITION}


def obs_relative(obs: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """wrappers.py:8-27.  The 'hypotenuse' of Box(-1,1) bounds is sqrt(1+1) in f32."""
    hyp = np.sqrt(np.float32(1.0) + np.float32(1.0))
    out = dict(obs)
    out["pedestrians_positions"] = (obs["pedestrians_positions"] - obs["agent_position"]) / hyp
    out["exit_position"] = (obs["exit_position"] - obs["agent_position"]) / hyp
    return out


def obs_statuses(status: np.ndarray, kind: str) -> Optional[np.ndarray]:
    """wrappers.py:47-57: code = 4 - Status.value (ESCAPED->0 ... VISCEK->3)."""
    code = 4 - status.astype(np.int64)
    if kind == "ohe":
        out = np.zeros((status.shape[0], 4))
        out[np.arange(status.shape[0]), code] = 1
        return out
    if kind == "cat":
        return code / 4
    return None


def obs_matrix(obs: Dict[str, np.ndarray], status: np.ndarray, kind: str) -> np.ndarray:
    """wrappers.py:77-96: rows = [agent; exit; pedestrians...]."""
    pos = np.vstack((obs["agent_position"], obs["exit_position"], obs["pedestrians_positions"]))
    if kind == "ohe":
        stat = np.vstack((np.array([0, 0, 0, 0], dtype=np.float32),
                          np.array([1, 0, 0, 0], dtype=np.float32),
                          obs_statuses(status, "ohe")))
        return np.hstack((pos, stat)).astype(np.float32)
    if kind == "cat":
        stat = np.hstack(([0, 1], obs_statuses(status, "cat")))
        return np.hstack((pos, stat[:, None])).astype(np.float32)
    if kind == "no":
        return pos
    raise ValueError(kind)


def grad_potential_pedestrians(agent_pos, pos, status, alpha, eps) -> np.ndarray:
    """gravity_encoding.py:8-25 (sum over VISCEK pedestrians; zeros(2) if there are none)."""
    m = status == VISCEK
    r = agent_pos[None, :] - pos[m, :]
    if len(r) != 0:
        norm = np.linalg.norm(r, axis=1)[:, None] + eps
        return (-alpha / norm ** (alpha + 2) * r).sum(axis=0)
    return np.zeros(2)


def grad_potential_exit(agent_pos, n_followers: int, exit_pos, alpha, eps) -> np.ndarray:
    """gravity_encoding.py:28-38.  agent and exit are both f32, so this is f32 arithmetic; the
    final product with the (int64) follower count promotes the stored value to f64."""
    r = agent_pos - exit_pos
    norm = np.linalg.norm(r) + eps
    g = -alpha / norm ** (alpha + 2) * r
    return g * np.int64(n_followers)


def obs_gravity(st: OracleState, alpha, eps) -> Dict[str, np.ndarray]:
    """gravity_encoding.py:59-81 (uses the post-step statuses)."""
    nf = int(np.sum(st.status == FOLLOWER))
    return {
        "agent_position": st.agent_pos,
        "grad_potential_pedestrians": grad_potential_pedestrians(st.agent_pos, st.pos, st.status, alpha, eps),
        "grad_potential_exit": grad_potential_exit(st.agent_pos, nf, EXIT_POSITION, alpha, eps),
    }


def observe(st: OracleState, positions: str = "abs", statuses: str = "no", type_: str = "Dict",
            alpha: float = 3, eps: float = 1e-8):
    """wrappers/config.py:46-93 dispatch table."""
    if positions == "grav":
        if type_ == "Dict":
            return obs_gravity(st, alpha, eps)
        if type_ == "Box":
            raise NotImplementedError
        raise ValueError(type_)
    obs = obs_abs(st)
    if positions == "rel":
        obs = obs_relative(obs)
    if type_ == "Box":
        return obs_matrix(obs, st.status, statuses)
    if statuses != "no":
        obs = dict(obs)
        obs["pedestrians_statuses"] = obs_statuses(st.status, statuses)
    return obs


# ------------------------------------------------------------------------------------------
# tie detector for f32-vs-f64 comparisons
# ------------------------------------------------------------------------------------------
def threshold_margin(pos: np.ndarray, agent_pos: np.ndarray, pre_pos: Optional[np.ndarray] = None,
                     width: float = 1.0, height: float = 1.0, status: Optional[np.ndarray] = None) -> float:
    """Smallest |distance - radius| over every comparison a step takes (neighbour test on the
    pre-step positions, status tests and wall test on the post-step positions).  An f32 step
    may legitimately flip a comparison whose margin is below ~1e-6; such envs are excluded
    from element-wise f32 parity (SURVEY.md section 7 'Parity definition')."""
    m = np.inf
    p64 = np.asarray(pos, dtype=np.float64)
    for dest, rad in ((agent_pos, R_LEADER), (EXIT_POSITION, R_EXIT), (EXIT_POSITION, R_ESCAPE)):
        d = pairwise_distance(p64, np.asarray(dest, dtype=np.float64)[None, :], np.float64)[:, 0]
        m = min(m, float(np.min(np.abs(d - rad))))
    # wall test: a reflected pedestrian sits as far inside the wall as it overshot; pedestrians
    # pinned on the exit (0,-1) sit ON the wall by construction and are not a tie
    free = np.ones(p64.shape[0], bool) if status is None else (np.asarray(status) != ESCAPED)
    if free.any():
        m = min(m, float(np.min(np.abs(np.abs(p64[free, 0]) - width))),
                float(np.min(np.abs(np.abs(p64[free, 1]) - height))))
    if pre_pos is not None:
        q = np.asarray(pre_pos, dtype=np.float64)
        dm = pairwise_distance(q, q, np.float64)
        iu = np.triu_indices(q.shape[0], 1)
        if len(iu[0]):
            m = min(m, float(np.min(np.abs(dm[iu] - R_PEDESTRIAN))))
    return m


def threshold_margins_per_pedestrian(pos: np.ndarray, agent_pos: np.ndarray, pre_pos: np.ndarray, width: float = 1.0,
                                     height: float = 1.0, status: Optional[np.ndarray] = None) -> np.ndarray:
    """`threshold_margin` resolved per pedestrian: the smallest |distance - radius| over the comparisons that involve pedestrian
    i (its status and wall tests on the post-step position, its pairs in the neighbour test on the pre-step positions).  In a
    teacher-forced single step a comparison of pedestrian i can only change i's own row, position and status -- and, through
    the status counts, the env-level outputs (rewards, gravity observation)."""
    p64 = np.asarray(pos, dtype=np.float64)
    n = p64.shape[0]
    m = np.full(n, np.inf)
    for dest, rad in ((agent_pos, R_LEADER), (EXIT_POSITION, R_EXIT), (EXIT_POSITION, R_ESCAPE)):
        d = pairwise_distance(p64, np.asarray(dest, dtype=np.float64)[None, :], np.float64)[:, 0]
        m = np.minimum(m, np.abs(d - rad))
    free = np.ones(n, bool) if status is None else (np.asarray(status) != ESCAPED)
    wall = np.minimum(np.abs(np.abs(p64[:, 0]) - width), np.abs(np.abs(p64[:, 1]) - height))
    m = np.where(free, np.minimum(m, wall), m)
    q = np.asarray(pre_pos, dtype=np.float64)
    dm = np.abs(pairwise_distance(q, q, np.float64) - R_PEDESTRIAN)
    np.fill_diagonal(dm, np.inf)
    return np.minimum(m, dm.min(axis=1)) if n > 1 else m

