import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import evacuation_amd as ea
import helpers as H
import test_gpu_parity as T
from oracle import evac_oracle as O
d = np.load(H.traj_files()[[os.path.basename(p) for p in H.traj_files()].index("traj_n60_s0.npz")])
p = H.load_params(d["params_json"])
K = len(d["action"])
pre = [H.state_at(d, k) for k in range(K)]
wrap = ea.EnvWrappersConfig(**T.WRAPS[0])
got = T.gpu_step_batch(ea, p, wrap, pre, d["action"], d["noise"])
bad = 0
for e in range(K):
    st = T.f32_state(pre[e])
    out = O.env_step(p, st, np.asarray(d["action"][e], dtype=np.float32), np.asarray(d["noise"][e], dtype=np.float32).astype(np.float64))
    ds = (got["status"][e] != st.status).sum()
    dp = np.abs(got["pos"][e] - st.pos).max()
    da = np.abs(got["agent_pos"][e] - st.agent_pos).max()
    if ds or dp > 1e-5 or da > 1e-6:
        bad += 1
        if bad < 6:
            print("env", e, "status diffs", ds, "pos err", dp, "agent err", da, "agent gpu", got["agent_pos"][e], "oracle", st.agent_pos, "pre", pre[e].agent_pos, "act", d["action"][e])
print(os.environ.get("EVAC_LIB"), "bad envs", bad, "of", K)
