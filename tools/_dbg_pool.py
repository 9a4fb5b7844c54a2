import os, sys, torch
sys.path.insert(0, "/root/repo")
import evacuation_amd as ea
def make(cfg, wrap, E, cu_wide):
    os.environ["EVAC_CU_WIDE"] = "1" if cu_wide else "0"; os.environ["EVAC_WORKSPACE"] = "0"
    try: return ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=7)
    finally: os.environ.pop("EVAC_CU_WIDE"); os.environ.pop("EVAC_WORKSPACE")
for n, E in ((200, 53), (200, 52), (256, 40), (256, 4), (129, 8), (192, 8)):
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=70, is_new_exiting_reward=True, intrinsic_reward_coef=0.5)
    wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
    a, b = make(cfg, wrap, E, False), make(cfg, wrap, E, True)
    a.reset(); b.reset()
    ra, rb = a.rollout(30), b.rollout(30)
    torch.cuda.synchronize()
    d = (ra["obs"].view(torch.int32) != rb["obs"].view(torch.int32)).any(dim=2)   # [T, E]
    bad = d.nonzero()
    print(n, E, b.kernel_variant("rollout"), "differing (t, env):", bad[:12].tolist(), "count", int(d.sum()))
