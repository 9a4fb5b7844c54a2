"""Hunt for an intermittent mismatch of chained launches: the parity test's pattern (R launches in flight, a join, repeated) with every
launch of every repetition compared; prints the first mismatch of a run.  GPU box: python tools/chain_flaky.py N E T runs [box|grav]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea
n, E, T, runs = (int(x) for x in sys.argv[1:5])
kind = sys.argv[5] if len(sys.argv) > 5 else "box"
foreign = int(sys.argv[6]) if len(sys.argv) > 6 else 1          # 1: the reference launches run concurrently with the chain (as in the test)
wrap = ea.EnvWrappersConfig(positions="rel", statuses="ohe", type="Box") if kind == "box" else ea.EnvWrappersConfig(positions="grav", alpha=3)
cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=45, is_new_exiting_reward=True, is_new_followers_reward=True)
bad_runs = 0
for run in range(runs):
    one = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=11, options=ea.KernelOptions(cu_wide=1))
    ch = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=11, options=ea.KernelOptions(cu_wide=1, chain=1))
    one.reset(); ch.reset()
    R = 12
    outs = [{"slab": torch.empty((T, E, one.obs_dim + 3), device=ch.device), "episode_stats": torch.zeros((T, E, ch.stats_words), device=ch.device)} for _ in range(R)]
    goes = [ch.rollout_launcher(T, o) for o in outs]
    first = None
    for rep in range(6):
        refs = [one.rollout(T) for _ in range(R)]
        if not foreign:
            torch.cuda.synchronize()
        for g in goes:
            g()
        ch.join(); torch.cuda.synchronize()
        for j in range(R):
            d = outs[j]["slab"] != refs[j]["slab"]
            if d.any() and first is None:
                idx = d.nonzero()[0].tolist()
                envs = d.any(dim=2).any(dim=0).nonzero().flatten().tolist()
                first = f"rep {rep} launch {j} (chain launch #{rep * R + j + 1}): {int(d.sum())} words, envs {envs[:12]}{'...' if len(envs) > 12 else ''} ({len(envs)}), first at t={idx[0]} env={idx[1]} col={idx[2]}; error word {ch.team_error(sync=False)}"
        if first:
            break
    if first:
        bad_runs += 1
        print(f"run {run}: MISMATCH {first}", flush=True)
    one.close(); ch.close()
print(f"{bad_runs} of {runs} runs mismatched (N={n} E={E} T={T} {kind} foreign={foreign})")
