"""Hunt for an intermittent mismatch of chained launches: the parity test's pattern (R launches in flight, a join, repeated) with every
launch of every repetition compared; prints the first mismatch of a run.
GPU box: python tools/chain_flaky.py N E T runs [box|grav] [foreign]      one shape, `runs` fresh pairs of handles
         python tools/chain_flaky.py mix runs [poison]                      the parity test's shapes in turn; poison = 1: memory the caching
                                                                            allocator recycles is filled with garbage between the runs"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea

BOX = dict(positions="rel", statuses="ohe", type="Box")
GRAV = dict(positions="grav", alpha=3)
MIX = [(60, 4096, GRAV, 20), (60, 512, GRAV, 7), (33, 64, BOX, 10), (64, 160, dict(positions="abs", statuses="cat", type="Dict"), 5),
       (256, 1024, GRAV, 20), (200, 52, BOX, 6)]


def poison(dev, seed):
    """garbage into whatever the allocator hands out next: blocks of the sizes the handles use, freed again at once"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    keep = []
    for nbytes in [1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24]:
        for _ in range(4):
            k = int(torch.randint(nbytes // 2, nbytes, (1,), generator=g))
            keep.append(torch.full((k // 4,), -7 - seed, dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    del keep


SYNC = os.environ.get("FLAKY_SYNC", "device")      # "stream": wait as the parity test does (join, then the current stream only)


def one_run(n, E, wrap_kw, T, foreign=1, reps=6):
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=45, is_new_exiting_reward=True, is_new_followers_reward=True)
    wrap = ea.EnvWrappersConfig(**wrap_kw)
    one = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=11, options=ea.KernelOptions(cu_wide=1))
    ch = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=11, options=ea.KernelOptions(cu_wide=1, chain=int(os.environ.get("FLAKY_CHAIN", "1"))))
    one.reset(); ch.reset()
    R = 12
    outs = [{"slab": torch.empty((T, E, one.obs_dim + 3), device=ch.device), "episode_stats": torch.zeros((T, E, ch.stats_words), device=ch.device)} for _ in range(R)]
    goes = [ch.rollout_launcher(T, o) for o in outs]
    first = None
    for rep in range(reps):
        refs = [one.rollout(T) for _ in range(R)]
        if not foreign:
            torch.cuda.synchronize()
        for g in goes:
            g()
        ch.join()
        if SYNC == "stream":
            torch.cuda.current_stream().synchronize()
        else:
            torch.cuda.synchronize()
        for j in range(R):
            d = outs[j]["slab"] != refs[j]["slab"]
            if d.any() and first is None:
                idx = d.nonzero()[0].tolist()
                envs = d.any(dim=2).any(dim=0).nonzero().flatten().tolist()
                steps = d.any(dim=2).any(dim=1).nonzero().flatten().tolist()
                cols = d.any(dim=0).any(dim=0).nonzero().flatten().tolist()
                first = (f"rep {rep} launch {j} (chain launch #{rep * R + j + 1}): {int(d.sum())} words, envs {envs[:12]}{'...' if len(envs) > 12 else ''} ({len(envs)}), "
                         f"steps {steps[:8]}, columns {cols[:6]}..{cols[-1]} ({len(cols)}), first at t={idx[0]} env={idx[1]} col={idx[2]}: "
                         f"{float(outs[j]['slab'][tuple(idx)])!r} vs {float(refs[j]['slab'][tuple(idx)])!r}; error word {ch.team_error(sync=False)}")
                import time
                time.sleep(0.3); torch.cuda.synchronize()
                first += f"; equal after 0.3 s and a device-wide wait: {bool(torch.equal(outs[j]['slab'], refs[j]['slab']))}"
        if first:
            break
    one.close(); ch.close()
    return first


if sys.argv[1] == "mix":
    runs, pois = int(sys.argv[2]), (int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    bad = 0
    for run in range(runs):
        for (n, E, kw, T) in MIX:
            if pois:
                poison("cuda:0", run)
            first = one_run(n, E, kw, T, reps=3)
            if first:
                bad += 1
                print(f"run {run} N={n} E={E} T={T}: MISMATCH {first}", flush=True)
        if run % 10 == 9:
            print(f"... {run + 1} runs, {bad} mismatches", flush=True)
    print(f"{bad} mismatches in {runs} x {len(MIX)} runs (poison={pois})")
else:
    n, E, T, runs = (int(x) for x in sys.argv[1:5])
    kind = sys.argv[5] if len(sys.argv) > 5 else "box"
    foreign = int(sys.argv[6]) if len(sys.argv) > 6 else 1          # 1: the reference launches run concurrently with the chain (as in the test)
    bad_runs = 0
    for run in range(runs):
        first = one_run(n, E, BOX if kind == "box" else GRAV, T, foreign)
        if first:
            bad_runs += 1
            print(f"run {run}: MISMATCH {first}", flush=True)
    print(f"{bad_runs} of {runs} runs mismatched (N={n} E={E} T={T} {kind} foreign={foreign})")
