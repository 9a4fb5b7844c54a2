#!/usr/bin/env python3
"""The reference trainer's rollout loop (src/agents/rpo_agent.py:180-203) with the env on the GPU.

A small MLP actor (the shape of RPOLinearNetwork, rpo_linear_agent_network.py:19-61: obs -> 64 -> 64 -> 2)
drives 4096 evacuation envs through ``NormalizedVectorEnv`` (= the trainer's wrapper chain on device).
The trainer's storage (rpo_agent.py:158-163: ``obs[step]``, ``actions[step]``, ``rewards[step]``, ``dones[step]``) is
written by the step kernel itself: every ``envs.step`` gets row ``t`` of the buffers as its destination
(``out_obs=obs[t + 1]``, ``out_reward=rewards[t]``, ...), so there is no per-step copy into the buffers and nothing
leaves HBM.  The whole T-step `policy -> step` rollout is captured once into a hipGraph and replayed.
Prints steps per second ("SPS" of rpo_agent.py:298-299).

    python examples/rollout_with_policy.py [--envs 4096] [--steps 128]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="step and normalisation chain as two launches (cross-check)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True)
    envs = ea.NormalizedVectorEnv.make(cfg, ea.EnvWrappersConfig(positions="grav", alpha=3), num_envs=args.envs, gamma=0.99)
    D = envs.obs_dim
    torch.manual_seed(0)
    actor = torch.nn.Sequential(torch.nn.Linear(D, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                                torch.nn.Linear(64, 2)).to(dev)
    T, E = args.steps, args.envs
    obs = torch.zeros((T + 1, E, D), device=dev)            # obs[t] is what the policy sees at step t
    actions = torch.zeros((T, E, 2), device=dev)
    rewards = torch.zeros((T, E), device=dev)
    terminated = torch.zeros((T, E), dtype=torch.uint8, device=dev)
    truncated = torch.zeros((T, E), dtype=torch.uint8, device=dev)
    first, _ = envs.reset()
    obs[0].copy_(first)

    def rollout():
        with torch.no_grad():
            for t in range(T):
                mean = actor(obs[t])
                torch.add(mean, torch.randn_like(mean), alpha=0.5, out=actions[t])
                envs.step(actions[t], out_obs=obs[t + 1], out_reward=rewards[t], out_terminated=terminated[t],
                          out_truncated=truncated[t], fused=not args.unfused)
            obs[0].copy_(obs[T])                             # the next rollout continues from here

    rollout()
    torch.cuda.synchronize()
    if args.no_graph:
        run = rollout
    else:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                rollout()
        torch.cuda.current_stream().wait_stream(side)
        run = g.replay
    torch.cuda.synchronize()
    reps = 4
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    done = (terminated | truncated).float()
    print(f"envs={E} steps={T} graph={not args.no_graph}: {E * T / dt:.3e} env-steps/s  ({dt / T * 1e6:.1f} us per vector step), "
          f"mean normalised reward {rewards.mean().item():.3f}, done fraction {done.mean().item():.4f}")
    envs.close()


if __name__ == "__main__":
    main()
