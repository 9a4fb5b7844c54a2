#!/usr/bin/env python3
"""The reference trainer's rollout loop (src/agents/rpo_agent.py:180-203) with the env on the GPU.

A small MLP actor (the shape of RPOLinearNetwork, rpo_linear_agent_network.py:19-61: obs -> 64 -> 64 -> 2)
drives 4096 evacuation envs through ``NormalizedVectorEnv`` (= the trainer's wrapper chain on device).
Observations, actions, rewards and done flags never leave HBM; the whole `policy -> step -> store` body
is captured once into a hipGraph and replayed.  Prints steps per second ("SPS" of rpo_agent.py:298-299).

    python examples/rollout_with_policy.py [--envs 4096] [--steps 256]
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
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True)
    envs = ea.NormalizedVectorEnv.make(cfg, ea.EnvWrappersConfig(positions="grav", alpha=3), num_envs=args.envs, gamma=0.99)
    D = envs.obs_dim
    torch.manual_seed(0)
    actor = torch.nn.Sequential(torch.nn.Linear(D, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                                torch.nn.Linear(64, 2)).to(dev)
    T, E = args.steps, args.envs
    obs_buf = torch.zeros((T, E, D), device=dev)
    act_buf = torch.zeros((T, E, 2), device=dev)
    rew_buf = torch.zeros((T, E), device=dev)
    done_buf = torch.zeros((T, E), device=dev)
    next_obs, _ = envs.reset()
    action = torch.zeros((E, 2), device=dev)
    slot = torch.zeros((), dtype=torch.long, device=dev)

    def body():
        with torch.no_grad():
            mean = actor(next_obs)
            action.copy_(mean + 0.5 * torch.randn_like(mean))
            obs_buf.index_copy_(0, slot.view(1), next_obs.unsqueeze(0))
            act_buf.index_copy_(0, slot.view(1), action.unsqueeze(0))
            obs, rew, term, trunc, _ = envs.step(action)          # same tensors every call: graph-safe
            rew_buf.index_copy_(0, slot.view(1), rew.unsqueeze(0))
            done_buf.index_copy_(0, slot.view(1), (term | trunc).float().unsqueeze(0))
            slot.add_(1).remainder_(T)

    for _ in range(3):
        body()
    torch.cuda.synchronize()
    if args.no_graph:
        run = body
    else:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                body()
        torch.cuda.current_stream().wait_stream(side)
        run = g.replay
    slot.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(T):
        run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"envs={E} steps={T} graph={not args.no_graph}: {E * T / dt:.3e} env-steps/s  ({dt / T * 1e6:.1f} us per vector step), "
          f"mean normalised reward {rew_buf.mean().item():.3f}, done fraction {done_buf.mean().item():.4f}")
    envs.close()


if __name__ == "__main__":
    main()
