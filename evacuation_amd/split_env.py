"""SplitBatchEnv: ONE batch of envs as several independent handles whose rollout launches go to streams of their own.

The envs of a batch do not depend on each other; only the launches of one HANDLE do (launch j + 1 continues the state launch j
left).  A rollout launch lasts as long as the heaviest env it carries, and between two launches of a stream lie the queue's
kernel boundary (~4 us) and the kernel's prologue (~3 us): with the driver's 20 steps per launch that is a fifth of the time
(profiles/r05_i_c2_launch_edges_prologue_and_boundary.txt).  Cut into P parts on P streams, each part waits only
for ITS heaviest workgroup and its boundary and prologue run under the other parts' steps: N = 60 x 4096 envs as two halves
1.92-1.96e9 env-steps/s against 1.79-1.85e9 as one handle (profiles/r05_k_one_batch_as_two_halves_on_two_streams_probe.txt).

Part k owns the envs [k E / P, (k + 1) E / P) of the batch and is created with that ``env_id_offset``: the on-device random streams
(reset draws, RandomAgent actions, pedestrian noise) are keyed by the GLOBAL env id, so the parts reproduce the single handle's
trajectories bit for bit (tests/test_gpu_parity.py::test_split_batch_equals_one_handle) -- the same property
ShardedEvacuationEnv uses across GPUs.  The price: the outputs are P slabs ``[T, E / P, D + 3]`` (``rollout()`` returns them and,
on request, their concatenation), and whoever consumes them must wait for all P streams (``join`` / ``synchronize``).

Since round 6 the same thing exists INSIDE the library for two parts -- ``KernelOptions(parts=2)`` / ``evac_options_t.parts``: one
handle, one slab ``[T, E, D + 3]``, two streams the handle owns (``BatchedEvacuationEnv.join``); that form is what bench.py times.
This class stays for P separate slabs and for experiments with other part counts."""
from __future__ import annotations

from typing import List, Optional

import torch

from .vector_env import BatchedEvacuationEnv


class SplitBatchEnv:
    """See the module docstring.  ``parts=2`` is the useful value on an MI355X: +3.4 % at 20 steps per launch, +3.9 % at 100; four or
    eight parts lose half and more (HIP has four hardware queues: parts that share one take turns, and 64- / 32-workgroup launches
    leave CUs idle -- profiles/r05_k_one_batch_as_two_halves_on_two_streams_probe.txt)."""

    def __init__(self, env_config, wrap_config=None, num_envs: int = 1, parts: int = 2, device=None, seed: int = 0, env_id_offset: int = 0,
                 cu_wide: Optional[bool] = True, **kw):
        from .distributed import side_stream
        if parts < 1 or num_envs % parts:
            raise ValueError(f"SplitBatchEnv: {num_envs} envs do not split into {parts} equal parts")
        self.num_envs, self.n_parts = int(num_envs), int(parts)
        per = self.num_envs // self.n_parts
        if device is None:
            device = f"cuda:{torch.cuda.current_device()}"
        self.device = torch.device(device)
        # (a part alone would not fill the device, and the automatic choice would give it 256-thread workgroups: the CU-wide form --
        # pace keeping, the in-kernel deal -- is what the parts are meant to keep: a create-time option of every part, evac_options_t)
        from .options import current_default
        opts = kw.pop("options", None) or current_default()
        if cu_wide is not None:
            opts = opts.replace(cu_wide=1 if cu_wide else 0)
        opts = opts.replace(parts=1)
        self.parts: List[BatchedEvacuationEnv] = [
            BatchedEvacuationEnv(env_config, wrap_config, num_envs=per, device=self.device, seed=seed, env_id_offset=env_id_offset + k * per,
                                 options=opts, **kw)
            for k in range(self.n_parts)]
        self.obs_dim, self.stats_words = self.parts[0].obs_dim, self.parts[0].stats_words
        main = torch.cuda.current_stream(self.device)
        self.streams = [main]
        for _ in range(1, self.n_parts):
            self.streams.append(side_stream(self.device, beside=self.streams[-1]))     # (a stream on another hardware queue)
        self._done = [torch.cuda.Event() for _ in self.parts]

    def reset(self, **kw):
        """Reset every part (on the current stream); returns the parts' observation tensors."""
        obs = [p.reset(**kw)[0] for p in self.parts]
        torch.cuda.current_stream(self.device).synchronize()      # the parts' streams start from the reset state
        return obs, {}

    def rollout_launcher(self, n_steps: int, outs: Optional[list] = None):
        """``(launch, outs)``: ``launch()`` enqueues one ``n_steps`` rollout launch of every part on the part's own stream (pre-bound
        calls, see BatchedEvacuationEnv.rollout_launcher); ``outs[k]`` is part k's output dict (``slab [T, E / P, D + 3]``, ...)."""
        T = int(n_steps)
        if outs is None:
            outs = [{"slab": torch.empty((T, p.num_envs, p.obs_dim + 3), dtype=torch.float32, device=self.device),
                     "episode_stats": torch.zeros((T, p.num_envs, p.stats_words), dtype=torch.float32, device=self.device)} for p in self.parts]
        # The outputs were allocated (and zero-filled) on the CURRENT stream, and a caller's ``outs`` were last touched on it: the
        # parts' streams must not start writing before that work is done (on a busy current stream the zero-fill could
        # land after a part's rollout and wipe its episode records).
        self.order_after_current()
        calls = [p.rollout_launcher(T, o, stream=s) for p, o, s in zip(self.parts, outs, self.streams)]

        def launch():
            for c in calls:
                c()
        return launch, outs

    def order_after_current(self):
        """Make every part's stream wait for what the current stream holds so far (output buffers allocated, filled or last read
        there).  ``rollout_launcher`` calls it once; callers that refill or reuse ``outs`` on their own stream call it again before
        the next ``launch()``."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            if s != cur:
                s.wait_stream(cur)

    def join(self, stream=None):
        """Make ``stream`` (default: the current stream) wait for everything the parts' streams have been given so far."""
        stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        for ev, s in zip(self._done, self.streams):
            if s != stream:
                ev.record(s)
                stream.wait_event(ev)

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def rollout(self, n_steps: int, concatenate: bool = True):
        """One launch per part, joined on the current stream; the parts' dicts, plus ``slab [T, E, D + 3]`` of the whole batch in
        global env order if ``concatenate`` (a copy)."""
        launch, outs = self.rollout_launcher(n_steps)
        launch()
        self.join()
        res = {"parts": outs}
        if concatenate:
            res["slab"] = torch.cat([o["slab"] for o in outs], dim=1)
            res["episode_stats"] = torch.cat([o["episode_stats"] for o in outs], dim=1)
        return res

    def kernel_variant(self, mode: str = "rollout") -> str:
        return self.parts[0].kernel_variant(mode)

    def team_error(self) -> bool:
        return any(p.team_error() for p in self.parts)

    def close(self):
        for p in self.parts:
            p.close()
