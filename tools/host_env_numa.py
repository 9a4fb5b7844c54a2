"""Is the zero-copy HostVectorEnv (the step kernel reading / writing pinned host memory) sensitive to WHERE that memory is?  One child
process per NUMA node that this process may run on: pinned to the node's CPUs before anything is allocated (pinned memory is placed on
the allocating thread's node), then 1000 steps each way.  GPU box."""
import glob
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpulist(text):
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def child(cpus):
    os.sched_setaffinity(0, cpus)
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import evacuation_amd as ea
    prop = torch.cuda.get_device_properties(0)
    bdf = f"{getattr(prop, 'pci_domain_id', 0):04x}:{prop.pci_bus_id:02x}:{prop.pci_device_id:02x}.0"
    try:
        gpu_node = open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip()
    except OSError:
        gpu_node = "?"
    E = 4096
    act = np.random.default_rng(0).uniform(-1, 1, (E, 2)).astype(np.float32)
    res = {}
    h = ea.HostVectorEnv.make(ea.EnvConfig(number_of_pedestrians=60), ea.EnvWrappersConfig(positions="grav"), num_envs=E, normalize=False, seed=3)
    h.reset()

    def per(fn, n=1000, warm=100):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e6

    parts = (per(h._launch), per(lambda: (h._launch(), h._sync())), per(lambda: h._np_obs.copy()), per(lambda: np.copyto(h._np_act, act)))
    h.close()
    for zero_copy in (True, False):
        h = ea.HostVectorEnv.make(ea.EnvConfig(number_of_pedestrians=60), ea.EnvWrappersConfig(positions="grav"), num_envs=E, normalize=False, seed=3,
                                  zero_copy=zero_copy)
        h.reset()
        for _ in range(100):
            h.step(act)
        t0 = time.perf_counter()
        for _ in range(1000):
            h.step(act)
        res[zero_copy] = (time.perf_counter() - t0) / 1000 * 1e6
        if zero_copy:     # the same instance, piece by piece
            same = (per(h._launch), per(lambda: (h._launch(), h._sync())), per(lambda: (np.copyto(h._np_act, act), h._launch(), h._sync())),
                    per(lambda: (h._launch(), h._sync(), h._np_obs.copy())), per(lambda: h._np_obs.copy()),
                    per(lambda: (h._launch(), h._sync(), h._np_reward.astype(np.float64), h._np_term != 0, h._np_trunc != 0)))
            ptrs = (h._h_out.data_ptr(), h._h_act.data_ptr())
        h.close()
    print(f"GPU {bdf} on node {gpu_node}; process on CPUs {min(cpus)}..{max(cpus)} ({len(cpus)}): zero-copy {res[True]:6.1f} us per step, staged {res[False]:6.1f}   "
          f"[launches back to back {parts[0]:5.1f}, launch + sync {parts[1]:5.1f}, obs copy {parts[2]:4.1f}, actions in {parts[3]:4.1f}; first instance]\n"
          f"      the timed instance piece by piece: launches back to back {same[0]:5.1f}, launch + sync {same[1]:5.1f}, actions in + launch + sync {same[2]:5.1f}, "
          f"launch + sync + obs copy {same[3]:5.1f}, obs copy alone {same[4]:4.1f}, launch + sync + reward / flags {same[5]:5.1f}; out buffer {ptrs[0]:#x} actions {ptrs[1]:#x}", flush=True)


def main():
    allowed = os.sched_getaffinity(0)
    print(f"allowed CPUs: {len(allowed)} ({min(allowed)}..{max(allowed)})")
    for path in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        cpus = cpulist(open(os.path.join(path, "cpulist")).read()) & allowed
        print(f"{os.path.basename(path)}: {len(cpus)} of the allowed CPUs")
        if not cpus:
            continue
        for _ in range(4):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", ",".join(str(c) for c in sorted(cpus))], check=False, timeout=300)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child({int(c) for c in sys.argv[2].split(",")})
    else:
        main()
