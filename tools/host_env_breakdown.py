"""Where a HostVectorEnv.step() goes (C2 batch: 4096 envs x 60 pedestrians): the launch alone, launch + stream synchronisation, the whole
step with and without copies, zero-copy (the kernel reads / writes the pinned buffers) against staged (a copy each way)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import evacuation_amd as ea  # noqa: E402


def per(fn, n=400, warm=40):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def main():
    E = 4096
    cfg = ea.EnvConfig(number_of_pedestrians=60)
    wrap = ea.EnvWrappersConfig(positions="grav")
    act = np.random.default_rng(0).uniform(-1, 1, (E, 2)).astype(np.float32)
    for normalize in (False, True):
        for zero_copy in (True, False):
            for copy in (True, False):
                h = ea.HostVectorEnv.make(cfg, wrap, num_envs=E, normalize=normalize, copy=copy, zero_copy=zero_copy, seed=3)
                h.reset()
                line = f"normalize={normalize!s:5} zero_copy={zero_copy!s:5} copy={copy!s:5}  step {per(lambda: h.step(act)):6.1f} us"
                if zero_copy and copy:
                    line += f"   [launch alone {per(h._launch):5.1f} us, launch + sync {per(lambda: (h._launch(), h._sync())):5.1f} us]"
                print(line, flush=True)
                h.close()




def flag_wait_experiment(E=4096):
    """launch + hipStreamSynchronize against launch + hipStreamWriteValue32 into a pinned word the host spins on."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamWriteValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint]
    cfg = ea.EnvConfig(number_of_pedestrians=60)
    h = ea.HostVectorEnv.make(cfg, ea.EnvWrappersConfig(positions="grav"), num_envs=E, normalize=False, seed=3)
    h.reset()
    flag_t = torch.zeros((16,), dtype=torch.int32).pin_memory()
    flag = flag_t.numpy()
    ptr = C.c_void_p(flag_t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = [0]

    def with_flag():
        n[0] += 1
        v = n[0]
        h._launch()
        rc = hip.hipStreamWriteValue32(st, ptr, v, 0)
        assert rc == 0, rc
        while flag[0] != v:
            pass

    print(f"launch + sync {per(lambda: (h._launch(), h._sync())):5.1f} us   launch + stream write + host spin {per(with_flag):5.1f} us", flush=True)
    h.close()


def distribution(E=4096, n=3000):
    """Per-step wall time of HostVectorEnv.step() (the distribution, not only the mean)."""
    cfg = ea.EnvConfig(number_of_pedestrians=60)
    act = np.random.default_rng(0).uniform(-1, 1, (E, 2)).astype(np.float32)
    for zero_copy in (True, False, True, False):
        h = ea.HostVectorEnv.make(cfg, ea.EnvWrappersConfig(positions="grav"), num_envs=E, normalize=False, seed=3, zero_copy=zero_copy)
        h.reset()
        for _ in range(100):
            h.step(act)
        t = np.zeros(n)
        for i in range(n):
            t0 = time.perf_counter()
            h.step(act)
            t[i] = time.perf_counter() - t0
        t *= 1e6
        print(f"zero_copy={zero_copy!s:5}  mean {t.mean():6.1f}  median {np.median(t):6.1f}  p90 {np.percentile(t, 90):6.1f}  p99 {np.percentile(t, 99):6.1f}  "
              f"max {t.max():7.1f} us", flush=True)
        h.close()


if __name__ == "__main__":
    if "--dist" in sys.argv:
        distribution()
    elif "--flag" in sys.argv:
        flag_wait_experiment()
    else:
        main()
