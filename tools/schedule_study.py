"""Offline study of the env-to-SIMD dealing rule of k_schedule (C2: four one-wave envs per SIMD, a launch lasts as long as
its heaviest SIMD).  Input: gpurun_out/loads.npz (tools/dump_loads.py).  Cost model of one env step in instruction slots:
no row needed B0; few rows (transposed) B1 + 54 per pair of rows; otherwise B1 + 15 + 8 per column (columns in fours)."""
import sys
import numpy as np

d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/loads.npz")
L = d["c2"].astype(np.int64)          # [phase, env, (v, f, x)]
B0, B1 = 250, 330


def cost_of(v, f, x):
    nc = v + f + x
    nc4 = (nc + 3) // 4 * 4
    pairs = (v + 1) // 2
    loop = B1 + 15 + 8 * nc4
    tr = B1 + 54 * pairs
    use_tr = (v <= 8) & (pairs * 54 < 20 + 8 * nc4)
    c = np.where(use_tr, tr, loop)
    return np.where(v == 0, B0, c)


def snake(c):
    s = np.sort(c); q = len(c) // 4
    return s[:q] + s[q:2 * q][::-1] + s[2 * q:3 * q] + s[3 * q:][::-1]


def lpt(c, slots=4):
    import heapq
    order = np.argsort(-c)
    nb = len(c) // slots
    heap = [(0, b, 0) for b in range(nb)]
    heapq.heapify(heap)
    sums = np.zeros(nb, np.int64)
    for i in order:
        s, b, k = heapq.heappop(heap)
        s += c[i]; k += 1
        sums[b] = s
        if k < slots:
            heapq.heappush(heap, (s, b, k))
    return sums


def rounds(c, slots=4):
    """Quartile rounds, adaptive: heaviest quartile one per bin; then the LIGHTEST quartile ascending to the bins by
    descending sum; then the two middle quartiles likewise (bins re-sorted before every round)."""
    s = np.sort(c); q = len(c) // 4
    sums = s[3 * q:][::-1].copy()
    for part in (s[:q], s[q:2 * q], s[2 * q:3 * q]):
        o = np.argsort(-sums, kind="stable")
        sums[o] += part
    return sums


def fold2(c):
    s = np.sort(c); n = len(c)
    p = s[:n // 2] + s[n // 2:][::-1]
    p = np.sort(p)
    return p[:n // 4] + p[n // 4:][::-1]


print("phase  mean   random  snake   rounds  fold2   lpt     (heaviest SIMD / mean SIMD)")
rng = np.random.default_rng(0)
tot = {k: 0.0 for k in ("random", "snake", "rounds", "fold2", "lpt")}
for ph in range(L.shape[0]):
    c = cost_of(L[ph, :, 0], L[ph, :, 1], L[ph, :, 2])
    mean4 = c.mean() * 4
    r = {"random": c[rng.permutation(len(c))].reshape(-1, 4).sum(1).max(), "snake": snake(c).max(), "rounds": rounds(c).max(),
         "fold2": fold2(c).max(), "lpt": lpt(c).max()}
    for k in tot:
        tot[k] += r[k]
    if ph % 4 == 0:
        print(f"t={ph * 50:5d} {c.mean():6.1f} " + " ".join(f"{r[k] / mean4:7.3f}" for k in ("random", "snake", "rounds", "fold2", "lpt")))
base = sum(cost_of(L[ph, :, 0], L[ph, :, 1], L[ph, :, 2]).mean() * 4 for ph in range(L.shape[0]))
print("episode: " + " ".join(f"{k} {tot[k] / base:.3f}" for k in tot))
