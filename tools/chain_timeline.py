#!/usr/bin/env python3
"""A rocprofv3 --kernel-trace CSV of a bench.py run as the timeline of its rollout launches: launches per hardware queue, the
start-to-start distance of consecutive launches (= the launch PERIOD the roofline is priced on), each kernel's own duration (a
chained launch is enqueued behind its queue's previous launch and ends two periods later), how many consecutive launches overlap, and
a sample of begin / end pairs.  A PERSISTENT rollout kernel (evac_options_t.chain = 2) carries all calls of a sweep: its duration over the
calls per sweep (third argument) is the call period.  usage: chain_timeline.py <dir with *_kernel_trace.csv> [kernel substring] [calls per kernel]"""
import csv, glob, statistics as st, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_rollout"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = [x for x in csv.DictReader(open(f)) if sub in x["Kernel_Name"]]
names = {}
for x in rows:
    names[x["Kernel_Name"]] = names.get(x["Kernel_Name"], 0) + 1
name = max(names, key=names.get)                 # the most frequent rollout kernel: the headline's
r = sorted((x for x in rows if x["Kernel_Name"] == name), key=lambda x: int(x["Start_Timestamp"]))
S = [int(x["Start_Timestamp"]) / 1e3 for x in r]; E = [int(x["End_Timestamp"]) / 1e3 for x in r]
q = {}
for x in r:
    q[x["Queue_Id"]] = q.get(x["Queue_Id"], 0) + 1
print(f"kernel: {name[:150]}")
if "persist" in name:
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    du_ = sorted((E[i] - S[i]) for i in range(len(r)))
    full = [x for x in du_ if x > 0.5 * du_[len(du_) * 9 // 10]]       # (whole sweeps: the joins' finishers -- a few us each --, the warm-up and the single-call kernels of the diagnostics left out)
    short = [x for x in du_ if x < 30.0]
    print(f"persistent kernels: {len(r)} in the trace, {len(full)} of them whole sweeps (the others: {len(short)} of under 30 us -- the finishers behind the joins, median "
          f"{st.median(short) if short else 0.0:.1f} us -- and single calls of the warm-up and of the diagnostics); "
          f"duration of a whole sweep [us]: mean {st.mean(full):.1f}, median {st.median(full):.1f}")
    if calls:
        print(f"  = {st.mean(full) / calls:.2f} (mean) / {st.median(full) / calls:.2f} (median) us per call, {calls} calls per kernel: kernel start, the calls' steps and STOP; "
              f"the sweep's join and its timing events lie outside the kernel")
    sys.exit(0)
print(f"launches: {len(r)}; per hardware queue: {q}")
# consecutive launches closer than 2.5 median distances belong to one back-to-back run (a sweep); the gaps between sweeps are left out
all_ss = [S[i + 1] - S[i] for i in range(len(r) - 1)]
ss = [x for x in all_ss if x < 2.5 * st.median(all_ss)]
du = [E[i] - S[i] for i in range(len(r))]
ov = sum(1 for i in range(len(r) - 1) if S[i + 1] < E[i])
print(f"start-to-start of consecutive launches inside a sweep [us]: mean {st.mean(ss):.2f}, median {st.median(ss):.2f}, p10 {sorted(ss)[len(ss) // 10]:.2f}, p90 {sorted(ss)[-len(ss) // 10]:.2f}  (n = {len(ss)})")
print(f"a kernel's own duration, begin -> end [us]: mean {st.mean(du):.2f}, median {st.median(du):.2f}  -> duration / period = {st.mean(du) / st.mean(ss):.2f} kernels in flight")
print(f"consecutive launches that overlap (the next begins before this one ends): {ov} of {len(r) - 1}")
k = len(r) * 3 // 4
t0 = S[k]
print("a sample late in the run (begin -> end, us, queue):")
for i in range(k, min(k + 10, len(r))):
    print(f"   {S[i] - t0:9.1f} -> {E[i] - t0:9.1f}   queue {r[i]['Queue_Id']}")
