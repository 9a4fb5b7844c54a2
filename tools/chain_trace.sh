#!/bin/bash
# kernel trace of a few chained sweeps: do consecutive launches overlap, and on which queues?  usage: tools/chain_trace.sh LIB OUTDIR
cd /tmp && export TMPDIR=/tmp
lib=$1; out=$2
cd $GRAFT_REPO_ROOT
EVAC_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 tools/steady_probe.py 2 20 e > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
r = [x for x in csv.DictReader(open(f)) if "k_rollout_chain" in x["Kernel_Name"]]
r.sort(key=lambda x: int(x["Start_Timestamp"]))
n = len(r)
ov = 0; gaps = []
for a, b in zip(r[200:400], r[201:401]):
    if int(b["Start_Timestamp"]) < int(a["End_Timestamp"]): ov += 1
    gaps.append((int(b["Start_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
q = {}
for x in r: q[x["Queue_Id"]] = q.get(x["Queue_Id"], 0) + 1
dur = sorted((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in r[200:400])
print(f"{n} chained launches; queues {q}; of 200 consecutive pairs {ov} overlap; start-to-start median {sorted(gaps)[100]:.1f} us; duration median {dur[100]:.1f} us")
t0 = int(r[300]["Start_Timestamp"])
for x in r[300:308]:
    print(f"   {(int(x['Start_Timestamp']) - t0) / 1e3:8.1f} -> {(int(x['End_Timestamp']) - t0) / 1e3:8.1f} us  queue {x['Queue_Id']}")
PY
grep steady $out.log | cut -c1-170
