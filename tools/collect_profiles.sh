#!/bin/bash
# Copies what tools/final_run.sh TAG left under gpurun_out/ into profiles/ under the names DESIGN.md cites (run here, after the
# gpurun call has merged its outputs back).  usage: tools/collect_profiles.sh r06_x
set -e
T=${1:?tag}
cd "$(dirname "$0")/.."
G=gpurun_out/$T
for f in $G/bench_*.json; do
  b=$(basename $f .json); [ -s $f ] && tail -1 $f > profiles/${T}_${b}.json
done
# the driver's command under the kernel trace: per-kernel stats, the timeline of its (chained) launches, the line the traced run printed
d=gpurun_out/${T}_c2_driver
if [ -d $d ]; then
  cp "$(ls -t $d/stats/*/*_kernel_stats.csv | head -1)" profiles/${T}_c2_driver_kernel_stats.csv      # (the newest: a re-run leaves the older process's tables beside it)
  { cat $d/command.txt; echo; cat $d/timeline.txt; } > profiles/${T}_c2_driver_timeline.txt
  tail -1 $d/bench_line.json > profiles/${T}_c2_driver_traced_line.json
fi
d=gpurun_out/${T}_c2_chained
if [ -d $d ]; then
  cp "$(ls -t $d/stats/*/*_kernel_stats.csv | head -1)" profiles/${T}_c2_chained_kernel_stats.csv
  { cat $d/command.txt; echo; cat $d/timeline.txt; } > profiles/${T}_c2_chained_timeline.txt
  tail -1 $d/bench_line.json > profiles/${T}_c2_chained_traced_line.json
fi
# every entry of `workloads` traced on its own (the trace covers the launches that entry times and no others)
for s in c2_one_kernel c3 c5_shard big_step; do
  d=gpurun_out/${T}_side_$s
  [ -d $d ] || continue
  cp "$(ls -t $d/stats/*/*_kernel_stats.csv | head -1)" profiles/${T}_side_${s}_kernel_stats.csv
  tail -1 $d/bench_line.json > profiles/${T}_side_${s}_traced_line.json
  [ -f $d/timeline.txt ] && { cat $d/command.txt; echo; cat $d/timeline.txt; } > profiles/${T}_side_${s}_timeline.txt
done
# counter passes
for k in c2_plain_pmc c2_step c3 c5 big_step; do
  d=gpurun_out/${T}_$k
  [ -d $d ] || continue
  cp "$(ls -t $d/stats/*/*_kernel_stats.csv | head -1)" profiles/${T}_${k}_kernel_stats.csv
  { cat $d/command.txt; echo; cat $d/summary.txt; } > profiles/${T}_${k}_pmc_summary.txt
done
for f in steady_probe subwave policy_example progress traffic_update; do [ -f $G/$f.txt ] && grep -v amdgpu.ids $G/$f.txt > profiles/${T}_$f.txt; done
[ -f $G/traffic.json ] && cp $G/traffic.json profiles/traffic.json
ls profiles | grep "^$T" | wc -l
