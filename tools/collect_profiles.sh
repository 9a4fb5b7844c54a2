#!/bin/bash
# Copies what tools/final_run.sh TAG left under gpurun_out/ into profiles/ under the names DESIGN.md cites (run here, after the
# gpurun call has merged its outputs back).  usage: tools/collect_profiles.sh r04_j
set -e
T=${1:?tag}
cd "$(dirname "$0")/.."
G=gpurun_out/$T
for f in $G/bench_*.json; do
  b=$(basename $f .json); tail -1 $f > profiles/${T}_${b}.json
done
for k in c2_driver c2_step c3 c5 big_step; do
  d=gpurun_out/${T}_$k
  [ -d $d ] || continue
  cp $d/stats/*/*_kernel_stats.csv profiles/${T}_${k}_kernel_stats.csv
  { cat $d/command.txt; echo; cat $d/summary.txt; } > profiles/${T}_${k}_pmc_summary.txt
done
for f in launch_intercept subwave policy_example gather_cost; do [ -f $G/$f.txt ] && grep -v amdgpu.ids $G/$f.txt > profiles/${T}_$f.txt; done
[ -f $G/mfma_4x4.txt ] && cp $G/mfma_4x4.txt profiles/${T}_mfma_4x4x1_microbench.txt
cp $G/traffic.json profiles/traffic.json
ls profiles | grep "^$T" | wc -l
