# The round's measurements (run ON the GPU box: gpurun -- bash tools/final_run.sh <tag> [part]): bench lines under gpurun_out/<tag>/,
# rocprofv3 kernel traces / stats and PMC passes under gpurun_out/<tag>_<workload>/, then profiles/traffic.json.
# part: all (default) | lines | profiles.  Every step appends a line to gpurun_out/<tag>/progress.txt (the box's watchdog wants output).
TAG=${1:-r06_a}; PART=${2:-all}
mkdir -p gpurun_out/$TAG
P=gpurun_out/$TAG/progress.txt
say() { echo "$(date +%T) $*" | tee -a $P; }
B="python bench.py"
if [ $PART = all ] || [ $PART = lines ]; then
say "driver line";            $B --steps 20 --warmup 5 > gpurun_out/$TAG/bench_c2_driver.json 2> gpurun_out/$TAG/bench_c2_driver.err
say "driver line, chained";   $B --steps 20 --warmup 5 --rollout-form chain --no-cpu-baseline --no-step-api > gpurun_out/$TAG/bench_c2_driver_chained_launches.json 2>/dev/null
say "driver line, plain";     $B --steps 20 --warmup 5 --rollout-form one --no-cpu-baseline --no-step-api > gpurun_out/$TAG/bench_c2_driver_one_kernel_per_launch.json 2>/dev/null
say "driver line, two parts"; $B --steps 20 --warmup 5 --rollout-form parts --no-cpu-baseline --no-step-api > gpurun_out/$TAG/bench_c2_driver_two_parts.json 2>/dev/null
say "default line";           $B --no-cpu-baseline > gpurun_out/$TAG/bench_c2_default.json 2>/dev/null
say "c3";                     $B --workload c3 --no-cpu-baseline --no-step-api > gpurun_out/$TAG/bench_c3.json 2>/dev/null
say "c5";                     $B --workload c5 --no-cpu-baseline --no-step-api > gpurun_out/$TAG/bench_c5.json 2>/dev/null
say "c5 one workgroup";       EVAC_TEAM=0 $B --workload c5 --no-cpu-baseline --no-step-api > gpurun_out/$TAG/bench_c5_one_workgroup.json 2>/dev/null
say "c5 256 envs";            $B --workload c5 --envs 256 --no-cpu-baseline --no-step-api > gpurun_out/$TAG/bench_c5_256envs.json 2>/dev/null
say "big rollout";            $B --workload big --steps 200 --warmup 40 --inner 20 --sweeps 1 --sustain-seconds 0 --no-cpu-baseline --no-step-api > gpurun_out/$TAG/bench_big.json 2>gpurun_out/$TAG/bench_big.err
say "c2 65536";               $B --workload c2 --envs 65536 --sustain-seconds 1 --no-cpu-baseline --no-step-api > gpurun_out/$TAG/bench_c2_65536.json 2>/dev/null
say "c2 256-thread";          EVAC_CU_WIDE=0 $B --steps 20 --warmup 5 --no-cpu-baseline --no-step-api > gpurun_out/$TAG/bench_c2_driver_256thread_workgroups.json 2>/dev/null
say "c2 generic kernel";      EVAC_SPECIALIZE=0 $B --steps 20 --warmup 5 --no-cpu-baseline --no-step-api > gpurun_out/$TAG/bench_c2_driver_generic_kernel.json 2>/dev/null
say "force gather rccl";      $B --force-gather --steps 20 --warmup 5 --no-cpu-baseline --no-step-api > gpurun_out/$TAG/bench_c2_driver_force_gather_rccl_world1.json 2>/dev/null
say "force gather auto";      $B --force-gather --gather auto --steps 20 --warmup 5 --no-cpu-baseline --no-step-api > gpurun_out/$TAG/bench_c2_driver_force_gather_auto_world1.json 2>/dev/null
say "steady probe";           python tools/steady_probe.py 300 20 aebc 2>&1 | grep -v "amdgpu.ids\|sweep [012]:" > gpurun_out/$TAG/steady_probe.txt
say "subwave";                python tools/subwave_bench.py > gpurun_out/$TAG/subwave.txt 2>&1
say "policy example";         python examples/rollout_with_policy.py > gpurun_out/$TAG/policy_example.txt 2>&1
fi
if [ $PART = all ] || [ $PART = profiles ]; then
ROOT=$(pwd)
# 1. the driver's command itself under the kernel trace (chained launches overlap as they do unprofiled): stats + the timeline of its launches
say "kernel trace of the driver's command"
D=gpurun_out/${TAG}_c2_driver; mkdir -p $D
echo "bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-step-api" > $D/command.txt
(cd /tmp && TMPDIR=/tmp timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$D/stats -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-step-api > $ROOT/$D/stats.log 2>&1)
python tools/chain_timeline.py $D/stats k_rollout 100 > $D/timeline.txt 2>&1      # (100 calls of 20 steps per sweep = per persistent kernel)
grep '^{"' $D/stats.log | tail -1 > $D/bench_line.json   # (rocprofv3 prints its own lines after the program's)
# 1b. the same command with CHAINED launches (evac_options_t.chain = 1): the timeline of two launches in flight
say "kernel trace of the driver's command, chained launches"
D=gpurun_out/${TAG}_c2_chained; mkdir -p $D
echo "bench.py --steps 20 --warmup 5 --rollout-form chain --no-cpu-baseline --no-step-api" > $D/command.txt
(cd /tmp && TMPDIR=/tmp timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$D/stats -- python3 $ROOT/bench.py --steps 20 --warmup 5 --rollout-form chain --no-cpu-baseline --no-step-api > $ROOT/$D/stats.log 2>&1)
python tools/chain_timeline.py $D/stats > $D/timeline.txt 2>&1
grep '^{"' $D/stats.log | tail -1 > $D/bench_line.json
# 2. every entry of `workloads` on its own: the trace covers the launches that entry times and no others
for s in c2_one_kernel c3 c5_shard big_step; do
  say "kernel trace of --side-only $s"
  D=gpurun_out/${TAG}_side_$s; mkdir -p $D
  echo "bench.py --side-only $s --steps 20" > $D/command.txt
  (cd /tmp && TMPDIR=/tmp timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$D/stats -- python3 $ROOT/bench.py --side-only $s --steps 20 > $ROOT/$D/stats.log 2>&1)
  grep '^{"' $D/stats.log | tail -1 > $D/bench_line.json   # (rocprofv3 prints its own lines after the program's)
  python tools/summarize_pmc.py $D > $D/summary.txt 2>&1
  python tools/chain_timeline.py $D/stats k_ 20 > $D/timeline.txt 2>&1      # (start-to-start = the launch period; c3: one persistent kernel per sweep of 20 calls)
done
# 3. counters (separate --pmc passes; short runs: every dispatch is serialised under the counters)
# (rollouts are counted in their PLAIN launches -- --rollout-form one: rocprofv3 --pmc runs every dispatch alone and a chained launch would wait
# for its predecessor for ever; tools/make_traffic_json.py records that and adds the chain's exchange records)
say "pmc c2 plain";   bash tools/profile_pmc.sh ${TAG}_c2_plain_pmc --gpus 1 --steps 20 --warmup 5 --sweeps 2 --sustain-seconds 0 --rollout-form one > /dev/null 2>&1
say "pmc c3";         bash tools/profile_pmc.sh ${TAG}_c3 --workload c3 --steps 400 --warmup 100 --sweeps 1 --sustain-seconds 0 --rollout-form one > /dev/null 2>&1
say "pmc c5";         bash tools/profile_pmc.sh ${TAG}_c5 --workload c5 --steps 400 --warmup 100 --sweeps 1 --sustain-seconds 0 --rollout-form one > /dev/null 2>&1
say "pmc c2 step";    bash tools/profile_pmc.sh ${TAG}_c2_step --mode step --steps 200 --warmup 50 --sweeps 1 --sustain-seconds 0 > /dev/null 2>&1
say "pmc big step";   bash tools/profile_pmc.sh ${TAG}_big_step --workload big --mode step --steps 100 --warmup 20 --sweeps 1 --blocks 3 --sustain-seconds 0 > /dev/null 2>&1
say "traffic.json"
python tools/make_traffic_json.py gpurun_out/${TAG}_c2_plain_pmc c2:rollout 4096 20 gpurun_out/${TAG}_c3 c3:rollout 1024 100 gpurun_out/${TAG}_c5 c5:rollout 32 100 gpurun_out/${TAG}_c2_step c2:step 4096 1 gpurun_out/${TAG}_big_step big:step 524288 1 > gpurun_out/$TAG/traffic_update.txt 2>&1
cp profiles/traffic.json gpurun_out/$TAG/traffic.json
# (gpurun merges at most 64 MiB back: the raw per-dispatch counter and trace tables go, the summaries and the stats tables stay)
find gpurun_out/${TAG}_* -name "*counter_collection.csv" -delete
find gpurun_out/${TAG}_* -name "*kernel_trace.csv" -delete
say "driver line with counters"; $B --steps 20 --warmup 5 --no-cpu-baseline --no-step-api > gpurun_out/$TAG/bench_c2_driver_with_counters.json 2>/dev/null
fi
say "done"; ls gpurun_out/$TAG | head -50; cat gpurun_out/${TAG}_c2_driver/timeline.txt 2>/dev/null; tail -5 gpurun_out/$TAG/traffic_update.txt 2>/dev/null
