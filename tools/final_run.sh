# The round's measurements (run ON the GPU box: gpurun -- bash tools/final_run.sh <tag>): bench lines under gpurun_out/<tag>/,
# rocprofv3 --stats + PMC passes under gpurun_out/<tag>_<workload>/ (tools/profile_pmc.sh), then profiles/traffic.json.
TAG=${1:-r05_a}
mkdir -p gpurun_out/$TAG; cd gpurun_out/$TAG
python ../../bench.py --steps 20 --warmup 5 > bench_c2_driver.json 2> bench_c2_driver.err
python ../../bench.py --no-cpu-baseline > bench_c2_default.json 2>/dev/null
python ../../bench.py --workload c3 --no-cpu-baseline --no-step-api > bench_c3.json 2>/dev/null
python ../../bench.py --workload c5 --no-cpu-baseline --no-step-api > bench_c5.json 2>/dev/null
EVAC_TEAM=0 python ../../bench.py --workload c5 --no-cpu-baseline --no-step-api > bench_c5_one_workgroup.json 2>/dev/null
python ../../bench.py --workload c5 --envs 256 --no-cpu-baseline --no-step-api > bench_c5_256envs.json 2>/dev/null
python ../../bench.py --workload big --steps 200 --warmup 40 --inner 20 --sweeps 1 --no-cpu-baseline --no-step-api > bench_big.json 2>bench_big.err
python ../../bench.py --workload big --mode step --steps 100 --warmup 20 --sweeps 1 --no-cpu-baseline --no-step-api > bench_big_step.json 2>bench_big_step.err
python ../../bench.py --workload c2 --envs 65536 --no-cpu-baseline --no-step-api > bench_c2_65536.json 2>/dev/null
EVAC_CU_WIDE=0 python ../../bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-step-api > bench_c2_driver_256thread_workgroups.json 2>/dev/null
EVAC_WORKSPACE=0 python ../../bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-step-api > bench_c2_driver_no_schedule.json 2>/dev/null
EVAC_SPECIALIZE=0 python ../../bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-step-api > bench_c2_driver_generic_kernel.json 2>/dev/null
python ../../bench.py --force-gather --steps 20 --warmup 5 --no-cpu-baseline --no-step-api > bench_c2_driver_force_gather_rccl_world1.json 2>/dev/null
python ../../bench.py --force-gather --gather peer --steps 20 --warmup 5 --no-cpu-baseline --no-step-api > bench_c2_driver_force_gather_peer_world1.json 2>/dev/null
python ../../bench.py --force-gather --device-wait --buffers 3 --steps 20 --warmup 5 --no-cpu-baseline --no-step-api > bench_c2_driver_force_gather_rccl_world1_device_side_wait.json 2>/dev/null
python ../../tools/gather_cost.py 2>&1 | grep "us per gather\|bracket" > gather_cost.txt
python ../../tools/subwave_bench.py > subwave.txt 2>&1
python ../../examples/rollout_with_policy.py > policy_example.txt 2>&1
python ../../tools/launch_intercept.py > launch_intercept.txt 2>&1
cd ../..
bash tools/profile_pmc.sh ${TAG}_c2_driver > /dev/null 2>&1
bash tools/profile_pmc.sh ${TAG}_c3 --workload c3 --steps 400 --warmup 100 --sweeps 1 > /dev/null 2>&1
bash tools/profile_pmc.sh ${TAG}_c5 --workload c5 --steps 400 --warmup 100 --sweeps 1 > /dev/null 2>&1
bash tools/profile_pmc.sh ${TAG}_c2_step --mode step --steps 200 --warmup 50 --sweeps 1 > /dev/null 2>&1
bash tools/profile_pmc.sh ${TAG}_big_step --workload big --mode step --steps 100 --warmup 20 --sweeps 1 --blocks 3 > /dev/null 2>&1
python tools/make_traffic_json.py gpurun_out/${TAG}_c2_driver c2:rollout 4096 20 gpurun_out/${TAG}_c3 c3:rollout 1024 100 gpurun_out/${TAG}_c5 c5:rollout 32 100 gpurun_out/${TAG}_c2_step c2:step 4096 1 gpurun_out/${TAG}_big_step big:step 524288 1 > gpurun_out/$TAG/traffic_update.txt 2>&1
cp profiles/traffic.json gpurun_out/$TAG/traffic.json
# (gpurun merges at most 64 MiB back: the raw per-dispatch counter and trace tables go, the summaries and the stats tables stay)
find gpurun_out/${TAG}_* -name "*counter_collection.csv" -delete
find gpurun_out/${TAG}_* -name "*kernel_trace.csv" -delete
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-step-api > gpurun_out/$TAG/bench_c2_driver_with_counters.json 2>/dev/null
ls gpurun_out/$TAG; tail -3 gpurun_out/${TAG}_c2_driver/summary.txt; tail -5 gpurun_out/$TAG/traffic_update.txt
