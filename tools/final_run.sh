# The round's measurements (run ON the GPU box: gpurun -- bash tools/final_run.sh <tag>): bench lines under gpurun_out/<tag>/,
# rocprofv3 --stats + PMC passes under gpurun_out/<tag>_<workload>/ (tools/profile_pmc.sh).
TAG=${1:-r03_g}
mkdir -p gpurun_out/$TAG; cd gpurun_out/$TAG
python ../../bench.py > bench_c2_default.json 2> bench_c2_default.err
python ../../bench.py --steps 20 --warmup 5 --no-cpu-baseline > bench_c2_driver.json 2>/dev/null
python ../../bench.py --workload c3 --no-cpu-baseline > bench_c3.json 2>/dev/null
python ../../bench.py --workload c5 --no-cpu-baseline > bench_c5.json 2>/dev/null
EVAC_TEAM=0 python ../../bench.py --workload c5 --no-cpu-baseline --no-step-api > bench_c5_one_workgroup.json 2>/dev/null
EVAC_TEAM_COOP=1 python ../../bench.py --workload c5 --no-cpu-baseline --no-step-api > bench_c5_cooperative_launch.json 2>/dev/null
[ -f ../../tools/ablate_libs/lib_counter.so ] && EVAC_LIB=../../tools/ablate_libs/lib_counter.so python ../../bench.py --workload c5 --no-cpu-baseline --no-step-api > bench_c5_counter_exchange.json 2>/dev/null
python ../../bench.py --workload c5 --envs 256 --no-cpu-baseline --no-step-api > bench_c5_256envs.json 2>/dev/null
python ../../bench.py --workload big --steps 200 --warmup 40 --inner 20 --blocks 3 --no-cpu-baseline --no-step-api > bench_big.json 2>bench_big.err
python ../../bench.py --workload big --mode step --steps 100 --warmup 20 --blocks 3 --no-cpu-baseline --no-step-api > bench_big_step.json 2>bench_big_step.err
python ../../bench.py --workload c2 --envs 65536 --no-cpu-baseline --no-step-api > bench_c2_65536.json 2>/dev/null
EVAC_CU_WIDE=0 python ../../bench.py --no-cpu-baseline --no-step-api > bench_c2_256thread_workgroups.json 2>/dev/null
EVAC_WORKSPACE=0 python ../../bench.py --no-cpu-baseline --no-step-api > bench_c2_no_schedule.json 2>/dev/null
EVAC_SPECIALIZE=0 python ../../bench.py --no-cpu-baseline > bench_c2_generic_kernel.json 2>/dev/null
EVAC_SPECIALIZE=0 python ../../bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-step-api > bench_c2_driver_generic_kernel.json 2>/dev/null
EVAC_BENCH_FORCE_DEVICE=0 EVAC_BENCH_BACKEND=gloo python ../../bench.py --gpus 2 --steps 20 --warmup 5 --no-step-api > bench_c2_two_ranks_one_gpu_gloo.json 2>/dev/null
EVAC_BENCH_FORCE_DEVICE=0 EVAC_BENCH_BACKEND=gloo python ../../bench.py --gpus 2 --steps 20 --warmup 5 --no-step-api --gather direct > bench_c2_two_ranks_one_gpu_direct.json 2>/dev/null
EVAC_BENCH_FORCE_DEVICE=0 EVAC_BENCH_BACKEND=gloo python ../../bench.py --gpus 2 --steps 20 --warmup 5 --no-step-api --gather peer > bench_c2_two_ranks_one_gpu_peer.json 2>/dev/null
python ../../tools/subwave_bench.py > subwave.txt 2>&1
python ../../examples/rollout_with_policy.py > policy_example.txt 2>&1
python ../../tools/moving_distribution.py > moving_distribution.txt 2>&1
python ../../tools/launch_intercept.py > launch_intercept.txt 2>&1
../../tools/microbench/team_barrier > team_barrier.txt 2>&1
../../tools/microbench/team_sentinel > team_sentinel.txt 2>&1
python ../../tools/gather_overlap.py > gather_overlap.txt 2>&1
python ../../tools/block_overhead.py 20 > block_overhead.txt 2>&1
../../tools/microbench/valu_rates > valu_rates.txt 2>&1
cd ../..
bash tools/profile_pmc.sh ${TAG}_c2_driver > /dev/null 2>&1
bash tools/profile_pmc.sh ${TAG}_c3 --workload c3 --steps 400 --warmup 100 --blocks 2 > /dev/null 2>&1
bash tools/profile_pmc.sh ${TAG}_c5 --workload c5 --steps 400 --warmup 100 --blocks 2 > /dev/null 2>&1
bash tools/profile_pmc.sh ${TAG}_c2_step --mode step --steps 200 --warmup 50 --blocks 2 > /dev/null 2>&1
python tools/make_traffic_json.py gpurun_out/${TAG}_c2_driver c2:rollout 4096 20 gpurun_out/${TAG}_c3 c3:rollout 1024 100 gpurun_out/${TAG}_c5 c5:rollout 32 100 gpurun_out/${TAG}_c2_step c2:step 4096 1 > gpurun_out/$TAG/traffic_update.txt 2>&1
cp profiles/traffic.json gpurun_out/$TAG/traffic.json
ls gpurun_out/$TAG; tail -3 gpurun_out/${TAG}_c2_driver/summary.txt
