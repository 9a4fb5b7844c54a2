mkdir -p gpurun_out/r02_f; cd gpurun_out/r02_f
python ../../bench.py > bench_c2_default.json 2> bench_c2_default.err
python ../../bench.py --steps 20 --warmup 5 --no-cpu-baseline > bench_c2_driver.json 2>/dev/null
python ../../bench.py --workload c3 --no-cpu-baseline > bench_c3.json 2>/dev/null
EVAC_CELLS=1 python ../../bench.py --workload c3 --no-cpu-baseline --no-step-api > bench_c3_cells.json 2>/dev/null
python ../../bench.py --workload c5 --no-cpu-baseline > bench_c5.json 2>/dev/null
EVAC_CELLS=0 python ../../bench.py --workload c5 --no-cpu-baseline --no-step-api > bench_c5_allpairs.json 2>/dev/null
python ../../bench.py --workload c5 --envs 256 --no-cpu-baseline --no-step-api > bench_c5_256envs.json 2>/dev/null
python ../../bench.py --workload big --steps 200 --warmup 40 --inner 20 --blocks 3 --no-cpu-baseline --no-step-api > bench_big.json 2>bench_big.err
python ../../bench.py --workload big --mode step --steps 100 --warmup 20 --blocks 3 --no-cpu-baseline --no-step-api > bench_big_step.json 2>bench_big_step.err
python ../../bench.py --workload c2 --envs 65536 --no-cpu-baseline --no-step-api > bench_c2_65536.json 2>/dev/null
python ../../tools/subwave_bench.py > subwave.txt 2>&1
python ../../examples/rollout_with_policy.py > policy_example.txt 2>&1
cd ../..
bash tools/profile_pmc.sh r02_d_c2_driver > /dev/null 2>&1
bash tools/profile_pmc.sh r02_d_c3 --workload c3 --steps 400 --warmup 100 --blocks 2 > /dev/null 2>&1
bash tools/profile_pmc.sh r02_d_c5 --workload c5 --steps 400 --warmup 100 --blocks 2 > /dev/null 2>&1
bash tools/profile_pmc.sh r02_d_c2_step --mode step --steps 200 --warmup 50 --blocks 2 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_d_big/stats -- python3 $GRAFT_REPO_ROOT/bench.py --workload big --steps 100 --warmup 20 --inner 20 --blocks 2 --no-cpu-baseline --no-step-api > $GRAFT_REPO_ROOT/gpurun_out/r02_d_big_stats.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/r02_f
