cd /tmp && export TMPDIR=/tmp
for m in 0 1 16; do
  LIB=$GRAFT_REPO_ROOT/tools/ab_libs/libevac_ablate_$m.so; [ $m = 0 ] && LIB=$GRAFT_REPO_ROOT/evacuation_amd/libevac.so
  EVAC_LIB=$LIB rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_conf_$m -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-step-api --steps 200 --warmup 200 --blocks 2 > /dev/null 2>&1
  python3 - <<PY
import csv, glob
v={}
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/r02_conf_$m/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_rollout" in r["Kernel_Name"]: v.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("ablate=$m", {k: sum(x)/len(x)/(4096*100) for k, x in v.items()})
PY
done
