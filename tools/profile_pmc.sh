#!/bin/bash
# rocprofv3 passes for the bench workload (run ON the GPU box via gpurun).  Separate --pmc passes
# (never combined with trace domains other than --kernel-trace), outputs under gpurun_out/<tag>/.
# usage: tools/profile_pmc.sh <tag> [bench args...]      (default bench args: the driver's `--gpus 1 --steps 20 --warmup 5`)
# The profiled command is bench.py itself with the same arguments (+ --no-cpu-baseline --no-step-api: the CPU leg and
# the one-launch-per-step side measurement launch other kernels and are not what the roofline block describes).
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- --gpus 1 --steps 20 --warmup 5; fi
ARGS="--no-cpu-baseline --no-step-api $*"
echo "bench.py $ARGS" > "$OUT/command.txt"
# (every pass under its own time limit and with a line of progress: a pass that hangs must not take the box's whole call with it)
echo "$(date +%T) $TAG: kernel trace" >> "$ROOT/gpurun_out/pmc_progress.txt"
timeout -k 10 ${PMC_PASS_LIMIT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1 || echo "$(date +%T) $TAG: kernel trace FAILED or timed out" >> "$ROOT/gpurun_out/pmc_progress.txt"
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" \
            "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_TRANS SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  echo "$(date +%T) $TAG: counter pass $i" | tee -a "$ROOT/gpurun_out/pmc_progress.txt"
  timeout -k 10 ${PMC_PASS_LIMIT:-300} rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc$i.log" 2>&1 || { echo "$(date +%T) $TAG: counter pass $i FAILED or timed out: no further pass" | tee -a "$ROOT/gpurun_out/pmc_progress.txt"; break; }
done
cd "$ROOT"
python3 tools/summarize_pmc.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
