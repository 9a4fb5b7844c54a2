#!/bin/bash
# the first processes of a call are where the now-and-then mismatches show: run the chained/parts selection in a few fresh processes
out=gpurun_out/canary.txt
{ rocm-smi --showserial 2>/dev/null | grep Serial; } > $out
for i in 1 2 3 4 5 6; do
  timeout -k 10 200 python tools/repeat_test.py 3 gpurun_out/canary_$i.txt tests/test_gpu_parity.py -m gpu -k "chained_launches_equal or two_parts" | tail -1 >> $out
  grep -h "^== iteration\|^FAILED\|AssertionError" gpurun_out/canary_$i.txt | cut -c1-6000 >> $out
done
