#!/bin/bash
# which tests of tests/test_gpu_parity.py must run in the same process for the now-and-then mismatches to show?
out=gpurun_out/flaky_context2.txt
: > $out
run() {   # name, iterations, -k expression
  echo "== $1: -k '$3'" >> $out
  timeout -k 10 400 python tools/repeat_test.py $2 gpurun_out/sel_$1.txt tests/test_gpu_parity.py -m gpu -k "$3" | tail -1 >> $out
  grep -E "^FAILED|side differs" gpurun_out/sel_$1.txt | cut -c1-400 >> $out
  echo "$1 done"
}
run s1 12 "two_parts or chained_launches_equal"
run s2 12 "split_batch or chained_launches_equal"
run s3 12 "sharded_handles or autoreset or chained_launches_equal"
run s4 12 "chained_launches_equal or lost_is_reported"
run s5 8 "teacher_forced or crafted or random_states or chained_launches_equal"
run s6 8 "not chained and not two_parts"
