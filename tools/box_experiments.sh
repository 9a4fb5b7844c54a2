#!/bin/bash
# On a box where the chained-launch tests fail now and then in fresh processes: which switch makes the failures go away?
# Every trial = P fresh processes of the parity test's chained/parts selection (3 iterations each); a box whose first trial is clean ends the script.
out=gpurun_out/boxexp.txt
sel="chained_launches_equal or two_parts"
P=${P:-6}
{ rocm-smi --showserial 2>/dev/null | grep Serial; } > $out
trial() {   # name, env assignments...
  name=$1; shift
  bad=0
  for i in $(seq $P); do
    r=$(env "$@" timeout -k 10 200 python tools/repeat_test.py 3 gpurun_out/boxexp_${name}_$i.txt tests/test_gpu_parity.py -m gpu -k "$sel" | tail -1)
    f=$(echo "$r" | cut -d" " -f1)
    [ "$f" != "0" ] && bad=$((bad + 1))
  done
  echo "$name: $bad of $P processes had a failure ($*)" >> $out
  echo "$name done: $bad"
}
tool() {   # name, env assignments...
  name=$1; shift
  bad=0
  for i in $(seq $P); do
    r=$(env "$@" timeout -k 10 100 python tools/chain_flaky.py mix 2 0 2>&1 | grep -c MISMATCH)
    [ "$r" != "0" ] && bad=$((bad + 1))
  done
  echo "$name: $bad of $P processes had a mismatch ($*)" >> $out
  echo "$name done: $bad"
}
trial nopools EVAC_DIAG_NO_POOLS=1
if grep -q "^nopools: 0 of" $out; then echo "good box: nothing to learn" >> $out; exit 0; fi
trial pools EVAC_NOP=1
trial nopools2 EVAC_DIAG_NO_POOLS=1
trial pools2 EVAC_NOP=1
trial nopools3 EVAC_DIAG_NO_POOLS=1
trial pools3 EVAC_NOP=1
