#!/bin/bash
# Sample the GPU's clocks / power / temperature next to a sustained run of bench.py (what settles over seconds: VERDICT r05 item 2).
# usage (GPU box): tools/clock_watch.sh OUT.txt -- python bench.py ...
out=$1; shift; shift
( while true; do date +%s.%N; rocm-smi --showclocks --showpower --showtemp --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge)|GPU use" ; sleep 0.25; done ) > "$out" &
watch_pid=$!
"$@"
rc=$?
kill $watch_pid
exit $rc
