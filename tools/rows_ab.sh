#!/bin/bash
# A/B of builds of libuchirp.so on the LIVE receiver step (the band kernel's ROWS build), one box, alternating:
#   bash tools/rows_ab.sh "libuchirp_ab_x.so libuchirp.so" [streams=4096,65536] [variant=rx_real] [reps=3]
libs="$1"; ns="${2:-4096,65536}"; v="${3:-rx_real}"; reps="${4:-3}"
cd "$(dirname "$0")/.."
for rep in $(seq $reps); do
  for L in $libs; do
    UCHIRP_LIB=$PWD/ultrasonic-communication_amd/$L python3 tools/run_live_async.py $ns $v 100 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    print('%-26s %-10s %6d streams  eager %.4f ms  graph %.4f ms' % ('$L', d['variant'], d['streams'], d['eager_ms_per_call'], d['graph_ms_per_call']), flush=True)"
  done
done
