#!/bin/bash
# A/B of two (or more) builds of libuchirp.so on ONE box, alternating: bench.py's rate per variant and library.
# usage (GPU box): bash tools/lib_ab.sh "libuchirp_base.so libuchirp.so" "rx_real stream iq_bb" [reps=3]
# (library names relative to ultrasonic-communication_amd/; rx_real = the contract line without its side blocks)
libs="$1"; variants="$2"; reps="${3:-3}"
cd "$(dirname "$0")/.."
for v in $variants; do
  for rep in $(seq $reps); do
    for L in $libs; do
      if [ "$v" = rx_real ]; then args="--no-configs --no-hello1 --no-cpu-baseline --no-receive --no-live-traffic --sustain-s 0"; else args="--variant $v"; fi
      UCHIRP_LIB=$PWD/ultrasonic-communication_amd/$L python3 bench.py $args 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-14s %-24s %.4g %s  %.4f ms' % ('$v', '$L', d['value'], d['unit'], d.get('ms_per_step', 0)), flush=True)"
    done
  done
done
