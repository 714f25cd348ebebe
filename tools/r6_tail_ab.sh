#!/bin/bash
# round 6: balanced tail of the static deal (ROWS build) -- library A (round-robin groups only) against B (tail shared out in units),
# alternated on one box; kept chunks; ms per call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6_tail_ab.txt
mkdir -p gpurun_out; : > $out
export UC_LIVE_SUSTAIN_S=0.3
for rep in 1 2 3; do
for lib in notail tail; do
  export UCHIRP_LIB=$GRAFT_REPO_ROOT/ultrasonic-communication_amd/libuchirp_ab_$lib.so
  for v in rx_real sync_cplx; do
    timeout -k 10 120 python3 tools/run_live_async.py 4096,16384,65536 $v 100 1 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$lib', '$v', d['streams'], 'eager %.4f graph %.4f ms' % (d['eager_ms_per_call'], d['graph_ms_per_call']), 'clock', d.get('smu_clock_MHz'))
" >> $out || exit 1
  done
done
done
cat $out
