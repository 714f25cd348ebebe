#!/bin/bash
# round 6: group size of the masked live steps (UC_BAND_GROUP 8 / 16 / 32), 65 536 and 4096 streams, keep and save
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6_group_ab.txt
mkdir -p gpurun_out; : > $out
export UC_TUNING=1 UC_LIVE_SUSTAIN_S=0.3
for rep in 1 2; do
for g in 32 16 8; do
  export UC_BAND_GROUP=$g
  for v in rx_real sync_cplx; do
    timeout -k 10 120 python3 tools/run_live_async.py 4096,65536 $v 100 1 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('group $g', '$v', d['streams'], 'keep' if d['keep_previous'] else 'save', 'eager %.4f graph %.4f ms' % (d['eager_ms_per_call'], d['graph_ms_per_call']), 'clock', d.get('smu_clock_MHz'))
" >> $out || exit 1
  done
done
done
cat $out
