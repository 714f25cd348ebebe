#!/bin/bash
# round 6: where the time of a masked live step goes -- every stream's need word forced to one value (UC_RX_NEED_FORCE: pricing only,
# results wrong), with the state save (default) and without it (uc_rx_state_keep_previous); 65 536 RX_REAL / SYNC_CPLX streams, ms per call
# back to back.  Then the per-kernel times of the real (unforced) idle step.    -> gpurun_out/r6_live_price.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6_live_price.txt
mkdir -p gpurun_out; : > $out
export UC_TUNING=1 UC_LIVE_SUSTAIN_S=0.3
for v in rx_real sync_cplx; do
  for keep in 0 1; do
    for need in none 0x1ff 0x0ad 0x052 0x080 0x040 0x001 0x000; do
      if [ $need = none ]; then unset UC_RX_NEED_FORCE; else export UC_RX_NEED_FORCE=$need; fi
      timeout -k 10 120 python3 tools/run_live_async.py 65536 $v 60 $keep 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$v', 'keep' if d['keep_previous'] else 'save', 'need $need', 'eager %.4f graph %.4f ms' % (d['eager_ms_per_call'], d['graph_ms_per_call']), 'clock', d.get('smu_clock_MHz'))
" >> $out || exit 1
    done
  done
done
unset UC_RX_NEED_FORCE
for keep in 0 1; do
  rm -rf /tmp/lp; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -- python3 tools/run_live_async.py 65536 rx_real 60 $keep > /dev/null 2>&1
  echo "kernel stats, rx_real 65536 idle, keep=$keep" >> $out
  python3 - "$(find /tmp/lp -name '*kernel_stats.csv' | head -1)" >> $out <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'uc::' in r['Name']: print('   %-90s calls %5s avg %.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
cat $out
