#!/bin/bash
# round 6, first GPU pass: the receive tests on the new ROWS walk, then the live-step timings (default and keep-previous)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_receive_many.py -x -q -m gpu > gpurun_out/r6a_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r6a_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for v in rx_real sync_cplx; do
  for keep in 0 1; do
    timeout -k 10 200 python tools/run_live_async.py 4096,65536 $v 100 $keep >> gpurun_out/r6a_live_async.jsonl 2>gpurun_out/r6a_err.log || exit 1
  done
done
cat gpurun_out/r6a_live_async.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['variant'], d['streams'], 'keep' if d['keep_previous'] else 'save', 'eager %.4f graph %.4f ms' % (d['eager_ms_per_call'], d['graph_ms_per_call']), d.get('smu_clock_MHz'))
"
