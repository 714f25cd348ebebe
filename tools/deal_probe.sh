#!/bin/bash
# static round-robin deal vs dynamic hand-out of frame groups, at several group sizes (bench.py, no CPU baseline)
cd "$GRAFT_REPO_ROOT"
for cfg in "UC_STATIC_DEAL=1 UC_BAND_GROUP=64" "UC_STATIC_DEAL=0 UC_BAND_GROUP=64" "UC_STATIC_DEAL=0 UC_BAND_GROUP=32" "UC_STATIC_DEAL=0 UC_BAND_GROUP=16" "UC_STATIC_DEAL=0 UC_BAND_GROUP=8"; do
  echo "== $cfg"
  env UC_TUNING=1 $cfg python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g frames/s  kernel %.3f ms  frac %.3f' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
