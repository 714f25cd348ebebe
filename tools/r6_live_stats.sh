#!/bin/bash
# per-kernel times of the live step at 4096 and 16384 streams (kept chunks): what a fused replay could save
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in rx_real sync_cplx; do
for ns in 4096 16384; do
  rm -rf /tmp/lp; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -- python3 tools/run_live_async.py $ns $v 100 1 > /tmp/lp.json 2>/dev/null
  echo "== $v $ns streams, kept chunks: $(python3 -c "import json;d=json.loads(open('/tmp/lp.json').read().strip().splitlines()[-1]);print('eager %.4f graph %.4f ms per call' % (d['eager_ms_per_call'], d['graph_ms_per_call']))")"
  python3 - "$(find /tmp/lp -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'uc::' in r['Name']: print('   %-80s calls %5s avg %.1f us' % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
done
