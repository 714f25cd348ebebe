#!/bin/bash
# usage: tools/pmc.sh "<counter list>" [frames_log2] [iters] ; prints per-dispatch averages of the band kernel
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set_="$1"; lg="${2:-19}"; it="${3:-3}"
n=$(echo $set_ | tr " " "_" | cut -c1-40)
rm -rf gpurun_out/pmc_$n
timeout 300 rocprofv3 --kernel-trace --pmc $set_ --output-format csv -d gpurun_out/pmc_$n -- python3 tools/run_band.py $lg $it > /dev/null 2>&1 || true
f=$(find gpurun_out/pmc_$n -name "*counter_collection.csv" | head -1)
python3 - "$f" <<PY
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(list)
for r in rows:
    if "band_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print("%-28s n=%d avg=%.6g" % (k, len(v), sum(v)/len(v)))
PY
