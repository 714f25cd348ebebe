#!/bin/bash
# usage: tools/pmc_any.sh "<counter list>" <kernel-name-substring> <python script> [args...]
# prints per-dispatch averages of the counters for the matching kernel (one --pmc pass, kernel trace only)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set_="$1"; kn="$2"; shift 2
n=$(echo ${set_}_$kn | tr " " "_" | cut -c1-48)
rm -rf gpurun_out/pmc_$n
timeout 300 rocprofv3 --kernel-trace --pmc $set_ --output-format csv -d gpurun_out/pmc_$n -- python3 "$@" > /dev/null 2>&1 || true
f=$(find gpurun_out/pmc_$n -name "*counter_collection.csv" | head -1)
python3 - "$f" "$kn" <<PY
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print("%-28s n=%d avg=%.6g" % (k, len(v), sum(v)/len(v)))
PY
