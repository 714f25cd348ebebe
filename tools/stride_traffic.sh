#!/bin/bash
# HBM traffic (FETCH_SIZE x 2 on gfx950) of the headline kernel at stride 2048 and at stride 256, separate --pmc passes:
#   bash tools/stride_traffic.sh  -> bytes fetched per frame of each leg (tools/run_stride.py 19 0.2 <leg>)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for leg in 2048 256; do
  d=gpurun_out/stride_pmc_$leg
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d -- python3 tools/run_stride.py 19 0.2 $leg > /dev/null 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$leg" <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "band_kernel" in r["Kernel_Name"]]
print("stride %s: FETCH_SIZE x 2 = %.0f B per frame (mean of %d launches of 2^19 frames)" % (sys.argv[2], sum(v) / len(v) * 1024 * 2 / (1 << 19), len(v)))
PY
  rm -rf $d
done
