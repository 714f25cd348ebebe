#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one --pmc pass each, kernel trace only) of the sibling kernels' bench side
# measurements.  usage (GPU box): bash tools/pmc_traffic.sh "iq1024 compress dechirp_down stream"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in ${1:-iq1024 compress dechirp_down stream}; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/pmct_${v}_$ctr
    rm -rf $d
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -- python3 bench.py --variant $v --steps 5 --warmup 2 > $d.json 2> $d.err
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$v" "$ctr" "$d.json" <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
key = {"iq1024": "iq1024_kernel", "iq": "iq_kernel", "compress": "compress_kernel", "dechirp_down": "band_kernel", "sync_cplx": "band_kernel", "stream": "stream_kernel"}[sys.argv[2]]
v = [float(r["Counter_Value"]) for r in rows if key in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[3]]
d = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
units = d.get("frames") or d["value"] * d["ms_per_step"] * 1e-3
mult = 2.0 if sys.argv[3] == "FETCH_SIZE" else 1.0   # gfx950: FETCH_SIZE counts 64 B per 128-B request
per = sum(v) / len(v) * 1024 * mult / units
alg = d["roofline"].get("bytes_per_frame") or d["roofline"].get("bytes_per_sample")
print("%-13s %-10s %d dispatches  raw avg %.6g KB  -> %.2f B per unit (%s, x%g)   algorithmic total %.1f B per unit" % (sys.argv[2], sys.argv[3], len(v), sum(v) / len(v), per, "frame" if d.get("frames") else "sample", mult, alg))
PY
    rm -rf $d
  done
done
