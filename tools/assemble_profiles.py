#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the committed summaries profiles/<tag>_*.
Usage: python tools/assemble_profiles.py r01_v8"""
import csv
import json
import sys

tag = sys.argv[1]
src, dst = "gpurun_out/prof_%s" % tag, "profiles"
variants = ("iq1024", "iq", "compress", "dechirp_down", "sync_cplx", "stream")
open("%s/%s_bench.json" % (dst, tag), "w").write(open("%s/bench.json" % src).read().strip().splitlines()[-1] + "\n")
rows = list(csv.reader(open("%s/bench_kernel_stats.csv" % src)))
with open("%s/%s_kernel_stats.csv" % (dst, tag), "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(rows[0])
    for r in rows[1:9]:
        w.writerow(r)
fetch = float(open("%s/pmc_fetch.txt" % src).read().split("avg=")[1])
write = float(open("%s/pmc_write.txt" % src).read().split("avg=")[1])
nf = 1 << 20
rd, wr = fetch * 1024 * 2, write * 1024
json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 tools/run_band.py 20 3"
                      "  (tools/pmc.sh, tools/profile_round.sh)",
           "kernel": "band_kernel<0,1,3> (rx_real, f32), %s" % tag, "frames_per_launch": nf,
           "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
           "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_frame": (rd + wr) / nf,
           "algorithmic_bytes_per_frame": 8193}, open("%s/%s_hbm_traffic.json" % (dst, tag), "w"), indent=1)
with open("%s/%s_variants_kernel_stats.csv" % (dst, tag), "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(["variant"] + rows[0])
    for v in variants:
        for r in list(csv.reader(open("%s/%s_kernel_stats.csv" % (src, v))))[1:]:
            if "uc::" in r[0]:
                w.writerow([v] + r)
with open("%s/%s_variants_bench.jsonl" % (dst, tag), "w") as f:
    for v in variants:
        f.write(open("%s/%s.json" % (src, v)).read().strip().splitlines()[-1] + "\n")
b = json.loads(open("%s/%s_bench.json" % (dst, tag)).read())
print("bench: %.4g frames/s, frac %.3f, kernel %.3f ms, cpu %.3g frames/s on %d threads"
      % (b["value"], b["roofline"]["frac"], b["roofline"]["kernel_ms"], b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"]))
for l in open("%s/%s_variants_bench.jsonl" % (dst, tag)):
    d = json.loads(l)
    print("  %-70s %.4g" % (d["metric"][:70], d["value"]))
