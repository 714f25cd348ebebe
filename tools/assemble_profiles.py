#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) and, if present, gpurun_out/pmcall_<round>/ (tools/pmc_all.sh)
into the committed summaries profiles/<tag>_*.
Usage: python tools/assemble_profiles.py r03_v1 [pmc round tag, e.g. r03]"""
import csv
import json
import os
import sys

tag = sys.argv[1]
src, dst = "gpurun_out/prof_%s" % tag, "profiles"
variants = ("sync_cplx", "compress", "dechirp_down", "iq", "iq_bb", "iq1024", "iq1024_bb", "stream")
open("%s/%s_bench.json" % (dst, tag), "w").write(open("%s/bench.json" % src).read().strip().splitlines()[-1] + "\n")
rows = list(csv.reader(open("%s/bench_kernel_stats.csv" % src)))
with open("%s/%s_kernel_stats.csv" % (dst, tag), "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(rows[0])
    for i, r in enumerate(rows[1:]):   # the 15 longest, and every kernel of the library or of bench.py's probes however short
        if i < 15 or "uc::" in r[0] or "anonymous namespace" in r[0]:
            w.writerow(r)
if os.path.exists("%s/headline_kernel_stats.csv" % src):   # the contract leg alone (tools/profile_round.sh)
    hr = list(csv.reader(open("%s/headline_kernel_stats.csv" % src)))
    with open("%s/%s_headline_kernel_stats.csv" % (dst, tag), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(hr[0])
        for r in hr[1:8]:
            w.writerow(r)
    open("%s/%s_headline_bench.json" % (dst, tag), "w").write(open("%s/headline.json" % src).read().strip().splitlines()[-1] + "\n")
    hb = json.loads(open("%s/%s_headline_bench.json" % (dst, tag)).read())
    print("headline alone: rocprofv3 average of %s = %.4f ms over %s launches (incl. ramp + warm-up); bench kernel_ms (HIP events, "
          "timed steps) = %.4f ms" % (hr[1][0][:60], float(hr[1][3]) / 1e6, hr[1][1], hb["roofline"]["kernel_ms"]))
with open("%s/%s_variants_kernel_stats.csv" % (dst, tag), "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(["variant"] + rows[0])
    for v in variants + ("hello",):
        p = "%s/%s_kernel_stats.csv" % (src, v)
        if not os.path.exists(p):
            continue
        for r in list(csv.reader(open(p)))[1:]:
            if "uc::" in r[0] or "ccl" in r[0].lower():
                w.writerow([v] + r)
with open("%s/%s_variants_bench.jsonl" % (dst, tag), "w") as f:
    for v in variants + ("hello",):
        p = "%s/%s.json" % (src, v)
        if os.path.exists(p) and open(p).read().strip():
            f.write(open(p).read().strip().splitlines()[-1] + "\n")
if len(sys.argv) > 2:
    pm = "gpurun_out/pmcall_%s" % sys.argv[2]
    for name in ("%s_pmc_all.json" % sys.argv[2], "%s_valu_insts.json" % sys.argv[2]):
        if os.path.exists(os.path.join(pm, name)):   # per target: a partial counter run updates, never drops entries
            new = json.load(open(os.path.join(pm, name)))
            old = json.load(open(os.path.join(dst, name))) if os.path.exists(os.path.join(dst, name)) else {}
            old.update(new)
            json.dump(old, open(os.path.join(dst, name), "w"), indent=1)
    # the band kernel's own traffic record, in the form bench.py reads (rNN_vM_hbm_traffic.json)
    allp = os.path.join(dst, "%s_pmc_all.json" % sys.argv[2])
    if os.path.exists(allp):
        d = json.load(open(allp))["band_rx_real_f32"]
        c, e = d["counters_per_dispatch"], d["derived"]
        json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 tools/run_target.py "
                              "band_rx_real_f32 --frames-log2 19 --iters 3  (tools/pmc_all.sh)",
                   "kernel": "band_kernel<0,1,3> (rx_real, f32), %s" % tag, "frames_per_launch": d["units_per_dispatch"],
                   "FETCH_SIZE_KB_raw": c["FETCH_SIZE"], "WRITE_SIZE_KB_raw": c["WRITE_SIZE"], "correction": e["correction"],
                   "hbm_read_bytes": e["hbm_read_bytes_per_unit"] * d["units_per_dispatch"],
                   "hbm_write_bytes": e["hbm_write_bytes_per_unit"] * d["units_per_dispatch"],
                   "hbm_bytes_per_frame": e["hbm_bytes_per_unit"], "algorithmic_bytes_per_frame": 8193},
                  open("%s/%s_hbm_traffic.json" % (dst, tag), "w"), indent=1)
b = json.loads(open("%s/%s_bench.json" % (dst, tag)).read())
print("bench: %.4g frames/s, frac %.3f (%s-bound, valu %.3f), kernel %.3f ms, cpu %.3g frames/s on %d threads"
      % (b["value"], b["roofline"]["frac"], b["roofline"].get("limiter", b["roofline"]["bound"]), b["roofline"].get("valu", {}).get("frac", float("nan")),
         b["roofline"]["kernel_ms"], b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"]))
c2, c3 = b["configs"]["configs[2]"], b["configs"]["configs[3]"]
print("  configs[2] base band %.4g frames/s (%.3f of firmware windows %.4g)" % (c2["baseband"]["value"], c2["baseband_over_firmware_windows"], c2["firmware_windows"]["value"]))
print("  configs[3] graph %.4g samples/s, eager %.4g" % (c3["graph_replay"]["value"], c3["eager"]["value"]))
print("  hello_world1 %.4g frames/s (%.4f of configs[1])" % (b["hello_world1"]["value"], b["hello_world1"]["over_configs1_value"]))
for l in open("%s/%s_variants_bench.jsonl" % (dst, tag)):
    d = json.loads(l)
    print("  %-70s %.4g" % (d["metric"][:70], d["value"]))
