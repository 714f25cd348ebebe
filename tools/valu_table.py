#!/usr/bin/env python3
"""Summaries of tools/pmc_all.sh: per kernel the SQ counters per dispatch and per unit of work, HBM traffic per unit
(FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md section HBM), the in-kernel clock, and the VALU issue bound
bench.py prints (wave-instructions x 4 cycles / (4 SIMDs x 256 CUs x clock)).
usage: python tools/valu_table.py <gpurun_out/pmcall_tag> <tag>  -> <dir>/<tag>_pmc_all.json, <dir>/<tag>_valu_insts.json
(entries of targets measured in this run replace their old ones; every other entry of an existing summary is kept)"""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
res, table = {}, {}
for info_f in sorted(glob.glob(os.path.join(src, "*.info.json"))):
    info = json.load(open(info_f))
    t, key, units = info["target"], info["kernel"], info["units"]
    ctr, dur = collections.defaultdict(list), []
    skip = int(info.get("skip_first_dispatches", 0))   # (live states: the first step runs on the power-on need word)
    for f in sorted(glob.glob(os.path.join(src, t + ".*.counters.csv"))):
        seen = collections.Counter()
        rows_f = sorted((r for r in csv.DictReader(open(f)) if key in r["Kernel_Name"]), key=lambda r: int(r.get("Dispatch_Id", 0)))
        for r in rows_f:
            seen[r["Counter_Name"]] += 1
            if seen[r["Counter_Name"]] > skip:
                ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in sorted(glob.glob(os.path.join(src, t + ".*.trace.csv"))):
        rows_f = sorted((r for r in csv.DictReader(open(f)) if key in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
        for r in rows_f[skip:]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    if not ctr:
        continue
    avg = {k: sum(v) / len(v) for k, v in sorted(ctr.items())}
    g = avg.get
    d = {"round": tag, "kernel": key, "units_per_dispatch": units, "unit": info["unit"], "alg_bytes_per_unit": info["alg_bytes_per_unit"],
         "dispatches_averaged": {k: len(v) for k, v in sorted(ctr.items())}, "counters_per_dispatch": avg}
    if dur:
        d["ms_per_dispatch_profiled"] = sum(dur) / len(dur)
    der = {}
    if g("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if g(k) is not None:
                der[k + "_over_WAVE_CYCLES"] = g(k) / g("SQ_WAVE_CYCLES")
    if g("SQ_LDS_IDX_ACTIVE"):
        der["LDS_BANK_CONFLICT_over_IDX_ACTIVE"] = (g("SQ_LDS_BANK_CONFLICT") or 0.0) / g("SQ_LDS_IDX_ACTIVE")
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD"):
        if g(k) is not None:
            der[k + "_per_unit"] = g(k) / units
    if g("SQ_INSTS_VALU") and g("SQ_ACTIVE_INST_VALU"):
        der["quad_cycles_per_VALU_inst"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU")
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        rd, wr = g("FETCH_SIZE") * 1024 * 2.0, g("WRITE_SIZE") * 1024
        der["hbm_read_bytes_per_unit"] = rd / units
        der["hbm_write_bytes_per_unit"] = wr / units
        der["hbm_bytes_per_unit"] = (rd + wr) / units
        der["hbm_over_algorithmic"] = (rd + wr) / units / info["alg_bytes_per_unit"]
        der["correction"] = "gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2; WRITE_SIZE exact"
    cf = os.path.join(src, t + ".clock.json")
    clock = None
    if os.path.exists(cf) and os.path.getsize(cf) > 0:
        c = json.load(open(cf))
        d["clock_probe"] = c
        clock = c["shader_clock_MHz_median"] / 1e3
        if g("SQ_INSTS_VALU"):
            per = g("SQ_INSTS_VALU") / units
            # units/s the VALU issue alone allows at this clock: every SIMD issues one VALU wave-instruction per 4 cycles
            der["valu_issue_bound_units_per_s"] = 4 * 256 * clock * 1e9 / (per * 4.0)
            der["valu_frac_at_measured_rate"] = c["units_per_s"] / der["valu_issue_bound_units_per_s"]
            der["hbm_frac_at_measured_rate"] = c["units_per_s"] * info["alg_bytes_per_unit"] / 8e12
    d["derived"] = der
    res[t] = d
    if g("SQ_INSTS_VALU") and clock:
        table[t] = {"valu_insts_per_unit": g("SQ_INSTS_VALU") / units, "clock_GHz": clock,
                    "source": "%s_pmc_all.json (tools/pmc_all.sh: SQ_INSTS_VALU per launch / units; in-kernel clock of the clock-stamp build)" % tag}
        if g("SQ_INSTS_LDS") is not None:
            table[t]["lds_insts_per_unit"] = g("SQ_INSTS_LDS") / units
# a run over a subset of the targets (or a directory whose raw CSVs are gone) UPDATES the summaries, it never empties them
for name, new in ((tag + "_pmc_all.json", res), (tag + "_valu_insts.json", table)):
    path = os.path.join(src, name)
    merged = json.load(open(path)) if os.path.exists(path) and os.path.getsize(path) > 2 else {}
    merged.update(new)
    json.dump(merged, open(path, "w"), indent=1)
for t, d in res.items():
    e = d["derived"]
    print("%-24s VALU/unit %8.1f  LDS/unit %7.1f  hbm/alg %s  clock %s  valu frac %s  hbm frac %s" % (
        t, e.get("SQ_INSTS_VALU_per_unit", float("nan")), e.get("SQ_INSTS_LDS_per_unit", float("nan")),
        "%.3f" % e["hbm_over_algorithmic"] if "hbm_over_algorithmic" in e else "-",
        "%.3f" % table[t]["clock_GHz"] if t in table else "-",
        "%.3f" % e["valu_frac_at_measured_rate"] if "valu_frac_at_measured_rate" in e else "-",
        "%.3f" % e["hbm_frac_at_measured_rate"] if "hbm_frac_at_measured_rate" in e else "-"))
