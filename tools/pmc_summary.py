#!/usr/bin/env python3
"""Per-dispatch averages of the counter passes of tools/pmc_round.sh, one JSON object per kernel, with the
derived figures DESIGN.md quotes.  Units (MI355X_MICROARCH.md): SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over waves (or SIMDs); GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import csv
import glob
import json
import os
import sys
import collections

src = sys.argv[1]
KEY = {"band": "band_kernel", "iq1024": "iq1024_kernel", "compress": "compress_kernel", "stream": "stream_kernel",
       "sinc5": "sinc5_kernel"}
res = {}
for target, key in KEY.items():
    ctr = collections.defaultdict(list)
    dur = []
    for f in sorted(glob.glob(os.path.join(src, target + "_*.counters.csv"))):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in sorted(glob.glob(os.path.join(src, target + "_*.trace.csv"))):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    if not ctr:
        continue
    avg = {k: sum(v) / len(v) for k, v in sorted(ctr.items())}
    d = {"kernel": key, "dispatches_averaged": {k: len(v) for k, v in sorted(ctr.items())}, "counters_per_dispatch": avg}
    if dur:
        d["ms_per_dispatch_profiled"] = sum(dur) / len(dur)
    der = {}
    g = avg.get
    if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
        pass
    if g("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                  "SQ_WAIT_INST_LDS"):
            if g(k) is not None:
                der[k + "_over_WAVE_CYCLES"] = g(k) / g("SQ_WAVE_CYCLES")
    if g("SQ_LDS_IDX_ACTIVE"):
        der["LDS_BANK_CONFLICT_over_IDX_ACTIVE"] = (g("SQ_LDS_BANK_CONFLICT") or 0.0) / g("SQ_LDS_IDX_ACTIVE")
    if g("GRBM_GUI_ACTIVE") and dur:
        # effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time (the guide: within 3 % of the in-kernel clock
        # on dispatches of >= 10 ms, reads high on short ones)
        der["effective_clock_MHz_GRBM"] = g("GRBM_GUI_ACTIVE") / 8.0 / (d["ms_per_dispatch_profiled"] * 1e-3) / 1e6
    if g("SQ_WAVES"):
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_MFMA"):
            if g(k) is not None:
                der[k + "_per_wave"] = g(k) / g("SQ_WAVES")
    d["derived"] = der
    res[target] = d
print(json.dumps(res, indent=1))
