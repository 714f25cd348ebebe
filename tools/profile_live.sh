#!/bin/bash
# Per-kernel times, launches per call and waves per launch of the LIVE receiver (uc_receive_streams_next, one new block of
# every stream per call, counts read back after every call):
#   bash tools/profile_live.sh <tag> [streams=65536] [variant=rx_real]
#     -> gpurun_out/live_<tag>_<variant>_<streams>.txt  (the tool's JSON line, kernel stats, launches per call, SQ_WAVES per launch)
# rocprofv3 --kernel-trace --stats for the times; a separate --pmc SQ_WAVES pass for the waves (never both with other traces).
set -e
tag="${1:-rXX}"; ns="${2:-65536}"; v="${3:-rx_real}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out="gpurun_out/live_${tag}_${v}_${ns}"
mkdir -p "$out.d"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out.d/t" -- python3 tools/run_receive_live.py "$ns" 1 "$v" > "$out.json" 2> "$out.err"
rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d "$out.d/p" -- python3 tools/run_receive_live.py "$ns" 1 "$v" > /dev/null 2>> "$out.err"
python3 - "$out" "$(find "$out.d/t" -name '*kernel_stats.csv' | head -1)" "$(find "$out.d/p" -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, json, sys
out, stats, ctr = sys.argv[1:4]
line = json.loads(open(out + ".json").read().strip().splitlines()[-1])
calls = line["calls"]
waves = {}
for r in csv.DictReader(open(ctr)):
    if "uc::" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_WAVES":
        waves.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
with open(out + ".txt", "w") as f:
    f.write(json.dumps(line) + "\n")
    f.write("kernels of the library in this run (%d calls of uc_receive_streams_next):\n" % calls)
    tot = 0.0
    for r in csv.DictReader(open(stats)):
        if "uc::" not in r["Name"]:
            continue
        n, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
        w = waves.get(r["Name"])
        per_call = n / calls
        if per_call >= 0.5:
            tot += avg * per_call
        f.write("  %-100s launches %5d = %.2f per call, average %.1f us%s\n"
                % (r["Name"][:100], n, per_call, avg, (", SQ_WAVES %.0f per launch" % (sum(w) / len(w))) if w else ""))
    f.write("kernel time per call (launches per call x average): %.1f us; the call incl. the host's read-back: %.1f us\n"
            % (tot, line["ms_per_call"] * 1e3))
print(open(out + ".txt").read())
PY
rm -rf "$out.d"
