#!/bin/bash
# Per-kernel times of the LIVE receiver (uc_receive_streams_next, one new block of every stream per call):
#   bash tools/profile_live.sh <tag> [streams=65536] [variant=rx_real]  -> gpurun_out/prof_live_<tag>_<variant>_<streams>_kernel_stats.csv
# (rocprofv3 --kernel-trace --stats; 176 calls: launches per call = Calls / 176)
set -e
tag="${1:-rXX}"; ns="${2:-65536}"; v="${3:-rx_real}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out="gpurun_out/prof_live_${tag}_${v}_${ns}"
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 tools/run_receive_live.py "$ns" 1 "$v" > "$out.json" 2> "$out.err"
f="$(find "$out" -name '*kernel_stats.csv' | head -1)"
test -n "$f" && cp "$f" "${out}_kernel_stats.csv"
rm -rf "$out"
cat "$out.json"; cut -c1-160 "${out}_kernel_stats.csv"
