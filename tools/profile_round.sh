#!/bin/bash
# One gpurun call that produces the per-round kernel-time evidence profiles/ holds:
#   - rocprofv3 --kernel-trace --stats of the default bench.py run (the driver's command): bench JSON line +
#     kernel_stats.csv -- since round 3 that one run holds band_kernel<rx_real> (configs[1] and hello_world1),
#     iq1024_kernel in both modes (configs[2]), stream_kernel eager and graph-replayed (configs[3]) and RCCL's kernels
#   - kernel stats of the sibling variants' side measurements
# Counters (SQ, FETCH/WRITE_SIZE) and in-kernel clocks of every kernel: tools/pmc_all.sh (separate --pmc passes).
# usage (on the GPU box): bash tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/
#
# NEVER profile `bench.py --gpus N` with N > 1: rocprofv3's preloaded library initialises the GPU in the parent, which
# then starts N child processes (and a GPU-initialised process must not fork/exec on this pool).  To profile the N > 1
# leg, profile ONE rank with the program directly behind `--`:
#   RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 UC_BENCH_HELLO=1 rocprofv3 --kernel-trace --stats ... -- python3 bench.py
set -e
tag="${1:-rXX}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out="gpurun_out/prof_$tag"
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench" -- python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
cp "$(find "$out/bench" -name '*kernel_stats.csv' | head -1)" "$out/bench_kernel_stats.csv"
echo "bench done"; head -c 400 "$out/bench.json"; echo
# the contract leg ALONE (configs[1]; no side legs: since round 4 the receive leg launches the same band kernel over other
# batch sizes, which would mix into its average): the rocprofv3 average of band_kernel<0,1,3> here is what bench.py's
# roofline.kernel_ms (HIP events) must agree with
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/headline" -- python3 bench.py --no-configs --no-hello1 --no-receive --no-cpu-baseline > "$out/headline.json" 2> "$out/headline.err"
cp "$(find "$out/headline" -name '*kernel_stats.csv' | head -1)" "$out/headline_kernel_stats.csv"
for v in sync_cplx compress dechirp_down iq iq_bb iq1024 iq1024_bb stream; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$v" -- python3 bench.py --variant $v > "$out/$v.json" 2> "$out/$v.err"
  cp "$(find "$out/$v" -name '*kernel_stats.csv' | head -1)" "$out/${v}_kernel_stats.csv"
  echo "$v done"
done
UC_BENCH_HELLO=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/hello" -- python3 bench.py > "$out/hello.json" 2> "$out/hello.err"
cp "$(find "$out/hello" -name '*kernel_stats.csv' | head -1)" "$out/hello_kernel_stats.csv"
# keep the merge small: drop the raw traces
find "$out" -mindepth 2 -type f -delete
find "$out" -mindepth 1 -type d -empty -delete
ls -la "$out"
