#!/bin/bash
# One gpurun call that produces everything profiles/ holds for a round:
#   - rocprofv3 --kernel-trace --stats of the default bench.py run (kernel_stats) + its JSON line
#   - FETCH_SIZE and WRITE_SIZE of band_kernel in two separate --pmc passes (kernel trace only)
#   - kernel stats of the sibling variants' side measurements
# usage (on the GPU box): bash tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/
set -e
tag="${1:-rXX}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python3 bench.py > $out/bench.json 2> $out/bench.err
cp "$(find $out/bench -name '*kernel_stats.csv' | head -1)" $out/bench_kernel_stats.csv
echo "bench done"; tail -c 600 $out/bench.json
bash tools/pmc.sh FETCH_SIZE 20 3 > $out/pmc_fetch.txt
bash tools/pmc.sh WRITE_SIZE 20 3 > $out/pmc_write.txt
cat $out/pmc_fetch.txt $out/pmc_write.txt
for v in iq1024 iq compress dechirp_down sync_cplx stream; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$v -- python3 bench.py --variant $v > $out/$v.json 2> $out/$v.err
  cp "$(find $out/$v -name '*kernel_stats.csv' | head -1)" $out/${v}_kernel_stats.csv
  echo "$v done"
done
# keep the merge small: drop the raw traces
find $out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
rm -rf gpurun_out/pmc_*
ls -la $out
