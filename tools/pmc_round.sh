#!/bin/bash
# Counter evidence for one round (every percentage DESIGN.md quotes must be recomputable from profiles/):
#   - in-kernel shader clock of band_kernel under sustained load (tools/clock_probe.py, diagnostic build)
#   - GRBM_GUI_ACTIVE and two SQ counter passes (--pmc with --kernel-trace only, one pass each) for
#     band_kernel<rx_real>, iq1024_kernel, compress_kernel, stream_kernel and sinc5_kernel
# usage (on the GPU box): bash tools/pmc_round.sh <tag>   -> gpurun_out/pmc_<tag>.json (+ clock_<tag>.json)
tag="${1:-rXX}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcr_$tag
rm -rf $out && mkdir -p $out
if [ -z "$PMC_NO_CLOCK" ]; then
python3 tools/clock_probe.py 20 2.5 > $out/clock_random.json 2> $out/clock.err && cat $out/clock_random.json
python3 tools/clock_probe.py 20 2.5 zeros > $out/clock_zeros.json 2>> $out/clock.err && cat $out/clock_zeros.json
fi
SQ_A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
SQ_B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM"
SQ_C="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES"
GR="GRBM_GUI_ACTIVE"
# keep only the counters this rocprofv3 knows (an unknown name fails the whole pass)
rocprofv3 -L > $out/counter_list.txt 2>&1
filter() { local o=""; for c in $1; do grep -qw "$c" $out/counter_list.txt && o="$o $c"; done; echo $o; }
SQ_A=$(filter "$SQ_A"); SQ_B=$(filter "$SQ_B"); SQ_C=$(filter "$SQ_C"); GR=$(filter "$GR")
echo "SQ_A: $SQ_A"; echo "SQ_B: $SQ_B"; echo "SQ_C: $SQ_C"; echo "GRBM: $GR"
run_pass() {  # <name> <counters> <program args...>
  local name="$1" ctr="$2"; shift 2
  rm -rf $out/raw
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/raw -- python3 "$@" > $out/$name.log 2>&1
  local f=$(find $out/raw -name "*counter_collection.csv" | head -1)
  local k=$(find $out/raw -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/$name.counters.csv
  [ -n "$k" ] && cp "$k" $out/$name.trace.csv
  rm -rf $out/raw
  echo "pass $name: $( [ -n "$f" ] && echo ok || echo NO DATA )"
}
for target in ${PMC_TARGETS:-band iq1024 compress stream sinc5}; do
  case $target in
    band)     export UC_VARIANT=0; unset UC_N; prog="tools/run_band.py 20 3" ;;
    iq1024)   export UC_VARIANT=4 UC_N=1024; prog="tools/run_band.py 20 3" ;;
    compress) export UC_VARIANT=2; unset UC_N; prog="tools/run_band.py 20 3" ;;
    stream)   unset UC_VARIANT UC_N; prog="bench.py --variant stream --steps 5 --warmup 2" ;;
    sinc5)    unset UC_VARIANT UC_N; prog="tools/run_cic.py 28 3" ;;
  esac
  run_pass ${target}_sqa "$SQ_A" $prog
  run_pass ${target}_sqb "$SQ_B" $prog
  run_pass ${target}_sqc "$SQ_C" $prog
  run_pass ${target}_grbm "$GR" $prog
done
unset UC_VARIANT UC_N
python3 tools/pmc_summary.py $out > gpurun_out/pmc_$tag.json && cat gpurun_out/pmc_$tag.json | head -c 3000
cp $out/clock_random.json gpurun_out/clock_${tag}_random.json 2>/dev/null
cp $out/clock_zeros.json gpurun_out/clock_${tag}_zeros.json 2>/dev/null
rm -f $out/*.trace.csv $out/*.counters.csv
