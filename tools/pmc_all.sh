#!/bin/bash
# Counter evidence for EVERY kernel of the library (each fraction recomputable from profiles/):
#   per target: two SQ counter passes, FETCH_SIZE and WRITE_SIZE (separate --pmc passes, --kernel-trace only: gpurun
#   refuses --pmc together with the other trace domains), and the in-kernel clock of the clock-stamp build.
# usage (on the GPU box): bash tools/pmc_all.sh <tag> ["target ..."]   -> gpurun_out/pmcall_<tag>/ (+ summary JSONs)
# Never wrap the program: rocprofv3 ... -- python3 <script> directly.
tag="${1:-rXX}"
targets="${2:-band_rx_real_f32 band_sync_cplx_f32 band_dechirp_down_f32 compress_f32 iq2048_fw_f32 iq2048_bb_f32 iq1024_fw_f32 iq1024_bb_f32 stream_d8_f32 sinc5 sinc5_streams rows_rx_real_f32 rows_sync_cplx_f32}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out="gpurun_out/pmcall_$tag"
mkdir -p "$out"
# (every gpurun call starts on a fresh box: seed the summaries from the committed record, so that a run over SOME targets
# updates it instead of replacing it -- round 6 lost a first pass that way)
for f in "${tag}_pmc_all.json" "${tag}_valu_insts.json"; do [ -f "profiles/$f" ] && [ ! -f "$out/$f" ] && cp "profiles/$f" "$out/$f"; done
SQ_A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
SQ_B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM"
rocprofv3 -L > "$out/counter_list.txt" 2>&1
filter() { local o=""; for c in $1; do grep -qw "$c" "$out/counter_list.txt" && o="$o $c"; done; echo $o; }
SQ_A=$(filter "$SQ_A"); SQ_B=$(filter "$SQ_B")
run_pass() {  # <target> <pass name> <counters>
  local t="$1" name="$2" ctr="$3"
  local raw="$out/raw_pass"
  mkdir -p "$raw"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$raw" -- python3 tools/run_target.py $t --frames-log2 19 --iters 3 > "$out/$t.$name.log" 2>&1
  local f=$(find "$raw" -name "*counter_collection.csv" | head -1)
  local k=$(find "$raw" -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/$t.$name.counters.csv"
  [ -n "$k" ] && cp "$k" "$out/$t.$name.trace.csv"
  find "$raw" -type f -delete
  echo "$t $name: $( [ -n "$f" ] && echo ok || echo NO DATA )"
}
for t in $targets; do
  python3 tools/run_target.py $t --frames-log2 19 --info > "$out/$t.info.json"
  run_pass $t sqa "$SQ_A"
  run_pass $t sqb "$SQ_B"
  run_pass $t fetch "FETCH_SIZE"
  run_pass $t write "WRITE_SIZE"
  if true; then
    timeout -k 10 120 python3 tools/run_target.py $t --frames-log2 20 --clock --seconds 2.0 > "$out/$t.clock.json" 2> "$out/$t.clock.err" && echo "$t clock: $(python3 -c "import json;d=json.load(open('$out/$t.clock.json'));print(d['shader_clock_MHz_median'], d['ms_last_launch_events'])")"
  fi
done
python3 tools/valu_table.py "$out" "$tag"
find "$out" -name "*.trace.csv" -delete
find "$out" -name "*.counters.csv" -delete
find "$out" -name "*.log" -delete
ls "$out" | head -60
