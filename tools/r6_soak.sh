#!/bin/bash
# round 6: the long live runs behind the rebuilt walk, under the poison switch (every passed-over statistic is HUGE):
# soaks of 3000 blocks with thousands of transmissions (eager, graph, kept chunks eager / two graphs, random chunks with dropped
# blocks), 600 random shapes, 65 536 streams.   -> gpurun_out/r6_soak_live.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r6_soak_live.txt
mkdir -p gpurun_out; : > $out
export UC_TUNING=1 UC_RX_POISON=1
echo "UC_TUNING=1 UC_RX_POISON=1 (passed-over statistics are 1e15 instead of zero)" >> $out
timeout -k 10 400 python3 tools/soak_live.py sync_cplx 3000 64 >> $out 2>&1 || { echo "SOAK FAILED" >> $out; tail -5 $out; exit 1; }
timeout -k 10 400 python3 tools/soak_live.py rx_real 3000 48 >> $out 2>&1 || { echo "SOAK FAILED" >> $out; tail -5 $out; exit 1; }
timeout -k 10 500 python3 tools/fuzz_live.py 600 2026 >> $out 2>&1 || { echo "FUZZ FAILED" >> $out; tail -5 $out; exit 1; }
timeout -k 10 300 python3 tools/scale_live.py >> $out 2>&1 || { echo "SCALE FAILED" >> $out; tail -5 $out; exit 1; }
tail -40 $out
