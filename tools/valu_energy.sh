#!/bin/bash
# Joules per wave-instruction by instruction type (tools/valu_energy_probe.hip): socket power (rocm-smi, every 0.4 s) while
# the whole chip issues one instruction type.  usage (GPU box): bash tools/valu_energy.sh [types="0 1 2 3 4 5 6 7 8"]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for f in ${1:-0 1 2 3 4 5 6 7 8}; do
  ( tools/valu_energy_probe $f > gpurun_out/ve_$f.log 2>&1 & echo $! > gpurun_out/ve.pid )
  sleep 0.3
  pid=$(cat gpurun_out/ve.pid)
  : > gpurun_out/ve_$f.smi
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed "s/.*: //" | tr -d "()MhzW" | tr "\n" " " >> gpurun_out/ve_$f.smi
    echo >> gpurun_out/ve_$f.smi
    sleep 0.4
  done
  python3 - $f <<'PY'
import sys, re, statistics
f = sys.argv[1]
rows = [l.split() for l in open("gpurun_out/ve_%s.smi" % f) if len(l.split()) >= 2]
busy = [(float(a), float(b)) for a, b in (r[:2] for r in rows) if float(a) > 1000]
log = open("gpurun_out/ve_%s.log" % f).read().strip()
m = re.search(r"= ([0-9.e+]+) per s", log)
if busy and m:
    p = statistics.median(b for _, b in busy)
    rate = float(m.group(1))
    clk = statistics.median(a for a, _ in busy)
    print("%s | sclk %.0f MHz, socket %.0f W, (W - 240) / rate = %.3f nJ per wave-instruction, %.2f cycles per instruction and SIMD"
          % (log, clk, p, (p - 240.0) / rate * 1e9, clk * 1e6 * 1024 / rate))
else:
    print(log, "| no busy samples")
PY
done
