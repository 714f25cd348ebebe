#!/bin/bash
# Joules per byte of the load flavours of tools/load_energy_probe.hip: socket power (rocm-smi, every 0.4 s) while each
# flavour streams 8 GiB for ~3 s.  usage (GPU box): bash tools/load_energy.sh [flavours="0 1 2 3 4 5"]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for f in ${1:-0 1 2 3 4 5}; do
  ( tools/load_energy_probe $f 8 2200 > gpurun_out/le_$f.log 2>&1 & echo $! > gpurun_out/le.pid )
  sleep 0.3
  pid=$(cat gpurun_out/le.pid)
  : > gpurun_out/le_$f.smi
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed "s/.*: //" | tr -d "()MhzW" | tr "\n" " " >> gpurun_out/le_$f.smi
    echo >> gpurun_out/le_$f.smi
    sleep 0.4
  done
  python3 - $f <<'PY'
import sys, re, statistics
f = sys.argv[1]
rows = [l.split() for l in open("gpurun_out/le_%s.smi" % f) if len(l.split()) >= 2]
busy = [(float(a), float(b)) for a, b in (r[:2] for r in rows) if float(a) > 1000]
log = open("gpurun_out/le_%s.log" % f).read().strip()
m = re.search(r"= ([0-9.]+) TB/s", log)
if busy and m:
    p = statistics.median(b for _, b in busy)
    tbs = float(m.group(1))
    print("%s | sclk %.0f MHz, socket %.0f W, (W - 240) / rate = %.3f nJ/B" % (log, statistics.median(a for a, _ in busy), p, (p - 240.0) / (tbs * 1e12) * 1e9))
else:
    print(log, "| no busy samples")
PY
done
