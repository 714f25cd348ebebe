#!/bin/bash
# usage: tools/power_probe.sh <label> <python script + args...>: runs the command in the background and samples
# sclk / socket power twice a second while it runs (rocm-smi, read-only); prints the median of the busy samples
label="$1"; shift
( python3 "$@" > gpurun_out/probe_$label.log 2>&1 & echo $! > gpurun_out/probe.pid )
sleep 0.2
pid=$(cat gpurun_out/probe.pid)
: > gpurun_out/probe_$label.smi
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed "s/.*: //" | tr -d "()MhzW" | tr "\n" " " >> gpurun_out/probe_$label.smi
  echo >> gpurun_out/probe_$label.smi
  sleep 0.4
done
python3 - "$label" <<'PY'
import sys, statistics
rows=[l.split() for l in open("gpurun_out/probe_%s.smi" % sys.argv[1]) if len(l.split())>=2]
busy=[(float(a),float(b)) for a,b in (r[:2] for r in rows) if float(a)>1000]
if busy:
    print("%-14s busy samples %3d: sclk median %.0f MHz, socket power median %.0f W" % (sys.argv[1], len(busy), statistics.median(a for a,_ in busy), statistics.median(b for _,b in busy)))
else:
    print("%-14s no busy samples" % sys.argv[1])
PY
tail -1 gpurun_out/probe_$label.log | cut -c1-200
