#!/usr/bin/env python3
"""Per-launch kernel time of the bench workload from the very first launch of a fresh process (is there a ramp?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
import torch, uchirp
from uchirp import synth
dev = torch.device("cuda:0")
frames, _ = synth.device_frames(1 << 20, dev, seed=1234, snr_db=-10.0)
e = uchirp.Engine(uchirp.RX_REAL, mag_mean=1000.0)
sym = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
torch.cuda.synchronize()
for a, b in ev:
    a.record(st); e.process(frames, want_stats=False, symbols_out=sym, stream=st.cuda_stream); b.record(st)
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in ev]
for lo in (0, 1, 2, 3, 5, 10, 20, 40, 80, 120, 160):
    hi = min(n, lo + (1 if lo < 5 else 10))
    if lo < n:
        print("launch %4d..%4d: %.4f ms" % (lo, hi - 1, sum(ms[lo:hi]) / (hi - lo)))
for lo in range(200, n, 200):
    hi = min(n, lo + 200)
    print("launch %4d..%4d: %.4f ms   (%.2f s after the first launch)" % (lo, hi - 1, sum(ms[lo:hi]) / (hi - lo), sum(ms[:lo]) / 1e3))
