#!/usr/bin/env python3
"""The clock the chip holds under band_kernel (MI355X_MICROARCH.md, 'DVFS give-back' item 6): the clock-stamped twin
inside libuchirp.so (uc_clock_probe: ONE s_memtime / s_memrealtime stamp
pair around the persistent loop), >= 2 s of back-to-back launches on random data, then
clock = d(s_memtime) / d(s_memrealtime) x 100 MHz of the LAST launch, median over workgroups.
Usage: python tools/clock_probe.py [frames_log2=20] [seconds=2.5] [zeros]   env UC_VARIANT, UC_BAND_WAVES"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
import numpy as np
import torch
import uchirp
from uchirp import synth

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
zeros = len(sys.argv) > 3 and sys.argv[3] == "zeros"
nf = 1 << lg
dev = torch.device("cuda:0")
frames, _ = synth.device_frames(nf, dev, seed=1)
if zeros:
    frames.zero_()
sym = torch.empty(nf, dtype=torch.uint8, device=dev)
variant = int(os.environ.get("UC_VARIANT", "0"))
e = uchirp.Engine(variant, mag_mean=1000.0)
e.clock_probe(True)
stream = torch.cuda.current_stream(dev)
want_sym = variant in (0, 1)
st = None if want_sym else torch.empty((nf, e.spf, 8), dtype=torch.float32, device=dev)


def launch():
    e.process(frames, want_symbols=want_sym, want_stats=not want_sym, symbols_out=sym if want_sym else None, stats_out=st,
              stream=stream.cuda_stream)


launch()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(50):
        launch()
    torch.cuda.synchronize()
    n += 50
wall = (time.perf_counter() - t0) / n
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(stream)
launch()
b.record(stream)
torch.cuda.synchronize()
d = e.clock_stamps().astype(np.int64)
d = d[d[:, 1] > 0]
d[:, 0] &= (1 << 40) - 1                      # (bits 40..51 of the cycle word name the CU: tools/run_target.py)
d = d.astype(np.float64)
t_first, t_last = d[:, 2].min(), d[:, 3].max()
start_us, end_us, life_us = (d[:, 2] - t_first) / 100.0, (d[:, 3] - t_first) / 100.0, d[:, 1] / 100.0
pct = lambda v: [float(np.percentile(v, q)) for q in (0, 10, 50, 90, 100)]
clk = d[:, 0] / d[:, 1] * 100.0  # MHz: s_memrealtime ticks at 100 MHz
out = {"kernel": "band_kernel variant %d (%s data)" % (variant, "zero" if zeros else "random"), "frames": nf,
       "launches_before_stamp": n, "seconds_of_back_to_back_launches": secs, "ms_per_launch_wall": wall * 1e3,
       "ms_last_launch_events": a.elapsed_time(b), "waves_stamped": int(d.shape[0]),
       "shader_clock_MHz_median": float(np.median(clk)), "shader_clock_MHz_p10": float(np.percentile(clk, 10)),
       "shader_clock_MHz_p90": float(np.percentile(clk, 90)),
       "loop_cycles_median": float(np.median(d[:, 0])), "loop_us_median": float(np.median(d[:, 1]) / 100.0),
       "frames_per_s": nf / wall,
       "grid_span_us_first_start_to_last_end": float((t_last - t_first) / 100.0),
       "wave_start_us_p0_10_50_90_100": pct(start_us), "wave_end_us_p0_10_50_90_100": pct(end_us),
       "wave_life_us_p0_10_50_90_100": pct(life_us)}
print(json.dumps(out))
