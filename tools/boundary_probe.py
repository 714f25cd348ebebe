#!/usr/bin/env python3
"""HISTORICAL (round 3): needs that round's separate diagnostic build libuchirp_clock.so with one stamp buffer per launch
(git show 1a04a99:ultrasonic-communication_amd/Makefile); since round 4 the stamped twin ships inside libuchirp.so
(uc_clock_probe / uc_clock_stamps, one buffer per context).  Its result is profiles/r03_boundary_tail.txt.
What a kernel boundary costs on this stack: two back-to-back launches of the same kernel on one stream, each with
its own stamp buffer (clock-stamp build): gap = first wave start of launch B - last wave end of launch A, in the
GPU's own 100 MHz s_memrealtime domain; beside it the launch period of a long back-to-back run.
Usage: python tools/boundary_probe.py [target=band_rx_real_f32] [frames_log2=20]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("UCHIRP_LIB", os.path.join(ROOT, "ultrasonic-communication_amd", "libuchirp_clock.so"))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ultrasonic-communication_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import uchirp  # noqa: E402
from uchirp import synth  # noqa: E402

target = sys.argv[1] if len(sys.argv) > 1 else "band_rx_real_f32"
nf = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev)
frames, _ = synth.device_frames(nf, dev, seed=1)
if target == "stream_d8_f32":
    e = uchirp.Engine(uchirp.STREAM)
    x = frames.reshape(-1)
    _, n_out, n_blocks, _ = e.stream_geometry(x.numel())
    comp = torch.empty(n_out, dtype=torch.float32, device=dev)
    pk = torch.empty((n_blocks, 2), dtype=torch.int32, device=dev)

    def launch():
        e.process_stream(x, compressed_out=comp, peaks_out=pk, stream=stream.cuda_stream)
else:
    e = uchirp.Engine(uchirp.RX_REAL, mag_mean=1000.0)
    sym = torch.empty(nf, dtype=torch.uint8, device=dev)

    def launch():
        e.process(frames, want_stats=False, symbols_out=sym, stream=stream.cuda_stream)

dbg = [torch.zeros(16384 * 4, dtype=torch.int64, device=dev) for _ in range(3)]
os.environ["UC_DEBUG_PTR"] = str(dbg[2].data_ptr())
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 1.5:
    for _ in range(50):
        launch()
    torch.cuda.synchronize()
    n += 50
t0 = time.perf_counter()
for _ in range(200):
    launch()
torch.cuda.synchronize()
period = (time.perf_counter() - t0) / 200 * 1e6
gaps, spans = [], []
for rep in range(10):
    for d in dbg:
        d.zero_()
    torch.cuda.synchronize()
    for k in range(3):
        os.environ["UC_DEBUG_PTR"] = str(dbg[k].data_ptr())
        launch()
    torch.cuda.synchronize()
    st = []
    for d in dbg:
        a = d.cpu().numpy().reshape(-1, 4)
        a = a[a[:, 1] > 0]
        st.append((a[:, 2].min(), a[:, 3].max()))
    gaps += [(st[1][0] - st[0][1]) / 100.0, (st[2][0] - st[1][1]) / 100.0]
    spans.append((st[1][1] - st[1][0]) / 100.0)
print(json.dumps({"target": target, "units": nf, "launch_period_us_back_to_back": period,
                  "grid_span_us_median": float(np.median(spans)),
                  "boundary_gap_us_median": float(np.median(gaps)), "boundary_gap_us_min_max": [float(min(gaps)), float(max(gaps))],
                  "note": "gap = first wave start of a launch - last wave end of the launch in front of it on the same stream"}))
