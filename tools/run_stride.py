#!/usr/bin/env python3
"""DESIGN section 5's cross-check with the power meter on: the headline kernel (band_kernel<rx_real>, fp32) over frames a
whole frame apart (every byte of every frame comes from HBM: 8193 B/frame) and over frames 256 samples apart (7/8 of every
frame's bytes are cache hits: 1025 B/frame from HBM) -- same frames per launch, same box, alternated; per leg the rate (HIP
events), the in-kernel shader clock (clock-stamped twin) and the socket power / SMU clock over the last second of a sustained run.
Usage: python tools/run_stride.py [frames_log2=20] [seconds=3] [only: 2048|256]   -> JSON lines"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
import torch
import uchirp
from bench import PowerSampler
from uchirp import synth

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
nf = 1 << lg
dev = torch.device("cuda:0")
frames, _ = synth.device_frames(nf, dev, seed=1234, snr_db=-10.0)
flat = frames.reshape(-1)
e = uchirp.Engine(uchirp.RX_REAL, mag_mean=1000.0)
stream = torch.cuda.current_stream(dev)
legs = [("stride 2048 (8193 B/frame from HBM)", 0, nf), ("stride 256 (1025 B/frame from HBM, the rest cache hits)", 256, nf)]
if len(sys.argv) > 3:
    legs = [l for l in legs if l[0].startswith("stride " + sys.argv[3] + " ")]
sym = torch.empty(nf, dtype=torch.uint8, device=dev)
for rep in range(2):
    for name, stride, n in legs:
        def launch():
            e.process(flat, n_frames=n, stride=stride, want_stats=False, symbols_out=sym, stream=stream.cuda_stream)
        for _ in range(40):
            launch()
        torch.cuda.synchronize()
        ps = PowerSampler(torch, dev)
        ps.start()
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(16):
                launch()
            torch.cuda.synchronize()
            k += 16
        t1 = time.perf_counter()
        power = ps.stop(t1 - 1.0, t1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(20):
            launch()
        b.record(stream)
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        e.clock_probe(True)
        launch()
        clk = e.clock_read()
        e.clock_probe(False)
        rec = {"leg": name, "rep": rep, "frames_per_launch": n, "ms_per_launch": ms, "frames_per_s": n / ms * 1e3,
               "sustained_frames_per_s": n * k / (t1 - t0), "shader_ghz_in_kernel": round(clk["shader_ghz"], 3), "power": power}
        if power and power.get("socket_W_mean"):
            rec["uJ_per_frame_socket"] = power["socket_W_mean"] / rec["sustained_frames_per_s"] * 1e6
        print(json.dumps(rec), flush=True)
