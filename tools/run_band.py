#!/usr/bin/env python3
"""Run the band kernel a few times on a device-resident batch (profiling target).
Usage: python tools/run_band.py [frames_log2=19] [iters=5] ; env UC_BAND_WAVES, UC_GRID, UC_VARIANT"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import uchirp
from bench import make_device_frames

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 19
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nf = 1 << lg
dev = torch.device("cuda:0")
frames, bits = make_device_frames(nf, dev, seed=1)
sym = torch.empty(nf, dtype=torch.uint8, device=dev)
e = uchirp.Engine(int(os.environ.get("UC_VARIANT", "0")), mag_mean=1000.0, **({"n": int(os.environ["UC_N"])} if "UC_N" in os.environ else {}))
for _ in range(iters):
    e.process(frames, want_stats=False, symbols_out=sym)
torch.cuda.synchronize()
print("done", nf, iters)
