#!/usr/bin/env python3
"""Run the band kernel a few times on a device-resident batch (profiling target).
Usage: python tools/run_band.py [frames_log2=19] [iters=5] ; env UC_BAND_WAVES, UC_GRID, UC_VARIANT"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
import torch
import uchirp
from bench import make_device_frames

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 19
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nf = 1 << lg
dev = torch.device("cuda:0")
frames, bits = make_device_frames(nf, dev, seed=1)
e = uchirp.Engine(int(os.environ.get("UC_VARIANT", "0")), mag_mean=1000.0, **({"n": int(os.environ["UC_N"])} if "UC_N" in os.environ else {}))
nfr = (frames.numel() - e.halo - e.n) // e.n + 1          # frames of THIS engine's length in the batch
want_sym = e.variant in (uchirp.RX_REAL, uchirp.SYNC_CPLX)
sym = torch.empty(nfr, dtype=torch.uint8, device=dev) if want_sym else None
st = None if want_sym else torch.empty((nfr, e.spf, 8), dtype=torch.float32, device=dev)
for _ in range(iters):
    e.process(frames, n_frames=nfr, want_symbols=want_sym, want_stats=not want_sym, symbols_out=sym, stats_out=st)
torch.cuda.synchronize()
print("done", nf, iters)
