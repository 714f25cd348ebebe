#!/usr/bin/env python3
"""Interleaved A/B timing of band-kernel build variants (UC_BAND_WAVES / UC_GRID knobs).
Usage: python tools/tune_band.py [frames_log2=19] [rounds=5]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import uchirp
from bench import make_device_frames, BYTES_PER_FRAME

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 19
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
variant = int(os.environ.get("UC_VARIANT", "0"))
nf = 1 << lg
dev = torch.device("cuda:0")
frames, bits = make_device_frames(nf, dev, seed=1)
sym = torch.empty(nf, dtype=torch.uint8, device=dev)
configs = []
for waves in (2, 3, 4):
    for grid in (0,):
        configs.append((waves, grid))
for waves, mult in ((2, 2), (2, 3), (3, 4), (3, 5), (4, 6), (4, 7)):
    configs.append((waves, 256 * mult))
os.environ["UC_TUNING"] = "1"   # the library reads its experiment switches only under UC_TUNING=1
engines = []
for waves, grid in configs:
    os.environ["UC_BAND_WAVES"] = str(waves)
    os.environ["UC_GRID"] = str(grid)
    engines.append(uchirp.Engine(variant, mag_mean=1000.0, **({"n": int(os.environ["UC_N"])} if "UC_N" in os.environ else {})))
stream = torch.cuda.current_stream(dev)
ref = None
times = {c: [] for c in configs}
for r in range(rounds + 1):
    for c, e in zip(configs, engines):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        e.process(frames, want_stats=False, symbols_out=sym if e.n == 2048 else None, want_symbols=(e.n == 2048), stream=stream.cuda_stream)
        b.record(stream)
        torch.cuda.synchronize()
        if r:
            times[c].append(a.elapsed_time(b))
        if ref is None:
            ref = sym.clone()
        assert torch.equal(ref, sym), "variant %s changes the symbols" % (c,)
for c in configs:
    t = np.array(times[c])
    print("waves=%d grid=%5d  median %.3f ms  min %.3f ms  -> %.1f Mframes/s  %.0f GB/s (%.1f%% of 8 TB/s)"
          % (c[0], c[1], np.median(t), t.min(), nf / np.median(t) / 1e3, nf * BYTES_PER_FRAME / np.median(t) / 1e6,
             nf * BYTES_PER_FRAME / np.median(t) / 1e6 / 80.0))
