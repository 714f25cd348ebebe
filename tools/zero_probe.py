#!/usr/bin/env python3
"""Power-cap probe: the same kernel, the same instruction stream, on all-zero frames (little switching
activity in the multipliers and the LDS) against noise frames.  A large gap = the kernel is limited by
the socket power cap, not by issue slots.  Usage: python tools/zero_probe.py [frames_log2=20]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import uchirp
from bench import make_device_frames

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nf = 1 << lg
dev = torch.device("cuda:0")
noise, _ = make_device_frames(nf, dev, seed=1)
zeros = torch.zeros_like(noise)
sym = torch.empty(nf, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream(dev)
for name, vid, kw in (("rx_real", uchirp.RX_REAL, {}), ("dechirp_down", uchirp.DECHIRP_DOWN, {}), ("compress", uchirp.COMPRESS, {}),
                      ("iq1024", uchirp.IQ, {"n": 1024}), ("iq", uchirp.IQ, {})):
    e = uchirp.Engine(vid, mag_mean=1000.0, **kw)
    nfr = (noise.numel() - e.halo - e.n) // e.n + 1
    st = torch.empty((nfr, e.spf, 8), dtype=torch.float32, device=dev)
    for label, x in (("noise", noise), ("zeros", zeros), ("noise", noise), ("zeros", zeros)):
        ts = []
        for r in range(40):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            if vid == uchirp.RX_REAL:
                e.process(x, want_stats=False, symbols_out=sym, stream=stream.cuda_stream)
            else:
                e.process(x, n_frames=nfr, want_symbols=False, want_stats=True, stats_out=st, stream=stream.cuda_stream)
            b.record(stream)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        t = np.median(ts[10:])
        print("%-13s %-6s median %.3f ms -> %.1f Mframes/s" % (name, label, t, nfr / t / 1e3), flush=True)
