#!/usr/bin/env python3
"""Soak: the same batch through every frame kernel many times; every launch must reproduce the first launch's bytes
(the dynamic hand-out changes WHICH workgroup does a group, never the result).  Usage: python tools/soak.py [launches=300]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
import torch
import uchirp
from uchirp import synth

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
nf = 1 << 18
frames, _ = synth.device_frames(nf, dev, seed=99, snr_db=-10.0)
flat = frames.reshape(-1)
bad = 0
for name, variant, kw in (("rx_real", uchirp.RX_REAL, {}), ("sync_cplx", uchirp.SYNC_CPLX, {}), ("dechirp_down", uchirp.DECHIRP_DOWN, {}),
                          ("compress", uchirp.COMPRESS, {}), ("iq", uchirp.IQ, {}), ("iq1024", uchirp.IQ, {"n": 1024}),
                          ("iq1024 base band", uchirp.IQ, {"n": 1024, "flags": uchirp.FLAG_IQ_BASEBAND, "fs": 100000.0,
                                                           "carrier": 18000.0, "f0": 16500.0, "f1": 19500.0, "time_frame": 1024 / 100000.0}),
                          ("iq base band", uchirp.IQ, {"flags": uchirp.FLAG_IQ_BASEBAND, "fs": 100000.0,
                                                       "carrier": 18000.0, "f0": 16500.0, "f1": 19500.0, "time_frame": 2048 / 100000.0}),
                          ("rx_real 41.7 kHz", uchirp.RX_REAL, {"fs": 125000.0 / 3.0, "time_frame": 2048 * 3.0 / 125000.0})):
    e = uchirp.Engine(variant, mag_mean=1000.0, **kw)
    n = e.n
    n_frames = (flat.numel() - e.halo - n) // n + 1
    sym = torch.empty(n_frames, dtype=torch.uint8, device=dev)
    st = torch.empty((n_frames, e.spf, 8), dtype=torch.float32, device=dev)
    e.process(flat, n_frames=n_frames, symbols_out=sym, stats_out=st)
    torch.cuda.synchronize()
    sym0, st0 = sym.clone(), st.clone()
    t0 = time.perf_counter()
    diff = 0
    for k in range(launches):
        e.process(flat, n_frames=n_frames, symbols_out=sym, stats_out=st)
        if k % 10 == 9 or k == launches - 1:   # compare on the device every 10th launch (the others overwrite in place)
            diff += int((sym != sym0).sum().item()) + int((st.view(torch.int32) != st0.view(torch.int32)).sum().item())
    torch.cuda.synchronize()
    print("%-18s %7d frames x %d launches: %d differing words  (%.1f s)" % (name, n_frames, launches, diff, time.perf_counter() - t0), flush=True)
    bad += diff
e = uchirp.Engine(uchirp.STREAM)
comp, pk = e.process_stream(flat)
torch.cuda.synchronize()
c0, p0 = comp.clone(), pk.clone()
diff = 0
for k in range(launches):
    e.process_stream(flat, compressed_out=comp, peaks_out=pk)
    if k % 10 == 9 or k == launches - 1:
        diff += int((comp.view(torch.int32) != c0.view(torch.int32)).sum().item()) + int((pk != p0).sum().item())
print("%-18s %7d samples x %d launches: %d differing words" % ("stream", flat.numel(), launches, diff), flush=True)
bad += diff
sys.exit(1 if bad else 0)
