#!/usr/bin/env python3
"""Eager launches vs replays of a captured hipGraph, dynamic hand-out vs static deal: UC_STREAM over 2^31 samples and
RX_REAL over 2^20 frames.  HIP events around every launch / replay, median of 20 after a 150 ms ramp."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ultrasonic-communication_amd")]
os.environ["UC_TUNING"] = "1"
import uchirp  # noqa: E402
from uchirp import synth  # noqa: E402
from bench import clock_ramp  # noqa: E402

dev = torch.device("cuda:0")
frames, _ = synth.device_frames(1 << 20, dev, seed=1)
x = frames.reshape(-1)


def med(launch, stream):
    clock_ramp(launch, torch, 150.0)
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        launch()
        b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


for static in ("0", "1"):
    os.environ["UC_STATIC_DEAL"] = static
    es = uchirp.Engine(uchirp.STREAM)
    er = uchirp.Engine(uchirp.RX_REAL, mag_mean=1000.0)
    _, n_out, n_blocks, _ = es.stream_geometry(x.numel())
    comp = torch.empty(n_out, dtype=torch.float32, device=dev)
    pk = torch.empty((n_blocks, 2), dtype=torch.int32, device=dev)
    sym = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    s1 = torch.cuda.current_stream(dev)
    for name, call in (("stream", lambda st: es.process_stream(x, compressed_out=comp, peaks_out=pk, stream=st.cuda_stream)),
                       ("rx_real", lambda st: er.process(frames, want_stats=False, symbols_out=sym, stream=st.cuda_stream))):
        t_e = med(lambda: call(s1), s1)
        s2 = torch.cuda.Stream(dev)
        g = torch.cuda.CUDAGraph()
        s2.wait_stream(s1)
        with torch.cuda.stream(s2):
            with torch.cuda.graph(g, stream=s2):
                call(s2)
            t_g = med(g.replay, s2)
            t_e2 = med(lambda: call(s2), s2)     # eager on the side stream
        print("%-8s static_deal=%s  eager %.4f ms  graph replay %.4f ms  eager on the capture stream %.4f ms"
              % (name, static, t_e, t_g, t_e2), flush=True)
