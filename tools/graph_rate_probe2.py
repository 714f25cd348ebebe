#!/usr/bin/env python3
"""Why bench.py saw graph replays of UC_STREAM 9 % slower than eager launches: buffers / back-to-back queueing variants."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ultrasonic-communication_amd")]
import uchirp  # noqa: E402
from uchirp import synth  # noqa: E402
from bench import clock_ramp, timed_launches  # noqa: E402

dev = torch.device("cuda:0")
frames, _ = synth.device_frames(1 << 20, dev, seed=1)
x = frames.reshape(-1)
es = uchirp.Engine(uchirp.STREAM)
_, n_out, n_blocks, _ = es.stream_geometry(x.numel())
bufs = [(torch.empty(n_out, dtype=torch.float32, device=dev), torch.empty((n_blocks, 2), dtype=torch.int32, device=dev))
        for _ in range(3)]
s1 = torch.cuda.current_stream(dev)


def eager(b, st):
    return lambda: es.process_stream(x, compressed_out=bufs[b][0], peaks_out=bufs[b][1], stream=st.cuda_stream)


clock_ramp(eager(0, s1), torch, 150.0)
for b in range(3):
    print("eager back-to-back, buffers %d: wall %.4f ms  events %.4f ms" % ((b,) + timed_launches(eager(b, s1), s1, torch, 10, 3)), flush=True)
s2 = torch.cuda.Stream(dev)
for b in (0, 1, 2):
    g = torch.cuda.CUDAGraph()
    s2.wait_stream(s1)
    with torch.cuda.stream(s2):
        with torch.cuda.graph(g, stream=s2):
            eager(b, s2)()
        print("graph back-to-back, buffers %d: wall %.4f ms  events %.4f ms" % ((b,) + timed_launches(g.replay, s2, torch, 10, 3)), flush=True)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("graph, sync after every replay, buffers %d: wall median %.4f ms" % (b, float(np.median(ts))), flush=True)
        print("eager on the capture stream, buffers %d: wall %.4f ms  events %.4f ms" % ((b,) + timed_launches(eager(b, s2), s2, torch, 10, 3)), flush=True)
