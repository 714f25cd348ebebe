#!/usr/bin/env python3
"""Two captured graphs of uc_process_batch replayed on two streams: concurrently / one after the other / from two contexts.
Counts replays whose output differs from the eager launch's (diagnostic for the graph-owned hand-out counters)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ultrasonic-communication_amd")]
os.environ["UC_TUNING"] = "1"
os.environ["UC_GRID"] = sys.argv[1] if len(sys.argv) > 1 else "3"
import uchirp  # noqa: E402
from uchirp import synth  # noqa: E402

dev = torch.device("cuda:0")
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
frames = torch.from_numpy(synth.make_frames(n_frames, seed=60, snr_db=-5.0)[0]).to(dev)


def capture(e, ss):
    sy = torch.zeros(n_frames, dtype=torch.uint8, device=dev)
    gg = torch.cuda.CUDAGraph()
    ss.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(ss):
        with torch.cuda.graph(gg, stream=ss):
            e.process(frames, want_stats=False, symbols_out=sy, stream=ss.cuda_stream)
    return gg, sy


def run(label, engines, concurrent):
    want, _ = engines[0].process(frames, want_stats=False)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    caps = [capture(e, s) for e, s in zip(engines, streams)]
    torch.cuda.synchronize()
    bad = 0
    for rep in range(50):
        for _, sy in caps:
            sy.zero_()
        torch.cuda.synchronize()
        for (gg, _), ss in zip(caps, streams):
            with torch.cuda.stream(ss):
                gg.replay()
            if not concurrent:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        for _, sy in caps:
            if not torch.equal(sy, want):
                bad += 1
    print("%-40s bad replays: %d of 100" % (label, bad), flush=True)


e1 = uchirp.Engine(uchirp.RX_REAL, mag_mean=1000.0)
e2 = uchirp.Engine(uchirp.RX_REAL, mag_mean=1000.0)
run("one context, serial replays", [e1, e1], False)
run("one context, concurrent replays", [e1, e1], True)
run("two contexts, concurrent replays", [e1, e2], True)
os.environ["UC_STATIC_DEAL"] = "1"
e3 = uchirp.Engine(uchirp.RX_REAL, mag_mean=1000.0)
run("static deal, concurrent replays", [e3, e3], True)
