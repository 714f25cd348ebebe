#!/usr/bin/env python3
"""The live receiver step without a host in the loop: K calls of uc_receive_streams_next back to back on one stream (no
read-back, no sync between calls), timed with HIP events -- what the kernels of a step cost when the device never waits for
the host; and the same step replayed from a captured hipGraph.
Usage: python tools/run_live_async.py [streams=4096,65536] [variant=rx_real] [calls=100] [keep=0|1]   -> JSON lines
keep=1: uc_rx_state_keep_previous -- the chunks are a ring the caller leaves alone until the next call has completed (they are:
this tool cycles through >= 4 buffers), so nothing is copied into the state; the graph form is then two captured steps (chunk 1
behind chunk 0, chunk 0 behind chunk 1) replayed in turn."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
import time

import torch
import uchirp
from bench import PowerSampler

N = 2048
counts = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4096,65536").split(",")]
vname = sys.argv[2] if len(sys.argv) > 2 else "rx_real"
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 100
keep = (int(sys.argv[4]) if len(sys.argv) > 4 else 0) != 0
dev = torch.device("cuda:0")
eng = uchirp.Engine(uchirp.RX_REAL if vname == "rx_real" else uchirp.SYNC_CPLX)
for ns in counts:
    nbuf = 4 if ns > 16384 else 16
    g = torch.Generator(device=dev)
    g.manual_seed(ns)
    chunks = [torch.randn((ns, N), generator=g, device=dev) * 50.0 for _ in range(nbuf)]
    live = eng.live(ns)
    if keep:
        live.keep_previous(True)
    text = torch.zeros((ns, 16), dtype=torch.uint8, device=dev)
    ntext = torch.zeros(ns, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for k in range(30):
            live.next_into(chunks[k % nbuf], text, ntext, stream=s.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for k in range(calls):
            live.next_into(chunks[k % nbuf], text, ntext, stream=s.cuda_stream)
        e1.record(s)
        e1.synchronize()
        eager_ms = e0.elapsed_time(e1) / calls
        # one step captured, replayed (the chunk buffer is the graph's: copy-in not timed here -- a live host DMA-s into it)
        class Pair:
            """two captured steps replayed in turn (keep=1), or one (keep=0)"""
            def __init__(self, graphs):
                self.g, self.k = graphs, 0
            def replay(self):
                self.g[self.k % len(self.g)].replay()
                self.k += 1
        graphs = []
        for c in ((1, 0) if keep else (0,)):
            gr1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr1, stream=s):
                live.next_into(chunks[c], text, ntext, stream=s.cuda_stream)
            graphs.append(gr1)
        gr = Pair(graphs)
        for k in range(10):
            gr.replay()
        e0.record(s)
        for k in range(calls):
            gr.replay()
        e1.record(s)
        e1.synchronize()
        graph_ms = e0.elapsed_time(e1) / calls
        # socket power / SMU clock over the last second of 2 s of graph replays
        ps = PowerSampler(torch, dev)
        ps.start()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < float(os.environ.get("UC_LIVE_SUSTAIN_S", "2.0")):
            for k in range(64):
                gr.replay()
            s.synchronize()
        t1 = time.perf_counter()
        power = ps.stop(t1 - 1.0, t1)
    print(json.dumps({"variant": vname, "streams": ns, "calls": calls, "keep_previous": keep, "eager_ms_per_call": eager_ms, "graph_ms_per_call": graph_ms,
                      "new_frames_per_call": ns * 8, "frames_per_s_eager": ns * 8 / eager_ms * 1e3,
                      "microphones_in_real_time_eager": int(ns * 26.2144 / eager_ms),
                      "socket_W": round(power["socket_W_mean"], 1) if power else None, "cap_W": power["cap_W"] if power else None,
                      "smu_clock_MHz": round(power["sclk_MHz_smu_mean"]) if power and power.get("sclk_MHz_smu_mean") else None,
                      "env": {k: v for k, v in os.environ.items() if k.startswith("UC_")},
                      "what": "uc_receive_streams_next back to back, one new block of every stream per call, no host sync"}), flush=True)
    live.close()
    del chunks
eng.close()
