#!/usr/bin/env python3
"""The ROWS build of the band kernel over the same 524 288 new FIFO offsets in different shapes (HIP events, back to back):
  live      65 536 streams x 1 block per call with a state (reads the state's newest block, stores the new one)
  recorded  n streams x k blocks in one call without a state (the block in front of block 0 is zeros, nothing is stored)
Usage: python tools/run_rows_shapes.py [variant=rx_real]  -> JSON lines"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
import torch
import uchirp

N = 2048
vname = sys.argv[1] if len(sys.argv) > 1 else "rx_real"
dev = torch.device("cuda:0")
eng = uchirp.Engine(uchirp.RX_REAL if vname == "rx_real" else uchirp.SYNC_CPLX)
s = torch.cuda.Stream()
g = torch.Generator(device=dev)
g.manual_seed(5)
x = torch.randn((65536, N), generator=g, device=dev) * 50.0            # 512 MiB
for ns, nb in ((65536, 1), (16384, 4), (4096, 16), (1024, 64), (256, 256)):
    text = torch.zeros((ns, 16), dtype=torch.uint8, device=dev)
    ntext = torch.zeros(ns, dtype=torch.int32, device=dev)
    xs = x.reshape(ns, nb * N)
    rows = []
    for mode in ("recorded", "live"):
        live = eng.live(ns) if mode == "live" else None
        with torch.cuda.stream(s):
            def call():
                if live is None:
                    eng.receive_many_into(xs, text, ntext, stream=s.cuda_stream)
                else:
                    live.next_into(xs, text, ntext, stream=s.cuda_stream)
            for _ in range(10):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(40):
                call()
            e1.record(s)
            e1.synchronize()
        rows.append((mode, e0.elapsed_time(e1) / 40))
        if live is not None:
            live.close()
    print(json.dumps({"variant": vname, "streams": ns, "blocks_per_call": nb, "new_offsets": ns * nb * 8,
                      **{m + "_ms": round(t, 4) for m, t in rows},
                      **{m + "_offsets_per_s": round(ns * nb * 8 / t * 1e3) for m, t in rows}}), flush=True)
eng.close()
