#!/usr/bin/env python3
"""Round 6: recorded streams in ONE launch over all their blocks (everything evaluated) against block-by-block steps (only what the
switch can look at), per receiver and stream count -- where the threshold of uc_receive_streams' stepped path belongs.
The bench's streams: 176 blocks, 40 of them before the transmission.  Usage: python tools/r6_steps_ab.py  -> text lines"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) < 2:
    # (UC_RX_STEP_MIN is read at uc_create: one child process per setting)
    for variant in ("sync_cplx", "rx_real"):
        for ns in ((512, 1024, 2048, 4096, 16384) if variant == "sync_cplx" else (4096, 8192, 16384, 32768)):
            row = []
            for step_min in ("1000000000", "1"):
                env = dict(os.environ, UC_TUNING="1", UC_RX_STEP_MIN=step_min)
                out = subprocess.run([sys.executable, os.path.abspath(__file__), variant, str(ns)], env=env, capture_output=True, text=True)
                row.append(out.stdout.strip().splitlines()[-1] if out.returncode == 0 else "FAILED " + out.stderr[-300:])
            print("%-9s %6d streams   one launch: %s   |   block by block: %s" % (variant, ns, row[0], row[1]), flush=True)
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "ultrasonic-communication_amd")]
import numpy as np
import torch
import uchirp
from uchirp import tx

N, NB = 2048, 176
vname, ns = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(ns)
x = torch.randn((ns, NB * N), generator=g, device=dev) * 50.0
tone = torch.from_numpy(tx.render("Hello World!", fs_rx=78125.0, amplitude=2000.0).astype(np.float32)).to(dev)
x[:, 40 * N + 777:40 * N + 777 + tone.numel()] += tone
eng = uchirp.Engine(uchirp.SYNC_CPLX if vname == "sync_cplx" else uchirp.RX_REAL)
text = torch.zeros((ns, 64), dtype=torch.uint8, device=dev)
ntext = torch.zeros(ns, dtype=torch.int32, device=dev)
for _ in range(2):
    eng.receive_many_into(x, text, ntext)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 4
for _ in range(reps):
    eng.receive_many_into(x, text, ntext)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
ok = sum(1 for r, k in zip(text.cpu().numpy(), ntext.cpu().numpy()) if b"Hello World!" in bytes(r[:k]))
print("%.2f ms  %.3g blocks/s  (%d of %d decode)" % (ms, ns * NB / ms * 1e3, ok, ns))
