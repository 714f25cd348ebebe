#!/usr/bin/env python3
"""The live receivers at the scale one GPU serves: 65 536 streams (64 different transmissions, each tiled 1024 times on the device),
140 blocks, one block per call -- the lane-per-stream replay, the idle masks, the state store of 65 536 rows -- every stream's text
against the recorded call of its base stream.  Usage: python tools/scale_live.py"""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, uchirp
from test_gpu_receive_many import _transmissions
N=2048
dev=torch.device("cuda:0")
x, busy, msgs = _transmissions(64, seed=77, blocks=140)
for variant in (uchirp.RX_REAL, uchirp.SYNC_CPLX):
    e = uchirp.Engine(variant)
    base_t, base_tr = e.receive_many(x)
    xd = torch.from_numpy(x).to(dev)
    ns = 65536
    for kept in (False, True):          # round 6: the second pass under uc_rx_state_keep_previous (the last two chunks are held)
        live = e.live(ns)
        if kept:
            live.keep_previous(True)
        held = []
        text = torch.zeros((ns, 8), dtype=torch.uint8, device=dev)
        ntext = torch.zeros(ns, dtype=torch.int32, device=dev)
        acc = torch.zeros((ns, 64), dtype=torch.uint8, device=dev)
        alen = torch.zeros(ns, dtype=torch.int64, device=dev)
        rows = torch.arange(ns, device=dev)
        for b in range(140):
            chunk = xd[:, b*N:(b+1)*N].repeat(ns // 64, 1).contiguous()
            held.append(chunk); del held[:-2]
            live.next_into(chunk, text, ntext)
            m = ntext > 0
            if bool(m.any()):
                acc[rows[m], alen[m]] = text[m, 0]
                alen[m] += 1
        torch.cuda.synchronize()
        a, l = acc.cpu().numpy(), alen.cpu().numpy()
        bad = 0
        for s in range(ns):
            if bytes(a[s, :l[s]]).decode("latin-1") != base_t[s % 64]:
                bad += 1
        print("variant %d%s: 65536 live streams x 140 blocks, one block per call: %d streams differ from the recorded call of their stream; %d of 64 base streams decode" % (variant, ", kept chunks" if kept else "", bad, sum(mm in t for mm, t in zip(msgs, base_t))), flush=True)
        live.close()
        if bad:
            sys.exit(1)
    e.close()
