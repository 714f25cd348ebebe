#!/usr/bin/env python3
"""A long live run: 96 microphones, 2400 blocks each (63 s of signal per microphone: transmissions with random gaps, noise),
through (a) ONE uc_receive_streams call over the whole recording, (b) uc_receive_streams_next one block per call, eager,
(c) the same step replayed 2400 times from ONE captured hipGraph, (d) chunks of random sizes with a busy mask against the
recorded call with the same mask; round 6: (e) one block per call under uc_rx_state_keep_previous, two chunk buffers in turn, eager,
(f) the same as TWO captured graphs replayed in turn.  Texts and every trace record must be equal, bit for bit; the hand-out counters must be zero.
Usage: python tools/soak_live.py [variant=sync_cplx] [blocks=2400] [streams=96]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
import numpy as np
import torch
import uchirp
from uchirp import tx

N = 2048
vname = sys.argv[1] if len(sys.argv) > 1 else "sync_cplx"
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 2400
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 96
dev = torch.device("cuda:0")
rng = np.random.default_rng(2024)
x = (rng.standard_normal((ns, blocks * N)) * 60.0).astype(np.float32)
sent = 0
for s in range(ns):
    at = int(rng.integers(25, 60)) * N + int(rng.integers(0, N))
    while True:
        msg = "".join(chr(int(c)) for c in rng.integers(48, 123, size=int(rng.integers(1, 9))))
        tone = tx.render(msg, fs_rx=78125.0, amplitude=float(rng.choice([800.0, 3000.0]))).astype(np.float32)
        if at + tone.size > x.shape[1]:
            break
        x[s, at:at + tone.size] += tone
        sent += 1
        at += tone.size + int(rng.integers(10, 80)) * N + int(rng.integers(0, N))
eng = uchirp.Engine(uchirp.SYNC_CPLX if vname == "sync_cplx" else uchirp.RX_REAL)
xd = torch.from_numpy(x).to(dev)
cap = 4096
w_text, w_trace = eng.receive_many(xd, text_cap=cap)
decoded = sum(t.count("\n") for t in w_text)
print("%s: %d streams x %d blocks, %d transmissions sent, %d messages ended in the recorded call" % (vname, ns, blocks, sent, decoded), flush=True)


def run_live(mode):
    live = eng.live(ns)
    keep = mode.startswith("keep")
    if keep:
        live.keep_previous(True)
    ring = [torch.zeros((ns, N), dtype=torch.float32, device=dev) for _ in range(2 if keep else 1)]
    chunk = ring[0]
    text = torch.zeros((ns, 8), dtype=torch.uint8, device=dev)
    ntext = torch.zeros(ns, dtype=torch.int32, device=dev)
    trace = torch.zeros((ns, 1, uchirp.RX_EVENT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    ntrace = torch.zeros(ns, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream()
    g = None
    graphs = None
    first_block = 0
    if mode == "keep_graph":
        # block 0 eagerly (it sizes the scratch; in front of it: the state's power-on FIFO), then one captured step per buffer
        ring[0].copy_(xd[:, :N])
        live.next_into(ring[0], text, ntext, trace=trace, n_trace=ntrace)
        torch.cuda.synchronize()
        first_block = 1
        graphs = []
        st.wait_stream(torch.cuda.current_stream())
        for k in (1, 0):
            gk = torch.cuda.CUDAGraph()
            with torch.cuda.stream(st):
                with torch.cuda.graph(gk, stream=st):
                    live.next_into(ring[k], text, ntext, trace=trace, n_trace=ntrace, stream=st.cuda_stream)
            graphs.append(gk)      # graphs[0]: ring[1] behind ring[0]; graphs[1]: ring[0] behind ring[1]
    if mode == "graph":
        live.next_into(chunk, text, ntext, trace=trace, n_trace=ntrace)
        live.reset()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            with torch.cuda.graph(g, stream=st):
                live.next_into(chunk, text, ntext, trace=trace, n_trace=ntrace, stream=st.cuda_stream)
    texts = [bytearray() for _ in range(ns)]
    recs = [[] for _ in range(ns)]
    # the outputs of every step are kept on the device and read back in bulk (the soak is about the state, not the host)
    all_text = torch.zeros((blocks, ns, 8), dtype=torch.uint8, device=dev)
    all_nt = torch.zeros((blocks, ns), dtype=torch.int32, device=dev)
    all_tr = torch.zeros((blocks, ns, uchirp.RX_EVENT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    if first_block:
        all_text[0].copy_(text)
        all_nt[0].copy_(ntext)
        all_tr[0].copy_(trace[:, 0])
    for b in range(first_block, blocks):
        chunk = ring[b % len(ring)]
        chunk.copy_(xd[:, b * N:(b + 1) * N])
        if graphs is not None:
            torch.cuda.current_stream().synchronize()
            graphs[(b + 1) % 2].replay()
            st.synchronize()
        elif g is not None:
            torch.cuda.current_stream().synchronize()
            g.replay()
            st.synchronize()
        else:
            live.next_into(chunk, text, ntext, trace=trace, n_trace=ntrace)
        all_text[b].copy_(text)
        all_nt[b].copy_(ntext)
        all_tr[b].copy_(trace[:, 0])
    torch.cuda.synchronize()
    at, an, atr = all_text.cpu().numpy(), all_nt.cpu().numpy(), all_tr.cpu().numpy()
    for s in range(ns):
        for b in np.nonzero(an[:, s])[0]:
            texts[s] += bytes(at[b, s, :an[b, s]])
        recs[s] = atr[:, s].copy().view(uchirp.RX_EVENT_DTYPE).reshape(-1)
    live.close()
    return [bytes(t).decode("latin-1") for t in texts], recs


bad = 0
for mode in ("eager", "graph", "keep", "keep_graph"):
    t, tr = run_live(mode)
    for s in range(ns):
        if t[s] != w_text[s] or not np.array_equal(tr[s].view(np.uint8), w_trace[s].view(np.uint8)):
            bad += 1
            print("FAIL %s: stream %d differs from the recorded call" % (mode, s), flush=True)
            break
    print("live, one block per call, %s: %s" % (mode, "equal to the recorded call, text and trace" if not bad else "DIFFERS"), flush=True)
# chunks of random sizes, dropped blocks
busy = (rng.random((ns, blocks)) < 0.04).astype(np.uint8)
b_text, b_trace = eng.receive_many(xd, busy=busy, text_cap=cap)
live = eng.live(ns)
texts, recs, b0 = [""] * ns, [[] for _ in range(ns)], 0
while b0 < blocks:
    nb = int(min(blocks - b0, rng.choice([1, 2, 3, 7, 16, 40])))
    t, tr = live.next(xd[:, b0 * N:(b0 + nb) * N].contiguous(), busy=np.ascontiguousarray(busy[:, b0:b0 + nb]), text_cap=64)
    for s in range(ns):
        texts[s] += t[s]
        recs[s].append(tr[s])
    b0 += nb
live.close()
for s in range(ns):
    if texts[s] != b_text[s] or not np.array_equal(np.concatenate(recs[s]).view(np.uint8), b_trace[s].view(np.uint8)):
        bad += 1
        print("FAIL chunks + busy mask: stream %d" % s, flush=True)
        break
print("live, random chunks with dropped blocks: %s" % ("equal to the recorded call with the same mask" if not bad else "DIFFERS"), flush=True)
print("hand-out counters not zero: %d" % eng.busy_counters())
bad += eng.busy_counters()
eng.close()
print("soak %s" % ("ok" if not bad else "FAILED"))
sys.exit(1 if bad else 0)
