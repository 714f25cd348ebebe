#!/usr/bin/env python3
"""Random shapes through the many-stream receivers (the ROWS build of the band kernel, the carried records, the replay kernels):
n_streams, blocks per call, row pitch (incl. pitches that are not multiples of 256 samples), busy masks, dtype (float32 / int32 DFSDM
words / PDM bit streams), host or device buffers, both receivers.  For every draw the chunked LIVE run (uc_receive_streams_next, chunk
sizes drawn per call) must give, bit for bit, the texts and traces of ONE call over the whole streams (uc_receive_streams), and that
call must equal uc_receive_stream[_isr] stream by stream on a sample of the streams.  Round 6: four draws in ten run the live
state under uc_rx_state_keep_previous (device chunks are then held for two calls: the promise; host / PDM / busy-masked calls mix
in as they are drawn).
Usage: python tools/fuzz_live.py [cases=60] [seed=1]"""
import ctypes as C
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
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
L = uchirp.lib()
dev = torch.device("cuda:0")
bad = 0
HELD = []


def streams(ns, blocks):
    x = np.zeros((ns, blocks * N), np.float32)
    for s in range(ns):
        msg = "".join(chr(int(c)) for c in rng.integers(48, 123, size=int(rng.integers(1, 4))))
        amp = float(rng.choice([500.0, 4000.0]))
        sigma = amp * float(rng.choice([0.02, 0.2]))
        tone = tx.render(msg, fs_rx=78125.0, amplitude=amp)
        row = rng.standard_normal(blocks * N) * sigma
        lead = int(rng.integers(20, 30)) * N + int(rng.integers(0, N))
        if rng.random() < 0.8 and lead + tone.size <= row.size:
            row[lead:lead + tone.size] += tone
        x[s] = row
    return x


def call(eng, state, buf, dt, ns, nsmp, pitch, busy, cap, trace_cap, device):
    """one uc_receive_streams[_next] call through ctypes (row pitch as given) -> (texts, traces)"""
    text = np.zeros((ns, cap), np.uint8)
    ntext = np.zeros(ns, np.uint32)
    trace = np.zeros((ns, max(trace_cap, 1)), uchirp.RX_EVENT_DTYPE)
    ntrace = np.zeros(ns, np.uint32)
    keep = None
    if device:
        keep = torch.from_numpy(buf).to(dev)
        HELD.append(keep)            # (uc_rx_state_keep_previous: a device chunk stays alive and unchanged for the next call too)
        del HELD[:-3]
        ptr = C.c_void_p(keep.data_ptr())
    else:
        ptr = buf.ctypes.data_as(C.c_void_p)
    bz = np.ascontiguousarray(busy, np.uint8) if busy is not None else None
    args = (ptr, dt, nsmp, pitch, bz.ctypes.data_as(C.c_void_p) if bz is not None else None, text.ctypes.data_as(C.c_void_p), cap,
            ntext.ctypes.data_as(C.c_void_p), trace.ctypes.data_as(C.c_void_p), max(trace_cap, 1), ntrace.ctypes.data_as(C.c_void_p), None)
    if state is None:
        rc = L.uc_receive_streams(eng._h, args[0], args[1], ns, *args[2:])
    else:
        rc = L.uc_receive_streams_next(eng._h, state._h, *args)
    assert rc == 0, L.uc_last_error()
    return [bytes(text[i, :ntext[i]]) for i in range(ns)], [trace[i, :ntrace[i]].copy() for i in range(ns)]


for case in range(cases):
    variant = int(rng.choice([uchirp.RX_REAL, uchirp.SYNC_CPLX]))
    ns = int(rng.choice([1, 2, 3, 7, 33, 64, 65, 130]))
    blocks = int(rng.integers(45, 80))
    x = streams(ns, blocks)
    kind = str(rng.choice(["f32", "i32", "pdm"], p=[0.45, 0.35, 0.2]))
    if kind == "f32":
        data, dt = x, uchirp.DTYPE_F32
    elif kind == "i32":
        data, dt = (np.round(x).astype(np.int64) * 256).astype(np.int32), uchirp.DTYPE_I32
    else:   # bit streams: no meaning as audio, a lot of meaning as a test of the sinc5 + ROWS + replay chain
        data, dt = rng.integers(-(1 << 31), (1 << 31) - 1, size=x.shape, dtype=np.int64).astype(np.int32), uchirp.DTYPE_PDM
    busy = (rng.random((ns, blocks)) < float(rng.choice([0.03, 0.25]))).astype(np.uint8) if rng.random() < 0.4 else None
    device = bool(rng.random() < 0.5)
    pad = int(rng.choice([0, 4, 100, 256, 1000])) if kind != "pdm" else int(rng.choice([0, 4, 256]))
    eng = uchirp.Engine(variant)
    # the whole streams in one call, rows `pad` words apart
    whole = np.zeros((ns, blocks * N + pad), data.dtype)
    whole[:, :blocks * N] = data
    whole[:, blocks * N:] = 12345
    w_t, w_tr = call(eng, None, whole, dt, ns, blocks * N, blocks * N + pad, busy, 32, blocks, device)
    # live, in chunks of random sizes
    live = eng.live(ns)
    kept = bool(rng.random() < 0.4)
    if kept:
        live.keep_previous(True)
    texts, traces, b0 = [b""] * ns, [[] for _ in range(ns)], 0
    while b0 < blocks:
        nb = int(min(blocks - b0, rng.choice([1, 1, 1, 2, 3, 5, 8, 13])))
        cpad = int(rng.choice([0, 4, 256])) if kind == "pdm" else int(rng.choice([0, 4, 52, 256]))
        chunk = np.zeros((ns, nb * N + cpad), data.dtype)
        chunk[:, :nb * N] = data[:, b0 * N:(b0 + nb) * N]
        cb = None if busy is None else busy[:, b0:b0 + nb]
        t, tr = call(eng, live, chunk, dt, ns, nb * N, nb * N + cpad, cb, 32, nb, bool(rng.random() < 0.5))
        for s in range(ns):
            texts[s] += t[s]
            traces[s].append(tr[s])
        b0 += nb
    ok = True
    for s in range(ns):
        got = np.concatenate(traces[s]) if traces[s] else np.zeros(0, uchirp.RX_EVENT_DTYPE)
        if texts[s] != w_t[s] or not np.array_equal(got.view(np.uint8), w_tr[s].view(np.uint8)):
            ok = False
            print("FAIL case %d (variant %d, %s, %d streams, busy %s, kept %s): live chunks differ from the whole call at stream %d" % (case, variant, kind, ns, busy is not None, kept, s), flush=True)
            break
    # the whole call against one stream at a time (the host replay), on a sample
    if ok and kind != "pdm":
        for s in rng.choice(ns, size=min(ns, 3), replace=False):
            t1, tr1 = eng.receive(data[s], busy=None if busy is None else busy[s])
            if t1.encode("latin-1") != w_t[s] or not np.array_equal(tr1.view(np.uint8), w_tr[s].view(np.uint8)):
                ok = False
                print("FAIL case %d: stream %d of the many-stream call differs from uc_receive_stream" % (case, s), flush=True)
                break
    bad += 0 if ok else 1
    live.close()
    eng.close()
    if case % 20 == 19:
        print("%d cases, %d failures" % (case + 1, bad), flush=True)
print("done: %d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
