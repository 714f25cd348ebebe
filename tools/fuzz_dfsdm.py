#!/usr/bin/env python3
"""Random shapes through the DFSDM front end (uc_dfsdm_sinc5_streams / uc_dfsdm_sinc5: the segment kernel + hist_kernel): number of
streams (1 .. 200 000), chunk lengths (1 .. 40 000 words, ragged, around the 256-word tile and the 8-tile segment), several chunks in
a row over one carried history array, row pitches of the input and the output wider than the chunk, host or device buffers, bit
patterns that saturate the filter next to random ones.  Every checked stream must equal, word for word, the oracle's DFSDM words of
the whole stream behind its first history; guard words behind every output row must survive; the history array must end as the last
four words of every stream.  The one-stream call is drawn as well.
Usage: python tools/fuzz_dfsdm.py [cases=100] [seed=1]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
import numpy as np
import torch
import uchirp
from oracle import uco

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
L = uchirp.lib()
dev = torch.device("cuda:0")
e = uchirp.Engine(uchirp.RX_REAL)
GUARD = -7654321
bad = 0


def words(ns, nw):
    w = rng.integers(0, 1 << 32, size=(ns, nw), dtype=np.uint64).astype(np.uint32)
    k = int(rng.integers(0, 4))
    if k == 1:
        w[:: max(1, ns // 7)] = 0xFFFFFFFF          # full scale: the clip
    elif k == 2:
        w[:: max(1, ns // 5)] = 0
    elif k == 3:
        w[:: max(1, ns // 3)] = 0xAAAAAAAA          # silence
    return w


def draw_len():
    k = int(rng.integers(0, 6))
    if k == 0:
        return int(rng.integers(1, 12))
    if k == 1:
        return int(rng.choice([252, 255, 256, 257, 260, 511, 512, 513]))
    if k == 2:
        return int(rng.choice([2044, 2047, 2048, 2049, 2052, 4096, 4100]))
    if k == 3:
        return 2048
    return int(rng.integers(1, 40000))


for case in range(cases):
    kind = int(rng.integers(0, 8))
    if kind == 0:
        # the one-stream call: history = the first four words of the buffer
        n = int(rng.integers(5, 3_000_000))
        w = words(1, n)[0]
        ref = uco.dfsdm_sinc5(w)
        if rng.random() < 0.5:
            got = e.dfsdm(w)
        else:
            t = torch.from_numpy(w.view(np.int32)).to(dev)
            out = torch.full((n,), GUARD, dtype=torch.int32, device=dev)
            e.dfsdm(t, out=out[:n - 4])
            torch.cuda.synchronize()
            o = out.cpu().numpy()
            if not (o[n - 4:] == GUARD).all():
                print("case %d: one stream of %d words stored past its end" % (case, n))
                bad += 1
            got = o[:n - 4]
        if not np.array_equal(got, ref):
            print("case %d: one stream of %d words differs" % (case, n))
            bad += 1
        continue
    ns = int(rng.choice([1, 2, 3, 17, 64, 300, 4096, 5000, 20000, 200000]))
    chunks = [draw_len() for _ in range(int(rng.integers(1, 5)))]
    while ns * sum(chunks) > 60_000_000:
        chunks = [max(1, c // 2) for c in chunks]
    on_device = rng.random() < 0.5
    if on_device:
        chunks = [max(4, c & ~3) for c in chunks]    # device rows stay 16-byte aligned
    total = sum(chunks)
    w = words(ns, total)
    hist0 = rng.integers(0, 1 << 32, size=(ns, 4), dtype=np.uint64).astype(np.uint32)
    check = range(ns) if ns <= 64 else sorted(set(rng.integers(0, ns, size=24).tolist()) | {0, ns - 1})
    ref = {s: uco.dfsdm_sinc5(np.concatenate([hist0[s], w[s]])) for s in check}
    got, b0, ok = [], 0, True
    if on_device:
        hist = torch.from_numpy(hist0.view(np.int32).copy()).to(dev)
        wd = torch.from_numpy(w.view(np.int32)).to(dev)
        for nw in chunks:
            pin = nw + 4 * int(rng.integers(0, 3))
            pout = nw + 4 * int(rng.integers(1, 3))
            src = torch.full((ns, pin), 0x5A5A5A5A, dtype=torch.int32, device=dev)
            src[:, :nw] = wd[:, b0:b0 + nw]
            out = torch.full((ns, pout), GUARD, dtype=torch.int32, device=dev)
            rc = L.uc_dfsdm_sinc5_streams(e._h, C.c_void_p(src.data_ptr()), ns, nw, pin, C.c_void_p(hist.data_ptr()),
                                          C.c_void_p(out.data_ptr()), pout, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                print("case %d: rc %d %s" % (case, rc, L.uc_last_error()))
                ok = False
                break
            torch.cuda.synchronize()
            o = out.cpu().numpy()
            if not (o[:, nw:] == GUARD).all():
                print("case %d: %d streams x %d words stored past the end of a row" % (case, ns, nw))
                ok = False
            got.append(o[:, :nw])
            b0 += nw
        hist_end = hist.cpu().numpy().view(np.uint32)
    else:
        hist = hist0.copy()
        for nw in chunks:
            got.append(e.dfsdm_streams(np.ascontiguousarray(w[:, b0:b0 + nw]), hist))
            b0 += nw
        hist_end = hist
    if ok:
        g = np.concatenate(got, axis=1)
        for s in check:
            if not np.array_equal(g[s], ref[s]):
                i = int(np.nonzero(g[s] != ref[s])[0][0])
                print("case %d: %d streams, chunks %s, %s: stream %d differs from word %d on" %
                      (case, ns, chunks, "device" if on_device else "host", s, i))
                ok = False
                break
        if not np.array_equal(hist_end, np.concatenate([hist0, w], axis=1)[:, -4:]):
            print("case %d: history not brought up to date" % case)
            ok = False
    bad += 0 if ok else 1
    if (case + 1) % 20 == 0:
        print("%d cases, %d failures" % (case + 1, bad), flush=True)
print("done: %d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
