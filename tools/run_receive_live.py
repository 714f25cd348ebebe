#!/usr/bin/env python3
"""Live operation of uc_receive_streams_next: every call brings ONE new 2048-sample block (26.2 ms of microphone signal) of
every stream, as the firmware's ISR does; ms per call, and how many such microphones one GPU serves in real time.
Usage: python tools/run_receive_live.py [streams=4096,65536] [blocks_per_call=1] [rx_real|sync_cplx|both]   -> JSON lines"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
import numpy as np
import torch
import uchirp
from uchirp import tx

N, FS, NB = 2048, 78125.0, 176
counts = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4096,65536").split(",")]
per_call = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = sys.argv[3] if len(sys.argv) > 3 else "both"
dev = torch.device("cuda:0")
L = uchirp.lib()
tone = torch.from_numpy(tx.render("Hello World!", fs_rx=FS, amplitude=2000.0).astype(np.float32)).to(dev)
for variant, vname in ((uchirp.SYNC_CPLX, "sync_cplx"), (uchirp.RX_REAL, "rx_real")):
    if only not in ("both", vname):
        continue
    eng = uchirp.Engine(variant)
    for ns in counts:
        g = torch.Generator(device=dev)
        g.manual_seed(ns)
        x = torch.randn((ns, NB * N), generator=g, device=dev) * 50.0
        lead = 40 * N + 777
        x[:, lead:lead + tone.numel()] += tone
        live = eng.live(ns)
        if os.environ.get("UC_KEEP") == "1":     # uc_rx_state_keep_previous: every chunk here is a buffer of its own, never rewritten
            live.keep_previous(True)
        cap = 16
        text = torch.zeros((ns, cap), dtype=torch.uint8, device=dev)
        ntext = torch.zeros(ns, dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream(dev)
        chunks = [x[:, b * N:(b + per_call) * N].contiguous() for b in range(0, NB, per_call)]
        got = [bytearray() for _ in range(min(ns, 8))]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for ch in chunks:
            rc = L.uc_receive_streams_next(eng._h, live._h, C.c_void_p(ch.data_ptr()), uchirp.DTYPE_F32, ch.shape[1], 0, None,
                                           C.c_void_p(text.data_ptr()), cap, C.c_void_p(ntext.data_ptr()), None, 0, None,
                                           C.c_void_p(stream.cuda_stream))
            assert rc == 0, L.uc_last_error()
            nt = ntext[:len(got)].cpu().numpy()            # (reading the counts back = one sync per call, as a live host does)
            tt = text[:len(got)].cpu().numpy()
            for s in range(len(got)):
                got[s] += bytes(tt[s, :nt[s]])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / len(chunks)
        ok = sum(1 for b in got if b"Hello World!" in bytes(b))
        print(json.dumps({"variant": vname, "streams": ns, "keep_previous": os.environ.get("UC_KEEP") == "1", "blocks_per_call": per_call, "calls": len(chunks), "ms_per_call": dt * 1e3,
                          "real_time_ms_per_call": per_call * N / FS * 1e3,
                          "headroom_x_real_time": per_call * N / FS / dt,
                          "microphones_served_in_real_time": int(ns * per_call * N / FS / dt),
                          "first_streams_decoding_hello_world": "%d of %d" % (ok, len(got)),
                          "what": "uc_receive_streams_next, one call per new block of every stream, device-resident chunks, "
                                  "texts read back after every call"}), flush=True)
        live.close()
        del x, chunks
    eng.close()
