#!/usr/bin/env python3
"""What the group API costs on top of the plain context, at world size 1 on real RCCL (all a one-GPU box can measure):
   * uc_group_receive_streams vs uc_receive_streams   -- 4096 recorded streams x 176 blocks, texts + counts gathered in place
   * uc_group_receive_streams_next vs uc_receive_streams_next -- the same streams live, one block per call
   * uc_group_process_stream vs uc_process_stream     -- UC_STREAM, 2^30 samples, peak records gathered in place
Usage: python tools/run_group_legs.py   -> JSON lines"""
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

N, FS = 2048, 78125.0
dev = torch.device("cuda:0")


def timed(fn, sync, reps):
    for _ in range(3):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def receive_legs(variant, ns=4096, nb=176, cap=64):
    tone = torch.from_numpy(tx.render("Hello World!", fs_rx=FS, amplitude=2000.0).astype(np.float32)).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(ns)
    x = torch.randn((ns, nb * N), generator=g, device=dev) * 50.0
    lead = 40 * N + 777
    x[:, lead:lead + tone.numel()] += tone
    text = torch.zeros((ns, cap), dtype=torch.uint8, device=dev)
    cnt = torch.zeros(ns, dtype=torch.int32, device=dev)
    L = uchirp.lib()
    import ctypes as C
    eng = uchirp.Engine(variant)
    grp = uchirp.Group(variant, devices=[0])
    st = torch.cuda.current_stream(dev)

    def plain():
        rc = L.uc_receive_streams(eng._h, C.c_void_p(x.data_ptr()), uchirp.DTYPE_F32, ns, nb * N, 0, None,
                                  C.c_void_p(text.data_ptr()), cap, C.c_void_p(cnt.data_ptr()), None, 0, None,
                                  C.c_void_p(st.cuda_stream))
        assert rc == 0, L.uc_last_error()

    t_plain = timed(plain, torch.cuda.synchronize, 5)
    want = text.cpu().numpy().copy()
    text.zero_()
    t_group = timed(lambda: grp.receive_streams([x], ns, nb * N, [text], cap, n_text=[cnt]),
                    lambda: (grp.synchronize(), torch.cuda.synchronize()), 5)
    same = bool(np.array_equal(text.cpu().numpy(), want))
    good = sum(1 for r, k in zip(text.cpu().numpy(), cnt.cpu().numpy()) if b"Hello World!" in bytes(r[:k]))
    out = {"leg": "recorded streams", "variant": int(variant), "streams": ns, "blocks_per_stream": nb,
           "uc_receive_streams_ms": t_plain * 1e3, "uc_group_receive_streams_ms": t_group * 1e3,
           "group_over_plain": t_group / t_plain, "blocks_per_s_through_the_group": ns * nb / t_group,
           "texts_equal": same, "streams_decoding": good}
    print(json.dumps(out), flush=True)
    # live: one new block of every stream per call
    live = uchirp.LiveStreams(eng, ns)
    state = grp.rx_state(0, ns)
    blk = [x[:, b * N:(b + 1) * N].contiguous() for b in range(nb)]
    k = [0]

    def plain_live():
        rc = L.uc_receive_streams_next(eng._h, live._h, C.c_void_p(blk[k[0] % nb].data_ptr()), uchirp.DTYPE_F32, N, 0, None,
                                       C.c_void_p(text.data_ptr()), cap, C.c_void_p(cnt.data_ptr()), None, 0, None,
                                       C.c_void_p(st.cuda_stream))
        assert rc == 0, L.uc_last_error()
        k[0] += 1

    t_pl = timed(plain_live, torch.cuda.synchronize, nb)
    k[0] = 0

    def group_live():
        grp.receive_streams([blk[k[0] % nb]], ns, N, [text], cap, n_text=[cnt], states=[state])
        k[0] += 1

    t_gl = timed(group_live, lambda: (grp.synchronize(), torch.cuda.synchronize()), nb)
    print(json.dumps({"leg": "live streams, one block per call", "variant": int(variant), "streams": ns,
                      "uc_receive_streams_next_ms": t_pl * 1e3, "uc_group_receive_streams_next_ms": t_gl * 1e3,
                      "group_over_plain": t_gl / t_pl, "real_time_ms_per_call": N / FS * 1e3,
                      "microphones_in_real_time_through_the_group": int(ns * (N / FS) / t_gl)}), flush=True)
    grp.rx_state_destroy(state)
    grp.close()
    eng.close()


def stream_leg(log2=30):
    eng = uchirp.Engine(uchirp.STREAM)
    grp = uchirp.Group(uchirp.STREAM, devices=[0])
    halo = eng.stream_geometry(0)[0]
    n = (1 << log2) + halo
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    x = torch.randn(n, generator=g, device=dev) * 1000.0
    _, n_out, n_blocks, hop = eng.stream_geometry(n)
    comp = torch.empty(n_out, dtype=torch.float32, device=dev)
    pk = torch.empty((n_blocks, 2), dtype=torch.int32, device=dev)
    t_plain = timed(lambda: eng.process_stream(x, compressed_out=comp, peaks_out=pk), torch.cuda.synchronize, 10)
    want = pk.cpu().numpy().copy()
    pk.zero_()
    t_group = timed(lambda: grp.process_stream([x], n, [pk], compressed=[comp]),
                    lambda: (grp.synchronize(), torch.cuda.synchronize()), 10)
    print(json.dumps({"leg": "UC_STREAM", "samples": n, "blocks": n_blocks, "uc_process_stream_ms": t_plain * 1e3,
                      "uc_group_process_stream_ms": t_group * 1e3, "group_over_plain": t_group / t_plain,
                      "samples_per_s_through_the_group": n / t_group,
                      "peaks_equal": bool(np.array_equal(pk.cpu().numpy(), want))}), flush=True)
    grp.close()
    eng.close()


if __name__ == "__main__":
    receive_legs(uchirp.RX_REAL)
    receive_legs(uchirp.SYNC_CPLX)
    stream_leg()
