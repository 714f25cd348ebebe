#!/usr/bin/env python3
"""The DFSDM front end next to this chip's own copy rate, in ONE process on ONE box: tools/hbm_probe.hip's 1:1 copy (4 B in,
4 B out per word -- the traffic shape of sinc5), uc_dfsdm_sinc5 on one long stream, uc_dfsdm_sinc5_streams on many
microphones (one 2048-word block each per call: the live shape; and longer chunks), alternated three times.  HIP events,
median; the in-kernel shader clock of each form from the clock-stamped twin (uc_clock_probe).
Usage: python tools/run_cic.py [words_log2=28] [iters=10]   -> JSON lines"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
import torch
import uchirp

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 1 << lg
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(1)
w = torch.randint(-(1 << 31), (1 << 31) - 1, (n + 4,), generator=gen, device=dev, dtype=torch.int64).to(torch.int32)
out = torch.empty(n + 4, dtype=torch.int32, device=dev)
e = uchirp.Engine(uchirp.RX_REAL)
L = uchirp.lib()
stream = torch.cuda.current_stream(dev)
P = C.CDLL(os.path.join(ROOT, "tools", "libhbm_probe.so"))
P.hbm_probe_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
blocks = torch.cuda.get_device_properties(dev).multi_processor_count * 8


def timed(fn):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        fn()
        b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(sorted(ts)[len(ts) // 2])


def clock(fn):
    e.clock_probe(True)
    fn()
    c = e.clock_read()
    e.clock_probe(False)
    return round(c["shader_ghz"], 3)


def streams_call(ns, nw, hist):
    rc = L.uc_dfsdm_sinc5_streams(e._h, C.c_void_p(w.data_ptr()), ns, nw, 0, C.c_void_p(hist.data_ptr()),
                                  C.c_void_p(out.data_ptr()), 0, C.c_void_p(stream.cuda_stream))
    assert rc == 0, L.uc_last_error()


shapes = [(n // 2048, 2048), (n // 65536, 65536), (4096, 2048), (16, n // 16)]
rows = {}
for rep in range(3):
    t = timed(lambda: P.hbm_probe_copy(w.data_ptr(), out.data_ptr(), n * 4, blocks, stream.cuda_stream))
    rows.setdefault("copy", []).append(8 * n / t / 1e6)
    t = timed(lambda: e.dfsdm(w, out=out[:n]))
    rows.setdefault("sinc5 one stream", []).append(8 * n / t / 1e6)
    for ns, nw in shapes:
        hist = torch.full((ns, 4), -1431655766, dtype=torch.int32, device=dev)
        t = timed(lambda: streams_call(ns, nw, hist))
        rows.setdefault("sinc5_streams %d x %d" % (ns, nw), []).append(8 * ns * nw / t / 1e6)
copy = sorted(rows["copy"])[1]
clk = {"sinc5 one stream": clock(lambda: e.dfsdm(w, out=out[:n]))}
for ns, nw in shapes:
    hist = torch.full((ns, 4), -1431655766, dtype=torch.int32, device=dev)
    clk["sinc5_streams %d x %d" % (ns, nw)] = clock(lambda: streams_call(ns, nw, hist))
for name, v in rows.items():
    med = sorted(v)[1]
    print(json.dumps({"what": name, "GBs_algorithmic_median_of_3": round(med, 1), "runs": [round(x, 1) for x in v],
                      "G_words_per_s": round(med / 8.0, 2), "frac_of_copy_probe": round(med / copy, 3),
                      "frac_of_8TBs": round(med / 8000.0, 3), "shader_ghz_in_kernel": clk.get(name),
                      "words_per_launch": n if "x" not in name else int(name.split()[1]) * int(name.split()[3])}), flush=True)
