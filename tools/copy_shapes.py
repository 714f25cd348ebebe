#!/usr/bin/env python3
"""Copy rate of tools/hbm_probe.hip at the footprints of the streaming kernels (read + write bytes / time)."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "tools", "libhbm_probe.so"))
L.hbm_probe_copy_shape.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
L.hbm_probe_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
n = 1 << 30
a = torch.empty(n // 4, dtype=torch.float32, device="cuda:0").normal_()
b = torch.empty_like(a)
st = torch.cuda.current_stream()
def timed(fn):
    ts = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); fn(); e1.record(st); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts[2:])[5]
for name, shape, blocks in (("1024 thr x 2 in flight, 1 block/CU", 0, 256), ("1024 x 4, 1/CU", 1, 256), ("1024 x 8, 1/CU", 2, 256),
                            ("256 x 2, 4/CU", 3, 1024), ("256 x 2, 8/CU", 3, 2048)):
    t = timed(lambda: L.hbm_probe_copy_shape(a.data_ptr(), b.data_ptr(), n, blocks, shape, st.cuda_stream))
    print("%-38s %.3f ms  %.0f GB/s" % (name, t, 2 * n / t / 1e6))
L.hbm_probe_copy_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
for name, al in (("sinc5's tiles: 1008-byte stride, 1024 thr, 1/CU", 0), ("the same with 1024-byte tiles", 1)):
    t = timed(lambda: L.hbm_probe_copy_tiles(a.data_ptr(), b.data_ptr(), n, 256, al, st.cuda_stream))
    print("%-48s %.3f ms  %.0f GB/s" % (name, t, 2 * n / t / 1e6))
L.hbm_probe_copy_runs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
for run in (2, 4, 8, 16, 64, 256):
    t = timed(lambda: L.hbm_probe_copy_runs(a.data_ptr(), b.data_ptr(), n, 256, run, st.cuda_stream))
    print("%-48s %.3f ms  %.0f GB/s" % ("1024-byte tiles, runs of %d per wave" % run, t, 2 * n / t / 1e6))
for run in (8, 16):
    t = timed(lambda: L.hbm_probe_copy_runs(a.data_ptr() + 16, b.data_ptr(), n - 1024, 256, run, st.cuda_stream))
    print("%-48s %.3f ms  %.0f GB/s" % ("runs of %d, loads 16 bytes off the line grid" % run, t, 2 * n / t / 1e6))
    t = timed(lambda: L.hbm_probe_copy_runs(a.data_ptr(), b.data_ptr() + 16, n - 1024, 256, run, st.cuda_stream))
    print("%-48s %.3f ms  %.0f GB/s" % ("runs of %d, STORES 16 bytes off the line grid" % run, t, 2 * n / t / 1e6))
t = timed(lambda: L.hbm_probe_copy(a.data_ptr(), b.data_ptr(), n, 2048, st.cuda_stream))
print("%-38s %.3f ms  %.0f GB/s" % ("256 x 8, 8/CU (bench probe)", t, 2 * n / t / 1e6))
