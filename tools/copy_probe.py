#!/usr/bin/env python3
"""Achievable HBM rate of a plain device copy (torch clone of 4 GiB), sustained for a few seconds."""
import sys, time, torch
n = 1 << 30
a = torch.empty(n, dtype=torch.float32, device="cuda:0").normal_()
b = torch.empty_like(a)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for _ in range(5):
    b.copy_(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    b.copy_(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print("copy 4 GiB: %.3f ms, %.2f TB/s (read + write)" % (dt * 1e3, 2 * 4 * n / dt / 1e12))
