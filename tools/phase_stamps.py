#!/usr/bin/env python3
"""Per-phase cycle shares of the band kernel from the diagnostic build
(make -C ultrasonic-communication_amd libuchirp_stamps.so).  Read the SHARES, not the length."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["UCHIRP_LIB"] = os.path.join(ROOT, "ultrasonic-communication_amd", "libuchirp_stamps.so")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import uchirp
from bench import make_device_frames

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 19
nf = 1 << lg
dev = torch.device("cuda:0")
frames, _ = make_device_frames(nf, dev, seed=1)
sym = torch.empty(nf, dtype=torch.uint8, device=dev)
e = uchirp.Engine(int(os.environ.get("UC_VARIANT", "0")), mag_mean=1000.0)
dbg = torch.zeros(4096 * 2 * 10, dtype=torch.int64, device=dev)
e.process(frames, want_stats=False, symbols_out=sym)
os.environ["UC_DEBUG_PTR"] = str(dbg.data_ptr())
e.process(frames, want_stats=False, symbols_out=sym)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 2, 10).astype(np.float64)
d = d[d[:, 0, :].sum(axis=1) > 0]
names = ["P1 wait-x+mul+dft16", "P1 lds store", "B1 wait", "P2 (finalise+)load+tw+dft16", "B2 wait",
         "P2 store + B3", "P3 prefetch-issue + pruned pass", "B4 wait", "windows + res", "loop top"]
frames_per_block = nf / d.shape[0]
tot = d.sum(axis=2).mean(axis=0)
print("blocks %d, frames/block %.1f, cycles/frame/wave: wave0 %.0f wave1 %.0f" %
      (d.shape[0], frames_per_block, tot[0] / frames_per_block, tot[1] / frames_per_block))
for k in [9, 0, 1, 2, 3, 4, 5, 6, 7, 8]:
    m = d[:, :, k].mean(axis=0) / frames_per_block
    print("%-34s wave0 %7.0f (%4.1f%%)   wave1 %7.0f (%4.1f%%)" %
          (names[k], m[0], 100 * m[0] * frames_per_block / tot[0], m[1], 100 * m[1] * frames_per_block / tot[1]))
