"""What uc_process_batch delivers when the caller hands over HOST buffers (the PCIe-inclusive rate DESIGN.md section 6 quotes;
never the bench's `value`).  Pageable numpy frames and pinned frames, symbols back to the host."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultrasonic-communication_amd"))
import torch  # noqa: E402
import uchirp  # noqa: E402
from uchirp import synth  # noqa: E402

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
frames, _ = synth.make_frames(1024, seed=3, snr_db=-10.0)
host = np.tile(frames, (nf // 1024, 1))
pinned_t = torch.empty(host.shape, dtype=torch.float32).pin_memory()
pinned_t.copy_(torch.from_numpy(host))
pinned = pinned_t.numpy()
eng = uchirp.Engine(uchirp.RX_REAL, device=0, mag_mean=1000.0)
for name, buf in (("pageable numpy", host), ("pinned (torch pin_memory)", pinned)):
    eng.process(buf, want_stats=False)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        sym, _ = eng.process(buf, want_stats=False)
    dt = (time.perf_counter() - t0) / reps
    print("%-28s %8d frames  %7.2f ms  %.3e frames/s  %.1f GB/s of frames over PCIe" % (name, nf, dt * 1e3, nf / dt, nf * 8192 / dt / 1e9))

# the reference's own granularity: ONE frame per call (uc_process_frame = one dsp() pair + decision, host in / host out)
pcm = (np.round(frames[0]).astype(np.int64) * 256).astype(np.int32)
for _ in range(50):
    eng.process_frame(pcm, 1000.0)
t0 = time.perf_counter()
reps = 2000
for _ in range(reps):
    eng.process_frame(pcm, 1000.0)
dt = (time.perf_counter() - t0) / reps
print("uc_process_frame (host words in, symbol + 2 histories out, synchronous): %.1f us per call = %.0f frames/s" % (dt * 1e6, 1.0 / dt))
