"""Synthetic frames for the parity tests and the bench.

Signal model = the transmitter's orthogonal chirp A*(cos(theta_b) + sin(theta_b))
(simulation/signal.py:45-53, generator/ChirpGenerator.ipynb cell 1) rendered at
the receiver's rate: one 2048-sample frame per symbol, sweep f0->f1 (up, bit 1)
or f1->f0 (down, bit 0) over the frame, plus white Gaussian noise of
sigma = A * 10^(-SNR/20).
"""
import numpy as np


def chirp_pair(n=2048, fs=78125.0, f0=16000.0, f1=19000.0, amp=1000.0, sweep_time=None):
    """(up, down) float64 arrays of the orthogonal chirp, t = i/fs."""
    t = np.arange(n, dtype=np.float64) / fs
    T = sweep_time if sweep_time else n / fs
    k = (f1 - f0) / T
    out = []
    for updown in ("up", "down"):
        f = f0 + k * t / 2.0 if updown == "up" else f1 - k * t / 2.0
        arg = 2.0 * np.pi * f * t - np.pi / 2.0
        out.append((np.cos(arg) + np.sin(arg)) * amp)
    return out[0], out[1]


def make_frames(n_frames, seed=1234, snr_db=None, n=2048, amp=1000.0, dtype=np.float32, **kw):
    """Returns (frames[n_frames, n], bits[n_frames]); bit 1 = up chirp."""
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2, size=n_frames, dtype=np.uint8)
    up, down = chirp_pair(n=n, amp=amp, **kw)
    x = np.where(bits[:, None] == 1, up[None, :], down[None, :])
    if snr_db is not None:
        sigma = amp * 10.0 ** (-snr_db / 20.0)
        x = x + sigma * rng.standard_normal((n_frames, n))
    if dtype == np.int32:
        # DFSDM words: 24-bit sample in bits 31:8 (agent/*.raw are multiples of 256)
        return (np.round(x).astype(np.int64) * 256).astype(np.int32), bits
    return x.astype(np.float32), bits
