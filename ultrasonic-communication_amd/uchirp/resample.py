"""Rational sample-rate conversion, for feeding the transmitter's 44.1 kHz WAV (generator/ChirpTone.wav) to a
receiver that samples at 78 125 Hz -- what the air between a loudspeaker and the DFSDM does to it, and what SURVEY A3 / K9
did with scipy.signal.resample_poly before handing the stream to the reference's `synchronization` build.

Own code (numpy only): a polyphase windowed-sinc interpolator.  y[m] = sum_k x[k] h[m M - k L + c] with L = up, M = down,
h a Kaiser-windowed sinc cut off at the narrower of the two Nyquist bands, unity pass-band gain, c its centre (zero delay):
each output sample is a dot product of `2 * half + 1` input samples (at most) with one of L filter phases.
"""
from math import gcd

import numpy as np

FS_RX = 78125
FS_TX = 44100


def design(up, down, half=24, beta=9.0):
    """The prototype low-pass at the UP-sampled rate: 2 * half * max(up, down) + 1 taps, Kaiser window (beta 9: ~ -90 dB)."""
    q = max(up, down)
    n = np.arange(-half * q, half * q + 1, dtype=np.float64)
    h = np.sinc(n / q) * np.kaiser(n.size, beta)
    return h * (up / h.sum())          # unity gain after zero stuffing by `up`


def resample(x, up, down, half=24, beta=9.0):
    """x (1-d, real) at rate fs -> rate fs * up / down, ceil(len(x) * up / down) samples, zero delay (sample 0 stays sample 0)."""
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    g = gcd(int(up), int(down))
    up, down = int(up) // g, int(down) // g
    if up == down:
        return x.copy()
    h = design(up, down, half, beta)
    centre = (h.size - 1) // 2
    n_out = -(-x.size * up // down)
    m = np.arange(n_out, dtype=np.int64)
    pos = m * down + centre                      # index into the zero-stuffed, filtered sequence
    k_hi = pos // up                             # newest input sample under the filter
    phase = pos - k_hi * up                      # h index that meets it
    taps = (h.size - 1) // up + 1                # input samples a dot product can touch
    hp = np.concatenate([h, np.zeros(taps * up)])
    xp = np.concatenate([np.zeros(taps), x, np.zeros(taps)])
    y = np.zeros(n_out)
    for j in range(taps):                        # (a few dozen passes over the output: vectorised over m)
        y += hp[phase + j * up] * xp[np.clip(k_hi - j, -taps, x.size + taps - 1) + taps]
    return y


def wav_to_receiver(samples_int16, fs_tx=FS_TX, fs_rx=FS_RX):
    """The transmitter's int16 samples as the receiver's DFSDM would see them: fs_tx -> fs_rx (44 100 -> 78 125: 3125 / 1764)."""
    return resample(np.asarray(samples_int16, dtype=np.float64), int(fs_rx), int(fs_tx))
