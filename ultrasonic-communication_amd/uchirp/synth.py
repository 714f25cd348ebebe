"""Synthetic receiver input: what the bench, the tools and the tests feed the kernels.

Signal model = the transmitter's orthogonal chirp A*(cos(theta_b) + sin(theta_b))
(simulation/signal.py:45-53, generator/ChirpGenerator.ipynb cell 1) rendered at
the receiver's rate: one 2048-sample frame per symbol, sweep f0->f1 (up, bit 1)
or f1->f0 (down, bit 0) over the frame, plus white Gaussian noise of
sigma = A * 10^(-SNR/20).

  chirp_pair / make_frames        numpy, host (tests, small cases)
  device_frames                   BASELINE configs[1]: random bits, generated on the device
  iq_symbol_pair / iq_stream / device_iq_stream
                                  BASELINE configs[2]: a continuous real pass-band stream at
                                  fs = 100 kHz, one n-sample symbol after the other, x = A cos(2 pi (carrier - f_b(t)) t)
                                  with the base-band chirp f_b = -+1.5 kHz -> +-1.5 kHz (the notebook's modulator,
                                  simulation/IQ_modulation.ipynb cell 4; generator/ChirpGeneratorIQmodulation.ipynb cell 5),
                                  behind 26 zeros of FIR history
  hello_kinds / device_hello_frames / decode_hello
                                  BASELINE configs[4]: the K7 wire format (G, 7 x H, L, 96 data bits of
                                  "Hello World!", 12 x G: generator/ChirpGenerator.ipynb cells 1-3) repeated to
                                  fill the batch, one pre-aligned frame per symbol
"""
import numpy as np

from . import tx

N = 2048
FS_RX = 78125.0


def chirp_pair(n=N, fs=FS_RX, f0=16000.0, f1=19000.0, amp=1000.0, sweep_time=None):
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


def make_frames(n_frames, seed=1234, snr_db=None, n=N, amp=1000.0, dtype=np.float32, **kw):
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


def _fill_device(kinds, device, g, snr_db, amp, n):
    """frames[i] = table[kinds[i]] + noise from generator g, on the device in chunks (kinds: 0 down, 1 up, 2 silence)."""
    import torch
    up, down = chirp_pair(n=n, amp=amp)
    tab = torch.tensor(np.stack([down, up, np.zeros(n)]), dtype=torch.float32, device=device)
    n_frames = kinds.numel()
    frames = torch.empty((n_frames, n), dtype=torch.float32, device=device)
    sigma = amp * 10.0 ** (-snr_db / 20.0)
    chunk = 1 << 15
    for s in range(0, n_frames, chunk):
        e = min(n_frames, s + chunk)
        frames[s:e] = tab[kinds[s:e]]
        frames[s:e] += sigma * torch.randn((e - s, n), generator=g, device=device)
    return frames


def device_frames(n_frames, device, seed, snr_db=-10.0, amp=1000.0, n=N):
    """configs[1]: random up/down symbols + AWGN, generated on the device.  Returns (frames, bits uint8)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    bits = torch.randint(0, 2, (n_frames,), generator=g, device=device, dtype=torch.int64)
    return _fill_device(bits, device, g, snr_db, amp, n), bits.to(torch.uint8)


def iq_symbol_pair(n, fs=100000.0, carrier=18000.0, bw=3000.0, amp=1000.0, inverted=True):
    """(up, down) float64 pass-band symbols of n samples.  inverted: the notebook's modulator
    x = A cos(2 pi (carrier - f_b(t)) t), f_b the base-band chirp -bw/2 .. +bw/2 (up, bit 1) or back (down, bit 0);
    else a plain pass-band chirp around the carrier."""
    t = np.arange(n, dtype=np.float64) / fs
    k = bw / (n / fs)
    out = []
    for up in (True, False):
        fb = (-bw / 2 + k * t / 2.0) if up else (bw / 2 - k * t / 2.0)
        out.append(amp * np.cos(2 * np.pi * ((carrier - fb) if inverted else (carrier + fb)) * t))
    return out[0], out[1]


def iq_stream(n_frames, n, fs=100000.0, carrier=18000.0, bw=3000.0, amp=1000.0, sigma=0.0, seed=5, inverted=True):
    """Host version: (stream float32 [26 + n_frames * n], bits uint8); the first 26 samples are zeros (FIR history)."""
    rng = np.random.default_rng(seed)
    up, down = iq_symbol_pair(n, fs, carrier, bw, amp, inverted)
    bits = rng.integers(0, 2, n_frames).astype(np.uint8)
    x = np.concatenate([np.zeros(26)] + [up if b else down for b in bits])
    x[26:] += sigma * rng.standard_normal(x.size - 26)
    return x.astype(np.float32), bits


def device_iq_stream(n_frames, n, device, seed, snr_db=-10.0, amp=1000.0, fs=100000.0, carrier=18000.0, bw=3000.0):
    """configs[2] on the device: (stream float32 [26 + n_frames * n], bits uint8).  Noise as configs[1]:
    sigma = A 10^(-SNR/20)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    bits = torch.randint(0, 2, (n_frames,), generator=g, device=device, dtype=torch.int64)
    up, down = iq_symbol_pair(n, fs, carrier, bw, amp)
    tab = torch.tensor(np.stack([down, up]), dtype=torch.float32, device=device)
    x = torch.empty(26 + n_frames * n, dtype=torch.float32, device=device)
    x[:26] = 0.0
    body = x[26:].view(n_frames, n)
    sigma = amp * 10.0 ** (-snr_db / 20.0)
    chunk = 1 << 16
    for s0 in range(0, n_frames, chunk):
        e0 = min(n_frames, s0 + chunk)
        body[s0:e0] = tab[bits[s0:e0]]
        body[s0:e0] += sigma * torch.randn((e0 - s0, n), generator=g, device=device)
    return x, bits.to(torch.uint8)


def hello_kinds(msg="Hello World!"):
    """One transmission as frame kinds: 1 = H (up), 0 = L (down), 2 = G (silence).
    tx.symbol_sequence: G, 7 x H, L, data bits MSB first, 12 x G (117 frames for 12 characters)."""
    seq = tx.symbol_sequence(msg)
    return np.where(seq < 0, 2, seq).astype(np.int64)


def hello_kind_stream(first_frame, n_frames, msg="Hello World!"):
    """Kinds of frames [first_frame, first_frame + n_frames) of the endless repetition of the transmission."""
    one = hello_kinds(msg)
    return one[(first_frame + np.arange(n_frames, dtype=np.int64)) % one.size]


def device_hello_frames(first_frame, n_frames, device, seed, snr_db=-10.0, amp=1000.0, msg="Hello World!", n=N):
    """configs[4]: this rank's contiguous share [first_frame, first_frame + n_frames) of the repeated
    'Hello World!' transmission, one pre-aligned frame per symbol, AWGN as configs[1].
    Returns (frames, kinds uint8: 1 up, 0 down, 2 silence)."""
    import torch
    kinds = torch.from_numpy(hello_kind_stream(first_frame, n_frames, msg)).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return _fill_device(kinds, device, g, snr_db, amp, n), kinds.to(torch.uint8)


def decode_hello(symbols, msg_len=12):
    """Texts of the whole transmissions in a gathered symbol stream that starts at frame 0.
    The data bits of transmission q are frames 117 q + 9 ... (G, 7 x H, L in front); bits are packed MSB first
    as the receiver does (msg = (msg << 1) + bit, receiver/Src/main.c:523-537); a frame below the SNR threshold
    (0xFF) reads as bit 1 here, as `(snr_down > snr_up) ? 0 : 1` would."""
    per = 1 + tx.N_PREAMBLE + 1 + 8 * msg_len + tx.N_GUARD
    s = np.asarray(symbols, dtype=np.uint8)
    nt = s.size // per
    if nt == 0:
        return []
    data = s[: nt * per].reshape(nt, per)[:, 1 + tx.N_PREAMBLE + 1: 1 + tx.N_PREAMBLE + 1 + 8 * msg_len]
    bits = (data != 0).astype(np.uint8)
    by = np.packbits(bits.reshape(nt, msg_len, 8), axis=2, bitorder="big").reshape(nt, msg_len)
    return [bytes(r).decode("latin-1") for r in by]
