"""Transmitter / wire format.

Restates generator/ChirpGenerator.ipynb cells 1-3 and simulation/signal.py:29-56:
fs_tx = 44100 Hz, one symbol = T = 0.0262 s -> int(T*fs) = 1155 samples,
symbol = A*(cos(arg) + sin(arg)) ("orthogonal chirp"), arg = 2*pi*f(t)*t - pi/2,
f(t) = f0 + k*t/2 (H, bit 1) or f1 - k*t/2 (L, bit 0), k = (f1 - f0)/T,
t = linspace(0, T, 1155) INCLUDING the endpoint (signal.py:30).
Frame = G, 7 x H (preamble), L (delimiter), data bits MSB first, 12 x G (guard).
Pinned bit-exactly by the sha256 of generator/ChirpTone.wav (tests/golden, K7).
"""
import numpy as np

FS_TX = 44100
T_SYMBOL = 0.0262
F0, F1 = 16000, 19000
AMPLITUDE = 20000
N_PREAMBLE, N_GUARD = 7, 12


def symbol(updown, fs=FS_TX, f0=F0, f1=F1, T=T_SYMBOL, A=AMPLITUDE):
    """One orthogonal-chirp symbol as float64 (Signal.chirp_orth, signal.py:45-53)."""
    t = np.linspace(0, T, int(T * fs))
    k = float(f1 - f0) / float(T)
    f = f0 + k * t / 2.0 if updown == "up" else f1 - k * t / 2.0
    arg = (2.0 * np.pi * f * t) - np.pi / 2.0
    return (np.cos(arg) + np.sin(arg)) * A


def bits_of(msg):
    """MSB-first bits of the ASCII message (ChirpGenerator.ipynb cell 1, `ascii`)."""
    out = []
    for ch in msg.encode("ascii"):
        for i in range(8):
            out.append(1 if (ch & (0b10000000 >> i)) else 0)
    return np.array(out, dtype=np.uint8)


def symbol_sequence(msg):
    """-1 = silence (G), 1 = H (up), 0 = L (down): G, 7 H, L, data, 12 G."""
    return np.concatenate([[-1], np.ones(N_PREAMBLE, int), [0], bits_of(msg).astype(int),
                           -np.ones(N_GUARD, int)]).astype(int)


def tone(msg="Hello World!", fs=FS_TX):
    """The transmit waveform as float64, exactly as cell 3 builds it."""
    H, L = symbol("up", fs), symbol("down", fs)
    G = np.zeros(int(T_SYMBOL * fs))
    return np.concatenate([{1: H, 0: L, -1: G}[int(s)] for s in symbol_sequence(msg)])


def tone_int16(msg="Hello World!"):
    """What Signal.play() writes to ChirpTone.wav: real(wave).astype(int16)."""
    return tone(msg).astype(np.int16)


def render(msg="Hello World!", fs_rx=78125.0, amplitude=AMPLITUDE, lead=0.0):
    """The same continuous-time waveform sampled at the receiver's rate (no resampling
    filter: the chirp law is evaluated at the receiver's sample instants).

    In the WAV, sample i of a symbol is the law at t = i*T/(n-1) (linspace endpoint) and is
    played at i/fs_tx; a receiver sample at real time tau inside the symbol therefore sees
    the law at t = tau * fs_tx * T/(n-1).  `lead` seconds of silence are prepended.
    """
    n_sym = int(T_SYMBOL * FS_TX)                  # 1155
    sym_dur = n_sym / float(FS_TX)                 # real duration of one symbol
    seq = symbol_sequence(msg)
    total = lead + len(seq) * sym_dur
    n = int(np.floor(total * fs_rx))
    tt = np.arange(n) / fs_rx - lead
    idx = np.floor((tt + 1e-10) / sym_dur).astype(int)  # (1e-10 s guards exact symbol boundaries)
    valid = (idx >= 0) & (idx < len(seq))
    tau = np.maximum(tt - idx * sym_dur, 0.0)
    t = tau * FS_TX * T_SYMBOL / (n_sym - 1)
    k = float(F1 - F0) / T_SYMBOL
    kind = np.where(valid, seq[np.clip(idx, 0, len(seq) - 1)], -1)
    f = np.where(kind == 1, F0 + k * t / 2.0, F1 - k * t / 2.0)
    arg = 2.0 * np.pi * f * t - np.pi / 2.0
    out = (np.cos(arg) + np.sin(arg)) * amplitude
    return np.where(kind >= 0, out, 0.0)
