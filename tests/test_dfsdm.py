"""DFSDM front end: sinc^5 / 32 of the 1-bit PDM stream, integer and bit-exact.

The peripheral is silicon (receiver/Src/dfsdm.c:59-61,69,78 only configure it), so the reference holds no
vector for it: the oracle's Hogenauer restatement is pinned against the direct 156-tap convolution in numpy
(CPU tests) and the HIP kernel against the oracle, bit for bit (GPU tests, through the C-ABI)."""
import numpy as np
import pytest

from uchirp import synth
from oracle import uco


def numpy_sinc5(words):
    w = np.asarray(words, np.uint32)
    bits = ((w[:, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(np.int64).reshape(-1) * 2 - 1
    h = np.ones(1, np.int64)
    for _ in range(5):
        h = np.convolve(h, np.ones(32, np.int64))
    assert h.size == 156 and h.sum() == 1 << 25
    y = np.convolve(bits, h)[np.arange(4, w.size) * 32 + 31]
    return (np.clip(y >> 2, -(1 << 23), (1 << 23) - 1) * 256).astype(np.int32)


def patterns(n, seed=0):
    rng = np.random.default_rng(seed)
    yield "random", rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    yield "ones", np.full(n, 0xFFFFFFFF, np.uint32)
    yield "zeros", np.zeros(n, np.uint32)
    yield "alternating", np.full(n, 0xAAAAAAAA, np.uint32)
    step = np.zeros(n, np.uint32)
    step[n // 2:] = 0xFFFFFFFF
    yield "step", step
    sparse = np.zeros(n, np.uint32)
    sparse[::7] = 1 << (np.arange(sparse[::7].size) % 32).astype(np.uint32)
    yield "single bits", sparse


def pdm_of_symbols(bits, amp=0.6, snr_db=None, seed=3):
    """Chirp symbols rendered at the PDM bit rate (2.5 MHz = 32 x 78125) and delta-sigma modulated."""
    n = 2048 * 32
    up, down = synth.chirp_pair(n=n, fs=78125.0 * 32, amp=1.0)   # same sweep, 32x oversampled
    x = np.concatenate([np.zeros(4 * 32)] + [(up if b else down) for b in bits]) * (amp / np.sqrt(2.0))
    if snr_db is not None:
        x = x + np.random.default_rng(seed).standard_normal(x.size) * amp * 10.0 ** (-snr_db / 20.0)
    return uco.pdm_modulate(np.clip(x, -1.0, 1.0).astype(np.float32))


# --------------------------------------------------------------------------- CPU

def test_oracle_sinc5_equals_the_direct_convolution():
    for name, w in patterns(300):
        assert np.array_equal(uco.dfsdm_sinc5(w), numpy_sinc5(w)), name
    assert (uco.dfsdm_sinc5(np.full(9, 0xFFFFFFFF, np.uint32)) == (2 ** 23 - 1) * 256).all()   # the one clipped value
    assert (uco.dfsdm_sinc5(np.zeros(9, np.uint32)) == -(2 ** 23) * 256).all()
    assert uco.dfsdm_sinc5(np.zeros(4, np.uint32)).size == 0
    out = uco.dfsdm_sinc5(next(patterns(64))[1])
    assert not (out & 0xFF).any()                      # 24-bit result in bits 31:8, as the agent/ *.raw captures


def test_oracle_sinc5_is_word_shift_invariant_and_chunks_continue():
    _, w = next(patterns(500, seed=5))
    full = uco.dfsdm_sinc5(w)
    assert np.array_equal(uco.dfsdm_sinc5(w[37:]), full[37:])
    a, b = uco.dfsdm_sinc5(w[:200]), uco.dfsdm_sinc5(w[196:])
    assert np.array_equal(np.concatenate([a, b]), full)


def test_pdm_front_end_feeds_the_receiver_oracle():
    """microphone bit stream -> DFSDM words -> dsp(): the decoded symbols are the transmitted ones."""
    bits = np.random.default_rng(1).integers(0, 2, size=12)
    words = uco.dfsdm_sinc5(pdm_of_symbols(bits))
    assert words.size == 12 * 2048
    o = uco.Oracle(uco.RX_REAL, mag_mean=float(np.abs(words).mean()) * 4)
    sym, st = o.process(words.reshape(12, 2048))
    assert np.array_equal(sym, bits)


# --------------------------------------------------------------------------- GPU

@pytest.fixture(scope="module")
def uchirp():
    import uchirp as m
    m.lib()
    return m


@pytest.mark.gpu
def test_sinc5_kernel_is_bit_exact(uchirp):
    e = uchirp.Engine(uchirp.RX_REAL)
    for n in (0, 4, 5, 6, 8, 255, 256, 257, 260, 4 + 252, 4 + 253, 4 + 4 * 252, 4 + 4 * 252 + 1, 5000, 70001):
        for name, w in patterns(n, seed=n):
            got = e.dfsdm(w)
            ref = uco.dfsdm_sinc5(w)
            assert got.shape == ref.shape and np.array_equal(got, ref), "%s n=%d" % (name, n)


@pytest.mark.gpu
def test_sinc5_ragged_tails_on_the_device_never_write_past_the_end(uchirp):
    """The kernel's stores are 16-byte buffer stores through a resource that covers exactly the tile's outputs: a lane
    whose four words straddle the end of the stream is clipped PER DWORD by the resource, lane 0 (whose outputs belong
    to the previous tile) sits at offset -16 = 0xFFFFFFF0 and is dropped whole.  Output counts with n_out % 4 in
    {1, 2, 3}, at and around wave-tile (252 outputs) and workgroup-tile boundaries, into a device buffer with guard
    words behind it: every output equals the oracle's, no guard word is touched."""
    import torch
    dev = torch.device("cuda:0")
    e = uchirp.Engine(uchirp.RX_REAL)
    tile = 252
    wg = tile * 16                                     # outputs per workgroup tile (16 waves)
    rng = np.random.default_rng(9)
    GUARD = -1234567
    for n_out in (1, 2, 3, 5, 6, 7, tile - 3, tile - 2, tile - 1, tile + 1, tile + 2, tile + 3, 2 * tile + 1, 3 * tile - 1,
                  wg - 1, wg + 1, wg + 2, wg + 3, 5 * wg + tile + 2, 70001, 70002, 70003):
        w = rng.integers(0, 1 << 32, size=n_out + 4, dtype=np.uint64).astype(np.uint32)
        ref = uco.dfsdm_sinc5(w)
        wd = torch.from_numpy(w.view(np.int32)).to(dev)
        buf = torch.full((n_out + 64,), GUARD, dtype=torch.int32, device=dev)
        e.dfsdm(wd, out=buf[:n_out])
        torch.cuda.synchronize()
        got = buf.cpu().numpy()
        assert np.array_equal(got[:n_out], ref), n_out
        assert (got[n_out:] == GUARD).all(), "n_out=%d: stored past the end" % n_out


@pytest.mark.gpu
def test_sinc5_device_path_at_scale_and_chunked(uchirp):
    import torch
    dev = torch.device("cuda:0")
    e = uchirp.Engine(uchirp.STREAM)          # any context will do
    n = (1 << 26) + 4                          # 256 MiB of PDM bits
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    w = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), generator=gen, device=dev, dtype=torch.int64).to(torch.int32)
    full = e.dfsdm(w)
    torch.cuda.synchronize()
    assert full.numel() == n - 4
    # head and tail against the oracle, bit for bit
    host = w.cpu().numpy().view(np.uint32)
    assert np.array_equal(full[:100000].cpu().numpy(), uco.dfsdm_sinc5(host[:100004]))
    assert np.array_equal(full[-100000:].cpu().numpy(), uco.dfsdm_sinc5(host[-100004:]))
    # word-shift invariance and chunk continuation, exactly, on the whole stream
    k = 4 * 1001
    assert torch.equal(e.dfsdm(w[k:].contiguous()), full[k:])
    cut = 4 * 3000001
    a, b = e.dfsdm(w[:cut + 4].contiguous()), e.dfsdm(w[cut:].contiguous())
    assert torch.equal(torch.cat([a, b]), full)
    assert int((full & 0xFF).abs().max()) == 0
    with pytest.raises(uchirp.UchirpError):
        e.dfsdm(w[1:100])                      # device pointer not 16-byte aligned


@pytest.mark.gpu
def test_pdm_to_symbols_on_the_gpu(uchirp):
    """The whole front of the receiver on the device: PDM bits -> uc_dfsdm_sinc5 -> uc_process_batch (int32
    DFSDM words) -> symbols, against the oracle chain and against the transmitted bits."""
    bits = np.random.default_rng(2).integers(0, 2, size=40)
    pdm = pdm_of_symbols(bits, snr_db=6.0)
    e = uchirp.Engine(uchirp.RX_REAL)
    words = e.dfsdm(pdm)
    ref_words = uco.dfsdm_sinc5(pdm)
    assert np.array_equal(words, ref_words)
    mm = float(np.abs(words).mean()) * 4
    e = uchirp.Engine(uchirp.RX_REAL, mag_mean=mm)
    o = uco.Oracle(uco.RX_REAL, mag_mean=mm)
    gs, _ = e.process(words.reshape(-1, 2048))
    rs, _ = o.process(ref_words.reshape(-1, 2048))
    assert np.array_equal(gs, rs) and np.array_equal(gs, bits)
