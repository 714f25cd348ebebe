"""Scale coverage: 64 Ki frames at -10 dB through every batch variant, EVERY record --
symbols, window magnitudes, peak indices -- against the float64 oracle (this box's CPU share), with the usual bars:
magnitudes within MAG_TOL x the frame's largest window magnitude, every index mismatch a proven near-tie, symbols
exact on every clear frame.  The 1 Mi-frame symbol gate of the bench's own batch is in test_gpu_parity.py."""
import numpy as np
import pytest

from uchirp import synth
from oracle import uco
from parity_util import MAG_TOL, check_history, clear_symbols

pytestmark = pytest.mark.gpu

N_FRAMES = 1 << 16


@pytest.fixture(scope="module")
def uchirp():
    import uchirp as m
    m.lib()
    return m


def _threads():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from bench import host_cpu_share
    return host_cpu_share()


def _frames(seed, dtype=np.float32, **kw):
    parts, bits = [], []
    for c in range(N_FRAMES // 8192):
        f, b = synth.make_frames(8192, seed=seed * 100 + c, snr_db=-10.0, dtype=dtype, **kw)
        parts.append(f)
        bits.append(b)
    return np.concatenate(parts), np.concatenate(bits)


@pytest.mark.parametrize("variant,dtype", [(uco.SYNC_CPLX, np.float32), (uco.RX_REAL, np.int32), (uco.SYNC_CPLX, np.int32)])
def test_64ki_frames_two_history_variants(uchirp, variant, dtype):
    import torch
    frames, bits = _frames(11 + variant, dtype=dtype)
    mm = 1000.0 * (256.0 if dtype == np.int32 else 1.0)
    o = uco.Oracle(variant, mag_mean=mm)
    e = uchirp.Engine(variant, mag_mean=mm)
    gs, gst = e.process(torch.from_numpy(frames).to("cuda:0"))
    torch.cuda.synchronize()
    gs, gst = gs.cpu().numpy(), uchirp.stats_from_tensor(gst)
    rs, rst = o.process(frames, precision=uco.F64, threads=_threads())
    clear = clear_symbols(rst)
    assert clear.mean() >= 0.995
    assert np.array_equal(gs[clear], rs[clear])
    ties = 0
    for h in (0, 1):
        ties += check_history(o, lambda f: frames[f], gst[:, h], rst[:, h], h, "variant %d hist%d" % (variant, h))
        np.testing.assert_array_equal(gst[:, h]["mag_mean"], rst[:, h]["mag_mean"])
    assert ties <= 0.01 * N_FRAMES
    print("variant %d %s: %d frames, %d unclear symbols, %d proven index near-ties" % (variant, dtype.__name__, N_FRAMES,
                                                                                     int((~clear).sum()), ties))


@pytest.mark.parametrize("variant", [uco.RX_REAL, uco.SYNC_CPLX])
def test_64ki_frames_wide_windows_at_41_7_khz(uchirp, variant):
    """The WIDE build of the band kernel (windows of 294 bins: the DFSDM rate of the reference's vacuum-cleaner captures)
    at the same scale and bars, matched sweep."""
    import torch
    fs = 125000.0 / 3.0
    frames, bits = _frames(51 + variant, fs=fs)
    kw = dict(fs=fs, time_frame=2048.0 / fs, mag_mean=1000.0)
    o = uco.Oracle(variant, **kw)
    e = uchirp.Engine(variant, **kw)
    assert e.bandwidth2 == 294
    gs, gst = e.process(torch.from_numpy(frames).to("cuda:0"))
    torch.cuda.synchronize()
    gs, gst = gs.cpu().numpy(), uchirp.stats_from_tensor(gst)
    rs, rst = o.process(frames, precision=uco.F64, threads=_threads())
    clear = clear_symbols(rst)
    assert clear.mean() >= 0.995
    assert np.array_equal(gs[clear], rs[clear])
    assert (gs == bits).mean() > 0.99
    ties = 0
    for h in (0, 1):
        ties += check_history(o, lambda f: frames[f], gst[:, h], rst[:, h], h, "wide variant %d hist%d" % (variant, h))
    assert ties <= 0.01 * N_FRAMES
    print("wide variant %d: %d frames, %d unclear symbols, %d proven index near-ties" % (variant, N_FRAMES, int((~clear).sum()), ties))


def test_64ki_frames_dechirp_down(uchirp):
    import torch
    frames, _ = _frames(31, fs=100000.0, f0=17000.0, f1=18000.0)
    o = uco.Oracle(uco.DECHIRP_DOWN, mag_mean=1000.0)
    e = uchirp.Engine(uchirp.DECHIRP_DOWN, mag_mean=1000.0)
    gs, gst = e.process(torch.from_numpy(frames).to("cuda:0"))
    torch.cuda.synchronize()
    gst = uchirp.stats_from_tensor(gst)
    rs, rst = o.process(frames, precision=uco.F64, threads=_threads())
    ties = check_history(o, lambda f: frames[f], gst[:, 0], rst[:, 0], 0, "dechirp_down", raw_idx=True)
    assert np.allclose(gst[:, 0]["snr"], rst[:, 0]["snr"], rtol=1e-4, atol=1e-4)
    assert ties <= 0.01 * N_FRAMES
    print("dechirp_down: %d frames, %d proven index near-ties" % (N_FRAMES, ties))


def test_64ki_frames_compress(uchirp):
    import torch
    rng = np.random.default_rng(5)
    o = uco.Oracle(uco.COMPRESS, mag_mean=1.0)
    e = uchirp.Engine(uchirp.COMPRESS, mag_mean=1.0)
    up = o.table(uco.TABLE_UP).astype(np.float32)
    parts = []
    for c in range(N_FRAMES // 8192):   # circularly shifted up chirps at -10 dB
        sh = rng.integers(0, 2048, size=8192)
        idx = (np.arange(2048)[None, :] - sh[:, None]) % 2048
        parts.append((1000.0 * up[idx] + 1000.0 * 10 ** 0.5 * rng.standard_normal((8192, 2048))).astype(np.float32))
    frames = np.concatenate(parts)
    gs, gst = e.process(torch.from_numpy(frames).to("cuda:0"))
    torch.cuda.synchronize()
    g = uchirp.stats_from_tensor(gst)[:, 0]
    rs, rst = o.process(frames, precision=uco.F64, threads=_threads())
    r = rst[:, 0]
    scale = np.abs(r["mag_max"].astype(np.float64))
    assert (np.abs(g["mag_max"].astype(np.float64) - r["mag_max"]) / scale).max() <= MAG_TOL
    bad = np.nonzero(g["max_freq"] != r["max_freq"])[0]
    for f in bad:  # every index mismatch must be a near-tie in the oracle's own compressed signal
        y = o.spectrum(frames[f])[0]
        assert y.max() - y[g["max_freq"][f]] <= MAG_TOL * abs(y.max()), f
    assert len(bad) <= 0.01 * N_FRAMES
    print("compress: %d frames, %d proven index near-ties" % (N_FRAMES, len(bad)))


@pytest.mark.parametrize("n", [1024, 2048])
def test_64ki_frames_iq_baseband(uchirp, n):
    import torch
    from test_gpu_iq_baseband import BB, iq_stream
    nf = N_FRAMES // 2
    x, bits = iq_stream(nf, n, sigma=1000.0 * 10 ** 0.5, seed=77)
    cfg = dict(BB, n=n, time_frame=n / BB["fs"], flags=uco.FLAG_IQ_BASEBAND, mag_mean=1000.0)
    o, e = uco.Oracle(uco.IQ, **cfg), uchirp.Engine(uchirp.IQ, **cfg)
    gs, gst = e.process(torch.from_numpy(x).to("cuda:0"), n_frames=nf)
    torch.cuda.synchronize()
    gs, gst = gs.cpu().numpy(), uchirp.stats_from_tensor(gst)
    rs, rst = o.process(x, halo=26, n_frames=nf, precision=uco.F64, threads=_threads())
    clear = clear_symbols(rst)
    assert np.array_equal(gs[clear], rs[clear]) and clear.mean() >= 0.995
    ties = 0
    for h in (0, 1):
        ties += check_history(o, lambda f: x[f * n: f * n + n + 26], gst[:, h], rst[:, h], h, "iq bb n=%d hist%d" % (n, h),
                              spectrum_kw={"halo": 26})
    assert ties <= 0.01 * nf
    print("iq base band n=%d: %d frames, BER %.4f, %d proven index near-ties" % (n, nf, float((gs != bits).mean()), ties))


_LIN = [("rx_real", uco.RX_REAL, {}), ("sync_cplx", uco.SYNC_CPLX, {}), ("dechirp_down", uco.DECHIRP_DOWN, {}),
        ("compress", uco.COMPRESS, {}), ("iq", uco.IQ, {}), ("iq1024", uco.IQ, {"n": 1024}),
        ("iq1024_bb", uco.IQ, dict(n=1024, fs=100000.0, carrier=18000.0, f0=16500.0, f1=19500.0, time_frame=1024 / 100000.0,
                                   flags=uco.FLAG_IQ_BASEBAND)),
        ("iq_bb", uco.IQ, dict(fs=100000.0, carrier=18000.0, f0=16500.0, f1=19500.0, time_frame=2048 / 100000.0,
                               flags=uco.FLAG_IQ_BASEBAND)),
        ("rx_real_41_7_khz", uco.RX_REAL, dict(fs=125000.0 / 3.0, time_frame=2048 * 3.0 / 125000.0))]


@pytest.mark.parametrize("name,variant,kw", _LIN, ids=[c[0] for c in _LIN])
def test_exact_x2_linearity_and_shard_invariance_of_every_frame_kernel(uchirp, name, variant, kw):
    """Oracle-free properties of every frame kernel on 128 Ki frames at -10 dB: doubling the input AND the noise floor
    doubles every magnitude exactly (a power of two commutes with every rounding of the pipeline), leaves every index,
    snr and symbol bit-identical; a batch cut in two (what a second GPU would get) gives the records of the whole."""
    import torch
    dev = torch.device("cuda:0")
    nf = 1 << 17
    frames, _ = synth.device_frames(nf, dev, seed=808, snr_db=-10.0)
    flat = frames.reshape(-1)
    e1 = uchirp.Engine(variant, mag_mean=1000.0, **kw)
    e2 = uchirp.Engine(variant, mag_mean=2000.0, **kw)
    n, halo = e1.n, e1.halo
    n_frames = (flat.numel() - halo - n) // n + 1
    s1, t1 = e1.process(flat, n_frames=n_frames)
    s2, t2 = e2.process(flat * 2.0, n_frames=n_frames)
    torch.cuda.synchronize()
    a, b = uchirp.stats_from_tensor(t1), uchirp.stats_from_tensor(t2)
    for fld in ("mag_max", "mag_max_left", "mag_max_right", "mag_mean"):
        np.testing.assert_array_equal(b[fld], 2.0 * a[fld], err_msg="%s %s" % (name, fld))
    for fld in ("max_freq", "max_freq_left", "max_freq_right"):
        np.testing.assert_array_equal(b[fld], a[fld], err_msg="%s %s" % (name, fld))
    np.testing.assert_array_equal(b["snr"].view(np.int32), a["snr"].view(np.int32))
    assert torch.equal(s1, s2)
    # two shards == the whole (the second shard starts `halo` samples early: its FIR history is the first shard's tail)
    h = n_frames // 2
    lo_s, lo_t = e1.process(flat[: halo + h * n], n_frames=h)
    hi_s, hi_t = e1.process(flat[h * n:], n_frames=n_frames - h)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([lo_s, hi_s]), s1)
    assert torch.equal(torch.cat([lo_t, hi_t]).view(torch.int32), t1.view(torch.int32))


def test_exact_x2_linearity_of_the_stream_kernel(uchirp):
    import torch
    dev = torch.device("cuda:0")
    frames, _ = synth.device_frames(1 << 15, dev, seed=909, snr_db=-10.0)
    flat = frames.reshape(-1)
    e = uchirp.Engine(uchirp.STREAM)
    c1, p1 = e.process_stream(flat)
    c2, p2 = e.process_stream(flat * 2.0)
    torch.cuda.synchronize()
    assert torch.equal(c2, 2.0 * c1)
    assert torch.equal(p1[:, 1], p2[:, 1])                                   # peak offsets
    assert torch.equal(p2[:, 0].view(torch.float32), 2.0 * p1[:, 0].view(torch.float32))   # peak values


@pytest.mark.parametrize("name,variant,kw", _LIN, ids=[c[0] for c in _LIN])
def test_int32_words_equal_their_float_values_in_every_frame_kernel(uchirp, name, variant, kw):
    """The ISR's cast (receiver/Src/main.c:664) fused into the loads: DFSDM words (24-bit samples in bits 31:8) and the
    float32 values of the same words give bit-identical records -- every kernel, 128 Ki frames."""
    import torch
    dev = torch.device("cuda:0")
    frames, _ = synth.device_frames(1 << 17, dev, seed=707, snr_db=-10.0)
    words = (torch.round(frames).to(torch.int32) * 256).reshape(-1)
    e = uchirp.Engine(variant, mag_mean=1000.0 * 256.0, **kw)
    n, halo = e.n, e.halo
    n_frames = (words.numel() - halo - n) // n + 1
    si, ti = e.process(words, n_frames=n_frames)
    sf, tf = e.process(words.to(torch.float32), n_frames=n_frames)
    torch.cuda.synchronize()
    assert torch.equal(si, sf)
    assert torch.equal(ti.view(torch.int32), tf.view(torch.int32))


@pytest.mark.parametrize("name,variant,kw", [c for c in _LIN if c[0] in ("rx_real", "sync_cplx", "rx_real_41_7_khz")],
                         ids=["rx_real", "sync_cplx", "rx_real_41_7_khz"])
def test_a_frame_does_not_depend_on_its_neighbours_or_its_place(uchirp, name, variant, kw):
    """One frame per transform: the batch in a random order gives the same records in that order, bit for bit (which
    workgroup, which group, which ring slot a frame meets never shows in its result)."""
    import torch
    dev = torch.device("cuda:0")
    nf = 1 << 16
    frames, _ = synth.device_frames(nf, dev, seed=606, snr_db=-10.0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    perm = torch.randperm(nf, generator=g, device=dev)
    e = uchirp.Engine(variant, mag_mean=1000.0, **kw)
    s, t = e.process(frames)
    sp, tp = e.process(frames[perm].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(sp, s[perm])
    assert torch.equal(tp.view(torch.int32), t[perm].view(torch.int32))
