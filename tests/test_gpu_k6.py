"""The HIP kernels against the reference's on-device captures (K6) DIRECTLY -- no oracle in between.

agent/chirp_experiment/* and agent/vaccum_cleaner/* hold what the STM32 itself computed with CMSIS-DSP
(experiments/basic/Src/main.c:107-174): `.raw` the DFSDM words, `.flt` = raw x Hann (arm_mult_f32 with the arm_cos_f32
window), `.fft` = |arm_rfft_fast_f32(flt)| / sqrt(N), bins below 1 kHz forced to 1.0.  They are the only real CMSIS-DSP
output the reference holds (tests/golden/k6_all.npz, copied by tests/golden/make_golden.py: arrays only).

  * uc_set_table(UP = DOWN = ones) turns RX_REAL -- the headline kernel family -- into exactly that chain:
    int32 ingest -> x Hann -> 2048-point real FFT -> magnitude; uc_window_spectrum returns the magnitudes of the
    bins its windows reach, which must be the device's at print precision.
  * SYNC_CPLX with a complex-exponential reference slides the same window over the WHOLE spectrum (shifts that
    cover bins 21 .. 1023), so every live bin of every capture meets a HIP kernel.
  * Both in TWO window geometries: 318 bins (the three-round WIDE build) and 190 bins -- the DEFAULT two-round build, the
    code of the kernel the bench times (uc_window_spectrum runs its instantiation with the bin stores added, and the
    statistics of the throughput instantiation are asserted to be the maxima of exactly those values, bit for bit).
  * The product's Hann table times the raw words must be the device's `.flt` to one float32 ulp (a3).

Also here: uc_window_spectrum / uc_set_table against the oracle on synthetic frames (per-bin parity, a5).
"""
import os

import numpy as np
import pytest

from uchirp import synth
from oracle import uco
from parity_util import MAG_TOL, check_history, clear_symbols

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
K6ALL = np.load(os.path.join(GOLD, "k6_all.npz"))
K6_MISMATCHED = "chirp_experiment/48.1(kHz)_M2A"      # the one trio whose files belong to different captures
N = 2048
# window geometry only (the tables are replaced): (f1 - f0) n / fs = 159.5 -> bandwidth 159, windows of 318 bins
GEOM = dict(fs=100000.0, f0=10000.0, f1=17788.0, mag_mean=1.0)
# (f1 - f0) n / fs = 95.2 -> bandwidth 95, windows of 190 bins: the default two-round build (bandwidth2 <= 191)
GEOM_DEFAULT = dict(fs=100000.0, f0=10000.0, f1=14650.0, mag_mean=1.0)
GEOMS = [pytest.param(GEOM, 318, id="wide318"), pytest.param(GEOM_DEFAULT, 190, id="default190")]
PRINT_STEP = 0.5e-6 * np.sqrt(N)                       # the device prints mag / sqrt(N) with six decimals


@pytest.fixture(scope="module")
def uchirp():
    import uchirp as m
    m.lib()
    return m


def _captures():
    names = [str(s) for s in K6ALL["names"]]
    for i, name in enumerate(names):
        raw = K6ALL["raw_%02d" % i].astype(np.int32)
        yield name, raw, K6ALL["flt_%02d" % i], K6ALL["fftfreq_%02d" % i], K6ALL["fftmag_%02d" % i] * np.sqrt(N)


def test_hann_table_times_raw_is_the_device_flt(uchirp):
    """a3: the product's own Hann table (uc_get_table) reproduces `.flt = .raw x hann` to one float32 ulp."""
    e = uchirp.Engine(uchirp.RX_REAL)
    hann = e.table(uchirp.TABLE_HANN)
    checked = 0
    for name, raw, flt, _, _ in _captures():
        got = (raw.astype(np.float32) * hann).astype(np.float64)
        tol = np.abs(raw).max() * 2.0 ** -23 + 1e-6        # one ulp of the product + the %f print step
        ok = np.abs(got - flt).max() <= tol
        if name == K6_MISMATCHED:
            continue
        assert ok, (name, np.abs(got - flt).max(), tol)
        checked += 1
    assert checked == 23


@pytest.mark.parametrize("geom,W", GEOMS)
def test_rx_real_with_unit_reference_is_the_devices_rfft_magnitude(uchirp, geom, W):
    """a1 + a3 + a5 of the HIP path on real CMSIS-DSP output: raw int32 words -> band_kernel<rx_real> with up = down =
    1 -> |X[k]|, k = 0 .. W, against `.fft x sqrt(N)` for every bin at or above 1 kHz; same arg-max.
    Tolerance: MAG_TOL x the frame's largest spectral magnitude (float32 FFT round-off scales with the DC region these
    captures are dominated by, 30 - 100 x the live peak) + the print step of the device's %f."""
    e = uchirp.Engine(uchirp.RX_REAL, **geom)
    assert e.bandwidth2 == W
    ones = np.ones(N, np.float32)
    e.set_table(uchirp.TABLE_UP, ones)
    e.set_table(uchirp.TABLE_DOWN, ones)
    caps = list(_captures())
    frames = np.stack([c[1] for c in caps])
    spec = e.window_spectrum(frames)                       # [24, 2, 2 W + 1]
    assert spec.shape == (len(caps), 2, 2 * W + 1)
    _, st = e.process(frames)
    checked = 0
    for i, (name, raw, flt, freq, mag_dev) in enumerate(caps):
        g = spec[i, 0, W:].astype(np.float64)              # bins 0 .. W
        # up == down reference: the same spectrum twice (one rides in the real, one in the imaginary part of the complex
        # transform: equal to round-off, not bit for bit)
        assert np.abs(spec[i, 0] - spec[i, 1]).max() <= 2e-6 * spec[i, 0].max()
        assert np.array_equal(spec[i, 0, :W][::-1], spec[i, 0, W + 1:])   # Hermitian mirror (Q1)
        live = np.nonzero(freq[:W + 1] >= 1000.0)[0]
        scale = g.max()                                    # (the DC region: this kernel's own largest bin)
        err = np.abs(g[live] - mag_dev[live])
        ok = err.max() <= MAG_TOL * scale + PRINT_STEP and np.argmax(g[live]) == np.argmax(mag_dev[live])
        if name == K6_MISMATCHED:
            assert not ok, "the mismatched trio is expected to disagree"
            continue
        assert ok, (name, err.max(), MAG_TOL * scale + PRINT_STEP, np.argmax(g[live]), np.argmax(mag_dev[live]))
        # the statistics of uc_process_batch -- at 190 bins band_kernel<rx_real, int32, 3 waves/SIMD>, the throughput
        # instantiation -- are the maxima of exactly these values, bit for bit, in both histories and both windows
        for h in (0, 1):
            assert st[i, h]["mag_max_right"] == spec[i, h, W:W + W].max()
            assert st[i, h]["mag_max_left"] == spec[i, h, :W].max()
        checked += 1
    assert checked == 23


def _slide(e, uchirp, shifts):
    """SYNC_CPLX references e^{-j 2 pi m t / n}, m = shifts[0] in the up slot, shifts[1] in the down slot."""
    t = np.arange(N, dtype=np.float64)
    for tid, m in zip((uchirp.TABLE_UP, uchirp.TABLE_DOWN), shifts):
        ph = -2.0 * np.pi * m * t / N
        e.set_table(tid, np.stack([np.cos(ph), np.sin(ph)], axis=1).astype(np.float32).reshape(-1))


@pytest.mark.parametrize("geom,W", GEOMS)
def test_every_live_bin_of_every_capture_through_the_complex_kernel(uchirp, geom, W):
    """band_kernel<sync_cplx> with reference e^{-j 2 pi m t / n}: Z[k] = X[k + m], so the +-W-bin window sits on bins
    m - W .. m + W of the capture's spectrum.  W = 318: m = 339 (up slot) and m = 705 (down slot) cover bins 21 .. 1023;
    W = 190 (the default two-round build): m = 211, 591, 833 in two passes.  Every bin the device printed at or above
    1 kHz, for all 23 consistent captures, plus the global arg-max."""
    e = uchirp.Engine(uchirp.SYNC_CPLX, **geom)
    assert e.bandwidth2 == W
    passes = [(339, 705)] if W == 318 else [(211, 591), (833, 833)]
    caps = list(_captures())
    frames = np.stack([c[1] for c in caps])
    specs, shifts = [], []
    for sh in passes:
        _slide(e, uchirp, sh)
        sp = e.window_spectrum(frames).astype(np.float64)          # [24, 2, 2 W + 1]
        for h in (0, 1):
            specs.append(sp[:, h])
            shifts.append(sh[h])
    top = max(float(np.nanmax(sp)) for sp in specs)
    checked = 0
    for i, (name, raw, flt, freq, mag_dev) in enumerate(caps):
        g = np.full(1024, np.nan)
        for sp, m in zip(specs, shifts):
            lo, hi = max(m - W, 0), min(m + W, 1023)
            seg = sp[i, lo - m + W: hi - m + W + 1]
            both = ~np.isnan(g[lo:hi + 1])
            # bins several shifts reach agree with each other to round-off
            assert np.all(np.abs(g[lo:hi + 1][both] - seg[both]) <= MAG_TOL * top)
            g[lo:hi + 1] = seg
        live = np.nonzero(freq >= 1000.0)[0]
        assert not np.isnan(g[live]).any()
        scale = np.abs(np.fft.rfft(raw.astype(np.float64) * np.hanning(N + 1)[:N])).max()    # the DC region (scale only)
        err = np.abs(g[live] - mag_dev[live])
        ok = err.max() <= MAG_TOL * scale + PRINT_STEP and np.argmax(g[live]) == np.argmax(mag_dev[live])
        if name == K6_MISMATCHED:
            assert not ok
            continue
        assert ok, (name, err.max(), MAG_TOL * scale + PRINT_STEP)
        checked += 1
    assert checked == 23


@pytest.mark.parametrize("name,kw", [("rx_real", {}), ("sync_cplx", {}), ("rx_real", dict(fs=125000.0 / 3.0)),
                                     ("dechirp_down", dict(fs=100000.0, f0=17000.0, f1=18000.0)),
                                     ("rx_real", dict(flags=2))])
def test_window_spectrum_matches_the_oracle_bin_by_bin(uchirp, name, kw):
    """a5 per bin: uc_window_spectrum == the float64 oracle's spectrum on every bin dsp() looks at, and the statistics
    of uc_process_batch are the maxima of exactly these values."""
    var = {"rx_real": uco.RX_REAL, "sync_cplx": uco.SYNC_CPLX, "dechirp_down": uco.DECHIRP_DOWN}[name]
    fk = {k: kw[k] for k in ("fs", "f0", "f1") if k in kw}
    for dtype in (np.float32, np.int32):
        frames, _ = synth.make_frames(257, seed=5, snr_db=-5.0, dtype=dtype, **fk)
        o = uco.Oracle(var, mag_mean=1000.0, **kw)
        e = uchirp.Engine(var, mag_mean=1000.0, **kw)
        bw2 = e.bandwidth2
        g = e.window_spectrum(frames)
        assert g.shape == (257, e.spf, 2 * bw2 + 1)
        _, st = e.process(frames)
        for f in range(257):
            ref = o.spectrum(frames[f])                    # [spf, n] float64
            for h in range(e.spf):
                want = np.concatenate([ref[h][N - bw2:], ref[h][:bw2 + 1]])
                assert np.abs(g[f, h] - want).max() <= MAG_TOL * want.max(), (f, h)
                # the statistics are the maxima of exactly these values, bit for bit: both calls run the same build (round 4:
                # the default two-round build has its own instantiation with the bin stores; it used to borrow the
                # three-round build, whose twiddles for bins >= 128 come from the table instead of being derived)
                mr, ml = g[f, h, bw2:2 * bw2].max(), g[f, h, :bw2].max()
                assert st[f, h]["mag_max_right"] == mr and st[f, h]["mag_max_left"] == ml, (f, h)
    # device tensors, overlapping frames
    import torch
    x = torch.from_numpy(frames.reshape(-1)[: 2048 * 9]).to("cuda:0")
    gd = e.window_spectrum(x, stride=256)
    torch.cuda.synchronize()
    gh = e.window_spectrum(frames.reshape(-1)[: 2048 * 9], stride=256)
    assert gd.shape[0] == 65 and np.array_equal(gd.cpu().numpy().view(np.uint32), gh.view(np.uint32))


@pytest.mark.parametrize("name", ["rx_real", "sync_cplx", "dechirp_down"])
def test_set_table_custom_reference_matches_the_oracle(uchirp, name):
    """A host-supplied reference (the measured chirp of a real channel, say): swapped up / down references with a
    Hamming window instead of the Hann -- product and oracle given the same tables agree as they do on their own."""
    var = {"rx_real": uco.RX_REAL, "sync_cplx": uco.SYNC_CPLX, "dechirp_down": uco.DECHIRP_DOWN}[name]
    kw = dict(fs=100000.0, f0=17000.0, f1=18000.0) if name == "dechirp_down" else dict(time_frame=2048.0 / 78125.0)
    o = uco.Oracle(var, mag_mean=1000.0, **kw)
    e = uchirp.Engine(var, mag_mean=1000.0, **kw)
    up, down = o.table(uco.TABLE_UP), o.table(uco.TABLE_DOWN)
    ham = (0.54 - 0.46 * np.cos(2 * np.pi * np.arange(N) / N)).astype(np.float32)
    for eng in (o, e):
        eng.set_table(uco.TABLE_UP, down)
        eng.set_table(uco.TABLE_DOWN, up)
        eng.set_table(uco.TABLE_HANN, ham)
    for tid in (uco.TABLE_UP, uco.TABLE_DOWN, uco.TABLE_HANN):
        assert np.array_equal(e.table(tid).view(np.uint32), o.table(tid).view(np.uint32))
    fk = {k: kw[k] for k in ("fs", "f0", "f1") if k in kw}
    frames, bits = synth.make_frames(400, seed=8, snr_db=-8.0, **fk)
    rs, rst = o.process(frames)
    gs, gst = e.process(frames)
    if o.spf == 2:
        clear = clear_symbols(rst)
        assert clear.mean() >= 0.995 and np.array_equal(gs[clear], rs[clear])
        assert (gs == 1 - bits).mean() > 0.97              # the references were swapped: every bit reads inverted
    for h in range(o.spf):
        check_history(o, lambda f: frames[f], gst[:, h], rst[:, h], h, "%s custom hist%d" % (name, h),
                      raw_idx=name == "dechirp_down")
    # error paths
    with pytest.raises(uchirp.UchirpError):
        e.set_table(uchirp.TABLE_UP, np.ones(7, np.float32))
    with pytest.raises(uchirp.UchirpError):
        e.set_table(uchirp.TABLE_FIR, np.ones(27, np.float32))
    with pytest.raises(uchirp.UchirpError):
        uchirp.Engine(uchirp.COMPRESS).set_table(uchirp.TABLE_UP, np.ones(N, np.float32))
    with pytest.raises(uchirp.UchirpError):
        uchirp.Engine(uchirp.COMPRESS).window_spectrum(frames)


def test_default_build_statistics_against_the_device_spectrum(uchirp):
    """The THROUGHPUT instantiation itself (uc_process_batch, band_kernel<sync_cplx, int32> two-round build, no spectrum
    stores) against the device's `.fft`: the reference e^{-j 2 pi m t / n} puts bins m .. m + 189 of the capture into the
    right window and m - 190 .. m - 1 into the left one; for every shift whose windows lie wholly at or above 1 kHz the
    statistics' mag_max_right / mag_max_left and their peak bins must be the device's maximum and arg-max over those bins."""
    W = 190
    e = uchirp.Engine(uchirp.SYNC_CPLX, **GEOM_DEFAULT)
    assert e.bandwidth2 == W
    to_bin_r = {e.idx2freq(k): k for k in range(W)}                 # idx2freq is injective inside a window
    to_bin_l = {e.idx2freq(N - k): k for k in range(1, W + 1)}      # left window: index n - k, k = 1 .. W
    caps = list(_captures())
    frames = np.stack([c[1] for c in caps])
    checked = 0
    for sh in ((211, 401), (591, 781), (833, 833)):
        _slide(e, uchirp, sh)
        _, st = e.process(frames)
        for i, (name, raw, flt, freq, mag_dev) in enumerate(caps):
            if name == K6_MISMATCHED:
                continue
            scale = np.abs(np.fft.rfft(raw.astype(np.float64) * np.hanning(N + 1)[:N])).max()
            tol = MAG_TOL * scale + PRINT_STEP
            for h, m in enumerate(sh):
                assert m + W - 1 <= 1023
                if freq[m - W] < 1000.0:       # (the 41.7 kHz captures: the device forced these bins to 1.0)
                    continue
                right, left = mag_dev[m:m + W], mag_dev[m - W:m]
                assert abs(st[i, h]["mag_max_right"] - right.max()) <= tol, (name, m)
                assert abs(st[i, h]["mag_max_left"] - left.max()) <= tol, (name, m)
                # the peak bin is the device's arg-max, or a bin the device itself printed within tolerance of it
                kr = to_bin_r[int(st[i, h]["max_freq_right"])]
                kl = to_bin_l[int(st[i, h]["max_freq_left"])]
                assert right.max() - right[kr] <= 2 * tol, (name, m, kr, int(np.argmax(right)))
                assert left.max() - left[W - kl] <= 2 * tol, (name, m, kl)
                checked += 1
    assert checked >= 23 * 4, checked
