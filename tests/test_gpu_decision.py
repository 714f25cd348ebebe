"""The bit decision of main() (receiver/Src/main.c:521-531) on EVERY frame, near-ties included.

The oracle comparisons of the other files mask out frames whose float64 decision margin is below 1e-3: there the float32
kernel and the float64 oracle may legitimately land on different sides.  What must hold on ALL frames -- and what those
tests cannot see -- is that the kernel's symbol is the firmware's rule applied to the kernel's OWN float32 snrs:

    snr      = (mag_max - mag_mean) / mag_mean                       main.c:229
    valid    = (snr_up >= SNR_THRESHOLD) || (snr_down >= SNR_THRESHOLD)    main.c:521
    symbol   = valid ? ((snr_down > snr_up) ? 0 : 1) : 0xFF          main.c:523-531, 539   (tie -> 1 / up)

bit for bit, including snr_up == snr_down, an snr exactly AT the threshold, zeros and NaNs.  Checked here at the
bench's scale (1 Mi frames, noise floor placed so that the threshold splits the batch) for RX_REAL, SYNC_CPLX and
base-band I/Q, and on crafted per-frame noise floors that make the ties and the at-threshold cases exact."""
import numpy as np
import pytest

from uchirp import synth

pytestmark = pytest.mark.gpu

N = 2048
F32 = np.float32


@pytest.fixture(scope="module")
def uchirp():
    import uchirp as m
    m.lib()
    return m


def rule(su, sd, thr):
    """main.c:521-531 on float32 arrays (NaN compares false everywhere, as on the MCU)."""
    su, sd, thr = su.astype(F32), sd.astype(F32), F32(thr)
    with np.errstate(invalid="ignore"):
        valid = (su >= thr) | (sd >= thr)
        return np.where(valid, np.where(sd > su, 0, 1), 0xFF).astype(np.uint8)


def snr_of(mag_max, mag_mean):
    """main.c:229 in float32: one subtraction, one division, each rounded once."""
    with np.errstate(invalid="ignore", divide="ignore"):
        return ((mag_max.astype(F32) - mag_mean.astype(F32)).astype(F32) / mag_mean.astype(F32)).astype(F32)


def check_records(sym, st, thr, label):
    su, sd = st[:, 0]["snr"], st[:, 1]["snr"]
    for h in (0, 1):
        want = snr_of(st[:, h]["mag_max"], st[:, h]["mag_mean"])
        same = (want.view(np.uint32) == st[:, h]["snr"].view(np.uint32)) | (np.isnan(want) & np.isnan(st[:, h]["snr"]))
        assert same.all(), "%s: snr of history %d is not (mag_max - mag_mean) / mag_mean in float32 on %d frames" % (
            label, h, int((~same).sum()))
    want = rule(su, sd, thr)
    bad = np.nonzero(want != sym)[0]
    assert bad.size == 0, "%s: %d symbols are not the rule of the kernel's own snrs, first frame %d: snr %r / %r -> %d" % (
        label, bad.size, int(bad[0]), float(su[bad[0]]), float(sd[bad[0]]), int(sym[bad[0]]))
    return want


def _engine_and_frames(uchirp, which, n_frames, device, seed):
    """(engine kwargs, frames tensor) of one two-history pipeline on its own workload."""
    if which in ("rx_real", "sync_cplx"):
        frames, _ = synth.device_frames(n_frames, device, seed, snr_db=-10.0)
        return (uchirp.RX_REAL if which == "rx_real" else uchirp.SYNC_CPLX), {}, frames.reshape(-1)
    n = 1024 if which == "iq1024_bb" else 2048
    x, _ = synth.device_iq_stream(n_frames, n, device, seed, snr_db=-10.0)
    kw = dict(n=n, fs=100000.0, carrier=18000.0, f0=16500.0, f1=19500.0, time_frame=n / 100000.0,
              flags=uchirp.FLAG_IQ_BASEBAND)
    return uchirp.IQ, kw, x


@pytest.mark.parametrize("which,n_frames", [("rx_real", 1 << 20), ("sync_cplx", 1 << 20), ("iq1024_bb", 1 << 21),
                                            ("iq2048_bb", 1 << 19)])
def test_symbol_is_the_rule_of_the_kernels_own_snrs_at_scale(uchirp, which, n_frames):
    """Every frame of a bench-sized batch, no frame masked.  The noise floor is set to a third of the median peak, so the
    threshold (2.0) cuts through the middle of the batch: all three symbol values occur in bulk."""
    import torch
    dev = torch.device("cuda", 0)
    var, kw, x = _engine_and_frames(uchirp, which, n_frames, dev, seed=77)
    e = uchirp.Engine(var, mag_mean=1000.0, **kw)
    _, st = e.process(x, n_frames=n_frames, want_symbols=False)
    torch.cuda.synchronize()
    peak = st[:, :, 0].max(dim=1).values
    floor = float(peak.median().item()) / 3.0
    e.close()
    del st
    e = uchirp.Engine(var, mag_mean=floor, **kw)
    sym, st = e.process(x, n_frames=n_frames)
    torch.cuda.synchronize()
    sym, st = sym.cpu().numpy(), uchirp.stats_from_tensor(st)
    want = check_records(sym, st, 2.0, which)
    counts = {v: int((want == v).sum()) for v in (0, 1, 0xFF)}
    assert min(counts.values()) >= n_frames // 20, counts       # the threshold and both bits are exercised in bulk
    assert np.array_equal(st[:, 0]["mag_mean"], np.full(n_frames, floor, F32))
    print("%s: %d frames, symbols == rule(own snrs) on all; down/up/none = %d/%d/%d" % (which, n_frames, counts[0], counts[1],
                                                                                      counts[0xFF]))
    e.close()


def _floor_for_snr(m, target):
    """Per element a float32 noise floor a with fl(fl(m - a) / a) == target exactly, searched among the few floats
    around m / (target + 1); NaN where none exists."""
    base = (m.astype(np.float64) / (target + 1.0)).astype(F32)
    out = np.full(m.shape, np.nan, F32)
    for d in range(-6, 7):
        a = base.copy()
        for _ in range(abs(d)):
            a = np.nextafter(a, F32(np.inf) if d > 0 else F32(-np.inf))
        hit = (snr_of(m, a) == F32(target)) & np.isnan(out)
        out[hit] = a[hit]
    return out


@pytest.mark.parametrize("which", ["rx_real", "sync_cplx", "iq1024_bb", "iq2048_bb"])
def test_exact_ties_and_the_threshold_itself(uchirp, which):
    """Crafted noise floors (two floats per frame, uc_process_batch's mag_mean argument) that make the comparisons of
    main.c:521-531 exact: snr_up == snr_down (tie -> up), either snr exactly AT the threshold (>= passes), one ulp below
    it (fails), down ahead by one ulp; plus all-zero and NaN frames.  A finaliser that used > for >=, or >= for >, or
    resolved the tie to down, fails here and nowhere else."""
    import torch
    dev = torch.device("cuda", 0)
    nf = 4096
    var, kw, x = _engine_and_frames(uchirp, which, nf, dev, seed=91)
    n = kw.get("n", N)
    halo = 26 if var == uchirp.IQ else 0
    # frames 0 .. 63 all zero, frames 64 .. 126 NaN (a NaN sample poisons the whole transform), frame 127 zero again so that
    # no NaN reaches the 26-sample FIR history of frame 128 (I/Q); frame 127 itself is not looked at
    body = x[halo:].view(nf, n)
    body[:64] = 0.0
    body[64:127] = float("nan")
    body[127] = 0.0
    e0 = uchirp.Engine(var, mag_mean=1000.0, **kw)
    _, st0 = e0.process(x, n_frames=nf, want_symbols=False)
    torch.cuda.synchronize()
    st0 = uchirp.stats_from_tensor(st0)
    e0.close()
    mu, md = st0[:, 0]["mag_max"].copy(), st0[:, 1]["mag_max"].copy()
    live = np.arange(nf) >= 128
    assert np.isnan(mu[64:127]).all() and np.isnan(md[64:127]).all()
    assert (mu[:64] == 0).all() and (md[:64] == 0).all()

    recipes = {}
    # A: both snrs exactly 1.0 (m - m/2 == m/2 exactly): the tie, at a threshold of 1.0 also AT the threshold
    recipes["tie_at_1"] = (mu / F32(2), md / F32(2))
    # B: up exactly 1.0, down one ulp of its floor ahead / behind
    recipes["down_ahead"] = (mu / F32(2), np.nextafter(md / F32(2), F32(0)))
    recipes["down_behind"] = (mu / F32(2), np.nextafter(md / F32(2), F32(np.inf)))
    # C: snr exactly 2.0 on one side, far below on the other; and the floats either side of that floor
    a2u, a2d = _floor_for_snr(mu, 2.0), _floor_for_snr(md, 2.0)
    big = F32(1e30)
    recipes["up_at_2"] = (a2u, np.full(nf, big, F32))
    recipes["down_at_2"] = (np.full(nf, big, F32), a2d)
    recipes["up_just_below_2"] = (np.nextafter(a2u, F32(np.inf)), np.full(nf, big, F32))
    recipes["down_just_above_2"] = (np.full(nf, big, F32), np.nextafter(a2d, F32(0)))
    # D: the same floor on both sides (what the firmware does: one mag_mean)
    recipes["one_floor"] = (np.maximum(mu, md) / F32(3), np.maximum(mu, md) / F32(3))

    seen = {"tie": 0, "at_thr": 0, "below_thr_none": 0, "zero_tie_at_thr": 0}
    for thr in (2.0, 1.0, float(np.nextafter(F32(1.0), F32(2.0))), -1.0):
        e = uchirp.Engine(var, mag_mean=1.0, snr_threshold=thr, **kw)
        for name, (au, ad) in recipes.items():
            mm = np.stack([au, ad], axis=1).astype(F32)
            mm[:128] = 1.0                                 # zero / NaN frames: floor 1 -> snr -1 / NaN
            mm[~np.isfinite(mm)] = 1.0                      # (no exact floor found for this frame: any floor will do)
            sym, st = e.process(x, n_frames=nf, mag_mean=torch.from_numpy(mm.reshape(-1)).to(dev))
            torch.cuda.synchronize()
            sym, st = sym.cpu().numpy(), uchirp.stats_from_tensor(st)
            assert np.array_equal(st[:, 0]["mag_mean"].view(np.uint32), mm[:, 0].view(np.uint32))
            assert np.array_equal(st[:, 1]["mag_mean"].view(np.uint32), mm[:, 1].view(np.uint32))
            check_records(sym, st, thr, "%s %s thr %r" % (which, name, thr))
            su, sd = st[:, 0]["snr"], st[:, 1]["snr"]
            assert (sym[64:127] == 0xFF).all()                                      # NaN frames never pass a compare
            assert (su[:64] == -1).all() and (sd[:64] == -1).all()                  # zero frames: (0 - 1) / 1, a tie
            if thr == -1.0:
                assert (sym[:64] == 1).all()                                        # tie AT the threshold: valid, up
                seen["zero_tie_at_thr"] += 64
            else:
                assert (sym[:64] == 0xFF).all()
            if name == "tie_at_1":
                assert (su[live] == 1).all() and (sd[live] == 1).all()
                assert (sym[live] == (1 if thr <= 1.0 else 0xFF)).all()
                seen["tie"] += int(live.sum())
                seen["below_thr_none"] += int(live.sum()) if thr > 1.0 else 0
            if name == "down_ahead" and thr <= 1.0:
                assert (sd[live] > 1).all() and (sym[live] == 0).all()
            if name == "down_behind" and thr <= 1.0:
                assert (sd[live] < 1).all() and (sym[live] == 1).all()
            if name in ("up_at_2", "down_at_2") and thr == 2.0:
                s_hit = (su if name == "up_at_2" else sd)[live]
                hit = s_hit == 2
                assert hit.sum() >= live.sum() // 8, hit.sum()      # an exact floor exists for a good share of the frames
                assert (sym[live][hit] == (1 if name == "up_at_2" else 0)).all()
                seen["at_thr"] += int(hit.sum())
            if name == "up_just_below_2" and thr == 2.0:
                s_hit = su[live]
                below = s_hit < 2
                assert below.sum() >= live.sum() // 8
                assert (sym[live][below] == 0xFF).all()
                seen["below_thr_none"] += int(below.sum())
        e.close()
    assert all(v > 0 for v in seen.values()), seen
    print("%s: %s" % (which, seen))
