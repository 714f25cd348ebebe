"""Shared bars of the GPU parity tests (BASELINE.md section 4 / SURVEY.md section 8d).

  * magnitudes: |gpu - oracle64| <= MAG_TOL x the frame's largest WINDOW magnitude (float64 oracle);
  * integer outputs bit-exact, except where the ORACLE's own float64 spectrum proves a near-tie: an index
    mismatch is legal only if the GPU's bin is within MAG_TOL x (that window's maximum) of the maximum
    (`prove_ties`); every mismatch goes through that proof, there is no unproved allowance;
  * symbols bit-exact on every frame whose oracle decision margin is >= MARGIN and whose snrs are not within
    1e-3 of the threshold (`clear_symbols`).
"""
import numpy as np

from oracle import uco

MAG_TOL = 2e-5
MARGIN = 1e-3


def clear_symbols(rst, thr=2.0):
    su, sd = rst["snr"][:, 0].astype(np.float64), rst["snr"][:, 1].astype(np.float64)
    margin = np.abs(su - sd) / np.maximum(np.maximum(np.abs(su), np.abs(sd)), 1e-30)
    near_thr = (np.abs(su - thr) < 1e-3 * max(abs(thr), 1e-30)) | (np.abs(sd - thr) < 1e-3 * max(abs(thr), 1e-30))
    return (margin >= MARGIN) & ~near_thr


def window_scale(r):
    return np.maximum(np.maximum(r["mag_max_left"], r["mag_max_right"]).astype(np.float64), 1e-30)


def check_magnitudes(g, r, label, scale=None):
    scale = window_scale(r) if scale is None else scale
    for fld in ("mag_max", "mag_max_left", "mag_max_right"):
        err = np.abs(g[fld].astype(np.float64) - r[fld].astype(np.float64)) / scale
        assert np.nanmax(err) <= MAG_TOL, "%s %s: rel err %.3g" % (label, fld, np.nanmax(err))


def index_mismatches(g, r):
    return np.nonzero((g["max_freq"] != r["max_freq"]) | (g["max_freq_left"] != r["max_freq_left"])
                      | (g["max_freq_right"] != r["max_freq_right"]))[0]


def _bins_of(o, raw_idx):
    """(field -> function frequency-or-index value -> FFT bin) and the two windows [(lo, hi)] of this oracle."""
    n = o.n
    right = (0, o.bandwidth2)
    left = (o.idx_left_zero, n)
    if raw_idx:  # DECHIRP_DOWN reports raw indices, the left one as bandwidth2 - local index
        return {"max_freq_right": lambda v: int(v), "max_freq_left": lambda v: n - int(v)}, right, left
    cand = list(range(right[0], right[1])) + list(range(left[0], left[1]))
    lut = {}
    for i in cand:
        lut.setdefault(o.idx2freq(i), []).append(i)
    # idx2freq is injective inside each window; bin 0 and a left-window bin never share a frequency
    return ({"max_freq_right": lambda v: [i for i in lut[int(v)] if right[0] <= i < right[1]][0],
             "max_freq_left": lambda v: [i for i in lut[int(v)] if left[0] <= i < left[1]][0]}, right, left)


def prove_ties(o, get_frame, bad, g, r, row, label, raw_idx=False, spectrum_kw=None):
    """Every frame in `bad` (index mismatch between GPU record g and oracle record r, one history each):
    the GPU's bin must be a near-tie of the maximum in the ORACLE's float64 spectrum of that frame."""
    to_bin, right, left = _bins_of(o, raw_idx)
    for f in bad:
        spec = o.spectrum(get_frame(f), **(spectrum_kw or {}))[row]
        wr, wl = spec[right[0]:right[1]], spec[left[0]:left[1]]
        for fld, win in (("max_freq_right", wr), ("max_freq_left", wl)):
            if g[fld][f] == r[fld][f]:
                continue
            gi = to_bin[fld](g[fld][f])
            assert win.max() - spec[gi] <= MAG_TOL * win.max(), \
                "%s frame %d %s: GPU bin %d is not a near-tie (%.6g vs max %.6g)" % (label, f, fld, gi, spec[gi], win.max())
        if g["max_freq"][f] != r["max_freq"][f]:
            # the overall winner is one of the GPU's own side winners; if it sits on the other side than the
            # oracle's, the two side maxima must themselves be a near-tie
            assert g["max_freq"][f] in (g["max_freq_left"][f], g["max_freq_right"][f]), "%s frame %d" % (label, f)
            side_same = (g["max_freq"][f] == g["max_freq_left"][f]) == (r["max_freq"][f] == r["max_freq_left"][f])
            if not side_same or (g["max_freq_left"][f] == g["max_freq_right"][f]):
                assert abs(wl.max() - wr.max()) <= MAG_TOL * max(wl.max(), wr.max()), \
                    "%s frame %d: left/right winner differs without a near-tie" % (label, f)
    return len(bad)


def check_history(o, get_frame, g, r, row, label, raw_idx=False, spectrum_kw=None):
    """Magnitudes within MAG_TOL and every index mismatch a proven near-tie; returns the number of near-ties."""
    check_magnitudes(g, r, label)
    bad = index_mismatches(g, r)
    return prove_ties(o, get_frame, bad, g, r, row, label, raw_idx=raw_idx, spectrum_kw=spectrum_kw)
