"""The last two records the reference holds about this path, through the HIP library (round 6):

  * K8 -- the on-device sync log, experiments/EXPERIMENT3.md:50-59: dsp() at the eight FIFO offsets of an acquisition sweep.
    uc_process_batch over frames 256 samples apart (the reference's own overlapping FIFO reads, synchronization/Src/main.c,
    receiver/Src/main.c:447-451) shows the law the log shows, and equals the oracle field for field.
  * K7 through a REAL rate conversion: the transmitter's 44.1 kHz int16 WAV samples, converted to the
    receiver's 78 125 samples/s (uchirp/resample.py), decode to "Hello World!" through uc_receive_stream and through live
    receivers fed one block per call; traces equal the oracle's literal loop.
(What the law is, and what is OUR choice in reproducing it, is argued in tests/test_oracle_golden.py.)"""
import numpy as np
import pytest

from oracle import uco
from test_oracle_golden import KNOWN, k7_resampled_stream, k8_law, k8_preamble_frames

pytestmark = pytest.mark.gpu
N = 2048


@pytest.fixture(scope="module")
def uchirp():
    import uchirp as m
    m.lib()
    return m


def test_k8_peak_walks_with_the_fifo_offset_as_in_the_device_log(uchirp):
    import torch
    rows = KNOWN["EXPERIMENT3_sync_log"]["rows"]
    log_step = np.diff([r["i"] for r in rows if r["max"] == r["max_r"]]).mean()          # 5.25 bins per n / 8 samples
    fs, tf = 100000.0, 0.03         # the experiments' rate (Divider 25); the sweep that makes a step 5.24 bins: OUR choice
    e = uchirp.Engine(uco.SYNC_CPLX, fs=fs, time_frame=tf, mag_mean=1.0)
    o = uco.Oracle(uco.SYNC_CPLX, fs=fs, time_frame=tf, mag_mean=1.0)
    for direction, hist, sign in (("down", 1, +1), ("up", 0, -1)):
        frames = k8_preamble_frames(fs, tf, 64, direction)
        # the eight frames as the firmware reads them: ONE buffer, frames 256 samples apart (stride < n: no copies)
        fifo = np.concatenate([frames[0], frames[7][-7 * 256:]]).astype(np.float32)
        assert all(np.array_equal(fifo[256 * q:256 * q + N], frames[q]) for q in range(8))
        sym, st = e.process(torch.from_numpy(fifo).to("cuda:0"), n_frames=8, stride=256)
        torch.cuda.synchronize()
        st = uchirp.stats_from_tensor(st)
        _, ost = o.process(frames)
        g, r = st[:, hist], ost[:, hist]
        # == the oracle: integer fields exactly, magnitudes to the library's bar
        for fld in ("max_freq", "max_freq_left", "max_freq_right"):
            assert np.array_equal(g[fld], r[fld]), (direction, fld)
        for fld in ("mag_max", "mag_max_left", "mag_max_right"):
            assert np.abs(g[fld].astype(np.float64) - r[fld]).max() <= 2e-5 * r["mag_max"].max(), (direction, fld)
        # the law of the log
        win, bins = k8_law(g, fs)
        first, second = ("R", "L") if sign > 0 else ("L", "R")
        assert "".join(win) == first * 4 + second * 4, (direction, win)
        for side in (first, second):
            d = np.diff([v for v, w in zip(bins, win) if w == side]) * sign
            assert (d >= 4).all() and (d <= 6).all() and abs(d.mean() - log_step) <= 1.0, (direction, side, bins)
    e.close()


def test_k7_wav_converted_to_the_receivers_rate_decodes_recorded_and_live(uchirp):
    import torch
    x = k7_resampled_stream()
    nb = x.size // N
    for variant in (uco.SYNC_CPLX, uco.RX_REAL):
        o = uco.Oracle(variant)
        e = uchirp.Engine(variant)
        text_o, tr_o = o.receive(x, precision=uco.F64)
        text_g, tr_g = e.receive(x)                                          # uc_receive_stream
        assert text_g == text_o and len(tr_g) == len(tr_o) == nb
        for fld in ("block", "state_before", "state_after", "bit", "sync_position"):
            assert np.array_equal(tr_g[fld], tr_o[fld]), (variant, fld)
        if variant == uco.SYNC_CPLX:
            assert text_g == "Hello World!\n"
        # live: one block per call (uc_receive_streams_next), the caller's two chunk buffers kept and not kept
        for keep in (False, True):
            live = e.live(1)
            live.keep_previous(keep)
            ring = [torch.zeros((1, N), dtype=torch.float32, device="cuda:0") for _ in range(2)]
            xd = torch.from_numpy(x[:nb * N].reshape(1, nb * N)).to("cuda:0")
            text, traces = "", []
            for b in range(nb):
                ring[b % 2].copy_(xd[:, b * N:(b + 1) * N])
                t, tr = live.next(ring[b % 2])
                text += t[0]
                traces.append(tr[0])
            tr_l = np.concatenate(traces)
            assert text == text_g, (variant, keep)
            assert np.array_equal(tr_l.view(np.uint8), tr_g.view(np.uint8)), (variant, keep)   # bit for bit, snrs included
            live.close()
        e.close()
