#!/usr/bin/env python3
"""uc_window_spectrum / uc_set_table on random configurations vs the oracle: random fs / band (windows of 2 .. 319 bins),
random reference tables (the generated chirps, or replaced by random tables through uc_set_table / uco_set_table), int32 /
float32, strides; every window bin within MAG_TOL x the window maximum of the float64 oracle spectrum, and the statistics
of uc_process_batch equal to the window maxima.  Usage: python tools/fuzz_spectrum.py [cases=200] [seed=1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "ultrasonic-communication_amd")]
import numpy as np
import uchirp
from oracle import uco
import test_gpu_parity as T
from parity_util import MAG_TOL

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
draws = T._random_configs(cases - cases // 2, seed=seed) + T._random_configs(cases // 2, seed=seed + 1, wide=True)
bad = 0
for case, (variant, cfg) in enumerate(draws):
    try:
        o = uco.Oracle(variant, **cfg)
        e = uchirp.Engine(variant, **cfg)
        n = 2048
        if rng.random() < 0.5:       # a host-supplied reference
            cplx = variant == uco.SYNC_CPLX
            for tid in (uco.TABLE_UP, uco.TABLE_DOWN):
                t = rng.standard_normal(n * (2 if cplx else 1)).astype(np.float32)
                o.set_table(tid, t)
                e.set_table(tid, t)
            if rng.random() < 0.5:
                w = rng.uniform(0.1, 1.0, n).astype(np.float32)
                o.set_table(uco.TABLE_HANN, w)
                e.set_table(uco.TABLE_HANN, w)
        nf = int(rng.choice([1, 2, 5, 33, 64]))
        stride = int(rng.choice([n, 256, 700, 2500]))
        dtype = np.int32 if rng.random() < 0.3 else np.float32
        x = rng.standard_normal((nf - 1) * stride + n) * 2000.0
        x = (np.round(x).astype(np.int64) * 256).astype(np.int32) if dtype == np.int32 else x.astype(np.float32)
        g = e.window_spectrum(x, n_frames=nf, stride=stride)
        _, st = e.process(x, n_frames=nf, stride=stride)
        bw2 = e.bandwidth2
        pair = variant == uco.DECHIRP_DOWN
        for f in range(nf):
            ref = o.spectrum(x[f * stride: f * stride + n])
            for h in range(e.spf):
                want = np.concatenate([ref[h][n - bw2:], ref[h][:bw2 + 1]])
                # (frame pairs: a frame's round-off scales with the larger frame of its pair)
                tol = MAG_TOL * want.max() * (4.0 if pair else 1.0)
                assert np.abs(g[f, h] - want).max() <= tol, ("bins", f, h, np.abs(g[f, h] - want).max(), tol)
                mr, ml = g[f, h, bw2:2 * bw2].max(), g[f, h, :bw2].max()
                assert abs(st[f, h]["mag_max_right"] - mr) <= 4e-6 * max(mr, ml) and abs(st[f, h]["mag_max_left"] - ml) <= 4e-6 * max(mr, ml), "stats"
    except Exception as ex:  # noqa: BLE001
        bad += 1
        print("case %d variant %d %r: %s" % (case, variant, cfg, str(ex)[:300]), flush=True)
    if (case + 1) % 50 == 0:
        print("%d cases, %d failures" % (case + 1, bad), flush=True)
print("done: %d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
