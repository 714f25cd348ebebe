#!/usr/bin/env python3
"""Extended random-configuration parity run: the body of tests/test_gpu_parity.py::test_randomised_configurations over
many more draws (not part of the suite: minutes of oracle time).  Usage: python tools/fuzz_parity.py [cases=300] [seed=1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
import numpy as np
import uchirp
from oracle import uco
import test_gpu_parity as T
from parity_util import check_history, MARGIN

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
# two thirds of the draws with windows of at most 191 bins (the default build), one third with 192 .. 319 (the WIDE build)
draws = T._random_configs(cases - cases // 3, seed=seed) + T._random_configs(cases // 3, seed=seed + 1, wide=True)
for case, (variant, cfg) in enumerate(draws):
    try:
        o = uco.Oracle(variant, **cfg)
        e = uchirp.Engine(variant, **cfg)
        for tid in (uco.TABLE_UP, uco.TABLE_DOWN, uco.TABLE_HANN):
            assert np.array_equal(e.table(tid).view(np.uint32), o.table(tid).view(np.uint32)), tid
        n_frames = int(np.random.default_rng(case).choice([1, 2, 7, 64, 65, 96]))
        frames, bits = T.synth.make_frames(n_frames, seed=1000 + case, snr_db=float(np.random.default_rng(case + 7).choice([-8.0, -3.0, 3.0])),
                                           fs=cfg["fs"], f0=cfg["f0"], f1=cfg["f1"], sweep_time=cfg["time_frame"] or None)
        rs, rst = o.process(frames)
        gs, gst = e.process(frames)
        for h in range(o.spf):
            check_history(o, lambda f: frames[f], gst[:, h], rst[:, h], h, "case %d hist%d" % (case, h), raw_idx=(variant == uco.DECHIRP_DOWN))
        if o.spf == 2:
            su, sd = rst["snr"][:, 0].astype(np.float64), rst["snr"][:, 1].astype(np.float64)
            thr = cfg["snr_threshold"]
            margin = np.abs(su - sd) / np.maximum(np.maximum(np.abs(su), np.abs(sd)), 1e-30)
            near = (np.abs(su - thr) < 1e-3 * abs(thr)) | (np.abs(sd - thr) < 1e-3 * abs(thr))
            clear = (margin >= MARGIN) & ~near
            assert np.array_equal(gs[clear], rs[clear])
    except AssertionError as ex:
        bad += 1
        print("FAIL case %d variant %d cfg %r: %s" % (case, variant, cfg, str(ex)[:300]), flush=True)
    if case % 50 == 49:
        print("%d cases, %d failures" % (case + 1, bad), flush=True)
print("done: %d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
