#!/usr/bin/env python3
"""UC_STREAM on random streams vs the oracle: decimation, template direction, length (ragged tails, fewer samples than a
block), dtype, noise, chunk sizes of the dynamic hand-out.  Usage: python tools/fuzz_stream.py [cases=120] [seed=1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
import numpy as np
import uchirp
from oracle import uco
from test_stream import _check_stream, make_stream

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
os.environ["UC_TUNING"] = "1"   # the library reads its experiment switches only under UC_TUNING=1
bad = 0
for case in range(cases):
    decim = int(rng.choice([4, 8, 16]))
    flags = int(rng.choice([0, uco.FLAG_STREAM_UP]))
    for k, v in (("UC_GRID", str(int(rng.choice([0, 1, 2, 5])))), ("UC_STREAM_CHUNK", str(int(rng.choice([1, 2, 4]))))):
        if v != "0":
            os.environ[k] = v
        else:
            os.environ.pop(k, None)
    o = uco.Oracle(uco.STREAM, decim=decim, flags=flags)
    e = uchirp.Engine(uchirp.STREAM, decim=decim, flags=flags)
    halo, _, _, hop = o.stream_geometry(0)
    n_sym = int(rng.integers(1, 60))
    dtype = np.int32 if rng.random() < 0.3 else np.float32
    x, _ = make_stream(n_sym, seed=int(rng.integers(1 << 30)), snr_db=float(rng.choice([-8.0, 0.0, 10.0])), dtype=dtype, lead=int(rng.integers(0, 4000)))
    cut = int(rng.integers(halo + decim, x.size + 1))
    x = x[:cut]
    try:
        cr, pr = o.process_stream(x)
        cg, pg = e.process_stream(x)
        assert cg.shape == cr.shape and pg.shape == pr.shape
        if cr.size:
            _check_stream(cg, pg, cr, pr, hop, "case %d" % case)
    except AssertionError as ex:
        bad += 1
        print("FAIL case %d decim %d flags %d samples %d dtype %s grid %s chunk %s: %s" % (case, decim, flags, x.size, np.dtype(dtype).name,
              os.environ.get("UC_GRID"), os.environ.get("UC_STREAM_CHUNK"), str(ex)[:300]), flush=True)
    if case % 20 == 19:
        print("%d cases, %d failures" % (case + 1, bad), flush=True)
print("done: %d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
