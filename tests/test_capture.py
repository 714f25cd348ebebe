"""Capture / log formats: uchirp.capture against the on-device K6 fixtures
(agent/ captures copied by tests/golden/make_golden.py) and, on the GPU, a capture file fed to the C-ABI."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "ultrasonic-communication_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from uchirp import capture  # noqa: E402
from oracle import uco      # noqa: E402

K6_DIR = os.path.join(HERE, "golden", "k6")
K6 = ["chirp_16000_18000_1m_100kHz_M1", "paper_100kHz_M1", "vacuum_1526445492_41.7kHz_M1"]


@pytest.mark.parametrize("name", K6)
def test_readers_parse_the_device_captures(name):
    raw = capture.read_raw(os.path.join(K6_DIR, name + ".raw"))
    flt = capture.read_flt(os.path.join(K6_DIR, name + ".flt"))
    f, mag, db = capture.read_fft(os.path.join(K6_DIR, name + ".fft"))
    assert raw.dtype == np.int32 and raw.size == 2048 and flt.size == 2048 and f.size == mag.size == db.size == 1024
    assert not (raw & 0xFF).any()                      # 24-bit sample in bits 31:8 (dfsdm.c:78)
    fs = capture.fs_from_fft(f)
    assert abs(fs - (41700.0 if "41.7kHz" in name else 100000.0)) / fs < 2e-3
    # the windowed samples are the raw words x the LUT Hann (pinned to 1 ulp in test_oracle_golden)
    w = uco.hann_periodic(2048)
    assert np.abs(flt - raw.astype(np.float32) * w).max() <= 1e-6 * np.abs(flt).max() + 5e-7


def test_uart_dump_round_trip():
    name = K6[1]
    raw = capture.read_raw(os.path.join(K6_DIR, name + ".raw"))
    flt = capture.read_flt(os.path.join(K6_DIR, name + ".flt"))
    fft = capture.read_fft(os.path.join(K6_DIR, name + ".fft"))
    i = int(np.argmax(fft[1]))
    text = capture.format_uart_dump("M1", float(fft[0][i]), float(fft[1][i]), fft, raw, flt)
    assert text.count("Index,Amplitude") == 2 and "EORAW\n" in text and text.endswith("EOFLT\n")
    d = capture.parse_uart_dump(text)
    assert d["mic"] == "M1" and abs(d["freq_max"] - fft[0][i]) < 0.06 and abs(d["mag_max"] - fft[1][i]) < 1e-3 * fft[1][i]
    assert np.array_equal(d["raw"], raw)
    assert np.allclose(d["flt"], flt, atol=1e-6) and np.allclose(d["fft"][1], fft[1], rtol=1e-6, atol=1e-6)
    # the file-level readers accept the same text sections
    with open(os.path.join(K6_DIR, name + ".raw")) as fh:
        assert np.array_equal(capture.read_raw(fh.read()), raw)


def test_malformed_captures_are_rejected():
    with pytest.raises(ValueError):
        capture.read_raw("Index,Amplitude\n0,1\n2,3\n")           # index gap
    with pytest.raises(ValueError):
        capture.read_raw("Frequency(Hz),Magnitude,Magnitude(dB)\n0.0,1,0\n")
    with pytest.raises(ValueError):
        capture.read_fft("Index,Amplitude\n0,1\n")
    with pytest.raises(ValueError):
        capture.parse_uart_dump("Frequency(Hz),Magnitude,Magnitude(dB)\n0.0,1,0\n\nIndex,Amplitude\n0,1\n")  # no EORAW
    with pytest.raises(ValueError):
        capture.read_raw("Index,Amplitude\n0,99999999999\n")


@pytest.mark.gpu
@pytest.mark.parametrize("name", K6)
def test_capture_file_through_the_c_abi(name):
    """A *.raw capture is a UC_DTYPE_I32 frame as is: device capture -> uc_process_frame == oracle."""
    import uchirp
    raw = capture.read_raw(os.path.join(K6_DIR, name + ".raw"))
    fs = capture.fs_from_fft(capture.read_fft(os.path.join(K6_DIR, name + ".fft"))[0])
    fs = 100000.0 if abs(fs - 100000.0) < 500 else fs
    # (41.7 kHz: 16-19 kHz is 2 x 147 = 294 bins wide there -- the WIDE build of the band kernel, tests/test_gpu_wide.py)
    e = uchirp.Engine(uchirp.RX_REAL, fs=fs, mag_mean=1000.0)
    o = uco.Oracle(uco.RX_REAL, fs=fs, mag_mean=1000.0)
    sym, st = e.process_frame(raw, mag_mean=1000.0)
    rs, rst = o.process(raw.reshape(1, -1))
    assert sym == rs[0]
    scale = max(float(rst["mag_max"].max()), 1e-30)
    for fld in ("mag_max", "mag_max_left", "mag_max_right"):
        assert np.abs(st[fld].astype(np.float64) - rst[0][fld]).max() <= 2e-5 * scale
    assert np.array_equal(st["max_freq"], rst[0]["max_freq"])
