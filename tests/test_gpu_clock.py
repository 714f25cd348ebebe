"""uc_clock_probe / uc_clock_read (include/uchirp.h): every kernel ships twice in libuchirp.so, the second build with one
s_memtime / s_memrealtime stamp pair per wave around its loop.  The twin must produce the bytes of the throughput build,
and its stamps must read as a plausible shader clock -- this is what bench.py's `roofline.valu` divides by."""
import numpy as np
import pytest

from uchirp import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def uchirp():
    import uchirp as m
    m.lib()
    return m


def _check_clock(c, label):
    assert 0.5 < c["shader_ghz"] < 3.0, (label, c)
    assert c["waves"] > 0 and c["wave_cycles"] > 0 and c["span_us"] > 0, (label, c)


def test_stamped_twin_of_every_frame_kernel_gives_the_same_bytes_and_a_clock(uchirp):
    import torch
    dev = torch.device("cuda", 0)
    nf = 1 << 15
    frames, _ = synth.device_frames(nf, dev, seed=3, snr_db=-10.0)
    for var, kw in ((uchirp.RX_REAL, {}), (uchirp.SYNC_CPLX, {}), (uchirp.DECHIRP_DOWN, {}), (uchirp.COMPRESS, {}),
                    (uchirp.RX_REAL, dict(fs=125000.0 / 3.0))):
        e = uchirp.Engine(var, mag_mean=1000.0, **kw)
        with pytest.raises(uchirp.UchirpError):
            e.clock_read()                                    # probe off: nothing to read
        s0, t0 = e.process(frames)
        e.clock_probe(True)
        with pytest.raises(uchirp.UchirpError):
            e.clock_read()                                    # probe on, no launch yet
        s1, t1 = e.process(frames)
        c = e.clock_read()
        raw = e.clock_stamps()
        assert raw.shape[1] == 4 and int((raw[:, 1] > 0).sum()) == c["waves"]
        e.clock_probe(False)
        s2, t2 = e.process(frames)
        torch.cuda.synchronize()
        assert torch.equal(s0, s1) and torch.equal(s0, s2)
        assert torch.equal(t0.view(torch.int32), t1.view(torch.int32)) and torch.equal(t0.view(torch.int32), t2.view(torch.int32))
        _check_clock(c, "variant %d" % var)
        e.close()
    # I/Q at both frame lengths, both modes
    for n in (1024, 2048):
        x, _ = synth.device_iq_stream(nf, n, dev, seed=4, snr_db=-10.0)
        for kw in (dict(n=n), dict(n=n, fs=100000.0, carrier=18000.0, f0=16500.0, f1=19500.0, time_frame=n / 100000.0,
                                   flags=uchirp.FLAG_IQ_BASEBAND)):
            e = uchirp.Engine(uchirp.IQ, mag_mean=1000.0, **kw)
            s0, t0 = e.process(x, n_frames=nf)
            e.clock_probe(True)
            s1, t1 = e.process(x, n_frames=nf)
            c = e.clock_read()
            e.clock_probe(False)
            torch.cuda.synchronize()
            assert torch.equal(s0, s1) and torch.equal(t0.view(torch.int32), t1.view(torch.int32))
            _check_clock(c, "iq n=%d" % n)
            e.close()


def test_stamped_twin_of_the_stream_and_sinc5_kernels(uchirp):
    import torch
    dev = torch.device("cuda", 0)
    x = (torch.randn(1 << 23, device=dev) * 1000).contiguous()
    e = uchirp.Engine(uchirp.STREAM)
    c0, p0 = e.process_stream(x)
    e.clock_probe(True)
    c1, p1 = e.process_stream(x)
    clk = e.clock_read()
    e.clock_probe(False)
    torch.cuda.synchronize()
    assert torch.equal(c0.view(torch.int32), c1.view(torch.int32)) and torch.equal(p0, p1)
    _check_clock(clk, "stream")
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    pdm = torch.randint(-2 ** 31, 2 ** 31 - 1, ((1 << 22) + 4,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    o0 = e.dfsdm(pdm)
    e.clock_probe(True)
    o1 = e.dfsdm(pdm)
    clk = e.clock_read()
    assert e.clock_stamps().shape[0] == clk["waves"] or e.clock_stamps().shape[0] > clk["waves"]
    e.clock_probe(False)
    torch.cuda.synchronize()
    assert torch.equal(o0, o1)
    _check_clock(clk, "sinc5")
    assert clk["waves"] % 16 == 0 and clk["waves"] >= 16        # sixteen waves per workgroup stamp (1024 threads)
    e.close()
