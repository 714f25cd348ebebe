"""The self-resetting hand-out counters (csrc/uc_dev.hpp: handout_leave): every dynamically dealt launch must leave its
counter at zero -- every workgroup, on every path out of every kernel, passes the exit exactly once.  A path that does not
would leave stale tickets and the NEXT launch on that counter would silently skip frame groups.  uc_debug_busy_counters
reads the counters back; here it is asserted behind every kernel family, ragged batch sizes, grids larger than the number
of groups (UC_GRID, the early-return path) and captured graphs, and the results of a second launch on the same counters
are compared with a statically dealt one."""
import numpy as np
import pytest

from uchirp import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def uchirp():
    import uchirp as m
    m.lib()
    return m


BB = dict(fs=100000.0, carrier=18000.0, f0=16500.0, f1=19500.0)


def _families(uchirp):
    return [("rx_real", uchirp.RX_REAL, {}), ("sync_cplx", uchirp.SYNC_CPLX, {}), ("dechirp_down", uchirp.DECHIRP_DOWN, {}),
            ("compress", uchirp.COMPRESS, {}), ("rx_real_wide", uchirp.RX_REAL, dict(fs=125000.0 / 3.0)),
            ("iq2048", uchirp.IQ, {}), ("iq1024", uchirp.IQ, dict(n=1024)),
            ("iq2048_bb", uchirp.IQ, dict(n=2048, time_frame=2048 / 1e5, flags=uchirp.FLAG_IQ_BASEBAND, **BB)),
            ("iq1024_bb", uchirp.IQ, dict(n=1024, time_frame=1024 / 1e5, flags=uchirp.FLAG_IQ_BASEBAND, **BB))]


@pytest.mark.parametrize("grid", [None, "7", "50000"])
def test_every_frame_kernel_leaves_its_counter_at_zero(uchirp, monkeypatch, grid):
    """grid None: the library's own grid; 7: few workgroups, many groups each; 50000: far more workgroups than groups (the
    early-return path of every kernel).  Sizes are ragged on purpose."""
    import torch
    monkeypatch.setenv("UC_TUNING", "1")
    if grid:
        monkeypatch.setenv("UC_GRID", grid)
    dev = torch.device("cuda", 0)
    for name, var, kw in _families(uchirp):
        n = kw.get("n", 2048)
        for nf in (1, 63, 4099, 70001):
            if var == uchirp.IQ:
                x, _ = synth.device_iq_stream(nf, n, dev, seed=nf, snr_db=-5.0)
            else:
                x, _ = synth.device_frames(nf, dev, seed=nf, snr_db=-5.0)
                x = x.reshape(-1)
            e = uchirp.Engine(var, mag_mean=1000.0, **kw)
            s0, t0 = e.process(x, n_frames=nf)
            assert e.busy_counters() == 0, (name, nf, grid)
            s1, t1 = e.process(x, n_frames=nf)            # the same counters again: nothing stale
            assert e.busy_counters() == 0, (name, nf, grid)
            torch.cuda.synchronize()
            assert torch.equal(s0, s1) and torch.equal(t0.view(torch.int32), t1.view(torch.int32)), (name, nf, grid)
            monkeypatch.setenv("UC_STATIC_DEAL", "1")
            es = uchirp.Engine(var, mag_mean=1000.0, **kw)
            monkeypatch.delenv("UC_STATIC_DEAL")
            s2, t2 = es.process(x, n_frames=nf)
            torch.cuda.synchronize()
            assert torch.equal(s0, s2) and torch.equal(t0.view(torch.int32), t2.view(torch.int32)), (name, nf, grid)
            es.close()
            e.close()


def test_stream_kernel_and_graph_replays_leave_their_counters_at_zero(uchirp, monkeypatch):
    import torch
    monkeypatch.setenv("UC_TUNING", "1")
    dev = torch.device("cuda", 0)
    for chunk, grid in (("2", None), ("1", "5"), ("8", "40000")):
        monkeypatch.setenv("UC_STREAM_CHUNK", chunk)
        if grid:
            monkeypatch.setenv("UC_GRID", grid)
        e = uchirp.Engine(uchirp.STREAM)
        for ns in (300000, (1 << 24) + 12345):
            x = (torch.randn(ns, device=dev) * 1000).contiguous()
            c0, p0 = e.process_stream(x)
            assert e.busy_counters() == 0, (chunk, grid, ns)
            c1, p1 = e.process_stream(x)
            assert e.busy_counters() == 0
            torch.cuda.synchronize()
            assert torch.equal(c0.view(torch.int32), c1.view(torch.int32)) and torch.equal(p0, p1)
        e.close()
        monkeypatch.delenv("UC_GRID", raising=False)
    # a captured launch owns a counter for the life of the context: zero after every replay
    e = uchirp.Engine(uchirp.RX_REAL, mag_mean=1000.0)
    nf = 50000
    frames, _ = synth.device_frames(nf, dev, seed=2, snr_db=-5.0)
    sym = torch.empty(nf, dtype=torch.uint8, device=dev)
    want, _ = e.process(frames, want_stats=False)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            e.process(frames, want_stats=False, symbols_out=sym, stream=s.cuda_stream)
        for _ in range(5):
            sym.zero_()
            g.replay()
            s.synchronize()
            assert torch.equal(sym, want)
            assert e.busy_counters() == 0
    del g
    e.close()


def test_eager_launch_while_another_stream_is_being_captured(uchirp):
    """an eager launch of a context that has already used two streams records and queries events; under a
    global-mode capture on another stream of the thread those calls are refused -- the launch must still succeed (dealt
    statically) and the capture must stay valid."""
    import torch
    dev = torch.device("cuda", 0)
    nf = 40000
    frames, _ = synth.device_frames(nf, dev, seed=4, snr_db=-5.0)
    e = uchirp.Engine(uchirp.RX_REAL, mag_mean=1000.0)
    want, _ = e.process(frames, want_stats=False)
    s1, s2, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    a, _ = e.process(frames, want_stats=False, stream=s1.cuda_stream)        # second stream: the context goes multi-stream
    b, _ = e.process(frames, want_stats=False, stream=s2.cuda_stream)
    torch.cuda.synchronize()
    other = uchirp.Engine(uchirp.RX_REAL, mag_mean=1000.0)
    out_g = torch.empty(nf, dtype=torch.uint8, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(sc):
        g.capture_begin()                                                     # global capture mode (torch's default)
        other.process(frames, want_stats=False, symbols_out=out_g, stream=sc.cuda_stream)
        # while `sc` captures: eager launches of `e` on its two streams
        for k in range(70):                                                   # (more than the 64 ring slots)
            st = s1 if k % 2 else s2
            c, _ = e.process(frames, want_stats=False, symbols_out=a if k % 2 else b, stream=st.cuda_stream)
        g.capture_end()
    torch.cuda.synchronize()
    assert torch.equal(a, want) and torch.equal(b, want)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_g, want)
    assert e.busy_counters() == 0 and other.busy_counters() == 0
    del g
    e.close()
    other.close()
