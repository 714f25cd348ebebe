"""UC_STREAM (BASELINE config 4): FIR-LPF decimate front-end + overlap-save chirp compression.

The pipeline is this build's composition of two reference stages (include/uchirp.h): the I/Q mixer +
27-tap low-pass of experiments/iq_modulation/Src/iq_modem.c:55-75 and the FFT x H x IFFT compression of
experiments/chirp_compression_time_domain/Src/chirp.c:78-83, run over a continuous stream.  The reference has
no vector for it, so the oracle (direct float64 time-domain sums) is pinned against numpy here (CPU tests),
and the HIP path against the oracle (GPU tests, through the C-ABI).
"""
import numpy as np
import pytest

from uchirp import synth
from oracle import uco

STREAM_TOL = 2e-5  # |GPU - oracle| relative to the largest compressed value of the stream


def make_stream(n_symbols, seed=7, snr_db=0.0, lead=3000, dtype=np.float32, amp=1000.0):
    """Back-to-back orthogonal chirp symbols (one per 2048 samples, as the transmitter sends them) after
    `lead` samples of noise; returns (samples, bits)."""
    frames, bits = synth.make_frames(n_symbols, seed=seed, snr_db=snr_db, amp=amp, dtype=np.float32)
    rng = np.random.default_rng(seed + 1)
    sigma = amp * 10.0 ** (-(snr_db if snr_db is not None else 60.0) / 20.0)
    head = (sigma * rng.standard_normal(lead)).astype(np.float32)
    x = np.concatenate([head, frames.reshape(-1)])
    if dtype == np.int32:
        return (np.round(x).astype(np.int64) * 256).astype(np.int32), bits
    return x, bits


def numpy_stream(o, x):
    """The definition of include/uchirp.h with numpy convolutions (float64)."""
    cfg = o.cfg
    D = cfg.decim
    fir = o.table(uco.TABLE_FIR).astype(np.float64)
    g = o.table(uco.TABLE_TEMPLATE).astype(np.float64)
    g = g[0::2] + 1j * g[1::2]
    L = g.size
    halo, n_out, _, _ = o.stream_geometry(x.size)
    r = np.arange(x.size, dtype=np.float64)
    mixed = x.astype(np.float64) * np.exp(-2j * np.pi * ((cfg.carrier / cfg.fs) * r % 1.0))
    zf = np.convolve(mixed, fir)[:x.size]
    z = zf[halo + np.arange(-(L - 1), n_out) * D]
    y = np.convolve(z, g)[L - 1:L - 1 + n_out]
    return np.abs(y)


# --------------------------------------------------------------------------- CPU: the oracle itself

@pytest.mark.parametrize("decim", [4, 8, 16])
def test_oracle_stream_equals_numpy_convolutions(decim):
    o = uco.Oracle(uco.STREAM, decim=decim)
    assert o.cfg.decim == decim
    x, _ = make_stream(12, seed=decim)
    halo, n_out, n_blocks, hop = o.stream_geometry(x.size)
    L = 2048 // decim
    assert (halo, hop) == ((L - 1) * decim + 26, 2048 - (L - 1))
    assert n_out == (x.size - halo) // decim and n_blocks == -(-n_out // hop)
    comp, peaks = o.process_stream(x)
    ref = numpy_stream(o, x)
    assert np.abs(comp - ref).max() <= 1e-6 * ref.max()
    for b in range(n_blocks):
        seg = comp[b * hop:(b + 1) * hop]
        assert peaks["offset"][b] == int(np.argmax(seg)) and peaks["value"][b] == seg.max()


def test_oracle_stream_compresses_symbols_to_one_peak_per_symbol():
    """Down template compresses UP chirps: one sharp peak per up symbol, L decimated samples apart."""
    o = uco.Oracle(uco.STREAM)
    D, L = 8, 256
    frames, _ = synth.make_frames(8, seed=1, snr_db=None)
    up, down = synth.chirp_pair()
    x = np.concatenate([np.zeros(o.stream_geometry(0)[0], np.float32)] + [up.astype(np.float32)] * 6)
    comp, _ = o.process_stream(x)
    pk = [int(np.argmax(comp[s * L:(s + 1) * L])) + s * L for s in range(1, 6)]
    assert all(b - a == L for a, b in zip(pk, pk[1:]))
    floor = np.median(comp[L:5 * L])
    assert comp[pk[0]] > 8 * floor
    # the same stream through the UP template stays flat (no compression of the wrong chirp)
    ou = uco.Oracle(uco.STREAM, flags=uco.FLAG_STREAM_UP)
    cu, _ = ou.process_stream(x)
    assert cu[L:5 * L].max() < 0.5 * comp[pk[0]]
    # ... and compresses the down chirp
    xd = np.concatenate([np.zeros(o.stream_geometry(0)[0], np.float32)] + [down.astype(np.float32)] * 6)
    cd, _ = ou.process_stream(xd)
    assert cd[L:5 * L].max() > 0.9 * comp[pk[0]]


def test_oracle_stream_chunks_continue_one_another():
    o = uco.Oracle(uco.STREAM)
    x, _ = make_stream(10, seed=3)
    halo, n_out, _, _ = o.stream_geometry(x.size)
    whole, _ = o.process_stream(x)
    cut = halo + 8 * 1000  # a multiple of D past the history
    a, _ = o.process_stream(x[:cut])
    b, _ = o.process_stream(x[cut - halo:])
    assert a.size == 1000
    joined = np.concatenate([a, b])
    assert joined.size == whole.size
    assert np.abs(joined - whole).max() <= 1e-6 * whole.max()


def test_oracle_stream_rejects_bad_decimation_and_other_entry_points():
    with pytest.raises(ValueError):
        uco.Oracle(uco.STREAM, decim=3)
    o = uco.Oracle(uco.STREAM)
    with pytest.raises(RuntimeError):
        o.process(np.zeros(4096, np.float32))
    c, p = o.process_stream(np.zeros(100, np.float32))
    assert c.size == 0 and p.size == 0


# --------------------------------------------------------------------------- GPU: HIP path vs oracle

@pytest.fixture(scope="module")
def uchirp():
    import uchirp as m
    m.lib()
    return m


def _check_stream(comp_g, peaks_g, comp_r, peaks_r, hop, label):
    scale = float(comp_r.max())
    err = np.abs(comp_g.astype(np.float64) - comp_r.astype(np.float64)).max()
    assert err <= STREAM_TOL * scale, "%s: |gpu - oracle| = %.3g of peak" % (label, err / scale)
    assert np.abs(peaks_g["value"].astype(np.float64) - peaks_r["value"]).max() <= STREAM_TOL * scale
    for b in np.nonzero(peaks_g["offset"] != peaks_r["offset"])[0]:
        # legal only as a near-tie of the oracle's own block maximum
        assert comp_r[b * hop + peaks_g["offset"][b]] >= peaks_r["value"][b] - STREAM_TOL * scale, \
            "%s: block %d peak offset %d is not a near-tie" % (label, b, peaks_g["offset"][b])


@pytest.mark.gpu
@pytest.mark.parametrize("decim", [4, 8, 16])
@pytest.mark.parametrize("dtype", [np.float32, np.int32])
def test_stream_matches_oracle(uchirp, decim, dtype):
    o = uco.Oracle(uco.STREAM, decim=decim)
    e = uchirp.Engine(uchirp.STREAM, decim=decim)
    assert e.stream_geometry(123456) == o.stream_geometry(123456)
    assert np.array_equal(e.table(uchirp.TABLE_TEMPLATE), o.table(uco.TABLE_TEMPLATE))
    assert np.array_equal(e.table(uchirp.TABLE_FIR), o.table(uco.TABLE_FIR))
    # ~40 overlap-save blocks plus a ragged tail, -5 dB
    _, _, _, hop = o.stream_geometry(0)
    x, _ = make_stream((40 * hop * decim) // 2048 + 3, seed=10 + decim, snr_db=-5.0, dtype=dtype, lead=1237)
    halo, n_out, n_blocks, hop = o.stream_geometry(x.size)
    assert n_blocks >= 40 and n_out % hop != 0
    cr, pr = o.process_stream(x)
    cg, pg = e.process_stream(x)
    assert cg.shape == cr.shape and pg.shape == pr.shape
    _check_stream(cg, pg, cr, pr, hop, "decim %d %s" % (decim, np.dtype(dtype).name))


@pytest.mark.gpu
def test_stream_edge_sizes(uchirp):
    o = uco.Oracle(uco.STREAM)
    e = uchirp.Engine(uchirp.STREAM)
    halo, _, _, hop = o.stream_geometry(0)
    rng = np.random.default_rng(5)
    for n_samples in (0, halo, halo + 7, halo + 8, halo + 8 * 5, halo + 8 * hop, halo + 8 * hop + 8, halo + 8 * (2 * hop - 1)):
        x = (1000.0 * rng.standard_normal(n_samples)).astype(np.float32)
        cr, pr = o.process_stream(x)
        cg, pg = e.process_stream(x)
        assert cg.shape == cr.shape and pg.shape == pr.shape, n_samples
        if cr.size:
            _check_stream(cg, pg, cr, pr, hop, "n_samples %d" % n_samples)
    # all-zero input: every output is exactly 0 and the first offset wins
    z = np.zeros(halo + 8 * (hop + 100), np.float32)
    cg, pg = e.process_stream(z)
    assert not cg.any() and not pg["offset"].any() and not pg["value"].any()
    # outputs are optional
    x = (1000.0 * rng.standard_normal(halo + 8 * 3000)).astype(np.float32)
    c0, p0 = e.process_stream(x)
    c1, none = e.process_stream(x, want_peaks=False)
    none2, p1 = e.process_stream(x, want_compressed=False)
    assert none is None and none2 is None and np.array_equal(c0, c1) and np.array_equal(p0, p1)


@pytest.mark.gpu
@pytest.mark.parametrize("decim", [4, 8, 16])
def test_stream_block_chunks_dynamic_hand_out(uchirp, decim, monkeypatch, uc_tuning):
    """The stream kernel deals overlap-save blocks in chunks of consecutive blocks; a workgroup's first chunk is fixed,
    every further one comes from an atomic counter asked one block ahead (csrc/uc_stream_kernel.hip).  Tiny grids
    (UC_GRID), chunk sizes 1 / 2 / 4 / 8 (UC_STREAM_CHUNK), the static partition (UC_STATIC_DEAL) and stream lengths
    around the chunk boundaries must give the same bytes as the default launch."""
    ref = uchirp.Engine(uchirp.STREAM, decim=decim)
    halo, _, _, hop = ref.stream_geometry(0)
    rng = np.random.default_rng(77)
    x = (1000.0 * rng.standard_normal(halo + decim * (41 * hop + 13))).astype(np.float32)
    for env in ({"UC_GRID": "1", "UC_STREAM_CHUNK": "1"}, {"UC_GRID": "2", "UC_STREAM_CHUNK": "2"},
                {"UC_GRID": "3", "UC_STREAM_CHUNK": "4"}, {"UC_GRID": "1", "UC_STREAM_CHUNK": "8"},
                {"UC_GRID": "2", "UC_STREAM_CHUNK": "1"}, {"UC_GRID": "3", "UC_STATIC_DEAL": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = uchirp.Engine(uchirp.STREAM, decim=decim)
        for k in env:
            monkeypatch.delenv(k)
        for blocks in (42, 41, 40, 33, 32, 31, 17, 16, 9, 8, 5, 4, 3, 2, 1):
            n = halo + decim * ((blocks - 1) * hop + 13)
            c0, p0 = ref.process_stream(x[:n])
            c1, p1 = e.process_stream(x[:n])
            assert len(p0) == blocks
            assert np.array_equal(c0.view(np.uint32), c1.view(np.uint32)), (env, blocks)
            assert np.array_equal(p0, p1), (env, blocks)


@pytest.mark.gpu
def test_stream_up_template_flag(uchirp):
    o = uco.Oracle(uco.STREAM, flags=uco.FLAG_STREAM_UP)
    e = uchirp.Engine(uchirp.STREAM, flags=uchirp.FLAG_STREAM_UP)
    x, _ = make_stream(20, seed=2, snr_db=0.0)
    cr, pr = o.process_stream(x)
    cg, pg = e.process_stream(x)
    _check_stream(cg, pg, cr, pr, o.stream_geometry(0)[3], "up template")


@pytest.mark.gpu
def test_stream_chunked_device_calls_and_graph_replay(uchirp):
    """Config 4's streaming loop: fixed-shape chunks with the history carried over, captured ONCE into a
    hipGraph and replayed per chunk; block-aligned chunks reproduce the one-shot result bit for bit."""
    import torch
    dev = torch.device("cuda:0")
    e = uchirp.Engine(uchirp.STREAM)
    halo, _, _, hop = e.stream_geometry(0)
    D = 8
    chunk_out = 24 * hop              # outputs per chunk: whole overlap-save blocks
    chunk_in = chunk_out * D          # new samples per chunk
    n_chunks = 5
    x, _ = make_stream((n_chunks * chunk_in) // 2048 + 2, seed=21, snr_db=-5.0, lead=halo)
    x = x[:halo + n_chunks * chunk_in]
    xd = torch.from_numpy(x).to(dev)
    whole, whole_pk = e.process_stream(xd)
    torch.cuda.synchronize()
    whole = whole.cpu().numpy()
    assert whole.size == n_chunks * chunk_out
    o = uco.Oracle(uco.STREAM)
    cr, pr = o.process_stream(x[:halo + chunk_in])
    _check_stream(whole[:chunk_out], uchirp.peaks_from_tensor(whole_pk)[:24], cr, pr, hop, "first chunk")

    # eager chunked calls
    buf = torch.zeros(halo + chunk_in, dtype=torch.float32, device=dev)
    out = torch.empty(chunk_out, dtype=torch.float32, device=dev)
    pk = torch.empty((24, 2), dtype=torch.int32, device=dev)
    got = []
    for c in range(n_chunks):
        buf.copy_(xd[c * chunk_in:c * chunk_in + halo + chunk_in])
        e.process_stream(buf, compressed_out=out, peaks_out=pk)
        got.append(out.cpu().numpy().copy())
    assert np.array_equal(np.concatenate(got), whole)

    # the same loop as ONE captured graph: copy-in, kernel, history carry
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    src = torch.zeros(chunk_in, dtype=torch.float32, device=dev)
    buf.zero_()
    buf[:halo].copy_(xd[:halo])
    torch.cuda.synchronize()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            buf[halo:].copy_(src)
            e.process_stream(buf, compressed_out=out, peaks_out=pk, stream=s.cuda_stream)
            buf[:halo].copy_(buf[chunk_in:chunk_in + halo].clone())
    buf.zero_()
    buf[:halo].copy_(xd[:halo])
    got = []
    for c in range(n_chunks):
        src.copy_(xd[halo + c * chunk_in:halo + (c + 1) * chunk_in])
        g.replay()
        torch.cuda.synchronize()
        got.append(out.cpu().numpy().copy())
    assert np.array_equal(np.concatenate(got), whole)

    # unaligned chunks (block boundaries move): same values within the float32 tolerance
    cut = halo + D * 10007
    a, _ = e.process_stream(xd[:cut].contiguous())
    b, _ = e.process_stream(xd[cut - halo:].contiguous())
    joined = torch.cat([a, b]).cpu().numpy()
    assert joined.size == whole.size
    assert np.abs(joined - whole).max() <= STREAM_TOL * whole.max()


@pytest.mark.gpu
def test_stream_properties_at_scale(uchirp):
    """2^26 samples (256 MiB) on the device: x2 scaling is exact, every up symbol compresses to a peak one
    symbol (n/D outputs) after the previous one, and the head agrees with the oracle."""
    import torch
    dev = torch.device("cuda:0")
    e = uchirp.Engine(uchirp.STREAM)
    halo, _, _, hop = e.stream_geometry(0)
    n_sym = (1 << 26) // 2048
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)
    up, down = synth.chirp_pair()
    sym = torch.from_numpy(np.stack([down, up]).astype(np.float32)).to(dev)
    bits = torch.randint(0, 2, (n_sym,), generator=gen, device=dev)
    x = sym[bits] + 1000.0 * torch.randn((n_sym, 2048), generator=gen, device=dev)
    x = torch.cat([torch.zeros(halo, device=dev), x.reshape(-1)])[:1 << 26].contiguous()
    c1, p1 = e.process_stream(x)
    c2, p2 = e.process_stream((2.0 * x).contiguous())
    torch.cuda.synchronize()
    assert torch.equal(c2, 2.0 * c1)
    assert torch.equal(p2[:, 1], p1[:, 1])
    n_out = c1.numel()
    assert n_out == ((1 << 26) - halo) // 8
    # symbol s ends at output 256 (s+1) - 1 (+- the filter's group delay): the per-symbol maximum of an UP symbol
    # sits at a fixed offset inside the symbol
    L = 256
    n_full = n_out // L - 1
    per = c1[L // 2:L // 2 + n_full * L].view(n_full, L)   # window k is centred on the end of symbol k
    arg = per.argmax(dim=1)
    upsym = bits[:n_full].bool()
    mode = int(torch.mode(arg[upsym]).values)
    frac = float(((arg[upsym] - mode).abs() <= 1).float().mean())
    assert frac > 0.98, frac
    # head against the oracle
    o = uco.Oracle(uco.STREAM)
    head = x[:halo + 8 * 20 * hop].cpu().numpy()
    cr, pr = o.process_stream(head)
    _check_stream(c1[:cr.size].cpu().numpy(), uchirp.peaks_from_tensor(p1)[:20], cr, pr, hop, "head at scale")


@pytest.mark.gpu
def test_stream_error_paths(uchirp):
    import torch
    with pytest.raises(uchirp.UchirpError):
        uchirp.Engine(uchirp.STREAM, decim=3)
    e = uchirp.Engine(uchirp.STREAM)
    with pytest.raises(uchirp.UchirpError):
        e.process(np.zeros(4096, np.float32))          # no frames in this variant
    r = uchirp.Engine(uchirp.RX_REAL)
    with pytest.raises(uchirp.UchirpError):
        r.process_stream(np.zeros(40000, np.float32))  # and no stream in the others
    x = torch.zeros(40001, dtype=torch.float32, device="cuda:0")
    with pytest.raises(uchirp.UchirpError):
        e.process_stream(x[1:])                         # device pointer not 16-byte aligned
