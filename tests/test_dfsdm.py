"""DFSDM front end: sinc^5 / 32 of the 1-bit PDM stream, integer and bit-exact.

The peripheral is silicon (receiver/Src/dfsdm.c:59-61,69,78 only configure it), so the reference holds no
vector for it: the oracle's Hogenauer restatement is pinned against the direct 156-tap convolution in numpy
(CPU tests) and the HIP kernel against the oracle, bit for bit (GPU tests, through the C-ABI)."""
import numpy as np
import pytest

from uchirp import synth
from oracle import uco


def numpy_sinc5(words):
    w = np.asarray(words, np.uint32)
    bits = ((w[:, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(np.int64).reshape(-1) * 2 - 1
    h = np.ones(1, np.int64)
    for _ in range(5):
        h = np.convolve(h, np.ones(32, np.int64))
    assert h.size == 156 and h.sum() == 1 << 25
    y = np.convolve(bits, h)[np.arange(4, w.size) * 32 + 31]
    return (np.clip(y >> 2, -(1 << 23), (1 << 23) - 1) * 256).astype(np.int32)


def patterns(n, seed=0):
    rng = np.random.default_rng(seed)
    yield "random", rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    yield "ones", np.full(n, 0xFFFFFFFF, np.uint32)
    yield "zeros", np.zeros(n, np.uint32)
    yield "alternating", np.full(n, 0xAAAAAAAA, np.uint32)
    step = np.zeros(n, np.uint32)
    step[n // 2:] = 0xFFFFFFFF
    yield "step", step
    sparse = np.zeros(n, np.uint32)
    sparse[::7] = 1 << (np.arange(sparse[::7].size) % 32).astype(np.uint32)
    yield "single bits", sparse


def pdm_of_symbols(bits, amp=0.6, snr_db=None, seed=3):
    """Chirp symbols rendered at the PDM bit rate (2.5 MHz = 32 x 78125) and delta-sigma modulated."""
    n = 2048 * 32
    up, down = synth.chirp_pair(n=n, fs=78125.0 * 32, amp=1.0)   # same sweep, 32x oversampled
    x = np.concatenate([np.zeros(4 * 32)] + [(up if b else down) for b in bits]) * (amp / np.sqrt(2.0))
    if snr_db is not None:
        x = x + np.random.default_rng(seed).standard_normal(x.size) * amp * 10.0 ** (-snr_db / 20.0)
    return uco.pdm_modulate(np.clip(x, -1.0, 1.0).astype(np.float32))


# --------------------------------------------------------------------------- CPU

def test_oracle_sinc5_equals_the_direct_convolution():
    for name, w in patterns(300):
        assert np.array_equal(uco.dfsdm_sinc5(w), numpy_sinc5(w)), name
    assert (uco.dfsdm_sinc5(np.full(9, 0xFFFFFFFF, np.uint32)) == (2 ** 23 - 1) * 256).all()   # the one clipped value
    assert (uco.dfsdm_sinc5(np.zeros(9, np.uint32)) == -(2 ** 23) * 256).all()
    assert uco.dfsdm_sinc5(np.zeros(4, np.uint32)).size == 0
    out = uco.dfsdm_sinc5(next(patterns(64))[1])
    assert not (out & 0xFF).any()                      # 24-bit result in bits 31:8, as the agent/ *.raw captures


def test_oracle_sinc5_is_word_shift_invariant_and_chunks_continue():
    _, w = next(patterns(500, seed=5))
    full = uco.dfsdm_sinc5(w)
    assert np.array_equal(uco.dfsdm_sinc5(w[37:]), full[37:])
    a, b = uco.dfsdm_sinc5(w[:200]), uco.dfsdm_sinc5(w[196:])
    assert np.array_equal(np.concatenate([a, b]), full)


def test_pdm_front_end_feeds_the_receiver_oracle():
    """microphone bit stream -> DFSDM words -> dsp(): the decoded symbols are the transmitted ones."""
    bits = np.random.default_rng(1).integers(0, 2, size=12)
    words = uco.dfsdm_sinc5(pdm_of_symbols(bits))
    assert words.size == 12 * 2048
    o = uco.Oracle(uco.RX_REAL, mag_mean=float(np.abs(words).mean()) * 4)
    sym, st = o.process(words.reshape(12, 2048))
    assert np.array_equal(sym, bits)


# --------------------------------------------------------------------------- GPU

@pytest.fixture(scope="module")
def uchirp():
    import uchirp as m
    m.lib()
    return m


@pytest.mark.gpu
def test_sinc5_kernel_is_bit_exact(uchirp):
    e = uchirp.Engine(uchirp.RX_REAL)
    for n in (0, 4, 5, 6, 8, 255, 256, 257, 260, 4 + 252, 4 + 253, 4 + 4 * 252, 4 + 4 * 252 + 1, 5000, 70001):
        for name, w in patterns(n, seed=n):
            got = e.dfsdm(w)
            ref = uco.dfsdm_sinc5(w)
            assert got.shape == ref.shape and np.array_equal(got, ref), "%s n=%d" % (name, n)


@pytest.mark.gpu
def test_sinc5_ragged_tails_on_the_device_never_write_past_the_end(uchirp):
    """The kernel's stores are 16-byte buffer stores through a resource that covers exactly the tile's outputs: a lane
    whose four words straddle the end of the stream is clipped PER DWORD by the resource, lane 0 (whose outputs belong
    to the previous tile) sits at offset -16 = 0xFFFFFFF0 and is dropped whole.  Output counts with n_out % 4 in
    {1, 2, 3}, at and around wave-tile (252 outputs) and workgroup-tile boundaries, into a device buffer with guard
    words behind it: every output equals the oracle's, no guard word is touched."""
    import torch
    dev = torch.device("cuda:0")
    e = uchirp.Engine(uchirp.RX_REAL)
    tile = 252
    wg = tile * 16                                     # outputs per workgroup tile (16 waves)
    rng = np.random.default_rng(9)
    GUARD = -1234567
    for n_out in (1, 2, 3, 5, 6, 7, tile - 3, tile - 2, tile - 1, tile + 1, tile + 2, tile + 3, 2 * tile + 1, 3 * tile - 1,
                  wg - 1, wg + 1, wg + 2, wg + 3, 5 * wg + tile + 2, 70001, 70002, 70003):
        w = rng.integers(0, 1 << 32, size=n_out + 4, dtype=np.uint64).astype(np.uint32)
        ref = uco.dfsdm_sinc5(w)
        wd = torch.from_numpy(w.view(np.int32)).to(dev)
        buf = torch.full((n_out + 64,), GUARD, dtype=torch.int32, device=dev)
        e.dfsdm(wd, out=buf[:n_out])
        torch.cuda.synchronize()
        got = buf.cpu().numpy()
        assert np.array_equal(got[:n_out], ref), n_out
        assert (got[n_out:] == GUARD).all(), "n_out=%d: stored past the end" % n_out


@pytest.mark.gpu
def test_sinc5_device_path_at_scale_and_chunked(uchirp):
    import torch
    dev = torch.device("cuda:0")
    e = uchirp.Engine(uchirp.STREAM)          # any context will do
    n = (1 << 26) + 4                          # 256 MiB of PDM bits
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    w = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), generator=gen, device=dev, dtype=torch.int64).to(torch.int32)
    full = e.dfsdm(w)
    torch.cuda.synchronize()
    assert full.numel() == n - 4
    # head and tail against the oracle, bit for bit
    host = w.cpu().numpy().view(np.uint32)
    assert np.array_equal(full[:100000].cpu().numpy(), uco.dfsdm_sinc5(host[:100004]))
    assert np.array_equal(full[-100000:].cpu().numpy(), uco.dfsdm_sinc5(host[-100004:]))
    # word-shift invariance and chunk continuation, exactly, on the whole stream
    k = 4 * 1001
    assert torch.equal(e.dfsdm(w[k:].contiguous()), full[k:])
    cut = 4 * 3000001
    a, b = e.dfsdm(w[:cut + 4].contiguous()), e.dfsdm(w[cut:].contiguous())
    assert torch.equal(torch.cat([a, b]), full)
    assert int((full & 0xFF).abs().max()) == 0
    with pytest.raises(uchirp.UchirpError):
        e.dfsdm(w[1:100])                      # device pointer not 16-byte aligned


@pytest.mark.gpu
def test_pdm_to_symbols_on_the_gpu(uchirp):
    """The whole front of the receiver on the device: PDM bits -> uc_dfsdm_sinc5 -> uc_process_batch (int32
    DFSDM words) -> symbols, against the oracle chain and against the transmitted bits."""
    bits = np.random.default_rng(2).integers(0, 2, size=40)
    pdm = pdm_of_symbols(bits, snr_db=6.0)
    e = uchirp.Engine(uchirp.RX_REAL)
    words = e.dfsdm(pdm)
    ref_words = uco.dfsdm_sinc5(pdm)
    assert np.array_equal(words, ref_words)
    mm = float(np.abs(words).mean()) * 4
    e = uchirp.Engine(uchirp.RX_REAL, mag_mean=mm)
    o = uco.Oracle(uco.RX_REAL, mag_mean=mm)
    gs, _ = e.process(words.reshape(-1, 2048))
    rs, _ = o.process(ref_words.reshape(-1, 2048))
    assert np.array_equal(gs, rs) and np.array_equal(gs, bits)


def _sinc5_with_history(words, hist0=None):
    """The oracle's DFSDM words of a stream behind its first filter history (four silent words unless given)."""
    h = np.full(4, 0xAAAAAAAA, np.uint32) if hist0 is None else np.asarray(hist0, np.uint32)
    return uco.dfsdm_sinc5(np.concatenate([h, np.asarray(words, np.uint32)]))


def test_oracle_silent_history_is_silence():
    """UC_PDM_SILENCE: alternating bits are the PDM stream of a silent microphone -- the filter output is 0 within the last
    bit of the 24-bit result -- so a stream that starts behind four such words starts from rest, not from a full-scale step."""
    y = uco.dfsdm_sinc5(np.full(64, 0xAAAAAAAA, np.uint32))
    assert np.abs(y.astype(np.int64)).max() <= 256


@pytest.mark.gpu
@pytest.mark.parametrize("n_streams", [1, 64, 4096])
def test_sinc5_streams_with_carried_history_bit_exact(uchirp, n_streams):
    """uc_dfsdm_sinc5_streams: many microphones, chunk after chunk, the 4-word filter history of every stream carried in an
    array the call reads and brings up to date.  Chunks of ANY sizes -- one word, sizes around the tile (256 words; 252 outputs in rounds 1-4), one
    2048-word block, sizes that are not multiples of 4 -- give, stream by stream, exactly the oracle's DFSDM words of the
    whole stream behind its first history; host arrays and device tensors; no stream's output touches its neighbour's."""
    import torch
    dev = torch.device("cuda:0")
    e = uchirp.Engine(uchirp.RX_REAL)
    rng = np.random.default_rng(100 + n_streams)
    sizes = [1, 2, 3, 5, 4, 251, 252, 253, 256, 2048, 7, 504, 505, 2048, 1000]
    total = sum(sizes)
    w = rng.integers(0, 1 << 32, size=(n_streams, total), dtype=np.uint64).astype(np.uint32)
    if n_streams >= 64:                                                       # the patterns of the single-stream test too
        for k, (_, pat) in enumerate(patterns(total, seed=n_streams)):
            w[k] = pat
    hist0 = rng.integers(0, 1 << 32, size=(n_streams, 4), dtype=np.uint64).astype(np.uint32)
    hist0[::2] = 0xAAAAAAAA
    check = range(n_streams) if n_streams <= 64 else list(range(0, n_streams, 97)) + [n_streams - 1]
    ref = {s: _sinc5_with_history(w[s], hist0[s]) for s in check}
    # host arrays (any chunk sizes: the library pads its staging rows)
    hist = hist0.copy()
    got, b0 = [], 0
    for nw in sizes:
        got.append(e.dfsdm_streams(np.ascontiguousarray(w[:, b0:b0 + nw]), hist))
        b0 += nw
        assert np.array_equal(hist[:, -min(nw, 4):], w[:, b0 - min(nw, 4):b0])   # the history is the stream's last words
    got = np.concatenate(got, axis=1)
    for s in check:
        assert np.array_equal(got[s], ref[s]), s
    # device tensors (rows must stay 16-byte aligned: chunk sizes that are multiples of 4), guard words behind every row
    hist_d = torch.from_numpy(hist0.view(np.int32).copy()).to(dev)
    wd = torch.from_numpy(w.view(np.int32)).to(dev)
    GUARD = -7654321
    got, b0 = [], 0
    for nw in [s4 for s4 in (4, 252, 256, 2048, 8, 504, 2048, 1000) if b0 + s4 <= total]:
        if b0 + nw > total:
            break
        out = torch.full((n_streams, nw + 4), GUARD, dtype=torch.int32, device=dev)
        chunk = wd[:, b0:b0 + nw].contiguous()
        L = uchirp.lib()
        import ctypes as C
        rc = L.uc_dfsdm_sinc5_streams(e._h, C.c_void_p(chunk.data_ptr()), n_streams, nw, 0, C.c_void_p(hist_d.data_ptr()),
                                      C.c_void_p(out.data_ptr()), nw + 4, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, L.uc_last_error()
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        assert (o[:, nw:] == GUARD).all(), "stored past the end of a row (chunk of %d)" % nw
        got.append(o[:, :nw])
        b0 += nw
    got = np.concatenate(got, axis=1)
    for s in check:
        assert np.array_equal(got[s], ref[s][:got.shape[1]]), s
    assert np.array_equal(hist_d.cpu().numpy().view(np.uint32), w[:, b0 - 4:b0])
    # every buffer on its own: device words with a row pitch wider than the chunk, HOST history, HOST outputs with a pitch
    if n_streams <= 64:
        import ctypes as C
        L = uchirp.lib()
        nw, pitch_in, pitch_out = 504, 512, 600
        wide = torch.zeros((n_streams, pitch_in), dtype=torch.int32, device=dev)
        wide[:, :nw] = wd[:, :nw]
        wide[:, nw:] = 0x5A5A5A5A                                                 # (never read: the filter sees nw words per row)
        h_host = hist0.copy()
        o_host = np.full((n_streams, pitch_out), 77, np.int32)
        rc = L.uc_dfsdm_sinc5_streams(e._h, C.c_void_p(wide.data_ptr()), n_streams, nw, pitch_in, h_host.ctypes.data_as(C.c_void_p),
                                      o_host.ctypes.data_as(C.c_void_p), pitch_out, None)
        assert rc == 0, L.uc_last_error()
        for s in check:
            assert np.array_equal(o_host[s, :nw], ref[s][:nw]), s
        assert (o_host[:, nw:] == 77).all() and np.array_equal(h_host, w[:, nw - 4:nw])
        assert L.uc_dfsdm_sinc5_streams(e._h, C.c_void_p(wide.data_ptr()), n_streams, nw, nw - 4, h_host.ctypes.data_as(C.c_void_p),
                                        o_host.ctypes.data_as(C.c_void_p), pitch_out, None) < 0          # rows overlap
    # refused: misaligned device rows, overlapping streams
    with pytest.raises(uchirp.UchirpError):
        e.dfsdm_streams(wd[:, 1:7].contiguous()[:, :5].contiguous(), hist_d)     # 5-word rows: stride not a multiple of 4
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_streams,n_words", [(300000, 8), (5000, 5000), (8192, 2052), (3, 70001), (700, 12292), (1, 600003),
                                               (4097, 2048), (33, 257)])
def test_sinc5_segment_geometries(uchirp, n_streams, n_words):
    """The kernel walks a stream in SEGMENTS of up to 8 tiles of 256 words, one wave per segment, the tile-to-tile carry in
    scalar registers, and looks up what lies in front of 64 of a wave's segments at once: shapes with more than 64 segments
    per wave (300 000 short streams), several segments per stream with a short last one (5000, 12 292, 70 001 words), a
    one-word ninth tile (2052), fewer segments than waves (3 streams: segments shortened to fill the grid), one long ragged
    stream, 257 words (a second tile of one word).  Random histories; every checked stream bit-exact against the oracle; the
    history array ends as the last four words of every stream."""
    e = uchirp.Engine(uchirp.RX_REAL)
    rng = np.random.default_rng(n_streams * 31 + n_words)
    w = rng.integers(0, 1 << 32, size=(n_streams, n_words), dtype=np.uint64).astype(np.uint32)
    hist = rng.integers(0, 1 << 32, size=(n_streams, 4), dtype=np.uint64).astype(np.uint32)
    hist0 = hist.copy()
    got = e.dfsdm_streams(w, hist)
    check = range(n_streams) if n_streams <= 64 else sorted(set(rng.integers(0, n_streams, size=60).tolist()) | {0, n_streams - 1})
    for s in check:
        assert np.array_equal(got[s], _sinc5_with_history(w[s], hist0[s])), s
    tail = np.concatenate([hist0, w], axis=1)[:, -4:]
    assert np.array_equal(hist, tail)
    e.close()


def _pdm_transmissions(count, blocks, seed):
    """`count` microphones: noise lead, one transmission each, rendered at the PDM bit rate (32 x 78125 Hz) and delta-sigma
    modulated -> (pdm uint32 [count, blocks * 2048], messages)."""
    from uchirp import tx
    rng = np.random.default_rng(seed)
    n_bits = blocks * 2048 * 32
    pdm = np.zeros((count, blocks * 2048), np.uint32)
    msgs = []
    for s in range(count):
        msg = "".join(chr(int(c)) for c in rng.integers(48, 123, size=int(rng.integers(2, 6))))
        tone = tx.render(msg, fs_rx=78125.0 * 32, amplitude=0.35)
        lead = (int(rng.integers(25, 40)) * 2048 + int(rng.integers(0, 2048))) * 32
        x = rng.standard_normal(n_bits) * 0.01
        assert lead + tone.size <= n_bits
        x[lead:lead + tone.size] += tone
        pdm[s] = uco.pdm_modulate(np.clip(x, -1.0, 1.0).astype(np.float32))
        msgs.append(msg)
    return pdm, msgs


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [uco.RX_REAL, uco.SYNC_CPLX])
def test_live_receivers_fed_with_the_microphones_bit_streams(uchirp, variant):
    """UC_DTYPE_PDM: the live chain of a node starts at the microphones' 1-bit PDM streams (receiver/Src/dfsdm.c:59-61,69,78 ->
    main.c:659-668 -> 417-554).  The DFSDM runs on the device in front of the ISR's FIFO, its filter history travels in the
    uc_rx_state.  Texts and traces -- block by block, in ragged chunks, with dropped blocks, recorded in one call -- equal,
    bit for bit, the receiver fed with the ORACLE's DFSDM words of the same bit streams; the messages decode.  The live step
    from bits also replays from one captured hipGraph."""
    import torch
    N = 2048
    blocks, ns = 112, 5
    pdm, msgs = _pdm_transmissions(ns, blocks, seed=40 + variant)
    words = np.stack([_sinc5_with_history(pdm[s]) for s in range(ns)])
    assert words.shape == (ns, blocks * N)
    e = uchirp.Engine(variant)
    ref_t, ref_tr = e.receive_many(words)
    assert sum(m in t for m, t in zip(msgs, ref_t)) >= ns - 2, (msgs, ref_t)   # (a bit slip at acquisition is the firmware's own)
    rng = np.random.default_rng(3)
    busy = (rng.random((ns, blocks)) < 0.05).astype(np.uint8)
    ref_bt, ref_btr = e.receive_many(words, busy=busy)
    # recorded, one call
    t, tr = e.receive_many(pdm, pdm=True)
    assert t == ref_t and all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(tr, ref_tr))
    t, tr = e.receive_many(pdm, busy=busy, pdm=True)
    assert t == ref_bt and all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(tr, ref_btr))
    # live: one block per call (device chunks); ragged chunks with dropped blocks (host chunks)
    for sizes, bz, want_t, want_tr in (([1] * blocks, None, ref_t, ref_tr), ([3, 1, 40, 2, 66], busy, ref_bt, ref_btr)):
        live = e.live(ns)
        texts, traces, b0 = [""] * ns, [[] for _ in range(ns)], 0
        for k, nb in enumerate(sizes):
            chunk = np.ascontiguousarray(pdm[:, b0 * N:(b0 + nb) * N])
            arg = torch.from_numpy(chunk.view(np.int32)).to("cuda:0") if bz is None else chunk
            t, tr = live.next(arg, busy=None if bz is None else np.ascontiguousarray(bz[:, b0:b0 + nb]), pdm=True)
            for s in range(ns):
                texts[s] += t[s]
                traces[s].append(tr[s])
            b0 += nb
        assert b0 == blocks
        for s in range(ns):
            assert texts[s] == want_t[s], s
            assert np.array_equal(np.concatenate(traces[s]).view(np.uint8), want_tr[s].view(np.uint8)), s
        live.close()
    # the same step -- DFSDM, history update, ROWS band launch, replay: four launches -- captured ONCE into a hipGraph and
    # replayed for every block of bits that arrives (one eager step first: tables and scratch are allocated by it)
    dev = torch.device("cuda:0")
    pd = torch.from_numpy(pdm.view(np.int32)).to(dev)
    live = e.live(ns)
    chunk = torch.zeros((ns, N), dtype=torch.int32, device=dev)
    text = torch.zeros((ns, 8), dtype=torch.uint8, device=dev)
    ntext = torch.zeros(ns, dtype=torch.int32, device=dev)
    live.next_into(chunk, text, ntext, pdm=True)
    live.reset()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream()
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        with torch.cuda.graph(gr, stream=cs):
            live.next_into(chunk, text, ntext, stream=cs.cuda_stream, pdm=True)
    texts = [b""] * ns
    for b in range(blocks):
        chunk.copy_(pd[:, b * N:(b + 1) * N])
        gr.replay()
        torch.cuda.synchronize()
        nt, tt = ntext.cpu().numpy(), text.cpu().numpy()
        for k in range(ns):
            texts[k] += bytes(tt[k, :nt[k]])
    assert [t.decode("latin-1") for t in texts] == ref_t
    live.close()
    # the node's form: uc_group_receive_streams_next takes the bit streams too (world size 1 here: a rank's whole code path)
    g = uchirp.Group(variant, devices=[0])
    st = g.rx_state(0, ns)
    cap = 64
    th, nh, acc, at = np.zeros((ns, cap), np.uint8), np.zeros(ns, np.uint32), [""] * ns, 0
    for nb in (2, 50, 60):
        chunk = np.ascontiguousarray(pdm[:, at * N:(at + nb) * N]).view(np.int32)
        g.receive_streams([chunk], ns, nb * N, [th], cap, n_text=[nh], states=[st], dtype=uchirp.DTYPE_PDM)
        acc = [a + bytes(th[i, :nh[i]]).decode("latin-1") for i, a in enumerate(acc)]
        at += nb
    assert at == blocks and acc == ref_t
    g.rx_state_destroy(st)
    g.close()
    # a state that began with PDM words refuses sample words in mid-stream
    live = e.live(ns)
    live.next(pdm[:, :N], pdm=True)
    with pytest.raises(uchirp.UchirpError):
        live.next(words[:, N:2 * N])
    live.close()
    e.close()
