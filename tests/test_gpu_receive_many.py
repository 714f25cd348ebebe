"""uc_receive_streams: the receiver's main loop (receiver/Src/main.c:417-554, resync 243-273, ISR FIFO 659-668) for many
recorded streams in one call -- pack kernel, one batched launch, the switch replayed on the device one lane per stream.

  * bit for bit the text and trace uc_receive_stream_isr gives each stream alone (the host replay of the same header),
    with and without dropped blocks, float32 and int32 words, host and device buffers;
  * against the oracle's literal sequential loop on 1000 random transmissions with dropped blocks -- random text, amplitude,
    noise, lead, skew, busy masks (the cases of tools/fuzz_receive.py): every trace equal, except where the ORACLE's own
    snrs show the diverging decision within round-off of a threshold or of a tie."""
import os
import subprocess

import numpy as np
import pytest

from uchirp import tx
from oracle import uco

pytestmark = pytest.mark.gpu

N = 2048
FIELDS = ("block", "state_before", "state_after", "bit", "sync_position")


@pytest.fixture(scope="module")
def uchirp():
    import uchirp as m
    m.lib()
    return m


def _transmissions(count, seed, blocks):
    """`count` streams of `blocks` blocks: noise lead (25 .. 45 blocks + a sample skew), one transmission of a random text,
    noise to the end; (streams float32 [count, blocks * N], busy uint8 [count, blocks], messages)."""
    rng = np.random.default_rng(seed)
    x = np.zeros((count, blocks * N), np.float32)
    busy = np.zeros((count, blocks), np.uint8)
    msgs = []
    for s in range(count):
        msg = "".join(chr(int(c)) for c in rng.integers(32, 127, size=int(rng.integers(1, 7))))
        amp = float(rng.choice([500.0, 2000.0, 8000.0]))
        sigma = amp * float(rng.choice([0.01, 0.05, 0.2, 0.5]))
        tone = tx.render(msg, fs_rx=78125.0, amplitude=amp)
        lead = int(rng.integers(25, 46)) * N + int(rng.integers(0, N))
        row = rng.standard_normal(blocks * N) * sigma
        assert lead + tone.size <= row.size
        row[lead:lead + tone.size] += tone
        x[s] = row.astype(np.float32)
        if rng.random() < 0.5:
            busy[s] = rng.random(blocks) < float(rng.choice([0.02, 0.1, 0.3]))
        msgs.append(msg)
    return x, busy, msgs


@pytest.mark.parametrize("variant", [uco.SYNC_CPLX, uco.RX_REAL])
def test_many_streams_equal_one_stream_at_a_time(uchirp, variant):
    import torch
    x, busy, msgs = _transmissions(70, seed=11 + variant, blocks=160)
    e = uchirp.Engine(variant)
    texts, traces = e.receive_many(x, busy=busy)
    decoded = 0
    for s in range(x.shape[0]):
        t1, tr1 = e.receive(x[s], busy=busy[s])
        assert texts[s] == t1, s
        assert len(traces[s]) == len(tr1) == int((busy[s] == 0).sum())
        assert np.array_equal(traces[s].view(np.uint8), tr1.view(np.uint8)), s     # every field, snrs included, bit for bit
        decoded += int(msgs[s] in t1)
    assert decoded >= 20
    # no busy mask; device-resident input; int32 DFSDM words
    t2, tr2 = e.receive_many(torch.from_numpy(x).to("cuda:0"))
    xi = (np.round(x).astype(np.int64) * 256).astype(np.int32)
    t3, tr3 = e.receive_many(xi)
    for s in range(0, x.shape[0], 7):
        t1, tr1 = e.receive(x[s])
        assert t2[s] == t1 and np.array_equal(tr2[s].view(np.uint8), tr1.view(np.uint8))
        t1i, tr1i = e.receive(xi[s])
        assert t3[s] == t1i and np.array_equal(tr3[s].view(np.uint8), tr1i.view(np.uint8))
    # streams NOT a multiple of 256 samples apart (the packed path without a busy mask): the ragged tail is ignored
    xr = np.concatenate([x[:9], np.full((9, 100), 7.0, np.float32)], axis=1)
    t4, tr4 = e.receive_many(xr)
    for s in range(9):
        t1, tr1 = e.receive(x[s])
        assert t4[s] == t1 and np.array_equal(tr4[s].view(np.uint8), tr1.view(np.uint8))
    # one stream, a stream shorter than a block, every block dropped, no trace wanted, a tiny text buffer
    t, tr = e.receive_many(x[:1])
    assert t[0] == texts[0] or busy[0].any()
    t, tr = e.receive_many(np.zeros((3, 100), np.float32))
    assert t == ["", "", ""] and all(len(q) == 0 for q in tr)
    t, tr = e.receive_many(x[:4], busy=np.ones((4, 160), np.uint8))
    assert t == [""] * 4 and all(len(q) == 0 for q in tr)
    t, tr = e.receive_many(x[:8], want_trace=False, text_cap=3)
    assert tr is None and all(len(q) <= 2 for q in t)
    # more streams than a grid dimension holds (70 000 one-block streams), with and without a busy mask
    many = np.tile(x[:7, :N], (10000, 1))
    t, tr = e.receive_many(many, want_trace=False)
    assert len(t) == 70000 and all(q == "" for q in t)
    t, tr = e.receive_many(many, busy=np.zeros((70000, 1), np.uint8))
    assert all(len(q) == 1 for q in tr) and all(q == "" for q in t)
    e.close()
    with pytest.raises(uchirp.UchirpError):
        uchirp.Engine(uchirp.COMPRESS).receive_many(x[:2])


SOFT_GAP = 2e-3     # a decision whose two sides are this close (relative) in the ORACLE may legitimately go the other way in float32


def classify_divergence(tr_g, tr_o, margin_o):
    """First block where the traces part, judged by the oracle's OWN decision margin there (uco_receive_stream_diag): every
    decision of the block -- the acquisition maximum against (1 + SNR_THRESHOLD) x mag_mean and against the runner-up of the
    eight maxima, snr_up / snr_down against the threshold and each other, resync()'s compares -- -> ("soft" | "bad", block, gap)."""
    n = min(len(tr_g), len(tr_o))
    d = [i for i in range(n) if any(tr_g[f][i] != tr_o[f][i] for f in FIELDS)]
    i = d[0] if d else n
    gap = float(margin_o[i]) if i < len(margin_o) else float("inf")
    return ("soft" if gap < SOFT_GAP else "bad"), i, gap


def test_a_thousand_random_transmissions_with_dropped_blocks_against_the_oracle(uchirp):
    x, busy, msgs = _transmissions(1000, seed=5, blocks=150)
    variants = np.random.default_rng(6).integers(0, 2, size=1000)        # 0 RX_REAL, 1 SYNC_CPLX
    bad = soft = decoded = 0
    for variant in (uco.RX_REAL, uco.SYNC_CPLX):
        idx = np.nonzero(variants == variant)[0]
        e, o = uchirp.Engine(variant), uco.Oracle(variant)
        texts, traces = e.receive_many(x[idx], busy=busy[idx])
        for k, s in enumerate(idx):
            text_o, tr_o, mg_o = o.receive(x[s], precision=uco.F64, busy=busy[s], margins=True)
            tr_g = traces[k]
            decoded += int(msgs[s] in text_o)
            same = texts[k] == text_o and len(tr_g) == len(tr_o) and all(np.array_equal(tr_g[f], tr_o[f]) for f in FIELDS)
            if same:
                continue
            # a decision within float32 round-off of going the other way may differ: the first diverging block is judged by
            # the ORACLE's own margin there -- acquisition blocks included (the block maximum against 3 x mag_mean, the two
            # largest of the eight maxima), not waved through
            kind, i, gap = classify_divergence(tr_g, tr_o, mg_o)
            if kind == "soft":
                soft += 1
                print("soft: stream %d variant %d block %d: the oracle's closest decision there had a relative gap of %.2e" % (s, variant, i, gap))
            else:
                bad += 1
                print("FAIL stream %d variant %d block %d gap %.3e: %r vs %r" % (s, variant, i, gap, text_o, texts[k]))
        e.close()
    print("1000 transmissions: %d failures, %d divergences at decisions within %.0e of a tie in the oracle, oracle decoded the text in %d"
          % (bad, soft, SOFT_GAP, decoded))
    assert bad == 0 and soft <= 3 and decoded >= 300


@pytest.mark.parametrize("variant", [uco.SYNC_CPLX, uco.RX_REAL])
def test_live_streams_chunk_after_chunk_equal_the_whole_stream(uchirp, variant):
    """uc_rx_state / uc_receive_streams_next: the firmware's own mode of operation -- blocks arrive for ever.  A stream cut
    into chunks of ANY sizes (one block at a time included) gives the text and the trace of the whole stream in one call,
    bit for bit: the FIFO's last two accepted blocks and main()'s locals travel in the state.  With and without dropped
    blocks (the packed and the copy-free path), float32 and int32 words, device-resident chunks."""
    import torch
    x, busy, msgs = _transmissions(40, seed=31 + variant, blocks=150)
    e = uchirp.Engine(variant)
    rng = np.random.default_rng(9)
    for use_busy, data in ((True, x), (False, x), (False, (np.round(x).astype(np.int64) * 256).astype(np.int32))):
        bz = busy if use_busy else None
        whole_t, whole_tr = e.receive_many(data, busy=bz)
        for sizes in ([1] * 150, [150], [7, 1, 1, 60, 2, 79], list(rng.integers(1, 12, size=60))):
            sizes = list(sizes)
            while sum(sizes) > 150:
                sizes.pop()
            if sum(sizes) < 150:
                sizes.append(150 - sum(sizes))
            live = e.live(data.shape[0])
            texts = [""] * data.shape[0]
            traces = [[] for _ in range(data.shape[0])]
            b0 = 0
            for k, nb in enumerate(sizes):
                chunk = np.ascontiguousarray(data[:, b0 * N:(b0 + nb) * N])
                cb = None if bz is None else np.ascontiguousarray(bz[:, b0:b0 + nb])
                arg = torch.from_numpy(chunk).to("cuda:0") if (k % 3 == 1) else chunk
                t, tr = live.next(arg, busy=cb)
                for s in range(data.shape[0]):
                    texts[s] += t[s]
                    traces[s].append(tr[s])
                b0 += nb
            for s in range(data.shape[0]):
                assert texts[s] == whole_t[s], (s, sizes[:6])
                got = np.concatenate(traces[s]) if traces[s] else np.zeros(0, whole_tr[s].dtype)
                assert np.array_equal(got.view(np.uint8), whole_tr[s].view(np.uint8)), (s, sizes[:6])
            live.close()
    # the lane-per-stream replay (more than 16 Ki streams) carries its state the same way: 20 000 streams, the WHOLE
    # transmission (acquisition, tracking, data), one block per call and eight
    many = torch.from_numpy(x[:8]).to("cuda:0").repeat(2500, 1)
    w_t, w_tr = e.receive_many(many)
    assert sum(m in t for m, t in zip(msgs[:8] * 2500, w_t)) >= 2500
    for per in (1, 8):
        live = e.live(many.shape[0])
        pick = range(0, many.shape[0], 397)
        texts, traces = {s: "" for s in pick}, {s: [] for s in pick}
        for b0 in range(0, 150, per):
            t, tr = live.next(many[:, b0 * N:min(b0 + per, 150) * N].contiguous())
            for s in pick:
                texts[s] += t[s]
                traces[s].append(tr[s])
        for s in pick:
            assert texts[s] == w_t[s], (per, s)
            assert np.array_equal(np.concatenate(traces[s]).view(np.uint8), w_tr[s].view(np.uint8)), (per, s)
        live.close()
    # reset = power-on again; a state is tied to its context and its stream count; partial blocks are refused
    live = e.live(3)
    t1, _ = live.next(x[:3, :40 * N])
    live.reset()
    t2, tr2 = live.next(x[:3, :40 * N])
    assert t1 == t2 and tr2[0]["block"][0] == 0
    with pytest.raises(uchirp.UchirpError):
        live.next(x[:3, :N + 5])
    with pytest.raises(ValueError):
        live.next(x[:4, :N])
    other = uchirp.Engine(variant)
    with pytest.raises(uchirp.UchirpError):
        other.receive_many(x[:3, :N], _state=live)
    live.close()
    other.close()
    e.close()


@pytest.mark.parametrize("variant,busy_mask", [(uco.RX_REAL, False), (uco.SYNC_CPLX, False), (uco.RX_REAL, True)])
def test_live_step_captured_into_a_graph_and_replayed_block_after_block(uchirp, variant, busy_mask):
    """One live step -- the new block of every stream in, texts and trace records out -- captured ONCE into a hipGraph and
    replayed for every block that arrives: everything a step carries (the newest block, the 9 FIFO records that survive the
    ISR's shift, main()'s locals, the stream's block count) lives on the device, so the replays continue one another.
    Texts and traces equal the eager chunked calls and the whole-stream call, bit for bit; with a busy mask too."""
    import torch
    dev = torch.device("cuda:0")
    blocks, ns = 120, 24
    x, busy, msgs = _transmissions(ns, seed=71 + variant, blocks=blocks)
    if not busy_mask:
        busy[:] = 0
    e = uchirp.Engine(variant)
    whole_t, whole_tr = e.receive_many(x, busy=busy if busy_mask else None)
    xd = torch.from_numpy(x).to(dev)
    bd = torch.from_numpy(busy).to(dev)
    live = e.live(ns)
    chunk = torch.zeros((ns, N), dtype=torch.float32, device=dev)
    bz = torch.zeros((ns, 1), dtype=torch.uint8, device=dev) if busy_mask else None
    text = torch.zeros((ns, 8), dtype=torch.uint8, device=dev)
    ntext = torch.zeros(ns, dtype=torch.int32, device=dev)
    trace = torch.zeros((ns, 1, uchirp.RX_EVENT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    ntrace = torch.zeros(ns, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    # one eager step sizes the state's scratch (nothing may be allocated during a capture); then back to power-on
    live.next_into(chunk, text, ntext, trace=trace, n_trace=ntrace, busy=bz)
    live.reset()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            live.next_into(chunk, text, ntext, trace=trace, n_trace=ntrace, busy=bz, stream=s.cuda_stream)
    texts = [b""] * ns
    traces = [[] for _ in range(ns)]
    for b in range(blocks):
        chunk.copy_(xd[:, b * N:(b + 1) * N])
        if busy_mask:
            bz.copy_(bd[:, b:b + 1])
        g.replay()
        torch.cuda.synchronize()
        nt, tt = ntext.cpu().numpy(), text.cpu().numpy()
        ntr = ntrace.cpu().numpy()
        tr = trace.cpu().numpy().reshape(ns, -1).view(uchirp.RX_EVENT_DTYPE)
        for k in range(ns):
            texts[k] += bytes(tt[k, :nt[k]])
            if ntr[k]:
                traces[k].append(tr[k, :1].copy())
    for k in range(ns):
        assert texts[k].decode("latin-1") == whole_t[k], k
        got = np.concatenate(traces[k]) if traces[k] else np.zeros(0, whole_tr[k].dtype)
        assert np.array_equal(got.view(np.uint8), whole_tr[k].view(np.uint8)), k
    assert sum(m in t for m, t in zip(msgs, whole_t)) >= (4 if busy_mask else 6)
    live.close()
    e.close()


@pytest.mark.parametrize("env", [{"UC_GRID": "5"}, {"UC_GRID": "3", "UC_BAND_GROUP": "1"}, {"UC_BAND_GROUP": "2"},
                                 {"UC_BAND_GROUP": "4", "UC_GRID": "64"}, {"UC_STATIC_DEAL": "1", "UC_BAND_GROUP": "8"}, {"UC_GRID": "1"}])
def test_live_receivers_under_odd_launch_geometries(uchirp, monkeypatch, env):
    """The ROWS build passes over the offsets an IDLE stream's switch cannot look at; the frame loop's bookkeeping (the parked
    group id, the ring drain at a group start, the loads handed on to the row's next needed unit) must hold for every group
    size and grid -- groups of one or two units that are passed over entirely, a single workgroup, tickets and the static deal.
    One block per call, both receivers, against the recorded call (which evaluates everything)."""
    monkeypatch.setenv("UC_TUNING", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    x, busy, msgs = _transmissions(21, seed=123, blocks=130)
    for variant in (uco.RX_REAL, uco.SYNC_CPLX):
        e = uchirp.Engine(variant)
        whole_t, whole_tr = e.receive_many(x)
        live = e.live(x.shape[0])
        texts, traces = [""] * x.shape[0], [[] for _ in range(x.shape[0])]
        for b in range(130):
            t, tr = live.next(np.ascontiguousarray(x[:, b * N:(b + 1) * N]))
            for s in range(x.shape[0]):
                texts[s] += t[s]
                traces[s].append(tr[s])
        for s in range(x.shape[0]):
            assert texts[s] == whole_t[s], (env, variant, s)
            assert np.array_equal(np.concatenate(traces[s]).view(np.uint8), whole_tr[s].view(np.uint8)), (env, variant, s)
        assert e.busy_counters() == 0
        live.close()
        e.close()


@pytest.mark.parametrize("variant", [uco.RX_REAL, uco.SYNC_CPLX])
def test_what_live_receivers_pass_over_is_never_looked_at(uchirp, monkeypatch, variant):
    """One-block live calls evaluate only the FIFO offsets main()'s switch can still look at: an IDLE stream 3 or 5 of the 8 its
    new block adds, the UP reference only (main.c:447-453); a SYNCHRONIZED / DATA_RECEIVING stream the positions around its
    sync_position now and one block on, plus the acquisition set it would fall back to (main.c:491-550, 243-273).  Under the tests'
    poison switch everything that is passed over gets a HUGE statistic instead of zero: a switch that looked at one of them would
    lock, move or decode differently.  Texts and traces still equal the recorded call (which evaluates everything), bit for bit."""
    monkeypatch.setenv("UC_TUNING", "1")
    monkeypatch.setenv("UC_RX_POISON", "1")
    x, busy, msgs = _transmissions(48, seed=500 + variant, blocks=170)
    e = uchirp.Engine(variant)
    for bz in (None, busy):
        whole_t, whole_tr = e.receive_many(x, busy=bz)
        live = e.live(x.shape[0])
        texts, traces = [""] * x.shape[0], [[] for _ in range(x.shape[0])]
        for b in range(170):
            t, tr = live.next(np.ascontiguousarray(x[:, b * N:(b + 1) * N]), busy=None if bz is None else np.ascontiguousarray(bz[:, b:b + 1]))
            for s in range(x.shape[0]):
                texts[s] += t[s]
                traces[s].append(tr[s])
        for s in range(x.shape[0]):
            assert texts[s] == whole_t[s], (variant, s)
            assert np.array_equal(np.concatenate(traces[s]).view(np.uint8), whole_tr[s].view(np.uint8)), (variant, s)
        live.close()
    assert sum(m in t for m, t in zip(msgs, whole_t)) >= 5      # (with dropped blocks; tracking states were visited)
    e.close()


@pytest.mark.parametrize("variant", [uco.RX_REAL, uco.SYNC_CPLX])
def test_live_receivers_whose_caller_keeps_the_previous_chunk(uchirp, monkeypatch, variant):
    """uc_rx_state_keep_previous: the caller receives into a ring of chunk buffers and leaves every chunk where it is until the
    next call has completed, so the block in front of a call's first block (what the ISR keeps in fifo_queue, main.c:662) is
    read from the previous chunk and nothing is copied into the state.  Texts and traces equal the recorded call bit for bit:
    one block per call (the masked walk) and chunks of any sizes, float32 and int32 words, rings of two and three buffers with
    strides larger than the chunk, under the poison switch, with busy-masked calls, calls on host memory and a switched-off
    contract mixed in, and after a reset."""
    import torch
    monkeypatch.setenv("UC_TUNING", "1")
    monkeypatch.setenv("UC_RX_POISON", "1")
    dev = torch.device("cuda:0")
    blocks = 150
    x, busy, msgs = _transmissions(37, seed=900 + variant, blocks=blocks)
    ns = x.shape[0]
    e = uchirp.Engine(variant)
    rng = np.random.default_rng(5)
    for data in (x, (np.round(x).astype(np.int64) * 256).astype(np.int32)):
        whole_t, whole_tr = e.receive_many(data)
        xd = torch.from_numpy(data).to(dev)
        for ring_n, sizes in ((2, [1] * blocks), (3, [1] * blocks), (2, [5, 1, 1, 40, 2, 1, 1, 1, 98]),
                              (3, list(rng.integers(1, 9, size=80)))):
            sizes = list(int(v) for v in sizes)
            while sum(sizes) > blocks:
                sizes.pop()
            if sum(sizes) < blocks:
                sizes.append(blocks - sum(sizes))
            cap = max(sizes)
            # a ring of buffers wider than the chunk (row stride > chunk): what lies beyond a chunk is never read
            ring = [torch.full((ns, cap * N + 64), 7, dtype=xd.dtype, device=dev) for _ in range(ring_n)]
            live = e.live(ns)
            live.keep_previous(True)
            texts, traces = [""] * ns, [[] for _ in range(ns)]
            b0 = 0
            for k, nb in enumerate(sizes):
                buf = ring[k % ring_n]
                buf[:, :nb * N].copy_(xd[:, b0 * N:(b0 + nb) * N])
                t, tr = live.next(buf[:, :nb * N])
                for s_ in range(ns):
                    texts[s_] += t[s_]
                    traces[s_].append(tr[s_])
                b0 += nb
            for s_ in range(ns):
                assert texts[s_] == whole_t[s_], (ring_n, sizes[:6], s_)
                assert np.array_equal(np.concatenate(traces[s_]).view(np.uint8), whole_tr[s_].view(np.uint8)), (ring_n, sizes[:6], s_)
            live.close()
    assert sum(m in t for m, t in zip(msgs, whole_t)) >= 10
    # mixed: kept device chunks, busy-masked calls, host chunks (staged by the library), the contract switched off and on again,
    # one block per call -- against the recorded call with the same busy mask
    whole_t, whole_tr = e.receive_many(x, busy=busy)
    xd = torch.from_numpy(x).to(dev)
    ring = [torch.zeros((ns, N), dtype=torch.float32, device=dev) for _ in range(2)]
    live = e.live(ns)
    live.keep_previous(True)
    texts, traces = [""] * ns, [[] for _ in range(ns)]
    for b in range(blocks):
        bz = np.ascontiguousarray(busy[:, b:b + 1])
        if b == 60:
            live.keep_previous(False)
        if b == 75:
            live.keep_previous(True)
        if b % 7 == 3:
            arg = np.ascontiguousarray(x[:, b * N:(b + 1) * N])     # host memory
        else:
            arg = ring[b % 2]
            arg.copy_(xd[:, b * N:(b + 1) * N])
        t, tr = live.next(arg, busy=bz if bz.any() else None)
        for s_ in range(ns):
            texts[s_] += t[s_]
            traces[s_].append(tr[s_])
    for s_ in range(ns):
        assert texts[s_] == whole_t[s_], s_
        assert np.array_equal(np.concatenate(traces[s_]).view(np.uint8), whole_tr[s_].view(np.uint8)), s_
    # reset = power-on again, also for what is kept
    live.reset()
    t2, tr2 = live.next(xd[:, :40 * N].contiguous())
    t1, tr1 = e.receive_many(x[:, :40 * N])
    assert t1 == t2 and all(np.array_equal(a.view(np.uint8), b_.view(np.uint8)) for a, b_ in zip(tr1, tr2))
    live.close()
    e.close()


@pytest.mark.parametrize("variant", [uco.RX_REAL, uco.SYNC_CPLX])
def test_kept_chunks_two_captured_steps_replayed_in_turn(uchirp, variant):
    """With uc_rx_state_keep_previous a captured step bakes in where its chunk and the chunk in front of it lie: a ring of two
    buffers is two graphs (A after B, B after A) replayed in turn.  Texts and traces equal the recorded call bit for bit."""
    import torch
    dev = torch.device("cuda:0")
    blocks, ns = 121, 24
    x, busy, msgs = _transmissions(ns, seed=171 + variant, blocks=blocks)
    e = uchirp.Engine(variant)
    whole_t, whole_tr = e.receive_many(x)
    xd = torch.from_numpy(x).to(dev)
    live = e.live(ns)
    live.keep_previous(True)
    ring = [torch.zeros((ns, N), dtype=torch.float32, device=dev) for _ in range(2)]
    text = torch.zeros((ns, 8), dtype=torch.uint8, device=dev)
    ntext = torch.zeros(ns, dtype=torch.int32, device=dev)
    trace = torch.zeros((ns, 1, uchirp.RX_EVENT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    ntrace = torch.zeros(ns, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    texts, traces = [b""] * ns, [[] for _ in range(ns)]

    def collect():
        torch.cuda.synchronize()
        nt, tt = ntext.cpu().numpy(), text.cpu().numpy()
        ntr = ntrace.cpu().numpy()
        tr = trace.cpu().numpy().reshape(ns, -1).view(uchirp.RX_EVENT_DTYPE)
        for k in range(ns):
            texts[k] += bytes(tt[k, :nt[k]])
            if ntr[k]:
                traces[k].append(tr[k, :1].copy())

    # block 0 eagerly (it sizes the scratch; the block in front of it is the state's power-on FIFO), then the two captures --
    # each capture IS a step: blocks 1 and 2 are in the buffers while the captures are made, and are run by the first replays
    ring[0].copy_(xd[:, :N])
    live.next_into(ring[0], text, ntext, trace=trace, n_trace=ntrace)
    collect()
    graphs = []
    s.wait_stream(torch.cuda.current_stream())
    for k in (1, 0):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                live.next_into(ring[k], text, ntext, trace=trace, n_trace=ntrace, stream=s.cuda_stream)
        graphs.append(g)                # graphs[0]: chunk in ring[1] behind ring[0]; graphs[1]: ring[0] behind ring[1]
    for b in range(1, blocks):
        ring[b % 2].copy_(xd[:, b * N:(b + 1) * N])
        graphs[(b + 1) % 2].replay()
        collect()
    for k in range(ns):
        assert texts[k].decode("latin-1") == whole_t[k], k
        got = np.concatenate(traces[k]) if traces[k] else np.zeros(0, whole_tr[k].dtype)
        assert np.array_equal(got.view(np.uint8), whole_tr[k].view(np.uint8)), k
    assert sum(m in t for m, t in zip(msgs, whole_t)) >= 6
    live.close()
    e.close()


def test_what_cannot_be_captured_is_refused_not_recorded(uchirp):
    """Only uc_receive_streams_next can be captured (a uc_rx_state owns its scratch): the state-less call shares the context's
    scratch through an event a capture cannot carry and answers -ENOTSUP; a live step whose scratch has not been sized by an
    eager call of the same shape would have to allocate inside the capture and answers -ENOBUFS.  Neither records anything, and
    the same objects work afterwards."""
    import torch
    dev = torch.device("cuda:0")
    ns = 6
    e = uchirp.Engine(uco.RX_REAL)
    x = torch.randn((ns, N), device=dev) * 50.0
    text = torch.zeros((ns, 8), dtype=torch.uint8, device=dev)
    ntext = torch.zeros(ns, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    live = e.live(ns)
    seen = []
    for what in ("stateless", "unsized"):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                try:
                    if what == "stateless":
                        e.receive_many_into(x, text, ntext, stream=s.cuda_stream)
                    else:
                        live.next_into(x, text, ntext, stream=s.cuda_stream)
                    seen.append((what, "accepted"))
                except uchirp.UchirpError as ex:
                    seen.append((what, str(ex)))
    assert "captured" in seen[0][1] and "ENOTSUP" not in seen[0][0], seen
    assert "eager call" in seen[1][1], seen
    torch.cuda.synchronize()
    # both forms still work eagerly, and the live step is capturable once sized
    e.receive_many_into(x, text, ntext)
    live.next_into(x, text, ntext)
    live.reset()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            live.next_into(x, text, ntext, stream=s.cuda_stream)
    g.replay()
    torch.cuda.synchronize()
    live.close()
    e.close()


def test_refused_live_calls_leave_the_state_as_it_was(uchirp):
    """uc_receive_streams_next checks every argument and sizes all scratch BEFORE its first launch: a refused call (partial blocks,
    overlapping streams, another dtype in mid-stream, NULL text) has enqueued nothing -- the streams continue behind it exactly
    as if it had not been made -- and uc_rx_state_reset behind anything puts the receivers back to power-on."""
    import ctypes as C
    import torch
    x, busy, msgs = _transmissions(9, seed=4242, blocks=140)
    ns = x.shape[0]
    e = uchirp.Engine(uco.SYNC_CPLX)
    whole_t, whole_tr = e.receive_many(x)
    live = e.live(ns)
    L = uchirp.lib()
    texts, traces = [""] * ns, [[] for _ in range(ns)]
    xd = torch.from_numpy(x).to("cuda:0")
    tbuf = torch.zeros((ns, 8), dtype=torch.uint8, device="cuda:0")
    for b in range(140):
        if b % 10 == 5:
            ch = xd[:, b * N:(b + 1) * N].contiguous()
            args = dict(samples=C.c_void_p(ch.data_ptr()), dtype=uchirp.DTYPE_F32, nsmp=N, stride=0, text=C.c_void_p(tbuf.data_ptr()))
            bad = [dict(nsmp=N + 3), dict(stride=N - 1), dict(dtype=uchirp.DTYPE_I32), dict(text=None), dict(dtype=9)]
            for k in bad:
                a = dict(args, **k)
                rc = L.uc_receive_streams_next(e._h, live._h, a["samples"], a["dtype"], a["nsmp"], a["stride"], None, a["text"], 8,
                                               None, None, 0, None, None)
                assert rc < 0 and rc != -5, (k, rc)                 # refused for its arguments (never -EIO)
        t, tr = live.next(np.ascontiguousarray(x[:, b * N:(b + 1) * N]))
        for s_ in range(ns):
            texts[s_] += t[s_]
            traces[s_].append(tr[s_])
    for s_ in range(ns):
        assert texts[s_] == whole_t[s_], s_
        assert np.array_equal(np.concatenate(traces[s_]).view(np.uint8), whole_tr[s_].view(np.uint8)), s_
    live.reset()
    t2, tr2 = live.next(x[:, :60 * N])
    t1, tr1 = e.receive_many(x[:, :60 * N])
    assert t1 == t2 and all(np.array_equal(a.view(np.uint8), b_.view(np.uint8)) for a, b_ in zip(tr1, tr2))
    live.close()
    e.close()


@pytest.mark.parametrize("variant", [uco.SYNC_CPLX, uco.RX_REAL])
def test_calls_of_several_blocks_served_block_by_block(uchirp, monkeypatch, variant):
    """A call of several blocks without a busy mask is served as one-block steps of the live form when a step fills the chip
    (from 1024 / 8192 streams on; here forced for few streams: UC_TUNING=1 UC_RX_STEP_MIN=1): every step evaluates only what the
    switch can still look at, the block in front of step b is block b - 1 of the same buffer.  Texts and traces equal the one-launch
    call (which evaluates everything) and one stream at a time, bit for bit: recorded calls (float32 / int32 / PDM words, host and
    device memory, row strides larger than the stream), live chunks of any sizes with and without kept chunks, busy-masked calls
    (served in one launch as ever) mixed in -- all under the poison switch."""
    import torch
    dev = torch.device("cuda:0")
    blocks = 140
    x, busy, msgs = _transmissions(19, seed=321 + variant, blocks=blocks)
    ns = x.shape[0]
    plain = uchirp.Engine(variant)                                   # one launch per call (19 streams: below every threshold)
    monkeypatch.setenv("UC_TUNING", "1")
    monkeypatch.setenv("UC_RX_STEP_MIN", "1")
    monkeypatch.setenv("UC_RX_POISON", "1")
    e = uchirp.Engine(variant)                                       # block by block
    xi = (np.round(x).astype(np.int64) * 256).astype(np.int32)
    for data in (x, xi):
        want_t, want_tr = plain.receive_many(data)
        for arg in (data, torch.from_numpy(data).to(dev)):
            got_t, got_tr = e.receive_many(arg)
            assert got_t == want_t
            assert all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(got_tr, want_tr))
        # a row stride larger than the stream
        wide = torch.full((ns, blocks * N + 96), 3, dtype=torch.from_numpy(data).dtype, device=dev)
        wide[:, :blocks * N] = torch.from_numpy(data).to(dev)
        got_t, got_tr = e.receive_many(wide[:, :blocks * N])
        assert got_t == want_t and all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(got_tr, want_tr))
        for s_ in (0, 7, 18):                                        # == uc_receive_stream of the stream alone
            t1, tr1 = e.receive(data[s_])
            assert t1 == want_t[s_] and np.array_equal(tr1.view(np.uint8), want_tr[s_].view(np.uint8)), s_
    assert sum(m in t for m, t in zip(msgs, want_t)) >= (5 if variant == uco.SYNC_CPLX else 0)
    # PDM words: the DFSDM first, then the steps over its words
    rng = np.random.default_rng(8)
    bits = rng.integers(-(1 << 31), (1 << 31) - 1, size=(ns, 20 * N), dtype=np.int64).astype(np.int32)
    w_t, w_tr = plain.receive_many(bits, pdm=True)
    g_t, g_tr = e.receive_many(bits, pdm=True)
    assert g_t == w_t and all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(g_tr, w_tr))
    # live states: chunks of several blocks are stepped too; kept chunks; busy-masked chunks in between (one launch each)
    want_t, want_tr = plain.receive_many(x, busy=busy)
    xd = torch.from_numpy(x).to(dev)
    for kept in (False, True):
        live = e.live(ns)
        live.keep_previous(kept)
        sizes = [3, 1, 8, 2, 1, 16, 5, 1, 1, 40, 7, 9, 2, 44]
        assert sum(sizes) == blocks
        ring = [torch.zeros((ns, 44 * N), dtype=torch.float32, device=dev) for _ in range(2)]
        texts, traces, b0 = [""] * ns, [[] for _ in range(ns)], 0
        for k, nb in enumerate(sizes):
            buf = ring[k % 2]
            buf[:, :nb * N].copy_(xd[:, b0 * N:(b0 + nb) * N])
            bz = np.ascontiguousarray(busy[:, b0:b0 + nb])
            t, tr = live.next(buf[:, :nb * N], busy=bz)              # (a mask of zeros is still a mask: the one-launch path)
            for s_ in range(ns):
                texts[s_] += t[s_]
                traces[s_].append(tr[s_])
            b0 += nb
        for s_ in range(ns):
            assert texts[s_] == want_t[s_], (kept, s_)
            assert np.array_equal(np.concatenate(traces[s_]).view(np.uint8), want_tr[s_].view(np.uint8)), (kept, s_)
        live.close()
    want_t, want_tr = plain.receive_many(x)
    for kept in (False, True):
        live = e.live(ns)
        live.keep_previous(kept)
        sizes = [3, 1, 8, 2, 1, 16, 5, 1, 1, 40, 7, 9, 2, 44]
        ring = [torch.zeros((ns, 44 * N), dtype=torch.float32, device=dev) for _ in range(2)]
        texts, traces, b0 = [""] * ns, [[] for _ in range(ns)], 0
        for k, nb in enumerate(sizes):
            buf = ring[k % 2]
            buf[:, :nb * N].copy_(xd[:, b0 * N:(b0 + nb) * N])
            t, tr = live.next(buf[:, :nb * N])
            for s_ in range(ns):
                texts[s_] += t[s_]
                traces[s_].append(tr[s_])
            b0 += nb
        for s_ in range(ns):
            assert texts[s_] == want_t[s_], (kept, s_)
            assert np.array_equal(np.concatenate(traces[s_]).view(np.uint8), want_tr[s_].view(np.uint8)), (kept, s_)
        live.close()
    e.close()
    plain.close()


def test_plain_c_host_runs_live_microphones(tmp_path):
    """tests/c/host_live.c (C99 -pedantic -Werror, libuchirp.so only): three synthetic microphones, one new block each per
    call of uc_receive_streams_next -- the firmware's own loop -- print the characters as they complete; every stream
    receives its own message."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_live")
    libdir = os.path.join(root, "ultrasonic-communication_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "c", "host_live.c"), "-o", exe, "-L" + libdir, "-luchirp", "-lm",
                           "-Wl,-rpath," + libdir])
    for mode in ([], ["pdm"]):        # DFSDM words; the microphones' 1-bit streams (UC_DTYPE_PDM: the DFSDM on the device)
        out = subprocess.run([exe, "4"] + mode, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        for s, m in enumerate(("Hello World!", "uchirp", "MI355X", "0123456789")):
            assert ('stream %d received "%s"' % (s, m)) in out.stdout
        print(mode, out.stdout.strip().splitlines()[-4:])
