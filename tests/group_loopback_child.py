"""Child process of tests/test_gpu_group_loopback.py: the group logic of uc_group.cpp at world 2 .. 8 on ONE GPU, with the
loop-back stand-in for RCCL (tests/stubs/loopback_rccl.cpp; UC_TUNING=1 UC_RCCL_LIB=... UC_GROUP_SHARE_DEVICES=1 set by the
parent).  Every "rank" lives on device 0; each decodes ITS shard of the batch into its slice of its own gathered buffer, the
loop-back library copies the slices between the ranks' buffers.  Checked against ONE plain context over the whole batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))
import numpy as np
import torch
import uchirp
from uchirp import synth
sys.path.insert(0, os.path.join(ROOT, "tests"))

dev = torch.device("cuda", 0)
N = 2048
checks = 0


def run(variant, kw, flat, n_frames, stride, halo, world, steps=7, dtype=uchirp.DTYPE_F32):
    """flat: the whole sample buffer on the device (halo samples in front of frame 0)."""
    global checks
    n = kw.get("n", N)
    eng = uchirp.Engine(variant, **kw)
    want, _ = eng.process(flat, n_frames=n_frames, stride=stride, want_stats=False)
    torch.cuda.synchronize()
    g = uchirp.Group(variant, devices=[0] * world, **kw)
    assert (g.world, g.n_local, g.first_rank) == (world, world, 0)
    ptrs, keep = [], []
    for r in range(world):
        first, count = uchirp.partition(n_frames, world, r)
        e0, ne = uchirp.frame_span(n, stride, halo, first, count)
        # the rank holds ITS span only: a private copy, so that a read outside it would hit unrelated memory
        mine = flat[e0:e0 + ne].clone() if count else flat[:1].clone()
        keep.append(mine)
        ptrs.append(mine.data_ptr() + 4 * halo)
    bufs = [[torch.full((n_frames,), 0x77, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(3)]
    # the shard copies and the fills above ran on torch's stream; the group launches on non-blocking streams of its own,
    # which do not wait for it: everything must have landed before the first step reads a shard or writes a slice
    torch.cuda.synchronize()
    for k in range(steps):
        g.process(ptrs, n_frames, bufs[k % 3], stride=stride, dtype=dtype)
    g.synchronize()
    for k in range(3):
        for r in range(world):
            assert torch.equal(bufs[k][r], want), (variant, world, n_frames, k, r)
            checks += 1
    # one buffer per rank reused every step: the write-after-gather guard serialises, results unchanged
    one = [torch.zeros(n_frames, dtype=torch.uint8, device=dev) for _ in range(world)]
    for k in range(4):
        g.process(ptrs, n_frames, one, stride=stride, dtype=dtype)
    g.synchronize()
    assert all(torch.equal(o, want) for o in one)
    g.close()
    eng.close()
    return want


# configs[4]: the 'Hello World!' stream, sharded; even and ragged shares (the grouped-broadcast path), more ranks than frames
for world in (2, 3, 8):
    for nf in (117 * 16, 117 * 16 + 5, 5):
        frames, _ = synth.device_hello_frames(0, nf, dev, seed=nf + world, snr_db=-10.0)
        sym = run(uchirp.RX_REAL, dict(mag_mean=1000.0, time_frame=N / 78125.0), frames.reshape(-1), nf, 0, 0, world)
        if nf >= 117:
            texts = synth.decode_hello(sym.cpu().numpy(), 12)
            assert texts and all(t == "Hello World!" for t in texts)
# ---- a call refused for its arguments has enqueued NOTHING: one local device of several gets a NULL shard / a NULL buffer --
# the call returns < 0 before any stream is touched and before any collective starts, the group stays usable (the next step
# gives the right stream) and uc_group_destroy returns without waiting for anybody
frames, _ = synth.device_hello_frames(0, 117 * 8, dev, seed=5, snr_db=-10.0)
nf = 117 * 8
eng = uchirp.Engine(uchirp.RX_REAL, mag_mean=1000.0, time_frame=N / 78125.0)
want, _ = eng.process(frames.reshape(-1), n_frames=nf, want_stats=False)
eng.close()
for world in (2, 3):
    g = uchirp.Group(uchirp.RX_REAL, devices=[0] * world, mag_mean=1000.0, time_frame=N / 78125.0)
    shards = []
    for r in range(world):
        first, count = uchirp.partition(nf, world, r)
        shards.append(frames.reshape(-1)[first * N:(first + count) * N].clone())
    outs = [torch.zeros(nf, dtype=torch.uint8, device=dev) for _ in range(world)]
    torch.cuda.synchronize()
    for bad in range(world):
        for what in ("frames", "gathered", "dtype"):
            p_ = [x.data_ptr() for x in shards]
            o_ = list(outs)
            kw = {}
            if what == "frames":
                p_[bad] = 0
            elif what == "gathered":
                o_[bad] = 0
            else:
                kw["dtype"] = 7
            try:
                g.process(p_, nf, o_, **kw)
                raise SystemExit("a NULL %s on local device %d of %d was accepted" % (what, bad, world))
            except uchirp.UchirpError as ex:
                assert "NULL" in str(ex) or "dtype" in str(ex), ex
        g.process([x.data_ptr() for x in shards], nf, outs)          # the group is as it was
        g.synchronize()
        assert all(torch.equal(o, want) for o in outs), (world, bad)
        checks += 1
    # a refused receive: one state missing
    try:
        g.receive_streams([shards[0]] * world, world, N, [torch.zeros((world, 8), dtype=torch.uint8, device=dev)] * world, 8,
                          states=[None] * world)
        raise SystemExit("NULL states accepted")
    except (uchirp.UchirpError, AttributeError, TypeError):
        pass
    g.process([x.data_ptr() for x in shards], nf, outs)
    g.synchronize()
    assert all(torch.equal(o, want) for o in outs)
    # refusals of uc_receive_streams_next ITSELF that only the LAST local device's arguments earn (they used to come
    # back after the kernels of the devices before it were enqueued): partial blocks, overlapping streams, a state of another
    # context, a misaligned UC_DTYPE_PDM buffer.  Nothing may have run: the text buffers keep their fill, and the states are
    # still at power-on (the live run behind it decodes from block 0)
    ns_l = 2 * world
    sts = [g.rx_state(l, 2) for l in range(world)]
    alien = g.rx_state(0, 2)                                       # two streams, but of local device 0's context
    xs = [torch.randn((2, N + 8), device=dev) * 50.0 for _ in range(world)]
    fill = [torch.full((ns_l, 8), 0x33, dtype=torch.uint8, device=dev) for _ in range(world)]
    last = world - 1
    for what in ("partial", "overlap", "alien", "pdm"):
        kw = dict(n_samples=N, stride=N + 8, dtype=uchirp.DTYPE_F32)
        st_, x_ = list(sts), [x.data_ptr() for x in xs]
        if what == "partial":
            kw["n_samples"] = N + 4
        elif what == "overlap":
            kw["stride"] = N - 4
        elif what == "alien":
            st_[last] = alien
        else:
            kw["dtype"] = uchirp.DTYPE_PDM
            x_[last] += 4                                          # 4 bytes off a 16-byte boundary
        try:
            g.receive_streams(x_, ns_l, kw["n_samples"], fill, 8, stride=kw["stride"], dtype=kw["dtype"], states=st_)
            raise SystemExit("uc_group_receive_streams_next accepted a %s call" % what)
        except uchirp.UchirpError:
            pass
        g.synchronize()
        assert all(bool((f == 0x33).all()) for f in fill), (world, what)
        checks += 1
    alien.close()
    for s_ in sts:
        s_.close()
    g.process([x.data_ptr() for x in shards], nf, outs)              # ... and the group still works
    g.synchronize()
    assert all(torch.equal(o, want) for o in outs)
    g.close()                                                        # returns: nothing is waiting for a peer
    checks += 1

# overlapping FIFO reads (stride 256): neighbouring shards overlap by n - 256 samples
flat = synth.device_frames(40, dev, seed=9, snr_db=-3.0)[0].reshape(-1)
nfr = (flat.numel() - N) // 256 + 1
for world in (2, 5):
    run(uchirp.SYNC_CPLX, dict(mag_mean=1000.0), flat, nfr, 256, 0, world)
# base-band I/Q: 26 samples of FIR history in front of every shard
BB = dict(n=1024, fs=100000.0, carrier=18000.0, f0=16500.0, f1=19500.0, time_frame=1024 / 1e5, flags=uchirp.FLAG_IQ_BASEBAND,
          mag_mean=1000.0)
x, _ = synth.device_iq_stream(999, 1024, dev, seed=4, snr_db=-5.0)
for world in (2, 4):
    run(uchirp.IQ, BB, x, 999, 0, 26, world)
# int32 DFSDM words
fi = (synth.device_frames(333, dev, seed=2, snr_db=0.0)[0].round().to(torch.int64) * 256).to(torch.int32).reshape(-1)
run(uchirp.RX_REAL, dict(mag_mean=256000.0, time_frame=N / 78125.0), fi, 333, 0, 0, 3, dtype=uchirp.DTYPE_I32)

# ---- uc_group_receive_streams: the streams of a node, block-partitioned over the ranks ("replicas across streams") ----------
from test_gpu_receive_many import _transmissions


def run_receive(variant, world, ns, blocks=150, cap=24):
    global checks
    x, busy, msgs = _transmissions(ns, seed=1000 + world + ns, blocks=blocks)
    eng = uchirp.Engine(variant)
    want, _ = eng.receive_many(x, busy=busy, text_cap=cap, want_trace=False)
    eng.close()
    g = uchirp.Group(variant, devices=[0] * world)
    shares = [uchirp.partition(ns, world, r) for r in range(world)]
    # every rank holds ITS streams only
    xs = [torch.from_numpy(np.ascontiguousarray(x[f:f + c])).to(dev) if c else torch.zeros(1, device=dev) for f, c in shares]
    bs = [torch.from_numpy(np.ascontiguousarray(busy[f:f + c])).to(dev) if c else None for f, c in shares]
    text = [torch.full((ns, cap), 0x33, dtype=torch.uint8, device=dev) for _ in range(world)]
    cnt = [torch.full((ns,), -1, dtype=torch.int32, device=dev) for _ in range(world)]
    for _ in range(2):                                      # twice into the same buffers: the hazard guard
        g.receive_streams(xs, ns, blocks * N, text, cap, n_text=cnt, busy=bs)
    g.synchronize()
    for r in range(world):
        t, c = text[r].cpu().numpy(), cnt[r].cpu().numpy()
        got = [bytes(t[i, :c[i]]).decode("latin-1") for i in range(ns)]
        assert got == want, (variant, world, ns, r)
        checks += 1
    # host buffers on every rank (staged through the group, complete when the call returns)
    th = [np.zeros((ns, cap), np.uint8) for _ in range(world)]
    ch = [np.zeros(ns, np.uint32) for _ in range(world)]
    xh = [np.ascontiguousarray(x[f:f + c]) if c else np.zeros(1, np.float32) for f, c in shares]
    bh = [np.ascontiguousarray(busy[f:f + c]) if c else None for f, c in shares]
    g.receive_streams(xh, ns, blocks * N, th, cap, n_text=ch, busy=bh)
    for r in range(world):
        assert [bytes(th[r][i, :ch[r][i]]).decode("latin-1") for i in range(ns)] == want, (variant, world, ns, r, "host")
        checks += 1
    # live, in chunks of unequal sizes: every rank's share keeps its receivers between the calls
    # (a rank that owns no stream -- more ranks than streams -- brings no state: NULL)
    states = [g.rx_state(r, c) if c else None for r, (f, c) in enumerate(shares)]
    acc = [""] * ns
    at = 0
    for nb in (3, 1, 40, blocks - 44):
        ch = [torch.from_numpy(np.ascontiguousarray(x[f:f + c, at * N:(at + nb) * N])).to(dev) for f, c in shares]
        bz = [torch.from_numpy(np.ascontiguousarray(busy[f:f + c, at:at + nb])).to(dev) for f, c in shares]
        g.receive_streams(ch, ns, nb * N, text, cap, n_text=cnt, busy=bz, states=states)
        g.synchronize()
        t, c = text[world - 1].cpu().numpy(), cnt[world - 1].cpu().numpy()
        acc = [a + bytes(t[i, :c[i]]).decode("latin-1") for i, a in enumerate(acc)]
        for r in range(world - 1):
            assert torch.equal(cnt[r], cnt[world - 1])
        at += nb
    assert acc == want, (variant, world, ns)
    checks += 1
    for s in states:
        if s is not None:
            g.rx_state_destroy(s)
    # the same live run with every rank's state under uc_rx_state_keep_previous (round 6): the chunks of a call stay alive and
    # unchanged until the next call has completed (held for two calls here), busy-masked calls mixed in
    states = [g.rx_state(r, c) if c else None for r, (f, c) in enumerate(shares)]
    for st_ in states:
        if st_ is not None:
            uchirp._check(uchirp.lib().uc_rx_state_keep_previous(st_._h, 1), "uc_rx_state_keep_previous")
    acc, at, held = [""] * ns, 0, []
    for k, nb in enumerate((3, 1, 40, 1, 1, blocks - 46)):
        ch = [torch.from_numpy(np.ascontiguousarray(x[f:f + c, at * N:(at + nb) * N])).to(dev) for f, c in shares]
        held.append(ch)
        del held[:-2]
        bz = [torch.from_numpy(np.ascontiguousarray(busy[f:f + c, at:at + nb])).to(dev) for f, c in shares] if k % 2 else None
        g.receive_streams(ch, ns, nb * N, text, cap, n_text=cnt, busy=bz, states=states)
        g.synchronize()
        t, c = text[world - 1].cpu().numpy(), cnt[world - 1].cpu().numpy()
        acc = [a + bytes(t[i, :c[i]]).decode("latin-1") for i, a in enumerate(acc)]
        at += nb
    # (the calls without a mask accepted every block: compare with the recorded call under the mask they add up to)
    mask2 = busy.copy()
    at = 0
    for k, nb in enumerate((3, 1, 40, 1, 1, blocks - 46)):
        if k % 2 == 0:
            mask2[:, at:at + nb] = 0
        at += nb
    eng = uchirp.Engine(variant)
    want2, _ = eng.receive_many(x, busy=mask2, text_cap=cap, want_trace=False)
    eng.close()
    assert acc == want2, (variant, world, ns, "kept chunks")
    checks += 1
    for s in states:
        if s is not None:
            g.rx_state_destroy(s)
    g.close()
    return sum(m in t for m, t in zip(msgs, want))



# ---- uc_group_process_stream: the overlap-save blocks of ONE stream over the ranks, peak records gathered -------------------
def run_stream(world, n_dec, dtype_i32=False):
    global checks
    eng = uchirp.Engine(uchirp.STREAM)
    halo = eng.stream_geometry(0)[0]
    x = (np.random.default_rng(world + n_dec).standard_normal(halo + 8 * n_dec) * 1000).astype(np.float32)
    if dtype_i32:
        x = (np.round(x).astype(np.int64) * 256).astype(np.int32)
    _, n_out, n_blocks, hop = eng.stream_geometry(x.size)
    want, want_pk = eng.process_stream(x)
    eng.close()
    g = uchirp.Group(uchirp.STREAM, devices=[0] * world)
    spans = [g.stream_span(x.size, r) for r in range(world)]
    # every rank holds ITS shard only (16-byte aligned copies)
    xs = [torch.from_numpy(np.ascontiguousarray(x[s0:s0 + ns])).to(dev) if ns else torch.zeros(4, device=dev) for s0, ns, q0, nq in spans]
    comp = [torch.zeros(max(nq, 1), dtype=torch.float32, device=dev) for s0, ns, q0, nq in spans]
    pk = [torch.full((n_blocks, 2), -1, dtype=torch.int32, device=dev) for _ in range(world)]
    for _ in range(3):
        g.process_stream(xs, x.size, pk, compressed=comp, dtype=uchirp.DTYPE_I32 if dtype_i32 else uchirp.DTYPE_F32)
    g.synchronize()
    got = np.zeros_like(want)
    for r, (s0, ns, q0, nq) in enumerate(spans):
        got[q0:q0 + nq] = comp[r].cpu().numpy()[:nq]
        assert np.array_equal(uchirp.peaks_from_tensor(pk[r]).view(np.uint8), want_pk.view(np.uint8)), (world, n_dec, r)
        checks += 1
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (world, n_dec)
    # host buffers on every rank
    ph = [np.zeros(n_blocks, uchirp.PEAK_DTYPE) for _ in range(world)]
    xh = [np.ascontiguousarray(x[s0:s0 + ns]) if ns else np.zeros(4, x.dtype) for s0, ns, q0, nq in spans]
    g.process_stream(xh, x.size, ph, dtype=uchirp.DTYPE_I32 if dtype_i32 else uchirp.DTYPE_F32)
    for r in range(world):
        assert np.array_equal(ph[r].view(np.uint8), want_pk.view(np.uint8)), (world, n_dec, r, "host")
        checks += 1
    g.close()


run_stream(2, 60000)
run_stream(3, 61000)            # ragged: the blocks do not divide by three
run_stream(8, 9000)             # fewer blocks than ranks
run_stream(4, 40000, dtype_i32=True)

decoded = run_receive(uchirp.RX_REAL, 2, 24) + run_receive(uchirp.SYNC_CPLX, 3, 25) + run_receive(uchirp.SYNC_CPLX, 4, 24)
run_receive(uchirp.RX_REAL, 8, 5)                           # more ranks than streams
assert decoded >= 25

# ---- random draws (tools/fuzz_round.sh: UC_LOOPBACK_FUZZ="<cases> <seed>"; the pytest run makes none) ---------------------
fz = os.environ.get("UC_LOOPBACK_FUZZ", "").split()
if fz:
    rng = np.random.default_rng(int(fz[1]))
    for case in range(int(fz[0])):
        world = int(rng.integers(1, 9))
        kind = int(rng.integers(0, 3))
        if kind == 0:      # frames: any count (fewer than ranks included), contiguous or overlapping FIFO reads, either reference
            nf = int(rng.choice([1, 3, world - 1 or 1, world + 1, 117, 1000, 4099]))
            variant = [uchirp.RX_REAL, uchirp.SYNC_CPLX][int(rng.integers(0, 2))]   # (not DECHIRP_DOWN: its frame PAIRS are
            # formed inside a shard, and a frame's round-off depends on its partner)
            stride = int(rng.choice([0, 256, 512]))
            fr = synth.device_frames(nf, dev, seed=int(rng.integers(1 << 30)), snr_db=float(rng.choice([-10.0, 0.0])))[0].reshape(-1)
            nfr = nf if stride == 0 else (fr.numel() - N) // stride + 1
            run(variant, dict(mag_mean=1000.0), fr, nfr, stride, 0, world, steps=int(rng.integers(3, 12)))       # (>= 3: the check reads all three rotating buffers)
        elif kind == 1:    # whole microphone streams
            run_receive([uchirp.RX_REAL, uchirp.SYNC_CPLX][int(rng.integers(0, 2))], world, int(rng.integers(1, 20)))
        else:              # the blocks of one UC_STREAM stream
            run_stream(world, int(rng.integers(1, 40000)), dtype_i32=bool(rng.integers(0, 2)))
    print("loopback fuzz: %d random cases, 0 failures" % int(fz[0]))
print("loopback ok: %d gathered buffers checked" % checks)
