"""The multi-GPU leg behind the C-ABI (include/uchirp.h: uc_group_*) on the one GPU this box has: a group of ONE device is
the whole code path of a rank -- partition, decode into the rank's slice of the gathered stream, RCCL all-gather in place
on the side stream behind an event, the write-after-gather guard, uc_group_wait_gather -- with a communicator of one.
(RCCL refuses two ranks on one device, so world > 1 runs only on the driver's multi-GPU node; the partition / halo
arithmetic of world > 1 is covered on the CPU, tests/test_group_cpu.py.)
tests/c/host_multi.c is the plain-C host of BASELINE configs[4] over this API."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from uchirp import shard, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 2048
MATCHED = N / 78125.0
MSG = "Hello World!"


@pytest.fixture(scope="module")
def uchirp():
    import uchirp as m
    m.lib()
    return m


def _hello(n_frames, seed=5):
    import torch
    frames, kinds = synth.device_hello_frames(0, n_frames, torch.device("cuda", 0), seed=seed, snr_db=-10.0, msg=MSG)
    return frames, kinds


@pytest.mark.parametrize("mode", ["devices", "rank"])
def test_group_of_one_device_equals_the_engine(uchirp, mode):
    """Both ways to build a group (one process drives the devices; one process per rank with a unique id), 9 steps over 3
    rotating buffers AND over one single buffer (the write-after-gather guard serialises those): the gathered stream is the
    engine's symbol stream, bit for bit, and decodes."""
    import torch
    nf = 117 * 60
    frames, _ = _hello(nf)
    eng = uchirp.Engine(uchirp.RX_REAL, mag_mean=1000.0, time_frame=MATCHED)
    want, _ = eng.process(frames, want_stats=False)
    torch.cuda.synchronize()
    if mode == "devices":
        g = uchirp.Group(uchirp.RX_REAL, devices=[0], mag_mean=1000.0, time_frame=MATCHED)
    else:
        g = uchirp.Group(uchirp.RX_REAL, world=1, rank=0, unique_id=uchirp.Group.unique_id(), device=0, mag_mean=1000.0,
                         time_frame=MATCHED)
    assert (g.world, g.n_local, g.first_rank) == (1, 1, 0)
    bufs = [torch.zeros(nf, dtype=torch.uint8, device="cuda:0") for _ in range(3)]
    for k in range(9):
        g.process([frames], nf, [bufs[k % 3]])
    g.synchronize()
    for b in bufs:
        assert torch.equal(b, want)
    one = torch.zeros(nf, dtype=torch.uint8, device="cuda:0")
    s = torch.cuda.Stream()
    for k in range(5):
        g.process([frames], nf, [one], streams=[s.cuda_stream])       # a caller's stream
    # a consumer stream waits for the gather on the device, not on the host
    s2 = torch.cuda.Stream()
    g.wait_gather(0, one, s2.cuda_stream)
    with torch.cuda.stream(s2):
        copy = one.clone()
    s2.synchronize()
    assert torch.equal(copy, want)
    texts = synth.decode_hello(one.cpu().numpy(), len(MSG))
    assert len(texts) == 60 and all(t == MSG for t in texts)
    with pytest.raises(uchirp.UchirpError):
        g.wait_gather(0, bufs[0][1:], s2.cuda_stream)                   # never gathered into
    g.close()
    eng.close()


def test_group_host_pointers_and_int32_words(uchirp):
    """Host buffers go through the staging path and come back complete when the call returns; int32 DFSDM words."""
    frames, bits = synth.make_frames(500, seed=3, snr_db=0.0, dtype=np.int32, sweep_time=MATCHED)
    eng = uchirp.Engine(uchirp.RX_REAL, mag_mean=256000.0, time_frame=MATCHED)
    want, _ = eng.process(frames)
    g = uchirp.Group(uchirp.RX_REAL, devices=[0], mag_mean=256000.0, time_frame=MATCHED)
    out = np.zeros(500, np.uint8)
    g.process([frames], 500, [out], dtype=uchirp.DTYPE_I32)
    assert np.array_equal(out, want) and np.array_equal(out, bits)
    # overlapping FIFO reads (stride 256) through the group
    flat = frames.reshape(-1)
    nfr = (flat.size - N) // 256 + 1
    want2, _ = eng.process(flat, stride=256)
    out2 = np.zeros(nfr, np.uint8)
    g.process([flat], nfr, [out2], stride=256, dtype=uchirp.DTYPE_I32)
    assert np.array_equal(out2, want2)
    g.close()
    eng.close()


def test_group_error_paths(uchirp):
    with pytest.raises(uchirp.UchirpError):
        uchirp.Group(uchirp.RX_REAL, devices=[0, 0])
    gs = uchirp.Group(uchirp.STREAM, devices=[0])                      # a UC_STREAM group has no frames ...
    with pytest.raises(uchirp.UchirpError):
        gs.process([np.zeros(4096, np.float32)], 2, [np.zeros(2, np.uint8)])
    gs.close()
    g0 = uchirp.Group(uchirp.RX_REAL, devices=[0])                     # ... and a frame group no overlap-save blocks
    with pytest.raises(uchirp.UchirpError):
        g0.process_stream([np.zeros(40000, np.float32)], 40000, [np.zeros((8, 2), np.uint32)])
    g0.close()
    with pytest.raises(uchirp.UchirpError):
        uchirp.Group(uchirp.RX_REAL, devices=[7])                      # no such device on this box
    with pytest.raises(ValueError):
        uchirp.Group(uchirp.RX_REAL, world=1, rank=0, unique_id=b"short")
    g = uchirp.Group(uchirp.RX_REAL, devices=[0])
    with pytest.raises(uchirp.UchirpError):
        g.process([None], 10, [np.zeros(10, np.uint8)])
    g.close()


def test_stream_span_equals_shard_py(uchirp):
    """uc_stream_span (whole overlap-save blocks per rank, halo shared read-only) == uchirp/shard.py::stream_span, and
    sharded UC_STREAM calls reproduce the one-GPU outputs bit for bit."""
    import ctypes as C
    eng = uchirp.Engine(uchirp.STREAM)
    rng = np.random.default_rng(1)
    halo = eng.stream_geometry(0)[0]
    # (halo + a multiple of the decimation: nothing lies behind the last output, in the whole stream or in a shard -- the
    # ragged last block's transform also sees whatever follows it in the buffer, which moves its round-off)
    x = rng.standard_normal(halo + 8 * 37000).astype(np.float32) * 1000
    halo, n_out, n_blocks, hop = eng.stream_geometry(x.size)
    assert n_out == 37000 and n_blocks > 8
    whole, _ = eng.process_stream(x)
    for world in (1, 2, 3, 8):
        got = np.zeros_like(whole)
        for r in range(world):
            v = [C.c_size_t() for _ in range(4)]
            rc = uchirp.lib().uc_stream_span(eng._h, x.size, world, r, *[C.byref(q) for q in v])
            assert rc == 0
            s0, ns, q0, nq = [q.value for q in v]
            p0, p1, pq0, pq1 = shard.stream_span(x.size, world, r, halo, hop, int(eng.cfg.decim))
            assert (s0, s0 + ns, q0, q0 + nq) == (p0, p1, pq0, pq1) or (ns == 0 and p1 == p0)
            if ns:
                part, _ = eng.process_stream(x[s0:s0 + ns])
                got[q0:q0 + nq] = part[:nq]
        assert np.array_equal(got.view(np.uint32), whole.view(np.uint32)), world
    eng.close()


def _build_host_multi(tmp_path):
    exe = str(tmp_path / "host_multi")
    libdir = os.path.join(ROOT, "ultrasonic-communication_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "host_multi.c"), "-o", exe, "-L" + libdir, "-luchirp", "-lm",
                           "-Wl,-rpath," + libdir])
    return exe


def test_plain_c_host_drives_the_group_and_decodes_hello_world(uchirp, tmp_path):
    """tests/c/host_multi.c (C99, no HIP header): configs[4] on a group of one device.  (i) its own generated frames decode
    to the text; (ii) given the SAME frames as the Python path (a file), it prints the sha256 digest of the symbol stream
    that uc_process_batch through ctypes produces."""
    exe = _build_host_multi(tmp_path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([exe, "-f", str(117 * 40), "-k", "8"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "transmissions decoded exactly: 40 of 40" in out.stdout and '"Hello World!"' in out.stdout
    assert "group: world 1, 1 local device(s), first rank 0" in out.stdout
    print(out.stdout)
    import torch
    nf = 117 * 70
    frames, _ = _hello(nf, seed=9)
    path = str(tmp_path / "frames.f32")
    frames.cpu().numpy().tofile(path)
    eng = uchirp.Engine(uchirp.RX_REAL, mag_mean=1000.0, time_frame=MATCHED)
    sym, _ = eng.process(frames, want_stats=False)
    torch.cuda.synchronize()
    digest = hashlib.sha256(sym.cpu().numpy().tobytes()).hexdigest()
    out = subprocess.run([exe, "-f", str(nf), "-i", path], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert ("sha256 of the gathered symbol stream: " + digest) in out.stdout, out.stdout
    assert "transmissions decoded exactly: 70 of 70" in out.stdout
    eng.close()


def _texts(text, ntext):
    text, ntext = np.asarray(text), np.asarray(ntext)
    assert all(text[i, ntext[i]] == 0 for i in range(len(ntext)))           # NUL-terminated
    return [bytes(text[i, :ntext[i]]).decode("latin-1") for i in range(len(ntext))]


@pytest.mark.parametrize("mode", ["devices", "rank"])
def test_group_receive_streams_of_one_device_equals_the_engine(uchirp, mode):
    """uc_group_receive_streams / _next at world size 1 (the whole code path of a rank: the share's receivers write into the
    rank's slice of the gathered texts and counts, two in-place gathers behind an event): the texts uc_receive_streams gives,
    from device and from host buffers, with dropped blocks, and LIVE in chunks of unequal sizes."""
    import torch
    from test_gpu_receive_many import _transmissions
    x, busy, msgs = _transmissions(48, seed=77, blocks=150)
    cap = 32
    for variant in (uchirp.RX_REAL, uchirp.SYNC_CPLX):
        eng = uchirp.Engine(variant)
        want, _ = eng.receive_many(x, busy=busy, text_cap=cap, want_trace=False)
        want_nb, _ = eng.receive_many(x, text_cap=cap, want_trace=False)
        assert sum(m in t for m, t in zip(msgs, want_nb)) >= 30
        if mode == "devices":
            g = uchirp.Group(variant, devices=[0])
        else:
            g = uchirp.Group(variant, world=1, rank=0, unique_id=uchirp.Group.unique_id(), device=0)
        ns, nsmp = x.shape
        xd = torch.from_numpy(x).to("cuda:0")
        bd = torch.from_numpy(busy).to("cuda:0")
        # device buffers, asynchronous; three text buffers in rotation, then the same one again (the hazard guard)
        tb = [torch.full((ns, cap), 0x55, dtype=torch.uint8, device="cuda:0") for _ in range(3)]
        nt = [torch.full((ns,), 9999, dtype=torch.int32, device="cuda:0") for _ in range(3)]
        for k in range(5):
            g.receive_streams([xd], ns, nsmp, [tb[k % 3]], cap, n_text=[nt[k % 3]], busy=[bd] if k % 2 == 0 else None)
        g.synchronize()
        assert _texts(tb[0].cpu().numpy(), nt[0].cpu().numpy()) == want_nb        # k = 3: no mask
        assert _texts(tb[1].cpu().numpy(), nt[1].cpu().numpy()) == want           # k = 4: masked
        assert _texts(tb[2].cpu().numpy(), nt[2].cpu().numpy()) == want           # k = 2
        # host buffers: complete when the call returns
        th, nh = np.zeros((ns, cap), np.uint8), np.zeros(ns, np.uint32)
        g.receive_streams([x], ns, nsmp, [th], cap, n_text=[nh], busy=[busy])
        assert _texts(th, nh) == want
        # without counts
        th2 = np.zeros((ns, cap), np.uint8)
        g.receive_streams([x], ns, nsmp, [th2], cap)
        assert [bytes(r).split(b"\0")[0].decode("latin-1") for r in th2] == want_nb
        # live: chunks of 1, 7, 30, 112 blocks add up to the recorded call
        st = g.rx_state(0, ns)
        acc = [""] * ns
        at = 0
        for nb in (1, 7, 30, 112):
            chunk = np.ascontiguousarray(x[:, at * N:(at + nb) * N])
            bz = np.ascontiguousarray(busy[:, at:at + nb])
            g.receive_streams([chunk], ns, nb * N, [th], cap, n_text=[nh], busy=[bz], states=[st])
            acc = [a + t for a, t in zip(acc, _texts(th, nh))]
            at += nb
        assert acc == want
        with pytest.raises(uchirp.UchirpError):                          # a state of the wrong size
            g.receive_streams([x[:5]], 5, nsmp, [th], cap, states=[st])
        g.rx_state_destroy(st)
        g.close()
        eng.close()
    with pytest.raises(uchirp.UchirpError):                              # no state machine in that variant
        gg = uchirp.Group(uchirp.COMPRESS, devices=[0])
        try:
            gg.receive_streams([x], x.shape[0], x.shape[1], [np.zeros((x.shape[0], cap), np.uint8)], cap)
        finally:
            gg.close()


def test_plain_c_host_serves_the_microphones_of_a_node(tmp_path):
    """tests/c/host_node_live.c (C99 -pedantic -Werror, libuchirp.so only): live microphones block-partitioned over the
    devices of a group (one here), one new block each per call of uc_group_receive_streams_next, the characters of every
    stream gathered into every device's arrays; each stream receives its own message."""
    exe = str(tmp_path / "host_node_live")
    libdir = os.path.join(ROOT, "ultrasonic-communication_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "host_node_live.c"), "-o", exe, "-L" + libdir, "-luchirp", "-lm",
                           "-Wl,-rpath," + libdir])
    for mode in ([], ["pdm"]):        # DFSDM words; the microphones' 1-bit streams (UC_DTYPE_PDM through the group)
        out = subprocess.run([exe, "6", "1"] + mode, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        for s, m in enumerate(("Hello World!", "uchirp", "MI355X", "0123456789", "Hello World!", "uchirp")):
            assert ('stream %d received "%s"' % (s, m)) in out.stdout
        print(mode, out.stdout.strip().splitlines()[-3:])


def test_group_process_stream_of_one_device_equals_the_engine(uchirp):
    """uc_group_process_stream at world size 1: the compressed envelope and the gathered peak records are the engine's, bit for
    bit, from device buffers (asynchronous, rotating and reused peak buffers) and from host buffers."""
    import torch
    eng = uchirp.Engine(uchirp.STREAM)
    halo = eng.stream_geometry(0)[0]
    x = (np.random.default_rng(3).standard_normal(halo + 8 * 50000) * 1000).astype(np.float32)
    _, n_out, n_blocks, hop = eng.stream_geometry(x.size)
    want, want_pk = eng.process_stream(x)
    g = uchirp.Group(uchirp.STREAM, devices=[0])
    assert g.stream_span(x.size, 0) == (0, x.size, 0, n_out)
    xd = torch.from_numpy(x).to("cuda:0")
    comp = torch.zeros(n_out, dtype=torch.float32, device="cuda:0")
    pk = [torch.zeros((n_blocks, 2), dtype=torch.int32, device="cuda:0") for _ in range(2)]
    for k in range(5):
        g.process_stream([xd], x.size, [pk[k % 2]], compressed=[comp])
    g.synchronize()
    assert np.array_equal(comp.cpu().numpy().view(np.uint32), want.view(np.uint32))
    for b in pk:
        assert np.array_equal(uchirp.peaks_from_tensor(b).view(np.uint8), want_pk.view(np.uint8))
    ph = np.zeros(n_blocks, uchirp.PEAK_DTYPE)
    ch = np.zeros(n_out, np.float32)
    g.process_stream([x], x.size, [ph], compressed=[ch])
    assert np.array_equal(ph.view(np.uint8), want_pk.view(np.uint8)) and np.array_equal(ch.view(np.uint32), want.view(np.uint32))
    g.process_stream([x], x.size, [ph])                                # peaks only
    assert np.array_equal(ph.view(np.uint8), want_pk.view(np.uint8))
    g.close()
    eng.close()
