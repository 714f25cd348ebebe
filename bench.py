#!/usr/bin/env python3
"""bench.py -- chirp frames/s of the receiver's per-frame DSP on MI355X.

One "step" = one pass of the hot path (uc_process_batch, variant RX_REAL:
ingest -> window*chirp -> 2048-pt FFT -> windowed peak pick -> up/down symbol)
over one batch of synthetic frames already resident in HBM.

  N = 1   BASELINE.json configs[1]: 1 Mi x 2048-sample fp32 frames, random orthogonal up/down
          chirps at -10 dB SNR, literal TIME_FRAME (the firmware's reference tables).
  N > 1   BASELINE.json configs[4]: the frame index space is block-partitioned (weak scaling:
          1 Mi frames per GPU, no data-path collective); the frames are the K7 wire format
          (G, 7 x H, L, 96 data bits of "Hello World!", 12 x G) repeated over the GLOBAL frame
          index at -10 dB, reference sweep matched to the frame; the decoded symbol stream
          (1 B/frame) is all-gathered over RCCL every step inside the timed region (the gather of
          step k overlaps the kernel of step k + 1) and decoded to text after it.

`python bench.py --gpus N` starts its N ranks itself (one process per GPU, before anything touches
the GPU); under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is
one of the ranks.  --gpus must equal the world size.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ultrasonic-communication_amd"))

N = 2048
BYTES_PER_FRAME = 8192 + 1          # fp32 frame in + 1 symbol byte out
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec
MATCHED_TIME_FRAME = N / 78125.0    # one symbol = one frame (generator: T = 0.0262 s)
MSG = "Hello World!"


def make_device_frames(n_frames, device, seed, snr_db=-10.0, amp=1000.0):
    """configs[1] frames (kept under this name for the tools)."""
    from uchirp import synth
    return synth.device_frames(n_frames, device, seed, snr_db=snr_db, amp=amp)


def host_cpu_share():
    """CPUs this process may actually use: the smaller of the affinity mask and the cgroup CPU
    quota (a one-GPU box exposes all 256 hardware threads but schedules 16 CPUs' worth of time:
    256 OpenMP threads there run 4x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())          # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(frames_host, mag_mean):
    """The CPU restatement (oracle, float32 butterflies like CMSIS-DSP) timed on the
    host cores of this box on a bounded sample of the same frames."""
    from oracle import uco
    o = uco.Oracle(uco.RX_REAL, mag_mean=mag_mean)
    cores = host_cpu_share()
    n = frames_host.shape[0]
    o.process(frames_host[:8192], precision=uco.F32, threads=cores)  # warm the thread pool
    passes, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < 10.0 and passes < 64:  # >= 10 s of CPU work, bounded
        o.process(frames_host, precision=uco.F32, threads=cores)
        passes += 1
        dt = time.perf_counter() - t0
    n = n * passes
    # one-thread figure on a smaller sample, and what the host is
    n1 = min(frames_host.shape[0], 1 << 15)
    t1 = time.perf_counter()
    o.process(frames_host[:n1], precision=uco.F32, threads=1)
    one_thread = n1 / (time.perf_counter() - t1)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    rs, _ = o.process(frames_host[:4096], precision=uco.F64, threads=cores)
    # SURVEY.md section 8d's optional NumPy line: the same decision with numpy.fft.rfft (pocketfft), one process
    up, down, hann = o.table(uco.TABLE_UP), o.table(uco.TABLE_DOWN), o.table(uco.TABLE_HANN)
    bw2 = o.bandwidth2
    xs = frames_host[: 1 << 14]
    t2 = time.perf_counter()
    def window_max(ref):  # both windows of dsp(): bins [0, bw2) and the mirror of [1, bw2]; mag[0] is the packed pair (Q2)
        X = np.fft.rfft((xs * ref) * hann, axis=1)
        m = np.abs(X[:, : bw2 + 1])
        m[:, 0] = np.hypot(X[:, 0].real, X[:, -1].real)
        return m.max(axis=1)
    mu, md = window_max(up), window_max(down)
    sym_np = np.where((np.maximum(mu, md) - mag_mean) / mag_mean >= 2.0, (md <= mu).astype(np.uint8), 255)
    numpy_rate = xs.shape[0] / (time.perf_counter() - t2)
    numpy_agree = float((sym_np[:4096] == rs).mean())
    return {"symbols_f64_oracle_head": rs, "value": n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "one_thread_value": one_thread, "cpu_model": model,
            "numpy_rfft_value": numpy_rate, "numpy_rfft_agrees_with_oracle": numpy_agree,
            "sample": "%d passes over the first %d frames of the same batch, oracle/uc_oracle.c "
                      "(float32 butterflies), OpenMP %d threads = this box's CPU share (%d hardware threads visible), "
                      "%.1f s" % (passes, n // passes, cores, os.cpu_count() or 1, dt)}


def achievable_hbm(frames, stream, torch):
    """What the simplest kernels get out of this chip's HBM right now (tools/hbm_probe.hip): a read-only
    non-temporal stream over the bench's own 8 GiB batch (the band kernel's traffic shape: 8192 B in, 1 B out)
    and a 1:1 copy of 4 GiB.  HIP events on the launch stream, median of 10."""
    import ctypes as C
    path = os.path.join(ROOT, "tools", "libhbm_probe.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.hbm_probe_read.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]
    L.hbm_probe_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    nbytes = frames.numel() * 4
    blocks = torch.cuda.get_device_properties(frames.device).multi_processor_count * 8
    sink = torch.zeros(blocks, dtype=torch.int32, device=frames.device)
    half = (nbytes // 2) & ~((1 << 15) - 1)
    dst = torch.empty(half // 4, dtype=torch.float32, device=frames.device)

    def timed(fn):
        ts = []
        for _ in range(13):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            rc = fn()
            b.record(stream)
            torch.cuda.synchronize()
            if rc != 0:
                return None
            ts.append(a.elapsed_time(b))
        return float(sorted(ts[3:])[len(ts[3:]) // 2])

    t_read = timed(lambda: L.hbm_probe_read(frames.data_ptr(), nbytes, sink.data_ptr(), blocks, stream.cuda_stream))
    t_copy = timed(lambda: L.hbm_probe_copy(frames.data_ptr(), dst.data_ptr(), half, blocks, stream.cuda_stream))
    if not t_read or not t_copy:
        return None
    return {"read_stream_GBs": nbytes / t_read / 1e6, "copy_GBs": 2 * half / t_copy / 1e6,
            "method": "tools/hbm_probe.hip: read-only nt stream over the %d MiB batch; 1:1 copy of %d MiB "
                      "(read + write bytes); HIP events, median of 10" % (nbytes >> 20, half >> 20)}


def stream_measurement(args, eng, frames, rank, torch):
    """BASELINE config 4 (side measurement): the batch read as ONE continuous stream through UC_STREAM."""
    x = frames.reshape(-1)
    halo, n_out, n_blocks, hop = eng.stream_geometry(x.numel())
    comp = torch.empty(n_out, dtype=torch.float32, device=x.device)
    pk = torch.empty((n_blocks, 2), dtype=torch.int32, device=x.device)
    t_r = time.perf_counter()
    while (time.perf_counter() - t_r) * 1e3 < args.ramp_ms:      # clock ramp: see main()
        eng.process_stream(x, compressed_out=comp, peaks_out=pk)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        eng.process_stream(x, compressed_out=comp, peaks_out=pk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.process_stream(x, compressed_out=comp, peaks_out=pk)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    byts = x.numel() * 4 + n_out * 4 + n_blocks * 8
    if rank == 0:
        print(json.dumps({"metric": "input samples/s (stream: FIR decimate + overlap-save compression, side measurement)",
                          "value": x.numel() / dt, "unit": "samples/s", "n_gpus": 1, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": dt * 1e3, "decim": int(eng.cfg.decim),
                          "blocks": n_blocks, "blocks_per_s": n_blocks / dt,
                          "roofline": {"bound": "hbm", "achieved": byts / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": byts / dt / 1e9 / HBM_PEAK_GBS,
                                       "bytes_per_sample": byts / x.numel()}}), flush=True)


def side_measurement(args, eng, frames, world, rank, torch):
    """Not the contract line: frames/s of one of the sibling variants on the same synthetic batch."""
    if args.variant == "stream":
        return stream_measurement(args, eng, frames, rank, torch)
    n = eng.n
    per_frame = {"sync_cplx": 8193, "compress": 8192 + 32, "dechirp_down": 8192 + 32, "iq": 8192 + 104 + 32,
                 "iq1024": 4096 + 104 + 32}[args.variant]
    nfr = (frames.numel() - eng.halo - n) // n + 1
    want_sym = args.variant == "sync_cplx"
    stats = None if want_sym else torch.empty((nfr, eng.spf, 8), dtype=torch.float32, device=frames.device)
    sym = torch.empty(nfr, dtype=torch.uint8, device=frames.device) if want_sym else None
    t_r = time.perf_counter()
    while (time.perf_counter() - t_r) * 1e3 < args.ramp_ms:      # clock ramp: see main()
        eng.process(frames, n_frames=nfr, want_symbols=want_sym, want_stats=not want_sym, symbols_out=sym, stats_out=stats)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        eng.process(frames, n_frames=nfr, want_symbols=want_sym, want_stats=not want_sym, symbols_out=sym, stats_out=stats)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.process(frames, n_frames=nfr, want_symbols=want_sym, want_stats=not want_sym, symbols_out=sym, stats_out=stats)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        v = nfr * args.steps / dt
        print(json.dumps({"metric": "chirp frames/s (%s, side measurement)" % args.variant, "value": v, "unit": "frames/s",
                          "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "frame_len": n, "frames": nfr,
                          "roofline": {"bound": "hbm", "achieved": v * per_frame / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": v * per_frame / 1e9 / HBM_PEAK_GBS, "bytes_per_frame": per_frame}}), flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=1 << 20, help="frames per GPU per step")
    ap.add_argument("--snr", type=float, default=-10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ramp-ms", type=float, default=150.0,
                    help="untimed kernel launches before the warm-up steps until the GPU's clocks have settled (0 = none)")
    ap.add_argument("--variant", default="rx_real",
                    choices=["rx_real", "sync_cplx", "compress", "dechirp_down", "iq", "iq1024", "stream"],
                    help="default rx_real = BASELINE configs[1]; the others are side measurements")
    return ap.parse_args(argv)


def launch_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as fresh child processes.
    This parent never imports torch and never touches the GPU (a process that has initialised the GPU
    must not be replaced or forked on this pool); it relays rank 0's JSON line and the worst exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile
    out0 = tempfile.TemporaryFile()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else sys.stderr))
    deadline = time.time() + float(os.environ.get("UC_BENCH_TIMEOUT", "900"))
    rc = 0
    while rc == 0 and any(p.poll() is None for p in procs):
        time.sleep(0.2)
        rc = next((p.returncode for p in procs if p.poll() not in (None, 0)), 0)
        if time.time() > deadline:
            rc = -1
    if rc:
        for p in procs:                      # a rank died or hung: the others wait in a collective; end exactly those
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    out0.seek(0)
    text = out0.read().decode()
    if rc:
        sys.stderr.write(text)
        raise SystemExit("bench.py: a rank failed (exit %s)" % rc)
    # stdout carries the JSON line(s) only; anything a library printed on rank 0's stdout goes to stderr
    for ln in text.splitlines():
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return launch_ranks(args)
    world = int(env_world or "1")
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d. Run `python bench.py --gpus N` (it starts its own ranks) "
                         "or `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`."
                         % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    # UC_BENCH_REHEARSE=1: plumbing rehearsal of the N > 1 path (spawn, rendezvous, symbol gather, concatenation
    # check, text decode, JSON line) with gloo.  On a ONE-GPU box every rank runs the real kernel on device 0; on a
    # box without a GPU NO kernel runs (the "decoded" symbols are the transmitted ones) and `value` is null.
    # Never a measurement.
    rehearse = os.environ.get("UC_BENCH_REHEARSE") == "1"
    have_gpu = torch.cuda.is_available()
    if not have_gpu and not rehearse:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if rehearse:
        local_rank = 0
    device = torch.device("cuda", local_rank) if have_gpu else torch.device("cpu")
    if have_gpu:
        torch.cuda.set_device(local_rank)
    dist = None
    # UC_BENCH_HELLO=1: run the N > 1 leg -- configs[4] framing, RCCL process group, async all-gather, digest check,
    # text decode -- with whatever world size there is, 1 included (what a one-GPU box can exercise of it on RCCL).
    multi = world > 1 or os.environ.get("UC_BENCH_HELLO") == "1"
    json_out = sys.stdout
    if multi:
        # RCCL prints its version banner on file descriptor 1; stdout must carry the JSON line and nothing else:
        # keep a private copy of the real stdout for that line and point descriptor 1 at stderr for everyone else
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        world = dist.get_world_size()          # n_gpus printed below is what the process group reports

    from uchirp import synth
    mag_mean = 1000.0
    nf = args.frames
    hello = multi and args.variant == "rx_real"
    eng = None
    if have_gpu:
        import uchirp
        vmap = {"rx_real": (uchirp.RX_REAL, {}), "sync_cplx": (uchirp.SYNC_CPLX, {}), "compress": (uchirp.COMPRESS, {}),
                "dechirp_down": (uchirp.DECHIRP_DOWN, {}), "iq": (uchirp.IQ, {}), "iq1024": (uchirp.IQ, {"n": 1024}),
                "stream": (uchirp.STREAM, {})}
        vid, vkw = vmap[args.variant]
        if hello:
            vkw = dict(vkw, time_frame=MATCHED_TIME_FRAME)
        eng = uchirp.Engine(vid, device=local_rank, mag_mean=mag_mean, **vkw)
    if hello:
        # rank r owns frames [r nf, (r + 1) nf) of the global stream (uchirp/shard.py::partition, equal shares)
        frames, sent = synth.device_hello_frames(rank * nf, nf, device, seed=1234 + rank, snr_db=args.snr, msg=MSG)
    else:
        frames, sent = synth.device_frames(nf, device, seed=1234 + rank, snr_db=args.snr)
    if args.variant != "rx_real":
        return side_measurement(args, eng, frames, world, rank, torch)
    # Two symbol buffers: the gather of step k (RCCL's own stream) overlaps the kernel of step k + 1;
    # a buffer is rewritten only after the gather that read it has finished (work.wait() orders the
    # launch stream behind it without blocking the host).
    sym2 = [torch.empty(nf, dtype=torch.uint8, device=device) for _ in range(2)]
    gathered2 = [torch.empty(world * nf, dtype=torch.uint8, device=device) for _ in range(2)] if multi else None
    works = [None, None]
    stream = torch.cuda.current_stream(device) if have_gpu else None

    def gather(b):
        if rehearse:  # gloo has no device all-gather: stage through the host (rehearsal only)
            host = [torch.empty(nf, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(host, sym2[b].cpu())
            gathered2[b].copy_(torch.cat(host))
            return None
        return dist.all_gather_into_tensor(gathered2[b], sym2[b], async_op=True)

    def step(k, e0=None, e1=None):
        b = k & 1
        if works[b] is not None:
            works[b].wait()
            works[b] = None
        if eng is None:                       # rehearsal without a GPU: no kernel, the transmitted symbols
            sym2[b].copy_(torch.where(sent == 2, torch.full_like(sent, 0xFF), sent))
        else:
            if e0 is not None:
                e0.record(stream)
            eng.process(frames, want_stats=False, symbols_out=sym2[b], stream=stream.cuda_stream)
            if e1 is not None:
                e1.record(stream)
        if multi:
            works[b] = gather(b)

    def drain():
        for b in range(2):
            if works[b] is not None:
                works[b].wait()
                works[b] = None

    def sync():
        if have_gpu:
            torch.cuda.synchronize()

    # Clock ramp (untimed, BEFORE the W warm-up steps; --ramp-ms, default 150): an idle MI355X needs ~20 launches
    # (~40 ms) of this kernel before its launch time settles -- 2.9, 3.0, 2.7, 2.5 ms for the first four launches,
    # 2.22 ms for launches 5-14, 2.10 ms from launch 20 on (tools/ramp_probe.py, profiles/r02_v5_ramp.txt).  The ramp
    # runs the kernel only (no gather), on the same resident batch, and is reported as `ramp_ms` in the JSON line.
    ramp_launches = 0
    if have_gpu and args.ramp_ms > 0:
        t_r = time.perf_counter()
        while (time.perf_counter() - t_r) * 1e3 < args.ramp_ms:
            for _ in range(4):
                eng.process(frames, want_stats=False, symbols_out=sym2[0], stream=stream.cuda_stream)
            torch.cuda.synchronize()
            ramp_launches += 4
    for k in range(args.warmup):
        step(k)
    drain()
    sync()
    if multi:
        dist.barrier()
    sync()

    # per-launch kernel time: HIP events on the stream the kernel is launched on
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if have_gpu else (None, None)
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, ev[k][0], ev[k][1])
    drain()
    sync()
    if multi:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    symbols = sym2[(args.steps - 1) & 1]
    gathered_host = None
    if multi:
        # Every rank holds the concatenation of all ranks' symbols, rank order: its own slice equals what it
        # decoded, and every rank's gathered buffer has the same digest (so the other slices are the owners').
        import hashlib
        g = gathered2[(args.steps - 1) & 1]
        assert torch.equal(g[rank * nf:(rank + 1) * nf], symbols), "gathered symbol stream differs from this rank's symbols"
        gathered_host = g.cpu().numpy()
        dig = torch.frombuffer(bytearray(hashlib.sha256(gathered_host.tobytes()).digest()), dtype=torch.uint8).clone()
        digs = [torch.empty(32, dtype=torch.uint8) for _ in range(world)]
        if rehearse:
            dist.all_gather(digs, dig)
        else:
            dd = [torch.empty(32, dtype=torch.uint8, device=device) for _ in range(world)]
            dist.all_gather(dd, dig.to(device))
            digs = [d.cpu() for d in dd]
        assert all(torch.equal(d, digs[0]) for d in digs), "ranks hold different gathered symbol streams"
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if have_gpu else None

    if rank == 0:
        total_frames = world * nf * args.steps
        value = total_frames / elapsed
        out = {
            "metric": "chirp frames/s (2048-pt FFT demod)", "value": value if have_gpu else None, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "ramp_ms": args.ramp_ms if have_gpu else 0.0, "ramp_launches": ramp_launches,
        }
        if hello:
            out["config"] = {"workload": "configs[4]: %d x 2048-sample fp32 frames per GPU, frame-sharded 'Hello World!' "
                                         "stream (K7 framing: G, 7 H, L, 96 data bits, 12 G, repeated over the global frame "
                                         "index), SNR %.0f dB, rx_real with the reference sweep matched to the frame"
                                         % (nf, args.snr),
                             "frames_per_gpu": nf, "frame_len": N, "variant": "rx_real", "time_frame": MATCHED_TIME_FRAME,
                             "parallelism": "frame-sharded x%d, RCCL all-gather of the symbol stream (1 B/frame) every step"
                                            % world}
        else:
            out["config"] = {"workload": "configs[1]: %d x 2048-sample fp32 frames per GPU, orthogonal up/down-chirp "
                                         "symbol decision (rx_real), SNR %.0f dB" % (nf, args.snr),
                             "frames_per_gpu": nf, "frame_len": N, "variant": "rx_real",
                             "parallelism": "frame-sharded x%d, RCCL all-gather of symbols" % world}
        if have_gpu:
            achieved = nf * BYTES_PER_FRAME / (kern_ms * 1e-3) / 1e9
            # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
            # collected with rocprofv3 --pmc in their own runs: profiles/*_hbm_traffic.json), scaled to
            # this launch's frame count; null if no profile is committed.
            traffic = None
            try:
                import glob
                import re
                # the band kernel's own passes are named rNN_vM_hbm_traffic.json (other kernels carry their name)
                tf = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json"))
                            if re.fullmatch(r"r\d+_v\d+_hbm_traffic\.json", os.path.basename(f)))[-1]
                traffic = json.load(open(tf))["hbm_bytes_per_frame"] * nf
            except Exception:
                traffic = None
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "kernel": "band_kernel<rx_real,f32>", "kernel_ms": kern_ms,
                               "bytes_per_frame": BYTES_PER_FRAME}
            if not multi:
                ach = achievable_hbm(frames, stream, torch)
                if ach:
                    out["roofline"]["achievable"] = ach
                    out["roofline"]["frac_of_achievable_read"] = achieved / ach["read_stream_GBs"]
        if hello:
            # config 5 gate: the gathered stream decodes to the transmitted text
            texts = synth.decode_hello(gathered_host, len(MSG))
            good = sum(1 for t in texts if t == MSG)
            out["decoded_text_first"] = texts[0] if texts else ""
            out["transmissions"] = len(texts)
            out["transmissions_decoded_exactly"] = good
            data = synth.hello_kind_stream(0, world * nf, MSG)
            m = data != 2
            out["bit_error_rate_vs_transmitted"] = float((gathered_host[m] != data[m]).mean())
            if not have_gpu:
                out["rehearsal"] = "plumbing only: no GPU, no kernel ran, value is null"
            elif rehearse:
                out["rehearsal"] = "every rank on device 0, gloo: not a measurement"
        else:
            # correctness gate on the measured run: decoded symbols vs transmitted bits
            out["bit_error_rate_vs_transmitted"] = float((symbols != sent).float().mean().item())
        if not multi and have_gpu and not args.no_cpu_baseline:
            cb = cpu_baseline(frames[: 1 << 19].cpu().numpy(), mag_mean)
            # the oracle as the checker: GPU symbols of the measured run vs float64 oracle
            head = cb.pop("symbols_f64_oracle_head")
            out["symbols_equal_oracle_head4096"] = float((symbols[:4096].cpu().numpy() == head).mean())
            out["cpu_baseline"] = cb
        print(json.dumps(out), file=json_out, flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
