#!/usr/bin/env python3
"""One kernel of the library on its own workload, as a profiling target and as the clock probe of every kernel.

  python tools/run_target.py <target> [--frames-log2 19] [--iters 3]            launches only (rocprofv3 passes)
  python tools/run_target.py <target> --clock [--seconds 2.5] [--zeros]         the clock-stamped twin inside libuchirp.so (uc_clock_probe):
        >= `seconds` of back-to-back launches, then the in-kernel clock d(s_memtime) / d(s_memrealtime) x 100 MHz of
        the last launch (median over waves), loop cycles, start / end skew of the grid
  --info prints {"target", "kernel", "units", "unit", "alg_bytes_per_unit"} of the launch and exits (no GPU work)

Targets: band_rx_real_f32 band_rx_real_i32 band_sync_cplx_f32 band_dechirp_down_f32 compress_f32 iq2048_fw_f32
iq2048_bb_f32 iq1024_fw_f32 iq1024_bb_f32 stream_d8_f32 sinc5 sinc5_streams rows_rx_real_f32 rows_sync_cplx_f32
(sinc5_streams: uc_dfsdm_sinc5_streams, one 2048-word block of every microphone per launch; rows_*: the live receivers' step,
uc_receive_streams_next with one new block of every stream per launch = 8 new FIFO offsets per stream: the band kernel's ROWS
build + the replay kernel)
rows_*_f32: EVERY offset evaluated (UC_RX_NEED_FORCE = 0x1ff: what a stream in a tracking state costs at most);
rows_*_f32_idle: noise streams with the idle masks (3 or 5 of the 8, UP only) -- the unit is still the offset INDEX;
rows_*_f32_idle_keep: the same with uc_rx_state_keep_previous (no block is copied into the state);
spectrum_rx_real_f32: uc_window_spectrum (the pipeline() boundary: the SPEC build, statistics + the window bins stored).
(keys of profiles/r*_valu_insts.json; the I/Q targets run on the pass-band stream of BASELINE configs[2]).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("target")
ap.add_argument("--frames-log2", type=int, default=19)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--clock", action="store_true")
ap.add_argument("--seconds", type=float, default=2.5)
ap.add_argument("--zeros", action="store_true")
ap.add_argument("--info", action="store_true")
args = ap.parse_args()
sys.path[:0] = [ROOT, os.path.join(ROOT, "ultrasonic-communication_amd")]

T = args.target
nf = 1 << args.frames_log2                     # 2048-sample frames' worth of samples
BB = dict(fs=100000.0, carrier=18000.0, f0=16500.0, f1=19500.0)
SPEC = {  # target -> (kernel-name substring, units, unit, algorithmic bytes per unit)
    "band_rx_real_f32": ("band_kernel", nf, "frame", 8193),
    "band_rx_real_i32": ("band_kernel", nf, "frame", 8193),
    "band_sync_cplx_f32": ("band_kernel", nf, "frame", 8193),
    "band_dechirp_down_f32": ("band_kernel", nf, "frame", 8192 + 32),
    "compress_f32": ("compress_kernel", nf, "frame", 8192 + 32),
    "iq2048_fw_f32": ("iq_kernel", nf, "frame", 8192 + 32),
    "iq2048_bb_f32": ("iq_kernel", nf, "frame", 8192 + 1),
    "iq1024_fw_f32": ("iq1024_kernel", 2 * nf, "frame", 4096 + 32),
    "iq1024_bb_f32": ("iq1024_kernel", 2 * nf, "frame", 4096 + 1),
    "stream_d8_f32": ("stream_kernel", nf * 2048, "sample", 4.5),
    "sinc5": ("sinc5_kernel", nf * 512, "word", 8.0),
    "sinc5_streams": ("sinc5_kernel", nf * 512, "word", 8.0),
    # a new block of a live stream: 8192 B of the block + 8192 B of the block in front of it read once (the 8 frames overlap in
    # cache), 8192 B kept for the next call, 8 x 8 B of records -- per NEW FIFO OFFSET (frame): 3080 B
    "rows_rx_real_f32": ("band_kernel", nf, "frame", 3080),
    "rows_sync_cplx_f32": ("band_kernel", nf, "frame", 3080),
    "rows_rx_real_f32_idle": ("band_kernel", nf, "frame", 3080),
    "rows_sync_cplx_f32_idle": ("band_kernel", nf, "frame", 3080),
    # ... with uc_rx_state_keep_previous: nothing kept, 2 x 8192 B read + 64 B of records per stream -- 2056 B per offset index
    "rows_rx_real_f32_idle_keep": ("band_kernel", nf, "frame", 2056),
    "rows_sync_cplx_f32_idle_keep": ("band_kernel", nf, "frame", 2056),
    # uc_window_spectrum: the frame in, 2 histories x (2 x 156 + 1) window bins out
    "spectrum_rx_real_f32": ("band_kernel", nf, "frame", 8192 + 2 * 313 * 4),
}
kname, units, unit, alg = SPEC[T]
if T in ("rows_rx_real_f32", "rows_sync_cplx_f32"):   # (pricing hook of the library, read at uc_create)
    os.environ["UC_TUNING"] = "1"
    os.environ["UC_RX_NEED_FORCE"] = "0x1ff"
ROWS = T.startswith("rows_")
if ROWS:
    # a live state's FIRST step runs on the power-on need word (0x052) whatever the target asks for, and idle streams alternate
    # between 3 and 5 offsets: the counters skip the first dispatch and average over whole pairs behind it
    args.iters = 5
if args.info:
    print(json.dumps({"target": T, "kernel": kname, "units": units, "unit": unit, "alg_bytes_per_unit": alg,
                      "skip_first_dispatches": 1 if ROWS else 0}))
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import uchirp  # noqa: E402
from uchirp import synth  # noqa: E402

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev)

if T.startswith("iq"):
    n = 2048 if T.startswith("iq2048") else 1024
    x, _ = synth.device_iq_stream(units, n, dev, seed=1)
    bb = "_bb_" in T
    cfg = dict(BB, n=n, time_frame=n / BB["fs"], flags=uchirp.FLAG_IQ_BASEBAND) if bb else dict(n=n)
    e = uchirp.Engine(uchirp.IQ, mag_mean=1000.0, **cfg)
    sym = torch.empty(units, dtype=torch.uint8, device=dev) if bb else None
    st = None if bb else torch.empty((units, e.spf, 8), dtype=torch.float32, device=dev)
    if args.zeros:
        x.zero_()

    def launch():
        e.process(x, n_frames=units, want_symbols=bb, want_stats=not bb, symbols_out=sym, stats_out=st, stream=stream.cuda_stream)
elif T == "sinc5_streams":
    import ctypes as C
    e = uchirp.Engine(uchirp.RX_REAL)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    ns = units // 2048
    pdm = torch.randint(-2 ** 31, 2 ** 31 - 1, (ns, 2048), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    out = torch.empty((ns, 2048), dtype=torch.int32, device=dev)
    hist = torch.full((ns, 4), -1431655766, dtype=torch.int32, device=dev)

    def launch():
        e.dfsdm_streams(pdm, hist, out=out, stream=stream.cuda_stream)
elif T.startswith("rows_"):
    e = uchirp.Engine(uchirp.RX_REAL if "rx_real" in T else uchirp.SYNC_CPLX)
    ns = nf // 8
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    chunk = torch.randn((ns, 2048), generator=g, device=dev) * 50.0
    if args.zeros:
        chunk.zero_()
    live = e.live(ns)
    keep = T.endswith("_keep")
    chunks = [chunk, chunk.clone()] if keep else [chunk]     # (keep_previous: a ring of two buffers, used in turn)
    if keep:
        live.keep_previous(True)
    text = torch.zeros((ns, 16), dtype=torch.uint8, device=dev)
    ntext = torch.zeros(ns, dtype=torch.int32, device=dev)
    turn = [0]

    def launch():
        live.next_into(chunks[turn[0] % len(chunks)], text, ntext, stream=stream.cuda_stream)
        turn[0] += 1
elif T == "sinc5":
    e = uchirp.Engine(uchirp.RX_REAL)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    pdm = torch.randint(-2 ** 31, 2 ** 31 - 1, (units + 4,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    out = torch.empty(units, dtype=torch.int32, device=dev)

    def launch():
        e.dfsdm(pdm, out=out, stream=stream.cuda_stream)
else:
    frames, _ = synth.device_frames(nf, dev, seed=1)
    if args.zeros:
        frames.zero_()
    if T == "band_rx_real_i32":
        frames = (frames.round().to(torch.int64) * 256).to(torch.int32)
    if T == "spectrum_rx_real_f32":
        e = uchirp.Engine(uchirp.RX_REAL, mag_mean=1000.0)
        import ctypes as C
        wb = uchirp.lib().uc_window_bins(e._h)
        spec = torch.empty((nf, e.spf, wb), dtype=torch.float32, device=dev)

        def launch():
            rc = uchirp.lib().uc_window_spectrum(e._h, C.c_void_p(frames.data_ptr()), uchirp.DTYPE_F32, nf, 0,
                                                 C.c_void_p(spec.data_ptr()), C.c_void_p(stream.cuda_stream))
            assert rc == 0, uchirp.lib().uc_last_error()
    elif T == "stream_d8_f32":
        e = uchirp.Engine(uchirp.STREAM)
        xs = frames.reshape(-1)
        _, n_out, n_blocks, _ = e.stream_geometry(xs.numel())
        comp = torch.empty(n_out, dtype=torch.float32, device=dev)
        pk = torch.empty((n_blocks, 2), dtype=torch.int32, device=dev)

        def launch():
            e.process_stream(xs, compressed_out=comp, peaks_out=pk, stream=stream.cuda_stream)
    else:
        var = {"band_rx_real_f32": uchirp.RX_REAL, "band_rx_real_i32": uchirp.RX_REAL, "band_sync_cplx_f32": uchirp.SYNC_CPLX,
               "band_dechirp_down_f32": uchirp.DECHIRP_DOWN, "compress_f32": uchirp.COMPRESS}[T]
        e = uchirp.Engine(var, mag_mean=1000.0)
        want_sym = var in (uchirp.RX_REAL, uchirp.SYNC_CPLX)
        sym = torch.empty(nf, dtype=torch.uint8, device=dev) if want_sym else None
        st = None if want_sym else torch.empty((nf, e.spf, 8), dtype=torch.float32, device=dev)

        def launch():
            e.process(frames, want_symbols=want_sym, want_stats=not want_sym, symbols_out=sym, stats_out=st,
                      stream=stream.cuda_stream)

if args.clock:
    e.clock_probe(True)
if not args.clock:
    for _ in range(args.iters):
        launch()
    torch.cuda.synchronize()
    print("done", T, units, args.iters)
    sys.exit(0)

launch()
torch.cuda.synchronize()
t0 = time.perf_counter()
n_l = 0
while time.perf_counter() - t0 < args.seconds:
    for _ in range(50):
        launch()
    torch.cuda.synchronize()
    n_l += 50
wall = (time.perf_counter() - t0) / n_l
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(stream)
launch()
if ROWS:
    launch()            # (idle streams: a 3-offset and a 5-offset step; the time below is the mean of the pair)
b.record(stream)
torch.cuda.synchronize()
raw = e.clock_stamps().astype(np.int64)
raw = raw[raw[:, 1] > 0]
cu_key = (raw[:, 0] >> 40) & 0xfff            # XCC_ID << 8 | HW_ID[15:8]: the CU that ran the wave
d = raw.astype(np.float64)
d[:, 0] = (raw[:, 0] & ((1 << 40) - 1)).astype(np.float64)
t_first, t_last = d[:, 2].min(), d[:, 3].max()
# per CU: when its LAST wave ended (the CU is out of work from then on) and when its first one started
cus = np.unique(cu_key)
cu_last = np.array([d[cu_key == c, 3].max() for c in cus])
cu_idle_us = (t_last - cu_last) / 100.0        # idle tail of each CU inside the grid span
start_us, end_us, life_us = (d[:, 2] - t_first) / 100.0, (d[:, 3] - t_first) / 100.0, d[:, 1] / 100.0


def pct(v):
    return [float(np.percentile(v, q)) for q in (0, 10, 50, 90, 100)]


clk = d[:, 0] / d[:, 1] * 100.0  # MHz: s_memrealtime ticks at 100 MHz
print(json.dumps({
    "target": T, "kernel": kname, "data": "zero" if args.zeros else "random", "units": units, "unit": unit,
    "launches_before_stamp": n_l, "seconds_of_back_to_back_launches": args.seconds, "ms_per_launch_wall": wall * 1e3,
    "ms_last_launch_events": a.elapsed_time(b) / (2.0 if ROWS else 1.0), "waves_stamped": int(d.shape[0]),
    "shader_clock_MHz_median": float(np.median(clk)), "shader_clock_MHz_p10": float(np.percentile(clk, 10)),
    "shader_clock_MHz_p90": float(np.percentile(clk, 90)), "loop_cycles_median": float(np.median(d[:, 0])),
    "loop_us_median": float(np.median(d[:, 1]) / 100.0), "units_per_s": units / wall,
    "grid_span_us_first_start_to_last_end": float((t_last - t_first) / 100.0),
    "wave_start_us_p0_10_50_90_100": pct(start_us), "wave_end_us_p0_10_50_90_100": pct(end_us),
    "wave_life_us_p0_10_50_90_100": pct(life_us),
    "cus_seen": int(cus.size), "cu_idle_tail_us_p0_10_50_90_100": pct(cu_idle_us), "cu_idle_tail_us_mean": float(cu_idle_us.mean()),
    "wave_idle_tail_us_mean": float(((t_last - d[:, 3]) / 100.0).mean())}))
