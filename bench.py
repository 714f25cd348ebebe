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

Beside the contract line's own fields the N = 1 line carries (after the timed region, which they do not touch):
  configs        BASELINE configs[2] (base-band I/Q, n = 1024, on the fs = 100 kHz pass-band stream; the firmware-window
                 mode beside it) and configs[3] (UC_STREAM, D = 8, replayed from a captured hipGraph; the eager rate
                 beside it), each with ms_per_step, kernel time by HIP events and both rooflines
  hello_world1   the N > 1 leg (configs[4]: K7 framing, RCCL all-gather every step, decode) at world size 1: the
                 like-for-like anchor of the driver's 1 -> N scaling efficiency
  roofline.valu  VALU issue fraction from the committed counter passes (profiles/r*_valu_insts.json); `bound` names the
                 larger of the HBM and the VALU fraction
Exit status: 3 if a correctness gate fails (symbols differ from the oracle on clear frames, a transmission does not
decode, the gathered stream is inconsistent) -- the JSON line is still printed.

Never run `--gpus N > 1` under rocprofv3: the profiler initialises the GPU in this parent, which then starts N children.
Profile one rank instead (RANK=0 WORLD_SIZE=1 ... rocprofv3 ... -- python3 bench.py), see tools/profile_round.sh.
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
# Symbol / gather buffers in rotation.  THREE, not two: the band kernel is persistent and fills every CU, so RCCL's gather
# kernel of step k only gets CUs when the kernel of step k + 1 drains; with two buffers the kernel of step k + 2 (which
# rewrites the buffer that gather reads) had to wait for it with the chip idle (20 us per step at world size 1, measured);
# with three it is the kernel of step k + 3 that depends on gather k, a whole kernel time later.
NBUF = 3


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
    rs, rst = o.process(frames_host[:4096], precision=uco.F64, threads=cores)
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
    return {"symbols_f64_oracle_head": rs, "clear_head": clear_frames(rst), "value": n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "one_thread_value": one_thread, "cpu_model": model,
            "numpy_rfft_value": numpy_rate, "numpy_rfft_agrees_with_oracle": numpy_agree,
            "sample": "%d passes over the first %d frames of the same batch, oracle/uc_oracle.c "
                      "(float32 butterflies), OpenMP %d threads = this box's CPU share (%d hardware threads visible), "
                      "%.1f s" % (passes, n // passes, cores, os.cpu_count() or 1, dt)}


def clear_frames(stats, thr=2.0, margin=1e-3):
    """Frames whose float64-oracle decision is not a near-tie (tests/parity_util.py: clear_symbols)."""
    su, sd = stats["snr"][:, 0].astype(np.float64), stats["snr"][:, 1].astype(np.float64)
    m = np.abs(su - sd) / np.maximum(np.maximum(np.abs(su), np.abs(sd)), 1e-30)
    near = (np.abs(su - thr) < 1e-3 * thr) | (np.abs(sd - thr) < 1e-3 * thr)
    return (m >= margin) & ~near


def clock_ramp(launch, torch, ramp_ms):
    """Untimed launches until `ramp_ms` have passed (an idle MI355X needs ~20 launches / ~40 ms of these kernels
    before its launch time settles: tools/ramp_probe.py, profiles/r02_v5_ramp.txt).  Returns the launch count."""
    n = 0
    if ramp_ms <= 0:
        return 0
    t_r = time.perf_counter()
    while (time.perf_counter() - t_r) * 1e3 < ramp_ms:
        for _ in range(4):
            launch()
        torch.cuda.synchronize()
        n += 4
    return n


def timed_launches(launch, stream, torch, steps, warmup):
    """W untimed + K timed launches: (wall ms per step, mean HIP-event ms per launch on `stream`)."""
    for _ in range(warmup):
        launch()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        launch()
        b.record(stream)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    return wall, float(np.mean([a.elapsed_time(b) for a, b in ev]))


_VALU = None


def valu_table():
    """profiles/r*_valu_insts.json (the newest): VALU wave-instructions per unit of work and the in-kernel shader clock
    of every kernel, from rocprofv3 --pmc passes and the clock-stamp build (tools/pmc_round.sh, tools/valu_table.py)."""
    global _VALU
    if _VALU is None:
        import glob
        fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_insts.json")))
        _VALU = (json.load(open(fs[-1])), os.path.relpath(fs[-1], ROOT)) if fs else ({}, None)
    return _VALU


def live_traffic(frames_log2=19, timeout_s=150):
    """HBM bytes per frame of the headline kernel from the PMC counters, collected IN THIS RUN: two child processes,
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md
    prescribes: they do not fit one pass) over `python3 tools/run_target.py band_rx_real_f32` -- the same kernel on 2^19
    frames of the same synthetic workload -- with the guide's gfx950 correction (FETCH_SIZE tallies a 128-byte request as 64
    bytes: x 2; WRITE_SIZE exact; both reported in KiB).  Returns None when rocprofv3 is not there or a pass fails."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    # (this process is itself being profiled -- tools/profile_round.sh: no profiler inside a profiler)
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return None
    raw = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="uc_pmc_", dir="/tmp")
        try:
            subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable,
                            os.path.join(ROOT, "tools", "run_target.py"), "band_rx_real_f32", "--frames-log2", str(frames_log2),
                            "--iters", "3"], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=timeout_s,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            vals = [float(r["Counter_Value"]) for f in files for r in csv.DictReader(open(f))
                    if "band_kernel" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
            if not vals:
                return None
            raw[ctr] = sum(vals) / len(vals)
        except (subprocess.SubprocessError, OSError, KeyError, ValueError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    nfr = 1 << frames_log2
    rd, wr = raw["FETCH_SIZE"] * 1024.0 * 2.0, raw["WRITE_SIZE"] * 1024.0
    return {"hbm_bytes_per_frame": (rd + wr) / nfr, "read_bytes_per_frame": rd / nfr, "write_bytes_per_frame": wr / nfr,
            "FETCH_SIZE_KiB_raw": raw["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": raw["WRITE_SIZE"], "frames_per_profiled_launch": nfr,
            "method": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two child processes, separate "
                      "passes) over tools/run_target.py band_rx_real_f32 --frames-log2 %d; gfx950 correction FETCH_SIZE x 2"
                      % frames_log2}


class PowerSampler:
    """Socket power and SMU-reported shader clock of the GPU this process drives, read from the amdgpu hwmon files
    (power1_input in microwatts, freq1_input in Hz, power1_cap) every millisecond by a thread while the timed region runs.
    Evidence for WHY the clock under these kernels is what uc_clock_read measures: the socket sits at its power cap."""

    def __init__(self, torch, device):
        import glob
        import threading
        self.dir, self.samples, self._stop, self._thread = None, [], threading.Event(), None
        try:
            bus = torch.cuda.get_device_properties(device).pci_bus_id
            if isinstance(bus, int):                      # (older torch: domain / bus / device numbers)
                p = torch.cuda.get_device_properties(device)
                bus = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
            for d in glob.glob("/sys/class/drm/card*/device"):
                if os.path.basename(os.path.realpath(d)).lower() == str(bus).lower():
                    hw = glob.glob(os.path.join(d, "hwmon", "hwmon*"))
                    if hw and os.path.exists(os.path.join(hw[0], "power1_input")):
                        self.dir = hw[0]
        except Exception:
            self.dir = None

    def _read(self, name):
        try:
            return int(open(os.path.join(self.dir, name)).read())
        except (OSError, ValueError):
            return None

    def start(self):
        import threading
        if not self.dir:
            return

        def run():
            while not self._stop.is_set():
                self.samples.append((time.perf_counter(), self._read("power1_input"), self._read("freq1_input")))
                time.sleep(0.001)
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def stop(self, t0, t1):
        """-> record of the samples taken inside [t0, t1] (perf_counter), or None."""
        if not self._thread:
            return None
        self._stop.set()
        self._thread.join()
        inside = [(p, f) for t, p, f in self.samples if t0 <= t <= t1 and p is not None]
        if not inside:
            return None
        pw = [p / 1e6 for p, _ in inside]
        fq = [f / 1e6 for _, f in inside if f]
        cap = self._read("power1_cap")
        return {"socket_W_mean": float(np.mean(pw)), "socket_W_max": float(np.max(pw)), "cap_W": cap / 1e6 if cap else None,
                "sclk_MHz_smu_mean": float(np.mean(fq)) if fq else None, "samples": len(inside),
                "source": "%s/power1_input, freq1_input sampled every ms" % self.dir}


_NUM_CU = {}


def num_cus(torch, device):
    """Compute units of the device the kernel runs on (hipDeviceProp.multiProcessorCount), not a constant."""
    key = str(device)
    if key not in _NUM_CU:
        _NUM_CU[key] = int(torch.cuda.get_device_properties(device).multi_processor_count)
    return _NUM_CU[key]


def live_clock(eng, launch, launches=12):
    """The shader clock the chip holds under `launch`'s kernel, in THIS run on THIS box: the clock-stamped twin of the
    kernel (uc_clock_probe, include/uchirp.h: one s_memtime / s_memrealtime stamp pair per wave), `launches` back-to-back
    launches right behind the timed region (the clocks are where the timed launches left them), read from the last one."""
    eng.clock_probe(True)
    try:
        for _ in range(launches):
            launch()
        c = eng.clock_read()
    finally:
        eng.clock_probe(False)
    c["method"] = ("uc_clock_read: median over %d waves of cycles / 100 MHz ticks in the clock-stamped twin of the kernel, last "
                   "of %d back-to-back launches right behind the timed region of this run" % (c["waves"], launches))
    return c


def roofline(kernel_key, kernel_name, units, bytes_per_unit, kern_ms, num_cu=256, clock=None):
    """Both roofs of one launch: HBM (algorithmic bytes / kernel time / 8 TB/s) and VALU issue (wave-instructions x 4
    cycles / (4 SIMDs x CUs x in-kernel clock x kernel time)).  `bound`, `achieved`, `peak`, `unit`, `frac` are the HBM roof's
    (the one BASELINE.json's metric names: "% HBM roofline"); `limiter` says which of the two measured fractions is the larger.
    clock: live_clock()'s record of this run; without it the clock of the committed counter pass is used and labelled so."""
    achieved = units * bytes_per_unit / (kern_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "kernel": kernel_name, "kernel_ms": kern_ms, "bytes_per_unit": bytes_per_unit}
    tab, src = valu_table()
    v = tab.get(kernel_key)
    if v:
        cyc = v["valu_insts_per_unit"] * 4.0 / 4.0                       # issue cycles per unit and CU: 4 SIMDs share it
        ghz = clock["shader_ghz"] if clock else v["clock_GHz"]
        frac = v["valu_insts_per_unit"] * units * 4.0 / (4.0 * num_cu * ghz * 1e9 * kern_ms * 1e-3)
        r["valu"] = {"insts_per_unit": v["valu_insts_per_unit"], "issue_cycles_per_unit": cyc,
                     "clock_GHz": ghz, "num_cu": num_cu,
                     "clock_source": clock["method"] if clock else "NOT measured in this run: the clock of the committed counter pass",
                     "clock_GHz_of_the_counter_pass": v["clock_GHz"],
                     "frac": frac, "source": "instructions per unit: " + src + ": " + v.get("source", "")}
        if clock:
            r["valu"]["wave_loop_cycles_median"] = clock["wave_cycles"]
        if "lds_insts_per_unit" in v:
            r["valu"]["lds_insts_per_unit"] = v["lds_insts_per_unit"]
        r["limiter"] = "valu" if frac > r["frac"] else "hbm"
    return r


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


def config2_iq(args, device, stream, torch, mag_mean):
    """BASELINE configs[2] on its stated workload: a continuous real pass-band stream at
    fs = 100 kHz, carrier 18 kHz, base-band chirps of +-1.5 kHz, n = 1024 samples per symbol, 26 samples of FIR
    history in front, generated on the device; UC_IQ with UC_FLAG_IQ_BASEBAND (the experiment's intended maths,
    simulation/IQ_modulation.ipynb cells 16-31: mix, 27-tap low-pass, two dechirp runs, up/down symbol) and, beside
    it, the firmware-window mode (experiments/iq_modulation/Src/main.c:283-285: one history, no symbol)."""
    import uchirp
    from uchirp import synth
    n = 1024
    nf = 2 * args.frames                                  # the same sample count as configs[1]
    x, bits = synth.device_iq_stream(nf, n, device, seed=4321, snr_db=args.snr)
    steps, warm = min(args.steps, 10), min(args.warmup, 3)
    cfg = dict(n=n, fs=100000.0, carrier=18000.0, f0=16500.0, f1=19500.0, time_frame=n / 100000.0, mag_mean=mag_mean)
    out = {"workload": "configs[2]: %d x %d-sample symbols of a continuous real pass-band stream, fs 100 kHz, carrier 18 kHz, "
                       "base-band chirp +-1.5 kHz, SNR %.0f dB, 26 samples of FIR history, generated on the device" % (nf, n, args.snr),
           "frames": nf, "frame_len": n, "steps": steps, "warmup": warm}
    # --- base-band mode: symbols out
    eng = uchirp.Engine(uchirp.IQ, device=device.index, flags=uchirp.FLAG_IQ_BASEBAND, **cfg)
    sym = torch.empty(nf, dtype=torch.uint8, device=device)

    def launch_bb():
        eng.process(x, n_frames=nf, want_stats=False, symbols_out=sym, stream=stream.cuda_stream)

    ramp = clock_ramp(launch_bb, torch, args.ramp_ms)
    wall, kern = timed_launches(launch_bb, stream, torch, steps, warm)
    ncu, clk = num_cus(torch, device), live_clock(eng, launch_bb)
    bb = {"mode": "UC_FLAG_IQ_BASEBAND: mix, 27-tap FIR, dechirp by conj(up) and conj(down), 2 x CFFT-1024, windows around DC, symbol",
          "value": nf / (wall * 1e-3), "unit": "frames/s", "ms_per_step": wall, "ramp_launches": ramp,
          "roofline": roofline("iq1024_bb_f32", "iq1024_kernel<f32,baseband>", nf, 4096 + 1, kern, ncu, clk),
          "bytes_note": "4096 B in + 1 B symbol out per frame (+ 104 B of FIR history once per launch)",
          "bit_error_rate_vs_transmitted": float((sym != bits).float().mean().item())}
    if not args.no_cpu_baseline:
        from oracle import uco
        o = uco.Oracle(uco.IQ, flags=uco.FLAG_IQ_BASEBAND, **cfg)
        head = x[: 26 + 4096 * n].cpu().numpy()
        rs, rst = o.process(head, halo=26, n_frames=4096)
        clear = clear_frames(rst)
        got = sym[:4096].cpu().numpy()
        # (all clear frames equal AND there are clear frames: the mean of an empty mask is NaN, which no comparison catches)
        bb["symbols_equal_oracle_head4096_clear"] = float((got[clear] == rs[clear]).mean()) if clear.any() else 0.0
        bb["oracle_head_clear_frames"] = int(clear.sum())
        bb["oracle_head_near_ties_excluded"] = int((~clear).sum())
        bb["oracle_head_bit_error_rate"] = float((rs != bits[:4096].cpu().numpy()).mean())
    out["baseband"] = bb
    eng.close()
    # --- firmware-window mode: one history record out, no symbol
    eng = uchirp.Engine(uchirp.IQ, device=device.index, n=n, mag_mean=mag_mean)
    st = torch.empty((nf, eng.spf, 8), dtype=torch.float32, device=device)

    def launch_fw():
        eng.process(x, n_frames=nf, want_symbols=False, stats_out=st, stream=stream.cuda_stream)

    clock_ramp(launch_fw, torch, args.ramp_ms / 3)
    wall, kern = timed_launches(launch_fw, stream, torch, steps, warm)
    clk = live_clock(eng, launch_fw)
    out["firmware_windows"] = {
        "mode": "the committed firmware's windows at bin (F1+F2) n / fs (iq_modulation/Src/main.c:215-219,283-285): one dechirp run, one history",
        "value": nf / (wall * 1e-3), "unit": "frames/s", "ms_per_step": wall,
        "roofline": roofline("iq1024_fw_f32", "iq1024_kernel<f32,firmware windows>", nf, 4096 + 32, kern, ncu, clk),
        "bytes_note": "4096 B in + 32 B history record out per frame"}
    out["baseband_over_firmware_windows"] = bb["value"] / out["firmware_windows"]["value"]
    eng.close()
    del x, sym, st
    return out


def config3_stream(args, frames, device, torch):
    """BASELINE configs[3]: the configs[1] batch read as ONE continuous stream through UC_STREAM (27-tap FIR low-pass,
    decimation by 8, overlap-save FFT x H x IFFT compression, include/uchirp.h), captured ONCE into a hipGraph and
    replayed; the eager launches beside it.  The captured launch deals its blocks dynamically like the eager one
    (a counter slot the graph owns)."""
    import uchirp
    eng = uchirp.Engine(uchirp.STREAM, device=device.index)
    x = frames.reshape(-1)
    halo, n_out, n_blocks, hop = eng.stream_geometry(x.numel())
    steps, warm = min(args.steps, 10), min(args.warmup, 3)
    comp = torch.empty(n_out, dtype=torch.float32, device=device)
    pk = torch.empty((n_blocks, 2), dtype=torch.int32, device=device)
    comp_g, pk_g = torch.empty_like(comp), torch.empty_like(pk)
    byts = (x.numel() * 4 + n_out * 4 + n_blocks * 8) / x.numel()
    out = {"workload": "configs[3]: %d samples (2^%.2f, the configs[1] batch as one continuous stream), decimation 8, FFT 2048, "
                       "hop %d, |y| for every decimated sample + one peak record per block" % (x.numel(), np.log2(x.numel()), hop),
           "samples": x.numel(), "blocks": n_blocks, "decim": int(eng.cfg.decim), "steps": steps, "warmup": warm,
           "bytes_per_sample": byts}
    s1 = torch.cuda.current_stream(device)

    def launch():
        eng.process_stream(x, compressed_out=comp, peaks_out=pk, stream=s1.cuda_stream)

    clock_ramp(launch, torch, args.ramp_ms)
    wall, kern = timed_launches(launch, s1, torch, steps, warm)
    ncu, clk = num_cus(torch, device), live_clock(eng, launch)
    out["eager"] = {"value": x.numel() / (wall * 1e-3), "unit": "samples/s", "ms_per_step": wall,
                    "roofline": roofline("stream_d8_f32", "stream_kernel<f32,8>", x.numel(), byts, kern, ncu, clk)}
    s2 = torch.cuda.Stream(device)
    g = torch.cuda.CUDAGraph()
    s2.wait_stream(s1)
    with torch.cuda.stream(s2):
        with torch.cuda.graph(g, stream=s2):
            eng.process_stream(x, compressed_out=comp_g, peaks_out=pk_g, stream=s2.cuda_stream)
        # (capture + instantiation leave the GPU idle for a while: the same clock ramp as in front of every timed loop --
        # without it the first replays run ~9 % slow, and so do eager launches at that moment: profiles/r03_graph_rate_probe.txt)
        clock_ramp(g.replay, torch, args.ramp_ms)
        wall, kern = timed_launches(g.replay, s2, torch, steps, warm)
    torch.cuda.synchronize()
    out["graph_replay"] = {"value": x.numel() / (wall * 1e-3), "unit": "samples/s", "ms_per_step": wall,
                           "roofline": roofline("stream_d8_f32", "stream_kernel<f32,8> (captured hipGraph, replayed)",
                                                x.numel(), byts, kern, ncu, clk)}
    out["graph_equals_eager"] = bool(torch.equal(comp, comp_g) and torch.equal(pk, pk_g))
    out["graph_over_eager"] = out["graph_replay"]["value"] / out["eager"]["value"]
    if not args.no_cpu_baseline:
        from oracle import uco
        o = uco.Oracle(uco.STREAM)
        nb = 6
        head = x[: halo + 8 * nb * hop].cpu().numpy()
        cr, pr = o.process_stream(head)
        got = comp_g[: cr.size].cpu().numpy()
        out["head_rel_err_vs_oracle"] = float(np.abs(got - cr).max() / cr.max())
        out["head_peak_offsets_equal_oracle"] = bool(np.array_equal(pk_g[:nb, 1].cpu().numpy().astype(np.uint32),
                                                                    pr["offset"][:nb]))
    eng.close()
    return out


def hello_world1(args, device, torch, mag_mean):
    """The N > 1 leg at world size 1, inside the N = 1 run: configs[4] framing, matched sweep, a uc_group of one device
    (include/uchirp.h: the RCCL communicator and the in-place all-gather of the symbol stream are made and called from C,
    on the group's gather stream, every step; three buffers in rotation as in the N > 1 run), decode.
    Gives the driver's 1 -> N efficiency a like-for-like anchor (the N = 1 contract line runs configs[1] without a gather)."""
    import uchirp
    from uchirp import synth
    nf = args.frames
    grp = uchirp.Group(uchirp.RX_REAL, devices=[device.index], mag_mean=mag_mean, time_frame=MATCHED_TIME_FRAME)
    try:
        frames, sent = synth.device_hello_frames(0, nf, device, seed=1234, snr_db=args.snr, msg=MSG)
        # a stream of its own: a NULL entry in uc_group_process_batch's stream list means "the group's own stream", and
        # torch's default stream IS the NULL stream -- the HIP events below must sit on the stream the kernel runs on
        stream = torch.cuda.Stream(device)
        gat2 = [torch.empty(nf, dtype=torch.uint8, device=device) for _ in range(NBUF)]
        torch.cuda.synchronize()

        def step(k, e0=None, e1=None):
            if e0 is not None:
                e0.record(stream)
            grp.process([frames], nf, [gat2[k % NBUF]], streams=[stream.cuda_stream])
            if e1 is not None:
                e1.record(stream)

        clock_ramp(lambda: step(0), torch, args.ramp_ms)
        for k in range(args.warmup):
            step(k)
        grp.synchronize()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k, ev[k][0], ev[k][1])
        grp.synchronize()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        kern = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        g = gat2[(args.steps - 1) % NBUF]
        # the checker: the same frames through a plain context (uc_process_batch), no group
        eng = uchirp.Engine(uchirp.RX_REAL, device=device.index, mag_mean=mag_mean, time_frame=MATCHED_TIME_FRAME)
        ref, _ = eng.process(frames, want_stats=False)
        torch.cuda.synchronize()
        ok = bool(torch.equal(g, ref))
        eng.close()
        texts = synth.decode_hello(g.cpu().numpy(), len(MSG))
        good = sum(1 for t in texts if t == MSG)
        ms = elapsed / args.steps * 1e3
        import hashlib
        out = {"workload": "configs[4] at world size 1: %d x 2048-sample frames, K7 'Hello World!' framing, SNR %.0f dB, matched "
                           "sweep, all-gather of the symbol stream every step" % (nf, args.snr),
               "gather_backend": "uc_group_process_batch: RCCL (ncclAllGather, in place) called from C on the group's gather stream",
               "value": nf * args.steps / elapsed, "unit": "frames/s", "ms_per_step": ms, "steps": args.steps,
               "kernel_ms": kern, "gather_ms_exposed": ms - kern, "gathered_equals_decoded": ok,
               "symbols_sha256": hashlib.sha256(g.cpu().numpy().tobytes()).hexdigest(),
               "transmissions": len(texts), "transmissions_decoded_exactly": good}
        del frames, gat2
        return out
    finally:
        grp.close()


def receive_leg(args, device, torch):
    """SURVEY.md section 8 f1 as a number: the WHOLE receiver (ISR FIFO, the 8 dsp() offsets x {up, down} of every block,
    main()'s switch and resync, byte assembly: receiver/Src/main.c:417-554, 243-273, 659-668) for thousands of recorded
    microphone streams at once -- uc_receive_streams: the band kernel's ROWS build over the 8 FIFO offsets every block adds
    (ONE launch over all blocks for few streams; from 1024 streams on block by block, each step evaluating only what the switch
    can still look at -- round 6: the 4096-stream call 21.7 -> 14.6 ms) (the other 9 of its FIFO were evaluated when the block before it arrived, main.c:662; frames read through two
    base addresses from the caller's buffer, nothing packed or copied), the switch replayed on the device one wave or lane per
    stream.  Streams: 40 blocks of noise + a sample skew,
    the K7 "Hello World!" transmission rendered at 78 125 Hz, noise; generated on the device.  Real time for ONE
    microphone is 38.1 blocks/s (the MCU keeps up with exactly one)."""
    import ctypes as C
    import uchirp
    from uchirp import tx
    fs, nb = 78125.0, 176
    tone = torch.from_numpy(tx.render(MSG, fs_rx=fs, amplitude=2000.0).astype(np.float32)).to(device)
    out = {"workload": "recorded streams of %d blocks (%.2f s of microphone signal each): 40 blocks of noise + 777 samples, the K7 "
                       "'%s' transmission at 78 125 Hz (amplitude 2000, noise sigma 50), noise" % (nb, nb * N / fs, MSG),
           "blocks_per_stream": nb, "real_time_blocks_per_s_per_stream": fs / N}
    L = uchirp.lib()
    stream = torch.cuda.current_stream(device)
    # (the complex-reference receiver -- the variant whose state machine decodes the whole text; its DSP launches
    # are band_kernel<sync_cplx>, so the rocprofv3 average of the HEADLINE kernel over this command stays the contract leg's.
    # The shipping real-reference receiver runs the same call at twice the rate: profiles/r04_receive_many.txt)
    for ns, var, name in ((4096, uchirp.SYNC_CPLX, "sync_cplx_4096_streams"), (64, uchirp.SYNC_CPLX, "sync_cplx_64_streams"),
                          (1, uchirp.SYNC_CPLX, "sync_cplx_1_stream")):
        g = torch.Generator(device=device)
        g.manual_seed(ns)
        x = torch.randn((ns, nb * N), generator=g, device=device) * 50.0
        lead = 40 * N + 777
        x[:, lead:lead + tone.numel()] += tone
        eng = uchirp.Engine(var, device=device.index)
        cap = 64
        text = torch.zeros((ns, cap), dtype=torch.uint8, device=device)
        ntext = torch.zeros(ns, dtype=torch.int32, device=device)

        def call():
            rc = L.uc_receive_streams(eng._h, C.c_void_p(x.data_ptr()), uchirp.DTYPE_F32, ns, nb * N, 0, None,
                                      C.c_void_p(text.data_ptr()), cap, C.c_void_p(ntext.data_ptr()), None, 0, None,
                                      C.c_void_p(stream.cuda_stream))
            if rc != 0:
                raise RuntimeError(L.uc_last_error().decode())

        for _ in range(3):
            call()
        torch.cuda.synchronize()
        reps = 5 if ns > 64 else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        texts = [bytes(r[:k]).decode("latin-1") for r, k in zip(text.cpu().numpy(), ntext.cpu().numpy())]
        out[name] = {"streams": ns, "ms_per_call": dt * 1e3, "blocks_per_s": ns * nb / dt,
                     "dsp_frames_per_s": ns * nb * 8 / dt,   # (FIFO offsets COVERED per second: from 1024 streams on not all are transformed)
                     "x_real_time": ns * nb / dt / (fs / N),
                     "streams_decoding_the_text": sum(1 for t in texts if MSG in t), "first_text": texts[0]}
        if ns == 4096:
            # the same microphones LIVE: one new block of every stream per call (uc_rx_state / uc_receive_streams_next), the
            # firmware's own mode of operation; the texts of the chunks must add up to the recorded-stream call's
            live = eng.live(ns)
            chunks = [x[:, b * N:(b + 1) * N].contiguous() for b in range(nb)]
            acc_buf, acc_len = np.zeros((ns, 4 * cap), np.uint8), np.zeros(ns, np.int64)   # what every stream has received so far
            rows_all = np.arange(ns)
            nt_host = torch.empty(ns, dtype=torch.int32).pin_memory()     # (a live host keeps pinned landing buffers)
            tx_host = torch.empty((ns, cap), dtype=torch.uint8).pin_memory()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for ch in chunks:
                rc = L.uc_receive_streams_next(eng._h, live._h, C.c_void_p(ch.data_ptr()), uchirp.DTYPE_F32, N, 0, None,
                                               C.c_void_p(text.data_ptr()), cap, C.c_void_p(ntext.data_ptr()), None, 0, None,
                                               C.c_void_p(stream.cuda_stream))
                if rc != 0:
                    raise RuntimeError(L.uc_last_error().decode())
                nt_host.copy_(ntext, non_blocking=True)        # (a live host reads its characters after every block:
                tx_host.copy_(text, non_blocking=True)         #  16 KiB of counts + 256 KiB of characters, one wait
                stream.synchronize()                           #  per 26.2 ms)
                nt = nt_host.numpy()
                if nt.any():
                    tt = tx_host.numpy()
                    for cpos in range(int(nt.max())):              # (a block completes at most one character per stream)
                        m = nt > cpos
                        acc_buf[rows_all[m], np.minimum(acc_len[m], 4 * cap - 1)] = tt[m, cpos]
                        acc_len[m] += 1
            torch.cuda.synchronize()
            dt_live = (time.perf_counter() - t0) / nb
            live.close()
            same = sum(1 for si in range(ns) if bytes(acc_buf[si, :acc_len[si]]).decode("latin-1") == texts[si])
            out["live_4096_streams"] = {"streams": ns, "blocks_per_call": 1, "calls": nb, "ms_per_call": dt_live * 1e3,
                                        "real_time_ms_per_call": N / fs * 1e3, "headroom_x_real_time": N / fs / dt_live,
                                        "microphones_served_in_real_time": int(ns * N / fs / dt_live),
                                        "streams_whose_chunks_add_up_to_the_recorded_call": same,
                                        "new_dsp_frames_per_call": ns * 8,
                                        "what": "one new block of every stream per call, the host reads the counts back after "
                                                "every call (a sync per block)"}
            # the same step with no host in the loop: back to back on one stream, and replayed from ONE captured hipGraph
            # (everything a step carries -- newest block, the 9 surviving FIFO records, main()'s locals, block counts -- lives
            # on the device); then the chain from the microphones' 1-bit PDM streams (UC_DTYPE_PDM: + the DFSDM, on the device)
            live = eng.live(ns)
            side = torch.cuda.Stream(device)
            reps_a = 60
            with torch.cuda.stream(side):
                for k in range(10):
                    live.next_into(chunks[k], text, ntext, stream=side.cuda_stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
                for k in range(reps_a):
                    live.next_into(chunks[10 + k], text, ntext, stream=side.cuda_stream)
                e1.record(side)
                e1.synchronize()
                eager_ms = e0.elapsed_time(e1) / reps_a
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    live.next_into(chunks[0], text, ntext, stream=side.cuda_stream)
                for k in range(5):
                    gr.replay()
                e0.record(side)
                for k in range(reps_a):
                    gr.replay()
                e1.record(side)
                e1.synchronize()
                graph_ms = e0.elapsed_time(e1) / reps_a
            live.reset()
            live.keep_previous(True)       # (every chunk is a buffer of its own here: uc_rx_state_keep_previous holds)
            with torch.cuda.stream(side):
                for k in range(10):
                    live.next_into(chunks[k], text, ntext, stream=side.cuda_stream)
                e0.record(side)
                for k in range(reps_a):
                    live.next_into(chunks[10 + k], text, ntext, stream=side.cuda_stream)
                e1.record(side)
                e1.synchronize()
                keep_ms = e0.elapsed_time(e1) / reps_a
            live.close()
            out["live_4096_streams"].update({"ms_per_call_back_to_back": eager_ms, "ms_per_call_graph_replay": graph_ms,
                                             "ms_per_call_back_to_back_keep_previous": keep_ms,
                                             "contracts": "ms_per_call, ..._back_to_back and ..._graph_replay: the default contract "
                                                          "(the state keeps a copy of every stream's newest block); "
                                                          "..._keep_previous: uc_rx_state_keep_previous (every chunk here is a "
                                                          "buffer of its own)",
                                             "microphones_served_in_real_time_back_to_back": int(ns * N / fs / (eager_ms * 1e-3))})
            live = eng.live(ns)
            gp = torch.Generator(device=device)
            gp.manual_seed(7)
            pdm = [torch.randint(-(1 << 31), (1 << 31) - 1, (ns, N), generator=gp, device=device, dtype=torch.int64).to(torch.int32)
                   for _ in range(4)]
            with torch.cuda.stream(side):
                for k in range(6):
                    live.next_into(pdm[k % 4], text, ntext, stream=side.cuda_stream, pdm=True)
                e0.record(side)
                for k in range(reps_a):
                    live.next_into(pdm[k % 4], text, ntext, stream=side.cuda_stream, pdm=True)
                e1.record(side)
                e1.synchronize()
            live.close()
            out["live_pdm_4096_streams"] = {"streams": ns, "ms_per_call_back_to_back": e0.elapsed_time(e1) / reps_a,
                                            "what": "one new block of every microphone per call as 2048 x 32 PDM bits "
                                                    "(UC_DTYPE_PDM, random bits: timing only; parity: tests/test_dfsdm.py): "
                                                    "sinc5 + history, ROWS band launch, replay"}
            del chunks, pdm
        eng.close()
        del x, text, ntext
    # the shipping receiver (RX_REAL) LIVE at the scale one GPU serves: 65 536 microphones, one new block each per call, back
    # to back.  Silent microphones here (noise: 94 GB would be needed for a transmission in every stream): an IDLE stream's
    # switch can look at 3 or 5 of the 8 offsets its new block adds (main.c:447-453), and only those are evaluated; a stream in
    # a tracking state costs all 8 (1.0 ms per block at this size, profiles/r05_live_async.txt)
    ns = 65536
    eng = uchirp.Engine(uchirp.RX_REAL, device=device.index)
    live = eng.live(ns)
    g = torch.Generator(device=device)
    g.manual_seed(99)
    bufs = [torch.randn((ns, N), generator=g, device=device) * 50.0 for _ in range(3)]
    text = torch.zeros((ns, 8), dtype=torch.uint8, device=device)
    ntext = torch.zeros(ns, dtype=torch.int32, device=device)
    side = torch.cuda.Stream(device)
    ms_by_contract = {}
    for contract in ("default", "keep_previous"):
        # default: the library copies every stream's newest block into the state on its way through the kernel (the caller may
        # overwrite `samples` at once); keep_previous: the caller leaves a chunk alone until the NEXT call has completed -- a ring
        # of >= 2 buffers, which this loop's three are -- and nothing is copied (uc_rx_state_keep_previous)
        live.reset()
        live.keep_previous(contract == "keep_previous")
        with torch.cuda.stream(side):
            for k in range(12):
                live.next_into(bufs[k % 3], text, ntext, stream=side.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for k in range(60):
                live.next_into(bufs[k % 3], text, ntext, stream=side.cuda_stream)
            e1.record(side)
            e1.synchronize()
        ms_by_contract[contract] = e0.elapsed_time(e1) / 60
    ms, ms_keep = ms_by_contract["default"], ms_by_contract["keep_previous"]
    live.close()
    eng.close()
    out["live_idle_rx_real_65536_streams"] = {"streams": ns, "ms_per_call_back_to_back": ms_keep,
                                              "contract": "uc_rx_state_keep_previous (this loop's ring of three chunk buffers keeps "
                                                          "it); the default contract -- the library copies every stream's newest "
                                                          "block into the state, what rounds 4-5 measured -- is "
                                                          "ms_per_call_back_to_back_default_contract",
                                              "ms_per_call_back_to_back_default_contract": ms,
                                              "real_time_ms_per_call": N / fs * 1e3,
                                              "microphones_served_in_real_time": int(ns * N / fs / (ms_keep * 1e-3)),
                                              "microphones_served_in_real_time_default_contract": int(ns * N / fs / (ms * 1e-3)),
                                              "what": "uc_receive_streams_next, one new block of each of 65 536 silent microphones "
                                                      "per call (IDLE streams: 3 or 5 of the 8 new FIFO offsets are evaluated, the "
                                                      "others cost nothing); keep_previous = uc_rx_state_keep_previous: the caller's "
                                                      "ring of chunk buffers is read in place, no block is copied into the state"}
    del bufs
    return out


def stream_measurement(args, eng, frames, rank, torch):
    """BASELINE config 4 (side measurement, eager launches): the batch read as ONE continuous stream through UC_STREAM."""
    x = frames.reshape(-1)
    halo, n_out, n_blocks, hop = eng.stream_geometry(x.numel())
    comp = torch.empty(n_out, dtype=torch.float32, device=x.device)
    pk = torch.empty((n_blocks, 2), dtype=torch.int32, device=x.device)
    stream = torch.cuda.current_stream(x.device)

    def launch():
        eng.process_stream(x, compressed_out=comp, peaks_out=pk, stream=stream.cuda_stream)

    clock_ramp(launch, torch, args.ramp_ms)
    wall, kern = timed_launches(launch, stream, torch, args.steps, args.warmup)
    ncu, clk = num_cus(torch, x.device), live_clock(eng, launch)
    byts = (x.numel() * 4 + n_out * 4 + n_blocks * 8) / x.numel()
    if rank == 0:
        print(json.dumps({"metric": "input samples/s (stream: FIR decimate + overlap-save compression, side measurement)",
                          "value": x.numel() / (wall * 1e-3), "unit": "samples/s", "n_gpus": 1, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": wall, "decim": int(eng.cfg.decim),
                          "blocks": n_blocks, "blocks_per_s": n_blocks / (wall * 1e-3),
                          "roofline": roofline("stream_d%d_f32" % int(eng.cfg.decim), "stream_kernel<f32,%d>" % int(eng.cfg.decim),
                                               x.numel(), byts, kern, ncu, clk)}), flush=True)


SIDE = {  # variant -> (algorithmic bytes per frame, key of profiles/r*_valu_insts.json, kernel)
    "sync_cplx": (8193, "band_sync_cplx_f32", "band_kernel<sync_cplx,f32>"),
    "compress": (8192 + 32, "compress_f32", "compress_kernel<f32>"),
    "dechirp_down": (8192 + 32, "band_dechirp_down_f32", "band_kernel<dechirp_down,f32>"),
    "iq": (8192 + 104 + 32, "iq2048_fw_f32", "iq_kernel<f32,firmware windows>"),
    "iq1024": (4096 + 104 + 32, "iq1024_fw_f32", "iq1024_kernel<f32,firmware windows>"),
    "iq_bb": (8192 + 104 + 1, "iq2048_bb_f32", "iq_kernel<f32,baseband>"),
    "iq1024_bb": (4096 + 104 + 1, "iq1024_bb_f32", "iq1024_kernel<f32,baseband>"),
}


def side_measurement(args, eng, frames, world, rank, torch):
    """Not the contract line: frames/s of one of the sibling variants on the same synthetic batch."""
    if args.variant == "stream":
        return stream_measurement(args, eng, frames, rank, torch)
    n = eng.n
    per_frame, key, kname = SIDE[args.variant]
    nfr = (frames.numel() - eng.halo - n) // n + 1
    want_sym = args.variant in ("sync_cplx", "iq_bb", "iq1024_bb")
    stats = None if want_sym else torch.empty((nfr, eng.spf, 8), dtype=torch.float32, device=frames.device)
    sym = torch.empty(nfr, dtype=torch.uint8, device=frames.device) if want_sym else None
    stream = torch.cuda.current_stream(frames.device)

    def launch():
        eng.process(frames, n_frames=nfr, want_symbols=want_sym, want_stats=not want_sym, symbols_out=sym, stats_out=stats,
                    stream=stream.cuda_stream)

    clock_ramp(launch, torch, args.ramp_ms)
    wall, kern = timed_launches(launch, stream, torch, args.steps, args.warmup)
    ncu, clk = num_cus(torch, frames.device), live_clock(eng, launch)
    if rank == 0:
        print(json.dumps({"metric": "chirp frames/s (%s, side measurement)" % args.variant, "value": nfr / (wall * 1e-3),
                          "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall,
                          "frame_len": n, "frames": nfr,
                          "roofline": roofline(key, kname, nfr, per_frame, kern, ncu, clk)}), flush=True)


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
                    choices=["rx_real", "sync_cplx", "compress", "dechirp_down", "iq", "iq1024", "iq_bb", "iq1024_bb", "stream"],
                    help="default rx_real = BASELINE configs[1]; the others are side measurements")
    ap.add_argument("--no-configs", action="store_true", help="N = 1: skip the configs[2] / configs[3] block")
    ap.add_argument("--no-hello1", action="store_true", help="N = 1: skip the configs[4] leg at world size 1")
    ap.add_argument("--no-receive", action="store_true", help="N = 1: skip the multi-stream receiver leg")
    ap.add_argument("--sustain-s", type=float, default=1.5,
                    help="N = 1: seconds of untimed back-to-back launches behind the timed region during which socket power and "
                         "the SMU clock are sampled (hwmon); 0 = none")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="N = 1: do not run the two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE); roofline.traffic then "
                         "comes from the committed record and says so")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak (default, the contract's line) = --frames per GPU; strong = --frames IN ALL, uc_partition shares "
                         "of --frames / N per GPU (N must divide it), the same gather -- the launch, tail and gather costs that a "
                         "fixed 1 Mi frames per GPU hides show here")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N in ONE process: a uc_group over N devices (include/uchirp.h), as a C host would drive the node "
                         "(tests/c/host_multi.c); the default is one process per GPU")
    return ap.parse_args(argv)


class Watchdog:
    """A rank that hangs (a peer died inside a collective, a rendezvous that never completes) must end the run LOUDLY and
    soon, not sit until the driver's own limit kills it without a line of output.  Every rank starts one: the main thread
    marks its progress (`mark`), a daemon thread checks the clock; past the limit it prints which rank, in which phase, after
    which step, for how long -- and ends the process with status 4 (os._exit: a thread cannot unblock a collective; the
    launcher then ends the other ranks).  UC_BENCH_TIMEOUT seconds WITHOUT PROGRESS (default 300: no phase of a healthy run --
    the first `import torch` on a fresh box, the rendezvous, a timed step -- takes that long; a slow but healthy 8-rank run
    keeps marking progress and is not cut off) and three times that for the whole run; never a retry."""

    def __init__(self, rank, world, limit_s):
        import threading
        self.rank, self.world, self.limit = rank, world, float(limit_s)
        self.t0 = time.time()
        self.phase, self.step, self.t_mark = "start", -1, self.t0
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()

    def mark(self, phase, step=-1):
        self.phase, self.step, self.t_mark = phase, step, time.time()

    def stop(self):
        self._stop.set()

    def _run(self):
        while not self._stop.wait(0.5):
            now = time.time()
            if now - self.t_mark > self.limit or now - self.t0 > 3.0 * self.limit:
                sys.stderr.write("bench.py WATCHDOG: rank %d of %d gave up after %.0f s (UC_BENCH_TIMEOUT): last progress %.1f s ago, "
                                 "phase '%s'%s -- a peer has probably died or never arrived; exiting with status 4\n"
                                 % (self.rank, self.world, now - self.t0, now - self.t_mark, self.phase,
                                    (", last completed step %d" % self.step) if self.step >= 0 else ""))
                sys.stderr.flush()
                os._exit(4)


def shard_frames(args, world):
    """frames per GPU and step: --frames (weak scaling, the contract's line) or --frames / world (strong: --frames in all, the
    equal uc_partition shares)"""
    if args.scaling == "strong" and world > 1:
        if args.frames % world:
            raise SystemExit("bench.py --scaling strong: %d GPUs do not divide %d frames" % (world, args.frames))
        return args.frames // world
    return args.frames


def bench_timeout():
    return float(os.environ.get("UC_BENCH_TIMEOUT", "300"))


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
        # This pool's host driver only supports dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL's intra-node setup
        # (hipIpcGetMemHandle) fails with "invalid argument".  The image exports it already; keep it if a caller's
        # environment dropped it.  (setdefault: an explicit value from the caller wins.)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("NCCL_DEBUG", "WARN")         # a failing RCCL call says why, on stderr, in the run that failed
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else sys.stderr))
    # every rank carries a watchdog of its own (UC_BENCH_TIMEOUT: 300 s without progress, 900 s in all) and reports where it hung;
    # this parent waits a little longer than the ranks' own cap, then says which ranks were still alive and ends exactly those
    deadline = time.time() + 3.0 * bench_timeout() + 15.0
    rc = 0
    while rc == 0 and any(p.poll() is None for p in procs):
        time.sleep(0.2)
        rc = next((p.returncode for p in procs if p.poll() not in (None, 0)), 0)
        if time.time() > deadline:
            rc = -1
    if rc:
        alive = [r for r, p in enumerate(procs) if p.poll() is None]
        ended = {r: p.returncode for r, p in enumerate(procs) if p.poll() is not None}
        sys.stderr.write("bench.py: %s; ranks still running: %s; ranks that had ended (exit status): %s -- ending the rest\n"
                         % ("timed out" if rc == -1 else "a rank failed (exit %s)" % rc, alive, ended))
        for p in procs:                      # a rank died or hung: the others wait in a collective; end exactly those
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    out0.seek(0)
    text = out0.read().decode()
    # stdout carries the JSON line(s) only; anything a library printed on rank 0's stdout goes to stderr.  Exit status 3
    # = a correctness gate failed on rank 0: its JSON line (with `gates_failed`) is still relayed.
    for ln in text.splitlines():
        (sys.stdout if ln.startswith("{") and rc in (0, 3) else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    if rc:
        raise SystemExit("bench.py: a rank failed (exit %s)" % rc if rc != 3 else 3)


def single_process(args):
    """`python bench.py --gpus N --single-process`: configs[4] with ONE host process driving all N GPUs through a uc_group
    (include/uchirp.h: uc_group_create = ncclCommInitAll, one launch stream + one gather stream per device) -- the shape a C
    host of the library has (tests/c/host_multi.c is that host in C99); same workload, same timed region, same JSON line as
    the one-process-per-GPU run.  torch only makes the frames and holds the buffers."""
    import hashlib
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    world = args.gpus
    # UC_BENCH_REHEARSE=1 (with UC_TUNING=1 UC_GROUP_SHARE_DEVICES=1 UC_RCCL_LIB=<tests/stubs/loopback_rccl.cpp built>): every
    # rank on device 0, the loop-back stand-in for RCCL -- the plumbing of this mode on a one-GPU box, never a measurement
    rehearse = os.environ.get("UC_BENCH_REHEARSE") == "1"
    if not rehearse and torch.cuda.device_count() < world:
        raise SystemExit("bench.py --single-process --gpus %d: only %d device(s) visible" % (world, torch.cuda.device_count()))
    import uchirp
    from uchirp import synth
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)                              # RCCL's banner and anything else on descriptor 1 goes to stderr
    nf, mag_mean = shard_frames(args, world), 1000.0
    devs = [torch.device("cuda", 0 if rehearse else d) for d in range(world)]
    wd = Watchdog(0, 1, bench_timeout())      # (one process: ncclCommInitAll and the in-process gathers can hang too)
    wd.mark("uc_group_create (ncclCommInitAll over %d devices)" % world)
    grp = uchirp.Group(uchirp.RX_REAL, devices=[d.index for d in devs], mag_mean=mag_mean, time_frame=MATCHED_TIME_FRAME)
    frames, streams = [], []
    for r, d in enumerate(devs):
        f, _ = synth.device_hello_frames(r * nf, nf, d, seed=1234 + r, snr_db=args.snr, msg=MSG)
        frames.append(f)
        streams.append(torch.cuda.Stream(d))
    gat = [[torch.empty(world * nf, dtype=torch.uint8, device=d) for d in devs] for _ in range(NBUF)]
    handles = [st.cuda_stream for st in streams]

    def step(k, ev=None):
        if ev is not None:
            for r in range(world):
                if k >= NBUF:                 # the write-after-gather wait in FRONT of the bracket: [e0, e1] = the kernel alone
                    grp.wait_gather(r, gat[k % NBUF][r], handles[r])
                ev[r][0].record(streams[r])
        grp.process(frames, world * nf, gat[k % NBUF], streams=handles)
        if ev is not None:
            for r in range(world):
                ev[r][1].record(streams[r])

    def sync():
        grp.synchronize()
        for d in devs:
            torch.cuda.synchronize(d)

    ramp_launches = 0
    t_r = time.perf_counter()
    while args.ramp_ms > 0 and (time.perf_counter() - t_r) * 1e3 < args.ramp_ms:
        for _ in range(4):
            step(0)
        sync()
        ramp_launches += 4
    wd.mark("warm-up")
    for k in range(args.warmup):
        step(k)
        wd.mark("warm-up", k)
    sync()
    evs = []
    for k in range(args.steps):
        row = []
        for d in devs:
            with torch.cuda.device(d):
                row.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
        evs.append(row)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, evs[k])
        wd.mark("timed region: enqueued", k)
    wd.mark("timed region: waiting for the devices (gathers included)", args.steps - 1)
    sync()
    elapsed = time.perf_counter() - t0
    wd.stop()
    km = np.array([[a.elapsed_time(b) for a, b in row] for row in evs]).mean(axis=0)       # kernel ms by device
    last = gat[(args.steps - 1) % NBUF]
    host = [g.cpu().numpy() for g in last]
    digs = [hashlib.sha256(h.tobytes()).hexdigest() for h in host]
    gate_failures = []
    if len(set(digs)) != 1:
        gate_failures.append("devices hold different gathered symbol streams")
    texts = synth.decode_hello(host[0], len(MSG))
    good = sum(1 for t in texts if t == MSG)
    if good != len(texts):
        gate_failures.append("hello: %d of %d transmissions decode" % (good, len(texts)))
    data = synth.hello_kind_stream(0, world * nf, MSG)
    m = data != 2
    value = world * nf * args.steps / elapsed
    out = {"metric": "chirp frames/s (2048-pt FFT demod)", "value": value, "unit": "frames/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
           "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "ramp_ms": args.ramp_ms,
           "ramp_launches": ramp_launches,
           "gather_backend": "uc_group_process_batch in ONE process: ncclCommInitAll, ncclAllGather in place per device inside "
                             "one ncclGroupStart/End, called from C",
           "config": {"workload": "configs[4]: %d x 2048-sample fp32 frames per GPU, frame-sharded 'Hello World!' stream (K7 "
                                  "framing repeated over the global frame index), SNR %.0f dB, rx_real with the reference "
                                  "sweep matched to the frame" % (nf, args.snr),
                      "frames_per_gpu": nf, "frame_len": N, "variant": "rx_real", "time_frame": MATCHED_TIME_FRAME,
                      "parallelism": "frame-sharded x%d, ONE host process (uc_group), RCCL all-gather of the symbol stream "
                                     "(1 B/frame) every step" % world},
           "roofline": roofline("band_rx_real_f32", "band_kernel<rx_real,f32>", nf, BYTES_PER_FRAME, float(km.max())),
           "per_rank": {"kernel_ms_by_rank": [float(v) for v in km]},
           "gather_ms_exposed": elapsed / args.steps * 1e3 - float(km.max()),
           "value_per_gpu": value / world, "symbols_sha256": digs[0],
           "decoded_text_first": texts[0] if texts else "", "transmissions": len(texts),
           "transmissions_decoded_exactly": good,
           "bit_error_rate_vs_transmitted": float((host[0][m] != data[m]).mean()),
           "gates_failed": gate_failures}
    if rehearse:
        out["rehearsal"] = "every rank on device 0, loop-back stand-in for RCCL: plumbing only, not a measurement"
    print(json.dumps(out), file=json_out, flush=True)
    grp.close()
    if gate_failures:
        sys.stderr.write("bench.py: correctness gate(s) failed: %s\n" % "; ".join(gate_failures))
        raise SystemExit(3)


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.single_process and env_world is None:
        return single_process(args)
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
    wd = Watchdog(rank, world, bench_timeout() if world > 1 else 1e9)
    # UC_BENCH_HELLO=1: run the N > 1 leg -- configs[4] framing, RCCL process group, async all-gather, digest check,
    # text decode -- with whatever world size there is, 1 included (what a one-GPU box can exercise of it on RCCL).
    multi = world > 1 or os.environ.get("UC_BENCH_HELLO") == "1"
    # RCCL prints its version banner on file descriptor 1 (the N = 1 run starts it too, for `hello_world1`); stdout must
    # carry the JSON line and nothing else: keep a private copy of the real stdout for that line and point descriptor 1
    # at stderr for everyone else
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:      # (UC_BENCH_HELLO=1 outside a launcher: one process, any free port)
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        os.environ.setdefault("NCCL_DEBUG", "WARN")  # (also under torch.distributed.run: RCCL's own reason for a failure, on stderr)
        wd.mark("rendezvous (torch.distributed.init_process_group)")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        world = dist.get_world_size()          # n_gpus printed below is what the process group reports
        wd.mark("rendezvous done")

    from uchirp import synth
    mag_mean = 1000.0
    nf = shard_frames(args, world)
    hello = multi and args.variant == "rx_real"
    eng = None
    if have_gpu:
        import uchirp
        vmap = {"rx_real": (uchirp.RX_REAL, {}), "sync_cplx": (uchirp.SYNC_CPLX, {}), "compress": (uchirp.COMPRESS, {}),
                "dechirp_down": (uchirp.DECHIRP_DOWN, {}), "iq": (uchirp.IQ, {}), "iq1024": (uchirp.IQ, {"n": 1024}),
                "stream": (uchirp.STREAM, {})}
        for nn in (2048, 1024):   # base-band I/Q, configs[2]'s constants (the batch is not its workload: timing only)
            vmap["iq_bb" if nn == 2048 else "iq1024_bb"] = (uchirp.IQ, dict(
                n=nn, fs=100000.0, carrier=18000.0, f0=16500.0, f1=19500.0, time_frame=nn / 100000.0,
                flags=uchirp.FLAG_IQ_BASEBAND))
        vid, vkw = vmap[args.variant]
        if hello:
            vkw = dict(vkw, time_frame=MATCHED_TIME_FRAME)
        eng = uchirp.Engine(vid, device=local_rank, mag_mean=mag_mean, **vkw)
    # The gather of the N > 1 leg goes through the C-ABI: a uc_group of one rank per process (include/uchirp.h,
    # uc_group_create_rank) -- the RCCL communicator is made from C out of a unique id that rank 0 draws and
    # torch.distributed merely carries to the others; every step decodes straight into the rank's slice of the gathered
    # stream and ncclAllGather runs in place on the group's gather stream.  torch.distributed stays the launcher's
    # rendezvous, the barrier and the max-over-ranks of the contract.  If any rank cannot build its group every rank
    # falls back to torch.distributed's all_gather_into_tensor (recorded as `gather_backend`); UC_BENCH_TORCH_GATHER=1
    # selects that path outright.
    # (UC_BENCH_REHEARSE=1 on a one-GPU box: the group path runs too when the loop-back stand-in for RCCL is named --
    # UC_TUNING=1 UC_RCCL_LIB=<tests/stubs/loopback_rccl.cpp built> -- otherwise the rehearsal gathers through gloo)
    grp, gather_backend = None, "torch.distributed all_gather_into_tensor"
    if (multi and args.variant == "rx_real" and have_gpu and os.environ.get("UC_BENCH_TORCH_GATHER") != "1"
            and (not rehearse or os.environ.get("UC_RCCL_LIB"))):
        ctl = torch.device("cpu") if rehearse else device           # where the launcher's own collectives live (gloo / RCCL)
        idt = torch.zeros(uchirp.GROUP_ID_BYTES, dtype=torch.uint8, device=ctl)
        why = ""
        # Preflight on EVERY rank, agreed on before anybody enters ncclCommInitRank: a rank that cannot load RCCL, select its
        # device or make a context would never arrive there, and the others would wait for it until the watchdog fires.
        wd.mark("uc_group_preflight")
        pre = 1
        try:
            uchirp.Group.preflight(uchirp.RX_REAL, device=local_rank, mag_mean=mag_mean, time_frame=MATCHED_TIME_FRAME)
        except Exception as ex:
            pre, why = 0, "preflight on rank %d: %s" % (rank, ex)
        pf = torch.tensor([pre], dtype=torch.int32, device=ctl)
        dist.all_reduce(pf, op=dist.ReduceOp.MIN)
        pre_all = int(pf.item()) == 1
        if rank == 0 and pre_all:
            try:
                idt.copy_(torch.frombuffer(bytearray(uchirp.Group.unique_id()), dtype=torch.uint8))
            except Exception as ex:                      # (an all-zero id tells the others)
                why = str(ex)
        dist.broadcast(idt, 0)
        ok = 0
        wd.mark("uc_group_create_rank (ncclCommInitRank)")
        if pre_all and bool(idt.any().item()):
            try:
                grp = uchirp.Group(uchirp.RX_REAL, world=world, rank=rank, unique_id=idt.cpu().numpy().tobytes(),
                                   device=local_rank, mag_mean=mag_mean, time_frame=MATCHED_TIME_FRAME)
                ok = 1
            except Exception as ex:
                why = str(ex)
        flag = torch.tensor([ok], dtype=torch.int32, device=ctl)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            gather_backend = "uc_group_process_batch: RCCL (ncclAllGather, in place) called from C on the group's gather stream"
        else:
            if grp is not None:
                grp.close()
                grp = None
            gather_backend += " (uc_group unavailable: %s)" % (why[:200] or "another rank failed")
    if hello:
        # rank r owns frames [r nf, (r + 1) nf) of the global stream (uc_partition, equal shares)
        frames, sent = synth.device_hello_frames(rank * nf, nf, device, seed=1234 + rank, snr_db=args.snr, msg=MSG)
    else:
        frames, sent = synth.device_frames(nf, device, seed=1234 + rank, snr_db=args.snr)
    if args.variant != "rx_real":
        sys.stdout = json_out                  # (the side measurements print their line themselves)
        return side_measurement(args, eng, frames, world, rank, torch)
    # NBUF symbol buffers in rotation: the gather of step k (RCCL's own stream) overlaps the kernels of the steps behind it;
    # a buffer is rewritten only after the gather that read it has finished (work.wait() orders the
    # launch stream behind it without blocking the host).
    sym2 = [torch.empty(nf, dtype=torch.uint8, device=device) for _ in range(NBUF)]
    gathered2 = [torch.empty(world * nf, dtype=torch.uint8, device=device) for _ in range(NBUF)] if multi else None
    works = [None] * NBUF
    stream = torch.cuda.current_stream(device) if have_gpu else None
    # the group's launches go to a stream of their own (a NULL entry in uc_group_process_batch's stream list means "the
    # group's own stream", and torch's default stream IS the NULL stream: the HIP events must sit where the kernel runs)
    gstream = torch.cuda.Stream(device) if grp is not None else None

    def gather(b):
        if rehearse:  # gloo has no device all-gather: stage through the host (rehearsal only)
            host = [torch.empty(nf, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(host, sym2[b].cpu())
            gathered2[b].copy_(torch.cat(host))
            return None
        return dist.all_gather_into_tensor(gathered2[b], sym2[b], async_op=True)

    def step(k, e0=None, e1=None):
        b = k % NBUF
        if grp is not None:                   # decode into this rank's slice of gathered2[b] + in-place all-gather, all in C
            if e0 is not None:
                # the write-after-gather wait for the gather that last used this buffer goes in FRONT of e0 (the library
                # would enqueue the same wait behind it): [e0, e1] brackets the kernel alone, and what a step costs beyond
                # the kernel -- that wait included -- shows up in gather_ms_exposed, where it belongs
                if k >= NBUF:
                    grp.wait_gather(0, gathered2[b], gstream.cuda_stream)
                e0.record(gstream)
            grp.process([frames], world * nf, [gathered2[b]], streams=[gstream.cuda_stream])
            if e1 is not None:
                e1.record(gstream)
            return
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
        if grp is not None:
            grp.synchronize()
        for b in range(NBUF):
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
    wd.mark("warm-up")
    for k in range(args.warmup):
        step(k)
        wd.mark("warm-up", k)
    drain()
    sync()
    wd.mark("barrier in front of the timed region")
    if multi:
        dist.barrier()
    sync()

    # per-launch kernel time: HIP events on the stream the kernel is launched on
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if have_gpu else (None, None)
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, ev[k][0], ev[k][1])
        wd.mark("timed region: enqueued", k)
    wd.mark("timed region: waiting for the device (gathers included)", args.steps - 1)
    drain()
    sync()
    wd.mark("barrier behind the timed region", args.steps - 1)
    if multi:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    wd.mark("checks behind the timed region", args.steps - 1)
    # Sustained run behind the timed region (untimed, N = 1 only): the SMU's power figure is a moving average over about a
    # second, so the 40 ms of the timed region cannot show it; 1.5 s of back-to-back launches can.  Reports the socket power
    # and the SMU's clock over the last 0.5 s, and the rate the kernel holds meanwhile.
    power_rec = None
    if eng is not None and not multi and rank == 0 and args.sustain_s > 0:
        power = PowerSampler(torch, device)
        power.start()
        t_s = time.perf_counter()
        n_s = 0
        while time.perf_counter() - t_s < args.sustain_s:
            for _ in range(16):
                eng.process(frames, want_stats=False, symbols_out=sym2[0], stream=stream.cuda_stream)
            torch.cuda.synchronize()
            n_s += 16
        t_e = time.perf_counter()
        power_rec = power.stop(t_e - 0.5, t_e)
        if power_rec:
            power_rec["sustained_frames_per_s"] = nf * n_s / (t_e - t_s)
            power_rec["sustained_s"] = t_e - t_s
    elapsed_local = elapsed
    if multi:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    symbols = sym2[(args.steps - 1) % NBUF]
    if grp is not None:                       # the kernel wrote this rank's symbols straight into its slice of the stream
        symbols = gathered2[(args.steps - 1) % NBUF][rank * nf:(rank + 1) * nf]
        if eng is not None:                   # the checker: the same shard through a plain context (uc_process_batch)
            eng.process(frames, want_stats=False, symbols_out=sym2[0], stream=stream.cuda_stream)
            torch.cuda.synchronize()
            assert torch.equal(symbols, sym2[0]), "the group's slice differs from uc_process_batch on the same shard"
    gathered_host = None
    if multi:
        # Every rank holds the concatenation of all ranks' symbols, rank order: its own slice equals what it
        # decoded, and every rank's gathered buffer has the same digest (so the other slices are the owners').
        import hashlib
        g = gathered2[(args.steps - 1) % NBUF]
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
    # the clock the chip held under the kernel of the timed region, measured now, on this box (every rank: its own)
    clk_live = None
    if eng is not None:
        clk_live = live_clock(eng, lambda: eng.process(frames, want_stats=False, symbols_out=sym2[0], stream=stream.cuda_stream))
    local_ms = elapsed_local / args.steps * 1e3
    per_rank = None
    if multi:
        # every rank's own kernel time and step time (not only rank 0's, not only the maximum): lets a reader of the
        # N > 1 line tell the gather's cost from a straggling rank
        mine = torch.tensor([kern_ms if kern_ms is not None else 0.0, local_ms], dtype=torch.float64,
                            device="cpu" if rehearse else device)
        alls = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(alls, mine)
        per_rank = np.array([a.cpu().numpy() for a in alls])       # [world, (kernel_ms, ms_per_step)]
        # who ran where: the device every rank holds (name, uuid, PCI bus id, LOCAL_RANK), the peer-access row the HIP runtime
        # reports from it to every visible device, and the clock its chip held -- so that a slow or mis-mapped rank of the first
        # real 8-GPU run can be named from the line alone
        topo = {"rank": rank, "local_rank": local_rank, "pid": os.getpid()}
        if have_gpu:
            try:
                pr = torch.cuda.get_device_properties(device)
                topo.update({"device": pr.name, "uuid": str(getattr(pr, "uuid", "")), "pci_bus_id": getattr(pr, "pci_bus_id", None),
                             "multi_processor_count": pr.multi_processor_count,
                             "can_access_peer": [bool(d == device.index or torch.cuda.can_device_access_peer(device.index, d))
                                                 for d in range(torch.cuda.device_count())],
                             "shader_clock_GHz_under_the_kernel": clk_live})
            except Exception as ex:             # (diagnostics must never fail the run)
                topo["error"] = str(ex)[:200]
        topo_all = [None] * world
        dist.all_gather_object(topo_all, topo)

    gate_failures = []
    if rank == 0:
        total_frames = world * nf * args.steps
        value = total_frames / elapsed
        out = {
            "metric": "chirp frames/s (2048-pt FFT demod)", "value": value if have_gpu else None, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "ramp_ms": args.ramp_ms if have_gpu else 0.0, "ramp_launches": ramp_launches,
        }
        if multi:
            out["gather_backend"] = gather_backend
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
            traffic, traffic_source = None, None
            try:
                import glob
                import re
                # the band kernel's own passes are named rNN_vM_hbm_traffic.json (other kernels carry their name)
                tf = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json"))
                            if re.fullmatch(r"r\d+_v\d+_hbm_traffic\.json", os.path.basename(f)))[-1]
                traffic = json.load(open(tf))["hbm_bytes_per_frame"] * nf
                traffic_source = ("NOT measured in this run: FETCH_SIZE x 2 + WRITE_SIZE per frame of the committed "
                                  "rocprofv3 --pmc passes in %s, scaled to this launch's %d frames"
                                  % (os.path.relpath(tf, ROOT), nf))
            except Exception:
                traffic = None
            out["roofline"] = roofline("band_rx_real_f32", "band_kernel<rx_real,f32>", nf, BYTES_PER_FRAME, kern_ms,
                                       num_cus(torch, device), clk_live)
            out["roofline"]["bytes_per_frame"] = BYTES_PER_FRAME
            # socket power / cap / SMU clock over the last 0.5 s of a 1.5 s sustained run behind the timed region (null: multi-GPU
            # run, --sustain-s 0, or no hwmon access)
            out["roofline"]["power"] = power_rec
            if not multi and not args.no_live_traffic:
                t_c = time.perf_counter()
                lt = live_traffic()
                if lt:
                    traffic, traffic_source = lt["hbm_bytes_per_frame"] * nf, lt.pop("method")
                    lt["wall_s"] = time.perf_counter() - t_c
                    out["roofline"]["traffic_counters"] = lt
                    out["roofline"]["traffic_over_algorithmic"] = lt["hbm_bytes_per_frame"] / BYTES_PER_FRAME
            out["roofline"]["traffic"] = traffic
            out["roofline"]["traffic_source"] = traffic_source
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
            if good != len(texts):
                gate_failures.append("hello: %d of %d transmissions decode" % (good, len(texts)))
            out["value_per_gpu"] = value / world if have_gpu else None
            km, sm = per_rank[:, 0], per_rank[:, 1]
            out["per_rank"] = {"kernel_ms": {"min": float(km.min()), "median": float(np.median(km)), "max": float(km.max())},
                               "ms_per_step": {"min": float(sm.min()), "median": float(np.median(sm)), "max": float(sm.max())},
                               "kernel_ms_by_rank": [float(v) for v in km]}
            # what a step costs beyond the kernel on the slowest rank: the gather that the next kernel does not hide,
            # launch gaps, and (N > 1) waiting for the slowest rank inside the collective
            out["kernel_ms_covers"] = ("[e0, e1] on the group's launch stream around uc_group_process_batch ALONE: the "
                                       "write-after-gather wait for the buffer's previous gather is enqueued in FRONT of e0 "
                                       "(uc_group_wait_gather) since round 5 -- not comparable with kernel_ms of rounds 1-4, "
                                       "which included it; gather_ms_exposed = ms_per_step - kernel_ms carries it now")
            out["per_rank"]["roofline_frac_by_rank"] = ([float(nf * BYTES_PER_FRAME / (v * 1e-3) / 1e9 / HBM_PEAK_GBS) if v > 0 else None
                                                         for v in km] if have_gpu else None)
            out["per_rank"]["ranks"] = topo_all
            out["gather_ms_exposed"] = float((sm - km).max()) if have_gpu else None
            out["gather_ms_exposed_by_rank"] = [float(v) for v in (sm - km)] if have_gpu else None
            # the world size the C group's RCCL communicator reports (n_gpus above is torch.distributed's)
            out["rccl_world"] = int(grp.world) if grp is not None else None
            out["scaling_note"] = ("no 1 -> 8 curve has been measured by the builder (one-GPU boxes only): the first real "
                                   "N > 1 numbers are the driver's")
        else:
            # correctness figure of the measured run: decoded symbols vs transmitted bits.  ~23 % is EXPECTED here:
            # configs[1] runs the firmware's literal TIME_FRAME = 0.0205 s reference tables against frames that
            # sweep over the whole 26.2 ms frame; GPU == oracle is the gate (below), the matched sweep is configs[4]
            out["bit_error_rate_vs_transmitted"] = float((symbols != sent).float().mean().item())
            out["bit_error_rate_note"] = ("expected ~0.23: literal TIME_FRAME reference (SURVEY Q4) vs full-frame sweeps; "
                                          "the oracle has the same errors (symbols_equal_oracle_head4096_clear)")
        if not multi and have_gpu and not args.no_configs:
            cfgs = {}
            for name, fn in (("configs[2]", lambda: config2_iq(args, device, stream, torch, mag_mean)),
                             ("configs[3]", lambda: config3_stream(args, frames, device, torch))):
                t_c = time.perf_counter()
                try:
                    cfgs[name] = fn()
                except Exception as ex:                      # the contract line must not die with a side leg
                    cfgs[name] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
                    gate_failures.append("%s failed: %s" % (name, type(ex).__name__))
                cfgs[name]["wall_s"] = time.perf_counter() - t_c
            out["configs"] = cfgs
            c2, c3 = cfgs["configs[2]"], cfgs["configs[3]"]
            if not c2.get("baseband", {}).get("symbols_equal_oracle_head4096_clear", 1.0) >= 1.0:   # (NaN fails too)
                gate_failures.append("configs[2]: base-band symbols differ from the oracle on clear frames (or none is clear)")
            if c3.get("graph_equals_eager") is False:
                gate_failures.append("configs[3]: graph replay differs from the eager launch")
            if c3.get("head_rel_err_vs_oracle", 0.0) > 2e-5 or c3.get("head_peak_offsets_equal_oracle") is False:
                gate_failures.append("configs[3]: stream head differs from the oracle")
        if not multi and have_gpu and not args.no_hello1:
            t_c = time.perf_counter()
            try:
                h1 = hello_world1(args, device, torch, mag_mean)
                if h1["transmissions_decoded_exactly"] != h1["transmissions"] or not h1["gathered_equals_decoded"]:
                    gate_failures.append("hello_world1: %d of %d transmissions decode"
                                         % (h1["transmissions_decoded_exactly"], h1["transmissions"]))
                h1["over_configs1_value"] = h1["value"] / value
            except Exception as ex:
                h1 = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
                gate_failures.append("hello_world1 failed: %s" % type(ex).__name__)
            h1["wall_s"] = time.perf_counter() - t_c
            out["hello_world1"] = h1
            if "value" in h1:
                # The N > 1 lines of the scaling curve run configs[4] (K7 framing, matched sweep, gather every step); THIS line
                # runs configs[1] (no gather).  Like for like, the curve's N = 1 point is this one: value(N) / N / anchor.
                out["scale_anchor"] = {"workload": "configs[4] at world size 1 (the workload of the --gpus N > 1 lines)",
                                       "n_gpus": 1, "value": h1["value"], "unit": "frames/s", "ms_per_step": h1["ms_per_step"],
                                       "how": "UC_BENCH_HELLO=1 python bench.py prints this leg as its contract line"}
        if not multi and have_gpu and not args.no_receive:
            t_c = time.perf_counter()
            try:
                rx = receive_leg(args, device, torch)
                for key in ("sync_cplx_4096_streams", "sync_cplx_64_streams", "sync_cplx_1_stream"):
                    if rx[key]["streams_decoding_the_text"] != rx[key]["streams"]:
                        gate_failures.append("receive: %d of %d streams decode the text"
                                             % (rx[key]["streams_decoding_the_text"], rx[key]["streams"]))
                if rx["live_4096_streams"]["streams_whose_chunks_add_up_to_the_recorded_call"] != 4096:
                    gate_failures.append("receive: live chunks differ from the recorded-stream call on %d streams"
                                         % (4096 - rx["live_4096_streams"]["streams_whose_chunks_add_up_to_the_recorded_call"]))
            except Exception as ex:
                rx = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
                gate_failures.append("receive failed: %s" % type(ex).__name__)
            rx["wall_s"] = time.perf_counter() - t_c
            out["receive"] = rx
        if not multi and have_gpu and not args.no_cpu_baseline:
            cb = cpu_baseline(frames[: 1 << 19].cpu().numpy(), mag_mean)
            # the oracle as the checker: GPU symbols of the measured run vs float64 oracle
            head, clear = cb.pop("symbols_f64_oracle_head"), cb.pop("clear_head")
            got = symbols[:4096].cpu().numpy()
            out["symbols_equal_oracle_head4096"] = float((got == head).mean())
            out["symbols_equal_oracle_head4096_clear"] = float((got[clear] == head[clear]).mean())
            if not np.array_equal(got[clear], head[clear]):
                gate_failures.append("configs[1]: symbols differ from the float64 oracle on clear frames")
            out["cpu_baseline"] = cb
        out["gates_failed"] = gate_failures
        print(json.dumps(out), file=json_out, flush=True)
    if grp is not None:
        grp.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if gate_failures:
        sys.stderr.write("bench.py: correctness gate(s) failed: %s\n" % "; ".join(gate_failures))
        raise SystemExit(3)


if __name__ == "__main__":
    main()
