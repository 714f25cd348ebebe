"""bench_telemetry.py -- what bench.py measures WITH: the constants of the contract line, the synthetic batch, the CPU baseline
(the oracle, timed on the box's host cores), clock ramp and timed launches, the VALU table of the committed counter passes,
the in-run PMC traffic passes, socket power / SMU clock sampling, the live clock probe, the roofline record, the HBM ceiling
probes.  Split from bench.py in round 6 (it had grown to 1600 lines); bench.py re-exports everything here."""
import json
import os
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

