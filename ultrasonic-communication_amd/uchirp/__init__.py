"""uchirp -- thin ctypes binding of libuchirp.so (include/uchirp.h).

The compute path is the HIP library and nothing else: if libuchirp.so is
missing or no MI355X is visible, importing works but creating an Engine raises
(there is no CPU fallback; the CPU restatement under oracle/ is test
infrastructure and is never imported from here).

PyTorch is optional plumbing: device tensors are passed by data_ptr() and run
on torch's current HIP stream; numpy arrays go through the library's own
staging copies.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)  # ultrasonic-communication_amd/
LIB_PATH = os.environ.get("UCHIRP_LIB") or os.path.join(_ROOT, "libuchirp.so")  # UCHIRP_LIB: diagnostic builds

RX_REAL, SYNC_CPLX, COMPRESS, DECHIRP_DOWN, IQ, STREAM = range(6)
DOWN_CHIRP, UP_CHIRP = 0, 1
DTYPE_I32, DTYPE_F32, DTYPE_PDM = 0, 1, 2
PDM_SILENCE = 0xAAAAAAAA   # UC_PDM_SILENCE: the filter history of a stream that starts
SYM_DOWN, SYM_UP, SYM_NONE = 0, 1, 0xFF
FLAG_LIBM_TRIG, FLAG_TRUE_DC, FLAG_STREAM_UP, FLAG_IQ_BASEBAND, FLAG_NO_FRAME_PAIRS = 1, 2, 8, 16, 32
(TABLE_UP, TABLE_DOWN, TABLE_HANN, TABLE_H_UP, TABLE_H_DOWN, TABLE_CARRIER_C, TABLE_CARRIER_S,
 TABLE_FIR, TABLE_TEMPLATE) = range(9)

EXPORTS = ["uc_abi_version", "uc_last_error", "uc_default_config", "uc_create", "uc_destroy",
           "uc_process_frame", "uc_process_batch", "uc_stats_per_frame", "uc_iq_halo",
           "uc_get_table", "uc_get_windows", "uc_idx2freq", "uc_receive_stream", "uc_receive_stream_isr",
           "uc_stream_geometry", "uc_process_stream", "uc_dfsdm_sinc5", "uc_set_table", "uc_window_bins",
           "uc_window_spectrum",
           "uc_partition", "uc_frame_span", "uc_stream_span", "uc_group_unique_id", "uc_group_create", "uc_group_create_rank",
           "uc_group_destroy", "uc_group_world", "uc_group_local_count", "uc_group_first_rank", "uc_group_ctx",
           "uc_group_process_batch", "uc_group_wait_gather", "uc_group_synchronize",
           "uc_device_count", "uc_device_malloc", "uc_device_free", "uc_device_copy", "uc_clock_probe", "uc_clock_read", "uc_clock_stamps", "uc_receive_streams", "uc_debug_busy_counters",
           "uc_rx_state_create", "uc_rx_state_reset", "uc_rx_state_destroy", "uc_receive_streams_next",
           "uc_rx_state_streams", "uc_rx_state_keep_previous", "uc_group_receive_streams", "uc_group_receive_streams_next", "uc_group_process_stream",
           "uc_dfsdm_sinc5_streams", "uc_group_preflight"]
GROUP_ID_BYTES = 128


class Config(C.Structure):
    """struct uc_config (include/uchirp.h)."""
    _fields_ = [("n", C.c_uint32), ("fs", C.c_float), ("f0", C.c_float), ("f1", C.c_float),
                ("time_frame", C.c_float), ("phase_deg", C.c_float), ("snr_threshold", C.c_float),
                ("mag_mean", C.c_float), ("carrier", C.c_float), ("variant", C.c_int32),
                ("device", C.c_int32), ("flags", C.c_uint32), ("decim", C.c_uint32)]


class Clock(C.Structure):
    """struct uc_clock (include/uchirp.h)."""
    _fields_ = [("shader_ghz", C.c_double), ("wave_cycles", C.c_double), ("span_us", C.c_double), ("waves", C.c_uint32)]


RX_EVENT_DTYPE = np.dtype([("block", "<u4"), ("sync_position", "<u4"), ("state_before", "u1"), ("state_after", "u1"),
                           ("bit", "i1"), ("reserved", "u1"), ("snr_up", "<f4"), ("snr_down", "<f4")])
PEAK_DTYPE = np.dtype([("value", "<f4"), ("offset", "<u4")])  # struct uc_peak
STATE_IDLE, STATE_SYNCHRONIZING, STATE_SYNCHRONIZED, STATE_DATA_RECEIVING = range(4)

STATS_DTYPE = np.dtype([("mag_max", "<f4"), ("mag_max_left", "<f4"), ("mag_max_right", "<f4"),
                        ("max_freq", "<i4"), ("max_freq_left", "<i4"), ("max_freq_right", "<i4"),
                        ("mag_mean", "<f4"), ("snr", "<f4")])


class UchirpError(RuntimeError):
    pass


def build(force=False):
    """Compile libuchirp.so for gfx950 with hipcc (in-tree)."""
    if os.environ.get("UCHIRP_LIB"):          # a diagnostic build named by the caller: it is what it is
        return LIB_PATH
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _ROOT] + (["-B"] if force else []) + ["libuchirp.so"])
    else:
        subprocess.check_call(["make", "-C", _ROOT, "libuchirp.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    """Load libuchirp.so; raises UchirpError if it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own
    # libamdhip64; if libuchirp.so pulled in /opt/rocm's copy first, a later
    # `import torch` would initialise a second runtime and one of the two loses
    # the device.  Importing torch first makes both bind to the same runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise UchirpError("libuchirp.so not built: run `make -C %s` (hipcc, gfx950); "
                          "there is no CPU fallback" % _ROOT)
    L = C.CDLL(LIB_PATH)
    L.uc_abi_version.restype = C.c_int
    L.uc_last_error.restype = C.c_char_p
    L.uc_default_config.argtypes = [C.c_int32, C.POINTER(Config)]
    L.uc_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.uc_destroy.argtypes = [C.c_void_p]
    L.uc_destroy.restype = None
    L.uc_process_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    L.uc_process_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.uc_stats_per_frame.argtypes = [C.c_void_p]
    L.uc_iq_halo.argtypes = [C.c_void_p]
    L.uc_get_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.uc_get_windows.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint32)] * 3
    L.uc_idx2freq.argtypes = [C.c_void_p, C.c_uint32]
    L.uc_idx2freq.restype = C.c_int32
    L.uc_receive_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_char_p, C.c_size_t,
                                    C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.uc_receive_stream_isr.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_char_p, C.c_size_t,
                                        C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.uc_stream_geometry.argtypes = [C.c_void_p, C.c_size_t] + [C.POINTER(C.c_size_t)] * 4
    L.uc_process_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.uc_dfsdm_sinc5.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.uc_dfsdm_sinc5_streams.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_void_p]
    L.uc_set_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.uc_window_bins.argtypes = [C.c_void_p]
    L.uc_window_spectrum.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
    SZP = C.POINTER(C.c_size_t)
    L.uc_partition.argtypes = [C.c_size_t, C.c_int, C.c_int, SZP, SZP]
    L.uc_frame_span.argtypes = [C.c_uint32, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, SZP, SZP]
    L.uc_stream_span.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, SZP, SZP, SZP, SZP]
    L.uc_group_unique_id.argtypes = [C.c_void_p, C.c_size_t]
    L.uc_group_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_void_p)]
    L.uc_group_create_rank.argtypes = [C.POINTER(Config), C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.uc_group_destroy.argtypes = [C.c_void_p]
    L.uc_group_destroy.restype = None
    L.uc_group_preflight.argtypes = [C.POINTER(Config)]
    for fn in (L.uc_group_world, L.uc_group_local_count, L.uc_group_first_rank, L.uc_group_synchronize):
        fn.argtypes = [C.c_void_p]
    L.uc_group_ctx.argtypes = [C.c_void_p, C.c_int]
    L.uc_group_ctx.restype = C.c_void_p
    L.uc_group_process_batch.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_size_t,
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.uc_group_wait_gather.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.uc_device_malloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
    L.uc_device_free.argtypes = [C.c_int, C.c_void_p]
    L.uc_device_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.uc_receive_streams.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.uc_debug_busy_counters.argtypes = [C.c_void_p]
    L.uc_rx_state_create.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.uc_rx_state_reset.argtypes = [C.c_void_p, C.c_void_p]
    L.uc_rx_state_destroy.argtypes = [C.c_void_p]
    L.uc_rx_state_destroy.restype = None
    L.uc_rx_state_streams.argtypes = [C.c_void_p]
    if hasattr(L, "uc_rx_state_keep_previous"):     # (UCHIRP_LIB may name an older diagnostic build)
        L.uc_rx_state_keep_previous.argtypes = [C.c_void_p, C.c_int]
    L.uc_rx_state_streams.restype = C.c_size_t
    VPP = C.POINTER(C.c_void_p)
    L.uc_group_receive_streams.argtypes = [C.c_void_p, VPP, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, VPP, VPP, C.c_size_t,
                                           VPP, VPP]
    L.uc_group_process_stream.argtypes = [C.c_void_p, VPP, C.c_int, C.c_size_t, VPP, VPP, VPP]
    L.uc_group_receive_streams_next.argtypes = [C.c_void_p, VPP, VPP, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, VPP, VPP,
                                                C.c_size_t, VPP, VPP]
    L.uc_receive_streams_next.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p,
                                          C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.uc_clock_probe.argtypes = [C.c_void_p, C.c_int]
    L.uc_clock_read.argtypes = [C.c_void_p, C.POINTER(Clock)]
    L.uc_clock_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    _lib = L
    return L


def _check(rc, what):
    if rc < 0:
        msg = lib().uc_last_error()
        raise UchirpError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))
    return rc


def default_config(variant=RX_REAL, **over):
    cfg = Config()
    _check(lib().uc_default_config(variant, C.byref(cfg)), "uc_default_config")
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class Engine:
    """One uc_ctx: the receiver's DSP state for one variant on one MI355X."""

    def __init__(self, variant=RX_REAL, device=0, **over):
        self.cfg = default_config(variant, device=device, **over)
        h = C.c_void_p()
        _check(lib().uc_create(C.byref(self.cfg), C.byref(h)), "uc_create")
        self._h = h
        self.n = int(self.cfg.n)
        self.variant = variant
        self.device = device
        self.spf = lib().uc_stats_per_frame(h)
        self.halo = lib().uc_iq_halo(h)
        bw, bw2, ilz = C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib().uc_get_windows(h, C.byref(bw), C.byref(bw2), C.byref(ilz))
        self.bandwidth, self.bandwidth2, self.idx_left_zero = bw.value, bw2.value, ilz.value

    def close(self):
        if getattr(self, "_h", None):
            lib().uc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def busy_counters(self):
        """uc_debug_busy_counters: hand-out counter words left non-zero with nothing in flight (always 0)."""
        return _check(lib().uc_debug_busy_counters(self._h), "uc_debug_busy_counters")

    def clock_probe(self, on=True):
        """uc_clock_probe: the following launches run the clock-stamped twin of their kernel (never time with it on)."""
        _check(lib().uc_clock_probe(self._h, 1 if on else 0), "uc_clock_probe")

    def clock_read(self):
        """uc_clock_read -> dict(shader_ghz, wave_cycles, span_us, waves) of the last launch (waits for the device)."""
        c = Clock()
        _check(lib().uc_clock_read(self._h, C.byref(c)), "uc_clock_read")
        return {"shader_ghz": c.shader_ghz, "wave_cycles": c.wave_cycles, "span_us": c.span_us, "waves": int(c.waves)}

    def clock_stamps(self):
        """uc_clock_stamps -> uint64 [waves, 4]: cycles (| CU id << 40), 100 MHz ticks, start tick, end tick of every wave."""
        nw = _check(lib().uc_clock_stamps(self._h, None, 0), "uc_clock_stamps")
        a = np.zeros(nw, np.uint64)
        _check(lib().uc_clock_stamps(self._h, a.ctypes.data_as(C.c_void_p), nw), "uc_clock_stamps")
        return a.reshape(-1, 4)

    def table(self, tid):
        buf = np.zeros(4 * self.n, np.float32)
        cnt = _check(lib().uc_get_table(self._h, tid, buf.ctypes.data_as(C.c_void_p), buf.size),
                     "uc_get_table")
        return buf[:cnt].copy()

    def idx2freq(self, idx):
        return lib().uc_idx2freq(self._h, int(idx))

    def set_table(self, tid, data):
        """uc_set_table: replace TABLE_UP / TABLE_DOWN / TABLE_HANN (synchronous)."""
        a = np.ascontiguousarray(data, np.float32).reshape(-1)
        _check(lib().uc_set_table(self._h, tid, a.ctypes.data_as(C.c_void_p), a.size), "uc_set_table")

    def window_spectrum(self, frames, n_frames=None, stride=0):
        """uc_window_spectrum: |X[k]|, k = -bandwidth2 .. +bandwidth2, of every history of every frame.
        numpy in -> numpy [n_frames, spf, 2 bandwidth2 + 1] (synchronous); torch device tensor in -> torch tensor."""
        wb = _check(lib().uc_window_bins(self._h), "uc_window_bins")
        st = stride or self.n
        if _is_torch(frames):
            import torch
            t = frames
            if not t.is_contiguous() or t.dtype not in (torch.int32, torch.float32) or t.device.type != "cuda":
                raise ValueError("frames must be a contiguous int32 / float32 GPU tensor")
            dt = DTYPE_I32 if t.dtype == torch.int32 else DTYPE_F32
            if n_frames is None:
                n_frames = (t.numel() - self.n) // st + 1 if t.numel() >= self.n else 0
            if n_frames and (n_frames - 1) * st + self.n > t.numel():
                raise ValueError("frames tensor too small for %d frames" % n_frames)
            out = torch.empty((n_frames, self.spf, wb), dtype=torch.float32, device=t.device)
            _check(lib().uc_window_spectrum(self._h, C.c_void_p(t.data_ptr()), dt, n_frames, st, C.c_void_p(out.data_ptr()),
                                            C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)),
                   "uc_window_spectrum")
            return out
        a = np.ascontiguousarray(frames)
        if a.dtype not in (np.int32, np.float32):
            raise TypeError("frames must be int32 or float32")
        dt = DTYPE_I32 if a.dtype == np.int32 else DTYPE_F32
        flat = a.reshape(-1)
        if n_frames is None:
            n_frames = (flat.size - self.n) // st + 1 if flat.size >= self.n else 0
        if n_frames and (n_frames - 1) * st + self.n > flat.size:
            raise ValueError("frames buffer too small for %d frames" % n_frames)
        out = np.zeros((n_frames, self.spf, wb), np.float32)
        _check(lib().uc_window_spectrum(self._h, flat.ctypes.data_as(C.c_void_p), dt, n_frames, st,
                                        out.ctypes.data_as(C.c_void_p), None), "uc_window_spectrum")
        return out

    def receive(self, samples, busy=None):
        """uc_receive_stream / uc_receive_stream_isr: the receiver's main loop over a recorded stream.
        samples: numpy int32/float32 array or a torch int32/float32 device tensor.
        busy: optional per-block flags (the consumer had not finished when the block arrived: the ISR drops it).
        Returns (text, trace)."""
        if _is_torch(samples):
            import torch
            if not samples.is_contiguous():
                raise ValueError("samples tensor must be contiguous")
            if samples.dtype not in (torch.int32, torch.float32):
                raise TypeError("samples must be int32 or float32")
            if samples.device.type != "cuda":
                raise ValueError("torch samples must live on the GPU (numpy arrays take the host path)")
            dt = DTYPE_I32 if samples.dtype == torch.int32 else DTYPE_F32
            # the call copies on the null stream and blocks: whatever produced `samples` on torch's stream must be done
            torch.cuda.current_stream(samples.device).synchronize()
            ptr, count = C.c_void_p(samples.data_ptr()), samples.numel()
        else:
            a = np.ascontiguousarray(samples).reshape(-1)
            if a.dtype not in (np.int32, np.float32):
                raise TypeError("samples must be int32 or float32")
            dt = DTYPE_I32 if a.dtype == np.int32 else DTYPE_F32
            ptr, count = a.ctypes.data_as(C.c_void_p), a.size
        nb = count // self.n
        trace = np.zeros(max(nb, 1), RX_EVENT_DTYPE)
        text = C.create_string_buffer(4096)
        nt = C.c_size_t(0)
        bz = None
        if busy is not None:
            bz = np.ascontiguousarray(busy, np.uint8).reshape(-1)
            if bz.size != nb:
                raise ValueError("busy must hold one flag per %d-sample block" % self.n)
        nch = _check(lib().uc_receive_stream_isr(self._h, ptr, dt, count, bz.ctypes.data_as(C.c_void_p) if bz is not None else None,
                                                 text, 4096, trace.ctypes.data_as(C.c_void_p), nb, C.byref(nt)),
                     "uc_receive_stream")
        # (the returned count, not the C string: a decoded byte may be 0)
        return text.raw[:nch].decode("latin-1"), trace[:nt.value]

    def live(self, n_streams):
        """uc_rx_state_create: n_streams live receivers at power-on; feed them chunk after chunk with LiveStreams.next()."""
        return LiveStreams(self, n_streams)

    @staticmethod
    def _rows_of(t):
        """(n_streams, n_samples, stream stride in elements) of a 2-d GPU tensor whose rows are contiguous: a contiguous tensor, or
        the first columns of a wider one (a chunk inside a ring buffer: stream_stride_elems > n_samples)."""
        import torch
        if (t.dim() != 2 or t.dtype not in (torch.int32, torch.float32) or t.device.type != "cuda" or
                (t.shape[1] > 1 and t.stride(1) != 1) or (t.shape[0] > 1 and t.stride(0) < t.shape[1])):
            raise ValueError("samples must be a 2-d int32 / float32 GPU tensor with contiguous rows")
        ns, nsmp = int(t.shape[0]), int(t.shape[1])
        return ns, nsmp, (int(t.stride(0)) if ns > 1 and t.stride(0) != nsmp else 0)

    def receive_many(self, samples, busy=None, text_cap=64, want_trace=True, stream=None, _state=None, pdm=False):
        """uc_receive_streams: samples [n_streams, n_samples] (numpy int32 / float32, or a contiguous torch device tensor);
        busy [n_streams, n_samples // n] or None.  Returns (texts: list of str, traces: list of RX_EVENT_DTYPE arrays or
        None).  Host results either way (the call waits).  pdm=True: the words are the microphones' 1-bit PDM streams
        (uint32 / int32 bit patterns, one word per sample: UC_DTYPE_PDM)."""
        if _is_torch(samples):
            import torch
            t = samples
            ns, nsmp, sstride = self._rows_of(t)
            dt = DTYPE_I32 if t.dtype == torch.int32 else DTYPE_F32
            ptr = C.c_void_p(t.data_ptr())
            if stream is None:
                stream = torch.cuda.current_stream(t.device).cuda_stream
        else:
            a = np.ascontiguousarray(samples)
            if pdm and a.dtype == np.uint32:
                a = a.view(np.int32)
            if a.ndim != 2 or a.dtype not in (np.int32, np.float32):
                raise TypeError("samples must be a 2-d int32 / float32 array")
            dt = DTYPE_I32 if a.dtype == np.int32 else DTYPE_F32
            ns, nsmp = a.shape
            sstride = 0
            ptr = a.ctypes.data_as(C.c_void_p)
        if pdm:
            if dt != DTYPE_I32:
                raise TypeError("PDM words are 32-bit integers")
            dt = DTYPE_PDM
        nb = nsmp // self.n
        bz = None
        if busy is not None:
            bz = np.ascontiguousarray(busy, np.uint8)
            if bz.shape != (ns, nb):
                raise ValueError("busy must be [n_streams, n_samples // n]")
        text = np.zeros((ns, text_cap), np.uint8)
        ntext = np.zeros(ns, np.uint32)
        trace = np.zeros((ns, max(nb, 1)), RX_EVENT_DTYPE) if want_trace else None
        ntrace = np.zeros(ns, np.uint32)
        tail = (bz.ctypes.data_as(C.c_void_p) if bz is not None else None, text.ctypes.data_as(C.c_void_p), text_cap,
                ntext.ctypes.data_as(C.c_void_p), trace.ctypes.data_as(C.c_void_p) if trace is not None else None, max(nb, 1),
                ntrace.ctypes.data_as(C.c_void_p), C.c_void_p(stream) if stream else None)
        if _state is None:
            _check(lib().uc_receive_streams(self._h, ptr, dt, ns, nsmp, sstride, *tail), "uc_receive_streams")
        else:
            if ns != _state.n_streams:
                raise ValueError("this state holds %d streams" % _state.n_streams)
            _check(lib().uc_receive_streams_next(self._h, _state._h, ptr, dt, nsmp, sstride, *tail), "uc_receive_streams_next")
        texts = [bytes(text[i, :ntext[i]]).decode("latin-1") for i in range(ns)]
        traces = [trace[i, :ntrace[i]] for i in range(ns)] if want_trace else None
        return texts, traces

    def receive_many_into(self, samples, text, n_text, trace=None, n_trace=None, busy=None, stream=None, _state=None, pdm=False):
        """uc_receive_streams[_next] with every buffer on the device (contiguous torch tensors): asynchronous on `stream`,
        nothing is staged, nothing is copied back -- the form a live host calls block after block, and the one that can be
        captured into a hipGraph (after one eager call of the same shape has sized the scratch).
        samples [n_streams, k * n] int32 / float32; text uint8 [n_streams, text_cap]; n_text int32 [n_streams];
        trace (optional) uint8 [n_streams, trace_cap, 20] (view as RX_EVENT_DTYPE on the host); n_trace int32 [n_streams];
        busy (optional) uint8 [n_streams, k]."""
        import torch
        t = samples
        ns, nsmp, sstride = self._rows_of(t)
        for name, x in (("text", text), ("n_text", n_text), ("trace", trace), ("n_trace", n_trace), ("busy", busy)):
            if x is not None and (x.device.type != "cuda" or not x.is_contiguous()):
                raise ValueError("%s must be a contiguous GPU tensor" % name)
        dt = DTYPE_I32 if t.dtype == torch.int32 else DTYPE_F32
        if pdm:
            if dt != DTYPE_I32:
                raise TypeError("PDM words are 32-bit integers")
            dt = DTYPE_PDM
        if text.shape[0] != ns or n_text.numel() != ns or (busy is not None and tuple(busy.shape) != (ns, nsmp // self.n)):
            raise ValueError("output / busy shapes do not match %d streams" % ns)
        trace_cap = int(trace.shape[1]) if trace is not None else 0
        if trace is not None and (trace.shape[0] != ns or trace[0, 0].numel() * trace.element_size() != RX_EVENT_DTYPE.itemsize):
            raise ValueError("trace must be [n_streams, trace_cap] records of %d bytes" % RX_EVENT_DTYPE.itemsize)
        if stream is None:
            stream = torch.cuda.current_stream(t.device).cuda_stream
        p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        tail = (p(busy), p(text), int(text.shape[1]), p(n_text), p(trace), trace_cap, p(n_trace), C.c_void_p(stream) if stream else None)
        if _state is None:
            _check(lib().uc_receive_streams(self._h, p(t), dt, ns, nsmp, sstride, *tail), "uc_receive_streams")
        else:
            if ns != _state.n_streams:
                raise ValueError("this state holds %d streams" % _state.n_streams)
            _check(lib().uc_receive_streams_next(self._h, _state._h, p(t), dt, nsmp, sstride, *tail), "uc_receive_streams_next")

    def stream_geometry(self, n_samples):
        """uc_stream_geometry -> (halo, n_out, n_blocks, hop) for a buffer of n_samples (UC_STREAM)."""
        v = [C.c_size_t() for _ in range(4)]
        _check(lib().uc_stream_geometry(self._h, int(n_samples), *[C.byref(x) for x in v]), "uc_stream_geometry")
        return tuple(x.value for x in v)

    def process_stream(self, samples, want_compressed=True, want_peaks=True, compressed_out=None, peaks_out=None,
                       stream=None):
        """uc_process_stream: FIR-decimate front-end + overlap-save compression of one buffer whose first
        `halo` samples are history.  numpy in -> numpy out (synchronous); torch device tensor in -> torch
        tensors out (compressed float32[n_out], peaks int32 view [n_blocks, 2]), asynchronous on `stream`."""
        if _is_torch(samples):
            import torch
            t = samples
            if not t.is_contiguous():
                raise ValueError("samples tensor must be contiguous")
            if t.dtype not in (torch.int32, torch.float32):
                raise TypeError("samples must be int32 or float32")
            if t.device.type != "cuda":
                raise ValueError("torch samples must live on the GPU (numpy arrays take the host path)")
            dt = DTYPE_I32 if t.dtype == torch.int32 else DTYPE_F32
            _, n_out, n_blocks, _ = self.stream_geometry(t.numel())
            comp = compressed_out
            if comp is None and want_compressed:
                comp = torch.empty(n_out, dtype=torch.float32, device=t.device)
            pk = peaks_out
            if pk is None and want_peaks:
                pk = torch.empty((n_blocks, 2), dtype=torch.int32, device=t.device)
            if stream is None:
                stream = torch.cuda.current_stream(t.device).cuda_stream
            _check(lib().uc_process_stream(self._h, C.c_void_p(t.data_ptr()), dt, t.numel(),
                                           C.c_void_p(comp.data_ptr()) if comp is not None else None,
                                           C.c_void_p(pk.data_ptr()) if pk is not None else None,
                                           C.c_void_p(stream)), "uc_process_stream")
            return comp, pk
        a = np.ascontiguousarray(samples).reshape(-1)
        if a.dtype not in (np.int32, np.float32):
            raise TypeError("samples must be int32 or float32")
        dt = DTYPE_I32 if a.dtype == np.int32 else DTYPE_F32
        _, n_out, n_blocks, _ = self.stream_geometry(a.size)
        comp = np.zeros(n_out, np.float32) if want_compressed else None
        pk = np.zeros(n_blocks, PEAK_DTYPE) if want_peaks else None
        _check(lib().uc_process_stream(self._h, a.ctypes.data_as(C.c_void_p), dt, a.size,
                                       comp.ctypes.data_as(C.c_void_p) if comp is not None else None,
                                       pk.ctypes.data_as(C.c_void_p) if pk is not None else None, None),
               "uc_process_stream")
        return comp, pk

    def dfsdm(self, pdm_words, out=None, stream=None):
        """uc_dfsdm_sinc5: packed 1-bit PDM words (first 4 = history) -> int32 DFSDM words (n - 4).
        numpy uint32 in -> numpy int32 out; torch int32 device tensor in -> torch int32 out (asynchronous)."""
        if _is_torch(pdm_words):
            import torch
            t = pdm_words
            if not t.is_contiguous() or t.dtype != torch.int32 or t.device.type != "cuda":
                raise ValueError("pdm_words must be a contiguous int32 GPU tensor (bit pattern of the uint32 words)")
            n = t.numel()
            if out is None:
                out = torch.empty(max(n - 4, 0), dtype=torch.int32, device=t.device)
            if stream is None:
                stream = torch.cuda.current_stream(t.device).cuda_stream
            _check(lib().uc_dfsdm_sinc5(self._h, C.c_void_p(t.data_ptr()), n, C.c_void_p(out.data_ptr()),
                                        C.c_void_p(stream)), "uc_dfsdm_sinc5")
            return out
        a = np.ascontiguousarray(pdm_words, np.uint32).reshape(-1)
        res = np.zeros(max(a.size - 4, 0), np.int32)
        _check(lib().uc_dfsdm_sinc5(self._h, a.ctypes.data_as(C.c_void_p), a.size, res.ctypes.data_as(C.c_void_p),
                                    None), "uc_dfsdm_sinc5")
        return res

    def dfsdm_streams(self, pdm_words, history, out=None, stream=None):
        """uc_dfsdm_sinc5_streams: pdm_words [n_streams, n_words] NEW words of every microphone, history [n_streams, 4] (in and
        out: updated in place) -> int32 DFSDM words [n_streams, n_words].  numpy uint32 arrays (synchronous) or contiguous
        torch int32 device tensors (asynchronous on `stream`)."""
        if _is_torch(pdm_words):
            import torch
            t, h = pdm_words, history
            for x in (t, h):
                if not x.is_contiguous() or x.dtype != torch.int32 or x.device.type != "cuda" or x.dim() != 2:
                    raise ValueError("pdm_words / history must be contiguous 2-d int32 GPU tensors (bit patterns of the words)")
            ns, nw = int(t.shape[0]), int(t.shape[1])
            if tuple(h.shape) != (ns, 4):
                raise ValueError("history must be [n_streams, 4]")
            if out is None:
                out = torch.empty((ns, nw), dtype=torch.int32, device=t.device)
            if stream is None:
                stream = torch.cuda.current_stream(t.device).cuda_stream
            _check(lib().uc_dfsdm_sinc5_streams(self._h, C.c_void_p(t.data_ptr()), ns, nw, 0, C.c_void_p(h.data_ptr()),
                                                C.c_void_p(out.data_ptr()), 0, C.c_void_p(stream)), "uc_dfsdm_sinc5_streams")
            return out
        a = np.ascontiguousarray(pdm_words, np.uint32)
        if a.ndim != 2 or history.dtype != np.uint32 or history.shape != (a.shape[0], 4) or not history.flags.c_contiguous:
            raise ValueError("pdm_words [n_streams, n_words], history uint32 [n_streams, 4] (contiguous: updated in place)")
        res = np.zeros(a.shape, np.int32)
        _check(lib().uc_dfsdm_sinc5_streams(self._h, a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1], 0,
                                            history.ctypes.data_as(C.c_void_p), res.ctypes.data_as(C.c_void_p), 0, None),
               "uc_dfsdm_sinc5_streams")
        return res

    def process_frame(self, pcm, mag_mean=1.0):
        """uc_process_frame: n int32 DFSDM words -> (symbol, stats[spf])."""
        pcm = np.ascontiguousarray(pcm, np.int32)
        if pcm.size != self.n:
            raise ValueError("frame must hold %d samples" % self.n)
        sym = C.c_uint8(SYM_NONE)
        st = np.zeros(2, STATS_DTYPE)
        _check(lib().uc_process_frame(self._h, pcm.ctypes.data_as(C.c_void_p), float(mag_mean),
                                      C.byref(sym), st.ctypes.data_as(C.c_void_p)), "uc_process_frame")
        return sym.value, st[:self.spf]

    def process(self, frames, n_frames=None, stride=0, mag_mean=None, want_symbols=True,
                want_stats=True, symbols_out=None, stats_out=None, stream=None, halo=None):
        """uc_process_batch.

        frames: numpy int32/float32 array (host path, synchronous) or a torch
        device tensor (device path, asynchronous on `stream` / torch's current
        stream; outputs are torch tensors on the same device).
        """
        halo = self.halo if halo is None else halo
        st = stride or self.n
        if _is_torch(frames):
            import torch
            t = frames
            if not t.is_contiguous():
                raise ValueError("frames tensor must be contiguous")
            if t.dtype == torch.int32:
                dt = DTYPE_I32
            elif t.dtype == torch.float32:
                dt = DTYPE_F32
            else:
                raise TypeError("frames must be int32 or float32")
            total = t.numel()
            if n_frames is None:
                n_frames = (total - halo - self.n) // st + 1 if total >= self.n + halo else 0
            if n_frames and halo + (n_frames - 1) * st + self.n > total:
                raise ValueError("frames tensor too small for %d frames" % n_frames)
            dev = t.device
            if dev.type != "cuda":
                raise ValueError("torch frames must live on the GPU (numpy arrays take the host path)")
            sym = symbols_out
            if sym is None and want_symbols:
                sym = torch.empty(n_frames, dtype=torch.uint8, device=dev)
            stt = stats_out
            if stt is None and want_stats:
                stt = torch.empty((n_frames, self.spf, 8), dtype=torch.float32, device=dev)
            # the kernels write n_frames records: a caller's buffer that is too small would be overrun on the device
            if sym is not None and (sym.dtype != torch.uint8 or sym.numel() < n_frames or not sym.is_contiguous()
                                    or sym.device != dev):
                raise ValueError("symbols_out must be a contiguous uint8 tensor of >= %d elements on %s" % (n_frames, dev))
            if stt is not None and (stt.dtype != torch.float32 or stt.numel() < n_frames * self.spf * 8
                                    or not stt.is_contiguous() or stt.device != dev):
                raise ValueError("stats_out must be a contiguous float32 tensor of >= %d x %d x 8 elements on %s"
                                 % (n_frames, self.spf, dev))
            mm_ptr = None
            if mag_mean is not None:
                mm = mag_mean.to(device=dev, dtype=torch.float32).contiguous()
                if mm.numel() != 2 * n_frames:
                    raise ValueError("mag_mean must hold 2 floats per frame")
                mm_ptr = C.c_void_p(mm.data_ptr())
                # `mm` may be a temporary and the launch is asynchronous on a stream torch's allocator may not know
                # about: keep it alive until the next call of this engine
                self._mm_keep = mm
            if stream is None:
                stream = torch.cuda.current_stream(dev).cuda_stream
            _check(lib().uc_process_batch(self._h, C.c_void_p(t.data_ptr() + 4 * halo), dt, n_frames, st,
                                          mm_ptr,
                                          C.c_void_p(sym.data_ptr()) if sym is not None else None,
                                          C.c_void_p(stt.data_ptr()) if stt is not None else None,
                                          C.c_void_p(stream)), "uc_process_batch")
            return sym, stt
        a = np.ascontiguousarray(frames)
        if a.dtype == np.int32:
            dt = DTYPE_I32
        elif a.dtype == np.float32:
            dt = DTYPE_F32
        else:
            raise TypeError("frames must be int32 or float32")
        flat = a.reshape(-1)
        if n_frames is None:
            n_frames = (flat.size - halo - self.n) // st + 1 if flat.size >= self.n + halo else 0
        if n_frames and halo + (n_frames - 1) * st + self.n > flat.size:
            raise ValueError("frames buffer too small for %d frames" % n_frames)
        sym = np.full(n_frames, SYM_NONE, np.uint8) if want_symbols else None
        stt = np.zeros((n_frames, self.spf), STATS_DTYPE) if want_stats else None
        mm = None
        if mag_mean is not None:
            mm = np.ascontiguousarray(mag_mean, np.float32).reshape(n_frames, 2)
        _check(lib().uc_process_batch(self._h, C.c_void_p(flat.ctypes.data + 4 * halo), dt, n_frames, st,
                                      mm.ctypes.data_as(C.c_void_p) if mm is not None else None,
                                      sym.ctypes.data_as(C.c_void_p) if sym is not None else None,
                                      stt.ctypes.data_as(C.c_void_p) if stt is not None else None,
                                      None), "uc_process_batch")
        return sym, stt


def peaks_from_tensor(t):
    """View a (n_blocks, 2) int32 torch peaks tensor as a numpy PEAK_DTYPE array."""
    return t.detach().cpu().numpy().view(PEAK_DTYPE).reshape(-1)


def stats_from_tensor(t):
    """View a (n_frames, spf, 8) float32 torch stats tensor as a numpy STATS_DTYPE array."""
    a = t.detach().cpu().numpy()
    return a.view(STATS_DTYPE).reshape(a.shape[0], a.shape[1])


def partition(n_units, world, rank):
    """uc_partition -> (first, count): the contiguous block partition (pure arithmetic, no GPU)."""
    a, b = C.c_size_t(), C.c_size_t()
    _check(lib().uc_partition(int(n_units), int(world), int(rank), C.byref(a), C.byref(b)), "uc_partition")
    return a.value, b.value


def frame_span(n, stride, halo, first_frame, count):
    """uc_frame_span -> (first_elem, n_elems) of the sample buffer a shard must hold."""
    a, b = C.c_size_t(), C.c_size_t()
    _check(lib().uc_frame_span(int(n), int(stride), int(halo), int(first_frame), int(count), C.byref(a), C.byref(b)),
           "uc_frame_span")
    return a.value, b.value


class Group:
    """One uc_group (include/uchirp.h): the frame-sharded multi-GPU leg -- a uc_ctx per device, an RCCL communicator called
    from C, the symbol stream all-gathered in place on a side stream.
      Group(variant, devices=[0, 1, ...])                         one process drives several GPUs
      Group(variant, world=W, rank=r, unique_id=id, device=d)     one process per GPU; id = Group.unique_id() of rank 0"""

    def __init__(self, variant=RX_REAL, devices=None, world=None, rank=None, unique_id=None, device=0, **over):
        h = C.c_void_p()
        if devices is not None:
            self.cfg = default_config(variant, device=int(devices[0]), **over)
            arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            _check(lib().uc_group_create(C.byref(self.cfg), arr, len(devices), C.byref(h)), "uc_group_create")
        else:
            if unique_id is None or len(unique_id) != GROUP_ID_BYTES:
                raise ValueError("unique_id must be the %d bytes of Group.unique_id()" % GROUP_ID_BYTES)
            self.cfg = default_config(variant, device=int(device), **over)
            _check(lib().uc_group_create_rank(C.byref(self.cfg), C.c_char_p(bytes(unique_id)), int(world), int(rank),
                                              C.byref(h)), "uc_group_create_rank")
        self._h = h
        self._states = []
        self.world = lib().uc_group_world(h)
        self.n_local = lib().uc_group_local_count(h)
        self.first_rank = lib().uc_group_first_rank(h)
        self.n = int(self.cfg.n)

    @staticmethod
    def preflight(variant=RX_REAL, device=0, **over):
        """uc_group_preflight: everything uc_group_create_rank does on this rank except the communicator's rendezvous; raises
        UchirpError where the real call would fail before it.  Agree on the result across ranks BEFORE anybody builds a group."""
        cfg = default_config(variant, device=int(device), **over)
        _check(lib().uc_group_preflight(C.byref(cfg)), "uc_group_preflight")

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(GROUP_ID_BYTES)
        _check(lib().uc_group_unique_id(buf, GROUP_ID_BYTES), "uc_group_unique_id")
        return buf.raw

    def close(self):
        for st in list(getattr(self, "_states", [])):     # states first: a state must be destroyed before its context
            st.close()
        if getattr(self, "_h", None):
            lib().uc_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _ptr(x):
        if x is None:
            return None
        if _is_torch(x):
            return x.data_ptr()
        if isinstance(x, np.ndarray):
            return x.ctypes.data
        return int(x)

    def process(self, frames, n_frames_total, gathered, stride=0, streams=None, dtype=DTYPE_F32):
        """uc_group_process_batch.  frames / gathered: one tensor / array / address per LOCAL device (a shard's first
        frame; the n_frames_total-byte buffer for the whole symbol stream); streams: HIP stream handles or None."""
        nl = self.n_local
        if len(frames) != nl or len(gathered) != nl:
            raise ValueError("one frames / gathered entry per local device (%d)" % nl)
        fa = (C.c_void_p * nl)(*[self._ptr(f) for f in frames])
        ga = (C.c_void_p * nl)(*[self._ptr(x) for x in gathered])
        sa = None
        if streams is not None:
            sa = (C.c_void_p * nl)(*[int(s) if s else None for s in streams])
        _check(lib().uc_group_process_batch(self._h, fa, dtype, int(n_frames_total), int(stride), ga, sa),
               "uc_group_process_batch")

    def receive_streams(self, samples, n_streams_total, n_samples, text, text_cap, n_text=None, busy=None, stride=0,
                        streams=None, dtype=DTYPE_F32, states=None):
        """uc_group_receive_streams (states=None) / uc_group_receive_streams_next (states = one rx-state handle per local
        device): samples / text / n_text / busy: one tensor / array / address per LOCAL device (the share's first stream;
        the gathered n_streams_total x text_cap bytes; the gathered n_streams_total uint32)."""
        nl = self.n_local
        if len(samples) != nl or len(text) != nl:
            raise ValueError("one samples / text entry per local device (%d)" % nl)
        arr = lambda xs: (C.c_void_p * nl)(*[self._ptr(x) for x in xs]) if xs is not None else None
        sa = None
        if streams is not None:
            sa = (C.c_void_p * nl)(*[int(s) if s else None for s in streams])
        if states is None:
            _check(lib().uc_group_receive_streams(self._h, arr(samples), dtype, int(n_streams_total), int(n_samples), int(stride),
                                                  arr(busy), arr(text), int(text_cap), arr(n_text), sa),
                   "uc_group_receive_streams")
        else:
            st = (C.c_void_p * nl)(*[x._h if hasattr(x, "_h") else x for x in states])
            _check(lib().uc_group_receive_streams_next(self._h, st, arr(samples), dtype, int(n_streams_total), int(n_samples),
                                                       int(stride), arr(busy), arr(text), int(text_cap), arr(n_text), sa),
                   "uc_group_receive_streams_next")

    def process_stream(self, samples, n_samples_total, peaks, compressed=None, streams=None, dtype=DTYPE_F32):
        """uc_group_process_stream: samples / peaks / compressed: one tensor / array / address per LOCAL device (the shard
        uc_stream_span names; the gathered n_blocks x 8 bytes of peak records; the rank's own outputs or None)."""
        nl = self.n_local
        if len(samples) != nl or len(peaks) != nl:
            raise ValueError("one samples / peaks entry per local device (%d)" % nl)
        arr = lambda xs: (C.c_void_p * nl)(*[self._ptr(x) for x in xs]) if xs is not None else None
        sa = None
        if streams is not None:
            sa = (C.c_void_p * nl)(*[int(s) if s else None for s in streams])
        _check(lib().uc_group_process_stream(self._h, arr(samples), dtype, int(n_samples_total), arr(compressed), arr(peaks), sa),
               "uc_group_process_stream")

    def stream_span(self, n_samples_total, rank):
        """uc_stream_span on the group's first context -> (first_sample, n_shard, first_out, n_out) of `rank`."""
        v = [C.c_size_t() for _ in range(4)]
        _check(lib().uc_stream_span(C.c_void_p(lib().uc_group_ctx(self._h, 0)), int(n_samples_total), self.world, int(rank),
                                    *[C.byref(x) for x in v]), "uc_stream_span")
        return tuple(x.value for x in v)

    def rx_state(self, local, n_streams):
        """uc_rx_state_create on the context of local device `local` -> a GroupRxState (close it, or rx_state_destroy it,
        BEFORE the group goes: a state must not outlive its context; the object keeps the group alive until then)."""
        h = C.c_void_p()
        _check(lib().uc_rx_state_create(C.c_void_p(lib().uc_group_ctx(self._h, int(local))), int(n_streams), C.byref(h)),
               "uc_rx_state_create")
        st = GroupRxState(self, h)
        self._states.append(st)
        return st

    @staticmethod
    def rx_state_destroy(st):
        if isinstance(st, GroupRxState):
            st.close()
        else:
            lib().uc_rx_state_destroy(st)

    def wait_gather(self, local, gathered, stream):
        _check(lib().uc_group_wait_gather(self._h, int(local), C.c_void_p(self._ptr(gathered)), C.c_void_p(int(stream) if stream else None)),
               "uc_group_wait_gather")

    def synchronize(self):
        _check(lib().uc_group_synchronize(self._h), "uc_group_synchronize")


class GroupRxState:
    """A uc_rx_state made on a context of a Group (Group.rx_state): holds the group, is closed with it at the latest."""

    def __init__(self, group, handle):
        self.group, self._h = group, handle

    def close(self):
        if getattr(self, "_h", None):
            lib().uc_rx_state_destroy(self._h)
            self._h = None
            if self in self.group._states:
                self.group._states.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LiveStreams:
    """uc_rx_state: n_streams receivers between calls (the FIFO's tail and main()'s locals of every stream, on the device)."""

    def __init__(self, engine, n_streams):
        self.engine, self.n_streams = engine, int(n_streams)
        h = C.c_void_p()
        _check(lib().uc_rx_state_create(engine._h, self.n_streams, C.byref(h)), "uc_rx_state_create")
        self._h = h

    def next(self, samples, busy=None, text_cap=64, want_trace=True, stream=None, pdm=False):
        """uc_receive_streams_next: the next whole blocks of every stream, [n_streams, k * n] -> (texts, traces) of this chunk."""
        return self.engine.receive_many(samples, busy=busy, text_cap=text_cap, want_trace=want_trace, stream=stream, _state=self,
                                        pdm=pdm)

    def next_into(self, samples, text, n_text, trace=None, n_trace=None, busy=None, stream=None, pdm=False):
        """The same with every buffer on the device, asynchronous, capturable (Engine.receive_many_into)."""
        return self.engine.receive_many_into(samples, text, n_text, trace=trace, n_trace=n_trace, busy=busy, stream=stream,
                                             _state=self, pdm=pdm)

    def reset(self, stream=None):
        _check(lib().uc_rx_state_reset(self._h, C.c_void_p(stream) if stream else None), "uc_rx_state_reset")

    def keep_previous(self, on=True):
        """uc_rx_state_keep_previous: the caller promises to leave the samples of every call where they are, unchanged, until the
        next call on this state has completed (a ring of >= 2 chunk buffers): nothing is copied into the state."""
        _check(lib().uc_rx_state_keep_previous(self._h, 1 if on else 0), "uc_rx_state_keep_previous")

    def close(self):
        if getattr(self, "_h", None):
            lib().uc_rx_state_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
