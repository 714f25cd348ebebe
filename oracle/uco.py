"""ctypes binding of the CPU oracle (oracle/libuc_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("UCO_LIB") or os.path.join(_HERE, "libuc_oracle.so")  # UCO_LIB: an instrumented build (tools/sanitize.sh)

RX_REAL, SYNC_CPLX, COMPRESS, DECHIRP_DOWN, IQ, STREAM = range(6)
DTYPE_I32, DTYPE_F32 = 0, 1
F32, F64 = 32, 64
FLAG_LIBM_TRIG, FLAG_TRUE_DC, FLAG_STREAM_UP, FLAG_IQ_BASEBAND = 1, 2, 8, 16
TABLE_UP, TABLE_DOWN, TABLE_HANN, TABLE_H_UP, TABLE_H_DOWN, TABLE_CARRIER_C, TABLE_CARRIER_S, TABLE_FIR,\
    TABLE_TEMPLATE = range(9)


class Config(C.Structure):
    _fields_ = [("n", C.c_uint32), ("fs", C.c_float), ("f0", C.c_float), ("f1", C.c_float),
                ("time_frame", C.c_float), ("phase_deg", C.c_float), ("snr_threshold", C.c_float),
                ("mag_mean", C.c_float), ("carrier", C.c_float), ("variant", C.c_int32),
                ("device", C.c_int32), ("flags", C.c_uint32), ("decim", C.c_uint32)]


PEAK_DTYPE = np.dtype([("value", "<f4"), ("offset", "<u4")])

RX_EVENT_DTYPE = np.dtype([("block", "<u4"), ("sync_position", "<u4"), ("state_before", "u1"), ("state_after", "u1"),
                           ("bit", "i1"), ("reserved", "u1"), ("snr_up", "<f4"), ("snr_down", "<f4")])
assert RX_EVENT_DTYPE.itemsize == 20

STATS_DTYPE = np.dtype([("mag_max", "<f4"), ("mag_max_left", "<f4"), ("mag_max_right", "<f4"),
                        ("max_freq", "<i4"), ("max_freq_left", "<i4"), ("max_freq_right", "<i4"),
                        ("mag_mean", "<f4"), ("snr", "<f4")])
assert STATS_DTYPE.itemsize == 32


def build(force=False):
    if os.environ.get("UCO_LIB"):
        return _LIB
    src = [os.path.join(_HERE, f) for f in ("uc_oracle.c", "uc_oracle.h")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in src)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-B", "libuc_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.uco_default_config.argtypes = [C.c_int32, C.POINTER(Config)]
        L.uco_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        L.uco_destroy.argtypes = [C.c_void_p]
        L.uco_destroy.restype = None
        L.uco_process_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.uco_receive_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t,
                                         C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.uco_receive_stream_isr.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_char_p,
                                             C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.uco_receive_stream_diag.argtypes = L.uco_receive_stream_isr.argtypes + [C.c_void_p]
        L.uco_stream_geometry.argtypes = [C.c_void_p, C.c_size_t] + [C.POINTER(C.c_size_t)] * 4
        L.uco_process_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        L.uco_dfsdm_sinc5.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.uco_pdm_modulate.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.uco_spectrum.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.uco_stats_per_frame.argtypes = [C.c_void_p]
        L.uco_get_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.uco_set_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.uco_get_windows.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint32)] * 3
        L.uco_idx2freq.argtypes = [C.c_void_p, C.c_uint32]
        L.uco_idx2freq.restype = C.c_int32
        L.uco_arm_cos_f32.argtypes = [C.c_float]
        L.uco_arm_cos_f32.restype = C.c_float
        L.uco_arm_sin_cos_f32.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.uco_arm_sin_cos_f32.restype = None
        L.uco_arm_max_f32.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        L.uco_arm_max_f32.restype = None
        L.uco_rfft_fast_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.uco_rfft_fast_f32.restype = None
        L.uco_hann_periodic.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.uco_hann_periodic.restype = None
        _lib = L
    return _lib


def default_config(variant, **over):
    cfg = Config()
    rc = lib().uco_default_config(variant, C.byref(cfg))
    if rc:
        raise ValueError("uco_default_config rc=%d" % rc)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """CPU restatement of the reference path for one variant/config."""

    def __init__(self, variant=RX_REAL, **over):
        self.cfg = default_config(variant, **over)
        h = C.c_void_p()
        rc = lib().uco_create(C.byref(self.cfg), C.byref(h))
        if rc:
            raise ValueError("uco_create rc=%d" % rc)
        self._h = h
        self.n = self.cfg.n
        self.spf = lib().uco_stats_per_frame(h)
        bw, bw2, ilz = C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib().uco_get_windows(h, C.byref(bw), C.byref(bw2), C.byref(ilz))
        self.bandwidth, self.bandwidth2, self.idx_left_zero = bw.value, bw2.value, ilz.value

    def close(self):
        if self._h:
            lib().uco_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def table(self, tid):
        buf = np.zeros(4 * self.n, np.float32)
        cnt = lib().uco_get_table(self._h, tid, _ptr(buf), buf.size)
        if cnt < 0:
            raise ValueError("uco_get_table rc=%d" % cnt)
        return buf[:cnt].copy()

    def idx2freq(self, idx):
        return lib().uco_idx2freq(self._h, int(idx))

    def set_table(self, tid, data):
        a = np.ascontiguousarray(data, np.float32).reshape(-1)
        rc = lib().uco_set_table(self._h, tid, _ptr(a), a.size)
        if rc:
            raise ValueError("uco_set_table rc=%d" % rc)

    def process(self, frames, n_frames=None, stride=0, mag_mean=None, precision=F64, threads=0,
                halo=0):
        """frames: 1-D or 2-D contiguous int32/float32 array.  Returns (symbols, stats)."""
        a = np.ascontiguousarray(frames)
        if a.dtype == np.int32:
            dt = DTYPE_I32
        elif a.dtype == np.float32:
            dt = DTYPE_F32
        else:
            raise TypeError("frames must be int32 or float32")
        flat = a.reshape(-1)
        st = stride or self.n
        if n_frames is None:
            n_frames = (flat.size - halo - self.n) // st + 1 if flat.size >= self.n + halo else 0
        if n_frames and halo + (n_frames - 1) * st + self.n > flat.size:
            raise ValueError("frames buffer too small")
        sym = np.empty(n_frames, np.uint8)
        stats = np.zeros((n_frames, self.spf), STATS_DTYPE)
        mm = None
        if mag_mean is not None:
            mm = np.ascontiguousarray(mag_mean, np.float32).reshape(n_frames, 2)
        base = flat.ctypes.data + 4 * halo
        rc = lib().uco_process_batch(self._h, C.c_void_p(base), dt, n_frames, st,
                                     _ptr(mm) if mm is not None else None, _ptr(sym), _ptr(stats),
                                     precision, threads)
        if rc:
            raise RuntimeError("uco_process_batch rc=%d" % rc)
        return sym, stats

    def receive(self, samples, precision=F64, busy=None, margins=False):
        """The receiver's main loop over a recorded stream -> (text, trace[RX_EVENT_DTYPE]).
        busy: optional per-block flags: the ISR drops those blocks (main.c:661).
        margins=True: -> (text, trace, margin float32[len(trace)]): per processed block the relative gap of the block's closest
        decision (uco_receive_stream_diag); 1e30 where the block decided nothing."""
        a = np.ascontiguousarray(samples).reshape(-1)
        if a.dtype not in (np.int32, np.float32):
            raise TypeError("samples must be int32 or float32")
        dt = DTYPE_I32 if a.dtype == np.int32 else DTYPE_F32
        nb = a.size // self.n
        trace = np.zeros(nb, RX_EVENT_DTYPE)
        text = C.create_string_buffer(4096)
        nt = C.c_size_t(0)
        bz = None if busy is None else np.ascontiguousarray(busy, np.uint8).reshape(-1)
        mg = np.full(max(nb, 1), 1e30, np.float32)
        rc = lib().uco_receive_stream_diag(self._h, _ptr(a), dt, a.size, _ptr(bz) if bz is not None else None, precision, text,
                                           4096, _ptr(trace), nb, C.byref(nt), _ptr(mg))
        if rc < 0:
            raise RuntimeError("uco_receive_stream rc=%d" % rc)
        if margins:
            return text.raw[:rc].decode("latin-1"), trace[:nt.value], mg[:nt.value]
        return text.raw[:rc].decode("latin-1"), trace[:nt.value]    # (the returned count: a decoded byte may be 0)

    def stream_geometry(self, n_samples):
        """(halo, n_out, n_blocks, hop) for a buffer of n_samples (UC_STREAM)."""
        v = [C.c_size_t() for _ in range(4)]
        rc = lib().uco_stream_geometry(self._h, n_samples, *[C.byref(x) for x in v])
        if rc:
            raise ValueError("uco_stream_geometry rc=%d" % rc)
        return tuple(x.value for x in v)

    def process_stream(self, samples, threads=0):
        """UC_STREAM over one buffer (first `halo` samples = history) -> (compressed, peaks)."""
        a = np.ascontiguousarray(samples).reshape(-1)
        if a.dtype not in (np.int32, np.float32):
            raise TypeError("samples must be int32 or float32")
        dt = DTYPE_I32 if a.dtype == np.int32 else DTYPE_F32
        _, n_out, n_blocks, _ = self.stream_geometry(a.size)
        comp = np.zeros(n_out, np.float32)
        peaks = np.zeros(n_blocks, PEAK_DTYPE)
        rc = lib().uco_process_stream(self._h, _ptr(a), dt, a.size, _ptr(comp), _ptr(peaks), threads)
        if rc:
            raise RuntimeError("uco_process_stream rc=%d" % rc)
        return comp, peaks

    def spectrum(self, frame, precision=F64, halo=0):
        a = np.ascontiguousarray(frame).reshape(-1)
        dt = DTYPE_I32 if a.dtype == np.int32 else DTYPE_F32
        if a.dtype not in (np.int32, np.float32):
            raise TypeError("frame must be int32 or float32")
        out = np.zeros(self.n * (2 if self.spf == 2 else 1), np.float64)
        rc = lib().uco_spectrum(self._h, C.c_void_p(a.ctypes.data + 4 * halo), dt, precision, _ptr(out))
        if rc:
            raise RuntimeError("uco_spectrum rc=%d" % rc)
        return out.reshape(-1, self.n)


def dfsdm_sinc5(pdm_words):
    """sinc^5 / 32 of a packed PDM stream (first 4 words = history) -> int32 DFSDM words."""
    w = np.ascontiguousarray(pdm_words, np.uint32).reshape(-1)
    out = np.zeros(max(w.size - 4, 0), np.int32)
    rc = lib().uco_dfsdm_sinc5(_ptr(w), w.size, _ptr(out))
    if rc:
        raise RuntimeError("uco_dfsdm_sinc5 rc=%d" % rc)
    return out


def pdm_modulate(x):
    """Test helper: 2nd-order delta-sigma modulation of x in [-1, 1] -> packed uint32 words."""
    x = np.ascontiguousarray(x, np.float32).reshape(-1)
    if x.size % 32:
        raise ValueError("length must be a multiple of 32")
    w = np.zeros(x.size // 32, np.uint32)
    rc = lib().uco_pdm_modulate(_ptr(x), x.size, _ptr(w))
    if rc:
        raise RuntimeError("uco_pdm_modulate rc=%d" % rc)
    return w


def arm_cos(x):
    return float(lib().uco_arm_cos_f32(float(x)))


def arm_sin_cos(theta_deg):
    s, c = C.c_float(), C.c_float()
    lib().uco_arm_sin_cos_f32(float(theta_deg), C.byref(s), C.byref(c))
    return s.value, c.value


def arm_max(v):
    v = np.ascontiguousarray(v, np.float32)
    m, i = C.c_float(), C.c_uint32()
    lib().uco_arm_max_f32(_ptr(v), v.size, C.byref(m), C.byref(i))
    return m.value, i.value


def rfft_fast(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().uco_rfft_fast_f32(_ptr(x), _ptr(out), x.size)
    return out


def hann_periodic(n, libm=False):
    w = np.empty(n, np.float32)
    lib().uco_hann_periodic(_ptr(w), n, 1 if libm else 0)
    return w
