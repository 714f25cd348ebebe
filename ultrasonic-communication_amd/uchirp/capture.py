"""Capture / log formats of the reference -- host-side readers and writers.

The firmware prints one capture over UART when the user button is pressed
(experiments/basic/Src/main.c:144-174) and the PC agent splits it into three CSV files
(agent/README.md:5-12):

    MEMS mic: <name>
    Frequency at max magnitude: <f>, Max magnitude: <m>
    Frequency(Hz),Magnitude,Magnitude(dB)      <- *.fft : n/2 rows  "%.1f,%f,%f"
    ...
    <blank>
    Index,Amplitude                            <- *.raw : n rows    "%lu,%ld"  raw DFSDM words
    ...
    EORAW
    Index,Amplitude                            <- *.flt : n rows    "%lu,%f"   Hann-windowed samples
    ...
    EOFLT

`raw` values are the int32 DFSDM words the ISR hands to the DSP (24-bit sample in bits 31:8:
multiples of 256), i.e. exactly what uc_process_batch takes with UC_DTYPE_I32.
"""
import io

import numpy as np

RAW_HEADER = "Index,Amplitude"
FFT_HEADER = "Frequency(Hz),Magnitude,Magnitude(dB)"


def _rows(lines, ncol):
    out = []
    for ln in lines:
        ln = ln.strip()
        if not ln:
            continue
        parts = ln.split(",")
        if len(parts) != ncol:
            raise ValueError("expected %d columns, got %r" % (ncol, ln))
        out.append(parts)
    return out


def _read_lines(path_or_text):
    if isinstance(path_or_text, (bytes, bytearray)):
        path_or_text = path_or_text.decode("ascii")
    if "\n" in path_or_text:
        return path_or_text.splitlines()
    with open(path_or_text, "r", encoding="ascii") as fh:
        return fh.read().splitlines()


def read_raw(path_or_text):
    """*.raw -> int32[n] DFSDM words (UC_DTYPE_I32 frames)."""
    lines = _read_lines(path_or_text)
    if not lines or lines[0].strip() != RAW_HEADER:
        raise ValueError("not a raw capture: first line %r" % (lines[0] if lines else ""))
    rows = _rows(lines[1:], 2)
    idx = np.array([int(r[0]) for r in rows], np.int64)
    if not np.array_equal(idx, np.arange(idx.size)):
        raise ValueError("raw capture: the index column is not 0..n-1")
    val = np.array([int(r[1]) for r in rows], np.int64)
    if val.size and (val.min() < -2 ** 31 or val.max() >= 2 ** 31):
        raise ValueError("raw capture: value outside int32")
    return val.astype(np.int32)


def read_flt(path_or_text):
    """*.flt -> float32[n] windowed samples."""
    lines = _read_lines(path_or_text)
    if not lines or lines[0].strip() != RAW_HEADER:
        raise ValueError("not a flt capture: first line %r" % (lines[0] if lines else ""))
    rows = _rows(lines[1:], 2)
    return np.array([float(r[1]) for r in rows], np.float32)


def read_fft(path_or_text):
    """*.fft -> (frequency_hz, magnitude, magnitude_db), float32[n/2] each."""
    lines = _read_lines(path_or_text)
    if not lines or lines[0].strip() != FFT_HEADER:
        raise ValueError("not an fft capture: first line %r" % (lines[0] if lines else ""))
    rows = _rows(lines[1:], 3)
    a = np.array([[float(c) for c in r] for r in rows], np.float32).reshape(-1, 3)
    return a[:, 0].copy(), a[:, 1].copy(), a[:, 2].copy()


def parse_uart_dump(text):
    """The whole UART dump of one capture -> dict(mic, freq_max, mag_max, fft=(f, m, db), raw, flt)."""
    if isinstance(text, (bytes, bytearray)):
        text = text.decode("ascii")
    lines = text.splitlines()
    out = {"mic": None, "freq_max": None, "mag_max": None}
    i = 0
    while i < len(lines) and lines[i].strip() != FFT_HEADER:
        ln = lines[i].strip()
        if ln.startswith("MEMS mic:"):
            out["mic"] = ln.split(":", 1)[1].strip()
        elif ln.startswith("Frequency at max magnitude:"):
            a, b = ln.split(",", 1)
            out["freq_max"] = float(a.split(":", 1)[1])
            out["mag_max"] = float(b.split(":", 1)[1])
        i += 1
    if i == len(lines):
        raise ValueError("UART dump: no FFT section")
    j = i + 1
    while j < len(lines) and lines[j].strip() != RAW_HEADER:
        j += 1
    out["fft"] = read_fft("\n".join(lines[i:j]) + "\n")
    try:
        k = next(x for x in range(j, len(lines)) if lines[x].strip() == "EORAW")
        m = next(x for x in range(k, len(lines)) if lines[x].strip() == "EOFLT")
    except StopIteration:
        raise ValueError("UART dump: EORAW / EOFLT marker missing")
    out["raw"] = read_raw("\n".join(lines[j:k]) + "\n")
    out["flt"] = read_flt("\n".join(lines[k + 1:m]) + "\n")
    return out


def format_uart_dump(mic, freq_max, mag_max, fft, raw, flt):
    """Inverse of parse_uart_dump with the firmware's printf formats (main.c:147-170)."""
    f, mag, db = fft
    s = io.StringIO()
    s.write("\nMEMS mic: %s\n" % mic)
    s.write("Frequency at max magnitude: %.1f, Max magnitude: %f\n" % (freq_max, mag_max))
    s.write(FFT_HEADER + "\n")
    for a, b, c in zip(f, mag, db):
        s.write("%.1f,%f,%f\n" % (a, b, c))
    s.write("\n")
    s.write(RAW_HEADER + "\n")
    for i, v in enumerate(np.asarray(raw, np.int64)):
        s.write("%d,%d\n" % (i, v))
    s.write("EORAW\n")
    s.write(RAW_HEADER + "\n")
    for i, v in enumerate(np.asarray(flt, np.float64)):
        s.write("%d,%f\n" % (i, v))
    s.write("EOFLT\n")
    return s.getvalue()


def fs_from_fft(freq_hz, n=None):
    """Sampling rate a *.fft frequency column implies: bin spacing x n (n = 2 x rows)."""
    freq_hz = np.asarray(freq_hz, np.float64)
    n = n or 2 * freq_hz.size
    return float(freq_hz[-1] / (freq_hz.size - 1) * n)
