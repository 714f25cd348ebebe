#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference.

Runs ONLY in the build container (needs /root/reference); the fixtures it
writes are committed, the reference never travels.  What it does:

 * imports the reference's Python generators (simulation/signal.py, dsp.py) by
   path.  Their module-level imports of `peakutils` and `IPython.display` --
   used only for printing / audio widgets -- are satisfied with EMPTY stub
   modules (no numerics in the stubs; peak picking below is this repo's own).
 * K1/K2/K3: evaluates the notebook cells' one-liners on the reference's own
   generated signals and stores inputs + float64 spectra, together with the
   peak frequencies the notebooks RECORDED in their stdout (parsed from the
   .ipynb JSON -- reference-held data).
 * K4: the FIR taps printed by "FIR LPF design.ipynb" cell 13.
 * K3: IQ_modulation.ipynb cells 4, 13, 16, 22, 28-31 on the reference's chirp.py (`chirp`, `lpf`): the modulated
   signals, the demodulated base-band signals and the Butterworth coefficients its `lpf` call designs.
 * K5: ChirpSimulation.ipynb's deterministic cells on the reference's dsp.py `Chirp`.
 * K6: copies three on-device capture triplets (agent/, data files) and parses ALL 24 into one .npz.
 * K7: sha256 + head of generator/ChirpTone.wav and the regeneration recipe's
   parameters.
 * K8: the eight `history` rows of the on-device sync log in experiments/EXPERIMENT3.md:50-59 (the lab note's text: data).
"""
import hashlib
import importlib.util
import json
import os
import re
import shutil
import sys
import types
import wave

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load_reference_module(name, path):
    for stub in ("peakutils", "IPython", "IPython.display"):
        if stub not in sys.modules:
            m = types.ModuleType(stub)
            if stub == "IPython.display":
                m.display = lambda *a, **k: None
                m.Audio = lambda *a, **k: None
            sys.modules[stub] = m
    os.environ.setdefault("MPLBACKEND", "Agg")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def notebook_peaks(nb_path):
    """{cell index: [recorded peak frequencies]} from 'Frequencies at peaks: [...] Hz' stdout."""
    nb = json.load(open(nb_path))
    out = {}
    for i, c in enumerate(nb["cells"]):
        if c["cell_type"] != "code":
            continue
        for o in c.get("outputs", []):
            if o.get("output_type") == "stream":
                txt = "".join(o["text"])
                m = re.search(r"Frequencies at peaks: \[(.*?)\] Hz", txt, re.S)
                if m:
                    out[str(i)] = [float(v) for v in m.group(1).split()]
    return out


def main():
    sig = _load_reference_module("ref_signal", os.path.join(REF, "simulation/signal.py"))
    known = {
        "ChirpSynchronization": notebook_peaks(os.path.join(REF, "simulation/ChirpSynchronization.ipynb")),
        "OrthogonalChirp": notebook_peaks(os.path.join(REF, "simulation/OrthogonalChirp.ipynb")),
        "IQ_modulation": notebook_peaks(os.path.join(REF, "simulation/IQ_modulation.ipynb")),
        "ChirpSimulation": notebook_peaks(os.path.join(REF, "simulation/ChirpSimulation.ipynb")),
    }

    arrays = {}
    # ---- K1: ChirpSynchronization.ipynb cells 3,5,7,9,11 -------------------
    s = sig.Signal(f0=16000, f1=19000, fs=100000, T=0.0205, A=20000)
    chirp = s.chirp()
    chirp_cos = s.chirp_cos()
    arrays["k1_chirp"] = chirp
    arrays["k1_chirp_cos"] = chirp_cos
    for cell, rate in (("5", 0.0), ("7", 1.0 / 8.0), ("9", 2.0 / 8.0), ("11", 4.0 / 8.0)):
        wave_ = sig.time_shift(chirp_cos, rate) * chirp
        arrays["k1_shift_cell%s" % cell] = sig.time_shift(chirp_cos, rate)
        arrays["k1_absfft_cell%s" % cell] = np.abs(np.fft.fft(wave_))
    # ---- K2: OrthogonalChirp.ipynb cells 2,8,11-13 (noise-free ones) --------
    sr = sig.Signal(f0=16000, f1=19000, fs=100000, T=0.0205, A=20000)
    W = np.real(sr.chirp()) + np.imag(sr.chirp())
    arrays["k2_W"] = W
    arrays["k2_orth"] = sr.chirp_orth()
    arrays["k2_absfft_cell8"] = np.abs(np.fft.fft(W))
    arrays["k2_absfft_cell12"] = np.abs(np.fft.fft(W * sr.chirp(updown="up")))
    arrays["k2_absfft_cell13"] = np.abs(np.fft.fft(W * sr.chirp(updown="down")))
    arrays["k2_chirp_down"] = sr.chirp(updown="down")
    np.savez_compressed(os.path.join(OUT, "notebook_vectors.npz"), **arrays)

    # ---- K3: IQ_modulation.ipynb (runs ./chirp.py: Fs = 44100, TIME_FRAME = 0.0262, AMPLITUDE = 20000) -----
    ch = _load_reference_module("ref_chirp", os.path.join(REF, "simulation/chirp.py"))
    from numpy import linspace, cos, sin, pi
    BW, CARRIER = 2000, 17000                      # cell 2
    F0, F1, T = -BW / 2, +BW / 2, ch.TIME_FRAME    # cell 3
    Fs, AMPLITUDE = ch.Fs, ch.AMPLITUDE

    def carrier_IQ(iq, f=CARRIER, phase=0.0):      # cell 4, verbatim maths on the reference's constants
        t = linspace(0, T, int(T * Fs))
        return cos(2 * pi * f * t + phase) if iq == 'I' else sin(2 * pi * f * t + phase)

    def chirp_x_carrier(f0=F0, f1=F1, updown="up"):  # cell 4
        t = linspace(0, T, int(T * Fs))
        k = float(f1 - f0) / float(T)
        f = f0 + k * t / 2.0 if updown == "up" else f1 - k * t / 2.0
        return cos(2 * pi * (CARRIER - f) * t) * AMPLITUDE

    WW, WWd = chirp_x_carrier(), chirp_x_carrier(updown="down")              # cell 13
    Ri, Rq = WW * carrier_IQ('I'), WW * carrier_IQ('Q')                      # cell 16
    Rid, Rqd = WWd * carrier_IQ('I'), WWd * carrier_IQ('Q')
    CUTOFF = 3000                                                            # cell 22
    R = ch.lpf(Ri, CUTOFF) + 1j * ch.lpf(Rq, CUTOFF)
    Rd = ch.lpf(Rid, CUTOFF) + 1j * ch.lpf(Rqd, CUTOFF)
    ref_chirp = ch.chirp(f0=F0, f1=F1)
    # the coefficients ch.lpf designs (chirp.py:139-146: buttord + butter with these arguments)
    from scipy.signal import buttord, butter
    WP = float(CUTOFF) / float(Fs / 2)
    Nb, Wn = buttord(wp=WP, ws=1.3 * WP, gpass=2, gstop=30, analog=0)
    bb, ab = butter(Nb, Wn, btype='low', analog=0, output='ba')
    k3 = {"k3_WW": WW, "k3_WWd": WWd, "k3_R": R, "k3_Rd": Rd, "k3_chirp": ref_chirp, "k3_butter_b": bb, "k3_butter_a": ab,
          "k3_absfft_cell28": np.abs(np.fft.fft(R * ref_chirp)), "k3_absfft_cell29": np.abs(np.fft.fft(R * ref_chirp.conjugate())),
          "k3_absfft_cell30": np.abs(np.fft.fft(Rd * ref_chirp)), "k3_absfft_cell31": np.abs(np.fft.fft(Rd * ref_chirp.conjugate()))}
    known["IQ_modulation_params"] = {"Fs": Fs, "T": T, "AMPLITUDE": AMPLITUDE, "BW": BW, "CARRIER": CARRIER, "CUTOFF": CUTOFF,
                                     "samples": int(T * Fs), "butter_order": int(Nb)}

    # ---- K5: ChirpSimulation.ipynb (runs ./dsp.py) cells 2, 6, 7, 13, 14, 20, 22 ------------------------------
    dsp = _load_reference_module("ref_dsp", os.path.join(REF, "simulation/dsp.py"))
    c = dsp.Chirp(f0=440, f1=1760, fs=44100, T=0.02, A=20000)
    cl = dsp.Chirp(f0=440, f1=1760, fs=44100, T=2.0, A=20000)
    cl2 = dsp.Chirp(f0=440, f1=1760, fs=44100, T=2.0, A=np.sqrt(20000))
    k5 = {"k5_c_chirp": c.chirp(), "k5_c_chirp_cos": c.chirp_cos(), "k5_c_chirp_down": c.chirp(updown="down"),
          # 88200-point signals: every 97th sample only (spot checks of this repo's generator)
          "k5_cl_chirp_s97": cl.chirp()[::97], "k5_cl_chirp_cos_s97": cl.chirp_cos()[::97],
          "k5_cl2_updown_s97": (cl2.chirp() * cl2.chirp(updown="down"))[::97]}
    np.savez_compressed(os.path.join(OUT, "notebook_vectors_k3k5.npz"), **k3, **k5)

    # ---- K4: FIR taps --------------------------------------------------------
    nb = json.load(open(os.path.join(REF, "simulation/FIR LPF design.ipynb")))
    taps = None
    for c in nb["cells"]:
        if c["cell_type"] == "code":
            for o in c.get("outputs", []):
                if o.get("output_type") == "stream":
                    txt = "".join(o["text"])
                    vals = re.findall(r"-?\d+\.\d+", txt)
                    if len(vals) == 27:
                        taps = [float(v) for v in vals]
    known["fir_taps_cell13"] = taps

    # ---- K7: transmit waveform ----------------------------------------------
    w = wave.open(os.path.join(REF, "generator/ChirpTone.wav"), "rb")
    raw = w.readframes(w.getnframes())
    samples = np.frombuffer(raw, dtype="<i2")
    known["ChirpTone"] = {
        "framerate": w.getframerate(), "nframes": w.getnframes(), "channels": w.getnchannels(),
        "sha256_le_int16": hashlib.sha256(samples.astype("<i2").tobytes()).hexdigest(),
        "samples_per_symbol": 1155,
        "first_nonzero_symbol_head": [int(v) for v in samples[np.nonzero(samples)[0][0]:][:6]],
    }
    # regenerate with the reference's own generator (generator/ChirpGenerator.ipynb cells 1,3)
    st = sig.Signal(f0=16000, f1=19000, fs=44100, T=0.0262, A=20000)
    H = st.chirp_orth(updown="up")
    L = st.chirp_orth(updown="down")
    known["ChirpTone"]["H_head_int16"] = [int(v) for v in H.astype(np.int16)[:6]]
    known["ChirpTone"]["L_head_int16"] = [int(v) for v in L.astype(np.int16)[:6]]
    # ---- K8: the on-device sync log (experiments/EXPERIMENT3.md:50-59) -------------------------------------
    # eight rows of `struct history`, printed by a revision of the complex-FFT build (experiments/synchronization) that is not
    # in the checkout: one row per FIFO offset (pos = N/2 + 256 k), fields as printed
    rows = []
    with open(os.path.join(REF, "experiments/EXPERIMENT3.md")) as f:
        for ln, line in enumerate(f, 1):
            m = re.match(r"max: ([\d.]+), max_r: ([\d.]+), max_l: ([\d.]+), s_time: (\d+), f_time: (\d+), i: (\d+), "
                         r"i_left: (\d+), i_right: (\d+)", line.strip())
            if m:
                g = m.groups()
                rows.append({"line": ln, "max": float(g[0]), "max_r": float(g[1]), "max_l": float(g[2]), "s_time": int(g[3]),
                             "f_time": int(g[4]), "i": int(g[5]), "i_left": int(g[6]), "i_right": int(g[7])})
    assert len(rows) == 8 and rows[0]["line"] == 51 and rows[-1]["line"] == 58, rows
    known["EXPERIMENT3_sync_log"] = {"source": "experiments/EXPERIMENT3.md:50-59", "rows": rows}
    json.dump(known, open(os.path.join(OUT, "known_answers.json"), "w"), indent=1, sort_keys=True)

    # ---- K6: on-device captures (data files) ---------------------------------
    k6 = os.path.join(OUT, "k6")
    os.makedirs(k6, exist_ok=True)
    picks = {
        "chirp_16000_18000_1m_100kHz_M1": "agent/chirp_experiment/16000_18000_1m_100.0(kHz)_M1",
        "paper_100kHz_M1": "agent/chirp_experiment/paper_100.0(kHz)_M1",
        "vacuum_1526445492_41.7kHz_M1": "agent/vaccum_cleaner/1526445492_41.7(kHz)_M1",
    }
    for dst, src in picks.items():
        for ext in ("raw", "flt", "fft"):
            shutil.copyfile(os.path.join(REF, src + "." + ext), os.path.join(k6, dst + "." + ext))
            os.chmod(os.path.join(k6, dst + "." + ext), 0o644)
    # all 24 triplets parsed into arrays (23 are consistent, `chirp_experiment/48.1(kHz)_M2A` is a
    # mismatched trio and `100.0(kHz)_M1/_M2A`, `48.1(kHz)_M1` lack files)
    import glob

    def col(path, ncol, which):
        rows = []
        with open(path) as f:
            next(f)
            for line in f:
                parts = line.strip().split(",")
                if len(parts) < ncol:
                    continue
                try:
                    rows.append(float(parts[which]))
                except ValueError:
                    continue
        return np.array(rows)

    allk6 = {}
    names = []
    for d in ("agent/chirp_experiment", "agent/vaccum_cleaner"):
        for raw in sorted(glob.glob(os.path.join(REF, d, "*.raw"))):
            stem = raw[:-4]
            if not (os.path.exists(stem + ".flt") and os.path.exists(stem + ".fft")):
                continue
            key = "%s/%s" % (os.path.basename(d), os.path.basename(stem))
            r, fl = col(raw, 2, 1), col(stem + ".flt", 2, 1)
            fq, fm = col(stem + ".fft", 3, 0), col(stem + ".fft", 3, 1)
            if r.size != 2048 or fl.size != 2048 or fm.size != 1024:
                continue
            i = len(names)
            names.append(key)
            allk6["raw_%02d" % i] = r.astype(np.int64)
            allk6["flt_%02d" % i] = fl
            allk6["fftfreq_%02d" % i] = fq
            allk6["fftmag_%02d" % i] = fm
    allk6["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "k6_all.npz"), **allk6)
    print("K6 triplets:", len(names))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
