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
 * K6: copies three on-device capture triplets (agent/, data files).
 * K7: sha256 + head of generator/ChirpTone.wav and the regeneration recipe's
   parameters.
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
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
