"""What the compiler made of the kernels that ship in libuchirp.so, read from the code objects' own metadata (no GPU, no recompile):
no kernel may spill a vector register or use scratch memory, and the kernels whose occupancy the design rests on must fit their
register budget.  (Round 5: two more live vector registers in the 168-register build of the band kernel meant 20 bytes of scratch per
lane and 10 % on every live step -- nothing failed, it was only slower.  This file makes that a failure.)"""
import os
import re
import shutil
import subprocess

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("UCHIRP_LIB") or os.path.join(ROOT, "ultrasonic-communication_amd", "libuchirp.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels(tmp_path):
    """{mangled kernel name: [metadata dict per code object that holds it]} of every gfx950 code object bundled in the library.
    (Every kernel file is compiled twice -- as it is, and with one clock-stamp pair per wave, csrc/*.clk.o -- and the kernels
    have internal linkage: the same name appears in two code objects.  The stamped twin needs a few registers more.)"""
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("ROCm's llvm-objdump / llvm-readelf not found")
    if not os.path.exists(LIB):
        pytest.skip("libuchirp.so not built")
    work = tmp_path / "co"
    work.mkdir()
    lib = shutil.copy(LIB, work / "lib.so")       # (llvm-objdump --offloading writes the bundles NEXT TO its input)
    subprocess.run([objdump, "--offloading", str(lib)], check=True, cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = {}
    for f in sorted(os.listdir(work)):
        if "amdgcn" not in f:
            continue
        assert "gfx950" in f, f                   # one target: no other code object may be bundled
        notes = subprocess.run([readelf, "--notes", str(work / f)], check=True, capture_output=True, text=True).stdout
        # the AMDGPU metadata note is a YAML document between "---" and "..."
        doc = notes[notes.index("---"):]
        doc = doc[:doc.index("\n...")] if "\n..." in doc else doc
        for k in yaml.safe_load(doc)["amdhsa.kernels"]:
            out.setdefault(k[".name"], []).append({key[1:]: val for key, val in k.items() if isinstance(val, int)})
    assert out, "no kernel metadata found"
    return out


# builds that are allowed to spill a few vector registers: none of them is a throughput path
#   band_kernel<.., 3 waves, SPEC>: uc_window_spectrum's instantiation (the statistics build + the bin stores)
#   band_kernel<.., 4 waves>: a tuning build (UC_BAND_WAVES=4)
#   stream_kernel<.., D = 4>: the 4-fold decimation (BASELINE's stream config is D = 8)
ALLOWED_TO_SPILL = (r"band_kernelILi\dELi[01]ELi3ELb0ELb1ELi0EE", r"band_kernelILi\dELi[01]ELi4ELb0ELb0ELi0EE", r"stream_kernelILi[01]ELi4EE")


def test_no_throughput_kernel_spills_vector_registers(tmp_path):
    ks = _kernels(tmp_path)
    # the plain build of a kernel is the entry with the fewest spills (its stamped twin carries two 64-bit stamps more)
    spilled = {k: min(e.get("vgpr_spill_count", 0) for e in v) for k, v in ks.items()}
    spilled = {k: n for k, n in spilled.items() if n}
    unexpected = {k: n for k, n in spilled.items() if not any(re.search(p, k) for p in ALLOWED_TO_SPILL)}
    assert not unexpected, unexpected
    # every kernel file is there twice
    names = " ".join(ks)
    for stem in ("band_kernel", "compress_kernel", "iq_kernel", "iq1024_kernel", "stream_kernel", "sinc5_kernel", "hist_kernel",
                 "replay_kernel"):
        assert stem in names, stem
    twice = [k for k, v in ks.items() if len(v) == 2]
    assert len(twice) >= 50, len(twice)           # (the state-machine kernels of uc_rx_kernel.hip have no twin)


def test_register_budgets_the_design_rests_on(tmp_path):
    ks = _kernels(tmp_path)

    def pick(pattern):
        hit = {k: v for k, v in ks.items() if re.search(pattern, k)}
        assert hit, pattern
        return hit

    # band_kernel<MODE, DTYPE, WAVES, WIDE, SPEC, FRAMES>: the RX_REAL builds run 3 waves per SIMD (512 / 3 -> 168 registers),
    # batch (FRAMES 0), live rows (1) and overlapping frames (2), both dtypes, plain and stamped
    for k, v in pick(r"band_kernelILi0ELi[01]ELi3ELb0ELb0ELi[012]EE").items():
        assert all(e["vgpr_count"] <= 168 for e in v), (k, v)
    # the 4-waves-per-SIMD tuning build: 128
    for k, v in pick(r"band_kernelILi0ELi[01]ELi4ELb0ELb0ELi0EE").items():
        assert all(e["vgpr_count"] <= 128 for e in v), (k, v)
    # sinc5: one 1024-thread workgroup per CU = 4 waves per SIMD (128 registers), both tables (128 KiB) in dynamic LDS
    for k, v in pick(r"sinc5_kernel").items():
        assert all(e["vgpr_count"] <= 128 and e["max_flat_workgroup_size"] == 1024 for e in v), (k, v)
    # the band kernel's LDS: one exchange tile + ring + the parked group id = 19 716 bytes -> 6 workgroups on a CU's 160 KiB
    for k, v in pick(r"band_kernelILi0ELi1ELi3ELb0ELb0ELi0EE").items():
        assert all(e["group_segment_fixed_size"] <= 160 * 1024 // 6 for e in v), (k, v)
    assert max(e["vgpr_count"] for v in ks.values() for e in v) <= 256


def test_machine_code_of_every_kernel_is_the_recorded_one():
    """tests/golden/kernel_digests.json holds the sha256 of every kernel's machine code (tools/kernel_digest.py).  A change that is
    not meant to touch a kernel -- a host-side refactoring, work on a sibling instantiation of the same template -- must leave the
    digests alone; one that is meant to re-records them (`python tools/kernel_digest.py --update`) and says so in its commit.
    Round 6: the batch builds of the band kernel (the headline kernel among them) are byte for byte what round 5 shipped while the
    ROWS builds next to them were rebuilt."""
    import importlib.util
    import json
    if not os.path.exists(LIB):
        pytest.skip("libuchirp.so not built")
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("ROCm's llvm-objdump not found")
    spec = importlib.util.spec_from_file_location("kernel_digest", os.path.join(ROOT, "tools", "kernel_digest.py"))
    kd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kd)
    with open(kd.RECORD) as f:
        recorded = json.load(f)
    built = kd.digests(LIB)
    changed = sorted(k for k in set(recorded) | set(built) if recorded.get(k) != built.get(k))
    assert not changed, "kernels whose machine code differs from tests/golden/kernel_digests.json: %s" % changed[:8]
