"""CPU tests of the drop-in boundary: libuchirp.so loads, exports every entry
point include/uchirp.h declares, and refuses to run without a GPU (no CPU path)."""
import ctypes as C
import errno
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "uchirp.h")


@pytest.fixture(scope="module")
def uchirp():
    import uchirp as m
    m.build()
    m.lib()
    return m


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uc_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(uchirp):
    decl = _declared_functions()
    assert len(decl) >= 12
    L = uchirp.lib()
    missing = [s for s in decl if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(uchirp.EXPORTS) == decl
    assert L.uc_abi_version() == 7


def test_struct_layouts(uchirp):
    assert C.sizeof(uchirp.Config) == 52
    assert uchirp.STATS_DTYPE.itemsize == 32


def test_default_configs_mirror_the_firmware_constants(uchirp):
    c = uchirp.default_config(uchirp.RX_REAL)
    assert (c.n, c.fs, c.f0, c.f1) == (2048, 78125.0, 16000.0, 19000.0)
    assert abs(c.time_frame - 0.0205) < 1e-9 and c.phase_deg == -90.0 and c.snr_threshold == 2.0
    c = uchirp.default_config(uchirp.COMPRESS)
    assert (c.fs, c.f0, c.f1, c.time_frame) == (100000.0, 17000.0, 18000.0, 0.0)
    c = uchirp.default_config(uchirp.IQ)
    assert (c.fs, c.carrier) == (100000.0, 18000.0)
    with pytest.raises(uchirp.UchirpError):
        uchirp.default_config(17)


def test_no_gpu_means_no_engine(uchirp):
    """The product path fails loudly instead of falling back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    cfg = uchirp.default_config(uchirp.RX_REAL)
    h = C.c_void_p()
    rc = uchirp.lib().uc_create(C.byref(cfg), C.byref(h))
    assert rc == -errno.ENODEV and not h.value
    assert b"no CPU path" in uchirp.lib().uc_last_error()
    with pytest.raises(uchirp.UchirpError):
        uchirp.Engine(uchirp.RX_REAL)


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "ultrasonic-communication_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                for needle in ("uc_oracle", "libuc_oracle", "from oracle", "import oracle", "uco_"):
                    assert needle not in txt, (f, needle)


def test_header_is_plain_c_and_a_c_host_fails_loudly_without_a_gpu(uchirp, tmp_path):
    """include/uchirp.h compiles as C99 (-pedantic -Werror); tests/c/host_main.c links against libuchirp.so.  Without
    a GPU uc_create reports the missing device (no CPU path) and the program says so."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "ultrasonic-communication_amd")
    exe = str(tmp_path / "host_main")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "c", "host_main.c"), "-o", exe, "-L" + libdir, "-luchirp", "-lm",
                           "-Wl,-rpath," + libdir])
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU suite runs the program (test_plain_c_host_through_the_c_abi)")
    out = subprocess.run([exe], capture_output=True, timeout=120)
    assert out.returncode == 0
    text = out.stdout.decode()
    assert "uc_abi_version 7 (header 7)" in text and "uc_create: -19" in text and "no CPU path" in text
