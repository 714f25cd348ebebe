"""uc_group at world > 1 on the ONE GPU this box has, with a loop-back stand-in for RCCL (tests/stubs/loopback_rccl.cpp: RCCL
itself refuses two ranks on one device).  What the real library does between GPUs is not what is tested here -- the group's
own logic is: which frames a rank decodes, where its slice lands, the in-place all-gather call and the ragged broadcast
path, several local devices in one RCCL group, buffer rotation under the write-after-gather guard, shards that hold only
their uc_frame_span.  Runs in a child process (the library choice is made once per process; the other tests use RCCL)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_group_logic_at_world_2_to_8_on_one_gpu(tmp_path):
    so = str(tmp_path / "libloopback_rccl.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "stubs", "loopback_rccl.cpp"), "-o", so, "-L/opt/rocm/lib", "-lamdhip64", "-lrt"],
                          stderr=subprocess.DEVNULL)
    env = dict(os.environ, UC_TUNING="1", UC_RCCL_LIB=so, UC_GROUP_SHARE_DEVICES="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "group_loopback_child.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "loopback ok" in p.stdout
    print(p.stdout.strip().splitlines()[-1])


def test_rehearsal_hooks_need_uc_tuning(tmp_path):
    """Without UC_TUNING=1 the hooks do nothing: the real RCCL is loaded and two ranks on one device are refused."""
    code = ("import sys; sys.path.insert(0, %r); import uchirp\n"
            "try:\n    uchirp.Group(uchirp.RX_REAL, devices=[0, 0])\n    print('CREATED')\n"
            "except uchirp.UchirpError as e:\n    print('REFUSED', e)\n") % os.path.join(ROOT, "ultrasonic-communication_amd")
    env = dict(os.environ, UC_RCCL_LIB="/nonexistent.so", UC_GROUP_SHARE_DEVICES="1")
    env.pop("UC_TUNING", None)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "REFUSED" in p.stdout and "named twice" in p.stdout, p.stdout + p.stderr
