"""bench.py's own N-rank launch (BASELINE configs[4]) rehearsed on the CPU: the parent starts
the ranks, they rendezvous over gloo, gather the symbol stream, check the concatenation on every rank and decode
the K7 framing.  Without a GPU no kernel runs (value is null): this covers the plumbing only."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=240):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_gpus_2_starts_two_ranks_and_decodes_hello_world():
    p = _run(["--gpus", "2", "--frames", "1170", "--steps", "3", "--warmup", "1"],
             {"UC_BENCH_REHEARSE": "1", "HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2
    assert d["config"]["workload"].startswith("configs[4]")
    assert d["decoded_text_first"] == "Hello World!"
    assert d["transmissions"] == 2 * 1170 // 117 and d["transmissions_decoded_exactly"] == d["transmissions"]
    assert d["value"] is None and "rehearsal" in d      # no GPU here: never a measurement
    # every rank's own kernel and step time ride in the N > 1 line
    assert len(d["per_rank"]["kernel_ms_by_rank"]) == 2 and d["per_rank"]["ms_per_step"]["max"] > 0 and d["gates_failed"] == []


def test_strong_scaling_line_shares_a_fixed_batch():
    """`--scaling strong`: --frames is the batch of the whole job, every rank takes its uc_partition share (--frames / N), the
    same gather; the line says so.  Plumbing on CPU (gloo, no kernel); a world that does not divide the batch is refused."""
    p = _run(["--gpus", "2", "--frames", "2340", "--steps", "3", "--warmup", "1", "--scaling", "strong"],
             {"UC_BENCH_REHEARSE": "1", "HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["frames_per_gpu"] == 1170
    assert d["transmissions"] == 2340 // 117 and d["transmissions_decoded_exactly"] == d["transmissions"] and d["gates_failed"] == []
    # who ran where rides in the N > 1 line
    assert [r["rank"] for r in d["per_rank"]["ranks"]] == [0, 1] and all("pid" in r for r in d["per_rank"]["ranks"])
    p = _run(["--gpus", "2", "--frames", "1171", "--steps", "2", "--warmup", "1", "--scaling", "strong"],
             {"UC_BENCH_REHEARSE": "1", "HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    assert p.returncode != 0 and "do not divide" in p.stderr


def test_gpus_must_equal_world_size():
    p = _run(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0
    assert "--gpus 2 but WORLD_SIZE=1" in p.stderr


def test_no_gpu_is_a_loud_failure():
    p = _run(["--gpus", "1", "--frames", "64"], {"HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    assert p.returncode != 0
    assert "no CPU fallback" in p.stderr


def test_stdout_is_exactly_one_json_line_for_n_ranks():
    """The driver reads ONE JSON line from stdout: whatever the libraries of the ranks print on descriptor 1 (RCCL's
    NCCL_DEBUG=VERSION banner on the GPU boxes, gloo's connection notes here) must not reach it."""
    p = _run(["--gpus", "2", "--frames", "234", "--steps", "2", "--warmup", "1"],
             {"UC_BENCH_REHEARSE": "1", "HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_hello_leg_runs_at_world_size_one():
    p = _run(["--frames", "1170", "--steps", "2", "--warmup", "1"],
             {"UC_BENCH_REHEARSE": "1", "UC_BENCH_HELLO": "1", "HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["workload"].startswith("configs[4]") and d["decoded_text_first"] == "Hello World!"


# ---- on the GPU box: the legs of bench.py the driver's 8-GPU run depends on, as fresh child processes -----------------

def _one_line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.gpu
def test_gpu_two_rank_rehearsal_runs_the_real_kernel_and_decodes():
    """`--gpus 2` as the driver's launcher would run it, both ranks on device 0 (one-GPU box), gloo gather: spawn,
    rendezvous, the real kernel on every rank, gather every step, digest agreement, decode, per-rank statistics."""
    d = _one_line(_run(["--gpus", "2", "--frames", "65536", "--steps", "3", "--warmup", "1"], {"UC_BENCH_REHEARSE": "1"}))
    assert d["n_gpus"] == 2 and d["config"]["workload"].startswith("configs[4]")
    assert d["transmissions"] == 2 * 65536 // 117 and d["transmissions_decoded_exactly"] == d["transmissions"]
    assert d["value"] > 0 and abs(d["value_per_gpu"] * 2 - d["value"]) < 1e-6 * d["value"]
    assert len(d["per_rank"]["kernel_ms_by_rank"]) == 2 and min(d["per_rank"]["kernel_ms_by_rank"]) > 0
    assert d["gather_ms_exposed"] is not None and d["gates_failed"] == []
    assert d["roofline"]["kernel"] == "band_kernel<rx_real,f32>" and "valu" in d["roofline"]


@pytest.mark.gpu
def test_gpu_hello_leg_on_rccl_at_world_size_one():
    """The N > 1 leg on RCCL itself (process group with device_id, asynchronous all_gather_into_tensor, digest check,
    decode) with the world a one-GPU box has."""
    d = _one_line(_run(["--frames", "65536", "--steps", "3", "--warmup", "1"], {"UC_BENCH_HELLO": "1"}))
    assert d["n_gpus"] == 1 and d["config"]["workload"].startswith("configs[4]")
    assert d["transmissions_decoded_exactly"] == d["transmissions"] == 65536 // 117
    assert d["decoded_text_first"] == "Hello World!" and "rehearsal" not in d and d["gates_failed"] == []
    # the gather is the C-ABI's: a uc_group of one rank per process, RCCL called from C (include/uchirp.h)
    assert d["gather_backend"].startswith("uc_group_process_batch")


@pytest.mark.gpu
def test_gpu_single_process_group_line():
    """`--gpus 1 --single-process`: one host process drives the devices through uc_group_create (ncclCommInitAll) -- the
    shape of a C host (tests/c/host_multi.c); same workload, same JSON line."""
    d = _one_line(_run(["--gpus", "1", "--single-process", "--frames", "65536", "--steps", "3", "--warmup", "1", "--ramp-ms", "20"], {}))
    assert d["n_gpus"] == 1 and d["config"]["workload"].startswith("configs[4]") and d["gates_failed"] == []
    assert d["transmissions_decoded_exactly"] == d["transmissions"] == 65536 // 117 and d["decoded_text_first"] == "Hello World!"
    assert d["gather_backend"].startswith("uc_group_process_batch in ONE process") and d["value"] > 0


@pytest.mark.gpu
def test_gpu_default_line_carries_every_single_gpu_config():
    """The N = 1 contract line (small batch): configs[1] headline + configs[2] (base-band I/Q and firmware windows) +
    configs[3] (eager and graph replay) + hello_world1 + cpu_baseline, all gates green."""
    d = _one_line(_run(["--frames", "65536", "--steps", "3", "--warmup", "1", "--ramp-ms", "20"], {}))
    assert d["n_gpus"] == 1 and d["config"]["workload"].startswith("configs[1]") and d["gates_failed"] == []
    # HBM traffic from the PMC counters, collected in this run (two rocprofv3 child passes): no wasted re-reads
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["limiter"] in ("hbm", "valu") and d["roofline"]["traffic_source"].startswith("measured in this run")
    assert 0.99 < d["roofline"]["traffic_over_algorithmic"] < 1.05
    pw = d["roofline"]["power"]                            # hwmon telemetry of a sustained run (null without sysfs access)
    assert pw is None or (pw["cap_W"] > 0 and 100 < pw["socket_W_mean"] <= pw["cap_W"] * 1.02 and pw["sustained_frames_per_s"] > 0)
    c2, c3 = d["configs"]["configs[2]"], d["configs"]["configs[3]"]
    assert c2["baseband"]["symbols_equal_oracle_head4096_clear"] == 1.0 and c2["baseband"]["bit_error_rate_vs_transmitted"] < 0.03
    assert c2["baseband"]["roofline"]["kernel"].startswith("iq1024_kernel") and c2["firmware_windows"]["value"] > 0
    assert c3["graph_equals_eager"] is True and c3["head_rel_err_vs_oracle"] < 2e-5 and c3["head_peak_offsets_equal_oracle"]
    assert c3["graph_replay"]["value"] > 0 and c3["eager"]["value"] > 0 and c3["samples"] == 65536 * 2048
    h = d["hello_world1"]
    assert h["transmissions_decoded_exactly"] == h["transmissions"] == 65536 // 117 and h["gathered_equals_decoded"]
    assert h["gather_backend"].startswith("uc_group_process_batch") and d["scale_anchor"]["value"] == h["value"]
    # the clock of the VALU roof is measured in this run (the stamped twin of the kernel), the CU count is the device's
    v = d["roofline"]["valu"]
    assert v["clock_source"].startswith("uc_clock_read") and 0.8 < v["clock_GHz"] < 3.0 and v["num_cu"] >= 64
    assert c2["baseband"]["roofline"]["valu"]["clock_source"].startswith("uc_clock_read")
    assert c3["eager"]["roofline"]["valu"]["clock_source"].startswith("uc_clock_read")
    assert d["symbols_equal_oracle_head4096_clear"] == 1.0 and d["cpu_baseline"]["kind"] == "port"


@pytest.mark.gpu
def test_gpu_single_process_three_ranks_rehearsed_on_one_gpu(tmp_path):
    """`--gpus 3 --single-process` on the one GPU of this box: every rank on device 0, the loop-back stand-in for RCCL
    (tests/stubs/loopback_rccl.cpp).  Plumbing of the one-host-process mode at N > 1: partition, per-device streams and
    events, gather, digests, decode, JSON line."""
    so = str(tmp_path / "libloopback_rccl.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "stubs", "loopback_rccl.cpp"), "-o", so, "-L/opt/rocm/lib", "-lamdhip64", "-lrt"],
                          stderr=subprocess.DEVNULL)
    d = _one_line(_run(["--gpus", "3", "--single-process", "--frames", "35100", "--steps", "4", "--warmup", "1", "--ramp-ms", "10"],
                       {"UC_BENCH_REHEARSE": "1", "UC_TUNING": "1", "UC_RCCL_LIB": so, "UC_GROUP_SHARE_DEVICES": "1"}))
    assert d["n_gpus"] == 3 and "rehearsal" in d and d["gates_failed"] == []
    assert d["transmissions"] == 3 * 35100 // 117 and d["transmissions_decoded_exactly"] == d["transmissions"]
    assert len(d["per_rank"]["kernel_ms_by_rank"]) == 3


@pytest.mark.gpu
def test_gpu_two_ranks_through_the_c_group_rehearsed_on_one_gpu(tmp_path):
    """`--gpus 2` exactly as the driver's launcher runs it -- two processes, the unique id carried by torch.distributed,
    uc_group_create_rank in every rank, every step decoded into the rank's slice and gathered through the group -- on the one
    GPU of this box: both ranks on device 0, gloo as the launcher's backend, the loop-back stand-in as the group's RCCL
    (shared-memory form).  The code path of the 8-GPU run, minus the real transport."""
    so = str(tmp_path / "libloopback_rccl.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "stubs", "loopback_rccl.cpp"), "-o", so, "-L/opt/rocm/lib", "-lamdhip64", "-lrt"],
                          stderr=subprocess.DEVNULL)
    d = _one_line(_run(["--gpus", "2", "--frames", "65520", "--steps", "4", "--warmup", "1", "--ramp-ms", "10"],
                       {"UC_BENCH_REHEARSE": "1", "UC_TUNING": "1", "UC_RCCL_LIB": so}))
    assert d["n_gpus"] == 2 and d["gather_backend"].startswith("uc_group_process_batch") and d["gates_failed"] == []
    assert d["transmissions"] == 2 * 65520 // 117 and d["transmissions_decoded_exactly"] == d["transmissions"]
    assert d["decoded_text_first"] == "Hello World!" and len(d["per_rank"]["kernel_ms_by_rank"]) == 2
    # round 5: the exposed gather per rank (kernel bracketed alone: the write-after-gather wait sits in front of it), the world
    # size the C group's communicator reports, and the statement that no curve was measured by the builder
    assert len(d["gather_ms_exposed_by_rank"]) == 2 and d["rccl_world"] == 2 and "no 1 -> 8 curve" in d["scaling_note"]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_gpu_strong_scaling_rehearsed_on_one_gpu(tmp_path, world):
    """`--gpus N --scaling strong --single-process` at world 2, 4 and 8 on the one GPU of this box (every rank on device 0, the
    loop-back stand-in for RCCL): a FIXED batch of 8 x 117 x 35 frames shared out by uc_partition, the real kernel, the C group's
    gather, every transmission decoded.  The first real 8-GPU run then differs by the transport alone."""
    so = str(tmp_path / "libloopback_rccl.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "stubs", "loopback_rccl.cpp"), "-o", so, "-L/opt/rocm/lib", "-lamdhip64", "-lrt"],
                          stderr=subprocess.DEVNULL)
    total = 8 * 117 * 35
    d = _one_line(_run(["--gpus", str(world), "--single-process", "--scaling", "strong", "--frames", str(total), "--steps", "4",
                        "--warmup", "1", "--ramp-ms", "10"],
                       {"UC_BENCH_REHEARSE": "1", "UC_TUNING": "1", "UC_RCCL_LIB": so, "UC_GROUP_SHARE_DEVICES": "1"}))
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["gates_failed"] == [] and "rehearsal" in d
    assert d["config"]["frames_per_gpu"] == total // world
    assert d["transmissions"] == total // 117 and d["transmissions_decoded_exactly"] == d["transmissions"]
