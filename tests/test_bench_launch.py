"""bench.py's own N-rank launch (BASELINE configs[4]) rehearsed on the CPU: the parent starts
the ranks, they rendezvous over gloo, gather the symbol stream, check the concatenation on every rank and decode
the K7 framing.  Without a GPU no kernel runs (value is null): this covers the plumbing only."""
import json
import os
import subprocess
import sys

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
             {"UC_BENCH_REHEARSE": "1", "UC_BENCH_HELLO": "1", "HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": "",
              "MASTER_PORT": "29541"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["workload"].startswith("configs[4]") and d["decoded_text_first"] == "Hello World!"
