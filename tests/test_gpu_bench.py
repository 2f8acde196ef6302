"""bench.py on the MI355X, in child processes (one GPU): the plain run, the RCCL rehearsal on one rank (the same
init_process_group("nccl") / all_gather_into_tensor / barrier / all_reduce calls an 8-rank run makes) and the refusals."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(extra, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True,
                          env=env, timeout=timeout)


def _line(p):
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_plain_run_prints_the_contract_line(hip_device):
    d = _line(_bench(["--steps", "5", "--warmup", "2", "--frames", "65536", "--cpu-seconds", "1"]))
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["config"]["frames_per_gpu"] == 65536
    assert d["config"]["final_allgather"] is False and d["config"]["dist"]["world_size"] == 1
    assert d["config"]["env"] == {k: v for k, v in os.environ.items() if k.startswith("MOLANN_")}
    assert "synced_launch_ms_min" in d["roofline"] and "launch_ms_min" not in d["roofline"]
    assert d["roofline"]["traffic"] is None      # the PMC figure is per 1M-frame launch, not this size
    assert d["cpu_baseline"]["kind"] == "port" and "diagnostic" not in d
    assert "molann_lane_jit" in d["config"]["kernels"]


def test_rccl_rehearsal_on_one_rank(hip_device):
    d = _line(_bench(["--steps", "4", "--warmup", "2", "--frames", "65536", "--no-cpu-baseline"],
                     {"MOLANN_BENCH_FORCE_DIST": "1"}))
    assert d["n_gpus"] == 1 and d["config"]["final_allgather"] is True
    assert d["config"]["dist"]["world_size"] == 1 and d["config"]["dist"]["backend"] == "nccl"
    assert len(d["config"]["dist"]["devices"]) == 1 and d["config"]["dist"]["devices"][0].startswith("cuda:0")
    ph = d["phases"]
    assert ph["allgather_mode"] in ("collective", "overlap") and set(ph["allgather_warmup_ms"]) == {"collective", "overlap"}
    assert ph["last_step_plus_allgather_ms"] > 0 and ph["compute_ms_per_step"] > 0
    assert d["config"]["env"].get("MOLANN_BENCH_FORCE_DIST") == "1"


def test_gpus_8_on_a_one_gpu_box_fails_loudly(hip_device):
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("8 GPUs visible")
    p = _bench(["--gpus", "8", "--steps", "2", "--warmup", "1"])
    assert p.returncode != 0 and "n_gpus" not in p.stdout and "--gpus 8" in p.stderr


def test_debug_switch_refused_unless_diagnostic(hip_device):
    p = _bench(["--steps", "2", "--warmup", "1", "--frames", "4096", "--no-cpu-baseline"],
               {"MOLANN_ELIDE_INVARIANT_ALIGNMENT": "1"})
    assert p.returncode != 0 and "MOLANN_ELIDE_INVARIANT_ALIGNMENT" in p.stderr and "value" not in p.stdout
    d = _line(_bench(["--steps", "2", "--warmup", "1", "--frames", "4096", "--no-cpu-baseline", "--diagnostic"],
                     {"MOLANN_ELIDE_INVARIANT_ALIGNMENT": "1"}))
    assert d["diagnostic"] is True and d["config"]["env"]["MOLANN_ELIDE_INVARIANT_ALIGNMENT"] == "1"


def test_counting_devices_leaves_the_parent_without_a_gpu_context(hip_device):
    """after the launcher has counted the GPUs, it holds no file descriptor on /dev/kfd (or a render
    node) and torch has not initialised HIP - so forking + exec'ing the ranks from it is safe on this pool."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import bench, torch
n = bench.visible_gpu_count()
fds = []
for fd in os.listdir("/proc/self/fd"):
    try:
        fds.append(os.readlink("/proc/self/fd/" + fd))
    except OSError:
        pass
bad = [f for f in fds if "/dev/kfd" in f or "/dev/dri" in f]
print("count", n, "initialized", torch.cuda.is_initialized(), "gpu_fds", bad)
assert n >= 1 and not torch.cuda.is_initialized() and not bad
''' % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, (p.stdout, p.stderr[-2000:])
    import torch
    assert ("count %d " % torch.cuda.device_count()) in p.stdout, p.stdout
