"""bench.py's host logic without a GPU: the launcher of a plain `--gpus N`, the environment certification, and its
rank code (sharding, timed region, both forms of the one all-gather, max over ranks, JSON line) driven at
world_size 2 over gloo with the kernels replaced by the oracle stand-in of tests/bench_standin.py."""

import json
import os
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_gpus_larger_than_device_count_is_an_error_not_a_smaller_run():
    started = []
    rc = bench.launch_ranks(8, [], device_count=1, popen=lambda *a, **k: started.append(a))
    assert rc != 0 and not started


def test_plain_gpus_n_exits_nonzero_on_this_box(tmp_path):
    """`python bench.py --gpus 8` where fewer GPUs exist: non-zero, no JSON line, says why (here: 0 or 1 GPU)."""
    if torch.cuda.device_count() >= 8:
        pytest.skip("8 GPUs visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0
    assert "n_gpus" not in p.stdout
    assert "--gpus 8" in p.stderr


class _FakeProc(object):
    def __init__(self, rc, log):
        self.rc, self.pid, self.log, self.terminated = rc, 1000 + len(log), log, False
        log.append(self)

    def poll(self):
        return self.rc

    def terminate(self):
        self.terminated = True


def test_launcher_starts_n_ranks_with_the_rendezvous_environment():
    log, envs = [], []

    def popen(cmd, env):
        envs.append(env)
        assert cmd[0] == sys.executable and cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "3"]
        return _FakeProc(0, log)
    assert bench.launch_ranks(4, ["--gpus", "4", "--steps", "3"], device_count=8, popen=popen) == 0
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)


def test_launcher_fails_when_a_rank_fails():
    log = []
    rcs = iter([0, 3])
    assert bench.launch_ranks(2, [], device_count=2, popen=lambda cmd, env: _FakeProc(next(rcs), log)) == 3


def test_world_size_mismatch_is_refused(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "8"])
    assert "WORLD_SIZE" in str(e.value)


def test_result_changing_switches_are_refused():
    args = bench.parse_args([])
    for k in ("MOLANN_ELIDE_INVARIANT_ALIGNMENT", "MOLANN_DEBUG_ABLATE", "MOLANN_DEBUG_JIT_WAVES"):
        with pytest.raises(SystemExit) as e:
            bench.check_env(args, {k: "1", "HOME": "/"})
        assert k in str(e.value)
    env, diag = bench.check_env(args, {"MOLANN_NO_JIT": "1", "MOLANN_DEBUG_ABLATE": "0", "PATH": "x"})
    assert env == {"MOLANN_DEBUG_ABLATE": "0", "MOLANN_NO_JIT": "1"} and diag is False
    env, diag = bench.check_env(bench.parse_args(["--diagnostic"]), {"MOLANN_DEBUG_ABLATE": "64"})
    assert diag is True and env == {"MOLANN_DEBUG_ABLATE": "64"}


def test_c1_cpu_baseline_runs_at_c1s_batch():
    from molann_amd import workloads as wl
    from bench_standin import CpuStandIn
    w = wl.get_workload("C1")
    model = CpuStandIn().build_model(w)
    rec = bench.cpu_baseline(w, model.module, 0.5)
    assert " 1024 frames in 1024-frame chunks" in rec["sample"] and rec["value"] > 0 and rec["kind"] == "port"


def _rank(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from bench_standin import CpuStandIn
    side = CpuStandIn()
    side.init_process_group()
    try:
        args = bench.parse_args(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--frames", "203", "--workload", "C3",
                                 "--gather-mode", mode, "--gather-chunks", "3"])
        lines = []
        rec = bench.run_rank(args, side, world, rank, True, env={}, emit=lines.append)
        # the gathered outputs of the last step, checked against the whole trajectory computed in one piece
        from build_util import oracle_for_workload
        from molann_amd import workloads as wl
        from molann_amd.dist import OutputGather
        w = wl.get_workload("C3")
        model = side.build_model(w)
        xs = [w.make_frames(203, seed=77 + r) for r in range(world)]
        g = OutputGather(203, w.out_dim(), torch.device("cpu"), world, rank, chunks=3)
        a = g.forward_overlapped(model, xs[rank]).clone()
        b = g.collective(model(xs[rank])).clone()
        want = torch.cat([oracle_for_workload(w, model.module, x) for x in xs])
        ok = torch.equal(a, b) and torch.allclose(a, want, atol=1e-6) and a.shape == (203 * world, 8)
        q.put((rank, ok, lines, rec is not None, side.calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["collective", "overlap", "auto"])
def test_rank_code_at_world_size_2_over_gloo(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = bench._free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert res[1][2] == [] and res[1][3] is False          # only rank 0 prints
    (line,) = res[0][2]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["config"]["final_allgather"] is True
    assert d["config"]["dist"] == {"world_size": 2, "backend": "gloo", "devices": ["cpu (oracle stand-in)"] * 2,
                                   "launched_by": d["config"]["dist"]["launched_by"], "cpu_affinity": d["config"]["dist"]["cpu_affinity"]}
    assert abs(d["value"] - 203 * 2 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]
    ph = d["phases"]
    assert ph["allgather_mode"] in ("collective", "overlap") and (mode == "auto" or ph["allgather_mode"] == mode)
    assert ph["allgather_bytes_received_per_rank"] == 203 * 8 * 4
    assert ph["last_step_plus_allgather_ms"] > 0 and ph["compute_ms_per_step"] > 0
    if mode == "auto":
        assert set(ph["allgather_warmup_ms"]) == {"collective", "overlap"}
    assert "cpu_baseline" not in d


def _fake_kfd(tmp_path, gpus=4):
    root = tmp_path / "nodes"
    for i in range(gpus + 1):
        d = root / str(i)
        d.mkdir(parents=True)
        simd = 0 if i == 0 else 1024        # node 0: the CPU
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nunique_id %d\ngfx_target_version %d\n"
                                      % (64 if i == 0 else 0, simd, 0 if i == 0 else 0xabc0 + i, 0 if i == 0 else 90500))
    return str(root)


def test_device_count_comes_from_sysfs_and_honours_visibility(tmp_path):
    """the launcher counts GPUs from the KFD topology in sysfs (no HIP call in the parent), narrowed by
    ROCR_VISIBLE_DEVICES, then HIP_VISIBLE_DEVICES (or CUDA_VISIBLE_DEVICES when HIP_ is absent)."""
    root = _fake_kfd(tmp_path, gpus=4)
    assert bench.visible_gpu_count({}, root) == 4
    assert bench.visible_gpu_count({"HIP_VISIBLE_DEVICES": "0,2"}, root) == 2
    assert bench.visible_gpu_count({"HIP_VISIBLE_DEVICES": "1,7,2"}, root) == 1          # the list ends at the first invalid entry
    assert bench.visible_gpu_count({"ROCR_VISIBLE_DEVICES": "1,2,3", "HIP_VISIBLE_DEVICES": "2"}, root) == 1
    assert bench.visible_gpu_count({"ROCR_VISIBLE_DEVICES": "1,2,3", "HIP_VISIBLE_DEVICES": "3"}, root) == 0
    assert bench.visible_gpu_count({"CUDA_VISIBLE_DEVICES": "0"}, root) == 1
    assert bench.visible_gpu_count({"CUDA_VISIBLE_DEVICES": "0", "HIP_VISIBLE_DEVICES": "0,1,2"}, root) == 3
    assert bench.visible_gpu_count({"ROCR_VISIBLE_DEVICES": "GPU-abc2,GPU-abc4"}, root) == 2
    assert bench.visible_gpu_count({"HIP_VISIBLE_DEVICES": ""}, root) == 0


def test_ranks_get_disjoint_core_slices():
    envs, log = [], []
    bench.launch_ranks(4, [], device_count=4, popen=lambda cmd, env: (envs.append(env), _FakeProc(0, log))[1])
    slices = [set(int(c) for c in e["MOLANN_BENCH_CPU_SLICE"].split(",")) for e in envs]
    have = len(os.sched_getaffinity(0))
    if have >= 4:
        assert all(len(s) == have // 4 for s in slices)
        assert len(set().union(*slices)) == sum(len(s) for s in slices)      # disjoint
    assert bench.cpu_slice(2, 4, range(16)) == [8, 9, 10, 11] and bench.cpu_slice(0, 8, range(4)) == [0, 1, 2, 3]


def test_traffic_is_quoted_only_for_the_kernel_it_was_measured_on():
    a = "molann_lane_jit<NL=2> (plan-specialised; 14 consumer waves + 2 loader, ring of 15 tiles) grid=256 block=1024 lds=150000"
    b = "molann_lane_jit<NL=2> (plan-specialised; 10 consumer waves + 2 loader, ring of 12 tiles) grid=256 block=768 lds=120000"
    assert bench.kernel_names(a) == bench.kernel_names(b) == "molann_lane_jit<NL=2>"
    assert bench.kernel_names("frames_ring_kernel<ND=8> (12 consumer) grid=256 || molann_mlp_chain<f32,FB=4> (plan-specialised) chunk=262144") \
        == "frames_ring_kernel<ND=8> || molann_mlp_chain<f32,FB=4>"
    assert bench.kernel_names("frames_lane_kernel<2,features_regs> grid=512") != bench.kernel_names(a)
