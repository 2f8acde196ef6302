"""The bench lines committed under profiles/ (written by bench.py on the GPU box; this round's and later) carry every key of the driver's
contract plus the `roofline` and `cpu_baseline` objects; guards the schema against accidental edits of bench.py."""

import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[2-9]_bench_*.json")))
CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"]


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_bench_line(path):
    d = json.load(open(path))
    for k in CONTRACT:
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["frames_per_gpu"] * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "synced_launch_ms_min" in r and "launch_ms_min" not in r
    # self-certifying: the product library, no result-changing switch, the world as torch.distributed saw it
    cfg = d["config"]
    assert cfg["library"] == "release" and "diagnostic" not in d
    assert not [k for k in cfg["env"] if k.startswith("MOLANN_DEBUG_") or k in ("MOLANN_ELIDE_INVARIANT_ALIGNMENT", "MOLANN_DIAG_LIB")]
    assert cfg["dist"]["world_size"] == d["n_gpus"] and len(cfg["dist"]["devices"]) == d["n_gpus"]
    assert str(cfg["frames_per_gpu"]) in cfg["workload"]            # the workload string states the size that ran
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "frames/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_default_workload_is_the_full_forward():
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_C3.json")))
    assert d["config"]["workload"].startswith("C3") and d["config"]["frames_per_gpu"] == 1 << 20
    assert d["config"]["mlp"] == [6, 32, 8] and d["config"]["align_atoms"] == 7
