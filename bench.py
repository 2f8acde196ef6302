#!/usr/bin/env python3
"""Throughput of the molann forward path on MI355X: frames/s (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3] [--frames F]

A "step" is one forward of the workload's model over one batch of synthetic frames that already sit in
HBM.  Default workload C3 = the full MolANN forward on the 22-atom alanine dipeptide (Kabsch on the 7
backbone atoms + 4 features + MLP [6,32,8]) over 1,048,576 frames per GPU; C1/C2/C4/C5 (BASELINE.json's
other configs) are selectable.  Inputs rotate over several distinct buffers (> 1 GiB in total for the
22-atom configs) so that timed reads come from HBM, not from the 256 MiB Infinity Cache.

N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL): frames are sharded, every rank
runs the same K steps on its own shard (weak scaling, no data-path collective) and ONE all-gather of the
last step's output shards closes the timed region (BASELINE.json: "RCCL all-gather of outputs ... only
at the end").  The time is the max over ranks; value = all frames of all ranks / that time.

One JSON line on rank 0; besides the contract's keys it carries
  roofline      the dominant kernel against the 8 TB/s HBM roof: algorithmic bytes per launch (SURVEY.md
                8(d): 12 B x touched atoms + 4 B x d_out per frame) / the average launch duration measured
                with HIP events on the launch stream around the timed steps
  cpu_baseline  the oracle (composite-PyTorch restatement of the reference's op sequence) timed on this
                box's host cores on a bounded sample (N = 1 only)
"""

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from molann_amd import workloads as wl  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0       # measured float4 copy (same guide)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C3", choices=wl.workload_names())
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU (default: the workload's)")
    ap.add_argument("--buffers", type=int, default=0, help="distinct input buffers to rotate over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-gather", action="store_true", help="N>1: leave the final all-gather out")
    return ap.parse_args()


def host_cores():
    """CPU cores this process may actually use: affinity mask and cgroup quota, not the host's total."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // per))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("MOLANN_BENCH_MAX_CORES", "16"))))


def cpu_baseline(w, model, seconds):
    """The oracle on the host cores, full batch in 64k-frame chunks."""
    from oracle import molann_oracle as mo
    cores = host_cores()
    torch.set_num_threads(cores)
    feats = [(t, [a - 1 for a in atoms]) for t, atoms in w.features]
    al = [a - 1 for a in w.align] if w.align is not None else None
    ref_x = mo.center_reference(torch.from_numpy(w.ref_xyz[[a - 1 for a in w.align]])) if al else None
    ws = bs = None
    if w.mlp_dims:
        lins = [m for m in model.ann_layers if isinstance(m, torch.nn.Linear)]
        ws = [l.weight.detach().float().cpu() for l in lins]
        bs = [l.bias.detach().float().cpu() for l in lins]
    small = w.n_atoms <= 64
    n = (1 << 18) if small else 2048
    chunk = (1 << 16) if small else 512
    x = w.make_frames(n, seed=4321)

    def one_pass():
        t0 = time.perf_counter()
        with torch.no_grad():
            for s in range(0, n, chunk):
                xs = x[s:s + chunk]
                if w.kind == "align":
                    mo.align_forward(xs, al, ref_x)
                elif w.mlp_dims:
                    mo.molann_forward(xs, feats, ws, bs, w.use_angle_value, al, ref_x)
                else:
                    mo.preprocessing_forward(xs, feats, w.use_angle_value, al, ref_x)
        return time.perf_counter() - t0

    t_start = time.perf_counter()
    one_pass()  # warm-up
    times = []
    while len(times) < 7 and (not times or (time.perf_counter() - t_start) < seconds):
        times.append(one_pass())
    times.sort()
    med = times[len(times) // 2]
    return {"value": n / med, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d passes over %d frames in %d-frame chunks, median; torch %s, %d threads, fp32, no_grad"
                      % (len(times), n, chunk, torch.__version__, cores)}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("MOLANN_BENCH_FORCE_DIST") == "1"   # the latter: rehearse the RCCL calls on one rank
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    w = wl.get_workload(args.workload)
    frames = args.frames if args.frames > 0 else w.frames
    model = wl.build_model(w, dev)
    model.requires_grad_(False)

    # ---- inputs resident in HBM before the clock starts ---------------------------------------
    buf_bytes = frames * w.n_atoms * 12
    nbuf = args.buffers if args.buffers > 0 else max(2, min(8, -(-(5 << 28) // buf_bytes)))  # > 1.25 GiB in total
    if buf_bytes > (8 << 30):
        nbuf = 2 if args.buffers <= 0 else nbuf
    xs = [w.make_frames(frames, device=dev, seed=w.seed + 1000 * rank + i) for i in range(nbuf)]
    torch.cuda.synchronize()

    def barrier():
        if distributed:
            dist.barrier(device_ids=[local_rank])

    with torch.no_grad():
        for i in range(args.warmup):
            y = model(xs[i % nbuf])
        if distributed and not args.no_gather:
            from molann_amd.dist import all_gather_outputs
            all_gather_outputs(y, frames * world)   # equal shards: no size exchange, one collective
        torch.cuda.synchronize()

        ev0, ev1, ev2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        for i in range(args.steps):
            y = model(xs[i % nbuf])
        ev1.record()
        if distributed and not args.no_gather:
            y_all = all_gather_outputs(y, frames * world)
        ev2.record()
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = ev0.elapsed_time(ev1) / args.steps   # average launch duration, launch stream
        gather_ms = ev1.elapsed_time(ev2)                # the one all-gather (0 when there is none)

        # per-launch durations (outside the timed region) for the spread
        per = []
        for i in range(min(args.steps, 20)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            model(xs[i % nbuf])
            b.record()
            b.synchronize()
            per.append(a.elapsed_time(b))
        per.sort()

    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        kmax = torch.tensor([kernel_ms, gather_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
        kernel_ms, gather_ms = float(kmax[0].item()), float(kmax[1].item())

    if rank == 0:
        total_frames = frames * world * args.steps
        value = total_frames / elapsed
        alg_bytes = w.algorithmic_bytes_per_frame()
        dense_bytes = w.dense_bytes_per_frame()
        achieved = alg_bytes * frames / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(w.name, {}).get("bytes_per_launch")
            except Exception:
                traffic = None
        from molann_amd.ann import last_launch_info
        plan_info = last_launch_info(model)
        rec = {
            "metric": "frames/sec (MolANN forward, 22-atom ala-dipeptide)" if w.n_atoms == 22 else
                      "frames/sec (MolANN forward, %d-atom system)" % w.n_atoms,
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if w.mlp_dtype != "bf16" else "f32 (Kabsch 3x3 in f64) + bf16 MLP",
            "data": "synthetic",
            "config": {"workload": "%s: %s" % (w.name, w.description), "frames_per_gpu": frames,
                       "n_atoms": w.n_atoms, "align_atoms": len(w.align) if w.align else 0,
                       "features": len(w.features), "feature_dim": w.feature_dim(),
                       "mlp": w.mlp_dims, "input_buffers": nbuf, "parallelism": "frames sharded x%d" % world,
                       "final_allgather": bool(distributed and not args.no_gather), "kernels": plan_info},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_frame": alg_bytes, "dense_bytes_per_frame": dense_bytes,
                         "dense_GBps": dense_bytes * frames / (kernel_ms * 1e-3) / 1e9,
                         "dense_frac_of_measured_copy_bw": dense_bytes * frames / (kernel_ms * 1e-3) / 1e9 / HBM_COPY_GBS,
                         "launch_ms_avg": kernel_ms, "launch_ms_min": per[0], "launch_ms_median": per[len(per) // 2]},
        }
        if distributed:
            # value (above) is the contract's number: K steps + the final all-gather + both barriers, max over ranks.
            # The same run split into its two phases (HIP events, max over ranks), because one all-gather of
            # 7 x [frames, d_out] per rank over xGMI costs as much as several 22-atom steps:
            rec["phases"] = {"compute_ms_per_step": kernel_ms, "compute_frames_per_s": frames * world / (kernel_ms * 1e-3),
                             "allgather_ms": gather_ms,
                             "allgather_bytes_received_per_rank": (world - 1) * frames * int(y.shape[1]) * 4}
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(w, model, args.cpu_seconds)
            rec["gpu_over_cpu"] = value / rec["cpu_baseline"]["value"]
        print(json.dumps(rec))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
