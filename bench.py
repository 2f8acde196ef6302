#!/usr/bin/env python3
"""Throughput of the molann forward path on MI355X: frames/s (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3] [--frames F]

A "step" is one forward of the workload's model over one batch of synthetic frames that already sit in
HBM.  Default workload C3 = the full MolANN forward on the 22-atom alanine dipeptide (Kabsch on the 7
backbone atoms + 4 features + MLP [6,32,8]) over 1,048,576 frames per GPU; C1/C2/C4/C5 (BASELINE.json's
other configs) are selectable.  Inputs rotate over several distinct buffers (> 1 GiB in total for the
22-atom configs) so that timed reads come from HBM, not from the 256 MiB Infinity Cache.

N > 1: one process per GPU over RCCL.  Either the driver starts the ranks (torch.distributed.run sets
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), or `python bench.py --gpus N` run plainly starts them itself:
the parent counts the devices in sysfs (KFD topology, narrowed by *_VISIBLE_DEVICES: no HIP call, /dev/kfd never opened),
spawns N children with that environment, each pinned to its own slice of the host's cores, and exits with their status
(a rank that fails takes the run down, non-zero).
Frames are sharded, every rank runs the same K steps on its own shard (weak scaling, no data-path
collective) and ONE all-gather of the last step's output shards closes the timed region (BASELINE.json:
"RCCL all-gather of outputs ... only at the end").  That gather either follows the last step as one
`all_gather_into_tensor` ("collective"), or runs beside it ("overlap": the last step goes in chunks and chunk i
travels - each peer's rows straight into their final place over that peer's own xGMI link - while chunk i+1
computes); the warm-up times both on the live communicator and the timed region uses the faster one.
The time is the max over ranks; value = all frames of all ranks / that time.

One JSON line on rank 0; besides the contract's keys it carries
  roofline      the dominant kernel against the 8 TB/s HBM roof: algorithmic bytes per launch (SURVEY.md
                8(d): 12 B x touched atoms + 4 B x d_out per frame) / the average launch duration measured
                with HIP events on the launch stream around the timed steps
  cpu_baseline  the oracle (composite-PyTorch restatement of the reference's op sequence) timed on this
                box's host cores on a bounded sample (N = 1 only)
  config.env    every MOLANN_* variable present; a variable that changes what is computed
                (MOLANN_ELIDE_INVARIANT_ALIGNMENT, MOLANN_DEBUG_*) makes the bench refuse to report a
                value unless --diagnostic is given, and the line then says "diagnostic": true
  config.dist   world size, backend and the device of every rank, as torch.distributed reports them
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from molann_amd import workloads as wl  # noqa: E402
from molann_amd import dist as mdist  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0       # measured float4 copy (same guide)

# variables that change WHAT is computed (or skip part of it): a line measured under one is not a result
RESULT_CHANGING_ENV = ("MOLANN_ELIDE_INVARIANT_ALIGNMENT", "MOLANN_DEBUG_", "MOLANN_DIAG_LIB", "MOLANN_JIT_EXTRA_FLAGS")


def parse_args(argv=None):
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
    ap.add_argument("--gather-mode", default="auto", choices=("auto", "collective", "overlap"),
                    help="N>1: how the one all-gather is issued (auto: both timed in the warm-up, faster one kept)")
    ap.add_argument("--gather-chunks", type=int, default=4, help="overlap mode: pieces the last step is cut into")
    ap.add_argument("--diagnostic", action="store_true",
                    help="allow result-changing MOLANN_* switches; the line is then marked diagnostic")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------
# environment certification
# ------------------------------------------------------------------------------------------------------

def molann_env(environ=None):
    environ = os.environ if environ is None else environ
    return {k: environ[k] for k in sorted(environ) if k.startswith("MOLANN_")}


def result_changing(env):
    """The MOLANN_* switches in `env` that alter or skip part of the computation (set to anything but ''/'0')."""
    bad = []
    for k, v in env.items():
        if any(k == p or (p.endswith("_") and k.startswith(p)) for p in RESULT_CHANGING_ENV) and v not in ("", "0"):
            bad.append(k)
    return bad


def check_env(args, environ=None):
    env = molann_env(environ)
    bad = result_changing(env)
    if bad and not args.diagnostic:
        raise SystemExit("bench.py: %s set - that changes what the kernels compute; no value is reported "
                         "(pass --diagnostic for a line marked as such)" % ", ".join(bad))
    return env, bool(bad)


def _kfd_gpu_nodes(root="/sys/class/kfd/kfd/topology/nodes"):
    """GPU nodes of the KFD topology (a node with SIMDs is a GPU, one without is a CPU), in node order; None if the
    topology is not readable.  Reading sysfs opens no device file."""
    import glob
    nodes = []
    paths = sorted(glob.glob(os.path.join(root, "*", "properties")), key=lambda q: int(os.path.basename(os.path.dirname(q))))
    if not paths:
        return None
    for q in paths:
        try:
            props = dict(ln.split(None, 1) for ln in open(q).read().splitlines() if " " in ln)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0:
            nodes.append(props)
    return nodes


def visible_gpu_count(environ=None, root="/sys/class/kfd/kfd/topology/nodes"):
    """How many GPUs a rank started from this environment will see, WITHOUT a HIP / HSA call in this process: the KFD
    topology in sysfs, narrowed by ROCR_VISIBLE_DEVICES and then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (each a
    comma-separated list of indices into the previous view, or GPU-<uuid> names; the list ends at the first invalid
    entry, as the runtimes read it).  Where sysfs is not there the count comes from a CHILD process."""
    environ = os.environ if environ is None else environ
    nodes = _kfd_gpu_nodes(root)
    if nodes is None:
        try:
            out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                                 text=True, timeout=300, env=dict(environ))
            return int(out.stdout.strip().splitlines()[-1])
        except Exception:
            return 0
    view = ["%x" % int(p.get("unique_id", "0")) for p in nodes]
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = environ.get(var)
        if val is None:
            continue
        if var == "CUDA_VISIBLE_DEVICES" and environ.get("HIP_VISIBLE_DEVICES") is not None:
            continue
        picked = []
        for tok in [t.strip() for t in val.split(",")]:
            if tok.isdigit() and int(tok) < len(view):
                picked.append(view[int(tok)])
            elif tok.upper().startswith("GPU-") and tok[4:].lower().lstrip("0") in [v.lstrip("0") for v in view]:
                picked.append(tok[4:].lower())
            else:
                break
        view = picked
    return len(view)


def core_ranges(cores):
    """'0-15,32-47' for a set of core numbers."""
    cs = sorted(cores)
    out, i = [], 0
    while i < len(cs):
        j = i
        while j + 1 < len(cs) and cs[j + 1] == cs[j] + 1:
            j += 1
        out.append("%d" % cs[i] if i == j else "%d-%d" % (cs[i], cs[j]))
        i = j + 1
    return ",".join(out)


def cpu_slice(rank, n_ranks, cores=None):
    """The contiguous share of this process's allowed cores that rank `rank` of `n_ranks` pins itself to."""
    cores = sorted(os.sched_getaffinity(0)) if cores is None else sorted(cores)
    per = len(cores) // max(1, n_ranks)
    if per < 1:
        return cores
    return cores[rank * per:(rank + 1) * per]


# ------------------------------------------------------------------------------------------------------
# the parent of a plain `bench.py --gpus N`: starts the ranks, never touches the GPU
# ------------------------------------------------------------------------------------------------------

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, device_count=None, popen=subprocess.Popen):
    """Start `n` rank processes of this script (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their environment) and
    return the exit status of the run: 0 only if every rank returned 0.  The caller has made no GPU call."""
    have = visible_gpu_count() if device_count is None else device_count   # sysfs: no HIP / HSA call, no /dev/kfd in this process
    if have < n:
        sys.stderr.write("bench.py: --gpus %d but this node shows %d GPU(s); not running a smaller job under "
                         "that name\n" % (n, have))
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   MOLANN_BENCH_CPU_SLICE=",".join(str(c) for c in cpu_slice(r, n)))
        procs.append(popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    status, live = 0, list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                sys.stderr.write("bench.py: rank process %d exited with %d; stopping the others\n" % (p.pid, rc))
                for q in live:
                    q.terminate()
        time.sleep(0.05)
    return status


# ------------------------------------------------------------------------------------------------------
# what a rank runs on: the MI355X (the only thing the command line ever builds)
# ------------------------------------------------------------------------------------------------------

class HipSide(object):
    """Device, model, frames, timers and synchronisation of one rank on its MI355X."""
    backend = "nccl"

    def __init__(self, local_rank):
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU path)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("bench.py: local rank %d but %d GPU(s) visible" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)
        self.index = local_rank

    def init_process_group(self):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "WORLD_SIZE" not in os.environ:      # the one-rank rehearsal (MOLANN_BENCH_FORCE_DIST=1) run plainly
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(self.index))
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist.init_process_group(backend=self.backend, device_id=self.device)

    def barrier(self):
        import torch.distributed as dist
        dist.barrier(device_ids=[self.index])

    def describe(self):
        p = torch.cuda.get_device_properties(self.device)
        return "cuda:%d %s (%s)" % (self.index, p.name, getattr(p, "gcnArchName", "?"))

    def build_model(self, w):
        model = wl.build_model(w, self.device)
        model.requires_grad_(False)
        return model

    def make_frames(self, w, n, seed):
        return w.make_frames(n, device=self.device, seed=seed)

    def sync(self):
        torch.cuda.synchronize()

    def mark(self):
        e = torch.cuda.Event(enable_timing=True)   # on the current stream = the stream the kernels are launched on
        e.record()
        return e

    def wait_mark(self, e):
        e.synchronize()

    def ms_between(self, a, b):
        return a.elapsed_time(b)

    def kernels(self, model):
        from molann_amd.ann import last_launch_info
        return last_launch_info(model)

    def library(self):
        from molann_amd import _capi
        return _capi.build_kind()      # "release": the product library reads no result-changing switch


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
    """The oracle on the host cores over the workload's own batch: the 1M-frame configs in 64k-frame chunks
    (a 262 144-frame sample), C1 at its batch of 1024, 5000-atom frames 2048 at a time."""
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
    n = min(w.frames, 1 << 18) if small else 2048
    chunk = min(n, 1 << 16) if small else 512
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
    most = 7 if n >= (1 << 16) else 2001
    while len(times) < most and (not times or (time.perf_counter() - t_start) < seconds):
        times.append(one_pass())
    times.sort()
    med = times[len(times) // 2]
    return {"value": n / med, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d passes over %d frames in %d-frame chunks, median; torch %s, %d threads, fp32, no_grad"
                      % (len(times), n, chunk, torch.__version__, cores)}


# ------------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------------

def kernel_names(info):
    """The kernels of a launch-info string with their template arguments, without the launch geometry: what identifies the code
    that ran ('molann_lane_jit<NL=2>', 'frames_ring_kernel<ND=8> || molann_mlp_chain<f32,FB=4>')."""
    import re
    return " || ".join(m.group(0) for m in re.finditer(r"\b(?:molann_|frames_|mlp_)\w+(?:<[^>]*>)?", info or ""))


def pick_buffers(args, buf_bytes):
    nbuf = args.buffers if args.buffers > 0 else max(2, min(8, -(-(5 << 28) // max(1, buf_bytes))))  # > 1.25 GiB in total
    if buf_bytes > (8 << 30) and args.buffers <= 0:
        nbuf = 2
    return nbuf


def run_rank(args, side, world, rank, distributed, env=None, diagnostic=False, emit=print):
    """K timed steps of the workload on this rank's shard (+ the one all-gather when distributed) and, on rank 0, the
    JSON line.  `side` supplies device, model, frames, timers (HipSide on the command line; tests hand in a CPU
    stand-in to drive this very code over gloo)."""
    import torch.distributed as dist
    w = wl.get_workload(args.workload)
    frames = args.frames if args.frames > 0 else w.frames
    model = side.build_model(w)
    gather = distributed and not args.no_gather

    # ---- inputs resident in HBM before the clock starts ---------------------------------------
    nbuf = pick_buffers(args, frames * w.n_atoms * 12)
    xs = [side.make_frames(w, frames, w.seed + 1000 * rank + i) for i in range(nbuf)]
    side.sync()
    barrier = side.barrier if distributed else (lambda: None)

    gat = None
    if gather:
        gat = mdist.OutputGather(frames, w.out_dim(), xs[0].device, world, rank, chunks=max(1, args.gather_chunks))

    def last_step(x, mode):
        """The K-th step and the all-gather of its outputs; returns the gathered [N, d_out]."""
        if mode == "overlap":
            return gat.forward_overlapped(model, x)
        return gat.collective(model(x))

    with torch.no_grad():
        for i in range(args.warmup):
            y = model(xs[i % nbuf])
        tune = None
        mode = args.gather_mode
        if gather:
            # both forms once untimed (communicator set-up, first-use allocations), then timed on the live communicator
            tune = {}
            for m in (("collective", "overlap") if mode == "auto" else (mode,)):
                last_step(xs[0], m)
                side.sync()
                barrier()
                ts = []
                for rep in range(3):
                    side.sync()
                    t0 = time.perf_counter()
                    last_step(xs[rep % nbuf], m)
                    side.sync()
                    ts.append(time.perf_counter() - t0)
                tune[m] = sorted(ts)[1] * 1e3
            if mode == "auto":
                t = torch.tensor([tune["collective"], tune["overlap"]], dtype=torch.float64, device=xs[0].device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)          # every rank takes the same decision
                tune = {"collective": float(t[0]), "overlap": float(t[1])}
                mode = "overlap" if tune["overlap"] < tune["collective"] else "collective"
        side.sync()

        # ---- the timed region: barrier + synchronize on both sides, EXACTLY K steps (+ the one all-gather) ------
        barrier()
        side.sync()
        t0 = time.perf_counter()
        ev0 = side.mark()
        n_plain = args.steps - 1 if gather else args.steps
        for i in range(n_plain):
            y = model(xs[i % nbuf])
        ev1 = side.mark()
        if gather:
            y_all = last_step(xs[n_plain % nbuf], mode)
        ev2 = side.mark()
        side.sync()
        elapsed = time.perf_counter() - t0      # this rank's K steps; the job's time is the max over ranks (below)
        barrier()
        kernel_ms = side.ms_between(ev0, ev1) / n_plain if n_plain > 0 else None
        tail_ms = side.ms_between(ev1, ev2) if gather else 0.0   # last step + all-gather, however they were interleaved
        if gather:
            assert tuple(y_all.shape) == (frames * world, w.out_dim())

        # individually synchronised launches (outside the timed region): each includes the host's launch latency
        per = []
        for i in range(min(args.steps, 20)):
            a = side.mark()
            model(xs[i % nbuf])
            b = side.mark()
            side.wait_mark(b)
            per.append(side.ms_between(a, b))
        per.sort()
        side.sync()
        if kernel_ms is None:       # --steps 1 with the gather: no plain step inside the timed region
            kernel_ms = per[len(per) // 2]

    devices = [side.describe()]
    if distributed:
        tmax = torch.tensor([elapsed, kernel_ms, tail_ms], dtype=torch.float64, device=xs[0].device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms, tail_ms = (float(v) for v in tmax)
        devices = [None] * world
        dist.all_gather_object(devices, side.describe())

    rec = None
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
                # counters taken on another kernel (or another launch size) say nothing about this run: the figure is quoted only
                # when the kernel that ran is the one the PMC passes profiled
                t = json.load(open(tfile)).get(w.name, {})
                if t.get("frames_per_launch") in (None, frames) and kernel_names(side.kernels(model)) == kernel_names(t.get("kernels", "")):
                    traffic = t.get("bytes_per_launch")
            except Exception:
                traffic = None
        rec = {
            "metric": "frames/sec (MolANN forward, 22-atom ala-dipeptide)" if w.n_atoms == 22 else
                      "frames/sec (MolANN forward, %d-atom system)" % w.n_atoms,
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if w.mlp_dtype != "bf16" else "f32 (Kabsch 3x3 in f64) + bf16 MLP",
            "data": "synthetic",
            "config": {"workload": "%s: %s, %d frames/GPU" % (w.name, w.description, frames), "frames_per_gpu": frames,
                       "n_atoms": w.n_atoms, "align_atoms": len(w.align) if w.align else 0,
                       "features": len(w.features), "feature_dim": w.feature_dim(),
                       "mlp": w.mlp_dims, "input_buffers": nbuf, "parallelism": "frames sharded x%d" % world,
                       "final_allgather": bool(gather), "kernels": side.kernels(model), "library": side.library(),
                       "env": env if env is not None else molann_env(),
                       "dist": {"world_size": dist.get_world_size() if distributed else 1,
                                "backend": dist.get_backend() if distributed else None,
                                "devices": devices, "launched_by": os.environ.get("MOLANN_BENCH_LAUNCHER", "self" if not distributed else "external"),
                                "cpu_affinity": core_ranges(os.sched_getaffinity(0))}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_frame": alg_bytes, "dense_bytes_per_frame": dense_bytes,
                         "dense_GBps": dense_bytes * frames / (kernel_ms * 1e-3) / 1e9,
                         "dense_frac_of_measured_copy_bw": dense_bytes * frames / (kernel_ms * 1e-3) / 1e9 / HBM_COPY_GBS,
                         "launch_ms_avg": kernel_ms,
                         "synced_launch_ms_min": per[0], "synced_launch_ms_median": per[len(per) // 2]},
        }
        if diagnostic:
            rec["diagnostic"] = True
        if distributed:
            # value (above) is the contract's number: K steps with the all-gather of the K-th step's outputs, max over
            # ranks.  The same run split into its phases (HIP events, max over ranks), because one all-gather of
            # 7 x [frames, d_out] per rank over xGMI costs as much as several 22-atom steps:
            rec["phases"] = {"compute_ms_per_step": kernel_ms, "compute_frames_per_s": frames * world / (kernel_ms * 1e-3),
                             "last_step_plus_allgather_ms": tail_ms,
                             "allgather_exposed_ms": max(0.0, tail_ms - kernel_ms) if gather else 0.0,
                             "allgather_mode": mode if gather else None,
                             "allgather_chunks": gat.chunks if (gather and mode == "overlap") else (1 if gather else 0),
                             "allgather_warmup_ms": tune,
                             "allgather_bytes_received_per_rank": (world - 1) * frames * w.out_dim() * 4 if gather else 0}
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(w, model, args.cpu_seconds)
            rec["gpu_over_cpu"] = value / rec["cpu_baseline"]["value"]
        emit(json.dumps(rec))
    return rec


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    env, diagnostic = check_env(args)
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        # plain `bench.py --gpus N`: become the launcher.  Nothing above has touched the GPU.
        os.environ["MOLANN_BENCH_LAUNCHER"] = "bench.py"
        sys.exit(launch_ranks(args.gpus, argv))
    world = int(world_env or "1")
    if world_env is not None and args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("MOLANN_BENCH_FORCE_DIST") == "1"   # the latter: rehearse the RCCL calls on one rank
    if os.environ.get("MOLANN_BENCH_CPU_SLICE"):      # the launcher's share of the host cores for this rank
        try:
            os.sched_setaffinity(0, {int(c) for c in os.environ["MOLANN_BENCH_CPU_SLICE"].split(",")})
        except (OSError, ValueError):
            pass
    side = HipSide(local_rank)
    if distributed:
        side.init_process_group()
    try:
        run_rank(args, side, world, rank, distributed, env=env, diagnostic=diagnostic)
    finally:
        if distributed:
            import torch.distributed as dist
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
