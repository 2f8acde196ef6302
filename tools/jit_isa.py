#!/usr/bin/env python3
"""Diagnostic (no GPU needed): the plan-specialised lane kernel of a workload as source and gfx950 ISA, with its
resource usage and instruction mix.   python tools/jit_isa.py [C3] [outdir] [--bwd | --mlp-bwd | --ring-bwd | --save-feat | --wide | --chain] [--no-mlp | --dims=6,32,32,8] [extra hipcc flags...]"""
import collections, ctypes, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from molann_amd import _capi, workloads as wl
_desc = _capi.workload_desc

args = [a for a in sys.argv[1:] if not a.startswith("-")]
flags = [a for a in sys.argv[1:] if a.startswith("-") and a not in ("--bwd", "--mlp-bwd", "--ring-bwd", "--no-mlp", "--save-feat", "--wide", "--chain", "--bf16") and not a.startswith("--dims=")]
name = args[0] if args else "C3"
out = args[1] if len(args) > 1 else "/tmp/jit_%s" % name
os.makedirs(out, exist_ok=True)
d, keep = _desc(wl.get_workload(name))
if "--no-mlp" in sys.argv:
    d.n_layers = 0
for a in sys.argv[1:]:
    if a.startswith("--dims="):            # another MLP behind the workload's features: --dims=6,32,32,8
        dims = [int(v) for v in a[7:].split(",")]
        ld = (ctypes.c_int32 * len(dims))(*dims)
        d.n_layers, d.layer_dims = len(dims) - 1, ld
        keep.append(ld)
if "--bf16" in sys.argv:
    d.mlp_precision = _capi.MLP_BF16
buf = ctypes.create_string_buffer(1 << 22)
rc = _capi.lib().molann_debug_jit(ctypes.byref(d), 4 if "--chain" in sys.argv else 128 if "--wide" in sys.argv else 32 if "--save-feat" in sys.argv else 18 if "--ring-bwd" in sys.argv else 10 if "--mlp-bwd" in sys.argv else 2 if "--bwd" in sys.argv else 0, buf, 1 << 22)
assert rc > 0, rc
src = os.path.join(out, "k.hip")
open(src, "w").write(buf.value.decode())
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-I", os.path.join(ROOT, "molann_amd", "csrc"),
       "-save-temps", "-Rpass-analysis=kernel-resource-usage", "-c", "k.hip", "-o", "k.o"] + flags
p = subprocess.run(cmd, cwd=out, capture_output=True, text=True)
for ln in p.stderr.splitlines():
    if "remark" in ln and any(k in ln for k in ("VGPRs:", "AGPRs", "Spill", "Occupancy", "LDS Size", "SGPRs:", "ScratchSize")):
        print(ln.split("remark:")[1].strip())
if p.returncode != 0:
    print(p.stderr[-3000:]); sys.exit(1)
asm = [f for f in os.listdir(out) if f.endswith(".s")][0]
mix = collections.Counter()
for ln in open(os.path.join(out, asm)):
    m = re.match(r"\s+([a-z_0-9]+)\s", ln)
    if m and not m.group(1).startswith("."):
        op = m.group(1)
        key = ("mfma" if "mfma" in op else "trans" if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op) else "v_pk" if op.startswith("v_pk_") else
               "valu_f64" if op.startswith("v_") and "f64" in op else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
               "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other")
        mix[key] += 1
print(out + "/" + asm, dict(mix))
