#!/usr/bin/env python3
"""Diagnostic: where a wave of frames_lane_kernel spends its cycles (s_memtime stamps, bit 32 of
MOLANN_DEBUG_ABLATE).  Shares only; the stamped build is slower than the production kernel."""
import ctypes, os, sys
os.environ["MOLANN_DIAG_LIB"] = "1"      # the diagnostics build of the library (make -C molann_amd/csrc diag)
os.environ["MOLANN_DEBUG_ABLATE"] = str(32 | int(os.environ.get("EXTRA_ABLATE", "0")))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import _capi, ann, workloads as wl
ann._RUN_OP.append(None)                  # ctypes path only: the operator library is linked against the release build
w = wl.get_workload(sys.argv[1] if len(sys.argv) > 1 else "C3")
dev = torch.device("cuda:0")
model = wl.build_model(w, dev).requires_grad_(False)
x = w.make_frames(w.frames, device=dev, seed=1)
buf = (ctypes.c_ulonglong * 16)()
with torch.no_grad():
    model(x); torch.cuda.synchronize()
    _capi.lib().molann_debug_read_stamps(buf)
    model(x); torch.cuda.synchronize()
    _capi.lib().molann_debug_read_stamps(buf)
names = ["wait DMA+stores", "reg fill + DMA issue", "covariance", "rotation solve", "feature table", "MLP + stores"]
waves = buf[7]; tiles = (w.frames + 63) // 64
tot = sum(buf[i] for i in range(6))
print("waves", waves, "tiles", tiles, "cycles per tile per wave:")
for i, n in enumerate(names):
    print("  %-22s %9.0f  (%4.1f%%)" % (n, buf[i] / tiles, 100.0 * buf[i] / max(1, tot)))
print("  %-22s %9.0f" % ("total", tot / tiles))
if buf[11]:
    lt = max(1, buf[12])
    print("loader waves %d, cycles per tile issued: slot wait / poll %.0f, DMA issue %.0f, landing wait + publish %.0f"
          % (buf[11], buf[8] / lt, buf[9] / lt, buf[10] / lt))
if buf[11]:
    print("loaders published their last tile after %.1f us on average; the last consumer of the chip left after %.1f us"
          % (buf[13] / buf[11] / 100.0, buf[15] / 100.0))
rt, ck = buf[6] >> 32, (buf[6] & 0xffffffff) << 8
if rt:
    print("shader clock held: %.0f MHz (sum over waves of s_memtime / s_memrealtime x 100 MHz); wave lifetime %.1f us" % (ck / rt * 100.0, rt / waves / 100.0))
