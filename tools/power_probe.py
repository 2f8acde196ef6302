#!/usr/bin/env python3
"""Diagnostic: board power and DPM clock levels (sysfs / hwmon, where an ordinary user may read them) while the C3
forward runs back to back for a few seconds.  MOLANN_DIAG_LIB=1 MOLANN_DEBUG_ABLATE=2048 (stream only) / 256
(arithmetic only) / unset (the real kernel) to compare what the chip draws and which clocks it holds in each."""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molann_amd import workloads as wl

dev = torch.device("cuda:0")
w = wl.get_workload("C3")
model = wl.build_model(w, dev).requires_grad_(False)
n = int(os.environ.get("FRAMES", 1 << 23))
xs = [w.make_frames(n, device=dev, seed=i) for i in range(2)]
base = sorted(glob.glob("/sys/class/drm/card*/device"))[0]
files = {"sclk": base + "/pp_dpm_sclk", "mclk": base + "/pp_dpm_mclk", "fclk": base + "/pp_dpm_fclk", "socclk": base + "/pp_dpm_socclk"}
hw = sorted(glob.glob(base + "/hwmon/hwmon*/power1_average") + glob.glob(base + "/hwmon/hwmon*/power1_input"))
samples, stop = [], False
def cur(path):
    try:
        for ln in open(path).read().splitlines():
            if ln.strip().endswith("*"):
                return ln.split(":", 1)[1].replace("*", "").strip()
    except Exception:
        return None
def sampler():
    while not stop:
        s = {k: cur(p) for k, p in files.items()}
        try:
            s["power_W"] = int(open(hw[0]).read()) / 1e6 if hw else None
        except Exception:
            s["power_W"] = None
        samples.append(s)
        time.sleep(0.02)
with torch.no_grad():
    for i in range(3): model(xs[i % 2])
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); k = 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    while time.perf_counter() - t0 < 3.0:
        for i in range(20): model(xs[i % 2])
        k += 20
        torch.cuda.synchronize()
    b.record(); b.synchronize()
    stop = True; th.join()
us = a.elapsed_time(b) / k * 1e3 * (1 << 20) / n
pw = [s["power_W"] for s in samples if s["power_W"] is not None]
print("%.2f us per 1M frames over %d launches; power %s W (mean of %d samples, max %s); clocks seen: %s" % (
    us, k, "%.0f" % (sum(pw) / len(pw)) if pw else "n/a", len(pw), "%.0f" % max(pw) if pw else "n/a",
    {key: sorted({s[key] for s in samples if s[key]}) for key in files}))
