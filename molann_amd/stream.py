"""Host-resident trajectories: pinned, double-buffered H2D copies overlapped with the forward.

At 264 B per 22-atom frame the PCIe Gen5 link (63 GB/s spec) carries ~0.24 G frames/s while the kernels
consume 13-20 G frames/s from HBM, so a trajectory that starts in host memory is transfer-bound by ~50x
(SURVEY.md 8(f)-4).  `stream_forward` keeps the link busy: chunk i+1 is staged into a pinned buffer and copied
on a copy stream while chunk i runs on the compute stream; outputs return through a second pinned buffer.
Selecting only the atoms a model touches on the host (`columns=`) shrinks 5000-atom frames before they cross
the link.
"""

import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


def _host_threads():
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def _parallel_copy(pool, nthreads, dst, src, columns):
    """pageable -> pinned staging copy split over host threads (one memcpy thread moves only ~5 GB/s)."""
    m = src.shape[0]
    step = -(-m // nthreads)

    def work(a):
        b = min(m, a + step)
        if columns is None:
            np.copyto(dst[a:b], src[a:b], casting="same_kind")
        else:
            np.copyto(dst[a:b], src[a:b][:, columns, :], casting="same_kind")

    list(pool.map(work, range(0, m, step)))


def stream_forward(model, frames, chunk_frames=1 << 18, device=None, out=None, columns=None):
    """Run ``model`` over a host array ``frames`` [N, n_atoms, 3] float32 (numpy array, np.memmap, or a CPU
    torch tensor).  A PINNED torch tensor (``torch.empty(...).pin_memory()``, e.g. filled by the trajectory
    reader) is copied to the device straight from where it lies: no staging copy, the link is the only bound.

    Returns a host numpy array [N, d_out].  ``columns``: optional list of atom indices to keep on the host
    (the model must then have been built for that reduced input group).
    """
    dev = torch.device(device if device is not None else "cuda")
    direct = isinstance(frames, torch.Tensor) and frames.is_pinned() and columns is None and \
        frames.dtype == torch.float32 and frames.is_contiguous()
    if isinstance(frames, torch.Tensor) and not direct:
        frames = frames.detach().numpy()
    n = int(frames.shape[0])
    n_atoms = len(columns) if columns is not None else int(frames.shape[1])
    if n == 0:
        with torch.no_grad():
            d_out = model(torch.zeros((0, n_atoms, 3), device=dev)).shape[1]
        return np.zeros((0, d_out), np.float32)
    chunk = int(min(chunk_frames, n))
    pin_in = [None, None] if direct else [torch.empty((chunk, n_atoms, 3), dtype=torch.float32).pin_memory() for _ in range(2)]
    dev_in = [torch.empty((chunk, n_atoms, 3), dtype=torch.float32, device=dev) for _ in range(2)]
    copy_s, comp_s, back_s = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    in_ready = [torch.cuda.Event() for _ in range(2)]     # H2D of slot done
    in_free = [torch.cuda.Event() for _ in range(2)]      # compute on slot done: device slot reusable
    out_done = [torch.cuda.Event() for _ in range(2)]     # D2H of slot done: pinned output slot reusable
    pin_out, result = None, out
    starts = list(range(0, n, chunk))
    nthreads = _host_threads()
    pool = ThreadPoolExecutor(max_workers=nthreads)
    pin_np = [None, None] if direct else [p.numpy() for p in pin_in]

    def stage(i):
        s = starts[i]
        m = min(chunk, n - s)
        slot = i & 1
        if i >= 2:
            in_free[slot].synchronize()                   # the pinned + device slot were consumed
        if not direct:
            _parallel_copy(pool, nthreads, pin_np[slot][:m], frames[s:s + m], columns)
        with torch.cuda.stream(copy_s):
            dev_in[slot][:m].copy_(frames[s:s + m] if direct else pin_in[slot][:m], non_blocking=True)
            in_ready[slot].record(copy_s)
        return m

    sizes = {0: stage(0)}
    with torch.no_grad():
        for i, s in enumerate(starts):
            slot = i & 1
            m = sizes[i]
            if i + 1 < len(starts):
                sizes[i + 1] = stage(i + 1)               # next chunk crosses the link while this one computes
            with torch.cuda.stream(comp_s):
                comp_s.wait_event(in_ready[slot])
                y = model(dev_in[slot][:m])
                in_free[slot].record(comp_s)
            if pin_out is None:
                d_out = int(y.shape[1])
                pin_out = [torch.empty((chunk, d_out), dtype=torch.float32).pin_memory() for _ in range(2)]
                if result is None:
                    result = np.empty((n, d_out), np.float32)
            if i >= 2:
                out_done[slot].synchronize()
                result[starts[i - 2]:starts[i - 2] + sizes[i - 2]] = pin_out[slot][:sizes[i - 2]].numpy()
            with torch.cuda.stream(back_s):
                back_s.wait_stream(comp_s)
                pin_out[slot][:m].copy_(y, non_blocking=True)
                y.record_stream(back_s)
                out_done[slot].record(back_s)
        for i in range(max(0, len(starts) - 2), len(starts)):
            slot = i & 1
            out_done[slot].synchronize()
            result[starts[i]:starts[i] + sizes[i]] = pin_out[slot][:sizes[i]].numpy()
    pool.shutdown(wait=True)
    return result
