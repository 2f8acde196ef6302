"""A CPU stand-in for bench.py's `HipSide`, for tests only: lets the gloo tests drive bench.run_rank - its sharding,
its timed region, the two forms of the final all-gather, the max-over-ranks reduction and the JSON line - without a
GPU.  The "model" is the oracle (test infrastructure); nothing under molann_amd/ or bench.py's command line can
reach this file."""

import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


class CpuStandIn(object):
    backend = "gloo"
    index = 0

    def __init__(self):
        self.device = torch.device("cpu")
        self.calls = 0

    def init_process_group(self):
        import torch.distributed as dist
        dist.init_process_group(backend=self.backend)

    def barrier(self):
        import torch.distributed as dist
        dist.barrier()

    def describe(self):
        return "cpu (oracle stand-in)"

    def build_model(self, w):
        from build_util import oracle_for_workload, workload_model
        module = workload_model(w)          # product module, CPU: only its weights are used

        def model(x):
            self.calls += 1
            return oracle_for_workload(w, module, x, torch.float32)
        model.module = module
        return model

    def make_frames(self, w, n, seed):
        return w.make_frames(n, seed=seed)

    def sync(self):
        pass

    def mark(self):
        return time.perf_counter()

    def wait_mark(self, e):
        pass

    def ms_between(self, a, b):
        return (b - a) * 1e3

    def library(self):
        return "none (oracle stand-in)"

    def kernels(self, model):
        return "oracle stand-in (tests only)"
