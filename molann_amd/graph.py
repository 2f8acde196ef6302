"""HIP-graph replay of a forward for small, latency-bound batches.

A 1024-frame batch (BASELINE config C1) is ~3 us of kernel time but ~16 us of Python + launch overhead per
call.  `GraphedForward` captures one forward (the kernel launches go to the capturing stream: the C ABI only
enqueues on the stream it is given, never allocates or synchronises) into a `torch.cuda.CUDAGraph`
(hipGraph on ROCm) and replays it on fresh input copied into a static buffer.

    g = GraphedForward(model, x_example)     # model on the GPU, parameters frozen / no_grad
    y = g(x)                                 # x.shape == x_example.shape; y is a static buffer, clone to keep

Weights are packed into the plan before capture; call `g.recapture()` after changing parameters.
"""

import torch


class GraphedForward(object):
    def __init__(self, model, example_x, warmup=3):
        assert example_x.is_cuda and example_x.dtype == torch.float32
        self.model = model
        self.static_x = example_x.detach().clone().contiguous()
        self.static_y = None
        self.graph = None
        self._warmup = warmup
        self.recapture()

    def recapture(self):
        side = torch.cuda.Stream(device=self.static_x.device)
        side.wait_stream(torch.cuda.current_stream(self.static_x.device))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(self._warmup):       # builds the plan, packs the weights, compiles the kernel
                self.model(self.static_x)
        torch.cuda.current_stream(self.static_x.device).wait_stream(side)
        torch.cuda.synchronize(self.static_x.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_y = self.model(self.static_x)
        return self

    def __call__(self, x):
        if x.shape != self.static_x.shape:
            raise ValueError("GraphedForward was captured for %s, got %s" % (tuple(self.static_x.shape), tuple(x.shape)))
        self.static_x.copy_(x)
        self.graph.replay()
        return self.static_y
