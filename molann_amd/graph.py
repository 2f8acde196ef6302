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


def _unpin_plans(desc, device):
    try:
        torch.ops.molann.unpin(desc, device)
    except Exception:   # interpreter shutdown / library gone
        pass


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
        self._pin()
        return self

    def _pin(self):
        """The captured launches hold raw pointers into the plan (packed weights, reference, code objects).  A plan the
        module owns (ctypes) lives as long as `self.model`; a MolANN served by the dispatcher operator has its plan in that
        library's LRU cache, which a replay never touches: pin it there for the life of this graph."""
        import weakref
        old = self.__dict__.pop("_unpin", None)
        if old is not None:
            old()
        st = getattr(self.model, "__dict__", {}).get("_fast")
        if st is not None and st.get("fused") and st.get("op") is not None:
            desc, dev = list(st["desc"]), self.static_x.device.index
            torch.ops.molann.pin(desc, dev)
            self._unpin = weakref.finalize(self, _unpin_plans, desc, dev)

    def __call__(self, x):
        if x.shape != self.static_x.shape:
            raise ValueError("GraphedForward was captured for %s, got %s" % (tuple(self.static_x.shape), tuple(x.shape)))
        self.static_x.copy_(x)
        self.graph.replay()
        return self.static_y


class GraphedForces(GraphedForward):
    """Forward and vector-Jacobian product as two HIP graphs, for callers that differentiate a small batch at every step (a
    collective variable inside an MD engine: the values first, then dV/dx for the cotangent dV/d(values) the engine forms).

        g = GraphedForces(model, x_example)      # a MolANN served by one fused plan, parameters frozen
        y = g(x)                                 # static buffer [n, d_out]
        dx = g.vjp(dy)                           # static buffer [n, n_inp, 3]: sum_k dy[:, k] d y[:, k] / d x, for the x of the last g(x)

    The full Jacobian of one frame's d_out values comes from the same two replays on a batch of d_out copies of the frame with the
    identity as cotangent: `g = GraphedForces(model, x.expand(d_out, -1, -1).contiguous()); y = g(xr)[0]; J = g.vjp(torch.eye(d_out))`.

    The backward graph holds one launch of `molann_backward_f32` (the one-pass kernel recomputes the forward from the static x:
    nothing else links the two graphs), built and warmed before capture.  Models on frames of a few hundred atoms with a small
    head (no one-pass kernel: wave-per-frame preprocessing) are captured as forward-with-kept-features and the two-launch backward
    on them (`_recapture_kept_features`).  `recapture()` after changing parameters."""

    def recapture(self):
        super(GraphedForces, self).recapture()
        x = self.static_x
        plan = self.model.plan_for(x) if hasattr(self.model, "plan_for") else None
        if plan is None or not plan.supports_backward():
            raise NotImplementedError("GraphedForces needs a model served by one fused plan with a backward kernel")
        self._kept = False
        if plan.backward_kind() == 1 and plan.kernel_family == 1:
            return self._recapture_kept_features(plan)
        if plan.backward_kind() != 2:
            # the three-launch backward orders its shared workspace with plan-owned events and a side stream: not capturable
            raise NotImplementedError("GraphedForces needs the one-pass backward kernel (molann_plan_backward_kind == 2); this plan's "
                                      "backward is the three-launch path")
        self._plan = plan
        self.static_dy = torch.zeros_like(self.static_y)
        self.static_dx = torch.empty_like(x)
        self.static_y2 = torch.empty_like(self.static_y)
        with torch.cuda.device(x.device):
            self.model.value_and_vjp(x, self.static_dy, into=(self.static_y2, self.static_dx))    # builds that kernel now
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side), torch.cuda.device(x.device):
            for _ in range(self._warmup):       # compiles the backward kernel outside the capture
                plan.backward(x, self.static_dy, self.static_dx, None)
        torch.cuda.current_stream(x.device).wait_stream(side)
        torch.cuda.synchronize(x.device)
        self.bwd_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.bwd_graph), torch.cuda.device(x.device):
            plan.backward(x, self.static_dy, self.static_dx, None)
        return self

    def _recapture_kept_features(self, plan):
        """Frames served by the wave-per-frame kernels with a small head (a peptide of a few hundred atoms): no one-pass kernel,
        but the three entry points of the kept-features backward - `molann_forward_train_f32`, `molann_mlp_backward_f32` without
        parameter gradients, `molann_features_backward_f32` - only enqueue on the stream they are given (no workspace, no side
        stream, no events), so they capture: the forward graph keeps the features, the backward graph is the other two."""
        x = self.static_x
        n = x.shape[0]
        self._plan, self._kept = plan, True
        self.static_y = torch.empty((n, plan.out_dim), dtype=torch.float32, device=x.device)
        self.static_y2 = self.static_y
        self.static_f = torch.empty((n, plan.feature_dim), dtype=torch.float32, device=x.device)
        self.static_gf = torch.empty_like(self.static_f)
        self.static_dy = torch.zeros_like(self.static_y)
        self.static_dx = torch.empty_like(x)

        def fwd():
            plan.forward_train(x, self.static_y, self.static_f)

        def bwd():
            plan.mlp_backward(self.static_f, self.static_dy, self.static_gf, None)
            plan.features_backward(x, self.static_gf, self.static_dx)

        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side), torch.cuda.device(x.device):
            for _ in range(self._warmup):       # compiles the MLP's backward kernel outside the capture
                fwd()
                bwd()
        torch.cuda.current_stream(x.device).wait_stream(side)
        torch.cuda.synchronize(x.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.cuda.device(x.device):
            fwd()
        self.bwd_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.bwd_graph), torch.cuda.device(x.device):
            bwd()
        return self

    def value_and_vjp(self, x, dy):
        """``(y, dx)`` of one launch of `molann_value_and_vjp_f32` on fresh `x` and `dy` (no graph: a direct launch from an idle
        stream costs less host time than a graph replay plus the two copies into its static buffers) into this object's
        static buffers - clone to keep.  For callers that know the cotangent before the values (a linear bias, or the Jacobian:
        the identity as cotangent on a batch of copies)."""
        if x.shape != self.static_x.shape or dy.shape != self.static_dy.shape:
            raise ValueError("GraphedForces was captured for %s / %s, got %s / %s" % (tuple(self.static_x.shape), tuple(self.static_dy.shape),
                                                                                    tuple(x.shape), tuple(dy.shape)))
        if self._kept:       # no single launch for these plans: the two replays
            self.static_x.copy_(x)
            self.static_dy.copy_(dy)
            self.graph.replay()
            self.bwd_graph.replay()
            return self.static_y, self.static_dx
        return self.model.value_and_vjp(x, dy, into=(self.static_y2, self.static_dx))

    def vjp(self, dy):
        if dy.shape != self.static_dy.shape:
            raise ValueError("GraphedForces was captured for cotangents %s, got %s" % (tuple(self.static_dy.shape), tuple(dy.shape)))
        self.static_dy.copy_(dy)
        self.bwd_graph.replay()
        return self.static_dx
