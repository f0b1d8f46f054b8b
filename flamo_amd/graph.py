"""hipGraph capture of a whole training / evaluation step.

The hot path's big kernels are few, but a step also runs the parameter maps in PyTorch
(``matrix_exp(skew(.))``, GEQ gain maps, the loss): a 16-channel FDN step is ~230 launches of which
~215 are tiny, so eager execution is launch-bound on the host (3.3 ms wall for 1.3 ms of GPU work).
``GraphedStep`` captures forward + backward (+ optimizer, if inside ``fn``) once into a HIP graph
through ``torch.cuda.CUDAGraph`` and replays it: the C-ABI kernels are launched on the capturing
stream like every other kernel, the library never allocates or synchronises, and all temporaries
come from PyTorch's graph-private pool, so capture needs nothing special from the kernels.

    step = GraphedStep(lambda x: loss_fn(model(x)), example_inputs=(x,), params=model.parameters())
    loss = step(x_new)            # copies x_new into the static input, replays, returns static loss
    step.grads                    # the parameters' .grad tensors are updated in place by the replay
"""
from __future__ import annotations

from typing import Callable, Iterable, Sequence

import torch


class GraphedStep:
    def __init__(self, fn: Callable[..., torch.Tensor], example_inputs: Sequence[torch.Tensor],
                 params: Iterable[torch.nn.Parameter] = (), warmup: int = 3,
                 allow_graph_packets: bool = False):
        import flamo_amd
        if not flamo_amd._graph_packets_off():
            # refused, not warned about: with ROCm's pre-built graph packets a captured torch reduction behind this library's
            # kernels has returned wrong -- deterministic, plausible -- values after eager launches between replays (DESIGN.md
            # section 4.5).  A training loop would go on silently with a wrong loss.
            msg = ("flamo_amd.graph.GraphedStep: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 is not in effect (flamo_amd was imported after "
                   "the HIP runtime had been initialised, or the variable is set to something else).  Export "
                   "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before starting the process, or import flamo_amd before the first "
                   "torch.cuda call.")
            if not allow_graph_packets:
                raise RuntimeError(msg + "  (GraphedStep(..., allow_graph_packets=True) captures anyway.)")
            import warnings
            warnings.warn(msg, RuntimeWarning, stacklevel=2)
        self.params = [p for p in params if p.requires_grad]
        self.static_inputs = [t.clone() for t in example_inputs]
        self._fn = fn
        self._seed = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):            # warm-up off the default stream: fills twiddle/constant caches
            for _ in range(warmup):
                self._run_eager()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: a communicator's watchdog thread (RCCL) may query events while this
        # thread captures; in the default global mode that would invalidate the capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.static_out = self._run_captured()
        # the gradients the captured backward produces live in the graph's pool and are rewritten
        # in place by every replay: hand them to the parameters as they are (no copy per step)
        for p, g in zip(self.params, self._static_grads):
            if g is not None:
                p.grad = g

    def _run_eager(self):
        from . import ops
        for p in self.params:
            p.grad = None
        with ops.step_scope():
            out = self._fn(*self.static_inputs)
            out.backward()
        # the all-ones gradient seed of the captured backward: allocated once here instead of being
        # filled by a launch in every replay
        self._seed = torch.ones_like(out)
        return out.detach()

    def _run_captured(self):
        from . import ops
        with ops.step_scope():
            out = self._fn(*self.static_inputs)
            seed = self._seed if (self._seed is not None and self._seed.shape == out.shape and self._seed.dtype == out.dtype) else None
            grads = torch.autograd.grad(out, self.params, grad_outputs=seed, allow_unused=True) if self.params else ()
        self._static_grads = [None if g is None else (g if g.is_contiguous() else g.contiguous()) for g in grads]
        return out.detach()

    @property
    def grads(self):
        return [p.grad for p in self.params]

    def replay(self) -> torch.Tensor:
        """Replay on the current contents of the static inputs (no host-side copies)."""
        self.graph.replay()
        return self.static_out

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        for dst, src in zip(self.static_inputs, inputs):
            if src is not dst:
                dst.copy_(src)
        self.graph.replay()
        return self.static_out
