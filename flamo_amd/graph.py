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

Data-parallel training: ``GraphedStep(..., grad_buckets=True)`` also packs the gradients, inside the captured graph, into
one of two flat buffers that alternate from replay to replay (``csrc/reduce.hip: fl_pack_toggle``); ``dist.BucketReducer``
sums the bucket of the last replay over the ranks while the next replay already fills the other one.
"""
from __future__ import annotations

from typing import Callable, Iterable, Sequence

import torch


class GraphedStep:
    def __init__(self, fn: Callable[..., torch.Tensor], example_inputs: Sequence[torch.Tensor],
                 params: Iterable[torch.nn.Parameter] = (), warmup: int = 3,
                 allow_graph_packets: bool = False, grad_buckets: bool = False):
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
        self._want_buckets = bool(grad_buckets)
        self.buckets = None           # [flat0, flat1] once captured with grad_buckets
        self.bucket_views = None      # per bucket: one view per parameter, shaped like its gradient (None: no gradient)
        self.replays = 0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):            # warm-up off the default stream: fills twiddle/constant caches
            for _ in range(warmup):
                self._run_eager()
        torch.cuda.current_stream().wait_stream(side)
        if self._want_buckets:
            self._plan_buckets()
        self.graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: a communicator's watchdog thread (RCCL) may query events while this
        # thread captures; in the default global mode that would invalidate the capture
        # constants of parameter values taken from the warm-up runs' caches instead of being recorded (ops.capture_scope): kept
        # alive here, their parameters' version counters checked before every replay
        self._constants = []
        from . import ops
        with ops.capture_scope(self._constants):
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.static_out = self._run_captured()
        # the gradients the captured backward produces live in the graph's pool and are rewritten
        # in place by every replay: hand them to the parameters as they are (no copy per step)
        for p, g in zip(self.params, self._static_grads):
            if g is not None:
                p.grad = g
        if self._want_buckets:
            self._fill_bucket_table()

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
        if self._want_buckets:
            self._pack_buckets()
        return out.detach()

    def _plan_buckets(self):
        """(before the capture) everything the packing launch points at: the two buckets sized for every parameter, the entry
        table and the device-side counter.  Nothing here may be created inside the capture -- a fill or a host copy would
        become a node of the graph and run again with every replay (the counter would restart from zero each time)."""
        if not self.params:
            raise ValueError("GraphedStep(grad_buckets=True): no parameters")
        dt, dev = self.params[0].dtype, self.params[0].device
        if any(p.dtype != dt for p in self.params) or dt not in (torch.float32, torch.float64):
            raise ValueError("GraphedStep(grad_buckets=True): the parameters must share one real dtype (float32 or float64)")
        total = sum(p.numel() for p in self.params)
        self.buckets = [torch.zeros(total, dtype=dt, device=dev) for _ in range(2)]
        self._bucket_state = torch.zeros(2, dtype=torch.int32, device=dev)
        self._bucket_table = torch.zeros((len(self.params), 3), dtype=torch.int64, device=dev)

    def _pack_buckets(self):
        """(inside the capture) the gradients -> the bucket of this replay: one launch, the bucket chosen on the device.  The
        table's rows are written after the capture (_fill_bucket_table): the launch only records its address."""
        from . import _lib, ops
        n = sum(g is not None for g in self._static_grads)
        if n == 0:
            raise ValueError("GraphedStep(grad_buckets=True): the step produces no parameter gradients")
        _lib.check(_lib.lib().fl_pack_toggle(self._bucket_table.data_ptr(), n, self.buckets[0].data_ptr(),
                                             self.buckets[1].data_ptr(), self._bucket_state.data_ptr(), ops._stream()), "pack_toggle")

    def _fill_bucket_table(self):
        """(after the capture) where the captured backward leaves each gradient, and where it goes in a bucket"""
        esz = self.buckets[0].element_size()
        rows, off = [], 0
        views = [[], []]
        for g in self._static_grads:
            if g is None:
                views[0].append(None)
                views[1].append(None)
                continue
            rows.append([g.data_ptr(), off * esz, g.numel() * esz])
            for b in range(2):
                views[b].append(self.buckets[b][off:off + g.numel()].view(g.shape))
            off += g.numel()
        self.bucket_views = views
        self._bucket_table[:len(rows)].copy_(torch.tensor(rows, dtype=torch.int64))
        torch.cuda.current_stream().synchronize()

    @property
    def bucket(self) -> int:
        """index of the bucket the LAST replay filled (replay i, counted from 0, fills bucket i & 1)"""
        if self.buckets is None or self.replays == 0:
            raise RuntimeError("GraphedStep.bucket: no replay has filled a gradient bucket (grad_buckets=True, then replay)")
        return (self.replays - 1) & 1

    @property
    def grads(self):
        return [p.grad for p in self.params]

    def _check_constants(self):
        for ref, version, _ in self._constants:
            p = ref()
            if p is None or p._version != version:
                raise RuntimeError("GraphedStep: a parameter whose response the captured graph holds as a constant (an integer delay) "
                                   "has been assigned a new value since the capture -- capture the step again")

    def replay(self) -> torch.Tensor:
        """Replay on the current contents of the static inputs (no host-side copies)."""
        self._check_constants()
        self.graph.replay()
        self.replays += 1
        return self.static_out

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        for dst, src in zip(self.static_inputs, inputs):
            if src is not dst:
                dst.copy_(src)
        self._check_constants()
        self.graph.replay()
        self.replays += 1
        return self.static_out
