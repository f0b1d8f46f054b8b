"""Drop-in counterparts of ``flamo.processor.system``: ``Series``, ``Recursion``, ``Parallel`` and
``Shell`` (citations: the reference's ``flamo/processor/system.py``).

The containers keep the reference's behaviour (key rules, nfft / alias_decay_db / dtype
coherence checks, I/O channel checks, ``ext_param`` routing, state_dict naming) and hand the
arithmetic to the HIP path.  ``Recursion`` differs in how it gets to the same numbers: the
closed-loop matrix ``A = I - F(B(I))`` is identical for every batch element (the reference
expands the identity to the batch and factors the same matrices B times, system.py:420-425);
here the identity is pushed through the two paths ONCE with batch 1 and the bin-parallel LU
kernel (``ops.solve``) factors each bin once and back-substitutes all B right-hand sides.
"""
from __future__ import annotations

import warnings
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import ops
from ..functional import signal_gallery
from .dsp import FFT, FFTAntiAlias, Transform, iFFT, iFFTAntiAlias


# Series-level fusion: adjacent per-bin products H_k ... H_2 H_1 X are evaluated as (H_k ... H_1) X --
# the small (N_out x N_in) responses are multiplied per bin ONCE, then the (B, M, N) signal makes a
# single pass through HBM instead of one per module (and one gradient pass instead of two per
# module on the way back).  Results are identical up to floating-point reassociation.
FUSE_SERIES = True
FUSE_MIN_COLUMNS = 4   # batch x trailing columns below which folding does not pay
# Build the folded response on a side stream, concurrently with the input transform of the
# enclosing Shell (see ops.fork_point).  The response depends on parameters only.
OVERLAP_RESPONSES = True
# Shell(FFT -> per-bin chain -> iFFT) as one fused operator (ops.spectral_apply) when the plan and channel counts allow
FUSE_SHELL = True
# Series(Matrix, cascade-type filter, ...): response and gradients of the pair from one fused operator (ops.*_rc)
FUSE_MATRIX_CASCADE = True
# Recursion whose loop is diag(g) D[f] U (per-bin delay matrix, per-channel gains, mixing matrix): the gains scale rows
# inside the solve and the backward pass forms no (M, N, N) gradient (ops.solve_scaled_loop)
SCALED_LOOP = True
# FDN loops whose feedforward path is one diagonal module without gradient (the delays): applied inside the factored solve
# and its one-pass backward (ops.solve_dud2) instead of as launches of its own
FDN_DIAGONAL_IN_SOLVE = True
# Series(Gain(N,1), Recursion, Gain(1,N)) on a one-channel spectrum as one operator (ops.fdn_core): the gains' gradients come
# out of the loop's one-pass backward
FDN_CORE = True
# Gradients of the parameters are then produced on the side stream while their AccumulateGrad
# nodes live on the main one; autograd synchronises the two correctly and merely warns about it.
_quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
if _quiet is not None:
    _quiet(False)


def _as_signal(H: torch.Tensor, diag: bool, M: int) -> torch.Tensor:
    """Response (const or per-bin, diagonal or full) -> matrix-valued signal (1, M, N_out, N_in)."""
    if diag:
        H = torch.diag_embed(H)
    if H.dim() == 2:
        H = H.unsqueeze(0).expand(M, *H.shape)
    return H.unsqueeze(0)


def _diag_mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """product of two diagonal responses, each per-bin (M, N) or constant (N,)"""
    if a.dtype != b.dtype:
        cd = torch.promote_types(a.dtype, b.dtype)
        a, b = a.to(cd), b.to(cd)
    if a.dim() == 2 and b.dim() == 2 and a.is_cuda and a.is_complex():
        return ops.mimo(a, b.unsqueeze(0), diag=True).squeeze(0)      # two per-bin diagonals: one pass on the per-bin kernel
    return a * b


def _compose(acc, nxt, M: int):
    """(H2, diag2) after (H1, diag1): the response of the cascade, H2[f] @ H1[f] per bin.

    The bin-planar memory of a per-bin response (N_out, N_in, M) IS the memory of a signal, so the
    small matrix products run on the per-bin kernels without copies: left factors act on the
    response read as a (1, M, N_out, N_in) signal, right factors on it read as (N_out, M, N_in)
    (output channel as batch)."""
    H1, d1 = acc
    H2, d2 = nxt
    if H1.dtype != H2.dtype:
        cd = torch.promote_types(H1.dtype, H2.dtype)
        H1, H2 = H1.to(cd), H2.to(cd)
    bin1, bin2 = H1.dim() == (2 if d1 else 3), H2.dim() == (2 if d2 else 3)
    if d1 and d2:
        return H2 * H1, True                                  # (M,N)*(N,), (N,)*(N,), ... broadcast
    if not bin1 and not bin2:                                 # both frequency independent: tiny matmul
        A = torch.diag_embed(H2) if d2 else H2
        Bm = torch.diag_embed(H1) if d1 else H1
        return A @ Bm, False
    if not d1 and bin1:                                       # per-bin full on the right: left-multiply it
        return ops.mimo(H2, H1.unsqueeze(0), diag=d2).squeeze(0), False   # squeeze, not [0]: select_backward zero-fills and copies the whole tensor
    if not d2 and bin2:                                       # per-bin full on the left: right-multiply it
        S2 = H2.permute(1, 0, 2)                              # (N_out, M, N_mid): rows of H2 as batch items
        R = ops.mimo(H1, S2, diag=True) if d1 else ops.mimo(H1.transpose(-1, -2), S2)
        return R.permute(1, 0, 2), False
    # a per-bin diagonal meets a constant full matrix: R[f, p, q] = W[p, q] h[f, p]  (rows scaled: diag(h2[f]) W)  or
    # W[p, q] h[f, q]  (columns scaled: W diag(h1[f])).  As ONE launch of the constant-matrix product kernel: the
    # diagonal response read as a (1, M, C) signal, times the (P*Q, C) selection matrix A[(p,q), c] = W[p,q] [c == p or q]
    # (built from P*Q*C parameter-sized scalars) -- no elementwise pass over the (M, P, Q) tensor in torch.
    W, h, rows = (H1, H2, True) if d2 else (H2, H1, False)
    P, Q = W.shape
    C = P if rows else Q
    eye = torch.eye(C, dtype=W.dtype, device=W.device)
    A = (W.unsqueeze(-1) * (eye.unsqueeze(1) if rows else eye.unsqueeze(0))).reshape(P * Q, C)
    R = ops.mimo(A, h.unsqueeze(0))                           # (1, M, P*Q), bin-planar
    return R.squeeze(0).unflatten(-1, (P, Q)), False


def _common_attribute(modules, attr, what="Series"):
    """Value of `attr` shared by all modules that have it (None + warning if nobody has it)."""
    value = None
    for m in modules:
        if hasattr(m, attr):
            value = getattr(m, attr)
            break
    if value is None:
        warnings.warn(f"Attribute {attr} not found in any of the modules.")
        return None
    for i, m in enumerate(modules):
        if hasattr(m, attr) and getattr(m, attr) != value:
            raise ValueError(
                f"All modules must have the same {attr} value. Module {m.__class__.__name__} at index {i} "
                f"is incoherent with the part of the {what} preceding it.")
    return value


# ============================================================================ Series
class Series(nn.Sequential):
    """Cascade of DSP modules (system.py:11-329)."""

    def __init__(self, *args):
        super().__init__(self.__unpack_modules(modules=args, current_keys=[]))
        self.__refresh()

    def __refresh(self):
        mods = list(self)
        self.nfft = _common_attribute(mods, "nfft")
        self.alias_decay_db = _common_attribute(mods, "alias_decay_db")
        self.dtype = _common_attribute(mods, "dtype")
        self.input_channels, self.output_channels = self.__check_io()

    def prepend(self, new_module) -> "Series":
        return self.insert(index=0, new_module=new_module)

    def append(self, new_module) -> "Series":
        for k, v in self.__unpack_modules((new_module,), [*self._modules.keys()]).items():
            self.add_module(k, v)
        self.__refresh()
        return self

    def insert(self, index: int, new_module) -> "Series":
        n = len(self._modules)
        if not (-n <= index <= n):
            raise IndexError("Index out of range.")
        if index < 0:
            index += n
        new_items = list(self.__unpack_modules((new_module,), [*self._modules.keys()]).items())
        items = list(self._modules.items())
        items[index:index] = new_items
        self._modules.clear()
        self._modules.update(items)
        self.__refresh()
        return self

    def __unpack_modules(self, modules: tuple, current_keys: list) -> OrderedDict:
        """Flatten nested Sequential / dict containers into one ordered {key: leaf module} map.
        Key rules (system.py:127-209): a unique custom key is kept; a duplicate custom key is an
        error; a missing key, or one that parses as an integer, becomes the position index."""
        out = OrderedDict()

        def taken():
            return [*current_keys, *out.keys()]

        for module in modules:
            if isinstance(module, nn.Sequential):
                out.update(self.__unpack_modules((module._modules,), taken()))
            elif isinstance(module, (OrderedDict, dict)):
                for k, v in module.items():
                    if isinstance(v, nn.Sequential):
                        out.update(self.__unpack_modules((v._modules,), taken()))
                    elif isinstance(v, (OrderedDict, dict)):
                        out.update(self.__unpack_modules((v,), taken()))
                    else:
                        numeric = True
                        try:
                            int(k)
                        except (TypeError, ValueError):
                            numeric = False
                        if numeric:
                            new_key = str(len(out) + len(current_keys))
                            out[new_key] = v
                            if k != new_key:
                                warnings.warn(f"Key {k} is an integer, it will be overwritten.")
                        else:
                            if k in taken():
                                raise ValueError(f"Key {k} is already present in the Series.")
                            out[k] = v
            elif isinstance(module, nn.Module):
                out[str(len(out) + len(current_keys))] = module
            else:
                raise ValueError("Modules must be nn.Module, nn.Sequential, or OrderedDict.")
        return out

    def __check_io(self):
        mods = [(i, m) for i, m in enumerate(self) if hasattr(m, "input_channels")]
        if not mods:
            return None, None
        first_i, first = mods[0]
        prev_name, prev_pos, prev_out = first.__class__.__name__, first_i, first.output_channels
        for j, m in mods[1:]:
            assert m.input_channels == prev_out, (
                f"Module {prev_name} at index {prev_pos} has {prev_out} output channels, but module "
                f"{m.__class__.__name__} at index {j} has {m.input_channels} input_channels.")
            prev_name, prev_pos, prev_out = m.__class__.__name__, j, getattr(m, "output_channels", None)
        return first.input_channels, prev_out

    def forward(self, input, ext_param=None):
        items = list(self._modules.items())
        i = 0
        while i < len(items):
            key, module = items[i]
            j = i
            if FDN_CORE and FDN_DIAGONAL_IN_SOLVE and ext_param is None and i + 2 < len(items) and isinstance(items[i + 1][1], Recursion):
                out = self.__fdn_core(items[i][1], items[i + 1][1], items[i + 2][1], input)
                if out is not None:
                    input = out
                    i += 3
                    continue
            if FUSE_SERIES and torch.is_tensor(input) and input.is_complex() and input.is_cuda and input.dim() >= 3:
                cols = input.shape[0]
                for d in input.shape[3:]:
                    cols *= d
                if cols >= FUSE_MIN_COLUMNS:
                    while j < len(items) and hasattr(items[j][1], "_fusable") and items[j][1]._fusable():
                        j += 1
            if j - i >= 2:
                input = self.__fused(items[i:j], input, ext_param)
                i = j
                continue
            if ext_param is not None and key in ext_param:
                input = module(input, ext_param[key])
            else:
                input = module(input)
            i += 1
        return input

    @staticmethod
    def __fdn_core(ig, rec, og, x):
        """Gain(N, 1) -> Recursion (FDN loop with a diagonal feedforward path) -> Gain(1, N) on a one-channel spectrum: one
        operator (ops.fdn_core) whose backward returns both gains' gradients from the loop's one-pass backward; None when
        the three modules are not that."""
        if not (torch.is_tensor(x) and x.is_cuda and x.is_complex() and x.dim() == 3 and x.shape[2] == 1):
            return None
        for g in (ig, og):
            if not (hasattr(g, "_mapped") and not g._diag and hasattr(g, "_fusable") and g._fusable()):
                return None
        N = rec.output_channels
        if (ig.input_channels, ig.output_channels, og.input_channels, og.output_channels) != (1, N, N, 1) or rec.input_channels != N:
            return None
        for m in (ig, rec, og):              # the modules' own input checks would have run
            if m.nfft // 2 + 1 != x.shape[1] and ops.bin_shard(m.nfft)[1] != x.shape[1]:
                return None
        b, c = ig._mapped(ig.param), og._mapped(og.param)
        if not (torch.is_tensor(b) and torch.is_tensor(c) and b.dim() == 2 and c.dim() == 2 and b.is_cuda and c.is_cuda):
            return None
        with ops.loop_scope():
            f = rec._fdn_factors(x.shape[1])
        if f is None:
            return None
        return ops.fdn_core(b, c, f[0], f[1], f[2], f[3], x)

    @staticmethod
    def _run_response(run, shape, ext_param, device):
        """(H, diag): response of a run of per-bin modules acting on a signal of the given shape, built on the
        side stream (concurrently with the input transform of the enclosing Shell) when a fork point is set."""
        M = shape[1]

        def ext_of(key):
            return ext_param[key] if (ext_param is not None and key in ext_param) else None

        def build():
            shp = list(shape)
            acc = None
            i = 0
            while i < len(run):
                key, module = run[i]
                if acc is None and FUSE_MATRIX_CASCADE and i + 1 < len(run):
                    # Matrix then cascade-type filter: one operator for (cascade response) @ (matrix), whose backward
                    # computes both factors' gradients inside the cascade's backward kernel
                    nxt_key, nxt = run[i + 1]
                    if hasattr(module, "_real_matrix") and not module._diag and hasattr(nxt, "_response_times_matrix"):
                        Wr = module._real_matrix(module._param_for_fusion(shp, ext_of(key)))
                        if Wr is not None:
                            shp2 = list(shp)
                            shp2[2] = module.output_channels
                            H = nxt._response_times_matrix(nxt._param_for_fusion(shp2, ext_of(nxt_key)), Wr)
                            if H is not None:
                                acc = (H, False)
                                shp[2] = nxt.output_channels
                                i += 2
                                continue
                resp = module._response_for_fusion(shp, ext_of(key))
                shp[2] = module.output_channels
                acc = resp if acc is None else _compose(acc, resp, M)
                i += 1
            return acc

        ev = ops.fork_event() if OVERLAP_RESPONSES else None
        if ev is None:
            return build()
        main = torch.cuda.current_stream(device)
        side = ops.side_stream(device)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            acc = build()
        main.wait_stream(side)
        # the response lives in the side stream's pool: keep it until the enclosing Shell.forward returns, so that a
        # later side-stream region of the same forward (which waits for the fork point only) cannot be handed its
        # memory while a main-stream kernel still reads it (no_grad / inference, where nothing else retains it)
        memo = ops.forward_memo()
        if memo is not None:
            memo.setdefault("_keep_alive", []).append(acc[0])
        return acc

    @staticmethod
    def __fused(run, x, ext_param):
        """Apply a run of per-bin modules as one product with the cascade's response."""
        acc = Series._run_response(run, list(x.shape), ext_param, x.device)
        return ops.mimo(acc[0], x, diag=acc[1])

    def _all_fusable(self) -> bool:
        return len(self) > 0 and all(hasattr(m, "_fusable") and m._fusable() for m in self)

    def probe(self, z: torch.Tensor):
        H = None
        for module in self:
            Hi = module.probe(z)
            H = Hi if H is None else Hi @ H
        return H

    def probe_w(self, w: torch.Tensor):
        H = None
        for module in self:
            Hi = module.probe_w(w)
            H = Hi if H is None else Hi @ H
        return H


# ============================================================================ Recursion
class Recursion(nn.Module):
    """Closed loop  Y = (I - F B)^-1 F X  per frequency bin (system.py:335-565)."""

    def __init__(self, fF, fB):
        nn.Module.__init__(self)
        self.feedforward = self.__as_series(fF, "Feedforward")
        self.feedback = self.__as_series(fB, "Feedback")
        self.nfft = self.__check_attribute("nfft")
        self.alias_decay_db = self.__check_attribute("alias_decay_db")
        self.dtype = self.__check_attribute("dtype")
        self.input_channels, self.output_channels = self.__check_io()
        # The closed-loop solve keeps a loop-matrix row per lane up to 64 (float32) / 32 (float64) channels -- the sizes with
        # the fused loop forms; above that one workgroup per bin factors the materialised matrix in LDS (fl_solve_max_n: 138 /
        # 97), and above THAT in a global-memory workspace (fl_solve_ws_*, to fl_solve_ws_max_n = 1024 channels).  The
        # reference's torch.linalg.solve has no bound; there is deliberately no torch fallback on this path.
        self._register_loop = self.output_channels <= (32 if self.dtype == torch.float64 else 64)
        # the static bound of the workspace solve; what THIS device's library answers is asked at the first forward, on the
        # tensor's device -- not here: constructing a module must not initialise the HIP runtime (fork-based data loaders),
        # and the device current now need not be the module's
        limit = 1024
        self._solve_limit_checked = set()
        assert self.output_channels <= limit, (
            f"Recursion: {self.output_channels} loop channels exceed the HIP solve kernels' limit of {limit}; see INTEGRATION.md")

    @staticmethod
    def __as_series(path, name):
        if isinstance(path, (nn.Sequential, OrderedDict)) and not isinstance(path, Series):
            warnings.warn(f"{name} path has been converted to a Series class instance.")
            return Series(path)
        return path

    def forward(self, X: torch.Tensor, ext_param: dict = None):
        ext_fb = ext_ff = None
        if ext_param is not None:
            for key, param in ext_param.items():
                if "feedback" in key:
                    ext_fb = param
                elif "feedforward" in key:
                    ext_ff = param
        if torch.is_tensor(X) and X.is_cuda:
            self.__check_solve_limit(X.device)
        with ops.loop_scope():
            return self.__forward_in_loop(X, ext_param, ext_fb, ext_ff)

    def __check_solve_limit(self, dev):
        """once per device: the loop's channel count against what the library's solve kernels take THERE"""
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        if key in self._solve_limit_checked:
            return
        from .. import _lib
        with torch.cuda.device(dev):
            limit = max(int(_lib.lib().fl_solve_max_n(int(self.dtype == torch.float64))), int(_lib.lib().fl_solve_ws_max_n()))
        if self.output_channels > limit:
            raise ValueError(f"Recursion: {self.output_channels} loop channels exceed the HIP solve kernels' limit of {limit} on "
                             f"device {key} ({'float64' if self.dtype == torch.float64 else 'float32'}); see INTEGRATION.md")
        self._solve_limit_checked.add(key)

    def __forward_in_loop(self, X, ext_param, ext_fb, ext_ff):
        # (external parameters -- system.py:409-415 -- take the same fused routes: every helper below hands each module its own entry)
        if FUSE_SERIES and FDN_DIAGONAL_IN_SOLVE and self._register_loop and torch.is_tensor(X) and X.is_cuda and X.is_complex():
            d2 = self._fdn_factors(X.shape[1], ext_fb, ext_ff) if (X.dim() >= 3 and X.shape[2] == self.output_channels) else None
            if d2 is not None:
                # FDN structure with a diagonal feedforward path (the delays): that diagonal scales l and the right-hand
                # side where the solve kernels load them (ops.solve_dud2) instead of in launches of its own
                return ops.solve_dud2(d2[0], d2[1], d2[2], d2[3], X)
        R = self.feedforward(X, ext_ff)
        if FUSE_SERIES and torch.is_tensor(R) and R.is_cuda:
            dud = self.__factored_loop(R, ext_fb, ext_ff) if self._register_loop else None
            if dud is not None:
                # FDN structure: P = diag(l) U diag(r) stays factored, A = I - P is built in registers
                return ops.solve_dud(dud[0], dud[1], dud[2], R)
            sl = self.__scaled_loop(R, ext_fb, ext_ff) if self._register_loop else None
            if sl is not None:
                return ops.solve_scaled_loop(sl[0], sl[1], sl[2], R)
            P = self.__composed_loop(R, ext_fb, ext_ff)
            if P is not None:
                return ops.solve(P, R, one_minus=True)
        # generic loop: P = F(B(I)) for ONE batch element (it does not depend on the batch)
        I = self.__identity_like(R)
        P = self.feedforward(self.feedback(I, ext_fb), ext_ff)
        return ops.solve(P, R, one_minus=True)

    @staticmethod
    def __path_items(path, ext):
        """[(module, its external parameter or None)] of a loop path, with the reference's routing (system.py:409-415 hands a
        path its `ext_param` entry: a dict keyed by module name for a Series, the tensor itself for a single module)."""
        if isinstance(path, Series):
            return [(m, (ext[k] if (isinstance(ext, dict) and k in ext) else None)) for k, m in path._modules.items()]
        return [(path, ext)]

    def __loop_chain(self, ext_fb=None, ext_ff=None):
        """The per-bin modules of feedback-then-feedforward in the order they act on the identity, each with its external
        parameter, or None if some module is not a plain per-bin product."""
        chain = []
        for path, ext in ((self.feedback, ext_fb), (self.feedforward, ext_ff)):
            for m, e in self.__path_items(path, ext):
                if isinstance(m, Series):
                    return None
                if not (hasattr(m, "_fusable") and m._fusable()):
                    return None
                chain.append((m, e))
        return chain

    def __composed_loop(self, R, ext_fb=None, ext_ff=None):
        """P[f] = F[f] B[f] from the modules' responses (the Series planner's composition) instead of pushing a
        (1, M, N, N) identity through them: the first module's product with the identity is its own response --
        one 1.6 GB product and its backward less at N = 32, nfft = 384000."""
        chain = self.__loop_chain(ext_fb, ext_ff)
        if chain is None:
            return None
        M = R.shape[1]
        shape = [1, M, self.output_channels, self.output_channels]
        acc = None
        for m, e in chain:
            resp = m._response_for_fusion(shape, e)
            shape[2] = m.output_channels
            acc = resp if acc is None else _compose(acc, resp, M)
        return _as_signal(acc[0], acc[1], M)

    def __scaled_loop(self, R, ext_fb=None, ext_ff=None):
        """If the loop is  diag(g) D[f] U  -- feedback = one constant full matrix U, feedforward = one per-bin full matrix D
        that carries no gradient (a matrix of integer delays) followed by constant per-channel gains g (the structure of
        the active-acoustics chain: Recursion(fF=Series(Delay((N,N)), parallelGain(N)), fB=Matrix)) -- return (g, D, U):
        P' = D U is formed once, the gains scale its rows inside the solve, and the backward pass needs no (M, N, N)
        gradient tensor (ops.solve_scaled_loop).  Else None."""
        if not SCALED_LOOP:
            return None
        chain = self.__loop_chain(ext_fb, ext_ff)
        if chain is None or len(chain) != 3:
            return None
        M = R.shape[1]
        shape = [1, M, self.output_channels, self.output_channels]
        resp = []
        for m, e in chain:
            resp.append(m._response_for_fusion(shape, e))
            shape[2] = m.output_channels
        (U, dU), (D, dD), (g, dg) = resp
        N = self.output_channels
        if dU or dD or not dg:
            return None
        if U.dim() != 2 or tuple(U.shape) != (N, N) or D.dim() != 3 or tuple(D.shape[1:]) != (N, N) or g.dim() != 1:
            return None
        if D.requires_grad:
            return None
        return g, D, U

    def _fdn_factors(self, M: int, ext_fb=None, ext_ff=None):
        """(l, l2, U, r) when the feedforward path is ONE diagonal per-bin module without gradient (l2: parallelDelay in
        every FDN of the reference) and the feedback path is one constant full matrix with per-bin diagonal factors around
        it (l: those applied after it, r: before); else None.  P = diag(l . l2) U diag(r),  R = l2 . X."""
        ffi = self.__path_items(self.feedforward, ext_ff)
        ff = [m for m, _ in ffi]
        if len(ff) != 1 or isinstance(ff[0], Series) or not (hasattr(ff[0], "_fusable") and ff[0]._fusable()):
            return None
        fbi = self.__path_items(self.feedback, ext_fb)
        for m, _ in fbi:
            if isinstance(m, Series) or not (hasattr(m, "_fusable") and m._fusable()):
                return None
        N = self.output_channels
        if ff[0].input_channels != N:
            return None
        shape = [1, M, N, N]
        l2, d2 = ff[0]._response_for_fusion(shape, ffi[0][1])
        if not d2 or l2.dim() != 2 or l2.requires_grad or not l2.is_complex():
            return None
        l = r = U = None
        for m, e in fbi:
            H, diag = m._response_for_fusion(shape, e)
            shape[2] = m.output_channels
            if diag:
                if H.dim() != 2:
                    return None                                    # constant diagonals keep the generic factored route
                if U is None:
                    r = H if r is None else _diag_mul(H, r)
                else:
                    l = H if l is None else _diag_mul(H, l)
            else:
                if U is not None or H.dim() != 2 or tuple(H.shape) != (N, N):
                    return None
                U = H
        return None if U is None else (l, l2, U, r)

    def __factored_loop(self, R, ext_fb=None, ext_ff=None):
        """If feedback-then-feedforward is a chain of per-bin modules with exactly one full,
        frequency-independent matrix U and otherwise diagonal factors (the structure of every FDN
        in flamo: delays and attenuation are diagonal, only the mixing matrix is full), return
        (l, U, r) with P = diag(l) U diag(r); else None."""
        chain = []
        for path, ext in ((self.feedback, ext_fb), (self.feedforward, ext_ff)):        # applied in this order to the identity
            for m, e in self.__path_items(path, ext):
                if not (hasattr(m, "_fusable") and m._fusable()):
                    return None
                chain.append((m, e))
        M = R.shape[1]
        shape = [1, M, self.output_channels, self.output_channels]
        l = r = U = None
        for m, e in chain:
            H, diag = m._response_for_fusion(shape, e)
            shape[2] = m.output_channels
            if diag:
                if U is None:
                    r = H if r is None else _diag_mul(H, r)
                else:
                    l = H if l is None else _diag_mul(H, l)
            else:
                if U is not None or H.dim() != 2 or H.shape[0] != H.shape[1]:
                    return None                                    # second full / per-bin full / non-square
                U = H
        return None if U is None else (l, U, r)

    def __identity_like(self, R: torch.Tensor) -> torch.Tensor:
        """(1, M_local, N, N) identity spectrum, bin-planar, cached per (device, dtype, bins)."""
        N, M = self.output_channels, R.shape[1]
        key = (R.device, R.dtype, M)
        cache = self.__dict__.setdefault("_I_cache", {})
        if key not in cache:
            eye = torch.eye(N, dtype=R.dtype, device=R.device)
            cache[key] = eye.unsqueeze(-1).expand(N, N, M).contiguous().movedim(-1, 0).unsqueeze(0)
        return cache[key]

    @property
    def I(self) -> torch.Tensor:  # noqa: E743
        """(M, N, N) complex identity, the attribute the reference builds eagerly in __init__
        (system.py:427-438).  Built on first access only: the forward pass never needs the
        M*N*N copy (1.5 GB at nfft=384000, N=32)."""
        if "_I_full" not in self.__dict__:
            N, M = self.output_channels, self.nfft // 2 + 1
            cd = torch.complex128 if self.dtype == torch.float64 else torch.complex64
            dev = self.alias_decay_db.device if isinstance(self.alias_decay_db, torch.Tensor) else None
            self.__dict__["_I_full"] = torch.eye(N, dtype=cd, device=dev).unsqueeze(0).repeat(M, 1, 1)
        return self.__dict__["_I_full"]

    def __check_attribute(self, attr: str):
        ff, fb = getattr(self.feedforward, attr, None), getattr(self.feedback, attr, None)
        if ff is None:
            warnings.warn(f"The feedforward pass does not possess the attribute {attr}.")
        if fb is None:
            warnings.warn(f"The feedback pass does not possess the attribute {attr}.")
        if ff is not None and fb is not None:
            assert ff == fb, (f"The feedforward pass has {attr} = {ff} and feedback pass has {attr} = {fb}. "
                              "They must have the same value.")
        return ff if ff is not None else fb

    def __check_io(self) -> tuple:
        chans = {}
        for path, label in ((self.feedforward, "feedforward"), (self.feedback, "feedback")):
            for end in ("input_channels", "output_channels"):
                v = getattr(path, end, None)
                if v is None:
                    raise ValueError(f"The {label} pass does not possess the attribute {end}.")
                chans[(label, end)] = v
        ff_in, ff_out = chans[("feedforward", "input_channels")], chans[("feedforward", "output_channels")]
        fb_in, fb_out = chans[("feedback", "input_channels")], chans[("feedback", "output_channels")]
        assert ff_out == fb_in, (f"Feedforward pass has {ff_out} output channels, but feedback pass has {fb_in} "
                                 "input channels. They must be the same.")
        assert fb_out == ff_in, (f"Feedforward pass {ff_in} input channels, but the feedback pass has {fb_out} "
                                 "output channels. They must be the same.")
        return ff_in, ff_out

    def probe(self, z: torch.Tensor):
        F, B = self.feedforward.probe(z), self.feedback.probe(z)
        A = torch.eye(F.shape[-1], dtype=F.dtype, device=F.device) - F @ B
        return torch.linalg.solve(A, F)

    def probe_recursion(self, z: torch.Tensor, include_shell_io: bool = False, **kwargs):
        F, B = self.feedforward.probe(z), self.feedback.probe(z)
        return torch.eye(F.shape[0], dtype=F.dtype, device=F.device) - F @ B

    def probe_recursion_w(self, w: torch.Tensor):
        F, B = self.feedforward.probe_w(w), self.feedback.probe_w(w)
        return torch.eye(F.shape[0], dtype=F.dtype, device=F.device) - F @ B


# ============================================================================ Parallel
class Parallel(nn.Module):
    """Two branches on the same input, summed or concatenated along channels (system.py:570-772)."""

    def __init__(self, brA, brB, sum_output: bool = True):
        nn.Module.__init__(self)
        self.branchA = Series(brA) if isinstance(brA, (nn.Sequential, OrderedDict)) and not isinstance(brA, Series) else brA
        self.branchB = Series(brB) if isinstance(brB, (nn.Sequential, OrderedDict)) and not isinstance(brB, Series) else brB
        self.sum_output = sum_output
        self.nfft = self.__shared("nfft")
        self.alias_decay_db = self.__shared("alias_decay_db")
        self.dtype = self.__shared("dtype")
        self.input_channels, self.output_channels = self.__check_io()

    def __shared(self, attr):
        a, b = getattr(self.branchA, attr, None), getattr(self.branchB, attr, None)
        if a is None:
            warnings.warn(f"The branch A does not possess the attribute {attr}.")
        if b is None:
            warnings.warn(f"The branch B does not possess the attribute {attr}.")
        if a is not None and b is not None:
            assert a == b, f"The branch A has {attr} = {a} and branch B has {attr} = {b}. They must have the same value."
        return a if a is not None else b

    def __check_io(self):
        a_in, a_out = getattr(self.branchA, "input_channels", None), getattr(self.branchA, "output_channels", None)
        b_in, b_out = getattr(self.branchB, "input_channels", None), getattr(self.branchB, "output_channels", None)
        for v, name in ((a_in, "branch A input"), (a_out, "branch A output"), (b_in, "branch B input"),
                        (b_out, "branch B output")):
            if v is None:
                raise ValueError(f"The {name} channels attribute is missing.")
        assert a_in == b_in, f"Branch A has {a_in} input channels, but branch B has {b_in} input channels."
        if self.sum_output:
            assert a_out == b_out, f"Branch A has {a_out} output channels, but branch B has {b_out}."
            return a_in, a_out
        return a_in, a_out + b_out

    def forward(self, X, ext_param: dict = None):
        # external parameters are routed by key: the entry whose key contains "branchA" / "branchB"
        # goes to that branch (system.py:640-652)
        ext_a = ext_b = None
        if ext_param is not None:
            for key, param in ext_param.items():
                if "branchA" in key:
                    ext_a = param
                elif "branchB" in key:
                    ext_b = param
        ya = self.branchA(X) if ext_a is None else self.branchA(X, ext_a)
        yb = self.branchB(X) if ext_b is None else self.branchB(X, ext_b)
        return ya + yb if self.sum_output else torch.cat((ya, yb), dim=2)

    def probe(self, z: torch.Tensor):
        """Transfer matrix at a complex z: H_A + H_B, or the two stacked along the output channels (system.py:740-758)."""
        HA, HB = self.branchA.probe(z), self.branchB.probe(z)
        return HA + HB if self.sum_output else torch.cat([HA, HB], dim=0)

    def probe_w(self, w: torch.Tensor):
        """The same in the w = 1/z variable (system.py:760-772)."""
        HA, HB = self.branchA.probe_w(w), self.branchB.probe_w(w)
        return HA + HB if self.sum_output else torch.cat([HA, HB], dim=0)


# ============================================================================ Shell
class Shell(nn.Module):
    """input_layer -> core -> output_layer, plus impulse/frequency response helpers
    (system.py:776-1153)."""

    def __init__(self, core, input_layer=nn.Identity(), output_layer=nn.Identity()):
        nn.Module.__init__(self)
        self.__core = self.__wrap(core, "Core")
        self.__input_layer = self.__wrap(input_layer, "Input layer")
        self.__output_layer = self.__wrap(output_layer, "Output layer")
        self.nfft = self.__check_attribute("nfft")
        self.alias_decay_db = self.__check_attribute("alias_decay_db")
        self.dtype = self.__check_attribute("dtype")
        self.input_channels, self.output_channels = self.__check_io()

    @staticmethod
    def __wrap(layer, name):
        if isinstance(layer, (nn.Sequential, OrderedDict)) and not isinstance(layer, Series):
            warnings.warn(f"{name} has been converted to a Series class instance.")
            return Series(layer)
        return layer

    def forward(self, x: torch.Tensor, ext_param: dict = None) -> torch.Tensor:
        with ops.fork_point(x):
            if FUSE_SHELL:
                y = self.__fused_forward(x, ext_param)
                if y is not None:
                    return y
            x = self.__input_layer(x)
            x = self.__core(x, ext_param) if ext_param is not None else self.__core(x)
            return self.__output_layer(x)

    def __fused_forward(self, x, ext_param):
        """FFT -> chain of per-bin products -> iFFT as ONE operator (ops.spectral_apply: three launches, the spectrum
        never makes a round trip through HBM between the transforms and the product, no layout conversion).  Returns
        None when this Shell / input is not of that form; the layered path then runs."""
        fin, fout, core = self.__input_layer, self.__output_layer, self.__core
        if not (type(fin) in (FFT, FFTAntiAlias) and type(fout) in (iFFT, iFFTAntiAlias)):
            return None
        if not (torch.is_tensor(x) and x.is_cuda and x.dtype in (torch.float32, torch.float64) and x.dim() == 3 and x.shape[0] > 0):
            return None
        if fin.transform is not fin._own_transform or fout.transform is not fout._own_transform:
            return None
        if fin.nfft != fout.nfft or fin.nfft != self.nfft:
            return None
        if isinstance(core, Series):
            if not core._all_fusable():
                return None
            run = list(core._modules.items())
        elif hasattr(core, "_fusable") and core._fusable():
            run = [("0", core)]
            if ext_param is not None:
                ext_param = {"0": ext_param}
        else:
            return None
        n_in, n_out = run[0][1].input_channels, run[-1][1].output_channels
        if x.shape[2] != n_in or not ops.spectral_supported(self.nfft, n_in, n_out, x.dtype):
            return None
        if ops.bin_shard(self.nfft) != (0, self.nfft // 2 + 1):
            return None
        nfft, M = self.nfft, self.nfft // 2 + 1
        if isinstance(fin, FFTAntiAlias):
            fin._check(x)
        # (a Matrix-then-cascade response's launch is recorded and rides in the input's column pass: ops.paired_launch)
        with ops.paired_launch(x.dtype == torch.float32) as pair:
            with ops.row_major_bins(nfft):      # the responses come out in the pipeline's bin order (no reordering pass)
                H, diag = Series._run_response(run, [x.shape[0], M, n_in], ext_param, x.device)
            cdt = torch.complex64 if x.dtype == torch.float32 else torch.complex128
            if diag or H.dim() == 2 or H.dtype != cdt:
                pair.flush()                    # torch operations read the response below
            if diag:
                H = torch.diag_embed(H)
            if H.dim() == 2:
                H = H.unsqueeze(0).expand(M, *H.shape)
            if H.dtype != cdt:
                H = H.to(cdt)
            return ops.spectral_apply(x, H, nfft, fin.norm, fout.norm, getattr(fin, "_alias_db", None),
                                      getattr(fout, "_alias_db", None))

    # ---- accessors
    def get_inputLayer(self):
        return self.__input_layer

    def set_inputLayer(self, input_layer: nn.Module = None) -> None:
        self.__input_layer = input_layer

    def get_outputLayer(self):
        return self.__output_layer

    def set_outputLayer(self, output_layer: nn.Module = None) -> None:
        self.__output_layer = output_layer

    def get_core(self):
        return self.__core

    def set_core(self, core: nn.Module) -> None:
        self.__core = core

    # ---- checks
    def __check_attribute(self, attr: str):
        core_v = getattr(self.__core, attr, None)
        if core_v is None:
            raise ValueError(f"The core does not possess the attribute {attr}.")
        in_v = getattr(self.__input_layer, attr, None)
        if in_v is not None:
            assert core_v == in_v, (f"The input layer has {attr} = {in_v} and the core has {attr} = {core_v}. "
                                    "They must have the same value.")
        out_v = getattr(self.__output_layer, attr, None)
        if out_v is not None:
            assert core_v == out_v, (f"The core has {attr} = {core_v} and the output layer has {attr} = {out_v}. "
                                     "They must have the same value.")
        return core_v

    def __check_io(self) -> tuple:
        core_in = getattr(self.__core, "input_channels", None)
        core_out = getattr(self.__core, "output_channels", None)
        if core_in is None:
            raise ValueError("The core does not possess the attribute input_channels.")
        il_out = getattr(self.__input_layer, "output_channels", None)
        if il_out is not None:
            assert core_in == il_out, (f"The core should receive {core_in} input channels, but {il_out} channels "
                                       "arrive from the input layer.")
        if core_out is None:
            raise ValueError("The core does not possess the attribute output_channels.")
        ol_in = getattr(self.__output_layer, "input_channels", None)
        if ol_in is not None:
            assert core_out == ol_in, (f"The core sends {core_out} output channels, but the output layer can only "
                                       f"receive {ol_in} channels.")
        il_in = getattr(self.__input_layer, "input_channels", None)
        ol_out = getattr(self.__output_layer, "output_channels", None)
        return (il_in if il_in is not None else core_in), (ol_out if ol_out is not None else core_out)

    def probe(self, z: torch.Tensor, include_shell_io: bool = False):
        H = self.__core.probe(z)
        if include_shell_io:
            for layer, left in ((self.__input_layer, False), (self.__output_layer, True)):
                Hl = layer.probe(z) if hasattr(layer, "probe") else None
                if Hl is None:
                    continue
                H = Hl if H is None else (Hl @ H if left else H @ Hl)
        return H

    # ---- responses
    def __device(self):
        for p in self.parameters():
            return p.device
        return self.alias_decay_db.device

    def __probe_signal(self, fs, identity):
        x = signal_gallery(batch_size=1, n_samples=self.nfft, n=self.input_channels, signal_type="impulse", fs=fs,
                           device=self.__device(), dtype=self.dtype)
        return x.diag_embed() if (identity and self.input_channels > 1) else x

    def __probe_spectrum(self, identity):
        """Spectrum of the probe: rfft of the unit impulse is exactly 1 at every bin, so the probe is
        written in the frequency domain directly -- ones (1, M, N_in), or with ``identity`` the identity
        matrix per bin (1, M, N_in, N_in) -- instead of transforming nfft x N_in (x N_in) zeros and ones."""
        N, M = self.input_channels, self.nfft // 2 + 1
        dev = self.__device()
        cd = torch.complex128 if self.dtype == torch.float64 else torch.complex64
        if identity and N > 1:
            X = ops._empty_planar((1, M, N, N), cd, dev)
            X.copy_(torch.eye(N, dtype=cd, device=dev).view(1, 1, N, N).expand(1, M, N, N))
            return X
        X = ops._empty_planar((1, M, N), cd, dev)
        X.fill_(1.0)
        return X

    def __with_layers(self, input_layer, output_layer, x):
        saved = (self.get_inputLayer(), self.get_outputLayer())
        self.set_inputLayer(input_layer)
        self.set_outputLayer(output_layer)
        try:
            with torch.no_grad():
                return self.forward(x)
        finally:
            self.set_inputLayer(saved[0])
            self.set_outputLayer(saved[1])

    def __anti_alias_db(self) -> float:
        return abs(float(self.alias_decay_db))

    def get_time_response(self, fs: int = 48000, identity: bool = False) -> torch.Tensor:
        """Impulse response irfft(core(rfft(delta))) * gamma^-t (system.py:1012-1079); with
        ``identity`` the input is diag_embed'ed so the full N_out x N_in response matrix comes out.
        The probe's spectrum is written directly (no input transform) and the envelope is fused into
        the inverse-FFT epilogue."""
        db = self.__anti_alias_db()
        out = Transform(lambda X: ops.irfft(X, self.nfft, "backward", db if db else None))
        if not self.__device().type == "cuda":
            return self.__with_layers(FFT(self.nfft, dtype=self.dtype), out, self.__probe_signal(fs, identity))
        return self.__with_layers(nn.Identity(), out, self.__probe_spectrum(identity))

    def get_freq_response(self, fs: int = 48000, identity: bool = False) -> torch.Tensor:
        """rfft(irfft(core(rfft(delta))) * gamma^-t) (system.py:1081-1153).  The time-domain round trip
        is what moves the response from the circle |z| = 1/gamma back to the unit circle; without
        anti-aliasing (gamma = 1) it is the identity up to the C2R convention (imaginary parts of the
        DC and Nyquist bins dropped), so neither transform is run; with it, both transforms and the
        envelope run as two fused HIP FFT calls.  The probe's spectrum is written directly."""
        db = self.__anti_alias_db()
        if db:
            out = Transform(lambda X: ops.rfft(ops.irfft(X, self.nfft, "backward", db), self.nfft))
        else:
            def out_fn(X):
                Y = X.clone(memory_format=torch.preserve_format)
                for k in (0, Y.shape[1] - 1):
                    torch.view_as_real(Y[:, k])[..., 1].zero_()
                return Y
            out = Transform(out_fn)
        if not self.__device().type == "cuda":
            return self.__with_layers(FFT(self.nfft, dtype=self.dtype), out, self.__probe_signal(fs, identity))
        return self.__with_layers(nn.Identity(), out, self.__probe_spectrum(identity))
