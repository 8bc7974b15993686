"""``torch.optim.Adam`` semantics on one multi-tensor HIP launch (include/gda_hip.h:
``gda_adam_multi_f32``).  Every trainer of the reference builds ``torch.optim.Adam(params, lr,
weight_decay)`` (pygda/models/a2gnn.py:290-294); torch's fused implementation chunks tensors by
65536 elements, which leaves a model with one 867k-element weight and a few small ones on ~20
workgroups (42 us per step at cfg-A).  Same update rule, 2048-element work items, device-resident
step counters (hipGraph-capturable by construction)."""
import ctypes

import torch

from . import _lib

MAX_TENSORS = 48


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameter")
        # `capturable` is what the hipGraph step looks for (pygda_amd/models/base.py)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=True))
        # id(parameter) -> a second leaf over the SAME storage whose .grad holds a second contribution to the parameter's
        # gradient (pygda_amd/nn/a2gnn_base.py::A2GNNBase.second_leaves): summed inside the update kernel
        self.grad_aliases = {}

    def _second(self, p):
        a = self.grad_aliases.get(id(p))
        return None if a is None else a.grad

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none)
        for a in self.grad_aliases.values():
            if a.grad is not None:
                if set_to_none:
                    a.grad = None
                else:
                    a.grad.zero_()

    def _state(self, p):
        st = self.state[p]
        if not st:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def bump_steps(self, counter=None):
        """START of a training step (captured steps: pygda_amd/hipgraph.py): increment every parameter's Adam step
        counter -- and ``counter``, the device step counter of the fused dropout kernels -- in ONE launch; the
        following :meth:`step` then launches the update alone.  The bump used to sit between the last gradient kernel
        and the update (8 us + a launch gap at the end of every replayed step).  Returns False, having done nothing,
        when the optimiser's shape does not allow it (a Parameter listed twice -- UDAGCN -- is updated in two rounds
        with two increments; more tensors than one launch takes).

        Only parameters that already HAVE optimiser state are bumped: like ``torch.optim.Adam`` (and this class's own
        eager path) state is created lazily, on the first step in which a parameter has a gradient -- ``state_dict()``
        of a captured run therefore holds the same entries as an eager run's, and a parameter that never receives a
        gradient costs a replayed step nothing.  A parameter that meets its first gradient in a
        bumped step gets that one increment inside :meth:`step`."""
        listed = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        if len({id(p) for p in listed}) != len(listed) or not 0 < len(listed) <= MAX_TENSORS:
            return False
        params = [p for p in listed if self.state.get(p)]
        ptrs = (ctypes.c_void_p * max(len(params), 1))(*[self.state[p]["step"].data_ptr() for p in params])
        _lib.check(_lib.lib().gda_step_bump(None if counter is None else counter.data_ptr(), ptrs, len(params),
                                            _lib.stream()), "gda_step_bump")
        self._bumped = {id(p) for p in params}
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        bumped, self._bumped = getattr(self, "_bumped", None), None
        if bumped is not None:
            # counters incremented by bump_steps(): torch steps only the parameters that HAVE a gradient, so one that
            # was bumped and has none gets its increment taken back (a plain device op: captured with the step)
            for group in self.param_groups:
                for p in group["params"]:
                    has_grad = p.grad is not None or self._second(p) is not None
                    if id(p) in bumped and not has_grad:
                        self.state[p]["step"].sub_(1.0)
                    elif id(p) not in bumped and has_grad:
                        self._state(p)["step"].add_(1.0)       # first gradient ever: state is created here, as torch does
        for group in self.param_groups:
            # A Parameter listed twice (UDAGCN / SpecReg hand the optimiser the conv weights their two
            # encoders share twice, pygda/models/udagcn.py:262-268) is updated twice per step, one update after
            # the other, by torch's per-tensor CPU loop.  The k-th occurrence of every tensor goes into the k-th
            # round of launches: rounds run in stream order, so the second update sees the first one's result.
            rounds, seen = [], {}
            for p in group["params"]:
                if p.grad is None and self._second(p) is None:
                    continue
                k = seen.get(id(p), 0)
                seen[id(p)] = k + 1
                while len(rounds) <= k:
                    rounds.append([])
                rounds[k].append(p)
            chunks = [r[i:i + MAX_TENSORS] for r in rounds for i in range(0, len(r), MAX_TENSORS)]
            for chunk in chunks:
                table = (_lib.AdamTensorStruct * len(chunk))()
                second = (ctypes.c_void_p * len(chunk))()
                keep = []
                for k, p in enumerate(chunk):
                    _lib.require_gpu_tensor(p, "parameter", torch.float32)
                    g, g2 = p.grad, self._second(p)
                    if g is None:
                        g, g2 = g2, None
                    if g2 is not None and seen.get(id(p), 0) > 1:
                        raise _lib.GdaError("Adam: a parameter listed twice cannot have a second gradient leaf")
                    if g.is_sparse or g.dtype != torch.float32:
                        raise _lib.GdaError("Adam: dense fp32 gradients only")
                    # the update is elementwise: any dense layout works as long as parameter, gradient and both
                    # moments share it (a weight stored gather-major, pygda_amd/sparse_features.py: [out, in] with
                    # transposed strides)
                    if not (p.is_contiguous() or (p.dim() == 2 and p.t().is_contiguous())):
                        raise _lib.GdaError("Adam: parameters must be dense (row- or column-major)")
                    if g.stride() != p.stride():
                        g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                    if g2 is not None:
                        if g2.is_sparse or g2.dtype != torch.float32:
                            raise _lib.GdaError("Adam: dense fp32 gradients only")
                        if g2.stride() != p.stride():
                            g2 = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g2)
                        second[k] = g2.data_ptr()
                    keep.append((g, g2))
                    st = self._state(p)
                    for name in ("exp_avg", "exp_avg_sq"):
                        if st[name].stride() != p.stride():          # state created before the weight was re-laid out
                            st[name] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(st[name])
                    table[k] = _lib.AdamTensorStruct(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                                     st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(), p.numel())
                b1, b2 = group["betas"]
                _lib.check(L.gda_adam_multi_sum_f32(table, second, len(chunk), float(group["lr"]), float(b1), float(b2),
                                                    float(group["eps"]), float(group["weight_decay"]),
                                                    1 if bumped is not None else 0, _lib.stream()),
                           "gda_adam_multi_sum_f32")
        return loss
