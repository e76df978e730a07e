"""The trainer's optimizer: ``torch.optim.Adam`` with its step as ONE launch of ``a3vt_adam_step`` over all parameter tensors.

Reference: ``optim.Adam(params, lr=self.args.lr, weight_decay=0)`` and ``self.optimizer.step()``,
pterotactyl/reconstruction/vision/train.py:64,148.  The class IS a ``torch.optim.Adam``: same constructor arguments, same
per-parameter state (``step``, ``exp_avg``, ``exp_avg_sq``) and therefore the same ``state_dict`` — checkpoints written by the
reference (``train.py:213``) or by torch's own Adam load here and the other way round.  Only ``step()`` differs: torch's fused
multi-tensor kernel takes 7 launches at 1.8 TB/s for the image model's 47 M parameters; the library walks a chunk table of all
tensors in one launch (csrc/adam.hip), with torch's single-tensor order of operations per element.

Whatever the kernel does not take (CPU or non-fp32 parameters, non-contiguous tensors, sparse gradients, amsgrad, maximize,
a closure that changes the parameter set) goes through ``torch.optim.Adam.step`` unchanged."""
import torch

from . import lib as _lib
from .ops import _stream


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, maximize=False):
        # the plain (single-tensor) flavour: `step` counters live on the host, as in the reference's optimizer
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize,
                         foreach=False, fused=False)
        self._tables = {}        # group index -> (key, device tables)
        self.library_steps = 0   # steps taken by a3vt_adam_step (tests, bench)

    # -- the tensors of one group, state created as torch's _init_group does ------------------------------------------------------
    def _collect(self, group):
        ps, gs, ms, vs, steps = [], [], [], [], []
        for p in group["params"]:
            if p.grad is None:
                continue
            g = p.grad
            if (g.is_sparse or not p.is_cuda or p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_contiguous()
                    or not g.is_contiguous() or g.device != p.device):
                return None
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            m, v = st["exp_avg"], st["exp_avg_sq"]
            if (m.dtype != torch.float32 or v.dtype != torch.float32 or not m.is_contiguous() or not v.is_contiguous()
                    or m.device != p.device or v.device != p.device):
                return None
            if st["step"].is_cuda:     # a checkpoint written by torch's fused flavour: the counter comes back to the host, once
                st["step"] = st["step"].detach().to("cpu", torch.float32)
            ps.append(p)
            gs.append(g)
            ms.append(m)
            vs.append(v)
            steps.append(st["step"])
        return ps, gs, ms, vs, steps

    def _table(self, gi, ps, gs, ms, vs):
        key = tuple((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()) for p, g, m, v in zip(ps, gs, ms, vs))
        hit = self._tables.get(gi)
        if hit is not None and hit[0] == key:
            return hit[1]
        L = _lib.load()
        chunk = int(L.a3vt_adam_chunk_elems())
        dev = ps[0].device
        ptrs = torch.tensor([[k[j] for k in key] for j in range(4)], dtype=torch.int64)
        numel = torch.tensor([k[4] for k in key], dtype=torch.int64)
        ct, co = [], []
        for t, k in enumerate(key):
            for off in range(0, k[4], chunk):
                ct.append(t)
                co.append(off)
        tabs = (ptrs.to(dev), numel.to(dev), torch.tensor(ct, dtype=torch.int32).to(dev), torch.tensor(co, dtype=torch.int64).to(dev), len(ct))
        self._tables[gi] = (key, tabs)
        return tabs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        plans = []
        for gi, group in enumerate(self.param_groups):
            got = None if (group["amsgrad"] or group["maximize"] or group.get("capturable") or group.get("differentiable")
                           or isinstance(group["lr"], torch.Tensor)) else self._collect(group)
            if got is None:
                plans = None
                break
            if got[0] and len({p.device for p in got[0]}) > 1:
                plans = None
                break
            plans.append(got)
        if plans is None:
            # everything the kernel does not take: torch's own step on the same state (the function under the step-hook wrapper: this
            # call is already inside that wrapper, and the hooks must fire once)
            getattr(torch.optim.Adam.step, "__wrapped__", torch.optim.Adam.step)(self)
            return loss
        L = _lib.load()
        for gi, (group, (ps, gs, ms, vs, steps)) in enumerate(zip(self.param_groups, plans)):
            if not ps:
                continue
            torch._foreach_add_(steps, 1.0)
            counts = {int(s.item()) for s in steps}
            if len(counts) != 1:                       # parameters that joined later: one launch per step count
                by = {}
                for i, s in enumerate(steps):
                    by.setdefault(int(s.item()), []).append(i)
                parts = [(c, [[x[i] for i in idx] for x in (ps, gs, ms, vs)]) for c, idx in sorted(by.items())]
                self._tables.pop(gi, None)
            else:
                parts = [(counts.pop(), [ps, gs, ms, vs])]
            beta1, beta2 = group["betas"]
            for count, (p_, g_, m_, v_) in parts:
                ptrs, numel, ct, co, n_chunks = self._table(gi if len(parts) == 1 else (gi, count), p_, g_, m_, v_)
                with torch.cuda.device(p_[0].device):
                    _lib.check(L.a3vt_adam_step(_lib.ptr(ptrs[0]), _lib.ptr(ptrs[1]), _lib.ptr(ptrs[2]), _lib.ptr(ptrs[3]), _lib.ptr(numel),
                                                _lib.ptr(ct), _lib.ptr(co), n_chunks, float(group["lr"]), float(beta1), float(beta2),
                                                float(group["eps"]), float(group["weight_decay"]), count, _stream()), "adam_step")
            self.library_steps += 1
        return loss


def make_adam(params, lr, library=True):
    """The trainer's optimizer (vision/train.py:64): the library's one-launch Adam, or — ``library=False``, for A/B — torch's fused
    (foreach where the build has no fused kernel) flavour.  Same state layout either way."""
    params = list(params)
    if library:
        return Adam(params, lr=lr, weight_decay=0)
    try:
        return torch.optim.Adam(params, lr=lr, weight_decay=0, fused=True)
    except (RuntimeError, TypeError):
        return torch.optim.Adam(params, lr=lr, weight_decay=0, foreach=True)
