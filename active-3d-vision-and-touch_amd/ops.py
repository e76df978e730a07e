"""torch.autograd wrappers over the C ABI (``include/a3vt.h``).  PyTorch is plumbing here: it owns device
memory, the current HIP stream and autograd bookkeeping; every op below is one C call into ``liba3vt.so``.

All tensors must be float32 / int32, contiguous, on a ROCm device.  There is no CPU path.
"""
import ctypes
import weakref

import numpy as np
import torch

from . import lib as _lib
from .mesh import CSRAdjacency


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"a3vt: `{name}` must be a tensor on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"a3vt: `{name}` must be {dtype}, got {t.dtype}")
    return t.contiguous()


_WORKSPACES = {}
# call counters (tests assert that forward-only callers never allocate the activation stash)
STATS = {"stack_calls": 0, "stack_stash_calls": 0}


def workspace(tag, nbytes, device):
    """Per-device scratch buffer that only grows; all users run on the current stream, in order."""
    key = (tag, device.index if device.index is not None else torch.cuda.current_device())
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((nbytes + 3) // 4 + 64, dtype=torch.float32, device=device)
        _WORKSPACES[key] = buf
    return buf


class DeviceCSR:
    """Row-normalised adjacency (and its transpose) as int32/float32 device tensors."""

    def __init__(self, host: CSRAdjacency, device):
        self.n, self.nnz = host.n, host.nnz
        L = _lib.load()
        _lib.check(L.a3vt_csr_validate(host.rowptr.ctypes.data, host.col.ctypes.data, host.n, host.nnz), "csr_validate")
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
        self.rowptr, self.col, self.val = to(host.rowptr), to(host.col), to(host.val)
        self.t_rowptr, self.t_col, self.t_val = to(host.t_rowptr), to(host.t_col), to(host.t_val)
        self.host = host
        self.max_degree = int(np.diff(host.rowptr).max()) if host.n else 0        # hub rows get their own launch
        self.t_max_degree = int(np.diff(host.t_rowptr).max()) if host.n else 0
        # the fused vision + touch matrix as P + a complete bipartite block (a3vt_adj_split): found on the host, PROVEN
        # against the CSR by the library, handed to the stack calls next to the CSR.  Plain meshes keep split = None
        # (their rows are short already: the generic kernels serve them bit for bit as before).
        self.split = self.split_struct = None
        sp = host.split() if self.max_degree > 8 else None
        if sp is not None:
            _lib.check(L.a3vt_adj_split_validate(host.rowptr.ctypes.data, host.col.ctypes.data, host.val.ctypes.data, host.n,
                                                 sp.rowptr.ctypes.data, sp.col.ctypes.data, sp.scale.ctypes.data,
                                                 sp.cls.ctypes.data), "adj_split_validate")
            self.split = (to(sp.rowptr), to(sp.col), to(sp.scale), to(sp.cls))
            self.split_struct = _lib.AdjSplit(*[t.data_ptr() for t in self.split], sp.max_degree, sp.n_seam, sp.n_centre)
        self.use_split = True     # test hook: False = the plain CSR everywhere (A/B of the two aggregations)

    def split_ref(self):
        """``const a3vt_adj_split *`` for a stack call (NULL without a split)."""
        return ctypes.byref(self.split_struct) if (self.split_struct is not None and self.use_split) else None

    @property
    def device(self):
        return self.val.device


def _wants_grad(*tensors):
    """Whether a backward pass can follow this call.  Decided OUTSIDE ``Function.forward``: inside it
    ``ctx.needs_input_grad`` is still True for parameters under ``torch.no_grad()`` (it mirrors ``requires_grad``, not the
    grad mode), which would make the forward-only callers (``Engine.validate``, ``policies/scoring.py``) allocate and write
    the whole activation stash."""
    return torch.is_grad_enabled() and any(t.requires_grad for t in tensors)


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


GEMM_MODES = {"fp32": 0, "bf16": 1, "bf16s": 2, "fp32x3": 3}


PATH_NAMES = ("stack_quad", "stack_rows", "rowgemm_adirect", "rowgemm3", "dw3", "dw_hybrid", "rowgemm16", "stack16_quad", "rowgemm_w", "dw_w", "stack_split", "csr16_tiles")


def path_counts(reset=False):
    """Launch decisions per kernel family since the last reset (``a3vt_dbg_path_counts``): a test hook — parity tests assert
    that their fixture reaches the kernels it claims to pin."""
    import ctypes
    buf = (ctypes.c_longlong * len(PATH_NAMES))()
    _lib.load().a3vt_dbg_path_counts(buf, len(PATH_NAMES), 1 if reset else 0)
    return dict(zip(PATH_NAMES, (int(x) for x in buf)))


def nn_work(enable):
    """Work counters of the pruned nearest-neighbour search (``a3vt_dbg_nn_work``; a test hook, synchronises the device):
    returns what was counted since they were switched on, then switches them on (True: cleared first) or off (False)."""
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    _lib.check(_lib.load().a3vt_dbg_nn_work(1 if enable else 0, buf), "dbg_nn_work")
    names = ("waves", "groups", "blocks", "box_tests", "max_groups_per_wave")
    return dict(zip(names, (int(x) for x in buf)))


def gemm_mode(bf16):
    """0 exact fp32 MFMA; 1 bf16 operands, fp32 storage; 2 bf16 storage (activations / gradients / weight images bf16 in
    HBM, fp32 accumulation); 3 "fp32x3": fp32 storage, the hidden-layer products as six bf16 MFMA passes on operands split
    exactly into three bf16 pieces (fp32-level error, not bit-identical to mode 0; csrc/gcn_gemm3.hip).  Accepts the mode
    number, a bool (True = 1) or the ``args.gemm_precision`` string."""
    if isinstance(bf16, str):
        return GEMM_MODES[bf16]
    return int(bf16)


class GradSink:
    """Where the gradient of one parameter lives (a view into a trainer's flat all-reduce bucket,
    ``distributed.FlatGradBucket``).  Library calls that produce parameter gradients write them there directly — the first
    use of a step overwrites, later uses (``mesh_deform_2`` serves stages 2 and 3) accumulate inside the kernel — instead of
    returning fresh tensors for autograd to assign, add up and the bucket to copy.  ``expect`` counts the forward uses of the
    step whose backward has not run yet; ``on_final`` fires when it returns to zero (the gradient is complete)."""
    __slots__ = ("view", "written", "expect", "on_final")

    def __init__(self, view, on_final=None):
        self.view, self.written, self.expect, self.on_final = view, False, 0, on_final

    def reset(self):
        self.written, self.expect = False, 0


GRAD_SINKS = {}     # id(parameter) -> (weakref to the parameter, GradSink); filled by FlatGradBucket, consulted by GCNStackFn


def register_grad_sink(param, view, on_final=None):
    sink = GradSink(view, on_final)
    key = id(param)
    # the weak reference both proves identity at lookup time (ids are reused once a Parameter is freed) and drops the
    # entry — and with it the reference to the owner's flat buffer — when the parameter dies
    def _gone(ref, key=key):
        if GRAD_SINKS.get(key, (None,))[0] is ref:
            del GRAD_SINKS[key]

    GRAD_SINKS[key] = (weakref.ref(param, _gone), sink)
    return sink


def unregister_grad_sink(param, sink=None):
    """Remove ``param``'s sink (only if it is ``sink``, when given: a newer bucket may have registered its own since)."""
    ent = GRAD_SINKS.get(id(param))
    if ent is not None and ent[0]() is param and (sink is None or ent[1] is sink):
        del GRAD_SINKS[id(param)]


def grad_sink_of(param):
    """The sink registered for exactly this parameter object, or None."""
    ent = GRAD_SINKS.get(id(param))
    return ent[1] if ent is not None and ent[0]() is param else None


class GCNStackFn(torch.autograd.Function):
    """One GCN (reference ``GCN.forward``, vision/model.py:316-331): feats (B,N,ld) -> update (B,N,3)."""

    @staticmethod
    def forward(ctx, feats, adj, in_features, hidden, cut_len, bf16, need_bwd, *params):
        L = _lib.load()
        feats = _req(feats, "feats")
        B, N, ld = feats.shape
        if N != adj.n:
            raise RuntimeError(f"a3vt: features have {N} vertices but the adjacency has {adj.n}")
        weights = [_req(p, "weight") for p in params[0::2]]
        biases = [_req(p, "bias") for p in params[1::2]]
        nl = len(weights)
        mode = gemm_mode(bf16)
        acts = masks = None
        STATS["stack_calls"] += 1
        if need_bwd and nl > 1:
            STATS["stack_stash_calls"] += 1
            ab, mb = ctypes.c_size_t(0), ctypes.c_size_t(0)
            _lib.check(L.a3vt_gcn_stack_stash_bytes(B, N, hidden, nl, cut_len, mode, ctypes.byref(ab), ctypes.byref(mb)),
                       "gcn_stack_stash_bytes")
            acts = torch.empty(ab.value, dtype=torch.uint8, device=feats.device)    # fp32 or bf16 rows, by mode
            masks = torch.empty(mb.value, dtype=torch.uint8, device=feats.device)
        nbytes = L.a3vt_gcn_stack_scratch_bytes_mode(B, N, in_features, hidden, nl, cut_len, 1 if need_bwd else 0, mode)
        scratch = workspace("gcn", nbytes, feats.device)
        update = torch.empty((B, N, 3), dtype=torch.float32, device=feats.device)
        wp, bp = _ptr_array(weights), _ptr_array(biases)
        _lib.check(L.a3vt_gcn_stack_fwd_adj(_lib.ptr(feats), ld, in_features, wp, bp, nl, hidden, cut_len,
                                            _lib.ptr(adj.rowptr), _lib.ptr(adj.col), _lib.ptr(adj.val),
                                            max(adj.max_degree, adj.t_max_degree), adj.split_ref(), N, B, mode,
                                            _lib.ptr(acts), _lib.ptr(masks), _lib.ptr(scratch), _lib.ptr(update), _stream()),
                   "gcn_stack_fwd")
        ctx.adj, ctx.dims, ctx.mode = adj, (in_features, hidden, cut_len, nl), mode
        ctx.acts, ctx.masks = acts, masks
        # gradients written where they live: every parameter of this call has a sink in the trainer's flat bucket
        sinks = [grad_sink_of(p) for p in params] if need_bwd else []
        ctx.sinks = sinks if sinks and all(k is not None and k.view.shape == p.shape and k.view.is_contiguous()
                                           for k, p in zip(sinks, params)) else None
        if ctx.sinks:
            for k in ctx.sinks:
                k.expect += 1
        ctx.save_for_backward(feats, *weights, *biases)
        return update

    @staticmethod
    def backward(ctx, grad_update):
        L = _lib.load()
        in_features, hidden, cut_len, nl = ctx.dims
        feats = ctx.saved_tensors[0]
        weights = list(ctx.saved_tensors[1:1 + nl])
        biases = list(ctx.saved_tensors[1 + nl:1 + 2 * nl])
        adj = ctx.adj
        B, N, ld = feats.shape
        grad_update = _req(grad_update, "grad_update")
        sinks = ctx.sinks
        if sinks and len({k.written for k in sinks}) != 1:     # mixed first / later uses: take the ordinary path
            for k in sinks:
                k.expect -= 1
            sinks = None
        if sinks:
            acc = 1 if sinks[0].written else 0
            gw = [k.view for k in sinks[0::2]]
            gb = [k.view for k in sinks[1::2]]
        else:
            acc = 0
            gw = [torch.empty_like(w) for w in weights]
            gb = [torch.empty_like(b) for b in biases]
        gfeats = torch.empty_like(feats)
        nbytes = L.a3vt_gcn_stack_scratch_bytes_mode(B, N, in_features, hidden, nl, cut_len, 1, ctx.mode)
        scratch = workspace("gcn", nbytes, feats.device)
        _lib.check(L.a3vt_gcn_stack_bwd_adj(_lib.ptr(feats), ld, in_features, _ptr_array(weights), _ptr_array(biases), nl,
                                            hidden, cut_len, _lib.ptr(adj.rowptr), _lib.ptr(adj.col), _lib.ptr(adj.val),
                                            _lib.ptr(adj.t_rowptr), _lib.ptr(adj.t_col), _lib.ptr(adj.t_val),
                                            max(adj.max_degree, adj.t_max_degree), adj.split_ref(), N, B, ctx.mode,
                                            _lib.ptr(ctx.acts), _lib.ptr(ctx.masks), _lib.ptr(grad_update), _ptr_array(gw),
                                            _ptr_array(gb),
                                            _lib.ptr(gfeats), _lib.ptr(scratch), acc, _stream()), "gcn_stack_bwd")
        ctx.acts = ctx.masks = None
        if sinks:   # the gradients are in the bucket already: nothing for autograd to assign or add
            for k in sinks:
                k.written = True
                k.expect -= 1
            for k in sinks:
                if k.expect == 0 and k.on_final is not None:
                    k.on_final()
            return (gfeats, None, None, None, None, None, None, *([None] * (2 * nl)))
        grads = []
        for w, b in zip(gw, gb):
            grads += [w, b]
        return (gfeats, None, None, None, None, None, None, *grads)


def gcn_stack(feats, adj, in_features, hidden, cut_len, weights, biases, bf16=False):
    """``bf16``: gemm mode of the per-vertex products (:func:`gemm_mode`): False / 0 = exact fp32 (default, the parity
    mode); True / 1 / "bf16" = operands rounded to bf16 on their way into the matrix pipe; 2 / "bf16s" = bf16 storage of
    activations and gradients as well (BASELINE configs[3]/[4])."""
    params = []
    for w, b in zip(weights, biases):
        params += [w, b]
    return GCNStackFn.apply(feats, adj, in_features, hidden, cut_len, gemm_mode(bf16), _wants_grad(feats, *params), *params)


class GCNLayerFn(torch.autograd.Function):
    """One GCN layer (reference ``GCN_layer.forward``, vision/model.py:351-363) for callers with their own layer loop:
    x (B,N,ld) -> y (B,N,out).  ``cut_len`` = out for a layer without the cut."""

    @staticmethod
    def forward(ctx, x, adj, weight, bias, cut_len, relu, bf16=False, need_bwd=True):
        L = _lib.load()
        x = _req(x, "features")
        weight, bias = _req(weight, "weight"), _req(bias, "bias")
        B, N, ld = x.shape
        if N != adj.n:
            raise RuntimeError(f"a3vt: features have {N} vertices but the adjacency has {adj.n}")
        kin, nout = weight.shape[-2], weight.shape[-1]
        if ld % 4 != 0 or ld < kin:
            raise RuntimeError(f"a3vt: feature row length {ld} must be a multiple of 4 and >= in_features={kin}")
        ldy = (nout + 3) // 4 * 4
        y = torch.empty((B, N, ldy), dtype=torch.float32, device=x.device)
        scratch = workspace("gcn", L.a3vt_gcn_layer_scratch_bytes(B, N, ld, nout, cut_len, 1 if need_bwd else 0),
                            x.device)
        _lib.check(L.a3vt_gcn_layer_fwd(_lib.ptr(x), ld, kin, _lib.ptr(weight), _lib.ptr(bias), nout, cut_len,
                                        1 if relu else 0, _lib.ptr(adj.rowptr), _lib.ptr(adj.col), _lib.ptr(adj.val),
                                        adj.max_degree, N, B, 1 if bf16 else 0, _lib.ptr(y), ldy, _lib.ptr(scratch),
                                        _stream()), "gcn_layer_fwd")
        ctx.adj, ctx.dims, ctx.bf16 = adj, (kin, nout, cut_len, bool(relu), ldy), bool(bf16)
        ctx.save_for_backward(x, weight, y)
        return y[..., :nout] if ldy != nout else y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.load()
        kin, nout, cut_len, relu, ldy = ctx.dims
        x, weight, y = ctx.saved_tensors
        adj = ctx.adj
        B, N, ld = x.shape
        gy = _req(gy, "grad_output")
        gw, gb, gx = torch.empty_like(weight), torch.empty(nout, dtype=torch.float32, device=x.device), torch.empty_like(x)
        scratch = workspace("gcn", L.a3vt_gcn_layer_scratch_bytes(B, N, ld, nout, cut_len, 1), x.device)
        _lib.check(L.a3vt_gcn_layer_bwd(_lib.ptr(x), ld, kin, _lib.ptr(weight), nout, cut_len, 1 if relu else 0,
                                        _lib.ptr(adj.t_rowptr), _lib.ptr(adj.t_col), _lib.ptr(adj.t_val),
                                        adj.t_max_degree, N, B, 1 if ctx.bf16 else 0,
                                        _lib.ptr(y), ldy, _lib.ptr(gy), gy.shape[-1], _lib.ptr(gw), _lib.ptr(gb),
                                        _lib.ptr(gx), _lib.ptr(scratch), _stream()), "gcn_layer_bwd")
        return gx, None, gw, gb, None, None, None, None


def gcn_layer(x, adj, weight, bias, cut_len, relu, bf16=False):
    """``bf16``: a gemm mode (:func:`gemm_mode`); a lone layer knows the exact products and the bf16 operand mode only
    (modes 1 and 2 -> operand mode, mode 3 "fp32x3" -> exact)."""
    return GCNLayerFn.apply(x, adj, weight, bias, cut_len, relu, gemm_mode(bf16) in (1, 2), _wants_grad(x, weight, bias))


class PosEncMaskFn(torch.autograd.Function):
    """Positional_Encoder + Mask_Encoder + add (vision/model.py:229-232 etc.): (B,N,3),(B,N,1) -> (B,N,ld)."""

    @staticmethod
    def forward(ctx, verts, mask, packed, input_size, ld, gemm_bf16=False):
        L = _lib.load()
        verts, mask, packed = _req(verts, "verts"), _req(mask, "mask"), _req(packed, "pe_params")
        if packed.numel() != L.a3vt_posenc_param_count(input_size):
            raise RuntimeError("a3vt: packed positional-encoder parameter count mismatch")
        B, N, _ = verts.shape
        feats = torch.empty((B, N, ld), dtype=torch.float32, device=verts.device)
        ctx.wide_acts = None
        if input_size == 50:
            _lib.check(L.a3vt_posenc_mask_fwd(_lib.ptr(verts), _lib.ptr(mask), B * N, input_size, _lib.ptr(packed),
                                              _lib.ptr(feats), ld, _stream()), "posenc_mask_fwd")
        else:   # wide inputs (448 of the image models): three products on the matrix pipe, activations kept for the backward
            if not L.a3vt_posenc_wide_supported(input_size) or ld != input_size:
                raise RuntimeError(f"a3vt: vertex-feature encoder: input_size={input_size} ld={ld} unsupported")
            acts = torch.empty(L.a3vt_posenc_wide_acts_bytes(B * N, input_size), dtype=torch.uint8, device=verts.device)
            need_bwd = ctx.needs_input_grad[0] or ctx.needs_input_grad[2]   # (grad mode is off inside forward)
            scratch = workspace("posenc", L.a3vt_posenc_wide_scratch_bytes(B * N, input_size, 0), verts.device)
            _lib.check(L.a3vt_posenc_wide_fwd(_lib.ptr(verts), _lib.ptr(mask), B * N, input_size, _lib.ptr(packed),
                                              _lib.ptr(feats), ld, _lib.ptr(acts), _lib.ptr(scratch), int(bool(gemm_bf16)),
                                              _stream()), "posenc_wide_fwd")
            ctx.wide_acts = acts if need_bwd else None
        ctx.save_for_backward(verts, mask, packed)
        ctx.dims = (input_size, ld)
        ctx.gemm_bf16 = int(bool(gemm_bf16))
        return feats

    @staticmethod
    def backward(ctx, gfeats):
        L = _lib.load()
        verts, mask, packed = ctx.saved_tensors
        input_size, ld = ctx.dims
        gfeats = _req(gfeats, "grad_feats")
        B, N, _ = verts.shape
        gverts = torch.empty_like(verts)
        gparams = torch.empty_like(packed)
        if input_size == 50:
            scratch = workspace("posenc", L.a3vt_posenc_scratch_bytes(B * N, input_size), verts.device)
            _lib.check(L.a3vt_posenc_mask_bwd(_lib.ptr(verts), _lib.ptr(mask), B * N, input_size, _lib.ptr(packed),
                                              _lib.ptr(gfeats), ld, _lib.ptr(gverts), _lib.ptr(gparams),
                                              _lib.ptr(scratch), _stream()), "posenc_mask_bwd")
        else:
            if ctx.wide_acts is None:
                raise RuntimeError("a3vt: vertex-feature encoder: backward without saved activations")
            scratch = workspace("posenc", L.a3vt_posenc_wide_scratch_bytes(B * N, input_size, 1), verts.device)
            _lib.check(L.a3vt_posenc_wide_bwd(_lib.ptr(verts), _lib.ptr(mask), B * N, input_size, _lib.ptr(packed),
                                              _lib.ptr(gfeats), ld, _lib.ptr(ctx.wide_acts), _lib.ptr(gverts),
                                              _lib.ptr(gparams), _lib.ptr(scratch), ctx.gemm_bf16, _stream()), "posenc_wide_bwd")
            ctx.wide_acts = None
        return gverts, None, gparams, None, None, None


class ImagePoolFn(torch.autograd.Function):
    """Image_Encoder.pooling (vision/model.py:70-103): verts (B,N,3) + feature maps (B,C_k,H_k,W_k) -> (B,N,sum C_k).
    Maps are consumed in torch's channels_last memory format (converted here if they are not already)."""

    @staticmethod
    def forward(ctx, verts, matrix, base, *maps):
        """``base`` (optional, (B,N,sum C_k) fp32): the result is ``base + pooled`` — the sum of vision/model.py:243,265,277 in
        the pooling's own pass (``a3vt_image_pool_fwd_add``); its gradient is the output gradient itself."""
        L = _lib.load()
        verts = _req(verts, "verts")
        B, N, _ = verts.shape
        cl = [m.contiguous(memory_format=torch.channels_last) for m in maps]
        for m in cl:
            if not m.is_cuda or m.dtype != torch.float32 or m.shape[0] != B:
                raise RuntimeError("a3vt: feature maps must be float32 (B,C,H,W) tensors on the GPU")
        chans = [int(m.shape[1]) for m in cl]
        ld = sum(chans)                                     # every C_k is a multiple of 4 (checked by the library)
        feats = torch.empty((B, N, ld), dtype=torch.float32, device=verts.device)
        if isinstance(matrix, torch.Tensor):               # host copy of the 3 x 4 camera matrix (a device tensor costs a sync)
            matrix = matrix.detach().cpu().reshape(-1).tolist()
        proj = (ctypes.c_float * 12)(*[float(x) for x in matrix])
        ints = lambda xs: (ctypes.c_int * len(xs))(*xs)  # noqa: E731
        geo = (ints(chans), ints([int(m.shape[2]) for m in cl]), ints([int(m.shape[3]) for m in cl]))
        if base is not None:
            base = _req(base, "base")
            if tuple(base.shape) != (B, N, ld):
                raise RuntimeError(f"a3vt: image_pool base must be {(B, N, ld)}, got {tuple(base.shape)}")
            _lib.check(L.a3vt_image_pool_fwd_add(_lib.ptr(verts), B, N, proj, len(cl), _ptr_array(cl), *geo, _lib.ptr(base),
                                                 _lib.ptr(feats), ld, _stream()), "image_pool_fwd_add")
        else:
            _lib.check(L.a3vt_image_pool_fwd(_lib.ptr(verts), B, N, proj, len(cl), _ptr_array(cl), *geo, _lib.ptr(feats), ld,
                                             _stream()), "image_pool_fwd")
        ctx.save_for_backward(verts, *cl)
        ctx.proj, ctx.geo, ctx.has_base = proj, geo, base is not None
        return feats

    @staticmethod
    def backward(ctx, gfeats):
        L = _lib.load()
        verts, *cl = ctx.saved_tensors
        B, N, _ = verts.shape
        gfeats = _req(gfeats, "grad_feats")
        gmaps = [torch.empty_like(m) for m in cl]          # keeps the channels_last strides
        gverts = torch.empty_like(verts)
        _lib.check(L.a3vt_image_pool_bwd(_lib.ptr(verts), B, N, ctx.proj, len(cl), _ptr_array(cl), *ctx.geo,
                                         _lib.ptr(gfeats), gfeats.shape[-1], _ptr_array(gmaps), _lib.ptr(gverts),
                                         _stream()), "image_pool_bwd")
        return (gverts, None, gfeats if ctx.has_base else None, *gmaps)


def image_pool(verts, matrix, maps, base=None):
    """Image_Encoder.pooling; with ``base``: ``base + pooling`` in one pass (the vertex-feature sum of the image models)."""
    return ImagePoolFn.apply(verts, matrix, base, *maps)


def bias_grad_nhwc(grad):
    """Sum of a (B,C,H,W) channels-last gradient over B, H, W -> fp32 (C,) (``a3vt_bias_grad_nhwc``)."""
    L = _lib.load()
    if not grad.is_cuda or grad.dim() != 4 or grad.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("a3vt: bias_grad_nhwc takes a float32 / bfloat16 (B,C,H,W) tensor on the GPU")
    g = grad.contiguous(memory_format=torch.channels_last)
    B, C, H, W = g.shape
    rows = B * H * W
    need = L.a3vt_bias_grad_scratch_bytes(rows, C)
    if need == 0:
        raise RuntimeError(f"a3vt: bias_grad_nhwc does not support {C} channels")
    scratch = torch.empty(need, dtype=torch.uint8, device=g.device)
    out = torch.empty(C, dtype=torch.float32, device=g.device)
    _lib.check(L.a3vt_bias_grad_nhwc(_lib.ptr(g), int(g.dtype == torch.bfloat16), rows, C, _lib.ptr(out), _lib.ptr(scratch),
                                     need, _stream()), "bias_grad_nhwc")
    return out


_BF16_COPIES = {}   # id(tensor) -> (weakref, version, data_ptr, optimizer epoch, bf16 copy)
_OPT_EPOCH = [0]    # moved by every torch optimizer step of the process once the cache is in use, and by invalidate_bf16_copies()
_BF16_HOOK = [None]  # None: not tried yet; True: the global step hook is registered; False: this torch has none (no caching)


def _optimizer_stepped(*_args, **_kwargs):
    _OPT_EPOCH[0] += 1


def invalidate_bf16_copies():
    """Forget every cached bf16 weight copy.  The cache notices in-place torch ops (``_version``), re-allocations
    (``data_ptr``) and optimizer steps (the step hook); writes through ``.data`` (``p.data.copy_``, ``.data.uniform_``) or a
    hand-written update move none of these and MUST be followed by this call — ``distributed.broadcast_parameters``,
    ``GCN_layer.reset_parameters`` and ``Engine.load`` (vision/train.py) do it themselves; ``load_state_dict`` alone is also
    caught (its ``copy_`` moves ``_version``)."""
    _OPT_EPOCH[0] += 1
    _BF16_COPIES.clear()


def _bf16_cache_on():
    """The process-wide optimizer post-step hook is registered on the FIRST use of the bf16 convolution branch, not at import
    (a host process that imports this module for the fp32 path gets no hook).  The trainer's optimizer — the library's one-launch
    Adam (optim.py) as well as torch's fused Adam — updates the weights in place WITHOUT moving their ``_version``; without a global step hook (older torch) nothing tells a
    stale copy from a fresh one, and every call casts."""
    if _BF16_HOOK[0] is None:
        try:
            from torch.optim.optimizer import register_optimizer_step_post_hook
            register_optimizer_step_post_hook(_optimizer_stepped)
            _BF16_HOOK[0] = True
        except ImportError:
            _BF16_HOOK[0] = False
    return _BF16_HOOK[0]


def _bf16_forget(key, ref):
    hit = _BF16_COPIES.get(key)
    if hit is not None and hit[0] is ref:       # the id can already belong to a new tensor with its own entry
        del _BF16_COPIES[key]


def _bf16_copy(t, channels_last):
    """bf16 (channels-last) copy of a convolution weight / bias, kept until the tensor can have changed: an in-place torch
    op (load_state_dict, copy_) moves ``t._version``; an optimizer step — neither the library's Adam nor torch's fused Adam moves it — moves the
    process-wide optimizer epoch (a global step post-hook); ``.data`` writers call :func:`invalidate_bf16_copies`.  A training
    loop casts once per step as before; the forward-only loops (validation, the K-candidate scoring of
    policies/environment.py:174-180) stop re-casting the 32 weights of the image pyramid on every call (64 copy launches per
    forward).  An entry dies with its tensor (weak-reference callback), so rebuilt models do not pin their copies."""
    on = _bf16_cache_on()
    key = id(t)
    hit = _BF16_COPIES.get(key) if on else None
    if (hit is not None and hit[0]() is t and hit[1] == t._version and hit[2] == t.data_ptr()
            and hit[3] == _OPT_EPOCH[0]):
        return hit[4]
    c = t.detach().to(torch.bfloat16)
    if channels_last:
        c = c.contiguous(memory_format=torch.channels_last)
    if on:
        ref = weakref.ref(t, lambda r, key=key: _bf16_forget(key, r))
        _BF16_COPIES[key] = (ref, t._version, t.data_ptr(), _OPT_EPOCH[0], c)
    return c


def prefetch_bf16_copies(items):
    """Refresh the cached bf16 copies of many convolution weights / biases at once: ``items`` = [(tensor, channels_last)].
    Everything whose cache entry is stale (see :func:`_bf16_copy`) is converted by ONE ``a3vt_cast_weights_bf16`` launch
    instead of two or three torch copy launches per tensor (84 per training step of the image model); the following
    :func:`_bf16_copy` calls hit the cache.  Tensors the kernel does not take (not fp32 / not contiguous) are left to them."""
    if not _bf16_cache_on():
        return
    stale = []
    for t, cl in items:
        hit = _BF16_COPIES.get(id(t))
        if (hit is not None and hit[0]() is t and hit[1] == t._version and hit[2] == t.data_ptr() and hit[3] == _OPT_EPOCH[0]):
            continue
        if t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() in (1, 4):
            stale.append((t, cl))
    L = _lib.load()
    for i0 in range(0, len(stale), 96):
        part = stale[i0:i0 + 96]
        n = len(part)
        dsts = []
        for t, cl in part:
            fmt = torch.channels_last if (cl and t.dim() == 4) else torch.contiguous_format
            dsts.append(torch.empty(t.shape, dtype=torch.bfloat16, device=t.device, memory_format=fmt))
        src = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in part])
        dst = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts])
        outer = (ctypes.c_longlong * n)(*[t.shape[0] for t, _ in part])
        # a channels-last (O, I, KH, KW) tensor is [O][KH KW][I] in memory; everything else keeps its order (inner = 1)
        inner = (ctypes.c_int * n)(*[(t.shape[1] if (cl and t.dim() == 4) else 1) for t, cl in part])
        hw = (ctypes.c_int * n)(*[((t.shape[2] * t.shape[3]) if (cl and t.dim() == 4) else (t.numel() // t.shape[0])) for t, cl in part])
        _lib.check(L.a3vt_cast_weights_bf16(n, src, dst, outer, inner, hw, _stream()), "cast_weights_bf16")
        for (t, _), d in zip(part, dsts):
            key = id(t)
            ref = weakref.ref(t, lambda r, key=key: _bf16_forget(key, r))
            _BF16_COPIES[key] = (ref, t._version, t.data_ptr(), _OPT_EPOCH[0], d)


_CONV5_IMAGES = {}   # (id(weight), flip) -> (weakref, version, data_ptr, optimizer epoch, image); as _BF16_COPIES


def _conv5_image(weight, flip):
    """The bf16 A-fragment image of a (cout,cin,5,5) weight for ``a3vt_conv5_nhwc`` (``flip``: the input-gradient form), cached
    until the weight can have changed (same rules as :func:`_bf16_copy`)."""
    L = _lib.load()
    on = _bf16_cache_on()
    key = (id(weight), flip)
    hit = _CONV5_IMAGES.get(key) if on else None
    if (hit is not None and hit[0]() is weight and hit[1] == weight._version and hit[2] == weight.data_ptr() and hit[3] == _OPT_EPOCH[0]):
        return hit[4]
    w = _req(weight.detach(), "conv weight")
    cout, cin = int(w.shape[0]), int(w.shape[1])
    img = torch.empty(L.a3vt_conv5_image_bytes(cout, cin, flip), dtype=torch.uint8, device=w.device)
    _lib.check(L.a3vt_conv5_weight_image(_lib.ptr(w), cout, cin, flip, _lib.ptr(img), _stream()), "conv5_weight_image")
    if on:
        ref = weakref.ref(weight, lambda r, key=key: _CONV5_IMAGES.pop(key, None))
        _CONV5_IMAGES[key] = (ref, weight._version, weight.data_ptr(), _OPT_EPOCH[0], img)
    return img


def conv5_nhwc(x, image, bias, cout, stride, pad):
    """``a3vt_conv5_nhwc`` on a channels-last bf16 (B,cin,H,W) tensor -> (B,cout,Ho,Wo), channels-last bf16."""
    L = _lib.load()
    B, C, H, W = x.shape
    Ho, Wo = (H + 2 * pad - 5) // stride + 1, (W + 2 * pad - 5) // stride + 1
    y = torch.empty((B, cout, Ho, Wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    _lib.check(L.a3vt_conv5_nhwc(_lib.ptr(x), B, H, W, C, cout, stride, pad, _lib.ptr(image), _lib.ptr(bias), _lib.ptr(y), _stream()),
               "conv5_nhwc")
    return y


def conv5_supported(weight, stride, padding):
    if weight.dim() != 4 or tuple(weight.shape[2:]) != (5, 5) or list(padding) != [1, 1] or stride[0] != stride[1]:
        return False
    return bool(_lib.load().a3vt_conv5_supported(int(weight.shape[1]), int(weight.shape[0]), int(stride[0])))


LIBRARY_CONV5 = [True]   # layers 2-6 of the bf16 image branch on a3vt_conv5_nhwc (False: MIOpen, for A/B)
LIBRARY_CONV5_WRW = [True]   # their weight gradients on a3vt_conv5_weight_grad (False: MIOpen's split-K kernel + fill + cast)
_CONV5_WRW_SCRATCH = {}   # (device index, stream) -> scratch of a3vt_conv5_weight_grad (per-workgroup partial images)


def _conv5_weight_grad(xb, gy, weight, stride, padding, wb):
    """fp32 (cout,cin,5,5) weight gradient of a layer ``conv5_supported`` takes, from the layer's channels-last bf16
    input ``xb`` and output gradient ``gy``: ``a3vt_conv5_weight_grad`` (fixed summation order, no fill / cast launches)."""
    if not LIBRARY_CONV5_WRW[0]:
        return torch.ops.aten.convolution_backward(gy, xb, wb, None, stride, padding, [1, 1], False, [0, 0], 1, [False, True, False])[1]
    L = _lib.load()
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    key = (xb.device.index, int(torch.cuda.current_stream().cuda_stream))
    need = L.a3vt_conv5_wrw_scratch_bytes(32, 32)
    buf = _CONV5_WRW_SCRATCH.get(key)
    if buf is None:
        buf = torch.empty(need, dtype=torch.uint8, device=xb.device)
        _CONV5_WRW_SCRATCH[key] = buf
    gw = torch.empty((cout, cin, 5, 5), dtype=torch.float32, device=xb.device)
    B, _, H, W = xb.shape
    _lib.check(L.a3vt_conv5_weight_grad(_lib.ptr(xb), _lib.ptr(gy), B, H, W, cin, cout, int(stride[0]), _lib.ptr(gw), _lib.ptr(buf), need,
                                        _stream()), "conv5_weight_grad")
    STATS["conv5_weight_grad"] = STATS.get("conv5_weight_grad", 0) + 1
    return gw


class ConvNHWCFn(torch.autograd.Function):
    """``nn.Conv2d`` of the image pyramid (vision/model.py:15-23) in the channels-last bf16 branch: MIOpen's NHWC bf16
    kernels for the convolution and its data / weight gradients, the bias gradient from the library (torch's own column
    reduction of the channels-last output gradient takes 1.3 ms on the 3-channel full-resolution map)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, add_bias=True):
        """``add_bias=False``: the output feeds a training-mode BatchNorm only (``BNReLUFn(..., pre_bias=bias)``), which removes
        the shift again: the bias-add launch is skipped; the bias gradient is formed as before."""
        xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wb = _bf16_copy(weight, True)
        # layers 2-6 of the pyramid (16 -> 16 and 32 -> 32 at stride 1, 16 -> 32 at stride 2): the library's direct convolution
        ctx.own = (LIBRARY_CONV5[0] and conv5_supported(weight, stride, padding) and weight.dtype == torch.float32
                   and bias.dtype == torch.float32 and xb.shape[2] >= 3 and xb.shape[3] >= 3)
        if ctx.own:
            y = conv5_nhwc(xb, _conv5_image(weight, 0), _req(bias.detach(), "conv bias") if add_bias else None,
                           int(weight.shape[0]), int(stride[0]), 1)
            ctx.weight = weight
        else:
            y = torch.ops.aten.convolution(xb, wb, _bf16_copy(bias, False) if add_bias else None, stride, padding, [1, 1], False, [0, 0], 1)
        ctx.save_for_backward(xb, wb)
        ctx.conf = (stride, padding, x.dtype, weight.dtype, bias.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        xb, wb = ctx.saved_tensors
        stride, padding, xdt, wdt, bdt = ctx.conf
        gy = gy.contiguous(memory_format=torch.channels_last)
        own = ctx.own and gy.dtype == torch.bfloat16
        need_gx = ctx.needs_input_grad[0]
        if not own:
            gx, gw, _ = torch.ops.aten.convolution_backward(gy, xb, wb, None, stride, padding, [1, 1], False, [0, 0], 1, [need_gx, True, False])
        else:
            if need_gx and list(stride) == [1, 1] and ctx.weight.shape[1] != 3:
                # the input gradient is the same convolution on gy with the weights transposed and flipped, padding 3
                gx = conv5_nhwc(gy, _conv5_image(ctx.weight, 1), None, int(ctx.weight.shape[1]), 1, 3)
            elif (need_gx and list(stride) == [2, 2] and tuple(ctx.weight.shape[:2]) == (16, 3)
                  and xb.shape[2] == 2 * gy.shape[2] + 2 and xb.shape[3] == 2 * gy.shape[3] + 2):
                # layer 1 (3 -> 16, stride 2): its input gradient as a stride-1 convolution of gy read as if upsampled with zeros
                L = _lib.load()
                gx = torch.empty_like(xb, memory_format=torch.channels_last)
                _lib.check(L.a3vt_conv5_input_grad_3x16s2(_lib.ptr(gy), gy.shape[0], gy.shape[2], gy.shape[3],
                                                          _lib.ptr(_conv5_image(ctx.weight, 1)), _lib.ptr(gx), _stream()), "conv5_input_grad_3x16s2")
            elif need_gx:      # (16 -> 32 at stride 2, 3 -> 3: MIOpen)
                gx = torch.ops.aten.convolution_backward(gy, xb, wb, None, stride, padding, [1, 1], False, [0, 0], 1, [True, False, False])[0]
            else:
                gx = None
            gw = _conv5_weight_grad(xb, gy, ctx.weight, stride, padding, wb)
        cs = getattr(gy, "_a3vt_colsum", None)
        if cs is not None and cs[1] == gy._version and cs[2] == gy.data_ptr() and cs[0].numel() == gy.shape[1]:
            gb = cs[0]            # formed by BNReLUFn.backward while it wrote gy
            STATS["bias_grad_from_bnrelu"] = STATS.get("bias_grad_from_bnrelu", 0) + 1
        else:
            gb = bias_grad_nhwc(gy)
        return (gx.to(xdt) if gx is not None else None), gw.to(wdt, memory_format=torch.contiguous_format), gb.to(bdt), None, None, None


_BNRELU_SCRATCH = {}   # (device index, stream) -> zero-initialised scratch of a3vt_bnrelu_*; the launches leave it zero
_BNRELU_MAX_C = 1024


def _bnrelu_scratch(dev):
    L = _lib.load()
    key = (dev.index, int(torch.cuda.current_stream().cuda_stream))
    buf = _BNRELU_SCRATCH.get(key)
    if buf is None:
        buf = torch.zeros(L.a3vt_bnrelu_scratch_bytes(_BNRELU_MAX_C), dtype=torch.uint8, device=dev)
        _BNRELU_SCRATCH[key] = buf
    return buf


class BNReLUFn(torch.autograd.Function):
    """Training-mode ``nn.BatchNorm2d`` followed by ``nn.ReLU`` of a ``CNN_layer`` (vision/model.py:15-23) on a channels-last
    bf16 map: ``a3vt_bnrelu_fwd / _bwd``, two launches each way.  Updates the module's running statistics and
    ``num_batches_tracked`` in place like the module does.  Returns a channels-last bf16 tensor.  ``pre_bias``: the bias of the
    convolution in front when that convolution did NOT add it (``ConvNHWCFn(..., add_bias=False)``): batch statistics remove a
    per-channel shift, so the output is the same and only ``running_mean`` takes the bias."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, num_batches, eps, momentum, pre_bias=None):
        L = _lib.load()
        if not x.is_cuda or x.dim() != 4 or x.dtype != torch.bfloat16:
            raise RuntimeError("a3vt: bnrelu takes a bfloat16 (B,C,H,W) tensor on the GPU")
        x = x.contiguous(memory_format=torch.channels_last)
        B, C, H, W = x.shape
        if C > _BNRELU_MAX_C:
            raise RuntimeError(f"a3vt: bnrelu supports up to {_BNRELU_MAX_C} channels")
        rows = B * H * W
        gamma, beta = _req(gamma, "bn weight"), _req(beta, "bn bias")
        y = torch.empty_like(x, memory_format=torch.channels_last)
        save = torch.empty((4, C), dtype=torch.float32, device=x.device)
        scratch = _bnrelu_scratch(x.device)
        if pre_bias is not None:
            pre_bias = _req(pre_bias.detach(), "pre_bias")
        _lib.check(L.a3vt_bnrelu_fwd(_lib.ptr(x), rows, C, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(pre_bias), float(eps), float(momentum),
                                     _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(num_batches),
                                     _lib.ptr(y), _lib.ptr(save), _lib.ptr(scratch), scratch.numel(), _stream()), "bnrelu_fwd")
        ctx.save_for_backward(x, save)
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.load()
        x, save = ctx.saved_tensors
        B, C, H, W = x.shape
        gy = gy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(x, memory_format=torch.channels_last)
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        colsum = torch.empty(C, dtype=torch.float32, device=x.device)
        scratch = _bnrelu_scratch(x.device)
        _lib.check(L.a3vt_bnrelu_bwd(_lib.ptr(gy), _lib.ptr(x), B * H * W, C, _lib.ptr(save), _lib.ptr(gx), _lib.ptr(dg),
                                     _lib.ptr(db), _lib.ptr(colsum), _lib.ptr(scratch), scratch.numel(), _stream()), "bnrelu_bwd")
        # x is a Conv2d output in the pyramid: that layer's bias gradient is the column sum of gx, which the dx launch formed
        # on its way (ConvNHWCFn.backward picks it up if THIS tensor, unmodified, arrives as its output gradient)
        gx._a3vt_colsum = (colsum, gx._version, gx.data_ptr())
        return gx, dg, db, None, None, None, None, None, None


class VertexUpdateFn(torch.autograd.Function):
    """vertices[:, :n_vision] += update[:, :n_vision] (vision/model.py:250,270,283), out of place."""

    @staticmethod
    def forward(ctx, verts, update, n_vision):
        L = _lib.load()
        verts, update = _req(verts, "verts"), _req(update, "update")
        B, N, _ = verts.shape
        out = torch.empty_like(verts)
        _lib.check(L.a3vt_vertex_update(_lib.ptr(verts), _lib.ptr(update), B, N, n_vision, _lib.ptr(out), _stream()),
                   "vertex_update")
        ctx.n_vision = n_vision
        return out

    @staticmethod
    def backward(ctx, g):
        nv = ctx.n_vision
        if nv == g.shape[1]:
            return g, g, None
        gu = g.clone()
        gu[:, nv:] = 0
        return g, gu, None


class SamplePointsFn(torch.autograd.Function):
    """batch_sample (utility/utils.py:152-187) for `draws` clouds at once: verts (B,N,3) -> (draws,B,num,3)."""

    @staticmethod
    def forward(ctx, verts, faces, num, draws, seed, offset, face_idx, u, v):
        L = _lib.load()
        verts = _req(verts, "verts")
        faces = _req(faces, "faces", torch.int32)
        B, N, _ = verts.shape
        F = faces.shape[0]
        dev = verts.device
        points = torch.empty((draws, B, num, 3), dtype=torch.float32, device=dev)
        cdf = None
        if face_idx is None:
            cdf = torch.empty((B, F), dtype=torch.float32, device=dev)
            _lib.check(L.a3vt_face_cdf(_lib.ptr(verts), _lib.ptr(faces), B, N, F, _lib.ptr(cdf), _stream()), "face_cdf")
            fi = torch.empty((draws, B, num), dtype=torch.int32, device=dev)
            uu = torch.empty((draws, B, num), dtype=torch.float32, device=dev)
            vv = torch.empty((draws, B, num), dtype=torch.float32, device=dev)
            _lib.check(L.a3vt_sample_points_fwd(_lib.ptr(verts), _lib.ptr(faces), _lib.ptr(cdf), B, N, F, draws, num,
                                                None, None, None, seed, offset, _lib.ptr(points), _lib.ptr(fi),
                                                _lib.ptr(uu), _lib.ptr(vv), _stream()), "sample_points_fwd")
        else:
            fi = _req(face_idx, "face_idx", torch.int32).reshape(draws, B, num)
            uu = _req(u, "u").reshape(draws, B, num)
            vv = _req(v, "v").reshape(draws, B, num)
            _lib.check(L.a3vt_sample_points_fwd(_lib.ptr(verts), _lib.ptr(faces), None, B, N, F, draws, num,
                                                _lib.ptr(fi), _lib.ptr(uu), _lib.ptr(vv), 0, 0, _lib.ptr(points),
                                                None, None, None, _stream()), "sample_points_fwd")
        ctx.save_for_backward(faces, fi, uu, vv)
        ctx.dims = (B, N, F, draws, num)
        return points

    @staticmethod
    def backward(ctx, gpoints):
        L = _lib.load()
        faces, fi, uu, vv = ctx.saved_tensors
        B, N, F, draws, num = ctx.dims
        gpoints = _req(gpoints, "grad_points")
        gverts = torch.empty((B, N, 3), dtype=torch.float32, device=gpoints.device)
        _lib.check(L.a3vt_sample_points_bwd(_lib.ptr(faces), B, N, F, draws, num, _lib.ptr(fi), _lib.ptr(uu),
                                            _lib.ptr(vv), _lib.ptr(gpoints), _lib.ptr(gverts), _stream()),
                   "sample_points_bwd")
        return gverts, None, None, None, None, None, None, None, None


def sample_points(verts, faces, num, draws, seed=0, offset=0, return_samples=False):
    """Forward-only area-weighted sampling on the Philox path; with ``return_samples`` also the draws themselves
    (face_idx int32, u, v — each (draws,B,num)), e.g. to re-inject them into another call."""
    L = _lib.load()
    verts = _req(verts, "verts")
    faces = _req(faces, "faces", torch.int32)
    B, N, _ = verts.shape
    F = faces.shape[0]
    dev = verts.device
    points = torch.empty((draws, B, num, 3), dtype=torch.float32, device=dev)
    cdf = torch.empty((B, F), dtype=torch.float32, device=dev)
    fi = torch.empty((draws, B, num), dtype=torch.int32, device=dev)
    uu = torch.empty((draws, B, num), dtype=torch.float32, device=dev)
    vv = torch.empty((draws, B, num), dtype=torch.float32, device=dev)
    _lib.check(L.a3vt_face_cdf(_lib.ptr(verts), _lib.ptr(faces), B, N, F, _lib.ptr(cdf), _stream()), "face_cdf")
    _lib.check(L.a3vt_sample_points_fwd(_lib.ptr(verts), _lib.ptr(faces), _lib.ptr(cdf), B, N, F, draws, num, None, None,
                                        None, seed, offset, _lib.ptr(points), _lib.ptr(fi), _lib.ptr(uu), _lib.ptr(vv),
                                        _stream()), "sample_points_fwd")
    return (points, fi, uu, vv) if return_samples else points


# Search algorithm ChamferFn asks the library for (include/a3vt.h: a3vt_chamfer_fwd_ws): 0 = let it choose.  All of them
# return the same bits; tests flip this to prove it end to end.
CHAMFER_ALGO = 0


class ChamferFn(torch.autograd.Function):
    """pytorch3d chamfer_distance(x, y, batch_reduction=None) averaged over draws (utility/utils.py:204-217).
    x (draws,B,P,3), y (B,Q,3) -> cd (B,).  Forward-only callers may pass y (E,Q,3) with B a multiple of E: mesh b is compared
    with y[b % E] (the candidate-major batches of ``policies/scoring.py``: the ground truth is sorted once, not per candidate)."""

    @staticmethod
    def forward(ctx, x, y):
        L = _lib.load()
        x, y = _req(x, "x"), _req(y, "y")
        draws, B, P, _ = x.shape
        Q = y.shape[1]
        E = y.shape[0]
        if E != B and (E <= 0 or B % E != 0):
            raise RuntimeError("a3vt: chamfer batch mismatch (a shared ground truth needs B % E == 0)")
        dev = x.device
        dxy = torch.empty((draws, B, P), dtype=torch.float32, device=dev)
        ixy = torch.empty((draws, B, P), dtype=torch.int32, device=dev)
        dyx = torch.empty((draws, B, Q), dtype=torch.float32, device=dev)
        iyx = torch.empty((draws, B, Q), dtype=torch.int32, device=dev)
        cd = torch.empty((B,), dtype=torch.float32, device=dev)
        nbytes = L.a3vt_chamfer_workspace_bytes(draws, B, P, Q)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        _lib.check(L.a3vt_chamfer_fwd_shared(_lib.ptr(x), _lib.ptr(y), draws, B, E, P, Q, _lib.ptr(dxy), _lib.ptr(ixy),
                                             _lib.ptr(dyx), _lib.ptr(iyx), _lib.ptr(cd), _lib.ptr(ws), nbytes, CHAMFER_ALGO,
                                             _stream()), "chamfer_fwd")
        ctx.save_for_backward(x, y, ixy, iyx)
        ctx.aux = (dxy, dyx)
        return cd

    @staticmethod
    def backward(ctx, gcd):
        L = _lib.load()
        x, y, ixy, iyx = ctx.saved_tensors
        draws, B, P, _ = x.shape
        Q = y.shape[1]
        if y.shape[0] != B:
            raise RuntimeError("a3vt: the shared-ground-truth Chamfer (y with fewer clouds than x has meshes) is forward-only "
                               "(scoring under no_grad); repeat y for a differentiable call")
        gcd = _req(gcd, "grad_cd")
        gx = torch.empty_like(x)
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        _lib.check(L.a3vt_chamfer_bwd(_lib.ptr(x), _lib.ptr(y), draws, B, P, Q, _lib.ptr(ixy), _lib.ptr(iyx),
                                      _lib.ptr(gcd), _lib.ptr(gx), _lib.ptr(gy), _stream()), "chamfer_bwd")
        return gx, gy


NN_ALGOS = {"auto": 0, "two_pass": 1, "sweep": 2, "pruned": 3}


def chamfer_nn(x, y, single_pass=True, algo=None):
    """Raw nearest-neighbour outputs (dist_xy, idx_xy, dist_yx, idx_yx, cd) — used by tests and scoring.
    ``algo``: "auto" | "two_pass" | "sweep" | "pruned" (include/a3vt.h: a3vt_chamfer_fwd_ws); by default the brute-force
    sweep, or with ``single_pass=False`` the two-pass search (no scratch).  The results are identical bit for bit."""
    L = _lib.load()
    x, y = _req(x, "x"), _req(y, "y")
    draws, B, P, _ = x.shape
    Q = y.shape[1]
    E = y.shape[0]                     # E < B: shared ground truth, mesh b against y[b % E] (a3vt_chamfer_fwd_shared)
    dev = x.device
    dxy = torch.empty((draws, B, P), dtype=torch.float32, device=dev)
    ixy = torch.empty((draws, B, P), dtype=torch.int32, device=dev)
    dyx = torch.empty((draws, B, Q), dtype=torch.float32, device=dev)
    iyx = torch.empty((draws, B, Q), dtype=torch.int32, device=dev)
    cd = torch.empty((B,), dtype=torch.float32, device=dev)
    if algo is None:
        algo = "sweep" if single_pass else "two_pass"
    nbytes = L.a3vt_chamfer_workspace_bytes(draws, B, P, Q) if algo != "two_pass" else 0
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev) if nbytes else None
    _lib.check(L.a3vt_chamfer_fwd_shared(_lib.ptr(x), _lib.ptr(y), draws, B, E, P, Q, _lib.ptr(dxy), _lib.ptr(ixy),
                                         _lib.ptr(dyx), _lib.ptr(iyx), _lib.ptr(cd), _lib.ptr(ws), nbytes, NN_ALGOS[algo],
                                         _stream()), "chamfer_fwd")
    return dxy, ixy, dyx, iyx, cd


def rowgemm(a, w, bf16=False):
    """C = A @ W on the MFMA kernel (tests / bench): a (M,K) with K % 4 == 0, w (K,N) with N <= 304."""
    L = _lib.load()
    a, w = _req(a, "a"), _req(w, "w")
    M, K = a.shape
    N = w.shape[1]
    wt = torch.empty((L.a3vt_wt_rows(N), L.a3vt_wt_ld(K)), dtype=torch.float32, device=a.device)
    _lib.check(L.a3vt_transpose_weight(_lib.ptr(w), K, N, _lib.ptr(wt), _stream()), "transpose_weight")
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _lib.check(L.a3vt_rowgemm(_lib.ptr(a), K, M, K, _lib.ptr(wt), N, 1 if bf16 else 0, _lib.ptr(c), N, _stream()),
               "rowgemm")
    return c


CSR_ALGOS = {"auto": 0, "rows": 1, "sliced": 2}


def dbg_csr_algo(name):
    """Test hook (``a3vt_dbg_csr_algo``): force the neighbour-aggregation kernels of the fp32 stacks — "auto" (default, by
    shape), "rows" (half-wave per vertex) or "sliced" (channel-sliced wherever a mesh slice fits LDS).  Same outputs."""
    _lib.check(_lib.load().a3vt_dbg_csr_algo(CSR_ALGOS[name]), "dbg_csr_algo")


def split3(x):
    """The exact operand split of gemm mode 3 (``a3vt_split3_bf16``): three int16 tensors of bf16 bit patterns with
    ``hi + mid + lo == x``."""
    L = _lib.load()
    x = _req(x, "x")
    hi, mid, lo = (torch.empty(x.shape, dtype=torch.int16, device=x.device) for _ in range(3))
    _lib.check(L.a3vt_split3_bf16(_lib.ptr(x), x.numel(), _lib.ptr(hi), _lib.ptr(mid), _lib.ptr(lo), _stream()), "split3_bf16")
    return hi, mid, lo


def check_finite(t, flag):
    """Deferred NaN/Inf check: ORs 1 into `flag` (int32 device scalar) without syncing the host."""
    L = _lib.load()
    _lib.check(L.a3vt_check_finite(_lib.ptr(t), t.numel(), _lib.ptr(flag), _stream()), "check_finite")
