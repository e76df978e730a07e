"""Drop-in for ``pterotactyl/reconstruction/vision/model.py`` (image-free modes) on the MI355X HIP path.

Public names, constructor signatures, forward semantics and the state-dict layout match the reference
(``Deformation`` :168-286, ``GCN`` :290-331, ``GCN_layer`` :335-363, ``Positional_Encoder`` :367-399,
``Mask_Encoder`` :403-414, ``prepare_mesh`` :418-439), so reference checkpoints load with
``load_state_dict`` and callers such as ``policies/environment.py:125-129,225`` work unchanged.
Parameters are created with the same torch initialisers in the same order as the reference constructor,
so the same ``torch.manual_seed`` gives bit-identical weights.

What differs underneath: a whole ``GCN`` (20 layers) is ONE call into the HIP library (fp32 MFMA
per-vertex products + CSR neighbour aggregation, activations saved for backward), the positional
encoder + mask embedding is one fused kernel, and nothing synchronises the host (the reference's NaN
trap at :326-329 becomes an optional deferred flag, ``Deformation.finite_flag``).

``use_img=True``: ``Image_Encoder`` (:27-164) keeps its convolutions as torch modules on MIOpen (SURVEY §2b
K13: not hand-written); the per-vertex projection + bilinear pooling is the fused HIP kernel of
``csrc/pooling.hip`` (``a3vt_image_pool_fwd/bwd``), and the 448-wide vertex features go through the same HIP
GCN stack.
"""
import math
import weakref

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from .... import ops as _ops
from ....mesh import CSRAdjacency


def _cut_len(out_features, cut):
    return round(out_features * cut)  # python banker's rounding, as the reference (:355)


def _csr_of(adj_info, key):
    """CSR handle for adj_info[key] ('adj' | 'origional'); converts + caches a dense-only (reference-made) dict."""
    ck = "csr" if key == "adj" else "csr_origional"
    if dict.__contains__(adj_info, ck):
        return dict.__getitem__(adj_info, ck)
    dense = adj_info[key]
    csr = _ops.DeviceCSR(CSRAdjacency.from_dense(dense.detach().cpu().numpy()), dense.device)
    try:
        adj_info[ck] = csr
    except TypeError:
        pass
    return csr


_DENSE_CSR_CACHE = {}


def _csr_from_arg(adj):
    """``GCN_layer.forward`` takes the adjacency as an argument: accept a CSR handle or a dense tensor."""
    if isinstance(adj, _ops.DeviceCSR):
        return adj
    if not isinstance(adj, torch.Tensor) or adj.dim() != 2 or adj.shape[0] != adj.shape[1]:
        raise RuntimeError("a3vt: GCN_layer needs a dense (N,N) adjacency tensor or an ops.DeviceCSR handle")
    ent = _DENSE_CSR_CACHE.get(id(adj))
    if ent is not None and ent[0]() is adj and ent[1] == adj._version:
        return ent[2]
    for k in [k for k, e in _DENSE_CSR_CACHE.items() if e[0]() is None]:  # drop entries of dead tensors
        del _DENSE_CSR_CACHE[k]
    csr = _ops.DeviceCSR(CSRAdjacency.from_dense(adj.detach().cpu().numpy()), adj.device)
    _DENSE_CSR_CACHE[id(adj)] = (weakref.ref(adj), adj._version, csr)
    return csr


def _cnn_block(f_in, f_out, k, stride=1, simple=False, padding=1):
    """BN -> ReLU -> Conv (or a bare Conv when ``simple``): same Sequential indices as the reference (:15-23), so
    state-dict keys read ``layers.<n>.0.*`` (BN or the bare conv) and ``layers.<n>.2.*`` (conv)."""
    mods = [] if simple else [nn.BatchNorm2d(int(f_in)), nn.ReLU(inplace=True)]
    mods.append(nn.Conv2d(int(f_in), int(f_out), kernel_size=k, padding=padding, stride=stride))
    return nn.Sequential(*mods)


# camera of the rendered dataset images (reference :50-64): intrinsics f = 221.7025, c = (128, 128)
_CAM_RT = [[-7.587616579485257e-08, -1.0000001192092896, 0.0, -2.2762851159541242e-08],
           [-0.7071068286895752, 7.587616579485257e-08, -0.7071068286895752, 0.0],
           [0.7071068286895752, 0.0, -0.7071067690849304, 0.4242640733718872]]
_CAM_F = 221.7025


class Image_Encoder(nn.Module):
    """Image pyramid + per-vertex feature pooling (reference :27-103).  The convolutions run on torch ops (MIOpen,
    SURVEY §2b K13); the pooling is the fused HIP kernel of csrc/pooling.hip."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        k = args.CNN_ker_size
        blocks = [_cnn_block(3, 3, k, stride=1, simple=True)]
        cur, nxt = 3, 16
        for _ in range(args.num_CNN_blocks):
            blocks.append(_cnn_block(cur, nxt, k, stride=2))
            cur, nxt = nxt, nxt * 2
            blocks += [_cnn_block(cur, cur, k) for _ in range(args.layers_per_block - 1)]
        self.layers = nn.ModuleList(blocks)
        K = np.array([[_CAM_F, 0, 128.0], [0, _CAM_F, 128.0], [0, 0, 1]])
        self.register_buffer("matrix", torch.FloatTensor(K.dot(np.array(_CAM_RT))), persistent=False)
        self._matrix_host = [float(x) for x in self.matrix.reshape(-1).tolist()]   # fp32 values, no device sync later

    def _bf16_branch(self, img):
        """The bf16 configurations (BASELINE configs[3] "bf16"; ``args.gemm_precision`` in {"bf16", "bf16s"}, overridable with
        ``args.cnn_precision``) run the convolution stack channels-last under bf16 autocast: fp32 master weights and
        BatchNorm statistics, bf16 activations — MIOpen picks its bf16 NHWC kernels, the NCHW <-> NHWC transposes
        between layers disappear and the BatchNorm passes move half the bytes.  fp32 stays the default (golden ``g8``)."""
        prec = getattr(self.args, "cnn_precision", None) or getattr(self.args, "gemm_precision", "fp32")
        return img.is_cuda and prec in ("bf16", "bf16s")

    library_bias_grad = True   # (tools flip it to time torch's own reduction)
    fused_bn_relu = True       # training BatchNorm2d + ReLU through a3vt_bnrelu_* (False: MIOpen's BatchNorm + torch's ReLU)
    fold_conv_bias = True      # a convolution in front of a fused BatchNorm leaves its bias to that operator (no bias-add launch)
    # MIOpen's training BatchNorm crashes the HOST on bf16 NHWC input at batch sizes below 4 (seen for 3 x 254^2, 3 x 64^2 and
    # 64 x 27^2 maps on the ROCm 7.2 image this was built on: tools/experiments/miopen_bn_nhwc_c3_crash.py).  Batches
    # smaller than this take the NCHW kernel instead (a last partial batch of an epoch, or a small per-rank shard: same
    # values to bf16 rounding, ~5 ms slower per step at batch 64).  A MIOpen build without the defect: set it to 0
    # (``Image_Encoder.bn_nhwc_min_batch = 0``); the fallback is reported once per process.
    bn_nhwc_min_batch = 4
    _bn_fallback_reported = False

    @staticmethod
    def _bn_fusable(m, x):
        """A training-mode BatchNorm2d the library's BatchNorm + ReLU operator takes (csrc/bnrelu.hip)."""
        return (Image_Encoder.fused_bn_relu and isinstance(m, nn.BatchNorm2d) and m.training and m.affine and m.track_running_stats
                and m.momentum is not None and x.dtype == torch.bfloat16 and m.running_mean.dtype == torch.float32
                and m.num_batches_tracked.dtype == torch.int64)

    @staticmethod
    def _block_nhwc(block, x, pre_bias=None, fold_bias=False):
        """One ``CNN_layer`` Sequential in the bf16 channels-last branch: BatchNorm / ReLU as they are (MIOpen under
        autocast), the convolution through ``ops.ConvNHWCFn`` (same MIOpen kernels, bias gradient from the library).
        ``pre_bias``: the bias the previous block's convolution did not add (this block's fused BatchNorm accounts for it);
        ``fold_bias``: this block's convolution leaves its bias to the next block's BatchNorm.  Returns (x, bias left out)."""
        mods, skip, left_out = list(block), False, None
        for i, m in enumerate(mods):
            if skip:          # the ReLU behind a fused BatchNorm
                skip = False
                continue
            if Image_Encoder._bn_fusable(m, x) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                # BatchNorm2d + ReLU as one operator (csrc/bnrelu.hip): three launches each way, any batch size
                x = _ops.BNReLUFn.apply(x, m.weight, m.bias, m.running_mean, m.running_var, m.num_batches_tracked, m.eps, m.momentum,
                                        pre_bias)
                pre_bias = None
                skip = True
            elif isinstance(m, nn.BatchNorm2d) and m.training and x.shape[0] < Image_Encoder.bn_nhwc_min_batch:
                if not Image_Encoder._bn_fallback_reported:
                    Image_Encoder._bn_fallback_reported = True
                    import warnings
                    warnings.warn(f"a3vt: batch of {x.shape[0]} < {Image_Encoder.bn_nhwc_min_batch}: training BatchNorm of the bf16 "
                                  "channels-last image encoder takes the NCHW kernel (MIOpen bf16 NHWC BatchNorm defect at "
                                  "tiny batches; Image_Encoder.bn_nhwc_min_batch = 0 disables the workaround)", stacklevel=2)
                x = m(x.contiguous()).contiguous(memory_format=torch.channels_last)
            elif not isinstance(m, nn.Conv2d):
                x = m(x)
            elif Image_Encoder.library_bias_grad and m.bias is not None and m.groups == 1 and tuple(m.dilation) == (1, 1):
                x = _ops.ConvNHWCFn.apply(x, m.weight, m.bias, list(m.stride), list(m.padding), not fold_bias)
                left_out = m.bias if fold_bias else None
            else:   # torch's own autocast convolution on a channels-last view of the weights
                x = nn.functional.conv2d(x, m.weight.contiguous(memory_format=torch.channels_last), m.bias, m.stride,
                                         m.padding, m.dilation, m.groups)
        if pre_bias is not None:      # (never: a bias is only left out in front of a BatchNorm this function fuses)
            raise RuntimeError("a3vt: a convolution bias was left to a BatchNorm that did not take it")
        return x, left_out

    def forward(self, img):
        """Feature maps of the three layers ``layers_per_block`` apart from the end, plus the last map reached
        before the spatial size drops below the kernel size (:147-164)."""
        n = len(self.layers)
        picks = {n - 1 - (i + 1) * self.args.layers_per_block for i in range(3)}
        low = self._bf16_branch(img)   # (parameters keep their NCHW strides — the flat gradient bucket and fused Adam see one
        #                                layout; the convolutions take a channels-last bf16 copy of their weights per call)
        x, maps = (img.contiguous(memory_format=torch.channels_last) if low else img), []
        if low and Image_Encoder.library_bias_grad:   # every weight / bias this call will use -> bf16 in one launch (cached per step)
            size, convs = img.shape[-1], []
            for layer in self.layers:
                if size < self.args.CNN_ker_size:
                    break
                for m in layer:
                    if isinstance(m, nn.Conv2d):
                        convs.append(m)
                        size = (size + 2 * m.padding[0] - m.kernel_size[0]) // m.stride[0] + 1
            _ops.prefetch_bf16_copies([(m.weight, True) for m in convs] + [(m.bias, False) for m in convs if m.bias is not None])
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=low):
            pre_bias = None
            for e, layer in enumerate(self.layers):
                if x.shape[-1] < self.args.CNN_ker_size:
                    break
                if low:
                    # This layer's convolution may leave its bias to the NEXT layer's fused BatchNorm (batch statistics remove a
                    # per-channel shift: same output, one launch per layer and step less) when nothing else reads its output:
                    # it is not pooled, the next layer runs, and that layer starts with a BatchNorm the operator takes.
                    fold = False
                    if self.fold_conv_bias and e not in picks and e + 1 < len(self.layers):
                        conv = [m for m in layer if isinstance(m, nn.Conv2d)][0]
                        out = (x.shape[-1] + 2 * conv.padding[0] - conv.kernel_size[0]) // conv.stride[0] + 1
                        nxt = list(self.layers[e + 1])
                        fold = (out >= self.args.CNN_ker_size and self.library_bias_grad and conv.bias is not None
                                and conv.groups == 1 and tuple(conv.dilation) == (1, 1)
                                and isinstance(nxt[0], nn.BatchNorm2d) and len(nxt) > 1 and isinstance(nxt[1], nn.ReLU)
                                and self._bn_fusable(nxt[0], torch.empty(0, dtype=torch.bfloat16)))
                    x, pre_bias = self._block_nhwc(layer, x, pre_bias, fold)
                else:
                    x = layer(x)
                if e in picks:
                    maps.append(x)
            maps.append(x)
        if low:   # the pooling kernel reads fp32 channels-last maps (three small tensors)
            maps = [m.float().contiguous(memory_format=torch.channels_last) for m in maps]
        return maps

    def pooling(self, blocks, verts_pos, base=None):
        """Project vertices with K.RT, bilinear-sample every map at the projected pixel, concatenate (:70-103) — one
        fused HIP kernel (``a3vt_image_pool_fwd/bwd``) instead of three ``grid_sample`` calls and the concatenations."""
        return _ops.image_pool(verts_pos.to(torch.float32), self._matrix_host, list(blocks), base=base)


_NERF_FREQS = [np.pi if i == 0 else np.pi * 2 * i for i in range(10)]   # reference :383-389


def _nerf_embedding(p):
    """[sin(f0 p), cos(f0 p), sin(f1 p), ...] for p (M,3) -> (M,60), frequencies as the reference (:381-391).  One
    broadcast product and one sin / cos each instead of 20 small launches and a 20-way cat (same values: the python
    scalars of the reference's loop are rounded to fp32 before the multiply either way)."""
    f = torch.tensor(_NERF_FREQS, dtype=p.dtype, device=p.device)
    ang = p[:, None, :] * f[None, :, None]                                   # (M,10,3)
    return torch.stack((torch.sin(ang), torch.cos(ang)), dim=2).reshape(p.shape[0], 60)


class GCN_layer(nn.Module):
    """Parameter holder with the reference's shapes/init (:336-349).  The arithmetic of a layer runs inside
    ``GCN.forward`` (one library call for the whole stack)."""

    def __init__(self, in_features, out_features, cut=0.33, do_cut=True):
        super().__init__()
        self.weight = Parameter(torch.empty(1, in_features, out_features))
        self.bias = Parameter(torch.empty(out_features))
        self.cut_size = cut
        self.do_cut = do_cut
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 6.0 / math.sqrt(self.weight.size(1) + self.weight.size(0))
        stdv *= 0.3
        self.weight.data.uniform_(-stdv, stdv)
        self.bias.data.uniform_(-0.1, 0.1)
        _ops.invalidate_bf16_copies()               # .data writes move no version counter

    def forward(self, features, adj, activation):
        """The reference's layer call (:351-363) for callers with their own layer loop (the auto-encoder encoder,
        the DDQN graph model).  ``adj`` is a dense row-normalised (N,N) tensor — converted to CSR once and cached — or
        an ``ops.DeviceCSR`` handle.  ReLU is fused; any other ``activation`` is applied to the layer output."""
        csr = _csr_from_arg(adj)
        nout = self.weight.shape[-1]
        c = _cut_len(nout, self.cut_size) if self.do_cut else nout
        relu = activation is F.relu or activation is torch.relu
        pad = (-features.shape[-1]) % 4
        if pad:
            features = F.pad(features, (0, pad))
        y = _ops.gcn_layer(features, csr, self.weight, self.bias, c, relu, getattr(self, "gemm_bf16", False))
        return y if relu else activation(y)


class GCN(nn.Module):
    def __init__(self, input_features, args, ignore_touch_matrix=False):
        super().__init__()
        self.ignore_touch_matrix = ignore_touch_matrix
        self.num_layers = args.num_GCN_layers
        self.input_features = input_features
        self.hidden = args.hidden_GCN_size
        self.cut = args.cut
        # new knob (no reference counterpart), BASELINE configs[3]/[4]: "bf16" rounds the operands of the per-vertex
        # products to bf16 on their way into the matrix pipe (everything stored stays fp32); "bf16s" also stores the
        # activations, their gradients and the weight images as bf16 (fp32 accumulation, fp32 master weights / Adam).
        # Default "fp32": exact fp32 MFMA, the parity mode.
        self.gemm_bf16 = _ops.gemm_mode(getattr(args, "gemm_precision", "fp32"))
        dims = [input_features] + [self.hidden] * (self.num_layers - 1) + [3]
        self.layers = nn.ModuleList(
            [GCN_layer(dims[i], dims[i + 1], args.cut, do_cut=i < self.num_layers - 1) for i in range(self.num_layers)])

    def forward(self, features, adj_info):
        """features (B,N,ld), ld >= input_features with zero pad columns -> (B,N,3).  The library takes rows of exactly
        pad4(input_features) floats (a3vt.h): narrower rows are padded here, wider ones cut back to that width."""
        adj = _csr_of(adj_info, "origional" if self.ignore_touch_matrix else "adj")
        ld, want = features.shape[-1], (self.input_features + 3) // 4 * 4
        if ld < want:    # callers outside Deformation may hand in exactly input_features columns
            features = torch.nn.functional.pad(features, (0, want - ld))
        elif ld > want:  # or rows padded further than the 4-float granule
            features = features[..., :want].contiguous()
        ws = [l.weight for l in self.layers]
        bs = [l.bias for l in self.layers]
        return _ops.gcn_stack(features, adj, self.input_features, self.hidden, _cut_len(self.hidden, self.cut), ws, bs,
                              bf16=self.gemm_bf16)


class Positional_Encoder(nn.Module):
    def __init__(self, input_size):
        super().__init__()
        self.input_size = input_size
        self.model = nn.Sequential(
            nn.Linear(63, input_size // 4), nn.ReLU(inplace=True),
            nn.Linear(input_size // 4, input_size // 2), nn.ReLU(inplace=True),
            nn.Linear(input_size // 2, input_size))

    def packed(self):
        m = self.model
        return [m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias]

    def forward(self, positions):
        """torch-op path (GPU); Deformation uses the fused HIP kernel instead when input_size == 50."""
        shape = positions.shape
        p = positions.contiguous().view(shape[0] * shape[1], -1)
        return self.model(torch.cat((_nerf_embedding(p), p), dim=-1)).view(shape[0], shape[1], -1)


class _TokenEmbedFn(torch.autograd.Function):
    """Row gather from a tiny table (4 mask tokens) whose backward is one matrix product, ``onehot^T @ grad``: torch's
    embedding backward sorts the 10^5 indices and scatters (3 ms per call at 64 x 1924 vertices x 448 channels, 12 % of a
    configs[3] step), and ``index_add_`` would be a float-atomics scatter.  The product is deterministic."""

    @staticmethod
    def forward(ctx, weight, idx):
        ctx.save_for_backward(idx)
        ctx.rows = weight.shape[0]
        return weight.index_select(0, idx)

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        onehot = torch.zeros(idx.numel(), ctx.rows, dtype=grad.dtype, device=grad.device)
        onehot.scatter_(1, idx.view(-1, 1), 1.0)
        return onehot.t() @ grad, None


class Mask_Encoder(nn.Module):
    def __init__(self, input_size):
        super().__init__()
        self.model = nn.Sequential(nn.Embedding(4, input_size))   # same state-dict key as the reference (:406)

    def forward(self, mask):
        shape = mask.shape
        idx = mask.contiguous().view(-1).long()
        return _TokenEmbedFn.apply(self.model[0].weight, idx).view(shape[0], shape[1], -1)


class Deformation(nn.Module):
    def __init__(self, adj_info, inital_positions, args, return_img=False, pass_img=False):
        super().__init__()
        self.adj_info = adj_info
        self.initial_positions = inital_positions
        self.args = args
        self.return_img = return_img
        self.pass_img = pass_img
        self.num_stages = getattr(args, "num_stages", 3)  # new knob; the reference always runs 3 (:216-283)
        # construction order = reference order (:180-201) so seeds reproduce the reference init
        if getattr(args, "use_img", False):
            self.img_encoder_global = Image_Encoder(args)
            self.img_encoder_local = Image_Encoder(args)
            with torch.no_grad():  # feature width = channels of the maps a 256x256 image yields (:183-190; 448 by default)
                maps = self.img_encoder_global(torch.zeros(1, 3, 256, 256))
                input_size = sum(m.shape[1] for m in maps)
        else:
            input_size = 50  # :193
        self.input_size = input_size
        self.ld_feats = (input_size + 3) // 4 * 4
        self.positional_encoder = Positional_Encoder(input_size)
        self.mask_encoder = Mask_Encoder(input_size)
        self.mesh_deform_1 = GCN(input_size, args, ignore_touch_matrix=getattr(args, "use_img", False))
        self.mesh_deform_2 = GCN(input_size, args)
        self.finite_flag = None  # set to an int32 device scalar to enable the deferred NaN/Inf check

    def _fused_encoder(self):
        if self.input_size == 50:
            return True
        from .... import lib as _alib
        return self.ld_feats == self.input_size and bool(_alib.load().a3vt_posenc_wide_supported(self.input_size))

    def _packed_encoder_params(self):
        return torch.cat([p.reshape(-1) for p in self.positional_encoder.packed()] +
                         [self.mask_encoder.model[0].weight.reshape(-1)])

    def _vertex_features(self, vertices, mask, packed, img_maps):
        if packed is not None:
            # I = 50: the LDS-resident fused kernel; wide inputs (448 of the image models): three products on the matrix
            # pipe (csrc/posenc_wide.hip) — both behind the same C-ABI pair and autograd function
            feats = _ops.PosEncMaskFn.apply(vertices, mask, packed, self.input_size, self.ld_feats,
                                            getattr(self.args, "gemm_precision", "fp32") in ("bf16", "bf16s"))
        else:  # input sizes the library does not take (not a multiple of 8): torch ops, padded to the 4-float row granule
            feats = self.positional_encoder(vertices) + self.mask_encoder(mask)
        if img_maps is not None:
            enc = self.img_encoder_global                                                  # always the global encoder's
            if feats.is_cuda and feats.dtype == torch.float32 and feats.shape[-1] == sum(int(m.shape[1]) for m in img_maps):
                feats = enc.pooling(img_maps, vertices, base=feats)     # the sum in the pooling's own pass
            else:
                feats = feats + enc.pooling(img_maps, vertices)
        if packed is None and self.ld_feats != self.input_size:                       # projection (:243,265,277)
            feats = F.pad(feats, (0, self.ld_feats - self.input_size))
        return feats.contiguous()

    def forward(self, img, charts, img_features=None):
        use_img = getattr(self.args, "use_img", False)
        if self.pass_img and img_features is not None:
            global_maps, local_maps = img_features
        elif use_img:
            img = img.to(charts["vision_charts"].device)
            if self.img_encoder_global._bf16_branch(img) and not img.requires_grad:
                # both encoders read the image channels-last in bf16: one conversion for the two (each made its own: 4 copies)
                img = img.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            global_maps, local_maps = self.img_encoder_global(img), self.img_encoder_local(img)
        else:
            global_maps, local_maps = [], []
        vertices, mask = self.deform_with_maps(charts, global_maps, local_maps)
        if self.return_img:
            return vertices, mask, [global_maps, local_maps]
        return vertices, mask

    def deform_with_maps(self, charts, global_maps, local_maps):
        """The three refinement stages (:216-283) on already computed image feature maps (empty lists without
        ``use_img``).  Split out so a caller scoring many candidate touches of the same images (policies/scoring.py)
        runs the image encoders once."""
        use_img, use_touch = getattr(self.args, "use_img", False), self.args.use_touch
        vc = charts["vision_charts"].shape[1]
        f32 = lambda t: t.to(torch.float32).contiguous()  # noqa: E731
        touch_in_stage1 = use_touch and not use_img       # touch-only models see the touch charts immediately (:218)
        if touch_in_stage1:
            vertices = f32(torch.cat((charts["vision_charts"], charts["touch_charts"]), dim=1))
            mask = f32(torch.cat((charts["vision_masks"], charts["touch_masks"]), dim=1))
        else:
            vertices, mask = f32(charts["vision_charts"]), f32(charts["vision_masks"])
        packed = self._packed_encoder_params() if self._fused_encoder() else None
        for stage in range(self.num_stages):
            if stage == 1 and use_touch and use_img:      # vision+touch models add the touch charts now (:254-259)
                vertices = f32(torch.cat((vertices, charts["touch_charts"]), dim=1))
                mask = f32(torch.cat((charts["vision_masks"], charts["touch_masks"]), dim=1))
            gcn = self.mesh_deform_1 if stage == 0 else self.mesh_deform_2  # stages 2 and 3 share weights (:268,281)
            maps = (global_maps if stage == 0 else local_maps) if use_img else None
            feats = self._vertex_features(vertices, mask, packed, maps)
            update = gcn(feats, self.adj_info)
            vertices = _ops.VertexUpdateFn.apply(vertices, update, vc)
            if self.finite_flag is not None:
                _ops.check_finite(update, self.finite_flag)
        return vertices, mask


def prepare_mesh(batch, vision_mesh, args):
    """Loader batch -> charts dict (:418-439).  Tensors are moved to the device of ``vision_mesh``."""
    s1 = batch["img"].shape[0]
    dev = vision_mesh.device
    vision_charts = vision_mesh.unsqueeze(0).repeat(s1, 1, 1)
    vision_masks = 3 * torch.ones(vision_charts.shape[:-1], device=dev).unsqueeze(-1)
    charts = {"vision_charts": vision_charts, "vision_masks": vision_masks}
    if args.use_touch:
        touch_info = batch["touch_charts"].to(dev).view(s1, -1, 4)
        charts["touch_charts"] = touch_info[:, :, :3]
        charts["touch_masks"] = touch_info[:, :, 3:]
    return charts
