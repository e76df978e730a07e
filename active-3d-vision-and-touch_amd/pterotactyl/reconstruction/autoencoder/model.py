"""``pterotactyl.reconstruction.autoencoder.model`` on MI355X — a consumer of the hot-path kernels.

Same public classes, constructor arguments and state-dict keys as the reference module
(``reconstruction/autoencoder/model.py``): ``AutoEncoder`` (:15-41), ``Encoder`` (:45-92), ``Decoder`` (:126-137),
the FoldingNet folds (:141-208).  What runs where:

* vertex features: the fused positional + mask encoder kernel (``a3vt_posenc_mask_fwd/bwd``), as in ``Deformation``;
* the encoder's graph layers: one ``a3vt_gcn_layer_fwd/bwd`` call per layer through ``vision.model.GCN_layer`` — every
  layer is hidden -> hidden here, the last one without the cut and without ReLU (:57-64, 87-89), shapes the stack
  entry points do not cover;
* vertex max-pool, the latent MLP and the FoldingNet decoder (1x1 convolutions on an 80x80 grid): torch ops on the
  GPU — they are not on the north_star path.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops as _ops
from ..vision.model import GCN_layer, Mask_Encoder, Positional_Encoder, _csr_of  # noqa: F401  (public names, :96,210,246)


class AutoEncoder(nn.Module):
    def __init__(self, adj_info, inital_positions, args, only_encode=False):
        super().__init__()
        self.adj_info = adj_info
        self.initial_positions = inital_positions
        self.args = args
        self.only_encode = only_encode
        input_size = 50  # :23
        self.positional_encoder = Positional_Encoder(input_size)
        self.mask_encoder = Mask_Encoder(input_size)
        self.encoder = Encoder(input_size, args)
        if not only_encode:
            self.decoder = Decoder(args)

    def forward(self, verts, mask, only_encode=False):
        packed = torch.cat([p.reshape(-1) for p in self.positional_encoder.packed()] +
                           [self.mask_encoder.model[0].weight.reshape(-1)])
        f32 = lambda t: t.to(torch.float32).contiguous()  # noqa: E731
        feats = _ops.PosEncMaskFn.apply(f32(verts), f32(mask), packed, 50, 52)  # pad columns are zero
        latent = self.encoder(feats, self.adj_info)
        if self.only_encode or only_encode:
            return latent
        return self.decoder(latent).permute(0, 2, 1), latent


class Encoder(nn.Module):
    def __init__(self, input_features, args):
        super().__init__()
        self.num_layers = args.num_GCN_layers
        dims = [input_features] + [args.hidden_GCN_size] * self.num_layers
        self.layers = nn.ModuleList([GCN_layer(dims[i], dims[i + 1], args.cut, do_cut=i < self.num_layers - 1)
                                     for i in range(self.num_layers)])
        widths = [args.hidden_GCN_size, 500, 400, 300, args.encoding_size]  # :69
        mlp = []
        for i in range(4):
            mods = [nn.Linear(widths[i], widths[i + 1])]
            if i < 3:
                mods.append(nn.ReLU())
            mlp.append(nn.Sequential(*mods))
        self.mlp = nn.Sequential(*mlp)

    def forward(self, features, adj_info):
        adj = _csr_of(adj_info, "adj")
        for i, layer in enumerate(self.layers):
            features = layer(features, adj, F.relu if i < self.num_layers - 1 else _identity)
        return self.mlp(features.max(dim=1)[0])


def _identity(x):
    return x


class Decoder(nn.Module):
    def __init__(self, args, rank=0):
        super().__init__()
        self.model = FoldingNetDec(rank=rank)
        self.initial = nn.Linear(args.encoding_size, 512)

    def forward(self, features):
        return self.model(self.initial(features))


class _Fold(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.conv1 = nn.Conv1d(in_channels, 512, 1)
        self.conv2 = nn.Conv1d(512, 512, 1)
        self.conv3 = nn.Conv1d(512, 3, 1)
        self.relu = nn.ReLU()

    def forward(self, x):
        return self.conv3(self.relu(self.conv2(self.relu(self.conv1(x)))))


class FoldingNetDecFold1(_Fold):
    def __init__(self):
        super().__init__(514)


class FoldingNetDecFold2(_Fold):
    def __init__(self):
        super().__init__(515)


def GridSamplingLayer(batch_size, meshgrid):
    """(batch, prod(n_i), ndim) float32 lattice; axis order of ``np.meshgrid`` (:160-170)."""
    axes = np.meshgrid(*[np.linspace(lo, hi, num=n) for lo, hi, n in meshgrid])
    grid = np.stack([a.reshape(-1) for a in axes], axis=1).astype(np.float32)
    return np.repeat(grid[None], batch_size, axis=0)


class FoldingNetDec(nn.Module):
    """Two folds of an 80 x 80 lattice in [-0.5, 0.5]^2 conditioned on the 512-d code (:188-208) -> (B, 3, 6400)."""

    def __init__(self, rank=0):
        super().__init__()
        self.rank = rank
        self.fold1 = FoldingNetDecFold1()
        self.fold2 = FoldingNetDecFold2()

    def forward(self, x):
        b = x.size(0)
        code = x.unsqueeze(2).expand(b, 512, 80 * 80)
        grid = torch.from_numpy(GridSamplingLayer(b, [[-0.5, 0.5, 80], [-0.5, 0.5, 80]])).to(x.device)
        folded = self.fold1(torch.cat((code, grid.transpose(2, 1)), dim=1))
        return self.fold2(torch.cat((code, folded), dim=1))
