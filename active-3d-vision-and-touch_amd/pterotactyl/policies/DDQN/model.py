"""``pterotactyl.policies.DDQN.model`` on MI355X — a consumer of the hot-path kernels.

``Graph_Model`` (reference ``policies/DDQN/model.py:65-128``): Q-network over the predicted mesh — per-vertex features
= [action embedding | positional embedding | mask embedding] (3 x 100), ``args.layers`` graph layers
(300 -> hidden_dim ... -> num_actions; the last aggregates every channel, no ReLU), max over vertices.  The graph
layers run as ``a3vt_gcn_layer_fwd/bwd`` calls through ``vision.model.GCN_layer``; the adjacency is the dense
``adj['adj']`` the reference keeps (:68), converted to CSR once.  ``Latent_Model`` (:15-61) is a plain MLP over
auto-encoder latents and stays on torch ops.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...reconstruction.vision.model import GCN_layer, Mask_Encoder, Positional_Encoder, _csr_of  # noqa: F401
from ...utility import utils


def _action_embedding(num_in, out):
    return nn.Sequential(nn.Sequential(nn.Linear(num_in, 200), nn.ReLU()),
                         nn.Sequential(nn.Linear(200, 100), nn.ReLU()),
                         nn.Sequential(nn.Linear(100, out)))


def _dev(t, device):
    return t.float().to(device)


class Latent_Model(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        latent_size = utils.load_model_config(args.auto_location)[0].encoding_size
        self.action_model = _action_embedding(args.num_actions, latent_size)
        sizes = [latent_size * 3] + [args.hidden_dim] * (args.layers - 1) + [args.num_actions]
        layers = []
        for i in range(args.layers):
            mods = [nn.Linear(sizes[i], sizes[i + 1])]
            if i < args.layers - 1:
                mods.append(nn.ReLU())
            layers.append(nn.Sequential(*mods))
        self.model = nn.Sequential(*layers)

    def forward(self, obs, next=False):
        dev = self.model[0][0].weight.device
        sfx = "_n" if next else ""
        action_input = self.action_model(_dev(obs["mask" + sfx], dev))
        full = torch.cat((action_input, _dev(obs["latent" + sfx], dev), _dev(obs["first_latent"], dev)), dim=-1)
        return self.model(full)


class Graph_Model(nn.Module):
    def __init__(self, args, adj):
        super().__init__()
        self.args = args
        self.num_layers = args.layers
        input_size = 100
        self.adj = adj["adj"]      # dense (N,N), as the reference keeps it (:68)
        self._adj_info = adj       # the CSR handle is taken from / cached in this dict
        self.action_model = _action_embedding(50, input_size)  # :75 (50 = number of candidate actions)
        self.positional_embedding = Positional_Encoder(input_size)
        self.mask_embedding = Mask_Encoder(input_size)
        sizes = [input_size * 3] + [args.hidden_dim] * (args.layers - 1) + [args.num_actions]
        self.layers = nn.ModuleList([GCN_layer(sizes[i], sizes[i + 1], cut=args.cut, do_cut=i != self.num_layers - 1)
                                     for i in range(args.layers)])

    def forward(self, obs, next=False):
        dev = self.layers[0].weight.device
        sfx = "_n" if next else ""
        action = self.action_model(_dev(obs["mask" + sfx], dev))
        mesh_all = _dev(obs["mesh" + sfx], dev)
        mesh, mask = mesh_all[:, :, :3], mesh_all[:, :, 3:]
        feats = torch.cat((action.unsqueeze(1).expand(-1, mesh.shape[1], -1), self.positional_embedding(mesh),
                           self.mask_embedding(mask)), dim=-1)
        adj = _csr_of(self._adj_info, "adj")
        x = feats
        for i, layer in enumerate(self.layers):   # layer 0 always takes the ReLU, also when it is the only layer (:118)
            x = layer(x, adj, F.relu if i == 0 or i != self.num_layers - 1 else _identity)
        return torch.max(x, dim=1)[0]


def _identity(x):
    return x
