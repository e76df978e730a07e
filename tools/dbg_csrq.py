import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import make_args, oracle_adj, rel_err, template
from a3vt_amd import mesh as amesh, ops
from oracle import gcn as og
cuda = torch.device('cuda', 0)
def run(adj, st, feats, gup, L, H, cut_len):
    ws = [st[f"mesh_deform_1.layers.{i}.weight"].to(cuda).requires_grad_(True) for i in range(L)]
    bs = [st[f"mesh_deform_1.layers.{i}.bias"].to(cuda).requires_grad_(True) for i in range(L)]
    fd = torch.nn.functional.pad(feats, (0, 2)).to(cuda).requires_grad_(True)
    out = ops.gcn_stack(fd, adj, 50, H, cut_len, ws, bs)
    (out * gup.to(cuda)).sum().backward()
    torch.cuda.synchronize()
    return out.detach(), fd.grad, [w.grad for w in ws], [b.grad for b in bs]
for tname, L, B in [("ico3", 2, 24), ("ico3", 4, 24), ("ico4", 3, 6)]:
    H = 300
    args = make_args(num_GCN_layers=L, hidden_GCN_size=H)
    verts, faces = template(tname)
    adj_o, _ = oracle_adj(verts, faces, args)
    n = verts.shape[0]
    st = og.init_state(50, H, L, seed=5)
    g = torch.Generator().manual_seed(21)
    feats = torch.randn(B, n, 50, generator=g) * 0.5
    gup = torch.randn(B, n, 3, generator=g)
    r, c = amesh.vision_pairs(faces, n)
    adj = ops.DeviceCSR(amesh.CSRAdjacency.from_pairs(r, c, n), cuda)
    ops.dbg_csr_algo("rows")
    ref = run(adj, st, feats, gup, L, H, 99)
    ops.dbg_csr_algo("auto")
    new = run(adj, st, feats, gup, L, H, 99)
    st64 = {k: v.double() for k, v in st.items()}
    out_o = og.gcn(feats.double(), st64, "mesh_deform_1", (adj_o[0], adj_o[1], adj_o[2].double()), L, 0.33)
    d = (new[0] - ref[0]).abs()
    print(tname, L, B, "out: max diff", d.max().item(), "n diff", (d > 0).sum().item(), "of", d.numel(),
          "| vs oracle new", rel_err(new[0], out_o), "ref", rel_err(ref[0], out_o))
    if (d > 0).any():
        idx = (d > 0).nonzero()[:8]
        print(" first diffs (b, v, ch):", idx.tolist())
    print("  gfeats diff", (new[1] - ref[1]).abs().max().item(), " dW diffs", [(a - b).abs().max().item() for a, b in zip(new[2], ref[2])],
          " db rel", [rel_err(a, b) for a, b in zip(new[3], ref[3])])
