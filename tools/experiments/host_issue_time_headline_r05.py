import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from a3vt_amd import mesh as amesh, distributed as adist
from a3vt_amd.pterotactyl.reconstruction.vision import model
from a3vt_amd.pterotactyl.utility import utils
from a3vt_amd.synthetic import gt_cloud, make_args
dev = torch.device("cuda", 0)
args = make_args(number_points=10000)
v, f = amesh.icosphere(4)
vt, ft = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
info = utils.adj_init(vt, ft, args)
torch.manual_seed(0)
net = model.Deformation(info, vt, args).to(dev)
params = list(net.parameters()); bucket = adist.FlatGradBucket(params); opt = torch.optim.Adam(params, lr=3e-4, fused=True)
img = torch.zeros(64, 1, device=dev); charts = model.prepare_mesh({"img": img}, vt, args); gt = gt_cloud(64, 10000, 0).to(dev)
def step():
    bucket.zero(); out = net(img, charts)[0]
    loss = 9000.0 * utils.chamfer_distance(out, info["faces_i32"], gt, num=10000).mean()
    loss.backward(); bucket.all_reduce_mean(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
cpu = []
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); cpu.append(1e3 * (time.perf_counter() - t0)); torch.cuda.synchronize()
print(f"headline step: host time to issue {sorted(cpu)[3]:.2f} ms (device ~52 ms)")
