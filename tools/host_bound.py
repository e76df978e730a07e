import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from a3vt_amd.synthetic import named_config, NamedStep
dev = torch.device("cuda", 0)
for which in (3, 4):
    cfg = named_config(which, dev, "bf16s")
    step = NamedStep(cfg)
    for _ in range(8): step()
    torch.cuda.synchronize()
    cpu, tot = [], []
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        cpu.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    free = (time.perf_counter() - t0) * 100
    print(f"configs[{which}]: host time to issue a step {sorted(cpu)[4]:.1f} ms, step with a sync after it {sorted(tot)[4]:.1f} ms, free-running {free:.1f} ms/step")
    step.close(); del cfg, step; torch.cuda.empty_cache()
