import sys, os, types, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from infodiffusion_amd.models import InfoDiff
a = types.SimpleNamespace(a_dim=32, batch=32, dtype='bf16')
margs = bench.make_args(a)
margs.diffusion_steps = 100; margs.deterministic = True
dev = torch.device('cuda', 0)
m = InfoDiff(margs, dev, (3, 64, 64)).eval()
x = torch.randn(256, 3, 64, 64, device=dev); av = torch.randn(256, 32, device=dev)
with torch.no_grad():
    for i in range(3): m(x, 50, av)
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(10): m(x, 50 - i, av)
    torch.cuda.synchronize(); print('eval ms %.2f' % ((time.time() - t0) / 10 * 1e3))
