import sys, os, types, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from infodiffusion_amd.models import InfoDiff, Diff
from infodiffusion_amd.optim import FusedClipAdamW
from infodiffusion_amd.sampling import DiffusionProcess, LatentDiffusionProcess, TwoPhaseDiffusionProcess
dev = torch.device('cuda', 0)
def args_for(dataset, a_dim, size, ch, inch):
    a = types.SimpleNamespace(a_dim=a_dim, batch=32, dtype='bf16')
    m = bench.make_args(a)
    m.dataset = dataset; m.input_size = size; m.unets_channels = ch; m.encoder_channels = ch; m.input_channels = inch
    return m
for name, (ds, ad, size, ch, inch) in {'celeba a_dim=256': ('celeba', 256, 64, 64, 3), 'cifar10 ch=64': ('cifar10', 32, 32, 64, 3),
                                      'fmnist ch=32': ('fmnist', 32, 32, 32, 1)}.items():
    m = args_for(ds, ad, size, ch, inch)
    model = InfoDiff(m, dev, (inch, size, size)).train()
    opt = FusedClipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
    x = torch.rand(32, inch, size, size, device=dev) * 2 - 1
    losses = []
    for i in range(4):
        loss = model.loss_fn(m, x); opt.zero_grad(); loss.backward(); opt.step(); losses.append(float(loss))
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(5):
        loss = model.loss_fn(m, x); opt.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize()
    print('%-18s losses %s  eager %.1f ms/step' % (name, ['%.4f' % l for l in losses], (time.time() - t0) / 5 * 1e3))
# cifar two-phase + latent sampling (config 5)
m = args_for('cifar10', 32, 32, 64, 3); m.diffusion_steps = 20; m.split_step = 10; m.model = 'diff'
model = InfoDiff(m, dev, (3, 32, 32)).eval()
m2 = args_for('cifar10', 32, 32, 64, 3); m2.diffusion_steps = 20; m2.model = 'vanilla'
van = Diff(m2, dev, (3, 32, 32)).eval()
m3 = args_for('cifar10', 32, 32, 64, 3); m3.diffusion_steps = 20; m3.is_latent = True
lat = Diff(m3, dev, (1, 32, 32)).eval()
with torch.no_grad():
    a = LatentDiffusionProcess(m3, lat, dev).sampling(sampling_number=16)
    out = TwoPhaseDiffusionProcess(m, model, van, dev, (3, 32, 32)).sampling(sampling_number=16)
    out2 = DiffusionProcess(m, model, dev, (3, 32, 32)).sampling(sampling_number=16, a=a)
print('latent', tuple(a.shape), 'two-phase', tuple(out.shape), bool(torch.isfinite(out.float()).all()), 'ddim with latent a', tuple(out2.shape), bool(torch.isfinite(out2.float()).all()))
