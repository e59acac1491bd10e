"""Host-side utilities the reference's run.py imports from utils.py (utils.py:11-172):
priors, experiment string, seeding, meters, the warm-up LR scheduler, the latent
dataset -- plus `compute_mmd`, which runs on the HIP kernels."""
import math
import random

import numpy as np
import torch
from torch.optim.lr_scheduler import _LRScheduler
from torch.utils.data import Dataset

from . import ops


def compute_mmd(x, y):
    """utils.py:85-90: mean K(x,x) + mean K(y,y) - 2 mean K(x,y), K = exp(-|x-y|^2/dim^2)."""
    return ops.mmd(x, y)


def gaussian_mixture(batch_size, n_dim=2, n_labels=10, x_var=0.5, y_var=0.1, label_indices=None):
    """utils.py:11-37: ring of `n_labels` 2-D Gaussians per coordinate pair (host numpy)."""
    if n_dim % 2 != 0:
        raise Exception("n_dim must be a multiple of 2.")
    x = np.random.normal(0, x_var, (batch_size, n_dim // 2))
    y = np.random.normal(0, y_var, (batch_size, n_dim // 2))
    z = np.empty((batch_size, n_dim), dtype=np.float32)
    for b in range(batch_size):
        for zi in range(n_dim // 2):
            label = label_indices[b] if label_indices is not None else np.random.randint(0, n_labels)
            if label >= n_labels:
                label = np.random.randint(0, n_labels)
            r = 2.0 * np.pi / float(n_labels) * float(label)
            cx, sx = math.cos(r), math.sin(r)
            z[b, zi * 2] = x[b, zi] * cx - y[b, zi] * sx + 1.4 * cx
            z[b, zi * 2 + 1] = x[b, zi] * sx + y[b, zi] * cx + 1.4 * sx
    return z


def swiss_roll(batch_size, noise=0.5):
    """utils.py:39-40."""
    from sklearn.datasets import make_swiss_roll
    return make_swiss_roll(n_samples=batch_size, noise=noise)[0][:, [0, 2]] / 5.


def cos(a, b):
    a = torch.nn.functional.normalize(a.view(-1), dim=0)
    b = torch.nn.functional.normalize(b.view(-1), dim=0)
    return (a * b).sum()


def generate_exp_string(args) -> str:
    """utils.py:49-61 -- defines the checkpoint / image directory names."""
    parts = ['%s_%sd' % (args.dataset, args.a_dim)]
    if args.kld_weight != 0:
        parts.append('%skld' % args.kld_weight)
        if args.use_C:
            parts.append('%sC' % args.C_max)
    if args.mmd_weight != 0:
        parts.append('%smmd' % args.mmd_weight)
    if args.prior != 'regular':
        parts.append('%s' % args.prior)
    if args.is_bottleneck:
        parts.append('bottleneck')
    return '_'.join(parts)


def seed_everything(r_seed):
    print("Set seed: ", r_seed)
    random.seed(r_seed)
    np.random.seed(r_seed)
    torch.manual_seed(r_seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(r_seed)
    # utils.py:71 sets torch.backends.cudnn.deterministic = True: the counterpart here is the ordered form of every accumulation that
    # otherwise uses fp32 atomics (weight gradients, GroupNorm parameter gradients) -- two runs of a step give the same bits
    ops.set_deterministic(True)


class AverageMeter:
    """Running value / mean tracker with the reference's printing format (utils.py:93-113)."""

    def __init__(self, name, fmt=':f'):
        self.name, self.fmt = name, fmt
        self.val = self.avg = self.sum = self.count = 0

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val, self.sum, self.count = val, self.sum + val * n, self.count + n
        self.avg = self.sum / self.count

    def __str__(self):
        spec = self.fmt
        return ('%s {v%s} ({a%s})' % (self.name, spec, spec)).format(v=self.val, a=self.avg)


class ProgressMeter:
    """`Epoch [ 3/50]<tab>Loss 0.1234 (0.1234)` lines, carriage-returned (utils.py:116-130)."""

    def __init__(self, num_batches, meters, prefix=""):
        width = len(str(int(num_batches)))
        self.template = '[{:%dd}/%s]' % (width, str(int(num_batches)).rjust(width))
        self.meters, self.prefix = list(meters), prefix

    def display(self, batch):
        cells = [self.prefix + self.template.format(batch)] + [str(m) for m in self.meters]
        print('\r' + '\t'.join(cells), end='')


class GradualWarmupScheduler(_LRScheduler):
    """utils.py:133-160: the rate climbs linearly from lr to multiplier*lr over `warm_epoch` epochs, then
    `after_scheduler` takes over with multiplier*lr as its base (stepped per epoch, run.py:209).  Same attribute
    names and the same float expressions as the reference, so the rate sequence is bit-identical
    (tests/test_cli_host.py pins it, the overshoot at the hand-over epoch included)."""

    def __init__(self, optimizer, multiplier, warm_epoch, after_scheduler=None):
        self.multiplier, self.total_epoch = multiplier, warm_epoch
        self.after_scheduler, self.finished = after_scheduler, False
        self.last_epoch = self.base_lrs = None          # filled in by the base class
        super().__init__(optimizer)

    def _scaled(self, factor):
        return [lr * factor for lr in self.base_lrs]

    def get_lr(self):
        if self.last_epoch <= self.total_epoch:          # still warming up
            return self._scaled((self.multiplier - 1.) * self.last_epoch / self.total_epoch + 1.)
        if not self.after_scheduler:
            return self._scaled(self.multiplier)
        if not self.finished:                            # hand-over: the follower starts from the warmed-up rate
            self.after_scheduler.base_lrs = self._scaled(self.multiplier)
            self.finished = True
        return self.after_scheduler.get_lr()

    def step(self, epoch=None, metrics=None):
        if not (self.finished and self.after_scheduler):
            return super().step(epoch)
        self.after_scheduler.step(epoch if epoch is None else epoch - self.total_epoch)


class LatentDataset(Dataset):
    """The `all_a` latents of a `save_latent` archive (.npz wire format between the two phases,
    utils.py:163-172)."""

    def __init__(self, data_path):
        with np.load(data_path) as archive:
            self.x = torch.as_tensor(archive['all_a'], dtype=torch.float32)

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, index):
        return self.x[index]
