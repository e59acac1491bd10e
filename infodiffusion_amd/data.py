"""Dataset configuration (reference data.py:63-102, reproduced verbatim in behaviour: it
mutates `args` and returns `shape`) and the batch sources available in this image.
torchvision is not installed here, so the real loaders (data.py:105-243) are replaced by
random-pixel batches in the reference's normalisation range ([-1, 1], data.py:170-171) or
by a user-supplied `.npy` array of images under `--data_dir`."""
import os

import numpy as np
import torch

_CFG = {   # dataset: (input_channels, unets/encoder channels, input_size)
    'fmnist': (1, 32, 32), 'mnist': (1, 32, 32), 'dsprites': (1, 32, 32), 'celeba': (3, 64, 64),
    'cifar10': (3, 64, 32), 'chairs': (3, 32, 64), 'ffhq': (3, 64, 64),
}


def get_dataset_config(args):
    c, ch, size = _CFG[args.dataset]
    args.input_channels = c
    args.unets_channels = ch
    args.encoder_channels = ch
    args.input_size = size
    return (args.input_channels, args.input_size, args.input_size)


class _Batches:
    """Iterable of `(images,)` tuples (the reference loops unpack `data[0]`, run.py:191-193)."""

    def __init__(self, args, shape, n_batches, device, rank=0, world=1):
        self.shape, self.n, self.bs, self.device = shape, n_batches, args.batch_size, device
        self.rank, self.world = rank, world
        self.array = None
        path = os.path.join(getattr(args, 'data_dir', './data'), '%s.npy' % args.dataset)
        if os.path.exists(path):
            arr = np.load(path, mmap_mode='r')       # uint8 NHWC or float NCHW in [0, 1]
            self.array = arr
            self.n = len(arr) // (self.bs * world)
        self.seed = getattr(args, 'r_seed', 0)
        self.epoch = 0
        # which loaders the reference shuffles: cifar10, dsprites, chairs always (data.py:197, 213, 230); CelebA only in the
        # three-loader modes (data.py:175-180; shuffle=False otherwise, :184); mnist / fmnist / ffhq never (data.py:130, 144:
        # DataLoader's default; :243)
        mode = getattr(args, 'mode', 'train')
        self.shuffle = args.dataset in ('cifar10', 'dsprites', 'chairs') or (
            args.dataset == 'celeba' and mode in ('attr_classification', 'eval_fid', 'reconstruction'))
        # RandomHorizontalFlip is part of the fmnist (data.py:138), CelebA (:165-166: `do_augment` defaults to True and no caller
        # clears it), cifar10 (:191), chairs (:224) and FFHQ (:237) transforms in every mode; mnist and dsprites have none
        self.augment = args.dataset in ('celeba', 'ffhq', 'fmnist', 'cifar10', 'chairs')

    def __len__(self):
        return self.n

    def set_epoch(self, epoch):
        """The epoch the next iteration draws for (the trainer's own counter: a resumed run continues the sequence of
        permutations / flip masks instead of replaying epoch 0's)."""
        self.epoch = int(epoch)

    def _seed(self, epoch, rank=None):
        """Collision-free over (seed, epoch, rank): rank < world keeps its own residue class."""
        s = (self.seed * 1000003 + epoch) * (self.world + 1)
        return (s + (self.world if rank is None else rank)) & ((1 << 62) - 1)

    def __iter__(self):
        # fresh draws every epoch (flip masks, random pixels), different per rank; ONE permutation per epoch shared by
        # all ranks for the datasets the reference shuffles
        epoch, self.epoch = self.epoch, self.epoch + 1
        g = torch.Generator(device='cpu')
        g.manual_seed(self._seed(epoch, self.rank))
        on_gpu = torch.device(self.device).type == 'cuda'
        if on_gpu and self.array is None:
            gd = torch.Generator(device=self.device)          # random pixels are drawn where they are consumed:
            gd.manual_seed(self._seed(epoch, self.rank))      # a host draw + H2D copy costs more than a step
        perm = None
        if self.array is not None and self.shuffle:
            gp = torch.Generator(device='cpu')
            gp.manual_seed(self._seed(epoch))                # shared by all ranks
            perm = torch.randperm(len(self.array), generator=gp).numpy()
        for i in range(self.n):
            if self.array is None and on_gpu:
                x = torch.rand(self.bs, *self.shape, generator=gd, device=self.device) * 2 - 1
            elif self.array is None:
                x = torch.rand(self.bs, *self.shape, generator=g) * 2 - 1
            else:
                lo = (i * self.world + self.rank) * self.bs
                rows = self.array[lo:lo + self.bs] if perm is None else self.array[np.sort(perm[lo:lo + self.bs])]
                a = torch.from_numpy(np.ascontiguousarray(rows))
                flip = (torch.rand(self.bs, generator=g) < 0.5) if self.augment else None      # one draw per image, every path
                if a.dtype == torch.uint8 and torch.device(self.device).type == 'cuda':
                    # bytes cross PCIe; ToTensor / RandomHorizontalFlip / Normalize run on the GPU (idf_prep_u8)
                    from . import ops
                    flip = flip.to(torch.uint8) if flip is not None else None
                    x = ops.prep_u8(a.pin_memory().to(self.device, non_blocking=True), flip)
                    yield (x, torch.zeros(self.bs, dtype=torch.long))
                    continue
                if a.dtype == torch.uint8:
                    a = a.permute(0, 3, 1, 2).float() / 255.0
                x = (a.float() - 0.5) / 0.5
                if flip is not None:             # float arrays / CPU batches: the same transform, NCHW (flip = reverse the columns)
                    x = torch.where(flip[:, None, None, None], x.flip(3), x)
            yield (x, torch.zeros(self.bs, dtype=torch.long))


def get_dataset(args, shape=None, device='cpu', rank=0, world=1):
    shape = shape or (args.input_channels, args.input_size, args.input_size)
    return _Batches(args, shape, getattr(args, 'steps_per_epoch', 100), device, rank, world)
