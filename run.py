#!/usr/bin/env python3
"""CLI with the reference's flags (reference run.py:25-97) driving the MI355X-native hot path.

    python run.py --model diff --mode train --mmd_weight 0.1 --a_dim 32 --epochs 50 --dataset celeba \
        --batch_size 32 --save_epochs 5 --deterministic --prior regular --r_seed 64        # reference run.sh:3
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 run.py ... (data parallel)

Modes: train, eval, eval_fid, save_latent, train_latent_ddim, and the latent-editing callers of the samplers
(interpolate / disentangle / latent_quality), plot_latent (matplotlib scatter) and save_original_img.
Extra flags: --act_dtype {bf16 (default), fp32}, --steps_per_epoch N (synthetic data), --graph {0,1}.
Images are written as .npy (torchvision is not available in this image).
"""
import argparse
import os

import numpy as np
import torch
import torch.distributed as dist

from infodiffusion_amd.data import get_dataset, get_dataset_config
from infodiffusion_amd.dist import GradSync, shard_range
from infodiffusion_amd.optim import FusedClipAdamW
from infodiffusion_amd.trainer import GraphedTrainStep
from infodiffusion_amd.models import VAE, Diff, InfoDiff
from infodiffusion_amd.sampling import DiffusionProcess, LatentDiffusionProcess, TwoPhaseDiffusionProcess
from infodiffusion_amd.utils import (AverageMeter, GradualWarmupScheduler, LatentDataset, ProgressMeter, cos,
                                     generate_exp_string, seed_everything)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--r_seed', type=int, default=0)
    p.add_argument('--img_id', type=int, default=0)
    p.add_argument('--model', required=True, choices=['diff', 'vae', 'vanilla'])
    p.add_argument('--mode', required=True,
                   choices=['train', 'eval', 'eval_fid', 'save_latent', 'disentangle', 'interpolate',
                            'save_original_img', 'latent_quality', 'train_latent_ddim', 'plot_latent'])
    p.add_argument('--prior', required=True, choices=['regular', '10mix', 'roll'])
    p.add_argument('--kld_weight', type=float, default=0)
    p.add_argument('--mmd_weight', type=float, default=0.1)
    p.add_argument('--use_C', action='store_true', default=False)
    p.add_argument('--C_max', type=float, default=25)
    p.add_argument('--dataset', required=True,
                   choices=['fmnist', 'mnist', 'celeba', 'cifar10', 'dsprites', 'chairs', 'ffhq'])
    p.add_argument('--img_folder', default='./imgs')
    p.add_argument('--log_folder', default='./logs')
    p.add_argument('-e', '--epochs', type=int, default=20)
    p.add_argument('--save_epochs', type=int, default=5)
    p.add_argument('--batch_size', type=int, default=64)
    p.add_argument('--learning_rate', type=float, default=0.0001)
    p.add_argument('--optimizer', default='adam', choices=['adam'])
    p.add_argument('--model_folder', default='./models')
    p.add_argument('--deterministic', action='store_true', default=False)
    p.add_argument('--input_channels', type=int, default=1)
    p.add_argument('--unets_channels', type=int, default=64)
    p.add_argument('--encoder_channels', type=int, default=64)
    p.add_argument('--input_size', type=int, default=32)
    p.add_argument('--a_dim', type=int, default=32, required=True)
    p.add_argument('--beta1', type=float, default=1e-5)
    p.add_argument('--betaT', type=float, default=1e-2)
    p.add_argument('--diffusion_steps', type=int, default=1000)
    p.add_argument('--split_step', type=int, default=500)
    p.add_argument('--sampling_number', type=int, default=16)
    p.add_argument('--data_dir', type=str, default='./data')
    p.add_argument('--tb_logger', action='store_true')
    p.add_argument('--is_latent', action='store_true')
    p.add_argument('--is_bottleneck', action='store_true')
    # extras
    p.add_argument('--act_dtype', default='bf16', choices=['fp32', 'bf16'],
                   help='activation / weight-shadow storage of the kernels: bf16 (MFMA bf16, ~9x faster; loss within 1e-2 of '
                        'the reference, epsilon-hat within 2e-2: DESIGN.md section 2) or fp32 (exact-f32 MFMA; within '
                        '1e-4 of the reference)')
    p.add_argument('--steps_per_epoch', type=int, default=100)
    p.add_argument('--graph', type=int, default=1, help='replay the train step from a captured hipGraph')
    return p.parse_args(argv)


def _dist_setup():
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise RuntimeError('run.py needs an MI355X: the HIP kernels have no CPU fallback')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group('nccl', device_id=dev)
    return world, rank, dev


def _model_root(args, latent=False):
    root = args.model_folder
    if args.model == 'vae':
        root = os.path.join(root, 'vae')
    elif args.model == 'vanilla':
        root = os.path.join(root, 'diff')
    root = os.path.join(root, generate_exp_string(args))
    return root + '_latent' if latent else root


def save_model(args, epoch, model, latent=False):
    root = _model_root(args, latent)
    os.makedirs(root, exist_ok=True)
    path = os.path.join(root, 'model-%d.pth' % epoch)
    torch.save(model.state_dict(), path)       # reference format: state_dict only (run.py:145-158)
    print('Saved PyTorch model state to %s' % path)


def _fit(args, model, batches, world, rank, latent=False):
    # reference run.py:177,199-200: AdamW(lr, weight_decay=1e-5) after clip_grad_norm_(1.0) -- here one fused
    # clip+AdamW kernel sequence over a device chunk table, gradients accumulated in the optimizer's arena
    opt = FusedClipAdamW(model.parameters(), lr=args.learning_rate, weight_decay=1e-5, max_norm=1.0)
    cosine = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer=opt, T_max=args.epochs, eta_min=0, last_epoch=-1)
    warm = GradualWarmupScheduler(optimizer=opt, multiplier=2., warm_epoch=1, after_scheduler=cosine)
    sync = GradSync(model, world, arena=opt.arena) if world > 1 else None
    if sync is not None:
        sync.broadcast_parameters()
    losses = AverageMeter('Loss', ':.4f')
    progress = ProgressMeter(args.epochs, [losses], prefix='Epoch ')
    step = GraphedTrainStep(model, args, opt, sync, use_graph=bool(args.graph))
    for epoch in range(args.epochs):
        total, n = torch.zeros((), device=model.device), 0      # accumulated on the device: no per-step host sync
        if hasattr(batches, 'set_epoch'):
            batches.set_epoch(epoch)      # per-epoch permutation / flip masks follow the trainer's epoch counter
        for data in batches:
            x = data[0] if isinstance(data, (tuple, list)) else data
            total += step(x.to(device=model.device), epoch)     # loss_fn, backward, [all-reduce], clip + AdamW
            n += 1
        # the synchronised convs' error word (trainer.GraphedTrainStep.check): a time-out rolls the weights back to the last
        # clean check or raises -- never a check-point, nor an epoch's loss line, from a poisoned step without a trace
        if step.check():
            print('epoch %d: synchronised-conv time-out(s); weights rolled back, training continues without that form'
                  % epoch, file=sys.stderr)
        losses.update(float(total) / max(n - 1, 1))   # reference divides by the last index (run.py:205)
        if rank == 0:
            progress.display(epoch)
        warm.step()
        opt.refresh_lr()        # the captured optimizer step reads the rate from device memory
        losses.reset()
        if (epoch + 1) % args.save_epochs == 0 and rank == 0:
            save_model(args, epoch + 1, model, latent)


_MODELS = {'diff': InfoDiff, 'vanilla': Diff, 'vae': VAE}      # reference run.py:171-176


def train(args):
    world, rank, dev = _dist_setup()
    seed_everything(args.r_seed + rank)
    shape = get_dataset_config(args)
    model = _MODELS[args.model](args, dev, shape)
    model.train()
    _fit(args, model, get_dataset(args, shape, dev, rank, world), world, rank)


def _load(model, path, dev, strict):
    print('Loading model from %s' % path)
    model.load_state_dict(torch.load(path, map_location=dev), strict=strict)


def _pick_batch(args, shape, dev, rank, world, index):
    for i, data in enumerate(get_dataset(args, shape, dev, rank, world)):
        if i == index:
            break
    return data[0].to(dev)


def _latent_edit(args, model, dev, shape, out_root, rank, world):
    """The callers that invert an image and re-generate it under an edited latent (reference run.py:310-337
    latent_quality, 366-414 disentangle, 444-481 interpolate): encoder -> reverse_sampling (DDIM inversion,
    the encoder is re-run on x_t every step as in sampling.py:84) -> sampling(xT=..., a=...)."""
    vae = args.model == 'vae'
    proc = None if vae else DiffusionProcess(args, model, dev, shape)
    data = _pick_batch(args, shape, dev, rank, world, 10 if args.mode == 'latent_quality' else args.img_id)
    with torch.no_grad():
        a, _, mu, log_var = model.encoder(data)
    if args.kld_weight != 0:
        a = mu + torch.exp(0.5 * log_var) if args.mode == 'latent_quality' else mu
    a = a.float()
    out_dir = os.path.join(out_root, args.mode)
    os.makedirs(out_dir, exist_ok=True)

    def emit(sample, k=0):
        np.save(os.path.join(out_dir, 'sample%05d.npy' % k), sample.float().cpu().numpy())

    if args.mode == 'latent_quality':
        # same latent, fresh x_T draws: how much of the image the latent alone pins down
        xT = proc.reverse_sampling(data, a)
        n = args.sampling_number
        batch = proc.sampling(xT=torch.randn_like(xT.repeat(n, 1, 1, 1)), a=a.repeat(n, 1))
        emit((torch.clip(batch.float(), min=-1, max=1) + 1) / 2)
    elif args.mode == 'disentangle':
        eta = [-1.5, -1.2, -0.9, -0.6, -0.3, 0.0, 0.3, 0.6, 0.9, 1.2, 1.5]
        xT = None if vae else proc.reverse_sampling(data, a).repeat(len(eta), 1, 1, 1)
        for k in range(args.a_dim):
            rows = a[:1].repeat(len(eta), 1)
            rows[:, k] = torch.tensor(eta, device=dev)         # traverse one latent coordinate
            with torch.no_grad():
                emit(model.decoder(rows) if vae else proc.sampling(xT=xT, a=rows), k)
    else:
        eta = [0.0, 0.11, 0.22, 0.33, 0.44, 0.55, 0.66, 0.77, 0.88, 1.0]
        intp_a = torch.stack([np.cos(e * np.pi / 2) * a[0] + np.sin(e * np.pi / 2) * a[1] for e in eta])
        if vae:
            with torch.no_grad():
                emit(model.decoder(intp_a))
            return
        xT = proc.reverse_sampling(data, a).float().contiguous()
        theta = torch.arccos(cos(xT[0], xT[1]))     # spherical interpolation of the two inverted x_T
        intp_x = torch.stack([(torch.sin((1 - e) * theta) * xT[0] + torch.sin(e * theta) * xT[1]) / torch.sin(theta)
                              for e in eta])
        emit(proc.sampling(xT=intp_x, a=intp_a))


def evaluate(args):
    world, rank, dev = _dist_setup()
    if args.mode == 'train_latent_ddim':
        seed_everything(args.r_seed + rank)
        ds = LatentDataset('%s_%s_latent.npz' % (args.model, generate_exp_string(args).replace('.', '_')))
        loader = torch.utils.data.DataLoader(ds, batch_size=args.batch_size, shuffle=True)
        model = Diff(args, dev, (1, args.a_dim, args.a_dim))
        model.train()
        _fit(args, model, loader, world, rank, latent=True)
        return
    seed_everything(args.r_seed + rank)
    shape = get_dataset_config(args)
    model = _MODELS[args.model](args, dev, shape)
    _load(model, os.path.join(_model_root(args), 'model-%d.pth' % args.epochs), dev, strict=False)
    model.eval()
    out_root = os.path.join(args.img_folder, 'vae' if args.model == 'vae' else '', generate_exp_string(args))
    if args.model == 'vae' and args.mode in ('eval', 'eval_fid'):
        # reference run.py:261-263, 297-300: images are decoder(randn) -- no diffusion process
        sub = 'eval' if args.mode == 'eval' else ('eval-fid-latent' if args.is_latent else 'eval-fid-fast')
        os.makedirs(os.path.join(out_root, sub), exist_ok=True)
        lo, hi = shard_range(args.sampling_number, rank, world)
        step = args.sampling_number if args.mode == 'eval' else args.batch_size
        with torch.no_grad():
            for n in range(lo, hi, step):
                a = torch.randn([min(step, hi - n), args.a_dim]).to(device=dev)
                img = model.decoder(a).float()
                if args.mode == 'eval_fid':
                    img = (torch.clip(img, min=-1, max=1) + 1) / 2
                np.save(os.path.join(out_root, sub, 'sample-%06d.npy' % n), img.cpu().numpy())
        print('DONE')
    elif args.mode == 'eval':
        proc = DiffusionProcess(args, model, dev, shape)
        os.makedirs(os.path.join(out_root, 'eval'), exist_ok=True)
        for n in range(0, args.sampling_number, args.batch_size):
            sample = proc.sampling(sampling_number=16)     # reference hard-codes 16 (run.py:259)
            np.save(os.path.join(out_root, 'eval', 'sample%05d.npy' % n), sample.float().cpu().numpy())
    elif args.mode == 'eval_fid':
        sub = 'eval-fid-latent' if args.is_latent else 'eval-fid-fast'
        os.makedirs(os.path.join(out_root, sub), exist_ok=True)
        proc = DiffusionProcess(args, model, dev, shape)
        if args.is_latent:
            model2 = Diff(args, dev, (1, args.a_dim, args.a_dim))
            _load(model2, os.path.join(_model_root(args, latent=True), 'model-%d.pth' % args.epochs), dev, True)
            model2.eval()
            proc_latent = LatentDiffusionProcess(args, model2, dev)
        else:
            model2 = Diff(args, dev, shape)
            _load(model2, './models/diff/%s_%dd/model-%d.pth' % (args.dataset, args.a_dim, args.epochs), dev, True)
            model2.eval()
            proc = TwoPhaseDiffusionProcess(args, model, model2, dev, shape)
        lo, hi = shard_range(args.sampling_number, rank, world)     # shard the images, no collective
        for n in range(lo, hi, args.batch_size):
            bs = min(args.batch_size, hi - n)
            if args.is_latent:
                batch = proc.sampling(sampling_number=bs, a=proc_latent.sampling(sampling_number=bs))
            else:
                batch = proc.sampling(sampling_number=bs)
            img = (torch.clip(batch.float(), min=-1, max=1) + 1) / 2
            np.save(os.path.join(out_root, sub, 'sample-%06d.npy' % n), img.cpu().numpy())
        print('DONE')
    elif args.mode == 'save_latent':
        # one file for the whole dataset (reference run.py:415-443): rank 0 encodes all of it, unsharded
        if rank == 0:
            all_a, all_attr = [], []
            for data in get_dataset(args, shape, dev, 0, 1):
                with torch.no_grad():
                    a, _, mu, _ = model.encoder(data[0].to(dev))
                all_a.append((mu if args.kld_weight != 0 else a).cpu().numpy())
                all_attr.append(np.asarray(data[1]))
            all_a = np.concatenate(all_a)
            np.savez('%s_%s_latent' % (args.model, generate_exp_string(args).replace('.', '_')),
                     all_a=all_a, all_attr=np.concatenate(all_attr).reshape(len(all_a), -1).squeeze(-1))
    elif args.mode in ('interpolate', 'disentangle', 'latent_quality'):
        _latent_edit(args, model, dev, shape, out_root, rank, world)
    elif args.mode == 'plot_latent':
        # reference run.py:338-365: scatter of the first two latent coordinates, coloured by the batch's labels
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
        if rank != 0:
            return
        all_a, all_attr = [], []
        for data in get_dataset(args, shape, dev, 0, 1):
            with torch.no_grad():
                a, _, mu, _ = model.encoder(data[0].to(dev))
            use_mu = args.kld_weight != 0 and args.mmd_weight == 0
            all_a.append((mu if use_mu else a).float().cpu().numpy())
            all_attr.append(np.asarray(data[1]))
        all_a, all_attr = np.concatenate(all_a), np.concatenate(all_attr)
        plt.scatter(all_a[:, 0], all_a[:, 1], c=all_attr, cmap='tab10', s=5)
        os.makedirs(out_root, exist_ok=True)
        plt.savefig(os.path.join(out_root, 'latent.png'))
    else:
        raise NotImplementedError('mode %s' % args.mode)


if __name__ == '__main__':
    args = parse_args()
    if args.mode == 'save_original_img':
        # reference run.py:540-549: the dataset's images in [0, 1] (here as .npy batches; no model, no GPU kernels)
        out = './%s_imgs/' % args.dataset
        os.makedirs(out, exist_ok=True)
        for i, data in enumerate(get_dataset(args, get_dataset_config(args), 'cpu')):
            np.save(os.path.join(out, '%06d.npy' % i), ((data[0] + 1) / 2).numpy())
    elif args.mode == 'train':
        train(args)
    else:
        if args.mode in ('disentangle', 'latent_quality'):
            args.batch_size = 1
        elif args.mode == 'interpolate':
            args.batch_size = 2
        evaluate(args)
    if dist.is_initialized():
        dist.destroy_process_group()
