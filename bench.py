#!/usr/bin/env python3
"""Headline benchmark: InfoDiffusion CelebA 64x64 training throughput (images/s) on
N MI355X, one process per GPU, data-parallel over RCCL.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = `InfoDiff.loss_fn` (q_sample + encoder + AdaGN UNet + eps-MSE + recon + MMD)
forward, backward, gradient all-reduce (N > 1), clip_grad_norm_(1.0), AdamW --
BASELINE.json configs[1]: CelebA 3x64x64, a_dim 32, mmd_weight 0.1, B = 32 per GPU,
T = 1000, dropout 0.1, bf16 activations.  Random-pixel batches resident in HBM.
Prints ONE JSON line (rank 0).  Extra keys: `roofline` (dominant kernel, HIP-event
timed), `cpu_baseline` (the CPU oracle on the host cores, rank 0 at N = 1),
`sampling` (DDIM-100 images/s, N = 1 only, --sampling).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch (reference run.sh: 32)')
    ap.add_argument('--a_dim', type=int, default=32)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--graph', type=int, default=1, help='replay the step from a captured hipGraph')
    ap.add_argument('--fused-opt', type=int, default=1, help='fused clip+AdamW kernel (0: clip_grad_norm_ + torch AdamW)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--sampling', action='store_true', help='(default now) also time DDIM-100 sampling (B=256)')
    ap.add_argument('--no-sampling', action='store_true', help='skip the DDIM-100 sampling measurement')
    ap.add_argument('--sampling-batch', type=int, default=256)
    ap.add_argument('--no-large-batch', action='store_true', help='skip the supplementary B=128 training rate')
    ap.add_argument('--cpu-baseline-worker', default=None, help=argparse.SUPPRESS)
    return ap.parse_args()


def make_args(a):
    from types import SimpleNamespace
    return SimpleNamespace(beta1=1e-5, betaT=1e-2, diffusion_steps=1000, input_size=64, input_channels=3,
                           is_bottleneck=False, unets_channels=64, encoder_channels=64, a_dim=a.a_dim,
                           mmd_weight=0.1, kld_weight=0.0, prior='regular', batch_size=a.batch, use_C=False,
                           C_max=25.0, epochs=50, deterministic=True, model='diff', split_step=500, mode='train',
                           is_latent=False, act_dtype=a.dtype, dataset='celeba')


CPU_THREADS = 16      # the oracle's convs stop scaling (and oversubscribe badly) far below the box's 256 cores
CPU_BATCH = 8


def cpu_baseline_worker(a_dim, out_path):
    """Child process: the CPU oracle (restatement of the reference's stock-ATen path, fp32 NCHW) doing
    the same training step on host cores; appends each step's seconds to `out_path` as it goes."""
    from oracle import infodiff_oracle as O
    torch.set_num_threads(CPU_THREADS)
    cfg = O.dataset_cfg('celeba', a_dim=a_dim, mmd_weight=0.1)
    down, mid, up, _ = O.unet_layout(64, [1, 2, 2, 2])
    from types import SimpleNamespace
    from infodiffusion_amd.models import InfoDiff
    margs = SimpleNamespace(**{**cfg.__dict__})
    with torch.no_grad():
        shapes = [(k, tuple(v.shape)) for k, v in InfoDiff(margs, 'cpu', cfg.shape).state_dict().items()]
    sd = O.synth_state_dict(shapes)
    params = []
    for k, v in sd.items():
        if not k.endswith('timembedding.0.weight'):
            v.requires_grad_(True)
            params.append(v)
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-5)
    sched = O.noise_schedule(cfg.beta1, cfg.betaT, cfg.diffusion_steps)
    g = torch.Generator(device='cpu')
    g.manual_seed(64)
    B = CPU_BATCH
    for it in range(3):
        x = torch.rand(B, 3, 64, 64, generator=g) * 2 - 1
        t0 = time.time()
        idx = torch.randint(0, 1000, (B,))
        eps = torch.randn_like(x)
        loss, _ = O.infodiff_loss(sd, cfg, x, idx, eps, sched, prior=torch.randn(B, cfg.a_dim), drop=O.Drop('torch'))
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 1.0)
        opt.step()
        with open(out_path, 'a') as f:
            f.write('%.4f\n' % (time.time() - t0))


def cpu_baseline(margs, budget_s=150):
    """Run the worker as a child process (bounded: killed by PID after `budget_s`) and report
    images/s from the timed steps it completed (first step = warm-up, excluded when more exist)."""
    import subprocess
    import tempfile
    out = tempfile.NamedTemporaryFile(prefix='idf_cpu_', suffix='.txt', delete=False).name
    env = dict(os.environ, CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='', OMP_NUM_THREADS=str(CPU_THREADS))
    proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', out,
                             '--a_dim', str(margs.a_dim)], env=env, stdout=subprocess.DEVNULL,
                            stderr=subprocess.DEVNULL)
    try:
        proc.wait(timeout=budget_s)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.wait()
    try:
        times = [float(l) for l in open(out).read().split()]
    except OSError:
        times = []
    finally:
        if os.path.exists(out):
            os.unlink(out)
    if not times:
        return {'value': None, 'unit': 'images/s', 'cores': CPU_THREADS, 'kind': 'port',
                'sample': 'CPU oracle did not finish one B=%d step within %d s' % (CPU_BATCH, budget_s)}
    timed = times[1:] if len(times) > 1 else times
    t = sum(timed) / len(timed)
    return {'value': round(CPU_BATCH / t, 3), 'unit': 'images/s', 'cores': CPU_THREADS, 'kind': 'port',
            'sample': 'CPU oracle (fp32 NCHW stock-ATen restatement of the reference), CelebA 64x64 train step '
                      '(fwd+bwd+clip+AdamW, dropout on) at B=%d on %d threads (host has %d cores): %d timed '
                      'step(s) after 1 warm-up, %.1f s/step' % (CPU_BATCH, CPU_THREADS, os.cpu_count() or 0,
                                                                 len(timed), t)}


def pmc_traffic(prefix):
    """Average HBM bytes per launch of the kernels named `prefix*`, from the committed rocprofv3 PMC
    passes (profiles/r01_pmc_traffic.json: FETCH_SIZE x2 + WRITE_SIZE), or None."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')) as f:
            d = json.load(f)
        ks = [v for k, v in d.items() if k.startswith(prefix)]
        n = sum(v['launches'] for v in ks)
        return round(sum(v['launches'] * v['hbm_bytes_avg'] for v in ks) / n) if n else None
    except (OSError, ValueError, KeyError):
        return None


class ConvTimer:
    """HIP-event timing of the dominant kernel (the halo 3x3 conv: forward + data-gradient launches).
    One eager step records every launch's arguments; each distinct launch configuration is then replayed
    n times back to back from a captured hipGraph -- exactly how the timed region issues it -- between one
    HIP event pair on that stream, and the per-launch durations are weighted by the step's launch counts.
    (An event pair around every single eager launch measures the events: +10 us per launch.)"""

    def __init__(self):
        self.calls = {}      # key -> [count, args]

    def install(self):
        from infodiffusion_amd import ops
        self.ops = ops
        self.orig = ops.conv_raw
        timer = self

        def recorded(x, w_fwd, bias, residual, sc, sh, seed, salt, p_drop, mode, taps, act, Cout, out_hw_=None, **kw):
            B, Cin, Hs, Ws = x.shape
            Ho, Wo = out_hw_ if out_hw_ is not None else ops.out_hw(mode, Hs, Ws)
            if ops.uses_halo_kernel(x.dtype, taps, act, mode, B, Cin, Cout, Ho, Wo):
                key = (B, Cin, Hs, Ws, Cout, mode, residual is not None, bias is not None)
                if key in timer.calls:
                    timer.calls[key][0] += 1
                else:
                    timer.calls[key] = [1, (x, w_fwd, bias, residual, sc, sh, seed, salt, p_drop, mode, taps, act, Cout,
                                            out_hw_)]
            return timer.orig(x, w_fwd, bias, residual, sc, sh, seed, salt, p_drop, mode, taps, act, Cout, out_hw_, **kw)
        ops.conv_raw = recorded

    def remove(self):
        self.ops.conv_raw = self.orig

    def summary(self, reps=10):
        """(launches per step, total ms per step, FLOPs per step, algorithmic bytes per step)"""
        n = tot_ms = fl = by = 0
        side = torch.cuda.Stream()
        for key, (count, args) in self.calls.items():
            x, w_fwd, _, residual = args[0], args[1], args[2], args[3]
            for _ in range(2):
                y = self.orig(*args)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side), torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
                for _ in range(reps):
                    y = self.orig(*args)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / (3 * reps)
            M = y.shape[0] * y.shape[2] * y.shape[3]
            n += count
            tot_ms += count * ms
            fl += count * 2.0 * M * args[12] * args[10] * x.shape[1]
            by += count * (x.numel() + y.numel() + (residual.numel() if residual is not None else 0)
                           + w_fwd.numel()) * x.element_size()
        return n, tot_ms, fl, by


def large_batch_rate(a, margs, dev, batch=128, steps=10):
    """Supplementary, NOT `value`: the same training step at a per-GPU batch of 128 through the product's own
    `GraphedTrainStep` (run.py's path) -- at B = 32 most launches are one wave of workgroups and the ~3.3 us
    per-launch floor is a fifth of the step; this shows the kernels with four times the work per launch."""
    import copy
    from infodiffusion_amd.models import InfoDiff
    from infodiffusion_amd.optim import FusedClipAdamW
    from infodiffusion_amd.trainer import GraphedTrainStep
    args = copy.copy(margs)
    args.batch_size = batch
    torch.manual_seed(65)
    model = InfoDiff(args, dev, (3, 64, 64)).train()
    opt = FusedClipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, max_norm=1.0)
    step = GraphedTrainStep(model, args, opt, use_graph=bool(a.graph))
    g = torch.Generator(device='cpu')
    g.manual_seed(65)
    pool = [(torch.rand(batch, 3, 64, 64, generator=g) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
            for _ in range(2)]
    for i in range(5):
        step(pool[i % 2], 0)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(steps):
        step(pool[i % 2], 0)
    torch.cuda.synchronize()
    dt = time.time() - t0
    del step, opt, model
    torch.cuda.empty_cache()
    return {'per_gpu_batch': batch, 'value': round(batch * steps / dt, 2), 'unit': 'images/s',
            'ms_per_step': round(dt / steps * 1e3, 3), 'steps': steps, 'note': 'supplementary; value above is B=32'}


def main():
    a = parse()
    if a.cpu_baseline_worker:
        cpu_baseline_worker(a.a_dim, a.cpu_baseline_worker)
        return
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: start the one-process-per-GPU job as a CHILD (nothing here has touched
        # the GPU yet) and leave with its exit code
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
               '--master-addr', '127.0.0.1', '--master-port', os.environ.get('MASTER_PORT', '29541'),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    force_sync = os.environ.get('IDF_FORCE_SYNC') == '1'      # exercise the DP code path on one GPU
    if world > 1 or force_sync:
        if force_sync and world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group('nccl', device_id=dev)
    from infodiffusion_amd.models import InfoDiff
    from infodiffusion_amd.dist import GradSync
    margs = make_args(a)
    torch.manual_seed(64 + rank)
    model = InfoDiff(margs, dev, (3, 64, 64))
    model.train()
    if a.fused_opt:
        from infodiffusion_amd.optim import FusedClipAdamW
        opt = FusedClipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, max_norm=1.0)
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, capturable=bool(a.graph))
    sync = GradSync(model, world, force=force_sync, arena=getattr(opt, 'arena', None)) if (world > 1 or force_sync) \
        else None
    if sync is not None:
        sync.broadcast_parameters()

    g = torch.Generator(device='cpu')
    g.manual_seed(64 + rank)
    # synthetic batches in the layout the input pipeline delivers (data.py / idf_prep_u8: NHWC-dense fp32)
    pool = [(torch.rand(a.batch, 3, 64, 64, generator=g) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
            for _ in range(8)]
    xbuf = pool[0].clone()

    def fwd_bwd():
        loss = model.loss_fn(margs, xbuf)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        return loss

    def tail():
        if sync is not None:
            sync.all_reduce_grads()
        if not a.fused_opt:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()      # FusedClipAdamW: global-norm clip (1.0) + AdamW in three launches

    def step_eager(i):
        xbuf.copy_(pool[i % 8])
        fwd_bwd()
        tail()

    graph = None
    used_graph = False
    # warm-up (eager) -- also builds allocator pools and weight shadows
    for i in range(max(2, a.warmup if not a.graph else 3)):
        step_eager(i)
    torch.cuda.synchronize()
    if a.graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step_eager(0)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            opt.zero_grad(set_to_none=True)
            # thread_local: RCCL's watchdog thread keeps querying events while this thread captures
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                fwd_bwd()
                if sync is None:
                    tail()
            used_graph = True
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print('graph capture failed (%s: %s); running eager' % (type(e).__name__, str(e)[:200]), file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    def step(i):
        if graph is not None:
            xbuf.copy_(pool[i % 8])
            graph.replay()
            if sync is not None:
                tail()
        else:
            step_eager(i)

    for i in range(a.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(a.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    # integrity of the timed region: the last step's global gradient norm (device scalar of the fused optimizer)
    gnorm = float(opt.total_norm()) if a.fused_opt else None
    if gnorm is not None and not (gnorm == gnorm and gnorm < 1e6):
        raise RuntimeError('bench: non-finite gradient norm %r in the timed training step' % gnorm)
    imgs = a.batch * world * a.steps
    out = {
        'metric': 'training images/sec, CelebA 64x64 (InfoDiff loss_fn fwd+bwd+clip+AdamW)',
        'value': round(imgs / dt, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: CelebA 3x64x64 a_dim=%d mmd_weight=0.1 T=1000 dropout=0.1 '
                               'train step, random-pixel batches, random-init weights' % a.a_dim,
                   'per_gpu_batch': a.batch, 'global_batch': a.batch * world,
                   'parallelism': 'dp%d' % world, 'hipgraph': used_graph,
                   'optimizer': 'fused clip+AdamW' if a.fused_opt else 'clip_grad_norm_ + torch AdamW',
                   'last_grad_norm': None if gnorm is None else round(gnorm, 4)},
    }

    if rank == 0 and not a.no_roofline:
        # dominant kernel: the halo 3x3 conv (forward + data-gradient launches)
        tm = ConvTimer()
        tm.install()
        fwd_bwd()       # every conv launch of a step; no exchange / optimizer: the other ranks are not in this block
        tm.remove()
        torch.cuda.synchronize()
        nrep = 1
        n, tot_ms, fl, by = tm.summary()
        ach = fl / (tot_ms * 1e-3) / 1e12
        peak = 2500.0 if a.dtype == 'bf16' else 157.3
        out['roofline'] = {'kernel': 'conv3x3_halo_bf16 (3x3 conv forward + data-gradient launches)', 'bound': 'mfma',
                           'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                           'traffic': pmc_traffic('conv3x3_halo_bf16'), 'launches_per_step': n // nrep,
                           'avg_launch_us': round(tot_ms * 1e3 / n, 2),
                           'algorithmic_gflop_per_step': round(fl / nrep / 1e9, 1),
                           'algorithmic_bytes_per_launch': round(by / n),
                           'algorithmic_gbs': round(by / (tot_ms * 1e-3) / 1e9, 1)}
    if world > 1:
        dist.barrier()

    if not a.no_sampling:
        # second headline metric (BASELINE configs[2]): DDIM-100 sampling, B = 256 per GPU.  The image batch is
        # sharded over the ranks with NO data-path collective; the barrier / MAX below only bracket the timing.
        from infodiffusion_amd.sampling import DiffusionProcess
        import copy
        sargs = copy.copy(margs)
        sargs.diffusion_steps = 100
        sargs.deterministic = True
        torch.manual_seed(64 + rank)
        smodel = InfoDiff(sargs, dev, (3, 64, 64)).eval()
        proc = DiffusionProcess(sargs, smodel, dev, (3, 64, 64))
        proc.sampling(8)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.time()
        proc.sampling(a.sampling_batch)
        torch.cuda.synchronize()
        ds = time.time() - t0
        if world > 1:
            tmax = torch.tensor([ds], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            ds = float(tmax)
        out['sampling'] = {'metric': 'DDIM-100 sampling images/sec (B=%d per GPU, 100 network evaluations, batch '
                                     'sharded over the GPUs, no collective)' % a.sampling_batch,
                           'value': round(a.sampling_batch * world / ds, 2), 'unit': 'images/s', 'n_gpus': world,
                           'seconds': round(ds, 3)}
        del proc, smodel
        torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not a.no_large_batch:
        out['large_batch'] = large_batch_rate(a, margs, dev)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(margs)
    if rank == 0:
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
