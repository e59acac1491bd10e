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
    ap.add_argument('--no-dp-probe', action='store_true', help='skip the one-rank data-parallel step (child process, IDF_FORCE_SYNC=1)')
    ap.add_argument('--cpu-baseline-worker', default=None, help=argparse.SUPPRESS)
    return ap.parse_args()


def make_args(a):
    from types import SimpleNamespace
    return SimpleNamespace(beta1=1e-5, betaT=1e-2, diffusion_steps=1000, input_size=64, input_channels=3,
                           is_bottleneck=False, unets_channels=64, encoder_channels=64, a_dim=a.a_dim,
                           mmd_weight=0.1, kld_weight=0.0, prior='regular', batch_size=a.batch, use_C=False,
                           C_max=25.0, epochs=50, deterministic=True, model='diff', split_step=500, mode='train',
                           is_latent=False, act_dtype=a.dtype, dataset='celeba')


# the oracle's convs stop scaling (and oversubscribe) far below the box's 256 cores: measured on the MI355X host at B = 32,
# s per train step (steps 2-4) / s per backbone evaluation -- 8 threads 3.8-4.7 / 0.28, 16 threads 3.0-4.6 / 0.16, 32 threads
# 4.7-5.8 / 0.31 (profiles/r04_cpu_baseline_threads.txt, tools/cpu_threads_sweep.sh); 64 threads 8.7-10, 128 threads 18-25
# (profiles/r03_cpu_baseline_threads.txt).  16 is the fastest measured: the default.  IDF_CPU_THREADS overrides for a sweep
CPU_THREADS = int(os.environ.get('IDF_CPU_THREADS', '16'))      # (infodiffusion_amd/knobs.py lists it; read here before the package loads)
CPU_BATCH = 32        # the benchmarked batch, CPU_WARMUP warm-ups + 3 timed steps; then 3 backbone evaluations
CPU_WARMUP = 3        # the series is still falling at step 4 with one warm-up (round-4 verdict)


def cpu_baseline_worker(a_dim, out_path):
    """Child process: the CPU oracle (restatement of the reference's stock-ATen path, fp32 NCHW) doing
    the same training step on host cores, then the sampler's network evaluation; appends one line per
    measurement to `out_path` as it goes ('train <s>' / 'eval <s>')."""
    from oracle import infodiff_oracle as O
    torch.set_num_threads(CPU_THREADS)
    cfg = O.dataset_cfg('celeba', a_dim=a_dim, mmd_weight=0.1)
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
    for it in range(CPU_WARMUP + 3):
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
            f.write('train %.4f\n' % (time.time() - t0))
    # sampling (sampling.py:41-60 calls the backbone once per step; models.py:705-723): 1 warm-up + 3 evaluations
    sdd = {k: v.detach() for k, v in sd.items()}
    xs = torch.randn(B, 3, 64, 64, generator=g)
    av = torch.randn(B, cfg.a_dim, generator=g)
    with torch.no_grad():
        for it in range(4):
            t0 = time.time()
            O.infodiff_eps(sdd, cfg, xs, 50, av)
            with open(out_path, 'a') as f:
                f.write('eval %.4f\n' % (time.time() - t0))


def cpu_baseline(margs, budget_s=240):
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
    train, evals = [], []
    try:
        for line in open(out).read().splitlines():
            kind, v = line.split()
            (train if kind == 'train' else evals).append(float(v))
    except (OSError, ValueError):
        pass
    finally:
        if os.path.exists(out):
            os.unlink(out)
    if not train:
        return {'value': None, 'unit': 'images/s', 'cores': CPU_THREADS, 'kind': 'port',
                'sample': 'CPU oracle did not finish one B=%d step within %d s' % (CPU_BATCH, budget_s)}
    # the first steps are still warming (allocator, thread pools: 6.5, 4.6, 3.8, 3.0 s in profiles/r04_cpu_baseline_threads.txt):
    # CPU_WARMUP of them are dropped, the mean of the rest is the baseline
    timed = train[CPU_WARMUP:] if len(train) > CPU_WARMUP else train[-1:]
    t = sum(timed) / len(timed)
    res = {'value': round(CPU_BATCH / t, 3), 'unit': 'images/s', 'cores': CPU_THREADS, 'kind': 'port',
           'sample': 'CPU oracle (fp32 NCHW stock-ATen restatement of the reference), CelebA 64x64 train step '
                     '(fwd+bwd+clip+AdamW, dropout on) at B=%d on %d threads (host has %d cores; ATen convs do not scale '
                     'past a few dozen threads: profiles/r04_cpu_baseline_threads.txt): %d timed step(s) after %d warm-ups, '
                     '%.2f s/step (every step: %s s)' % (CPU_BATCH, CPU_THREADS, os.cpu_count() or 0, len(timed),
                                                         min(CPU_WARMUP, len(train) - len(timed)), t,
                                                         ' '.join('%.2f' % v for v in train))}
    if len(evals) > 1:
        te = sum(evals[1:]) / len(evals[1:])
        res['sampling'] = {'value': round(CPU_BATCH / (100 * te), 4), 'unit': 'images/s',
                           'sample': 'DDIM-100 = 100 backbone evaluations per image batch: %d timed evaluations at B=%d '
                                     'after 1 warm-up, %.2f s each, x100' % (len(evals) - 1, CPU_BATCH, te)}
    return res


def pmc_traffic_file(names):
    """Average HBM bytes per launch over the kernels whose name contains one of `names`, from the committed rocprofv3
    PMC passes of the latest round (profiles/rNN_pmc_traffic.json: FETCH_SIZE x2 + WRITE_SIZE), or None."""
    try:
        files = sorted(fn for fn in os.listdir(os.path.join(ROOT, 'profiles')) if fn.endswith('_pmc_traffic.json'))
        with open(os.path.join(ROOT, 'profiles', files[-1])) as f:      # the latest round's passes
            d = json.load(f)
        ks = [v for k, v in d.items() if any(n in k for n in names)]
        n = sum(v['launches'] for v in ks)
        return round(sum(v['launches'] * v['hbm_bytes_avg'] for v in ks) / n) if n else None
    except (OSError, ValueError, KeyError):
        return None


def parity_file():
    """The measured parity of the benchmarked configuration against the reference's fp32 outputs (tools/parity_summary.py on an
    MI355X -> profiles/rNN_parity.json, latest round), or None: the tolerance the number is quoted at travels with the number."""
    try:
        files = sorted(fn for fn in os.listdir(os.path.join(ROOT, 'profiles')) if fn.endswith('_parity.json'))
        with open(os.path.join(ROOT, 'profiles', files[-1])) as f:
            d = json.load(f)
        d['source'] = 'profiles/' + files[-1]
        return d
    except (OSError, ValueError, IndexError):
        return None


def dp_timeline(trainer, sync, pool, steps=10):
    """Where a data-parallel step's time goes (N > 1, or IDF_FORCE_SYNC on one rank): HIP events around the three graphs on the
    compute stream and around both all-reduces on the exchange stream, over `steps` extra replays OUTSIDE the timed region.
    exchange_exposed_ms = compute-stream time between the end of G2 and the start of G3 (the join: whatever of either collective
    the encoder's backward pass did not cover, + the packed buckets)."""
    if not isinstance(trainer.graph, tuple):
        return None
    trainer.trace, sync.trace = [], {'early': [], 'late': []}
    for i in range(steps):
        trainer(pool[i % len(pool)], 0)
    torch.cuda.synchronize()
    tr, st = trainer.trace, sync.trace
    trainer.trace = sync.trace = None
    tr = [m for m in tr if len(m) == 5]
    if not tr:
        return None

    def med(v):
        v = sorted(v)
        return round(v[len(v) // 2], 4) if v else None
    out = {'steps': len(tr),
           'g1_fwd_backbone_bwd_ms': med([m[0].elapsed_time(m[1]) for m in tr]),
           'g2_encoder_bwd_ms': med([m[1].elapsed_time(m[2]) for m in tr]),
           'exchange_exposed_ms': med([m[2].elapsed_time(m[3]) for m in tr]),
           'g3_clip_adamw_ms': med([m[3].elapsed_time(m[4]) for m in tr]),
           'step_ms': med([m[0].elapsed_time(m[4]) for m in tr])}
    for which in ('early', 'late'):
        ev = st[which]
        out['allreduce_%s_ms' % which] = med([a.elapsed_time(b) for a, b, _ in ev])
        out['allreduce_%s_bytes' % which] = ev[0][2] if ev else None
    if st['early'] and len(st['early']) == len(tr):
        # how much of the early (backbone-slice) collective ran under G2: its end against G2's end on the compute stream
        tail = [max(0.0, m[2].elapsed_time(e[1])) for m, e in zip(tr, st['early'])]
        out['allreduce_early_tail_past_g2_ms'] = med(tail)
    return out


def one_rank_dp_cost(a):
    """N = 1 only: the same bench in a CHILD process with IDF_FORCE_SYNC=1 (RCCL world size 1): the three-graph step with both
    collectives, the synchronised convs off in the encoder's backward pass -- the kernel set and launch structure a rank of an
    N > 1 run executes, minus the wire.  -> its ms_per_step and timeline (the expected per-rank step next to `value`)."""
    env = dict(os.environ)
    env['IDF_FORCE_SYNC'] = '1'
    env.setdefault('MASTER_PORT', '29547')
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', str(a.steps), '--warmup', str(a.warmup),
           '--no-roofline', '--no-sampling', '--no-large-batch', '--no-cpu-baseline', '--no-dp-probe',
           '--batch', str(a.batch), '--a_dim', str(a.a_dim), '--dtype', a.dtype, '--graph', str(int(a.graph)),
           '--fused-opt', str(int(a.fused_opt))]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=180)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
        d = json.loads(line)
        return {'ms_per_step': d['ms_per_step'], 'ms_per_step_median': d.get('ms_per_step_median'), 'value': d['value'],
                'timeline': d.get('dp_timeline'), 'ranks_seen': d.get('ranks_seen'),
                'note': 'IDF_FORCE_SYNC=1 child process, RCCL world size 1: three graphs + two all-reduces per step, synchronised '
                        'convs off in the encoder backward (trainer._shares_chip) -- the per-rank step of an N > 1 run without the wire'}
    except Exception as e:  # noqa: BLE001
        return {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}


class LaunchRecorder:
    """Cold HIP-event timing of the step's hot kernels.  One eager step records every C-ABI call of the kernel
    families below; each distinct launch configuration is then replayed from a captured hipGraph between one HIP event
    pair -- every replayed call on its OWN set of activation buffers, >= 4 sets and > 256 MB (the Infinity Cache)
    together, so no call finds its operands in L2 / MALL because the previous call left them there.  (An event pair
    around every single eager launch measures the events: +10 us per launch; replaying one launch on the same buffers
    measures the caches.)  The per-launch times are weighted by the step's launch counts."""

    def __init__(self):
        self.calls = {}      # (name, signature) -> [count, args]
        self.order = []

    # ---- per entry point: (family, [(arg index, bytes)] of the activation-sized buffers, work)
    @staticmethod
    def _spec(name, a):
        if name == 'idf_conv3x3_bf16':
            B, H, W, Cin, Cout, mode = a[5:11]
            Hs, Ws = (2 * H, 2 * W) if mode == 1 else ((H // 2, W // 2) if mode >= 2 else (H, W))
            yb = B * H * W * Cout * 2
            bufs = [(0, B * Hs * Ws * Cin * 2), (4, yb)] + ([(3, yb)] if a[3] else [])
            return 'conv3x3', bufs, 2.0 * B * H * W * Cout * 9 * Cin / (4 if mode == 3 else 1)
        if name in ('idf_conv_gn_bf16', 'idf_conv_gn_sc_bf16'):
            C1 = a[2]
            B, H, W, Cin, Cout, taps = a[29:35]
            if taps != 9:
                return None
            px = B * H * W
            bufs = [(0, px * (C1 if a[1] else Cin) * 2), (21, px * Cout * 2)]
            if a[1]:
                bufs.append((1, px * (Cin - C1) * 2))
            if a[20]:
                bufs.append((20, px * Cout * 2))
            if a[22]:
                bufs.append((22, px * Cin * 2))
            work = 2.0 * px * Cout * 9 * Cin
            if name == 'idf_conv_gn_sc_bf16':       # the block's 1x1 shortcut rides in the launch: its output and its FLOPs
                bufs.append((38, px * a[39] * 2))
                work += 2.0 * px * a[39] * Cin
            return 'conv3x3', bufs, work
        if name == 'idf_conv_dgrad_gn_bf16':      # small maps: data-gradient conv + GroupNorm backward in one launch
            B, H, W, Cin, Cout, taps = a[25:31]
            if taps != 9:
                return None
            px = B * H * W
            bufs = [(0, px * Cin * 2), (2, px * Cout * 2), (5, px * Cout * 2)]
            bufs += [(i, px * Cout * 2) for i in (3, 4) if a[i]]
            return 'conv3x3', bufs, 2.0 * px * Cout * 9 * Cin
        if name == 'idf_conv_dgrad_chain_bf16':    # big maps: data-gradient conv with the du epilogue (and / or the dy prologue)
            B, H, W, Cin, Cout, taps = a[31:37]
            if taps != 9:
                return None
            px = B * H * W
            bufs = [(0, px * Cin * 2), (29, px * Cout * 2)]
            if a[1]:
                bufs.append((1, px * Cin * 2))
            if a[18]:
                bufs.append((18, px * Cin * 2))
            if a[20]:
                C1 = a[22] if a[21] else Cout
                bufs.append((20, px * C1 * 2))
                if a[21]:
                    bufs.append((21, px * (Cout - C1) * 2))
            return 'conv3x3', bufs, 2.0 * px * Cout * 9 * Cin
        if name == 'idf_conv_rs_gn_bf16':          # round 5: the GroupNorm-prologue conv of the big maps in the row-reuse form
            B, H, W, Cin, Cout = a[24:29]
            px = B * H * W
            bufs = [(0, px * Cin * 2), (17, px * Cout * 2)]
            if a[16]:
                bufs.append((16, px * Cout * 2))
            if a[18]:
                bufs.append((18, px * Cin * 2))
            return 'conv3x3', bufs, 2.0 * px * Cout * 9 * Cin
        if name == 'idf_conv_rs_dgrad_chain_bf16':  # ... and the data-gradient conv with the du epilogue
            B, H, W, Cin, Cout = a[13:18]
            px = B * H * W
            C1 = a[4] if a[3] else Cout
            bufs = [(0, px * Cin * 2), (11, px * Cout * 2), (2, px * C1 * 2)]
            if a[3]:
                bufs.append((3, px * (Cout - C1) * 2))
            return 'conv3x3', bufs, 2.0 * px * Cout * 9 * Cin
        if name == 'idf_conv_rs_dgrad_gn_bf16':     # ... and with the whole GroupNorm backward behind it (du never written):
            B, H, W, Cin, Cout = a[30:35]           # credited with the conv's FLOPs only, like the du-epilogue launches
            px = B * H * W
            C1 = a[4] if a[3] else Cout
            bufs = [(0, px * Cin * 2), (2, px * C1 * 2), (13, px * C1 * 2)]
            if a[3]:
                bufs += [(3, px * (Cout - C1) * 2), (14, px * (Cout - C1) * 2)]
            for i in (11, 12):
                if a[i]:
                    bufs.append((i, px * Cout * 2))
            return 'conv3x3', bufs, 2.0 * px * Cout * 9 * Cin
        if name == 'idf_conv_dgrad_chain_sc_bf16':  # ... with the block shortcut's data gradient riding in the launch
            B, H, W, Cin, Cout = a[13:18]
            sc_Cin = a[22]
            px = B * H * W
            C1 = a[4] if a[3] else Cout
            bufs = [(0, px * Cin * 2), (11, px * Cout * 2), (2, px * C1 * 2), (19, px * sc_Cin * 2), (21, px * Cout * 2)]
            if a[3]:
                bufs.append((3, px * (Cout - C1) * 2))
            return 'conv3x3', bufs, 2.0 * px * Cout * 9 * Cin + 2.0 * px * Cout * sc_Cin
        # UpSample / DownSample in their sub-pixel forms: the ALGORITHMIC work stays the reference's (a 3x3 conv over the up-sampled
        # image, its data gradient, the transposed stride-2 conv) -- the launches execute 4/9, 4/9 and all of it respectively
        if name == 'idf_upconv_bf16':
            B, Hl, Wl, Cin, Cout = a[5:10]
            px = B * 4 * Hl * Wl
            return 'conv3x3', [(0, B * Hl * Wl * Cin * 2), (3, px * Cout * 2)], 2.0 * px * Cout * 9 * Cin
        if name == 'idf_upconv_dgrad_bf16':
            B, Hl, Wl, Cin, Cout = a[3:8]
            px = B * 4 * Hl * Wl
            return 'conv3x3', [(0, px * Cout * 2), (2, B * Hl * Wl * Cin * 2)], 2.0 * px * Cout * 9 * Cin
        if name == 'idf_downconv_dgrad_bf16':
            B, Hl, Wl, Cin, Cout = a[4:9]
            px = B * Hl * Wl
            bufs = [(0, px * Cout * 2), (3, 4 * px * Cin * 2)] + ([(2, 4 * px * Cin * 2)] if a[2] else [])
            return 'conv3x3', bufs, 2.0 * px * Cout * 9 * Cin
        if name == 'idf_conv3x3_fewc_bf16':      # the head conv (Cin <= 3): one MFMA K-step per output tile
            B, H, W, Cin, Cout = a[4:9]
            px = B * H * W
            return 'conv3x3', [(0, px * Cin * 2), (3, px * Cout * 2)], 2.0 * px * Cout * 9 * Cin
        if name == 'idf_conv_wr_gn_bf16':        # small maps, fragment-major weights: GroupNorm-prologue conv
            C1 = a[2]
            B, H, W, Cin, Cout = a[27:32]
            px = B * H * W
            bufs = [(0, px * (C1 if a[1] else Cin) * 2), (20, px * Cout * 2)]
            if a[1]:
                bufs.append((1, px * (Cin - C1) * 2))
            if a[19]:
                bufs.append((19, px * Cout * 2))
            if a[21]:
                bufs.append((21, px * Cin * 2))
            return 'conv3x3', bufs, 2.0 * px * Cout * 9 * Cin
        if name == 'idf_conv_wr_dgrad_gn_bf16':  # ... data-gradient conv + GroupNorm backward
            B, H, W, Cin, Cout = a[24:29]
            px = B * H * W
            bufs = [(0, px * Cin * 2), (2, px * Cout * 2), (5, px * Cout * 2)]
            bufs += [(i, px * Cout * 2) for i in (3, 4) if a[i]]
            return 'conv3x3', bufs, 2.0 * px * Cout * 9 * Cin
        if name == 'idf_resblock_small_fwd':     # image-resident 8x8 ResBlock: 2-3 convs (+ the 1x1 shortcut) in one launch
            A = a[0]._obj
            px, n = A.B * 64, A.nstage
            bufs = [('x', px * (A.C1 if A.x2 else A.Cin) * 2), ('y', px * 128 * 2)]
            if A.x2:
                bufs.append(('x2', px * (A.Cin - A.C1) * 2))
            for i in range(n):
                if A.s[i].a_out:
                    bufs.append((('s', i, 'a_out'), px * (A.Cin if i == 0 else 128) * 2))
                if A.s[i].h_out and i < n - 1:           # (the last stage's h_out IS y)
                    bufs.append((('s', i, 'h_out'), px * 128 * 2))
            work = 2.0 * px * 128 * 9 * (A.Cin + 128 * (n - 1)) + (2.0 * px * 128 * A.Cin if A.w_sc else 0.0)
            return 'conv3x3', bufs, work
        if name == 'idf_resblock_small_bwd':     # ... its data-gradient convs + GroupNorm backwards
            A = a[0]._obj
            px, n = A.B * 64, A.nstage - A.first
            bufs = [('dy', px * 128 * 2)]
            if A.dres2:
                bufs.append(('dres2', px * 128 * 2))
            for i in range(A.first, A.nstage):
                bufs += [(('s', i, 'x'), px * 128 * 2), (('s', i, 'dx'), px * 128 * 2)]
            return 'conv3x3', bufs, 2.0 * px * 128 * 9 * 128 * n
        if name == 'idf_gn_bwd_apply':
            C1 = a[5]
            B, HW, C = a[24:27]
            c1 = C1 if a[4] else C
            bufs = [(0, B * HW * C * 2), (3, B * HW * c1 * 2), (8, B * HW * c1 * 2)]
            if a[4]:
                bufs += [(4, B * HW * (C - c1) * 2), (9, B * HW * (C - c1) * 2)]
            bufs += [(i, B * HW * C * 2) for i in (6, 7) if a[i]]
            return 'gn_bwd', bufs, float(sum(n for _, n in bufs))
        if name == 'idf_gn_fused_bwd':
            C1 = a[3]
            B, HW, C, dt = a[27:31]
            esz = 2 if dt == 1 else 4
            c1 = C1 if a[2] else C
            bufs = [(0, B * HW * C * esz), (1, B * HW * c1 * esz), (6, B * HW * c1 * esz)]
            if a[2]:
                bufs += [(2, B * HW * (C - c1) * esz), (7, B * HW * (C - c1) * esz)]
            if a[4]:
                bufs.append((4, B * HW * C * esz))
            if a[5]:
                bufs.append((5, B * HW * C * esz))
            return 'gn_bwd', bufs, float(sum(n for _, n in bufs))
        if name == 'idf_attn_fwd':
            B, N, C = a[3:6]
            return 'attn', [(0, B * N * 3 * C * 2), (1, B * N * C * 2)], 4.0 * B * N * N * C
        if name == 'idf_attn_fwd_res':           # attention with the residual and the statistics in its epilogue (proj folded into V)
            B, N, C = a[6:9]
            bufs = [(0, B * N * 3 * C * 2), (1, B * N * C * 2), (4, B * N * C * 2)] + ([(2, B * N * C * 2)] if a[2] else [])
            return 'attn', bufs, 4.0 * B * N * N * C
        return None

    @staticmethod
    def _stream_idx(name, args):
        """Index of the stream argument (the last one, except where a rider's arguments follow it)."""
        return {'idf_conv_gn_sc_bf16': 35, 'idf_conv_dgrad_chain_sc_bf16': 18}.get(name, len(args) - 1)

    def install(self):
        from infodiffusion_amd import ops
        self.ops = ops
        self.orig = ops.call
        rec = self

        def recorded(name, *args):
            sp = rec._spec(name, args)
            if sp is not None and name.startswith('idf_resblock_small_'):
                import ctypes
                A = args[0]._obj
                key = (name, A.B, A.nstage, getattr(A, 'first', 0), getattr(A, 'Cin', 128), bool(getattr(A, 'w_sc', None)),
                       tuple(n for _, n in sp[1]))
                if key in rec.calls:
                    rec.calls[key][0] += 1
                else:          # a private copy of the argument block: the caller's goes away with its autograd node
                    rec.calls[key] = [1, (name, (ctypes.byref(type(A).from_buffer_copy(A)), args[1]))]
            elif sp is not None:
                ptr = {i for i, _ in sp[1]}
                si = rec._stream_idx(name, args)
                key = (name,) + tuple((v is not None) if (i in ptr or isinstance(v, int) and v > (1 << 32)) else v
                                      for i, v in enumerate(args) if i != si)
                if key in rec.calls:
                    rec.calls[key][0] += 1
                else:
                    rec.calls[key] = [1, (name, args)]
            return rec.orig(name, *args)
        ops.call = recorded

    def remove(self):
        self.ops.call = self.orig

    def measure(self, dev, reps_min=8):
        """family -> (launches per step, ms per step, work per step, bytes per step)"""
        fam = {}
        side = torch.cuda.Stream()
        for key, (count, (name, args)) in self.calls.items():
            family, bufs, work = self._spec(name, args)
            si = self._stream_idx(name, args)
            per_set = sum(n for _, n in bufs)
            K = max(4, min(16, -(-(320 << 20) // per_set)))
            sets = []
            for k in range(K if name.startswith('idf_resblock_small_') else 0):
                import ctypes
                A = type(args[0]._obj).from_buffer_copy(args[0]._obj)
                keep = [A]
                for path, n in bufs:
                    t = torch.randn(n // 2, device=dev, dtype=torch.bfloat16)
                    keep.append(t)
                    if isinstance(path, tuple):
                        setattr(A.s[path[1]], path[2], t.data_ptr())
                    else:
                        setattr(A, path, t.data_ptr())
                if name == 'idf_resblock_small_fwd':
                    A.s[A.nstage - 1].h_out = A.y
                if name == 'idf_resblock_small_bwd':     # not into the live gradient arena: per-sample sums to scratch
                    for i in range(A.first, A.nstage):
                        t = torch.empty(A.B * 2 * 128, device=dev, dtype=torch.float32)
                        keep.append(t)
                        A.s[i].dgb, A.s[i].dgamma_acc, A.s[i].dbeta_acc = t.data_ptr(), None, None
                sets.append(([ctypes.byref(A), None], keep))
            for k in range(0 if sets else K):
                al = list(args)
                keep = []
                for i, n in bufs:
                    t = torch.randn(n // 2, device=dev, dtype=torch.bfloat16)
                    keep.append(t)
                    al[i] = t.data_ptr()
                if name == 'idf_gn_fused_bwd':      # not into the live gradient arena: per-sample sums to scratch
                    B_, C_ = al[27], al[29]
                    t = torch.empty(B_ * 2 * C_, device=dev, dtype=torch.float32)
                    keep.append(t)
                    al[20], al[21], al[22] = t.data_ptr(), None, None
                if name == 'idf_conv_dgrad_gn_bf16':
                    B_, C_ = al[25], al[29]
                    t = torch.empty(B_ * 2 * C_, device=dev, dtype=torch.float32)
                    keep.append(t)
                    al[18], al[19], al[20] = t.data_ptr(), None, None
                if name == 'idf_conv_wr_dgrad_gn_bf16':
                    B_, C_ = al[24], al[28]
                    t = torch.empty(B_ * 2 * C_, device=dev, dtype=torch.float32)
                    keep.append(t)
                    al[18], al[19], al[20] = t.data_ptr(), None, None
                if name == 'idf_gn_bwd_apply':       # not into the live gradient arena: per-sample sums to scratch
                    B_, C_ = al[24], al[26]
                    t = torch.empty(B_ * 2 * C_, device=dev, dtype=torch.float32)
                    keep.append(t)
                    al[21], al[22], al[23] = t.data_ptr(), None, None
                if name == 'idf_conv_dgrad_chain_bf16' and al[1]:
                    B_, C_ = al[31], al[34]
                    t = torch.empty(B_ * 2 * C_, device=dev, dtype=torch.float32)
                    keep.append(t)
                    al[15], al[16], al[17] = t.data_ptr(), None, None
                al[si] = None
                sets.append((al, keep))
            reps = max(1, -(-reps_min // K))

            def run_all(stream_ptr):
                for _ in range(reps):
                    for al, _k in sets:
                        al[si] = stream_ptr
                        self.orig(name, *al)
            run_all(torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side), torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
                run_all(side.cuda_stream)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / (3 * reps * K)
            f = fam.setdefault(family, [0, 0.0, 0.0, 0.0])
            f[0] += count
            f[1] += count * ms
            f[2] += count * work
            f[3] += count * per_set
            del sets, g
        return fam


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

    # the product's own training step (run.py's): loss_fn -> zero_grad -> backward -> [exchange] -> clip + AdamW, two
    # eager steps, then the step captured once and replayed (trainer.GraphedTrainStep: ONE graph on one GPU; with a
    # gradient exchange THREE graphs -- forward + backbone backward | encoder backward | clip + AdamW -- with the two
    # all-reduces issued eagerly between them, the first overlapping the encoder's backward pass; a capture that fails
    # on any rank takes every rank to eager steps together)
    from infodiffusion_amd.trainer import GraphedTrainStep
    pre = None if a.fused_opt else (lambda: torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0))
    trainer = GraphedTrainStep(model, margs, opt, sync=sync, use_graph=bool(a.graph), pre_step=pre)

    def step(i):
        trainer(pool[i % 8], 0)

    for i in range(max(4, a.warmup)):      # eager warm-up, capture (if any), first replay
        step(i)
    torch.cuda.synchronize()
    used_graph = trainer.graph is not None

    for i in range(a.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.time()
    marks[0].record()
    for i in range(a.steps):
        step(i)
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
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
        'ms_per_step': round(dt / a.steps * 1e3, 3), 'ms_per_step_median': round(median_ms, 3),
        'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: CelebA 3x64x64 a_dim=%d mmd_weight=0.1 T=1000 dropout=0.1 '
                               'train step, random-pixel batches, random-init weights' % a.a_dim,
                   'per_gpu_batch': a.batch, 'global_batch': a.batch * world,
                   'parallelism': 'dp%d' % world, 'hipgraph': used_graph,
                   'exchange': None if sync is None else (
                       'three graphs (forward + backbone backward | encoder backward | clip + AdamW), all-reduce of the backbone '
                       'slice of the gradient arena eagerly between the first two (overlapping the encoder backward), the rest before the third'
                       if (used_graph and trainer.split) else 'all-reduce of the gradient arena between backward and the optimizer'),
                   'optimizer': 'fused clip+AdamW' if a.fused_opt else 'clip_grad_norm_ + torch AdamW',
                   'last_grad_norm': None if gnorm is None else round(gnorm, 4)},
    }

    par = parity_file()
    if par is not None:
        out['parity'] = par
    if sync is not None:
        out['ranks_seen'] = dist.get_world_size()       # after RCCL init: the ranks that actually take part in the exchange
        out['dp_timeline'] = dp_timeline(trainer, sync, pool)
    if rank == 0 and not a.no_roofline:
        # dominant kernel family: the 3x3 convs (forward incl. the GroupNorm-prologue form, and data gradients)
        rec = LaunchRecorder()
        rec.install()
        trainer.forward_backward(pool[0])       # every launch of a step; no exchange / optimizer: the other ranks are not in this block
        rec.remove()
        torch.cuda.synchronize()
        fam = rec.measure(dev)
        peak = 2500.0 if a.dtype == 'bf16' else 157.3
        if 'conv3x3' in fam:
            n, ms, fl, by = fam['conv3x3']
            ach = fl / (ms * 1e-3) / 1e12
            out['roofline'] = {'kernel': '3x3 conv family (conv_rs_bf16 / conv_ps_bf16 / conv_dlds_bf16 / conv3x3_halo_bf16 / conv3x3_fewc_bf16, the sub-pixel UpSample / DownSample kernels upconv_bf16 / '
                                         'upconv_dgrad_bf16 / downconv_dgrad_bf16 -- counted with the reference\'s 3x3 FLOPs, of which they execute 4/9, 4/9 and 1/1 --, and on the '
                                         'small maps conv_wr_kernel / resblock8_fwd_kernel / resblock8_bwd_kernel, whose launches hold 2-3 convs: '
                                         'forward incl. GroupNorm-prologue launches + data-gradient launches incl. those whose epilogue is the '
                                         'GroupNorm backward (small maps; round 5: also the big maps, behind an in-launch wait of the image\'s workgroups -- conv_rs_bf16 EPI 3) or its du / partial-sum half (the big-map stages that form does not cover))',
                               'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                               'frac': round(ach / peak, 4),
                               'traffic': pmc_traffic_file(['conv_rs_bf16', 'conv_ps_bf16', 'conv_dlds_bf16', 'conv3x3_halo_bf16', 'conv3x3_fewc_bf16', 'conv_wr_kernel', 'upconv_', 'downconv_',
                                                            'resblock8_fwd_kernel', 'resblock8_bwd_kernel']),
                               'launches_per_step': n, 'avg_launch_us': round(ms * 1e3 / n, 2),
                               'timing': 'cold: each replayed launch on its own buffer set, sets > 256 MB together',
                               'algorithmic_gflop_per_step': round(fl / 1e9, 1),
                               'algorithmic_bytes_per_launch': round(by / n),
                               'algorithmic_gbs': round(by / (ms * 1e-3) / 1e9, 1)}
        if 'gn_bwd' in fam:
            n, ms, _, by = fam['gn_bwd']
            gbs = by / (ms * 1e-3) / 1e9
            out['roofline_hbm'] = {'kernel': 'gn_bwd_apply (the streaming half of the GroupNorm + FiLM + SiLU + dropout backward: '
                                             'the ResBlock elementwise pass that remains a kernel of its own; + the two tail '
                                             'GroupNorms still in gn_small_bwd)', 'bound': 'hbm',
                                   'achieved': round(gbs, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(gbs / 8000.0, 4),
                                   'traffic': pmc_traffic_file(['gn_bwd_apply', 'gn_small_bwd']), 'launches_per_step': n,
                                   'avg_launch_us': round(ms * 1e3 / n, 2),
                                   'algorithmic_bytes_per_launch': round(by / n),
                                   'note': 'bytes = du + x (+ branch gradients) read, dx written, once each'}
        if 'attn' in fam:
            n, ms, fl, by = fam['attn']
            ach = fl / (ms * 1e-3) / 1e12
            out['roofline_attn'] = {'kernel': 'attn_fwd_res_kernel (N = 256 tokens at 16x16, N = 64 at the 8x8 middle block; d = 128: QK^T, softmax, PV, + x, statistics of y in one launch; the proj conv is folded into V)',
                                    'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                                    'frac': round(ach / peak, 4), 'traffic': pmc_traffic_file(['attn_fwd_res_kernel', 'attn_fwd_kernel']),
                                    'launches_per_step': n, 'avg_launch_us': round(ms * 1e3 / n, 2),
                                    'note': '4 N^2 d FLOP per image; launch-bound at B = 32 (32 x 4 workgroups)'}
    if world > 1:
        dist.barrier()

    sampler_capture_failed = False
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
        small_stats = dict(proc.graph_stats)
        # the reference's eval flows sample batch after batch (run.py:255-259, 284-287): the step graph of a batch shape is captured by
        # the first batch and replayed by the following ones -- the timed batch is such a following one
        torch.cuda.synchronize()
        t0 = time.time()
        proc.sampling(a.sampling_batch)
        torch.cuda.synchronize()
        ds_first = time.time() - t0
        if world > 1:
            dist.barrier()
        before = dict(proc.graph_stats)
        t0 = time.time()
        proc.sampling(a.sampling_batch)
        torch.cuda.synchronize()
        ds = time.time() - t0
        big_stats = {k: proc.graph_stats[k] - before[k] for k in before}
        if world > 1:
            tmax = torch.tensor([ds], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            ds = float(tmax)
        out['sampling'] = {'metric': 'DDIM-100 sampling images/sec (B=%d per GPU, 100 network evaluations, batch '
                                     'sharded over the GPUs, no collective; steady state of batch-after-batch sampling: the '
                                     'step graph was captured by the preceding, untimed batch of the same shape)' % a.sampling_batch,
                           'value': round(a.sampling_batch * world / ds, 2), 'unit': 'images/s', 'n_gpus': world,
                           'seconds': round(ds, 3),
                           # the batch before it: the first of its shape in the process (step-graph capture + first-touch costs inside;
                           # rounds <= 4 reported THIS kind of number -- compare like with like)
                           'first_batch': {'value': round(a.sampling_batch * world / ds_first, 2), 'seconds': round(ds_first, 3),
                                           'note': 'first batch of this shape (capture of the step graph included; this rank)'},
                           # inner steps replayed from a captured step (98 of the 100: first and last run eagerly)?
                           'graphed': big_stats['replays'] > 0 and proc.graph_stats['fallback'] == 0,
                           'graph_stats': {'warmup_b8': small_stats, 'timed_batch': big_stats}}
        if proc.graph_stats['fallback']:
            # a capture the sampler expected to work fell back to eager stepping: a product defect, not a number to report
            print('bench: the sampler\'s step capture FAILED (%r)' % (proc.graph_stats,), file=sys.stderr)
            sampler_capture_failed = True
        del proc, smodel
        torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not a.no_large_batch:
        out['large_batch'] = large_batch_rate(a, margs, dev)
        if a.dtype == 'bf16':
            # the same step in the reference's own arithmetic (fp32 activations, exact-f32 MFMA): supplementary
            import copy
            fa = copy.copy(margs)
            fa.act_dtype = 'fp32'
            r = large_batch_rate(a, fa, dev, batch=a.batch, steps=5)
            out['fp32_value'] = {'value': r['value'], 'unit': 'images/s', 'ms_per_step': r['ms_per_step'],
                                 'note': 'fp32 activations / weights (1e-4 parity path), B=%d, graph replay' % a.batch}
            # ... and with fp32 atomics where the default (round 5: bit-reproducible steps, as the reference's seed_everything asks of
            # cuDNN) uses ordered reductions: the A/B of ops.set_deterministic
            from infodiffusion_amd import ops as _ops
            was = _ops._WGRAD_DET
            _ops.set_deterministic(False)
            try:
                r = large_batch_rate(a, margs, dev, batch=a.batch, steps=20)
            finally:
                _ops.set_deterministic(was)
            out['atomics_value'] = {'value': r['value'], 'unit': 'images/s', 'ms_per_step': r['ms_per_step'],
                                    'note': 'same step with fp32 atomics for the weight gradients and the GroupNorm parameter gradients '
                                            '(ops.set_deterministic(False)); `value` is the bit-reproducible default, B=%d, graph replay' % a.batch}
    if rank == 0 and world == 1 and sync is None and not a.no_dp_probe:
        out['dp_one_rank'] = one_rank_dp_cost(a)
        if 'ms_per_step' in out['dp_one_rank']:
            out['dp_one_rank']['delta_ms_vs_value'] = round(out['dp_one_rank']['ms_per_step'] - out['ms_per_step'], 3)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(margs)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    # workgroups of the synchronised data-gradient conv that ever gave up waiting for their group (a grid that was not resident at once):
    # a launch with a time-out computed garbage, so the run is void
    from infodiffusion_amd import ops as _ops
    sync_timeouts = _ops.rs_sync_timeouts(False) + trainer.timeouts      # (the trainer's health check zeroes the word when it rolls back)
    out['config']['rs_sync_timeouts'] = sync_timeouts
    if rank == 0:
        # the LAST line of stdout: RCCL printf()s its version banner into the C library's stdout buffer, which would
        # otherwise be flushed at exit, after this line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if sampler_capture_failed:
        sys.exit(3)
    if sync_timeouts:
        print('idf_conv_rs_dgrad_gn_bf16: %d workgroup(s) timed out waiting for their group' % sync_timeouts, file=sys.stderr)
        sys.exit(4)


if __name__ == '__main__':
    main()
