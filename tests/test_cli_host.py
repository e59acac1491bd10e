"""CPU: host-side pieces of the CLI (flag set, dataset config, exp string, schedulers)."""
import types

import torch


def test_cli_flags_match_reference_defaults():
    import run
    a = run.parse_args(['--model', 'diff', '--mode', 'train', '--prior', 'regular', '--dataset', 'celeba',
                        '--a_dim', '32', '--save_epoch', '7'])       # prefix matching as eval_fid.sh:9 relies on
    assert (a.mmd_weight, a.kld_weight, a.beta1, a.betaT, a.diffusion_steps) == (0.1, 0, 1e-5, 1e-2, 1000)
    assert a.save_epochs == 7 and a.batch_size == 64 and a.learning_rate == 1e-4 and a.split_step == 500
    assert not a.deterministic and not a.is_latent and a.act_dtype == 'bf16'      # our extra flag: fast path by default


def test_dataset_config_and_exp_string():
    from infodiffusion_amd.data import get_dataset_config
    from infodiffusion_amd.utils import generate_exp_string
    a = types.SimpleNamespace(dataset='fmnist')
    assert get_dataset_config(a) == (1, 32, 32) and a.unets_channels == 32      # 28 -> 32 (data.py:64-68)
    a = types.SimpleNamespace(dataset='celeba')
    assert get_dataset_config(a) == (3, 64, 64) and a.encoder_channels == 64
    a = types.SimpleNamespace(dataset='celeba', a_dim=32, kld_weight=0, use_C=False, C_max=25, mmd_weight=0.1,
                              prior='regular', is_bottleneck=False)
    assert generate_exp_string(a) == 'celeba_32d_0.1mmd'
    a.kld_weight, a.use_C, a.prior = 0.01, True, 'roll'
    assert generate_exp_string(a) == 'celeba_32d_0.01kld_25C_0.1mmd_roll'


def test_warmup_cosine_schedule():
    from infodiffusion_amd.utils import GradualWarmupScheduler
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-4)
    cos = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10, eta_min=0, last_epoch=-1)
    w = GradualWarmupScheduler(opt, multiplier=2., warm_epoch=1, after_scheduler=cos)
    lrs = []
    for _ in range(5):
        lrs.append(opt.param_groups[0]['lr'])
        opt.step()
        w.step()
    # the sequence the reference's scheduler pair produces (utils.py:133-160 + CosineAnnealingLR),
    # measured by running the reference class in the build container -- overshoot at epoch 2 included
    want = [1e-4, 2e-4, 0.00020501712618738332, 2e-4, 0.00018543973270544474]
    assert all(abs(a - b) < 1e-12 for a, b in zip(lrs, want)), lrs


def test_synthetic_batches_are_normalised_and_sharded():
    from infodiffusion_amd.data import get_dataset
    a = types.SimpleNamespace(dataset='celeba', batch_size=4, steps_per_epoch=3, data_dir='/nonexistent', r_seed=1)
    b0 = [x for x, _ in get_dataset(a, (3, 8, 8), 'cpu', 0, 2)]
    b1 = [x for x, _ in get_dataset(a, (3, 8, 8), 'cpu', 1, 2)]
    assert len(b0) == 3 and b0[0].shape == (4, 3, 8, 8)
    assert float(b0[0].min()) >= -1 and float(b0[0].max()) <= 1
    assert not torch.equal(b0[0], b1[0])


def test_flip_augmentation_follows_the_reference_transforms(tmp_path):
    """RandomHorizontalFlip is in the reference's fmnist / celeba / cifar10 / chairs / ffhq transforms (data.py:138, 166, 191,
    224, 237) and absent from mnist / dsprites (:125, :203-207); float arrays and CPU batches are flipped like uint8 ones."""
    import numpy as np
    from infodiffusion_amd.data import get_dataset
    n = 64
    ramp = np.tile(np.linspace(0.0, 1.0, 8, dtype=np.float32), (n, 1, 8, 1))        # every row of every image rises left -> right
    for ds, flips in (('fmnist', True), ('cifar10', True), ('chairs', True), ('celeba', True), ('ffhq', True),
                      ('mnist', False), ('dsprites', False)):
        ch = 3 if ds in ('cifar10', 'chairs', 'celeba', 'ffhq') else 1
        np.save(tmp_path / ('%s.npy' % ds), np.repeat(ramp, ch, axis=1))
        a = types.SimpleNamespace(dataset=ds, batch_size=n, steps_per_epoch=1, data_dir=str(tmp_path), r_seed=3, mode='train')
        (x, _), = list(get_dataset(a, (ch, 8, 8), 'cpu'))
        rising = (x[:, 0, 0, -1] > x[:, 0, 0, 0])
        assert x.shape == (n, ch, 8, 8) and float(x.min()) == -1.0 and float(x.max()) == 1.0
        if flips:
            assert 8 < int(rising.sum()) < n - 8, (ds, int(rising.sum()))          # about half the images are mirrored
        else:
            assert bool(rising.all()), ds


def test_save_original_img_mode_runs_without_a_gpu(tmp_path):
    """--mode save_original_img (reference run.py:540-549): dataset images in [0, 1], no model involved."""
    import glob
    import os
    import subprocess
    import sys

    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(tmp_path / 'data')
    rng = np.random.default_rng(0)
    raw = rng.integers(0, 256, (8, 32, 32, 1), dtype=np.uint8)
    np.save(tmp_path / 'data' / 'fmnist.npy', raw)
    r = subprocess.run([sys.executable, os.path.join(root, 'run.py'), '--model', 'diff', '--mode', 'save_original_img',
                        '--prior', 'regular', '--dataset', 'fmnist', '--a_dim', '32', '--batch_size', '4',
                        '--data_dir', str(tmp_path / 'data')], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    files = sorted(glob.glob(str(tmp_path / 'fmnist_imgs' / '*.npy')))
    assert len(files) == 2
    img = np.load(files[0])
    assert img.shape == (4, 1, 32, 32)
    # whatever the order (the reference's fmnist loader does not shuffle, data.py:144; cifar10 / chairs / dsprites do):
    # each saved image is one of the source images, every source image is saved exactly once
    allimgs = np.concatenate([np.load(f) for f in files])
    src = raw.transpose(0, 3, 1, 2) / 255.0
    # ... possibly mirrored: the reference's fmnist transform carries RandomHorizontalFlip (data.py:138) in every mode
    match = [int(np.argmin([min(np.abs(im - s_).max(), np.abs(im[..., ::-1] - s_).max()) for s_ in src])) for im in allimgs]
    assert sorted(match) == list(range(8))
    assert all(np.allclose(im, src[j], atol=1e-6) or np.allclose(im[..., ::-1], src[j], atol=1e-6) for im, j in zip(allimgs, match))


def test_latent_dataset_reads_a_reference_written_archive():
    """The wire format between the two phases of config 5 (`{model}_{exp}_latent.npz`, /root/reference/run.py:415-443 ->
    utils.py:163-172): tests/golden/ref_diff_latent.npz was WRITTEN by the reference's statements on the reference's own encoder
    (tools/gen_golden.py latent_archive) and read back there through the reference's LatentDataset -- the product's LatentDataset
    must deliver the same rows (values, order, dtype, length), and the oracle's encoder must reproduce them from the inputs."""
    import os
    import numpy as np
    from infodiffusion_amd.utils import LatentDataset
    from oracle import infodiff_oracle as O
    from tests.helpers import GOLD, gold, manifest
    ds = LatentDataset(os.path.join(GOLD, 'ref_diff_latent.npz'))
    g = gold('ref_diff_latent_inputs')
    assert len(ds) == int(g['n']) == g['rows'].shape[0] == 15
    rows = torch.stack([ds[i] for i in range(len(ds))])
    assert rows.dtype == torch.float32 and torch.equal(rows, g['rows'])
    with np.load(os.path.join(GOLD, 'ref_diff_latent.npz')) as z:          # the archive also carries the labels, one per row
        assert sorted(z.files) == ['all_a', 'all_attr'] and z['all_attr'].shape == (15,)
        assert np.array_equal(z['all_attr'], g['attr'].numpy())
    cfg = O.dataset_cfg('fmnist', a_dim=32, mmd_weight=0.1)
    sd = O.synth_state_dict(manifest('manifest_fmnist'))
    with torch.no_grad():
        a = O.encoder(sd, 'encoder', g['x'], cfg.encoder_channels, O.ch_mult_for(cfg))[0]
    assert float((a - rows).abs().max()) <= 2e-5 * float(rows.abs().max())
