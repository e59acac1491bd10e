"""CPU: the C-ABI library builds, loads, and exports every symbol include/infodiff_hip.h
declares (no compute calls without a GPU); the product refuses to run without it."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'infodiff_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(idf_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from infodiffusion_amd import build, _lib
    build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), 'missing export: ' + n
    # the ctypes table binds exactly the declared entry points
    assert sorted(_lib.SIGNATURES) == names
    _lib.load()
    assert _lib.load().idf_version() >= 100
    assert _lib.load().idf_last_error() is not None


def test_error_channel_without_gpu():
    from infodiffusion_amd import _lib
    lib = _lib.load()
    rc = lib.idf_sampler_step(None, None, None, None, None, None, None, 7, 0, 0, None)   # bad mode: rejected before any launch
    assert rc == _lib.ERR_BADARG
    assert b'bad mode' in lib.idf_last_error()
    with pytest.raises(_lib.HipKernelError):
        _lib.check(rc, 'idf_sampler_step')


def test_missing_library_fails_loudly(monkeypatch):
    from infodiffusion_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libinfodiff_hip.so')
    with pytest.raises(ImportError):
        _lib.load()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'infodiffusion_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            assert 'oracle' not in open(os.path.join(pkg, fn)).read().replace('no oracle', ''), fn


def test_state_dict_matches_reference_manifest():
    import json
    import types
    import torch
    from infodiffusion_amd.models import InfoDiff
    args = types.SimpleNamespace(beta1=1e-5, betaT=1e-2, diffusion_steps=1000, input_size=64, is_bottleneck=False,
                                 unets_channels=64, encoder_channels=64, a_dim=32, mmd_weight=0.1, kld_weight=0.0)
    m = InfoDiff(args, 'cpu', (3, 64, 64))
    man = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_celeba.json')))
    sd = m.state_dict()
    assert [k for k, _ in man] == list(sd.keys())
    assert all(tuple(s) == tuple(sd[k].shape) for k, s in man)
    assert len(sd) == 969
