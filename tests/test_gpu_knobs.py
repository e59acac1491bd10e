"""Every surviving environment switch (infodiffusion_amd/knobs.py + the library's table in csrc/idf_capi.hip) at its NON-default value:
the switches are read once per process, so each case runs a reference-pinned test in a child interpreter with the switch set --
the CelebA bf16 training step against the reference fixture (loss, gradient norm, every named gradient, epsilon-hat) for the
kernel-path switches, the bf16 graphed-sampler test for the sampler's.  A switch nobody exercises is a path nobody has run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEP = 'tests/test_gpu_model.py::test_bf16_train_step_celeba[32]'
SAMPLER = 'tests/test_gpu_model.py::test_graphed_sampler_captures_in_bf16_on_celeba'
TRACES = 'tests/test_gpu_model.py::test_samplers_real_model'

CASES = [
    # kernel-path switches of the package
    ('IDF_CONV_RS', '0', STEP), ('IDF_CONV_RS', '2', STEP), ('IDF_CONV_RS_FWD', '1', STEP), ('IDF_CONV_WR', '0', STEP),
    ('IDF_RB_SMALL', '0', STEP), ('IDF_RB_SMALL_MAXB', '1', STEP), ('IDF_GN_FUSE', '0', STEP), ('IDF_BWD_CHAIN', '0', STEP),
    ('IDF_BWD_LAZY', '1', STEP), ('IDF_DGRAD_GN', '0', STEP), ('IDF_SC_FUSE', '0', STEP), ('IDF_WGRAD_BATCH', '0', STEP), ('IDF_DETERMINISTIC', '0', STEP),
    ('IDF_ATTN_FOLD', '0', STEP), ('IDF_ATTN_BLOCK_MINB', '1', STEP), ('IDF_UPCONV', '0', STEP), ('IDF_TEMB_FUSED', '0', STEP),
    # the library's own
    ('IDF_CONV_RS_SYNC', '0', STEP), ('IDF_CONV_PS', '0', STEP), ('IDF_CONV_DLDS_MIN', '1', STEP), ('IDF_WGRAD_KR3', '0', STEP), ('IDF_WGRAD_TPB3', '32', STEP), ('IDF_WGRAD_RING', '0', STEP),
    # the samplers'
    ('IDF_SAMPLER_GRAPH', '0', TRACES), ('IDF_TRAJ_CACHE', '0', TRACES), ('IDF_SAMPLER_GRAPH_MAXPIX', '1', TRACES), ('IDF_SAMPLER_GRAPH_STRICT', '1', SAMPLER),
]


def test_every_switch_is_listed():
    """The table in knobs.py, the library's table and this file's cases name the same switches (IDF_LIB / IDF_FORCE_SYNC /
    IDF_CPU_THREADS select a library, a bench mode and a thread count: nothing to run)."""
    from infodiffusion_amd import knobs
    lib = {'IDF_CONV_RS', 'IDF_CONV_RS_SYNC', 'IDF_CONV_PS', 'IDF_CONV_DLDS_MIN', 'IDF_WGRAD_KR3', 'IDF_WGRAD_TPB3', 'IDF_WGRAD_RING'}
    capi = open(os.path.join(ROOT, 'infodiffusion_amd', 'csrc', 'idf_capi.hip')).read()
    assert all('"%s"' % n in capi for n in lib)
    covered = {c[0] for c in CASES} | {'IDF_LIB', 'IDF_FORCE_SYNC', 'IDF_CPU_THREADS'}
    assert covered == set(knobs.TABLE) | lib, sorted(covered ^ (set(knobs.TABLE) | lib))
    assert len(set(knobs.TABLE) | lib) <= 31
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    assert all('`%s`' % n in doc for n in set(knobs.TABLE) | lib), [n for n in set(knobs.TABLE) | lib if '`%s`' % n not in doc]
    # ... and nothing else in the product reads an IDF_* name from the environment
    import glob
    import re
    for f in glob.glob(os.path.join(ROOT, 'infodiffusion_amd', 'csrc', '*')):
        if not f.endswith('idf_capi.hip'):
            assert 'getenv' not in open(f).read(), f
    for f in glob.glob(os.path.join(ROOT, 'infodiffusion_amd', '*.py')):
        if not f.endswith(('knobs.py', 'build.py')):
            assert not re.search(r'environ[^\n]*IDF_', open(f).read()), f


@pytest.mark.gpu
@pytest.mark.parametrize('name,value,test', CASES)
def test_switch_at_its_non_default_value(name, value, test):
    env = dict(os.environ)
    env[name] = value
    env['IDF_TEST_NONDEFAULT'] = '1'       # test_gpu_model.py: the non-default paths keep the looser epsilon-hat bound
    r = subprocess.run([sys.executable, '-m', 'pytest', test, '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider'], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, '%s=%s: %s' % (name, value, (r.stdout + r.stderr)[-3000:])
