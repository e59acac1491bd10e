#!/usr/bin/env python3
"""Phase shares of the register-weights / row-reuse conv (diagnostic build: tools/build_variant.sh rsstamp idf_conv_rs.hip
-DIDF_RS_STAMP, run with IDF_LIB=infodiffusion_amd/variants/libinfodiff_hip_rsstamp.so): one shape run repeatedly on rotating
buffers; prints the average s_memtime ticks per workgroup (wave 0) in each phase.
Usage: rs_stamps.py B Cin Cout H kind      kind: fwd (GroupNorm-prologue forward, training outputs) | bwd (du-epilogue dgrad)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infodiffusion_amd import _lib, ops
from tools.bench_rs import Frag

B, Cin, Cout, H = [int(v) for v in sys.argv[1:5]]
kind = sys.argv[5]
DEV, CL = 'cuda', torch.channels_last
lib = _lib.load()
lib.idf_debug_rs_stamps.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
NSET = 8
seed = torch.tensor([1234567], dtype=torch.int64, device=DEV)
if kind == 'fwd':
    sh = Frag(torch.randn(Cout, Cin, 3, 3, device=DEV) / (9 * Cin) ** 0.5)
    gam, bet, bias = torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV), torch.zeros(Cout, device=DEV)
    ft, fa = 0.1 * torch.randn(B, 2 * Cin, device=DEV), 0.1 * torch.randn(B, 2 * Cin, device=DEV)
    xs = [torch.randn(B, Cin, H, H, device=DEV).bfloat16().contiguous(memory_format=CL) for _ in range(NSET)]
    sts = [ops.conv_raw(torch.randn(B, 64, H, H, device=DEV).bfloat16().contiguous(memory_format=CL),
                        ops.pack_weight(torch.randn(Cin, 64, 3, 3, device=DEV) / 24, torch.bfloat16, True, False)[0], None, None, None,
                        None, None, 0, 0.0, ops.S1, 9, 0, Cin, want_stats=True) for _ in range(NSET)]
    xs = [s[0] for s in sts]
    sts = [s[1] for s in sts]
    assert sts[0].shape[1] <= 16, sts[0].shape

    def run(i):
        ops.conv_gn_raw(xs[i % NSET], None, sts[i % NSET], None, gam, bet, ft, fa, seed, 7, 0.1, 2, sh.val[0], bias, None, Cout, 9,
                        keep_a=True, keep_coef=True, want_stats=True, shadows=sh)
else:
    sh = Frag(torch.randn(Cin, Cout, 3, 3, device=DEV) / (9 * Cout) ** 0.5)
    xs = [torch.randn(B, Cout, H, H, device=DEV).bfloat16().contiguous(memory_format=CL) for _ in range(NSET)]
    dys = [torch.randn(B, Cin, H, H, device=DEV).bfloat16().contiguous(memory_format=CL) for _ in range(NSET)]
    sc, shf = torch.randn(B, Cout, device=DEV), torch.randn(B, Cout, device=DEV)

    def run(i):
        ops.conv_dgrad_chain_raw(dys[i % NSET], sh.val[1], 9, Cout, x=xs[i % NSET], sc=sc, sh=shf, seed=seed, salt=3, p_drop=0.1, act=2,
                                 shadows=sh)

for i in range(NSET):
    run(i)
torch.cuda.synchronize()
_addr = ctypes.c_void_p()
assert lib.idf_debug_rs_stamps(ctypes.byref(_addr)) == 0
_hip = ctypes.CDLL('libamdhip64.so')
_hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
_raw = (ctypes.c_ulonglong * 1024)()


def stamps(reset):
    assert _hip.hipMemcpy(_raw, _addr, 8192, 2) == 0
    out = [sum(_raw[s * 16 + i] for s in range(64)) for i in range(16)]
    if reset:
        zero = (ctypes.c_ulonglong * 1024)()
        assert _hip.hipMemcpy(_addr, zero, 8192, 1) == 0
    return out


stamps(1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 40
e0.record()
for i in range(N):
    run(i)
e1.record()
torch.cuda.synchronize()
buf = stamps(0)
nwg, ntile = max(1, buf[12]), max(1, buf[11])
names = ['issue (plan, fold + row loads)', 'fold', 'rows landed + transform + ds_write', 'first barrier', 'tile: issue x / next rows',
         'tile: MFMA loop', 'tile: barrier 1', 'tile: acc -> LDS, next rows -> image', 'tile: barrier 2', 'tile: epilogue tail']
print('%s B %d %d->%d @%d: %.1f us per launch (stamped build, eager), %d workgroups, %.2f tiles each' % (
    kind, B, Cin, Cout, H, e0.elapsed_time(e1) / N * 1e3, nwg // N, ntile / nwg))
for i, n in enumerate(names):
    print('  %-40s %8.0f ticks per workgroup' % (n, buf[i] / nwg))
print('  %-40s %8.0f ticks per workgroup (s_memtime ticks = shader cycles)' % ('workgroup life', buf[10] / nwg))
