#!/usr/bin/env python3
"""Conditioning path (idf_temb_film_fwd / _bwd) timed back to back: CelebA sizes (dim 256, d_model 64, a_dim 32,
Nt = Na = 4992, B = 32).  Usage: python tools/bench_temb_film.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infodiffusion_amd import ops

DEV = 'cuda'
B, T, d_model, dim, a_dim, Nt = 32, 1000, 64, 256, 32, 4992
mk = lambda *s: torch.randn(*s, device=DEV) * 0.05
table, t = mk(T, d_model), torch.randint(0, T, (B,), device=DEV)
P = [mk(dim, d_model), mk(dim), mk(dim, dim), mk(dim), mk(dim, a_dim), mk(dim), mk(Nt, dim), mk(Nt), mk(Nt, dim), mk(Nt)]
for p in P:
    p.requires_grad_(True)
a = mk(B, a_dim).requires_grad_(True)
gt, ga = mk(B, Nt), mk(B, Nt)


def fwd():
    return ops._TembFilm.apply(t, a, table, *P, False, None)


def timed(fn, n=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    g = torch.cuda.CUDAGraph()
    fwd()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(10):
            fwd()
    print('forward (3 launches), replayed graph of 10: %.1f us per call' % (timed(g.replay, 50) / 10))


def fb():
    ft, fa = fwd()
    torch.autograd.backward([ft, fa], [gt, ga])


fb()
torch.cuda.synchronize()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    for _ in range(10):
        fb()
print('forward + backward (3 + 4 launches + autograd glue), replayed graph of 10: %.1f us per call' % (timed(g2.replay, 50) / 10))

# diagnostic build only (IDF_LIB=.../libinfodiff_hip_gmstamp.so): phase cycle sums of the GEMM blocks
import ctypes
from infodiffusion_amd import _lib
lib = _lib.load()
if hasattr(lib, 'idf_debug_gm_stamps'):
    lib.idf_debug_gm_stamps.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    addr = ctypes.c_void_p()
    assert lib.idf_debug_gm_stamps(ctypes.byref(addr)) == 0
    hip = ctypes.CDLL('libamdhip64.so')
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    raw = (ctypes.c_ulonglong * 8)()
    zero = (ctypes.c_ulonglong * 8)()
    torch.cuda.synchronize()
    hip.hipMemcpy(addr, zero, 64, 1)
    with torch.no_grad():
        for _ in range(20):
            fwd()
    torch.cuda.synchronize()
    hip.hipMemcpy(raw, addr, 64, 2)
    nb = max(1, raw[5])
    for i, n in enumerate(['job lookup', 'wait loads + commit + barrier', 'issue next + MFMA + barrier', 'epilogue', 'whole block']):
        print('  %-32s %8.0f ticks per block (100 MHz s_memtime: x10 ns)' % (n, raw[i] / nb))
    print('  blocks', nb)
