#!/usr/bin/env python3
"""Phase shares of the direct-to-LDS conv (diagnostic build: tools/build_variant.sh stamp idf_conv3x3.hip -DIDF_DLDS_STAMP,
run with IDF_LIB=infodiffusion_amd/variants/libinfodiff_hip_stamp.so): one GroupNorm-prologue (or plain) conv shape run
repeatedly; prints the average cycles per block spent in prologue / load wait / transform / MFMA / epilogue.
Usage: dlds_stamps.py B Cin Cout H pro"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infodiffusion_amd import _lib, ops

B, Cin, Cout, H, pro = [int(v) for v in sys.argv[1:6]]
DEV, CL = 'cuda', torch.channels_last
lib = _lib.load()
lib.idf_debug_dlds_stamps.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
w = torch.randn(Cout, Cin, 3, 3, device=DEV) / (9 * Cin) ** 0.5
wf, _ = ops.pack_weight(w, torch.bfloat16, True, False)
g, bt = torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV)
xs = [torch.randn(B, Cin, H, H, device=DEV).bfloat16().contiguous(memory_format=CL) for _ in range(4)]
sts = [ops.gn_partials_raw(x) for x in xs]


def run(i):
    x = xs[i % 4]
    if pro:
        ops.conv_gn_raw(x, None, sts[i % 4], None, g, bt, None, None, None, 3, 0.0, 2, wf, None, None, Cout, 9, want_stats=True)
    else:
        ops.conv_raw(x, wf, None, None, None, None, None, 0, 0.0, ops.S1, 9, 0, Cout, want_stats=True)


for i in range(4):
    run(i)
torch.cuda.synchronize()
_addr = ctypes.c_void_p()
assert lib.idf_debug_dlds_stamps(ctypes.byref(_addr)) == 0
_hip = ctypes.CDLL('libamdhip64.so')
_hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
_raw = (ctypes.c_ulonglong * 512)()


def stamps(reset):
    """Sum of the 64 shards; optionally clear them."""
    assert _hip.hipMemcpy(_raw, _addr, 4096, 2) == 0                  # device to host
    out = [sum(_raw[s * 8 + i] for s in range(64)) for i in range(8)]
    if reset:
        zero = (ctypes.c_ulonglong * 512)()
        assert _hip.hipMemcpy(_addr, zero, 4096, 1) == 0              # host to device
    return out


stamps(1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N = 20
for i in range(N):
    run(i)
e1.record()
torch.cuda.synchronize()
buf = stamps(0)
nb = max(1, buf[5])
names = ['prologue', 'load wait', 'transform', 'mfma', 'epilogue']
tot = sum(buf[i] for i in range(5))
print('B %d %d->%d @%dx%d pro %d: %.1f us per launch (stamped build), %d blocks per launch' % (B, Cin, Cout, H, H, pro, e0.elapsed_time(e1) / N * 1e3, nb // N))
for i, n in enumerate(names):
    print('  %-10s %8.0f ticks per block  %5.1f %%' % (n, buf[i] / nb, 100.0 * buf[i] / tot))
print('  sum        %8.0f ticks per block (s_memtime ticks; 100 MHz counter x clock ratio -- shares matter, not totals)' % (tot / nb))
