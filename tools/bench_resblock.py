#!/usr/bin/env python3
"""Forward time of one ResBlock at 8x8 (B = 32, bf16, training outputs on): the image-resident launch (idf_resblock_small_fwd)
against the per-op launches of the same module, graph-replayed.  usage: tools/bench_resblock.py [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infodiffusion_amd import modules, ops

DEV, CL = 'cuda', torch.channels_last
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def make(kind, cin):
    torch.manual_seed(0)
    blk = (modules.AuxResBlock(cin, 128, tdim=256, dropout=0.1) if kind == 'aux' else modules.ResBlock_encoder(cin, 128, dropout=0.1)).to(DEV)
    for m in blk.modules():
        if isinstance(m, torch.nn.Conv2d) and m.kernel_size != (1, 1):
            m.weight.data = m.weight.data.contiguous(memory_format=CL)
    blk.ctx.act_dtype = torch.bfloat16
    blk.ctx.seed = torch.tensor([7], dtype=torch.int64, device=DEV)
    return blk.train()


def timed(fn, reps=20):
    fn()
    fn()       # (a stand-alone block's first call asks for the fragment-major layouts, its second packs them: both outside the capture)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


def main():
    for kind, cin in (('aux', 128), ('aux', 256), ('enc', 128), ('enc', 256)):
        blk = make(kind, cin)
        x1 = torch.randn(B, 128, 8, 8, device=DEV).bfloat16().contiguous(memory_format=CL).requires_grad_(True)
        x2 = torch.randn(B, cin - 128, 8, 8, device=DEV).bfloat16().contiguous(memory_format=CL).requires_grad_(True) if cin > 128 else None
        x1._gn = ops.gn_partials_raw(x1)
        if x2 is not None:
            x2._gn = ops.gn_partials_raw(x2)
        ft = torch.randn(B, 256, device=DEV) * 0.1
        fa = torch.randn(B, 256, device=DEV) * 0.1
        xin = (x1, x2) if x2 is not None else x1

        def fwd():
            if kind == 'aux':
                blk._film = {'t': ft, 'a': fa}
                return blk(xin, None, None)
            return blk(xin)
        res = []
        for fused in (False, True):
            ops._RB_SMALL = fused
            res.append(timed(fwd))
        with torch.no_grad():                      # inference: no a / h / coefficient outputs
            for fused in (False, True):
                ops._RB_SMALL = fused
                res.append(timed(fwd))
        print('%s Cin %d B %d: per-op %.1f us   fused %.1f us   | no_grad: per-op %.1f us   fused %.1f us' % (kind, cin, B, *res))

    # diagnostic build (tools/build_variant.sh rbstamp idf_resblock.hip -DIDF_RB_STAMP, IDF_LIB=...): phase shares of the fused kernel
    import ctypes
    from infodiffusion_amd import _lib
    lib = _lib.load()
    if hasattr(lib, 'idf_debug_rb_stamps'):
        lib.idf_debug_rb_stamps.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        addr = ctypes.c_void_p()
        assert lib.idf_debug_rb_stamps(ctypes.byref(addr)) == 0
        hip = ctypes.CDLL('libamdhip64.so')
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        ops._RB_SMALL = True
        for kind, cin in (('aux', 128), ('aux', 256), ('enc', 128)):
            blk = make(kind, cin)
            x1 = torch.randn(B, 128, 8, 8, device=DEV).bfloat16().contiguous(memory_format=CL).requires_grad_(True)
            x2 = torch.randn(B, cin - 128, 8, 8, device=DEV).bfloat16().contiguous(memory_format=CL).requires_grad_(True) if cin > 128 else None
            x1._gn = ops.gn_partials_raw(x1)
            if x2 is not None:
                x2._gn = ops.gn_partials_raw(x2)
            ft = torch.randn(B, 256, device=DEV) * 0.1
            xin = (x1, x2) if x2 is not None else x1

            def fwd():
                if kind == 'aux':
                    blk._film = {'t': ft, 'a': ft}
                    return blk(xin, None, None)
                return blk(xin)
            for _ in range(3):
                fwd()
            torch.cuda.synchronize()
            hip.hipMemset(addr, 0, 64)
            for _ in range(10):
                fwd()
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * 8)()
            hip.hipMemcpy(buf, addr, 64, 2)
            n = buf[7]
            print('%s Cin %d stamps per block (ticks): stage0 %d | conv %s | epilogue %s' % (
                kind, cin, buf[0] // n, [buf[1 + 2 * k] // n for k in range(3)], [buf[2 + 2 * k] // n for k in range(3)]))


if __name__ == '__main__':
    main()
