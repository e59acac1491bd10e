#!/usr/bin/env python3
"""One conv shape launched repeatedly (for rocprofv3 --pmc passes). Usage: one_conv.py Cin Cout H [B] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infodiffusion_amd import ops

Cin, Cout, H = (int(v) for v in sys.argv[1:4])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 30
x = torch.randn(B, Cin, H, H, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
w = torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.05
wf, _ = ops.pack_weight(w, torch.bfloat16, True, False)
bias = torch.zeros(Cout, device='cuda')
for _ in range(reps):
    y = ops.conv_raw(x, wf, bias, None, None, None, None, 0, 0.0, ops.S1, 9, 0, Cout)
torch.cuda.synchronize()
