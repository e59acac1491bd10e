"""NN blocks of the AVDM UNet on the HIP kernels (mirror of the reference's
modules.py: same class names, constructor arguments, module tree and therefore
state_dict keys; forward passes are re-designed around fused NHWC kernels).

Stock `nn.Conv2d / nn.GroupNorm / nn.Linear / nn.Embedding` objects are kept only
as *parameter holders* (so checkpoints interchange with the reference and the
initial weights under a seed are identical); their own forward is never called.
Every block computes

    conv( dropout( SiLU( GroupNorm(x) [* (1+s_t) + b_t] [* (1+s_a) + b_a] ) ) ) + bias [+ residual]

as ONE conv launch whose input staging applies the per-(sample, channel) affine
that `idf_gn_coef_fwd` folds from the GroupNorm statistics and the FiLM pairs.
"""
import math
import types
from typing import Union

import torch
import torch.nn as nn
from torch.nn import init

from . import ops

_ACT_NONE, _ACT_AFFINE, _ACT_SILU = 0, 1, 2


class RunCtx(types.SimpleNamespace):
    """Per-network run state shared by reference (not a Module): the activation
    dtype and the current step's dropout seed (1-element int64 CUDA tensor or None)."""

    def __init__(self, act_dtype=torch.float32):
        super().__init__(act_dtype=act_dtype, seed=None)


class _Shadows:
    """Kernel-layout copies of conv weights in persistent buffers (fixed addresses, so captured
    graphs stay valid), refreshed when a master weight changes (optimizer step / load_state_dict
    bump `_version`) -- normally by the owning network's ONE batched launch (`ShadowSet.refresh`)."""

    def __init__(self, *convs):
        self.convs = convs
        self.key = None
        self.owners = []            # the ShadowSets that re-pack this shadow: told when its layout state changes mid-forward
        self.val = [None, None, None, None]      # forward, data-gradient, [fragment-major forward / data-gradient: want_frag]
        # the image-resident ResBlock kernel (ops.resblock_small) reads its weights fragment-major; a conv it has met once
        # gets that third shadow from the next re-pack on
        self.want_frag = False
        # UpSample's conv: the summed sub-pixel weights of idf_upconv_bf16 (bf16 only), re-packed with the other shadows
        self.want_sub = False
        self.sub = None
        self.subd = None            # ... and their data-gradient form (rows = cins), idf_upconv_dgrad_bf16
        # AttnBlock with the proj conv folded into V: the shadows are packed from (Wq, Wk, Wv' = Wp Wv) -- `srcs` replaces the
        # convs' own weights as the pack sources, `fold` = (wp, bp, wv, bv, bq, bk, wvf, bf) feeds idf_attn_fold_batched first
        self.srcs = None
        self.fold = None
        # several convs applied as one (q, k, v): their weights / biases live adjacently so the
        # concatenation is a view and its gradient is written once (grad_arena.ParamGroup)
        self.wgroup = ops.ParamGroup([c.weight for c in convs]) if len(convs) > 1 else None
        self.bgroup = ops.ParamGroup([c.bias for c in convs]) if len(convs) > 1 else None

    def weight(self):
        if len(self.convs) == 1:
            return self.convs[0].weight
        return ops.cat_params(self.wgroup)

    def bias(self):
        if len(self.convs) == 1:
            return self.convs[0].bias
        return ops.cat_params(self.bgroup)

    def current_key(self, dtype):
        key = tuple((c.weight.data_ptr(), c.weight._version) for c in self.convs) + (dtype,)
        if self.fold is not None:          # the folded weights follow proj's parameters and the biases too
            key += tuple((t.data_ptr(), t._version) for t in self.fold[:6])
        return key

    def pack_sources(self):
        """The fp32 weight tensors the shadows are packed from (the convs' own, unless a fold replaces some)."""
        return self.srcs if self.srcs is not None else [c.weight for c in self.convs]

    def run_fold(self):
        """Stand-alone use (a block outside a network's ShadowSet, which folds all its blocks in one launch): the folded V weights /
        biases by the same kernel on a one-row table -- no stock-library product anywhere in the product path."""
        import numpy as np
        from ._lib import call
        key = tuple(t.data_ptr() for t in self.fold)
        if getattr(self, '_fold_tab_key', None) != key:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('attention fold: run one eager forward before graph capture')
            dt = np.dtype([('p%d' % i, '<i8') for i in range(8)] + [('C', '<i4'), ('pad', '<i4')])
            row = np.array([key + (self.fold[0].shape[0], 0)], dtype=dt)
            self._fold_tab = torch.from_numpy(row.view(np.uint8).reshape(1, -1).copy()).to(self.fold[0].device)
            self._fold_tab_key = key
        call('idf_attn_fold_batched', self._fold_tab.data_ptr(), 1, self.fold[0].shape[0], torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _new(shape, dtype, dev):
        if torch.cuda.is_current_stream_capturing():
            # a buffer allocated now would live in the capture's private pool and die with the graph
            raise RuntimeError('ShadowSet: run one eager forward before graph capture')
        return torch.empty(shape, dtype=dtype, device=dev)

    def ensure_buffers(self, dtype, need_dgrad):
        O = sum(c.weight.shape[0] for c in self.convs)
        _, I, kh, kw = self.convs[0].weight.shape
        dev = self.convs[0].weight.device
        for j, shape in enumerate(((O, kh * kw, I), (I, kh * kw, O))):
            if j == 1 and not need_dgrad:
                continue
            v = self.val[j]
            if v is None or v.dtype != dtype or v.device != dev:
                self.val[j] = self._new(shape, dtype, dev)
                self.key = None
        if self.want_sub and dtype == torch.bfloat16 and O % 16 == 0 and I % 64 == 0 and (kh, kw) == (3, 3) and (
                self.sub is None or self.sub.device != dev):
            self.sub = self._new((O * 16 * I,), dtype, dev)
            self.subd = self._new((O * 16 * I,), dtype, dev) if (O % 64 == 0 and I % 16 == 0) else None
            self.key = None
        if self.want_frag:
            for j, fits, on in ((2, O % 16 == 0 and I % 64 == 0, True), (3, I % 16 == 0 and O % 64 == 0, need_dgrad)):
                if fits and on and (self.val[j] is None or self.val[j].dtype != dtype or self.val[j].device != dev):
                    self.val[j] = self._new((O * kh * kw * I,), dtype, dev)
                    self.key = None

    def request_frag(self):
        """The fragment-major shadows (forward: O % 16 == 0, I % 64 == 0; data gradient: I % 16 == 0, O % 64 == 0), from the
        next re-pack on (one 3x3 conv, or the attention block's three 1x1 convs as one)."""
        w = self.convs[0].weight
        O = sum(c.weight.shape[0] for c in self.convs)
        plain = len(self.convs) == 1 and w.shape[2:] == (3, 3)
        qkv = len(self.convs) == 3 and w.shape[2:] == (1, 1)       # the attention block's q | k | v (ops.attn_block_fwd_raw)
        if not self.want_frag and (plain or qkv) and O % 16 == 0 and w.shape[1] % 16 == 0:
            self.want_frag = True
            self.key = None
            self.touch()

    def touch(self):
        """The set of shadows / pack sources changed while a forward pass was running: the owning networks re-pack at the END of
        that pass (`ShadowSet.settle`), so the next pass -- the one a sampler or trainer may capture -- finds nothing stale."""
        for o in self.owners:
            o.dirty = True

    def stale(self, dtype, need_dgrad):
        if self.want_sub and self.sub is None and dtype == torch.bfloat16:
            w = self.convs[0].weight
            if w.shape[0] % 16 == 0 and w.shape[1] % 64 == 0 and w.shape[2:] == (3, 3):
                return True
        if self.want_frag:
            O, I = sum(c.weight.shape[0] for c in self.convs), self.convs[0].weight.shape[1]
            if (O % 16 == 0 and I % 64 == 0 and self.val[2] is None) or (need_dgrad and I % 16 == 0 and O % 64 == 0
                                                                         and self.val[3] is None):
                return True
        return (self.key != self.current_key(dtype) or self.val[0] is None or self.val[0].dtype != dtype
                or (need_dgrad and self.val[1] is None))

    def __call__(self, dtype, need_dgrad):
        if self.stale(dtype, need_dgrad):        # stand-alone use (block tests): individual pack
            self.ensure_buffers(dtype, need_dgrad)
            with torch.no_grad():
                if self.fold is not None:
                    self.run_fold()
                wsrc = self.weight() if self.srcs is None else torch.cat([t.detach() for t in self.srcs], dim=0)
                wf, wd = ops.pack_weight(wsrc, dtype, True, self.val[1] is not None)
                self.val[0].copy_(wf)
                if wd is not None:
                    self.val[1].copy_(wd)
                for j, m in ((2, wf), (3, wd)):       # [N][taps][K] -> [K / 64][N / 16][tap][half][fq][fr][8]
                    if self.val[j] is not None and m is not None:
                        N, taps, K = m.shape
                        self.val[j].copy_(m.view(N // 16, 16, taps, K // 64, 2, 4, 8).permute(3, 0, 2, 4, 5, 1, 6).reshape(-1))
                if self.sub is not None:
                    self.sub.copy_(ops.upconv_pack(self.convs[0].weight))
                    if self.subd is not None:
                        self.subd.copy_(ops.upconv_pack(self.convs[0].weight, dgrad=True))
            self.key = self.current_key(dtype)
        return self.val


_PACK_DT = None


def _pack_dtype():
    global _PACK_DT
    if _PACK_DT is None:
        import numpy as np
        _PACK_DT = np.dtype([('src', '<i8'), ('wf', '<i8'), ('wd', '<i8'), ('so', '<i8'), ('si', '<i8'), ('st', '<i8'),
                             ('O', '<i4'), ('I', '<i4'), ('taps', '<i4'), ('Ototal', '<i4'), ('o0', '<i4'),
                             ('e0', '<i8'), ('wfrag', '<i8'), ('wdfrag', '<i8')], align=True)
        assert _PACK_DT.itemsize == 96
    return _PACK_DT


class ShadowSet:
    """Every conv shadow of a network, re-packed by ONE kernel launch per optimizer step
    (`idf_pack_conv_weights_batched`) instead of one launch per conv."""

    ROW = 16384

    def __init__(self, net):
        seen, self.items = set(), []
        for m in net.modules():
            for v in vars(m).values():
                sh = v.get('shadows') if isinstance(v, dict) else v
                if isinstance(sh, _Shadows) and id(sh) not in seen:
                    seen.add(id(sh))
                    self.items.append(sh)
                    sh.owners.append(self)
        self.dirty = False          # a shadow changed its layout state during a forward pass (see _Shadows.touch)
        self.tkey = None
        self.table = None
        self.pinned = None
        self.up_key = None          # the UpSample convs' summed sub-pixel weights: their own table, one more launch
        self.up_table = None

    def settle(self, dtype, need_dgrad):
        """End of a forward pass: if a block switched a shadow's layout state during the pass (fragment-major request, the
        attention fold), re-pack NOW -- tables and buffers are (re)built outside any capture, and the next pass starts clean."""
        if self.dirty and not torch.cuda.is_current_stream_capturing():
            self.dirty = False
            self.refresh(dtype, need_dgrad, force=True)       # the table too, even if another set has re-packed the shadows

    def refresh(self, dtype, need_dgrad, force=False):
        if not force and not any(s.stale(dtype, need_dgrad) for s in self.items):
            return
        import numpy as np
        for s in self.items:
            s.ensure_buffers(dtype, need_dgrad)
        tkey = (dtype, tuple((w.data_ptr(), s.val[0].data_ptr(), s.val[1].data_ptr() if s.val[1] is not None else 0,
                              s.val[2].data_ptr() if s.val[2] is not None else 0,
                              s.val[3].data_ptr() if s.val[3] is not None else 0)
                             for s in self.items for w in s.pack_sources()))
        if tkey != self.tkey:
            rows = []
            for s in self.items:
                Ot, o0 = sum(c.weight.shape[0] for c in s.convs), 0
                for w in s.pack_sources():
                    O, I, kh, kw = w.shape
                    taps = kh * kw
                    if taps > 1 and w.stride(2) != kw * w.stride(3):
                        raise RuntimeError('conv weight layout not packable in place')
                    st = w.stride(3) if taps > 1 else 0
                    for tap in range(taps):
                        for ot in range(-(-O // 32)):
                            for it in range(-(-I // 64)):
                                rows.append((w.data_ptr(), s.val[0].data_ptr(),
                                             s.val[1].data_ptr() if s.val[1] is not None else 0,
                                             w.stride(0), w.stride(1), st, O, I, taps, Ot, o0,
                                             tap | (ot << 8) | (it << 32),
                                             s.val[2].data_ptr() if s.val[2] is not None else 0,
                                             s.val[3].data_ptr() if s.val[3] is not None else 0))
                    o0 += O
            host = torch.from_numpy(np.array(rows, dtype=_pack_dtype()).view(np.uint8).reshape(len(rows), -1).copy())
            dev = self.items[0].val[0].device
            capturing = torch.cuda.is_current_stream_capturing()
            if self.table is None or self.table.shape != host.shape:
                if capturing:
                    raise RuntimeError('ShadowSet: run one eager forward before graph capture')
                self.table = torch.empty(host.shape, dtype=torch.uint8, device=dev)
                self.pinned = torch.empty(host.shape, dtype=torch.uint8).pin_memory()
            if capturing:
                self.pinned.copy_(host)
                self.table.copy_(self.pinned, non_blocking=True)
            else:
                self.table.copy_(host.to(dev))
            self.tkey = tkey
        from ._lib import call, F32, BF16
        folds = [s for s in self.items if s.fold is not None]
        if folds:          # the attention blocks' folded V weights / biases first: the pack below reads them
            fkey = tuple(tuple(t.data_ptr() for t in s.fold) for s in folds)
            if fkey != getattr(self, 'fold_key', None):
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError('ShadowSet: run one eager forward before graph capture')
                dt = np.dtype([('p%d' % i, '<i8') for i in range(8)] + [('C', '<i4'), ('pad', '<i4')])
                rows = [tuple(t.data_ptr() for t in s.fold) + (s.fold[0].shape[0], 0) for s in folds]
                self.fold_table = torch.from_numpy(np.array(rows, dtype=dt).view(np.uint8).reshape(len(rows), -1).copy()).to(
                    self.items[0].val[0].device)
                self.fold_key = fkey
                self.fold_C = max(s.fold[0].shape[0] for s in folds)
            call('idf_attn_fold_batched', self.fold_table.data_ptr(), len(folds), self.fold_C, torch.cuda.current_stream().cuda_stream)
        call('idf_pack_conv_weights_batched', self.table.data_ptr(), self.table.shape[0],
             F32 if dtype == torch.float32 else BF16, torch.cuda.current_stream().cuda_stream)
        ups = [s for s in self.items if s.sub is not None]
        if ups:
            ukey = tuple((s.convs[0].weight.data_ptr(), s.sub.data_ptr(), s.subd.data_ptr() if s.subd is not None else 0) for s in ups)
            if ukey != self.up_key:
                rows = []
                for s in ups:
                    w = s.convs[0].weight
                    rows.append((w.data_ptr(), s.sub.data_ptr(), s.subd.data_ptr() if s.subd is not None else 0, w.stride(0), w.stride(1),
                                 w.stride(3), w.shape[0], w.shape[1]))
                    if w.stride(2) != 3 * w.stride(3):
                        raise RuntimeError('conv weight layout not packable in place')
                dt = np.dtype([('src', '<i8'), ('dst', '<i8'), ('dstd', '<i8'), ('so', '<i8'), ('si', '<i8'), ('st', '<i8'), ('O', '<i4'), ('I', '<i4')])
                host = torch.from_numpy(np.array(rows, dtype=dt).view(np.uint8).reshape(len(rows), -1).copy())
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError('ShadowSet: run one eager forward before graph capture')
                self.up_table = host.to(self.items[0].val[0].device)
                self.up_key = ukey
                self.up_pairs = max(s.convs[0].weight.shape[0] * s.convs[0].weight.shape[1] for s in ups)
            call('idf_upconv_pack_batched', self.up_table.data_ptr(), len(ups), self.up_pairs, torch.cuda.current_stream().cuda_stream)
        for s in self.items:
            s.key = s.current_key(dtype)


def _cfg(shadows, mode, taps, act, p_drop=0.0, salt=0):
    return dict(shadows=shadows, mode=mode, taps=taps, act=act, p_drop=p_drop, salt=salt)


def _xavier_all(module):
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            init.xavier_uniform_(m.weight)
            init.zeros_(m.bias)


class TimeEmbedding(nn.Module):
    """modules.py:9-38: frozen interleaved sin/cos table -> Linear -> SiLU -> Linear."""

    def __init__(self, T, d_model, dim):
        assert d_model % 2 == 0
        super().__init__()
        freq = torch.exp(-(torch.arange(0, d_model, step=2) / torch.Tensor([d_model]) * math.log(10000)))
        ang = torch.arange(T).float()[:, None] * freq[None, :]
        table = torch.stack([torch.sin(ang), torch.cos(ang)], dim=-1).view(T, d_model)
        self.timembedding = nn.Sequential(
            nn.Embedding.from_pretrained(table), nn.Linear(d_model, dim), nn.SiLU(), nn.Linear(dim, dim))
        _xavier_all(self)

    def forward(self, t):
        tab, l1, _, l2 = self.timembedding
        e = ops.gather_rows(tab.weight, t)
        e = ops.linear(e, l1.weight, l1.bias)
        return ops.linear(e, l2.weight, l2.bias, silu_in=True)


_freq_cache = {}


def timestep_embedding(timesteps, dim, max_period=10000):
    """modules.py:41-60 ([cos..., sin...]), used by LatentUNet only.  The frequency table is evaluated with the
    reference's CPU expression once per (device, dim) and kept on the device: no H2D copy per call, so a denoise
    step can be captured into a graph."""
    half = dim // 2
    key = (timesteps.device, half, max_period)
    freqs = _freq_cache.get(key)
    if freqs is None:
        freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) /
                          half).to(device=timesteps.device)
        _freq_cache[key] = freqs
    ang = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class DownSample(nn.Module):
    """modules.py:63-75: conv3x3 stride 2."""

    def __init__(self, in_ch):
        super().__init__()
        self.main = nn.Conv2d(in_ch, in_ch, 3, stride=2, padding=1)
        _xavier_all(self)
        self._cfg = _cfg(_Shadows(self.main), ops.S2, 9, _ACT_NONE)

    def forward(self, x, temb=None, aemb=None, want_alias=False):
        """want_alias: also return an alias of x for the skip connection (its gradient joins this conv's
        data-gradient epilogue instead of an autograd add)."""
        if want_alias:
            return ops.fused_conv(x, self.main.weight, self.main.bias, self._cfg, passthrough=1, want_stats=True)
        return ops.fused_conv(x, self.main.weight, self.main.bias, self._cfg, want_stats=True)


class UpSample(nn.Module):
    """modules.py:78-93: nearest x2 then conv3x3 -- the upsample is folded into the
    conv's read addressing, the 4x tensor never exists."""

    def __init__(self, in_ch):
        super().__init__()
        self.main = nn.Conv2d(in_ch, in_ch, 3, stride=1, padding=1)
        _xavier_all(self)
        self._cfg = _cfg(_Shadows(self.main), ops.UP2, 9, _ACT_NONE)
        self._cfg['shadows'].want_sub = True

    def forward(self, x, temb=None, aemb=None):
        w = self.main.weight
        tiles = ops.upconv_tiles(x, w.shape[0])
        if tiles:
            # bf16: four 2x2 convs on the low-resolution input with summed weights (idf_upconv_bf16: 16 tap products per four
            # outputs instead of 36); the summed weights are one more shadow, re-packed with the others.  The backward pass is
            # the 3x3 conv's (data gradient through the fused up-sampling read, weight gradient of the UP2 class).
            train = torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)
            sh = self._cfg['shadows']
            sh(x.dtype, train)
            if sh.sub is not None:
                y, st = ops.upconv_raw(x, sh.sub, self.main.bias, w.shape[0], tiles)
                if not train:
                    return ops._tag(y, st)
                return ops.fused_conv(x, w, self.main.bias, self._cfg, want_stats=True, pre=(y, None, None, None, None, None, st))
        return ops.fused_conv(x, w, self.main.bias, self._cfg, want_stats=True)


class AttnBlock(nn.Module):
    """modules.py:129-164: GN -> q,k,v 1x1 -> softmax(q k^T / sqrt(C)) v -> 1x1 proj -> + x.
    q,k,v run as one GN-prologue 1x1 conv with the three weights concatenated."""

    def __init__(self, in_ch):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, in_ch)
        self.proj_q = nn.Conv2d(in_ch, in_ch, 1, stride=1, padding=0)
        self.proj_k = nn.Conv2d(in_ch, in_ch, 1, stride=1, padding=0)
        self.proj_v = nn.Conv2d(in_ch, in_ch, 1, stride=1, padding=0)
        self.proj = nn.Conv2d(in_ch, in_ch, 1, stride=1, padding=0)
        self.initialize()
        self._qkv = _Shadows(self.proj_q, self.proj_k, self.proj_v)
        self._cfg_qkv = _cfg(self._qkv, ops.S1, 1, _ACT_AFFINE)
        self._cfg_proj = _cfg(_Shadows(self.proj), ops.S1, 1, _ACT_NONE)

    def initialize(self):
        for m in (self.proj_q, self.proj_k, self.proj_v, self.proj):
            init.xavier_uniform_(m.weight)
            init.zeros_(m.bias)
        init.xavier_uniform_(self.proj.weight, gain=1e-5)

    def _fold(self, on, dev):
        """Switch the q | k | v shadows between the convs' own weights and (Wq, Wk, Wv' = Wp Wv) with biases (bq | bk | Wp bv + bp)."""
        sh = self._qkv
        if not on:
            if sh.fold is not None:
                sh.srcs = sh.fold = None
                sh.key = None
                sh.touch()
                self._cfg_qkv.pop('bias_values', None)
            return
        if sh.fold is None or sh.fold[6].device != dev:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('AttnBlock: run one eager forward before graph capture')
            C = self.proj.weight.shape[0]
            wvf = torch.empty((C, C, 1, 1), dtype=torch.float32, device=dev)
            bf = torch.empty((3 * C,), dtype=torch.float32, device=dev)
            sh.srcs = [self.proj_q.weight, self.proj_k.weight, wvf]
            sh.fold = (self.proj.weight, self.proj.bias, self.proj_v.weight, self.proj_v.bias, self.proj_q.bias, self.proj_k.bias,
                       wvf, bf)
            sh.key = None
            sh.touch()
            self._cfg_qkv['bias_values'] = bf

    def forward(self, x):
        gn = self.group_norm
        B, C, H, W = x.shape
        # bf16, shapes the fused attention covers: the proj conv is folded into V (Wv' = Wp Wv, b' = Wp bv + bp -- the rows of the
        # softmax sum to one, so the bias passes through the product): y = x + P V', no proj launch / data gradient / weight
        # gradient; the chain rule back to proj and proj_v runs once per backward pass (ops._FoldProjV)
        tiles = 0
        if ops._ATTN_FOLD and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4:
            tiles = int(ops._lib.load().idf_attn_res_tiles(B, H * W, C, ops.BF16))
        self._fold(tiles > 0, x.device)
        wq, bq = self._qkv.weight(), self._qkv.bias()
        if tiles:
            wq, bq = ops.fold_proj_v(wq, bq, self.proj.weight, self.proj.bias, self.proj_v.weight, self.proj_v.bias)
            pre_q = pre_a = None
            if ops.attn_block_ok(x):
                # 256 tokens x 128 channels from B = 256 up: the whole block is ONE launch (idf_attnblock_fwd); while gradients are
                # recorded it also leaves what the two ops below would have saved, and they only record their backward passes
                train = torch.is_grad_enabled() and x.requires_grad
                val = self._qkv(x.dtype, train)
                if val[2] is None:
                    self._qkv.request_frag()          # fragment-major q | k | v' weights come with the next re-pack
                else:
                    y, st, qkv, h, o, lse, mean, rstd, sc, sh = ops.attn_block_fwd_raw(
                        x, ops.stats_of(x), gn.weight, gn.bias, val[2], self._qkv.fold[7], train)
                    if not train:
                        return ops._tag(y, st)
                    pre_q, pre_a = (qkv, h, mean, rstd, sc, sh, None), (y, st, o, lse)
            qkv, xa = ops.fused_conv(x, wq, bq, self._cfg_qkv, gn.weight, gn.bias, passthrough=1, pre=pre_q)
            return ops.attention_res(qkv, xa, tiles, pre=pre_a)
        qkv, x = ops.fused_conv(x, wq, bq, self._cfg_qkv, gn.weight, gn.bias,
                                passthrough=1)      # the residual branch's gradient joins the GN backward
        o = ops.attention(qkv)
        return ops.fused_conv(o, self.proj.weight, self.proj.bias, self._cfg_proj, residual=x, want_stats=True)


class CrossAttnBlock(nn.Module):
    """modules.py:167-203.  Constructed by every AuxResBlock but never executed by
    the reference (crossattn=False at every call site); kept for its state_dict keys."""

    def __init__(self, in_ch):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, in_ch)
        self.proj_q = nn.Conv2d(in_ch, in_ch, 1, stride=1, padding=0)
        self.proj_k = nn.Conv2d(in_ch, in_ch, 1, stride=1, padding=0)
        self.proj_v = nn.Conv2d(in_ch, in_ch, 1, stride=1, padding=0)
        self.proj = nn.Conv2d(in_ch, in_ch, 1, stride=1, padding=0)
        AttnBlock.initialize(self)

    def forward(self, x, a):
        raise NotImplementedError('CrossAttnBlock is dead code in the reference (never executed)')


def _gn_act_conv(in_ch, out_ch, dropout=None):
    layers = [nn.GroupNorm(32, in_ch), nn.SiLU()]
    if dropout is not None:
        layers.append(nn.Dropout(dropout))
    layers.append(nn.Conv2d(in_ch, out_ch, 3, stride=1, padding=1))
    return nn.Sequential(*layers)


class _ResBase(nn.Module):
    """Shared machinery of ResBlock / AuxResBlock / ResBlock_encoder."""

    def _finish_init(self, in_ch, out_ch, attn):
        self.shortcut = nn.Conv2d(in_ch, out_ch, 1, stride=1, padding=0) if in_ch != out_ch else nn.Identity()
        self.use_attn = attn
        self.attn = AttnBlock(out_ch) if attn else nn.Identity()

    def _setup(self, dropout):
        self._film = {}      # FiLM pairs the parent network projected in one batched GEMM (consumed per call)
        self.ctx = RunCtx()
        self.salt = 0
        self.p_drop = dropout
        for name in ('block1', 'block2', 'block3'):
            blk = getattr(self, name, None)
            if blk is not None:
                setattr(self, '_sh_' + name, _Shadows(blk[-1]))
        if isinstance(self.shortcut, nn.Conv2d):
            self._cfg_sc = _cfg(_Shadows(self.shortcut), ops.S1, 1, _ACT_NONE)

    def _gn_conv(self, name, x, film_t=None, film_a=None, drop_site=None, residual=None, passthrough=False, single=False):
        blk = getattr(self, name)
        gn, conv = blk[0], blk[-1]
        seed = self.ctx.seed if (drop_site is not None and self.training) else None
        cfg = _cfg(getattr(self, '_sh_' + name), ops.S1, 9, _ACT_SILU, self.p_drop, self.salt + (drop_site or 0))
        # every conv of a block feeds a GroupNorm (the next stage, the next block, the AttnBlock or the tail):
        # its epilogue leaves the statistics of its output behind
        # single: x is the previous stage's conv output and nobody else reads it -- the GroupNorm backward may then be
        # folded into that conv's data-gradient launch (ops.LazyGrad)
        return ops.fused_conv(x, conv.weight, conv.bias, cfg, gn.weight, gn.bias, film_t, film_a, residual, seed,
                              passthrough, want_stats=True, x_single_use=single)

    def _small(self, x, ft, fa, want_alias):
        """The whole block as one launch where a workgroup can own an image (8x8 maps, ops.resblock_small): y [, alias of
        x], or None when the fused kernel does not cover this call."""
        x1, x2 = x if isinstance(x, tuple) else (x, None)
        names = [n for n in ('block1', 'block2', 'block3') if hasattr(self, n)]
        has_sc = isinstance(self.shortcut, nn.Conv2d)
        out_ch = getattr(self, names[-1])[-1].weight.shape[0]
        if (x2 is not None and not has_sc) or not ops.resblock_small_ok(x1, x2, out_ch, len(names)):
            return None
        if not has_sc and x1.shape[1] != out_ch:
            return None
        stages = []
        for k, name in enumerate(names):
            blk = getattr(self, name)
            stages.append((blk[-1], blk[0], _cfg(getattr(self, '_sh_' + name), ops.S1, 9, _ACT_SILU, self.p_drop, self.salt + k)))
        seed = self.ctx.seed if self.training else None
        film_stage = 1 if (ft is not None or fa is not None) else -1
        return ops.resblock_small(x1, x2, stages, (self.shortcut, self._cfg_sc) if has_sc else None, ft, fa, film_stage, seed,
                                  self.p_drop, [False] + [True] * (len(names) - 1), want_alias)

    def _block1(self, x, want_alias=False):
        """(h, residual[, alias of x]) of the block's first stage.  x may be the pair (h_prev, skip) of an
        up-path block: the concatenation is then read in place by the two-source kernels instead of being
        materialised.  want_alias: a second alias of x for the skip connection that branches off here."""
        if isinstance(x, tuple):
            blk = self.block1
            if isinstance(self.shortcut, nn.Conv2d) and ops.block_entry_cat_ok(x[0], x[1], blk[-1].weight,
                                                                               self.shortcut.weight):
                cfg = _cfg(self._sh_block1, ops.S1, 9, _ACT_SILU, self.p_drop, self.salt)
                return ops.block_entry_cat(x[0], x[1], blk[-1], blk[0], self.shortcut, cfg, self._cfg_sc)
            x = torch.cat(x, dim=1)
        if want_alias:
            h, x, alias = self._gn_conv('block1', x, passthrough=2)
            return h, self._shortcut(x), alias
        h, x = self._gn_conv('block1', x, passthrough=1)     # x: the residual branch's gradient joins in block1
        return h, self._shortcut(x)

    def _shortcut(self, x):
        if isinstance(self.shortcut, nn.Conv2d):
            return ops.fused_conv(x, self.shortcut.weight, self.shortcut.bias, self._cfg_sc)
        return x


class ResBlock(_ResBase):
    """modules.py:206-258 (vanilla UNet block: FiLM on t only, three 3x3 convs)."""

    def __init__(self, in_ch, out_ch, tdim, dropout, attn=False):
        super().__init__()
        self.temb_proj = nn.Sequential(nn.SiLU(), nn.Linear(tdim, 2 * out_ch))
        self.block1 = _gn_act_conv(in_ch, out_ch)
        self.block2 = _gn_act_conv(out_ch, out_ch, dropout)
        self.block3 = _gn_act_conv(out_ch, out_ch, dropout)
        self._finish_init(in_ch, out_ch, attn)
        _xavier_all(self)
        self._setup(dropout)

    def forward(self, x, temb, want_alias=False):
        ft = self._film.pop('t', None) if self._film else None
        if ft is None:
            ft = ops.linear(temb, self.temb_proj[1].weight, self.temb_proj[1].bias, silu_in=True)
        small = self._small(x, ft, None, want_alias)
        if small is not None:
            h, *alias = small if want_alias else (small,)
            h = self.attn(h)
            return (h, alias[0]) if want_alias else h
        h, res, *alias = self._block1(x, want_alias)
        h = self._gn_conv('block2', h, film_t=ft, drop_site=1, single=True)
        h = self._gn_conv('block3', h, drop_site=2, residual=res, single=True)
        h = self.attn(h)
        return (h, alias[0]) if want_alias else h


class AuxResBlock(_ResBase):
    """modules.py:261-328: AdaGN block conditioned on t and on the auxiliary latent a."""

    def __init__(self, in_ch, out_ch, tdim, dropout, attn=False, crossattn: Union[bool, nn.Module] = False):
        super().__init__()
        self.block1 = _gn_act_conv(in_ch, out_ch)
        self.temb_proj = nn.Sequential(nn.SiLU(), nn.Linear(tdim, 2 * out_ch))
        self.aemb_proj = nn.Sequential(nn.SiLU(), nn.Linear(tdim, 2 * out_ch))
        self.block2 = _gn_act_conv(out_ch, out_ch, dropout)
        self.block3 = _gn_act_conv(out_ch, out_ch, dropout)
        self._finish_init(in_ch, out_ch, attn)
        self.use_crossattn = bool(crossattn)
        self.crossattn = CrossAttnBlock(out_ch)
        _xavier_all(self)   # also resets attn.proj to gain 1 (reference quirk 9)
        self._setup(dropout)

    def forward(self, x, temb, aemb=None, want_alias=False):
        ft = self._film.pop('t', None) if self._film else None
        fa = self._film.pop('a', None) if self._film else None
        if ft is None:
            ft = ops.linear(temb, self.temb_proj[1].weight, self.temb_proj[1].bias, silu_in=True)
        if fa is None:
            fa = ops.linear(aemb, self.aemb_proj[1].weight, self.aemb_proj[1].bias, silu_in=True)
        small = self._small(x, ft, fa, want_alias)
        if small is not None:
            h, *alias = small if want_alias else (small,)
            h = self.attn(h)
            if self.use_crossattn:
                h = self.crossattn(h, aemb)
            return (h, alias[0]) if want_alias else h
        h, res, *alias = self._block1(x, want_alias)
        h = self._gn_conv('block2', h, film_t=ft, film_a=fa, drop_site=1, single=True)
        h = self._gn_conv('block3', h, drop_site=2, residual=res, single=True)
        h = self.attn(h)
        if self.use_crossattn:
            h = self.crossattn(h, aemb)
        return (h, alias[0]) if want_alias else h


class ResBlock_encoder(_ResBase):
    """modules.py:331-366: unconditioned block, two 3x3 convs."""

    def __init__(self, in_ch, out_ch, dropout, attn=False):
        super().__init__()
        self.block1 = _gn_act_conv(in_ch, out_ch)
        self.block2 = _gn_act_conv(out_ch, out_ch, dropout)
        self._finish_init(in_ch, out_ch, attn)
        _xavier_all(self)
        self._setup(dropout)

    def forward(self, x, want_alias=False):
        small = self._small(x, None, None, want_alias)
        if small is not None:
            h, *alias = small if want_alias else (small,)
            h = self.attn(h)
            return (h, alias[0]) if want_alias else h
        h, res, *alias = self._block1(x, want_alias)
        h = self._gn_conv('block2', h, drop_site=1, residual=res, single=True)
        h = self.attn(h)
        return (h, alias[0]) if want_alias else h


def bind_context(net, ctx):
    """Share one RunCtx across a network and give every dropout site a unique salt."""
    for i, m in enumerate(net.modules()):
        if isinstance(m, _ResBase):
            m.ctx = ctx
            m.salt = 4 * i


def film_groups(blocks, which):
    """(weight group, bias group) of every block's FiLM projection: adjacent storage, so the batched
    GEMM reads the concatenated weights as a view.  Built once (the network's _post), kept on block 0."""
    name = 'temb_proj' if which == 't' else 'aemb_proj'
    groups = getattr(blocks[0], '_film_groups_' + which, None)
    if groups is None:
        lins = [getattr(b, name)[1] for b in blocks]
        groups = (ops.ParamGroup([l.weight for l in lins]), ops.ParamGroup([l.bias for l in lins]))
        setattr(blocks[0], '_film_groups_' + which, groups)
    return groups


def batched_film(blocks, emb, which):
    """All blocks' FiLM projections Linear(SiLU(emb)) (modules.py:269-276, 312, 316) as ONE
    GEMM over the concatenated weights; each block then reads its [B, 2C] column slice in
    place (row stride = total width).  `which`: 't' (temb_proj) or 'a' (aemb_proj)."""
    name = 'temb_proj' if which == 't' else 'aemb_proj'
    lins = [getattr(b, name)[1] for b in blocks]
    groups = film_groups(blocks, which)
    out = ops.linear(emb, ops.cat_params(groups[0]), ops.cat_params(groups[1]), silu_in=True)
    for blk, chunk in zip(blocks, out.split([l.weight.shape[0] for l in lins], dim=1)):
        blk._film[which] = chunk


def fused_film(time_embedding, t, blocks_t, a=None, fc=None, fc_silu=False, blocks_a=None):
    """The whole conditioning path in one call (ops.temb_film): TimeEmbedding(t), fc_a(a) and every block's FiLM
    projections; each block finds its [B, 2C] column slices in `_film` as after `batched_film`.  False when the fused
    entry does not apply (the caller then takes the per-product path)."""
    tab, l1, _, l2 = time_embedding.timembedding
    if not torch.is_tensor(t) or t.dtype != torch.long:
        return False
    if not ops.temb_film_ok(tab.weight, l1.weight, a, fc.weight if fc is not None else None):
        return False
    gt = film_groups(blocks_t, 't')
    film_t = (ops.cat_params(gt[0]), ops.cat_params(gt[1]))
    film_a = None
    if a is not None and blocks_a:
        ga = film_groups(blocks_a, 'a')
        film_a = (ops.cat_params(ga[0]), ops.cat_params(ga[1]))
    out_t, out_a = ops.temb_film(t, tab.weight, l1, l2, film_t, a if film_a is not None else None,
                                 fc if film_a is not None else None, fc_silu, film_a)
    for blk, chunk in zip(blocks_t, out_t.split([b.temb_proj[1].weight.shape[0] for b in blocks_t], dim=1)):
        blk._film['t'] = chunk
    if film_a is not None:
        for blk, chunk in zip(blocks_a, out_a.split([b.aemb_proj[1].weight.shape[0] for b in blocks_a], dim=1)):
            blk._film['a'] = chunk
    return True

