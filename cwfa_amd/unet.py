"""UNet of the LRNN on the HIP implicit-GEMM convolution (reference: unet.py:9-113,161-195).

Same module tree as the reference (``down_path.{i}.block.{0..5}``, ``up_path.{i}.{up,conv_block}``, ``last.{0,1}``) so
checkpoints load unchanged; torch.nn layers are used only as PARAMETER CONTAINERS (and to consume the torch RNG
identically at construction) -- ``forward`` never calls them.

Execution (per UNetConvBlock conv -> PReLU -> BatchNorm, unet.py:99-107):
  conv kernel with bias + PReLU epilogue  ->  [train: per-channel statistics kernel]  ->  fold BatchNorm into
  (scale, shift)  ->  applied on the LOAD side of whichever kernel consumes the tensor next (conv / conv-transpose /
  max-pool).  Zero padding is inserted after that affine, so this is exact at the borders.  The skip connection is an
  ADD (unet.py:190) fused into the consumer conv's load; ConvTranspose2d(k2,s2) runs as a 1x1 conv with a
  pixel-shuffle store.  ``F.dropout2d`` (always active in the reference, unet.py:80,86) becomes a per-(sample,channel)
  factor folded into the same load-side affine.

Not provided (unused by CWFA): up_mode='upsample', UNetConv3DBlock, UNetPullBlock, store_activations.
"""
import torch
from torch import nn

from . import ops

__all__ = ["UNet", "UNetConvBlock", "UNetUpBlock"]


_MATERIALIZE_UP = False  # the same for the transposed convolutions of the up path: measured, no gain (tools/bench_variants.sh)
_MATERIALIZE = True      # see UNetConvBlock.run (False: BatchNorm applied inside the consumer convolution's load)


class _Packed:
    """Cache of kernel-layout filter banks keyed by the parameter they were built from."""

    def __init__(self):
        self._c = {}

    def get(self, conv, transposed=False):
        w = conv.weight
        pc = self._c.get(id(conv))
        if pc is None or pc.version != w._version or pc.src_ptr != w.data_ptr() or pc.epoch != ops.pack_epoch():
            pc = self._c[id(conv)] = ops.pack_conv_weight(w, transposed=transposed)
        return pc


class _Drop:
    """A pending F.dropout2d(x, p) (training=True, the functional default the reference relies on, unet.py:80,86): the raw uniform
    draw [B,C]; the factor (u >= p) / (1 - p) is formed inside the kernel that folds it into a load-side affine (ops.bn_finish)."""
    __slots__ = ("u", "p")

    def __init__(self, u, p):
        self.u, self.p = u, p

    def mask(self):
        return ((self.u >= self.p).to(torch.float32) / (1.0 - self.p)).contiguous()


STATS_IN_EPILOGUE = True      # (tuning / ablation) False: train-mode BatchNorm statistics always by a separate pass (ops.channel_stats)


def _bn_batch_stats(bn):
    return bn.training or not bn.track_running_stats


def _bn_stats_buffer(bn, device):
    """The layer's own float64 [2C] statistics buffer, zeroed (bn_finish clears it again after use; an exception in between
    leaves the dirty flag set and the buffer is rebuilt).  One buffer per (device, stream): forwards of the same module on
    different streams accumulate into different buffers."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    bufs = bn.__dict__.setdefault("_cwfa_stats", {})
    st = bufs.get(key)
    if st is None or getattr(bn, "_cwfa_stats_dirty", False):
        st = bufs[key] = torch.zeros(2 * bn.num_features, dtype=torch.float64, device=device)
    bn._cwfa_stats_dirty = True
    return st


def _bn_affine(bn, y, drop=None, stats=None):
    """(scale, shift) of BatchNorm2d ``bn`` for its input ``y`` (raw conv+PReLU output), times the
    pending dropout factor ``drop`` (a _Drop or a ready [B,C] mask).  Train mode: the statistics come from the producing
    convolution's epilogue (``stats``: the buffer it added into, ops.conv2d(out_stats=)) or from one pass over ``y`` into the
    layer's own (kept zeroed) float64 buffer; then ONE launch for fold + running-buffer bookkeeping + dropout factor + re-zeroing."""
    C = bn.num_features
    mu, mb, p = (drop.u, None, drop.p) if isinstance(drop, _Drop) else (None, drop, 0.0)
    if _bn_batch_stats(bn):
        st = stats
        if st is None:
            st = _bn_stats_buffer(bn, y.device)
            ops.channel_stats(y, out=st)
        n = y.numel() // C
        track = bn.track_running_stats and bn.momentum is not None       # buffer bookkeeping, as nn.BatchNorm2d does
        out = ops.bn_finish(C, bn.weight, bn.bias, bn.eps, stats=st, count=float(n), running_mean=bn.running_mean if track else None,
                            running_var=bn.running_var if track else None, momentum=bn.momentum if track else None,
                            num_batches_tracked=bn.num_batches_tracked if track else None, mask_bc=mb, mask_u=mu, drop_p=p,
                            zero_stats=True)
        bn._cwfa_stats_dirty = False
        return out
    return ops.bn_finish(C, bn.weight, bn.bias, bn.eps, running_mean=bn.running_mean, running_var=bn.running_var, mask_bc=mb,
                         mask_u=mu, drop_p=p)


def _drop_mask(p, B, C, device):
    """F.dropout2d(x, p) as a ready keep-mask / (1-p) tensor (the training path keeps it for its backward)."""
    if not p:
        return None
    return _Drop(torch.rand(B, C, device=device), p).mask()


def _drop(p, B, C, device, pool=None):
    """The same draw, left pending (inference path).  ``pool``: a _DrawPool to slice the uniforms from (one RNG launch per forward)."""
    if not p:
        return None
    return _Drop(pool.take(B, C) if pool is not None else torch.rand(B, C, device=device), p)


class _DrawPool:
    """All uniform draws of one forward pass from ONE torch.rand launch (six dropout2d masks per UNet pass otherwise cost six
    launches of a few hundred numbers each)."""

    def __init__(self, n, device):
        self.u = torch.rand(n, device=device) if n > 0 else None
        self.pos = 0

    def take(self, B, C):
        out = self.u[self.pos:self.pos + B * C].view(B, C)
        self.pos += B * C
        return out


def _drop_affine(C, drop):
    """A bare dropout factor as a load-side affine."""
    if isinstance(drop, _Drop):
        return ops.bn_finish(C, mask_u=drop.u, drop_p=drop.p)
    return ops.bn_finish(C, mask_bc=drop)


class UNetConvBlock(nn.Module):
    """conv3x3 -> act -> BN -> conv3x3 -> act -> BN.  unet.py:94-113."""

    def __init__(self, in_size, out_size, padding, batch_norm, kernel_size=3, use_bias=False, stride=1,
                 activation=nn.LeakyReLU):
        super().__init__()
        if stride != 1 or kernel_size != 3 or not padding:
            raise NotImplementedError("cwfa_amd UNetConvBlock: 3x3, stride 1, padded convolutions only")
        block = [nn.Conv2d(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=int(padding), bias=use_bias),
                 activation()]
        if batch_norm:
            block.append(nn.BatchNorm2d(out_size))
        block += [nn.Conv2d(out_size, out_size, kernel_size=kernel_size, stride=1, padding=int(padding), bias=use_bias),
                  activation()]
        if batch_norm:
            block.append(nn.BatchNorm2d(out_size))
        self.block = nn.Sequential(*block)
        self.batch_norm = batch_norm
        self._packed = _Packed()

    def _layers(self):
        mods = list(self.block)
        step = 3 if self.batch_norm else 2
        return [(mods[i], mods[i + 1], mods[i + 2] if self.batch_norm else None) for i in (0, step)]

    @staticmethod
    def _act(a):
        if isinstance(a, nn.PReLU):
            if a.weight.numel() != 1:
                raise NotImplementedError("per-channel PReLU")
            return "prelu", a.weight
        if isinstance(a, nn.ELU):
            return "elu", None
        if isinstance(a, nn.ReLU):
            return "relu", None
        raise NotImplementedError(f"activation {type(a).__name__} has no HIP epilogue")

    def run(self, x, in_affine=None, in_add=None, out_mask=None):
        """-> (y, affine): y is the RAW output of the last conv+act; ``affine`` = pending BatchNorm (x mask) the consumer
        must apply on load (None when the block has no BatchNorm and no mask)."""
        aff = in_affine
        layers = self._layers()
        packs = [self._packed.get(conv) for conv, _, _ in layers]
        for li, (conv, act, bn) in enumerate(layers):
            kind, alpha = self._act(act)
            sc, sh = aff if aff is not None else (None, None)
            add = in_add if li == 0 else None
            if (_MATERIALIZE and ops._split_bf16 < 2 and (sc is not None or add is not None)
                    and (x.shape[2] * x.shape[3]) % 4 == 0):         # (the split / bf16 kernels apply the prologue for free)
                # one streaming pass writes the BatchNorm (x mask, + skip) output, and the convolution runs without its
                # load-side prologue: the plain kernels are 8-10 % faster than the prologue ones and the 2-D Winograd
                # kernel (x1.2 on the 512 / 1024-channel layers) only pays off without it -- the pass costs 2-7 %
                x = ops.plane_affine(x, sc, sh, add=add)
                sc = sh = add = None
            # train-mode BatchNorm behind the convolution: its statistics come out of the convolution's epilogue where the
            # kernel can do that (split-bf16 3x3, NCHW output)
            st = None
            if (STATS_IN_EPILOGUE and bn is not None and _bn_batch_stats(bn)
                    and ops.conv_writes_stats(packs[li], kind)):
                st = _bn_stats_buffer(bn, x.device)
            x = ops.conv2d(x, packs[li], bias=conv.bias, act=kind, prelu_alpha=alpha, in_scale=sc,
                           in_shift=sh, in_add=add, out_stats=st)
            last = li == len(layers) - 1
            m = out_mask if last else None
            if bn is not None:
                aff = _bn_affine(bn, x, m, stats=st)
            elif m is not None:
                aff = _drop_affine(x.shape[1], m)
            else:
                aff = None
        return x, aff

    def forward(self, x):
        y, aff = self.run(x)
        if aff is None:
            return y
        return ops.channel_affine(y, aff[0], aff[1])


class UNetUpBlock(nn.Module):
    """ConvTranspose2d(k2,s2) -> (+ skip) -> UNetConvBlock.  unet.py:161-195."""

    def __init__(self, in_size, out_size, up_mode, padding, batch_norm, use_bias=False, skip_conn=True,
                 activation=nn.Softplus):
        super().__init__()
        self.skip_conn = skip_conn
        if up_mode != 'upconv':
            raise NotImplementedError("cwfa_amd UNetUpBlock: up_mode='upconv' only (what the LRNN uses, networks.py:532)")
        self.up = nn.ConvTranspose2d(in_size, out_size, kernel_size=2, stride=2, bias=use_bias)
        in_size = out_size if not skip_conn else in_size // 2
        self.conv_block = UNetConvBlock(in_size, out_size, padding, batch_norm, use_bias=use_bias, activation=activation)
        self._packed = _Packed()

    @staticmethod
    def center_crop(layer, target_size):
        _, _, h, w = layer.size()
        dy, dx = (h - target_size[0]) // 2, (w - target_size[1]) // 2
        return layer[:, :, dy:(dy + target_size[0]), dx:(dx + target_size[1])]

    def run(self, x, bridge, in_affine=None, out_mask=None):
        sc, sh = in_affine if in_affine is not None else (None, None)
        if _MATERIALIZE_UP and ops._split_bf16 < 1 and sc is not None and (x.shape[2] * x.shape[3]) % 4 == 0:
            x = ops.plane_affine(x, sc, sh)
            sc = sh = None
        up = ops.conv2d(x, self._packed.get(self.up, transposed=True), bias=self.up.bias, in_scale=sc, in_shift=sh)
        add = self.center_crop(bridge, up.shape[2:]) if self.skip_conn else None
        return self.conv_block.run(up, in_add=add, out_mask=out_mask)

    def forward(self, x, bridge):
        y, aff = self.run(x, bridge)
        return y if aff is None else ops.channel_affine(y, aff[0], aff[1])


class UNet(nn.Module):
    """unet.py:9-91."""

    def __init__(self, in_channels=1, n_classes=2, depth=5, wf=6, padding=True, batch_norm=True, up_mode='upsample',
                 drop_out=0, use_bias=False, skip_conn=False, activation=nn.PReLU):
        super().__init__()
        assert up_mode in ('upconv', 'upsample')
        self.padding, self.depth, self.skip_conn, self.drop_out = padding, depth, skip_conn, drop_out
        prev = in_channels
        self.down_path = nn.ModuleList()
        for i in range(depth):
            self.down_path.append(UNetConvBlock(prev, 2 ** (wf + i), padding, batch_norm, use_bias=use_bias,
                                                activation=activation))
            prev = 2 ** (wf + i)
        self.up_path = nn.ModuleList()
        for i in reversed(range(depth - 1)):
            self.up_path.append(UNetUpBlock(prev, 2 ** (wf + i), up_mode, padding, batch_norm, use_bias=use_bias,
                                            skip_conn=skip_conn, activation=activation))
            prev = 2 ** (wf + i)
        self.last = nn.Sequential(nn.Conv2d(prev, n_classes, kernel_size=1, bias=use_bias), activation())
        self._packed = _Packed()

    def forward(self, x, store_activations=False, in_affine=None):
        if store_activations:
            raise NotImplementedError("store_activations is a debugging aid of the reference, not on the hot path")
        B, dev = x.shape[0], x.device
        skips, aff = [], in_affine
        n_draw = 0
        if self.drop_out:
            n_draw = B * (sum(d.block[0].out_channels for d in list(self.down_path)[:-1]) +
                          sum(u.conv_block.block[0].out_channels for u in self.up_path))
        pool = _DrawPool(n_draw, dev)
        for i, down in enumerate(self.down_path):
            x, aff = down.run(x, in_affine=aff)
            if i != len(self.down_path) - 1:
                sc, sh = aff if aff is not None else (None, None)
                # adaptive max-pool to (W//2, W//2) (unet.py:79); the BatchNorm affine is applied before the max and
                # the normalised full-resolution map is written once for the skip connection
                x, full = ops.maxpool(x, x.shape[-1] // 2, x.shape[-1] // 2, sc, sh, want_full=True)
                skips.append(full)
                m = _drop(self.drop_out, B, x.shape[1], dev, pool)
                aff = _drop_affine(x.shape[1], m) if m is not None else None
        for i, up in enumerate(self.up_path):
            m = _drop(self.drop_out, B, up.conv_block.block[0].out_channels, dev, pool)
            x, aff = up.run(x, skips[-i - 1], in_affine=aff, out_mask=m)
        kind, alpha = UNetConvBlock._act(self.last[1])
        sc, sh = aff if aff is not None else (None, None)
        return ops.conv2d(x, self._packed.get(self.last[0]), bias=self.last[0].bias, act=kind, prelu_alpha=alpha,
                          in_scale=sc, in_shift=sh)
