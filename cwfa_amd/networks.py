"""Model builders of the hot path on HIP kernels (reference: networks.py).

Provided (same names, signatures, module trees and therefore state_dict keys as the reference):
  conditional_wavelet_flow (:264-368), wavelet_flow_subnetwork / ...2D / ...2D_first (:586-706),
  cond_network / ResidualBlock (:165-242), GlobalAttention (:244-262), drop_path (:370-385), ConvNeXt (:468-503),
  LRNN (:505-555), Encoder (:557-584), the weight initialisers (:19-96), reset_ActNorm / reset_perm (:137-163),
  serialize_INN_step / load_INN_steps (:708-756).
Not provided (unused by the CWFA path): subnet_conv, subnet_conv_half, LayerNorm, Block,
XLFMNet.

torch.nn layers appear only as PARAMETER CONTAINERS: they give the reference's state_dict layout and consume the torch
RNG in the reference's order at construction (including the reference's double construction of the sub-networks,
networks.py:605 + :682, and the discarded Conv3d stack of LRNN, :521-526).  ``forward`` never calls them; all
arithmetic runs in libcwfa_hip.so.
"""
import glob
import math
import re

import torch
import torch.nn as nn

from . import autograd as AG
from . import ops
from .FrEIA import framework as Ff
from .FrEIA import modules as Fm
from .INN_utils import HaarTransform1D, PermuteDim
from .unet import UNet, _Packed

__all__ = ["conditional_wavelet_flow", "wavelet_flow_subnetwork", "wavelet_flow_subnetwork2D",
           "wavelet_flow_subnetwork2D_first", "cond_network", "ResidualBlock", "GlobalAttention", "drop_path", "ConvNeXt",
           "LRNN", "Encoder", "subnet_initialization", "subnet_initialization_small", "zero_initialization",
           "subnet_initialization_positive", "reset_ActNorm", "reset_perm", "serialize_INN_step", "load_INN_steps",
           "HaarTransform1D", "PermuteDim", "UNet"]

# carried into the sub-network constructors exactly like the reference's module-level global (networks.py:272-274,604)
networks_n_chans = 64

_CONVS = (nn.Conv2d, nn.Conv3d, nn.Linear)


# --------------------------------------------------------------------------------------------------- initialisers
def subnet_initialization(m):
    """Kaiming-uniform weights, bias x 0.1.  networks.py:19-27."""
    ops.invalidate_packs()          # `.data` edits below do not bump the version counters the pack caches are keyed on
    if isinstance(m, _CONVS):
        nn.init.kaiming_uniform_(m.weight.data)
        if m.bias is not None:
            m.bias.data *= 0.1


def subnet_initialization_small(m):
    """Xavier-uniform (gain 0.01) weights, bias x 0.01.  networks.py:29-38."""
    ops.invalidate_packs()          # `.data` edits below do not bump the version counters the pack caches are keyed on
    if isinstance(m, _CONVS):
        nn.init.xavier_uniform_(m.weight.data, 0.01)
        if m.bias is not None:
            m.bias.data *= 0.01


def zero_initialization(m):
    """networks.py:40-49."""
    ops.invalidate_packs()          # `.data` edits below do not bump the version counters the pack caches are keyed on
    if isinstance(m, _CONVS):
        nn.init.constant_(m.weight.data, 0.0)
        if m.bias is not None:
            m.bias.data *= 0.0


def subnet_initialization_positive(m):
    """|Xavier-uniform (gain 0.1)| weights, bias x 0.1.  networks.py:51-62."""
    ops.invalidate_packs()          # `.data` edits below do not bump the version counters the pack caches are keyed on
    if isinstance(m, _CONVS):
        nn.init.xavier_uniform_(m.weight.data, 0.1)
        m.weight.data = m.weight.data.abs()
        if m.bias is not None:
            m.bias.data *= 0.1


def reset_ActNorm(network, n_to_reset=50):
    """Re-arm the data-dependent init of the first ``n_to_reset`` ActNorm layers.  networks.py:137-151."""
    n = 0
    ops.invalidate_packs()
    for mod in next(network.named_children())[1]:
        if isinstance(mod, Fm.ActNorm):
            mod.init_on_next_batch = True
            n += 1
            if n_to_reset and n >= n_to_reset:
                break
    return network, n


def reset_perm(network):
    """A no-op in the reference too (it rebinds a loop variable, networks.py:159-163); kept for API compatibility."""
    return network


# --------------------------------------------------------------------------------------------------- coupling sub-networks
class wavelet_flow_subnetwork(nn.Module):
    """1x1(c_in->n) -> 3 x [3x3(n->n), ELU, 1x1(n->n), +residual, ELU] -> 3x3(n->c_out).  networks.py:586-671.

    HIP execution: the 1x1 (direct MFMA kernel), three fused layer launches (3x3 -> ELU -> 1x1 -> +x -> ELU in one kernel:
    fp32 Winograd, or both convolutions split-bf16 with channel-blocked maps in between) and the output 3x3 -- which, for
    the data-dependent block types, applies the coupling in its epilogue (``couple``): 5 launches per sub-network.
    ``normal=False`` (the ``_first`` variant) takes
    cat(mean, omega): the conv stack sees only the omega half and the mean half is passed through as ``-mean/sqrt(2)``.
    """

    def __init__(self, c_in, c_out, c_internal=32):
        super().__init__()
        self.c_in, self.c_out = c_in, c_out
        self.n_ch = networks_n_chans
        self.init_blocks(nn.Conv3d, nn.BatchNorm3d)      # as the reference: consumes the RNG, then gets replaced
        self.normal = True
        self._packed = _Packed()
        self._panels = {}

    def init_blocks(self, conv_type, bn=nn.BatchNorm2d, use_bias=True):
        self.conv_type = conv_type
        k, n = 3, self.n_ch
        self.bn = None
        self.act = nn.ELU
        cv = lambda i, o, ks: conv_type(i, o, ks, padding=ks // 2, bias=use_bias)   # noqa: E731
        self.block_grad_up = cv(self.c_in // 2, self.c_in, 3)       # dead layers of the reference, kept for the
        self.block1 = cv(self.c_in // 2, n, 1)                      # state_dict layout (block_grad_up is never used,
        self.block12 = cv(self.c_in, n, 1)                          # block1/block7 only by `_first`, block12/72 otherwise)
        self.block2 = nn.Sequential(cv(n, n, k), self.act(), cv(n, n, 1))
        self.block3 = self.act()
        self.block4 = nn.Sequential(cv(n, n, k), self.act(), cv(n, n, 1))
        self.block5 = self.act()
        self.block6 = nn.Sequential(cv(n, n, k), self.act(), cv(n, n, 1))
        self.block7 = nn.Sequential(self.act(), cv(n, self.c_out // 2, k))
        self.block72 = nn.Sequential(self.act(), cv(n, self.c_out, k))

    # ---- HIP path
    def _stack(self, u, conv_in, conv_out, out=None, couple=None):
        """``couple`` = (x, out, clamp_kind, clamp, pre_scale, rev, logdet): the last convolution applies the coupling to
        ``x`` from its accumulators (ops.conv3x3_couple) instead of writing [s_raw | t]."""
        if self.conv_type is not nn.Conv2d:
            raise NotImplementedError("3-D sub-networks are not used by CWFA (every graph uses the 2-D subclasses)")
        if couple is None and AG.tracking(u, conv_in, conv_out, self.block2, self.block4, self.block6):
            # torch would record a graph here (training, CWFA.py:1002-1006): the whole stack as ONE autograd node whose backward is
            # training.subnet_backward; cat(half, condition) is materialised through a differentiable concatenation
            a = AG.subnet(self, AG.concat(list(u)) if isinstance(u, (list, tuple)) else u, conv_in, conv_out)
            if out is not None:
                raise NotImplementedError("wavelet_flow_subnetwork: `out=` views are an inference-path feature")
            return a
        P = self._packed.get
        fused = self.n_ch == 64 and all(blk[0].bias is not None and blk[2].bias is not None
                                        for blk in (self.block2, self.block4, self.block6))
        H_, W_ = (u[0] if isinstance(u, (list, tuple)) else u).shape[2:4]
        split_layers = fused and ops._split_bf16 >= 2 and 64 * H_ * W_ * 4 < 2 ** 31
        # ``u`` may be the list [half, condition] of a coupling block's input: the 1x1 then reads cat(half, condition) from the
        # two tensors (ops.pack_conv_weight_cat) instead of a materialised concatenation (coupling_layers.py:74-87)
        two = isinstance(u, (list, tuple))
        u1 = None
        if (two and split_layers and ops.FIRST_LAYER_COMPOSED and ops.FIRST_LAYER_FUSED_X and conv_in.kernel_size == (1, 1)
                and conv_in.in_channels <= 31 and sum(t.shape[1] for t in u) == conv_in.in_channels):
            # few input channels (the coarse steps' GLOW / AllInOne blocks): the composed first layer with its fused first map takes
            # cat(half, condition, 1) as ONE small tensor -- no 1x1 launch, half the convolution steps
            # (the LAST tensor is the condition, the same for every block of the step: its (condition | 1) is built once per step)
            u1 = ops.concat_channels(list(u[:-1]) + [ops.with_ones(u[-1])])
            u, two = u1[:, :-1], False
        if two and not (len(u) == 2 and ops.VIRTUAL_CAT and conv_in.kernel_size == (1, 1) and conv_in.out_channels <= 64):
            u, two = ops.concat_channels(list(u)), False
        pc_in = self._cat_bank(conv_in, u[0].shape[1]) if two else P(conv_in)
        # the first map is channel-blocked already when the 1x1 kernel that writes it can do so (see below)
        blocked = bool(split_layers and ops.BLOCKED_MAPS and not pc_in.split and pc_in.ks == 1 and pc_in.cout == 64)
        # the first layer in its composed form (3x3 o 1x1 over the sub-network's few input channels + a ones channel: half the
        # convolution steps) where the input is one tensor of <= 31 channels; with FIRST_LAYER_FUSED_X that launch also forms the
        # first map conv_in(u) itself (its residual), so the 1x1 below does not run at all
        first_form = (split_layers and ops.FIRST_LAYER_COMPOSED and not two and conv_in.kernel_size == (1, 1) and conv_in.in_channels <= 31
                      and not pc_in.split)
        fused_x = first_form and ops.FIRST_LAYER_FUSED_X
        pre = ops.first_map_of(self)             # computed by the plan together with the other sub-networks' (ops.first_map_scope)
        if fused_x:
            b = None
        elif pre is not None and pre[0] is u and pre[1] is conv_in and blocked:
            b = pre[2]
        elif two:
            b = ops.conv2d(u[0], pc_in, bias=conv_in.bias, out_blocked=blocked, cat=u[1])
        else:
            b = ops.conv2d(u, pc_in, bias=conv_in.bias, out_blocked=blocked)
        # the maps between the layers are private to this stack: on the split-bf16 kernels they are kept CHANNEL-BLOCKED
        # ([8][H][W][8]: 16-byte accesses in the layer kernel, see cwfa_subnet_layer_split_f32) whenever their consumer reads
        # that layout -- the next layer, and the last convolution if it runs on the split-bf16 3x3 kernel
        pc_out = None if couple is not None else P(conv_out)
        last_reads_blocked = split_layers and ops.BLOCKED_MAPS and (couple is not None or (pc_out.split and pc_out.ks == 3))
        for i, blk in enumerate((self.block2, self.block4, self.block6)):
            if fused:       # 3x3 -> ELU -> 1x1 -> +b -> ELU in one launch, hidden map stays in registers
                if split_layers and i == 0 and first_form:
                    out_blocked = bool(ops.BLOCKED_MAPS)
                    b = ops.subnet_layer_first(u1 if u1 is not None else ops.with_ones(u), b, self._first3(conv_in, blk[0], blk[2]), blk[0].bias, blk[2].bias,
                                               layout=(0 if fused_x else int(blocked)) | (int(out_blocked) << 1))
                    blocked = out_blocked
                    continue
                if split_layers:                                        # both convs on the bf16 matrix pipe
                    out_blocked = ops.BLOCKED_MAPS and (i < 2 or last_reads_blocked)
                    b = ops.subnet_layer(b, self._split3(blk[0], blk[2]), blk[0].bias, None, blk[2].bias,
                                         layout=int(blocked) | (int(out_blocked) << 1))
                    blocked = out_blocked
                else:
                    b = ops.subnet_layer(b, P(blk[0]), blk[0].bias, self._panel(blk[2]), blk[2].bias)
                continue
            h = ops.conv2d(b, P(blk[0]), bias=blk[0].bias, act="elu")
            b = ops.conv2d(h, P(blk[2]), bias=blk[2].bias, residual=b, act2="elu")   # ELU = block3 / block5 / block7x[0]
        if couple is not None:
            return ops.conv3x3_couple(b, self._couple_bank(conv_out), *couple, in_blocked=blocked)
        return ops.conv2d(b, pc_out, bias=conv_out.bias, out=out, in_blocked=blocked)

    def _cat_bank(self, conv, c1):
        w = conv.weight
        pc = self._panels.get(("cat", id(conv), c1))
        if pc is None or pc.version != w._version or pc.src_ptr != w.data_ptr() or pc.epoch != ops.pack_epoch():
            pc = self._panels[("cat", id(conv), c1)] = ops.pack_conv_weight_cat(w, c1)
        return pc

    def _couple_bank(self, conv):
        w, bias = conv.weight, conv.bias
        hit = self._panels.get(("c", id(conv)))
        if hit is not None:
            pc = hit[0]
            stale = (pc.version != w._version or pc.src_ptr != w.data_ptr() or pc.epoch != ops.pack_epoch() or
                     (bias is not None and (pc.version1 != bias._version or pc.src_ptr1 != bias.data_ptr())))
        if hit is None or stale:
            hit = self._panels[("c", id(conv))] = ops.pack_couple_weight(w, bias)
        return hit

    def couple(self, parts, x, out, clamp_kind, clamp, pre_scale, rev, logdet):
        """The whole coupling ``out = A(x | net(cat(parts)))`` with s, t kept in the accumulators of the last convolution
        (coupling_layers.py:87-110; all_in_one_block.py:206-224).  Returns False when this form does not apply (precision
        mode without the split-bf16 kernel, the ``_first`` variant, more than 64 active channels) -- the caller then takes
        affine_parts + ops.affine."""
        n = x.shape[1]
        if AG.tracking(list(parts), x, self):       # the coupling epilogue is an inference form: autograd takes affine_parts + AG.affine
            return False
        if (not self.normal or not ops.couple_fused() or self.conv_type is not nn.Conv2d or self.c_out != 2 * n or n > 64
                or clamp_kind is None or (max(self.n_ch, n) + 64) * x.shape[2] * x.shape[3] * 4 >= 2 ** 31):
            return False
        u = parts[0] if len(parts) == 1 else list(parts)
        self._stack(u, self.block12, self.block72[1], couple=(x, out, clamp_kind, clamp, pre_scale, rev, logdet))
        return True

    def fused_layers(self):
        """The three residual layers run on the fused layer kernel (64 channels, biases present)."""
        return self.n_ch == 64 and all(blk[0].bias is not None and blk[2].bias is not None for blk in (self.block2, self.block4, self.block6))

    def first_conv_of(self, parts):
        """(input tensor, 1x1 module) of this sub-network's first convolution for the condition list ``parts`` if that is ONE tensor
        read by a plain 1x1 with bias (what MergedFirstMaps can stack), else None."""
        if self.conv_type is not nn.Conv2d or self.n_ch != 64:
            return None
        if self.normal:
            if len(parts) != 1:
                return None
            u, conv = parts[0], self.block12
        else:
            n = self.c_in // 2
            if not (len(parts) == 2 and parts[1].shape[1] == n):
                return None
            u, conv = parts[1], self.block1
        if conv.kernel_size != (1, 1) or conv.bias is None or conv.in_channels != u.shape[1]:
            return None
        return u, conv

    def _first3(self, conv0, conv3, conv1):
        srcs = [conv0.weight, conv3.weight, conv1.weight] + ([conv0.bias] if conv0.bias is not None else [])
        key = tuple((t._version, t.data_ptr()) for t in srcs) + (ops.pack_epoch(), bool(ops.FIRST_LAYER_FUSED_X))
        hit = self._panels.get(("f", id(conv3)))
        if hit is None or hit[0] != key:
            hit = self._panels[("f", id(conv3))] = (key, ops.pack_first_layer_weight(conv0.weight, conv0.bias, conv3.weight, conv1.weight))
        return hit[1]

    def _split3(self, conv3, conv1):
        w, w1 = conv3.weight, conv1.weight
        pc = self._panels.get(("s", id(conv3)))
        if (pc is None or pc.version != w._version or pc.src_ptr != w.data_ptr() or pc.version1 != w1._version
                or pc.src_ptr1 != w1.data_ptr() or pc.epoch != ops.pack_epoch()):
            pc = self._panels[("s", id(conv3))] = ops.pack_split_layer_weight(w, w1)
        return pc

    def _panel(self, conv):
        w = conv.weight
        pc = self._panels.get(id(conv))
        if pc is None or pc.version != w._version or pc.src_ptr != w.data_ptr() or pc.epoch != ops.pack_epoch():
            pc = self._panels[id(conv)] = ops.pack_1x1_panel(w)
        return pc

    def affine_parts(self, parts, n_s):
        """(s_raw, t, t_neg_div_sqrt2) for a coupling block, without materialising any concatenation."""
        if self.normal:
            u = parts[0] if len(parts) == 1 else list(parts)
            a = self._stack(u, self.block12, self.block72[1])
            return a[:, :n_s], a[:, n_s:], False
        n = self.c_in // 2
        if len(parts) == 2 and parts[1].shape[1] == n and self.c_out // 2 == n_s:
            mean, om = parts
        else:
            u = parts[0] if len(parts) == 1 else ops.concat_channels(parts)
            mean, om = u[:, :-n], u[:, -n:]
            if self.c_out // 2 != n_s:
                a = self.forward(u)
                return a[:, :n_s], a[:, n_s:], False
        return self._stack(om, self.block1, self.block7[1]), mean, True

    def forward(self, input):
        if self.normal:
            return self._stack(input, self.block12, self.block72[1])
        n = self.c_in // 2
        mean, om = input[:, :-n], input[:, -n:]
        if AG.tracking(input, self):
            return AG.concat([self._stack(om, self.block1, self.block7[1]), AG.neg_div_sqrt2(mean)])
        B, _, H, W = input.shape
        co = self.c_out // 2
        out = torch.empty((B, co + mean.shape[1], H, W), dtype=torch.float32, device=input.device)
        self._stack(om, self.block1, self.block7[1], out=out[:, :co])
        # tail = -mean / sqrt(2)  (networks.py:671): one strided plane pass, y = (x - 0) / (-sqrt2)
        from . import _lib
        import ctypes as C
        mean_p, mbs = ops.planes(mean, "mean")
        div = torch.full((mean.shape[1],), -math.sqrt(2), dtype=torch.float32, device=input.device)
        zero = torch.zeros_like(div)
        _lib.check(_lib.lib().cwfa_channel_affine_f32(
            C.c_void_p(mean_p.data_ptr()), C.c_void_p(out.data_ptr() + 4 * co * H * W), C.c_void_p(div.data_ptr()),
            C.c_void_p(zero.data_ptr()), 1, None, None, B, mean.shape[1], H * W, mbs, out.shape[1] * H * W,
            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "subnet_first tail")
        return out


MERGE_FIRST_MAPS = True      # (tuning / ablation) False: every sub-network runs its own first 1x1 convolution

_merged_banks = {}


def merged_first_maps(jobs):
    """``jobs``: [(sub-network, condition list)] of the blocks of a CAT step.  Where at least two sub-networks read the same tensor
    through a plain 1x1 (they all do: the condition), their banks are stacked and ONE launch writes all first maps, each a
    channel-blocked 64-channel chunk of one tensor.  Returns the dict for ops.first_map_scope (empty when the form does not apply:
    precision other than split / bf16, 3-D sub-networks, ...)."""
    if not (MERGE_FIRST_MAPS and ops._split_bf16 >= 2 and ops.BLOCKED_MAPS):
        return {}
    groups = {}
    for net, parts in jobs:
        fc = net.first_conv_of(parts) if hasattr(net, "first_conv_of") else None
        if fc is not None and ops.FIRST_LAYER_COMPOSED and ops.FIRST_LAYER_FUSED_X and fc[1].in_channels <= 31 and net.fused_layers():
            continue                             # that sub-network's first layer forms its first map itself (subnet_layer_first, x = None)
        if fc is not None:
            groups.setdefault(id(fc[0]), []).append((net, fc[0], fc[1]))
    out = {}
    for members in groups.values():
        u = members[0][1]
        if len(members) < 2 or (64 * len(members) + 64) * u.shape[2] * u.shape[3] * 4 >= 2 ** 31:
            continue
        convs = [m[2] for m in members]
        key = tuple((c.weight._version, c.weight.data_ptr(), c.bias._version, c.bias.data_ptr()) for c in convs) + (ops.pack_epoch(),)
        hit = _merged_banks.get(id(convs[0]))
        if hit is None or hit[0] != key:
            w = torch.cat([c.weight.detach() for c in convs], 0).contiguous()
            # (the fp32 MFMA 1x1 kernel, as for the separate 64-output banks: the launch is bound by its 64 n output planes)
            hit = _merged_banks[id(convs[0])] = (key, ops.pack_conv_weight(w, direct=True), torch.cat([c.bias.detach() for c in convs]).contiguous())
        X = ops.conv2d(u, hit[1], bias=hit[2], out_blocked=True)
        for k, (net, _, conv) in enumerate(members):
            out[id(net)] = (u, conv, X[:, 64 * k:64 * (k + 1)])
    return out


class wavelet_flow_subnetwork2D(wavelet_flow_subnetwork):
    """networks.py:673-682."""

    def __init__(self, c_in, c_out, c_internal=[]):
        super().__init__(c_in, c_out, c_internal)
        self.c_internal = c_internal
        self.init_blocks(nn.Conv2d)


class wavelet_flow_subnetwork2D_first(wavelet_flow_subnetwork):
    """networks.py:684-706: ``normal=False`` and a near-zero last conv (Xavier gain 0.01)."""

    def __init__(self, c_in, c_out, c_internal=[]):
        super().__init__(c_in, c_out, c_internal)
        self.c_internal = c_internal
        self.init_blocks(nn.Conv2d)
        self.normal = False
        self.block7[-1].apply(subnet_initialization_small)


# --------------------------------------------------------------------------------------------------- condition nets
_SHARED_PRELU = nn.PReLU()      # the reference's default-argument instance: ONE parameter shared by every ResidualBlock


class ResidualBlock(nn.Module):
    """2-D residual block 29 -> C followed by Conv3d 1 -> K -> 1 over (H, W, depth).  networks.py:198-242.

    HIP execution: three MFMA convs (PReLU / residual-add / PReLU in the epilogues) + ONE fused stencil kernel for the
    3-D part whose K-channel hidden volume never leaves the CU (cwfa_conv3d_1k1_f32)."""

    def __init__(self, in_channels, out_channels, chans_3D=32, stride=1, downsample=None, activation=_SHARED_PRELU):
        super().__init__()
        if stride != 1:
            raise NotImplementedError("cwfa_amd ResidualBlock: stride 1 only")
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1), activation)
        self.conv2 = nn.Sequential(nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1))
        self.downsample = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1))
        self.relu = activation
        self.conv3d = nn.Sequential(nn.Conv3d(1, chans_3D, kernel_size=3, stride=stride, padding=1), activation,
                                    nn.Dropout3d(), nn.Conv3d(chans_3D, 1, kernel_size=3, stride=stride, padding=1))
        self.bn_out = None
        self.out_channels = out_channels
        self._packed = _Packed()

    def forward(self, x):
        if not isinstance(self.relu, nn.PReLU) or self.relu.weight.numel() != 1:
            raise NotImplementedError("ResidualBlock activation must be a single-parameter PReLU")
        if AG.tracking(x, self):
            # training (CWFA.py:859,1002-1012): one autograd node; in train mode its Dropout3d draws a per-(sample, hidden channel)
            # mask that is folded into the second Conv3d's weights (training.cond_forward_train)
            return AG.cond_net(self, x)
        if self.training:                         # train mode without a graph (networks.py:224: Dropout3d(0.5) is live)
            from . import training
            return training.cond_forward_train(self, x)[0]
        a = self.relu.weight
        P = self._packed.get
        c1, c2, ds = self.conv1[0], self.conv2[0], self.downsample[0]
        hit = _omega_first.get(id(self)) if _omega_first is not None else None
        if hit is not None and hit[0] is x:          # conv1 / downsample of all steps' condition nets ran as one launch (omega_first_scope)
            o, r = hit[1], hit[2]
        else:
            o = ops.conv2d(x, P(c1), bias=c1.bias, act="prelu", prelu_alpha=a)
            r = ops.conv2d(x, P(ds), bias=ds.bias)
        o = ops.conv2d(o, P(c2), bias=c2.bias, residual=r, act2="prelu", prelu_alpha=a)
        k1, k2 = self.conv3d[0], self.conv3d[3]
        return ops.conv3d_1k1(o, k1.weight, k1.bias, a, k2.weight, k2.bias)


MERGE_OMEGA_FIRST = True     # (tuning / ablation) False: every ResidualBlock runs its own conv1 / downsample launches

_omega_first = None          # inside omega_first_scope: {id(ResidualBlock): (views tensor, conv1 output, downsample output)}


class omega_first_scope:
    """The condition nets of all flow steps read the SAME light-field views (CWFA.py:890-899: ``cond_nets[n](cond_input)`` for every
    n), and each ResidualBlock starts with two 3x3 convolutions of that tensor, conv1 (+ PReLU) and downsample (networks.py:212-219,
    229-233).  Within this scope those 2 x n_steps banks run as ONE convolution 29 -> sum of their outputs on the split-bf16 3x3 kernel
    (per-channel PReLU slopes: the blocks' slope on conv1's channels, 1.0 = identity on downsample's), each block then reads its
    two maps as channel-slice views.  Applies in split / bf16 precision to eval-mode blocks outside autograd; anything else leaves
    the blocks to their own launches."""

    def __init__(self, nets, x):
        self.maps = {}
        blocks = []
        for net in nets:
            blk = net.subnetworks[0] if isinstance(net, cond_network) and len(net.subnetworks) == 1 else None
            if not isinstance(blk, ResidualBlock):
                return
            blocks.append(blk)
        if not (MERGE_OMEGA_FIRST and len(blocks) >= 2 and ops._split_bf16 >= 2 and torch.is_tensor(x) and x.dim() == 4 and x.is_cuda):
            return
        cin = blocks[0].conv1[0].in_channels
        total = 2 * sum(b.out_channels for b in blocks)
        if (x.shape[1] != cin or any(b.training or b.conv1[0].in_channels != cin or b.conv1[0].bias is None or b.downsample[0].bias is None
                                     or not isinstance(b.relu, nn.PReLU) or b.relu.weight.numel() != 1 for b in blocks)
                or AG.tracking(x, *blocks) or (total + 64) * x.shape[2] * x.shape[3] * 4 >= 2 ** 31):
            return
        tensors = [t for b in blocks for t in (b.conv1[0].weight, b.conv1[0].bias, b.downsample[0].weight, b.downsample[0].bias, b.relu.weight)]
        key = tuple((id(b),) for b in blocks) + tuple((t._version, t.data_ptr()) for t in tensors) + (ops.pack_epoch(),)
        hit = getattr(blocks[0], "_cwfa_omega_first", None)          # cached ON the first block (dies with it)
        if hit is None or hit[0] != key:
            w = torch.cat([c.weight.detach() for b in blocks for c in (b.conv1[0], b.downsample[0])], 0).contiguous()
            bias = torch.cat([c.bias.detach() for b in blocks for c in (b.conv1[0], b.downsample[0])]).contiguous()
            slopes = torch.cat([v for b in blocks for v in (b.relu.weight.detach().reshape(1).expand(b.out_channels),
                                                            torch.ones(b.out_channels, dtype=torch.float32, device=w.device))]).contiguous()
            pc = ops.pack_conv_weight(w)
            if not (pc.split and pc.ks == 3):
                return
            hit = (key, pc, bias, slopes)
            object.__setattr__(blocks[0], "_cwfa_omega_first", hit)
        y = ops.conv2d(x, hit[1], bias=hit[2], act="prelu", prelu_alpha=hit[3])
        off = 0
        for b in blocks:
            C_ = b.out_channels
            self.maps[id(b)] = (x, y[:, off:off + C_], y[:, off + C_:off + 2 * C_])
            off += 2 * C_

    def __enter__(self):
        global _omega_first
        self.prev, _omega_first = _omega_first, (self.maps or None)
        return self

    def __exit__(self, *exc):
        global _omega_first
        _omega_first = self.prev
        return False


class cond_network(nn.Module):
    """Condition net Omega: light-field views [B,29,H,W] -> [B,C_n,H,W].  networks.py:165-196."""

    def __init__(self, c_in, c_out, n_steps, max_steps=7, n_channels=[], cond_chans=32, net_constructor=None):
        super().__init__()
        self.n_steps = n_steps
        self.global_attention = None
        self.subnetworks = nn.Sequential(ResidualBlock(c_in, c_out, chans_3D=cond_chans))

    def forward(self, lf_img):
        return [self.subnetworks[0](lf_img)]


class GlobalAttention(nn.Module):
    """Conv1d(k3) -> ReLU -> Conv1d(k1) -> sigmoid over the flattened H*W sequence.  networks.py:244-262."""

    def __init__(self, n_chans):
        super().__init__()
        self.m = nn.Sequential(nn.Conv1d(n_chans, n_chans, 3, 1, 1), nn.ReLU(), nn.Conv1d(n_chans, n_chans, 1, 1, 0),
                               nn.Sigmoid())

    def weights_init(self, m):
        if isinstance(m, nn.Conv1d):
            torch.nn.init.xavier_uniform_(m.weight.data)

    def reset(self):
        self.apply(self.weights_init)

    def combine(self, mean, m=None, x=None):
        """att(mean), or fused ``x + m*2*(att-0.5)`` (networks.py:552-554)."""
        return ops.attention_combine(mean, self.m[0].weight, self.m[0].bias, self.m[2].weight, self.m[2].bias, m, x)

    def forward(self, input):
        return self.combine(input)


def drop_path(x, drop_prob: float = 0., training: bool = False):
    """Per-sample stochastic depth: x/keep * Bernoulli(keep).  networks.py:370-385."""
    if drop_prob == 0. or not training:
        return x
    keep = 1 - drop_prob
    B, C = x.shape[0], x.shape[1]
    # floor(keep + U) is Bernoulli(keep) (networks.py:381-383): drawn directly, two tiny launches instead of four
    gate = torch.empty(B, 1, dtype=x.dtype, device=x.device).bernoulli_(keep).div_(keep)
    return ops.scale_channels(x, gate.expand(B, C).contiguous())


class ConvNeXt(nn.Module):
    """u = 1x1(x);  out = GELU(1x1(LayerNorm_{C,H,W}(7x7(u)))) + drop_path(u).  networks.py:468-503."""

    def __init__(self, c_in, c_out, drop_prob=0.1, size=512):
        super().__init__()
        self.drop_prob = drop_prob
        self.input = nn.Conv2d(c_in, c_out, 1, 1)
        self.m = nn.Sequential(nn.Conv2d(c_out, c_out, 7, 1, 3), nn.LayerNorm([c_out, size, size]),
                               nn.Conv2d(c_out, c_out, 1, 1), nn.GELU())
        self._packed = _Packed()

    def forward(self, input):
        P = self._packed.get
        u = ops.conv2d(input, P(self.input), bias=self.input.bias)
        v = ops.conv2d(u, P(self.m[0]), bias=self.m[0].bias)
        ln = self.m[1]
        if tuple(v.shape[1:]) != tuple(ln.normalized_shape):
            raise RuntimeError(f"Given normalized_shape={list(ln.normalized_shape)}, expected input with shape "
                               f"[*, {', '.join(map(str, ln.normalized_shape))}], but got input of size{list(v.shape)}")
        v = ops.layernorm_apply(v, ops.sample_stats(v), ln.weight, ln.bias, ln.eps)
        res = drop_path(u, self.drop_prob, training=self.training)
        return ops.conv2d(v, P(self.m[2]), bias=self.m[2].bias, act="gelu", residual=res)


class LRNN(nn.Module):
    """Low-resolution network: 1x1 conv + UNet(depth 3, wf 8) on the views, plus the mean-volume branch.
    networks.py:505-555."""

    def __init__(self, ch_in, n_depths, use_bias=False, activation=nn.Softplus()):
        super().__init__()
        n_ch = 3
        # the reference builds (and throws away) this stack first; it consumes the torch RNG (networks.py:521-526)
        self.conv3d = nn.Sequential(nn.Conv3d(1, n_ch, 3, 1, 1), activation, nn.Conv3d(n_ch, n_ch, 3, 1, 1), activation,
                                    nn.Conv3d(n_ch, 1, 3, 1, 1))
        self.conv3d = nn.Sequential(ConvNeXt(n_depths, 64, 0.05), ConvNeXt(64, n_depths, 0.05))
        self.attention_3d = GlobalAttention(n_depths)
        self.deconv = nn.Sequential(
            nn.Conv2d(ch_in, n_depths, 1, stride=1, padding=0, bias=use_bias),
            UNet(n_depths, n_depths, depth=3, wf=8, drop_out=0.005, use_bias=use_bias, skip_conn=True, up_mode='upconv',
                 batch_norm=True))
        self.deconv[0].apply(subnet_initialization_positive)
        self._packed = _Packed()

    def forward(self, x_in, mean_vol=None):
        if AG.tracking(x_in, mean_vol, self):
            # training (CWFA.py:880-886,936-950,1002-1006): the whole network as one autograd node (training.lrnn_forward_train /
            # lrnn_backward); BatchNorm on batch statistics in train mode (CWFA.py:532), on the running ones in eval mode
            return AG.lrnn(self, x_in, mean_vol)
        c0 = self.deconv[0]
        x = self.deconv[1](ops.conv2d(x_in, self._packed.get(c0), bias=c0.bias))
        if mean_vol is not None:
            m = self.conv3d[1](self.conv3d[0](mean_vol))
            x = self.attention_3d.combine(mean_vol, m, x)
        return x


class Encoder(nn.Module):
    """networks.py:557-584."""

    def __init__(self, c_in, c_out, n_steps, n_channels=[], use_bias=False):
        super().__init__()
        self.net = LRNN(c_in, c_out, use_bias)

    def forward(self, im_in, mean_vol=None):
        return [self.net(im_in)] if mean_vol is None else [self.net(im_in, mean_vol)]


# --------------------------------------------------------------------------------------------------- the flow builder
def conditional_wavelet_flow(input_volume_shape, condition_shape, st_subnet, conditional_network, n_down_steps=2,
                             use_permutations=False, block_type='RNVP', n_internal_ch=128, n_blocks=1,
                             disable_low_res_input=False, device='cpu'):
    """One GraphINN per wavelet scale; only the last one carries the conditioned flow.  networks.py:264-368.
    Returns ``(cond_net, [GraphINN, ...])``.  The node names, node order, numpy RNG call order and module arguments are
    the reference's, so permutation tables and state_dict keys coincide."""
    global networks_n_chans
    networks_n_chans = n_internal_ch
    cond_net = None
    if conditional_network is None:
        cond_shape_c = list(condition_shape[1:])
    else:
        cond_net = conditional_network().to(device)
        # the reference dry-runs the net on torch.rand to learn this shape (networks.py:281-283); the condition net keeps
        # the spatial size and emits `out_channels` maps, so the shape is known without running anything
        blk = cond_net.subnetworks[0] if hasattr(cond_net, "subnetworks") else None
        if blk is not None and hasattr(blk, "out_channels"):
            torch.rand(condition_shape)                            # keep the torch RNG stream aligned with the reference
            cond_shape_c = [blk.out_channels, condition_shape[2], condition_shape[3]]
        else:
            with torch.no_grad():
                cond_shape_c = list(cond_net(torch.rand(condition_shape, device=device))[-1].shape[1:])
    args_conv_block = {'subnet_constructor': st_subnet}
    INN_block = {'RNVP': Fm.RNVPCouplingBlock, 'GLOW': Fm.GLOWCouplingBlock, 'GIN': Fm.GINCouplingBlock,
                 'AI1': Fm.AllInOneBlock, 'CAT': Fm.ConditionalAffineTransform}.get(block_type, Fm.RNVPCouplingBlock)
    permute_function = PermuteDim

    subnetworks = []
    for k in range(n_down_steps):
        nodes = [Ff.InputNode(*input_volume_shape, name=F'input {k}')]
        nodes.append(Ff.Node(nodes[-1], HaarTransform1D, module_args={'order_by_wavelet': True}, name=F'down_sampling_{k}'))
        n_ch = nodes[-1].output_dims[0][0]
        out0 = int(n_ch * 0.5)
        split1 = Ff.Node(nodes[-1], Fm.Split, {'section_sizes': (out0, n_ch - out0), 'dim': 0}, name=F'Split {k}')
        nodes.append(split1)
        if k == n_down_steps - 1:
            cond = [Ff.ConditionNode(*cond_shape_c, name=F'Condition {k - 1}')]
            if not disable_low_res_input:
                cond.append(Ff.ConditionNode(*cond_shape_c, name=F'Condition I {k - 1}'))
                nodes.append(cond[1])
            nodes.append(cond[0])
            first_subnet = wavelet_flow_subnetwork2D if disable_low_res_input else wavelet_flow_subnetwork2D_first
            nodes.append(Ff.Node(split1.out1, Fm.ConditionalAffineTransform, {'subnet_constructor': first_subnet},
                                 conditions=cond, name=F'Block_net{k}_input'))
            for nn_ in range(1, n_blocks + 1):
                nodes.append(Ff.Node(nodes[-1], permute_function if nn_ % 2 == 0 else Fm.PermuteRandom, {'seed': k + nn_},
                                     name=F'Permute_net{k}_{nn_}'))
                nodes.append(Ff.Node(nodes[-1], INN_block, args_conv_block, conditions=[cond[-1]],
                                     name=F'Block_net{k}_{nn_}'))
            if use_permutations:
                nodes.append(Ff.Node(nodes[-1], Fm.PermuteRandom, {}, name='Permute_final2'))
        nodes.append(Ff.OutputNode(nodes[-1] if k == n_down_steps - 1 else nodes[-1].out1, name=F'Output WVF{k}'))
        nodes.append(Ff.OutputNode(split1.out0, name=F'Output_net{k}'))
        input_volume_shape = split1.output_dims[0]
        subnetworks.append(Ff.GraphINN(nodes))
    return cond_net, subnetworks


# --------------------------------------------------------------------------------------------------- checkpoints
def serialize_INN_step(INN, cond, optimizer, std_train_stats, args, epoch, path, posfix=''):
    """Checkpoint layout of the reference.  networks.py:708-730."""
    path += '/model_step_' + str(args.INN_down_steps) + '__ep_' + str(epoch) + posfix
    torch.save({'epoch': epoch, 'args': args,
                'INN_state_dict': INN.state_dict() if INN else None,
                'condition_state_dict': cond.state_dict() if cond else None,
                'optimizer_state_dict': optimizer.state_dict() if optimizer else None,
                'training_statistics': std_train_stats}, path)


def load_INN_steps(path, prefix='model_step_*__ep_*', epoch=-1):
    """{step: [epoch, file]} with the highest epoch per step (or exactly ``epoch``).  networks.py:732-756."""
    found = {}
    for m in glob.glob(path + '/' + prefix):
        step, it = map(int, re.findall(r'\d+', m.split('/')[-1]))
        if epoch == -1:
            if step in found and it < found[step][0]:
                continue
            found[step] = [it, m]
        elif it == epoch:
            found[step] = [it, m]
    return found
