"""CPU oracle for the CWFA inverse / forward-NLL hot path.

TEST INFRASTRUCTURE ONLY -- not the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  ``cwfa_amd`` never does (tests/test_boundary.py enforces it).

This is a *functional restatement* of the reference's algorithm for the path:
plain functions over a flat ``state_dict`` (same keys as the reference's
modules produce), built from stock PyTorch CPU ops in fp32.  Every function
cites the reference lines it follows (paths relative to the reference root).

Pinning: checked against golden vectors produced by importing the reference
itself in the build container (``oracle/make_golden.py`` ->
``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).  The reference ships
no tests or fixtures of its own, so these generated
vectors are the pin.  Convolution arithmetic itself is PyTorch ATen (the
reference pins torch==1.12.1; oracle and fixtures ran on torch 2.10.0 CPU).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

INV_SQRT2 = 1.0 / math.sqrt(2)


# --------------------------------------------------------------------------- wavelets
def haar1d(x: Tensor, rev: bool) -> Tuple[Tensor, float]:
    """Orthonormal 1-D Haar along the channel (= depth) axis.  INN_utils.py:142-161.

    fwd: lo=(even+odd)/sqrt2 -> channels [0,h), hi=(even-odd)/sqrt2 -> [h,2h).
    rev: even=(lo+hi)/sqrt2, odd=(lo-hi)/sqrt2.  log-det is 0 for rebalance=1
    (INN_utils.py:136-140: (log16 + 4 log .5)/4).
    """
    h = x.shape[1] // 2
    out = torch.empty_like(x)
    if not rev:
        out[:, :h] = x[:, 0::2] + x[:, 1::2]
        out[:, h:] = x[:, 0::2] - x[:, 1::2]
    else:
        out[:, 0::2] = x[:, :h] + x[:, h:]
        out[:, 1::2] = x[:, :h] - x[:, h:]
    return out * INV_SQRT2, 0.0


def haar2d(x: Tensor, rev: bool, order_by_wavelet: bool = False, rebalance: float = 1.0) -> Tuple[Tensor, float]:
    """FrEIA HaarDownsampling.  FrEIA/modules/reshapes.py:191-300.

    fwd [B,C,H,W]->[B,4C,H/2,W/2]: per channel a=(p00+p01+p10+p11), second=(p00-p01+p10-p11),
    third=(p00+p01-p10-p11), d=(p00-p01-p10+p11), times fac_fwd=0.5*rebalance; channel order c*4+j, or
    j*C+c when order_by_wavelet.  rev: transpose of the same matrix times fac_rev=0.5/rebalance.
    The returned jac is a python float = numel * jac_{fwd,rev} (reshapes.py:278,290; NOT negated in rev).
    """
    fac_fwd, fac_rev = 0.5 * rebalance, 0.5 / rebalance
    jac_fwd = (np.log(16.0) + 4 * np.log(fac_fwd)) / 4.0
    jac_rev = (np.log(16.0) + 4 * np.log(fac_rev)) / 4.0
    ndims = x[0].numel()
    sgn = torch.tensor([[1, 1, 1, 1], [1, -1, 1, -1], [1, 1, -1, -1], [1, -1, -1, 1]], dtype=x.dtype)
    if not rev:
        B, C, H, W = x.shape
        p = torch.stack([x[:, :, 0::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 0::2], x[:, :, 1::2, 1::2]], 2)
        y = torch.einsum("jq,bcqhw->bcjhw", sgn, p)              # [B,C,4,h,w]
        y = y.transpose(1, 2).reshape(B, 4 * C, H // 2, W // 2) if order_by_wavelet \
            else y.reshape(B, 4 * C, H // 2, W // 2)
        return y * fac_fwd, ndims * jac_fwd
    B, C4, h, w = x.shape
    C = C4 // 4
    y = x.reshape(B, 4, C, h, w).transpose(1, 2) if order_by_wavelet else x.reshape(B, C, 4, h, w)
    y = y * fac_rev
    p = torch.einsum("jq,bcjhw->bcqhw", sgn, y)
    out = torch.empty(B, C, 2 * h, 2 * w, dtype=x.dtype)
    out[:, :, 0::2, 0::2], out[:, :, 0::2, 1::2] = p[:, :, 0], p[:, :, 1]
    out[:, :, 1::2, 0::2], out[:, :, 1::2, 1::2] = p[:, :, 2], p[:, :, 3]
    return out, ndims * jac_rev


# --------------------------------------------------------------------------- permutations
def invert_perm(perm: np.ndarray) -> np.ndarray:
    inv = np.zeros_like(perm)
    inv[perm] = np.arange(len(perm))
    return inv


def gather_axis(x: Tensor, idx: Tensor, axis: int) -> Tensor:
    """y[..., i, ...] = x[..., idx[i], ...] along ``axis`` (1=channels, 2=rows, 3=cols).
    PermuteRandom: fixed_transforms.py:37-41; PermuteDim: INN_utils.py:73-81."""
    return x.index_select(axis, idx.to(torch.long))


def perm_recipe(k: int, C: int, H: int, W: int, block_type: str = "CAT", n_blocks: int = 4,
                use_perm: bool = True) -> List[dict]:
    """Replays the numpy *global legacy RNG* call sequence of one flow step (step index k, C flow channels).

    networks.py:341-357 with PermuteRandom (fixed_transforms.py:26-28: reseed then permutation(C)),
    PermuteDim (INN_utils.py:61-64: axis drawn BEFORE the reseed, then permutation(H or W)) and, for AI1,
    AllInOneBlock's own unseeded permutation(C) (all_in_one_block.py:147).  Returns one dict per permute
    node, in graph order: {'axis': 1|2|3, 'perm': int64[...]} (+ 'ai1_perm' drawn right after it).
    NOTE: mutates numpy's global RNG exactly as the reference does.
    """
    out = []
    for nn in range(1, n_blocks + 1):
        if nn % 2 == 1:
            np.random.seed(k + nn)
            ent = {"axis": 1, "perm": np.random.permutation(C)}
        else:
            axis = [2, 3][np.random.randint(0, 2)]
            np.random.seed(k + nn)
            ent = {"axis": axis, "perm": np.random.permutation(H if axis == 2 else W)}
        if block_type == "AI1":
            ent["ai1_perm"] = np.random.permutation(C)
        out.append(ent)
    if use_perm:
        out.append({"axis": 1, "perm": np.random.permutation(C)})
    return out


# --------------------------------------------------------------------------- coupling maths
def soft_clamp(a: Tensor, kind: str, clamp: float) -> Tensor:
    """coupling_layers.py:50-60 and its use `s = clamp * f_clamp(a)` (:210,:249,:282,:493)."""
    if kind == "ATAN":
        return clamp * (0.636 * torch.atan(a))
    if kind == "TANH":
        return clamp * torch.tanh(a)
    if kind == "SIGMOID":
        return clamp * (2.0 * (torch.sigmoid(a) - 0.5))
    if kind == "NONE":
        return clamp * a
    raise ValueError(kind)


def affine(x: Tensor, s: Tensor, t: Tensor, rev: bool) -> Tuple[Tensor, Tensor]:
    """fwd y = e^s x + t, j = sum s;  rev y = (x - t) e^{-s}, j = -sum s.  coupling_layers.py:211-217."""
    j = s.sum(dim=tuple(range(1, s.ndim)))
    if rev:
        return (x - t) * torch.exp(-s), -j
    return torch.exp(s) * x + t, j


def subnet(sd: SD, p: str, x: Tensor, first: bool = False) -> Tensor:
    """wavelet_flow_subnetwork2D / _first.  networks.py:641-671 (layers: :621-638).

    normal: b1 = 1x1(block12); three times b <- ELU?(1x1(ELU(3x3(b))) + b); out = 3x3(ELU(b6)) (block72).
    first : input = cat(mean, omega) halves; conv stack on the omega half via block1, out = cat(block7(...), -mean/sqrt2).
    """
    def cv(name, v, pad):
        return F.conv2d(v, sd[p + name + ".weight"], sd.get(p + name + ".bias"), padding=pad)

    if first:
        n = x.shape[1] // 2                                  # networks.py:654-657 (n = c_in//2)
        mean, om = x[:, :-n], x[:, -n:]
        b = cv("block1", om, 0)
    else:
        b = cv("block12", x, 0)
    for i, blk in enumerate(("block2", "block4", "block6")):
        b = cv(blk + ".2", F.elu(cv(blk + ".0", b, 1)), 0) + b
        if i < 2:
            b = F.elu(b)                                     # block3 / block5
    if first:
        b7 = cv("block7.1", F.elu(b), 1)
        return torch.cat((b7, -mean / math.sqrt(2)), 1)
    return cv("block72.1", F.elu(b), 1)


def block_cat(sd: SD, p: str, x: Tensor, c: Sequence[Tensor], rev: bool, first: bool = False,
              clamp: float = 2.0, kind: str = "ATAN") -> Tuple[Tensor, Tensor]:
    """ConditionalAffineTransform.  coupling_layers.py:475-500."""
    cond = torch.cat(list(c), 1) if len(c) > 1 else c[0]
    a = subnet(sd, p + "subnet.", cond, first)
    C = x.shape[1]
    return affine(x, soft_clamp(a[:, :C], kind, clamp), a[:, C:], rev)


def block_two_sided(sd: SD, p: str, x: Tensor, c: Sequence[Tensor], rev: bool, variant: str,
                    clamp: float = 2.0, kind: str = "ATAN") -> Tuple[Tensor, Tensor]:
    """GLOW / RNVP / GIN / NICE.  _BaseCouplingBlock.forward coupling_layers.py:62-87; variants :124-381.

    split (C//2, C-C//2); fwd: y1 = A(x1 | net2(x2,c)), y2 = A(x2 | net1(y1,c)); rev: coupling2 first.
    """
    C = x.shape[1]
    l1, l2 = C // 2, C - C // 2
    x1, x2 = x[:, :l1], x[:, l1:]
    cc = list(c)

    def st(which: int, u: Tensor, n_out: int):
        if variant == "RNVP":                                  # subnet_s{w}, subnet_t{w}  (:193-196)
            return subnet(sd, f"{p}subnet_s{which}.", u), subnet(sd, f"{p}subnet_t{which}.", u)
        if variant == "NICE":                                  # F (which=2) / G (which=1)     (:144-145)
            return None, subnet(sd, p + ("F." if which == 2 else "G."), u)
        a = subnet(sd, f"{p}subnet{which}.", u)               # GLOW / GIN                    (:267-268)
        return a[:, :n_out], a[:, n_out:]

    def couple(xa: Tensor, u: Tensor, which: int, n_out: int):
        s_raw, t = st(which, torch.cat([u] + cc, 1) if cc else u, n_out)
        if variant == "NICE":
            return (xa - t if rev else xa + t), torch.zeros(x.shape[0])
        s = soft_clamp(s_raw, kind, clamp)
        if variant == "GIN":
            s = s - s.mean(1, keepdim=True)                    # :355,:372
            y, _ = affine(xa, s, t, rev)
            return y, torch.zeros(x.shape[0])
        return affine(xa, s, t, rev)

    if not rev:
        y1, j1 = couple(x1, x2, 2, l1)
        y2, j2 = couple(x2, y1, 1, l2)
    else:
        y2, j2 = couple(x2, x1, 1, l2)
        y1, j1 = couple(x1, y2, 2, l1)
    return torch.cat((y1, y2), 1), j1 + j2


def block_onesided(sd: SD, p: str, x: Tensor, c: Sequence[Tensor], rev: bool, clamp: float = 2.0,
                   kind: str = "ATAN") -> Tuple[Tensor, Tensor]:
    """AffineCouplingOneSided.  coupling_layers.py:412-437."""
    C = x.shape[1]
    l1, l2 = C // 2, C - C // 2
    x1, x2 = x[:, :l1], x[:, l1:]
    a = subnet(sd, p + "subnet.", torch.cat([x1] + list(c), 1) if len(c) else x1)
    y2, j = affine(x2, soft_clamp(a[:, :l2], kind, clamp), a[:, l2:], rev)
    return torch.cat((x1, y2), 1), j


def block_ai1(sd: SD, p: str, x: Tensor, c: Sequence[Tensor], rev: bool, gin: bool = False,
              clamp: float = 2.0) -> Tuple[Tensor, Tensor]:
    """AllInOneBlock (stored permutation matrix -- hard or soft --, SOFTPLUS global affine).  all_in_one_block.py:181-268.
    (Householder / reverse-permutation / SIGMOID / EXP variants are pinned on the GPU side directly by the g18 fixtures.)

    fwd: split [C-C//2, C//2]; a = 0.1*subnet(cat(x1,c)); x2 <- x2*exp(clamp*tanh(a_s)) + a_t; then
    (x*scale + offset) through the 0/1 1x1 conv w_perm; scale = 0.1*softplus_{beta=.5}(global_scale).
    rev: (conv1x1(x, w_perm_inv) - offset)/scale first, then the inverse coupling.
    log-det: coupling sum(s) (0 for GIN) + (-1)^rev * HW * sum(log scale) (0 for GIN).
    """
    C = x.shape[1]
    l1, l2 = C - C // 2, C // 2
    n_pix = x[0, :1].numel()
    if gin:
        scale, gj = 1.0, torch.zeros(())
    else:
        scale = 0.1 * F.softplus(sd[p + "global_scale"], beta=0.5)
        gj = torch.log(scale).sum()
    if rev:
        x = (F.conv2d(x, sd[p + "w_perm_inv"]) - sd[p + "global_offset"]) / scale
    x1, x2 = x[:, :l1], x[:, l1:]
    a = subnet(sd, p + "subnet.", torch.cat([x1] + list(c), 1) if len(c) else x1) * 0.1
    s = clamp * torch.tanh(a[:, :l2])
    if gin:
        s = s - s.mean(dim=(1, 2, 3), keepdim=True)
    y2, j = affine(x2, s, a[:, l2:], rev)
    out = torch.cat((x1, y2), 1)
    if not rev:
        out = F.conv2d(out * scale + sd[p + "global_offset"], sd[p + "w_perm"])
    return out, j + (-1) ** int(rev) * n_pix * gj


def actnorm_init(x: Tensor) -> Tuple[Tensor, Tensor]:
    """Data-dependent init.  invertible_resnet.py:54-66: scale = log(1/std_c) (unbiased), bias = -mean_c(x e^scale)."""
    C = x.shape[1]
    flat = x.transpose(0, 1).contiguous().view(C, -1)
    scale = torch.log(1 / flat.std(dim=-1))
    bias = -(flat * scale.exp()[:, None]).mean(dim=-1)
    shp = [1, C] + [1] * (x.ndim - 2)
    return scale.view(shp), bias.view(shp)


def actnorm(scale: Tensor, bias: Tensor, x: Tensor, rev: bool) -> Tuple[Tensor, Tensor]:
    """invertible_resnet.py:68-81."""
    j = (scale.sum() * np.prod(x.shape[2:])).repeat(x.shape[0])
    if rev:
        return (x - bias) / scale.exp(), -j
    return x * scale.exp() + bias, j


# --------------------------------------------------------------------------- condition nets
def omega_net(sd: SD, x: Tensor, p: str = "subnetworks.0.") -> Tensor:
    """cond_network -> ResidualBlock (eval: Dropout3d off).  networks.py:195-196, 229-242.

    2-D: out = PReLU(conv2(PReLU(conv1(x))) + downsample(x));  3-D: [B,1,H,W,D] Conv3d 1->K, PReLU, Conv3d K->1.
    All PReLUs are the ONE shared parameter (networks.py:209 default-arg instance).
    """
    a = sd[p + "relu.weight"]
    out = F.prelu(F.conv2d(x, sd[p + "conv1.0.weight"], sd[p + "conv1.0.bias"], padding=1), a)
    out = F.conv2d(out, sd[p + "conv2.0.weight"], sd[p + "conv2.0.bias"], padding=1)
    out = out + F.conv2d(x, sd[p + "downsample.0.weight"], sd[p + "downsample.0.bias"], padding=1)
    out = F.prelu(out, a)
    v = out.permute(0, 2, 3, 1).unsqueeze(1)
    v = F.prelu(F.conv3d(v, sd[p + "conv3d.0.weight"], sd[p + "conv3d.0.bias"], padding=1), a)
    v = F.conv3d(v, sd[p + "conv3d.3.weight"], sd[p + "conv3d.3.bias"], padding=1)
    return v[:, 0].permute(0, 3, 1, 2)


def _bn(sd: SD, p: str, x: Tensor, train: bool) -> Tensor:
    if train:       # batch statistics (biased variance), eps 1e-5; running stats are not needed for the output
        return F.batch_norm(x, None, None, sd[p + "weight"], sd[p + "bias"], True, 0.0, 1e-5)
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        False, 0.0, 1e-5)


def _unet_convblock(sd: SD, p: str, x: Tensor, train: bool) -> Tensor:
    """UNetConvBlock: conv3x3 -> PReLU -> BN -> conv3x3 -> PReLU -> BN.  unet.py:94-113."""
    for i in (0, 3):
        x = F.conv2d(x, sd[f"{p}block.{i}.weight"], sd.get(f"{p}block.{i}.bias"), padding=1)
        x = F.prelu(x, sd[f"{p}block.{i + 1}.weight"])
        x = _bn(sd, f"{p}block.{i + 2}.", x, train)
    return x


def unet(sd: SD, x: Tensor, p: str = "", depth: int = 3, train: bool = False, drop_u=None, drop_p: float = 0.0) -> Tensor:
    """UNet.forward, skip **add**, upconv.  unet.py:72-91,161-195.  ``F.dropout2d(x, self.drop_out)`` (unet.py:80 after every
    pooling, :86 after every up block; functional default training=True, so it is live in eval mode too) is stochastic: by default
    it is left out (drop_out = 0); with ``drop_u`` -- one uniform [B, C] tensor per site in call order -- it is applied as what it
    computes, x * (u >= p) / (1 - p) per (sample, channel), so a test can hand both sides the same draws."""
    skips, site = [], [0]

    def drop(t):
        if drop_u is None:
            return t
        u = drop_u[site[0]]
        site[0] += 1
        return t * ((u >= drop_p).to(t.dtype) / (1.0 - drop_p)).view(t.shape[0], t.shape[1], 1, 1)

    for i in range(depth):
        x = _unet_convblock(sd, f"{p}down_path.{i}.", x, train)
        if i != depth - 1:
            skips.append(x)
            x = drop(F.adaptive_max_pool2d(x, x.shape[-1] // 2))                # unet.py:79-80
    for i in range(depth - 1):
        up = F.conv_transpose2d(x, sd[f"{p}up_path.{i}.up.weight"], sd.get(f"{p}up_path.{i}.up.bias"), stride=2)
        x = drop(_unet_convblock(sd, f"{p}up_path.{i}.conv_block.", up + skips[-i - 1], train))      # unet.py:85-86
    x = F.conv2d(x, sd[p + "last.0.weight"], sd.get(p + "last.0.bias"))
    return F.prelu(x, sd[p + "last.1.weight"])


def convnext(sd: SD, p: str, x: Tensor, drop_scale: float = 1.0) -> Tensor:
    """ConvNeXt (networks.py:494-503): u = 1x1(x); out = GELU(1x1(LN_{C,H,W}(7x7(u)))) + u * drop_scale.
    drop_scale = 1 in eval; in train drop_path gives u/0.95 w.p. .95 else 0 (networks.py:379-385) -- stochastic, caller's choice."""
    u = F.conv2d(x, sd[p + "input.weight"], sd[p + "input.bias"])
    v = F.conv2d(u, sd[p + "m.0.weight"], sd[p + "m.0.bias"], padding=3)
    v = F.layer_norm(v, v.shape[1:], sd[p + "m.1.weight"], sd[p + "m.1.bias"], 1e-5)
    v = F.gelu(F.conv2d(v, sd[p + "m.2.weight"], sd[p + "m.2.bias"]))
    return v + u * drop_scale


def global_attention(sd: SD, p: str, x: Tensor) -> Tensor:
    """GlobalAttention (networks.py:249-262): Conv1d k3 over the flattened H*W sequence, ReLU, Conv1d k1, sigmoid."""
    s = x.view(x.shape[0], x.shape[1], -1)
    s = F.relu(F.conv1d(s, sd[p + "m.0.weight"], sd[p + "m.0.bias"], padding=1))
    s = torch.sigmoid(F.conv1d(s, sd[p + "m.2.weight"], sd[p + "m.2.bias"]))
    return s.view(x.shape)


def lrnn(sd: SD, x: Tensor, mean_vol: Optional[Tensor] = None, p: str = "net.", train: bool = False) -> Tensor:
    """Encoder -> LRNN.forward.  networks.py:544-555, 573-584.
    x = UNet(1x1(x));  if mean: x += ConvNeXt2(ConvNeXt1(mean)) * 2 * (attention(mean) - 0.5)."""
    y = F.conv2d(x, sd[p + "deconv.0.weight"], sd.get(p + "deconv.0.bias"))
    y = unet(sd, y, p + "deconv.1.", 3, train)
    if mean_vol is not None:
        m = convnext(sd, p + "conv3d.1.", convnext(sd, p + "conv3d.0.", mean_vol))
        y = y + m * 2 * (global_attention(sd, p + "attention_3d.", mean_vol) - 0.5)
    return y


# --------------------------------------------------------------------------- one flow step (GraphINN semantics)
def step_layout(block_type: str = "CAT", n_blocks: int = 4, use_perm: bool = True) -> List[Tuple[str, int]]:
    """module_list order of one conditional step (networks.py:305-366 + topological order graph_inn.py:429-473):
    0 Haar1D, 1 Split, 2 first CAT, then (permute, block) x n_blocks, then the final PermuteRandom."""
    lay = [("haar", 0), ("split", 1), ("cat_first", 2)]
    i = 3
    for nn in range(1, n_blocks + 1):
        lay.append(("perm", i))
        lay.append((block_type, i + 1))
        i += 2
    if use_perm:
        lay.append(("perm", i))
    return lay


def flow_step(sd: SD, inputs, c: Sequence[Tensor], rev: bool, axes: Dict[int, int], block_type: str = "CAT",
              n_blocks: int = 4, use_perm: bool = True) -> Tuple[object, Tensor]:
    """One GraphINN step.  graph_inn.py:242-326 over the graph of networks.py:305-366.

    c = [omega ('Condition I'), mean detail ('Condition')] (graph condition_nodes order, networks.py:333-335).
    fwd: x [B,D,H,W] -> ((z, low), logdet[B]);  rev: (z, low) -> (x, logdet[B]).
    ``axes``: module index -> gather axis (1 channels / 2 rows / 3 cols) for every permute node (PermuteDim's axis is
    not in the state_dict, INN_utils.py:61).
    """
    om, mean = c[0], c[1]
    lay = step_layout(block_type, n_blocks, use_perm)[2:]
    B = (inputs if torch.is_tensor(inputs) else inputs[0]).shape[0]
    jac = torch.zeros(B)

    def run(kind, i, v):
        p = f"module_list.{i}."
        if kind == "perm":
            return gather_axis(v, sd[p + ("perm_inv" if rev else "perm")], axes[i]), torch.zeros(B)
        if kind == "cat_first":
            return block_cat(sd, p, v, [mean, om], rev, first=True)
        if kind == "CAT":
            return block_cat(sd, p, v, [om], rev)
        if kind in ("GLOW", "RNVP", "GIN"):
            return block_two_sided(sd, p, v, [om], rev, kind)
        if kind == "AI1":
            return block_ai1(sd, p, v, [om], rev)
        raise ValueError(kind)

    if not rev:
        y, _ = haar1d(inputs, False)
        h = y.shape[1] // 2
        low, v = y[:, :h], y[:, h:]
        for kind, i in lay:
            v, j = run(kind, i, v)
            jac = jac + j
        return (v, low), jac
    v, low = inputs
    for kind, i in reversed(lay):
        v, j = run(kind, i, v)
        jac = jac + j
    x, _ = haar1d(torch.cat((low, v), 1), True)
    return x, jac


# --------------------------------------------------------------------------- graph topology / fixed 1x1
def concat(xs: Sequence[Tensor], rev_input: Optional[Tensor] = None, sizes: Optional[Sequence[int]] = None):
    """Concat along the channel axis: fwd ``torch.cat`` (graph_topology.py:136-143), rev ``torch.split`` into ``sizes``;
    log-det 0."""
    if rev_input is not None:
        return list(torch.split(rev_input, list(sizes), dim=1)), 0
    return torch.cat(list(xs), dim=1), 0


def fixed1x1conv(M: Tensor, x: Tensor, rev: bool) -> Tuple[Tensor, float]:
    """Fixed1x1Conv (fixed_transforms.py:95-133): fwd ``conv2d(x, M^T[...,None,None])``, rev with (M^T)^-1;
    log-det = +-log|det M| * H*W."""
    w = M.t().inverse() if rev else M.t()
    n_pix = x.shape[2] * x.shape[3]
    j = float(torch.slogdet(M)[1]) * n_pix
    return F.conv2d(x, w.reshape(*M.shape, 1, 1)), (-j if rev else j)


# --------------------------------------------------------------------------- pipelines
def pyramid_forward(x: Tensor, n_steps: int) -> List[Tensor]:
    """gt_cache of evaluate_INN_forward (CWFA.py:146-195): level n+1 = low half of Haar1D(level n)."""
    out = [x]
    for _ in range(n_steps):
        y, _ = haar1d(out[-1], False)
        out.append(y[:, : y.shape[1] // 2])
    return out


def nll_terms(z: Tensor, logdet: Tensor) -> Tuple[float, float, int]:
    """Shard-local sums for the NLL (CWFA.py:970-978): (sum z^2, sum logdet, B) in float64."""
    return float(z.double().pow(2).sum()), float(logdet.double().sum()), int(z.shape[0])


def nll_from_terms(sumsq: float, sumlogdet: float, B: int, numel_total: int) -> float:
    """NLL = (0.5*||Z||^2 - mean_b logdet) / numel(volume batch).  CWFA.py:978."""
    return (0.5 * sumsq - sumlogdet / B) / numel_total


def step_log_likelihood(z: Tensor, logdet: Tensor, numel_per_sample: int) -> Tensor:
    """Per-sample log-likelihood of one step, float64[B]: -(0.5*||z_b||^2 - logdet_b) / numel_b -- CWFA.py:183-186 with
    the norm taken per volume (for B = 1 it is -curr_LL_loss); the score main.py:78-80 thresholds."""
    return -(0.5 * z.double().flatten(1).pow(2).sum(1) - logdet.double()) / numel_per_sample


def inverse_pass(steps: Sequence[dict], low: Tensor, cond_input: Tensor, mean_cache: Sequence[Tensor],
                 lrnn_sd: Optional[SD] = None, lrnn_train: bool = False) -> List[Tensor]:
    """The reconstruction loop, T=0 (z = 0), n_samples=1.  CWFA.py:865-924.

    steps[n] = {'inn': sd, 'omega': sd, 'axes': {...}, 'block_type', 'n_blocks', 'use_perm'} for n = 0..S-2.
    If lrnn_sd is given, ``low`` is ignored and the lowest-resolution volume is LRNN(cond_input, mean_cache[S-2])
    (CWFA.py:882); otherwise ``low`` is the synthetic lowest-resolution volume (configs 1-2).
    Returns the list of volumes from coarsest to finest (last = full-resolution reconstruction).
    """
    S1 = len(steps)
    up = lrnn(lrnn_sd, cond_input, mean_cache[S1 - 1], train=lrnn_train) if lrnn_sd is not None else low
    vols = [up]
    for n in range(S1 - 1, -1, -1):
        st = steps[n]
        om = omega_net(st["omega"], cond_input)
        z = torch.zeros(up.shape[0], up.shape[1], up.shape[2], up.shape[3])
        up, _ = flow_step(st["inn"], (z, up), [om, mean_cache[n]], True, st["axes"], st.get("block_type", "CAT"),
                          st.get("n_blocks", 4), st.get("use_perm", True))
        vols.append(up)
    return vols


# ---------------------------------------------------------------------------------------------------------------------
# Mean-volume cache and output step
def mean_volume_cache(levels: Sequence[Tensor]) -> List[Tensor]:
    """``[gt[0, ::2] - gt[0, 1::2] for gt in gt_cache]`` (CWFA.py:655): even minus odd depth planes of the first sample."""
    return [gt[0, ::2] - gt[0, 1::2] for gt in levels]


def denormalise_prediction(stored: Tensor, std_vols: Tensor, mean_vols: Tensor) -> Tensor:
    """CWFA.py:1041: ``(stored[0] * 2**len(stored)) * std + mean`` (len = the batch length of the stored tensor)."""
    return (stored[0] * 2 ** len(stored)) * std_vols + mean_vols


def denormalise_ground_truth(gt: Tensor, std_vols: Tensor, mean_vols: Tensor) -> Tensor:
    """CWFA.py:1037-1038: ``gt[0] * std + mean`` minus its minimum."""
    v = gt[0] * std_vols + mean_vols
    return v - v.min()


# ---------------------------------------------------------------------------------------------------------------------
# Lenslet views
def extract_views(image: Tensor, coords_yx, subimage_shape, mean: float = 0.0, std: float = 1.0) -> Tensor:
    """XLFMDatasetFull.extract_views XLFMDataset.py:212-242, then (v - mean) / std (CWFA.py:796-797).
    Window of subimage_shape around each lenslet, lower bounds clamped to 0 (:236-237), upper bounds clipped by slicing
    (:238), the clipped patch written into the bottom-right corner of a zero view (:239)."""
    sh, sw = int(subimage_shape[0]), int(subimage_shape[1])
    hh, hw = sh // 2, sw // 2
    out = torch.zeros((image.shape[0], len(coords_yx), sh, sw), dtype=image.dtype)
    for n, (cy, cx) in enumerate(coords_yx):
        cy, cx = int(cy), int(cx)
        patch = image[:, 0, max(cy - hh, 0): cy + hh, max(cx - hw, 0): cx + hw]
        out[:, n, sh - patch.shape[1]:, sw - patch.shape[2]:] = patch
    return (out - mean) / std
