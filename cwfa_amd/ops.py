"""Tensor-level wrappers over the C ABI (include/cwfa_hip.h).  PyTorch-ROCm provides device memory and the current
HIP stream; every computation happens in libcwfa_hip.so.  Inputs must be fp32 tensors on a HIP device -- a CPU tensor
raises (there is no CPU path in the product; the CPU restatement lives in oracle/ and is test infrastructure)."""
import ctypes as C

import torch

from . import _lib
from ._lib import AffineStage, Chain, ConvOpts, check

__all__ = ["haar1d", "haar2d", "gather", "affine", "channel_affine", "chain_inv", "chain_fwd", "pack_conv_weight",
           "conv2d", "conv2d_wgrad", "elu_bwd", "set_precision", "gelu_add", "gelu_bwd", "layernorm_bwd", "attention_bwd", "plane_affine", "bn_bwd_stats", "bn_act_bwd", "maxpool2_bwd", "chain_bwd", "chain_inv_bwd", "prelu_bwd", "conv3d_1k1_backward", "pack_1x1_panel", "pack_split_layer_weight", "subnet_layer", "conv3d_1k1", "channel_stats", "bn_fold", "bn_running_update", "maxpool", "sample_stats", "layernorm_apply",
           "attention_combine", "scale_channels", "axpby", "stage"]


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _dev(t, name="tensor"):
    if not torch.is_tensor(t):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: cwfa_amd runs on MI355X only -- got a {t.device} tensor (no CPU fallback exists)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t


def planes(t, name="tensor"):
    """Return (tensor, batch_stride) with contiguous [H,W] planes and channel stride H*W (channel-sliced views of a
    contiguous NCHW tensor qualify without a copy)."""
    _dev(t, name)
    if t.dim() != 4:
        raise ValueError(f"{name}: expected [B,C,H,W], got {tuple(t.shape)}")
    B, Cc, H, W = t.shape
    ok = (W == 1 or t.stride(3) == 1) and (H == 1 or t.stride(2) == W) and (Cc == 1 or t.stride(1) == H * W)
    if not ok or (B > 1 and t.stride(0) < Cc * H * W):
        t = t.contiguous()
    return t, (t.stride(0) if B > 1 else Cc * H * W)


def _idx(perm, name="perm"):
    if perm is None:
        return None
    if not perm.is_cuda or perm.dtype != torch.int64:
        raise RuntimeError(f"{name}: index table must be an int64 tensor on the HIP device")
    return perm if perm.is_contiguous() else perm.contiguous()


# ------------------------------------------------------------------------------------------------ wavelets
def haar1d(x, rev=False, lo=None, hi=None):
    """fwd: x[B,D,H,W] -> one [B,D,H,W] tensor (lo | hi halves).  rev: that layout (or separate lo, hi) -> x."""
    L = _lib.lib()
    if not rev:
        x, xbs = planes(x, "x")
        B, D, H, W = x.shape
        out = torch.empty((B, D, H, W), dtype=x.dtype, device=x.device)
        h = D // 2
        check(L.cwfa_haar1d_fwd_f32(_p(x), _p(out), C.c_void_p(out.data_ptr() + 4 * h * H * W), B, D, H * W, xbs,
                                    D * H * W, D * H * W, _stream()), "haar1d_fwd")
        return out
    if lo is None:
        x, xbs = planes(x, "x")
        B, D, H, W = x.shape
        h = D // 2
        lo_p, hi_p, lbs, hbs = _p(x), C.c_void_p(x.data_ptr() + 4 * h * H * W), xbs, xbs
        keep = (x,)
    else:
        lo, lbs = planes(lo, "lo")
        hi, hbs = planes(hi, "hi")
        B, h, H, W = lo.shape
        D = 2 * h
        lo_p, hi_p = _p(lo), _p(hi)
        keep = (lo, hi)
    out = torch.empty((B, D, H, W), dtype=torch.float32, device=keep[0].device)
    check(L.cwfa_haar1d_inv_f32(lo_p, hi_p, _p(out), B, D, H * W, lbs, hbs, D * H * W, _stream()), "haar1d_inv")
    return out


def haar2d(x, rev, order_by_wavelet, fac):
    L = _lib.lib()
    x = _dev(x, "x").contiguous()
    if not rev:
        B, Cc, H, W = x.shape
        out = torch.empty((B, 4 * Cc, H // 2, W // 2), dtype=x.dtype, device=x.device)
        check(L.cwfa_haar2d_fwd_f32(_p(x), _p(out), B, Cc, H, W, int(order_by_wavelet), float(fac), _stream()),
              "haar2d_fwd")
        return out
    B, C4, h, w = x.shape
    out = torch.empty((B, C4 // 4, 2 * h, 2 * w), dtype=x.dtype, device=x.device)
    check(L.cwfa_haar2d_inv_f32(_p(x), _p(out), B, C4 // 4, 2 * h, 2 * w, int(order_by_wavelet), float(fac), _stream()),
          "haar2d_inv")
    return out


def haar3d(x, rev=False, order_by_wavelet=True, fac=0.5):
    """The 2 x 2 x 2 Haar tile in one launch: fwd x[B,D,H,W] -> haar2d(haar1d(x)) [B,4D,H/2,W/2]; rev the inverse.
    Bit-identical to the two-launch composition (``fac`` = haar2d's factor: 0.5 * rebalance)."""
    L = _lib.lib()
    if not rev:
        x, xbs = planes(x, "x")
        B, D, H, W = x.shape
        out = torch.empty((B, 4 * D, H // 2, W // 2), dtype=torch.float32, device=x.device)
        check(L.cwfa_haar3d_fwd_f32(_p(x), _p(out), B, D, H, W, int(order_by_wavelet), float(fac), xbs, _stream()), "haar3d_fwd")
        return out
    y = _dev(x, "y").contiguous()
    B, D4, h, w = y.shape
    if D4 % 8:
        raise ValueError(f"haar3d: {D4} channels are not 4 x an even depth")
    out = torch.empty((B, D4 // 4, 2 * h, 2 * w), dtype=torch.float32, device=y.device)
    check(L.cwfa_haar3d_inv_f32(_p(y), _p(out), B, D4 // 4, 2 * h, 2 * w, int(order_by_wavelet), float(fac), out.stride(0), _stream()),
          "haar3d_inv")
    return out


def gather(x, perm, axis):
    L = _lib.lib()
    x, xbs = planes(x, "x")
    perm = _idx(perm)
    B, Cc, H, W = x.shape
    if perm.numel() != x.shape[axis]:
        raise ValueError(f"perm of length {perm.numel()} does not match axis {axis} of {tuple(x.shape)}")
    out = torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device)
    check(L.cwfa_gather_f32(_p(x), _p(perm), _p(out), B, Cc, H, W, axis, xbs, Cc * H * W, _stream()), "gather")
    return out


# ------------------------------------------------------------------------------------------------ affine / chains
def stage(s_raw=None, t=None, clamp_kind="ATAN", clamp=2.0, pre_scale=1.0, t_neg_div_sqrt2=False, perm=None, axis=1,
          gin=False):
    """Describe one coupling stage (see cwfa_affine_stage).  Returns (ctypes struct, keep-alive tuple)."""
    st = AffineStage()
    keep = []
    if s_raw is not None:
        s_raw, sbs = planes(s_raw, "s_raw")
        st.s_raw, st.s_bs = s_raw.data_ptr(), sbs
        keep.append(s_raw)
    if t is not None:
        t, tbs = planes(t, "t")
        st.t, st.t_bs = t.data_ptr(), tbs
        keep.append(t)
    st.clamp_kind = _lib.CLAMP[clamp_kind]
    st.clamp = float(clamp)
    st.pre_scale = float(pre_scale)
    st.t_neg_div_sqrt2 = int(bool(t_neg_div_sqrt2))
    if perm is not None:
        perm = _idx(perm)
        st.perm = perm.data_ptr()
        keep.append(perm)
    st.perm_axis = int(axis)
    st.gin = int(bool(gin))
    return st, tuple(keep)


def affine(x, st_keep, rev, shape=None, logdet=None, sumsq=None, out=None):
    """y = A(gather(x)) for one stage.  x may be None (zeros) if `shape` is given.  `out` may be a channel-slice view
    (also x itself when the stage has no gather)."""
    L = _lib.lib()
    st, keep = st_keep
    if x is not None:
        x, xbs = planes(x, "x")
        shape = tuple(x.shape)
        dev = x.device
    else:
        xbs = 0
        dev = keep[0].device
    B, Cc, H, W = shape
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=dev)
        ybs = Cc * H * W
    else:
        o2, ybs = planes(out, "out")
        if o2.data_ptr() != out.data_ptr() or tuple(out.shape) != tuple(shape):
            raise ValueError("affine: `out` must be a [B,C,H,W] view with contiguous planes")
    check(L.cwfa_affine_f32(_p(x), _p(out), C.byref(st), int(bool(rev)), B, Cc, H, W, xbs, ybs, _p(logdet),
                            _p(sumsq), _stream()), "affine")
    return out


def affine_bwd(x, g, st_keep, rev, gld=None, want=(True, True, True)):
    """Backward of ``affine(x, st, rev)`` for a stage without a gather: (dL/dx, dL/d s_raw, dL/d t) from g = dL/dy and the
    upstream gradient ``gld`` [B] of the per-sample log-det; entries of ``want`` switch the three outputs."""
    L = _lib.lib()
    st, keep = st_keep
    x, xbs = planes(x, "x")
    g, gbs = planes(g, "g")
    B, Cc, H, W = x.shape
    if tuple(g.shape) != (B, Cc, H, W):
        raise ValueError("affine_bwd: g and x differ in shape")
    mk = lambda on: torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device) if on else None    # noqa: E731
    gx, gs, gt = mk(want[0]), mk(want[1] and bool(st.s_raw)), mk(want[2] and bool(st.t))
    if gld is not None:
        gld = _dev(gld, "gld").contiguous()
    check(L.cwfa_affine_bwd_f32(_p(x), _p(g), C.byref(st), int(bool(rev)), B, Cc, H, W, xbs, gbs, _p(gld), _p(gx), _p(gs), _p(gt),
                                _stream()), "affine_bwd")
    return gx, gs, gt


def channel_affine(x, scale, shift, inverse=False, perm_in=None, perm_out=None):
    L = _lib.lib()
    x, xbs = planes(x, "x")
    B, Cc, H, W = x.shape
    for t in (scale, shift, perm_in, perm_out):
        if t is not None and t.numel() != Cc:
            raise ValueError("channel_affine: scale / shift / permutation tables must hold one entry per channel")
    out = torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device)
    check(L.cwfa_channel_affine_f32(_p(x), _p(out), _p(scale), _p(shift), int(bool(inverse)), _p(_idx(perm_in)),
                                    _p(_idx(perm_out)), B, Cc, H * W, xbs, Cc * H * W, _stream()), "channel_affine")
    return out


def _chain(stages, tables=None):
    if len(stages) > _lib.CHAIN_MAX:
        raise ValueError(f"chain of {len(stages)} stages exceeds CWFA_CHAIN_MAX={_lib.CHAIN_MAX}")
    ch = Chain()
    ch.n_stages = len(stages)
    keep = []
    for k, (st, kp) in enumerate(stages):
        ch.stage[k] = st
        keep.append(kp)
    if tables is not None:
        ch.src_c, ch.src_h = (t.data_ptr() for t in tables)
        keep.append(tables)
    return ch, keep


def chain_tables(perms, final_perm, C_, H, W, device):
    """The gathers of a chain composed per axis (see cwfa_chain.src_* in include/cwfa_hip.h).  ``perms``: per stage in
    execution order (table or None, axis).  Returns two int32 device tensors [n+1, C], [n+1, H]; built once per plan and
    direction (a handful of tiny index ops) and reused by every launch.  (Column permutations are not composed: the
    kernels move the values between threads instead.)"""
    idx = {1: torch.arange(C_, device=device), 2: torch.arange(H, device=device)}
    if final_perm is not None:
        idx[1] = final_perm.to(device)[idx[1]]
    n = len(perms)
    rows = {1: [None] * (n + 1), 2: [None] * (n + 1)}
    for k in range(n - 1, -1, -1):
        for ax in (1, 2):
            rows[ax][k] = idx[ax]
        table, axis = perms[k]
        if table is not None and axis in (1, 2):
            idx[axis] = table.to(device)[idx[axis]]
    for ax in (1, 2):
        rows[ax][n] = idx[ax]
    return tuple(torch.stack(rows[ax]).to(torch.int32).contiguous() for ax in (1, 2))


def chain_inv(z, low, stages, logdet=None, tables=None):
    """x[B,2C,H,W] = Haar1D^-1(cat[low, A_{K-1}^-1(...A_0^-1(z))]); z may be None (= zeros).  ``tables``: chain_tables()
    of the stages' gathers (optional, speed only)."""
    L = _lib.lib()
    low, lbs = planes(low, "low")
    B, Cc, H, W = low.shape
    zbs = 0
    if z is not None:
        z, zbs = planes(z, "z")
        if tuple(z.shape) != tuple(low.shape):
            raise ValueError(f"z {tuple(z.shape)} and low {tuple(low.shape)} differ")
    ch, keep = _chain(stages, tables)
    out = torch.empty((B, 2 * Cc, H, W), dtype=torch.float32, device=low.device)
    rec = chain_event_sink
    if rec is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(L.cwfa_chain_inv_f32(_p(z), _p(low), _p(out), C.byref(ch), B, Cc, H, W, zbs, lbs, 2 * Cc * H * W, _p(logdet),
                               _stream()), "chain_inv")
    if rec is not None:
        e1.record()
        rec.append(("inv", B, Cc, H, W, len(stages), z is not None, e0, e1))
    return out


def chain_fwd(x, stages, final_perm=None, logdet=None, sumsq=None, tables=None):
    """(z, low) for x[B,2C,H,W].  ``tables``: chain_tables() of the stages' gathers INCLUDING final_perm (optional)."""
    L = _lib.lib()
    x, xbs = planes(x, "x")
    B, D, H, W = x.shape
    Cc = D // 2
    ch, keep = _chain(stages, tables)
    low = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
    z = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
    rec = chain_event_sink
    if rec is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(L.cwfa_chain_fwd_f32(_p(x), _p(low), _p(z), C.byref(ch), _p(_idx(final_perm)), B, Cc, H, W, xbs, Cc * H * W,
                               Cc * H * W, _p(logdet), _p(sumsq), _stream()), "chain_fwd")
    if rec is not None:
        e1.record()
        rec.append(("fwd", B, Cc, H, W, len(stages), True, e0, e1))
    return z, low


chain_event_sink = None     # list collecting (direction, B, C, H, W, stages, z_read, start_event, end_event); set by bench.py only


def _chain_grads(grads, shape):
    B, Cc, H, W = shape
    gr = _lib.ChainGrads()
    for k, (ds, dt) in enumerate(grads):
        for name, t in (("ds", ds), ("dt", dt)):
            if t is None:
                continue
            t2, bs = planes(t, name)
            if t2.data_ptr() != t.data_ptr() or tuple(t.shape) != (B, Cc, H, W):
                raise ValueError(f"chain backward: {name}[{k}] must be a [B,C,H,W] view with contiguous planes")
            getattr(gr, name)[k] = t.data_ptr()
            getattr(gr, name + "_bs")[k] = bs
    return gr


def chain_inv_bwd(xhat, gt, stages, grads, gscale, loss_kind=2, accumulate=False, want_latent_grad=False, want_low_grad=False):
    """Backward of gscale-weighted sum |xhat - gt|^p (p = loss_kind) through the inverse pass that produced ``xhat``;
    ``stages`` in FORWARD order.  ``loss_kind`` 0: ``gt`` IS an upstream gradient dL/dxhat (times gscale).  Returns the float64[1]
    tensor sum |xhat - gt|^p -- or, with ``want_latent_grad`` / ``want_low_grad``, the tuple (sum, dL/dz, dL/dlow) of the inverse
    pass's two inputs (None where not wanted; dL/dlow needs loss_kind 0)."""
    L = _lib.lib()
    xhat, xbs = planes(xhat, "xhat")
    gt, gbs = planes(gt, "gt")
    if tuple(xhat.shape) != tuple(gt.shape):
        raise ValueError("chain_inv_bwd: xhat and gt differ in shape")
    B, D, H, W = xhat.shape
    Cc = D // 2
    ch, keep = _chain(stages)
    gr = _chain_grads(grads, (B, Cc, H, W))
    loss = torch.zeros(1, dtype=torch.float64, device=xhat.device)
    if want_low_grad and loss_kind != 0:
        raise ValueError("chain_inv_bwd: dL/dlow is formed from an upstream gradient (loss_kind 0)")
    gz = torch.empty((B, Cc, H, W), dtype=torch.float32, device=xhat.device) if want_latent_grad else None
    glow = torch.empty((B, Cc, H, W), dtype=torch.float32, device=xhat.device) if want_low_grad else None
    check(L.cwfa_chain_inv_bwd_f32(_p(xhat), _p(gt), C.byref(ch), C.byref(gr), B, Cc, H, W, xbs, gbs, float(gscale), int(loss_kind),
                                   int(bool(accumulate)), _p(loss), _p(gz), _p(glow), _stream()), "chain_inv_bwd")
    if want_latent_grad or want_low_grad:
        return loss, gz, glow
    return loss


def chain_bwd(z, stages, grads, final_perm=None, gscale=0.0, ldscale=0.0, gz=None, want_input_grad=False, accumulate=False, gld=None):
    """Backward of L = gscale*0.5*sum z^2 - ldscale*sum logdet (+ <gz, z>) through the chain that produced ``z``
    (``chain_fwd`` with the same ``stages`` / ``final_perm``).  ``grads[k] = (ds_raw_k, dt_k)``: preallocated
    [B,C,H,W] views (contiguous planes) or None.  Returns dL/d(detail band) if ``want_input_grad``."""
    L = _lib.lib()
    z, zbs = planes(z, "z")
    B, Cc, H, W = z.shape
    ch, keep = _chain(stages)
    gr = _chain_grads(grads, (B, Cc, H, W))
    gzbs = 0
    if gz is not None:
        gz, gzbs = planes(gz, "gz")
    gv0 = torch.empty((B, Cc, H, W), dtype=torch.float32, device=z.device) if want_input_grad else None
    if gld is not None:                      # upstream gradient of the per-sample log-det (autograd)
        gld = _dev(gld, "gld").contiguous()
        if gld.numel() != B:
            raise ValueError("chain_bwd: gld must hold one value per sample")
    check(L.cwfa_chain_bwd_f32(_p(z), _p(gz), C.byref(ch), C.byref(gr), _p(_idx(final_perm)), _p(gv0), B, Cc, H, W, zbs, gzbs,
                               Cc * H * W, float(gscale), float(ldscale), int(bool(accumulate)), _p(gld), _stream()), "chain_bwd")
    return gv0


# ------------------------------------------------------------------------------------------------ convolutions
class PackedConv:
    """Kernel-layout image of one filter bank (built once per weight version on the device)."""
    __slots__ = ("packed", "cout", "cin", "ks", "transposed", "version", "src_ptr", "split", "epoch", "version1", "src_ptr1", "cat_c1",
                 "cat_from", "short")

    def __init__(self, packed, cout, cin, ks, transposed, version, src_ptr, split=False):
        self.packed, self.cout, self.cin, self.ks = packed, cout, cin, ks
        self.transposed, self.version, self.src_ptr = transposed, version, src_ptr
        self.split = split          # weights held as three bf16 pieces for the split-bf16 kernels (set_precision)
        self.epoch = _pack_epoch    # set_option() generation this image was built under (caches re-pack on a change)


def pack_conv_weight(w, transposed=False, direct=False):
    """w: [Cout,Cin,k,k] (k in 1,3,7), or with transposed=True a ConvTranspose2d weight [Cin,Co,2,2].  ``direct``: the fp32 MFMA
    kernels' layout whatever the precision mode (banks whose launch is bound by its output stores, not by the matrix pipe)."""
    L = _lib.lib()
    w = _dev(w, "weight").detach().contiguous()
    if transposed:
        cin, co, kh, kw = w.shape
        if (kh, kw) != (2, 2):
            raise ValueError("only 2x2 stride-2 transposed convolutions are supported")
        cout, ks = 4 * co, 1
    else:
        cout, cin, ks, kw = w.shape
        if ks != kw:
            raise ValueError("square kernels only")
    if direct:
        pass
    elif _split_bf16 and ks == 1 and cout >= 128:
        packed = torch.empty(L.cwfa_conv_split_packed_bytes(cout, cin, ks), dtype=torch.uint8, device=w.device)
        check(L.cwfa_conv_split_pack_f32(_p(w), _p(packed), cout, cin, ks, int(transposed), _stream()), "conv_split_pack")
        return PackedConv(packed, cout, cin, ks, transposed, w._version, w.data_ptr(), split=True)
    # (bf16 mode: one product per multiply-add -- there the 32-channel tiling beats the fp32 Winograd kernel for 17 .. 32 outputs too)
    narrow_max = 32 if _plain_bf16 else SPLIT_3X3_NARROW_MAX
    if not direct and _split_bf16 >= 2 and ks == 3 and cout >= SPLIT_3X3_MIN_COUT and (cout > 32 or (cout <= narrow_max and cin >= 29)):
        packed = torch.empty(L.cwfa_conv3x3_split_packed_bytes(cout, cin), dtype=torch.uint8, device=w.device)
        check(L.cwfa_conv3x3_split_pack_f32(_p(w), _p(packed), cout, cin, _stream()), "conv3x3_split_pack")
        return PackedConv(packed, cout, cin, ks, transposed, w._version, w.data_ptr(), split=True)
    if not direct and SPLIT_7X7 and _split_bf16 >= 2 and ks == 7 and cout <= 64 and cin >= 32:   # the ConvNeXt convolution of the LRNN (64 -> 64)
        packed = torch.empty(L.cwfa_conv7x7_split_packed_bytes(cout, cin), dtype=torch.uint8, device=w.device)
        check(L.cwfa_conv7x7_split_pack_f32(_p(w), _p(packed), cout, cin, _stream()), "conv7x7_split_pack")
        return PackedConv(packed, cout, cin, ks, transposed, w._version, w.data_ptr(), split=True)
    n = L.cwfa_conv2d_packed_floats(cout, cin, ks)
    if n <= 0:
        raise ValueError(f"unsupported filter bank {tuple(w.shape)}")
    packed = torch.empty(n, dtype=torch.float32, device=w.device)
    check(L.cwfa_conv2d_pack_f32(_p(w), _p(packed), cout, cin, ks, int(transposed), _stream()), "conv2d_pack")
    return PackedConv(packed, cout, cin, ks, transposed, w._version, w.data_ptr())


def pack_conv_weight_cat(w, c1):
    """1x1 bank [Cout <= 64, c1 + c2, 1, 1] for conv2d(x, pc, cat=x2): the input cat(x, x2) is read from its two tensors.  The
    columns of the first source are padded with zeros to a multiple of 16 (one K chunk of the kernel)."""
    w = _dev(w, "weight").detach()
    cout, cin, ks, kw = w.shape
    if ks != 1 or kw != 1 or cout > 64 or not (0 < c1 < cin):
        raise ValueError(f"pack_conv_weight_cat: a 1x1 bank with <= 64 outputs and 0 < c1 < Cin is needed, got {tuple(w.shape)}, c1 = {c1}")
    start = (c1 + 15) // 16 * 16
    wz = torch.zeros((cout, start + cin - c1, 1, 1), dtype=torch.float32, device=w.device)
    wz[:, :c1] = w[:, :c1]
    wz[:, start:] = w[:, c1:]
    pc = pack_conv_weight(wz)                       # (cout <= 64: always the direct kernel's layout)
    pc.version, pc.src_ptr = w._version, w.data_ptr()
    pc.cat_c1, pc.cat_from = c1, start
    return pc


def conv_writes_stats(pc, act=None, residual=None, act2=None, out_blocked=False):
    """True when conv2d(..., out_stats=) is available for this bank / epilogue: the split-bf16 3x3 kernel with an NCHW output and
    a bias / PReLU epilogue takes the per-channel (sum, sum of squares) of its output from the accumulators."""
    return bool(pc.split and pc.ks == 3 and act in (None, "prelu") and residual is None and act2 is None and not out_blocked)


def conv2d(x, pc, bias=None, act=None, prelu_alpha=None, residual=None, act2=None, in_scale=None, in_shift=None,
           in_add=None, out=None, in_blocked=False, out_blocked=False, cat=None, out_stats=None):
    """y = act2(act(conv(x') + bias) + residual), x' = x*in_scale[c] + in_shift[c] + in_add.  Transposed banks
    (ConvTranspose2d k2 s2) write the pixel-shuffled [B,Co,2H,2W] output.  ``out_stats`` (float64 [2*Cout], see
    conv_writes_stats): the launch adds (sum y, sum y^2) per channel -- the statistics of a BatchNorm behind the convolution."""
    L = _lib.lib()
    x, xbs = planes(x, "x")
    B, Cin, H, W = x.shape
    o = ConvOpts()
    keep = [x]
    if cat is not None:                     # the input is cat(x, cat) read from its two tensors (pack_conv_weight_cat)
        cat, cbs = planes(cat, "cat")
        c1 = getattr(pc, "cat_c1", None)
        if c1 is None or Cin != c1 or tuple(cat.shape) != (B, pc.cin - pc.cat_from, H, W) or in_scale is not None or in_add is not None:
            raise ValueError("conv2d: `cat` needs a bank from pack_conv_weight_cat for exactly these two inputs, and no load-side prologue")
        o.in_cat, o.in_cat_bs, o.in_cat_from, o.in_cat_c1 = cat.data_ptr(), cbs, pc.cat_from, c1
        keep.append(cat)
        Cin = pc.cin
    if Cin != pc.cin:
        raise ValueError(f"conv2d: input has {Cin} channels, filter bank expects {pc.cin}")
    up = bool(pc.transposed)
    oshape = (B, pc.cout // 4, 2 * H, 2 * W) if up else (B, pc.cout, H, W)
    if out is None:
        out = torch.empty(oshape, dtype=torch.float32, device=x.device)
        ybs = oshape[1] * oshape[2] * oshape[3]
    else:
        if tuple(out.shape) != oshape:
            raise ValueError(f"conv2d: out has shape {tuple(out.shape)}, expected {oshape}")
        out_c, ybs = planes(out, "out")
        if out_c.data_ptr() != out.data_ptr():
            raise ValueError("conv2d: `out` must have contiguous planes")
    if bias is not None:
        o.bias = _dev(bias, "bias").data_ptr()
    o.act, o.act2 = _lib.ACT[act], _lib.ACT[act2]
    if prelu_alpha is not None:
        prelu_alpha = _dev(prelu_alpha, "prelu_alpha")
        o.prelu_alpha = prelu_alpha.data_ptr()
        if prelu_alpha.numel() != 1:             # one slope per output channel (1.0 = no activation there): split-bf16 3x3 kernel only
            if prelu_alpha.numel() != pc.cout or not prelu_alpha.is_contiguous() or not (pc.split and pc.ks == 3):
                raise ValueError("conv2d: per-channel PReLU slopes need [Cout] contiguous values and the split-bf16 3x3 kernel")
            o.prelu_per_channel = 1
    elif "prelu" in (act, act2):
        raise ValueError("conv2d: PReLU needs prelu_alpha")
    if residual is not None:
        residual, rbs = planes(residual, "residual")
        if tuple(residual.shape) != oshape:
            raise ValueError(f"conv2d: residual {tuple(residual.shape)} != output {oshape}")
        o.residual, o.res_bs = residual.data_ptr(), rbs
        keep.append(residual)
    if in_scale is not None:
        o.in_scale, o.in_shift = _dev(in_scale).data_ptr(), _dev(in_shift).data_ptr()
        if in_scale.numel() not in (Cin, B * Cin) or in_shift.numel() != in_scale.numel():
            raise ValueError("conv2d: in_scale / in_shift must be [Cin] or [B,Cin]")
        o.in_affine_bs = Cin if (in_scale.numel() == B * Cin and B > 1) else 0
    if in_add is not None:
        in_add, abs_ = planes(in_add, "in_add")
        if tuple(in_add.shape) != tuple(x.shape):
            raise ValueError("conv2d: in_add shape mismatch")
        o.in_add, o.in_add_bs = in_add.data_ptr(), abs_
        keep.append(in_add)
    o.upshuffle2 = int(up)
    if in_blocked:                          # x is a channel-blocked map of the fused layer kernel (subnet_layer, layout bit 1)
        if not (pc.split and pc.ks == 3):
            raise ValueError("conv2d: channel-blocked input is read by the split-bf16 3x3 kernel only")
        o.in_blocked8 = 1
    if out_blocked:                         # y leaves channel-blocked: the split-bf16 3x3 kernel (bias / PReLU epilogue), or plain 1x1
        ok3 = (pc.split and pc.ks == 3 and pc.cout % 8 == 0 and residual is None and act2 is None and act in (None, "prelu")
               and not split3x3_narrow(pc.cout))
        ok1 = not pc.split and pc.ks == 1 and pc.cout >= 33 and pc.cout % 8 == 0 and not up
        if not (ok3 or ok1):                # banks with 33..64 outputs on the fp32 MFMA kernel
            raise ValueError("conv2d: channel-blocked output is written by the split-bf16 3x3 kernel (bias / PReLU) and by the "
                             "direct 1x1 kernel with 40..64 output channels only")
        o.out_blocked8 = 1
    if out_stats is not None:
        if not conv_writes_stats(pc, act, residual, act2, out_blocked):
            raise ValueError("conv2d: out_stats needs the split-bf16 3x3 kernel with an NCHW output and a bias / PReLU epilogue")
        if out_stats.dtype != torch.float64 or out_stats.numel() != 2 * pc.cout or not out_stats.is_cuda:
            raise ValueError("conv2d: out_stats must be a float64 [2*Cout] tensor on the HIP device")
        o.out_stats = out_stats.data_ptr()
    rec = conv_event_sink
    if rec is not None:                    # bench.py: HIP events around selected launches, on the launch stream
        # the last field names the kernel instantiation the C side dispatches to (prologue / epilogue variant)
        key = (pc.ks, Cin, pc.cout, H, W, B, "%s|%s|%s|%s|%s" % ("pro" if (in_scale is not None or in_add is not None) else "",
                                                                 act or "", "res" if residual is not None else "", act2 or "",
                                                                 ("up" if up else "") + ("+split" if pc.split else "")),
               in_add is not None)
        if rec.want(key):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
    if pc.split and pc.ks == 7:
        # the same kernel with a 3-pixel halo and 49 taps (bias-only epilogue, no prologue: what the ConvNeXt block needs)
        if act or act2 or residual is not None or in_scale is not None or in_add is not None:
            raise ValueError("conv2d: the split-bf16 7x7 kernel has a bias-only epilogue and no load-side prologue")
        check(L.cwfa_conv7x7_split_f32(_p(x), _p(pc.packed), _p(out), B, Cin, H, W, pc.cout, xbs, ybs, C.byref(o), _stream()),
              "conv7x7_split")
    elif pc.split and pc.ks == 3:
        if split3x3_narrow(pc.cout) and (in_scale is not None or in_add is not None):
            # the narrow tilings take no load-side prologue: one streaming pass materialises it (no such layer in CWFA's graphs)
            if (H * W) % 4 == 0:
                x = plane_affine(x, in_scale, in_shift, add=in_add)
            else:                               # odd plane sizes: per-channel affine pass(es) + an add pass
                if in_scale is not None:
                    if in_scale.numel() == Cin:
                        x = channel_affine(x, in_scale.reshape(-1), in_shift.reshape(-1))
                    else:
                        x = torch.cat([channel_affine(x[i:i + 1], in_scale.reshape(B, Cin)[i].contiguous(), in_shift.reshape(B, Cin)[i].contiguous())
                                       for i in range(B)], 0)
                if in_add is not None:
                    x = axpby(x.contiguous(), 1.0, in_add, 1.0)
            x, xbs = planes(x, "x")
            o.in_scale = o.in_shift = o.in_add = None
        # fp32-accurate conv on the bf16 pipe, the kernel splits x on the way into LDS (prologue included)
        check(L.cwfa_conv3x3_split_f32(_p(x), _p(pc.packed), _p(out), B, Cin, H, W, pc.cout, xbs, ybs, C.byref(o),
                                       _stream()), "conv3x3_split")
    elif pc.split:
        # fp32-accurate GEMM on the bf16 pipe: one pass splits x (with the load-side prologue) into three bf16 planes
        ws = torch.empty(L.cwfa_split_workspace_bytes(B, Cin, H * W), dtype=torch.uint8, device=x.device)
        per_sample = in_scale is not None and in_scale.numel() == B * Cin and B > 1
        check(L.cwfa_split_input_f32(_p(x), _p(ws), B, Cin, H * W, xbs, _p(in_scale), _p(in_shift), Cin if per_sample else 0,
                                     _p(in_add), in_add.stride(0) if in_add is not None else 0, _stream()), "split_input")
        o.in_scale = o.in_shift = o.in_add = None
        check(L.cwfa_conv_split_f32(_p(ws), _p(pc.packed), _p(out), B, Cin, H, W, pc.cout, pc.ks, ybs, C.byref(o), _stream()),
              "conv_split")
    else:
        check(L.cwfa_conv2d_f32(_p(x), _p(pc.packed), _p(out), B, Cin, H, W, pc.cout, pc.ks, xbs, ybs, C.byref(o),
                                _stream()), "conv2d")
    if rec is not None and rec.want(key):
        e1.record()
        rec.add(key, e0, e1)
    return out


def pack_couple_weight(w, bias):
    """Last convolution of a coupling sub-network, [2n,Cin,3,3] (+ bias [2n]): rows interleaved (cwfa_couple_rows) so that
    s_j and t_j of a pixel land in the same lane of the split-bf16 kernel, whose epilogue then applies the coupling."""
    L = _lib.lib()
    w = _dev(w, "weight").detach()
    n2, cin, ks, kw = w.shape
    if ks != 3 or kw != 3 or n2 % 2 or n2 // 2 > 64:
        raise ValueError(f"pack_couple_weight: a [2n,Cin,3,3] bank with n <= 64 is needed, got {tuple(w.shape)}")
    n = n2 // 2
    total = L.cwfa_couple_rows(n, None)
    rows_c = (C.c_int * total)()
    L.cwfa_couple_rows(n, rows_c)
    rows = torch.tensor(list(rows_c), dtype=torch.int64, device=w.device)
    live = (rows >= 0)
    wr = torch.zeros((total, cin, 3, 3), dtype=torch.float32, device=w.device)
    wr[live] = w[rows[live]]
    br = torch.zeros(total, dtype=torch.float32, device=w.device)
    if bias is not None:
        br[live] = _dev(bias, "bias").detach()[rows[live]]
    packed = torch.empty(L.cwfa_conv3x3_split_packed_bytes(total, cin), dtype=torch.uint8, device=w.device)
    check(L.cwfa_conv3x3_split_pack_f32(_p(wr), _p(packed), total, cin, _stream()), "conv3x3_split_pack (coupling bank)")
    pc = PackedConv(packed, n2, cin, 3, False, w._version, w.data_ptr(), split=True)
    pc.version1, pc.src_ptr1 = (bias._version, bias.data_ptr()) if bias is not None else (None, None)
    return pc, br


def conv3x3_couple(u, pc_bias, x, out, clamp_kind, clamp, pre_scale, rev, logdet=None, in_blocked=False):
    """out = coupling(x | s, t) with [s_raw | t] = conv3x3(u) + bias taken from the accumulators of the convolution (they never
    reach memory).  ``x`` / ``out``: the active half [B,n,H,W] (channel-slice views allowed; ``out`` may be ``x``)."""
    L = _lib.lib()
    pc, br = pc_bias
    u, ubs = planes(u, "u")
    B, Cin, H, W = u.shape
    n = pc.cout // 2
    if Cin != pc.cin:
        raise ValueError(f"conv3x3_couple: input has {Cin} channels, filter bank expects {pc.cin}")
    x2, xbs = planes(x, "x")
    o2, obs = planes(out, "out")
    if tuple(x.shape) != (B, n, H, W) or tuple(out.shape) != (B, n, H, W) or o2.data_ptr() != out.data_ptr():
        raise ValueError("conv3x3_couple: x / out must be [B,n,H,W] views with contiguous planes")
    cp = _lib.Couple()
    cp.x, cp.y, cp.x_bs, cp.y_bs, cp.n = x2.data_ptr(), out.data_ptr(), xbs, obs, n
    cp.clamp_kind, cp.clamp, cp.pre_scale, cp.rev = _lib.CLAMP[clamp_kind], float(clamp), float(pre_scale), int(bool(rev))
    if logdet is not None:
        cp.logdet = logdet.data_ptr()
    cp.in_blocked8 = int(bool(in_blocked))
    rec = conv_event_sink
    if rec is not None:
        key = (3, Cin, pc.cout, H, W, B, "||||couple+split", False)
        if rec.want(key):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
    check(L.cwfa_conv3x3_split_couple_f32(_p(u), _p(pc.packed), _p(br), B, Cin, H, W, ubs, C.byref(cp), _stream()),
          "conv3x3_split_couple")
    if rec is not None and rec.want(key):
        e1.record()
        rec.add(key, e0, e1)
    return out


BLOCKED_MAPS = True          # (tuning / ablation) False: the maps between the split-bf16 sub-network layers stay NCHW
SPLIT_7X7 = True             # (tuning / ablation) False: 7x7 convolutions stay on the fp32 MFMA kernel in split / bf16 precision
VIRTUAL_CAT = True           # (tuning / ablation) False: the input cat(half, condition) of a coupling sub-network is materialised
COUPLE_EPILOGUE = True       # (tuning / ablation) False: sub-networks write [s_raw | t] and a separate affine launch applies them


def couple_fused():
    """True when the coupling epilogue is available in the active precision mode (split / bf16: the split-bf16 3x3 kernel)."""
    return COUPLE_EPILOGUE and _split_bf16 >= 2


def split3x3_narrow(cout):
    """Banks the split-bf16 3x3 kernel runs on its narrow tilings (16 / 32 / 48 / 96 channels per block on the 16-row tile: no zero
    rows, more m-tiles per B fragment; csrc/conv_split3x3.hip: mpw_of / wm_of): no load-side prologue, NCHW output."""
    return cout <= 48 or 64 < cout <= 96


SPLIT_3X3_MIN_COUT = 1      # 3x3 banks with at least this many outputs take the split-bf16 kernel when "split_bf16" >= 2 (banks with <= 32 /
#                             <= 16 outputs: its narrow tilings of 32 / 16 channels per block; 33 = round 2's rule, those banks on fp32 Winograd)
SPLIT_3X3_NARROW_MAX = 16   # banks with <= this many outputs (and >= 29 inputs) take the 16-channel tiling of the split-bf16 kernel; 17..32
#                             outputs stay on the fp32 Winograd kernel.  Measured @512^2 (tools/small_conv_time.py, fp32 Winograd vs split): 64 -> 24:
#                             71 vs 76 us, 64 -> 12: 66 vs 58, 64 -> 6: 62 vs 56, 29 -> 12: 38 vs 32, 29 -> 6: 35 vs 31, 24 -> 24: 34 vs 39, 6 -> 6: 15 vs 29
#                             -- with so few MFMAs per step the per-step costs of the kernel (barrier, weight DMA, operand reads) decide
conv_event_sink = None      # object with want(key)->bool and add(key, start_event, end_event); set by bench.py only


def pack_1x1_panel(w):
    """[64,64,1,1] filter -> the lane-ordered A-operand image the fused sub-network layer consumes."""
    L = _lib.lib()
    w = _dev(w, "weight").detach().contiguous()
    if tuple(w.shape) != (64, 64, 1, 1):
        raise ValueError("pack_1x1_panel: the fused layer kernel is specialised for 64 channels")
    panel = torch.empty(4096, dtype=torch.float32, device=w.device)
    check(L.cwfa_subnet_pack1x1_f32(_p(w), _p(panel), _stream()), "subnet_pack1x1")
    return PackedConv(panel, 64, 64, 1, False, w._version, w.data_ptr())


def pack_split_layer_weight(w3, w1):
    """[64,64,3,3] and [64,64,1,1] filters -> the split-bf16 image of the fused layer (40 slices of three bf16 pieces)."""
    L = _lib.lib()
    w3 = _dev(w3, "weight").detach().contiguous()
    w1 = _dev(w1, "weight").detach().contiguous()
    if tuple(w3.shape) != (64, 64, 3, 3) or tuple(w1.shape) != (64, 64, 1, 1):
        raise ValueError("pack_split_layer_weight: 64x64x3x3 and 64x64x1x1 only")
    packed = torch.empty(L.cwfa_subnet_layer_split_packed_bytes(), dtype=torch.uint8, device=w3.device)
    check(L.cwfa_subnet_layer_split_pack_f32(_p(w3), _p(w1), _p(packed), _stream()), "subnet_layer_split_pack")
    pc = PackedConv(packed, 64, 64, 3, False, w3._version, w3.data_ptr(), split=True)
    pc.version1, pc.src_ptr1 = w1._version, w1.data_ptr()
    return pc


FIRST_LAYER_COMPOSED = True   # (tuning / ablation) False: the first layer of a sub-network runs like the other two (K = 9 x 64)
FIRST_LAYER_FUSED_X = True      # ... and its residual conv1x1(u) + b0 formed inside that launch (no first 1x1 launch, no first map in memory)


def pack_first_layer_weight(w0, b0, w3, w1, short=None):
    """The composed bank of a sub-network's FIRST layer (cwfa_subnet_layer_first_f32): conv3x3(conv1x1(u, w0) + b0, w3) =
    conv3x3(u | 1, w3c) with w3c[o][i][tap] = sum_m w3[o][m][tap] [w0 | b0][m][i] (formed in float64, rounded once), zero-padded to
    32 input channels; w0 [64,cin,1,1] with cin <= 31, b0 [64] or None, w3 [64,64,3,3], w1 [64,64,1,1]."""
    L = _lib.lib()
    w0, w3, w1 = _dev(w0, "w0").detach(), _dev(w3, "w3").detach(), _dev(w1, "w1").detach().contiguous()
    cin = w0.shape[1]
    if tuple(w0.shape) != (64, cin, 1, 1) or cin > 31 or tuple(w3.shape) != (64, 64, 3, 3) or tuple(w1.shape) != (64, 64, 1, 1):
        raise ValueError("pack_first_layer_weight: 64-channel layers with a 1x1 bank of <= 31 inputs in front only")
    w0p = torch.cat([w0.reshape(64, cin).double(), (b0.detach().double() if b0 is not None else torch.zeros(64, dtype=torch.float64, device=w0.device)).reshape(64, 1)], 1)
    wc = torch.einsum("omt,mi->oit", w3.double().reshape(64, 64, 9), w0p)
    w3c = torch.zeros((64, 32, 3, 3), dtype=torch.float32, device=w0.device)
    w3c[:, :cin + 1] = wc.reshape(64, cin + 1, 3, 3).to(torch.float32)
    w0c = torch.zeros((64, 32), dtype=torch.float32, device=w0.device)          # [W0 | b0 | 0]: the first map as a third k step of the 1x1 phase
    w0c[:, :cin + 1] = w0p.to(torch.float32)
    packed = torch.empty(L.cwfa_subnet_layer_first_packed_bytes(), dtype=torch.uint8, device=w0.device)
    # short form (u has one 16-channel chunk: five conv steps instead of nine) -- only launched with x = None (subnet_layer_first)
    short = (FIRST_LAYER_FUSED_X and cin + 1 <= 16) if short is None else bool(short)
    if short and cin + 1 > 16:
        raise ValueError("pack_first_layer_weight: the short form needs cin + 1 <= 16")
    check(L.cwfa_subnet_layer_first_pack_f32(_p(w3c), _p(w1), _p(w0c), int(short), _p(packed), _stream()), "subnet_layer_first_pack")
    pc = PackedConv(packed, 64, cin + 1, 3, False, w3._version, w3.data_ptr(), split=True)
    pc.short = short
    pc.version1, pc.src_ptr1 = w1._version, w1.data_ptr()
    return pc


def subnet_layer_first(u1, x, pc, b3, b1, layout=0):
    """y = ELU(conv1x1(ELU(conv3x3'(u1) + b3)) + b1 + x): the first layer of a sub-network with its 3x3 composed with the 1x1 in
    front (pack_first_layer_weight).  ``u1``: the sub-network's input plus a constant-one channel [B, cin + 1 <= 32, H, W];
    ``x`` = conv1x1(u) + b0, the residual ([B,64,H,W]; ``layout`` bits as in subnet_layer), or None: the kernel then forms it itself
    as a third k step of its 1x1 phase (the first 1x1 launch of the sub-network and its map are never needed)."""
    L = _lib.lib()
    u1, ubs = planes(u1, "u1")
    B, Cu, H, W = u1.shape
    xbs = 0
    if x is not None:                       # (None: the kernel forms the first map itself from u1 and the packed [W0 | b0]: FIRST_LAYER_FUSED_X)
        x, xbs = planes(x, "x")
    if (x is not None and tuple(x.shape) != (B, 64, H, W)) or Cu != pc.cin:
        raise ValueError(f"subnet_layer_first: u1 {tuple(u1.shape)} / x {None if x is None else tuple(x.shape)} do not match the composed "
                         f"bank ({pc.cin} inputs)")
    short = bool(getattr(pc, "short", False))
    if short and x is not None:
        raise ValueError("subnet_layer_first: a short-form image (pack_first_layer_weight(short=True)) forms its first map itself: x must be None")
    out = torch.empty((B, 64, H, W), dtype=torch.float32, device=u1.device)
    rec = conv_event_sink
    if rec is not None:
        key = ("L", 16 if short else 32, 64, H, W, B, "layer1+split" if x is not None else "layer1x+split", False)
        if rec.want(key):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
    check(L.cwfa_subnet_layer_first_f32(_p(u1), _p(x), _p(pc.packed), _p(_dev(b3)), _p(_dev(b1)), _p(out), B, Cu, H, W, ubs, xbs,
                                        64 * H * W, int(layout) | (4 if short else 0), _stream()), "subnet_layer_first")
    if rec is not None and rec.want(key):
        e1.record()
        rec.add(key, e0, e1)
    return out


_ones_scope = None
_first_maps = None          # {id(sub-network): its first map x = conv1x1(u) + b0, channel-blocked} while a plan has them merged


class first_map_scope:
    """The first 1x1 convolutions of several sub-networks that read the SAME tensor (the five blocks of a CAT step all read the
    condition: coupling_layers.py:475-500, networks.py:621-623) as ONE launch with their banks stacked: the five 64-channel maps
    leave as consecutive channel-blocked chunks of one [B, 64 n, H, W] tensor."""

    def __init__(self, maps):
        self.maps = maps

    def __enter__(self):
        global _first_maps
        self.prev, _first_maps = _first_maps, self.maps
        return self

    def __exit__(self, *exc):
        global _first_maps
        _first_maps = self.prev
        return False


def first_map_of(net):
    return None if _first_maps is None else _first_maps.get(id(net))


class ones_channel_scope:
    """Within this scope ``with_ones(t)`` is computed once per tensor OBJECT (the five sub-networks of a CAT step read the same
    condition): cat(t, 1) through the strided plane-copy kernel."""

    def __enter__(self):
        global _ones_scope
        self.prev, _ones_scope = _ones_scope, {}
        return self

    def __exit__(self, *exc):
        global _ones_scope
        _ones_scope = self.prev
        return False


_ones_planes = {}


def ones_plane(B, H, W, device):
    """A constant [B,1,H,W] tensor of ones (read-only: the ones channel of the composed first layers), built once per shape."""
    key = (B, H, W, str(device))
    hit = _ones_planes.get(key)
    if hit is None:
        if len(_ones_planes) > 8:
            _ones_planes.clear()
        hit = _ones_planes[key] = torch.ones((B, 1, H, W), dtype=torch.float32, device=device)
    return hit


def with_ones(t):
    hit = _ones_scope.get(id(t)) if _ones_scope is not None else None
    if hit is not None and hit[0] is t:
        return hit[1]
    B, _, H, W = t.shape
    u1 = concat_channels([t, ones_plane(B, H, W, t.device)])
    if _ones_scope is not None:
        _ones_scope[id(t)] = (t, u1)
    return u1


def subnet_layer(x, pc3, b3, panel1, b1, want_hidden=False, layout=0):
    """y = ELU(conv1x1(ELU(conv3x3(x) + b3)) + b1 + x), 64 channels, one launch.  ``want_hidden`` (training forward):
    returns (y, h) with h = ELU(conv3x3(x) + b3), written by the same launch.  With ``pc3`` from
    pack_split_layer_weight (both banks in one split-bf16 image) ``panel1`` is unused; ``layout`` (split image only): bit 0 /
    bit 1 = input / output tensor is channel-blocked [B][8][H][W][8] (cwfa_subnet_layer_split_f32) -- the tensor object keeps
    the shape [B,64,H,W], only its memory order differs; the consumer must be told (layout bit 0, or ``in_blocked``)."""
    L = _lib.lib()
    x, xbs = planes(x, "x")
    B, Cc, H, W = x.shape
    if Cc != 64 or pc3.cin != 64 or pc3.cout != 64 or pc3.ks != 3:
        raise ValueError("subnet_layer: 64-channel 3x3 layers only")
    out = torch.empty((B, 64, H, W), dtype=torch.float32, device=x.device)
    if want_hidden:
        hid = torch.empty((B, 64, H, W), dtype=torch.float32, device=x.device)
        if pc3.split:
            if layout:
                raise ValueError("subnet_layer: the tape form keeps NCHW maps")
            check(L.cwfa_subnet_layer_split_tape_f32(_p(x), _p(pc3.packed), _p(_dev(b3)), _p(_dev(b1)), _p(out), _p(hid), B, H, W,
                                                     xbs, 64 * H * W, 64 * H * W, _stream()), "subnet_layer_split_tape")
            return out, hid
        check(L.cwfa_subnet_layer_tape_f32(_p(x), _p(pc3.packed), _p(_dev(b3)), _p(panel1.packed), _p(_dev(b1)), _p(out), _p(hid), B, H, W,
                                           xbs, 64 * H * W, 64 * H * W, _stream()), "subnet_layer_tape")
        return out, hid
    rec = conv_event_sink
    if rec is not None:
        key = ("L", 64, 64, H, W, B, "layer+split" if pc3.split else "layer", False)
        if rec.want(key):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
    if layout and not pc3.split:
        raise ValueError("subnet_layer: channel-blocked maps exist on the split-bf16 layer kernel only")
    if pc3.split:
        check(L.cwfa_subnet_layer_split_f32(_p(x), _p(pc3.packed), _p(_dev(b3)), _p(_dev(b1)), _p(out), B, H, W,
                                            xbs, 64 * H * W, int(layout), _stream()), "subnet_layer_split")
    else:
        check(L.cwfa_subnet_layer_f32(_p(x), _p(pc3.packed), _p(_dev(b3)), _p(panel1.packed), _p(_dev(b1)), _p(out), B, H, W,
                                      xbs, 64 * H * W, _stream()), "subnet_layer")
    if rec is not None and rec.want(key):
        e1.record()
        rec.add(key, e0, e1)
    return out


# ------------------------------------------------------------------------------------------------ backward of the sub-networks
_wgrad_ws = {}


def conv2d_wgrad(x, dy, ks, out=None, accumulate=False, bias_out=None, want_bias=False):
    """dW [Cout,Cin,ks,ks] of a stride-1 'same' convolution y = conv(x, W): sum over batch and pixels of dy (x) shifted x.
    With ``out`` and ``accumulate`` the result is added to ``out`` (a .grad buffer).  ``want_bias`` / ``bias_out``: also the
    bias gradient sum(dy) [Cout] from the same pass (returned as second value; accumulated like ``out``)."""
    L = _lib.lib()
    x, xbs = planes(x, "x")
    dy, dbs = planes(dy, "dy")
    B, Cin, H, W = x.shape
    Cout = dy.shape[1]
    if tuple(dy.shape) != (B, Cout, H, W):
        raise ValueError(f"conv2d_wgrad: dy {tuple(dy.shape)} does not match x {tuple(x.shape)}")
    if out is None:
        out = torch.empty((Cout, Cin, ks, ks), dtype=torch.float32, device=x.device)
        accumulate = False
    elif tuple(out.shape) != (Cout, Cin, ks, ks) or not out.is_contiguous():
        raise ValueError("conv2d_wgrad: `out` must be a contiguous [Cout,Cin,ks,ks] tensor")
    if bias_out is None and want_bias:
        if accumulate:
            raise ValueError("conv2d_wgrad: accumulate needs an existing bias_out")
        bias_out = torch.empty(Cout, dtype=torch.float32, device=x.device)
    if bias_out is not None and (tuple(bias_out.shape) != (Cout,) or not bias_out.is_contiguous()):
        raise ValueError("conv2d_wgrad: `bias_out` must be a contiguous [Cout] tensor")
    nbytes = L.cwfa_conv2d_wgrad_workspace_bytes(B, Cin, H, W, Cout, ks)
    key = (x.device, torch.cuda.current_stream().cuda_stream)
    ws = _wgrad_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _wgrad_ws[key] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    check(L.cwfa_conv2d_wgrad_f32(_p(x), _p(dy), _p(out), _p(bias_out), _p(ws), B, Cin, H, W, Cout, ks, xbs, dbs,
                                  1.0 if accumulate else 0.0, _stream()), "conv2d_wgrad")
    return (out, bias_out) if bias_out is not None else out


def elu_bwd(g, a, add=None, out=None):
    """g * ELU'(q) from the activation output a = ELU(q) (+ add); may run in place on g."""
    L = _lib.lib()
    g, gbs = planes(g, "g")
    a, abs_ = planes(a, "a")
    B = g.shape[0]
    n = g[0].numel()
    if tuple(a.shape) != tuple(g.shape) or (add is not None and tuple(add.shape) != tuple(g.shape)):
        raise ValueError("elu_bwd: g, a (and add) differ in shape")
    addbs = 0
    if add is not None:
        add, addbs = planes(add, "add")
    if out is None:
        out = torch.empty(tuple(g.shape), dtype=torch.float32, device=g.device)
    o2, obs = planes(out, "out")
    if o2.data_ptr() != out.data_ptr():
        raise ValueError("elu_bwd: `out` must have contiguous planes")
    check(L.cwfa_elu_bwd_f32(_p(g), _p(a), _p(add), _p(out), B, n, gbs, abs_, addbs, obs, _stream()), "elu_bwd")
    return out


def plane_affine(x, scale=None, shift=None, add=None):
    """x * scale + shift (+ add) with [C] or [B,C] tables: a BatchNorm (x dropout mask) output as a tensor."""
    L = _lib.lib()
    x, xbs = planes(x, "x")
    B, Cc, H, W = x.shape
    per = 0
    if scale is not None:
        scale, shift = _dev(scale).contiguous(), _dev(shift).contiguous()
        if scale.numel() not in (Cc, B * Cc) or shift.numel() != scale.numel():
            raise ValueError("plane_affine: scale / shift must be [C] or [B,C]")
        per = int(scale.numel() == B * Cc and B > 1)
    abs_ = 0
    if add is not None:
        add, abs_ = planes(add, "add")
        if tuple(add.shape) != (B, Cc, H, W):       # (a centre-cropped skip of another size: the reference's `up + crop` raises too)
            raise ValueError(f"plane_affine: add {tuple(add.shape)} does not match x {(B, Cc, H, W)}")
    out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
    check(L.cwfa_plane_affine_f32(_p(x), _p(scale), _p(shift), per, _p(add), _p(out), B, Cc, H * W, xbs, abs_, Cc * H * W, _stream()),
          "plane_affine")
    return out


def bn_bwd_stats(g, y, mask_bc=None):
    """float64 [C,2]: (sum g*m, sum g*m*y) over (B,H,W)."""
    L = _lib.lib()
    g, gbs = planes(g, "g")
    y, ybs = planes(y, "y")
    B, Cc, H, W = g.shape
    if tuple(y.shape) != (B, Cc, H, W) or (mask_bc is not None and mask_bc.numel() != B * Cc):
        raise ValueError(f"bn_bwd_stats: g {tuple(g.shape)}, y {tuple(y.shape)} / mask do not match")
    st = torch.zeros(2 * Cc, dtype=torch.float64, device=g.device)
    check(L.cwfa_bn_bwd_stats_f32(_p(g), _p(y), _p(None if mask_bc is None else _dev(mask_bc).contiguous()), _p(st), B, Cc, H * W, gbs,
                                  ybs, _stream()), "bn_bwd_stats")
    return st.view(Cc, 2)


def bn_act_bwd(g, y, A, Bc, Cc_, alpha=None, dalpha=None, out=None):
    """(A*g + Bc + Cc*y) * PReLU'(y) (alpha None: identity activation); A is [C] or [B,C]."""
    L = _lib.lib()
    g, gbs = planes(g, "g")
    y, ybs = planes(y, "y")
    B, Cc, H, W = g.shape
    A = _dev(A).contiguous()
    if tuple(y.shape) != (B, Cc, H, W) or A.numel() not in (Cc, B * Cc) or Bc.numel() != Cc or Cc_.numel() != Cc:
        raise ValueError(f"bn_act_bwd: g {tuple(g.shape)}, y {tuple(y.shape)} or the coefficient tables do not match")
    per = int(A.numel() == B * Cc and B > 1)
    if out is None:
        out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=g.device)
    o2, obs = planes(out, "out")
    if o2.data_ptr() != out.data_ptr():
        raise ValueError("bn_act_bwd: `out` must have contiguous planes")
    check(L.cwfa_bn_act_bwd_f32(_p(g), _p(y), _p(A), per, _p(_dev(Bc).contiguous()), _p(_dev(Cc_).contiguous()),
                                _p(None if alpha is None else _dev(alpha)), _p(out), _p(dalpha), B, Cc, H * W, gbs, ybs, obs, _stream()),
          "bn_act_bwd")
    return out


def maxpool2_bwd(full, g_pool, g_skip=None):
    L = _lib.lib()
    full = _dev(full, "full").contiguous()
    g_pool = _dev(g_pool, "g_pool").contiguous()
    B, Cc, H, W = full.shape
    if tuple(g_pool.shape) != (B, Cc, H // 2, W // 2):
        raise ValueError(f"maxpool2_bwd: g_pool {tuple(g_pool.shape)} is not the 2x2-pooled shape of {tuple(full.shape)}")
    if g_skip is not None:
        g_skip = _dev(g_skip, "g_skip").contiguous()
        if tuple(g_skip.shape) != tuple(full.shape):
            raise ValueError("maxpool2_bwd: g_skip and full differ in shape")
    out = torch.empty_like(full)
    check(L.cwfa_maxpool2_bwd_f32(_p(full), _p(g_pool), _p(g_skip), _p(out), B, Cc, H, W, _stream()), "maxpool2_bwd")
    return out


def gelu_add(p, res=None):
    """GELU(p) + res (the unfused form of the ConvNeXt tail, kept for training: the backward needs p)."""
    L = _lib.lib()
    p = _dev(p, "p").contiguous()
    res = None if res is None else _dev(res).contiguous()
    if res is not None and tuple(res.shape) != tuple(p.shape):
        raise ValueError("gelu_add: p and res differ in shape")
    out = torch.empty_like(p)
    check(L.cwfa_gelu_f32(_p(p), _p(res), _p(out), p.numel(), 0, _stream()), "gelu")
    return out


def gelu_bwd(g, p):
    L = _lib.lib()
    p, g = _dev(p, "p").contiguous(), _dev(g, "g").contiguous()
    if tuple(g.shape) != tuple(p.shape):
        raise ValueError("gelu_bwd: g and p differ in shape")
    out = torch.empty_like(p)
    check(L.cwfa_gelu_f32(_p(p), _p(g), _p(out), p.numel(), 1, _stream()), "gelu_bwd")
    return out


def layernorm_bwd(g, v, weight, mean, invstd, dw, db):
    """dL/dv of a LayerNorm over (C,H,W); ``dw`` / ``db`` (shape of the affine) are accumulated in place."""
    L = _lib.lib()
    g, v = _dev(g, "g").contiguous(), _dev(v, "v").contiguous()
    B, n = v.shape[0], v[0].numel()
    if tuple(g.shape) != tuple(v.shape) or weight.numel() != n or mean.numel() != B or invstd.numel() != B or dw.numel() != n or db.numel() != n:
        raise ValueError("layernorm_bwd: operand shapes do not match")
    st = torch.zeros(2 * B, dtype=torch.float64, device=v.device)
    gv = torch.empty_like(v)
    check(L.cwfa_layernorm_bwd_f32(_p(g), _p(v), _p(_dev(weight).contiguous()), _p(_dev(mean).contiguous()), _p(_dev(invstd).contiguous()),
                                   _p(st), _p(gv), _p(dw), _p(db), B, n, _stream()), "layernorm_bwd")
    return gv


def attention_bwd(mean, w1, b1, w2, b2, m, g):
    """Backward of ``attention_combine(mean, ..., m, x)`` given g = dL/dout: (dL/dm, float64 [w1|b1|w2|b2] gradients); dL/dx = g."""
    L = _lib.lib()
    mean, m, g = _dev(mean).contiguous(), _dev(m).contiguous(), _dev(g).contiguous()
    if tuple(m.shape) != tuple(mean.shape) or tuple(g.shape) != tuple(mean.shape):
        raise ValueError("attention_bwd: mean, m and g differ in shape")
    B, Cc = mean.shape[:2]
    HW = mean[0, 0].numel()
    gm = torch.empty_like(m)
    pg = torch.zeros(Cc * Cc * 3 + Cc + Cc * Cc + Cc, dtype=torch.float64, device=mean.device)
    check(L.cwfa_attention_bwd_f32(_p(mean), _p(_dev(w1).contiguous()), _p(_dev(b1)), _p(_dev(w2).contiguous()), _p(_dev(b2)), _p(m),
                                   _p(g), _p(gm), _p(pg), B, Cc, HW, _stream()), "attention_bwd")
    return gm, pg


def prelu_bwd(g, o, alpha, dalpha=None, out=None):
    """g * PReLU'(q) from the activation output o (single alpha > 0); ``dalpha`` (float64[1]) accumulates sum g*min(q,0)."""
    L = _lib.lib()
    g, gbs = planes(g, "g")
    o, obs = planes(o, "o")
    if tuple(o.shape) != tuple(g.shape):
        raise ValueError("prelu_bwd: g and o differ in shape")
    B, n = g.shape[0], g[0].numel()
    if out is None:
        out = torch.empty(tuple(g.shape), dtype=torch.float32, device=g.device)
    o2, ybs = planes(out, "out")
    if o2.data_ptr() != out.data_ptr():
        raise ValueError("prelu_bwd: `out` must have contiguous planes")
    check(L.cwfa_prelu_bwd_f32(_p(g), _p(o), _p(_dev(alpha)), _p(out), _p(dalpha), B, n, gbs, obs, ybs, _stream()), "prelu_bwd")
    return out


def conv3d_1k1_backward(x, dy, w1, b1, alpha, w2, want_input_grad=True):
    """Backward of ``conv3d_1k1`` (y = W2 * PReLU(W1 * x + b1) + b2): returns (dx or None, dW1, db1, dW2, db2, dalpha[float64 1])."""
    L = _lib.lib()
    x = _dev(x, "x").contiguous()
    dy = _dev(dy, "dy").contiguous()
    B, D, H, W = x.shape
    K = w1.shape[0]
    w1c, w2c = _dev(w1).detach().contiguous(), _dev(w2).detach().contiguous()
    st = _stream()
    q = torch.empty((B, K, D, H, W), dtype=torch.float32, device=x.device)
    m = torch.empty_like(q)
    dalpha = torch.zeros(1, dtype=torch.float64, device=x.device)
    ws = torch.empty(L.cwfa_conv3d_wgrad_workspace_bytes(B, D, H, W), dtype=torch.uint8, device=x.device)
    g2 = torch.empty(32, 32, dtype=torch.float32, device=x.device)
    g1 = torch.empty(32, 32, dtype=torch.float32, device=x.device)
    check(L.cwfa_conv3d_hidden_fwd_f32(_p(x), _p(w1c), _p(_dev(b1)), _p(q), B, D, H, W, K, st), "conv3d_hidden_fwd")
    check(L.cwfa_conv3d_wgrad_f32(_p(q), _p(dy), _p(_dev(alpha)), _p(g2), _p(ws), B, D, H, W, K, -1, 0, 0.0, st), "conv3d_wgrad")
    check(L.cwfa_conv3d_hidden_bwd_f32(_p(dy), _p(w2c), _p(q), _p(_dev(alpha)), _p(m), _p(dalpha), B, D, H, W, K, st), "conv3d_hidden_bwd")
    check(L.cwfa_conv3d_wgrad_f32(_p(m), _p(x), None, _p(g1), _p(ws), B, D, H, W, K, 1, 1, 0.0, st), "conv3d_wgrad")
    dx = None
    if want_input_grad:
        dx = torch.empty_like(x)
        check(L.cwfa_conv3d_input_bwd_f32(_p(m), _p(w1c), _p(dx), B, D, H, W, K, st), "conv3d_input_bwd")
    dW1 = g1[:K, :27].reshape(K, 1, 3, 3, 3).contiguous()
    db1 = g1[:K, 27].contiguous()
    dW2 = g2[:K, :27].reshape(1, K, 3, 3, 3).contiguous()
    db2 = sample_stats(dy.reshape(1, -1, 1, 1))[0:1].to(torch.float32)
    return dx, dW1, db1, dW2, db2, dalpha


SPLIT_CONV3D = True          # (tuning / ablation) False: the fused Conv3d of the condition nets stays on the fp32 MFMA kernel in split / bf16 precision


def conv3d_1k1(x, w1, b1, alpha, w2, b2):
    """Conv3d(1->K) -> PReLU -> Conv3d(K->1) over the (H, W, depth) volume of a [B,D,H,W] tensor (networks.py:221-225,239), one
    launch; in split / bf16 precision on the bf16 matrix cores (cwfa_conv3d_1k1_split_f32), else fp32 MFMA."""
    L = _lib.lib()
    x = _dev(x, "x").contiguous()
    B, D, H, W = x.shape
    K = w1.shape[0]
    out = torch.empty_like(x)
    if SPLIT_CONV3D and _split_bf16 >= 2 and K <= 32 and (D + 4) * H * W * 4 < 2 ** 31:
        check(L.cwfa_conv3d_1k1_split_f32(_p(x), _p(_dev(w1).contiguous()), _p(_dev(b1)), _p(_dev(alpha)), _p(_dev(w2).contiguous()),
                                          _p(_dev(b2)), _p(out), B, D, H, W, K, _stream()), "conv3d_1k1_split")
        return out
    check(L.cwfa_conv3d_1k1_f32(_p(x), _p(_dev(w1).contiguous()), _p(_dev(b1)), _p(_dev(alpha)), _p(_dev(w2).contiguous()),
                                _p(_dev(b2)), _p(out), B, D, H, W, K, _stream()), "conv3d_1k1")
    return out


# ------------------------------------------------------------------------------------------------ LRNN helpers
def channel_stats(x, out=None):
    """double[2*C]: per-channel (sum, sumsq) over (B,H,W).  ``out``: a ZEROED float64[2*C] buffer to add into (bn_finish clears
    it again) instead of a fresh one."""
    L = _lib.lib()
    x, xbs = planes(x, "x")
    B, Cc, H, W = x.shape
    st = out if out is not None else torch.zeros(2 * Cc, dtype=torch.float64, device=x.device)
    check(L.cwfa_channel_stats_f32(_p(x), _p(st), B, Cc, H * W, xbs, _stream()), "channel_stats")
    return st


def bn_running_update(stats, count, momentum, running_mean, running_var, num_batches_tracked=None):
    """nn.BatchNorm2d's train-mode update of its buffers from channel_stats() sums, in place, one launch."""
    L = _lib.lib()
    check(L.cwfa_bn_running_update_f32(_p(stats), float(count), float(momentum), _p(_dev(running_mean)), _p(_dev(running_var)),
                                       _p(num_batches_tracked), running_mean.numel(), _stream()), "bn_running_update")


def bn_fold(C_, weight=None, bias=None, eps=1e-5, stats=None, count=0.0, running_mean=None, running_var=None,
            mask_bc=None):
    """(scale, shift) of a BatchNorm (batch or running statistics), optionally times a per-(sample,channel) mask."""
    L = _lib.lib()
    ref = next(t for t in (weight, stats, running_mean, mask_bc) if t is not None)
    B = 0
    n = C_
    if mask_bc is not None:
        mask_bc = _dev(mask_bc, "mask").contiguous()
        B = mask_bc.numel() // C_
        n = B * C_
    scale = torch.empty(n, dtype=torch.float32, device=ref.device)
    shift = torch.empty(n, dtype=torch.float32, device=ref.device)
    check(L.cwfa_bn_fold_f32(_p(stats), float(count), _p(running_mean), _p(running_var), _p(weight), _p(bias),
                             float(eps), _p(mask_bc), B, _p(scale), _p(shift), C_, _stream()), "bn_fold")
    return scale, shift


def bn_finish(C_, weight=None, bias=None, eps=1e-5, stats=None, count=0.0, running_mean=None, running_var=None, momentum=None,
              num_batches_tracked=None, mask_bc=None, mask_u=None, drop_p=0.0, zero_stats=False):
    """bn_fold + bn_running_update (``momentum`` not None) in ONE launch; the dropout factor may be given as the raw uniform draw
    ``mask_u`` [B,C] with ``drop_p`` (m = (u >= p) / (1 - p), exactly F.dropout2d's keep / (1 - p)); ``zero_stats`` clears the
    statistics buffer for its next accumulation."""
    L = _lib.lib()
    ref = next(t for t in (weight, stats, running_mean, mask_bc, mask_u) if t is not None)
    B, n = 0, C_
    mk = mask_bc if mask_bc is not None else mask_u
    if mk is not None:
        mk = _dev(mk, "mask").contiguous()
        B = mk.numel() // C_
        n = B * C_
    scale = torch.empty(n, dtype=torch.float32, device=ref.device)
    shift = torch.empty(n, dtype=torch.float32, device=ref.device)
    upd = momentum is not None and stats is not None and running_mean is not None
    check(L.cwfa_bn_finish_f32(_p(stats), float(count), _p(running_mean), _p(running_var), _p(num_batches_tracked),
                               float(momentum or 0.0), int(upd), _p(weight), _p(bias), float(eps), _p(mk if mask_bc is not None else None),
                               _p(mk if mask_u is not None else None), float(drop_p), float(1.0 - drop_p), B, _p(scale), _p(shift), C_,
                               int(bool(zero_stats)), _stream()), "bn_finish")
    return scale, shift


def maxpool(x, Ho, Wo, scale=None, shift=None, want_full=False):
    L = _lib.lib()
    x = _dev(x, "x").contiguous()
    B, Cc, H, W = x.shape
    for t in (scale, shift):
        if t is not None and t.numel() != Cc:
            raise ValueError("maxpool: the load-side affine must be a [C] table")
    y = torch.empty((B, Cc, Ho, Wo), dtype=x.dtype, device=x.device)
    full = torch.empty_like(x) if want_full else None
    check(L.cwfa_maxpool_f32(_p(x), _p(y), _p(full), _p(scale), _p(shift), B, Cc, H, W, Ho, Wo, _stream()), "maxpool")
    return (y, full) if want_full else y


def sample_stats(x):
    L = _lib.lib()
    x = _dev(x, "x").contiguous()
    B = x.shape[0]
    st = torch.zeros(2 * B, dtype=torch.float64, device=x.device)
    check(L.cwfa_sample_stats_f32(_p(x), _p(st), B, x[0].numel(), _stream()), "sample_stats")
    return st


def layernorm_apply(x, stats, weight, bias, eps):
    L = _lib.lib()
    x = _dev(x, "x").contiguous()
    if stats.numel() != 2 * x.shape[0] or weight.numel() != x[0].numel() or bias.numel() != x[0].numel():
        raise ValueError("layernorm_apply: statistics / affine do not match x")
    out = torch.empty_like(x)
    check(L.cwfa_layernorm_apply_f32(_p(x), _p(stats), _p(weight), _p(bias), float(eps), _p(out), x.shape[0],
                                     x[0].numel(), _stream()), "layernorm_apply")
    return out


def attention_combine(mean, w1, b1, w2, b2, m=None, x=None):
    L = _lib.lib()
    mean = _dev(mean, "mean").contiguous()
    B, Cc = mean.shape[:2]
    HW = mean[0, 0].numel()
    out = torch.empty_like(mean)
    m = None if m is None else _dev(m).contiguous()
    x = None if x is None else _dev(x).contiguous()
    for t in (m, x):
        if t is not None and tuple(t.shape) != tuple(mean.shape):
            raise ValueError("attention_combine: mean, m and x must share one shape")
    if w1.numel() != Cc * Cc * 3 or b1.numel() != Cc or w2.numel() != Cc * Cc or b2.numel() != Cc:
        raise ValueError("attention_combine: the Conv1d banks do not match the channel count")
    check(L.cwfa_attention_combine_f32(_p(mean), _p(_dev(w1).contiguous()), _p(_dev(b1)), _p(_dev(w2).contiguous()),
                                       _p(_dev(b2)), _p(m), _p(x), _p(out), B, Cc, HW, _stream()), "attention_combine")
    return out


def scale_channels(x, scale_bc):
    L = _lib.lib()
    x = _dev(x, "x").contiguous()
    B, Cc = x.shape[:2]
    if scale_bc.numel() != B * Cc:
        raise ValueError(f"scale_channels: the table must hold B*C = {B * Cc} factors")
    out = torch.empty_like(x)
    check(L.cwfa_scale_channels_f32(_p(x), _p(_dev(scale_bc).contiguous()), _p(out), B, Cc, x[0, 0].numel(), _stream()),
          "scale_channels")
    return out


def axpby(x, a, z=None, b=0.0):
    L = _lib.lib()
    x = _dev(x, "x").contiguous()
    z = None if z is None else _dev(z).contiguous()
    if z is not None and z.numel() != x.numel():
        raise ValueError("axpby: x and z differ in size")
    out = torch.empty_like(x)
    check(L.cwfa_axpby_f32(_p(x), _p(z), float(a), float(b), _p(out), x.numel(), _stream()), "axpby")
    return out


def extract_views(image, coords_yx, subimage_shape, mean=0.0, std=1.0):
    """XLFMDatasetFull.extract_views (XLFMDataset.py:212-242) + (v - mean) / std (CWFA.py:796-797) in one launch.
    image [B,1,Hs,Ws] fp32 on the HIP device; coords_yx: n x 2 integers (row, column) of the lenslet centres."""
    L = _lib.lib()
    image = _dev(image, "image")
    if image.dim() != 4 or image.shape[1] != 1:
        raise ValueError(f"extract_views: image must be [B,1,H,W], got {tuple(image.shape)}")
    if image.dtype != torch.float32:
        raise TypeError("extract_views: fp32 images only")
    image = image if image[0].is_contiguous() else image.contiguous()
    B, _, Hs, Ws = image.shape
    co = torch.as_tensor(coords_yx, dtype=torch.int32).reshape(-1, 2)
    sh, sw = int(subimage_shape[0]), int(subimage_shape[1])
    ys, xs = co[:, 0], co[:, 1]
    if len(co) and (bool(((torch.minimum(ys + sh // 2, torch.tensor(Hs)) - torch.clamp(ys - sh // 2, min=0)) <= 0).any()) or
                    bool(((torch.minimum(xs + sw // 2, torch.tensor(Ws)) - torch.clamp(xs - sw // 2, min=0)) <= 0).any())):
        raise ValueError("extract_views: a lenslet window lies outside the image")    # the reference raises on the empty patch
    cod = co.contiguous().to(image.device)
    out = torch.empty((B, len(co), sh, sw), dtype=torch.float32, device=image.device)
    check(L.cwfa_extract_views_f32(_p(image), _p(cod), _p(out), B, Hs, Ws, len(co), sh, sw, float(mean), float(std),
                                   image.stride(0), _stream()), "extract_views")
    return out


_split_bf16 = 0
_plain_bf16 = False      # set_precision("bf16"): the split kernels with ONE product (BASELINE.json configs[4])
_pack_epoch = 0


def pack_epoch():
    """Generation counter of the packing-relevant options; a cached PackedConv with another epoch must be rebuilt."""
    return _pack_epoch


def invalidate_packs():
    """Drop every cached kernel-layout image (filter banks, packed biases, composed chain tables): they are keyed on
    ``tensor._version`` / ``data_ptr()``, which an in-place write through ``.data`` (``m.bias.data *= 0.1``,
    ``nn.init.*_(m.weight.data)`` -- the reference's own initialisers, networks.py:19-62) does NOT change.  The initialisers and
    reset helpers of cwfa_amd.networks call this; call it yourself after any other ``.data`` edit of a parameter that has
    already been used in a forward pass."""
    global _pack_epoch
    _pack_epoch += 1


WINOGRAD_2D_DEFAULT = 512     # library default of the "winograd_2d" option (output-channel threshold of the 2-D kernel)


def set_option(name, value):
    """Process-wide tuning option (see cwfa_set_option in include/cwfa_hip.h).  Filter banks packed before a change of
    "winograd_min_cout" / "winograd_2d" / "split_bf16" must be re-packed.
    "split_bf16" (host-side switch, default 0): 1x1 convolutions and ConvTranspose2d(k2,s2) with >= 128 output channels
    run as an fp32-accurate GEMM on the bf16 matrix pipe (cwfa_split_input_f32 + cwfa_conv_split_f32)."""
    global _split_bf16, _pack_epoch
    _pack_epoch += 1
    if name == "split_bf16":          # 0 off, 1: 1x1 / transposed convs, 2: also 3x3 convs and the fused 64-channel layers
        _split_bf16 = int(value)
        return
    check(_lib.lib().cwfa_set_option(name.encode(), int(value)), "set_option")


def set_precision(mode):
    """"fp32" (library default: plain fp32 MFMA / Winograd kernels) | "split_bf16" (the benchmark's arithmetic, fp32-equivalent:
    three bf16 pieces per operand, six products, fp32 accumulation) | "bf16" (BASELINE.json configs[4]: the same kernels with ONE
    product, i.e. plain bf16 operands and fp32 accumulation).  The split / bf16 kernels take: 1x1 and transposed convolutions with
    >= 128 outputs, 3x3 convolutions with >= SPLIT_3X3_MIN_COUT outputs, the 64-channel fused layers; wavelets, couplings,
    permutations, the Conv3d of the condition nets and the remaining small convolutions stay fp32."""
    if mode not in ("fp32", "split_bf16", "bf16"):
        raise ValueError(f"set_precision: unknown mode {mode!r}")
    global _plain_bf16
    _plain_bf16 = mode == "bf16"
    set_option("split_products", 1 if mode == "bf16" else 6)
    set_option("split_bf16", 0 if mode == "fp32" else 2)
    set_option("wgrad_split", 0 if mode == "fp32" else 1)        # training: 3x3 weight gradients in the same arithmetic (csrc/conv_bwd.hip)


def copy_channels(src, dst):
    """dst[...] = src for [B,c,H,W] views with contiguous planes (channel slices of NCHW tensors): one strided plane-copy launch."""
    L = _lib.lib()
    s_, sbs = planes(src, "src")
    d_, dbs = planes(dst, "dst")
    if d_.data_ptr() != dst.data_ptr() or tuple(src.shape) != tuple(dst.shape):
        raise ValueError("copy_channels: dst must be a [B,c,H,W] view with contiguous planes of src's shape")
    B, Cc, H, W = s_.shape
    check(L.cwfa_channel_affine_f32(_p(s_), _p(dst), None, None, 0, None, None, B, Cc, H * W, sbs, dbs, _stream()), "copy_channels")
    return dst


def concat_channels(parts):
    """torch.cat(parts, 1) through the strided plane-copy kernel (coupling_layers.py:74-87 materialise this too)."""
    L = _lib.lib()
    parts = [planes(t, "part") for t in parts]
    B, _, H, W = parts[0][0].shape
    Ct = sum(t.shape[1] for t, _ in parts)
    out = torch.empty((B, Ct, H, W), dtype=torch.float32, device=parts[0][0].device)
    c0 = 0
    for t, bs in parts:
        if t.shape[0] != B or t.shape[2:] != (H, W):
            raise ValueError("concat_channels: shape mismatch")
        check(L.cwfa_channel_affine_f32(_p(t), C.c_void_p(out.data_ptr() + 4 * c0 * H * W), None, None, 0, None, None, B,
                                        t.shape[1], H * W, bs, Ct * H * W, _stream()), "concat_channels")
        c0 += t.shape[1]
    return out
