#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (where /root/reference
exists); never on the GPU box.  No reference source is copied: the reference is
imported as a library with harness-side ``sys.modules`` shims for packages that
are absent and off the hot path (recipe: SURVEY.md Appendix A), executed on
seeded synthetic inputs, and only inputs / weights / outputs are written out as
small ``.npz`` fixtures.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Fixture index:
  g01_haar1d        HaarTransform1D fwd / rev                (INN_utils.py:126-174)
  g02_haar2d_*      HaarDownsampling fwd / rev, 4 option sets (reshapes.py:191-318)
  g03_perms         permutation tables + axes per step/block  (networks.py:336-357)
  g04_<block>_<cl>  coupling blocks fwd / rev                 (coupling_layers.py)
  g05_ai1*          AllInOneBlock fwd / rev                   (all_in_one_block.py)
  g06_actnorm       ActNorm data init + fwd / rev             (invertible_resnet.py:11-85)
  g07_subnet*       wavelet_flow_subnetwork2D(_first)         (networks.py:586-706)
  g08_omega         cond_network / ResidualBlock (eval)       (networks.py:165-242)
  g09_step_<bt>_k<k> one full GraphINN step per block type    (networks.py:264-368)
  g10_pipeline      2-step inverse pipeline + evaluate_INN_forward (CWFA.py:134-196,865-924)
  g11_unet_*, g11_convnext, g11_attention, g11_lrnn_small, g11_lrnn_full
  g12_extract_views_*   XLFMDatasetFull.extract_views (needs only torch: imported from the reference file directly)
  g13_step_grad_*       autograd gradients of the training NLL of one CAT step (CWFA.py:966-978,1002-1006)
  g20_blockgrad_*       the same for steps built with the data-dependent block types (GLOW, AI1, RNVP, GIN)
  g14_unet_grad_*       autograd gradients of the UNet in train mode (unet.py:72-113,161-195)
  g15_meanbranch_grad   autograd gradients of the LRNN's mean-volume branch (networks.py:468-503,244-262,552-554)
  g16_concat            Concat fwd / rev, three inputs              (graph_topology.py:92-152)
  g17_fixed1x1conv      Fixed1x1Conv fwd / rev + log-det            (fixed_transforms.py:95-133)
  g18_ai1_opt_*         AllInOneBlock options: soft permutation, learned householder, reverse permutation, SIGMOID / EXP
                        global affine                               (all_in_one_block.py:122-196)
  g19_meanvol / g19_denorm  mean-volume cache detail bands and the output de-normalisation (CWFA.py:637-655,1035-1044)
"""
import os
import sys
import types
import argparse
import warnings

import numpy as np

warnings.filterwarnings("ignore")
REF = "/root/reference"
# CWFA_GOLDEN_OUT: write somewhere else (tests/test_make_golden.py regenerates into a temp dir and compares)
OUT = os.environ.get("CWFA_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    sys.path.insert(0, REF)
    m = types.ModuleType("numpy.lib.arraysetops")
    m.isin = np.isin
    sys.modules["numpy.lib.arraysetops"] = m
    for name in ["tifffile", "multipagetiff", "torchvision", "lion_pytorch",
                 "torch.utils.tensorboard"]:
        mod = types.ModuleType(name)
        mod.imsave = mod.imread = None
        mod.Lion = object
        mod.SummaryWriter = object
        sys.modules[name] = mod
    import torch  # noqa
    import FrEIA.framework as Ff
    import FrEIA.modules as Fm
    import INN_utils
    import networks
    import unet
    import CWFA
    return Ff, Fm, INN_utils, networks, unet, CWFA


def fresh_process_state(networks):
    """What a generator relies on must not depend on which generators ran before it in the same process: autograd on
    (main() switches it off for the forward-only fixtures) and the ONE PReLU instance every ResidualBlock of the process
    shares (a default argument, networks.py:209; G8 sets it to 0.2) back at its initial slope."""
    import torch
    torch.set_grad_enabled(True)
    with torch.no_grad():
        networks.ResidualBlock.__init__.__defaults__[-1].weight.fill_(0.25)


def npy(t):
    return t.detach().cpu().numpy().copy()


def sd_arrays(module, prefix="sd/"):
    return {prefix + k: npy(v) for k, v in module.state_dict().items()}


def dump(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name:32s} {os.path.getsize(path)/1024:8.1f} KiB  ({len(arrs)} arrays)")


def main():
    import torch
    Ff, Fm, INN_utils, networks, unet, CWFA = import_reference()
    torch.set_num_threads(8)
    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(1234)

    def rn(*shape, scale=1.0):
        return torch.randn(*shape, generator=g) * scale

    # ---------------------------------------------------------------- G1
    x = rn(2, 8, 6, 10)
    mod = INN_utils.HaarTransform1D([(8, 6, 10)], order_by_wavelet=True)
    (yf,), jf = mod((x,), rev=False)
    (yr,), jr = mod((x,), rev=True)
    dump("g01_haar1d", x=npy(x), y_fwd=npy(yf), y_rev=npy(yr),
         jac_fwd=np.float64(jf), jac_rev=np.float64(jr))

    # ---------------------------------------------------------------- G2
    for obw in (False, True):
        for reb in (1.0, 0.5):
            x = rn(2, 3, 8, 12)
            mod = Fm.HaarDownsampling([(3, 8, 12)], order_by_wavelet=obw, rebalance=reb)
            (yf,), jf = mod((x.clone(),), rev=False)
            z = rn(2, 12, 4, 6)
            (xr,), jr = mod((z.clone(),), rev=True)
            dump(f"g02_haar2d_obw{int(obw)}_reb{reb}", x=npy(x), y_fwd=npy(yf), z=npy(z),
                 x_rev=npy(xr), jac_fwd=np.float64(jf), jac_rev=np.float64(jr),
                 order_by_wavelet=np.int64(obw), rebalance=np.float64(reb))

    # ---------------------------------------------------------------- G3 / G9
    # Steps built exactly as run_CWFA does (CWFA.py:498-510), tiny channel counts.
    S = 3
    D, H, W = 16, 12, 16
    n_ch, cond_ch = 8, 4
    perms = {}
    for bt in ["CAT", "GLOW", "RNVP", "GIN", "AI1"]:
        for ix in range(S - 1):
            torch.manual_seed(100 + ix)
            np.random.seed(7)
            Cn = D // 2 ** (ix + 1)
            cond_net, inns = networks.conditional_wavelet_flow(
                [D, H, W], [1, 29, H, W], networks.wavelet_flow_subnetwork2D,
                lambda: networks.cond_network(29, Cn, ix + 1, S, [], cond_ch),
                n_internal_ch=n_ch, n_down_steps=ix + 1, use_permutations=True,
                block_type=bt, n_blocks=4)
            inn = inns[ix].eval()
            Dn = D // 2 ** ix
            # perturb biases / the near-zero `_first` last conv so every path is exercised
            for p in inn.parameters():
                if p.requires_grad and p.dtype == torch.float32:
                    p.add_(torch.randn(p.shape, generator=g) * 0.05)
            meta = {}
            for i, mdl in enumerate(inn.module_list):
                cls = type(mdl).__name__
                meta[f"meta/module_{i}"] = np.array(cls)
                if cls == "PermuteDim":
                    meta[f"meta/axis_{i}"] = np.int64(mdl.dims_to_permute[1])
                    perms[f"{bt}/k{ix}/m{i}/axis"] = np.int64(mdl.dims_to_permute[1])
                if hasattr(mdl, "perm") and mdl.perm is not None and cls != "HaarDownsampling":
                    perms[f"{bt}/k{ix}/m{i}/perm"] = npy(mdl.perm)
                if cls == "AllInOneBlock":
                    perms[f"{bt}/k{ix}/m{i}/w_perm"] = npy(mdl.w_perm)[:, :, 0, 0]
            meta["meta/node_names"] = np.array([n.name for n in inn.node_list])
            meta["meta/cond_node_names"] = np.array([n.name for n in inn.condition_nodes])
            meta["meta/out_node_names"] = np.array([n.name for n in inn.out_nodes])
            meta["meta/dims_c"] = np.array(inn.dims_c)
            meta["meta/global_out_shapes"] = np.array(inn.global_out_shapes)
            x = rn(2, Dn, H, W)
            c = [rn(2, Cn, H, W), rn(2, Cn, H, W, scale=0.3)]   # [Omega -> 'Condition I', mean detail -> 'Condition']
            (z, low), jf = inn(x, c=c)
            xr, jr = inn([z, low], c=c, rev=True)
            x0, j0 = inn([torch.zeros_like(z), low], c=c, rev=True)
            dump(f"g09_step_{bt}_k{ix}", x=npy(x), c0=npy(c[0]), c1=npy(c[1]), z=npy(z), low=npy(low),
                 jac_fwd=npy(jf), x_rev=npy(xr), jac_rev=npy(jr), x_rev_z0=npy(x0), jac_rev_z0=npy(j0),
                 D=np.int64(D), H=np.int64(H), W=np.int64(W), ix=np.int64(ix), S=np.int64(S),
                 n_ch=np.int64(n_ch), cond_ch=np.int64(cond_ch), **meta, **sd_arrays(inn))
    dump("g03_perms", **perms)

    # ---------------------------------------------------------------- G4
    networks.networks_n_chans = 8
    C, Hc, Wc, Cc = 6, 8, 10, 5
    blocks = {
        "CAT": Fm.ConditionalAffineTransform, "GLOW": Fm.GLOWCouplingBlock, "RNVP": Fm.RNVPCouplingBlock,
        "GIN": Fm.GINCouplingBlock, "NICE": Fm.NICECouplingBlock, "ONESIDED": Fm.AffineCouplingOneSided,
    }
    for bname, cls in blocks.items():
        clamps = ["ATAN", "TANH", "SIGMOID"] if bname in ("CAT", "GLOW") else ["ATAN"]
        for cl in clamps:
            torch.manual_seed(5)
            kw = {"subnet_constructor": networks.wavelet_flow_subnetwork2D}
            if bname != "NICE":
                kw.update(clamp=1.5, clamp_activation=cl)
            blk = cls([(C, Hc, Wc)], dims_c=[(Cc, Hc, Wc)], **kw).eval()
            for p in blk.parameters():
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
            x = rn(2, C, Hc, Wc)
            c = rn(2, Cc, Hc, Wc)
            (yf,), jf = blk((x,), c=(c,), rev=False)
            (yr,), jr = blk((x,), c=(c,), rev=True)
            jf = jf if torch.is_tensor(jf) else torch.full((2,), float(jf))
            jr = jr if torch.is_tensor(jr) else torch.full((2,), float(jr))
            dump(f"g04_{bname}_{cl}", x=npy(x), c=npy(c), y_fwd=npy(yf), jac_fwd=npy(jf), y_rev=npy(yr),
                 jac_rev=npy(jr), clamp=np.float64(1.5), **sd_arrays(blk))

    # ---------------------------------------------------------------- G5
    for gin in (False, True):
        for cond in (True, False):
            torch.manual_seed(6)
            np.random.seed(11)
            C = 7
            blk = Fm.AllInOneBlock([(C, Hc, Wc)], dims_c=[(Cc, Hc, Wc)] if cond else [],
                                   subnet_constructor=networks.wavelet_flow_subnetwork2D, gin_block=gin).eval()
            for n_, p in blk.named_parameters():
                if p.requires_grad:
                    p.add_(torch.randn(p.shape, generator=g) * 0.2)
            x = rn(2, C, Hc, Wc)
            c = (rn(2, Cc, Hc, Wc),) if cond else ()
            (yf,), jf = blk((x,), c=c, rev=False)
            (yr,), jr = blk((x,), c=c, rev=True)
            dump(f"g05_ai1_gin{int(gin)}_cond{int(cond)}", x=npy(x), c=npy(c[0]) if cond else np.zeros(0, np.float32),
                 y_fwd=npy(yf), jac_fwd=npy(jf), y_rev=npy(yr), jac_rev=npy(jr), **sd_arrays(blk))

    # ---------------------------------------------------------------- G6
    x = rn(3, 5, 6, 8) * 2.5 + 0.7
    an = Fm.ActNorm([(5, 6, 8)])
    (yf,), jf = an((x,), rev=False)        # triggers data-dependent init
    z = rn(3, 5, 6, 8)
    (yr,), jr = an((z,), rev=True)
    dump("g06_actnorm", x=npy(x), y_fwd=npy(yf), jac_fwd=npy(jf), z=npy(z), x_rev=npy(yr), jac_rev=npy(jr),
         **sd_arrays(an))

    # ---------------------------------------------------------------- G7
    for first in (False, True):
        for nch in (8, 64):
            networks.networks_n_chans = nch
            torch.manual_seed(8)
            ci, co = (12, 12) if first else (9, 10)
            ctor = networks.wavelet_flow_subnetwork2D_first if first else networks.wavelet_flow_subnetwork2D
            net = ctor(ci, co).eval()
            for p in net.parameters():
                p.add_(torch.randn(p.shape, generator=g) * 0.03)
            x = rn(2, ci, 9, 11)
            y = net(x)
            dump(f"g07_subnet_first{int(first)}_ch{nch}", x=npy(x), y=npy(y), c_in=np.int64(ci), c_out=np.int64(co),
                 n_ch=np.int64(nch), **sd_arrays(net))
    networks.networks_n_chans = 8

    # ---------------------------------------------------------------- G8
    for cout, chans3d in ((6, 4), (8, 32)):
        torch.manual_seed(9)
        net = networks.cond_network(29, cout, 1, 5, [], chans3d).eval()
        with torch.no_grad():
            net.subnetworks[0].relu.weight.fill_(0.2)      # shared PReLU (networks.py:209)
        x = rn(2, 29, 10, 12)
        y = net(x)[-1]
        dump(f"g08_omega_c{cout}_k{chans3d}", x=npy(x), y=npy(y), c_out=np.int64(cout), chans3d=np.int64(chans3d),
             **sd_arrays(net))

    # ---------------------------------------------------------------- G10
    S, D, H, W = 3, 16, 16, 16
    args = argparse.Namespace(INN_max_down_steps=S, force_all_steps_NF=0, n_depths=D, volume_side_size=H)
    conv_inn, cond_nets = [], []
    torch.manual_seed(21)
    np.random.seed(3)
    for ix in range(S - 1):
        Cn = D // 2 ** (ix + 1)
        cn, inns = networks.conditional_wavelet_flow(
            [D, H, W], [1, 29, H, W], networks.wavelet_flow_subnetwork2D,
            lambda: networks.cond_network(29, Cn, ix + 1, S, [], 4),
            n_internal_ch=8, n_down_steps=ix + 1, use_permutations=True, block_type="CAT", n_blocks=4)
        for p in inns[ix].parameters():
            if p.requires_grad:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
        conv_inn.append(inns[ix].eval())
        cond_nets.append(cn.eval())
    B = 2
    gt = rn(B, D, H, W) + 0.1 * torch.arange(D).view(1, D, 1, 1)   # no empty depths
    views = rn(B, 29, H, W)
    stats = (torch.tensor(0.1), torch.tensor(1.3), None, None, torch.tensor(0.0), torch.tensor(1.0))
    losses, gt_cache, prior, logj = CWFA.evaluate_INN_forward(conv_inn, cond_nets, args, [args] * S, gt.clone(),
                                                              views, stats)
    cond_input = (views - stats[0]) / stats[1]
    mean_cache = [rn(B, D // 2 ** (n + 1), H, W, scale=0.1) for n in range(S - 1)]
    low = gt_cache[S - 1].clone()
    up = low
    ups = {}
    for n in range(S - 2, -1, -1):
        cp = [cond_nets[n](cond_input)[-1].float(), mean_cache[n]]
        z = CWFA.sample_z_truncated((B,) + tuple(conv_inn[n].global_out_shapes[0]), temperature=0)
        up, lj = conv_inn[n]([z, up], c=cp, rev=True)
        ups[f"up_{n}"] = npy(up)
        ups[f"omega_{n}"] = npy(cp[0])
    fx = dict(gt=npy(gt), views=npy(views), mean_imgs=np.float32(0.1), std_imgs=np.float32(1.3), low=npy(low),
              losses=np.array([float(l) for l in losses]), prior=np.array([float(l) for l in prior]),
              logjac=np.array([float(l) for l in logj]), S=np.int64(S), **ups)
    for n in range(S):
        fx[f"gt_cache_{n}"] = npy(gt_cache[n])
    for n in range(S - 1):
        fx[f"mean_cache_{n}"] = npy(mean_cache[n])
        fx.update(sd_arrays(conv_inn[n], f"inn{n}/"))
        fx.update(sd_arrays(cond_nets[n], f"omega{n}/"))
    dump("g10_pipeline", **fx)

    # ---------------------------------------------------------------- G11
    for bias in (False, True):
        torch.manual_seed(31)
        u = unet.UNet(5, 4, depth=3, wf=3, drop_out=0, use_bias=bias, skip_conn=True, up_mode="upconv",
                      batch_norm=True)
        for m_ in u.modules():
            if isinstance(m_, torch.nn.BatchNorm2d):
                m_.running_mean.copy_(torch.randn(m_.running_mean.shape, generator=g) * 0.1)
                m_.running_var.copy_(torch.rand(m_.running_var.shape, generator=g) + 0.5)
                m_.weight.copy_(torch.rand(m_.weight.shape, generator=g) + 0.5)
                m_.bias.copy_(torch.randn(m_.bias.shape, generator=g) * 0.1)
        x = rn(2, 5, 16, 16)
        sd0 = sd_arrays(u)
        u.eval()
        y_eval = u(x)
        u.train()
        y_train = u(x)                       # batch statistics (B=2), running stats get updated
        y_train_b1 = u(x[:1])                # B=1: instance statistics
        dump(f"g11_unet_bias{int(bias)}", x=npy(x), y_eval=npy(y_eval), y_train=npy(y_train), y_train_b1=npy(y_train_b1),
             **sd0)

    torch.manual_seed(32)
    cnx = networks.ConvNeXt(6, 10, drop_prob=0.05, size=16).eval()
    cnx.m[1].weight.add_(torch.randn(cnx.m[1].weight.shape, generator=g) * 0.1)
    cnx.m[1].bias.add_(torch.randn(cnx.m[1].bias.shape, generator=g) * 0.1)
    x = rn(2, 6, 16, 16)
    dump("g11_convnext", x=npy(x), y=npy(cnx(x)), **sd_arrays(cnx))

    torch.manual_seed(33)
    att = networks.GlobalAttention(6).eval()
    x = rn(2, 6, 8, 12)
    dump("g11_attention", x=npy(x), y=npy(att(x)), **sd_arrays(att))

    # LRNN: weights are NOT shipped (63.7 M params).  The fixture pins (a) the construction-time RNG stream through
    # per-tensor checksums of the default-initialised state_dict under torch.manual_seed(41), (b) outputs for seeded
    # inputs that the test regenerates with the same torch CPU generator.
    torch.manual_seed(41)
    enc = networks.Encoder(29, 6, 5, 64, True)
    sums = {}
    for k, v in enc.state_dict().items():
        vf = v.double()
        sums["chk/" + k] = np.array([vf.sum().item(), vf.abs().sum().item(), float(v.numel())])
    gi = torch.Generator().manual_seed(4242)
    x_small = torch.randn(2, 29, 16, 16, generator=gi)
    enc.eval()
    # UNet dropout2d p=0.005 is ALWAYS active in the reference (F.dropout2d default training=True, unet.py:80,86) and
    # draws from torch's CPU RNG -> switched off for the deterministic fixtures
    enc.net.deconv[1].drop_out = 0
    y_small_eval = enc(x_small)[-1]
    dump("g11_lrnn_small", y_eval=npy(y_small_eval), seed_init=np.int64(41), seed_input=np.int64(4242), **sums)
    # full-size (512x512) with the mean-volume branch, eval mode (running BN stats, no drop_path), B=1.
    gi = torch.Generator().manual_seed(4343)
    x_full = torch.randn(1, 29, 512, 512, generator=gi)
    mean_full = torch.randn(1, 6, 512, 512, generator=gi) * 0.1
    # give the LayerNorm affine a non-trivial value (deterministic, regenerated in the test)
    gl = torch.Generator().manual_seed(4444)
    for cn_ in enc.net.conv3d:
        cn_.m[1].weight.copy_(1 + 0.1 * torch.randn(cn_.m[1].weight.shape, generator=gl))
        cn_.m[1].bias.copy_(0.1 * torch.randn(cn_.m[1].bias.shape, generator=gl))
    y_full = enc(x_full, mean_full)[-1]
    y_nomean = enc(x_full)[-1]
    dump("g11_lrnn_full", y_sub=npy(y_full[:, :, ::23, ::29]), y_nomean_sub=npy(y_nomean[:, :, ::23, ::29]),
         y_sum=np.float64(y_full.double().sum()), y_abs=np.float64(y_full.double().abs().sum()),
         seed_init=np.int64(41), seed_input=np.int64(4343), seed_ln=np.int64(4444))


def gen_topology():
    """G16 Concat, G17 Fixed1x1Conv (FrEIA API surface; VERDICT r1: no test anywhere)."""
    import torch
    Ff, Fm, INN_utils, networks, unet, CWFA = import_reference()
    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(1616)
    dims = [(3, 5, 7), (2, 5, 7), (4, 5, 7)]
    xs = [torch.randn(2, *d, generator=g) for d in dims]
    cat = Fm.Concat(dims, dim=0)
    (y,), jf = cat(xs, rev=False)
    z = torch.randn(2, 9, 5, 7, generator=g)
    parts, jr = cat((z,), rev=True)
    dump("g16_concat", x0=npy(xs[0]), x1=npy(xs[1]), x2=npy(xs[2]), y_fwd=npy(y), z=npy(z),
         r0=npy(parts[0]), r1=npy(parts[1]), r2=npy(parts[2]), jac_fwd=np.float64(jf), jac_rev=np.float64(jr),
         out_dims=np.array(cat.output_dims(dims)[0], np.int64))
    M = torch.randn(6, 6, generator=g) + 2.0 * torch.eye(6)
    m = Fm.Fixed1x1Conv([(6, 9, 11)], M=M)
    x = torch.randn(2, 6, 9, 11, generator=g)
    (yf,), jf = m((x,), rev=False)
    (xr,), jr = m((x,), rev=True)
    dump("g17_fixed1x1conv", M=npy(M), x=npy(x), y_fwd=npy(yf), y_rev=npy(xr), jac_fwd=np.float64(float(jf)),
         jac_rev=np.float64(float(jr)), **sd_arrays(m))


def gen_ai1_options():
    """G18: AllInOneBlock options outside CWFA's defaults: soft permutation, learned householder, reverse permutation,
    GIN, SIGMOID / EXP global affine (all_in_one_block.py:122-196)."""
    import torch
    Ff, Fm, INN_utils, networks, unet, CWFA = import_reference()
    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(1818)
    C, H, W, Cc = 6, 8, 10, 3

    class Sub(torch.nn.Module):
        def __init__(self, cin, cout):
            super().__init__()
            self.c = torch.nn.Conv2d(cin, cout, 3, padding=1)

        def forward(self, t):
            return self.c(t)

    cases = {"soft": dict(permute_soft=True), "house": dict(learned_householder_permutation=2),
             "revperm": dict(reverse_permutation=True), "gin": dict(gin_block=True),
             "sigmoid": dict(global_affine_type="SIGMOID", global_affine_init=0.7), "exp": dict(global_affine_type="EXP", global_affine_init=1.3),
             "soft_rev_gin": dict(permute_soft=True, reverse_permutation=True, gin_block=True)}
    for name, kw in cases.items():
        torch.manual_seed(40)
        np.random.seed(41)
        m = Fm.AllInOneBlock([(C, H, W)], dims_c=[(Cc, H, W)], subnet_constructor=Sub, **kw)
        with torch.no_grad():
            m.global_offset.copy_(0.1 * torch.randn(m.global_offset.shape, generator=g))
            m.global_scale.add_(0.3 * torch.randn(m.global_scale.shape, generator=g))
        x = torch.randn(2, C, H, W, generator=g)
        c = torch.randn(2, Cc, H, W, generator=g)
        (yf,), jf = m((x.clone(),), c=[c], rev=False)
        (yr,), jr = m((x.clone(),), c=[c], rev=True)
        arrs = {k2: npy(v) for k2, v in m.state_dict().items()}
        dump(f"g18_ai1_opt_{name}", x=npy(x), c=npy(c), y_fwd=npy(yf), y_rev=npy(yr), jac_fwd=npy(jf), jac_rev=npy(jr),
             **{"sd/" + k2: v for k2, v in arrs.items()})


def gen_cache_and_output():
    """G19: the mean-volume cache (the detail bands of the forward pyramid of a mean volume, CWFA.py:646-655; stored as
    {'mean_vol_gt_cache': [...]} by main.py:366-377) and the output de-normalisation of the evaluation branch
    (CWFA.py:1035-1044, with its 2**len(...) factor).  The pyramid comes from the reference's evaluate_INN_forward; the two
    post-processing expressions are the reference's lines applied to its tensors."""
    import io
    import torch
    Ff, Fm, INN_utils, networks, unet, CWFA = import_reference()
    fresh_process_state(networks)
    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(1919)
    S, D, H, W = 3, 16, 12, 16
    args = argparse.Namespace(INN_max_down_steps=S, force_all_steps_NF=0, n_depths=D, volume_side_size=H)
    conv_inn, cond_nets = [], []
    torch.manual_seed(23)
    np.random.seed(5)
    for ix in range(S - 1):
        Cn = D // 2 ** (ix + 1)
        cn, inns = networks.conditional_wavelet_flow(
            [D, H, W], [1, 29, H, W], networks.wavelet_flow_subnetwork2D,
            lambda: networks.cond_network(29, Cn, ix + 1, S, [], 4),
            n_internal_ch=8, n_down_steps=ix + 1, use_permutations=True, block_type="CAT", n_blocks=4)
        conv_inn.append(inns[ix].eval())
        cond_nets.append(cn.eval())
    conv_inn.append(None)
    mean_vols_stack = torch.randn(1, D, H, W, generator=g) + 0.1 * torch.arange(D).view(1, D, 1, 1)
    mean_vols, std_vols = torch.tensor(0.3), torch.tensor(1.7)
    gt_volume = (mean_vols_stack - mean_vols) / std_vols                                   # CWFA.py:646
    views = torch.rand(1, 29, H, W, generator=g)
    stats = (torch.tensor(0.1), torch.tensor(1.3), mean_vols, std_vols, torch.tensor(0.0), torch.tensor(1.0))
    _, gt_cache, _, _ = CWFA.evaluate_INN_forward(conv_inn[:S - 1], cond_nets, args, [args] * S, gt_volume.clone(), views, stats)   # :653
    levels = [t for t in gt_cache if t is not None]
    cache = [gt[0, ::2, ...] - gt[0, 1::2, ...] for gt in levels]                            # :655
    buf = io.BytesIO()
    torch.save({'mean_vol_gt_cache': [v.cpu() for v in cache]}, buf)                        # main.py:377
    arrs = {"gt_volume": npy(gt_volume)}
    arrs.update({f"level_{i}": npy(t) for i, t in enumerate(levels)})
    arrs.update({f"cache_{i}": npy(t) for i, t in enumerate(cache)})
    arrs["file_bytes"] = np.frombuffer(buf.getvalue(), dtype=np.uint8).copy()
    dump("g19_meanvol", **arrs)
    # output step: B = 2 stored volumes -> the factor is 2**2
    stored0 = torch.randn(2, D, H, W, generator=g)
    gt0 = torch.randn(2, D, H, W, generator=g)
    vol_out = gt0[0] * std_vols + mean_vols                                                  # :1037
    vol_out -= vol_out.min()                                                                 # :1038
    vol_out_pred = (stored0[0] * 2 ** len(stored0)) * std_vols + mean_vols                   # :1041
    dump("g19_denorm", stored0=npy(stored0), gt0=npy(gt0), mean_vols=npy(mean_vols), std_vols=npy(std_vols),
         vol_out=npy(vol_out), vol_out_pred=npy(vol_out_pred))


def gen_extract_views():
    """g12: the reference's XLFMDatasetFull.extract_views on small frames, windows clipped at every border."""
    import_reference()
    import torch
    from XLFMDataset import XLFMDatasetFull
    g = torch.Generator().manual_seed(12)
    cases = {
        "even": ((2, 1, 40, 50), (16, 16), [(20, 25), (3, 4), (38, 47), (8, 45), (33, 2), (0, 1), (39, 49)]),
        "odd": ((1, 1, 31, 29), (15, 9), [(15, 14), (2, 1), (30, 28), (7, 3)]),
        "wide": ((1, 1, 24, 90), (8, 64), [(12, 45), (1, 10), (22, 80)]),
    }
    for name, (ishape, sub, coords) in cases.items():
        img = torch.randn(*ishape, generator=g)
        views = XLFMDatasetFull.extract_views(img, coords, list(sub), debug=False)
        mean, std = 0.37, 1.9
        dump(f"g12_extract_views_{name}", image=npy(img), coords=np.asarray(coords, dtype=np.int64), sub=np.asarray(sub),
             views=npy(views), normalized=npy((views - mean) / std), mean=np.float32(mean), std=np.float32(std))


def gen_step_grad():
    """g13: gradients of the training NLL (CWFA.py:966-978) of one CAT step w.r.t. every parameter and both
    conditions, from the reference's own graph + torch autograd -- what `full_loss.backward()` (CWFA.py:1002-1006)
    produces for the log-likelihood term.  Steps built as run_CWFA does (CWFA.py:498-510), small channel counts."""
    Ff, Fm, INN_utils, networks, unet, CWFA = import_reference()
    fresh_process_state(networks)
    import torch
    torch.set_num_threads(8)
    g = torch.Generator().manual_seed(1313)
    S, D, H, W = 3, 16, 12, 16
    for ix, n_ch in ((0, 8), (1, 8), (0, 64)):
        torch.manual_seed(300 + ix)
        np.random.seed(7)
        networks.networks_n_chans = n_ch
        Cn = D // 2 ** (ix + 1)
        cond_net, inns = networks.conditional_wavelet_flow(
            [D, H, W], [1, 29, H, W], networks.wavelet_flow_subnetwork2D,
            lambda: networks.cond_network(29, Cn, ix + 1, S, [], 4),
            n_internal_ch=n_ch, n_down_steps=ix + 1, use_permutations=True, block_type="CAT", n_blocks=4)
        inn = inns[ix].train()
        with torch.no_grad():
            for p in inn.parameters():
                if p.requires_grad and p.dtype == torch.float32:
                    p.add_(torch.randn(p.shape, generator=g) * 0.05)
        meta = {}
        for i, mdl in enumerate(inn.module_list):
            if type(mdl).__name__ == "PermuteDim":
                meta[f"meta/axis_{i}"] = np.int64(mdl.dims_to_permute[1])
        Dn = D // 2 ** ix
        B = 3
        x = torch.randn(B, Dn, H, W, generator=g)
        c = [torch.randn(B, Cn, H, W, generator=g).requires_grad_(), (0.3 * torch.randn(B, Cn, H, W, generator=g)).requires_grad_()]
        Z, log_jac_det = inn(x, c=c)
        loss = (0.5 * torch.norm(Z[0]) ** 2 - log_jac_det.mean()) / x.numel()      # upsampled_vol has x's shape (CWFA.py:911,978)
        loss.backward()
        grads = {"grad/" + k: npy(p.grad) for k, p in inn.named_parameters() if p.grad is not None}
        if n_ch == 8:
            # the default training loss: 0.40984 * F.mse_loss(curr_gt, upsampled_vol) + 0.59016 * NLL, upsampled_vol from the
            # inverse pass with a sampled z (CWFA.py:905-911,952-959,978,987; main.py:43,107); L1 variant as well
            import torch.nn.functional as F
            w_c = 0.40984
            z_in = 0.5 * torch.randn(B, Cn, H, W, generator=g)
            low_in = torch.randn(B, Cn, H, W, generator=g)
            for kind, fn in (("l2", F.mse_loss), ("l1", F.l1_loss)):
                for p in inn.parameters():
                    p.grad = None
                c2 = [t.detach().clone().requires_grad_() for t in c]
                xhat, _ = inn([z_in, low_in], c=c2, rev=True)
                Z2, ld2 = inn(x, c=c2)
                nll2 = (0.5 * torch.norm(Z2[0]) ** 2 - ld2.mean()) / xhat.numel()
                full = w_c * fn(x, xhat) + (1 - w_c) * nll2
                full.backward()
                grads.update({f"grad_{kind}/" + k: npy(p.grad) for k, p in inn.named_parameters() if p.grad is not None})
                grads.update({f"full_{kind}/loss": np.float64(full.item()), f"full_{kind}/recon": np.float64(fn(x, xhat).item()),
                              f"full_{kind}/gc0": npy(c2[0].grad), f"full_{kind}/gc1": npy(c2[1].grad)})
            grads.update({"full/z_in": npy(z_in), "full/low_in": npy(low_in), "full/xhat": npy(xhat), "full/w_c": np.float64(w_c)})
            # ... and with the condition computed by the step's condition net (eval mode, CWFA.py:527-528,893), so that the
            # gradients reach its parameters too (`optimizer_cond`, CWFA.py:1008-1012)
            cond_net.eval()
            with torch.no_grad():
                for p in cond_net.parameters():
                    p.add_(torch.randn(p.shape, generator=g) * 0.05)
            views = torch.randn(B, 29, H, W, generator=g)
            for p in list(inn.parameters()) + list(cond_net.parameters()):
                p.grad = None
            om = cond_net(views)[-1]
            c3 = [om, c[1].detach()]
            xhat3, _ = inn([z_in, low_in], c=c3, rev=True)
            Z3, ld3 = inn(x, c=c3)
            full3 = w_c * F.mse_loss(x, xhat3) + (1 - w_c) * (0.5 * torch.norm(Z3[0]) ** 2 - ld3.mean()) / xhat3.numel()
            full3.backward()
            grads.update({"cond/views": npy(views), "cond/omega": npy(om), "cond/loss": np.float64(full3.item())})
            grads.update({"condgrad/" + k: npy(p.grad) for k, p in cond_net.named_parameters() if p.grad is not None})
            grads.update({"flowgrad_cond/" + k: npy(p.grad) for k, p in inn.named_parameters() if p.grad is not None})
            grads.update(sd_arrays(cond_net, "condsd/"))
            # ... and with the condition net in TRAIN mode, as the reference runs the optimised step (CWFA.py:768,859):
            # Dropout3d(0.5) on the hidden Conv3d channels (networks.py:224).  The draw is pinned harness-side: the Dropout3d
            # module is swapped for a fixed keep / scale table with the same semantics (whole channels per sample, 1/(1-p)).
            if ix == 0 and n_ch == 8:
                blk = cond_net.subnetworks[0]
                K = blk.conv3d[0].out_channels
                mask = (torch.rand(B, K, generator=g) >= 0.5).float() / 0.5

                class FixedDrop(torch.nn.Module):
                    def forward(self, t):
                        return t * mask.view(B, K, 1, 1, 1)
                real_drop = blk.conv3d[2]
                assert isinstance(real_drop, torch.nn.Dropout3d) and real_drop.p == 0.5
                blk.conv3d[2] = FixedDrop()
                for p in list(inn.parameters()) + list(cond_net.parameters()):
                    p.grad = None
                om4 = cond_net(views)[-1]
                c4 = [om4, c[1].detach()]
                xhat4, _ = inn([z_in, low_in], c=c4, rev=True)
                Z4, ld4 = inn(x, c=c4)
                full4 = w_c * F.mse_loss(x, xhat4) + (1 - w_c) * (0.5 * torch.norm(Z4[0]) ** 2 - ld4.mean()) / xhat4.numel()
                full4.backward()
                blk.conv3d[2] = real_drop
                grads.update({"drop/mask": npy(mask), "drop/omega": npy(om4), "drop/loss": np.float64(full4.item())})
                grads.update({"dropgrad/" + k: npy(p.grad) for k, p in cond_net.named_parameters() if p.grad is not None})
        dump(f"g13_step_grad_k{ix}_ch{n_ch}", x=npy(x), c0=npy(c[0]), c1=npy(c[1]), z=npy(Z[0]), low=npy(Z[1]), loss=np.float64(loss.item()),
             gc0=npy(c[0].grad), gc1=npy(c[1].grad), D=np.int64(D), H=np.int64(H), W=np.int64(W), ix=np.int64(ix), S=np.int64(S),
             n_ch=np.int64(n_ch), **meta, **grads, **sd_arrays(inn))


def gen_block_grad():
    """g20: the default training loss of a flow step built with the DATA-DEPENDENT block types (main.py --INN_block_type: GLOW, AI1,
    RNVP, GIN): 0.40984 * mse(gt, xhat) + 0.59016 * NLL with xhat from the inverse pass on a sampled z (CWFA.py:905-911,
    952-987), forward values and the gradient of every parameter and of both conditions from the reference's own graph + torch
    autograd (coupling_layers.py:124-437, all_in_one_block.py:206-268 under `full_loss.backward()`, CWFA.py:1002-1006)."""
    Ff, Fm, INN_utils, networks, unet, CWFA = import_reference()
    fresh_process_state(networks)
    import torch
    import torch.nn.functional as F
    torch.set_num_threads(8)
    g = torch.Generator().manual_seed(2020)
    S, D, H, W, ix, n_ch, B = 3, 16, 12, 16, 0, 8, 2
    for bt in ("GLOW", "AI1", "RNVP", "GIN"):
        torch.manual_seed(400 + len(bt))
        np.random.seed(11)
        networks.networks_n_chans = n_ch
        Cn = D // 2 ** (ix + 1)
        cond_net, inns = networks.conditional_wavelet_flow(
            [D, H, W], [1, 29, H, W], networks.wavelet_flow_subnetwork2D,
            lambda: networks.cond_network(29, Cn, ix + 1, S, [], 4),
            n_internal_ch=n_ch, n_down_steps=ix + 1, use_permutations=True, block_type=bt, n_blocks=4)
        inn = inns[ix].train()
        with torch.no_grad():
            for p in inn.parameters():
                if p.requires_grad and p.dtype == torch.float32:
                    p.add_(torch.randn(p.shape, generator=g) * 0.05)
        meta = {}
        for i, mdl in enumerate(inn.module_list):
            if type(mdl).__name__ == "PermuteDim":
                meta[f"meta/axis_{i}"] = np.int64(mdl.dims_to_permute[1])
        x = torch.randn(B, D, H, W, generator=g)
        c = [torch.randn(B, Cn, H, W, generator=g).requires_grad_(), (0.3 * torch.randn(B, Cn, H, W, generator=g)).requires_grad_()]
        z_in = 0.5 * torch.randn(B, Cn, H, W, generator=g)
        low_in = torch.randn(B, Cn, H, W, generator=g)
        w_c = 0.40984
        xhat, _ = inn([z_in, low_in], c=c, rev=True)
        Z, ld = inn(x, c=c)
        nll = (0.5 * torch.norm(Z[0]) ** 2 - ld.mean()) / xhat.numel()
        full = w_c * F.mse_loss(x, xhat) + (1 - w_c) * nll
        full.backward()
        grads = {"grad/" + k: npy(p.grad) for k, p in inn.named_parameters() if p.grad is not None}
        dump(f"g20_blockgrad_{bt}", x=npy(x), c0=npy(c[0]), c1=npy(c[1]), z_in=npy(z_in), low_in=npy(low_in), xhat=npy(xhat), z=npy(Z[0]),
             low=npy(Z[1]), logdet=npy(ld), loss=np.float64(full.item()), nll=np.float64(nll.item()), w_c=np.float64(w_c),
             gc0=npy(c[0].grad), gc1=npy(c[1].grad), ix=np.int64(ix), n_ch=np.int64(n_ch), **meta, **grads, **sd_arrays(inn))


def gen_unet_grad():
    """g14: gradients of every UNet parameter (and of its input) in train mode -- batch-statistics BatchNorm, PReLU,
    max-pool, skip additions, transposed convolutions (unet.py:72-113,161-195) -- from the reference's own modules and
    torch autograd, for an arbitrary upstream gradient.  Dropout off (the reference's dropout draws from torch's RNG)."""
    Ff, Fm, INN_utils, networks, unet, CWFA = import_reference()
    fresh_process_state(networks)
    import torch
    torch.set_num_threads(8)
    g = torch.Generator().manual_seed(1414)
    for bias in (False, True):
        torch.manual_seed(41)
        u = unet.UNet(5, 4, depth=3, wf=3, drop_out=0, use_bias=bias, skip_conn=True, up_mode="upconv", batch_norm=True)
        with torch.no_grad():
            for m_ in u.modules():
                if isinstance(m_, torch.nn.BatchNorm2d):
                    m_.weight.copy_(torch.rand(m_.weight.shape, generator=g) + 0.5)
                    m_.bias.copy_(torch.randn(m_.bias.shape, generator=g) * 0.1)
                if isinstance(m_, torch.nn.PReLU):
                    m_.weight.copy_(0.1 + 0.3 * torch.rand(m_.weight.shape, generator=g))
        sd0 = sd_arrays(u)
        u.train()
        x = torch.randn(3, 5, 16, 16, generator=g).requires_grad_()      # square: the reference pools to (W//2, W//2), unet.py:79
        dy = torch.randn(3, 4, 16, 16, generator=g)
        y = u(x)
        (y * dy).sum().backward()
        grads = {"grad/" + k: npy(p.grad) for k, p in u.named_parameters() if p.grad is not None}
        dump(f"g14_unet_grad_bias{int(bias)}", x=npy(x), dy=npy(dy), y=npy(y), gx=npy(x.grad), **grads, **sd0)


def gen_meanbranch_grad():
    """g15: gradients through the LRNN's mean-volume branch at a small size -- m = ConvNeXt(ConvNeXt(mean)),
    att = GlobalAttention(mean), out = x + m*2*(att - 0.5) (networks.py:468-503,244-262,552-554) -- from the reference's
    modules and torch autograd.  drop_path off (it draws from torch's RNG)."""
    Ff, Fm, INN_utils, networks, unet, CWFA = import_reference()
    fresh_process_state(networks)
    import torch
    torch.set_num_threads(8)
    g = torch.Generator().manual_seed(1515)
    torch.manual_seed(51)
    cn1 = networks.ConvNeXt(6, 10, drop_prob=0.0, size=16).train()
    cn2 = networks.ConvNeXt(10, 6, drop_prob=0.0, size=16).train()
    ga = networks.GlobalAttention(6).train()
    with torch.no_grad():
        for cn in (cn1, cn2):
            cn.m[1].weight.copy_(1 + 0.2 * torch.randn(cn.m[1].weight.shape, generator=g))
            cn.m[1].bias.copy_(0.1 * torch.randn(cn.m[1].bias.shape, generator=g))
    mean = torch.randn(2, 6, 16, 16, generator=g)
    x = torch.randn(2, 6, 16, 16, generator=g).requires_grad_()
    dy = torch.randn(2, 6, 16, 16, generator=g)
    m = cn2(cn1(mean))
    att = ga(mean.view(2, 6, -1)).view(2, 6, 16, 16)
    out = x + m * 2 * (att - 0.5)
    (out * dy).sum().backward()
    arrs = {"mean": npy(mean), "x": npy(x), "dy": npy(dy), "out": npy(out), "m": npy(m), "gx": npy(x.grad)}
    for tag, mod in (("cn1", cn1), ("cn2", cn2), ("ga", ga)):
        arrs.update(sd_arrays(mod, f"sd_{tag}/"))
        arrs.update({f"grad_{tag}/" + k: npy(p.grad) for k, p in mod.named_parameters() if p.grad is not None})
    dump("g15_meanbranch_grad", **arrs)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "meanbranch_grad":
        gen_meanbranch_grad()
    elif len(sys.argv) > 1 and sys.argv[1] == "unet_grad":
        gen_unet_grad()
    elif len(sys.argv) > 1 and sys.argv[1] == "extract_views":
        gen_extract_views()
    elif len(sys.argv) > 1 and sys.argv[1] == "step_grad":
        gen_step_grad()
    elif len(sys.argv) > 1 and sys.argv[1] == "block_grad":
        gen_block_grad()
    elif len(sys.argv) > 1 and sys.argv[1] == "topology":
        gen_topology()
    elif len(sys.argv) > 1 and sys.argv[1] == "ai1_options":
        gen_ai1_options()
    elif len(sys.argv) > 1 and sys.argv[1] == "cache_output":
        gen_cache_and_output()
    else:
        main()
        gen_topology()
        gen_ai1_options()
        gen_cache_and_output()
        gen_extract_views()
        gen_step_grad()
        gen_block_grad()
        gen_unet_grad()
        gen_meanbranch_grad()
