"""CPU: the oracle (oracle/cwfa_oracle.py) against the golden vectors generated from the imported reference."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_close, load_golden, sd_of
from oracle import cwfa_oracle as O

T = torch.from_numpy
TOL = 2e-6   # same ATen kernels on both sides; only op association differs


def test_haar1d_exact():
    fx = load_golden("g01_haar1d")
    yf, jf = O.haar1d(T(fx["x"]), False)
    yr, jr = O.haar1d(T(fx["x"]), True)
    assert torch.equal(yf, T(fx["y_fwd"])) and torch.equal(yr, T(fx["y_rev"]))
    assert abs(jf - float(fx["jac_fwd"])) < 1e-12 and abs(jr - float(fx["jac_rev"])) < 1e-12


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(GOLDEN + "/g02_*.npz")))
def test_haar2d(name):
    fx = load_golden(name)
    kw = dict(order_by_wavelet=bool(fx["order_by_wavelet"]), rebalance=float(fx["rebalance"]))
    yf, jf = O.haar2d(T(fx["x"]), False, **kw)
    xr, jr = O.haar2d(T(fx["z"]), True, **kw)
    assert_close(yf, fx["y_fwd"], TOL, "fwd")
    assert_close(xr, fx["x_rev"], TOL, "rev")
    assert abs(jf - float(fx["jac_fwd"])) < 1e-9 and abs(jr - float(fx["jac_rev"])) < 1e-9


def test_perm_recipe_bit_exact():
    fx = load_golden("g03_perms")
    D, H, W = 16, 12, 16
    for bt in ["CAT", "GLOW", "RNVP", "GIN", "AI1"]:
        for k in range(2):
            C = D // 2 ** (k + 1)
            np.random.seed(999)        # the result must not depend on the incoming global state
            rec = O.perm_recipe(k, C, H, W, bt, 4, True)
            for j, ent in enumerate(rec):
                m = 3 + 2 * j
                assert np.array_equal(ent["perm"], fx[f"{bt}/k{k}/m{m}/perm"]), (bt, k, m)
                if f"{bt}/k{k}/m{m}/axis" in fx:
                    assert ent["axis"] == int(fx[f"{bt}/k{k}/m{m}/axis"])
                else:
                    assert ent["axis"] == 1
                if bt == "AI1" and j < 4:
                    w = fx[f"AI1/k{k}/m{m + 1}/w_perm"]
                    assert np.array_equal(np.argmax(w, 1), ent["ai1_perm"])


G4 = sorted(os.path.basename(p)[:-4] for p in glob.glob(GOLDEN + "/g04_*.npz"))


@pytest.mark.parametrize("name", G4)
def test_coupling_blocks(name):
    fx = load_golden(name)
    _, bname, cl = name.split("_")
    sd = sd_of(fx)
    x, c = T(fx["x"]), [T(fx["c"])]
    for rev, key in ((False, "fwd"), (True, "rev")):
        if bname == "CAT":
            y, j = O.block_cat(sd, "", x, c, rev, False, 1.5, cl)
        elif bname == "ONESIDED":
            y, j = O.block_onesided(sd, "", x, c, rev, 1.5, cl)
        else:
            y, j = O.block_two_sided(sd, "", x, c, rev, bname, 1.5, cl)
        assert_close(y, fx["y_" + key], TOL, f"{name} y {key}")
        assert_close(j, fx["jac_" + key], 1e-5, f"{name} jac {key}") if np.abs(fx["jac_" + key]).max() > 0 \
            else None


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(GOLDEN + "/g05_*.npz")))
def test_all_in_one(name):
    fx = load_golden(name)
    sd = sd_of(fx)
    c = [T(fx["c"])] if fx["c"].size else []
    gin = "gin1" in name
    for rev, key in ((False, "fwd"), (True, "rev")):
        y, j = O.block_ai1(sd, "", T(fx["x"]), c, rev, gin)
        assert_close(y, fx["y_" + key], 5e-6, f"{name} y {key}")
        assert_close(j, fx["jac_" + key], 1e-5, f"{name} jac {key}")


def test_actnorm():
    fx = load_golden("g06_actnorm")
    scale, bias = O.actnorm_init(T(fx["x"]))
    assert_close(scale, fx["sd/scale"], TOL)
    assert_close(bias, fx["sd/bias"], 1e-5)
    y, j = O.actnorm(scale, bias, T(fx["x"]), False)
    xr, jr = O.actnorm(scale, bias, T(fx["z"]), True)
    assert_close(y, fx["y_fwd"], TOL)
    assert_close(xr, fx["x_rev"], TOL)
    assert_close(j, fx["jac_fwd"], TOL)
    assert_close(jr, fx["jac_rev"], TOL)


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(GOLDEN + "/g07_*.npz")))
def test_subnet(name):
    fx = load_golden(name)
    y = O.subnet(sd_of(fx), "", T(fx["x"]), "first1" in name)
    assert_close(y, fx["y"], TOL, name)


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(GOLDEN + "/g08_*.npz")))
def test_omega(name):
    fx = load_golden(name)
    assert_close(O.omega_net(sd_of(fx), T(fx["x"])), fx["y"], TOL, name)


def _axes(fx):
    return {int(k.split("_")[-1]): int(v) for k, v in fx.items() if k.startswith("meta/axis_")}


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(GOLDEN + "/g09_*.npz")))
def test_flow_step(name):
    fx = load_golden(name)
    bt = name.split("_")[2]
    sd = sd_of(fx)
    axes = {i: 1 for i in range(3, 12, 2)}
    axes.update(_axes(fx))
    c = [T(fx["c0"]), T(fx["c1"])]
    (z, low), jf = O.flow_step(sd, T(fx["x"]), c, False, axes, bt)
    assert_close(z, fx["z"], 1e-5, "z")
    assert torch.equal(low, T(fx["low"]))
    assert_close(jf, fx["jac_fwd"], 1e-5, "jac fwd")
    xr, jr = O.flow_step(sd, (T(fx["z"]), T(fx["low"])), c, True, axes, bt)
    assert_close(xr, fx["x_rev"], 1e-5, "x_rev")
    assert_close(jr, fx["jac_rev"], 1e-5, "jac rev")
    x0, _ = O.flow_step(sd, (torch.zeros_like(z), T(fx["low"])), c, True, axes, bt)
    assert_close(x0, fx["x_rev_z0"], 1e-5, "x_rev_z0")
    # the recipe reproduces the axes the reference drew
    rec = O.perm_recipe(int(fx["ix"]), z.shape[1], int(fx["H"]), int(fx["W"]), bt)
    for j, ent in enumerate(rec):
        assert ent["axis"] == axes[3 + 2 * j]


def test_pipeline_and_nll():
    fx = load_golden("g10_pipeline")
    S = int(fx["S"])
    gt = T(fx["gt"])
    pyr = O.pyramid_forward(gt, S - 1)
    for n in range(S):
        assert torch.equal(pyr[n], T(fx[f"gt_cache_{n}"]))
    cond_input = (T(fx["views"]) - float(fx["mean_imgs"])) / float(fx["std_imgs"])
    steps = []
    for n in range(S - 1):
        C = gt.shape[1] // 2 ** (n + 1)
        rec = O.perm_recipe(n, C, gt.shape[2], gt.shape[3], "CAT")
        steps.append({"inn": sd_of(fx, f"inn{n}/"), "omega": sd_of(fx, f"omega{n}/"),
                      "axes": {3 + 2 * j: e["axis"] for j, e in enumerate(rec)}})
    mean_cache = [T(fx[f"mean_cache_{n}"]) for n in range(S - 1)]
    vols = O.inverse_pass(steps, T(fx["low"]), cond_input, mean_cache)
    for i, n in enumerate(range(S - 2, -1, -1)):
        assert_close(vols[i + 1], fx[f"up_{n}"], 1e-5, f"up_{n}")
    # evaluate_INN_forward (CWFA.py:134-196): zero conditions, per-step loss / prior / logjac
    x = gt
    for n in range(S - 1):
        C = x.shape[1] // 2
        zc = torch.zeros(x.shape[0], C, x.shape[2], x.shape[3])
        (z, low), jac = O.flow_step(steps[n]["inn"], x, [zc, zc], False, steps[n]["axes"])
        numel = low.numel()
        err = float(torch.norm(z) ** 2)
        loss = float(((0.5 * err - jac) / numel).mean())
        assert abs(loss - fx["losses"][n]) <= 1e-5 * abs(fx["losses"][n])
        assert abs(0.5 * err / numel - fx["prior"][n]) <= 1e-5 * abs(fx["prior"][n])
        assert abs(float(jac.mean()) / numel - fx["logjac"][n]) <= 1e-5 * abs(fx["logjac"][n]) + 1e-9
        # shard-decomposed NLL equals the batch formula of CWFA.py:978
        t0 = O.nll_terms(z[:1], jac[:1])
        t1 = O.nll_terms(z[1:], jac[1:])
        nll = O.nll_from_terms(t0[0] + t1[0], t0[1] + t1[1], t0[2] + t1[2], x.numel())
        ref = float((0.5 * torch.norm(z) ** 2 - jac.mean()) / x.numel())       # CWFA.py:978: / upsampled_vol.numel()
        assert abs(nll - ref) <= 1e-5 * abs(ref)
        # per-sample log-likelihoods (the OOD score) sum back to the reference's batch quantities
        ll = O.step_log_likelihood(z, jac, low[0].numel())
        want = numel * (fx["prior"][n] - x.shape[0] * fx["logjac"][n])
        assert abs(float(-(ll * low[0].numel()).sum()) - want) <= 1e-5 * abs(want)
        x = low


@pytest.mark.parametrize("bias", [0, 1])
def test_unet(bias):
    fx = load_golden(f"g11_unet_bias{bias}")
    sd = sd_of(fx)
    x = T(fx["x"])
    assert_close(O.unet(sd, x, train=False), fx["y_eval"], 1e-5, "eval")
    assert_close(O.unet(sd, x, train=True), fx["y_train"], 1e-5, "train")
    assert_close(O.unet(sd, x[:1], train=True), fx["y_train_b1"], 1e-5, "train b1")


def test_convnext_attention():
    fx = load_golden("g11_convnext")
    assert_close(O.convnext(sd_of(fx), "", T(fx["x"])), fx["y"], 1e-5)
    fx = load_golden("g11_attention")
    assert_close(O.global_attention(sd_of(fx), "", T(fx["x"])), fx["y"], TOL)


@pytest.mark.parametrize("name", ["even", "odd", "wide"])
def test_extract_views(name):
    """the lenslet crop in front of the path; bit-exact (it is a gather + one fp32 sub/div)."""
    fx = load_golden(f"g12_extract_views_{name}")
    img, coords, sub = T(fx["image"]), fx["coords"].tolist(), fx["sub"].tolist()
    assert np.array_equal(O.extract_views(img, coords, sub).numpy(), fx["views"])
    assert np.array_equal(O.extract_views(img, coords, sub, float(fx["mean"]), float(fx["std"])).numpy(), fx["normalized"])


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(GOLDEN + "/g13_*.npz")))
def test_step_gradients_vs_reference_autograd(name):
    """The oracle is plain differentiable torch: autograd through it must give the gradients the reference's own graph
    gave for the training NLL of CWFA.py:966-978 (fixture g13: every parameter with a gradient, both conditions)."""
    fx = load_golden(name)
    sd = {k: (v.clone().requires_grad_() if v.dtype.is_floating_point else v) for k, v in sd_of(fx).items()}
    axes = {i: 1 for i in range(3, 12, 2)}
    axes.update(_axes(fx))
    x = T(fx["x"])
    c = [T(fx["c0"]).requires_grad_(), T(fx["c1"]).requires_grad_()]
    (z, low), jac = O.flow_step(sd, x, c, False, axes, "CAT")
    loss = (0.5 * torch.norm(z) ** 2 - jac.mean()) / x.numel()
    assert abs(float(loss) - float(fx["loss"])) <= 1e-5 * abs(float(fx["loss"]))
    loss.backward()
    names = [k[len("grad/"):] for k in fx if k.startswith("grad/")]
    assert len(names) == 80
    for k in names:
        assert_close(sd[k].grad, fx["grad/" + k], 1e-4, k)
    assert_close(c[0].grad, fx["gc0"], 1e-4, "d loss / d omega")
    assert_close(c[1].grad, fx["gc1"], 1e-4, "d loss / d mean detail")


@pytest.mark.parametrize("name", ["g13_step_grad_k0_ch8", "g13_step_grad_k1_ch8"])
@pytest.mark.parametrize("kind", ["l2", "l1"])
def test_full_training_loss_gradients_vs_reference_autograd(name, kind):
    """The default training loss of a flow step (0.40984 * mse / l1(gt, xhat) + 0.59016 * NLL, CWFA.py:905-911,952-987)
    through the oracle's inverse + forward pass against the reference's autograd."""
    import torch.nn.functional as F
    fx = load_golden(name)
    sd = {k: (v.clone().requires_grad_() if v.dtype.is_floating_point else v) for k, v in sd_of(fx).items()}
    axes = {i: 1 for i in range(3, 12, 2)}
    axes.update(_axes(fx))
    x = T(fx["x"])
    c = [T(fx["c0"]).requires_grad_(), T(fx["c1"]).requires_grad_()]
    w_c = float(fx["full/w_c"])
    xhat, _ = O.flow_step(sd, (T(fx["full/z_in"]), T(fx["full/low_in"])), c, True, axes, "CAT")
    assert_close(xhat, fx["full/xhat"], 1e-5, "xhat")
    (z, low), jac = O.flow_step(sd, x, c, False, axes, "CAT")
    recon = (F.mse_loss if kind == "l2" else F.l1_loss)(x, xhat)
    full = w_c * recon + (1 - w_c) * (0.5 * torch.norm(z) ** 2 - jac.mean()) / x.numel()
    assert abs(float(full) - float(fx[f"full_{kind}/loss"])) <= 1e-5 * abs(float(fx[f"full_{kind}/loss"]))
    full.backward()
    for k in [k[len(f"grad_{kind}/"):] for k in fx if k.startswith(f"grad_{kind}/")]:
        assert_close(sd[k].grad, fx[f"grad_{kind}/" + k], 1e-4 if kind == "l2" else 5e-4, k)
    assert_close(c[0].grad, fx[f"full_{kind}/gc0"], 5e-4, "d loss / d omega")


@pytest.mark.parametrize("bias", [0, 1])
def test_unet_gradients_vs_reference_autograd(bias):
    """Autograd through the oracle's train-mode UNet against the reference's own gradients (fixture g14)."""
    fx = load_golden(f"g14_unet_grad_bias{bias}")
    sd = {k: (v.clone().requires_grad_() if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd_of(fx).items()}
    x = T(fx["x"]).requires_grad_()
    y = O.unet(sd, x, train=True)
    assert_close(y, fx["y"], 1e-5, "train-mode forward")
    (y * T(fx["dy"])).sum().backward()
    assert_close(x.grad, fx["gx"], 1e-4, "input gradient")
    for k in [k[len("grad/"):] for k in fx if k.startswith("grad/")]:
        assert_close(sd[k].grad, fx["grad/" + k], 1e-3 if fx["grad/" + k].size == 1 else 1e-4, k)


def test_mean_branch_gradients_vs_reference_autograd():
    """Autograd through the oracle's ConvNeXt / GlobalAttention / combine against the reference's gradients (fixture g15)."""
    fx = load_golden("g15_meanbranch_grad")
    sds = {t: {k: v.clone().requires_grad_() for k, v in sd_of(fx, f"sd_{t}/").items()} for t in ("cn1", "cn2", "ga")}
    mean, x = T(fx["mean"]), T(fx["x"]).requires_grad_()
    m = O.convnext(sds["cn2"], "", O.convnext(sds["cn1"], "", mean))
    att = O.global_attention(sds["ga"], "", mean)
    out = x + m * 2 * (att - 0.5)
    assert_close(out, fx["out"], 1e-5, "combined output")
    (out * T(fx["dy"])).sum().backward()
    for t in ("cn1", "cn2", "ga"):
        for k in [k[len(f"grad_{t}/"):] for k in fx if k.startswith(f"grad_{t}/")]:
            assert_close(sds[t][k].grad, fx[f"grad_{t}/" + k], 1e-4, f"{t}.{k}")


def test_concat_and_fixed1x1conv_golden():
    """FrEIA API surface outside the default graphs: Concat (graph_topology.py:92-152), Fixed1x1Conv (fixed_transforms.py:95-133)."""
    fx = load_golden("g16_concat")
    y, j = O.concat([T(fx["x0"]), T(fx["x1"]), T(fx["x2"])])
    assert torch.equal(y, T(fx["y_fwd"])) and j == float(fx["jac_fwd"])
    parts, jr = O.concat(None, rev_input=T(fx["z"]), sizes=[3, 2, 4])
    assert all(torch.equal(p_, T(fx[f"r{i}"])) for i, p_ in enumerate(parts)) and jr == float(fx["jac_rev"])
    fx = load_golden("g17_fixed1x1conv")
    y, j = O.fixed1x1conv(T(fx["M"]), T(fx["x"]), False)
    assert_close(y, fx["y_fwd"], 2e-6)
    assert abs(j - float(fx["jac_fwd"])) <= 1e-5 * abs(float(fx["jac_fwd"]))
    y, j = O.fixed1x1conv(T(fx["M"]), T(fx["x"]), True)
    assert_close(y, fx["y_rev"], 2e-5)
    assert abs(j - float(fx["jac_rev"])) <= 1e-5 * abs(float(fx["jac_rev"]))


def test_mean_volume_cache_and_output_step_golden():
    """the mean-volume cache (CWFA.py:646-655) from the pyramid levels and the output de-normalisation
    (CWFA.py:1035-1044, 2**len factor)."""
    fx = load_golden("g19_meanvol")
    levels = [T(fx[f"level_{i}"]) for i in range(3)]
    for i, v in enumerate(O.mean_volume_cache(levels)):
        assert torch.equal(v, T(fx[f"cache_{i}"]))
    for i, lv in enumerate(O.pyramid_forward(T(fx["gt_volume"]), 2)):
        assert_close(lv, fx[f"level_{i}"], 2e-6)
    fx = load_golden("g19_denorm")
    assert torch.equal(O.denormalise_prediction(T(fx["stored0"]), T(fx["std_vols"]), T(fx["mean_vols"])), T(fx["vol_out_pred"]))
    assert torch.equal(O.denormalise_ground_truth(T(fx["gt0"]), T(fx["std_vols"]), T(fx["mean_vols"])), T(fx["vol_out"]))
