"""CPU: host-side logic of the product -- graph construction, module/state_dict layout, permutation tables, plan lowering,
construction-time RNG parity -- against the golden fixtures taken from the imported reference."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, sd_of

import cwfa_amd
from cwfa_amd import networks as N
from cwfa_amd.FrEIA import framework as Ff
from cwfa_amd.FrEIA import modules as Fm


def build_step(bt, ix, D=16, H=12, W=16, S=3, n_ch=8, cond_ch=4):
    Cn = D // 2 ** (ix + 1)
    cond_net, inns = N.conditional_wavelet_flow(
        [D, H, W], [1, 29, H, W], N.wavelet_flow_subnetwork2D, lambda: N.cond_network(29, Cn, ix + 1, S, [], cond_ch),
        n_internal_ch=n_ch, n_down_steps=ix + 1, use_permutations=True, block_type=bt, n_blocks=4)
    return cond_net, inns[ix]


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(GOLDEN + "/g09_*.npz")))
def test_step_graph_matches_reference(name):
    fx = load_golden(name)
    bt, ix = name.split("_")[2], int(name[-1])
    np.random.seed(12345)                       # tables must not depend on the incoming numpy state
    cond_net, g = build_step(bt, ix)
    assert [n.name for n in g.node_list] == list(fx["meta/node_names"])
    assert [n.name for n in g.condition_nodes] == list(fx["meta/cond_node_names"])
    assert [n.name for n in g.out_nodes] == list(fx["meta/out_node_names"])
    assert np.array_equal(np.array(g.dims_c), fx["meta/dims_c"])
    assert np.array_equal(np.array(g.global_out_shapes), fx["meta/global_out_shapes"])
    ref_sd = sd_of(fx)
    sd = g.state_dict()
    assert list(sd.keys()) == list(ref_sd.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(ref_sd[k].shape) and sd[k].dtype == ref_sd[k].dtype, k
    for i, m in enumerate(g.module_list):
        assert type(m).__name__ == str(fx[f"meta/module_{i}"])
        if hasattr(m, "perm") and type(m).__name__ != "HaarDownsampling":
            assert torch.equal(m.perm, ref_sd[f"module_list.{i}.perm"]), f"perm of module {i} differs"
            assert torch.equal(m.perm_inv, ref_sd[f"module_list.{i}.perm_inv"])
        if f"meta/axis_{i}" in fx:
            assert m.axis == int(fx[f"meta/axis_{i}"])
        if type(m).__name__ == "AllInOneBlock":
            assert torch.equal(m.w_perm, ref_sd[f"module_list.{i}.w_perm"])
    g.load_state_dict(ref_sd)
    # every step graph lowers to a plan: all-CAT steps to ONE fused chain launch, the data-dependent block types to the mixed
    # plan (fused Haar / Split / first CAT / permutation prefix + blocks with the coupling in the conv epilogue)
    assert g._plan is not None and hasattr(g._plan, "rest") == (bt != "CAT")
    if bt != "CAT":
        assert [k for k, _ in g._plan.fused] == ["cat", "perm"] and sum(k == "blk" for k, _ in g._plan.rest) == 4


def test_condition_net_layout_and_shared_prelu():
    fx = load_golden("g08_omega_c8_k32")
    net = N.cond_network(29, 8, 1, 5, [], 32)
    ref = sd_of(fx)
    assert list(net.state_dict().keys()) == list(ref.keys())
    net.load_state_dict(ref)
    other = N.cond_network(29, 4, 2, 5, [], 32)
    assert other.subnetworks[0].relu is net.subnetworks[0].relu          # networks.py:209 default-arg aliasing


def test_lrnn_default_init_matches_reference_rng_stream():
    """Same torch seed -> same 63.7 M default-initialised parameters as the reference's Encoder (checksums)."""
    fx = load_golden("g11_lrnn_small")
    torch.manual_seed(int(fx["seed_init"]))
    enc = N.Encoder(29, 6, 5, 64, True)
    sd = enc.state_dict()
    keys = [k[4:] for k in fx if k.startswith("chk/")]
    assert list(sd.keys()) == keys
    for k in keys:
        v = sd[k].double()
        got = np.array([v.sum().item(), v.abs().sum().item(), float(v.numel())])
        assert np.allclose(got, fx["chk/" + k], rtol=1e-12, atol=1e-12), k


def test_unet_layout():
    from cwfa_amd.unet import UNet
    for bias in (0, 1):
        fx = load_golden(f"g11_unet_bias{bias}")
        u = UNet(5, 4, depth=3, wf=3, drop_out=0, use_bias=bool(bias), skip_conn=True, up_mode="upconv", batch_norm=True)
        assert list(u.state_dict().keys()) == list(sd_of(fx).keys())
        u.load_state_dict(sd_of(fx))


def test_graph_errors_and_generic_topology():
    a = Ff.InputNode(4, 8, 8, name="in")
    p = Ff.Node(a, Fm.PermuteRandom, {"seed": 3}, name="p")
    s = Ff.Node(p, Fm.Split, {"section_sizes": (1, 3), "dim": 0}, name="s")
    o0, o1 = Ff.OutputNode(s.out0, name="o0"), Ff.OutputNode(s.out1, name="o1")
    g = Ff.GraphINN([a, p, s, o0, o1])
    assert g._plan is None and g.global_out_shapes == [(1, 8, 8), (3, 8, 8)]
    with pytest.raises(ValueError, match="Got 2 inputs, but expected 1"):
        g([torch.zeros(1, 4, 8, 8)] * 2)
    with pytest.raises(ValueError, match="Got 1 conditions, but expected 0"):
        g(torch.zeros(1, 4, 8, 8), c=[torch.zeros(1)])
    with pytest.raises(ValueError, match="not in the node_list"):
        Ff.GraphINN([a, s, o0, o1])
    with pytest.warns(DeprecationWarning):
        Ff.ReversibleGraphNet([a, p, s, o0, o1], verbose=False)
    seq = Ff.SequenceINN(4, 8, 8)
    seq.append(Fm.PermuteRandom, seed=1)
    seq.append(Fm.HaarDownsampling, order_by_wavelet=True)
    assert seq.shapes[-1] == (16, 4, 4)


def test_install_aliases():
    import sys
    saved = {k: sys.modules.get(k) for k in ("FrEIA", "FrEIA.framework", "FrEIA.modules", "INN_utils", "networks", "unet")}
    try:
        cwfa_amd.install()
        import FrEIA.framework as F2
        import networks as n2
        assert F2.GraphINN is Ff.GraphINN and n2.conditional_wavelet_flow is N.conditional_wavelet_flow
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_checkpoint_layout_roundtrip(tmp_path):
    """the reference's checkpoint files (networks.py:708-756): file naming, dict keys, discovery
    of the newest epoch per step, and state_dicts that load back into freshly built nets."""
    import argparse
    torch.manual_seed(0)
    np.random.seed(0)
    build = lambda: N.conditional_wavelet_flow([8, 8, 8], [1, 29, 8, 8], N.wavelet_flow_subnetwork2D,
                                               lambda: N.cond_network(29, 4, 1, 3, [], 4), n_internal_ch=8, n_down_steps=1,
                                               use_permutations=True, block_type="CAT", n_blocks=2)
    cond, inns = build()
    args = argparse.Namespace(INN_down_steps=0, lr=1e-3)
    stats = (0.1, 1.5, 0.0, 1.0, 0.2, 2.0)
    for ep in (3, 11, 7):
        N.serialize_INN_step(inns[0], cond[0] if isinstance(cond, (list, tuple)) else cond, None, stats, args, ep, str(tmp_path))
    args.INN_down_steps = 1
    N.serialize_INN_step(None, None, None, stats, args, 5, str(tmp_path), posfix="")
    found = N.load_INN_steps(str(tmp_path))
    assert sorted(found) == [0, 1] and found[0][0] == 11 and found[1][0] == 5
    assert found[0][1].endswith("model_step_0__ep_11")
    assert N.load_INN_steps(str(tmp_path), epoch=7)[0][0] == 7
    ck = torch.load(found[0][1], weights_only=False)          # our own file (argparse.Namespace inside, as the reference writes)
    assert set(ck) == {"epoch", "args", "INN_state_dict", "condition_state_dict", "optimizer_state_dict", "training_statistics"}
    torch.manual_seed(1)
    np.random.seed(0)                                          # permutations come from numpy's global RNG: same graph
    cond2, inns2 = build()
    inns2[0].load_state_dict(ck["INN_state_dict"])
    for (k, a), (_, b) in zip(inns[0].state_dict().items(), inns2[0].state_dict().items()):
        assert torch.equal(a, b), k
    assert ck["training_statistics"] == stats and ck["optimizer_state_dict"] is None


def test_training_host_helpers_without_a_process_group():
    """Host-side pieces of the training path that need neither a GPU nor torch.distributed."""
    import torch
    from cwfa_amd import CWFA, ops, training
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.tensor([1.0, 2.0, 3.0])
    assert training.allreduce_gradients([p]) == 0                  # no process group: nothing to exchange
    training.sgd_step([p], 0.5)
    assert torch.equal(p.detach(), torch.tensor([0.5, 0.0, -0.5]))
    scores = torch.tensor([[-1.0, -2.0], [-1.5, -0.5]], dtype=torch.float64)
    assert CWFA.allgather_scores(scores) is scores
    assert CWFA.detect_ood(scores, 0, -1.33).tolist() == [False, True]
    assert CWFA.detect_ood(scores, 1, -1.33).tolist() == [True, False]
    with pytest.raises(ValueError):
        ops.set_precision("fp8")
    with pytest.raises(NotImplementedError):
        training.step_backward(type("G", (), {"_plan": None})(), None, [])


def test_mean_volume_cache_file_round_trip(tmp_path):
    """The cache file of main.py:377 as the reference wrote it (fixture bytes) loads with the weights-only loader, the
    discovery rule of CWFA.py:637-640 finds it, and our writer produces a file the same loader reads back identically."""
    import numpy as np
    from conftest import load_golden
    from cwfa_amd import CWFA
    fx = load_golden("g19_meanvol")
    p = tmp_path / "mean_vol_7Imgs_ds_3_train"
    p.write_bytes(bytes(np.asarray(fx["file_bytes"], dtype=np.uint8)))
    got = CWFA.load_mean_volume_cache(str(tmp_path), dataset_id=3, split="train")
    assert got is not None and len(got) == 3
    for i, v in enumerate(got):
        assert torch.equal(v, torch.from_numpy(fx[f"cache_{i}"]))
    assert CWFA.load_mean_volume_cache(str(tmp_path), dataset_id=4, split="train") is None
    q = tmp_path / "mean_vol_7Imgs_ds_9_val"
    CWFA.save_mean_volume_cache(str(q), got)
    again = CWFA.load_mean_volume_cache(str(q))
    assert all(torch.equal(a, b) for a, b in zip(again, got))


def test_plan_lowering_of_actnorm_and_mixed_graphs_on_cpu():
    """Graph construction and plan lowering are host logic (no kernel runs): a step graph with ActNorm nodes keeps a plan whose
    chain carries 'act' stages and asks for the node walk until the ActNorms are initialised; a GLOW graph lowers to the mixed
    plan with the [CAT, permutation] prefix fused; a graph whose Haar node has rebalance != 1 falls back to the node walk."""
    from cwfa_amd import networks as N, INN_utils
    from cwfa_amd.FrEIA import framework as Ff, modules as Fm
    N.networks_n_chans = 8
    np.random.seed(0)
    torch.manual_seed(0)

    def graph(block, actnorm, rebalance=1.0):
        D, H, W = 8, 8, 16
        nodes = [Ff.InputNode(D, H, W, name="input")]
        nodes.append(Ff.Node(nodes[-1], INN_utils.HaarTransform1D, {"order_by_wavelet": True, "rebalance": rebalance}, name="haar"))
        split = Ff.Node(nodes[-1], Fm.Split, {"section_sizes": (D // 2, D // 2), "dim": 0}, name="split")
        nodes.append(split)
        cond = Ff.ConditionNode(D // 2, H, W, name="cond")
        nodes.append(cond)
        nodes.append(Ff.Node(split.out1, Fm.ConditionalAffineTransform, {"subnet_constructor": N.wavelet_flow_subnetwork2D},
                             conditions=[cond], name="cat0"))
        if actnorm:
            nodes.append(Ff.Node(nodes[-1], Fm.ActNorm, {}, name="an0"))
        nodes.append(Ff.Node(nodes[-1], Fm.PermuteRandom, {"seed": 1}, name="p1"))
        nodes.append(Ff.Node(nodes[-1], block, {"subnet_constructor": N.wavelet_flow_subnetwork2D}, conditions=[cond], name="b1"))
        nodes.append(Ff.OutputNode(nodes[-1], name="z"))
        nodes.append(Ff.OutputNode(split.out0, name="low"))
        return Ff.GraphINN(nodes)

    g = graph(Fm.ConditionalAffineTransform, True)
    assert [k for k, _ in g._plan.chain] == ["cat", "act", "perm", "cat"] and g._plan.needs_walk() and not hasattr(g._plan, "rest")
    g.load_state_dict(g.state_dict())                      # loading a checkpoint switches the data-dependent initialisation off
    assert not g._plan.needs_walk()
    g = graph(Fm.GLOWCouplingBlock, False)
    assert [k for k, _ in g._plan.fused] == ["cat", "perm"] and [k for k, _ in g._plan.rest] == ["blk"]
    try:
        g = graph(Fm.ConditionalAffineTransform, False, rebalance=0.5)
    except TypeError:
        return                                             # this HaarTransform1D takes no rebalance argument: nothing to check
    assert g._plan is None


# ------------------------------------------------------------------------------------------------ bench.py launcher plumbing
def test_bench_self_launch_plumbing(monkeypatch, tmp_path):
    """`python bench.py --gpus N` without a launcher must start the driver's own launch (torch.distributed.run, one rank per
    GPU, rendezvous on 127.0.0.1) BEFORE touching the GPU, forward the arguments unchanged and return the ranks' status."""
    import importlib
    import subprocess
    import sys as _sys
    bench = importlib.import_module("bench")
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "7", "--warmup", "2"], port=29512)
    assert cmd[:3] == [_sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29512"
    assert cmd[-7].endswith("bench.py") and cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert isinstance(int(bench.launch_command(2, [])[bench.launch_command(2, []).index("--master-port") + 1]), int)   # a free port is picked

    seen = {}

    def fake_run(c, env=None, **kw):
        seen["cmd"], seen["env"] = c, env
        return subprocess.CompletedProcess(c, 3)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(_sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    import torch
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: (_ for _ in ()).throw(AssertionError("GPU touched before the launch")))
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 3                                  # the ranks' status is the launcher's status
    else:
        raise AssertionError("main() must exit through the launcher")
    assert seen["cmd"][-4:] == ["--gpus", "2", "--steps", "1"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher with a different world size: refuse, never re-launch
    monkeypatch.setenv("WORLD_SIZE", "4")
    seen.clear()
    try:
        bench.main()
    except SystemExit as e:
        assert "must agree" in str(e.code) and not seen
