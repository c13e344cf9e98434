#!/usr/bin/env python3
"""Headline benchmark: inverse-pass (reconstruction) throughput, volumes/s, 512x512x96 fp32 (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one full inverse pass over one batch of synthetic light-field views resident in HBM:
LRNN (train mode, as CWFA.py:532) -> for n = 3..0: condition net Omega_n, 5 coupling sub-networks, fused flow chain
(config 3 of BASELINE.json: 4-scale CWFA + LRNN, CAT blocks, main.py defaults).  Every rank reconstructs its own
independent volumes (weak scaling, no data-path collective); value = all volumes of all ranks / max-over-ranks time.

Arithmetic of the timed region (`--precision`, default "split"): fp32 tensors, fp32 accumulation, every convolution
product formed from an EXACT three-way bf16 split of both fp32 operands (six bf16 MFMA products per fp32 product, dropped
terms <= 2^-24 relative): fp32-equivalent arithmetic on the bf16 matrix pipe; full-size parity under the fp32 bound in
tests/test_gpu_parity.py::test_full_config3_inverse_vs_oracle.  The same step on the plain fp32 MFMA kernels is measured
after the timed region and stays on the line as `fp32_mfma`.

Extra objects on the JSON line: `roofline` / `roofline_second` (the two conv kernels with the largest time share, HIP
events on the launch stream), `roofline_dwt` (the in-path fused chain kernel that holds the inverse Haar; the standalone
Haar kernel as a secondary field), `forward_nll` (BASELINE.json configs[3]: batch 4 per GPU, four flow steps + condition
nets, one all-reduce of the NLL sums at N > 1), `fp32_mfma`, `bf16` (configs[4]), `experiment_train_step`,
`cpu_baseline` (the CPU oracle timed on this host, rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np      # noqa: E402
import torch            # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense (AMD's 5 PF headline includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0
SPLIT_PRODUCTS = 6                 # bf16 MFMA products issued per fp32 product in "split" precision

WINO2D_MIN = 512      # the library's "winograd_2d" default (ops.WINOGRAD_2D_DEFAULT); --wino2d overrides both


ROWS16 = True         # the library's "split3x3_rows16" option
SPLIT_SIX = True      # six bf16 products per fp32 product (False in --precision bf16); part of the split kernels' template arguments


def family(key):
    """The kernel instantiation a conv launch runs (mirrors the dispatch in csrc/conv2d.hip / conv_wino.hip /
    conv_split_layer.hip), i.e. ONE kernel name in a rocprofv3 trace: launches of one family differ only in Cin/H/W/B."""
    ks, cin, cout, H, W, B, tag = key[:7]
    if ks == "L":
        return "split_layer_kernel" if tag.endswith("+split") else "wino_layer_kernel"
    if tag.endswith("+split") and ks == 3:
        # exactly the rocprofv3 name: conv3x3_split_kernel<MPW, SIX, ADD, ACT1, KS, RPW> (csrc/conv_split3x3.hip: launch_epi)
        pro, act, res, act2, _ = tag.split("|")
        plain = not res and not act2
        act1 = 2 if (plain and act == "prelu") else 0 if (plain and not act and not key[7]) else -1
        if tag.endswith("couple+split"):
            act1 = -2
        # (m-tiles per wave, channel groups per block): csrc/conv_split3x3.hip mpw_of / wm_of
        mpw = 4 if cout > 128 else 2 if cout > 96 else 3 if cout > 64 else 1 if cout > 48 else 3 if cout > 32 else 2 if cout > 16 else 1
        wm = 4 if cout > 96 else 2 if cout > 64 else 4 if cout > 48 else 1
        rpw = 4
        if wm == 4 and mpw == 1 and ROWS16 and H > 8 and not pro and not key[7]:   # 64-channel tiling without a load-side prologue: 16-row tiles,
            rpw, act1 = 8, (act1 if act1 in (0, -2) else -1)                        # bias-only / coupling epilogue compiled in, anything else at run time
            if act1 != -2:
                mpw, wm = 2, 2                                                     # two m-tiles per wave x two channel groups (not the coupling form)
        if wm != 4:
            rpw = 8
        if wm != 4 and not 48 < cout <= 64:
            act1 = 0 if (plain and not act) else 2 if (plain and act == "prelu") else -1
        return "conv3x3_split_kernel<%d, %s, %s, %d, 3, %d, %d>" % (mpw, "true" if SPLIT_SIX else "false", "true" if key[7] else "false", act1, rpw, wm)
    if tag.endswith("+split") and ks == 7:
        return "conv3x3_split_kernel<2, %s, false, 0, 7, 4, 2>" % ("true" if SPLIT_SIX else "false")
    if tag.endswith("+split"):
        return "conv%dx%d_split_kernel[%s]" % (ks, ks, tag)
    if ks == 3 and cout > 64 and WINO2D_MIN and cout >= WINO2D_MIN:
        return "conv3x3_wino2d_kernel[%s]" % tag
    if ks == 3 and cout >= 33:
        return "conv3x3_wino_kernel<%s>[%s]" % ("W64" if cout <= 64 else "W128", tag)
    return "conv2d_mfma_kernel<k%d,%s>[%s]" % (ks, "co<=32" if cout <= 32 else "co<=64" if cout <= 64 else "co>64", tag)


# kernel family (see family()) -> substrings of the rocprofv3 kernel names it covers (tools/pmc_traffic.py averages FETCH_SIZE /
# WRITE_SIZE over the launches of all of them; the layer kernel has one instantiation per map layout)
ROCPROF_NAMES = {
    "conv3x3_split_kernel<4, true, false, 2, 3, 4, 4>": ["conv3x3_split_kernel<4, true, false, 2, 3, 4, 4>"],
    "conv3x3_split_kernel<4, true, true, 2, 3, 4, 4>": ["conv3x3_split_kernel<4, true, true, 2, 3, 4, 4>"],
    "conv3x3_split_kernel<4, false, false, 2, 3, 4, 4>": ["conv3x3_split_kernel<4, false, false, 2, 3, 4, 4>"],
    "split_layer_kernel": ["split_layer_kernel<"],
    "wino_layer_kernel": ["wino_layer_kernel<false>"],
    "conv3x3_wino2d_kernel[|prelu|||]": ["conv3x3_wino2d_kernel<2, false, true>"],
    "conv3x3_wino_kernel<W128>[pro|prelu|||]": ["conv3x3_wino_kernel<WCfg<8, 2, 2, 4>, 3, true>"],
    "conv3x3_wino_kernel<W128>[|prelu|||]": ["conv3x3_wino_kernel<WCfg<8, 2, 2, 4>, 3, false>"],
}


def issued_factor(fam):
    """Matrix-core products actually issued per algorithmic multiply-add of a kernel family, and the pipe they run on.
    fp32 MFMA kernels: Winograd issues fewer (2/3 for F(2,3), 4/9 for F(2x2,3x3), 0.7 for the fused layer = nine F(2,3)
    taps + one direct 1x1 tap of ten).  Split kernels: six bf16 products (one in plain-bf16 mode) per fp32 product."""
    if "split" in fam:
        return float(SPLIT_PRODUCTS), "bf16"
    if fam == "wino_layer_kernel":
        return 0.7, "f32"
    if "wino2d" in fam:
        return 4.0 / 9.0, "f32"
    if "wino" in fam:
        return 2.0 / 3.0, "f32"
    return 1.0, "f32"


class ConvEvents:
    """Event sink for ops.conv2d / ops.subnet_layer: everything (selection pass) or the launches of some kernel families."""

    def __init__(self, only=None):
        self.only, self.rows = only, []

    def want(self, key):
        return self.only is None or family(key) in self.only

    def add(self, key, e0, e1):
        self.rows.append((key, e0, e1))

    def totals(self):
        """family -> [ms, launches, flops, {shape: launches}]"""
        tot = {}
        for key, e0, e1 in self.rows:
            t = tot.setdefault(family(key), [0.0, 0, 0.0, {}, {}])
            ms = e0.elapsed_time(e1)
            t[0] += ms
            t[1] += 1
            t[2] += conv_flops(key)
            t[3][key] = t[3].get(key, 0) + 1
            t[4][key] = t[4].get(key, 0.0) + ms            # per-shape time (roofline_of: per_shape)
        return tot


def conv_flops(key):
    ks, cin, cout, H, W, B = key[:6]
    if ks == "L" and key[6].startswith("layer1x"):  # ... with the first map as a third k step: + 64 x 32 (cin = 16: the short form)
        return 2.0 * cout * (cin * 9 + cout + 32) * H * W * B
    if ks == "L" and key[6].startswith("layer1"):   # first layer in composed form: 3x3 over the (padded) 32 input channels + the 64 x 64 1x1
        return 2.0 * cout * (cin * 9 + cout) * H * W * B
    taps = 10 if ks == "L" else ks * ks          # "L": fused 3x3 + 1x1 sub-network layer (9 + 1 taps)
    return 2.0 * cout * cin * taps * H * W * B


def conv_bytes(key):
    """Algorithmic HBM bytes of one launch: every input tensor (x, and the skip tensor of a load-side add) read once, the
    output written once; the fused layer's residual is the x tile it already holds.  Weights are L2-resident noise."""
    ks, cin, cout, H, W, B = key[:6]
    if ks == "L" and key[6].startswith("layer1x"):  # u (+ ones channel) is read, y is written
        return 4.0 * (cin + cout) * H * W * B
    if ks == "L" and key[6].startswith("layer1"):   # u (+ ones channel) and the residual map x are read, y is written
        return 4.0 * (cin + 2 * cout) * H * W * B
    return 4.0 * (cin * (2 if key[7] else 1) + cout) * H * W * B


def roofline_of(fam, t_ms, n, flops, shapes, products, share=None, all_conv_ms=None, shape_ms=None):
    """Roofline object of one conv kernel family.  `algorithmic_tflops` = direct-convolution FLOPs (2*Cout*Cin*taps*H*W
    per launch) / summed HIP-event time; `achieved` = the matrix-core FLOPs the kernel ISSUES for
    them (algorithmic x issued factor) / the same time, against the dense peak of the pipe it runs on: `frac` <= 1."""
    factor, pipe = issued_factor(fam)
    if pipe == "bf16":
        factor = float(products)
    peak = PEAK_BF16_MFMA_TFLOPS if pipe == "bf16" else PEAK_FP32_MFMA_TFLOPS
    alg = flops / (t_ms * 1e-3) / 1e12
    inst = ("v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16, %d bf16 products per fp32 product" % products) if pipe == "bf16" else \
        "v_mfma_f32_32x32x2_f32" + (", Winograd: %.2fx the algorithmic MFMA count is issued" % factor if factor != 1.0 else "")
    r = {"bound": "mfma", "achieved": alg * factor, "peak": peak, "unit": "TFLOP/s", "frac": alg * factor / peak, "traffic": None,
         "kernel": "%s (%s)" % (fam, inst), "algorithmic_tflops": alg, "issued_per_algorithmic": factor,
         "shapes": [dict(zip(("ks", "cin", "cout", "H", "W", "B", "launches"), (*k[:6], c)))
                    for k, c in sorted(shapes.items(), key=lambda kv: -kv[1])][:6],
         "flops_per_launch": flops / n, "algorithmic_bytes_per_launch": sum(conv_bytes(k) * c for k, c in shapes.items()) / n,
         "avg_launch_ms": t_ms / n, "launches_timed": n}
    if shape_ms:                             # the family's shapes one by one (a family mixes e.g. full layers and composed first layers)
        r["per_shape"] = [{"cin": k[1], "cout": k[2], "H": k[3], "form": k[6], "launches": shapes[k], "avg_launch_ms": shape_ms[k] / shapes[k],
                           "frac": conv_flops(k) * shapes[k] / (shape_ms[k] * 1e-3) / 1e12 * factor / peak}
                          for k in sorted(shape_ms, key=lambda kk: -shape_ms[kk])][:6]
    if share is not None:
        r["share_of_conv_time"], r["all_conv_ms_per_step"] = share, all_conv_ms
    return r


def launch_command(n, argv, port=None):
    """The command `python bench.py --gpus N ...` turns itself into when it is started without a launcher: the same launch the
    driver uses (one process per GPU, rendezvous on 127.0.0.1 -- the container hostname may not resolve)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_launch(n, argv):
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(launch_command(n, argv), env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="volumes per step per GPU")
    ap.add_argument("--side", type=int, default=512)
    ap.add_argument("--depths", type=int, default=96)
    ap.add_argument("--precision", choices=("split", "fp32", "bf16"), default="split",
                    help="arithmetic of the timed region: split = fp32-equivalent (exact three-way bf16 split of both operands, "
                         "six products, fp32 accumulation) on the bf16 matrix cores [default]; fp32 = plain fp32 MFMA kernels; "
                         "bf16 = BASELINE.json configs[4] (bf16 conv operands, restated tolerance; NOT the headline metric)")
    ap.add_argument("--block-type", default="CAT", choices=("CAT", "GLOW", "AI1", "RNVP", "GIN"),
                    help="coupling block of the flow steps (main.py --INN_block_type; CAT is the reference's default and the "
                         "headline configuration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="the complete SURVEY 8d protocol on the full volume "
                    "(warm-up + median of 3, all cores and 8 threads): several minutes")
    ap.add_argument("--no-lrnn", action="store_true", help="(diagnostic) flows + condition nets only")
    ap.add_argument("--no-experiment", action="store_true", help="skip the extra measurements after the timed region")
    ap.add_argument("--split3x3-min", type=int, default=None, help="(tuning) smallest output-channel count of a 3x3 bank that takes "
                    "the split-bf16 kernel in split / bf16 precision (library default: ops.SPLIT_3X3_MIN_COUT)")
    ap.add_argument("--no-couple-epilogue", action="store_true", help="(ablation, block types other than CAT) sub-networks write "
                    "[s | t] and a separate affine launch applies the coupling")
    ap.add_argument("--no-virtual-cat", action="store_true", help="(ablation, block types other than CAT) materialise the input "
                    "cat(half, condition) of every coupling sub-network instead of reading it from its two tensors")
    ap.add_argument("--no-split7x7", action="store_true", help="(ablation) the ConvNeXt 7x7 convolution on the fp32 MFMA kernel")
    ap.add_argument("--no-rows16", action="store_true", help="(ablation) 8-row tiles for the 64-channel tiling of the split 3x3 kernel")
    ap.add_argument("--no-xcd-map", action="store_true", help="(ablation) plain block order in the split 3x3 kernel")
    ap.add_argument("--no-first-composed", action="store_true", help="(ablation) the first layer of every sub-network like the other two "
                    "(K = 9 x 64) instead of its composed form")
    ap.add_argument("--no-fused-first-map", action="store_true", help="(ablation) the composed first layer reads its residual conv1x1(u) + b0 "
                    "from a map written by a 1x1 launch instead of forming it itself")
    ap.add_argument("--no-merge-first", action="store_true", help="(ablation) every sub-network runs its own first 1x1 convolution")
    ap.add_argument("--no-merge-omega", action="store_true", help="(ablation) every condition net runs its own conv1 / downsample launches")
    ap.add_argument("--wino2d", type=int, default=None, help="(tuning) override the 2-D Winograd output-channel threshold (0 = off)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: become the launcher (one rank per GPU over RCCL) BEFORE anything touches the GPU in
        # this process, forward the ranks' output (rank 0 prints the JSON line) and exit with their status
        raise SystemExit(self_launch(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} under a launcher with WORLD_SIZE={world}: the two must agree")
    import torch.distributed as dist
    # rehearsal switch (not used by the driver): several ranks on ONE card over gloo, to exercise the N > 1 control flow
    # on a single-GPU box (RCCL refuses two ranks on one device)
    rehearse = os.environ.get("CWFA_BENCH_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm

    from cwfa_amd import CWFA, ops
    if a.wino2d is not None:
        global WINO2D_MIN
        WINO2D_MIN = a.wino2d
        ops.set_option("winograd_2d", a.wino2d)
    if a.no_first_composed:
        ops.FIRST_LAYER_COMPOSED = False
    if a.no_fused_first_map:
        ops.FIRST_LAYER_FUSED_X = False
    if a.no_merge_first:
        from cwfa_amd import networks as _N
        _N.MERGE_FIRST_MAPS = False
    if a.no_merge_omega:
        from cwfa_amd import networks as _N
        _N.MERGE_OMEGA_FIRST = False
    if a.no_couple_epilogue:
        ops.COUPLE_EPILOGUE = False
    if a.no_virtual_cat:
        ops.VIRTUAL_CAT = False
    if a.no_split7x7:
        ops.SPLIT_7X7 = False
    if a.no_xcd_map:
        ops.set_option("split3x3_xcd_map", 0)
    if a.no_rows16:
        global ROWS16
        ROWS16 = False
        ops.set_option("split3x3_rows16", 0)
    if a.split3x3_min is not None:
        ops.SPLIT_3X3_MIN_COUT = a.split3x3_min
    PREC = {"split": "split_bf16", "fp32": "fp32", "bf16": "bf16"}
    ops.set_precision(PREC[a.precision])
    products = {"split": SPLIT_PRODUCTS, "bf16": 1, "fp32": SPLIT_PRODUCTS}[a.precision]
    global SPLIT_SIX
    SPLIT_SIX = a.precision != "bf16"
    S = 5                                                    # INN_max_down_steps (main.py:106): 4 flow steps + LRNN
    torch.manual_seed(0)
    np.random.seed(0)
    conv_inn, cond_nets = CWFA.build_networks(a.depths, a.side, S, block_type=a.block_type, with_lrnn=not a.no_lrnn, device=dev)
    g = torch.Generator().manual_seed(1 + rank)
    B = a.batch
    cond_input = torch.randn(B, 29, a.side, a.side, generator=g).to(dev)
    mean_cache = [(0.1 * torch.randn(B, a.depths // 2 ** (n + 1), a.side, a.side, generator=g)).to(dev)
                  for n in range(S - 1)]
    low = torch.randn(B, a.depths // 2 ** (S - 1), a.side, a.side, generator=g).to(dev) if a.no_lrnn else None

    def step():
        with torch.no_grad():
            return CWFA.inverse_pass(conv_inn, cond_nets, cond_input, mean_cache, low=low)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps):
        fn(); fn()                                           # re-pack affected filter banks, warm up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    out = step()                                             # cold pass: weight packing, allocator
    assert out.shape == (B, a.depths, a.side, a.side) and bool(torch.isfinite(out).all())
    # selection pass: which conv kernels dominate?
    sel = ops.conv_event_sink = ConvEvents()
    step()
    torch.cuda.synchronize()
    tot = sel.totals()
    all_conv_ms = sum(t[0] for t in tot.values())
    ranked = sorted(tot, key=lambda k: -tot[k][0])
    dom = ranked[0]                                          # the kernel (one instantiation) with the largest time share
    ops.conv_event_sink = None
    for _ in range(max(a.warmup - 2, 0)):
        step()

    sink = ops.conv_event_sink = ConvEvents(only={dom})      # only the dominant kernel carries events in the timed region
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    ops.conv_event_sink = None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)

    # BASELINE.json configs[3] / the metric's second half: forward NLL, batch-sharded, one all-reduce of the NLL sums
    fwd = None
    if not a.no_experiment and not a.no_lrnn:
        try:
            fwd = forward_nll_leg(CWFA, ops, conv_inn, cond_nets, a, dev, rank, world, sync_all, products)
        except Exception as exc:                    # noqa: BLE001  (after the timed region; must not cost the line)
            fwd = {"error": repr(exc)[:300]}

    res = None
    if rank == 0:
        t_dom, n_dom, f_dom, shapes, shape_ms = sink.totals()[dom]
        baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
        dtype = {"split": "f32 (storage, accumulation, wavelets, couplings); convolution products on the bf16 matrix cores from an "
                          "exact three-way bf16 split of both fp32 operands, six products per fp32 product: fp32-equivalent; "
                          "plain-fp32-MFMA figure: fp32_mfma",
                 "fp32": "f32 (v_mfma_f32_32x32x2_f32 / Winograd)",
                 "bf16": "bf16 conv operands, f32 accumulate, f32 wavelets / couplings (BASELINE.json configs[4]; tolerance restated)"}[a.precision]
        res = {
            "metric": baseline["metric"], "value": world * a.steps * B / elapsed,
            "unit": "volumes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            # what `value` is, for a reader of `metric` / `value` alone (the metric string is BASELINE.json's and says "fp32")
            "value_label": {"split": "fp32-equivalent (bf16 x 3 split, six products per fp32 product, fp32 accumulate); plain fp32 MFMA: fp32_mfma",
                            "fp32": "plain fp32 MFMA (v_mfma_f32_32x32x2_f32 / Winograd)",
                            "bf16": "bf16 conv operands (BASELINE.json configs[4]), NOT the fp32 headline"}[a.precision],
            "multi_gpu": ("this line is N = 1; the 1/2/4/8-GPU half of the metric and the RCCL NLL all-reduce are UNMEASURED on hardware "
                          "so far (builders cannot launch multi-GPU runs; `python bench.py --gpus N` starts its own ranks)") if world == 1 else
                         f"{world} ranks, one per GPU, {'gloo on ONE card (rehearsal)' if rehearse else 'RCCL'}; see forward_nll.nll_check",
            "config": {"workload": f"{a.side}x{a.side}x{a.depths} volume, 4-scale CWFA ({a.block_type} blocks, 64 ch) + "
                                   f"{'LRNN' if not a.no_lrnn else 'synthetic low-res (NO LRNN: diagnostic)'} inverse, z=0, "
                                   f"batch {B}/GPU, random-init weights (BASELINE.json configs[2])",
                       "precision": a.precision,
                       "parallelism": f"replicated x{world} (independent volumes per GPU, no collective)"},
            "roofline": roofline_of(dom, t_dom, n_dom, f_dom, shapes, products, tot[dom][0] / all_conv_ms, all_conv_ms, shape_ms),
            "reference_readme": {"volumes_per_s": 6.25, "note": "README.md:29, unstated CUDA GPU, fp16 autocast; not "
                                 "this fp32 metric, hence vs_baseline is null"},
        }
        res["roofline"]["traffic"], res["roofline"]["traffic_detail"] = pmc_traffic(dom, a)
        if len(ranked) > 1:                                        # the runner-up kernel, same definitions, timed in three
            sink2 = ops.conv_event_sink = ConvEvents(only={ranked[1]})   # extra steps AFTER the timed region (its many small
            for _ in range(3):                                     # launches would otherwise put event gaps into `value`)
                step()
            torch.cuda.synchronize()
            ops.conv_event_sink = None
            t2, n2, f2, sh2, sm2 = sink2.totals()[ranked[1]]
            res["roofline_second"] = roofline_of(ranked[1], t2, n2, f2, sh2, products, tot[ranked[1]][0] / all_conv_ms, shape_ms=sm2)
            res["roofline_second"]["traffic"], _ = pmc_traffic(ranked[1], a)
        res["roofline_dwt"] = dwt_roofline(ops, step, a, dev)
        if fwd is not None:
            res["forward_nll"] = fwd
        if world == 1 and not a.no_lrnn and not a.no_experiment:
            for name, mode in (("fp32_mfma", "fp32"), ("bf16", "bf16"), ("split", "split")):
                if mode == a.precision:
                    continue
                try:                                          # never let a side measurement cost the headline line
                    ops.set_precision(PREC[mode])
                    dt = timed(step, max(a.steps // 2, 3))
                    roof = None
                    if mode == "bf16":                        # configs[4] carries its own roofline object: dominant kernel of THIS mode
                        roof = mode_roofline(ops, step, "bf16")
                    res[name] = {"value": B / dt, "unit": "volumes/s", "ms_per_step": 1e3 * dt, "steps": max(a.steps // 2, 3),
                                 "note": {"fp32": "the same step on the plain fp32 MFMA kernels (Winograd F(2,3) / F(2x2,3x3)), after the timed region",
                                          "bf16": "BASELINE.json configs[4]: bf16 conv operands, fp32 accumulation; parity max-rel <= 1e-2, "
                                                  "L2-rel <= 5e-3 vs the fp32 oracle (tests/test_gpu_parity.py::test_full_config3_inverse_vs_oracle)",
                                          "split": "fp32-equivalent split-bf16 arithmetic (see --precision)"}[mode]}
                    if roof is not None:
                        res[name]["roofline"] = roof
                except Exception as exc:                      # noqa: BLE001
                    res[name] = {"error": repr(exc)[:300]}
                finally:
                    ops.set_precision(PREC[a.precision])
            try:
                res["experiment_train_step"] = train_experiment(conv_inn, cond_nets, dev, a, max(a.steps // 4, 3), ops)
            except Exception as exc:                          # noqa: BLE001
                res["experiment_train_step"] = {"error": repr(exc)[:300]}
            finally:
                ops.set_precision(PREC[a.precision])
        if world == 1 and not a.no_cpu_baseline and not a.no_lrnn and a.block_type == "CAT":
            try:
                res["cpu_baseline"] = cpu_baseline(conv_inn, cond_nets, cond_input, mean_cache, a.cpu_baseline_full)
            except Exception as exc:                          # noqa: BLE001  (host trouble must not lose the GPU measurement)
                res["cpu_baseline"] = {"value": None, "unit": "volumes/s", "cores": 0, "kind": "port", "sample": "failed: " + repr(exc)[:200]}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return res


def mode_roofline(ops, step, mode):
    """Roofline object of the dominant conv kernel of another precision mode (ops.set_precision already done): one selection
    pass with events on every conv launch, then three passes with events on the dominant family only.  In bf16 mode the split
    kernels run their one-product instantiations (SIX = false): issued = algorithmic FLOPs, peak = the bf16 dense peak."""
    global SPLIT_SIX
    import torch
    six, SPLIT_SIX = SPLIT_SIX, mode != "bf16"
    try:
        sel = ops.conv_event_sink = ConvEvents()
        step()
        torch.cuda.synchronize()
        tot = sel.totals()
        all_ms = sum(t[0] for t in tot.values())
        dom = max(tot, key=lambda k: tot[k][0])
        sink = ops.conv_event_sink = ConvEvents(only={dom})
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        ops.conv_event_sink = None
        t, n, f, sh, sm = sink.totals()[dom]
        return roofline_of(dom, t, n, f, sh, 1 if mode == "bf16" else SPLIT_PRODUCTS, tot[dom][0] / all_ms, all_ms, sm)
    finally:
        ops.conv_event_sink = None
        SPLIT_SIX = six


def forward_nll_leg(CWFA, ops, conv_inn, cond_nets, a, dev, rank, world, sync_all, products, per_gpu=4, steps=5):
    """BASELINE.json configs[3]: forward NLL of a batch of 512x512x96 volumes over the four flow steps with their condition
    nets (CWFA.py:966-978), batch sharded over the ranks (4 volumes per GPU: 32 over 8 GPUs), ONE all-reduce of the
    float64[12] NLL sums per pass (RCCL over xGMI).  After the headline timed region; same barrier / max-over-ranks rule.
    At N > 1 `nll_check` compares the sharded NLL with the same NLL computed by one rank alone over the first two shards."""
    import torch.distributed as dist
    D, Sd = a.depths, a.side
    nst = len(conv_inn)

    def shard(r):
        gen = torch.Generator().manual_seed(3 + 7 * r)
        x = torch.randn(per_gpu, D, Sd, Sd, generator=gen).to(dev)
        views = torch.randn(per_gpu, 29, Sd, Sd, generator=gen).to(dev)
        means = [(0.1 * torch.randn(per_gpu, D // 2 ** (n + 1), Sd, Sd, generator=gen)).to(dev) for n in range(nst)]
        return x, views, means

    x, views, means = shard(rank)

    def one():
        with torch.no_grad():
            return CWFA.forward_nll_pass(conv_inn, cond_nets[:nst], x, views, means)

    nll, _ = one()
    sel = ops.conv_event_sink = ConvEvents()
    chain = ops.chain_event_sink = []
    one()
    torch.cuda.synchronize()
    ops.conv_event_sink = ops.chain_event_sink = None
    tot = sel.totals()
    dom = max(tot, key=lambda k: tot[k][0])
    chain_ms = sum(e0.elapsed_time(e1) for *_, e0, e1 in chain)
    chain_bytes = sum(4.0 * (4 + 2 * ns) * Bc * C * H * W for _, Bc, C, H, W, ns, *_ in chain)   # x (2C) + s,t of the stages + z, low (2C)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        nll, _ = one()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    out = {"value": world * per_gpu * steps / dt, "unit": "volumes/s", "batch_per_gpu": per_gpu, "global_batch": world * per_gpu,
           "ms_per_step": 1e3 * dt / steps, "steps": steps, "scaling": "weak",
           "workload": f"forward NLL, {Sd}x{Sd}x{D} volumes, four flow steps + condition nets (no LRNN), BASELINE.json configs[3]",
           "collective": "one all_reduce(SUM) of float64[%d] per pass (nccl = RCCL)" % (3 * nst) if world > 1 else "none (N = 1)",
           "nll_per_step": [float(v) for v in nll],
           "roofline": roofline_of(dom, tot[dom][0], tot[dom][1], tot[dom][2], tot[dom][3], products,
                                   tot[dom][0] / sum(t[0] for t in tot.values())),
           "chain_fwd": {"bound": "hbm", "achieved": chain_bytes / (chain_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": chain_bytes / (chain_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "launches": len(chain), "ms": chain_ms}}
    if world > 1:
        # the same NLL by ONE process over the volumes of ranks 0 and 1 (no collective) vs the sharded value of a 2-rank
        # sub-problem is not available here (the all-reduce spans all ranks), so: every rank recomputes the global sums
        # from all shards sequentially (world x the work, after the timed region) and compares
        with torch.no_grad():
            rows = torch.zeros(nst, 3, dtype=torch.float64, device=dev)
            for r in range(world):
                xr, vr, mr = shard(r)
                gt = xr
                for n, gph in enumerate(conv_inn):
                    Z, logdet, sumsq = CWFA.nll_terms(gph, gt, [cond_nets[n](vr)[-1], mr[n]])
                    rows[n] += torch.stack([sumsq[0], logdet.double().sum(), torch.tensor(float(per_gpu), dtype=torch.float64, device=dev)])
                    gt = Z[1]
            numel = torch.tensor([float(D * Sd * Sd) / 2 ** n for n in range(nst)], dtype=torch.float64, device=dev)
            ref = (0.5 * rows[:, 0] - rows[:, 1] / rows[:, 2]) / (rows[:, 2] * numel)
        rel = float(((nll - ref).abs() / ref.abs()).max())
        lo, hi = nll.clone(), nll.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        out["nll_check"] = {"nll_sharded": [float(v) for v in nll], "nll_single_process": [float(v) for v in ref], "max_rel_diff": rel,
                            "identical_on_all_ranks": bool((lo == hi).all()), "volumes": world * per_gpu, "tolerance": 1e-6}
    return out


def train_experiment(conv_inn, cond_nets, dev, a, steps, ops):
    """NOT the headline: SURVEY.md 8(f) row 1 -- one training iteration over the WHOLE pyramid on one synthetic volume, in
    the reference's order (CWFA.py:865-1027): LRNN step (L2), then the four flow steps with their condition nets (inverse +
    forward + backward of 0.40984 * mse + 0.59016 * NLL); gradients computed, no optimiser update.  Split precision: forward
    (sub-network layers in their tape form), data-gradient convolutions and the 3x3 weight gradients on the bf16 matrix cores in the
    fp32-equivalent split arithmetic (csrc/conv_bwd.hip: conv_wgrad_split_kernel); 1x1 / 1x7 weight gradients and the Conv3d backward on
    the fp32 kernels; `fp32_kernels_ms_per_step` = the same iteration with every convolution on the fp32 kernels."""
    from cwfa_amd import training
    ops.set_precision("split_bf16")
    B, D, S = 1, a.depths, a.side
    gen = torch.Generator().manual_seed(17)
    gt = torch.randn(B, D, S, S, generator=gen).to(dev)
    views = torch.randn(B, 29, S, S, generator=gen).to(dev)
    means = [(0.1 * torch.randn(B, D // 2 ** (n + 1), S, S, generator=gen)).to(dev) for n in range(len(conv_inn))]

    def one():
        return training.train_iteration(conv_inn, cond_nets, gt, views, means)

    one(); one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n_par = sum(p.numel() for m in list(conv_inn) + list(cond_nets) for p in m.parameters() if p.requires_grad)
    ops.set_precision("fp32")
    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    dt32 = (time.perf_counter() - t0) / steps
    # ... and the same iteration as the reference writes it (modules as autograd nodes, loss with torch operators, full_loss.backward():
    # what `cwfa_amd.install()` gives an unmodified CWFA.py), split precision
    ops.set_precision("split_bf16")
    auto = lambda: training.train_iteration_autograd(conv_inn, cond_nets, gt, views, means)     # noqa: E731
    auto(); auto()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        auto()
    torch.cuda.synchronize()
    dta = (time.perf_counter() - t0) / steps
    for m in list(conv_inn) + list(cond_nets):
        for p_ in m.parameters():
            p_.grad = None
    return {"value": B / dt, "unit": "volumes/s", "ms_per_step": 1e3 * dt, "steps": steps, "fp32_kernels_ms_per_step": 1e3 * dt32,
            "through_autograd_ms_per_step": 1e3 * dta,
            "full_loss_per_pyramid_step": [float(v) for v in res["losses"]], "trainable_parameters": n_par,
            "note": "LRNN + 4 flow steps + condition nets, forward with tape + backward; forward, data gradients and 3x3 weight gradients in "
                    "split-bf16 arithmetic (fp32-equivalent), 1x1 / 1x7 weight gradients and the Conv3d backward on the fp32 kernels; gradients pinned to the "
                    "reference's autograd by tests/test_gpu_backward.py (fixtures g13-g15)"}


def pmc_traffic(dom, a):
    """HBM bytes per launch of a kernel.  FETCH_SIZE / WRITE_SIZE cannot be collected from inside this process
    (rocprofv3 --pmc, separate passes); profiles/*_pmc_traffic.json holds the latest such passes over THIS command, averaged
    over the same launches as `achieved`, with the calibration described in DESIGN.md section 6.  (None, None) if the
    committed record does not cover this kernel / workload."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files or (a.side, a.depths, a.batch) != (512, 96, 1):
        return None, None
    rec = json.load(open(files[-1]))
    hit = rec.get("kernels", {}).get(dom)
    if not hit:
        return None, None
    return hit["hbm_bytes_per_launch"], {"source": os.path.basename(files[-1]), **hit}


def dwt_roofline(ops, step, a, dev, reps=20):
    """The DWT stage as it runs IN the path: the inverse depth Haar lives in the fused chain kernel (one launch per flow step:
    5 gathers + 5 affines + Split/cat + Haar1D^-1).  Algorithmic bytes per launch = (low C + s,t of five blocks 10 C + output
    2 C) x H x W x 4 B = 13 C HW 4 (z = 0 is never read); HIP events on the launch stream around every chain launch of three
    steps.  `standalone_haar`: the four inverse Haar levels of one volume as separate haar1d kernels (377.5 MB algorithmic
    at 512x512x96, SURVEY.md 8d: 8 bytes per element) -- what the north_star's 60 % target was first measured on."""
    chain = ops.chain_event_sink = []
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ops.chain_event_sink = None
    # algorithmic planes of C x H x W floats per launch: low band + (z if it is read) + s, t of every stage + the 2C output
    rows = [(kind, B, C, H, W, e0.elapsed_time(e1), 1 + int(zr) + 2 * ns + 2) for kind, B, C, H, W, ns, zr, e0, e1 in chain]
    ms = sum(r[5] for r in rows)
    nbytes = sum(4.0 * pl * B * C * H * W for _, B, C, H, W, _, pl in rows)
    big = [r for r in rows if r[2] == max(r2[2] for r2 in rows)]
    gbs = nbytes / (ms * 1e-3) / 1e9
    big_bytes = 4.0 * big[0][6] * big[0][1] * big[0][2] * big[0][3] * big[0][4]
    out = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
           "kernel": "chain_rows4_kernel<true> (in-path: inverse Haar1D + Split/cat + %d x (gather, affine), one launch per flow step)"
                     % ((big[0][6] - 3) // 2),
           "bytes_per_volume": nbytes / 3 / a.batch, "us_per_volume": 1e3 * ms / 3 / a.batch, "launches_timed": len(rows),
           "largest_level": {"bytes_per_launch": big_bytes, "avg_launch_us": 1e3 * sum(r[5] for r in big) / len(big),
                             "GBps": big_bytes / (sum(r[5] for r in big) / len(big) * 1e-3) / 1e9}}
    levels = [a.depths // 2 ** n for n in range(4)]
    bufs = [torch.randn(1, d, a.side, a.side, device=dev) for d in levels]
    for b in bufs:
        ops.haar1d(b, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for b in bufs:
            ops.haar1d(b, True)
    e1.record()
    torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / reps
    nb2 = sum(8.0 * b.numel() for b in bufs)
    out["standalone_haar"] = {"kernel": "haar1d_inv_kernel<4> (NOT launched by the path)", "achieved": nb2 / (ms2 * 1e-3) / 1e9, "unit": "GB/s",
                              "frac": nb2 / (ms2 * 1e-3) / 1e9 / PEAK_HBM_GBS, "bytes_per_volume": nb2, "us_per_volume": 1e3 * ms2}
    # the one-pass 2 x 2 x 2 Haar tile (cwfa_haar3d_*: depth Haar + spatial Haar of every band; API surface, the path itself
    # transforms along depth only): forward + inverse of one 512x512x96 volume, 8 bytes per element and direction
    vol = torch.randn(1, a.depths, a.side, a.side, device=dev)
    coef = ops.haar3d(vol)
    ops.haar3d(coef, True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        coef = ops.haar3d(vol)
        ops.haar3d(coef, True)
    e1.record()
    torch.cuda.synchronize()
    ms3 = e0.elapsed_time(e1) / reps
    nb3 = 2 * 8.0 * vol.numel()
    out["haar3d_tile"] = {"kernel": "haar3d_fwd_kernel<2> + haar3d_inv_kernel<2> (NOT launched by the path)", "achieved": nb3 / (ms3 * 1e-3) / 1e9,
                          "unit": "GB/s", "frac": nb3 / (ms3 * 1e-3) / 1e9 / PEAK_HBM_GBS, "bytes": nb3, "us": 1e3 * ms3}
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(conv_inn, cond_nets, cond_input, mean_cache, full=False):
    """The CPU oracle (oracle/cwfa_oracle.py, a port of the reference's op sequence onto torch CPU ops) on the same weights and
    inputs, timed per stage (LRNN, then per flow step its condition net and its inverse) on one full 512x512x96 volume.
    Default (bounded: ~20 s of CPU work): one warm-up of the coarsest flow step (threads, oneDNN primitives), then ONE timed
    full volume with torch threads = all host cores.  (The sample cannot be a spatial crop: the row / column permutations of
    the flow steps are tables of length 512.)
    `--cpu-baseline-full`: the SURVEY.md 8d protocol -- one full warm-up, median of three full volumes, with torch threads =
    all host cores AND = 8 (the reference's default, main.py:75); several minutes.  Its latest result on the GPU box is kept
    in profiles/*_cpu_baseline_full.json and attached to the default line as `protocol_full`."""
    from oracle import cwfa_oracle as O
    ncores = min(os.cpu_count() or 1, 64)
    cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}   # noqa: E731
    steps = []
    for n, g in enumerate(conv_inn):
        axes = {i: (m.axis if hasattr(m, "axis") else 1) for i, m in enumerate(g.module_list) if hasattr(m, "perm")}
        steps.append({"inn": cpu(g.state_dict()), "omega": cpu(cond_nets[n].state_dict()), "axes": axes})
    lrnn_sd = cpu(cond_nets[-1].state_dict())
    ci = cond_input[:1].cpu()
    mc = [m[:1].cpu() for m in mean_cache]

    def flow(n, up):
        t0 = time.perf_counter()
        om = O.omega_net(steps[n]["omega"], ci)
        t1 = time.perf_counter()
        up, _ = O.flow_step(steps[n]["inn"], (torch.zeros_like(up), up), [om, mc[n]], True, steps[n]["axes"])
        return up, t1 - t0, time.perf_counter() - t1

    def run_once():
        t = {}
        with torch.no_grad():
            t0 = time.perf_counter()
            up = O.lrnn(lrnn_sd, ci, mc[-1], train=True)
            t["lrnn"] = time.perf_counter() - t0
            for n in range(len(steps) - 1, -1, -1):
                up, t[f"omega{n}"], t[f"flow{n}"] = flow(n, up)
        t["total"] = sum(t.values())
        return t

    def protocol(threads, runs, warm_full):
        torch.set_num_threads(threads)
        with torch.no_grad():
            if warm_full:
                run_once()
            else:
                flow(len(steps) - 1, torch.zeros(1, mc[-1].shape[1], ci.shape[2], ci.shape[3]))
        ts = [run_once() for _ in range(runs)]
        med = {k: float(np.median([t[k] for t in ts])) for k in ts[0]}
        return {"threads": torch.get_num_threads(), "runs": runs, "seconds_per_volume": med["total"], "value": 1.0 / med["total"],
                "stages_s": {k: round(v, 3) for k, v in med.items() if k != "total"}}

    alln = protocol(ncores, 3 if full else 1, full)
    out = {"value": alln["value"], "unit": "volumes/s", "cores": alln["threads"], "kind": "port", "cpu_model": cpu_model(),
           "sample": f"1 full volume (LRNN + 4 flow steps + condition nets), " +
                     (f"one full warm-up then median of {alln['runs']} runs" if full else "one timed run after a warm-up of the coarsest flow step") +
                     f", {alln['seconds_per_volume']:.1f} s per volume, torch {torch.__version__} CPU fp32",
           "stages_s": alln["stages_s"]}
    if full:
        out["threads_8"] = protocol(8, 3, True)
        torch.set_num_threads(ncores)
    else:
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_cpu_baseline_full.json")))
        if files:
            out["protocol_full"] = {"source": os.path.basename(files[-1]), **json.load(open(files[-1]))}
    return out


if __name__ == "__main__":
    main()
