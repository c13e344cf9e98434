#!/usr/bin/env python3
"""Headline benchmark: inverse-pass (reconstruction) throughput, volumes/s, 512x512x96 fp32 (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one full inverse pass over one batch of synthetic light-field views resident in HBM:
LRNN (train mode, as CWFA.py:532) -> for n = 3..0: condition net Omega_n, 5 coupling sub-networks, fused flow chain
(config 3 of BASELINE.json: 4-scale CWFA + LRNN, CAT blocks, main.py defaults).  Every rank reconstructs its own
independent volumes (weak scaling, no data-path collective); value = all volumes of all ranks / max-over-ranks time.

Extra objects on the JSON line: `roofline` (dominant kernel, HIP events on the launch stream inside the timed region),
`roofline_dwt` (the standalone Haar kernels of the DWT stage, the north_star's 60 %-of-HBM target), `cpu_baseline`
(the CPU oracle timed on this host, rank 0, N=1 only).
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

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0


def family(key):
    """The kernel instantiation a conv launch runs (mirrors the dispatch in csrc/conv2d.hip / conv_wino.hip), i.e. ONE
    kernel name in a rocprofv3 trace: launches of one family differ only in Cin / H / W / B."""
    ks, cin, cout, H, W, B, tag = key[:7]
    if ks == "L":
        return "wino_layer_kernel"
    if tag.endswith("+split"):
        return "conv%dx%d_split_kernel[%s]" % (ks, ks, tag)
    if ks == 3 and cout > 64 and WINO2D_MIN and cout >= WINO2D_MIN:
        return "conv3x3_wino2d_kernel[%s]" % tag
    if ks == 3 and cout >= 33:
        return "conv3x3_wino_kernel<%s>[%s]" % ("W64" if cout <= 64 else "W128", tag)
    return "conv2d_mfma_kernel<k%d,%s>[%s]" % (ks, "co<=32" if cout <= 32 else "co<=64" if cout <= 64 else "co>64", tag)


WINO2D_MIN = 512      # the library's "winograd_2d" default (ops.WINOGRAD_2D_DEFAULT); --wino2d overrides both


def issued_factor(fam):
    """MFMA work actually issued / algorithmic FLOPs of a kernel family (Winograd kernels issue fewer)."""
    if fam == "wino_layer_kernel":
        return 0.7                     # nine F(2,3) taps (x 2/3) + one direct 1x1 tap, of ten algorithmic
    if "wino2d" in fam:
        return 4.0 / 9.0               # F(2x2,3x3): 16 products per 4 outputs instead of 36
    if "wino" in fam:
        return 2.0 / 3.0               # F(2,3) along one axis
    return 1.0


# family -> substring of the kernel's name in a rocprofv3 trace (template arguments: config, epilogue id, prologue flag)
ROCPROF_NAMES = {
    "wino_layer_kernel": "wino_layer_kernel",
    "conv3x3_wino_kernel<W128>[pro|prelu|||]": "conv3x3_wino_kernel<WCfg<8, 2, 2, 4>, 3, true>",
    "conv3x3_wino_kernel<W128>[|prelu|||]": "conv3x3_wino_kernel<WCfg<8, 2, 2, 4>, 3, false>",
    "conv3x3_wino_kernel<W128>[|elu|||]": "conv3x3_wino_kernel<WCfg<8, 2, 2, 4>, 2, false>",
    "conv3x3_wino_kernel<W64>[|elu|||]": "conv3x3_wino_kernel<WCfg<8, 2, 1, 8>, 2, false>",
    "conv3x3_wino2d_kernel[|prelu|||]": "conv3x3_wino2d_kernel<2, false, true>",
}


class ConvEvents:
    """Event sink for ops.conv2d: everything (selection pass) or the launches of one kernel family (timed region)."""

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
            t = tot.setdefault(family(key), [0.0, 0, 0.0, {}])
            t[0] += e0.elapsed_time(e1)
            t[1] += 1
            t[2] += conv_flops(key)
            t[3][key] = t[3].get(key, 0) + 1
        return tot


def conv_flops(key):
    ks, cin, cout, H, W, B = key[:6]
    taps = 10 if ks == "L" else ks * ks          # "L": fused 3x3 + 1x1 sub-network layer (9 + 1 taps)
    return 2.0 * cout * cin * taps * H * W * B


def conv_bytes(key):
    """Algorithmic HBM bytes of one launch: every input tensor (x, and the skip tensor of a load-side add) read once, the
    output written once; the fused layer's residual is the x tile it already holds.  Weights are L2-resident noise."""
    ks, cin, cout, H, W, B = key[:6]
    return 4.0 * (cin * (2 if key[7] else 1) + cout) * H * W * B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="volumes per step per GPU")
    ap.add_argument("--side", type=int, default=512)
    ap.add_argument("--depths", type=int, default=96)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lrnn", action="store_true", help="(diagnostic) flows + condition nets only")
    ap.add_argument("--no-experiment", action="store_true", help="skip the extra split-bf16 measurement after the timed region")
    ap.add_argument("--wino2d", type=int, default=None, help="(tuning) override the 2-D Winograd output-channel threshold (0 = off)")
    ap.add_argument("--no-materialize-up", action="store_true", help="(tuning) the same for the UNet's transposed convolutions")
    ap.add_argument("--no-materialize", action="store_true", help="(tuning) UNet BatchNorm applied on load instead of materialised")
    ap.add_argument("--bf16", action="store_true", help="BASELINE.json configs[4] (NOT the headline configuration): bf16 operands in "
                    "the heavy convolutions, fp32 accumulation; the line's dtype says so")
    ap.add_argument("--split-bf16", type=int, default=0, choices=(0, 1, 2),
                    help="(experiment, NOT the headline configuration) fp32-accurate split-bf16 matrix-core kernels: 1 = 1x1 / "
                         "transposed convolutions with >= 128 outputs, 2 = also the 3x3 convolutions with >= 192 outputs; the "
                         "JSON line then says so in `dtype`")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})")
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
    if a.no_materialize:
        from cwfa_amd import unet as _unet
        _unet._MATERIALIZE = False
    if a.no_materialize_up:
        from cwfa_amd import unet as _unet
        _unet._MATERIALIZE_UP = False
    if a.bf16:
        ops.set_precision("bf16")
        a.split_bf16 = 2
    elif a.split_bf16:
        ops.set_option("split_bf16", a.split_bf16)
    S = 5                                                    # INN_max_down_steps (main.py:106): 4 flow steps + LRNN
    torch.manual_seed(0)
    np.random.seed(0)
    conv_inn, cond_nets = CWFA.build_networks(a.depths, a.side, S, with_lrnn=not a.no_lrnn, device=dev)
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

    out = step()                                             # cold pass: weight packing, allocator
    assert out.shape == (B, a.depths, a.side, a.side) and bool(torch.isfinite(out).all())
    # selection pass: which conv kernel/shape dominates?
    sel = ops.conv_event_sink = ConvEvents()
    step()
    torch.cuda.synchronize()
    tot = sel.totals()
    all_conv_ms = sum(t[0] for t in tot.values())
    dom = max(tot, key=lambda k: tot[k][0])                  # the kernel (one instantiation) with the largest time share
    ops.conv_event_sink = None
    for _ in range(max(a.warmup - 2, 0)):
        step()

    ranked = sorted(tot, key=lambda k: -tot[k][0])
    sink = ops.conv_event_sink = ConvEvents(only={dom})       # only the dominant kernel carries events in the timed region
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

    nll_check = None
    if world > 1:                                   # the metric's second half: "NLL match vs ref (1/2/4/8 GPU)"
        try:
            nll_check = sharded_nll_check(CWFA, conv_inn[0], a, dev, rank, world)
        except Exception as exc:                    # noqa: BLE001  (after the timed region; must not cost the line)
            nll_check = {"error": repr(exc)[:300]}

    res = None
    if rank == 0:
        t_dom, n_dom, f_dom, shapes = sink.totals()[dom]
        avg_ms = t_dom / n_dom
        tf = f_dom / (t_dom * 1e-3) / 1e12
        alg_bytes = sum(conv_bytes(k) * n for k, n in shapes.items()) / n_dom
        res = {
            "metric": "volumes/sec inverse-pass @512x512x96 fp32", "value": world * a.steps * B / elapsed,
            "unit": "volumes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 (conv operands; f32 accumulate, f32 wavelets / couplings)" if a.bf16 else "f32" if not a.split_bf16 else f"f32 (split level {a.split_bf16}: 3-way split bf16 operands, six products, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"{a.side}x{a.side}x{a.depths} volume, 4-scale CWFA (CAT x5 per scale, 64 ch) + "
                                   f"{'LRNN' if not a.no_lrnn else 'synthetic low-res (NO LRNN: diagnostic)'} inverse, z=0, "
                                   f"batch {B}/GPU, random-init weights (BASELINE.json configs[2])",
                       "parallelism": f"replicated x{world} (independent volumes per GPU, no collective)"},
            # achieved = algorithmic conv FLOPs (2*Cout*Cin*taps*H*W per launch) of ALL launches of
            # the dominant kernel inside the timed region / their summed HIP-event durations; avg_launch_ms is the figure
            # to compare with the rocprofv3 kernel-stats average for the same kernel (profiles/).
            "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                         "kernel": dom + (" (v_mfma_f32_32x32x16_bf16, six split products per algorithmic FMA)" if "split" in dom else
                                          " (v_mfma_f32_32x32x2_f32" + (", Winograd: %.2fx the algorithmic MFMA count is issued"
                                                                         % issued_factor(dom) if "wino" in dom else "") + ")"),
                         "mfma_issued_frac": tf * issued_factor(dom) / PEAK_FP32_MFMA_TFLOPS,
                         "shapes": [dict(zip(("ks", "cin", "cout", "H", "W", "B", "launches"), (*k[:6], n)))
                                    for k, n in sorted(shapes.items(), key=lambda kv: -kv[1])],
                         "flops_per_launch": f_dom / n_dom, "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": avg_ms, "launches_timed": n_dom,
                         "share_of_conv_time": tot[dom][0] / all_conv_ms, "all_conv_ms_per_step": all_conv_ms},
            "reference_readme": {"volumes_per_s": 6.25, "note": "README.md:29, unstated CUDA GPU, fp16 autocast; not "
                                 "this fp32 metric, hence vs_baseline is null"},
        }
        res["roofline"]["traffic"], res["roofline"]["traffic_detail"] = pmc_traffic(dom, a)
        if len(ranked) > 1:                                        # the runner-up kernel, same definitions, timed in three
            sink2 = ops.conv_event_sink = ConvEvents(only={ranked[1]})   # extra steps AFTER the timed region (its many small
            for _ in range(3):                                     # launches would otherwise put ~1 % of event gaps into `value`)
                step()
            torch.cuda.synchronize()
            ops.conv_event_sink = None
            t2, n2, f2, sh2 = sink2.totals()[ranked[1]]
            tf2 = f2 / (t2 * 1e-3) / 1e12
            tr2, td2 = pmc_traffic(ranked[1], a)
            res["roofline_second"] = {"bound": "mfma", "achieved": tf2, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                      "frac": tf2 / PEAK_FP32_MFMA_TFLOPS, "traffic": tr2, "kernel": ranked[1],
                                      # issued / algorithmic MFMA work: 2/3 for Winograd 3x3; the fused layer = 9 Winograd taps + 1 direct tap of 10
                                      "mfma_issued_frac": tf2 * issued_factor(ranked[1]) / PEAK_FP32_MFMA_TFLOPS,
                                      "flops_per_launch": f2 / n2, "avg_launch_ms": t2 / n2, "launches_timed": n2,
                                      "algorithmic_bytes_per_launch": sum(conv_bytes(k) * n for k, n in sh2.items()) / n2,
                                      "share_of_conv_time": tot[ranked[1]][0] / all_conv_ms}
        if nll_check is not None:
            res["nll_check"] = nll_check
        res["roofline_dwt"] = dwt_roofline(ops, a, dev)
        if world == 1 and not a.split_bf16 and not a.no_lrnn and not a.no_experiment:
            try:                                              # never let the side experiment cost the headline line
                res["experiment_split_bf16"] = split_experiment(ops, step, max(a.steps // 2, 3), B)
            except Exception as exc:                          # noqa: BLE001
                res["experiment_split_bf16"] = {"error": repr(exc)[:300]}
        if world == 1 and not a.split_bf16 and not a.no_lrnn and not a.no_experiment:
            try:                                              # BASELINE.json configs[4]: bf16 operands in the heavy convolutions
                res["experiment_bf16"] = bf16_experiment(ops, step, max(a.steps // 2, 3), B)
            except Exception as exc:                          # noqa: BLE001
                res["experiment_bf16"] = {"error": repr(exc)[:300]}
        if world == 1 and not a.no_lrnn and not a.no_experiment:
            try:
                res["experiment_train_step"] = train_experiment(conv_inn, cond_nets, dev, a, max(a.steps // 4, 3))
            except Exception as exc:                          # noqa: BLE001
                res["experiment_train_step"] = {"error": repr(exc)[:300]}
        if world == 1 and not a.no_cpu_baseline and not a.no_lrnn:
            try:
                res["cpu_baseline"] = cpu_baseline(conv_inn, cond_nets, cond_input, mean_cache)
            except Exception as exc:                          # noqa: BLE001  (host trouble must not lose the GPU measurement)
                res["cpu_baseline"] = {"value": None, "unit": "volumes/s", "cores": 0, "kind": "port", "sample": "failed: " + repr(exc)[:200]}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return res


def sharded_nll_check(CWFA, g0, a, dev, rank, world):
    """BASELINE.json configs[3] on the ranks of this run: `world` synthetic volumes (same seed everywhere), rank r takes
    volume r, the forward / NLL step of the finest flow, ONE all-reduce of the float64[3] shard sums (RCCL over xGMI) --
    against the same NLL computed by this rank alone over all volumes (no collective).  After the timed region."""
    import torch.distributed as dist
    gen = torch.Generator().manual_seed(3)
    D, S = a.depths, a.side
    x = torch.randn(world, D, S, S, generator=gen).to(dev)
    c = [torch.randn(world, D // 2, S, S, generator=gen).to(dev), (0.1 * torch.randn(world, D // 2, S, S, generator=gen)).to(dev)]
    with torch.no_grad():
        nll, _, _ = CWFA.nll_step(g0, x[rank:rank + 1].contiguous(), [t[rank:rank + 1].contiguous() for t in c])
        _, logdet, sumsq = CWFA.nll_terms(g0, x, c)
    ref = (0.5 * float(sumsq[0]) - float(logdet.double().sum()) / world) / (world * x[0].numel())
    vals = torch.tensor([float(nll)], dtype=torch.float64, device=dev)
    lo, hi = vals.clone(), vals.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return {"nll_sharded": float(nll), "nll_single_process": ref, "rel_diff": abs(float(nll) - ref) / abs(ref),
            "identical_on_all_ranks": bool(float(lo) == float(hi)), "volumes": world, "tolerance": 1e-6}


def split_experiment(ops, step, steps, batch):
    """NOT the headline: the same step with the opt-in fp32-accurate split-bf16 matrix-core kernels (level 2: 1x1 /
    transposed convs and the 3x3 convs with >= 192 outputs), measured after the timed region, for the record."""
    ops.set_option("split_bf16", 2)
    try:
        step(); step()                                       # re-pack the affected filter banks, warm up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    finally:
        ops.set_option("split_bf16", 0)
    return {"value": batch / dt, "unit": "volumes/s", "ms_per_step": 1e3 * dt, "steps": steps,
            "note": "opt-in, not the headline configuration: operands split exactly into three bf16 pieces, six partial "
                    "products on v_mfma_f32_32x32x16_bf16, fp32 accumulation; parity tests hold the fp32 path's bounds "
                    "(tests/test_gpu_parity.py::test_split_bf16_*)"}


def bf16_experiment(ops, step, steps, batch):
    """NOT the headline (which is fp32): BASELINE.json configs[4], the same step with plain bf16 operands in the heavy
    convolutions (fp32 accumulation), measured after the timed region."""
    ops.set_precision("bf16")
    try:
        step(); step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    finally:
        ops.set_precision("fp32")
    return {"value": batch / dt, "unit": "volumes/s", "ms_per_step": 1e3 * dt, "steps": steps, "dtype": "bf16 operands, f32 accumulate",
            "note": "configs[4]; parity (max-rel <= 1e-2, L2-rel <= 5e-3 vs the fp32 oracle) in "
                    "tests/test_gpu_parity.py::test_full_config3_inverse_vs_oracle"}


def train_experiment(conv_inn, cond_nets, dev, a, steps):
    """NOT the headline: SURVEY.md 8(f) row 1 -- one training iteration over the WHOLE pyramid on one synthetic volume, in
    the reference's order (CWFA.py:865-1027): LRNN step (L2), then the four flow steps with their condition nets (inverse +
    forward + backward of 0.40984 * mse + 0.59016 * NLL); gradients computed, no optimiser update.  After the timed region."""
    from cwfa_amd import training
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
    return {"value": B / dt, "unit": "volumes/s", "ms_per_step": 1e3 * dt, "steps": steps,
            "full_loss_per_pyramid_step": [float(v) for v in res["losses"]], "trainable_parameters": n_par,
            "note": "LRNN + 4 flow steps + condition nets, forward with tape + backward; gradients pinned to the reference's autograd "
                    "by tests/test_gpu_backward.py (fixtures g13-g15)"}


def pmc_traffic(dom, a):
    """HBM bytes per launch of the dominant kernel.  FETCH_SIZE / WRITE_SIZE cannot be collected from inside this process
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


def dwt_roofline(ops, a, dev, reps=20):
    """The DWT stage on its own: the four inverse depth-Haar levels of one volume (377.5 MB algorithmic at 512x512x96,
    SURVEY.md 8d: 8 bytes per element read+written), standalone kernels, HIP events on the launch stream."""
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
    ms = e0.elapsed_time(e1) / reps
    nbytes = sum(8.0 * b.numel() for b in bufs)
    gbs = nbytes / (ms * 1e-3) / 1e9
    # the largest level alone (per-launch figure for the rocprof cross-check)
    e0.record()
    for _ in range(reps):
        ops.haar1d(bufs[0], True)
    e1.record()
    torch.cuda.synchronize()
    ms0 = e0.elapsed_time(e1) / reps
    return {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
            "traffic": None, "kernel": "haar1d_inv_kernel<4>", "bytes_per_volume": nbytes, "us_per_volume": 1e3 * ms,
            "largest_level": {"bytes_per_launch": 8.0 * bufs[0].numel(), "avg_launch_us": 1e3 * ms0,
                              "GBps": 8.0 * bufs[0].numel() / (ms0 * 1e-3) / 1e9}}


def cpu_baseline(conv_inn, cond_nets, cond_input, mean_cache):
    """The CPU oracle (oracle/cwfa_oracle.py, a port of the reference's op sequence onto torch CPU ops) on the same
    weights and inputs: ONE full 512x512x96 volume, all host cores PyTorch gives us."""
    from oracle import cwfa_oracle as O
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}   # noqa: E731
    steps = []
    for n, g in enumerate(conv_inn):
        axes = {i: (m.axis if hasattr(m, "axis") else 1) for i, m in enumerate(g.module_list) if hasattr(m, "perm")}
        steps.append({"inn": cpu(g.state_dict()), "omega": cpu(cond_nets[n].state_dict()), "axes": axes})
    lrnn_sd = cpu(cond_nets[-1].state_dict())
    ci = cond_input[:1].cpu()
    mc = [m[:1].cpu() for m in mean_cache]
    t0 = time.perf_counter()
    with torch.no_grad():
        vols = O.inverse_pass(steps, None, ci, mc, lrnn_sd=lrnn_sd, lrnn_train=True)
    dt = time.perf_counter() - t0
    assert vols[-1].shape[1] == conv_inn[0].dims_in[0][0]
    return {"value": 1.0 / dt, "unit": "volumes/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 full volume (LRNN + 4 flow steps + condition nets), single cold run, {dt:.1f} s, "
                      f"torch {torch.__version__} CPU fp32"}


if __name__ == "__main__":
    main()
