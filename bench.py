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


class ConvEvents:
    """Event sink for ops.conv2d: everything (selection pass) or one kernel/shape key (timed region)."""

    def __init__(self, only=None):
        self.only, self.rows = only, []

    def want(self, key):
        return self.only is None or key == self.only

    def add(self, key, e0, e1):
        self.rows.append((key, e0, e1))

    def totals(self):
        tot = {}
        for key, e0, e1 in self.rows:
            t, n = tot.get(key, (0.0, 0))
            tot[key] = (t + e0.elapsed_time(e1), n + 1)
        return tot


def conv_flops(key):
    ks, cin, cout, H, W, B = key
    taps = 10 if ks == "L" else ks * ks          # "L": fused 3x3 + 1x1 sub-network layer (9 + 1 taps)
    return 2.0 * cout * cin * taps * H * W * B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="volumes per step per GPU")
    ap.add_argument("--side", type=int, default=512)
    ap.add_argument("--depths", type=int, default=96)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lrnn", action="store_true", help="(diagnostic) flows + condition nets only")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})")
    import torch.distributed as dist
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)       # nccl == RCCL on ROCm

    from cwfa_amd import CWFA, ops
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
    all_conv_ms = sum(t for t, _ in tot.values())
    dom = max(tot, key=lambda k: tot[k][0])
    ops.conv_event_sink = None
    for _ in range(max(a.warmup - 2, 0)):
        step()

    sink = ops.conv_event_sink = ConvEvents(only=dom)
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

    res = None
    if rank == 0:
        t_dom, n_dom = sink.totals()[dom]
        avg_ms = t_dom / n_dom
        tf = conv_flops(dom) / (avg_ms * 1e-3) / 1e12
        res = {
            "metric": "volumes/sec inverse-pass @512x512x96 fp32", "value": world * a.steps * B / elapsed,
            "unit": "volumes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.side}x{a.side}x{a.depths} volume, 4-scale CWFA (CAT x5 per scale, 64 ch) + "
                                   f"{'LRNN' if not a.no_lrnn else 'synthetic low-res (NO LRNN: diagnostic)'} inverse, z=0, "
                                   f"batch {B}/GPU, random-init weights (BASELINE.json configs[2])",
                       "parallelism": f"replicated x{world} (independent volumes per GPU, no collective)"},
            "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                         "kernel": ("subnet_layer_kernel" if dom[0] == "L" else "conv2d_mfma_kernel") +
                                   " (v_mfma_f32_32x32x2_f32)",
                         "shape": dict(zip(("ks", "cin", "cout", "H", "W", "B"), dom)),
                         "flops_per_launch": conv_flops(dom), "avg_launch_ms": avg_ms, "launches_timed": n_dom,
                         "share_of_conv_time": tot[dom][0] / all_conv_ms,
                         "all_conv_ms_per_step": all_conv_ms},
            "reference_readme": {"volumes_per_s": 6.25, "note": "README.md:29, unstated CUDA GPU, fp16 autocast; not "
                                 "this fp32 metric, hence vs_baseline is null"},
        }
        res["roofline"]["traffic"] = pmc_traffic(res["roofline"]["kernel"], dom)
        res["roofline_dwt"] = dwt_roofline(ops, a, dev)
        if world == 1 and not a.no_cpu_baseline and not a.no_lrnn:
            res["cpu_baseline"] = cpu_baseline(conv_inn, cond_nets, cond_input, mean_cache)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return res


def pmc_traffic(kernel, dom):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
    cannot be collected from inside this process; profiles/*_pmc_traffic.json holds the latest separate-pass numbers,
    taken on the same shape).  None if no matching record."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files or dom[3:5] != (512, 512):
        return None
    rec = json.load(open(files[-1]))["kernels"]
    name = kernel.split(" ")[0]
    for k, v in rec.items():
        if k.startswith(name) and (name != "conv2d_mfma_kernel" or "true" in k):
            return {"bytes": (v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024.0, "source": os.path.basename(files[-1]),
                    "algorithmic_bytes": 2 * 4.0 * dom[1] * dom[3] * dom[4] * dom[5]}
    return None


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
