"""CPU, 2 processes over gloo: the batch-sharded NLL of SURVEY.md 8(e).  Each rank owns a contiguous slice of the batch,
computes its shard sums (here with the CPU oracle standing in for the GPU kernels), one all-reduce of a float64[3]
vector combines them, and every rank must obtain the single-process NLL of CWFA.py:978."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden, sd_of


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cwfa_amd.CWFA import allreduce_nll
        from oracle import cwfa_oracle as O
        fx = load_golden("g09_step_CAT_k0")
        sd = sd_of(fx)
        axes = {i: 1 for i in range(3, 12, 2)}
        axes.update({int(k.split("_")[-1]): int(v) for k, v in fx.items() if k.startswith("meta/axis_")})
        x, c = torch.from_numpy(fx["x"]), [torch.from_numpy(fx["c0"]), torch.from_numpy(fx["c1"])]
        B = x.shape[0]
        lo, hi = rank * B // world, (rank + 1) * B // world            # contiguous batch split
        (z, low), jac = O.flow_step(sd, x[lo:hi], [t[lo:hi] for t in c], False, axes)
        s, j, n = O.nll_terms(z, jac)
        terms = allreduce_nll(torch.tensor([s, j, float(n)], dtype=torch.float64))
        nll = O.nll_from_terms(float(terms[0]), float(terms[1]), int(terms[2]), int(terms[2]) * x[0].numel())
        # single-process reference over the whole batch (CWFA.py:970-978)
        (zf, lowf), jf = O.flow_step(sd, x, c, False, axes)
        ref = float((0.5 * torch.norm(zf) ** 2 - jf.mean()) / x.numel())         # CWFA.py:978: / upsampled_vol.numel()
        q.put((rank, nll, ref, int(terms[2])))
    finally:
        dist.destroy_process_group()


def test_sharded_nll_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    for _, nll, ref, n in res:
        assert n == 2
        assert abs(nll - ref) <= 1e-5 * abs(ref), (nll, ref)
    assert res[0][1] == res[1][1], "every rank must hold the identical global NLL"


def _score_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cwfa_amd.CWFA import allgather_scores, detect_ood
        from oracle import cwfa_oracle as O
        fx = load_golden("g09_step_CAT_k0")
        sd = sd_of(fx)
        axes = {i: 1 for i in range(3, 12, 2)}
        axes.update({int(k.split("_")[-1]): int(v) for k, v in fx.items() if k.startswith("meta/axis_")})
        x, c = torch.from_numpy(fx["x"]), [torch.from_numpy(fx["c0"]), torch.from_numpy(fx["c1"])]
        x, c = torch.cat([x, 0.5 * x, -x], 0), [torch.cat([t, t, t], 0) for t in c]      # 3B samples, uneven shards
        B = x.shape[0]
        lo, hi = (0, B // 3) if rank == 0 else (B // 3, B)
        (z, low), jac = O.flow_step(sd, x[lo:hi], [t[lo:hi] for t in c], False, axes)
        local = O.step_log_likelihood(z, jac, low[0].numel()).view(-1, 1)
        scores = allgather_scores(local)
        (zf, lowf), jf = O.flow_step(sd, x, c, False, axes)
        ref = O.step_log_likelihood(zf, jf, lowf[0].numel()).view(-1, 1)
        flags = detect_ood(scores, 0, float(ref.median()))
        q.put((rank, scores.tolist(), ref.tolist(), flags.tolist()))
    finally:
        dist.destroy_process_group()


def test_sharded_ood_scores_two_ranks_gloo():
    """SURVEY.md 8(f)4 over 8(e)'s sharding: uneven batch shards, per-sample step log-likelihoods all-gathered in rank
    order; every rank holds the single-process scores of the whole batch."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_score_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, scores, ref, flags in res:
        assert len(scores) == len(ref)
        assert torch.allclose(torch.tensor(scores), torch.tensor(ref), rtol=1e-9, atol=0)
        assert 0 < sum(flags) < len(flags)
    assert res[0][1] == res[1][1]


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cwfa_amd.training import allreduce_gradients
        g = torch.Generator().manual_seed(7)
        shapes = [(64, 64, 3, 3), (64,), (24, 64, 3, 3), (5,), (64, 58, 1, 1)]
        params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        frozen = torch.nn.Parameter(torch.zeros(3), requires_grad=False)
        unused = torch.nn.Parameter(torch.zeros(4))                        # no rank's loss reaches it: .grad stays None
        full = [torch.randn(world, *s, generator=g) for s in shapes]          # every rank draws the same numbers
        for p, f in zip(params, full):
            p.grad = f[rank].clone()
        if rank == 1:
            params[3].grad = None                                            # a parameter this rank's shard never reached
        nb = allreduce_gradients(params[:2] + [unused] + params[2:] + [frozen], bucket_bytes=64 * 64 * 9 * 4 + 1024)
        want = [f.sum(0) if i != 3 else f[0] for i, f in enumerate(full)]
        ok = all(torch.allclose(p.grad, w, rtol=1e-6, atol=1e-6) for p, w in zip(params, want))
        q.put((rank, ok, nb, frozen.grad is None and unused.grad is None))
    finally:
        dist.destroy_process_group()


def test_bucketed_gradient_allreduce_two_ranks_gloo():
    """the ranks' gradients are SUMMED (the loss is normalised by the global batch) through flat
    buckets; small bucket size here so that the five tensors need several messages."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, ok, nb, frozen_untouched in res:
        assert ok and frozen_untouched
        assert nb == 2, nb            # [w0 b0] [w1 b1 w2]: a bucket closes where the next tensor no longer fits
