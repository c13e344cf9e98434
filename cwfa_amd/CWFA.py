"""Hot-path drivers (reference: CWFA.py): z sampling (:47-64), the forward pyramid + NLL evaluation
(``evaluate_INN_forward`` :134-196), the per-volume inverse loop (:865-924) and the training-time NLL (:966-978),
plus the batch-sharded multi-GPU NLL (new; the reference is single-device).

Out of scope: experiment management of ``run_CWFA`` -- checkpoint discovery, optimisers,
TensorBoard, metrics, figures, TIFF export.
"""
import math

import torch

from . import ops

__all__ = ["sample_z_truncated", "check_empty_depths", "evaluate_INN_forward", "inverse_pass", "nll_step",
           "nll_terms", "allreduce_nll", "build_networks", "step_log_likelihoods", "allgather_scores", "detect_ood",
           "forward_nll_pass", "mean_volume_cache", "save_mean_volume_cache", "load_mean_volume_cache",
           "denormalise_prediction", "denormalise_ground_truth"]


def _no_grad_trunc_normal_(tensor, mean=0., std=1., a=-1., b=1.):
    """Truncated normal by inverse-CDF sampling (utils.py:42-82 semantics).  Only reached for temperature != 0, a path
    that raises NameError in the reference itself; provided for completeness and checked
    statistically only."""
    def norm_cdf(v):
        return (1. + math.erf(v / math.sqrt(2.))) / 2.
    with torch.no_grad():
        lo, up = norm_cdf((a - mean) / std), norm_cdf((b - mean) / std)
        tensor.uniform_(2 * lo - 1, 2 * up - 1)
        tensor.erfinv_()
        tensor.mul_(std * math.sqrt(2.))
        tensor.add_(mean)
        tensor.clamp_(min=a, max=b)
        return tensor


def sample_z_truncated(x, device="cpu", temperature=1):
    """Latent sample; ``temperature == 0`` (the default of main.py:109) returns zeros.  CWFA.py:47-64.
    The fused inverse treats ``None`` as an all-zero z and never reads it (``inverse_pass`` does that itself)."""
    shape_like = torch.is_tensor(x)
    if temperature == 0:
        return torch.zeros_like(x, device=device) if shape_like else torch.zeros(x, device=device)
    base = torch.zeros_like(x, device=device) if shape_like else torch.zeros(x, device=device)
    return _no_grad_trunc_normal_(base, a=-temperature, b=temperature)


def check_empty_depths(gt_volume):
    """Add tiny noise to depth columns that are constant over depth (avoids degenerate statistics).  CWFA.py:84-96.
    Bookkeeping on the input volume (torch ops), not part of the kernel path."""
    empty = gt_volume.std(dim=1) == 0
    n_depths = gt_volume.shape[1]
    empty = empty.view(gt_volume.shape[0], empty.shape[0] // gt_volume.shape[0], empty.shape[1], empty.shape[2])
    if empty.any():
        sel = empty.repeat(1, n_depths, 1, 1)
        gt_volume[sel] += torch.normal(0, 0.001, gt_volume[sel].size(), device=gt_volume.device)
    return gt_volume


def nll_terms(graph, x, c):
    """Forward one step and return the shard-local NLL sums: (Z tuple, logdet[B], sum ||Z0||^2 as float64[1] tensor).
    The sum of squares comes out of the fused chain kernel (no extra pass over Z)."""
    sumsq = torch.zeros(1, dtype=torch.float64, device=x.device)
    Z, logdet = graph(x, c=c, sumsq=sumsq) if getattr(graph, "_plan", None) is not None else graph(x, c=c)
    if getattr(graph, "_plan", None) is None:
        st = ops.sample_stats(Z[0].reshape(1, -1, 1, 1))
        sumsq = st[1:2]
    return Z, logdet, sumsq


def evaluate_INN_forward(conv_inn, cond_nets, args_general, args_nets, gt_volume, input_views, train_statistics,
                         extra_cond_in=None):
    """Forward pyramid with per-step likelihood terms.  CWFA.py:134-196.
    Returns (losses, gt_cache, prior_errors, log_jacobians) with the reference's normalisations."""
    device = gt_volume.device
    gt_volume = check_empty_depths(gt_volume)
    mean_imgs, std_imgs = train_statistics[0], train_statistics[1]
    losses, prior_errors, log_jacobians = [], [], []
    gt_cache = args_general.INN_max_down_steps * [None]
    gt_cache[0] = gt_volume
    cond_input = (input_views - mean_imgs) / std_imgs
    B = gt_volume.shape[0]
    for n_net in range(len(conv_inn)):
        g = conv_inn[n_net]
        is_last_step = n_net == args_general.INN_max_down_steps - 1
        if is_last_step:
            cond_in = [] if args_general.force_all_steps_NF else [cond_nets[n_net](cond_input)[-1]]
        else:
            cond_in = [torch.zeros((B,) + tuple(g.dims_c[0]), device=device)] if len(g.dims_c) > 0 else []
        if len(g.dims_c) > 1:
            if extra_cond_in is None:
                cond_in.append(torch.zeros((B,) + tuple(g.dims_c[1]), device=device))
            else:
                cond_in.append(extra_cond_in[n_net].clone())
        Z, log_jac_det, sumsq = nll_terms(g, gt_volume, cond_in)
        error_on_prior = sumsq[0].to(torch.float32)
        numel = Z[-1].numel()
        curr = (0.5 * error_on_prior - log_jac_det) / numel
        losses.append(curr.mean())
        prior_errors.append(0.5 * error_on_prior.mean() / numel)
        log_jacobians.append(log_jac_det.mean() / numel)
        if not is_last_step:
            gt_volume = Z[1]
            gt_cache[n_net + 1] = gt_volume
    return losses, gt_cache, prior_errors, log_jacobians


def step_log_likelihoods(conv_inn, cond_nets, args_general, gt_volume, input_views, train_statistics, extra_cond_in=None,
                         group=None):
    """Per-sample, per-step log-likelihoods of a batch under the flows: LL[b, n] = -(0.5*||z_b||^2 - logdet_b) / numel_b
    -- the per-volume form of ``evaluate_INN_forward``'s losses (CWFA.py:183-186; for a batch of one, LL[0, n] is
    exactly -losses[n]).  This is the score the reference thresholds for out-of-distribution detection
    (main.py:78-80: ``--step_LL_to_use``, ``--step_LL_ths_to_use``; its evaluator ``main_OOD`` is not part of the
    released sources, main.py:16,401).  One fused forward chain per step; the per-sample sums of squares are one
    extra read of Z0 (``sample_stats``).  With a process group every rank scores its shard and the [B_local, S] blocks
    are all-gathered in rank order (RCCL on MI355X): every rank returns the scores of the global batch.
    Returns a float64 tensor [B, len(conv_inn)]."""
    device = gt_volume.device
    gt_volume = check_empty_depths(gt_volume)
    cond_input = (input_views - train_statistics[0]) / train_statistics[1]
    B = gt_volume.shape[0]
    cols = []
    for n_net, g in enumerate(conv_inn):
        is_last_step = n_net == args_general.INN_max_down_steps - 1
        if is_last_step:
            cond_in = [] if args_general.force_all_steps_NF else [cond_nets[n_net](cond_input)[-1]]
        else:
            cond_in = [torch.zeros((B,) + tuple(g.dims_c[0]), device=device)] if len(g.dims_c) > 0 else []
        if len(g.dims_c) > 1:
            cond_in.append(torch.zeros((B,) + tuple(g.dims_c[1]), device=device) if extra_cond_in is None
                           else extra_cond_in[n_net].clone())
        Z, logdet = g(gt_volume, c=cond_in)
        sumsq = ops.sample_stats(Z[0]).view(B, 2)[:, 1]
        cols.append(-(0.5 * sumsq - logdet.to(torch.float64)) / Z[-1][0].numel())
        if not is_last_step:
            gt_volume = Z[1]
    return allgather_scores(torch.stack(cols, 1), group)


def allgather_scores(scores, group=None):
    """Concatenate the ranks' [B_local, S] score blocks in rank order (shards may differ in size); no-op without
    torch.distributed.  Two small all-gathers: the shard sizes, then the blocks padded to the largest shard."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return scores
    device, B = scores.device, scores.shape[0]
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(dist.get_world_size(group))]
    dist.all_gather(counts, torch.tensor([B], dtype=torch.int64, device=device), group=group)
    counts = [int(c_) for c_ in counts]
    padded = torch.zeros(max(counts), scores.shape[1], dtype=scores.dtype, device=device)
    padded[:B] = scores
    blocks = [torch.empty_like(padded) for _ in counts]
    dist.all_gather(blocks, padded, group=group)
    return torch.cat([blk[:c_] for blk, c_ in zip(blocks, counts)], 0)


def detect_ood(scores, step_LL_to_use=0, step_LL_ths_to_use=-1.33):
    """Out-of-distribution flags from ``step_log_likelihoods``: a sample is flagged when the log-likelihood of step
    ``step_LL_to_use`` falls below the threshold (defaults of main.py:79-80)."""
    return scores[:, step_LL_to_use] < step_LL_ths_to_use


def inverse_pass(conv_inn, cond_nets, cond_input, mean_vols_cache, low=None, temperature=0.0, n_samples=1,
                 keep_all=False):
    """The reconstruction loop for one batch of views.  CWFA.py:865-924.

    ``cond_nets`` has one condition net per flow step plus, if ``low`` is None, the LRNN encoder as its last entry
    (CWFA.py:495-496,882).  ``mean_vols_cache[n]`` is the mean-volume detail at step n (CWFA.py:655,899).
    Returns the full-resolution volume (or every level, coarse -> fine, with ``keep_all``)."""
    S1 = len(conv_inn)
    if low is None:
        up = cond_nets[S1](cond_input, mean_vols_cache[S1 - 1])[-1]
    else:
        up = low
    vols = [up]
    from .networks import omega_first_scope
    with omega_first_scope(list(cond_nets[:S1]), cond_input):      # the steps' condition nets share their input: first convs in one launch
        return _inverse_steps(conv_inn, cond_nets, cond_input, mean_vols_cache, up, vols, temperature, n_samples, keep_all)


def _inverse_steps(conv_inn, cond_nets, cond_input, mean_vols_cache, up, vols, temperature, n_samples, keep_all):
    S1 = len(conv_inn)
    for n in range(S1 - 1, -1, -1):
        g = conv_inn[n]
        cond_processed = [cond_nets[n](cond_input)[-1], mean_vols_cache[n]]
        if n_samples > 1:
            cond_processed = [cc.repeat(n_samples, 1, 1, 1) for cc in cond_processed]
            up = up.repeat(n_samples, 1, 1, 1)
        if temperature == 0 and getattr(g, "_plan", None) is not None:
            z = None                              # z == 0: the fused chain never reads it (CWFA.py:54-55)
        else:
            z = sample_z_truncated((up.shape[0],) + tuple(g.global_out_shapes[0]), device=up.device,
                                   temperature=temperature)
        up, _ = g([z, up], c=cond_processed, rev=True, jac=getattr(g, "_plan", None) is None)   # log-det is discarded
        if n_samples > 1:                         # CWFA.py:913-914: average the samples
            parts = up.view(n_samples, -1, *up.shape[1:])
            acc = ops.axpby(parts[0], 1.0 / n_samples)
            for i in range(1, n_samples):
                acc = ops.axpby(parts[i], 1.0 / n_samples, acc, 1.0)
            up = acc
        vols.append(up)
    return vols if keep_all else up


def nll_step(graph, x, c, group=None):
    """Training-time NLL of one step, CWFA.py:966-978:  (0.5*||Z0||^2 - mean_b logdet) / numel(batch volume),
    with the norm taken over the WHOLE (global) batch.  The divisor is ``upsampled_vol.numel()`` = B*D_n*H*W, the step's
    full-depth volume (CWFA.py:911,978) -- twice ``Z[-1].numel()``, which ``evaluate_INN_forward`` divides by (:186).  With a process group the three shard sums are all-reduced
    (RCCL over xGMI on MI355X: one float64[3] message) and every rank returns the identical global value."""
    Z, logdet, sumsq = nll_terms(graph, x, c)
    terms = torch.stack([sumsq[0], logdet.to(torch.float64).sum(),
                         torch.tensor(float(x.shape[0]), dtype=torch.float64, device=x.device)])
    terms = allreduce_nll(terms, group)
    numel_total = terms[2] * x[0].numel()      # `upsampled_vol.numel()`: the step's whole input volume (CWFA.py:911,978)
    nll = (0.5 * terms[0] - terms[1] / terms[2]) / numel_total
    return nll, Z, logdet


def forward_nll_pass(conv_inn, cond_nets, gt_volume, cond_input, mean_vols_cache, group=None):
    """The forward / NLL twin of ``inverse_pass`` over the whole pyramid for one (shard of a) batch -- BASELINE.json
    configs[3]: for n = 0 .. S-2: condition net Omega_n, ``Z, logdet = conv_inn[n](gt_n, c=[Omega_n(views), mean_n])``
    (CWFA.py:895-899,966), ``gt_{n+1} = Z[1]`` (the low band: what CWFA.py:821 caches), and
    ``NLL_n = (0.5 * ||Z0||^2 - mean_b logdet) / numel(batch volume)`` (CWFA.py:970-978) with the norm and the mean
    taken over the GLOBAL batch.  With a process group the S-1 triples {sum z^2, sum logdet, B_local} are summed over
    the ranks in ONE all-reduce of a float64[3(S-1)] vector (RCCL over xGMI on MI355X) and every rank returns the same
    values.  Returns (nll float64[S-1], low-resolution volume gt_{S-1})."""
    gt = gt_volume
    rows = []
    from .networks import omega_first_scope
    with omega_first_scope(list(cond_nets[:len(conv_inn)]), cond_input):
        for n, g in enumerate(conv_inn):
            Z, logdet, sumsq = nll_terms(g, gt, [cond_nets[n](cond_input)[-1], mean_vols_cache[n]])
            rows.append(torch.stack([sumsq[0], logdet.to(torch.float64).sum(),
                                     torch.tensor(float(gt.shape[0]), dtype=torch.float64, device=gt.device)]))
            gt = Z[1]
    terms = allreduce_nll(torch.stack(rows).reshape(-1), group).view(len(conv_inn), 3)
    numel = torch.tensor([float(gt_volume[0].numel()) / 2 ** n for n in range(len(conv_inn))], dtype=torch.float64,
                         device=gt_volume.device)                    # per-sample numel of the step's input volume
    nll = (0.5 * terms[:, 0] - terms[:, 1] / terms[:, 2]) / (terms[:, 2] * numel)
    return nll, gt


def allreduce_nll(terms, group=None):
    """Sum the [sum z^2, sum logdet, B] vector over the data-parallel ranks (no-op without torch.distributed)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(terms, op=dist.ReduceOp.SUM, group=group)
    return terms


def build_networks(n_depths=96, side=512, max_down_steps=5, block_type="CAT", n_blocks=4, internal_chans=64,
                   cond_chans=32, use_perm=True, use_bias=True, with_lrnn=True, device="cuda"):
    """Build the model list the way run_CWFA does (CWFA.py:478-532): one (GraphINN, cond_network) per flow step and the
    LRNN Encoder as the last condition net.  Flow nets and condition nets in eval mode, the LRNN in train mode."""
    from . import networks as N
    conv_inn, cond_nets = [], []
    for ix in range(max_down_steps - 1):
        cn = n_depths // (2 ** (ix + 1))
        cond_net, inns = N.conditional_wavelet_flow(
            input_volume_shape=[n_depths, side, side], condition_shape=[1, 29, side, side],
            st_subnet=N.wavelet_flow_subnetwork2D,
            conditional_network=lambda: N.cond_network(29, cn, ix + 1, max_down_steps, [], cond_chans),
            n_internal_ch=internal_chans, n_down_steps=ix + 1, use_permutations=use_perm, block_type=block_type,
            n_blocks=n_blocks, device="cpu")
        conv_inn.append(inns[ix].eval().to(device))
        cond_nets.append(cond_net.eval().to(device))
    if with_lrnn:
        enc = N.Encoder(29, n_depths // (2 ** (max_down_steps - 1)), max_down_steps, internal_chans, use_bias).to(device)
        enc.train()                                     # CWFA.py:532: BatchNorm batch statistics, dropout, drop_path
        cond_nets.append(enc)
    return conv_inn, cond_nets


# ---------------------------------------------------------------------------------------------------------------------
# Mean-volume cache and output step
# ---------------------------------------------------------------------------------------------------------------------
def mean_volume_cache(gt_levels):
    """The per-step second condition of the flows: for every level of the forward pyramid of a (normalised) mean volume
    -- ``evaluate_INN_forward``'s gt_cache -- the difference of the even and odd depth planes of its FIRST sample,
    ``gt[0, ::2] - gt[0, 1::2]`` (CWFA.py:655).  One channel gather (even planes | odd planes) and one subtraction per
    level on the HIP kernels; returns tensors [D_n / 2, H, W] like the reference."""
    out = []
    for gt in gt_levels:
        if gt is None:
            continue
        D = gt.shape[1]
        order = torch.cat([torch.arange(0, D, 2), torch.arange(1, D, 2)]).to(gt.device)
        eo = ops.gather(gt[:1], order, 1)                                  # [1, D, H, W]: even planes, then odd planes
        out.append(ops.axpby(eo[:, :D // 2], 1.0, eo[:, D // 2:], -1.0)[0])
    return out


def save_mean_volume_cache(path, cache):
    """``torch.save({'mean_vol_gt_cache': [...cpu tensors...]}, path)`` -- the file main.py:377 writes (named
    ``mean_vol_{N}Imgs_ds_{id}_{split}`` there)."""
    torch.save({'mean_vol_gt_cache': [v.detach().cpu() for v in cache]}, path)


def load_mean_volume_cache(path_or_dir, device="cpu", dataset_id=None, split=None):
    """Read a mean-volume cache as CWFA.py:637-640 does: ``path_or_dir`` is the file, or with ``dataset_id`` and ``split`` the
    directory searched for ``mean_vol_*ds_{dataset_id}_{split}`` (first match).  The file holds plain tensors: it is read
    with ``weights_only=True`` (nothing in it is executed).  Returns the list moved to ``device``, or None if no file matches."""
    import glob
    import os
    path = path_or_dir
    if dataset_id is not None:
        hits = sorted(glob.glob(os.path.join(str(path_or_dir), f"mean_vol_*ds_{dataset_id}_{split}")))
        if not hits:
            return None
        path = hits[0]
    data = torch.load(path, map_location="cpu", weights_only=True)
    return [v.to(device) for v in data['mean_vol_gt_cache']]


def denormalise_prediction(stored_volume, std_vols, mean_vols):
    """The evaluation branch's predicted output volume, CWFA.py:1041:
    ``(stored_volumes[0][0] * 2**len(stored_volumes[0])) * std_vols + mean_vols`` -- first sample of the finest reconstruction;
    the factor is 2 to the power of the BATCH length of that tensor, exactly as the reference has it.  One per-channel
    affine launch (scale and shift are the same scalar for every depth; a power of two commutes with the rounding of the
    product, and the kernel multiplies, then adds, like the reference)."""
    x = stored_volume[:1]
    C_ = x.shape[1]
    scale = (torch.as_tensor(std_vols, dtype=torch.float32).reshape(()) * float(2 ** len(stored_volume))).to(x.device)
    shift = torch.as_tensor(mean_vols, dtype=torch.float32).reshape(()).to(x.device)
    return ops.channel_affine(x, scale.expand(C_).contiguous(), shift.expand(C_).contiguous())[0]


def denormalise_ground_truth(gt_volume, std_vols, mean_vols):
    """The ground-truth volume of the evaluation branch, CWFA.py:1037-1038: ``gt[0] * std + mean`` shifted so that its minimum
    is 0.  The affine is the HIP kernel; the minimum and its subtraction are two torch reductions / elementwise ops on the
    result (output bookkeeping after the path, like the metrics that follow it in the reference)."""
    x = gt_volume[:1]
    C_ = x.shape[1]
    scale = torch.as_tensor(std_vols, dtype=torch.float32).reshape(()).to(x.device)
    shift = torch.as_tensor(mean_vols, dtype=torch.float32).reshape(()).to(x.device)
    v = ops.channel_affine(x, scale.expand(C_).contiguous(), shift.expand(C_).contiguous())[0]
    return v - v.min()
