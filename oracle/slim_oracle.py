"""CPU oracle for the SLiM per-layer compression hot path.

TEST INFRASTRUCTURE ONLY.  This module restates, with torch CPU ops, the algorithm of
the reference implementation so that the HIP kernels can be checked against it.  It may
be imported only by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline
leg.  The product package (slim_amd) never imports it and has no CPU fallback.

Parity pin: every function below is checked against golden vectors produced by running
the reference itself in the development container (tests/golden/make_golden.py ->
tests/golden/*.npz, test file tests/test_oracle_golden.py).

Each function cites the reference file:line it follows (paths relative to the reference
checkout).
"""
import math

import torch

__all__ = [
    "add_batch_scaler_row", "add_batch_gram", "shift_zeros", "saliency", "prune_nm",
    "histogram_density", "compute_average_error", "find_optimal_quantization_cap",
    "num_bins_for", "quantize_per_tensor", "dequantize_per_tensor",
    "group_quant_params", "group_quantize", "group_dequantize",
    "lora_rank_for", "add_lora", "important_columns", "group_fake_quantize", "quantize_input_fp8", "quantize_input_int",
    "compress_linear",
]


# --------------------------------------------------------------------------------------
# Calibration statistics
# --------------------------------------------------------------------------------------
def add_batch_scaler_row(scaler_row, nsamples, inp):
    """One WrappedGPT.add_batch call (slim/layerwrapper.py:25-41).

    inp: [B, S, K] or [S, K] activations of one forward call.  Returns the updated
    (scaler_row fp32 [K], nsamples).  The squared norm is formed as sqrt-then-square,
    exactly as `torch.norm(inp, p=2, dim=1) ** 2` does.
    """
    if inp.dim() == 2:
        inp = inp.unsqueeze(0)
    b = inp.shape[0]
    x = inp.reshape(-1, inp.shape[-1]).t()                       # [K, B*S]
    scaler_row = scaler_row * (nsamples / (nsamples + b))
    nsamples += b
    x = x.type(torch.float32)
    scaler_row = scaler_row + torch.norm(x, p=2, dim=1) ** 2 / nsamples
    return scaler_row, nsamples


def add_batch_gram(H, nsamples, inp):
    """One SparseGPT.add_batch call (slim/sparsegpt.py:28-39): H = (2/n) * sum X^T X in fp32."""
    if inp.dim() == 2:
        inp = inp.unsqueeze(0)
    b = inp.shape[0]
    x = inp.reshape(-1, inp.shape[-1]).t()
    H = H * (nsamples / (nsamples + b))
    nsamples += b
    x = math.sqrt(2 / nsamples) * x.float()
    H = H + x.matmul(x.t())
    return H, nsamples


def shift_zeros(x):
    """slim/utils.py:64-72: add the smallest non-zero entry (zeros count as 1) to every element."""
    m = x.clone()
    m[m == 0] = 1
    return x + m.min()


def saliency(W, scaler_row):
    """slim/prune.py:264-265: |W| (bf16) * sqrt(scaler_row) (fp32) -> fp32 [N, K]."""
    return torch.abs(W) * torch.sqrt(scaler_row.reshape((1, -1)))


def prune_nm(mat, n, m):
    """slim/utils.py:75-88: in every aligned group of m columns mark the n smallest True.

    The reference calls topk(largest=False) once per column group; ties are resolved by
    that routine (std::nth_element over (value, index) pairs).  One topk over all groups
    at once performs the identical per-row selection.
    """
    N, K = mat.shape
    g = mat.float().reshape(N * (K // m), m)
    idx = torch.topk(g, n, dim=1, largest=False)[1]
    mask = torch.zeros_like(g, dtype=torch.bool)
    mask.scatter_(1, idx, True)
    return mask.reshape(N, K)


def prune_unstructured(mat, sparsity_ratio):
    """slim/prune.py:275-279: per row, a stable ascending sort and the first int(K * ratio) indices pruned."""
    order = torch.sort(mat.float(), dim=-1, stable=True)[1]
    idx = order[:, :int(mat.shape[1] * sparsity_ratio)]
    mask = torch.zeros(mat.shape, dtype=torch.bool)
    mask.scatter_(1, idx, True)
    return mask


def compress_linear_plain(W, W_mask, num_bits=4, quantize_first=True):
    """slim/prune.py:330-357 with a per-tensor SLiM-Quant quantizer and no adapters: quantise then mask, or mask
    then quantise.  Returns (weight bf16, scaling factor)."""
    W = W.clone()
    if not quantize_first:
        W[W_mask] = 0
    codes, sf = quantize_per_tensor(W, num_bits, slim_quant=True)
    out = dequantize_per_tensor(codes, sf).to(torch.bfloat16)
    if quantize_first:
        out[W_mask] = 0
    return out, sf


# --------------------------------------------------------------------------------------
# SLiM-Quant: histogram scale search + per-tensor symmetric quantisation
# --------------------------------------------------------------------------------------
def num_bins_for(numel):
    """slim/quantization/quantization.py:187."""
    return max(512, min(numel // 1000, 20000))


def histogram_density(absw_f32, num_bins):
    """The density histogram of slim/quantization/quantization.py:36 (range [min, max])."""
    pdf, edges = torch.histogram(absw_f32.flatten(), bins=num_bins, density=True)
    return pdf, edges


def compute_average_error(pdf, val, index, q):
    """slim/quantization/quantization.py:6-20."""
    alpha = val[index]
    dx = val[1] - val[0]
    step = alpha / (2 ** q)
    qv = (val[:index] // step) * step
    quantization_loss = torch.sum(pdf[:index] * (val[:index] - qv) ** 2) * dx
    clipping_loss = torch.sum(pdf[index:] * (val[index:] - alpha) ** 2) * dx
    return quantization_loss + clipping_loss


def find_optimal_quantization_cap(mat, num_bits, num_bins, return_trace=False):
    """slim/quantization/quantization.py:34-62 (integrate=True branch).

    Coarse scan at range(0, bins, bins // 10) into 11 slots (unused slots stay inf with
    index 0), then a fine scan of [indices[tp-1], indices[min(tp+1, 10)]).
    """
    pdf, edges = histogram_density(mat.data.abs().float(), num_bins)
    val = (edges[:-1] + edges[1:]) / 2
    total_loss = torch.zeros(num_bins) + torch.inf
    losses = torch.zeros(11) + torch.inf
    indices = torch.zeros(11).to(torch.int)
    trace = []
    j = 0
    for i in range(0, num_bins, num_bins // 10):
        losses[j] = compute_average_error(pdf, val, i, num_bits)
        indices[j] = i
        trace.append((i, float(losses[j])))
        j += 1
    _, tp = torch.min(losses, 0)
    start = indices[max(tp - 1, 0)]
    end = indices[min(tp + 1, 10)]
    for i in range(start, end):
        total_loss[i] = compute_average_error(pdf, val, i, num_bits)
        trace.append((i, float(total_loss[i])))
    _, idx = torch.min(total_loss, 0)
    if return_trace:
        return val[idx], int(idx), pdf, val, trace
    return val[idx]


def important_columns(W, scaler_row):
    """slim/lora.py:132-136 (and slim/prune.py:331-335): the 1 % of input channels with the largest
    scaler_row * sum_rows |W| (the column sums are a bf16 tensor), as an index tensor in topk's order."""
    metric = scaler_row * W.abs().sum(dim=0)
    return metric.topk(int(0.01 * metric.numel()), largest=True, sorted=False)[1]


def quantize_per_tensor(mat, num_bits, slim_quant=True, abs_max=None, important=None):
    """Quantizer.quantize_block, slim/quantization/quantization.py:173-200.

    Returns (codes int8, scaling_factor reshaped (1, 1)).  `mat * scaling_factor` keeps
    mat's dtype (a 0-d fp32 tensor does not promote bf16), so the product is rounded to
    bf16 before round-half-even.
    important (index tensor, quantize_weight's important_columns): those columns are divided by 1/2 in
    mat's dtype AFTER the clipping threshold has been found on the unscaled matrix (:184-192 with :160-168;
    the reference scales in place and undoes it at :156).
    """
    max_q = 2 ** (num_bits - 1) - 1
    if abs_max is None:
        if slim_quant:
            abs_max = find_optimal_quantization_cap(mat, num_bits, num_bins_for(mat.numel()))
        else:
            abs_max = mat.abs().max()
    if important is not None:
        mat = mat.clone()
        mat[:, important] /= 0.5
    scaling_factor = max_q / abs_max
    q = torch.round((mat * scaling_factor).float())
    q = torch.clamp(q, -max_q - 1, max_q)
    return q.to(torch.int8), scaling_factor.reshape(1, 1)


def dequantize_per_tensor(codes, scaling_factor, important=None):
    """Quantizer.dequantize_absmax, (1, 1) branch, slim/quantization/quantization.py:211-215 (the important
    columns, quantised at twice their size, are halved again)."""
    deq = codes / scaling_factor
    if important is not None:
        deq[:, important] *= 0.5
    return deq


def group_fake_quantize(mat, block, num_bits, important=None):
    """Quantizer.quantize_weight, block branch, followed by dequantize_absmax
    (slim/quantization/quantization.py:139-148, :213-215): the group scales come from the unscaled matrix, the
    important columns are doubled before the codes are taken and halved after dequantisation (bf16)."""
    alphas = group_quant_params(mat, block)
    src = mat
    if important is not None:
        src = mat.clone()
        src[:, important] /= 0.5
    deq = group_dequantize(group_quantize(src, alphas, num_bits), alphas, num_bits, dtype=mat.dtype)
    if important is not None:
        deq[:, important] *= 0.5
    return deq, alphas


# --------------------------------------------------------------------------------------
# Group quantisation (the reference's Triton kernels, slim/quantization/utils.py)
# --------------------------------------------------------------------------------------
def group_quant_params(x, block):
    """compute_quantization_params(x, block, 1, symmetric=True), utils.py:254-271 /
    kernel :216-251: alpha[g, c] = max |x[g*block:(g+1)*block, c]| in x.dtype."""
    M, N = x.shape
    return x.abs().reshape(M // block, block, N).amax(dim=1)


def _maxq(q):
    # utils.py:8-9,20: exp(log(2) * (q - 1)) - 1 evaluated in fp32
    two = torch.tensor(2.0, dtype=torch.float32)
    return float(torch.exp(torch.log(two) * torch.tensor(float(q - 1), dtype=torch.float32)) - 1)


def group_quantize(x, alphas, q):
    """quantize_tensor, utils.py:123-128 / kernel :66-120 / quantize :12-31.

    alpha is cast bf16 -> fp16 (:98); x / alpha * maxq in fp32; clamp; truncate toward
    zero; +-1 when the dropped fraction is strictly beyond 0.5 (round half toward zero).
    """
    M, N = x.shape
    block = M // alphas.shape[0]
    maxq = _maxq(q)
    a = alphas.to(torch.float16).float().repeat_interleave(block, dim=0)
    v = x.float() / a * maxq
    v = torch.clamp(v, -(maxq + 1), maxq)
    vi = torch.trunc(v)
    up = ((v - vi) > 0.5).to(torch.int8) - ((v - vi) < -0.5).to(torch.int8)
    return vi.to(torch.int8) + up


def group_dequantize(codes, alphas, q, dtype=torch.bfloat16):
    """dequantize_tensor, utils.py:188-193 / kernel :131-185 / dequantize :34-47.

    fp16(code) / maxq * alpha in fp32, rounded to `dtype` (round-to-nearest-even, as a
    GPU build of the Triton kernel does; the Triton CPU interpreter truncates instead).
    """
    M, N = codes.shape
    block = M // alphas.shape[0]
    maxq = _maxq(q)
    a = alphas.float().repeat_interleave(block, dim=0)
    v = codes.to(torch.float16).float() / maxq * a
    return v.to(dtype)


# --------------------------------------------------------------------------------------
# Low-rank adapters
# --------------------------------------------------------------------------------------
def lora_rank_for(shape, rank_ratio, tile):
    """slim/lora.py:174-181."""
    rank = int(rank_ratio * min(shape))
    if tile is not None:
        res = rank % tile
        if res != 0:
            rank += tile - res
    return rank


def add_lora(W, W_mask, scaler_row, rank_ratio, num_bits=4, tile=None, slim_lora=True, quantize=True,
             scale_important_weights=False, slim_quant=True, block=None, quantize_first=True):
    """slim/lora.py:105-213, branch separate_lora=True, no prune_lora.  block: the tiled group quantiser with
    that tile instead of the per-tensor one.  Returns dict(weight bf16, lora_left [K,r], lora_right [r,N],
    scaling_factor (1,1) of the second quantisation (None when tiled), error_mat fp32, rank)."""
    use_slim = slim_lora and not bool((scaler_row == 0).any())
    sf = None
    important = important_columns(W, scaler_row) if scale_important_weights else None

    def fake_quant(x):
        if block is not None:
            return group_fake_quantize(x, block, num_bits, important)[0], None
        codes, sf_ = quantize_per_tensor(x, num_bits, slim_quant=slim_quant, important=important)
        return dequantize_per_tensor(codes, sf_, important), sf_

    if use_slim:
        s = torch.sqrt(scaler_row.reshape((1, -1)))
        W_metric = W * s
        if quantize:
            src = W if quantize_first else W.masked_fill(W_mask, 0)
            new_w, sf = fake_quant(src)
            new_w = new_w * s
        else:
            new_w = W_metric.clone()
        new_w[W_mask] = 0
        error_mat = W_metric - new_w
    else:
        new_w = W.clone()
        if quantize:
            new_w, sf = fake_quant(new_w)
        new_w[W_mask] = 0
        error_mat = W - new_w
    U, S, V = torch.svd(error_mat.float())
    rank = lora_rank_for(error_mat.shape, rank_ratio, tile)
    L = (torch.diag_embed(S[:rank]).to(torch.bfloat16) @ V[:, :rank].to(torch.bfloat16).T).t()
    R = U[:, :rank].to(torch.bfloat16).t()
    if use_slim:
        denom = torch.sqrt(scaler_row.reshape((1, -1))).to(torch.bfloat16)
        L = L / denom.t()
    low_rank = R.t() @ L.t()
    new_w = W - low_rank
    new_w[W_mask] = 0
    if quantize:
        new_w, sf = fake_quant(new_w)
    return dict(weight=new_w.to(torch.bfloat16).contiguous(), lora_left=L.to(torch.bfloat16).contiguous(),
                lora_right=R.to(torch.bfloat16).contiguous(), scaling_factor=sf, error_mat=error_mat.float(),
                rank=rank, singular_values=S)


def prune_and_optimize_lora(L, R, num_iters=1000, lr_end_factor=1e-4, return_losses=False):
    """slim/lora.py:7-53 without autograd: 2:4 mask over the SIGNED entries of L^T (prune_nm marks the two smallest
    of every four along K, :27), then num_iters Adam steps (betas 0.9 / 0.999, eps 1e-8, lr = 1e6 / min(K, N)^2
    decaying linearly to lr * lr_end_factor, :31-32) on fp32 copies of L [K, r] and R [r, N] for the loss
    ||L R - target||_F / ||target||_F with target = float(bf16 L @ bf16 R) (:25), masked entries of L reset to zero
    after every step (:41).  Returns (L bf16, R bf16, mask[, losses])."""
    target = torch.matmul(L, R).float()
    tnorm = torch.norm(target).item()
    mask = prune_nm(L.t(), 2, 4).t()
    L = L.clone()
    L[mask] = 0
    Lp, Rp = L.float(), R.float()
    lr0 = 1e6 / min(L.shape[0], R.shape[1]) ** 2
    b1, b2, eps = 0.9, 0.999, 1e-8
    mL, vL, mR, vR = (torch.zeros_like(t) for t in (Lp, Lp, Rp, Rp))
    losses = []
    for it in range(num_iters):
        lr = lr0 * (1.0 + (lr_end_factor - 1.0) * min(it, num_iters) / num_iters)
        D = torch.matmul(Lp, Rp) - target
        nrm = torch.norm(D)
        losses.append((nrm / tnorm).item())
        G = D / (nrm * tnorm)                      # d loss / d (L R)
        gL, gR = G @ Rp.t(), Lp.t() @ G
        t = it + 1
        for p_, g_, m_, v_ in ((Lp, gL, mL, vL), (Rp, gR, mR, vR)):
            m_.mul_(b1).add_(g_, alpha=1 - b1)
            v_.mul_(b2).addcmul_(g_, g_, value=1 - b2)
            denom = (v_.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
            p_.addcdiv_(m_, denom, value=-lr / (1 - b1 ** t))
        Lp[mask] = 0
    out = (Lp.to(torch.bfloat16), Rp.to(torch.bfloat16), mask)
    return out + (losses,) if return_losses else out


# --------------------------------------------------------------------------------------
# Input quantisation hooks
# --------------------------------------------------------------------------------------
def quantize_input_fp8(x):
    """8-bit branch of input_quantization_pre_hook, slim/quantization/quantization.py:286-290."""
    max_val = x.abs().max()
    dtype = torch.float8_e4m3fn if max_val < 488 else torch.float8_e5m2
    fmax = torch.finfo(dtype).max
    xq = (x / max_val * (fmax - 1)).to(dtype)
    return xq.to(x.dtype) / (fmax - 1) * max_val


def quantize_input_int(x, num_bits, group_size, eps=1e-4):
    """Quantizer.quantize_input + dequantize_input (symmetric), quantization.py:219-263."""
    shape = x.shape
    mat = x.reshape(-1, group_size) if group_size != -1 else x
    max_q = 2 ** (num_bits - 1) - 1
    rng = mat.abs().max(dim=-1, keepdim=True)[0]
    rng[rng.abs() < (eps * max_q)] = 1.0
    scale = max_q / rng
    q = torch.clamp(torch.round(mat * scale), -max_q - 1, max_q)
    return (q / scale).reshape(shape)


# --------------------------------------------------------------------------------------
# Whole per-linear pipeline (what prune_wanda does for one nn.Linear, recipe flags)
# --------------------------------------------------------------------------------------
def compress_linear(W, scaler_row, rank_ratio=0.1, num_bits=4, tile=256, shift_zero_metrics=True):
    """slim/prune.py:259-327 for one linear in the SLiM recipe: shift_zeros -> saliency ->
    2:4 mask -> add_lora -> sqrt(rank) pre-scaling of the adapters."""
    if shift_zero_metrics:
        scaler_row = shift_zeros(scaler_row)
    mask = prune_nm(saliency(W, scaler_row), 2, 4)
    out = add_lora(W, mask, scaler_row, rank_ratio, num_bits=num_bits, tile=tile)
    r = torch.tensor(out["lora_left"].shape[1])
    out["lora_left"] = out["lora_left"] * torch.sqrt(r)
    out["lora_right"] = out["lora_right"] * torch.sqrt(r)
    out["mask"] = mask
    out["scaler_row"] = scaler_row
    return out
