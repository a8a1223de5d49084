"""Saliency-weighted low-rank adapters (mirror of reference slim/lora.py: add_lora :105-213,
quantize_lora :56-101), with the full SVD replaced by a randomized rank-r fit on MFMA."""
import torch
import tqdm.auto as tqdm

from . import ops
from .quantization.quantization import Quantizer as AutoQuantizer
from .utils import get_layers_list, find_layers

# Randomized-fit parameters: subspace dimension l = rank + max(FIT_MIN_OVERSAMPLE, rank // 4), and the
# number of subspace iterations.  With these the reconstruction error of the adapters stays within
# 1e-3 (relative) of the truncated SVD on SLiM error matrices (tests/test_hip_kernels.py).
FIT_OVERSAMPLE_DIV = 4
FIT_MIN_OVERSAMPLE = 64
FIT_ITERS = 6
FIT_SEED = 0


def lora_rank(shape, rank_ratio, lora_tile_size=None):
    """lora.py:174-181: int(ratio * min(shape)) padded up to a multiple of the tile size."""
    rank = int(rank_ratio * min(shape))
    if lora_tile_size is not None:
        residue = rank % lora_tile_size
        if residue != 0:
            rank = rank + (lora_tile_size - residue)
        assert rank % lora_tile_size == 0
    return rank


def quantize_lora(model, bitwidth=8, lora_tile_size=256):
    """Group-quantise every adapter in place and mark the layers (lora.py:56-101).  One shared
    Quantizer(block_quantization=True) instance is attached to every linear as `lora_quantizer`."""
    quantizer = AutoQuantizer("weight", num_bits=bitwidth, block_quantization=True, block_dim=lora_tile_size)
    layers = get_layers_list(model)
    progress_bar = tqdm.tqdm(range(len(layers)))
    for i in progress_bar:
        subset = find_layers(layers[i])
        for name in subset:
            progress_bar.set_description(f"Layer {i} - Quantizing LoRA for {name}")
            lin = subset[name]
            for attr in ("lora_left", "lora_right"):
                p = getattr(lin, attr)
                if not p.is_cuda:
                    raise RuntimeError("quantize_lora needs the adapters on the GPU (slim_amd has no CPU path); "
                                       "move the model to the device first")
                # same asserts and state as Quantizer.quantize_weight + dequantize_absmax
                q = quantizer.quantize_weight(p.data)
                p.data = quantizer.dequantize_absmax(q).to(lin.weight.dtype)
            lin.lora_quantizer = quantizer


def add_lora(module, W_mask, rank_ratio=0.01, slim_lora=False, activations=None, quantizer=None, prune_lora=False,
             separate_lora=True, lora_tile_size=None, quantize_first=False, scale_important_weights=False):
    """Fit low-rank adapters to the compression error of `module` and re-quantise the residual weight
    (lora.py:105-213).  Sets module.lora_left [K, r], module.lora_right [r, N] and module.weight.data.

    Implemented: quantizer None or per-tensor (SLiM-Quant / abs-max), slim_lora on or off,
    quantize_first=True, separate_lora=True.  Other flag combinations raise NotImplementedError.
    """
    if scale_important_weights:
        raise NotImplementedError("scale_important_weights is not implemented")
    if prune_lora:
        raise NotImplementedError("prune_lora is not implemented")
    if not separate_lora:
        raise NotImplementedError("separate_lora=False is not implemented")
    if quantizer is not None and not quantize_first:
        raise NotImplementedError("quantize_first=False is not implemented")
    if quantizer is not None and quantizer.block_quantization:
        raise NotImplementedError("tiled weight quantisation inside add_lora is not implemented")
    W = module.weight.data
    if W.dtype != torch.bfloat16 or not W.is_cuda:
        raise RuntimeError("add_lora expects a bf16 weight on the GPU")
    W = W.contiguous()
    N, K = W.shape
    bits = quantizer.num_bits if quantizer is not None else 4
    has_zero = getattr(activations, "has_zero", None)       # precomputed by the driver (one sync per block)
    if has_zero is None:
        has_zero = ops.any_zero(activations.scaler_row)
    use_slim = bool(slim_lora) and not has_zero                                 # lora.py:139
    s = ops.sqrt(activations.scaler_row) if use_slim else None

    # first quantisation (scale search on W) and error matrix  (lora.py:146-156 / :158-169)
    sf1 = None
    if quantizer is not None:
        sf1 = _weight_scale(W, quantizer).reshape(1)
    _, E = ops.error_matrix(W, W_mask, s, sf1, bits)
    Et = ops.transpose(E)

    rank = lora_rank((N, K), rank_ratio, lora_tile_size)
    rank = min(rank, min(N, K))      # S[:rank] of a min(N, K)-long spectrum (lora.py:184)
    over = max(FIT_MIN_OVERSAMPLE, rank // FIT_OVERSAMPLE_DIV)
    fit = ops.lowrank_fit(E, Et, rank, over, FIT_ITERS, seed=FIT_SEED, s=s)
    del E, Et
    lora_left, lora_right, U = fit["L"], fit["Ut"], fit["U"]

    # residual weight W - R^T L^T, masked (lora.py:199-201), then second quantisation (:202-204)
    new_weight = ops.lowrank_residual(U, lora_left, W, W_mask)
    if quantizer is not None:
        sf2 = _weight_scale(new_weight, quantizer)
        new_weight = ops.fake_quant_pt(new_weight, sf2.reshape(1), bits)
        quantizer.scaling_factor = (sf2 if quantizer.slim_quant else sf2.to(torch.bfloat16)).reshape(1, 1)
    module.lora_left = torch.nn.Parameter(lora_left).to(torch.bfloat16).contiguous()
    module.lora_right = torch.nn.Parameter(lora_right).to(torch.bfloat16).contiguous()
    module.weight.data = new_weight.to(torch.bfloat16).contiguous()
    module.lora_fit_info = fit["info"]


def _weight_scale(W, quantizer):
    """Per-tensor scaling factor of `W` under the quantizer's policy, as a 0-d fp32 device tensor."""
    hist = ops.hist_abs(W)
    if quantizer.slim_quant:
        bins = max(512, min(W.numel() // 1000, 20000))
        _, _, sf, _ = ops.quant_search(hist, bins, quantizer.num_bits)
    else:
        _, sf = ops.absmax_scale(hist, quantizer.num_bits)
    return sf
