"""Group quantisation front-ends (mirror of reference slim/quantization/utils.py launchers
:123-128, :188-193, :254-271), backed by the HIP group-quant kernels instead of Triton."""
import torch

from .. import ops


def compute_quantization_params(x, block_size_row, block_size_col, symmetric=False):
    if not symmetric or block_size_col != 1:
        raise NotImplementedError("only symmetric scales over (block_size_row x 1) column strips are implemented")
    return ops.group_absmax(x.reshape(-1, x.shape[-1]), block_size_row), None


def quantize_tensor(x, alphas, betas, q):
    if betas is not None:
        raise NotImplementedError("asymmetric group quantisation is not implemented")
    return ops.group_quantize(x, alphas, q)


def dequantize_tensor(x, alphas, betas, q, dtype=torch.bfloat16):
    if betas is not None:
        raise NotImplementedError("asymmetric group quantisation is not implemented")
    if dtype != torch.bfloat16:
        raise NotImplementedError("group dequantisation produces bf16")
    return ops.group_dequantize(x, alphas, q)
