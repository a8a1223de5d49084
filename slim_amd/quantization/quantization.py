"""SLiM-Quant and the quantisation hooks (mirror of reference slim/quantization/quantization.py)."""
import torch

from .. import ops
from ..utils import get_layers_list, find_layers
from .utils import compute_quantization_params, quantize_tensor, dequantize_tensor


def find_optimal_quantiztion_cap(mat, num_bits=8, num_bins=4096, integrate=True):
    """Clipping threshold minimising the histogram-integrated quantisation + clipping error
    (quantization.py:34-62).  Returns a 0-d fp32 tensor on mat's device; no host synchronisation."""
    if not integrate:
        raise NotImplementedError("the brute-force (integrate=False) search is not implemented")
    if mat.dtype != torch.bfloat16:
        raise NotImplementedError("the scale search runs on bf16 weights")
    hist = ops.hist_abs(mat.data)
    alpha, _, _, _ = ops.quant_search(hist, num_bins, num_bits)
    return alpha


class Quantizer:
    """Same constructor and state as the reference Quantizer (quantization.py:88-108): the instance is
    shared across layers and `scaling_factor` / `dtype` are overwritten by every quantize_weight call."""

    def __init__(self, matrix_type, num_bits=8, group_size=-1, symmetric=True, eps=1e-4, slim_quant=False,
                 block_quantization=False, block_dim=16):
        self.matrix_type = matrix_type
        self.num_bits = num_bits
        self.group_size = group_size
        self.symmetric = symmetric
        self.eps = eps
        self.slim_quant = slim_quant
        self.block_quantization = block_quantization
        self.block_dim = block_dim
        self.important_columns_scaling_factor = (1 / 2.)

    def quantize(self, mat, num_bits=8):
        if self.matrix_type == "weight":
            return self.quantize_weight(mat)
        elif self.matrix_type == "input":
            return self.quantize_input(mat)

    def get_dtype(self, num_bits):
        return torch.int8 if num_bits <= 8 else (torch.int16 if num_bits <= 16 else torch.int32)

    def quantize_weight(self, mat, important_columns=None):
        """quantization.py:133-157.  Group branch: one bf16 scale per (block_dim rows x 1 column).
        Per-tensor branch: SLiM-Quant (histogram search) or abs-max scale, int8 codes.
        important_columns (index tensor, or a bool [K] tensor of flags): the scale(s) come from `mat` as it is, the
        codes from `mat` with those columns divided by important_columns_scaling_factor (:147, :192); dequantize_absmax
        multiplies them back (:215).  The reference scales `mat` in place and undoes it (:156); here `mat` is not
        touched."""
        self.important_columns = self._column_flags(mat, important_columns)
        if self.num_bits > 8:
            raise NotImplementedError("weight quantisation above 8 bits is not implemented")
        if self.block_quantization:
            assert mat.shape[0] % self.block_dim == 0 and mat.shape[1] % self.block_dim == 0, \
                "Input matrix size is not divisible by block size"
            if self.slim_quant:
                raise NotImplementedError("SLiM-Quant is not supported for block quantization")
            self.dtype = mat.dtype
            self.scaling_factor, _ = compute_quantization_params(mat, self.block_dim, 1, symmetric=True)
            return quantize_tensor(self._scaled(mat), self.scaling_factor, None, self.num_bits)
        quantized_mat, scaling_factor = self.quantize_block(mat, self.num_bits, self.slim_quant)
        self.scaling_factor = scaling_factor.reshape(1, 1)
        return quantized_mat

    @staticmethod
    def _column_flags(mat, important_columns):
        if important_columns is None:
            return None
        if important_columns.dtype == torch.bool:
            return important_columns.to(mat.device)
        flags = torch.zeros(mat.shape[1], dtype=torch.bool, device=mat.device)
        flags[important_columns.to(mat.device).long()] = True
        return flags

    def _scaled(self, mat):
        """`mat` with the important columns divided by their factor (quantization.py:168), a copy."""
        if getattr(self, "important_columns", None) is None:
            return mat
        return ops.scale_columns(mat, self.important_columns, 1.0 / self.important_columns_scaling_factor)

    def quantize_block(self, mat, num_bits=8, slim_quant=False, important_columns=None):
        """quantization.py:173-200: (int8 codes, 0-d scaling factor maxq / alpha)."""
        if mat.dtype != torch.bfloat16:
            raise NotImplementedError("per-tensor weight quantisation runs on bf16 weights")
        hist = ops.hist_abs(mat)
        if slim_quant:
            bins = max(512, min(torch.numel(mat) // 1000, 20000))
            _, _, sf, _ = ops.quant_search(hist, bins, num_bits)
        else:
            _, sf = ops.absmax_scale(hist, num_bits)
        codes = ops.quantize_pt(self._scaled(mat), sf.reshape(1), num_bits)
        # abs-max scales are bf16 in the reference (bf16 max -> bf16 quotient); SLiM-Quant scales are fp32
        return codes, (sf if slim_quant else sf.to(torch.bfloat16))

    def dequantize_absmax(self, quantized_mat, scaling_factor=None):
        """quantization.py:203-217."""
        if scaling_factor is None:
            scaling_factor = self.scaling_factor
        flags = getattr(self, "important_columns", None)
        if scaling_factor.shape == (1, 1):
            as_bf16 = scaling_factor.dtype == torch.bfloat16
            deq = ops.dequantize_pt(quantized_mat, scaling_factor.float().reshape(1), out_bf16=as_bf16)
            if flags is not None:
                # deq_mat[:, cols] *= 1/2 (:215): exact in either dtype
                if as_bf16:
                    ops.scale_columns(deq, flags, self.important_columns_scaling_factor, out=deq)
                else:
                    ops.scale_columns_f32_(deq, flags, self.important_columns_scaling_factor)
            return deq
        deq = dequantize_tensor(quantized_mat, scaling_factor, None, self.num_bits, dtype=self.dtype)
        if flags is not None:
            ops.scale_columns(deq, flags, self.important_columns_scaling_factor, out=deq)
        return deq

    def quantize_input(self, mat):
        """quantization.py:219-251 fused with dequantize_input :253-263: returns the DEQUANTISED
        activations (the only way the reference's hooks use the pair)."""
        if not self.symmetric:
            raise NotImplementedError("asymmetric input quantisation is not implemented")
        self._fake_quantized = True
        return ops.input_fake_quant_int(mat, self.num_bits, self.group_size, self.eps)

    def dequantize_input(self, mat):
        return mat


def attach_input_quantization_hooks(model, num_bits=8, input_group_size=-1, bitwidth=None):
    """Forward pre-hooks that fake-quantise the input of every block linear (quantization.py:266-304).
    8 bits: per-tensor FP8 emulation; otherwise symmetric integer group quantisation.
    `bitwidth` is accepted as an alias of num_bits (the reference README uses that name)."""
    if bitwidth is not None:
        num_bits = bitwidth

    def input_quantization_pre_hook(module, input):
        if module.quantize_input:
            x = input[0]
            if x.dtype != torch.bfloat16:
                raise TypeError("input quantisation hooks expect bf16 activations")
            if num_bits != 8:
                deq = ops.input_fake_quant_int(x, num_bits, input_group_size, module.quantizer.eps)
            else:
                deq = ops.input_fake_quant_fp8(x)
            input[0].data = deq

    for layer in get_layers_list(model):
        for lin in find_layers(layer).values():
            lin.register_forward_pre_hook(input_quantization_pre_hook)
            lin.quantize_input = True
            if num_bits != 8:
                lin.input_group_size = input_group_size
                lin.quantizer = Quantizer("input", num_bits=num_bits, group_size=input_group_size)


class QuantizedMatmul(torch.autograd.Function):
    """input @ deq(quant(weight)) with the weight re-quantised on every call (quantization.py:307-328).
    The backward is that of the reference (straight-through to the dequantised weight)."""

    @staticmethod
    def forward(ctx, input, weight, quantizer):
        if quantizer is not None:
            if quantizer.block_quantization:
                dequantized_weight = ops.group_fake_quant(weight.data, quantizer.block_dim, quantizer.num_bits)
            else:
                dequantized_weight = quantizer.dequantize_absmax(quantizer.quantize_weight(weight.data))
        else:
            dequantized_weight = weight
        ctx.save_for_backward(input, dequantized_weight)
        return torch.matmul(input, dequantized_weight)

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        grad_input = torch.matmul(grad_output, weight.t())
        grad_weight = torch.matmul(input.view(-1, input.shape[-1]).t(), grad_output.view(-1, grad_output.shape[-1]))
        return grad_input, grad_weight, None
