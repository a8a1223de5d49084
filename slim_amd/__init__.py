"""slim_amd: the SLiM per-layer compression hot path (2:4 Wanda mask, SLiM-Quant, saliency-weighted
low-rank adapters, emulated compressed forward) as hand-written HIP kernels for MI355X behind the
reference's Python API.  `slim_amd.prune.prune_and_quantize` and `slim_amd.lora.quantize_lora` are the
drop-in entry points; the top-level package `slim` re-exports them under the reference's module names.
"""
__all__ = ["ops", "prune", "lora", "eval", "utils", "layerwrapper", "quantization"]
