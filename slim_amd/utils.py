"""Model traversal helpers and the mask / shift primitives (mirror of reference slim/utils.py)."""
import torch

from . import ops


def get_layers_list(model):
    """Decoder blocks of a HF causal LM (utils.py:44-61): LLaMA-style, OPT-style or GPT-2-style."""
    if hasattr(model, "model"):
        inner = model.model
        if hasattr(inner, "layers"):
            return inner.layers
        if hasattr(inner, "decoder"):
            return inner.decoder.layers
        raise NotImplementedError
    if hasattr(model, "transformer"):
        return model.transformer.h
    raise NotImplementedError


def find_layers(module, layers=(torch.nn.Linear,), name=""):
    """Depth-first {qualified name: module} of the given layer types (utils.py:102-121)."""
    if type(module) in tuple(layers):
        return {name: module}
    found = {}
    for child_name, child in module.named_children():
        found.update(find_layers(child, layers=layers, name=f"{name}.{child_name}" if name else child_name))
    return found


def check_sparsity(model):
    """Fraction of exactly-zero weights over all block linears (utils.py:4-32)."""
    use_cache = model.config.use_cache
    model.config.use_cache = False
    zeros, total = 0, 0
    for i, layer in enumerate(get_layers_list(model)):
        lz, lt = 0, 0
        for lin in find_layers(layer).values():
            W = lin.weight.data
            lz += (W == 0).sum().item()
            lt += W.numel()
        print(f"Layer {i} Sparsity Ratio: {float(lz) / lt:.2f}")
        zeros += lz
        total += lt
    model.config.use_cache = use_cache
    return float(zeros) / total


def report_gpu_memory(message=""):
    torch.cuda.empty_cache()
    print(message, f" - Allocated Memory: {(torch.cuda.memory_allocated() / 1024 / 1024 / 1024):.2f}GB")


def shift_zeros(x):
    """x + min(x with zeros read as 1) (utils.py:64-72), on the GPU."""
    return ops.shift_zeros(x)


def prune_nm(mat, n, m):
    """N:M mask of an arbitrary metric matrix (utils.py:75-88); True = pruned."""
    return ops.prune_nm_f32(mat.float(), n, m)
