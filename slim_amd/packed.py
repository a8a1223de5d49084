"""Inference from packed storage: a compressed nn.Linear (2:4 sparse 4-bit weight + low-rank adapters, as
slim.prune.prune_and_quantize / slim.lora.quantize_lora leave it) is replaced by a module that keeps the
weight at 0.375 byte per element and runs Linear + adapter hook (reference slim/prune.py:303-321) as two
fused launches.  The reference itself only emulates this forward with dense bf16 tensors."""
import math

import torch

from . import ops
from .utils import get_layers_list, find_layers


# From this many rows on, the forward expands the packed weight into a transient dense bf16 tensor (one streaming
# kernel, 2 B written per element) and multiplies with the library GEMM (hipBLASLt through torch.matmul).  Off by
# default: the library's own large-batch kernel (128 x 256 tile, sparse MFMA) is level with or ahead of the dense
# emulated forward from 512 rows on, and the expansion costs 0.1 ms per call on top of the dense product (28672 x 8192:
# 0.75 x the emulated forward at 512-1024 rows).  Storage stays packed either way.
LARGE_BATCH = int(__import__("os").environ.get("SLIM_PACKED_LARGE_BATCH", str(1 << 30)))
# Keep 4-bit group-quantised adapters as codes + scales for the small-batch path (0.5 instead of 2 bytes per
# adapter element in HBM).  Off by default: rebuilding bf16(code / 7 * alpha) in registers costs more VALU time
# than the bytes save on MI355X (28672 x 8192, rank 1024, batch 1: 81 us against 63 us with bf16 operands), so it
# is a memory-footprint option, not a speed one.
Q4_ADAPTERS = __import__("os").environ.get("SLIM_PACKED_Q4_ADAPTERS", "0") == "1"


class PackedLinear(torch.nn.Module):
    """Drop-in for a compressed nn.Linear (inference only).

    Built from the attributes the compression leaves on the layer: weight.data (dequantised values),
    scaling_factor, and optionally lora_left [K, r], lora_right [r, N], lora_rank, lora_quantizer.
    """

    def __init__(self, lin=None):
        super().__init__()
        self._ws = None
        if lin is None:              # filled by from_tensors
            return
        W = lin.weight.data
        if not W.is_cuda or W.dtype != torch.bfloat16:
            raise RuntimeError("PackedLinear needs a bf16 weight on the GPU (slim_amd has no CPU path)")
        sf = getattr(lin, "scaling_factor", None)
        if sf is None:
            raise ValueError("the layer has no scaling_factor: it was not quantised by prune_and_quantize")
        self.out_features, self.in_features = W.shape
        self.sf = sf.detach().to(torch.float32).reshape(1).contiguous()
        self.vals, self.meta = ops.pack24(W.contiguous(), self.sf)
        self.bias = None if lin.bias is None else lin.bias.data.to(torch.bfloat16).contiguous()
        self.rank = 0
        self.a_div = 1.0
        self.LeffT = self.ReffT = None
        self.qL = self.qR = None       # 4-bit adapter operands of the small-batch path
        self.qblock = 0
        if hasattr(lin, "lora_left"):
            r = int(lin.lora_rank)
            sqrt_r = torch.sqrt(lin.lora_rank)        # on the device the reference computes it on (lora_rank's own)
            L, R = lin.lora_left.data, lin.lora_right.data
            q = getattr(lin, "lora_quantizer", None)
            if q is not None:
                # QuantizedMatmul re-quantises the stored adapters on every call (quantization.py:313-317)
                # and the hook divides the activations by sqrt(rank) (prune.py:305-312)
                if not q.block_quantization:
                    raise NotImplementedError("PackedLinear supports group-quantised adapters only")
                L, aL, cL = ops.group_fake_quant(L.contiguous(), q.block_dim, q.num_bits, want_codes=True)
                R, aR, cR = ops.group_fake_quant(R.contiguous(), q.block_dim, q.num_bits, want_codes=True)
                self.a_div = float(sqrt_r)
                if Q4_ADAPTERS and q.num_bits == 4 and q.block_dim % 32 == 0:
                    # codes and scales themselves, laid out for the small-batch kernel (0.5 byte per element)
                    self.qblock = int(q.block_dim)
                    self.qL = ops.pack_q4_operand(cL.t().contiguous(), aL.t().contiguous(), -(-r // 128) * 128)
                    self.qR = ops.pack_q4_operand(cR.t().contiguous(), aR.t().contiguous())
            else:
                # prune.py:317-321 divides the adapters by sqrt(rank) on every call
                sqrt_r = sqrt_r if sqrt_r.dim() == 0 and not sqrt_r.is_cuda else sqrt_r.to(W.device)
                L, R = L / sqrt_r, R / sqrt_r
            r8, r128 = -(-r // 8) * 8, -(-r // 128) * 128
            self.LeffT = torch.zeros(r128, self.in_features, dtype=torch.bfloat16, device=W.device)
            self.LeffT[:r] = L.t()
            self.ReffT = torch.zeros(self.out_features, r8, dtype=torch.bfloat16, device=W.device)
            self.ReffT[:, :r] = R.t()
            self.rank = r
        for hook_id, hook in lin._forward_pre_hooks.items():      # input fake-quantisation hooks carry over
            self._forward_pre_hooks[hook_id] = hook

    TENSORS = ("vals", "meta", "sf", "bias", "LeffT", "ReffT", "qL_codes", "qL_alpha", "qR_codes", "qR_alpha")

    @classmethod
    def from_tensors(cls, tensors, out_features, in_features, rank, a_div, device, qblock=0):
        """Rebuild from the tensors `state()` returns (a checkpoint written by save_packed)."""
        self = cls()
        self.out_features, self.in_features, self.rank, self.a_div = int(out_features), int(in_features), int(rank), float(a_div)
        self.qblock = int(qblock)
        got = {k: (None if tensors.get(k) is None else tensors[k].to(device).contiguous()) for k in cls.TENSORS}
        for k in ("vals", "meta", "sf", "bias", "LeffT", "ReffT"):
            setattr(self, k, got[k])
        self.qL = (got["qL_codes"], got["qL_alpha"]) if got["qL_codes"] is not None else None
        self.qR = (got["qR_codes"], got["qR_alpha"]) if got["qR_codes"] is not None else None
        if self.vals.numel() * 4 != self.out_features * self.in_features or self.meta.numel() * 8 != self.vals.numel() * 4:
            raise ValueError("packed tensors do not match the recorded shape")
        return self

    def state(self):
        out = {k: getattr(self, k) for k in ("vals", "meta", "sf", "bias", "LeffT", "ReffT") if getattr(self, k) is not None}
        if self.qL is not None:
            out.update(qL_codes=self.qL[0], qL_alpha=self.qL[1], qR_codes=self.qR[0], qR_alpha=self.qR[1])
        return out

    def packed_bytes(self):
        """Bytes the small-batch forward streams: packed weight + adapters (4-bit when quantised, else bf16)."""
        n = self.vals.numel() + self.meta.numel()
        if self.qL is not None:
            n += sum(t.numel() * t.element_size() for t in self.qL + self.qR)
        elif self.rank:
            n += 2 * (self.LeffT.numel() + self.ReffT.numel())
        return n

    def dense_weight(self):
        """The dequantised bf16 weight (what the reference keeps in weight.data)."""
        return ops.unpack24(self.vals, self.meta, self.sf, self.out_features, self.in_features)

    def forward(self, x):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        if x2.dtype != torch.bfloat16:
            x2 = x2.to(torch.bfloat16)
        if x2.shape[0] >= LARGE_BATCH:
            y = self._forward_expanded(x2)
        elif self.qL is not None and x2.shape[0] <= 16:
            y = ops.linear24_forward_q4(x2.contiguous(), self.vals, self.meta, self.sf, self.out_features, self.bias,
                                        self.qL, self.qR, self.rank, self.qblock, self.a_div)
        else:
            y = ops.linear24_forward(x2.contiguous(), self.vals, self.meta, self.sf, self.out_features, self.bias,
                                     self.LeffT, self.ReffT, self.rank, self.a_div)
        return y.reshape(*shape[:-1], self.out_features)

    def _forward_expanded(self, x2):
        """Large batches: transient dense weight + library GEMMs, adapter terms as in prune.py:303-321."""
        y = torch.nn.functional.linear(x2, self.dense_weight(), self.bias)
        if self.rank:
            r = self.rank
            xs = x2 if self.a_div == 1.0 else x2 / self.a_div
            t = torch.matmul(xs, self.LeffT[:r].t())
            if self.a_div != 1.0:
                t = t / self.a_div
            y += torch.matmul(t, self.ReffT[:, :r].t())
        return y

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, rank={self.rank}, packed 2:4 int4"


def pack_model(model):
    """Replace every compressed Linear of the decoder blocks by a PackedLinear, in place.  Returns
    (dense bytes, packed bytes) of the replaced layers."""
    dense = packed = 0
    for layer in get_layers_list(model):
        subset = find_layers(layer)
        for name, lin in subset.items():
            parent = layer
            parts = name.split(".")
            for p in parts[:-1]:
                parent = getattr(parent, p)
            pl = PackedLinear(lin)
            setattr(parent, parts[-1], pl)
            dense += lin.weight.numel() * 2
            packed += pl.packed_bytes()
    return dense, packed


FORMAT = "slim_amd.packed24/1"


def _packed_modules(model):
    for i, layer in enumerate(get_layers_list(model)):
        for name, mod in layer.named_modules():
            if isinstance(mod, PackedLinear):
                yield f"{i}.{name}", layer, name, mod


def save_packed(model, path):
    """Write every PackedLinear of `model` to one safetensors file: per layer `<block>.<name>.{vals,meta,sf,bias,
    LeffT,ReffT}` plus the shapes, rank and activation divisor in the file's metadata.  The rest of the model
    (embeddings, norms, head) is not touched -- save it the usual way."""
    import json
    from safetensors.torch import save_file
    tensors, index = {}, {}
    for key, _, _, mod in _packed_modules(model):
        for k, t in mod.state().items():
            tensors[f"{key}.{k}"] = t.detach().cpu().contiguous()
        index[key] = dict(out_features=mod.out_features, in_features=mod.in_features, rank=mod.rank, a_div=mod.a_div,
                          qblock=mod.qblock)
    if not index:
        raise ValueError("the model holds no PackedLinear; call pack_model first")
    save_file(tensors, path, metadata={"format": FORMAT, "layers": json.dumps(index)})
    return index


def load_packed(model, path, device="cuda"):
    """Replace the linears named in a save_packed file by PackedLinear modules built from it (the model needs the
    right architecture, not the right weights).  Returns the number of layers replaced."""
    import json
    from safetensors import safe_open
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = f.metadata() or {}
        if meta.get("format") != FORMAT:
            raise ValueError(f"{path} is not a {FORMAT} file")
        index = json.loads(meta["layers"])
        names = set(f.keys())
        layers = get_layers_list(model)
        for key, info in index.items():
            block, name = key.split(".", 1)
            tensors = {k: f.get_tensor(f"{key}.{k}") for k in PackedLinear.TENSORS if f"{key}.{k}" in names}
            parent = layers[int(block)]
            parts = name.split(".")
            for p in parts[:-1]:
                parent = getattr(parent, p)
            setattr(parent, parts[-1], PackedLinear.from_tensors(tensors, device=device, **info))
    return len(index)
