#!/usr/bin/env python3
"""End-to-end run of slim.prune.prune_and_quantize on a random-init HF model of a given shape, with
synthetic calibration tokens (no network): wall-clock of the whole drop-in call, not only of the kernels.

    python tools/e2e_compress.py --model llama2-7b --nsamples 128 --calib-batch 4
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

SHAPES = {
    "llama2-7b": dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                      num_key_value_heads=32, vocab_size=32000),
    "llama2-13b": dict(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40,
                       num_key_value_heads=40, vocab_size=32000),
    "tiny": dict(hidden_size=512, intermediate_size=1536, num_hidden_layers=2, num_attention_heads=8,
                 num_key_value_heads=8, vocab_size=1000),
}


def run_e2e(model="llama2-7b", layers=0, nsamples=128, seqlen=2048, calib_batch=8, streams=7, pack=False, ppl=False):
    """prune_and_quantize (the drop-in entry) on a random-init model of the given shape; returns the record main() prints."""
    return _run(argparse.Namespace(model=model, layers=layers, nsamples=nsamples, seqlen=seqlen, calib_batch=calib_batch,
                                   streams=streams, pack=pack, ppl=ppl))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama2-7b", choices=sorted(SHAPES))
    ap.add_argument("--layers", type=int, default=0, help="truncate to this many decoder blocks (0 = all)")
    ap.add_argument("--nsamples", type=int, default=128)
    ap.add_argument("--seqlen", type=int, default=2048)
    ap.add_argument("--calib-batch", type=int, default=8)
    ap.add_argument("--streams", type=int, default=7)
    ap.add_argument("--pack", action="store_true", help="afterwards replace the linears by PackedLinear and report sizes / decode latency")
    ap.add_argument("--ppl", action="store_true", help="also evaluate synthetic-token perplexity before/after")
    args = ap.parse_args()
    print(json.dumps(_run(args)))


def _run(args):
    from transformers import LlamaConfig, LlamaForCausalLM
    import slim.prune as P
    from slim.eval import eval_ppl_wikitext
    from slim_amd.data import get_synthetic

    cfg = dict(SHAPES[args.model])
    if args.layers:
        cfg["num_hidden_layers"] = args.layers
    cfg["max_position_embeddings"] = args.seqlen
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    t0 = time.time()
    torch.set_default_dtype(torch.bfloat16)
    with torch.device(dev):
        model = LlamaForCausalLM(LlamaConfig(**cfg))
    torch.set_default_dtype(torch.float32)
    model.eval()
    model.config.max_position_embeddings = args.seqlen
    torch.cuda.synchronize()
    t_build = time.time() - t0
    calib, test = get_synthetic(args.nsamples, 0, args.seqlen, cfg["vocab_size"])
    saved = (P.get_loaders, P.CALIBRATION_BATCH, P.KEEP_ON_DEVICE, P.COMPRESS_STREAMS)
    P.get_loaders = lambda *a, **k: (calib, None)
    P.CALIBRATION_BATCH = args.calib_batch
    P.KEEP_ON_DEVICE = True
    P.COMPRESS_STREAMS = args.streams
    nbytes = sum(2 * m.weight.numel() for blk in model.model.layers for m in blk.modules() if isinstance(m, torch.nn.Linear))
    out = {"model": args.model, "layers": cfg["num_hidden_layers"], "linear_weight_GB": nbytes / 1e9,
           "nsamples": args.nsamples, "seqlen": args.seqlen, "calib_batch": args.calib_batch, "build_s": round(t_build, 2)}
    if args.ppl:
        test.input_ids = test.input_ids[:, :8 * args.seqlen]
        out["ppl_dense_fp32loss"] = eval_ppl_wikitext(model, test, 1, dev, fp32_loss=True)
    # passes over (part of) a block's linears: one per block when every fit was sized well, one more for a block that had to
    # repeat fits (hint of the previous block too low, filter interval too high)
    import slim_amd.prune as _prune
    passes = [0]
    one_pass = _prune.compress_linears_unchecked

    def counted(*a, **k):
        passes[0] += 1
        return one_pass(*a, **k)
    _prune.compress_linears_unchecked = counted
    torch.cuda.synchronize()
    t0 = time.time()
    try:
        P.prune_and_quantize(model, None, bitwidth=4, slim_quant=True, prune_method="wanda", sparsity_ratio=0.5,
                             sparsity_type="2:4", quantize_weight=True, nsamples=args.nsamples, lora_rank=0.1,
                             slim_lora=True, quantize_lora=True, lora_tile_size=256)
    finally:
        _prune.compress_linears_unchecked = one_pass
    torch.cuda.synchronize()
    dt = time.time() - t0
    out["blocks_with_repeated_fits"] = passes[0] - cfg["num_hidden_layers"]
    P.get_loaders, P.CALIBRATION_BATCH, P.KEEP_ON_DEVICE, P.COMPRESS_STREAMS = saved
    out.update(compress_wallclock_s=round(dt, 2), weight_GBps=round(nbytes / 1e9 / dt, 3),
               peak_mem_GB=round(torch.cuda.max_memory_allocated() / 1e9, 1))
    if args.ppl:
        out["ppl_slim_fp32loss"] = eval_ppl_wikitext(model, test, 1, dev, fp32_loss=True)
    if args.pack:
        from slim.lora import quantize_lora
        from slim_amd.packed import pack_model
        quantize_lora(model, bitwidth=4, lora_tile_size=256)
        ids = torch.randint(0, cfg["vocab_size"], (1, 64), device=dev)
        with torch.no_grad():
            ref = model(ids).logits
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3):
                model(ids[:, :1])
            torch.cuda.synchronize()
            out["emulated_1token_forward_ms"] = round((time.time() - t0) / 3 * 1e3, 2)
            dense, packed = pack_model(model)
            got = model(ids).logits
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3):
                model(ids[:, :1])
            torch.cuda.synchronize()
            out["packed_1token_forward_ms"] = round((time.time() - t0) / 3 * 1e3, 2)
        out.update(dense_GB=round(dense / 1e9, 2), packed_GB=round(packed / 1e9, 2),
                   packed_logits_identical_frac=round((got == ref).float().mean().item(), 4),
                   packed_logits_max_abs_diff=round((got.float() - ref.float()).abs().max().item(), 4))
        if args.ppl:
            out["ppl_packed_fp32loss"] = eval_ppl_wikitext(model, test, 1, dev, fp32_loss=True)
    return out


if __name__ == "__main__":
    main()
