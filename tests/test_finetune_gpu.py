"""Adapter fine-tuning of a model compressed by slim.prune.prune_and_quantize, on the GPU: the compressed weights stay as
they were (frozen, same values), the adapters learn, the loss on learnable synthetic blocks falls."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("quantized_adapters", [False, True])
def test_fine_tune_after_compression(quantized_adapters):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from transformers import LlamaConfig, LlamaForCausalLM
    import slim.prune as P
    import slim.fine_tune as F
    from slim.lora import quantize_lora
    from slim_amd.data import get_synthetic
    from test_finetune import learnable_blocks
    torch.manual_seed(0)
    vocab, seq = 256, 32
    m = LlamaForCausalLM(LlamaConfig(vocab_size=vocab, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                                     num_attention_heads=4, num_key_value_heads=4, max_position_embeddings=seq))
    m = m.to(torch.bfloat16).cuda().eval()
    calib, _ = get_synthetic(8, 0, seq, vocab)
    saved = P.get_loaders, P.KEEP_ON_DEVICE
    P.get_loaders = lambda *a, **k: (calib, None)
    P.KEEP_ON_DEVICE = True
    try:
        P.prune_and_quantize(m, None, bitwidth=4, slim_quant=True, prune_method="wanda", sparsity_ratio=0.5, sparsity_type="2:4",
                             quantize_weight=True, nsamples=8, lora_rank=0.1, slim_lora=True, quantize_lora=quantized_adapters,
                             lora_tile_size=256 if quantized_adapters else None)
    finally:
        P.get_loaders, P.KEEP_ON_DEVICE = saved
    if quantized_adapters:
        quantize_lora(m, bitwidth=8, lora_tile_size=256)
    lins = {n: mod for n, mod in m.named_modules() if isinstance(mod, torch.nn.Linear) and hasattr(mod, "lora_left")}
    assert len(lins) == 14
    before = {n: (mod.weight.data.clone(), mod.lora_left.data.clone(), mod.lora_right.data.clone()) for n, mod in lins.items()}
    train, evalb = learnable_blocks(96, seq, vocab), learnable_blocks(8, seq, vocab, seed=4)
    saved = F.get_token_blocks, F.LEARNING_RATE
    F.get_token_blocks = lambda *a, **k: (train, evalb)
    F.LEARNING_RATE = 2e-3
    try:
        loss0 = F.evaluate(m, evalb.cuda(), 4, lambda: torch.autocast("cuda", dtype=torch.bfloat16))
        F.fine_tune(m, None, global_batch_size=8, local_batch_size=4)
    finally:
        F.get_token_blocks, F.LEARNING_RATE = saved
    steps, _, loss1 = F.LAST_LOG[-1]
    print(f"quantized adapters {quantized_adapters}: eval loss {loss0:.4f} -> {loss1:.4f} after {steps} updates")
    assert steps == 12 and loss1 < loss0 - 0.2, (loss0, F.LAST_LOG)
    moved = 0
    for n, mod in lins.items():
        w0, l0, r0 = before[n]
        assert not mod.weight.requires_grad and mod.weight.dtype == torch.bfloat16
        assert torch.equal(mod.weight.data, w0), n
        assert mod.lora_left.requires_grad and mod.lora_right.requires_grad
        moved += int(not torch.equal(mod.lora_left.data, l0)) + int(not torch.equal(mod.lora_right.data, r0))
    assert moved == 2 * len(lins)
    # the model still runs outside training (hooks, adapters in place)
    with torch.no_grad():
        out = m(input_ids=evalb[:2].cuda()).logits
    assert bool(torch.isfinite(out.float()).all())
