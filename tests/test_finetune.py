"""Adapter fine-tuning (slim_amd/fine_tune.py): the two helpers against the reference's recorded behaviour
(tests/golden/g14_finetune.npz, made by make_golden_finetune.py), the training loop on a tiny model, and two ranks over
gloo against one rank on the same global batches."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))


def _bf16(bits):
    return torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16)


@pytest.mark.parametrize("fp16_container", [True, False])
def test_helpers_match_reference_record(fp16_container, monkeypatch):
    """fp16_container=True: exactly what the reference leaves, dtypes included.  False (this build's default): the frozen
    tensors keep their dtype (see fine_tune.FROZEN_WEIGHT_FP16), every value and flag as recorded."""
    from make_golden_finetune import build
    import slim_amd.fine_tune as FT
    from slim_amd.fine_tune import disable_linear_layer_grads, requantize
    monkeypatch.setattr(FT, "FROZEN_WEIGHT_FP16", fp16_container)
    d = np.load(os.path.join(HERE, "golden", "g14_finetune.npz"), allow_pickle=False)
    m, g = build()
    disable_linear_layer_grads(m)

    def dtype_ok(got, recorded, frozen):
        return str(got) == (str(recorded) if (fp16_container or not frozen) else "torch.bfloat16")
    for name, mod in m.named_children():
        frozen = hasattr(mod, "lora_left")
        assert mod.weight.requires_grad == bool(d[f"{name}/weight_requires_grad"])
        assert dtype_ok(mod.weight.dtype, d[f"{name}/weight_dtype"], frozen)
        if mod.bias is not None:
            assert mod.bias.requires_grad == bool(d[f"{name}/bias_requires_grad"])
            assert dtype_ok(mod.bias.dtype, d[f"{name}/bias_dtype"], frozen)
        assert hasattr(mod, "weight_mask") == bool(d[f"{name}/has_weight_mask"])
        if hasattr(mod, "weight_mask"):
            assert np.array_equal(mod.weight_mask.numpy(), d[f"{name}/weight_mask"])
        assert len(mod._forward_pre_hooks) == int(d[f"{name}/n_pre_hooks"])
    for name, mod in m.named_children():
        drift = torch.from_numpy(d[f"{name}/drift"])
        if not hasattr(mod, "lora_left"):
            mod.weight.data = (mod.weight.data.float() + drift).to(mod.weight.dtype)
    m.b.lora_left.data = m.b.lora_left.data + 1.0
    for name, mod in m.named_children():
        for hook in mod._forward_pre_hooks.values():
            hook(mod, (None,))
        assert torch.equal(mod.weight.data.float().to(torch.bfloat16), _bf16(d[f"{name}/weight_after_hooks_bf16"])), name
    assert torch.equal(m.b.lora_left.data.to(torch.bfloat16), _bf16(d["b/lora_left_after_hooks_bf16"]))
    requantize(m)
    for name, mod in m.named_children():
        assert torch.equal(mod.weight.data.float().to(torch.bfloat16), _bf16(d[f"{name}/weight_after_requantize_bf16"])), name
        assert dtype_ok(mod.weight.dtype, d[f"{name}/weight_dtype_after_requantize"], hasattr(mod, "lora_left"))
    e = torch.nn.Sequential()
    lin = torch.nn.Linear(8, 8, bias=False)
    lin.scaling_factor = None
    e.add_module("x", lin)
    for fn, key in ((disable_linear_layer_grads, "err_disable"), (requantize, "err_requantize")):
        with pytest.raises(NotImplementedError) as ex:
            fn(e)
        assert f"NotImplementedError: {ex.value}" == str(d[key])


def test_group_texts_and_block_size():
    from slim_amd.fine_tune import group_texts, pick_block_size
    ex = {"input_ids": [[1, 2, 3], [4, 5, 6, 7], [8]], "attention_mask": [[1, 1, 1], [1, 1, 1, 1], [1]]}
    out = group_texts(ex, 3)
    assert out["input_ids"] == [[1, 2, 3], [4, 5, 6]] and out["labels"] == out["input_ids"]      # remainder dropped
    assert group_texts({"input_ids": [[1, 2]]}, 3)["input_ids"] == [[1, 2]]                       # shorter than a block: kept
    assert pick_block_size(None, 4096) == 1024 and pick_block_size(None, 512) == 512
    assert pick_block_size(2048, 1024) == 1024 and pick_block_size(256, 1024) == 256


def test_lr_schedule():
    from slim_amd.fine_tune import lr_factor
    assert lr_factor(0, 100) == 0.0 and lr_factor(5, 100) == 1.0
    assert abs(lr_factor(52, 100) - (100 - 52) / 95) < 1e-12 and lr_factor(100, 100) == 0.0


def tiny_model(seed=0, sparse=True):
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(seed)
    m = LlamaForCausalLM(LlamaConfig(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                                     num_attention_heads=4, num_key_value_heads=4, max_position_embeddings=16))
    if sparse:
        g = torch.Generator().manual_seed(seed + 1)
        for blk in m.model.layers:
            for mod in blk.modules():
                if isinstance(mod, torch.nn.Linear):
                    mod.weight.data[torch.rand(mod.weight.shape, generator=g) < 0.5] = 0
    return m


def learnable_blocks(n, block=16, vocab=64, seed=3):
    """Sequences that count upwards from a random start: next-token prediction is learnable in a few updates."""
    g = torch.Generator().manual_seed(seed)
    start = torch.randint(0, vocab, (n, 1), generator=g)
    return (start + torch.arange(block).reshape(1, -1)) % vocab


def test_loop_learns_and_keeps_the_zero_pattern():
    import slim_amd.fine_tune as F
    m = tiny_model()
    zeros = {n: (p.data == 0) for n, p in m.named_parameters() if "layers" in n and p.dim() == 2}
    saved = F.get_token_blocks, F.LEARNING_RATE
    F.get_token_blocks = lambda *a, **k: (learnable_blocks(64), learnable_blocks(8, seed=4))
    F.LEARNING_RATE = 5e-3
    try:
        before = F.evaluate(m.float(), learnable_blocks(8, seed=4), 4, __import__("contextlib").nullcontext)
        F.fine_tune(m, None, global_batch_size=4, local_batch_size=2)
    finally:
        F.get_token_blocks, F.LEARNING_RATE = saved
    steps, train_loss, eval_loss = F.LAST_LOG[-1]
    assert steps == 16 and eval_loss < before - 0.3, (before, F.LAST_LOG)
    for n, p in m.named_parameters():
        if n in zeros:
            # the pre-hooks zero the pattern before every forward, the last update may have moved it again; one forward restores it
            pass
    m(input_ids=learnable_blocks(1))
    for n, p in m.named_parameters():
        if n in zeros:
            assert bool((p.data[zeros[n]] == 0).all()), n
    assert not m.training


WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch.distributed as dist
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
import slim_amd.fine_tune as F
from test_finetune import tiny_model, learnable_blocks
m = tiny_model()
m.model.never_reached = torch.nn.Parameter(torch.ones(4))     # trainable, in no forward: its grad stays None on every rank
F.get_token_blocks = lambda *a, **k: (learnable_blocks(32), learnable_blocks(4, seed=4))
F.LEARNING_RATE = 5e-3
F.fine_tune(m, None, global_batch_size=int(sys.argv[2]), local_batch_size=1)
rank = dist.get_rank() if world > 1 else 0
torch.save({n: p.data for n, p in m.named_parameters()}, sys.argv[3] + f".{rank}")
'''


def test_two_ranks_equal_one_rank(tmp_path):
    """Two ranks with 4 micro-batches each per update against one rank with 8: the same samples per update, gradients
    averaged over the ranks, so the same parameters up to fp32 summation order."""
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    subprocess.run([sys.executable, str(worker), ROOT, "8", str(tmp_path / "one")], check=True, env=env, timeout=600)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                    "--master-port", "29641", str(worker), ROOT, "4", str(tmp_path / "two")], check=True, env=env, timeout=600)
    one = torch.load(str(tmp_path / "one.0"))
    two = [torch.load(str(tmp_path / f"two.{r}")) for r in range(2)]
    assert torch.equal(two[0]["model.never_reached"], torch.ones(4))
    for n in one:
        assert torch.equal(two[0][n], two[1][n]), n                       # both ranks hold the same model
        assert float((one[n] - two[0][n]).abs().max()) <= 2e-4 * max(1.0, float(one[n].abs().max())), n
