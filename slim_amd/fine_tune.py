"""Adapter fine-tuning after compression (mirror of reference slim/fine_tune.py: fine_tune :148-422,
disable_linear_layer_grads :24-48, requantize :51-67).

The contract is the reference's: with adapters the compressed weights (and biases) are frozen and only what is left
trainable (adapters, embeddings, norms, un-compressed linears under their zero-pattern mask) is updated; without adapters
the weights themselves are trained under their zero-pattern mask and snapped back to their quantisation grid at the end.
The mechanism is this build's: the reference hands the model to the HuggingFace Trainer (one process, DataParallel /
accelerate around it); here the loop is a plain one -- token blocks resident in HBM, bf16 autocast, gradient
checkpointing, gradient accumulation to the global batch, AdamW with the Trainer's defaults for the arguments the
reference sets (learning rate 5e-5, linear decay after 5 warm-up steps, gradient clipping at 1.0, one epoch, seed 42) --
and one process per GPU when torch.distributed is initialised: every rank takes its share of each global batch and the
gradients are summed in flat buckets with one all-reduce per bucket (RCCL over xGMI), the same update on every rank.

Data.  The reference tokenises a C4 shard it downloads (or finds as `<cache_dir>/c4-raw.pt`); machines without network
access can provide that directory, or replace `get_token_blocks` (same pattern as slim_amd.prune.get_loaders).
"""
import os
from itertools import chain

import torch

KNOWN_MODULES = {"Linear", "Conv1d"}
# the Trainer defaults behind the reference's TrainingArguments (fine_tune.py:197-219)
LEARNING_RATE = 5e-5
WARMUP_STEPS = 5
MAX_GRAD_NORM = 1.0
SEED = 42
LOGGING_STEPS = 100
EVAL_STEPS = 100
# gradients are summed over the ranks in flat buckets of about this many bytes
BUCKET_BYTES = 256 << 20
# filled by fine_tune: [(step, training loss averaged since the last entry, evaluation loss or None)]
LAST_LOG = []


def _mask_lora(module, inputs):
    module.lora_left.data[module.lora_left_mask] = 0


def _mask_weight(module, inputs):
    module.weight.data[module.weight_mask] = 0


# The reference re-stores the frozen weights and biases of the adapter layers as fp16 (fine_tune.py:31-33).  For the compressed
# values (bf16 numbers of ordinary magnitude) that changes the container, not the values -- but a bf16 model then only runs
# under autocast: outside it F.linear refuses bf16 activations with an fp16 weight, so the reference's own evaluation after
# fine_tune (main.py:191-197) cannot run on a bf16 model.  Here the frozen tensors keep their dtype unless this is set.
FROZEN_WEIGHT_FP16 = os.environ.get("SLIM_FINETUNE_FP16_WEIGHTS", "0") == "1"


def disable_linear_layer_grads(model):
    """fine_tune.py:24-48.  Linear / Conv1d modules with adapters: weight and bias frozen (as fp16 only with
    FROZEN_WEIGHT_FP16), pruned adapters (lora_left_mask) re-zeroed before every forward.  Modules without adapters: the
    weight stays trainable under the mask of its current zeros; a tiled-quantised weight (scaling_factor None) is refused.
    The mask lives on the weight's device (the reference keeps it on the CPU and indexes the device tensor with it)."""
    for _, module in model.named_modules():
        if type(module).__name__ not in KNOWN_MODULES:
            continue
        if hasattr(module, "lora_left"):
            module.weight.requires_grad = False
            if FROZEN_WEIGHT_FP16:
                module.weight.data = module.weight.half()
            if module.bias is not None:
                if FROZEN_WEIGHT_FP16:
                    module.bias.data = module.bias.half()
                module.bias.requires_grad = False
            if hasattr(module, "lora_left_mask"):
                module.register_forward_pre_hook(_mask_lora)
        else:
            module.weight.requires_grad = True
            module.weight_mask = (module.weight.data.clone().detach() == 0)
            module.register_forward_pre_hook(_mask_weight)
            if hasattr(module, "scaling_factor") and module.scaling_factor is None:
                raise NotImplementedError("Group quantization is not supported for fine-tuning")


def requantize(model, q=4):
    """fine_tune.py:51-67: weights that were trained (modules without adapters) go back to their grid,
    round((w * scaling_factor).float()) / scaling_factor, without clamping (`q` is unused there too).  The product keeps the
    dtype torch's promotion gives it, as in the reference: fp32 with a SLiM-Quant scale (fp32 (1, 1)), but bf16 with an abs-max
    scale (bf16 (1, 1)) under a bf16 weight -- it is then rounded to bf16 before it is rounded to an integer.  A one-off
    elementwise pass in torch on the weight's device."""
    for _, module in model.named_modules():
        if type(module).__name__ not in KNOWN_MODULES or hasattr(module, "lora_left") or not hasattr(module, "scaling_factor"):
            continue
        if module.scaling_factor is None:
            raise NotImplementedError("Group quantization is not supported for fine-tuning")
        sf = module.scaling_factor.to(device=module.weight.device)
        w = torch.round((module.weight.data * sf).float())
        module.weight.data = (w / sf).to(module.weight.dtype)


def group_texts(examples, block_size):
    """fine_tune.py:321-335: concatenate the tokenised texts of a batch and cut them into blocks (the remainder is dropped)."""
    concatenated = {k: list(chain(*examples[k])) for k in examples.keys()}
    total = len(concatenated[list(examples.keys())[0]])
    if total >= block_size:
        total = (total // block_size) * block_size
    result = {k: [t[i:i + block_size] for i in range(0, total, block_size)] for k, t in concatenated.items()}
    result["labels"] = result["input_ids"].copy()
    return result


def pick_block_size(block_size, model_max_length):
    """fine_tune.py:303-318."""
    if block_size is None:
        block_size = model_max_length
        if block_size > 1024:
            print("The chosen tokenizer supports a `model_max_length` that is longer than the default `block_size` value"
                  " of 1024. If you would like to use a longer `block_size` up to `tokenizer.model_max_length` you can"
                  " override this default with `--block_size xxx`.")
            block_size = 1024
        return block_size
    if block_size > model_max_length:
        print(f"The block_size passed ({block_size}) is larger than the maximum length for the model"
              f"({model_max_length}). Using block_size={model_max_length}.")
    return min(block_size, model_max_length)


def get_token_blocks(tokenizer, dataset_name="c4", dataset_config_name=None, validation_split_percentage=5, streaming=False,
                     preprocessing_num_workers=None, overwrite_cache=False, block_size=None, max_train_samples=30000,
                     max_eval_samples=128, cache_dir="data"):
    """(train ids int64 [n_train, block], eval ids int64 [n_eval, block]) as fine_tune.py:221-373 builds them: the raw C4
    shard saved under `<cache_dir>/c4-raw.pt` (the reference writes it there after its first download), tokenised, concatenated
    and cut into blocks, truncated to max_*_samples.  Without that directory the data would have to be downloaded, which this
    build does not assume."""
    path = f"{cache_dir}/c4-raw.pt"
    if not os.path.exists(path):
        raise NotImplementedError(
            f"'{path}' not found: the reference downloads the C4 shards en/c4-train.00000-of-01024 and "
            "en/c4-validation.00000-of-00008 there; without network access provide that directory (datasets.save_to_disk) "
            "or replace slim_amd.fine_tune.get_token_blocks")
    from datasets import load_from_disk
    raw = load_from_disk(path)
    columns = list(raw["train"].features)
    text = "text" if "text" in columns else columns[0]
    kw = {} if streaming else dict(num_proc=preprocessing_num_workers, load_from_cache_file=not overwrite_cache)
    tokenized = raw.map(lambda ex: tokenizer(ex[text]), batched=True, remove_columns=columns, **kw)
    block = pick_block_size(block_size, tokenizer.model_max_length)
    lm = tokenized.map(lambda ex: group_texts(ex, block), batched=True, **kw)
    if "validation" not in lm:
        raise ValueError("--do_eval requires a validation dataset")
    out = []
    for split, cap in (("train", max_train_samples), ("validation", max_eval_samples)):
        ds = lm[split]
        n = len(ds) if cap is None else min(len(ds), cap)
        out.append(torch.tensor(ds.select(range(n))["input_ids"], dtype=torch.int64))
    return out[0], out[1]


def make_optimizer(name, params):
    """The `optim` names of the Trainer this build covers."""
    if name in ("adamw_torch", "adamw_hf", "adamw_torch_fused"):
        return torch.optim.AdamW(params, lr=LEARNING_RATE, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                                 fused=(name == "adamw_torch_fused") or None)
    if name == "sgd":
        return torch.optim.SGD(params, lr=LEARNING_RATE)
    if name == "adagrad":
        return torch.optim.Adagrad(params, lr=LEARNING_RATE)
    if name == "adafactor":
        from transformers.optimization import Adafactor
        return Adafactor(params, lr=LEARNING_RATE, scale_parameter=False, relative_step=False)
    raise NotImplementedError(f"optimizer '{name}' is not implemented in slim_amd.fine_tune")


def lr_factor(step, total_steps, warmup=WARMUP_STEPS):
    """Linear warm-up then linear decay to zero (the Trainer's default schedule); `step` counts finished updates."""
    if step < warmup:
        return step / max(1, warmup)
    return max(0.0, (total_steps - step) / max(1, total_steps - warmup))


def all_reduce_grads(params, world):
    """Sum the gradients over the ranks and divide by their number: flat buckets of one dtype, one all-reduce each."""
    import torch.distributed as dist
    buckets, size = {}, {}
    params = list(params)
    # a parameter no rank's batches reached keeps grad None, as on one rank, where the optimizer then skips it (a zero
    # gradient would still move it on stale momentum): one flag per parameter, summed over the ranks
    has = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], device=params[0].device if params else None)
    if params:
        dist.all_reduce(has)
    reached = has.tolist()
    for p, n in zip(params, reached):
        if n == 0:
            continue
        if p.grad is None:
            p.grad = torch.zeros_like(p)          # a parameter this rank's batches did not reach, but another's did
        key = (p.grad.dtype, p.grad.device)
        if size.get(key, 0) + p.grad.numel() * p.grad.element_size() > BUCKET_BYTES and buckets.get(key):
            _reduce_bucket(buckets.pop(key), world, dist)
            size[key] = 0
        buckets.setdefault(key, []).append(p.grad)
        size[key] = size.get(key, 0) + p.grad.numel() * p.grad.element_size()
    for grads in buckets.values():
        _reduce_bucket(grads, world, dist)


def _reduce_bucket(grads, world, dist):
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    flat /= world
    o = 0
    for g in grads:
        g.copy_(flat[o:o + g.numel()].view_as(g))
        o += g.numel()


def _loss(model, ids, autocast):
    with autocast():
        return model(input_ids=ids, labels=ids).loss


def evaluate(model, eval_ids, batch, autocast):
    was_training = model.training
    model.eval()
    total, n = 0.0, 0
    with torch.no_grad():
        for i in range(0, eval_ids.shape[0], batch):
            ids = eval_ids[i:i + batch]
            total += float(_loss(model, ids, autocast)) * ids.shape[0]
            n += ids.shape[0]
    model.train(was_training)
    return total / max(n, 1)


def fine_tune(model, tokenizer, dataset_name="c4", dataset_config_name=None, validation_split_percentage=5, streaming=False,
              preprocessing_num_workers=None, overwrite_cache=False, block_size=None, max_train_samples=30000,
              max_eval_samples=128, cache_dir="data", optimizer="adamw_torch", global_batch_size=64, local_batch_size=1):
    """Fine-tune a compressed model for one epoch (fine_tune.py:148-422); same arguments, returns None, the model is updated
    in place.  With adapters the compressed weights stay frozen and the adapters (and whatever else is trainable) learn;
    otherwise the weights learn under their zero pattern and are re-quantised at the end."""
    import contextlib
    import torch.distributed as dist
    device = next(model.parameters()).device
    bf16 = device.type == "cuda" and torch.cuda.is_bf16_supported()
    model = model.to(torch.bfloat16) if bf16 else model.float()              # fine_tune.py:192-196
    autocast = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if bf16 else contextlib.nullcontext
    train_ids, eval_ids = get_token_blocks(tokenizer, dataset_name, dataset_config_name, validation_split_percentage, streaming,
                                           preprocessing_num_workers, overwrite_cache, block_size, max_train_samples,
                                           max_eval_samples, cache_dir)
    train_ids, eval_ids = train_ids.to(device), eval_ids.to(device)          # resident for the whole run
    disable_linear_layer_grads(model)
    model.config.use_cache = False
    model.train()
    if hasattr(model, "gradient_checkpointing_enable"):
        model.gradient_checkpointing_enable()
        if hasattr(model, "enable_input_require_grads"):
            model.enable_input_require_grads()
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    accum = max(1, global_batch_size // local_batch_size)                    # fine_tune.py:213 (per process, as there)
    per_update = local_batch_size * accum * world
    n_train = train_ids.shape[0]
    total_steps = max(1, n_train // per_update) if n_train >= per_update else 1
    params = [p for p in model.parameters() if p.requires_grad]
    opt = make_optimizer(optimizer, params)
    order = torch.randperm(n_train, generator=torch.Generator().manual_seed(SEED)).to(device)
    del LAST_LOG[:]
    running, since = 0.0, 0
    for step in range(total_steps):
        for group in opt.param_groups:
            group["lr"] = LEARNING_RATE * lr_factor(step, total_steps)
        base = step * per_update
        for a in range(accum):
            # micro-batch `a` of this update: consecutive ranks take consecutive slices
            lo = base + (a * world + rank) * local_batch_size
            idx = order[lo:lo + local_batch_size]
            if idx.numel() == 0:
                continue
            loss = _loss(model, train_ids[idx], autocast) / accum
            loss.backward()
            running += float(loss.detach())
        if world > 1:
            all_reduce_grads(params, world)
        torch.nn.utils.clip_grad_norm_(params, MAX_GRAD_NORM)
        opt.step()
        opt.zero_grad(set_to_none=True)
        since += 1
        done = step + 1
        if done % LOGGING_STEPS == 0 or done == total_steps:
            ev = evaluate(model, eval_ids, local_batch_size, autocast) if (done % EVAL_STEPS == 0 or done == total_steps) and eval_ids.numel() else None
            LAST_LOG.append((done, running / since, ev))
            if rank == 0:
                print(f"step {done}/{total_steps}: loss {running / since:.4f}" + (f", eval loss {ev:.4f}" if ev is not None else ""))
            running, since = 0.0, 0
    if hasattr(model, "gradient_checkpointing_disable"):
        model.gradient_checkpointing_disable()
    model.eval()
    requantize(model)
