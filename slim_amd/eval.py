"""Perplexity harness (mirror of reference slim/eval.py:52-116)."""
import torch
import torch.nn as nn
import tqdm.auto as tqdm


@torch.no_grad()
def eval_ppl_wikitext(model, testenc, bs=1, device=None, fp32_loss=False):
    """exp(mean NLL) over consecutive windows of model.config.max_position_embeddings tokens.

    As in the reference the cross entropy is taken in the logits' dtype (bf16 models give a
    bf16-quantised loss per batch); fp32_loss=True evaluates the same quantity on fp32 logits."""
    ids = testenc.input_ids
    seqlen = model.config.max_position_embeddings
    nsamples = ids.numel() // seqlen
    nlls = []
    model.eval()
    for i in tqdm.tqdm(range(0, nsamples, bs)):
        j = min(i + bs, nsamples)
        inputs = ids[:, (i * seqlen):(j * seqlen)].to(device).reshape(j - i, seqlen)
        logits = model(inputs).logits
        if fp32_loss:
            logits = logits.float()
        shift_logits = logits[:, :-1, :].contiguous()
        shift_labels = inputs[:, 1:]
        loss = nn.CrossEntropyLoss()(shift_logits.reshape(-1, shift_logits.size(-1)), shift_labels.reshape(-1))
        nlls.append(loss.float() * seqlen * (j - i))
    return torch.exp(torch.stack(nlls).sum() / (nsamples * seqlen)).item()


def eval_ppl(model, tokenizer, eval_dataset, eval_batch_size):
    """eval.py:12-49: perplexity on a named dataset through get_loaders."""
    from .data import get_loaders
    _, testloader = get_loaders(eval_dataset, seed=0, seqlen=model.config.max_position_embeddings,
                                tokenizer=tokenizer, vocab_size=getattr(model.config, "vocab_size", None))
    return eval_ppl_wikitext(model, testloader, eval_batch_size, model.device)
