"""Calibration / evaluation token streams.  The reference (slim/data.py) downloads C4, WikiText-2,
OpenWebText or SlimPajama through `datasets`; this build targets machines without network access and
provides a synthetic stream of the same shapes.  Real datasets can be plugged in by replacing
`slim_amd.prune.get_loaders` (same signature and return value as the reference's get_loaders)."""
import torch


class TokenizerWrapper:
    def __init__(self, input_ids):
        self.input_ids = input_ids


def get_synthetic(nsamples, seed, seqlen, vocab_size):
    """[(ids [1, seqlen], None)] * nsamples and a TokenizerWrapper with [1, 64 * seqlen] evaluation ids."""
    g = torch.Generator().manual_seed(int(seed) + 1)
    train = [(torch.randint(0, vocab_size, (1, seqlen), generator=g), None) for _ in range(nsamples)]
    g = torch.Generator().manual_seed(int(seed) + 2)
    test = TokenizerWrapper(torch.randint(0, vocab_size, (1, 64 * seqlen), generator=g))
    return train, test


def get_loaders(name, nsamples=128, seed=0, seqlen=2048, tokenizer=None, vocab_size=None):
    if name.startswith("synthetic"):
        if vocab_size is None:
            vocab_size = getattr(tokenizer, "vocab_size", None)
        if vocab_size is None:
            raise ValueError("synthetic calibration data needs vocab_size (or a tokenizer with .vocab_size)")
        return get_synthetic(nsamples, seed, seqlen, vocab_size)
    raise NotImplementedError(
        f"dataset '{name}' needs network access to the HuggingFace hub, which this build does not assume; "
        "use calibration_dataset='synthetic' or replace slim_amd.prune.get_loaders")
