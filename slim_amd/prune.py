"""Public entry point and per-block driver (mirror of reference slim/prune.py:
prune_and_quantize :1419-1709, prune_wanda :148-372, prepare_calibration_input :26-78).

Differences in mechanism, not in contract: the calibration activations stay resident in HBM for the
whole run (the reference shuttles every sample CPU<->GPU twice per block), the block forwards may be
batched while the statistics keep the reference's one-sample-at-a-time accumulation order, and every
per-linear step (statistics, mask, scale search, quantisation, low-rank fit) is a HIP kernel.
"""
import os
import threading
import time

import torch
import torch.nn as nn
import tqdm.auto as tqdm

from . import ops, parallel
from .data import get_loaders
from .layerwrapper import WrappedGPT, shared_inputs
from .lora import add_lora  # noqa: F401  (re-exported: slim.prune.add_lora in the reference)
from .quantization.quantization import Quantizer as AutoQuantizer, QuantizedMatmul
from .utils import get_layers_list, find_layers

# samples per block forward during calibration (statistics are accumulated per sample regardless)
CALIBRATION_BATCH = int(os.environ.get("SLIM_CALIBRATION_BATCH", "8"))
# leave compressed blocks on the accelerator instead of returning them to the CPU like the reference
KEEP_ON_DEVICE = False
# optional callable(i, name, module, wrapped) invoked after each linear is compressed (tests, tracing)
LAYER_CALLBACK = None
# optional callable(i, name, wrapped) invoked once the statistics of a block's linear are final (after shift_zeros) and
# before anything is derived from them; it may replace wrapped.scaler_row (tests inject recorded statistics)
STATS_CALLBACK = None
# HIP streams over which the independent linears of one block are compressed concurrently.  The l x l
# eigen-solve of one linear is a long chain of tiny kernels; on its own stream it overlaps the GEMM-heavy
# phases of the other linears.  Each stream is fed by its own host thread (run_on_streams); 7 = one per linear
# of a LLaMA block.  1 = everything on the current stream.
COMPRESS_STREAMS = int(os.environ.get("SLIM_COMPRESS_STREAMS", "7"))
# the eigenproblems of a block's fits share one batched tridiagonalisation chain (SLIM_SHARED_EIGH=0: every fit solves its own)
SHARED_EIGH = os.environ.get("SLIM_SHARED_EIGH", "1") == "1"
# with the shared eigen-solver: linears of a block with the same short dimension, rank and Newton-Schulz depth (all seven of a
# LLaMA-2-7B block) run their fits as lockstep chains (SLIM_GROUP_FITS=0: one chain per linear)
GROUP_FITS = os.environ.get("SLIM_GROUP_FITS", "1") == "1"
# at most this many members per chain (0: no limit).  Measured on a LLaMA-2-7B-shaped block (seven linears with one short
# dimension): two chains of 4 + 3 members 22.4 ms, three of 3 + 3 + 1 22.4 ms, one chain of 7 25.0 ms (nothing overlaps its
# eigen-solver launches), four chains 29 ms; letting the second chain's front wait for the first one's (so that it runs
# behind the first chain's eigen-solver) 25.7 ms: the fronts, chains of small launches themselves, gain more side by side
GROUP_MAX = int(os.environ.get("SLIM_GROUP_MAX", "4"))
# with lockstep groups: every group runs its own eigen-solver chain right after its front instead of all groups meeting
# in one batched chain (SLIM_PIPELINE_GROUPS=0 restores that; measured 23.6 against 24.3 ms per block)
PIPELINE_GROUPS = os.environ.get("SLIM_PIPELINE_GROUPS", "1") == "1"
# which of the side streams carry the chains when there are at most that many: HIP spreads a process's streams over four
# in-order hardware queues, and these four of the seven side streams sit on four different ones
GROUP_STREAM_IDS = [int(x) for x in os.environ.get("SLIM_GROUP_STREAMS", "0,1,2,6").split(",") if x != ""]
# several GPUs: cut every linear of a block into row (or column) blocks over all ranks (sharded.py) where the recipe
# allows it; SLIM_SHARD_LINEARS=0: whole linears on their owners
SHARD_LINEARS = os.environ.get("SLIM_SHARD_LINEARS", "1") == "1"
# the first phase of a linear (mask, first scale search, error matrix, split-off of outlier columns) as one library call
# where the recipe allows (lora.prepare_fusable); SLIM_FUSED_PREPARE=0: the same steps as separate calls
FUSED_PREPARE = os.environ.get("SLIM_FUSED_PREPARE", "1") == "1"
_side_streams = {}


def _streams_for(dev, n, pool=None):
    key = (dev.index, n, pool)
    if key not in _side_streams:
        _side_streams[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _side_streams[key]


def block_stats(wrapped_list, rank_of=None):
    """Per-linear scalars the host needs before compressing a block, fetched with ONE synchronisation:
    has_zero (the any(scaler_row == 0) guard), the indices of outlier channels (saliency scale far above the
    median, which the low-rank fit deflates) and the spread of the remaining scales (sizes the Newton-Schulz
    orthonormalisation).  rank_of: callable(wrapped) -> adapter rank of that linear (default: a tenth of its smaller
    dimension, the recipe's ratio), which limits how many columns are split off."""
    from .lora import MAX_OUTLIER_COLS, _outliers_from_host
    if rank_of is None:
        rank_of = lambda w: max(1, min(w.rows, w.columns) // 10)
    if not wrapped_list:            # a rank that owns no linear of this block
        return
    top = MAX_OUTLIER_COLS + 1
    dev = wrapped_list[0].scaler_row.device
    vals = torch.empty(len(wrapped_list), 2 + top, dtype=torch.float32, device=dev)
    idxs = torch.empty(len(wrapped_list), top, dtype=torch.int32, device=dev)
    for j, w in enumerate(wrapped_list):
        ops.outlier_stats(w.scaler_row, top, vals[j], idxs[j])
    host = vals.cpu()                                               # the block's one synchronisation
    for j, (w, row) in enumerate(zip(wrapped_list, host.tolist())):
        k = min(top, w.scaler_row.numel())
        w.has_zero = bool(row[0])
        w.outliers, w.scale_spread = _outliers_from_host(row[2:2 + k], row[1], idxs[j, :k], rank_of(w))


# SLIM_STREAM_TRACE=1: run_on_streams records, per job, when its host thread started / finished enqueueing and when its
# stream started / finished executing it (ms after the call), into LAST_STREAM_TRACE; costs a device sync per call
STREAM_TRACE = os.environ.get("SLIM_STREAM_TRACE", "0") == "1"
# SLIM_STREAM_THREADS=0: the calling thread feeds the side streams one after the other (slower: one thread enqueues
# everything; for profilers that do not tolerate dispatches from several threads)
# Under rocprofv3 (its launcher exports ROCPROF_* variables to the profiled process) the streams are always fed by the
# calling thread: dispatches from several host threads while the profiler intercepts them took a GPU node down once.
PROFILER_ATTACHED = any(k.startswith(("ROCPROF_", "ROCPROFILER_")) for k in os.environ)
STREAM_THREADS = os.environ.get("SLIM_STREAM_THREADS", "1") != "0" and not PROFILER_ATTACHED
LAST_STREAM_TRACE = []


def run_on_streams(jobs, dev, n_streams=None, stream_ids=None, pool=None):
    """Run independent GPU jobs (callables that only enqueue work and return the tensors they produced) on side
    streams that start after the current stream's pending work; the current stream then waits for all of them.

    Each side stream is fed by its own host thread.  One compression job issues thousands of small launches
    (the eigen-solver alone is a storm of tiny kernels), so a single host thread, not the GPU, is the limit;
    the library calls release the GIL, which lets the threads enqueue concurrently.  Jobs are assigned
    round-robin in order, so the stream (and the result) of every job is deterministic.  pool: names a set of side
    streams of its own (a job that itself fans work out must not share streams with the chains running beside it)."""
    n = COMPRESS_STREAMS if n_streams is None else n_streams
    main = torch.cuda.current_stream(dev)
    if n <= 1 or len(jobs) <= 1:
        for job in jobs:
            job()
        return
    if stream_ids and len(jobs) <= len(stream_ids) and max(stream_ids) < n:
        full = _streams_for(dev, n, pool)      # a chosen subset of the full set of side streams
        streams = [full[i] for i in stream_ids[:len(jobs)]]
    else:
        streams = _streams_for(dev, min(n, len(jobs)), pool)
    trace = STREAM_TRACE and pool is None      # the outermost call only: nested fan-outs would overwrite its record
    ev = torch.cuda.Event(enable_timing=trace)
    ev.record(main)
    errors = []
    t_base = time.perf_counter()
    records = []

    def feed(si):
        try:
            torch.cuda.set_device(dev)
            st = streams[si]
            st.wait_event(ev)
            with torch.cuda.stream(st), torch.no_grad():
                for idx in range(si, len(jobs), len(streams)):
                    if trace:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(st)
                        h0 = time.perf_counter()
                    produced = jobs[idx]() or ()
                    if trace:
                        e1.record(st)
                        records.append((idx, si, h0 - t_base, time.perf_counter() - t_base, e0, e1))
                    for t in produced:          # these outlive the side stream's scope and are used on `main`
                        if isinstance(t, torch.Tensor) and t.is_cuda:
                            t.record_stream(main)
        except BaseException as e:              # re-raised on the calling thread
            errors.append(e)

    if STREAM_THREADS:
        threads = [threading.Thread(target=feed, args=(si,), name=f"slim-stream-{si}") for si in range(len(streams))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    else:
        for si in range(len(streams)):
            feed(si)
    for st in streams:
        main.wait_stream(st)
    if errors:
        raise errors[0]
    if trace:
        torch.cuda.synchronize(dev)
        LAST_STREAM_TRACE[:] = [dict(job=idx, stream=si, host_start_ms=h0 * 1e3, host_end_ms=h1 * 1e3,
                                     gpu_start_ms=ev.elapsed_time(e0), gpu_end_ms=ev.elapsed_time(e1))
                                for idx, si, h0, h1, e0, e1 in sorted(records)]


# SLIM_PHASE_TRACE=1 (debugging aid): every chain of compress_linears waits for the GPU after each of its phases and appends
# (chain, phase, milliseconds since the chain started) to LAST_PHASE_TRACE; the waits serialise host and GPU, so the run
# is slower than an untraced one
PHASE_TRACE = os.environ.get("SLIM_PHASE_TRACE", "0") == "1"
LAST_PHASE_TRACE = []


def _phase_marker(chain):
    if not PHASE_TRACE:
        return lambda name: None
    t0 = time.perf_counter()

    def mark(name):
        torch.cuda.current_stream().synchronize()
        LAST_PHASE_TRACE.append((chain, name, (time.perf_counter() - t0) * 1e3))
    return mark


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("slim_amd needs a ROCm GPU: the compression path has no CPU implementation")
    return torch.device("cuda", torch.cuda.current_device())


def prepare_calibration_input(model, dataloader, nsamples=128, device=None):
    """Capture the inputs of decoder block 0 for every calibration sample (prune.py:26-78).

    Returns (inps, outs, kwargs): inps/outs are [nsamples, seqlen, hidden] in the model dtype, held on
    `device` (the reference keeps them on the CPU), kwargs are the block's keyword arguments."""
    use_cache = model.config.use_cache
    model.config.use_cache = False
    layers = get_layers_list(model)
    dtype = next(iter(model.parameters())).dtype
    device = device or _device()
    inps = torch.zeros((nsamples, model.config.max_position_embeddings, model.config.hidden_size), dtype=dtype,
                       device=device)
    inps.requires_grad = False
    cache = {"i": 0, "attention_mask": None, "position_ids": None}

    class Catcher(nn.Module):
        def __init__(self, module):
            super().__init__()
            self.module = module

        def forward(self, inp, **kwargs):
            inps[cache["i"]] = inp.to(device)
            cache["i"] += 1
            for key in kwargs:
                cache[key] = kwargs[key]
            raise ValueError

    layers[0] = Catcher(layers[0])
    model_device = next(iter(model.parameters())).device
    for batch in dataloader:
        try:
            model(batch[0].to(model_device))
        except ValueError:
            pass
    layers[0] = layers[0].module
    outs = torch.zeros_like(inps)
    model.config.use_cache = use_cache
    del cache["i"]
    return inps, outs, cache


def _to_device(kwargs, device):
    out = {}
    for key, v in kwargs.items():
        if isinstance(v, torch.Tensor):
            v = v.to(device)
        elif isinstance(v, tuple):
            v = tuple(k.to(device) if isinstance(k, torch.Tensor) else k for k in v)
        out[key] = v
    return out


def add_lora_hook(module, input, output):
    """Forward hook adding the adapter path to the Linear's output in place (prune.py:303-321)."""
    if hasattr(module, "lora_quantizer"):
        xl = QuantizedMatmul.apply(input[0].to(module.lora_left.dtype) / torch.sqrt(module.lora_rank),
                                   module.lora_left, module.lora_quantizer)
        xlr = QuantizedMatmul.apply(xl / torch.sqrt(module.lora_rank), module.lora_right, module.lora_quantizer)
        output += xlr
    else:
        output += torch.matmul(
            torch.matmul(input[0].to(module.lora_left.dtype), module.lora_left / torch.sqrt(module.lora_rank)),
            module.lora_right / torch.sqrt(module.lora_rank))


class _StopBlock(Exception):
    """Raised by the statistics hook of the last linear of a block: nothing after it is needed."""


def _run_block(layer, src, dst, kwargs, batch, after_batch=None):
    """layer(src) in batches of `batch` samples into dst.  A batch whose forward ends with _StopBlock (raised by a hook)
    leaves its part of dst as it is; after_batch(index) runs after every batch."""
    n = src.shape[0]
    with torch.no_grad():
        for b, j in enumerate(range(0, n, batch)):
            try:
                with shared_inputs():       # q/k/v (gate/up) read one tensor: its column sums are computed once
                    out = layer(src[j:j + batch], **kwargs)
                if isinstance(out, (tuple, list)):      # older decoder layers return (hidden_states, ...)
                    out = out[0]
                dst[j:j + batch] = out
            except _StopBlock:
                pass
            if after_batch is not None:
                after_batch(b)


# The statistics pass of a block only needs the INPUTS of its linears; its output is overwritten by the pass that follows
# the compression (prune.py:362-364).  So from the second batch on, the linear that the first batch showed to run last
# takes its statistics in a pre-hook and ends the block's forward there: its own product (in LLaMA the 11008 -> 4096
# down projection, a fifth of the block's GEMM work) and whatever follows it are never computed.  Statistics are
# unchanged (the same tensors reach the same linears).
STATS_EARLY_EXIT = os.environ.get("SLIM_STATS_EARLY_EXIT", "1") == "1"


def gather_block_statistics(layer, subset, wrapped_layers, inps, outs, kwargs, batch):
    """The forward pass of prune.py:247-254: every linear's wrapper sees every calibration sample's input."""
    order = []              # the linears in the order they ran in the current batch

    def add_batch(name):
        def tmp(_, inp, out):
            order.append(name)
            wrapped_layers[name].add_batch(inp[0].data, out.data, per_sample=True)
        return tmp

    handles = {name: subset[name].register_forward_hook(add_batch(name)) for name in wrapped_layers}

    def after_batch(b):
        first = list(order)
        order.clear()
        if b != 0 or not STATS_EARLY_EXIT or sorted(first) != sorted(wrapped_layers):
            return          # (a linear that ran twice or not at all: keep the plain pass)
        last = first[-1]

        others = set(wrapped_layers) - {last}

        def pre(_, args):
            wrapped_layers[last].add_batch(args[0].data, None, per_sample=True)
            if set(order) == others:                        # every other linear has had its input: nothing left to see
                raise _StopBlock
            order.append(last)                              # (a block that changed its order: run on)

        handles[last].remove()
        handles[last] = subset[last].register_forward_pre_hook(pre)

    try:
        _run_block(layer, inps, outs, kwargs, batch, after_batch)
    finally:
        for h in handles.values():
            h.remove()


COND_WORD = 63        # SLIM_FIT_INFO_COND_WORD of the last 64-word status block of a fit


def check_fit_status(items, levels=None):
    """Raise if the low-rank fit of any of these linears reported a failure (a status word of its eigen-solver or of an
    orthonormalisation is non-zero): such a fit leaves arbitrary adapters and a residual weight quantised against them.
    Returns {index in items: level} for the fits that ask to be repeated with orthonormalisations sized for a wider
    spread of singular values (their last status word).  levels (a dict, optional) receives {index in items: the spread
    the fit measured, where its cut sits} (status words SLIM_FIT_INFO_LEVEL_WORD and SLIM_FIT_INFO_RATIO_WORD; fits that
    report none are left out).  One host
    synchronisation for the whole block."""
    named = [(i, name, lin.lora_fit_info) for i, (name, lin, _) in enumerate(items)
             if getattr(lin, "lora_fit_info", None) is not None]
    if not named:
        return {}
    # one stacked tensor and a handful of launches for the whole block (per-linear reductions cost the host more than the GPU)
    words = torch.stack([info.reshape(-1)[-64:].to(torch.int32) for _, _, info in named])
    extra = [info.reshape(-1)[:-64].ne(0).any() for _, _, info in named if info.numel() > 64]
    # (the last three words are reports, not errors: conditioning request, measured spread, where the cut sits)
    flags = torch.cat([words[:, :-3].ne(0).any(1, keepdim=True).to(torch.int32), words[:, -1:], words[:, -2:-1], words[:, -3:-2]], 1)
    if extra:
        flags[[i for i, (_, _, info) in enumerate(named) if info.numel() > 64], 0] |= torch.stack(extra).to(torch.int32)
    flags = flags.cpu().tolist()
    failed = [(name, info[:-3].cpu().tolist()) for (_, name, info), f in zip(named, flags) if f[0]]
    if failed:
        from ._lib import SlimHipError, ERR_SOLVER
        raise SlimHipError("low-rank fit", ERR_SOLVER, "; ".join(
            f"{name}: status words {[w for w in words if w][:8]}" for name, words in failed))
    if levels is not None:
        levels.update({i: (int(f[2]), int(f[3])) for (i, _, _), f in zip(named, flags) if f[2] > 0})
    return {i: int(f[1]) for (i, _, _), f in zip(named, flags) if f[1] > 0}


def _reset_linear(lin, weight):
    """Undo what compress_linears left on a linear (for a repeated fit): original weight, no adapters, no adapter hook."""
    lin.weight.data = weight
    for attr in ("lora_left", "lora_right", "lora_rank", "scaling_factor", "lora_fit_info", "lora_left_mask"):
        if hasattr(lin, attr):
            delattr(lin, attr)
    for key in [k for k, h in lin._forward_hooks.items() if h is add_lora_hook]:
        del lin._forward_hooks[key]


def compress_linears(items, dev, *args, level_hints=None, **kwargs):
    """compress_linears_unchecked, then the status words of the fits (one host synchronisation per block): failures
    raise; fits whose orthonormalisations turned out to be sized for too narrow a spectrum are repeated once with the
    level they report.
    level_hints (a dict the caller keeps from block to block, optional): the fits of this block are sized from what the
    same linears (same name and shape) measured in the last block that used the dict, and leave their own measurements in
    it.  Neighbouring decoder blocks have similar error spectra, and a measured spread is tighter than the estimate from
    the saliency scales, which saves Newton-Schulz steps; a fit whose hint turns out too low is repeated without it."""
    from . import lora as _lora
    originals = [lin.weight.data for _, lin, _ in items]
    masks = [getattr(lin.weight, "mask", None) for _, lin, _ in items]
    keys = [(name, tuple(lin.weight.shape)) for name, lin, _ in items]
    for key, (_, _, wrapped) in zip(keys, items):
        wrapped.fit_level_hint = level_hints.get(key, 0) if level_hints is not None else 0
        wrapped.fit_cond_floor = 0
    compress_linears_unchecked(items, dev, *args, **kwargs)
    found = {}
    again = check_fit_status(items, found)
    if again:
        for i in again:
            _reset_linear(items[i][1], originals[i])
            if masks[i] is not None:
                items[i][1].weight.mask = masks[i]
        order = sorted(again)
        for i in order:                     # the repeated fits are sized for the widest spectrum any of them reported
            items[i][2].fit_cond_floor = max(again.values())
        compress_linears_unchecked([items[i] for i in order], dev, *args, **kwargs)
        refound = {}
        check_fit_status([items[i] for i in order], refound)      # failures still raise; a second request is not followed
        found.update({order[j]: w for j, w in refound.items()})
    if level_hints is not None:
        for i, word in found.items():
            level_hints[keys[i]] = _lora.level_hint_from_status(*word)


class _MemberPhases:
    """The per-linear phases of one compress_linears call, bound to its recipe: mask -> first quantisation and error matrix
    (prepare, optionally with the split-off of the outlier columns) ... the fit, run by the caller ... residual weight, second
    quantisation, adapter attributes and hook (finish); or, without adapters, mask and quantisation (plain).  contexts[i] is
    the LoraContext of items[i] between its prepare and its finish."""

    def __init__(self, items, dev, prune_n, prune_m, sparsity_ratio, quantizer, lora_rank, slim_lora, prune_lora, tile, separate_lora,
                 quantize_first, scale_important_weights, tiled_weight_quantization, callback, describe):
        self.items, self.dev, self.prune_n, self.prune_m, self.sparsity_ratio = items, dev, prune_n, prune_m, sparsity_ratio
        self.quantizer, self.lora_rank, self.slim_lora, self.prune_lora, self.tile = quantizer, lora_rank, slim_lora, prune_lora, tile
        self.separate_lora, self.quantize_first, self.scale_important_weights = separate_lora, quantize_first, scale_important_weights
        self.tiled_weight_quantization, self.callback, self.describe = tiled_weight_quantization, callback, describe
        self.contexts = [None] * len(items)

    def given_mask(self, lin):
        """A precomputed (MaskLLM-style) mask attached to the weight, prune.py:269-271 (taken off the weight), or None."""
        if not hasattr(lin.weight, "mask"):
            return None
        W_mask = lin.weight.mask.to(self.dev)
        del lin.weight.mask
        return W_mask

    def mask_of(self, lin, wrapped):
        W_mask = self.given_mask(lin)
        if W_mask is not None:
            return W_mask
        if self.prune_n != 0:
            return ops.mask_nm(lin.weight.data, ops.sqrt(wrapped.scaler_row), self.prune_n, self.prune_m)
        # unstructured (prune.py:275-279): per row the int(K * ratio) smallest saliencies, ties to the lower index
        K = lin.weight.shape[1]
        return ops.mask_unstructured(lin.weight.data.contiguous(), ops.sqrt(wrapped.scaler_row), int(K * self.sparsity_ratio))

    @staticmethod
    def produced(lin):
        """The tensors a job leaves on its linear (run_on_streams hands them back to the caller's stream)."""
        return [lin.weight.data, getattr(lin, "scaling_factor", None)] + \
               [p.data for p in (getattr(lin, "lora_left", None), getattr(lin, "lora_right", None)) if p is not None]

    def plain(self, idx):
        """No adapters (prune.py:330-357): quantise the dense weight then zero the mask, or the other way round."""
        name, lin, wrapped = self.items[idx]
        quantizer = self.quantizer
        if self.describe:
            self.describe(name)
        W_mask = self.mask_of(lin, wrapped)
        important = None
        if self.scale_important_weights:
            # prune.py:331-335 reads `scaler_row` off the nn.Linear, which has none: the reference raises here
            important = ops.important_columns(lin.weight.data, lin.scaler_row)

        def quantise():
            if quantizer is not None:
                codes = quantizer.quantize_weight(lin.weight.data, important)
                lin.weight.data = quantizer.dequantize_absmax(codes).to(torch.bfloat16)
                lin.scaling_factor = None if self.tiled_weight_quantization else quantizer.scaling_factor
        if self.quantize_first:
            quantise()
            lin.weight.data[W_mask] = 0
        else:
            lin.weight.data[W_mask] = 0
            quantise()
        if self.callback is not None:
            self.callback(name, lin, wrapped)
        return self.produced(lin)

    def prepare(self, idx):
        from .lora import add_lora_prepare
        name, lin, wrapped = self.items[idx]
        if self.describe:
            self.describe(name)
        self.contexts[idx] = add_lora_prepare(lin, self.mask_of(lin, wrapped), self.lora_rank, self.slim_lora, wrapped, self.quantizer,
                                              self.prune_lora, self.separate_lora, self.tile, self.quantize_first,
                                              self.scale_important_weights)
        return self.contexts[idx].tensors()

    def prepare_and_deflate(self, idx):
        """prepare plus the split-off of the outlier columns; as ONE library call where the recipe allows."""
        from . import lora as _lora
        name, lin, wrapped = self.items[idx]
        if FUSED_PREPARE and _lora.prepare_fusable(lin, self.quantizer, self.slim_lora, self.prune_lora, self.separate_lora,
                                                   self.quantize_first, self.scale_important_weights, self.prune_n):
            if self.describe:
                self.describe(name)
            self.contexts[idx] = _lora.add_lora_prepare_fused(lin, self.prune_n, self.prune_m, self.lora_rank, self.slim_lora, wrapped,
                                                              self.quantizer, self.tile, self.given_mask(lin))
        else:
            self.prepare(idx)
            _lora.add_lora_deflate(self.contexts[idx])
        return self.contexts[idx].tensors()

    def finish(self, idx):
        from .lora import add_lora_finish
        name, lin, wrapped = self.items[idx]
        sf = add_lora_finish(self.contexts[idx], prescale=self.separate_lora)     # prune.py:296-301: adapters stored times sqrt(rank)
        if self.quantizer is not None:
            lin.scaling_factor = None if self.tiled_weight_quantization else sf
        if self.separate_lora:
            lin.register_forward_hook(add_lora_hook)
        if self.callback is not None:
            self.callback(name, lin, wrapped)
        self.contexts[idx] = None
        return self.produced(lin)


def _run_fit_chains(phases, plans, n_streams):
    """The adapters of a block through lockstep chains around a shared Rayleigh-Ritz stage.

    The fits of a block pose one l x l eigenproblem each, whose solver is a chain of dependent launches: seven chains side by
    side take seven times the launches on four in-order hardware queues.  So linears with equal plans (plans[i] = (key, l)
    from lora.planned_fit_group: same short dimension, rank, Newton-Schulz depth and filter class -- the shapes may differ)
    run as ONE chain whose subspace iteration is a single batched launch sequence; GROUP_MAX caps the members of a chain,
    and several chains overlap on streams of their own."""
    from . import lora as _lora
    items, dev, ctxs = phases.items, phases.dev, phases.contexts
    fit_groups, seen = {}, {}
    for idx, (key, l) in enumerate(plans):
        n_seen = seen.get((key, l), 0)
        seen[(key, l)] = n_seen + 1
        part = n_seen // GROUP_MAX if GROUP_MAX > 0 else 0          # at most GROUP_MAX members per chain
        fit_groups.setdefault((key, l, part) if GROUP_FITS else (key, l, idx), []).append(idx)
    fit_groups = list(fit_groups.items())
    count, where = {}, []
    for gkey, members in fit_groups:
        l = gkey[1]
        where.append((l, count.get(l, 0)))
        count[l] = count.get(l, 0) + len(members)
    Mb = {l: torch.empty(n, l, l, dtype=torch.float32, device=dev) for l, n in count.items()}
    split = len(items) < 3          # fewer fits than that do not fill the chip by themselves

    def fit_front(g):
        members = fit_groups[g][1]
        l, b0 = where[g]
        if len(members) == 1:
            _lora.add_lora_fit_front(ctxs[members[0]], Mb[l][b0], sketch_split=split)
        else:
            _lora.add_lora_fit_front_group([ctxs[i] for i in members], Mb[l][b0:b0 + len(members)], sketch_split=split)

    def fit_back(g, evals, vecs, tail):
        members = fit_groups[g][1]
        if len(members) == 1:
            _lora.add_lora_fit_back(ctxs[members[0]], evals[0], None, None, vecs[0], tail[0:1])
        else:
            _lora.add_lora_fit_back_group([ctxs[i] for i in members], evals, None, None, vecs, tail)

    def fit_rank(g):
        return ctxs[fit_groups[g][1][0]].pending["ctx"]["r"]

    ids = GROUP_STREAM_IDS if len(fit_groups) <= len(GROUP_STREAM_IDS) else None
    if PIPELINE_GROUPS and len(fit_groups) <= 4:
        # every chain runs whole -- its members' first phases, the lockstep front, its own Rayleigh-Ritz stage, the factors, its
        # members' last phases -- on its stream without waiting for the others: the latency-bound launches of one chain
        # overlap the GEMMs of the others.  (More host threads than chains do not pay: spreading the members' first phases
        # over seven threads was slower than running them in the chain.)
        def whole_chain(g):
            members = fit_groups[g][1]
            l, b0 = where[g]
            mark = _phase_marker(g)
            for i in members:
                phases.prepare_and_deflate(i)
            mark("prepare")
            fit_front(g)
            mark("front")
            Mg = Mb[l][b0:b0 + len(members)]
            evals, vecs, tail = _lora.rayleigh_ritz(Mg, fit_rank(g), [_lora.fit_keep(ctxs[i]) for i in members],
                                                    max(ctxs[i].cond_floor for i in members))
            mark("rayleigh_ritz")
            fit_back(g, evals, vecs, tail)
            mark("back")
            out = [evals, vecs, tail]
            for i in members:
                out += phases.finish(i)
            mark("finish")
            return out
        run_on_streams([lambda g=g: whole_chain(g) for g in range(len(fit_groups))], dev, n_streams, ids)
        return

    # more chains than streams to pipeline them on: phase by phase, the eigenproblems of the block size by size in between
    def front(g):
        for i in fit_groups[g][1]:
            phases.prepare(i)
        fit_front(g)
        return [x for i in fit_groups[g][1] for x in ctxs[i].tensors()]
    run_on_streams([lambda g=g: front(g) for g in range(len(fit_groups))], dev, n_streams, ids)
    top, keeps = {}, {}
    for g in range(len(fit_groups)):
        l = where[g][0]
        top[l] = max(top.get(l, 0), fit_rank(g))
        keeps.setdefault(l, []).extend(_lora.fit_keep(ctxs[i]) for i in fit_groups[g][1])
    floor = max(c.cond_floor for c in ctxs)
    rr = {l: _lora.rayleigh_ritz(Mb[l], top[l], keeps[l], floor) for l in sorted(Mb)}

    def back(g):
        members = fit_groups[g][1]
        l, b0 = where[g]
        sl = slice(b0, b0 + len(members))
        evals, vecs, tails = rr[l]
        fit_back(g, evals[sl], vecs[sl], tails[sl])
        return [x for i in members for x in ctxs[i].tensors()]
    run_on_streams([lambda g=g: back(g) for g in range(len(fit_groups))], dev, n_streams, ids)
    run_on_streams([lambda i=i: phases.finish(i) for i in range(len(items))], dev, n_streams)


def compress_linears_unchecked(items, dev, prune_n, prune_m, sparsity_ratio=0.5, quantizer=None, lora_rank=0., slim_lora=True, prune_lora=False,
                     quantize_lora=False, pad_lora=False, lora_tile_size=256, separate_lora=True, quantize_first=True,
                     scale_important_weights=False, tiled_weight_quantization=False, callback=None, describe=None,
                     n_streams=None, shard=False):
    """Compress the linears of one block after their statistics pass (the body of the per-linear loop of
    prune.py:256-357).  items: list of (name, nn.Linear, WrappedGPT) with block_stats() already applied.

    Three ways through, by recipe: no adapters -- mask and quantisation per linear (_MemberPhases.plain); separate, unpruned
    adapters on more than one linear -- lockstep chains (_run_fit_chains); otherwise -- the three phases of every linear back
    to back on a stream of its own.
    shard=True (several ranks, every rank passing the same items): the row-block partition of sharded.py instead."""
    tile = lora_tile_size if (quantize_lora or pad_lora) else None
    if shard:
        # every rank holds a block of every linear (the caller has checked sharded.supported on these items)
        from . import sharded
        sharded.compress_linears_sharded(items, dev, prune_n, prune_m, quantizer, lora_rank, slim_lora, tile, callback, describe)
        return
    phases = _MemberPhases(items, dev, prune_n, prune_m, sparsity_ratio, quantizer, lora_rank, slim_lora, prune_lora, tile, separate_lora,
                           quantize_first, scale_important_weights, tiled_weight_quantization, callback, describe)
    if not lora_rank > 0.:
        # the shared quantizer keeps its last scale as state, so these jobs stay on one thread
        run_on_streams([lambda i=i: phases.plain(i) for i in range(len(items))], dev, 1)
        return
    for _, _, wrapped in items:
        if getattr(wrapped, "has_zero", None) is None:
            block_stats([wrapped])
    if SHARED_EIGH and separate_lora and not prune_lora and len(items) > 1:
        from . import lora as _lora
        plans = [_lora.planned_fit_group(lin, lora_rank, tile, slim_lora, wrapped) for _, lin, wrapped in items]
        if all(p is not None for p in plans):
            _run_fit_chains(phases, plans, n_streams)
            return
    # a single linear, pruned or merged adapters, or an eigenproblem the shared stage does not take
    from .lora import add_lora_fit

    def whole(idx):
        phases.prepare(idx)
        add_lora_fit([phases.contexts[idx]])
        return phases.finish(idx)
    run_on_streams([lambda i=i: whole(i) for i in range(len(items))], dev, n_streams)


def _exchange_block(names, subset, owner, rank, dev, lora_rank, tiled_rank, lora_tile_size, has_scale, separate_lora=True,
                    scale_dtype=torch.float32, prune_lora=False, recipe=(0, 0, 0)):
    """Every rank receives the compressed form of the linears it does not own (weight, adapters, scale, mask of pruned
    adapters) and gives them the attributes and the hook the owner's copy has."""
    from .lora import lora_rank as rank_of
    # everything an owner has to send goes out as ONE broadcast (parallel.broadcast_coalesced)
    outgoing = {}            # owner -> tensors
    completions = []
    for j, n in enumerate(names):
        lin = subset[n]
        N, K = lin.weight.shape
        bucket = outgoing.setdefault(owner[j], [])
        if lora_rank > 0. and separate_lora:
            r = min(rank_of((N, K), lora_rank, lora_tile_size if tiled_rank else None), min(N, K))
            mine = owner[j] == rank
            # a 2:4 sparse 4-bit weight travels packed (0.375 instead of 2 bytes per element), recipe = (n, m, bits)
            completions.append(parallel.exchange_linear(lin, owner[j], rank, (N, K, r), dev, has_scale=has_scale,
                                                        scale_dtype=scale_dtype, has_left_mask=prune_lora,
                                                        packed=has_scale and parallel.packable(N, K, *recipe), defer=bucket))
            if not mine:
                lin.lora_rank = torch.tensor(r)
                lin.register_forward_hook(add_lora_hook)
        else:
            bucket.append(lin.weight.data)
            if has_scale:
                if owner[j] != rank:
                    lin.scaling_factor = torch.empty(1, 1, dtype=scale_dtype, device=dev)
                bucket.append(lin.scaling_factor)
    for src in sorted(outgoing):
        parallel.broadcast_coalesced(outgoing[src], src)
    for complete in completions:
        complete()
    # the owners' packing status words, read with one synchronisation
    # the owners' packing status words, one per linear (0 where nothing was packed), combined over the ranks so that every
    # rank raises together: the receivers have unpacked what the owner sent and would otherwise run on into the next
    # block's collectives
    words = torch.zeros(len(names), dtype=torch.int32, device=dev)
    for j, n in enumerate(names):
        st = subset[n].__dict__.pop("pack_status", None)
        if st is not None:
            words[j] = st.reshape(-1)[0].to(torch.int32)
    words = parallel.all_reduce_max(words).cpu().tolist()
    bad = [(n, st) for n, st in zip(names, words) if st]
    if bad:
        raise ValueError("; ".join(f"{n}: the compressed weight does not have the packed 2:4 int4 form (status {st})"
                                   for n, st in bad))


def prune_wanda(model, tokenizer, sparsity_ratio=0.5, prune_n=0, prune_m=0, quantize_weight=False, bitwidth=4,
                slim_quant=False, tiled_weight_quantization=False, weight_tile_size=256, shift_zero_metrics=True,
                lora_rank=0., slim_lora=True, prune_lora=False, quantize_lora=False, lora_tile_size=256,
                separate_lora=True, nsamples=128, seed=0, calibration_dataset="c4", pad_lora=False,
                quantize_first=True, scale_important_weights=False):
    """Wanda N:M pruning + weight quantisation + low-rank adapters, block by block (prune.py:148-372)."""
    dev = _device()
    use_cache = model.config.use_cache
    model.config.use_cache = False

    dataloader, _ = get_loaders(calibration_dataset, nsamples=nsamples, seed=seed,
                                seqlen=model.config.max_position_embeddings, tokenizer=tokenizer)
    with torch.no_grad():
        inps, outs, kwargs = prepare_calibration_input(model, dataloader, nsamples, device=dev)
    kwargs = _to_device(kwargs, dev)
    # Several GPUs (one process each, torch.distributed initialised by the caller): every rank keeps a contiguous
    # share of the calibration samples and runs the block forwards on it; per-sample statistics are gathered into
    # sample order, so scaler_row is bit-identical to a single-GPU run; the linears of a block are compressed by
    # their owners (cost-balanced) and broadcast, so every rank holds the whole compressed block again.
    rank, world = parallel.world_info()
    if world > 1:
        if nsamples % world != 0:
            raise ValueError(f"nsamples ({nsamples}) must be divisible by the number of ranks ({world})")
        n_local = nsamples // world
        inps = inps[rank * n_local:(rank + 1) * n_local].clone()
        outs = outs[rank * n_local:(rank + 1) * n_local].clone()

    quantizer = AutoQuantizer("weight", num_bits=bitwidth, slim_quant=slim_quant,
                              block_quantization=tiled_weight_quantization,
                              block_dim=weight_tile_size) if quantize_weight else None
    layers = get_layers_list(model)
    progress_bar = tqdm.tqdm(range(len(layers)))
    level_hints = {}        # what the fits of a block measured, for the same linears of the next block (compress_linears)
    for i in progress_bar:
        progress_bar.set_description(f"Layer {i} - Gathering data")
        layer = layers[i].to(dev)
        subset = find_layers(layer)
        wrapped_layers = {name: WrappedGPT(subset[name], defer=world > 1) for name in subset}

        gather_block_statistics(layer, subset, wrapped_layers, inps, outs, kwargs, CALIBRATION_BATCH)

        # shift_zeros (prune.py:261-262), then the `any(scaler_row == 0)` guard of add_lora (lora.py:139, which
        # sees the shifted statistics) for every linear with a single host synchronisation
        if world > 1:
            for name in subset:
                wrapped_layers[name].finish_deferred(parallel.gather_sample_stats)
        if shift_zero_metrics:
            for name in subset:
                wrapped_layers[name].scaler_row = ops.shift_zeros(wrapped_layers[name].scaler_row)
        if STATS_CALLBACK is not None:
            for name in subset:
                STATS_CALLBACK(i, name, wrapped_layers[name])
        from .lora import lora_rank as _rank_rule
        tile_rule = lora_tile_size if (quantize_lora or pad_lora) else None
        block_stats(list(wrapped_layers.values()),
                    (lambda w: min(_rank_rule((w.rows, w.columns), lora_rank, tile_rule), w.rows, w.columns)) if lora_rank > 0. else None)

        names = list(subset)
        opts = dict(sparsity_ratio=sparsity_ratio, quantizer=quantizer,
                    lora_rank=lora_rank, slim_lora=slim_lora, prune_lora=prune_lora, quantize_lora=quantize_lora, pad_lora=pad_lora,
                    lora_tile_size=lora_tile_size, separate_lora=separate_lora, quantize_first=quantize_first,
                    scale_important_weights=scale_important_weights, tiled_weight_quantization=tiled_weight_quantization,
                    callback=(lambda name, lin, wrapped, i=i: LAYER_CALLBACK(i, name, lin, wrapped)) if LAYER_CALLBACK else None,
                    describe=lambda name, i=i: progress_bar.set_description(f"Layer {i} - Pruning and Quantizing {name}"))
        everything = [(n, subset[n], wrapped_layers[n]) for n in names]
        # a precomputed mask need not be 2:4, and important columns carry levels of half the size: neither has the packed form
        given_masks = any(hasattr(subset[n].weight, "mask") for n in names)
        from . import sharded
        if world > 1 and SHARD_LINEARS and not scale_important_weights and sharded.supported(everything, world, prune_n, prune_m, quantizer, lora_rank, slim_lora,
                                                             prune_lora, separate_lora, quantize_first, tile_rule):
            # every linear cut into blocks over all ranks; every rank ends with the whole compressed block
            compress_linears(everything, dev, prune_n, prune_m, shard=True, level_hints=level_hints, **opts)
        else:
            owner = parallel.partition_by_cost([parallel.linear_cost(*subset[n].weight.shape) for n in names], world)
            compress_linears([it for j, it in enumerate(everything) if owner[j] == rank], dev, prune_n, prune_m,
                             level_hints=level_hints, **opts)
            if world > 1:
                _exchange_block(names, subset, owner, rank, dev, lora_rank, quantize_lora or pad_lora, lora_tile_size,
                                quantizer is not None and not tiled_weight_quantization, separate_lora,
                                torch.float32 if slim_quant else torch.bfloat16, prune_lora,
                                (prune_n, prune_m, bitwidth if (quantizer is not None and quantize_first and not given_masks
                                                                and not scale_important_weights) else 0))

        if i + 1 < len(layers):         # (the last block's outputs feed nothing)
            progress_bar.set_description(f"Layer {i} - Evaluating Output")
            _run_block(layer, inps, outs, kwargs, CALIBRATION_BATCH)
            inps, outs = outs, inps
        if not KEEP_ON_DEVICE:
            layers[i] = layer.cpu()
        del layer

    model.config.use_cache = use_cache
    torch.cuda.empty_cache()


def prune_and_quantize(model, tokenizer, bitwidth=4, slim_quant=True, weight_tiled_quantization=False,
                       weight_tile_size=256, prune_method="wanda", sparsity_ratio=0.5, sparsity_type="2:4",
                       quantize_weight=False, nsamples=128, shift_zero_metrics=True, lora_rank=0., slim_lora=True,
                       prune_lora=False, quantize_lora=False, lora_tile_size=256, separate_lora=True, seed=0,
                       joint_pq_mixing_factor=2.1, calibration_dataset="c4", pad_lora=False,
                       scale_important_weights=False, mask_checkpoint=None, quantize_first=True):
    """Prune, quantise and add low-rank adapters to `model` in place; returns None (prune.py:1419-1709).

    Same signature, flag validation and error behaviour as the reference.  The Wanda / SLiM path
    ("wanda", and "maskllm" when the weights already carry a `.mask`) is implemented; the alternative
    baselines of the reference (magnitude, sparsegpt, joint_pq, pruner_zero, prune_gblm) and the dense
    quantisation-only mode raise NotImplementedError after the reference's own argument checks."""
    if sparsity_ratio == 0. or sparsity_type == "dense":
        if quantize_weight:
            print("Quantizing the dense model:")
            if lora_rank > 0:
                raise NotImplementedError("LoRA approximation not implemented for quantization only - "
                                          "Please use pruning with low sparsity ratio for quantization only.")
            raise NotImplementedError("dense quantization-only mode is not implemented in slim_amd")
        print("Using original dense model.")
        return
    print("Sparsity Ratio: ", sparsity_ratio)
    print("Pruning Structure: ", sparsity_type)
    prune_n, prune_m = 0, 0
    if sparsity_type != "unstructured":
        prune_n, prune_m = map(int, sparsity_type.split(":"))
        prune_n = prune_m - prune_n
        assert sparsity_ratio == prune_n / prune_m, \
            f"Sparsity ratio must be {prune_n / prune_m} for structured N:M sparsity"
    if prune_method in ["wanda", "maskllm", "pruner_zero", "prune_gblm"]:
        if quantize_weight:
            method = "SLiM-Quant" if slim_quant else ("Tiled Group AbsMax" if weight_tiled_quantization else "AbsMax")
            print(f"Pruning the model with {prune_method} and quantizing the weights using {method}.")
        else:
            print(f"Pruning the model with {prune_method}.")
        if lora_rank > 0:
            print(f"Adding {'SLiM' if slim_lora else 'Naive'}-LoRA approximation with rank ratio {lora_rank}.")
        if prune_lora and not (prune_n == 2 and prune_m == 4):
            raise NotImplementedError("Pruning LoRA is only supported for 2:4 sparsity ratio")
        if prune_method == "maskllm":
            assert mask_checkpoint is not None, "Mask checkpoint must be provided for MaskLLM pruning"
            assert prune_n == 2 and prune_m == 4, "MaskLLM pruning only supports 2:4 sparsity ratio"
            if not any(hasattr(p, "mask") for p in model.parameters()):
                raise NotImplementedError("downloading MaskLLM masks needs network access; attach the masks as "
                                          "`param.mask` (True = pruned) before calling")
        if prune_method in ("pruner_zero", "prune_gblm"):
            raise NotImplementedError(f"Pruning method {prune_method} is not implemented in slim_amd")
        prune_wanda(model, tokenizer, sparsity_ratio, prune_n, prune_m, quantize_weight, bitwidth, slim_quant,
                    weight_tiled_quantization, weight_tile_size, shift_zero_metrics, lora_rank, slim_lora, prune_lora,
                    quantize_lora, lora_tile_size, separate_lora, nsamples, seed, calibration_dataset, pad_lora,
                    scale_important_weights=scale_important_weights, quantize_first=quantize_first)
    elif prune_method == "magnitude":
        if scale_important_weights and quantize_weight:
            raise NotImplementedError("Scaling important weights not implemented for magnitude pruning and "
                                      "quantization")
        if lora_rank > 0:
            raise NotImplementedError("LoRA approximation not implemented for magnitude pruning")
        raise NotImplementedError("Pruning method magnitude is not implemented in slim_amd")
    elif prune_method == "sparsegpt":
        if scale_important_weights and quantize_weight:
            raise NotImplementedError("Scaling important weights not implemented for magnitude pruning and "
                                      "quantization")
        if lora_rank > 0:
            raise NotImplementedError("LoRA approximation not implemented for SparseGPT")
        if slim_quant:
            raise NotImplementedError("SparseGPT can only support OPTQ (GPTQ) quantization")
        raise NotImplementedError("Pruning method sparsegpt is not implemented in slim_amd")
    elif prune_method == "joint_pq":
        if weight_tiled_quantization is False:
            raise NotImplementedError("Joint pruning and quantization only supports block quantization")
        if slim_quant:
            raise NotImplementedError("Joint pruning and quantization only supports AbsMax")
        if quantize_weight is False:
            raise NotImplementedError("Joint pruning and quantization requires quantizing weights")
        raise NotImplementedError("Pruning method joint_pq is not implemented in slim_amd")
    else:
        raise NotImplementedError(f"Pruning method {prune_method} not implemented")
