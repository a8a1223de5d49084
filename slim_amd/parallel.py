"""Multi-GPU partitioning of the compression path: one process per GPU, torch.distributed (RCCL on the
GPUs, gloo in CPU tests).  The reference has no distributed code; this is new design.

Two partitions of a decoder block's compression.  Layer-level (this file + prune.prune_wanda): the linear layers are
independent once their calibration statistics exist, so they are assigned to ranks by cost; the owner compresses a
linear and broadcasts the result (weight, adapters, scale), because every rank needs the whole compressed block for the
next forward pass.  Row-block level (sharded.py): every linear is cut over all ranks; the collectives it needs are the
staged wrappers below."""
import torch
import torch.distributed as dist


def partition_by_cost(costs, world_size):
    """Longest-processing-time-first: owner[i] = rank that compresses item i.  Deterministic on every rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world_size
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world_size), key=lambda j: (load[j], j))
        owner[i] = r
        load[r] += costs[i]
    return owner


def linear_cost(n, k):
    """Relative cost of compressing an n x k linear: dominated by passes over the n*k matrix."""
    return float(n) * float(k)


def _staged(group):
    """gloo moves host memory only: stage device tensors through the host (CPU tests, single-GPU rehearsals)."""
    return dist.get_backend(group) == "gloo"


def _alone(group):
    return not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1


def broadcast(t, src, group=None):
    if _alone(group):
        return t
    if t.is_cuda and _staged(group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def broadcast_coalesced(tensors, src, group=None):
    """One broadcast for several tensors of any dtypes and shapes (identical lists on every rank; the receivers' tensors
    are overwritten).  The owner copies them into one byte buffer, the receivers copy them out: a compressed decoder block
    is some thirty-five tensors, and a collective costs tens of microseconds before its first byte moves, which is what
    a step of a few milliseconds notices; the extra copies run at HBM speed."""
    tensors = [t for t in tensors if t is not None]
    if _alone(group) or not tensors:
        return
    offs, total = [], 0
    for t in tensors:
        offs.append(total)
        total += -(-t.numel() * t.element_size() // 16) * 16
    flat = torch.empty(total, dtype=torch.uint8, device=tensors[0].device)
    mine = dist.get_rank(group) == src
    if mine:
        for t, o in zip(tensors, offs):
            n = t.numel() * t.element_size()
            flat[o:o + n].copy_(t.contiguous().reshape(-1).view(torch.uint8))
    broadcast(flat, src, group)
    if not mine:
        for t, o in zip(tensors, offs):
            n = t.numel() * t.element_size()
            t.copy_(flat[o:o + n].view(t.dtype).reshape(t.shape))


def _op(op):
    return {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX}[op]


def all_reduce(t, group=None, op="sum"):
    """In place; t is identical on every rank afterwards."""
    if _alone(group):
        return t
    if t.is_cuda and _staged(group):
        h = t.cpu()
        dist.all_reduce(h, op=_op(op), group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=_op(op), group=group)
    return t


def all_reduce_max(t, group=None):
    """all_reduce with MAX (status words: every rank sees the worst)."""
    return all_reduce(t, group, op="max")


def reduce(t, dst, group=None):
    """In place sum onto rank dst (the other ranks' tensors are left undefined)."""
    if _alone(group):
        return t
    if t.is_cuda and _staged(group):
        h = t.cpu()
        dist.reduce(h, dst=dst, group=group)
        if dist.get_rank(group) == dst:
            t.copy_(h)
    else:
        dist.reduce(t, dst=dst, group=group)
    return t


def all_gather(t, group=None):
    """[world, *t.shape]: every rank's t (same shape and dtype everywhere)."""
    t = t.contiguous()
    if _alone(group):
        return t.unsqueeze(0)
    world = dist.get_world_size(group)
    if _staged(group):
        h = t.cpu()
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h, group=group)
        return torch.stack(parts).to(t.device)
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out


def packable(N, K, prune_n, prune_m, bits):
    """Whether a compressed weight of this shape and recipe has the packed 2:4 int4 form (0.375 byte per element)."""
    return prune_n == 2 and prune_m == 4 and bits == 4 and N % 128 == 0 and K % 64 == 0


def exchange_linear(lin, owner, rank, shapes, device, group=None, has_scale=True, scale_dtype=torch.float32,
                    has_left_mask=False, packed=False, defer=None):
    """Broadcast one compressed linear from its owner.  shapes = (N, K, r).  Non-owners allocate receive
    buffers; afterwards every rank holds weight.data [N, K], lora_left [K, r], lora_right [r, N],
    scaling_factor (1, 1) in scale_dtype (fp32 under SLiM-Quant, bf16 under abs-max: what the owner's quantiser
    leaves) and, for pruned adapters, lora_left_mask [K, r] (bool).
    packed=True (a 2:4 sparse 4-bit weight, see packable): the weight travels in its packed form -- codes and positions,
    0.375 byte per element instead of 2 -- and the receivers expand it; pack and unpack are exact inverses on such
    weights up to the sign of zeros (a level that rounds to -0 arrives as +0), so the receivers' tensor equals the
    owner's under ==.
    defer (a list, optional): the tensors are appended to it instead of being broadcast, and the function returns the
    callable that completes the linear once the caller has exchanged them (broadcast_coalesced: one collective for all
    linears of an owner)."""
    from . import ops
    N, K, r = shapes
    dt = lin.weight.dtype
    if owner != rank:
        if not packed:
            lin.weight.data = torch.empty(N, K, dtype=dt, device=device)
        lin.lora_left = torch.nn.Parameter(torch.empty(K, r, dtype=dt, device=device), requires_grad=False)
        lin.lora_right = torch.nn.Parameter(torch.empty(r, N, dtype=dt, device=device), requires_grad=False)
        lin.scaling_factor = torch.empty(1, 1, dtype=scale_dtype, device=device) if has_scale else None
        if has_left_mask:
            lin.lora_left_mask = torch.empty(K, r, dtype=torch.bool, device=device)
    elif has_scale and lin.scaling_factor.dtype != scale_dtype:
        raise TypeError(f"scaling factor of the owner is {lin.scaling_factor.dtype}, the exchange expects {scale_dtype}")
    tensors = [lin.lora_left.data, lin.lora_right.data]
    if has_scale:
        tensors.append(lin.scaling_factor)
    if packed:
        if not has_scale:
            raise ValueError("the packed exchange needs the weight's scaling factor")
        if owner == rank:
            # (the status word -- not 2:4, or not 4-bit levels of this scale -- is left on the module for the caller to
            # read once per block)
            vals, meta, lin.pack_status = ops.pack24(lin.weight.data, lin.scaling_factor.float().reshape(1), check=False)
        else:
            vb, mb = ops.packed24_bytes(N, K)
            vals = torch.empty(vb, dtype=torch.uint8, device=device)
            meta = torch.empty(mb, dtype=torch.uint8, device=device)
        tensors += [vals, meta]
    else:
        tensors.append(lin.weight.data)
    m8 = lin.lora_left_mask.to(torch.uint8) if has_left_mask else None    # bool has no broadcast support in every backend
    if m8 is not None:
        tensors.append(m8)

    def complete():
        if packed and owner != rank:
            lin.weight.data = ops.unpack24(vals, meta, lin.scaling_factor.float().reshape(1), N, K)
        if m8 is not None:
            lin.lora_left_mask = m8.bool()
        return lin
    if defer is not None:
        defer.extend(tensors)
        return complete
    for t in tensors:
        broadcast(t, owner, group)
    return complete()


def gather_sample_stats(local_sumsq, group=None):
    """All-gather per-sample column sums of squares ([n_local, K] fp32 each) into sample order
    [n_local * world, K], so that every rank can apply the scaler_row recurrence in the same order as a
    single-GPU run.  Rank r is expected to hold samples r * n_local ... (r + 1) * n_local - 1."""
    src = local_sumsq.contiguous()
    if _alone(group):
        return src
    world = dist.get_world_size(group)
    if src.is_cuda and _staged(group):
        h = src.cpu()
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h, group=group)
        return torch.cat(parts, 0).to(src.device)
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src, group=group)
    return torch.cat(parts, 0)


def world_info(group=None):
    """(rank, world_size) of the initialised process group, (0, 1) without one."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1
