"""Multi-GPU layout of the batched solve: independent agents, contiguous shards, one process per
GPU, no collective inside a solve; the only exchange is the final gather of the controls
(SURVEY 8e).  Works on any torch.distributed backend ("nccl" = RCCL over xGMI on the GPU node,
"gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_bounds(B, rank, world):
    """Contiguous split of B agents over `world` ranks; the first B % world ranks get one more."""
    if world < 1 or not (0 <= rank < world) or B < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_GATHER_OK = {}   # (backend, group id) -> bool: decided once, collectively


def gather_supported(group=None, device=None):
    """Does this backend build implement `gather`?  Asked ONCE per process group, by a probe every rank
    takes part in: a one-element gather to rank 0, then an all_reduce(MIN) of "it worked here" so that all
    ranks use the same collective from then on.  (A backend without gather refuses the call on every
    rank alike, before anything is sent; the all_reduce makes the agreement explicit instead of assumed.)
    Errors of the real gather later on are errors: they are raised, never mapped to another collective."""
    key = (dist.get_backend(group), id(group))
    if key in _GATHER_OK:
        return _GATHER_OK[key]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    on_gpu = dist.get_backend(group) == "nccl"
    dev = device if (on_gpu and device is not None) else (torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
    probe = torch.zeros(1, dtype=torch.float64, device=dev)
    ok = 1
    try:
        dst_global = 0 if group is None else dist.get_global_rank(group, 0)
        bufs = [torch.empty_like(probe) for _ in range(world)] if rank == 0 else None
        dist.gather(probe, gather_list=bufs, dst=dst_global, group=group)
    except (RuntimeError, NotImplementedError):
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    _GATHER_OK[key] = bool(int(flag.item()))
    return _GATHER_OK[key]


def gather_controls(local, B_total, dst=0, group=None):
    """Gather row shards [b_r, k] (rank order = agent order) into [B_total, k] on rank `dst`
    (None elsewhere).  Ragged shards are padded to the largest one for the collective."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    k = local.shape[1]
    sizes = [shard_bounds(B_total, r, world) for r in range(world)]
    bmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(bmax, k, dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    if dist.get_backend(group) == "gloo" and pad.is_cuda:
        pad = pad.cpu()   # rehearsal on one GPU box: gloo moves host tensors
    # a gather to `dst` ("no collective beyond a final gather"): every rank sends its shard
    # once, only `dst` receives -- 1/world of the bytes an all_gather would move over xGMI
    if gather_supported(group, local.device if local.is_cuda else None):
        dst_global = dst if group is None else dist.get_global_rank(group, dst)
        bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, gather_list=bufs, dst=dst_global, group=group)
    else:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
    if rank != dst:
        return None
    return torch.cat([bufs[r][:hi - lo] for r, (lo, hi) in enumerate(sizes)], 0).to(local.device)
