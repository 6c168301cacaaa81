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
    dst_global = dst if group is None else dist.get_global_rank(group, dst)
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    try:
        dist.gather(pad, gather_list=bufs, dst=dst_global, group=group)
    except (RuntimeError, NotImplementedError) as exc:
        # a backend build without gather refuses the call on every rank alike, before anything is sent:
        # the all_gather below is then the one code path all ranks take
        if "gather" not in str(exc).lower() and "not supported" not in str(exc).lower() and "implemented" not in str(exc).lower():
            raise
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
    if rank != dst:
        return None
    return torch.cat([bufs[r][:hi - lo] for r, (lo, hi) in enumerate(sizes)], 0).to(local.device)
