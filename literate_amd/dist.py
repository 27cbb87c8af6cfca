"""Chain sharding and trace gathering for one process per GPU (torch.distributed; backend
"nccl" is RCCL on ROCm, "gloo" in the CPU tests).  Chains are independent (the reference pools
them only post hoc, plotRJforward.v3.py:307-350), so the only collective is the gather of the
fixed-width trace rows at sampling time."""
import torch
import torch.distributed as dist


def shard_chains(total_chains, world_size, rank):
    """Contiguous block partition: returns (chain_offset, n_local).  The first `total % world`
    ranks get one extra chain.  Global chain id = chain_offset + local id = the Philox key, so a
    chain's trajectory does not depend on the sharding."""
    base, extra = divmod(int(total_chains), int(world_size))
    n_local = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, n_local


def gather_traces(local, total_chains=None, dst=0, group=None):
    """local: [samples, n_local, width] on this rank.  Returns [samples, total, width] on `dst`
    (chains in global order), None elsewhere.  Ragged shards are padded to the largest one."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if total_chains is None:
        n = torch.tensor([local.shape[1]], device="cpu" if dist.get_backend(group) == "gloo" else local.device)
        dist.all_reduce(n, group=group)
        total_chains = int(n.item())
    n_max = -(-total_chains // world)
    pad = local
    if local.shape[1] < n_max:
        pad = torch.cat([local, local.new_full((local.shape[0], n_max - local.shape[1], local.shape[2]), float("nan"))], 1)
    pad = pad.contiguous()
    # gloo (CPU tests, single-GPU rehearsals of the multi-rank path) gathers host tensors only
    device = pad.device
    if dist.get_backend(group) == "gloo" and pad.is_cuda:
        pad = pad.cpu()
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    parts = [bufs[r][:, :shard_chains(total_chains, world, r)[1]].to(device) for r in range(world)]
    return torch.cat(parts, 1)


def agree_status(status, device, group=None):
    """The largest engine status word over the ranks (0 = every rank is fine).  A rank whose team exchange timed out must
    not leave the others blocked in the next collective: every rank learns of it here and raises alike."""
    if not (dist.is_available() and dist.is_initialized()):
        return int(status)
    t = torch.tensor([int(status)], dtype=torch.int32, device="cpu" if dist.get_backend(group) == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())
