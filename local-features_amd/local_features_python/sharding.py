"""Multi-GPU plumbing of the describe path: one process per GPU, frames shard by image with
no exchange; the only collective is the all-gather of descriptor shards that feeds the cross-image
brute-force match stage.  torch.distributed is used as transport only (nccl = RCCL on GPUs, gloo on CPU)."""
import torch
import torch.distributed as dist


def frames_of_rank(n_frames, rank, world):
    """Frame f belongs to GPU f mod world; returns this rank's frame indices."""
    return list(range(rank, n_frames, world))


def patch_slice_of_rank(n_total, rank, world):
    """Patch mode: contiguous n/world slices, the first (n mod world) ranks take one more."""
    base, extra = divmod(n_total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_descriptors(desc_local, group=None):
    """[n_i,128] per rank -> ([sum n_i,128] on every rank, counts).  Shard sizes differ, so shards are
    padded to the largest one for a single all_gather_into_tensor (one bucket per rank, direct over xGMI
    with RCCL) and compacted afterwards."""
    world = dist.get_world_size(group)
    n_local = torch.tensor([desc_local.shape[0]], device=desc_local.device, dtype=torch.int64)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    padded = torch.zeros((cap, desc_local.shape[1]), device=desc_local.device, dtype=desc_local.dtype)
    padded[:desc_local.shape[0]] = desc_local
    gathered = torch.empty((world * cap, desc_local.shape[1]), device=desc_local.device, dtype=desc_local.dtype)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    parts = [gathered[r * cap:r * cap + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0), counts


def max_over_ranks(seconds, device):
    """bench.py's timing rule: the slowest rank defines the step time."""
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
