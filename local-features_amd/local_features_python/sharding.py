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


def cross_image_match(desc_local, image_sizes_local, match_fn, ratio=0.8, group=None):
    """The match stage of BASELINE configs[3]: every rank holds the descriptors of its own images
    (desc_local [n_i,128], image_sizes_local = descriptors per image, in storage order); descriptor shards are
    all-gathered (the path's one collective), and each rank matches ITS descriptors against ALL descriptors
    except those of the same image -- no second exchange, results stay sharded like the queries.

    match_fn(a, b, exclude_lo, exclude_hi, ratio) -> int32 [len(a)] is the engine: the product passes
    `gpu_match_fn(handle)` (lf_mkd_match_device); the CPU rehearsal in tests/ injects the oracle.

    Returns (match [n_i] int64 GLOBAL row indices into the gathered set or -1, gathered descriptors,
    global offset of this rank's first row)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    sizes = torch.as_tensor(list(image_sizes_local), dtype=torch.int64)
    if int(sizes.sum()) != desc_local.shape[0]:
        raise ValueError("image_sizes_local must add up to the number of local descriptors")
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        gathered, counts = all_gather_descriptors(desc_local, group)
    else:
        gathered, counts = desc_local, [desc_local.shape[0]]
    base = sum(counts[:rank])
    starts = base + torch.cumsum(sizes, 0) - sizes                       # global row of each local image's first row
    lo = torch.repeat_interleave(starts, sizes).to(torch.int32)
    hi = torch.repeat_interleave(starts + sizes, sizes).to(torch.int32)
    match = match_fn(desc_local, gathered, lo.to(desc_local.device), hi.to(desc_local.device), ratio)
    return match.to(torch.int64), gathered, base


def gpu_match_fn(handle):
    """Engine for cross_image_match backed by lf_mkd_match_device (tensors on the handle's GPU)."""
    def fn(a, b, lo, hi, ratio):
        a, b = a.contiguous(), b.contiguous()
        lo, hi = lo.contiguous(), hi.contiguous()
        out = torch.empty((a.shape[0],), dtype=torch.int32, device=a.device)
        if a.shape[0]:
            handle.match_device(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], out.data_ptr(), ratio,
                                lo.data_ptr(), hi.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
            torch.cuda.current_stream().synchronize()
        return out
    return fn


def max_over_ranks(seconds, device):
    """bench.py's timing rule: the slowest rank defines the step time."""
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
