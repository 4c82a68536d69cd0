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


def shard_counts(n_local, device, group=None):
    """Rows held by every rank (a tiny all-gather of one integer each)."""
    world = dist.get_world_size(group)
    mine = torch.tensor([int(n_local)], device=device, dtype=torch.int64)
    counts = torch.zeros((world,), device=device, dtype=torch.int64)
    dist.all_gather_into_tensor(counts, mine, group=group)
    return [int(c) for c in counts.tolist()]


def gathered_buffer(counts, rank, width=128, device="cuda", dtype=torch.float32):
    """The final gathered set [sum counts, width] and this rank's own rows in it (a view).  A producer that writes its
    descriptors straight into the view (bench.py does) makes the all-gather copy-free on the sending side too."""
    total = sum(counts)
    buf = torch.empty((total, width), device=device, dtype=dtype)
    lo = sum(counts[:rank])
    return buf, buf[lo:lo + counts[rank]]


def make_comm(handle, group=None):
    """An RCCL communicator of the C boundary (lf_mkd_comm_create) over the ranks of `group`: rank 0 draws the identifier,
    torch.distributed -- whatever its backend -- only carries its 128 bytes to the others.  Collective."""
    from ._lib import Comm, comm_unique_id
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return Comm(handle, box[0], world, rank)


def all_gather_descriptors(desc_local, group=None, mode="direct", out=None, counts=None, comm=None):
    """[n_i,128] per rank -> ([sum n_i,128] on every rank, counts).  Every shard lands at its final place (rank order)
    in ONE buffer: no padding to the largest shard, no second copy.

    mode "direct": every rank sends its shard to each peer and receives each peer's shard in one group of point-to-point
                   transfers (ncclGroupStart .. End under batch_isend_irecv).  xGMI is a full point-to-point mesh, so the
                   7 sends of a rank leave on 7 different links at once; shard sizes may differ.
    mode "ring":   one all_gather_into_tensor (RCCL's ring: every byte crosses world-1 links in turn); needs equal
                   shards, falls back to "direct" otherwise.
    out / counts:  a buffer from gathered_buffer() whose own-rank view already holds desc_local (then nothing is copied
                   locally); counts may be passed when the caller already knows them.
    comm:          a communicator from make_comm(): the gather then goes through the C boundary
                   (lf_mkd_allgather_descriptors: grouped ncclSend / ncclRecv, or ncclAllGather) on torch's current stream --
                   the same call the Rust crate makes; torch.distributed moves no descriptor."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if counts is None:
        counts = shard_counts(desc_local.shape[0], desc_local.device, group)
    if counts[rank] != desc_local.shape[0]:
        raise ValueError("counts[rank] must be the number of local descriptors")
    offs = [sum(counts[:r]) for r in range(world + 1)]
    if out is None:
        out = torch.empty((offs[-1], desc_local.shape[1]), device=desc_local.device, dtype=desc_local.dtype)
    elif out.shape[0] != offs[-1] or out.shape[1] != desc_local.shape[1] or not out.is_contiguous():
        raise ValueError("out must be a contiguous [sum(counts), width] buffer")
    mine = out[offs[rank]:offs[rank + 1]]
    if mine.data_ptr() != desc_local.data_ptr() and counts[rank]:
        mine.copy_(desc_local)
    if comm is not None:
        if not out.is_cuda:
            raise ValueError("the C-boundary gather moves device memory")
        from ._lib import GATHER_DIRECT, GATHER_RING
        if mode not in ("direct", "ring"):
            raise ValueError(f"unknown all-gather mode {mode!r}")
        s = torch.cuda.current_stream(out.device)
        if s.cuda_stream == 0:          # the ABI reads handle 0 as the library's own stream: order the two by hand
            s.synchronize()
        comm.allgather_descriptors(out.data_ptr(), counts, GATHER_RING if mode == "ring" else GATHER_DIRECT,
                                   s.cuda_stream or None)
        return out, counts
    # gloo moves host memory only: a CUDA shard on a gloo group (the one-GPU rehearsal of bench.py) goes through the host
    via_host = out.is_cuda and dist.get_backend(group) == "gloo"
    buf = out.cpu() if via_host else out
    if mode == "ring" and len(set(counts)) == 1 and counts[0] > 0:
        # NCCL / RCCL gathers in place (the shard already sits at its offset in the output); other backends get a copy of
        # the shard.  Decided from the backend alone, before the collective: every rank takes the same branch, whatever
        # happens -- a rank that re-issued a collective after an exception would desynchronise the group.
        mine_in = buf[offs[rank]:offs[rank + 1]]
        if dist.get_backend(group) != "nccl":
            mine_in = mine_in.clone()
        dist.all_gather_into_tensor(buf, mine_in, group=group)
    elif mode in ("direct", "ring"):
        ops = []
        for step in range(1, world):       # peer order staggered by rank: at any step the pairs are disjoint
            dst, src = (rank + step) % world, (rank - step) % world
            if counts[rank]:
                ops.append(dist.P2POp(dist.isend, buf[offs[rank]:offs[rank + 1]], dst, group))
            if counts[src]:
                ops.append(dist.P2POp(dist.irecv, buf[offs[src]:offs[src + 1]], src, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
    else:
        raise ValueError(f"unknown all-gather mode {mode!r}")
    if via_host:
        out.copy_(buf)
    return out, counts


def exclusion_ranges(image_sizes_local, base, device):
    """Per local descriptor, the rows [lo, hi) of the gathered set that belong to its own image (which a cross-image
    match must not consider); `base` = global row of this rank's first descriptor."""
    sizes = torch.as_tensor(list(image_sizes_local), dtype=torch.int64)
    starts = base + torch.cumsum(sizes, 0) - sizes                       # global row of each local image's first row
    lo = torch.repeat_interleave(starts, sizes).to(torch.int32)
    hi = torch.repeat_interleave(starts + sizes, sizes).to(torch.int32)
    return lo.to(device), hi.to(device)


def cross_image_match(desc_local, image_sizes_local, match_fn, ratio=0.8, group=None, mode="direct", out=None, comm=None):
    """The match stage of BASELINE configs[3]: every rank holds the descriptors of its own images
    (desc_local [n_i,128], image_sizes_local = descriptors per image, in storage order); descriptor shards are
    all-gathered (the path's one collective), and each rank matches ITS descriptors against ALL descriptors
    except those of the same image -- no second exchange, results stay sharded like the queries.

    match_fn(a, b, exclude_lo, exclude_hi, ratio) -> int32 [len(a)] is the engine: the product passes
    `gpu_match_fn(handle)` (lf_mkd_match_device); the CPU rehearsal in tests/ injects the oracle.

    Returns (match [n_i] int64 GLOBAL row indices into the gathered set or -1, gathered descriptors,
    global offset of this rank's first row)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if sum(int(s) for s in image_sizes_local) != desc_local.shape[0]:
        raise ValueError("image_sizes_local must add up to the number of local descriptors")
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or comm is not None):
        # (a world of one with a communicator still goes through the transport: lf_mkd_allgather_descriptors is then the
        #  identity on the buffer, but it is the call a larger world makes)
        gathered, counts = all_gather_descriptors(desc_local, group, mode=mode, out=out, comm=comm)
    else:
        gathered, counts = desc_local, [desc_local.shape[0]]
    base = sum(counts[:rank])
    lo, hi = exclusion_ranges(image_sizes_local, base, desc_local.device)
    match = match_fn(desc_local, gathered, lo, hi, ratio)
    return match.to(torch.int64), gathered, base


def gpu_match_fn(handle):
    """Engine for cross_image_match backed by lf_mkd_match_device (tensors on the handle's GPU)."""
    def fn(a, b, lo, hi, ratio):
        a, b = a.contiguous(), b.contiguous()
        lo, hi = lo.contiguous(), hi.contiguous()
        out = torch.empty((a.shape[0],), dtype=torch.int32, device=a.device)
        if a.shape[0]:
            # torch's default stream is handle 0, which the ABI reads as "the library's own (non-blocking) stream": what
            # produced a / b must be complete before, and the library's stream drained after
            torch.cuda.current_stream().synchronize()
            handle.match_device(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], out.data_ptr(), ratio,
                                lo.data_ptr(), hi.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
            torch.cuda.current_stream().synchronize()
            handle.synchronize()
        return out
    return fn


def max_over_ranks(seconds, device):
    """bench.py's timing rule: the slowest rank defines the step time."""
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
