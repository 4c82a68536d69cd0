"""The path's one collective on the C boundary (lf_mkd_comm_*, lf_mkd_allgather_descriptors; RCCL bound at run time), with the
one rank a one-GPU box has.  More ranks need more GPUs -- the driver's 8-GPU run; the logic above the transport (shard
offsets, uneven and empty shards, exclusion ranges, cross-image match) is rehearsed over gloo in tests/test_distributed.py."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lfp():
    import local_features_python as m
    return m


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "these tests need the MI355X"
    return t


def test_one_rank_communicator_both_gather_forms(lfp, torch):
    """With one rank both forms are the identity on the gathered buffer -- which still goes through ncclGetUniqueId,
    ncclCommInitRank, the grouped point-to-point form (no peer to post to) and ncclAllGather in place."""
    h = lfp.MkdHandle(max_features=64)
    comm = lfp.Comm(h, lfp.comm_unique_id(), 1, 0)
    version, n_ranks, rank = comm.info()
    assert version >= 20000 and (n_ranks, rank) == (1, 0)
    n = 3000
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.nn.functional.normalize(torch.randn((n, 128), device="cuda", generator=g), dim=1)
    want = x.clone()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for mode in (lfp.GATHER_DIRECT, lfp.GATHER_RING):
        comm.allgather_descriptors(x.data_ptr(), [n], mode, side.cuda_stream)
        side.synchronize()
        assert torch.equal(x, want)
    comm.allgather_descriptors(x.data_ptr(), [n], lfp.GATHER_DIRECT)          # the handle's own stream
    assert torch.equal(x, want)
    comm.allgather_descriptors(0, [0], lfp.GATHER_RING)                       # nothing to gather: not an error
    with pytest.raises(RuntimeError, match="mode"):
        comm.allgather_descriptors(x.data_ptr(), [n], 7)
    with pytest.raises(RuntimeError, match="rank"):
        lfp.Comm(h, lfp.comm_unique_id(), 2, 5)
    comm.close()


def test_loopback_drives_the_grouped_send_recv_branch_on_one_rank(lfp, torch):
    """The DIRECT form of the gather is one group of ncclSend / ncclRecv per peer -- a loop that is empty on a one-rank
    communicator, so that before round 6 those four bound symbols, their argument order and units had never met a real
    librccl.  lf_mkd_comm_loopback posts the same group through the same routine with this rank as its own peer (a self
    send/recv inside a group is legal): the rows must arrive, for no rows (the empty group), one row and the per-GPU share of
    BASELINE configs[3] (2^20 rows = 512 MiB), on the handle's own stream and on a caller's, repeatedly on one communicator,
    and with the gather itself before and after."""
    h = lfp.MkdHandle(max_features=64)
    comm = lfp.Comm(h, lfp.comm_unique_id(), 1, 0)
    assert comm.last_form() == -1
    g = torch.Generator(device="cuda").manual_seed(11)
    side = torch.cuda.Stream()
    for rnd, n in enumerate((0, 1, 1 << 20, 777, 1)):
        src = torch.randn((max(n, 1), 128), device="cuda", generator=g)
        dst = torch.full((max(n, 1) + 2, 128), -7.0, device="cuda")            # guard rows before and after
        torch.cuda.synchronize()
        stream = side.cuda_stream if rnd % 2 else None
        comm.loopback(src.data_ptr(), dst[1:].data_ptr(), n, stream)
        if stream:
            side.synchronize()
        assert torch.equal(dst[1:1 + n], src[:n]), n
        assert bool((dst[0] == -7.0).all()) and bool((dst[1 + n:] == -7.0).all()), n      # exactly n rows were written
        if n == 777:      # a gather on the same communicator in between leaves it usable
            comm.allgather_descriptors(src.data_ptr(), [n], lfp.GATHER_DIRECT)
            assert comm.last_form() == lfp.GATHER_DIRECT
            comm.allgather_descriptors(src.data_ptr(), [n], lfp.GATHER_RING)
            assert comm.last_form() == lfp.GATHER_RING
    with pytest.raises(RuntimeError, match="differ"):
        comm.loopback(src.data_ptr(), src.data_ptr(), 1)
    with pytest.raises(RuntimeError, match="null buffer"):
        comm.loopback(0, 0, 5)
    comm.close()


def test_cross_image_match_over_the_boundary_transport(lfp, torch):
    """sharding.cross_image_match with a communicator of the C boundary (world of one): torch.distributed carries the
    identifier only, the gather is lf_mkd_allgather_descriptors, the result the plain match with own-image exclusion."""
    import torch.distributed as dist
    from local_features_python import sharding
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        h = lfp.MkdHandle(max_features=64)
        comm = sharding.make_comm(h)
        rng = np.random.default_rng(3)
        things = rng.normal(size=(3, 20, 128)).astype(np.float32)
        sizes = [100, 60, 140]
        parts = []
        for i, m in enumerate(sizes):       # image i shows its own 20 things and, perturbed, those of image i-1, plus clutter:
            d = np.concatenate([things[i], things[(i - 1) % 3] + 0.03 * rng.normal(size=(20, 128)),   # every thing is in
                                rng.normal(size=(m - 40, 128))])                                        # exactly two images
            parts.append(d / np.linalg.norm(d, axis=1, keepdims=True))
        local = torch.from_numpy(np.concatenate(parts).astype(np.float32)).cuda()
        buf, mine = sharding.gathered_buffer([len(local)], 0)
        mine.copy_(local)
        with torch.cuda.stream(torch.cuda.Stream()):
            match, gathered, base = sharding.cross_image_match(mine, sizes, sharding.gpu_match_fn(h), out=buf, comm=comm)
        torch.cuda.synchronize()
        assert base == 0 and gathered.data_ptr() == buf.data_ptr() and torch.equal(gathered, local)
        assert comm.calls >= 1          # the gather went through lf_mkd_allgather_descriptors, world of one or not
        m = match.cpu().numpy()
        starts = np.cumsum([0] + sizes)
        img_of = np.repeat(np.arange(3), sizes)
        assert (m >= 0).sum() >= 100
        assert (img_of[m[m >= 0]] != img_of[m >= 0]).all()                    # never the query's own image
        lo, hi = sharding.exclusion_ranges(sizes, 0, "cuda")
        want = torch.empty(len(local), dtype=torch.int32, device="cuda")
        h.match_device(local.data_ptr(), len(local), local.data_ptr(), len(local), want.data_ptr(), 0.8, lo.data_ptr(),
                       hi.data_ptr())
        assert np.array_equal(m, want.cpu().numpy())
        comm.close()
    finally:
        dist.destroy_process_group()


_TWO_RANK_CHILD = r"""
import os, sys
rank, world, port, root = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
sys.path.insert(0, os.path.join(root, "local-features_amd"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch
import torch.distributed as dist
import local_features_python as lfp
from local_features_python import sharding
torch.cuda.set_device(rank)
dist.init_process_group("gloo", rank=rank, world_size=world)          # carries the 128-byte identifier and the checks
h = lfp.MkdHandle(max_features=64, device=rank)
comm = sharding.make_comm(h)
assert comm.info()[1:] == (world, rank)
torch.cuda.set_stream(torch.cuda.Stream())
def rows(r, n):            # rank r's shard: recognisable rows
    return (torch.arange(n, dtype=torch.float32)[:, None] + 1000.0 * r + torch.arange(128, dtype=torch.float32)[None, :] / 256).cuda()
for counts in ([3000, 3000], [4097, 130], [0, 777], [513, 0]):
    want = torch.cat([rows(r, c) for r, c in enumerate(counts)])
    for mode in ("direct", "ring"):
        buf, mine = sharding.gathered_buffer(counts, rank)
        buf.fill_(-1.0)
        mine.copy_(rows(rank, counts[rank]))
        out, got_counts = sharding.all_gather_descriptors(mine, mode=mode, out=buf, counts=counts, comm=comm)
        torch.cuda.synchronize()
        assert got_counts == counts and torch.equal(out, want), (rank, counts, mode)
# and the cross-image match over it equals the single-process answer
g = torch.Generator().manual_seed(7)
allv = torch.nn.functional.normalize(torch.randn((600, 128), generator=g), dim=1)
allv[300:320] = torch.nn.functional.normalize(allv[10:30] + 0.02 * torch.randn((20, 128), generator=g), dim=1)
sizes = [[100, 200], [120, 180]]
lo = 0 if rank == 0 else 300
local = allv[lo:lo + 300].cuda()
match, gathered, base = sharding.cross_image_match(local, sizes[rank], sharding.gpu_match_fn(h), comm=comm)
torch.cuda.synchronize()
assert base == lo and torch.equal(gathered.cpu(), allv)
elo, ehi = sharding.exclusion_ranges(sizes[rank], base, "cuda")
want = torch.empty(300, dtype=torch.int32, device="cuda")
h.match_device(local.data_ptr(), 300, gathered.data_ptr(), 600, want.data_ptr(), 0.8, elo.data_ptr(), ehi.data_ptr())
h.synchronize()
assert torch.equal(match.cpu(), want.cpu().long())
if rank == 1:
    assert int((match[0:20].cpu() == torch.arange(10, 30)).sum()) == 20       # the planted partners, across the rank boundary
comm.close()
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok", flush=True)
"""


def test_two_ranks_over_rccl_uneven_and_empty_shards():
    """lf_mkd_allgather_descriptors with MORE than one rank: two fresh child processes, one GPU each, started
    before anything in them touches the GPU; both forms, equal / uneven / empty shards against the concatenation, and the
    cross-image match over the transport against the direct call.  Needs two GPUs: skipped on a one-GPU box (where the
    grouped send / recv offsets are covered over gloo by tests/test_distributed.py)."""
    import subprocess
    import sys
    import torch as t
    if t.cuda.device_count() < 2:          # (counting devices does not initialise the GPU on this image)
        pytest.skip("needs two GPUs")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", _TWO_RANK_CHILD, str(r), "2", str(port), root], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append("TIMEOUT " + p.communicate()[0])
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
