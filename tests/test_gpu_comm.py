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
