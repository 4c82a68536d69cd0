"""CPU rehearsal of the N>1 path: two gloo ranks shard frames/patches and all-gather descriptor shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cat(parts):
    return np.concatenate(list(parts) or [np.zeros((0, 128), np.float32)])


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))
    from local_features_python import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # shard 7 frames of different keypoint counts by image
    kp_per_frame = [5, 0, 3, 9, 1, 4, 2]
    mine = sharding.frames_of_rank(len(kp_per_frame), rank, world)
    rng = np.random.default_rng(1234)
    all_desc = [rng.random((k, 128)).astype(np.float32) for k in kp_per_frame]   # same on every rank
    local = torch.from_numpy(_cat(all_desc[f] for f in mine))
    gathered, counts = sharding.all_gather_descriptors(local)           # "direct": point-to-point, uneven shards
    expect = _cat(_cat(all_desc[f] for f in sharding.frames_of_rank(7, r, world)) for r in range(world))
    ok = gathered.shape == (sum(kp_per_frame), 128) and np.array_equal(gathered.numpy(), expect)
    ok = ok and counts == [sum(kp_per_frame[f] for f in sharding.frames_of_rank(7, r, world)) for r in range(world)]
    # the same set into a caller-owned final buffer whose own-rank view already holds the shard (nothing copied locally)
    buf, mine_view = sharding.gathered_buffer(counts, rank, device="cpu")
    mine_view.copy_(local)
    g2, _ = sharding.all_gather_descriptors(mine_view, out=buf, counts=counts)
    ok = ok and g2.data_ptr() == buf.data_ptr() and np.array_equal(buf.numpy(), expect)
    # "ring" (one all_gather_into_tensor) on equal shards, and its fall-back to point-to-point on uneven ones
    eq = torch.full((4, 128), float(rank + 1))
    g3, c3 = sharding.all_gather_descriptors(eq, mode="ring")
    ok = ok and c3 == [4] * world and all(bool((g3[4 * r:4 * r + 4] == r + 1).all()) for r in range(world))
    g4, _ = sharding.all_gather_descriptors(local, mode="ring")
    ok = ok and np.array_equal(g4.numpy(), expect)
    tmax = sharding.max_over_ranks(1.0 + rank, "cpu")
    ok = ok and tmax == float(world)
    # match stage of configs[3]: local queries against the gathered set, own image excluded; the engine is injected
    # (the oracle here, lf_mkd_match_device on the GPUs), the sharding logic is what is rehearsed
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import MkdOracle
    orc = MkdOracle(os.path.join(ROOT, "local-features_amd", "models", "mkd", "concat-pca-liberty.safetensors"))

    def engine(a, b, lo, hi, ratio):
        m = orc.match(a.numpy(), b.numpy(), ratio, exclude=(lo.numpy().view(np.uint32), hi.numpy().view(np.uint32)))[0]
        return torch.from_numpy(m)
    things = rng.normal(size=(7, 3, 128)).astype(np.float32)
    def image(f):   # image f shows 3 things of its own, the 3 things of image f-1 (slightly perturbed), and clutter
        g = np.random.default_rng(100 + f)
        d = np.concatenate([things[f], things[(f - 1) % 7] + 0.02 * g.normal(size=(3, 128)),
                            g.normal(size=(kp_per_frame[f], 128))])
        return (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    imgs = [image(f) for f in range(7)]
    local = torch.from_numpy(_cat(imgs[f] for f in mine))
    match, gathered2, base = sharding.cross_image_match(local, [len(imgs[f]) for f in mine], engine)
    # single-process answer over the same global order
    order = [f for r in range(world) for f in sharding.frames_of_rank(7, r, world)]
    glob = np.concatenate([imgs[f] for f in order])
    sizes = np.array([len(imgs[f]) for f in order])
    starts = np.cumsum(sizes) - sizes
    img_of = np.repeat(np.arange(7), sizes)
    want = orc.match(glob, glob, 0.8, exclude=(starts[img_of].astype(np.uint32),
                                               (starts + sizes)[img_of].astype(np.uint32)))[0]
    ok = ok and np.array_equal(gathered2.numpy(), glob)
    ok = ok and np.array_equal(match.numpy(), want[base:base + len(local)])
    ok = ok and (match.numpy() >= 0).sum() >= 6 * len(mine)                 # every shared thing finds its partner
    sel = match.numpy() >= 0
    ok = ok and (img_of[match.numpy()[sel]] != img_of[base:base + len(local)][sel]).all()
    a, b = sharding.patch_slice_of_rank(1001, rank, world)
    q.put((rank, ok, a, b))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_ranks_shard_and_all_gather(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    base, extra = divmod(1001, world)
    want = [(r * base + min(r, extra), r * base + min(r, extra) + base + (1 if r < extra else 0)) for r in range(world)]
    assert [(a, b) for _, _, a, b in res] == want and want[0][0] == 0 and want[-1][1] == 1001


def test_slices_cover_everything():
    sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))
    from local_features_python import sharding
    for n in (0, 1, 7, 1 << 20):
        for world in (1, 2, 3, 8):
            cuts = [sharding.patch_slice_of_rank(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            frames = sorted(f for r in range(world) for f in sharding.frames_of_rank(n % 97, r, world))
            assert frames == list(range(n % 97))
