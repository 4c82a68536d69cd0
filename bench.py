#!/usr/bin/env python3
"""bench.py -- MKD descriptors/second on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the describe path (lf_mkd_describe_patches_device through the C ABI) over one
batch of synthetic 32x32 f32 patches already resident in HBM.  Default workload: 2^20 patches per GPU
(the per-GPU share of BASELINE.json configs[3]; SURVEY.md 8(d) headline "patch mode").  For N > 1 there is one
process per GPU (torch.distributed, backend nccl = RCCL): either the driver starts them (torch.distributed.run sets
WORLD_SIZE) or, when `--gpus N` is given without that environment, this script starts them itself before it touches
the GPU.  Patches shard by rank with no data-path collective, so scaling is "weak".

After the timed describe steps every rank runs the rest of configs[3] once on the descriptors it just produced, and
rank 0 reports it under "match_stage": the all-gather of the descriptor shards (the path's one collective; both the
point-to-point form and RCCL's ring) and the cross-image brute-force match of the rank's own descriptors against the
gathered set.

Extra objects on the line:
  roofline      the describe kernel (mkd_pool: blur .. whitening .. L2 fused in one launch) vs the
                8 TB/s HBM roof: algorithmic bytes per launch (4608 B x descriptors) /
                its mean launch time from HIP events recorded by the library on the launch stream
                (lf_mkd_kernel_times)
  cpu_baseline  the CPU oracle (a port of the reference's algorithm) timed on this host, rank 0, N=1,
                on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_DESC = 4608          # 4096 B patch read + 512 B descriptor write
FLOP_PER_DESC = 2 * 1024 * 238 + 2 * 238 * 128   # pooling (238 sums over 1024 pixels) + whitening, as f32 multiply-adds
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 matrix peak


def host_cores():
    """Threads for the CPU baseline: the affinity mask, capped by the cgroup CPU quota and by the GPU box's
    per-GPU CPU share (16; its affinity mask shows all 256 host cores but the job only gets 16 of them)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("LF_BENCH_CPU_THREADS", "16"))))


def cpu_baseline(seconds, seed):
    """The CPU legs, the only place bench.py touches oracle/: the CPU-organised port of the path (oracle/mkd_cpu_fast.c: one
    sine / cosine per pixel, pooling as register-blocked AVX2 dot products) on the host cores for about `seconds`, one thread
    beside it, and the shader-structured oracle proper (the parity checker) for comparison."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import MkdOracle
    cores = host_cores()
    o = MkdOracle(os.path.join(ROOT, "local-features_amd", "models", "mkd", "concat-pca-liberty.safetensors"))
    pool = np.random.default_rng(seed).random((65536, 32, 32), dtype=np.float32)      # 256 MiB, swept repeatedly
    o.describe_patches_fast(pool[:4096], nthreads=cores)                              # warm, calibrate
    t0 = time.perf_counter()
    o.describe_patches_fast(pool[:16384], nthreads=cores)
    rate = 16384 / (time.perf_counter() - t0)
    sweeps = max(1, int(round(seconds * rate / len(pool))))
    t0 = time.perf_counter()
    for _ in range(sweeps):
        o.describe_patches_fast(pool, nthreads=cores)
    dt = time.perf_counter() - t0
    n = sweeps * len(pool)
    n1 = max(1024, min(len(pool), int(2.0 * n / dt / max(cores, 1))))                 # one thread, ~2 s
    t1 = time.perf_counter()
    o.describe_patches_fast(pool[:n1], nthreads=1)
    dt1 = time.perf_counter() - t1
    n2 = max(1024, min(len(pool), int(n / dt / 6)))                                   # the oracle proper, ~2 s on all cores
    t2 = time.perf_counter()
    o.describe_patches(pool[:n2], nthreads=cores)
    dt2 = time.perf_counter() - t2
    return {"value": n / dt, "unit": "descriptors/s", "cores": cores, "kind": "port",
            "single_thread_value": n1 / dt1, "shader_structured_oracle_value": n2 / dt2,
            "sample": f"{n} uniform-random 32x32 patches ({sweeps} sweeps of {len(pool)}), {cores} pthreads, {dt:.1f} s, "
                      f"oracle/mkd_cpu_fast.c (AVX2 + FMA, -march=x86-64-v3; single thread: {n1} patches, {dt1:.1f} s); "
                      f"the parity oracle oracle/mkd_oracle.c on the same cores: {n2} patches, {dt2:.1f} s; "
                      "LUTs and whitening matrix built once (the reference rebuilds them per patch)"}


def smooth_frames(torch, count, h, w, sigma, seed):
    """`count` frames of smooth noise in [0, 1] (uniform noise blurred with a Gaussian of `sigma`), synthesised on the GPU
    (keypoint-mode inputs)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.rand((count, 1, h, w), device="cuda", generator=g)
    r = int(3 * sigma)
    k = torch.exp(-0.5 * (torch.arange(-r, r + 1, device="cuda") / sigma) ** 2)
    k /= k.sum()
    x = torch.nn.functional.conv2d(x, k.view(1, 1, 1, -1), padding=(0, r))
    x = torch.nn.functional.conv2d(x, k.view(1, 1, -1, 1), padding=(r, 0))
    lo, hi = x.amin(dim=(2, 3), keepdim=True), x.amax(dim=(2, 3), keepdim=True)
    return ((x - lo) / (hi - lo))[:, 0].contiguous()


def configs3_keypoint_leg(args, lfp, torch, sharding, rank, world, local_rank):
    """BASELINE configs[3] in its own form, this rank's share: frames shard by image (sharding.frames_of_rank: global frame
    f lives on rank f mod N), every rank builds the pyramids of ITS frames only and describes their given
    keypoints -- 128 frames of 1920x1080 with 8192 keypoints each per GPU = 2^20 descriptors, one set_images + one describe
    call per step (the reference's unit of work is an image: vulkan/mod.rs:363-453).  No collective inside: the rank's own
    HIP-event time; the caller gathers the per-rank results."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_golden import random_keypoints
    w, h = 1920, 1080
    nf, nk = args.frames_per_gpu, args.kpts_per_image
    mine = sharding.frames_of_rank(nf * world, rank, world)              # global indices of this rank's frames
    n = nf * nk
    hnd = lfp.MkdHandle(max_features=n, max_image_width=w, max_image_height=h, max_frames=nf, device=local_rank,
                        flags=lfp.FLAG_KERNEL_TIMING)
    # 16 distinct frames per rank (seeded by the global frame index they stand for), each used for nf / 16 of the rank's
    # frames; every frame has its own keypoints (seeded by ITS global index)
    nd = min(nf, 16)
    base = smooth_frames(torch, nd, h, w, 2.0, 3000 + mine[0])
    imgs = base[torch.arange(nf, device="cuda") % nd].contiguous()
    del base
    kps = torch.from_numpy(np.concatenate(
        [np.concatenate([random_keypoints(nk, w, h, 7000 + g, margin=64.0), np.zeros((nk, 1), np.float32)], axis=1)
         for g in mine]).astype(np.float32)).cuda()
    fid = torch.arange(nf, device="cuda", dtype=torch.int32).repeat_interleave(nk).contiguous()
    out = torch.empty((n, 128), device="cuda")
    side = torch.cuda.current_stream()
    s = side.cuda_stream

    def one():
        hnd.set_images_device(imgs.data_ptr(), nf, w, h, s)
        hnd.describe_keypoints_frames_device(kps.data_ptr(), fid.data_ptr(), n, out.data_ptr(), s)
    for _ in range(2):
        one()
    side.synchronize()
    hnd.kernel_times()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = max(2, min(args.steps, 5))
    e0.record(side)
    for _ in range(iters):
        one()
    e1.record(side)
    side.synchronize()
    ms = e0.elapsed_time(e1) / iters
    pool_ms, _, launches = hnd.kernel_times()
    nrm = out.norm(dim=1)
    ok = bool(torch.isfinite(out).all().item()) and float((nrm - 1).abs().max().item()) < 1e-4
    alg = n * (16 + 512) + nf * w * h * 4
    return {"ok": ok, "ms_per_call": ms, "descriptors": n, "frames": nf, "first_global_frames": mine[:3],
            "describe_kernel_ms": pool_ms / max(launches, 1), "algorithmic_bytes_per_call": alg}


def pipeline_extras(lfp, torch, device):
    """Secondary figures (rank 0, N = 1, never allowed to break the headline line): the rows built around the describe
    path, on BASELINE.json's other configurations.  Smooth-noise frames synthesised on the GPU."""
    def frames(count, h, w, sigma, seed):
        return smooth_frames(torch, count, h, w, sigma, seed)

    side = torch.cuda.Stream()       # the library reads a NULL stream as "its own": time on a real one
    out = {}
    with torch.cuda.stream(side):
        s = side.cuda_stream
        # configs[4]: 4K frame, detect + describe as one hipGraph launch per frame
        w, h, top_n = 3840, 2160, 6000
        cap = 2 * top_n
        hnd = lfp.MkdHandle(max_features=cap, max_image_width=w, max_image_height=h, pool_mode=lfp.POOL_F16X3,
                            max_blobs=1 << 16, device=device)
        imgs = frames(2, h, w, 2.5, 7)
        d_img = torch.empty((h, w), device="cuda")
        kps, desc = torch.empty((cap, 5), device="cuda"), torch.empty((cap, 128), device="cuda")
        cnt = torch.zeros((8,), dtype=torch.int64, device="cuda")
        hnd.stream_create(w, h, top_n, 0.0, cap, d_img.data_ptr(), kps.data_ptr(), desc.data_ptr(), cnt.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lat = []
        for f in range(12):
            d_img.copy_(imgs[f % 2])
            e0.record(side)
            hnd.stream_frame(s)
            e1.record(side)
            side.synchronize()
            if f >= 2:
                lat.append(e0.elapsed_time(e1))
        lat.sort()
        out["configs4_4k_frame_hipgraph"] = {"ms_per_frame": lat[len(lat) // 2], "keypoints": int(cnt[3].item()),
                                             "extrema": int(cnt[0].item()), "what": "detect top-6000 + orient + describe"}
        del hnd, imgs, d_img, kps, desc
        # configs[2]: 256 frames 640x480 in one batch, detect + describe
        w, h, top_n, nf = 640, 480, 1400, 256
        cap = 2 * top_n * nf
        hnd = lfp.MkdHandle(max_features=cap, max_image_width=w, max_image_height=h, pool_mode=lfp.POOL_F16X3,
                            max_frames=nf, device=device)
        imgs = frames(nf, h, w, 1.8, 8)
        kps, desc = torch.empty((cap, 5), device="cuda"), torch.empty((cap, 128), device="cuda")
        fo = torch.empty((cap,), dtype=torch.int32, device="cuda")
        run = lambda: hnd.detect_frames_device(imgs.data_ptr(), nf, w, h, top_n, 0.0, kps.data_ptr(), fo.data_ptr(),
                                               desc.data_ptr(), cap, s)
        run()
        side.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            m, _, _ = run()
        side.synchronize()
        dt = (time.perf_counter() - t0) / 3
        out["configs2_256x640x480_batch"] = {"ms_per_batch": dt * 1e3, "keypoints": m, "descriptors_per_s": m / dt,
                                             "what": "detect top-1400 per frame + orient + describe, one call"}
        del hnd, imgs, kps, desc, fo
        # matcher (the stage the configs[3] all-gather feeds): 65536 x 65536 descriptors
        n = 65536
        g = torch.Generator(device="cuda").manual_seed(9)
        a = torch.nn.functional.normalize(torch.randn((n, 128), device="cuda", generator=g), dim=1)
        b = torch.nn.functional.normalize(torch.randn((n, 128), device="cuda", generator=g), dim=1)
        mt = torch.empty(n, dtype=torch.int32, device="cuda")
        hnd = lfp.MkdHandle(max_features=64, device=device)
        hnd.match_device(a.data_ptr(), n, b.data_ptr(), n, mt.data_ptr(), 0.8, stream=s)
        side.synchronize()
        e0.record(side)
        for _ in range(3):
            hnd.match_device(a.data_ptr(), n, b.data_ptr(), n, mt.data_ptr(), 0.8, stream=s)
        e1.record(side)
        side.synchronize()
        ms = e0.elapsed_time(e1) / 3
        # (two passes: every pair is screened with one f16 MFMA term -- 256 flop -- and the few survivors re-scored in f32)
        out["matcher_65536x65536"] = {"ms": ms, "similarities_per_s": n * n / (ms * 1e-3),
                                      "screen_f16_mfma_pflops": n * n * 128 * 2 / (ms * 1e-3) / 1e15}
        del hnd, a, b, mt
        # keypoint mode, describe only: pyramid + sampling + describe of GIVEN keypoints, with the algorithmic
        # bytes of that mode: 16 B keypoint + 512 B descriptor + the frame's bytes spread over its keypoints
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from gen_golden import random_keypoints
        import numpy as np
        kp_traffic = {}
        try:
            kj = json.load(open(os.path.join(ROOT, "profiles", "traffic_keypoint_mode.json")))
            if kj.get("source_sha256") == keypoint_source_stamp():
                kp_traffic = kj.get("hbm_bytes_per_call", {})
        except Exception:
            kp_traffic = {}
        # (third row: the reference's own settings -- examples/match_images/src/main.rs:62-76: top_n 2000, max_features 3000 --
        #  on one 1080p frame: a launch of at most 8192 keypoints takes the 2 + 2-wave form of the describe kernel)
        for tag, w, h, nk, nf, iters in (("configs1_1080p_10k_keypoints", 1920, 1080, 10000, 1, 30),
                                         ("configs2_256x640x480_2k_keypoints", 640, 480, 2000, 256, 5),
                                         ("reference_defaults_1080p_3000_keypoints", 1920, 1080, 3000, 1, 30)):
            n = nk * nf
            hnd = lfp.MkdHandle(max_features=n, max_image_width=w, max_image_height=h, max_frames=nf, device=device)
            imgs = frames(nf, h, w, 2.0, 11)
            base = [np.concatenate([random_keypoints(nk, w, h, 200 + f, margin=64.0 if nf == 1 else 8.0),
                                    np.zeros((nk, 1), np.float32)], axis=1) for f in range(min(nf, 8))]
            kps = torch.from_numpy(np.concatenate([base[f % len(base)] for f in range(nf)]).astype(np.float32)).cuda()
            fid = torch.arange(nf, device="cuda", dtype=torch.int32).repeat_interleave(nk).contiguous()
            o = torch.empty((n, 128), device="cuda")

            def one():
                hnd.set_images_device(imgs.data_ptr(), nf, w, h, s)
                hnd.describe_keypoints_frames_device(kps.data_ptr(), fid.data_ptr(), n, o.data_ptr(), s)
            for _ in range(2):
                one()
            side.synchronize()
            e0.record(side)
            for _ in range(iters):
                one()
            e1.record(side)
            side.synchronize()
            ms = e0.elapsed_time(e1) / iters
            e0.record(side)                     # the describe launch alone (pyramid already built)
            for _ in range(iters):
                hnd.describe_keypoints_frames_device(kps.data_ptr(), fid.data_ptr(), n, o.data_ptr(), s)
            e1.record(side)
            side.synchronize()
            ms_d = e0.elapsed_time(e1) / iters
            alg = n * (16 + 512) + nf * w * h * 4
            ach = alg / (ms * 1e-3) / 1e9
            out[f"keypoint_mode_{tag}"] = {
                "ms_per_call": ms, "descriptors_per_s": n / (ms * 1e-3),
                "describe_only_ms": ms_d, "describe_only_descriptors_per_s": n / (ms_d * 1e-3),
                "what": "set_images (pyramid) + describe of given keypoints (one launch: patches are sampled by producer waves "
                        "of the describe kernel into its LDS ring), one call per batch; describe_only = that launch alone",
                "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                             "traffic": kp_traffic.get(tag), "algorithmic_bytes_per_call": alg,
                             "algorithmic_bytes_per_descriptor": alg / n,
                             "why_traffic_exceeds_algorithmic": WHY_KEYPOINT_TRAFFIC,
                             "kernels": "pyr_* + mkd_pool<.., keypoints> (no patch crosses HBM)"}}
            del hnd, imgs, kps, fid, o
        # the reference's own benchmark (benches/bench.rs:41-112: detect_top_n on houses.jpg 4096 x 3072, feature-count and
        # image-scale sweeps at n_scales 3 and 5) through the call its callers make -- host image in, host results out --
        # as f32 (lf_mkd_detect), as the 8-bit frame it was made from (lf_mkd_detect_u8), and stage by stage as before round 5
        try:
            import bench_reference_sweep as brs
            rows = brs.sweep(full=True, calls=12)
            pcie = 56.0     # GB/s host -> device measured on this pool for pinned and for pageable memory alike (profiles/r05_upload_probe.txt)
            out["reference_bench_houses"] = {
                "what": "benches/bench.rs on tests/golden/houses.jpg (the reference's sample_data/houses.jpg): detect_top_n(image, "
                        "max_features, 0.) with max_blobs = 5 x max_features, host f32 / host u8 frame in, keypoints + descriptors in "
                        "host arrays out; median wall ms of 12 calls; upload / compute / readback = the library's own split "
                        "(lf_mkd_detect_times); every form returns the same bits (checked in the run)",
                "rows": rows,
                "pcie_floor_ms_4096x3072": {"f32": 4096 * 3072 * 4 / pcie / 1e6, "u8": 4096 * 3072 / pcie / 1e6,
                                            "what": "bytes over the 56 GB/s this pool's PCIe link moves (pinned or pageable, DMA or a "
                                                    "kernel reading host memory: profiles/r05_upload_probe*.txt): the f32 call cannot "
                                                    "be shorter than this plus the part of the pipeline that needs the whole frame"}}
        except Exception as e:
            out["reference_bench_houses"] = {"error": f"{type(e).__name__}: {e}"}
        # patch mode at the keypoint counts of configs[1] and configs[2]: the same describe call, smaller n
        for n in (10000, 512000):
            g = torch.Generator(device="cuda").manual_seed(10)
            p = torch.rand((n, 32, 32), device="cuda", generator=g)
            o = torch.empty((n, 128), device="cuda")
            hnd = lfp.MkdHandle(max_features=n, pool_mode=lfp.POOL_F16X3, device=device)
            for _ in range(3):
                hnd.describe_patches_device(p.data_ptr(), n, o.data_ptr(), s)
            side.synchronize()
            e0.record(side)
            for _ in range(10):
                hnd.describe_patches_device(p.data_ptr(), n, o.data_ptr(), s)
            e1.record(side)
            side.synchronize()
            ms = e0.elapsed_time(e1) / 10
            out[f"patch_mode_n{n}"] = {"ms": ms, "descriptors_per_s": n / (ms * 1e-3)}
            del hnd, p, o
    return out


def launch_ranks(n, limit_s=None):
    """`bench.py --gpus N` without a launcher's environment: start the N ranks here, as children of a parent that never
    touches the GPU (a process that has initialised HIP must not be replaced or forked), and pass on their exit code.
    The children get a wall-clock limit (LF_BENCH_LAUNCH_LIMIT_S, default 1500 s): a rank stuck in a collective its peers
    never entered is killed with its whole process group and the parent exits non-zero -- it never re-launches."""
    import signal
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if limit_s is None:
        limit_s = float(os.environ.get("LF_BENCH_LAUNCH_LIMIT_S", "1500"))
    child = subprocess.Popen(cmd, env=env, start_new_session=True)      # its own process group: exactly what we may kill
    try:
        return child.wait(timeout=limit_s)
    except subprocess.TimeoutExpired:
        sys.stderr.write(f"bench.py: the {n} ranks did not finish within {limit_s:.0f} s; killing them\n")
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(child.pid, sig)
            except ProcessLookupError:
                break
            try:
                child.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        return 124


def planted_descriptors(torch, n, per_img, rank, world):
    """Descriptors with cross-image correspondences for the match stage's second leg: image g (global index, per_img rows)
    is random unit vectors, except that its first quarter is the second quarter of image g-1 perturbed at 1e-2 -- so every
    such row has exactly one partner, in a neighbouring image, in either direction (half of all rows) -- and, on rank 0, a
    block of near-duplicates of one vector that crowds the candidate rings of 3 x per_img / 2 query rows (they are redone
    by the full-precision scan; all of it in rows that carry no planted correspondence).  Returns (descriptors [n,128], expected global match or -1 per row, crowded-row mask)."""
    n_img = n // per_img
    g_total = n_img * world
    q = per_img // 4

    def base(g):
        gen = torch.Generator(device="cuda").manual_seed(1000 + g)
        return torch.nn.functional.normalize(torch.randn((per_img, 128), device="cuda", generator=gen), dim=1)

    out = torch.empty((n, 128), device="cuda")
    want = torch.full((n,), -1, dtype=torch.int64, device="cuda")
    ar = torch.arange(q, device="cuda")
    for j in range(n_img):
        g = rank * n_img + j
        d = base(g)
        gen = torch.Generator(device="cuda").manual_seed(5000 + g)
        d[:q] = torch.nn.functional.normalize(base((g - 1) % g_total)[q:2 * q]
                                              + 1e-2 * torch.randn((q, 128), device="cuda", generator=gen) / 128 ** 0.5, dim=1)
        out[j * per_img:(j + 1) * per_img] = d
        want[j * per_img:j * per_img + q] = ((g - 1) % g_total) * per_img + q + ar           # partner in image g-1
        want[j * per_img + q:j * per_img + 2 * q] = ((g + 1) % g_total) * per_img + ar         # partner in image g+1
    crowded = torch.zeros((n,), dtype=torch.bool, device="cuda")
    if rank == 0 and n_img >= 6:
        gen = torch.Generator(device="cuda").manual_seed(77)
        hub = torch.nn.functional.normalize(torch.randn((1, 128), device="cuda", generator=gen), dim=1)
        cl = slice(per_img + 2 * q, per_img + 2 * q + 1024)                      # image 1: 1024 near-duplicates, contiguous
        out[cl] = torch.nn.functional.normalize(hub + 2e-4 * torch.randn((1024, 128), device="cuda", generator=gen), dim=1)
        crowded[cl] = True
        for j in (3, 4, 5):         # 3 x 4096 = 12 288 queries next to the hub: the second halves (no planted rows) of three images
            qs = slice(j * per_img + 2 * q, (j + 1) * per_img)
            out[qs] = torch.nn.functional.normalize(hub + 1e-4 * torch.randn((2 * q, 128), device="cuda", generator=gen), dim=1)
            crowded[qs] = True
    return out, want, crowded


class StageFailed(Exception):
    """A self-check of the match stage failed on some rank; raised on EVERY rank by match_stage's agree()."""


def match_stage(args, lfp, torch, dist, sharding, rank, world, local_rank, rehearsal, gathered, out, n):
    """configs[3] after the describe: descriptor shards -> all-gather -> cross-image brute-force match, once, timed.
    Every rank's `out` is already its own view of `gathered` (the describe wrote there), so the gather copies nothing
    locally.  Images: --kpts-per-image descriptors each, in storage order; a query never matches its own image.
    Two legs: the descriptors the describe step just produced (uniform-random patches: nothing passes the ratio test --
    the screening pass's best case) and descriptors with planted cross-image correspondences plus a crowded block."""
    per_img = max(2, min(args.kpts_per_image, n))
    sizes = [per_img] * (n // per_img) + ([n % per_img] if n % per_img else [])
    counts = [n] * world
    res = {"ranks_seen": world, "world_size": dist.get_world_size() if dist is not None else 1,
           "descriptors_per_rank": n, "keypoints_per_image": per_img,
           "gathered_bytes": world * n * 512, "received_bytes_per_rank": (world - 1) * n * 512}
    hm = lfp.MkdHandle(max_features=64, device=local_rank)
    s = torch.cuda.current_stream().cuda_stream

    def per_rank(value):
        """every rank's value, in rank order (a tiny object gather)"""
        if dist is None:
            return [value]
        box = [None] * world
        dist.all_gather_object(box, value)
        return box

    def timed(fn):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        mine = (time.perf_counter() - t0) * 1e3
        return sharding.max_over_ranks(mine / 1e3, "cpu" if rehearsal else "cuda") * 1e3, mine

    def agree(ok_local, what):
        """Every rank calls this at the same points with its own verdict: one all-reduce (MIN) of an ok flag, so that a
        failed self-check on ONE rank makes EVERY rank leave the stage together instead of the failing rank stopping to
        issue collectives while its peers wait in the next one."""
        good = bool(ok_local)
        if dist is not None:
            flag = torch.tensor([1 if ok_local else 0], dtype=torch.int32, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            good = bool(int(flag.item()))
        if not good:
            raise StageFailed(what + ("" if ok_local else f" [failed on rank {rank}]"))

    # Before anything depends on it: the point-to-point branch of the gather against the librccl this process finds.  A
    # one-rank communicator of this rank's own (lf_mkd_comm_loopback: one group of ncclSend + ncclRecv to itself, through
    # the routine the DIRECT form posts its transfers with) moves a shard-sized block and the rows must arrive -- at N = 1
    # too, where the gather itself has no peer to talk to.
    def loopback(c, rows, tag):
        src = out[:rows]
        dst = torch.full((rows + 1, 128), -3.0, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c.loopback(src.data_ptr(), dst.data_ptr(), rows, s)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        good = bool(torch.equal(dst[:rows], src)) and bool((dst[rows] == -3.0).all().item())
        res[tag] = {"rows": rows, "ms": round(ms, 3), "arrived": good, "gb_s": rows * 512 / ms / 1e6}
        return good
    ok_loop = True
    try:
        self_comm = lfp.Comm(hm, lfp.comm_unique_id(), 1, 0)
        res["rccl_version"] = self_comm.info()[0]
        ok_loop = loopback(self_comm, n, "rccl_loopback_own_communicator")
        self_comm.close()
    except Exception as e:
        res["rccl_loopback_error"] = f"{type(e).__name__}: {e}"
        ok_loop = False
    agree(ok_loop, "RCCL loopback (grouped ncclSend / ncclRecv to this rank): the rows did not arrive")
    comm = None
    if world > 1:
        # the gather goes through the C boundary (lf_mkd_allgather_descriptors over RCCL: what the Rust crate calls);
        # torch.distributed carries the 128-byte identifier only.  (Rehearsal on one GPU: gloo through the host.)
        if not rehearsal:
            try:
                comm = sharding.make_comm(hm)
                res["rccl_version"] = comm.info()[0]
            except Exception as e:
                res["comm_error"] = f"{type(e).__name__}: {e}"
                comm = None
            # every rank takes the same branch (sharding.py): the C-boundary transport only if EVERY rank has its
            # communicator; otherwise those that have one close it and all fall back to torch.distributed together
            made = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(made, op=dist.ReduceOp.MIN)
            if not int(made.item()) and comm is not None:
                comm.close()
                comm = None
                res["comm_error"] = "another rank could not create its communicator"
        res["allgather_transport"] = "lf_mkd_allgather_descriptors (RCCL, C boundary)" if comm else "torch.distributed"
        if comm:      # the same self-transfer on the N-rank communicator the gather is about to use
            try:
                ok_gc = loopback(comm, min(n, 65536), "rccl_loopback_gather_communicator")
            except Exception as e:          # (every rank still reaches agree(): nobody is left alone in a collective)
                res["rccl_loopback_gather_communicator"] = {"error": f"{type(e).__name__}: {e}"}
                ok_gc = False
            agree(ok_gc, "RCCL loopback on the gather's communicator: the rows did not arrive")
        res["gather_form_run"] = {}
        for mode in ("direct", "ring"):
            run = lambda: sharding.all_gather_descriptors(out, mode=mode, out=gathered, counts=counts, comm=comm)
            run()                                                       # first call sets up the communicator's channels
            best, mine = min(timed(run) for _ in range(3))
            res[f"allgather_{mode}_ms"] = best
            res[f"allgather_{mode}_ms_per_rank"] = per_rank(round(mine, 3))
            res[f"allgather_{mode}_gbs_per_rank"] = res["received_bytes_per_rank"] / best / 1e6
            # which form the library ran for this request (RING falls back to DIRECT on unequal shards)
            res["gather_form_run"][mode] = ({lfp.GATHER_DIRECT: "direct (grouped ncclSend / ncclRecv)", lfp.GATHER_RING: "ring (ncclAllGather)"}
                                            .get(comm.last_form(), "?") if comm else "torch.distributed all_gather")
        res["allgather_ms"] = min(res["allgather_direct_ms"], res["allgather_ring_ms"])
        # priced against xGMI: a rank receives (N - 1) shards, and in the direct form they arrive on N - 1 of its 7 point-to-point
        # links at once (~153 GB/s per link and direction); a ring moves the same bytes
        # over ONE link per hop, so its ceiling is a single link's rate
        link = 153.0
        ach = res["received_bytes_per_rank"] / res["allgather_direct_ms"] / 1e6
        res["allgather_roofline"] = {"bound": "xgmi", "achieved": ach, "peak": link * min(world - 1, 7), "unit": "GB/s",
                                     "frac": ach / (link * min(world - 1, 7)),
                                     "what": f"direct form (grouped send/recv): bytes a rank receives over its time; peak = "
                                             f"{min(world - 1, 7)} links x {link:.0f} GB/s; the ring form's ceiling is one link "
                                             f"({link:.0f} GB/s)", "ring_achieved": res["received_bytes_per_rank"] / res["allgather_ring_ms"] / 1e6,
                                     "rehearsal": bool(rehearsal)}
        # every shard must have arrived: row norms of the whole gathered set are 1
        nrm = gathered.norm(dim=1)
        agree(bool(((nrm - 1).abs() < 1e-4).all().item()), "gathered descriptors are not all unit norm (a shard did not arrive)")
    else:
        res["allgather_ms"] = 0.0
    lo, hi = sharding.exclusion_ranges(sizes, rank * n, "cuda")
    m = torch.empty((n,), dtype=torch.int32, device="cuda")
    best = torch.empty((n,), device="cuda")

    def leg(tag, a, b_all):
        run = lambda: hm.match_device(a.data_ptr(), n, b_all.data_ptr(), world * n, m.data_ptr(), 0.8, lo.data_ptr(),
                                      hi.data_ptr(), best.data_ptr(), None, s)
        run()                                                           # warm: the first call allocates the scratch
        ms, mine = timed(run)
        r = {"match_ms": ms, "match_ms_per_rank": per_rank(round(mine, 3)),
             "rows_redone_by_full_scan": int(hm.match_overflowed(s))}
        # the match stage is matrix-core work: one 128-long f16 dot product per pair in the screening pass (NOTEBOOK.md 4d)
        tf = 2.0 * 128 * float(n) * n * world / (ms * 1e-3) / 1e12          # per GPU
        r["roofline"] = {"bound": "mfma", "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / MFMA_F16_PEAK_TFLOPS, "flop_per_pair": 256, "traffic": None,
                         "what": "per GPU, whole match call (split + screen + verify + redone rows) on its clock"}
        r["similarities_per_s"] = float(n) * n * world * world / (ms * 1e-3)
        # accepted matches point outside the query's own image, and every best similarity is a valid cosine
        acc = m >= 0
        bad = acc & (m >= lo) & (m < hi)
        agree(not bool(bad.any().item()) and bool((best.abs() <= 1.0 + 1e-4).all().item()),
              f"cross-image match ({tag}) returned a candidate inside the query's own image")
        r["accepted_fraction"] = float(acc.float().mean().item())
        return r, acc

    # leg 1: the descriptors of the timed describe steps.  Nothing passes ratio 0.8 there (accepted_fraction 0), so the
    # exclusion rule is checked on a second call with ratio 0 -- every row then returns its best candidate -- and a few
    # hundred rows of that call against a plain matrix product over the gathered set
    r1, _ = leg("random", out, gathered)
    hm.match_device(out.data_ptr(), n, gathered.data_ptr(), world * n, m.data_ptr(), 0.0, lo.data_ptr(), hi.data_ptr(),
                    best.data_ptr(), None, s)
    torch.cuda.synchronize()
    agree(bool((m >= 0).all().item()) and not bool(((m >= lo) & (m < hi)).any().item()),
          "with ratio 0 every row must return a candidate outside its own image")
    pick = torch.arange(0, n, max(1, n // 64), device="cuda")[:64]
    sim = out[pick] @ gathered.T
    for i, row in enumerate(pick.tolist()):
        sim[i, int(lo[row]):int(hi[row])] = -2.0
    ref = sim.argmax(dim=1)
    same = (ref == m[pick].long()) | ((sim.gather(1, m[pick].long()[:, None])[:, 0] - sim.max(dim=1).values).abs() < 2e-6)
    agree(bool(same.all().item()), "the matcher's best candidate differs from a plain matrix product")
    res.update(r1)
    res["what"] = ("each rank: its descriptors x the gathered set, own image excluded, ratio 0.8 "
                   "(examples/match_images/src/main.rs:8-27); ms = slowest rank; best candidates checked at ratio 0")
    # leg 2: planted correspondences and a crowded block (gathered the same way)
    if n % per_img == 0 and n // per_img >= 6:
        mine_p, want, crowded = planted_descriptors(torch, n, per_img, rank, world)
        gathered_p, view = sharding.gathered_buffer([n] * world, rank)
        view.copy_(mine_p)
        del mine_p
        if world > 1:
            sharding.all_gather_descriptors(view, mode="direct", out=gathered_p, counts=counts, comm=comm)
        r2, acc = leg("planted", view, gathered_p)
        planted = want >= 0
        hit = (m.long() == want) & planted
        r2["planted_rows_fraction"] = float(planted.float().mean().item())
        r2["planted_rows_recovered"] = float(hit.sum().item()) / max(1.0, float(planted.sum().item()))
        r2["crowded_query_rows"] = int(crowded.sum().item())
        r2["what"] = ("image g's first quarter = image g-1's second quarter perturbed at 1e-2 (one partner per planted row, "
                      "either direction); rank 0 also holds 1024 near-duplicates of one vector and 12 288 queries next to it, "
                      "whose candidate rings overflow: those rows are redone by the full-precision scan")
        agree(r2["planted_rows_recovered"] >= 0.999,
              f"only {r2['planted_rows_recovered']:.4f} of the planted correspondences were matched")
        res["planted"] = r2
        del gathered_p, view
    if comm is not None:
        comm.close()
    return res


def headline_line(args, world, n, total, dt, kern_s, achieved, alt, alt_f32, alt_fp6, clock_mhz, wg0_ms):
    """The bench line without its match_stage / pipelines / cpu_baseline objects."""
    # HBM traffic of the kernel comes from rocprofv3 PMC passes (tools/profile_round.sh), which cannot run inside
    # this process: the committed figure is quoted only if it was measured on this very kernel source and workload
    traffic = mfma_busy = projection = counters_from = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if (tj.get("patches") == n and tj.get("pool") == args.pool and tj.get("angle") == args.angle
                    and tj.get("source_sha256") == source_stamp()):
                traffic = tj.get("hbm_bytes_per_launch")
                mfma_busy = tj.get("mfma_busy_frac")          # SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles)
                projection = tj.get("projection")             # the whitening stage on its own (phase clocks)
                counters_from = f"profiles/traffic_latest.json@{tj.get('source_sha256')} (rocprofv3 --pmc, builder's box)"
        except Exception:
            traffic = None
    line = {
        "metric": "MKD descriptors/sec (32x32 patch, 128-D)", "value": total / dt, "unit": "descriptors/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if args.pool == "f32" else "f32 (f16x3 split MFMA pooling)",
        "data": "synthetic",
        "config": {"workload": f"patch mode: {n} uniform-random 32x32 f32 patches per GPU resident in HBM "
                               "(per-GPU share of BASELINE configs[3]; SURVEY 8(d) headline), PCA=liberty",
                   "patches_per_gpu": n, "angle_mode": args.angle, "pool_mode": args.pool,
                   "parallelism": f"shard-by-rank x{world}, no collective in the describe path"},
        "exact_zero_angle_mode_value": alt,
        "f32_pool_mode_value": alt_f32,
        "fp6_cross_pool_mode_value": alt_fp6,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "mkd_pool", "kernel_ms": kern_s * 1e3,
                     "algorithmic_bytes_per_launch": BYTES_PER_DESC * n,
                     # the clock the chip held under this very kernel (in-kernel stamps: shader-clock counter over the
                     # constant 100 MHz counter; MI355X_MICROARCH.md DVFS give-back) -- boxes differ by several per cent
                     "shader_clock_mhz": clock_mhz, "workgroup0_ms": wg0_ms,
                     "mfma_busy_frac": mfma_busy, "projection": projection,
                     # achieved / kernel_ms / shader_clock_mhz are measured in THIS run; traffic, mfma_busy_frac and
                     # projection are quoted from the committed rocprofv3 PMC passes of the same kernel source
                     "counters_from": counters_from,
                     # NOT a roofline of the kernel that ran (it issues f16 MFMAs): where an f32-input MFMA formulation of
                     # the same arithmetic would be capped -- the path's nominal f32 flops over the f32 matrix peak
                     "f32_formulation_bound": {"what": "bound of an f32-MFMA formulation (not the one that ran): nominal f32 "
                                                       "flops of the path / this launch's time, against the f32 matrix peak",
                                               "flop_per_descriptor": FLOP_PER_DESC,
                                               "nominal_tflops": FLOP_PER_DESC * n / kern_s / 1e12, "f32_mfma_peak": 157.3,
                                               "unit": "TFLOP/s"}},
        "match_stage": None,
    }
    return line


def source_stamp():
    """What the traffic counters in profiles/traffic_latest.json were measured on: sha256 of the describe kernel's source."""
    import hashlib
    hsh = hashlib.sha256()
    for f in ("mkd_describe.hip", "mkd_sample.h", "mkd_device.h"):
        hsh.update(open(os.path.join(ROOT, "local-features_amd", "csrc", f), "rb").read())
    return hsh.hexdigest()[:16]


WHY_KEYPOINT_TRAFFIC = ("the algorithmic bytes count a frame once and 528 B per keypoint; the counters see the pyramid written and "
                        "read (1.33 x the frame, plus level 0 re-read for level 1) and the producers' 8-byte gathers of 1024 bilinear "
                        "taps per keypoint from a pyramid larger than L2.  Sorting the keypoints by (frame, level, 64 x 64 tile) cuts "
                        "the describe kernel's FETCH_SIZE by 27 % (configs[3]) / 46 % (configs[1]) and its time by < 0.5 % "
                        "(profiles/r05_gather_locality.txt): the kernel is bound by instruction issue, not by these bytes, so no "
                        "binning pass was added")


def keypoint_source_stamp():
    import hashlib
    hsh = hashlib.sha256()
    for f in ("mkd_describe.hip", "mkd_pyramid.hip", "mkd_sample.h", "mkd_device.h", "lf_mkd.cpp"):
        hsh.update(open(os.path.join(ROOT, "local-features_amd", "csrc", f), "rb").read())
    return hsh.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patches", type=int, default=1 << 20, help="patches per GPU per step")
    ap.add_argument("--angle", choices=["shader", "exact", "exact_zero"], default="shader")
    ap.add_argument("--pool", choices=["f32", "f16x3"], default=os.environ.get("LF_MKD_POOL", "f16x3"))
    ap.add_argument("--cpu-sample", type=int, default=-1, help="seconds of CPU work for the CPU baseline (default ~12; 0: skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary pipeline figures")
    ap.add_argument("--no-match", action="store_true", help="skip the all-gather + cross-image match stage")
    ap.add_argument("--kpts-per-image", type=int, default=8192, help="descriptors per image in the match stage and "
                    "keypoints per frame in the configs[3] keypoint-mode leg (configs[3]: 8M keypoints over 1024 frames)")
    ap.add_argument("--frames-per-gpu", type=int, default=128, help="1920x1080 frames per GPU in the configs[3] "
                    "keypoint-mode leg (configs[3]: 1024 frames over 8 GPUs); 0: skip the leg")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))       # nothing has touched the GPU yet in this process

    import torch
    import local_features_python as lfp

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the describe path has no CPU fallback")
    # LF_BENCH_REHEARSAL=1 (development only): ranks share the visible GPUs and meet over gloo, so that the N > 1 code
    # path can be run on a one-GPU box; the figures of such a run mean nothing
    rehearsal = os.environ.get("LF_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # Everything below runs on ONE explicit stream: the ABI reads stream handle 0 (which is what torch's default stream
    # reports) as "the library's own stream", which is not ordered with torch's work.
    torch.cuda.set_stream(torch.cuda.Stream())
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    n = args.patches
    gen = torch.Generator(device="cuda").manual_seed(0x4D4B44 + rank)
    patches = torch.rand((n, 32, 32), device="cuda", generator=gen)
    from local_features_python import sharding
    # the descriptors are written straight into this rank's rows of the buffer the match stage gathers into
    gathered, out = sharding.gathered_buffer([n] * world, rank)
    h = lfp.MkdHandle(pca="liberty", max_features=n, device=local_rank,
                      angle_mode={"shader": lfp.ANGLE_SHADER, "exact": lfp.ANGLE_EXACT,
                                  "exact_zero": lfp.ANGLE_EXACT_ZERO}[args.angle],
                      pool_mode=lfp.POOL_F32 if args.pool == "f32" else lfp.POOL_F16X3,
                      flags=lfp.FLAG_KERNEL_TIMING)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        h.describe_patches_device(patches.data_ptr(), n, out.data_ptr(), stream)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    h.kernel_times()                       # drop warm-up events
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    pool_ms, whiten_ms, launches = h.kernel_times()
    clock_mhz, wg0_ms = h.kernel_clock(stream)     # stamps of the last timed launch's workgroup 0
    dt_mine, kern_mine = dt, pool_ms / 1e3 / max(launches, 1)
    dt = sharding.max_over_ranks(dt, "cpu" if rehearsal else "cuda")
    # every rank's own figures, so that a straggler is visible on the N > 1 line without a second run
    per_rank_headline = [{"rank": rank, "descriptors_per_s": n * args.steps / dt_mine, "kernel_ms": kern_mine * 1e3}]
    if dist is not None:
        per_rank_headline = [None] * world
        dist.all_gather_object(per_rank_headline, {"rank": rank, "descriptors_per_s": n * args.steps / dt_mine,
                                                   "kernel_ms": kern_mine * 1e3})

    # secondary figure: the same workload with the exact gradient direction (plus the shader's angle 0 at gx == 0)
    # instead of the shader's polynomial atan2: LF_MKD_ANGLE_EXACT_ZERO, within 1e-4 of the shader reference on every
    # patch (tests/test_gpu_parity.py) -- the headline stays on the most faithful mode
    def side_figure(angle_mode, pool_mode):
        h2 = lfp.MkdHandle(pca="liberty", max_features=n, device=local_rank, angle_mode=angle_mode, pool_mode=pool_mode)
        out2 = torch.empty_like(out)
        for _ in range(2):
            h2.describe_patches_device(patches.data_ptr(), n, out2.data_ptr(), stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            h2.describe_patches_device(patches.data_ptr(), n, out2.data_ptr(), stream)
        torch.cuda.synchronize()
        return n * args.steps / (time.perf_counter() - t1)

    alt = alt_f32 = alt_fp6 = None
    if world == 1 and args.angle == "shader" and args.pool == "f16x3":
        alt = side_figure(lfp.ANGLE_EXACT_ZERO, lfp.POOL_F16X3)
        # the same workload with the pooling contraction in exact f32 arithmetic (LF_MKD_POOL_F32, the verification mode)
        alt_f32 = side_figure(lfp.ANGLE_SHADER, lfp.POOL_F32)
        # ... and with the harmonics' cross terms in e2m3 (LF_MKD_POOL_F16_FP6: round 4's formulation experiment, kept as a
        # mode; 51 instead of 81 matrix instructions per wave-row at ~6 x the default mode's error, NOTEBOOK.md section 11)
        alt_fp6 = side_figure(lfp.ANGLE_SHADER, lfp.POOL_F16_FP6)

    # sanity on the timed output: finite, unit norm (a wrong-but-fast kernel must not pass silently)
    nrm = out.norm(dim=1)
    ok = bool(torch.isfinite(out).all().item()) and float((nrm - 1).abs().max().item()) < 1e-4
    if not ok:
        raise SystemExit("bench.py: descriptors are not finite / unit norm")

    # The headline is complete at this point; what follows (the collective and the match stage, then the secondary pipelines)
    # must never cost it.  An exception is reported in place of the stage; a HANG -- a rank stuck in a collective its peers
    # never entered: no N > 1 run of this code has ever met real RCCL before the driver's -- is cut by a watchdog: rank 0
    # prints the line with the stage marked as timed out, and every rank leaves without waiting for the others.
    total = n * world * args.steps
    kern_s = pool_ms / 1e3 / max(launches, 1)
    achieved = BYTES_PER_DESC * n / kern_s / 1e9 if kern_s > 0 else 0.0
    line = {}

    def emit_and_leave(reason, key="match_stage"):
        """Watchdog: a rank stuck in a collective its peers never entered.  Rank 0 prints the headline line with the stage
        marked, then every rank leaves at once with a NON-ZERO status (no destroy_process_group: it would wait for the stuck
        collective; os._exit from this timer thread ends the process without re-executing anything)."""
        if rank == 0:
            line[key] = {"error": reason}      # the stage that hung, under its own key
            print(json.dumps(line), flush=True)
        sys.stderr.write(f"bench.py rank {rank}: {reason}\n")
        sys.stderr.flush()
        os._exit(3)

    import threading
    # the other ranks leave a little later than rank 0, so that the launcher does not tear rank 0 down before it has printed
    limit = float(os.environ.get("LF_BENCH_STAGE_LIMIT_S", "300")) + (0.0 if rank == 0 else 20.0)
    stage = None
    failed = None          # why this run must end with a non-zero status although its line was printed
    desync = False         # an exception on this rank alone: its peers may be waiting in a collective
    if rank == 0:
        line.update(headline_line(args, world, n, total, dt, kern_s, achieved, alt, alt_f32, alt_fp6, clock_mhz, wg0_ms))
        line["per_rank"] = {"descriptors_per_s": [round(r["descriptors_per_s"]) for r in per_rank_headline],
                            "kernel_ms": [round(r["kernel_ms"], 4) for r in per_rank_headline],
                            "what": "each rank's own 2^20-patch steps over its own wall time between the two fences, and its own "
                                    "mean describe-kernel time (HIP events); value = all ranks' descriptors over the slowest rank's time"}

    # configs[3] in its own form (keypoint mode, frames sharded by image): every rank runs its share, with no collective
    # inside; one object gather afterwards, reached by every rank whatever happened locally
    if args.frames_per_gpu > 0:
        dog = threading.Timer(limit, emit_and_leave, args=(f"configs[3] keypoint-mode leg did not finish within {limit:.0f} s", "configs3_keypoint_mode"))
        dog.daemon = True
        dog.start()
        try:
            leg = configs3_keypoint_leg(args, lfp, torch, sharding, rank, world, local_rank)
        except Exception as e:
            leg = {"ok": False, "error": f"{type(e).__name__}: {e}"}
        legs = [leg]
        if dist is not None:
            legs = [None] * world
            dist.all_gather_object(legs, leg)
        dog.cancel()
        torch.cuda.empty_cache()
        if rank == 0:
            good = [g for g in legs if g.get("ok")]
            kp = {"what": "BASELINE configs[3] as written, per GPU: frames sharded by image (frame f on rank f mod N), "
                          f"{args.frames_per_gpu} frames 1920x1080 x {args.kpts_per_image} given keypoints, keypoint mode: "
                          "set_images (pyramids of the rank's own frames) + describe (one launch, patches sampled inside the "
                          "describe kernel), one call each per step; every rank's own HIP-event time, no collective",
                  "ranks_ok": len(good), "ranks": world,
                  "ms_per_call_per_rank": [round(g["ms_per_call"], 3) if g.get("ok") else None for g in legs],
                  "descriptors_per_s_per_rank": [round(g["descriptors"] / (g["ms_per_call"] * 1e-3)) if g.get("ok") else None
                                                 for g in legs],
                  "describe_kernel_ms_per_rank": [round(g["describe_kernel_ms"], 3) if g.get("ok") else None for g in legs],
                  "errors": [g.get("error") for g in legs if not g.get("ok")] or None}
            if len(good) == world:
                slow = max(g["ms_per_call"] for g in good)
                ndesc = sum(g["descriptors"] for g in good)
                alg = good[0]["algorithmic_bytes_per_call"]
                ach = alg / (slow * 1e-3) / 1e9
                kp_traffic = None        # PMC counters of the same call, quoted if they were taken on this very source
                try:
                    kj = json.load(open(os.path.join(ROOT, "profiles", "traffic_keypoint_mode.json")))
                    if (kj.get("source_sha256") == keypoint_source_stamp() and args.frames_per_gpu == 128
                            and args.kpts_per_image == 8192):
                        kp_traffic = kj.get("hbm_bytes_per_call", {}).get("configs3_128x1080p_8192_keypoints")
                except Exception:
                    kp_traffic = None
                kp.update({"descriptors_per_s": ndesc / (slow * 1e-3), "descriptors_per_s_per_gpu": ndesc / world / (slow * 1e-3),
                           "describe_only_descriptors_per_s_per_gpu":
                               good[0]["descriptors"] / (max(g["describe_kernel_ms"] for g in good) * 1e-3),
                           "frames_of_rank0": good[0]["first_global_frames"],
                           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": ach / HBM_PEAK_GBS, "traffic": kp_traffic,
                                        "counters_from": "profiles/traffic_keypoint_mode.json (rocprofv3 --pmc, builder's box)"
                                                         if kp_traffic else None,
                                        "algorithmic_bytes_per_call": alg,
                                        "algorithmic_bytes_per_descriptor": alg / good[0]["descriptors"],
                                        "why_traffic_exceeds_algorithmic": WHY_KEYPOINT_TRAFFIC,
                                        "what": "per GPU: 528 B per keypoint (16 in, 512 out) + the frames' bytes, over the "
                                                "slowest rank's set_images + describe time",
                                        "kernels": "pyr_* + mkd_pool<.., keypoints> (no patch crosses HBM)"}})
            else:
                failed = "configs[3] keypoint-mode leg failed on a rank"
            line["configs3_keypoint_mode"] = kp

    if not args.no_match:
        dog = threading.Timer(limit, emit_and_leave, args=(f"match stage did not finish within {limit:.0f} s",))
        dog.daemon = True
        dog.start()
        try:
            stage = match_stage(args, lfp, torch, dist, sharding, rank, world, local_rank, rehearsal, gathered, out, n)
        except StageFailed as e:         # a failed self-check, agreed by every rank (match_stage.agree): all are in step
            stage = {"error": f"FAILED CHECK: {e}"}
            failed = f"match stage failed a self-check: {e}"
            sys.stderr.write(f"bench.py rank {rank}: {failed}\n")
        except Exception as e:           # on this rank alone: reported in the stage's place; the peers may be mid-collective
            stage = {"error": f"{type(e).__name__}: {e}"}
            failed = f"match stage raised {type(e).__name__}"
            desync = world > 1
        finally:
            dog.cancel()

    if rank == 0:
        line["match_stage"] = stage
        if world == 1 and not args.no_extras:
            del patches, out, gathered
            torch.cuda.empty_cache()
            try:
                line["pipelines"] = pipeline_extras(lfp, torch, local_rank)
            except Exception as e:       # secondary figures must never cost the headline
                line["pipelines"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1:
            if args.cpu_sample != 0:                                  # seconds of CPU work (default about 12; 0: skip)
                line["cpu_baseline"] = cpu_baseline(12.0 if args.cpu_sample < 0 else float(args.cpu_sample), 0x4D4B44)
        print(json.dumps(line), flush=True)
    # The line is out.  A failed stage ends the run with status 3 so that the driver's rc shows it (the line itself carries
    # the reason).  If only this rank failed, its peers may sit in a collective: leave without the group's teardown, which
    # would wait for them (their own watchdogs end them, also with 3).  Never a re-exec: a plain exit of this process.
    if desync:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3)
    if dist is not None:
        dist.destroy_process_group()
    if failed:
        sys.stderr.write(f"bench.py rank {rank}: exiting 3: {failed}\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
