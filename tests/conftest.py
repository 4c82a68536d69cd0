import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "local-features_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
MODELS = os.path.join(ROOT, "local-features_amd", "models", "mkd")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the product library and the oracle are build artefacts (git-ignored): make sure both exist and are current, so
    # that a fresh checkout can run the suite without a separate build step (hipcc cross-compiles without a GPU)
    import subprocess
    for target in (os.path.join(ROOT, "local-features_amd"), os.path.join(ROOT, "oracle")):
        subprocess.check_call(["make", "-s", "-j8", "-C", target])


def rel_l2(a, b):
    """Per-vector relative L2 error, the metric BASELINE.json's north_star states (gate 1e-4)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b, axis=-1) / np.linalg.norm(b, axis=-1)


@pytest.fixture(scope="session")
def oracle():
    from oracle import MkdOracle
    return MkdOracle(os.path.join(MODELS, "concat-pca-liberty.safetensors"))


@pytest.fixture(scope="session")
def oracles():
    from oracle import MkdOracle
    return {n: MkdOracle(os.path.join(MODELS, f"concat-pca-{n}.safetensors"))
            for n in ("liberty", "notredame", "yosemite")}


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


# --- parity in the presence of atan2.glsl's discontinuity ---------------------------------------------------------
# The shader's atan2 returns 0 when x == 0 (atan2.glsl:33-38), so a pixel whose gx rounds to exactly 0 flips its angle
# by 90 degrees, and whether it does depends on the last bit of the blur, which the reference leaves to its GLSL
# compiler (no `precise`; mul+add may or may not become fma).  About 2 in 1000 random patches hold such a pixel.
# The HIP kernel evaluates the blur as an fma chain in the shader's tap order, i.e. oracle mode BLUR_CONTRACT:
#   * same input bits  -> every descriptor must meet the gate against the contracted oracle, and every descriptor of a
#     patch on which the two readings agree (`settled`) must meet it against the uncontracted oracle too;
#   * different input bits (keypoint mode: the sampler rounds differently) -> descriptors of unsettled patches,
#     in either side's patch, are set aside (and must be few).
GATE = 1e-4
# Drift bar: the worst SETTLED error any parity helper has seen stays well inside the gate (round 3's worst: 3.5e-5, the 4K
# frame end to end, where a sample coordinate near 3840 has a 2.4e-4-texel ulp).  A helper call whose settled rows exceed
# this fails its own test, so a drift toward the gate shows up long before the gate does.
DRIFT = 5e-5
WORST = {"value": 0.0, "what": "-", "test": "-"}


def _note_worst(value, what):
    if value > WORST["value"]:
        WORST.update(value=float(value), what=str(what),
                     test=os.environ.get("PYTEST_CURRENT_TEST", "-").split(" ")[0])


# every parity helper call leaves one line here; printed at the end of the run (also under -q) and appended to
# gpurun_out/parity_report.txt, so that a drift in the number of patches set aside is visible from the test log
PARITY_REPORT = []


def _report(line):
    PARITY_REPORT.append(line)
    print(line)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_report.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def pytest_terminal_summary(terminalreporter):
    if PARITY_REPORT:
        terminalreporter.write_sep("-", "parity helpers: patches set aside, and why")
        for line in PARITY_REPORT:
            terminalreporter.write_line(line)
        line = (f"[parity margin] worst settled relative L2 of this run: {WORST['value']:.2e} ({WORST['what']}; {WORST['test']}); "
                f"drift bar {DRIFT:.0e}, gate {GATE:.0e}")
        terminalreporter.write_line(line)
        try:
            with open(os.path.join(ROOT, "gpurun_out", "parity_report.txt"), "a") as f:
                f.write(line + "\n")
        except OSError:
            pass


# --- the forms of the keypoint kernel ---------------------------------------------------------------------------------
# Requests of at most 4096 keypoints take the row-split form (round 6: 2 or 4 workgroups per batch of 32 keypoints, each
# pooling its share of the 32 patch rows, the partial sums added in a fixed order), larger ones the whole-patch form (one
# chain of 32 row sums).  Within a form a descriptor's bits depend on its keypoint and frame alone; between forms the sums
# round differently: ~3e-6 relative L2 after whitening (measured worst 4.3e-6 over tools/check_split.py's sizes, 6.5e-6 over
# tools/soak_split.py's 32 000 launches).
# LF_MKD_KP_SPLIT=1 / 2 / 4 in the environment (read per launch) forces a form where it fits.
CROSS_FORM = 1e-5


class kp_form:
    """with kp_form("1"): ...   -- the launches inside take the whole-patch (1) or a row-split (2, 4) form"""

    def __init__(self, form):
        self.form, self.old = str(form), None

    def __enter__(self):
        self.old = os.environ.get("LF_MKD_KP_SPLIT")
        os.environ["LF_MKD_KP_SPLIT"] = self.form

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("LF_MKD_KP_SPLIT", None)
        else:
            os.environ["LF_MKD_KP_SPLIT"] = self.old


def assert_same_descriptors(a, b, what=""):
    """descriptors of the same keypoints from requests of different sizes, i.e. possibly from different forms of the kernel"""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    e = rel_l2(a, b).max(initial=0.0)
    _report(f"[cross-form] {what or '-'}: {len(a)} descriptors, worst relative L2 between the two requests {e:.2e}"
            + (" (same bits)" if np.array_equal(a, b) else ""))
    assert e < CROSS_FORM, (what, e)


def settled_detail(oracle, patches, atan_mode, gate=GATE):
    """(settled mask, contracted descriptors, uncontracted descriptors, mask of patches with a pixel on the shader's
    gx == 0 discontinuity, mask of patches on which the reference's two readings of the blur differ by more than a quarter
    of the gate).  Unsettled are: patches with a pixel on the discontinuity, and -- a second, milder way in which the
    blur's last bit reaches the descriptor -- patches in which a pixel whose gradient is null up to rounding (a critical
    point of the blurred patch: its angle is rounding noise, its magnitude still the floor 0.01) carries visible weight,
    i.e. low-contrast patches; told by the two readings themselves differing."""
    from oracle import ATAN_SHADER, BLUR_CONTRACT
    ref_c = oracle.describe_patches(patches, atan_mode=atan_mode | BLUR_CONTRACT, nthreads=8)
    ref_s = oracle.describe_patches(patches, atan_mode=atan_mode, nthreads=8)
    quirk = oracle.quirk_pixels(patches) != 0 if atan_mode == ATAN_SHADER else np.zeros(len(patches), bool)
    reading = rel_l2(ref_c, ref_s) >= gate / 4
    return ~quirk & ~reading, ref_c, ref_s, quirk, reading


def settled(oracle, patches, atan_mode, gate=GATE):
    """(mask of patches on which the reference's two readings of the blur give the same descriptor, the contracted and
    the uncontracted descriptors); see settled_detail."""
    ok, ref_c, ref_s, _, _ = settled_detail(oracle, patches, atan_mode, gate)
    return ok, ref_c, ref_s


def _aside(quirk, reading):
    n = len(quirk)
    return (f"{int((quirk | reading).sum())}/{n} set aside (gx==0 pixel: {int(quirk.sum())}, "
            f"blur readings differ: {int((reading & ~quirk).sum())})")


def assert_patch_parity(oracle, patches, desc, atan_mode, gate=GATE, what="", min_settled=0.97):
    patches = np.asarray(patches, np.float32).reshape(-1, 32, 32)
    clean, ref_c, ref_s, quirk, reading = settled_detail(oracle, patches, atan_mode, gate)
    e_c = rel_l2(desc, ref_c)
    e_s = rel_l2(desc, ref_s)
    _report(f"[patch parity] {what or '-'}: {_aside(quirk, reading)}; worst vs contracted reading (all {len(patches)}) "
            f"{e_c.max(initial=0.0):.2e}, vs uncontracted (settled) {e_s[clean].max(initial=0.0):.2e}")
    assert e_c.max(initial=0.0) < gate, (what, "contracted", int(e_c.argmax()), e_c.max())
    assert clean.mean() > min_settled, (what, clean.mean())
    assert e_s[clean].max(initial=0.0) < gate, (what, "uncontracted", e_s[clean].max())
    worst = max(e_c.max(initial=0.0), e_s[clean].max(initial=0.0))
    _note_worst(worst, what or "patch parity")
    assert worst < min(gate, DRIFT), (what, "drift bar", worst)
    return worst


def assert_keypoint_parity(oracle, handle, img, kps5, desc, gate=GATE, what="", patch_tol=1e-5, min_settled=0.97,
                           patch_scale_factor=24.0):
    """desc = the library's descriptors of keypoints kps5 [n,5] on img (already set on `handle`, which was created with
    this patch_scale_factor).
    patch_tol: the sampling positions of the two sides differ by ~1e-5 texel (different libm sin/cos/exp2); on smooth
    test frames that is 1e-6 in the sampled values, on a sharp photograph up to the local gradient times that."""
    import torch
    from oracle import ATAN_SHADER
    what = str(what)
    img = np.ascontiguousarray(img, np.float32)
    kps5 = np.ascontiguousarray(kps5, np.float32)
    hgt, w = img.shape
    n = len(kps5)
    d_k = torch.from_numpy(kps5).cuda()
    d_p = torch.empty((n, 32, 32), device="cuda")
    torch.cuda.synchronize()
    handle.sample_patches_device(d_k.data_ptr(), n, d_p.data_ptr())
    handle.synchronize()
    got_p = d_p.cpu().numpy()
    ref_p = oracle.sample_patches(oracle.build_pyramid(img), w, hgt, kps5[:, :4], patch_scale_factor)
    assert np.abs(got_p - ref_p).max(initial=0.0) < patch_tol, (what, "sampled patches", np.abs(got_p - ref_p).max())
    assert_patch_parity(oracle, got_p, desc, ATAN_SHADER, gate, what + " (describe stage, GPU-sampled bits)", min_settled)
    clean_ref, _, ref_s, q_ref, r_ref = settled_detail(oracle, ref_p, ATAN_SHADER, gate)
    clean_got, _, _, q_got, r_got = settled_detail(oracle, got_p, ATAN_SHADER, gate)
    clean = clean_got & clean_ref
    e = rel_l2(desc, ref_s)                                                  # end to end
    _report(f"[keypoint parity] {what or '-'}: end to end {_aside(q_ref | q_got, r_ref | r_got)} (either side's patch); "
            f"worst settled {e[clean].max(initial=0.0):.2e}")
    assert clean.mean() > min_settled, (what, clean.mean())
    assert e[clean].max(initial=0.0) < gate, (what, "end to end", e[clean].max())
    _note_worst(e[clean].max(initial=0.0), (what or "keypoint parity") + " (end to end)")
    assert e[clean].max(initial=0.0) < min(gate, DRIFT), (what, "drift bar, end to end", e[clean].max())
