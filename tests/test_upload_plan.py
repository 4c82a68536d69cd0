"""The banded upload's plan (lf_mkd_plan_upload: host only, no device): the pieces lf_mkd_detect / lf_mkd_detect_u8 would upload a
frame in, chosen from a model of the link and of the pipeline's front (csrc/lf_mkd.cpp plan_cuts).  What the plan must always
satisfy, and the shapes it takes for the reference benchmark's frame (benches/bench.rs: houses.jpg, 4096 x 3072)."""
import ctypes

import pytest

import local_features_python as lfp


def _check(cuts, height):
    assert cuts == sorted(set(cuts)) and all(c % 4 == 0 and 16 <= c <= height - 16 for c in cuts), cuts


@pytest.mark.parametrize("n_scales", [3, 4, 5, 6])
def test_the_reference_benchmarks_frame(n_scales, monkeypatch):
    for k in ("LF_MKD_DETECT_BANDS", "LF_MKD_BAND_SPLIT", "LF_MKD_BAND_PIECES"):
        monkeypatch.delenv(k, raising=False)
    w, h = 4096, 3072
    f32, t_f32, one_f32 = lfp.plan_upload(w, h, 4, n_scales)
    u8, t_u8, one_u8 = lfp.plan_upload(w, h, 1, n_scales)
    for cuts in (f32, u8):
        _check(cuts, h)
        assert 1 <= len(cuts) <= 5
    # a plan is only taken when the model puts it 4 % ahead of one piece
    assert t_f32 < 0.96 * one_f32 and t_u8 < 0.96 * one_u8
    # an f32 frame is upload-bound (50 MB over the link against a front of a few hundred us): a large first piece, little
    # left to do when the last byte lands; an 8-bit frame's front takes longer than its upload: it must start early
    sizes = lambda c: [b - a for a, b in zip([0] + c, c + [h])]
    assert sizes(f32)[0] > sizes(f32)[-1] and sizes(f32)[0] >= h // 3
    assert sizes(u8)[0] <= h // 2 and u8[0] < f32[0]
    # the upload alone bounds any plan from below (56 GB/s)
    assert t_f32 > w * h * 4 / 56e3 and t_u8 > w * h / 56e3 / 2


def test_small_and_unaligned_frames_travel_in_one_piece(monkeypatch):
    for k in ("LF_MKD_DETECT_BANDS", "LF_MKD_BAND_SPLIT", "LF_MKD_BAND_PIECES"):
        monkeypatch.delenv(k, raising=False)
    for w, h, bpp in ((640, 480, 4), (1920, 1080, 1), (1024, 768, 4), (1920, 1080, 4)):       # below 6 MB of upload
        cuts, t, one = lfp.plan_upload(w, h, bpp, 4)
        assert cuts == [] and t == one
    for w, h in ((3838, 2160), (3840, 2161)):          # the row-tiled kernels need widths of 4 and even heights
        assert lfp.plan_upload(w, h, 4, 4)[0] == []
    assert lfp.plan_upload(3840, 2160, 4, 4)[0] != []


def test_environment_overrides(monkeypatch):
    monkeypatch.setenv("LF_MKD_DETECT_BANDS", "0")
    assert lfp.plan_upload(4096, 3072, 4, 3)[0] == []
    monkeypatch.delenv("LF_MKD_DETECT_BANDS")
    monkeypatch.setenv("LF_MKD_BAND_PIECES", "4")
    assert lfp.plan_upload(640, 480, 1, 4)[0] == [120, 240, 360]                  # whatever the frame's size
    assert lfp.plan_upload(4096, 3072, 4, 5)[0] == [768, 1536, 2304]
    monkeypatch.setenv("LF_MKD_BAND_PIECES", "1")
    assert lfp.plan_upload(4096, 3072, 4, 5)[0] == []
    monkeypatch.delenv("LF_MKD_BAND_PIECES")
    monkeypatch.setenv("LF_MKD_BAND_SPLIT", "0.9,0.1,0.5")                        # any order; rounded to multiples of four rows
    cuts = lfp.plan_upload(640, 480, 1, 4)[0]
    assert cuts == [48, 240, 432]
    monkeypatch.setenv("LF_MKD_BAND_SPLIT", "0.0,1.0")                            # clamped 16 rows inside either end
    cuts = lfp.plan_upload(640, 480, 4, 4)[0]
    _check(cuts, 480)
    assert cuts == [16, 464]
    monkeypatch.setenv("LF_MKD_BAND_SPLIT", "0.5,0.5,0.501")                      # duplicates collapse
    assert lfp.plan_upload(640, 480, 4, 4)[0] == [240]


def test_bad_arguments():
    L = lfp.load_library()
    n = ctypes.c_uint32()
    cuts = (ctypes.c_uint32 * 4)()
    assert L.lf_mkd_plan_upload(4096, 3072, 4, 3, cuts, 4, None, None, None) == -1        # n_cuts is required
    assert L.lf_mkd_plan_upload(4096, 3072, 2, 3, cuts, 4, ctypes.byref(n), None, None) == -1   # 1 or 4 bytes per pixel
    assert L.lf_mkd_plan_upload(4096, 3072, 4, 9, cuts, 4, ctypes.byref(n), None, None) == -1   # n_scales <= 6
    assert L.lf_mkd_plan_upload(4096, 3072, 4, 3, None, 4, ctypes.byref(n), None, None) == -1   # room promised, none given
    assert L.lf_mkd_plan_upload(4096, 3072, 4, 5, None, 0, ctypes.byref(n), None, None) == 0 and n.value == 3   # count only
    assert L.lf_mkd_plan_upload(4096, 3072, 4, 5, cuts, 2, ctypes.byref(n), None, None) == 0 and n.value == 3 and cuts[1] > cuts[0] > 0
