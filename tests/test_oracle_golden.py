"""CPU: the oracle restatement reproduces the committed reference vectors (tests/golden, made by
oracle/make_golden.py from the imported reference modules)."""
import os

import numpy as np
import pytest

from oracle import depth_oracle as O
from prisma_amd import synth


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def test_encode_vectors(golden_dir):
    z = np.load(os.path.join(golden_dir, "encode.npz"))
    assert np.array_equal((O.heat_to_rgb(z["ramp"]) * 255).astype(np.uint8), z["heat"])
    # known answers from SURVEY.md section 8 a-4
    assert z["heat"][0, 0].tolist() == [0, 25, 255] and z["heat"][-1, -1].tolist() == [255, 0, 0]
    rgb, dmin, dmax = O.encode_depth_video(z["pred"], flip=True)
    assert np.array_equal(rgb, z["vid"])
    assert dmin == float(z["pred"].min()) and dmax == float(z["pred"].max())


def test_encode_degenerate_frame():
    rgb, lo, hi = O.encode_depth_video(np.full((4, 5), 3.0, np.float32))
    assert lo == hi == 3.0 and rgb.shape == (4, 5, 3)          # 0/0 -> NaN -> uint8 0 (reference behaviour)


@pytest.mark.parametrize("wh,expect", [((1280, 720), (924, 518)), ((1920, 1080), (924, 518)),
                                        ((934, 440), (1106, 518)), ((518, 518), (518, 518)),
                                        ((128, 96), (686, 518)), ((120, 90), (686, 518))])
def test_net_size(wh, expect):
    assert O.net_size(*wh) == expect


@pytest.mark.parametrize("name,cfg", [("depth_vits_96x128", "vits"), ("depth_vitl_d4_90x120", "vitl_d4")])
def test_small_model_matches_reference(golden_dir, name, cfg):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    c = synth.DEPTH_CFGS[cfg]
    w = synth.depth_anything_weights(c, seed=1234)
    h, wd = [int(v) for v in z["frame_hw"]]
    frame = synth.frames(1, h, wd, seed=int(z["frame_seed"]))[0]
    x = O.preprocess(frame)[None]
    d_net, st = O.model_forward(w, x, c.depth, c.heads, return_stages=True)
    assert rel(d_net[0, ::4, ::4], z["net_depth_s4"]) < 2e-5
    for k in [k[3:] for k in z.files if k.startswith("st_")]:
        v = st[k]
        s = np.array([v.astype(np.float64).sum(), np.abs(v.astype(np.float64)).sum()])
        assert np.allclose(s, z["sum_" + k], rtol=1e-4, atol=1e-2), k
        g = z["st_" + k]
        if v.size > 40000:
            v = v.reshape(-1)[:: max(1, v.size // 20000)]
        assert rel(v, g) < 2e-5, k
    assert rel(O.infer(w, frame, c.depth, c.heads), z["depth"]) < 2e-5


def test_full_vitl_720p_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "depth_vitl_720p.npz"))
    c = synth.DEPTH_CFGS["vitl"]
    w = synth.depth_anything_weights(c, seed=1234)
    frame = synth.frames(1, 720, 1280, seed=int(z["frame_seed"]))[0]
    d = O.infer(w, frame, c.depth, c.heads)
    assert rel(d[::8, ::8], z["depth_s8"]) < 5e-5
    rgb, lo, hi = O.encode_depth_video(d)
    assert abs(lo - z["minmax"][0]) < 1e-4 and abs(hi - z["minmax"][1]) < 1e-4
    # heat bytes may flip by one count where fp32 round-off crosses a truncation edge
    assert (np.abs(rgb[::8, ::8].astype(int) - z["rgb_s8"].astype(int)) > 1).mean() < 1e-3


def test_write_depth_png_bytes_match_the_reference(golden_dir, tmp_path):
    """the still-image / --subpath encode.  tests/golden/write_depth.npz holds what the REAL reference
    write_depth (bands/common/io.py:138-172, encode.py:73-95,141-146) hands to cv2.imwrite for a seeded depth map (relative:
    flipped; metric: not flipped; 16-bit); the band's writer must produce PNGs with exactly those pixels - heat ramp, Sobel-edge
    saturation, min / max packed in pixels (0,0), (0,1), uint8 truncation.  (cv2.Sobel itself is the one unpinned step.)"""
    import os
    import sys
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "bands"))
    from common import io as IO
    z = np.load(os.path.join(golden_dir, "write_depth.npz"))
    depth = z["depth"]
    import depth_anything as band
    heat = band.heat_to_rgb              # the band script's own ramp (bands/depth_anything.py), not the oracle's
    for name, flip in (("rel_rgb", True), ("met_rgb", False)):
        p = str(tmp_path / (name + ".png"))
        IO.write_depth(p, depth.copy(), heat, normalize=True, flip=flip, heatmap=True, encode_range=True)
        got = np.asarray(Image.open(p))
        assert got.shape == z[name].shape and np.array_equal(got, z[name]), name
    # the range pixels decode back to min / max (viewer contract, view.py:186-210)
    rgb = z["rel_rgb"].astype(np.float64)
    dec = lambda px: (px[0] + px[1] * 256 + px[2] * 65536) / (256 ** 3 - 1) * 1000.0
    assert abs(dec(rgb[0, 0]) - depth.min()) < 1e-4 and abs(dec(rgb[0, 1]) - depth.max()) < 1e-4
    p = str(tmp_path / "u16.png")
    IO.write_depth(p, depth.copy(), heat, normalize=True, flip=False, heatmap=False)
    assert np.array_equal(np.asarray(Image.open(p)), z["u16"])
