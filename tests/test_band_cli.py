"""Band script: folder/metadata contract on CPU, full CLI run on the GPU."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bands"))

from common import meta  # noqa: E402
from common.io import float_to_rgb  # noqa: E402


def test_metadata_contract(tmp_path):
    folder = str(tmp_path / "clip")
    data = meta.create_metadata(folder)
    assert data == {"bands": {}}
    # reference behaviour (bands/common/meta.py:70-93): target name = <band>.<ext of the rgba url>
    meta.add_band(data, "rgba", url="rgba.mp4")
    rgba = os.path.join(folder, "rgba.mp4")
    assert meta.get_url(folder, data, "rgba") == rgba
    assert meta.get_target(rgba, data, band="depth_anything", force_extension="png") == os.path.join(folder, "depth_anything.mp4")
    assert data["bands"]["depth_anything"]["url"] == "depth_anything.mp4"
    assert meta.get_target(os.path.join(folder, "rgba.png"), data, band="mask", force_extension="png").endswith("mask.png")
    assert meta.get_target(rgba, data, band="flow_raft", force_extension="csv").endswith("flow_raft.csv")
    meta.write_metadata(folder, data)
    assert json.load(open(os.path.join(folder, "metadata.json")))["bands"]["rgba"]["url"] == "rgba.mp4"
    meta.set_default_band(folder, "depth", "depth_anything")
    assert meta.load_metadata(folder)["bands"]["depth"]["url"] == "depth_anything.mp4"
    assert meta.is_video("x.mp4") and not meta.is_video("x.png")


def test_float_to_rgb_known_answer(golden_dir):
    z = np.load(os.path.join(golden_dir, "encode.npz"))
    assert np.allclose(float_to_rgb(12.5, 0.0, 1000.0), z["float_to_rgb"], rtol=0, atol=0)


@pytest.mark.gpu
def test_cli_video_and_image(tmp_path):
    import depth_anything as band
    from prisma_amd import synth
    folder = tmp_path / "clip"
    folder.mkdir()
    frames = synth.frames(5, 90, 160, seed=3)
    np.save(folder / "rgba.npy", frames)
    (folder / "metadata.json").write_text(json.dumps({"bands": {"rgba": {"url": "rgba.npy"}}}))
    os.environ["PRISMA_OVERWRITE"] = "1"
    band.model = None
    band.main(["-i", str(folder), "--encoder", "vits"])
    out = np.load(folder / "depth_anything.npy")
    assert out.shape == frames.shape and out.dtype == np.uint8
    lo = [float(x) for x in open(folder / "depth_anything_min.csv")]
    hi = [float(x) for x in open(folder / "depth_anything_max.csv")]
    assert len(lo) == len(hi) == 5 and all(b > a for a, b in zip(lo, hi))
    md = json.load(open(folder / "metadata.json"))
    assert md["bands"]["depth_anything"]["url"] == "depth_anything.npy"
    assert md["bands"]["depth_anything"]["values"]["min"] == {"type": "float", "url": "depth_anything_min.csv"}
    # per-frame API agrees with the batched video path
    d = band.infer(frames[2])
    assert abs(d.min() - lo[2]) < 1e-6 and abs(d.max() - hi[2]) < 1e-6
    # still image
    from PIL import Image
    Image.fromarray(frames[0]).save(tmp_path / "img.png")
    band.main(["-i", str(tmp_path / "img.png"), "--encoder", "vits"])
    png = np.asarray(Image.open(tmp_path / "depth_anything.png"))
    assert png.shape == (90, 160, 3)
    # the PNG the band wrote on the GPU box = the reference's write_depth arithmetic (common/io.py:138-172
    # restated in bands/common/io.py, byte-pinned to the real one in tests/test_oracle_golden.py) applied to the engine's own float
    # depth through the BAND's heat ramp, byte for byte; its range pixels decode back to the depth's min / max (view.py:186-210)
    from common import io as IO
    d0 = band.infer(frames[0])
    IO.write_depth(str(tmp_path / "host.png"), d0.copy(), band.heat_to_rgb, normalize=True, flip=True, heatmap=True, encode_range=True)
    assert np.array_equal(png, np.asarray(Image.open(tmp_path / "host.png")))
    dec = lambda px: (float(px[0]) + float(px[1]) * 256 + float(px[2]) * 65536) / (256 ** 3 - 1) * 1000.0
    assert abs(dec(png[0, 0]) - d0.min()) < 1e-4 and abs(dec(png[0, 1]) - d0.max()) < 1e-4
    band.model.close()
    band.model = None


@pytest.mark.gpu
def test_flow_cli_video(tmp_path):
    import flow_raft as band
    from prisma_amd import synth
    folder = tmp_path / "clip"
    folder.mkdir()
    frames = synth.frame_pair_sequence(4, 176, 256, seed=6)
    np.save(folder / "rgba.npy", frames)
    (folder / "metadata.json").write_text(json.dumps({"bands": {"rgba": {"url": "rgba.npy"}}}))
    os.environ["PRISMA_OVERWRITE"] = "1"
    band.model = None
    band.main(["-i", str(folder), "--iterations", "4", "--scale", "1.0", "-b"])
    out = np.load(folder / "flow_raft.npy")
    assert out.shape == (4, 176, 256, 3) and out.dtype == np.uint8 and not out[-1].any()
    assert np.load(folder / "flow_raft_bwd.npy").shape == out.shape
    dist = [float(x) for x in open(folder / "flow_raft.csv")]
    assert len(dist) == 4 and dist[-1] == 0.0 and all(d > 0 for d in dist[:-1])
    md = json.load(open(folder / "metadata.json"))
    assert md["bands"]["flow_raft"]["values"]["dist"] == {"type": "float", "url": "flow_raft.csv"}
    assert md["bands"]["flow_raft_bwd"]["url"] == "flow_raft_bwd.npy"
    band.model.close()
    band.model = None


@pytest.mark.gpu
def test_flow_gmflow_cli_video(tmp_path):
    """bands/flow_gmflow.py (prisma's default flow band): same folder contract as flow_raft under its own names, the reference's
    architecture flags accepted at their defaults and rejected otherwise."""
    import flow_gmflow as band
    from prisma_amd import synth
    folder = tmp_path / "clip"
    folder.mkdir()
    frames = synth.frame_pair_sequence(4, 176, 256, seed=6)
    np.save(folder / "rgba.npy", frames)
    (folder / "metadata.json").write_text(json.dumps({"bands": {"rgba": {"url": "rgba.npy"}}}))
    os.environ["PRISMA_OVERWRITE"] = "1"
    band.model = None
    band.main(["-i", str(folder), "--scale", "1.0", "-b", "--mask", "--num_transformer_layers", "6", "--attn_splits_list", "2"])
    out = np.load(folder / "flow_gmflow.npy")
    assert out.shape == (4, 176, 256, 3) and out.dtype == np.uint8 and not out[-1].any() and out[0].any()
    assert np.load(folder / "flow_gmflow_bwd.npy").shape == out.shape
    assert np.load(folder / "flow_gmflow_mask.npy").shape == out.shape
    dist = [float(x) for x in open(folder / "flow_gmflow.csv")]
    assert len(dist) == 4 and dist[-1] == 0.0 and all(d > 0 for d in dist[:-1])
    md = json.load(open(folder / "metadata.json"))
    assert md["bands"]["flow_gmflow"]["values"]["dist"] == {"type": "float", "url": "flow_gmflow.csv"}
    assert md["bands"]["flow_gmflow_bwd"]["url"] == "flow_gmflow_bwd.npy" and md["bands"]["flow_gmflow_mask"]["url"] == "flow_gmflow_mask.npy"
    # module API of the reference (:66-118): infer(args, prev, curr) on CHW float frames
    import types
    a = types.SimpleNamespace(backwards=True, output_mask="", subpath_mask="")
    fwd, bwd, fm, bm = band.infer(a, frames[0].transpose(2, 0, 1).astype(np.float32), frames[1].transpose(2, 0, 1).astype(np.float32))
    assert fwd.shape == (176, 256, 2) and bwd.shape == (176, 256, 2) and fm is None and bm is None
    assert abs(float(np.sqrt((fwd ** 2).sum(-1)).max()) - dist[0]) < 1e-4
    with pytest.raises(SystemExit, match="only the band's default GMFlow"):
        band.main(["-i", str(folder), "--num_scales", "2"])
    band.model.close()
    band.model = None
    # --inference_size H W (reference :76-100): another network size, same output size; bad sizes are refused before the model loads
    first = out.copy()
    band.main(["-i", str(folder), "--scale", "1.0", "--inference_size", "128", "192"])
    out = np.load(folder / "flow_gmflow.npy")
    assert out.shape == first.shape and not np.array_equal(out, first) and out[0].any() and not out[-1].any()
    with pytest.raises(SystemExit, match="multiples of 16"):
        band.main(["-i", str(folder), "--inference_size", "100", "192"])
    band.model.close()
    band.model = None


def test_flow_file_writers(tmp_path):
    """.flo layout (bands/common/io.py:175-197) and the 16-bit flow+mask PNG (encode.py:105-110 through cv2.imwrite)."""
    from common.io import encode_flow, write_flo, write_flow_png
    from PIL import Image
    rng = np.random.default_rng(5)
    flow = (rng.standard_normal((6, 9, 2)) * 40).astype(np.float32)
    flow[0, 0] = (200.0, 0.0)          # 2^15 + 200 * 256 overflows uint16 -> invalid
    mask = rng.random((6, 9)) > 0.3
    write_flo(str(tmp_path / "a.flo"), flow)
    raw = open(tmp_path / "a.flo", "rb").read()
    assert np.frombuffer(raw[:4], np.float32)[0] == np.float32(202021.25)
    assert tuple(np.frombuffer(raw[4:12], np.int32)) == (9, 6)
    assert np.array_equal(np.frombuffer(raw[12:], np.float32).reshape(6, 9, 2), flow)
    enc = encode_flow(flow, mask)
    assert enc.dtype == np.uint16 and enc.shape == (6, 9, 3) and enc[0, 0, 2] == 0
    assert enc[1, 1, 0] == np.uint16(np.float32(32768) + flow[1, 1, 0] * np.float32(256))
    assert np.array_equal(enc[..., 2] == 65535, mask & (enc[..., 2] == 65535)) and enc[..., 2][mask][1:].all()
    write_flow_png(str(tmp_path / "a.png"), flow, mask)
    png = np.asarray(Image.open(tmp_path / "a.png"))         # PIL reads the high bytes
    assert np.array_equal(png, (enc[..., ::-1] >> 8).astype(np.uint8))


@pytest.mark.gpu
def test_flow_cli_masks_and_dumps(tmp_path):
    import flow_raft as band
    from prisma_amd import synth
    folder = tmp_path / "clip"
    folder.mkdir()
    frames = synth.frame_pair_sequence(3, 176, 256, seed=6)
    np.save(folder / "rgba.npy", frames)
    (folder / "metadata.json").write_text(json.dumps({"bands": {"rgba": {"url": "rgba.npy"}}}))
    os.environ["PRISMA_OVERWRITE"] = "1"
    band.model = None
    band.main(["-i", str(folder), "--iterations", "3", "--scale", "1.0", "-b", "--mask", "--subpath", "flo", "--subpath_mask", "fm"])
    m = np.load(folder / "flow_raft_mask.npy")
    assert m.shape == (3, 176, 256, 3) and set(np.unique(m)) <= {0, 255} and not m[-1].any()
    assert np.load(folder / "flow_raft_mask_bwd.npy").shape == m.shape
    assert sorted(os.listdir(folder / "flo_fwd")) == ["0000.flo", "0001.flo", "0002.flo"]
    assert sorted(os.listdir(folder / "fm_bwd")) == ["0000.png", "0001.png", "0002.png"]
    md = json.load(open(folder / "metadata.json"))
    assert md["bands"]["flow_raft_mask"]["url"] == "flow_raft_mask.npy"
    assert md["bands"]["flow_raft_mask_bwd"]["url"] == "flow_raft_mask_bwd.npy"
    assert md["bands"]["flow_raft"]["folder"].endswith("flo")
    # module API: infer() hands back the masks when a mask output is requested (reference :63-64)
    import argparse
    a = argparse.Namespace(iterations=3, output_mask="x", subpath_mask="", subpath="", backwards=False)
    i1 = np.stack([frames[0], frames[1]]).transpose(0, 3, 1, 2).astype(np.float32)
    i2 = i1[::-1].copy()
    fwd, bwd, mf, mb = band.infer(a, i1, i2)
    assert fwd.shape == (176, 256, 2) and mf.dtype == np.bool_ and mf.shape == (176, 256)
    band.model.close()
    band.model = None


@pytest.mark.gpu
def test_mask_cli_video_and_image(tmp_path):
    import mask_mmdet as band
    from prisma_amd import synth
    from PIL import Image
    folder = tmp_path / "clip"
    folder.mkdir()
    frames = synth.frames(3, 180, 300, seed=5)
    np.save(folder / "rgba.npy", frames)
    (folder / "metadata.json").write_text(json.dumps({"bands": {"rgba": {"url": "rgba.npy"}}}))
    os.environ["PRISMA_OVERWRITE"] = "1"
    band.model = None
    band.main(["-i", str(folder), "--arch", "tiny", "--subpath", "mask_frames", "--sdf"])
    out = np.load(folder / "mask.npy")
    assert out.shape == frames.shape and out.dtype == np.uint8
    assert np.array_equal(out[..., 0], out[..., 2]) and out[..., 0].any()       # R == B = accumulated masks, G = SDF
    assert not np.array_equal(out[..., 0], out[..., 1])
    colmap = np.asarray(Image.open(folder / "mask_frames" / "00001.png"))
    assert np.array_equal(colmap[..., 0], 255 - out[1, ..., 0])
    md = json.load(open(folder / "metadata.json"))
    assert md["bands"]["mask"] == {"url": "mask.npy", "ids": band.CLASSES}
    # still image, no metadata folder
    Image.fromarray(frames[0]).save(tmp_path / "img.png")
    band.main(["-i", str(tmp_path / "img.png"), "--arch", "tiny"])
    png = np.asarray(Image.open(tmp_path / "mask.png"))
    # batch of 3 vs batch of 1: the mask ids of a frame do not depend on the frames it shares a launch with
    assert png.shape == (180, 300, 3) and np.array_equal(png[..., 0], out[0, ..., 0])
    band.model.close()
    band.model = None


@pytest.mark.gpu
def test_cli_metric_depth(tmp_path):
    """`--metric indoor`: ZoeDepth head, not flipped (reference bands/depth_anything.py:52-57,106-119,150,188)."""
    import depth_anything as band
    from prisma_amd import synth
    folder = tmp_path / "clip"
    folder.mkdir()
    frames = synth.frames(2, 120, 200, seed=3)
    np.save(folder / "rgba.npy", frames)
    (folder / "metadata.json").write_text(json.dumps({"bands": {"rgba": {"url": "rgba.npy"}}}))
    os.environ["PRISMA_OVERWRITE"] = "1"
    band.model = None
    band.main(["-i", str(folder), "--metric", "indoor"])
    out = np.load(folder / "depth_anything.npy")
    assert out.shape == frames.shape
    lo = [float(x) for x in open(folder / "depth_anything_min.csv")]
    hi = [float(x) for x in open(folder / "depth_anything_max.csv")]
    assert all(0.1 < a < b < 20.0 for a, b in zip(lo, hi))            # metres, not the relative model's arbitrary scale
    d = band.infer(frames[1])
    assert d.shape == (120, 200) and abs(d.min() - lo[1]) < 1e-6 and abs(d.max() - hi[1]) < 1e-6
    far = np.unravel_index(d.argmax(), d.shape)
    assert tuple(out[1][far]) == (255, 0, 0)                            # farthest = heat 1 = red: not flipped
    band.model.close()
    band.model = None
    band.args = None
