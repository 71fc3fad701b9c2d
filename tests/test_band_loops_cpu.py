"""CPU: the band scripts' video loops (chunking, prefetch / async writes, output files) driven by fake engines.
The numeric path needs the GPU (tests/test_band_cli.py, -m gpu); here the model object is a stand-in whose output encodes
the frame index, so ordering or chunk-boundary mistakes in the host logic show up without a device."""
import json
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bands"))


def _clip(tmp_path, n, h=12, w=16):
    frames = np.zeros((n, h, w, 3), np.uint8)
    frames[:, 0, 0, 0] = np.arange(n)                 # the frame index rides in one pixel
    folder = tmp_path / "clip"
    folder.mkdir()
    np.save(folder / "rgba.npy", frames)
    (folder / "metadata.json").write_text(json.dumps({"bands": {"rgba": {"url": "rgba.npy"}}}))
    return folder, frames


class _Ranks:
    rank, world, main, device = 0, 1, True, 0

    def frames(self, n, halo=0):
        return 0, n

    def close(self):
        pass


@pytest.mark.parametrize("n,batch", [(7, 3), (4, 4), (1, 32)])
def test_depth_loop(tmp_path, monkeypatch, n, batch):
    import depth_anything as band
    folder, frames = _clip(tmp_path, n)

    class Fake:
        calls = []

        def infer_batch(self, fr, want_depth=True, want_rgb=True, flip=True):
            idx = fr[:, 0, 0, 0].astype(np.float32)
            Fake.calls.append(len(fr))
            ramp = np.linspace(0.0, 0.5, fr.shape[2], dtype=np.float32)[None, None, :]        # non-constant, or write_depth divides 0 / 0
            depth = (idx[:, None, None] + np.broadcast_to(ramp, fr.shape[:3])).astype(np.float32) if want_depth else None
            return depth, fr.copy(), idx, idx + 1.0

    class FakeStill:                # the --subpath PNG encode runs on the GPU (pb_depth_encode_still); here only the file plumbing is under test
        def encode_still(self, depth, flip=True, encode_range=True):
            return np.zeros(depth.shape + (3,), np.uint8), float(depth.min()), float(depth.max())

    monkeypatch.setattr(band, "BATCH", batch)
    monkeypatch.setattr(band, "model", Fake())
    monkeypatch.setattr(band, "_still", FakeStill())
    monkeypatch.setattr(band, "ranks", _Ranks())
    monkeypatch.setattr(band, "data", json.load(open(folder / "metadata.json")))
    band.data["bands"][band.BAND] = {"url": "depth_anything.npy"}
    a = types.SimpleNamespace(input=str(folder / "rgba.npy"), output=str(folder / "depth_anything.npy"), npy=True, subpath="frames",
                              metric="none")
    monkeypatch.setattr(band, "args", a)
    band.process_video(a)
    out = np.load(folder / "depth_anything.npy")
    assert np.array_equal(out, frames)                                   # every frame, in order
    assert Fake.calls == [min(batch, n - s) for s in range(0, n, batch)]
    lo = [float(x) for x in open(folder / "depth_anything_min.csv")]
    hi = [float(x) for x in open(folder / "depth_anything_max.csv")]
    assert lo == list(map(float, range(n))) and hi == [v + 1.0 for v in lo]
    for i in range(n):                                                   # per-frame dumps written by the sink thread
        assert float(np.load(folder / "frames" / "{:05d}.npy".format(i))[0, 0]) == float(i)
        assert os.path.exists(folder / "frames" / "{:05d}.png".format(i))


def test_mask_loop(tmp_path, monkeypatch):
    import mask_mmdet as band
    folder, frames = _clip(tmp_path, 5)

    class Fake:
        sdf = []

        def set_sdf(self, on):           # --sdf is the engine's job now (green channel written on the GPU); the loop only switches it
            Fake.sdf.append(bool(on))

        def infer_batch(self, fr, confidence, keep):
            return fr.copy()

    monkeypatch.setattr(band, "BATCH", 2)
    monkeypatch.setattr(band, "model", Fake())
    monkeypatch.setattr(band, "ranks", _Ranks())
    monkeypatch.setattr(band, "data", {"bands": {"rgba": {"url": "rgba.npy"}}})
    a = types.SimpleNamespace(input=str(folder / "rgba.npy"), output=str(folder / "mask.npy"), confidence=0.5, sdf=False, subpath="")
    band.process_video(a)
    assert np.array_equal(np.load(folder / "mask.npy")[:, 0, 0, 0], np.arange(5)) and Fake.sdf == [False]
    a.sdf, a.subpath = True, "colmap"
    os.environ["PRISMA_OVERWRITE"] = "1"
    band.process_video(a)
    assert Fake.sdf == [False, True]
    from PIL import Image
    png = np.asarray(Image.open(folder / "colmap" / "00003.png"))        # COLMAP frames: 255 - the pre-SDF image (its red channel three times)
    assert np.array_equal(png, 255 - np.repeat(frames[3][..., :1], 3, axis=-1))


def test_flow_loop_pairs_and_halo(tmp_path, monkeypatch):
    import flow_raft as band
    folder, frames = _clip(tmp_path, 6)
    seen = []

    class Fake:
        def infer_sequence(self, fr, scale=1.0, iters=12, backward=False, want_flow=False, want_rgb=True):
            idx = fr[:, 0, 0, 0]
            seen.append(list(map(int, idx)))                              # chunk = its pairs + one halo frame
            n = len(fr) - 1
            rgb = np.zeros((n, 2 if backward else 1) + fr.shape[1:], np.uint8)
            rgb[:, 0, 0, 0, 0] = idx[:-1]
            mx = np.zeros((n, 2 if backward else 1), np.float32)
            mx[:, 0] = idx[:-1] + 0.5
            return None, rgb, mx

    monkeypatch.setattr(band, "CHUNK", 2)
    monkeypatch.setattr(band, "model", Fake())
    monkeypatch.setattr(band, "ranks", _Ranks())
    monkeypatch.setattr(band, "data", {"bands": {"rgba": {"url": "rgba.npy"}, band.BAND: {"url": "flow_raft.npy"}}})
    monkeypatch.setattr(band.engine, "flow_out_size", lambda h, w, s: (h, w))
    a = types.SimpleNamespace(input=str(folder / "rgba.npy"), output=str(folder / "flow_raft.npy"), scale=1.0, iterations=12,
                              backwards=False, mask=False, output_mask="", subpath="", subpath_mask="")
    band.process_video(a)
    assert seen == [[0, 1, 2], [2, 3, 4], [4, 5]]
    out = np.load(folder / "flow_raft.npy")
    assert out.shape[0] == 6 and list(out[:5, 0, 0, 0]) == [0, 1, 2, 3, 4] and not out[5].any()   # last frame: zero flow
    disp = [float(x) for x in open(folder / "flow_raft.csv")]
    assert disp == [0.5, 1.5, 2.5, 3.5, 4.5, 0.0]
