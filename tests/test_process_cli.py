"""process.py (the L5 orchestrator the north_star keeps): BASELINE configs[0] "plumbing only" - folder / metadata.json contract,
band order and per-band arguments of /root/reference/process.py:60-73,76-99,172-290 - on a synthetic 934x440 PNG (data/gog.jpg
and MiDaS need the network).  CPU: the GPU bands are replaced by a recorder, rgba runs for real.  GPU: everything runs."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import process  # noqa: E402


def _png(tmp_path, w=934, h=440):
    from PIL import Image
    rng = np.random.default_rng(0)
    p = tmp_path / "gog.png"
    Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(p)
    return p


def _fake_bands(monkeypatch):
    real = subprocess.run

    def fake(cmd, **kw):
        if os.path.basename(cmd[1]) == "rgba.py":
            return real(cmd, **kw)
        return subprocess.CompletedProcess(cmd, 0)
    monkeypatch.setattr(process.subprocess, "run", fake)


def test_image_plumbing_metadata_and_commands(tmp_path, monkeypatch):
    _fake_bands(monkeypatch)
    src = _png(tmp_path)
    folder = process.main(["-i", str(src), "-d", "depth_anything"])
    assert folder == str(tmp_path / "gog") and os.path.exists(os.path.join(folder, "rgba.png"))
    md = json.load(open(os.path.join(folder, "metadata.json")))
    assert md["width"] == 934 and md["height"] == 440
    assert md["principal_point"] == [467.0, 220.0]
    assert abs(md["focal_length"] - (934 * 440) ** 0.5) < 1e-9 and abs(md["focal_length"] - 641.06) < 0.01       # process.py:186-187
    assert abs(md["field_of_view"] - 2 * np.degrees(np.arctan(0.5 * 440 / md["focal_length"]))) < 1e-9
    assert abs(md["field_of_view"] - 37.88) < 0.01                                                              # process.py:188-189
    assert md["bands"]["rgba"]["url"] == "rgba.png" and "fps" not in md
    names = [os.path.basename(c[1]) for c in process.COMMANDS]
    assert names == ["rgba.py", "mask_mmdet.py", "depth_anything.py"]                 # image: no flow, no camera (process.py:205-245)
    rgba, mask, depth = process.COMMANDS
    assert rgba[2:] == ["-i", str(src), "--output", os.path.join(folder, "rgba.png"), "--subpath", "images"]
    assert mask[2:] == ["-i", folder, "--sdf", "--subpath", "mask"]                  # EXTRA_ARGS + subpath=True
    assert depth[2:] == ["-i", folder, "--metric", "outdoor"]                        # process.py:53; --extra 0: no --subpath


def test_extra_levels_and_video_band_order(tmp_path, monkeypatch):
    _fake_bands(monkeypatch)
    clip = tmp_path / "clip.npy"
    np.save(clip, np.zeros((3, 48, 64, 3), np.uint8))
    folder = process.main(["-i", str(clip), "-e", "2", "-b", "-m", "--fps", "30"])
    md = json.load(open(os.path.join(folder, "metadata.json")))
    assert (md["width"], md["height"], md["frames"]) == (64, 48, 3) and md["duration"] == 3 / md["fps"]
    names = [os.path.basename(c[1]) for c in process.COMMANDS]
    assert names == ["rgba.py", "mask_mmdet.py", "depth_anything.py", "flow_gmflow.py"]        # reference default flow band (:23); camera_colmap is not built: skipped
    depth, flow = process.COMMANDS[2], process.COMMANDS[3]
    assert depth[2:] == ["-i", folder, "--ply", "--metric", "outdoor", "--subpath", "depth_anything"]    # -e >= 1 adds --ply (:196-197, 208-210)
    assert flow[2:] == ["-i", folder, "--backwards", "--mask", "--subpath", "flow_gmflow"]                # -e >= 2 adds --flo -> subpath (:199-200, 268)
    process.main(["-i", str(clip), "-f", "flow_raft"])
    assert [os.path.basename(c[1]) for c in process.COMMANDS][-1] == "flow_raft.py" and process.COMMANDS[-1][2:] == ["-i", folder]
    assert os.path.exists(os.path.join(folder, "images", "000002.png"))                                  # rgba --subpath images


def test_unbuilt_band_is_reported_not_run(tmp_path, monkeypatch, capsys):
    _fake_bands(monkeypatch)
    process.main(["-i", str(_png(tmp_path)), "-d", "depth_midas"])
    assert "band 'depth_midas' is not built" in capsys.readouterr().out
    assert [os.path.basename(c[1]) for c in process.COMMANDS] == ["rgba.py", "mask_mmdet.py"]


def test_failed_band_fails_the_run(tmp_path, monkeypatch, capsys):
    """a band that exits non-zero (missing checkpoint ...) must not leave `process.py` exiting 0 with a partial folder."""
    real = subprocess.run

    def fake(cmd, **kw):
        name = os.path.basename(cmd[1])
        if name == "rgba.py":
            return real(cmd, **kw)
        return subprocess.CompletedProcess(cmd, 3 if name == "depth_anything.py" else 0)
    monkeypatch.setattr(process.subprocess, "run", fake)
    with pytest.raises(SystemExit) as e:
        process.main(["-i", str(_png(tmp_path)), "-d", "depth_anything"])
    assert e.value.code == 1
    assert "depth_anything (exit 3)" in capsys.readouterr().err
    assert ("mask_mmdet", 0) in process.RESULTS and ("depth_anything", 3) in process.RESULTS


def test_missing_checkpoint_is_an_error_without_opt_in(monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "bands"))
    import flow_raft
    monkeypatch.delenv("PRISMA_SYNTH", raising=False)
    flow_raft._SYNTH[0] = False
    with pytest.raises(SystemExit, match="not found"):
        flow_raft.load_weights("models/does-not-exist.pth")
    assert flow_raft.MODEL == "models/raft-sintel.pth"                    # reference bands/flow_raft.py:31
    monkeypatch.setenv("PRISMA_SYNTH", "1")
    assert "fnet.conv1.weight" in flow_raft.load_weights("models/does-not-exist.pth")


@pytest.mark.gpu
def test_image_end_to_end_on_gpu(tmp_path):
    src = _png(tmp_path)
    env = dict(os.environ, PRISMA_SYNTH="1", PRISMA_OVERWRITE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "process.py"), "-i", str(src), "-d", "depth_anything"], env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    folder = tmp_path / "gog"
    md = json.load(open(folder / "metadata.json"))
    from PIL import Image
    for band, f in (("rgba", "rgba.png"), ("mask", "mask.png"), ("depth_anything", "depth_anything.png")):
        assert md["bands"][band]["url"] == f and np.asarray(Image.open(folder / f)).shape[:2] == (440, 934), band
    assert md["bands"]["depth"] == md["bands"]["depth_anything"]                     # set_default_band (process.py:243-252)
    v = md["bands"]["depth_anything"]["values"]
    assert 0.0 < v["min"]["value"] < v["max"]["value"]                                # --metric outdoor: metres
