"""CPU: the SOLOv2 oracle (oracle/solov2_oracle.py, PARITY UNPINNED - mmcv / cv2 / config absent) against
hand-computable cases and independent loop restatements of the vendored post-processing, plus the host side of the
mask band."""
import os
import sys

import numpy as np
import torch

from oracle import solov2_oracle as SO
from prisma_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bands"))


def test_rescale_and_pad_geometry():
    # 1080p and 720p both -> 1333 x 750 -> padded 1344 x 768
    assert SO.rescale_size(1080, 1920, 1333, 800)[:2] == (750, 1333)
    assert SO.rescale_size(720, 1280, 1333, 800)[:2] == (750, 1333)
    assert SO.rescale_size(440, 934, 1333, 800)[:2] == (628, 1333)
    x, meta = SO.preprocess(np.zeros((108, 192, 3), np.uint8), synth.MASK_CFGS["tiny"])
    assert meta == {"img_shape": (180, 320), "ori_shape": (108, 192), "pad_shape": (192, 320)}
    assert x.shape == (1, 3, 192, 320) and not x[0, :, 180:].any()
    mean = np.asarray(SO.MEAN, np.float32)
    std = np.asarray(SO.STD, np.float64)
    assert np.allclose(x[0, :, 0, 0], (0 - mean) * (1.0 / std).astype(np.float32), rtol=0, atol=0)


def test_linear_resize_u8():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(SO.cv_resize_linear_u8(img, 37, 53), img)                      # identity size
    flat = np.full((20, 30, 3), 201, np.uint8)
    assert np.array_equal(SO.cv_resize_linear_u8(flat, 33, 47), np.full((33, 47, 3), 201, np.uint8))
    # against float bilinear with the same half-pixel geometry: fixed point stays within one grey level
    out = SO.cv_resize_linear_u8(img, 61, 90).astype(np.float64)
    t = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
    ref = torch.nn.functional.interpolate(t, size=(61, 90), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(out - ref).max() <= 1.0
    down = SO.cv_resize_linear_u8(img, 20, 31).astype(np.float64)                        # no antialias: plain 2x2 taps
    ref = torch.nn.functional.interpolate(t, size=(20, 31), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(down - ref).max() <= 1.0


def test_points_nms_hand_case():
    logit = torch.full((1, 1, 3, 3), -3.0)
    logit[0, 0, 1, 1] = 2.0
    logit[0, 0, 1, 2] = 1.0      # right neighbour of the peak: its window (up-left 2x2) contains the peak -> dropped
    logit[0, 0, 0, 0] = 0.5      # top-left: window is itself only -> kept
    s = SO.points_nms_scores(logit)[0, :, 0].reshape(3, 3)
    assert s[1, 1] == torch.sigmoid(torch.tensor(2.0)) and s[1, 2] == 0 and s[0, 0] == torch.sigmoid(torch.tensor(0.5))
    assert s[2, 2] == 0 and s[0, 1] == 0                     # their windows contain (1,1) resp. (0,0)
    assert s[2, 0] == torch.sigmoid(torch.tensor(-3.0))      # window {(1,0), (2,0)}: a tie keeps the cell


def test_matrix_nms_against_reference_vectors(golden_dir):
    """tests/golden/matrix_nms.npz: inputs and outputs of the REAL mask_matrix_nms (reference bands/mmdet/core/post_processing/
    matrix_nms.py:5-121, loaded by oracle/make_golden.py `nms` - the file imports only torch).  PINS the oracle's Matrix NMS."""
    import dataclasses
    import os
    z = np.load(os.path.join(golden_dir, "matrix_nms.npz"))
    for case in range(3):
        n, h, w = [int(v) for v in z[f"shape{case}"]]
        masks = np.unpackbits(z[f"masks{case}"], axis=-1)[..., :w].astype(bool)
        nms_pre, max_num, filter_thr, sigma = z[f"cfg{case}"]
        cfg = dataclasses.replace(synth.MASK_CFGS["tiny"], nms_pre=int(nms_pre), max_per_img=int(max_num), filter_thr=float(filter_thr),
                                  sigma=float(sigma))
        sc, lb, keep = SO.matrix_nms(torch.from_numpy(masks), torch.from_numpy(z[f"labels{case}"]), torch.from_numpy(z[f"scores{case}"]),
                                     torch.from_numpy(z[f"areas{case}"]), cfg)
        assert np.array_equal(sc.numpy(), z[f"out_scores{case}"]) and np.array_equal(lb.numpy(), z[f"out_labels{case}"])
        assert np.array_equal(keep.numpy(), z[f"out_keep{case}"])
        assert len(keep) == min(int(max_num), int((sc.numpy() >= filter_thr).sum()))


def _matrix_nms_loops(masks, labels, scores, areas, cfg):
    """Independent restatement of core/post_processing/matrix_nms.py:52-121 with explicit loops."""
    order = sorted(range(len(scores)), key=lambda i: -scores[i])[:cfg.nms_pre]
    m = [masks[i].reshape(-1).astype(np.float64) for i in order]
    n = len(order)
    iou = np.zeros((n, n))
    for i in range(n):
        for j in range(i + 1, n):
            if labels[order[i]] == labels[order[j]]:
                inter = float(m[i] @ m[j])
                iou[i, j] = inter / (areas[order[i]] + areas[order[j]] - inter)
    comp = iou.max(0)
    out = []
    for j in range(n):
        coeff = min(np.exp(-cfg.sigma * iou[i, j] ** 2) / np.exp(-cfg.sigma * comp[i] ** 2) for i in range(n))
        out.append(scores[order[j]] * coeff)
    keep = [(out[j], order[j]) for j in range(n) if out[j] >= cfg.filter_thr]
    keep.sort(key=lambda t: -t[0])
    return keep[:cfg.max_per_img]


def test_matrix_nms_against_loops():
    cfg = synth.MASK_CFGS["tiny"]
    rng = np.random.default_rng(3)
    n, h, w = 40, 24, 32
    yy, xx = np.mgrid[0:h, 0:w]
    masks = np.stack([((yy - rng.uniform(4, 20)) ** 2 + (xx - rng.uniform(4, 28)) ** 2) < rng.uniform(9, 60) for _ in range(n)])
    labels = rng.integers(0, 3, n)
    scores = rng.uniform(0.06, 0.9, n).astype(np.float32)
    areas = masks.reshape(n, -1).sum(1).astype(np.float32)
    sc, lb, keep = SO.matrix_nms(torch.from_numpy(masks), torch.from_numpy(labels), torch.from_numpy(scores),
                                 torch.from_numpy(areas), cfg)
    ref = _matrix_nms_loops(masks, labels, scores, areas, cfg)
    assert [k for _, k in ref] == keep.tolist()
    assert np.allclose([s for s, _ in ref], sc.numpy(), rtol=1e-5)
    assert np.array_equal(lb.numpy(), labels[keep.numpy()])


def test_band_accumulation_wraps_modulo_256():
    H, W = 6, 8
    masks = np.zeros((3, H, W), bool)
    masks[0, :4, :4] = True
    masks[1, 2:, 2:] = True
    masks[2, :, :] = True
    names = ("person", "car", "dog")
    img = SO.band_mask(np.array([0.9, 0.8, 0.7], np.float32), np.array([0, 2, 1]), masks, names, ("person", "dog"), 0.5, (H, W))
    assert img.shape == (H, W, 3) and img[0, 0, 0] == 255 and img[3, 3, 1] == 254 and img[5, 7, 2] == 255 and img[0, 7, 0] == 0
    # score below getTotalMasks' fixed 0.5 is never drawn, whatever --confidence says (mask_mmdet.py:43-49)
    img = SO.band_mask(np.array([0.45], np.float32), np.array([0]), masks[:1], names, ("person",), 0.1, (H, W))
    assert not img.any()
    img = SO.band_mask(np.array([0.65], np.float32), np.array([0]), masks[:1], names, ("person",), 0.7, (H, W))
    assert not img.any()


def test_oracle_end_to_end_self_vector(golden_dir):
    """Regression pin of the oracle against its own committed output (NOT a reference vector)."""
    z = np.load(os.path.join(golden_dir, "solov2_tiny_180x300.npz"))
    cfg = synth.MASK_CFGS["tiny"]
    w = synth.solov2_weights(cfg)
    fr = synth.frames(1, 180, 300, seed=int(z["frame_seed"]))[0]
    x, meta = SO.preprocess(fr, cfg)
    kps, cps, mf = SO.network(w, cfg, x)
    assert np.allclose(cps[4].numpy(), z["cls_logit4"], rtol=1e-4, atol=1e-4)
    assert np.allclose(mf.numpy()[0, ::16, ::4, ::4], z["mask_feats_sub"], rtol=1e-4, atol=1e-4)
    sc, lb, mk = SO.get_results(cfg, kps, cps, mf, meta["img_shape"], meta["ori_shape"])
    assert np.array_equal(lb.numpy(), z["labels"]) and np.allclose(sc.numpy(), z["scores"], rtol=1e-4)
    img = SO.band_mask(sc, lb, mk, synth.COCO_CLASSES, synth.BAND_CLASSES, 0.5, meta["ori_shape"])
    assert (img != z["mask_image"]).mean() < 1e-3


def test_mask_band_host_helpers():
    import mask_mmdet as band
    assert band.BAND == "mask" and len(band.CLASSES) == 11 and band.keep_ids()[:3] == [0, 14, 15]
    m = np.zeros((40, 40, 3), np.uint8)
    m[10:30, 10:30] = 255
    assert np.array_equal(band._colmap(m)[..., 1], 255 - m[..., 0])


def test_band_sdf_restatement_and_tables():
    """The --sdf image (reference mask_mmdet.py:64-69,150-152): oracle restatement on a square, and the (side, squared distance) ->
    byte tables the engine looks up (prisma_amd.engine.sdf_tables) against it - every pixel of the restated image is the table entry
    of its exact squared distance, which is all the device has to compute."""
    from scipy.ndimage import distance_transform_edt
    from prisma_amd.engine import SDF_NTAB, sdf_tables
    m = np.zeros((200, 260, 3), np.uint8)
    m[60:150, 80:200] = 255
    m[0:5, 250:260] = 254                                   # a wrapped overlap count is still "inside"
    img = SO.band_sdf(m)
    g = img[..., 1].astype(int)
    assert np.array_equal(img[..., 0], m[..., 0]) and np.array_equal(img[..., 2], m[..., 2])
    assert g[0, 0] < g[59, 79] < g[60, 80] < g[100, 140]
    assert g[70, 140] == int((1.0 - ((127.0 - 11.0) / 255.0 - 0.25) * 2.0) * 255)       # 11 px from the background, inside
    to, ti = sdf_tables()
    assert len(to) == len(ti) == SDF_NTAB and to[-1] == 0 and ti[-1] == 255
    assert (to[4065:] == 0).all() and to[4064] > 0 and (ti[4001:] == 255).all() and ti[4000] < 255
    inside = m[..., 0] != 0
    n_out = np.rint(distance_transform_edt(~inside) ** 2).astype(np.int64)
    n_in = np.rint(distance_transform_edt(inside) ** 2).astype(np.int64)
    want = np.where(inside, ti[np.minimum(n_in, SDF_NTAB - 1)], to[np.minimum(n_out, SDF_NTAB - 1)])
    assert np.array_equal(want, img[..., 1])
    # degenerate frames: snowy.generate_sdf keeps its INF = 1e20 start value when a class is empty, the remap saturates
    assert (SO.band_sdf(np.zeros((40, 50, 3), np.uint8))[..., 1] == 0).all()
    assert (SO.band_sdf(np.full((40, 50, 3), 255, np.uint8))[..., 1] == 255).all()
