#!/usr/bin/env python3
"""mask (mmdet / SOLOv2) band - drop-in for /root/reference/bands/mask_mmdet.py on MI355X.

Same CLI (reference :164-199), same outputs (mask.png | mask.mp4 with the accumulated instance masks of the 11
animate COCO classes, optional COLMAP black/white frames in --subpath, optional SDF in the green channel,
metadata entry `bands.mask = {url, ids}` :113-115,158-161), same module-level names (BAND, CLASSES, init_model()).
mmdet's init_detector / inference_detector are replaced by libprisma_bands.so through prisma_amd.engine; frames
are pushed in batches instead of one by one.
"""
import argparse
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_ROOT, os.path.dirname(os.path.abspath(__file__))):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from common.io import FrameReader, VideoWriter, check_overwrite, create_folder, open_rgb, write_rgb  # noqa: E402
from common.ckpt import load_checkpoint  # noqa: E402
from common.meta import get_target, get_url, is_video, load_metadata, write_metadata  # noqa: E402
from common.pipe import AsyncSink, prefetch  # noqa: E402
from prisma_amd import engine, shard, synth  # noqa: E402

BAND = "mask"
MODEL = "models/solov2_r101_fpn_3x_coco_20220511_095119-c559a076.pth"
CLASSES = list(synth.BAND_CLASSES)
CONFIDENCE_THRESHOLD = 0.5
BATCH = int(os.environ.get("PRISMA_BATCH", "32"))

model = None
data = None
_SYNTH = [False]      # --synthetic
ranks = None          # shard.Ranks(): one process per GPU under torchrun, world 1 otherwise


def load_weights(path, cfg):
    """mmdet checkpoints keep the tensors under 'state_dict'."""
    if path and os.path.exists(path):
        return load_checkpoint(path, wrappers=("state_dict",))
    if not shard.synthetic_allowed(_SYNTH[0]):
        raise SystemExit(f"[{BAND}] checkpoint {path!r} not found; pass --weights, or --synthetic / PRISMA_SYNTH=1 for seeded synthetic weights")
    print(f"[{BAND}] checkpoint {path!r} not found; using seeded synthetic weights (--synthetic)", file=sys.stderr)
    return synth.solov2_weights(cfg)


def init_model(arch="r101", weights=MODEL, device=0, max_batch=BATCH):
    global model
    cfg = synth.MASK_CFGS[arch]
    model = engine.MaskMMDet(load_weights(weights, cfg), cfg, device=device, max_batch=max_batch)
    model.CLASSES = synth.COCO_CLASSES
    return model


def keep_ids():
    return [synth.COCO_CLASSES.index(c) for c in CLASSES]


def set_sdf(on):
    """--sdf (reference :64-69,150-152; process.py passes it on every run): the engine writes the clamped signed distance field of
    each id image into its green channel on the GPU (engine.MaskMMDet.set_sdf; the host restatement the tests compare it with is
    oracle/solov2_oracle.py band_sdf)."""
    model.set_sdf(bool(on))


def _colmap(masks_u8):
    """COLMAP wants black objects on white (reference :149-150 writes 255 - masks BEFORE the SDF goes into G): the id image has the
    same byte in all three channels, so the pre-SDF image is its red channel three times."""
    return 255 - np.repeat(masks_u8[..., :1], 3, axis=-1)


def process_image(args):
    img = open_rgb(args.input)
    set_sdf(args.sdf)
    masks = model.infer_batch(img[None], args.confidence, keep_ids())[0]
    write_rgb(args.output, masks)
    data["bands"][BAND] = {"url": os.path.basename(args.output), "ids": CLASSES}


def process_video(args):
    """Frames shard by rank (no cross-frame state, reference :131-154).  Rank 0 writes its own chunks to the video as they
    finish (sink thread) and then muxes the other ranks' chunks in frame order through shard.Relay; nothing is gathered
    and no list holds the whole video."""
    rk = ranks or shard.Ranks()
    src = FrameReader(args.input)
    n = len(src)
    h, w = src[0].shape[:2]
    if args.subpath:
        args.subpath = os.path.join(os.path.dirname(args.output), args.subpath)
        create_folder(args.subpath)
    first, last = rk.frames(n)
    out = VideoWriter(width=w, height=h, frame_rate=src.fps, filename=args.output) if rk.main else None
    relay = shard.Relay(rk, args.output, est_bytes=(n - (last - first)) * h * w * 3)

    set_sdf(args.sdf)

    def emit(s, masks):              # sink thread, chunks in order (the SDF, if asked for, is already in G: set_sdf)
        if rk.main:
            for f in masks:
                out.write(f)
        else:
            relay.put(s, {"mask": masks})
        if args.subpath:
            for j in range(len(masks)):
                write_rgb(os.path.join(args.subpath, "{:05d}.png".format(s + j)), _colmap(masks[j]))

    sink = AsyncSink(depth=2)
    load = lambda s: np.stack([src[i] for i in range(s, min(last, s + BATCH))])      # noqa: E731
    for s, frames in prefetch(load, range(first, last, BATCH)):      # the next chunk decodes while this one is on the GPU
        sink.submit(emit, s, model.infer_batch(frames, args.confidence, keep_ids()))
    sink.close()
    if rk.world > 1 and rk.main:
        relay.drain(n, BATCH, lambda s, c: [out.write(f) for f in c["mask"]])
    relay.close()
    if not rk.main:
        return
    out.close()
    # the reference replaces the whole entry here, which also drops the `folder` key it set earlier (:127-129,158-161)
    data["bands"][BAND] = {"url": os.path.basename(args.output), "ids": CLASSES}


def main(argv=None):
    global data, ranks
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", "-i", help="input", type=str, required=True)
    ap.add_argument("--output", "-o", help="output", type=str, default="")
    ap.add_argument("--confidence", "-c", help="confidence threshold", type=float, default=CONFIDENCE_THRESHOLD)
    ap.add_argument("--sdf", "-s", help="Encode SDF on GREEN channel", action="store_true")
    ap.add_argument("--subpath", help="Mask Subpath to frames", type=str, default="")
    ap.add_argument("--arch", help="backbone / geometry preset (prisma_amd.synth.MASK_CFGS)", default="r101")
    ap.add_argument("--weights", help="mmdet checkpoint (.pth) or .npz state dict", default=MODEL)
    ap.add_argument("--synthetic", action="store_true", help="seeded synthetic weights when the checkpoint is missing (tests / benchmarks)")
    args = ap.parse_args(argv)
    _SYNTH[0] = args.synthetic
    data = load_metadata(args.input)
    meta_path = args.input
    if data:
        print("PRISMA metadata found and loaded")
        args.input = get_url(meta_path, data, "rgba")
        args.output = get_target(args.input, data, band=BAND, target=args.output, force_extension="png")
    else:
        data = {"bands": {}}
        if not args.output:
            ext = os.path.basename(args.input).rsplit(".", 1)[1]
            args.output = os.path.join(os.path.dirname(args.input), BAND + "." + (ext if is_video(args.input) else "png"))
    ranks = shard.Ranks()
    if ranks.main:
        check_overwrite(args.output)
    init_model(args.arch, args.weights, device=ranks.device)
    if is_video(args.output):
        process_video(args)
    elif ranks.main:
        process_image(args)
    if ranks.main:
        write_metadata(meta_path, data)
    ranks.close()


if __name__ == "__main__":
    main()
