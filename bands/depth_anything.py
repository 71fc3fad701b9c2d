#!/usr/bin/env python3
"""depth_anything band - drop-in for /root/reference/bands/depth_anything.py on MI355X.

Same CLI (reference :254-292), same outputs (<BAND>.mp4|png, <BAND>_min.csv, <BAND>_max.csv,
metadata.json entries, :157-166, :232-251), same module-level API (BAND, init_model(), infer()).
The PyTorch model and the numpy post-process are replaced by libprisma_bands.so through
prisma_amd.engine (C ABI, include/prisma_bands.h); frames are pushed in batches instead of one by one.
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

BAND = "depth_anything"
BATCH = int(os.environ.get("PRISMA_BATCH", "32"))

model = None
data = None
args = None
ranks = None          # shard.Ranks(): one process per GPU under torchrun, world 1 otherwise
_still = None         # engine.Ops ctx of the still-image encodes (write_depth_png)


def heat_to_rgb(heat):
    """bands/common/encode.py:13-33 on the host: the ramp common.io.write_depth takes as an argument (tests compare the GPU encodes
    with it; both the video path and the still-image path encode on the GPU)."""
    hue = (1.0 - np.asarray(heat, np.float64)) * 0.65
    rgb = np.stack([hue * 6.0, hue * 6.0 + 4.0, hue * 6.0 + 2.0], axis=-1)
    return np.clip(np.abs(np.mod(rgb, 6.0) - 3.0) - 1.0, 0.0, 1.0)


def load_weights(encoder, path=""):
    """{reference state_dict name: float32 ndarray}.  The reference pulls LiheYoung/depth_anything_*14
    from the HF hub (:60); offline we read models/depth_anything_<enc>14.{npz,pth} if present and fall
    back to the seeded synthetic weights ONLY with --synthetic / PRISMA_SYNTH=1 (tests, benchmarks); otherwise a missing
    checkpoint is an error."""
    cands = [path] if path else [os.path.join("models", f"depth_anything_{encoder}14.npz"),
                                 os.path.join("models", f"depth_anything_{encoder}14.pth")]
    for c in cands:
        if c and os.path.exists(c):
            return load_checkpoint(c)
    if not shard.synthetic_allowed(getattr(args, "synthetic", False)):
        raise SystemExit(f"[{BAND}] no checkpoint found ({cands}); pass --weights, or --synthetic / PRISMA_SYNTH=1 for seeded synthetic weights")
    print(f"[{BAND}] no checkpoint found ({cands}); using seeded synthetic weights (--synthetic)", file=sys.stderr)
    return synth.depth_anything_weights(encoder, seed=1234)


def init_model(encoder=None, weights="", device=0, max_batch=BATCH):
    global model
    encoder = encoder or (args.encoder if args else "vitl")
    metric = getattr(args, "metric", "none") if args is not None else "none"
    if metric != "none":
        # reference :52-57: ZoeDepth (eval config) over the ViT-L core, checkpoints models/depth_anything_metric_depth_{indoor,outdoor}.pt
        path = weights or os.path.join("models", f"depth_anything_metric_depth_{metric}.pt")
        model = engine.DepthAnything(load_metric_weights(path), "vitl", device=device, max_batch=max_batch, metric=True)
    else:
        model = engine.DepthAnything(load_weights(encoder, weights), encoder, device=device, max_batch=max_batch)
    return model


def load_metric_weights(path):
    """ZoeDepth state dict (`core.core.*` + the metric head); `model_io.load_state_from_resource` keeps it under 'model'."""
    if path and os.path.exists(path):
        return load_checkpoint(path, wrappers=("model",))
    if not shard.synthetic_allowed(getattr(args, "synthetic", False)):
        raise SystemExit(f"[{BAND}] metric checkpoint {path!r} not found; pass --weights, or --synthetic / PRISMA_SYNTH=1")
    print(f"[{BAND}] metric checkpoint {path!r} not found; using seeded synthetic weights (--synthetic)", file=sys.stderr)
    return synth.zoe_weights()


def _flip():
    """The relative model encodes near = hot (flip), the metric models do not (reference :150,188)."""
    return getattr(args, "metric", "none") == "none" if args is not None else True


def infer(img, normalize=False):
    """uint8 RGB HxWx3 -> float32 HxW relative depth (reference :100-143)."""
    if model is None:
        init_model()
    return model.infer(img, normalize=normalize)


def write_depth_png(path, depth):
    """write_depth(path, depth, normalize=True, flip, heatmap=True, encode_range=True) (reference :176-180, :221-225;
    bands/common/io.py:138-172) with the encode on the GPU (pb_depth_encode_still): bytes equal common.io.write_depth's."""
    global _still
    if _still is None:      # its own ctx and stream: the --subpath dumps run on the sink thread while the band's ctx computes the next chunk
        _still = engine.Ops(device=ranks.device if ranks else 0)
    rgb, _, _ = _still.encode_still(depth, flip=_flip(), encode_range=True)
    write_rgb(path, rgb)


def process_image(a):
    if getattr(a, "ply", False):       # reference :168-174 write_pcl (camera intrinsics + plyfile): geometry export, SURVEY section 2 out of scope
        print(f"[{BAND}] --ply (point cloud export, reference :168-174) is not built (SURVEY.md section 2.1 geom); writing the depth image only",
              file=sys.stderr)
    img = open_rgb(a.input)
    out_folder = os.path.dirname(a.output)
    pred = infer(img)
    if data:
        data["bands"][BAND]["values"] = {"min": {"value": float(pred.min()), "type": "float"},
                                         "max": {"value": float(pred.max()), "type": "float"}}
    if a.npy:
        np.save(os.path.join(out_folder, BAND + ".npy"), pred)
    write_depth_png(a.output, pred)


def process_video(a):
    """Frames shard by rank: every rank encodes its contiguous block on its own GPU.  Rank 0 writes its own
    chunks to the video as they finish and then muxes the other ranks' chunks in frame order (shard.Relay: one chunk in
    memory at a time, nothing gathered); only the per-frame (min, max) scalars go through a collective."""
    rk = ranks or shard.Ranks()
    if getattr(a, "ply", False):
        print(f"[{BAND}] --ply only applies to still images (reference :168-174); ignored for video", file=sys.stderr)
    src = FrameReader(a.input)
    n = len(src)
    h, w = src[0].shape[:2]
    out = VideoWriter(width=w, height=h, frame_rate=src.fps, filename=a.output) if rk.main else None
    out_folder = os.path.dirname(a.output)
    if a.subpath:
        if data:
            data["bands"][BAND]["folder"] = a.subpath
        a.subpath = os.path.join(out_folder, a.subpath)
        create_folder(a.subpath)
    if model is None:
        init_model(device=rk.device)
    first, last = rk.frames(n)
    lo, hi = [], []
    want_depth = bool(a.npy or a.subpath)
    relay = shard.Relay(rk, a.output, est_bytes=(n - (last - first)) * h * w * 3)

    def emit(s, depth, rgb):
        # runs on the sink thread, chunk after chunk in order: video frames, .npy / .png dumps (reference :215-225)
        if not rk.main:
            relay.put(s, {"rgb": rgb})
        for j in range(len(rgb)):
            if rk.main:
                out.write(rgb[j])
            if a.npy and a.subpath:
                np.save(os.path.join(a.subpath, "{:05d}.npy".format(s + j)), depth[j])
            if a.subpath:
                write_depth_png(os.path.join(a.subpath, "{:05d}.png".format(s + j)), depth[j])

    # the decode of chunk i+1 and the encode / writes of chunk i-1 overlap the engine's work on chunk i
    sink = AsyncSink(depth=2)
    load = lambda s: np.stack([src[i] for i in range(s, min(last, s + BATCH))])      # noqa: E731
    for s, frames in prefetch(load, range(first, last, BATCH)):
        depth, rgb, mn, mx = model.infer_batch(frames, want_depth=want_depth, want_rgb=True, flip=_flip())
        sink.submit(emit, s, depth, rgb)
        lo += [float(v) for v in mn]
        hi += [float(v) for v in mx]
    sink.close()
    if rk.world > 1:
        # scalars first (every rank gets here when its own compute is done), then rank 0 muxes the other ranks' chunks while they
        # wait on a file signal in relay.close() - no collective is pending during the mux
        # (with a bounded spool, PRISMA_SPOOL_MAX_CHUNKS, the drain has to run DURING the gather: Relay.drain_begin)
        if rk.main:
            relay.drain_begin(n, BATCH, lambda s, c: [out.write(f) for f in c["rgb"]])
        mm = rk.gather(np.asarray([lo, hi], np.float32).T.reshape(-1, 2), n, ctx=model)
        if rk.main:
            lo, hi = [float(v) for v in mm[:, 0]], [float(v) for v in mm[:, 1]]
            relay.drain_end()
    relay.close()
    if not rk.main:
        return
    out.close()
    with open(os.path.join(out_folder, BAND + "_min.csv"), "w") as f:
        f.writelines("{}\n".format(v) for v in lo)
    with open(os.path.join(out_folder, BAND + "_max.csv"), "w") as f:
        f.writelines("{}\n".format(v) for v in hi)
    if data:
        data["bands"][BAND]["values"] = {"min": {"type": "float", "url": BAND + "_min.csv"},
                                         "max": {"type": "float", "url": BAND + "_max.csv"}}


def main(argv=None):
    global args, data, ranks
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", "-i", help="Input image/video", type=str, required=True)
    ap.add_argument("--output", "-o", help="Output image/video", type=str, default="")
    ap.add_argument("--npy", "-n", help="Save numpy data", action="store_true")
    ap.add_argument("--ply", "-p", help="Create point cloud PLY", action="store_true")
    ap.add_argument("--subpath", "-d", help="subpath to frames", type=str, default="")
    ap.add_argument("--encoder", type=str, default="vitl", choices=["vits", "vitb", "vitl"])
    ap.add_argument("--metric", help="Use a metric model", type=str, default="none", choices=["none", "indoor", "outdoor"])
    ap.add_argument("--weights", help="checkpoint (.npz / .pth state dict); default models/depth_anything_<encoder>14.*", default="")
    ap.add_argument("--synthetic", help="seeded synthetic weights when no checkpoint is found (tests / benchmarks)", action="store_true")
    args = ap.parse_args(argv)
    data = load_metadata(args.input)
    if data:
        print("PRISMA metadata found and loaded")
        folder = args.input
        args.input = get_url(folder, data, "rgba")
        args.output = get_target(args.input, data, band=BAND, target=args.output, force_extension="png")
        meta_path = folder
    else:
        meta_path = args.input
        if not args.output:
            ext = os.path.basename(args.input).rsplit(".", 1)[1]
            args.output = os.path.join(os.path.dirname(args.input), BAND + "." + (ext if is_video(args.input) else "png"))
    ranks = shard.Ranks()
    if ranks.main:
        check_overwrite(args.output)
    init_model(args.encoder, args.weights, device=ranks.device)
    if is_video(args.output):
        process_video(args)
    elif ranks.main:
        process_image(args)
    if ranks.main:
        write_metadata(meta_path, data)
    ranks.close()


if __name__ == "__main__":
    main()
