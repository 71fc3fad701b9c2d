#!/usr/bin/env python3
"""flow_raft band - drop-in for /root/reference/bands/flow_raft.py on MI355X.

Same CLI (reference :169-225), same outputs (<BAND>.mp4, <BAND>.csv with the per-frame max
displacement, optional <BAND>_bwd.mp4, metadata entries :143-166), same module API (BAND, init_model(),
infer()).  Frames are pushed to libprisma_bands.so in overlapping chunks; every frame is encoded by
fnet / cnet once instead of twice per pair.
"""
import argparse
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_ROOT, os.path.dirname(os.path.abspath(__file__))):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from common.io import FrameReader, VideoWriter, check_overwrite, write_flo, write_flow_png  # noqa: E402
from common.meta import get_target, get_url, is_video, load_metadata, write_metadata  # noqa: E402
from prisma_amd import engine, synth  # noqa: E402

BAND = "flow_raft"
MODEL = "models/raft-things.pth"
ITERATIONS = 20
CHUNK = int(os.environ.get("PRISMA_BATCH", "16"))

model = None
data = None


def load_weights(path):
    """Checkpoint keys carry a `module.` prefix from DataParallel (reference :42-44): strip it."""
    if path and os.path.exists(path):
        if path.endswith(".npz"):
            z = np.load(path)
            sd = {k: z[k] for k in z.files}
        else:
            import torch
            sd = {k: v.numpy() for k, v in torch.load(path, map_location="cpu").items()}
        return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    print(f"[{BAND}] checkpoint {path!r} not found; using seeded synthetic weights", file=sys.stderr)
    return synth.raft_weights(seed=4321)


def init_model(args=None, device=0):
    global model
    if args is not None and (getattr(args, "small", False) or getattr(args, "alternate_corr", False)):
        raise NotImplementedError("--small / --alternate_corr select other RAFT variants; only the basic model is built")
    model = engine.FlowRaft(load_weights(getattr(args, "model", MODEL) if args else MODEL), device=device)
    return model


def infer(args, image1, image2):
    """Reference signature (:51-66): image1 = [prev, curr], image2 = [curr, prev] as float CHW 0..255 (already
    scaled).  Returns (fwd, bwd, None, None) with flows as float32 [H', W', 2]."""
    if model is None:
        init_model(args)
    a = np.ascontiguousarray(np.asarray(image1)[0].transpose(1, 2, 0)).astype(np.uint8)
    b = np.ascontiguousarray(np.asarray(image2)[0].transpose(1, 2, 0)).astype(np.uint8)
    if getattr(args, "output_mask", "") or getattr(args, "subpath_mask", ""):
        flow, _, _, mask = model.infer_sequence_masks(np.stack([a, b]), scale=1.0, iters=args.iterations, want_rgb=False)
        return flow[0, 0], flow[0, 1], mask[0, 0], mask[0, 1]
    flow, _, _ = model.infer_sequence(np.stack([a, b]), scale=1.0, iters=args.iterations, backward=True, want_rgb=False)
    return flow[0, 0], flow[0, 1], None, None


def _mask_rgb(mask):
    """bands/common/flow.py:75-78: True -> 255, three equal channels."""
    return np.repeat((mask.astype(np.uint8) * 255)[..., None], 3, axis=-1)


def process_video(args):
    src = FrameReader(args.input)
    n = len(src)
    h, w = src[0].shape[:2]
    base = args.output.rsplit(".", 1)[0]
    ext = args.output.rsplit(".", 1)[1]
    want_mask = bool(args.output_mask or args.subpath_mask)
    both = args.backwards or want_mask or bool(args.subpath)
    fwd_video = VideoWriter(width=w, height=h, frame_rate=src.fps, filename=args.output)
    bwd_video = VideoWriter(width=w, height=h, frame_rate=src.fps, filename=base + "_bwd." + ext) if args.backwards else None
    fwd_mask_video = bwd_mask_video = None
    if args.output_mask:
        mbase, mext = args.output_mask.rsplit(".", 1)
        fwd_mask_video = VideoWriter(width=w, height=h, frame_rate=src.fps, filename=args.output_mask)
        if args.backwards:
            bwd_mask_video = VideoWriter(width=w, height=h, frame_rate=src.fps, filename=mbase + "_bwd." + mext)
    if model is None:
        init_model(args)
    max_disps = []
    sh, sw = engine.flow_out_size(h, w, args.scale)
    want_flow = bool(args.subpath or args.subpath_mask)

    def emit(idx, rgb_f, rgb_b, mx, flow_f, flow_b, mask_f, mask_b):
        fwd_video.write(rgb_f)
        max_disps.append(mx)
        if fwd_mask_video:
            fwd_mask_video.write(_mask_rgb(mask_f))
        if bwd_mask_video:
            bwd_mask_video.write(_mask_rgb(mask_b))
        if bwd_video:
            bwd_video.write(rgb_b)
        if args.subpath:        # the reference crashes here (common/flow.py:91 shadows io.write_flow); write the .flo it meant to
            write_flo(os.path.join(args.subpath + "_fwd", "%04d.flo" % idx), flow_f)
            if args.backwards:
                write_flo(os.path.join(args.subpath + "_bwd", "%04d.flo" % idx), flow_b)
        if args.subpath_mask:
            write_flow_png(os.path.join(args.subpath_mask + "_fwd", "%04d.png" % idx), flow_f, mask_f)
            if args.backwards:
                write_flow_png(os.path.join(args.subpath_mask + "_bwd", "%04d.png" % idx), flow_b, mask_b)

    for s in range(0, n - 1, CHUNK):
        frames = np.stack([src[i] for i in range(s, min(n, s + CHUNK + 1))])      # 1-frame halo
        mask = None
        if want_mask:
            flow, rgb, mx, mask = model.infer_sequence_masks(frames, scale=args.scale, iters=args.iterations,
                                                             want_flow=want_flow, want_rgb=True)
        else:
            flow, rgb, mx = model.infer_sequence(frames, scale=args.scale, iters=args.iterations, backward=both,
                                                 want_flow=want_flow, want_rgb=True)
        for j in range(len(frames) - 1):
            emit(s + j, rgb[j, 0], rgb[j, 1] if both else None, float(mx[j, 0]),
                 flow[j, 0] if want_flow else None, flow[j, 1] if want_flow and both else None,
                 mask[j, 0] if want_mask else None, mask[j, 1] if want_mask else None)
    # last frame: zero flow -> 0/0 -> NaN -> uint8 0, max displacement 0.0, all-False masks (reference :116-131)
    zero = np.zeros((sh, sw, 3), np.uint8)
    zf = np.zeros((sh, sw, 2), np.float32)
    zm = np.zeros((sh, sw), bool)
    emit(n - 1, zero, zero, 0.0, zf, zf, zm, zm)
    for v in (fwd_video, bwd_video, fwd_mask_video, bwd_mask_video):
        if v:
            v.close()
    with open(base + ".csv", "w") as f:
        f.writelines("{}\n".format(e) for e in max_disps)
    if data:
        data["bands"][BAND] = {"url": BAND + "." + ext, "values": {"dist": {"type": "float", "url": BAND + ".csv"}}}
        if args.subpath:
            data["bands"][BAND]["folder"] = args.subpath
        if args.backwards:
            data["bands"][BAND + "_bwd"] = {"url": BAND + "_bwd." + ext}
            if args.subpath:
                data["bands"][BAND + "_bwd"]["folder"] = args.subpath + "_bwd"
        if args.output_mask:
            data["bands"][BAND + "_mask"] = {"url": BAND + "_mask." + ext}
            if args.backwards:
                data["bands"][BAND + "_mask_bwd"] = {"url": BAND + "_mask_bwd." + ext}


def main(argv=None):
    global data
    ap = argparse.ArgumentParser()
    ap.add_argument("-input", "-i", "--input", dest="input", help="input", type=str, required=True)
    ap.add_argument("-output", "-o", "--output", dest="output", help="output", type=str, default="")
    ap.add_argument("--subpath", "-d", help="Subpath to frames", type=str, default="")
    ap.add_argument("--backwards", "-b", help="Backward video", action="store_true")
    ap.add_argument("--mask", help="Compute consistency mask", action="store_true")
    ap.add_argument("--output_mask", help="Mask video", type=str, default="")
    ap.add_argument("--subpath_mask", help="Subpath to mask frames", type=str, default="")
    ap.add_argument("--iterations", help="number of iterations", type=int, default=ITERATIONS)
    ap.add_argument("--model", "-m", help="model path", type=str, default=MODEL)
    ap.add_argument("--scale", type=float, default=0.75, help="scale factor")
    ap.add_argument("--small", action="store_true", help="use small model")
    ap.add_argument("--mixed_precision", action="store_true", help="use mixed precision")
    ap.add_argument("--alternate_corr", action="store_true", help="use efficent correlation implementation")
    args = ap.parse_args(argv)
    data = load_metadata(args.input)
    if data:
        folder = args.input
        args.input = get_url(folder, data, "rgba")
        args.output = get_target(args.input, data, band=BAND, target=args.output)
        if args.mask:
            args.output_mask = get_target(args.input, data, band=BAND + "_mask")
        meta_path = folder
    else:
        meta_path = args.input
        if not args.output:
            args.output = os.path.join(os.path.dirname(args.input), BAND + "." + os.path.basename(args.input).rsplit(".", 1)[1])
    if not is_video(args.output):
        raise SystemExit(f"[{BAND}] needs a video input")
    check_overwrite(args.output)
    input_folder = os.path.dirname(args.input)
    for attr in ("subpath", "subpath_mask"):
        if getattr(args, attr):
            setattr(args, attr, os.path.join(input_folder, getattr(args, attr)))
            os.makedirs(getattr(args, attr) + "_fwd", exist_ok=True)
            if args.backwards:
                os.makedirs(getattr(args, attr) + "_bwd", exist_ok=True)
    init_model(args)
    process_video(args)
    write_metadata(meta_path, data)


if __name__ == "__main__":
    main()
