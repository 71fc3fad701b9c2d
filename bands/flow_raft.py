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
from common.ckpt import load_checkpoint  # noqa: E402
from common.meta import get_target, get_url, is_video, load_metadata, write_metadata  # noqa: E402
from common.pipe import AsyncSink, prefetch  # noqa: E402
from prisma_amd import engine, shard, synth  # noqa: E402

BAND = "flow_raft"
MODEL = "models/raft-sintel.pth"            # reference :31
ITERATIONS = 20
CHUNK = int(os.environ.get("PRISMA_BATCH", "16"))

model = None
data = None
_SYNTH = [False]      # --synthetic
ranks = None          # shard.Ranks(): one process per GPU under torchrun, world 1 otherwise
SUBPATH_NEEDS_BOTH = True   # flow_gmflow's wrapper clears it around its call of process_video


def load_weights(path):
    """Checkpoint keys carry a `module.` prefix from DataParallel (reference :42-44): strip it."""
    if path and os.path.exists(path):
        return load_checkpoint(path, strip_prefix="module.")
    if not shard.synthetic_allowed(_SYNTH[0]):
        raise SystemExit(f"[{BAND}] checkpoint {path!r} not found; pass --model, or --synthetic / PRISMA_SYNTH=1 for seeded synthetic weights")
    print(f"[{BAND}] checkpoint {path!r} not found; using seeded synthetic weights (--synthetic)", file=sys.stderr)
    return synth.raft_weights(seed=4321)


def init_model(args=None, device=0):
    global model
    if args is not None and getattr(args, "small", False):
        # reference raft.py:28-53: both `if args.small:` branches sit inside string literals (dead code) - RAFT(args) builds BasicEncoder /
        # BasicUpdateBlock (hidden 128, corr radius 4, 5.26 M parameters) whatever the flag says, and loads args.model into it.  Drop-in
        # behaviour is therefore: accept the flag, run the basic model (round 4 refused it as "another network"; it never was one here)
        print(f"[{BAND}] --small: the reference builds the basic RAFT regardless (bands/raft/raft.py:28-53, the small branches are commented out); flag ignored", file=sys.stderr)
    if args is not None and getattr(args, "alternate_corr", False):
        # reference raft.py:103-106: AlternateCorrBlock computes the same correlations on the fly (memory saving, needs the alt_cuda_corr
        # extension); results are those of CorrBlock, and the volume is no memory problem in 288 GB - run the normal path
        print(f"[{BAND}] --alternate_corr: same result as the default correlation block (reference raft.py:103-106); flag ignored", file=sys.stderr)
    if args is not None and getattr(args, "mixed_precision", False):
        print(f"[{BAND}] --mixed_precision: the engine's precision is set by PRISMA_PRECISION (split-fp16 by default); flag ignored", file=sys.stderr)
    _SYNTH[0] = bool(getattr(args, "synthetic", False))
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
    """Pairs (i, i+1) shard by rank in contiguous blocks with a one-frame halo.  Every chunk is written as soon
    as it is done: rank 0 feeds its own chunks to the VideoWriters from a sink thread (the next chunk is already on the
    GPU) and then muxes the other ranks' chunks in frame order through shard.Relay; only the per-pair max displacements
    (4 bytes each) go through a collective.  No list or tensor ever holds the whole video."""
    rk = ranks or shard.Ranks()
    src = FrameReader(args.input)
    n = len(src)
    h, w = src[0].shape[:2]
    base = args.output.rsplit(".", 1)[0]
    ext = args.output.rsplit(".", 1)[1]
    want_mask = bool(args.output_mask or args.subpath_mask)
    # flow_gmflow predicts the backward flow only for --backwards / masks (reference flow_gmflow.py:86); with --subpath alone (what
    # process.py passes) a second global matching would be computed and thrown away
    both = args.backwards or want_mask or (SUBPATH_NEEDS_BOTH and bool(args.subpath))
    want_flow = bool(args.subpath or args.subpath_mask)
    if model is None:
        init_model(args, device=rk.device)
    sh, sw = engine.flow_out_size(h, w, args.scale)
    first, last = rk.frames(n - 1)                       # pair indices owned by this rank
    fwd_video = bwd_video = fwd_mask_video = bwd_mask_video = None
    if rk.main:
        fwd_video = VideoWriter(width=w, height=h, frame_rate=src.fps, filename=args.output)
        bwd_video = VideoWriter(width=w, height=h, frame_rate=src.fps, filename=base + "_bwd." + ext) if args.backwards else None
        if args.output_mask:
            mbase, mext = args.output_mask.rsplit(".", 1)
            fwd_mask_video = VideoWriter(width=w, height=h, frame_rate=src.fps, filename=args.output_mask)
            if args.backwards:
                bwd_mask_video = VideoWriter(width=w, height=h, frame_rate=src.fps, filename=mbase + "_bwd." + mext)
    streams = (1 + bool(args.backwards)) * (1 + bool(args.output_mask))
    relay = shard.Relay(rk, args.output, est_bytes=((n - 1) - (last - first)) * sh * sw * 3 * streams)
    mxs = []

    def write_chunk(_s, c):          # rank 0 only: one chunk of encoded pairs into every open video, in order
        for j in range(len(c["rgb_f"])):
            fwd_video.write(c["rgb_f"][j])
            if bwd_video:
                bwd_video.write(c["rgb_b"][j])
            if fwd_mask_video:
                fwd_mask_video.write(_mask_rgb(c["mask_f"][j]))
            if bwd_mask_video:
                bwd_mask_video.write(_mask_rgb(c["mask_b"][j]))

    def emit(s, flow, rgb, mask):    # sink thread, chunks in order
        c = {"rgb_f": rgb[:, 0]}
        if args.backwards:
            c["rgb_b"] = rgb[:, 1]
        if args.output_mask:
            c["mask_f"] = mask[:, 0]
            if args.backwards:
                c["mask_b"] = mask[:, 1]
        if rk.main:
            write_chunk(s, c)
        else:
            relay.put(s, c)
        for j in range(len(rgb)):
            if args.subpath:    # the reference crashes here (common/flow.py:91 shadows io.write_flow); write the .flo it meant to
                write_flo(os.path.join(args.subpath + "_fwd", "%04d.flo" % (s + j)), flow[j, 0])
                if args.backwards:
                    write_flo(os.path.join(args.subpath + "_bwd", "%04d.flo" % (s + j)), flow[j, 1])
            if args.subpath_mask:
                write_flow_png(os.path.join(args.subpath_mask + "_fwd", "%04d.png" % (s + j)), flow[j, 0], mask[j, 0])
                if args.backwards:
                    write_flow_png(os.path.join(args.subpath_mask + "_bwd", "%04d.png" % (s + j)), flow[j, 1], mask[j, 1])

    sink = AsyncSink(depth=2)
    load = lambda s: np.stack([src[i] for i in range(s, min(last, s + CHUNK) + 1)])  # noqa: E731  (1-frame halo)
    for s, frames in prefetch(load, range(first, last, CHUNK)):     # the next chunk decodes while this one is on the GPU
        mask = None
        if want_mask:
            flow, rgb, mx, mask = model.infer_sequence_masks(frames, scale=args.scale, iters=args.iterations,
                                                             want_flow=want_flow, want_rgb=True)
        else:
            flow, rgb, mx = model.infer_sequence(frames, scale=args.scale, iters=args.iterations, backward=both,
                                                 want_flow=want_flow, want_rgb=True)
        sink.submit(emit, s, flow, rgb, mask)
        mxs += [np.float32(v) for v in mx[:, 0]]
    sink.close()
    mx_all = np.asarray(mxs, np.float32)
    if rk.world > 1:
        # the gather comes before the drain: no collective pending while rank 0 muxes - unless the spool is bounded
        # (PRISMA_SPOOL_MAX_CHUNKS), in which case the drain has to run during the gather (Relay.drain_begin)
        if rk.main:
            relay.drain_begin(n - 1, CHUNK, write_chunk)
        mx_all = rk.gather(mx_all, n - 1, ctx=model)
        if rk.main:
            relay.drain_end()
    relay.close()
    if not rk.main:
        return
    zero = np.zeros((sh, sw, 3), np.uint8)
    # last frame: zero flow -> 0/0 -> NaN -> uint8 0, max displacement 0.0, all-False masks (reference :116-131)
    for v in (fwd_video, bwd_video, fwd_mask_video, bwd_mask_video):
        if v:
            v.write(zero)
    # the reference appends np.float32 scalars and formats them with "{}" (:138-141): '12.148', not the float64 repr
    max_disps = [np.float32(v) for v in np.asarray(mx_all).reshape(-1)] + [np.float32(0.0)]
    zf = np.zeros((sh, sw, 2), np.float32)
    if args.subpath:
        write_flo(os.path.join(args.subpath + "_fwd", "%04d.flo" % (n - 1)), zf)
        if args.backwards:
            write_flo(os.path.join(args.subpath + "_bwd", "%04d.flo" % (n - 1)), zf)
    if args.subpath_mask:
        zm = np.zeros((sh, sw), bool)
        write_flow_png(os.path.join(args.subpath_mask + "_fwd", "%04d.png" % (n - 1)), zf, zm)
        if args.backwards:
            write_flow_png(os.path.join(args.subpath_mask + "_bwd", "%04d.png" % (n - 1)), zf, zm)
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
    global data, ranks
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
    # reference :184: parsed and never read (init_model loads args.model, :40) - accepted so a reference command line runs unchanged
    ap.add_argument("--raft_model", default="models/raft-things.pth", help="[RAFT] restore checkpoint (unused by the reference as well)")
    ap.add_argument("--small", action="store_true", help="use small model")
    ap.add_argument("--mixed_precision", action="store_true", help="use mixed precision")
    ap.add_argument("--alternate_corr", action="store_true", help="use efficent correlation implementation")
    ap.add_argument("--synthetic", action="store_true", help="seeded synthetic weights when the checkpoint is missing (tests / benchmarks)")
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
    ranks = shard.Ranks()
    if ranks.main:
        check_overwrite(args.output)
    input_folder = os.path.dirname(args.input)
    for attr in ("subpath", "subpath_mask"):
        if getattr(args, attr):
            setattr(args, attr, os.path.join(input_folder, getattr(args, attr)))
            os.makedirs(getattr(args, attr) + "_fwd", exist_ok=True)
            if args.backwards:
                os.makedirs(getattr(args, attr) + "_bwd", exist_ok=True)
    init_model(args, device=ranks.device)
    process_video(args)
    if ranks.main:
        write_metadata(meta_path, data)
    ranks.close()


if __name__ == "__main__":
    main()
