#!/usr/bin/env python3
"""flow_gmflow band - drop-in for /root/reference/bands/flow_gmflow.py on MI355X (prisma's DEFAULT flow band, process.py:23).

Same CLI (reference :223-255; the GMFlow architecture flags are accepted and must equal the band's defaults, which is the model the
engine builds), same outputs (<BAND>.mp4, <BAND>.csv with the per-frame max displacement, optional <BAND>_bwd / _mask / _mask_bwd
videos, .flo / 16-bit PNG dumps, metadata entries :195-218), same module API (BAND, init_model(), infer()).  The frame loop, the file
writers and the multi-rank relay are flow_raft's (bands/flow_raft.py process_video: the two reference scripts share them line for line,
flow_gmflow.py:121-218 vs flow_raft.py:69-166); the model is libprisma_bands.so's GmflowEngine through prisma_amd.engine.FlowGMFlow.
"""
import argparse
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_ROOT, os.path.dirname(os.path.abspath(__file__))):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import flow_raft as _loop  # noqa: E402  (process_video and its writers; its module globals carry this band's model / metadata)
from common.io import check_overwrite  # noqa: E402
from common.ckpt import load_checkpoint  # noqa: E402
from common.meta import get_target, get_url, is_video, load_metadata, write_metadata  # noqa: E402
from prisma_amd import engine, shard, synth  # noqa: E402

BAND = "flow_gmflow"
MODEL = "models/gmflow_sintel-0c07dcb3.pth"      # reference :35
# the model flags of reference :239-249 and the only values the engine implements (the band's defaults)
ARCH = {"feature_channels": 128, "num_scales": 1, "upsample_factor": 8, "num_head": 1, "attention_type": "swin", "ffn_dim_expansion": 4,
        "num_transformer_layers": 6, "attn_splits_list": [2], "corr_radius_list": [-1], "prop_radius_list": [-1], "padding_factor": 16}

model = None
data = None
ranks = None
_SYNTH = [False]


def load_weights(path):
    """reference :57-61: torch.load(checkpoint)['model'] if present, else the dict itself."""
    if path and os.path.exists(path):
        return load_checkpoint(path, wrappers=("model",))
    if not shard.synthetic_allowed(_SYNTH[0]):
        raise SystemExit(f"[{BAND}] checkpoint {path!r} not found; pass --model, or --synthetic / PRISMA_SYNTH=1 for seeded synthetic weights")
    print(f"[{BAND}] checkpoint {path!r} not found; using seeded synthetic weights (--synthetic)", file=sys.stderr)
    return synth.gmflow_weights(seed=2468)


def check_arch(args):
    bad = {k: getattr(args, k) for k, v in ARCH.items() if hasattr(args, k) and getattr(args, k) != v}
    if bad:
        raise SystemExit(f"[{BAND}] only the band's default GMFlow is built ({ARCH}); got {bad}")
    isz = getattr(args, "inference_size", None)
    if isz and (len(isz) != 2 or any(v < 32 or v % 16 for v in isz)):
        raise SystemExit(f"[{BAND}] --inference_size takes H W, multiples of 16 (the reference's 2 x 2 window split of the 1/8 grid fails otherwise); got {isz}")


def init_model(args=None, device=0):
    global model
    if args is not None:
        check_arch(args)
        _SYNTH[0] = bool(getattr(args, "synthetic", False))
    model = engine.FlowGMFlow(load_weights(getattr(args, "model", MODEL) if args else MODEL), device=device)
    model.set_inference_size(getattr(args, "inference_size", None) if args is not None else None)      # reference :76-100
    return model


def infer(args, image1, image2):
    """Reference signature (:66-118): image1 = prev, image2 = curr as float CHW 0..255 tensors (already scaled).
    Returns (fwd, bwd | None, fwd_mask | None, bwd_mask | None), flows as float32 [H', W', 2]."""
    if model is None:
        init_model(args)
    a = np.ascontiguousarray(np.asarray(image1).transpose(1, 2, 0)).astype(np.uint8)
    b = np.ascontiguousarray(np.asarray(image2).transpose(1, 2, 0)).astype(np.uint8)
    want_mask = bool(getattr(args, "output_mask", "") or getattr(args, "subpath_mask", ""))
    if want_mask:
        flow, _, _, mask = model.infer_sequence_masks(np.stack([a, b]), scale=1.0, want_rgb=False)
        return flow[0, 0], flow[0, 1], mask[0, 0], mask[0, 1]
    both = bool(getattr(args, "backwards", False))
    flow, _, _ = model.infer_sequence(np.stack([a, b]), scale=1.0, backward=both, want_rgb=False)
    return flow[0, 0], (flow[0, 1] if both else None), None, None


def process_video(args):
    """The shared flow loop with this band's name, model, metadata and ranks."""
    saved = (_loop.BAND, _loop.model, _loop.data, _loop.ranks, _loop.SUBPATH_NEEDS_BOTH)
    _loop.BAND, _loop.model, _loop.data, _loop.ranks, _loop.SUBPATH_NEEDS_BOTH = BAND, model, data, ranks, False
    args.iterations = 1                    # GMFlow is not iterative; the loop passes it through to the engine, which ignores it
    try:
        _loop.process_video(args)
    finally:                               # a process may run both flow bands (tests do): leave flow_raft's globals as they were
        _loop.BAND, _loop.model, _loop.data, _loop.ranks, _loop.SUBPATH_NEEDS_BOTH = saved


def main(argv=None):
    global data, ranks
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", "-i", help="input", type=str, required=True)
    ap.add_argument("--output", "-o", help="output", type=str, default="")
    ap.add_argument("--subpath", help="path to flo files", type=str, default="")
    ap.add_argument("--backwards", "-b", help="Backward video", action="store_true")
    ap.add_argument("--mask", action="store_true", help="Compute mask as well")
    ap.add_argument("--output_mask", help="output dense", type=str, default="")
    ap.add_argument("--subpath_mask", help="path to flo files", type=str, default="")
    ap.add_argument("--scale", type=float, default=0.75)
    ap.add_argument("--model", "-m", help="model path", type=str, default=MODEL)
    ap.add_argument("--feature_channels", default=128, type=int)
    ap.add_argument("--num_scales", default=1, type=int)
    ap.add_argument("--upsample_factor", default=8, type=int)
    ap.add_argument("--num_head", default=1, type=int)
    ap.add_argument("--attention_type", default="swin", type=str)
    ap.add_argument("--ffn_dim_expansion", default=4, type=int)
    ap.add_argument("--num_transformer_layers", default=6, type=int)
    ap.add_argument("--attn_splits_list", default=[2], type=int, nargs="+")
    ap.add_argument("--corr_radius_list", default=[-1], type=int, nargs="+")
    ap.add_argument("--prop_radius_list", default=[-1], type=int, nargs="+")
    ap.add_argument("--strict_resume", action="store_true")
    ap.add_argument("--inference_size", default=None, type=int, nargs="+")
    ap.add_argument("--padding_factor", default=16, type=int)
    ap.add_argument("--local_rank", default=0, type=int)
    ap.add_argument("--synthetic", action="store_true", help="seeded synthetic weights when the checkpoint is missing (tests / benchmarks)")
    args = ap.parse_args(argv)
    check_arch(args)
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
