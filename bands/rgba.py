#!/usr/bin/env python3
"""rgba band: bring the input into the prisma folder as rgba.png / rgba.mp4 (what every other band reads).

Re-statement of the plain path of /root/reference/bands/rgba.py (process_image :104-110: open_float_rgb -> write_rgb as PNG;
process_video -> prune :77-101: frame pass-through that drops audio, optional --subpath frame dump `255 - frame`); same
flags for that path.  The Record3D / side-by-side RGB-D split (--rgbd left|right|top|bottom, --encoding_depth) is not
built: it only serves prisma's capture formats, not the three bands of SURVEY section 8.  No model, no GPU.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common.io import FrameReader, VideoWriter, check_overwrite, create_folder, open_rgb, write_rgb  # noqa: E402
from common.meta import get_target, is_video, load_metadata, write_metadata  # noqa: E402

BAND = "rgba"


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--input", "-i", type=str, required=True)
    p.add_argument("--tmp", "-t", type=str, default="tmp")
    p.add_argument("--fps", "-r", type=float, default=24)
    p.add_argument("--output", "-o", type=str, default="")
    p.add_argument("--subpath", type=str, default=None)
    p.add_argument("--rgbd", choices=["none", "left", "right", "top", "bottom"], default="none")
    p.add_argument("--encoding_depth", choices=["none", "hue"], default="none")
    p.add_argument("--output_depth", type=str, default="depth")
    p.add_argument("--subpath_depth", type=str, default=None)
    args = p.parse_args(argv)
    if args.rgbd != "none":
        raise SystemExit("rgba: the RGB-D split (--rgbd) is not built in this repo (SURVEY section 2: capture formats are out of scope)")
    ext = args.input.rsplit(".", 1)[1]
    if not is_video(args.input):
        ext = "png"
    if args.output == "" or os.path.isdir(args.output):
        folder = args.output if args.output else os.path.dirname(args.input)
        args.output = os.path.join(folder, BAND + "." + ext)
    check_overwrite(args.output)
    if is_video(args.input):
        src = FrameReader(args.input)
        sub = None
        if args.subpath:
            sub = os.path.join(os.path.dirname(args.output), args.subpath)
            create_folder(sub)
        h, w = src[0].shape[:2]
        out = VideoWriter(width=w, height=h, frame_rate=args.fps, filename=args.output)
        for i in range(len(src)):
            f = src[i]
            if sub:
                write_rgb(os.path.join(sub, str(i).zfill(6) + ".png"), (255 - f).astype(np.uint8))     # rgba.py:96
            out.write(f)
        out.close()
    else:
        write_rgb(args.output, open_rgb(args.input))
    data = load_metadata(os.path.dirname(args.output))
    if data is not None:
        get_target(args.output, data, band=BAND, target=args.output)
        write_metadata(os.path.dirname(args.output), data)


if __name__ == "__main__":
    main()
