#!/usr/bin/env python3
"""process.py - prisma's L5 orchestrator over this repo's band scripts.

Re-statement of /root/reference/process.py: same flags (:76-99), same folder / metadata.json contract (:101-189: folder next to
the input, rgba band first, width / height / fps / frames / duration, principal_point, focal_length = sqrt(W H),
field_of_view), same band order (mask -> depth -> flow -> camera, :205-290), same per-band extra arguments (:47-58:
`--sdf` for mask_mmdet, `--metric outdoor` for depth_anything, `--ply` / `--npy` / `--subpath` from `--extra`), same
default-band aliases (`depth`, `flow`, `flow_bwd`, `flow_mask`, `flow_mask_bwd`, :243-287).  It shells out to
`bands/<band>.py` exactly like the reference's run() (:60-73), with `sys.executable` instead of a bare `python3`.

Bands this repo builds: rgba, depth_anything, flow_raft, flow_gmflow (the reference's default flow band, :23 -
and this script's), mask_mmdet.  The reference's default for still images (depth_patchfusion) and camera_colmap are out of scope
(SURVEY section 2): a request for a band that is not built is reported and skipped; a default that is not built falls back to the
built band of the same kind.
"""
import argparse
import os
import shlex
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from bands.common.io import get_image_size, get_video_data  # noqa: E402
from bands.common.meta import (add_band, create_metadata, is_video, load_metadata, set_default_band,  # noqa: E402
                               write_metadata)

# Default BANDS & MODELS (reference :17-30; defaults narrowed to what is built here)
BUILT = ("rgba", "depth_anything", "flow_raft", "flow_gmflow", "mask_mmdet")
DEPTH_VIDEO_DEFAULT = "depth_anything"
DEPTH_IMAGE_DEFAULT = "depth_anything"          # reference: depth_patchfusion (not built)
DEPTH_BANDS = ["depth_midas", "depth_marigold", "depth_zoedepth", "depth_patchfusion", "depth_anything"]
DEPTH_OPTIONS = DEPTH_BANDS + ["all"]
FLOW_DEFAULT = "flow_gmflow"                    # reference :23
FLOW_BANDS = ["flow_gmflow", "flow_raft"]
FLOW_OPTIONS = FLOW_BANDS + ["all"]
MASK_DEFAULT = "mask_mmdet"

SUBFOLDERS = {"rgba": "images", "mask_mmdet": "mask", "flow_raft": "flow_raft", "flow_gmflow": "flow_gmflow",
              "depth_zoedepth": "depth_zoedepth", "depth_midas": "depth_midas", "depth_marigold": "depth_marigold",
              "depth_patchfusion": "depth_patchfusion", "depth_anything": "depth_anything", "camera_colmap": "sparse"}
EXTRA_ARGS = {"rgba": "", "mask_mmdet": "--sdf ", "depth_midas": " ", "depth_marigold": "", "depth_zoedepth": "",
              "depth_patchfusion": "", "depth_anything": "--metric outdoor ", "flow_raft": "", "flow_gmflow": ""}

COMMANDS = []        # every command run() issued, in order (tests read it)
RESULTS = []         # (band, return code) of every BUILT band run() launched; unbuilt bands are "skipped", not failures


def build_command(band, input_folder, output_file="", subpath=False, extra_args=""):
    """The argv of reference run() (:60-73): bands/<band>.py -i <input> [--output <file>] <extra> [--subpath <SUBFOLDERS[band]>]."""
    cmd = [sys.executable, os.path.join(ROOT, "bands", band + ".py"), "-i", input_folder]
    if output_file != "":
        cmd += ["--output", output_file]
    if extra_args != "":
        cmd += shlex.split(extra_args)
    if subpath:
        cmd += ["--subpath", SUBFOLDERS[band]]
    return cmd


def run(band, input_folder, output_file="", subpath=False, extra_args=""):
    print("\n# ", band.upper())
    if band not in BUILT:
        print(f"band '{band}' is not built in this repo (out of scope, SURVEY section 2): skipped")
        return 1
    cmd = build_command(band, input_folder, output_file, subpath, extra_args)
    COMMANDS.append(cmd)
    print(" ".join(shlex.quote(c) for c in cmd), "\n")
    rc = subprocess.run(cmd, cwd=ROOT).returncode
    RESULTS.append((band, rc))
    return rc


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--input", "-i", help="input file", type=str, required=True)
    parser.add_argument("--output", help="folder name", type=str, default="")
    parser.add_argument("--record3d", help="Record3D video", action="store_true")
    parser.add_argument("--fps", "-r", help="fix framerate", type=float, default=24)
    parser.add_argument("--extra", "-e", help="Save extra data [>0 frames|PLYs; >1 FLOs; >2 NPY]", type=int, default=0)
    parser.add_argument("--rgbd", help="Where the depth is", type=str, default=None)
    parser.add_argument("--depth", "-d", help="Depth bands", type=str, default=None, choices=DEPTH_OPTIONS)
    parser.add_argument("--ply", "-p", help="Save ply for images", action="store_true")
    parser.add_argument("--npy", "-n", help="Save npy version of files", action="store_true")
    parser.add_argument("--flow", "-f", help="Flow bands", type=str, default=None, choices=FLOW_OPTIONS)
    parser.add_argument("--flo", help="Save flo files for raft", action="store_true")
    parser.add_argument("--flow_backwards", "-b", help="Save backwards video", action="store_true")
    parser.add_argument("--flow_mask", "-m", help="Save mask of videos", action="store_true")
    args = parser.parse_args(argv)
    if args.record3d or args.rgbd:
        raise SystemExit("process.py: --record3d / --rgbd (side-by-side RGB-D captures) are not built in this repo")
    del COMMANDS[:]
    del RESULTS[:]

    # 1. input parameters, 2. folder + metadata (reference :101-117)
    input_path = args.input
    input_folder = os.path.dirname(input_path)
    input_basename = os.path.basename(input_path).rsplit(".", 1)[0]
    folder_name = args.output if args.output else os.path.join(input_folder, input_basename)
    data = create_metadata(folder_name)
    video = is_video(input_path)
    extension = input_path.rsplit(".", 1)[1] if video else "png"        # "mp4" in the reference; .npy frame stacks offline
    name_rgba = "rgba." + extension
    path_rgba = os.path.join(folder_name, name_rgba)

    # 3. extract RGBA (reference :160-171)
    add_band(data, "rgba", url=name_rgba)
    extra_rgba_args = EXTRA_ARGS["rgba"]
    if video:
        extra_rgba_args += " --fps " + str(args.fps)
    write_metadata(folder_name, data)
    run("rgba", input_path, path_rgba, subpath=True, extra_args=extra_rgba_args)
    data = load_metadata(folder_name)

    # 4. metadata: sizes and reconstructed intrinsics (reference :174-191)
    if video:
        data["width"], data["height"], data["fps"], data["frames"] = get_video_data(path_rgba)
        data["duration"] = float(data["frames"]) / float(data["fps"])
    else:
        data["width"], data["height"] = get_image_size(path_rgba)
    if "principal_point" not in data:
        data["principal_point"] = [float(data["width"] / 2), float(data["height"] / 2)]
    if "focal_length" not in data:
        data["focal_length"] = float(data["height"] * data["width"]) ** 0.5
    if "field_of_view" not in data:
        data["field_of_view"] = float(2 * np.arctan(0.5 * data["height"] / data["focal_length"]) * 180 / np.pi)
    write_metadata(folder_name, data)

    # 5. bands (reference :196-290)
    if args.extra > 0:
        args.ply = True
    if args.extra > 1:
        args.flo = True
    if args.extra > 2:
        args.npy = True

    run("mask_mmdet", folder_name, subpath=True, extra_args=EXTRA_ARGS["mask_mmdet"])

    depth_args = ""
    if args.ply:
        depth_args = "--ply "
    if args.npy:
        depth_args += "--npy "
    if args.depth is None:
        args.depth = DEPTH_VIDEO_DEFAULT if video else DEPTH_IMAGE_DEFAULT
    for band in (DEPTH_BANDS if args.depth == "all" else [args.depth]):
        extra_args = depth_args + EXTRA_ARGS.get(band, "")
        if band == "depth_patchfusion" and video:
            extra_args += "--mode=p49 "
        run(band, folder_name, subpath=args.extra, extra_args=extra_args)
    set_default_band(folder_name, "depth", (DEPTH_VIDEO_DEFAULT if video else DEPTH_IMAGE_DEFAULT) if args.depth == "all" else args.depth)

    if video:
        if args.flow is None:
            args.flow = FLOW_DEFAULT
        flow_args = ""
        if args.flow_backwards:
            flow_args += "--backwards "
        if args.flow_mask:
            flow_args += "--mask "
        for band in (FLOW_BANDS if args.flow == "all" else [args.flow]):
            run(band, folder_name, subpath=args.flo, extra_args=flow_args + EXTRA_ARGS.get(band, ""))
        fdef = FLOW_DEFAULT if args.flow == "all" else args.flow
        set_default_band(folder_name, "flow", fdef)
        set_default_band(folder_name, "flow_bwd", fdef + "_bwd")
        set_default_band(folder_name, "flow_mask", fdef + "_mask")
        set_default_band(folder_name, "flow_mask_bwd", fdef + "_mask_bwd")
        run("camera_colmap", folder_name, subpath=True)
    # a band that failed (missing checkpoint, bad input ...) leaves a PRISMA folder without its entries: say so and fail the run
    # (the reference's os.system() ignores band failures; a drop-in that now refuses to run without weights must not exit 0 on them)
    failed = [(b, rc) for b, rc in RESULTS if rc != 0]
    if failed:
        print("\nprocess.py: %d band(s) FAILED: %s - %s is incomplete" % (len(failed), ", ".join("%s (exit %d)" % f for f in failed), folder_name),
              file=sys.stderr)
        raise SystemExit(1)
    return folder_name


if __name__ == "__main__":
    main()
