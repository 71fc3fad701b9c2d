"""prisma folder contract: <folder>/metadata.json + one file per band.

Re-statement of the behaviour of /root/reference/bands/common/meta.py (load_metadata :27-32,
create_metadata :35-58, is_video :65-67, get_target :70-93, get_url :96-104, add_band :109-121,
write_metadata :124-134, set_default_band :137-146) - same function names, arguments and JSON
layout, so files written by either implementation are interchangeable.
"""
import json
import os

META_FILE = "metadata.json"


def get_metadata_path(path):
    if os.path.isfile(path):
        return path if path.endswith(".json") else get_metadata_path(os.path.dirname(path))
    if os.path.isdir(path):
        return os.path.join(path, META_FILE)
    return None


def load_metadata(path):
    mp = get_metadata_path(path)
    if mp and os.path.exists(mp):
        with open(mp) as f:
            return json.load(f)
    return None


def create_metadata(path):
    folder = os.path.dirname(path) if os.path.isfile(path) else path
    os.makedirs(folder, exist_ok=True)
    mp = os.path.join(folder, META_FILE)
    if not os.path.exists(mp):
        with open(mp, "w") as f:
            f.write(json.dumps({"bands": {}}, indent=4))
    return load_metadata(mp)


def is_video(path):
    # the reference tests the suffix only (:65-67); .npy frame stacks are this repo's offline stand-in
    return path.endswith(".mp4") or path.endswith(".npy")


def add_band(metadata, band, url="", folder=""):
    b = metadata.setdefault("bands", {}).setdefault(band, {})
    if url != "":
        b["url"] = url
    if folder != "":
        b["folder"] = folder


def get_target(path, metadata, band="rgba", target="", force_extension=None):
    folder = target if os.path.isdir(target) else os.path.dirname(path)
    ext = os.path.basename(path).rsplit(".", 1)[1]
    if force_extension and (not is_video(path) or force_extension == "csv"):
        ext = force_extension
    name = band + "." + ext
    if target == "" or os.path.isdir(target):
        target = os.path.join(folder, name)
    if metadata:
        add_band(metadata, band, url=name)
    return target


def get_url(path, metadata, band):
    if os.path.isdir(path) and metadata:
        url = metadata.get("bands", {}).get(band, {}).get("url")
        if url:
            return os.path.join(path, url)
    return path


def write_metadata(path, metadata):
    if metadata is None:
        return
    mp = get_metadata_path(path)
    if mp and os.path.exists(mp):
        with open(mp, "w") as f:
            f.write(json.dumps(metadata, indent=4))


def set_default_band(path, band, band_default):
    data = load_metadata(path)
    if data and band_default in data.get("bands", {}):
        data["bands"][band] = data["bands"][band_default]
        write_metadata(path, data)
