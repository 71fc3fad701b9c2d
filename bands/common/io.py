"""Frame sources / sinks of the band scripts (host side, stays Python like the reference).

Re-states the file behaviour of /root/reference/bands/common/io.py: open_rgb (:78-83),
check_overwrite (:35-41), VideoWriter (:246-305, libx264 crf 15 yuv420p), write_depth (:138-172).
decord / PyAV / OpenCV are optional here (absent in the build image): .mp4 needs them, .npy
frame stacks ([n,H,W,3] uint8) and .png work everywhere.
"""
import os

import numpy as np


def create_folder(path):
    os.makedirs(path, exist_ok=True)


def check_overwrite(path, assume_yes=None):
    """Reference asks interactively (io.py:35-41); PRISMA_OVERWRITE=1 or a non-tty answers yes."""
    if not path or not os.path.exists(path):
        return
    if assume_yes is None:
        assume_yes = os.environ.get("PRISMA_OVERWRITE", "") == "1" or not os.isatty(0)
    if not assume_yes and input(f"File {path} already exists. Overwrite? [y/N] ").lower() != "y":
        raise SystemExit(0)


def open_rgb(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def write_rgb(path, rgb_u8):
    from PIL import Image
    Image.fromarray(rgb_u8).save(path)


def get_image_size(path):
    """(width, height) of an image file (io.py:57-60)."""
    from PIL import Image
    with Image.open(path) as im:
        return im.size[0], im.size[1]


def get_video_data(path):
    """(width, height, fps, total_frames) of a video (io.py:63-67)."""
    v = FrameReader(path)
    f0 = v[0]
    return f0.shape[1], f0.shape[0], v.fps, len(v)


class FrameReader:
    """len() / [i] -> uint8 HxWx3 RGB; fps.  decord for .mp4 (reference), numpy memmap for .npy."""

    def __init__(self, path):
        self.fps = 24.0
        if path.endswith(".npy"):
            self._a = np.load(path, mmap_mode="r")
            assert self._a.ndim == 4 and self._a.shape[-1] == 3 and self._a.dtype == np.uint8
            self._get = lambda i: np.asarray(self._a[i])
            self._n = self._a.shape[0]
        else:
            try:
                import decord
            except ImportError as e:
                raise RuntimeError("reading .mp4 needs decord (reference dependency); use a .npy frame stack here") from e
            self._v = decord.VideoReader(path)
            self.fps = self._v.get_avg_fps()
            self._get = lambda i: self._v[i].asnumpy()
            self._n = len(self._v)

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        return self._get(i)


class VideoWriter:
    """write(uint8 HxWx3) / close().  .mp4 -> PyAV libx264 crf 15 (io.py:246-305); .npy -> frame stack."""

    def __init__(self, width, height, frame_rate, filename):
        self.filename, self._frames, self._av = filename, None, None
        if filename.endswith(".npy"):
            self._frames = []
        else:
            try:
                import av
            except ImportError as e:
                raise RuntimeError("writing .mp4 needs PyAV (reference dependency); write a .npy stack here") from e
            # reference io.py:262-289: sides clamped to 3840 keeping the aspect, made even; rate as '%.2f' (29.97 stays 29.97, so the
            # band videos stay frame-aligned with rgba.mp4); libx264 crf 15, frame threading AUTO
            max_size = 3840
            if width > max_size or height > max_size:
                aspect = height / width
                width, height = (max_size, round(max_size * aspect)) if aspect < 1 else (round(max_size / aspect), max_size)
            self._av = av.open(filename, mode="w")
            self._st = self._av.add_stream("libx264", rate="%.2f" % frame_rate, options={"crf": "15"})
            self._w, self._h = 2 * round(width / 2), 2 * round(height / 2)
            self._st.width, self._st.height, self._st.pix_fmt = self._w, self._h, "yuv420p"
            self._st.thread_type = "AUTO"

    def write(self, rgb):
        if self._frames is not None:
            self._frames.append(np.ascontiguousarray(rgb))
            return
        import av
        frame = av.VideoFrame.from_ndarray(np.ascontiguousarray(rgb, np.uint8), format="rgb24")
        for pkt in self._st.encode(frame.reformat(width=self._w, height=self._h)):     # io.py:297: frames may be scaled
            self._av.mux(pkt)

    def close(self):
        if self._frames is not None:
            np.save(self.filename, np.stack(self._frames) if self._frames else np.zeros((0, 0, 0, 3), np.uint8))
            return
        for pkt in self._st.encode():
            self._av.mux(pkt)
        self._av.close()


def _sobel_mag_u8(img_u8):
    """cv2.Sobel(img, CV_64F, 1, 0, ksize=1) / (0, 1): central differences, reflect-101 border.
    PARITY UNPINNED (third-party opencv-python absent); call site encode.py:81-95."""
    a = img_u8.astype(np.float64)
    p = np.pad(a, 1, mode="reflect")
    gx = p[1:-1, 2:] - p[1:-1, :-2]
    gy = p[2:, 1:-1] - p[:-2, 1:-1]
    return np.sqrt(gx * gx + gy * gy)


def float_to_rgb(value, min_value=0.0, max_value=1.0, base=256):
    """encode.py:141-146: 24-bit fixed point of a scalar in three channels."""
    L = np.clip((value - min_value) / (max_value - min_value), 0.0, 1.0) * (base ** 3 - 1)
    return (np.floor(L % base) / (base - 1), np.floor(L / base) % base / (base - 1),
            np.floor(L / (base * base)) % base / (base - 1))


def write_depth(path, depth, heat_rgb_fn, normalize=True, flip=False, heatmap=True, encode_range=True):
    """Still-image encode (io.py:138-172): heat ramp, Sobel edge in the saturation, min/max packed in
    pixels (0,0) and (0,1).  heat_rgb_fn(float64 HxW in 0..1) -> float64 HxWx3 (the band's ramp)."""
    dmin, dmax = depth.min(), depth.max()
    if normalize:
        depth = (depth - dmin) / (dmax - dmin)
    if flip:
        depth = 1.0 - depth
    if not heatmap:
        from PIL import Image
        Image.fromarray((depth * 65535).astype("uint16")).save(path)
        return
    mag = _sobel_mag_u8((depth * 255).astype(np.uint8))
    edge = mag * (255.0 / mag.max()) / 255.0 if mag.max() > 0 else mag
    rgb = heat_rgb_fn(depth.astype(np.float64))
    sat = (1.0 - edge)[..., None]
    rgb = rgb * sat + (1.0 - sat)
    if encode_range:
        rgb[0, 0] = float_to_rgb(dmin, 0.0, 1000.0)
        rgb[0, 1] = float_to_rgb(dmax, 0.0, 1000.0)
    write_rgb(path, (rgb * 255).astype(np.uint8))


def write_flo(path, flow):
    """Middlebury .flo (io.py:175-197): float32 magic 202021.25, int32 width, int32 height, float32 HxWx2."""
    flow = np.ascontiguousarray(flow, np.float32)
    with open(path, "wb") as f:
        np.array([202021.25], np.float32).tofile(f)
        np.array([flow.shape[1], flow.shape[0]], np.int32).tofile(f)
        flow.tofile(f)


def encode_flow(flow, mask):
    """encode.py:105-110: 8.8 fixed point around 2^15 in uint16, validity in the third channel."""
    q = np.float32(2 ** 15) + flow.astype(np.float32) * np.float32(2 ** 8)
    ok = mask.astype(bool) & (q.max(axis=-1) < (2 ** 16 - 1)) & (0 < q.min(axis=-1))
    return np.concatenate([q.astype(np.uint16), ok[..., None].astype(np.uint16) * (2 ** 16 - 1)], axis=-1)


def write_png16(path, img_u16):
    """16-bit RGB PNG, channel order as given (PIL cannot write this mode; cv2 is absent)."""
    import struct
    import zlib
    a = np.ascontiguousarray(img_u16, np.uint16)
    h, w, c = a.shape
    assert c == 3
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.astype(">u2").view(np.uint8).reshape(h, w * 6)], axis=1).tobytes()

    def chunk(tag, payload):
        return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_flow_png(path, flow, mask):
    """cv2.imwrite(path, encode_flow(flow, mask)) (common/flow.py:96-99): OpenCV stores the array as B, G, R, so
    the file's R channel is the validity mask, G is v and B is u."""
    write_png16(path, encode_flow(flow, mask)[..., ::-1])
