# bands/depth_anything.py  (reference lines 48-76 and 100-143 replaced; self-contained: ctypes + numpy only)
import ctypes as C, os, numpy as np

_lib = C.CDLL(os.environ.get("PRISMA_BANDS_LIB", "libprisma_bands.so"))
_lib.pb_last_error.restype = C.c_char_p

class pb_tensor(C.Structure):                      # include/prisma_bands.h pb_tensor
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 6), ("data", C.c_void_p)]

class pb_depth_cfg(C.Structure):                   # include/prisma_bands.h pb_depth_cfg
    _fields_ = [("embed_dim", C.c_int32), ("depth", C.c_int32), ("heads", C.c_int32),
                ("features", C.c_int32), ("out_channels", C.c_int32 * 4),
                ("pos_grid", C.c_int32), ("max_batch", C.c_int32), ("metric", C.c_int32),
                ("precision", C.c_int32)]

# a stale binding fails here, not by reading shifted struct fields (include/prisma_bands.h PB_ABI_VERSION, pb_struct_size)
assert _lib.pb_abi_version() == 3 and _lib.pb_struct_size(0) == C.sizeof(pb_tensor) and _lib.pb_struct_size(1) == C.sizeof(pb_depth_cfg)

_CFG = {"vits": (384, 12, 6, 64, (48, 96, 192, 384)),
        "vitb": (768, 12, 12, 128, (96, 192, 384, 768)),
        "vitl": (1024, 24, 16, 256, (256, 512, 1024, 1024))}
BATCH = 1
model = None

def init_model(sd, encoder="vitl"):     # was: DepthAnything.from_pretrained(...).to(DEVICE).eval()
    """sd: the same checkpoint as {name: float32 ndarray} (reference key names, `pretrained.*`, `depth_head.*`)."""
    global model
    e = _CFG[encoder]
    cfg = pb_depth_cfg(e[0], e[1], e[2], e[3], (C.c_int32 * 4)(*e[4]), 37, BATCH, 0, 1)   # relative model, split-fp16
    arr, keep = (pb_tensor * len(sd))(), []
    for i, (k, v) in enumerate(sd.items()):
        v = np.ascontiguousarray(v, np.float32); keep.append(v)
        arr[i].name, arr[i].dtype, arr[i].ndim, arr[i].data = k.encode(), 0, v.ndim, v.ctypes.data
        for j, s in enumerate(v.shape): arr[i].shape[j] = s
    model = C.c_void_p()
    rc = _lib.pb_create(C.byref(model), 0, b"depth_anything", arr, len(sd), C.byref(cfg), C.c_size_t(C.sizeof(cfg)))
    if rc: raise RuntimeError(_lib.pb_last_error().decode())
    return model

def infer(img, normalize=False):        # img: uint8 HxWx3 RGB from decord / open_rgb
    h, w = img.shape[:2]
    depth = np.empty((h, w), np.float32)
    frame = np.ascontiguousarray(img, np.uint8)
    rc = _lib.pb_depth_infer_batch(model, frame.ctypes.data_as(C.c_void_p), 1, h, w,
                                   depth.ctypes.data_as(C.c_void_p), None, None, None, 1)
    if rc: raise RuntimeError(_lib.pb_last_error().decode())
    if normalize:                       # unchanged numpy (reference :135-141)
        lo, hi = depth.min(), depth.max()
        if hi - lo > np.finfo("float").eps: depth = (depth - lo) / (hi - lo)
    return depth
