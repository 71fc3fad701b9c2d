"""Python host side of the bands engine: thin, typed wrappers over the C ABI.

`DepthAnything` mirrors what bands/depth_anything.py builds in init_model() and calls in
infer() (reference lines 48-76, 100-143); all arithmetic runs in libprisma_bands.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import check
from .synth import DEPTH_CFGS, MASK_CFGS, DepthCfg, MaskCfg


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def device_count() -> int:
    return _lib.load().pb_device_count()


def net_size(H: int, W: int) -> Tuple[int, int]:
    """(net_h, net_w) of the network input for an HxW frame (transform.py:100-166)."""
    nh, nw = C.c_int(), C.c_int()
    check(_lib.load().pb_depth_net_size(H, W, C.byref(nh), C.byref(nw)))
    return nh.value, nw.value


def run_concurrently(jobs):
    """Drive several band contexts at once: jobs = [(ctx, enqueue), ...] where `enqueue()` calls one of ctx's asynchronous device-pointer
    entry points (`infer_dev`, `infer_sequence_dev`, `infer_batch_dev`).  Every job is enqueued on its own ctx stream from this thread
    (an enqueue returns in a few milliseconds; the GPU then shares its CUs between the streams), then each ctx is waited for in order.
    Returns the seconds from the first enqueue to each ctx's completion.  Results are the bytes of running the jobs one after the other:
    contexts share no buffers (tests/test_gpu_edges.py::test_bands_run_concurrently_equal_sequential).  This replaces the reference's
    strictly sequential band order (process.py:205-290) where the bands of one video are independent."""
    import time
    t0 = time.perf_counter()
    started = []
    try:
        for ctx, enqueue in jobs:
            started.append(ctx)
            enqueue()
    except BaseException:
        for ctx in started:                 # an enqueue failed: nothing may stay in flight on the contexts that did start
            try:
                ctx.sync()
            except Exception:               # noqa: BLE001 - the first error is the one to report
                pass
        raise
    done = []
    for ctx, _ in jobs:
        ctx.sync()
        done.append(time.perf_counter() - t0)
    return done


class _Ctx:
    def __init__(self):
        self.lib = _lib.load()
        self.ctx = C.c_void_p()

    def close(self):
        if self.ctx:
            self.lib.pb_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # RCCL all-gather of per-frame scalars through the C ABI (pb_comm_*, pb_gather_scalars)
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(_lib.load().pb_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, comm_id: bytes, rank: int, world: int):
        assert len(comm_id) == 128
        check(self.lib.pb_comm_init(self.ctx, C.create_string_buffer(comm_id, 128), rank, world))
        self._comm_world = world

    def gather_scalars(self, local: np.ndarray) -> np.ndarray:
        """float32 [n_local, ...] of this rank -> [world, n_local, ...] on every rank (same n_local everywhere)."""
        local = _f32(local)
        out = np.empty((self._comm_world,) + local.shape, np.float32)
        check(self.lib.pb_gather_scalars(self.ctx, _ptr(local), local.size, _ptr(out)))
        return out

    # device memory helpers (frames resident in HBM for bench / pipelines)
    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        check(self.lib.pb_dev_alloc(self.ctx, C.byref(p), nbytes))
        return p.value

    def dev_free(self, ptr: int):
        check(self.lib.pb_dev_free(self.ctx, C.c_void_p(ptr)))

    def h2d(self, ptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        check(self.lib.pb_memcpy_h2d(self.ctx, C.c_void_p(ptr), _ptr(arr), arr.nbytes))

    def d2h(self, arr: np.ndarray, ptr: int):
        assert arr.flags.c_contiguous
        check(self.lib.pb_memcpy_d2h(self.ctx, _ptr(arr), C.c_void_p(ptr), arr.nbytes))

    def sync(self):
        check(self.lib.pb_sync(self.ctx))

    def wait(self):
        """Blocks until the OLDEST submit_* call of this context not yet waited for has its results in the caller's (page-locked) arrays
        and returns what that call returned (pb_wait)."""
        check(self.lib.pb_wait(self.ctx))
        return self._inflight.pop(0)[0] if getattr(self, "_inflight", None) else None

    def _hold(self, result, *arrays):
        # the arrays of a submission must outlive it: keep references until wait() hands the results back
        if not hasattr(self, "_inflight"):
            self._inflight = []
        self._inflight.append((result, arrays))

    def encode_still(self, depth: np.ndarray, flip: bool = True, encode_range: bool = True):
        """write_depth(heatmap=True) of one float32 depth map on the GPU (bands/common/io.py:138-172): -> (rgb u8 [H,W,3], min, max)."""
        depth = _f32(depth)
        H, W = depth.shape
        rgb = np.empty((H, W, 3), np.uint8)
        lo, hi = C.c_float(), C.c_float()
        check(self.lib.pb_depth_encode_still(self.ctx, _ptr(depth), H, W, int(flip), int(encode_range), _ptr(rgb), C.byref(lo), C.byref(hi)))
        return rgb, lo.value, hi.value

    def set_option(self, key: str, value: int):
        check(self.lib.pb_set_option(self.ctx, key.encode(), int(value)))


    def set_profiling(self, timing: bool = True, debug_stages: bool = False, accumulate: bool = False):
        """per-launch HIP-event timing (read with kernel_stats), debug stage snapshots, and `accumulate`: keep the records of
        earlier infer calls so that a timed loop can be queried once after it ends"""
        check(self.lib.pb_set_profiling(self.ctx, (1 if timing else 0) | (2 if debug_stages else 0) | (4 if accumulate else 0)))

    def kernel_stats(self) -> List[dict]:
        arr = (_lib.pb_kernel_stat * 64)()
        n = check(self.lib.pb_get_kernel_stats(self.ctx, arr, 64))
        return [dict(name=arr[i].name.decode(), ms=arr[i].ms, flops=arr[i].flops, exec_flops=arr[i].exec_flops, bytes=arr[i].bytes,
                     launches=arr[i].launches) for i in range(n)]


class Ops(_Ctx):
    """Single-kernel entry points (pb_op_*) used by the parity tests."""

    def __init__(self, device: int = 0):
        super().__init__()
        check(self.lib.pb_create(C.byref(self.ctx), device, b"ops", None, 0, None, 0))

    def gemm(self, A, W, bias=None, act: int = 0, tile: int = 0) -> np.ndarray:
        A, W = _f32(A), _f32(W)
        M, K = A.shape
        N = W.shape[0]
        out = np.empty((M, N), np.float32)
        b = None if bias is None else _f32(bias)
        check(self.lib.pb_op_gemm(self.ctx, _ptr(A), _ptr(W), _ptr(b), _ptr(out), M, N, K, act, tile))
        return out

    def corr_volume(self, A, W, ldo: int = 0, guard_rows: int = 32) -> np.ndarray:
        """the flow band's all-pairs correlation kernel alone: [M + guard_rows, ldo] floats, NaN wherever the kernel must not write."""
        A, W = _f32(A), _f32(W)
        M, N = A.shape[0], W.shape[0]
        ldo = ldo or N
        out = np.empty((M + guard_rows, ldo), np.float32)
        check(self.lib.pb_op_corr_volume(self.ctx, _ptr(A), M, _ptr(W), N, ldo, guard_rows, _ptr(out)))
        return out

    def gemm_bench(self, M: int, N: int, K: int, tile: int = 0, epi: int = 0, iters: int = 20) -> float:
        """mean milliseconds per launch on device-resident random data."""
        ms = C.c_double()
        check(self.lib.pb_op_gemm_bench(self.ctx, M, N, K, tile, epi, iters, C.byref(ms)))
        return ms.value

    def attention_bench(self, B: int, heads: int, N: int, variant: int = 0, iters: int = 10) -> float:
        """mean milliseconds per launch of the fused attention kernel on device-resident random Q, K, V."""
        ms = C.c_double()
        check(self.lib.pb_op_attention_bench(self.ctx, B, heads, N, variant, iters, C.byref(ms)))
        return ms.value

    def layernorm(self, x, g, b) -> np.ndarray:
        x, g, b = _f32(x), _f32(g), _f32(b)
        out = np.empty_like(x)
        check(self.lib.pb_op_layernorm(self.ctx, _ptr(x), _ptr(g), _ptr(b), _ptr(out), x.shape[0], x.shape[1]))
        return out

    def attention128(self, q, k, v, region=None) -> np.ndarray:
        """softmax(q k^T / sqrt(128) + mask) v for q, k, v [B, L, 128] (one head); region [B, L] int8: keys of another region get -100"""
        q, k, v = _f32(q), _f32(k), _f32(v)
        B, L, D = q.shape
        assert D == 128 and k.shape == q.shape and v.shape == q.shape
        out = np.empty_like(q)
        rg = None if region is None else np.ascontiguousarray(region, np.int8)
        check(self.lib.pb_op_attention128(self.ctx, _ptr(q), _ptr(k), _ptr(v), _ptr(rg), _ptr(out), B, L))
        return out

    def attention128_split(self, q, k, v, region=None, kxor: int = 0) -> np.ndarray:
        """the same in the flow_gmflow band's split precision; v [B, L, 128 or 32]; region [nreg, L] int8; keys / values of b are those of b ^ kxor"""
        q, k, v = _f32(q), _f32(k), _f32(v)
        B, L, D = q.shape
        vc = v.shape[2]
        assert D == 128 and k.shape == q.shape and v.shape[:2] == (B, L) and vc in (32, 128)
        out = np.empty((B, L, vc), np.float32)
        rg = None if region is None else np.ascontiguousarray(region, np.int8)
        check(self.lib.pb_op_attention128_split(self.ctx, _ptr(q), _ptr(k), _ptr(v), _ptr(rg), 0 if rg is None else rg.shape[0], _ptr(out),
                                                B, L, vc, kxor))
        return out

    def attention(self, q, k, v) -> np.ndarray:
        q, k, v = _f32(q), _f32(k), _f32(v)
        B, Hh, N, d = q.shape
        assert d == 64
        out = np.empty_like(q)
        check(self.lib.pb_op_attention(self.ctx, _ptr(q), _ptr(k), _ptr(v), _ptr(out), B, Hh, N))
        return out

    def conv2d(self, x, w, bias=None, stride=1, pad=None, relu_in=False, relu_out=False) -> np.ndarray:
        x, w = _f32(x), _f32(w)
        B, Ci, H, W = x.shape
        Co, _, ks, _ = w.shape
        pad = ks // 2 if pad is None else pad
        OH, OW = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        out = np.empty((B, Co, OH, OW), np.float32)
        b = None if bias is None else _f32(bias)
        check(self.lib.pb_op_conv2d(self.ctx, _ptr(x), _ptr(w), _ptr(b), _ptr(out), B, Ci, H, W, Co, ks, stride, pad,
                                    int(relu_in), int(relu_out)))
        return out

    def bilinear(self, x, OH: int, OW: int, align_corners: bool) -> np.ndarray:
        x = _f32(x)
        B, Cc, H, W = x.shape
        out = np.empty((B, Cc, OH, OW), np.float32)
        check(self.lib.pb_op_bilinear(self.ctx, _ptr(x), _ptr(out), B, Cc, H, W, OH, OW, int(align_corners)))
        return out

    def preprocess(self, frame: np.ndarray) -> np.ndarray:
        frame = np.ascontiguousarray(frame, np.uint8)
        H, W = frame.shape[:2]
        nh, nw = net_size(H, W)
        out = np.empty((3, nh, nw), np.float32)
        check(self.lib.pb_op_preprocess(self.ctx, _ptr(frame), H, W, _ptr(out), nh, nw))
        return out

    def encode_depth(self, depth, flip: bool = True):
        depth = _f32(depth)
        if depth.ndim == 2:
            depth = depth[None]
        n, H, W = depth.shape
        rgb = np.empty((n, H, W, 3), np.uint8)
        mn, mx = np.empty(n, np.float32), np.empty(n, np.float32)
        check(self.lib.pb_op_encode_depth(self.ctx, _ptr(depth), n, H, W, int(flip), _ptr(rgb), _ptr(mn), _ptr(mx)))
        return rgb, mn, mx


class DepthAnything(_Ctx):
    """Depth-Anything (DINOv2 ViT + DPT) band on one GPU.

    weights: reference state_dict naming -> float32 ndarray (bands/d_anything/dpt.py:139-171).
    """

    def __init__(self, weights: Dict[str, np.ndarray], cfg: DepthCfg | str = "vitl", device: int = 0,
                 max_batch: int = 1, metric: bool = False, precision: Optional[int] = None):
        super().__init__()
        self.cfg = DEPTH_CFGS[cfg] if isinstance(cfg, str) else cfg
        self.metric = bool(metric)
        self.precision = _lib.default_precision() if precision is None else int(precision)
        if self.metric:      # ZoeDepth state dict: the core's tensors sit under `core.core.`
            weights = {(k[len("core.core."):] if k.startswith("core.core.") else k): v for k, v in weights.items()}
        c = _lib.pb_depth_cfg(self.cfg.embed_dim, self.cfg.depth, self.cfg.heads, self.cfg.features,
                              (C.c_int32 * 4)(*self.cfg.out_channels), self.cfg.pos_grid, max_batch, int(self.metric),
                              self.precision)
        keep: List[np.ndarray] = []
        arr = (_lib.pb_tensor * len(weights))()
        for i, (name, w) in enumerate(weights.items()):
            w = _f32(w)
            keep.append(w)
            arr[i].name = name.encode()
            arr[i].dtype = 0
            arr[i].ndim = w.ndim
            for j, s in enumerate(w.shape):
                arr[i].shape[j] = s
            arr[i].data = w.ctypes.data
        check(self.lib.pb_create(C.byref(self.ctx), device, b"depth_anything", arr, len(weights), C.byref(c),
                                 C.sizeof(c)))
        self.max_batch = max_batch

    def infer_batch(self, frames: np.ndarray, want_depth: bool = True, want_rgb: bool = True, flip: bool = True,
                    out_depth: Optional[np.ndarray] = None, out_rgb: Optional[np.ndarray] = None):
        """frames uint8 [n,H,W,3] RGB -> (depth f32 [n,H,W] | None, rgb u8 [n,H,W,3] | None, min [n], max [n]).
        out_depth / out_rgb: caller-owned result arrays (e.g. views of page-locked memory, which the library's copy engines then
        write directly - no staging copy)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        n, H, W, ch = frames.shape
        assert ch == 3
        depth = (out_depth if out_depth is not None else np.empty((n, H, W), np.float32)) if want_depth else None
        rgb = (out_rgb if out_rgb is not None else np.empty((n, H, W, 3), np.uint8)) if want_rgb else None
        assert depth is None or (depth.dtype == np.float32 and depth.shape == (n, H, W) and depth.flags.c_contiguous)
        assert rgb is None or (rgb.dtype == np.uint8 and rgb.shape == (n, H, W, 3) and rgb.flags.c_contiguous)
        mn, mx = np.empty(n, np.float32), np.empty(n, np.float32)
        check(self.lib.pb_depth_infer_batch(self.ctx, _ptr(frames), n, H, W, _ptr(depth), _ptr(rgb), _ptr(mn),
                                            _ptr(mx), int(flip)))
        return depth, rgb, mn, mx

    def submit_batch(self, frames: np.ndarray, out_rgb: Optional[np.ndarray] = None, out_depth: Optional[np.ndarray] = None,
                     out_min: Optional[np.ndarray] = None, out_max: Optional[np.ndarray] = None, flip: bool = True):
        """Asynchronous infer_batch (pb_depth_submit_batch): every array - `frames` and each given output - must be a C-contiguous view of
        PAGE-LOCKED memory (torch `pin_memory()` + `.numpy()`); returns after the enqueue, `wait()` returns (depth, rgb, min, max) once they are
        filled.  Up to two submissions may be in flight per context: the second one's uploads run under the first one's kernels."""
        n, H, W, ch = frames.shape
        assert ch == 3 and frames.dtype == np.uint8 and frames.flags.c_contiguous
        for a, dt, shp in ((out_rgb, np.uint8, (n, H, W, 3)), (out_depth, np.float32, (n, H, W)), (out_min, np.float32, (n,)), (out_max, np.float32, (n,))):
            assert a is None or (a.dtype == dt and a.shape == shp and a.flags.c_contiguous)
        check(self.lib.pb_depth_submit_batch(self.ctx, _ptr(frames), n, H, W, _ptr(out_depth), _ptr(out_rgb), _ptr(out_min), _ptr(out_max), int(flip)))
        self._hold((out_depth, out_rgb, out_min, out_max), frames, out_depth, out_rgb, out_min, out_max)

    def infer(self, img: np.ndarray, normalize: bool = False) -> np.ndarray:
        """bands/depth_anything.py:100-143 `infer(img, normalize)` for the relative model."""
        d = self.infer_batch(img[None], want_depth=True, want_rgb=False)[0][0]
        if normalize:
            lo, hi = d.min(), d.max()
            if hi - lo > np.finfo("float").eps:
                d = (d - lo) / (hi - lo)
        return d

    def infer_dev(self, frames_ptr: int, n: int, H: int, W: int, depth_ptr: int = 0, rgb_ptr: int = 0,
                  min_ptr: int = 0, max_ptr: int = 0, flip: bool = True):
        """All pointers are device addresses; asynchronous on the ctx stream (call sync())."""
        v = lambda p: C.c_void_p(p) if p else None
        check(self.lib.pb_depth_infer_batch_dev(self.ctx, v(frames_ptr), n, H, W, v(depth_ptr), v(rgb_ptr),
                                                v(min_ptr), v(max_ptr), int(flip)))

    def stage(self, name: str, cap: int = 1 << 26) -> np.ndarray:
        out = np.empty(cap, np.float32)
        shape = (C.c_int64 * 4)()
        n = check(self.lib.pb_depth_get_stage(self.ctx, name.encode(), _ptr(out), cap, shape))
        dims = [int(s) for s in shape]
        while len(dims) > 1 and dims[-1] == 1:
            dims.pop()
        return out[:n].reshape(dims).copy()


def flow_out_size(H: int, W: int, scale: float) -> Tuple[int, int]:
    sh, sw = C.c_int(), C.c_int()
    check(_lib.load().pb_flow_out_size(H, W, C.c_float(scale), C.byref(sh), C.byref(sw)))
    return sh.value, sw.value


class FlowRaft(_Ctx):
    """RAFT optical-flow band on one GPU (bands/flow_raft.py:38-66 init_model / infer).

    weights: reference checkpoint naming without the `module.` prefix (fnet.*, cnet.*, update_block.*).
    """

    BAND = b"flow_raft"

    def __init__(self, weights: Dict[str, np.ndarray], device: int = 0, precision: Optional[int] = None):
        super().__init__()
        self.precision = _lib.default_precision() if precision is None else int(precision)
        fc = _lib.pb_flow_cfg(self.precision)
        keep = {k: _f32(v) for k, v in weights.items() if np.asarray(v).dtype.kind == "f"}
        arr = (_lib.pb_tensor * len(keep))()
        for i, (name, w) in enumerate(keep.items()):
            arr[i].name = name.encode()
            arr[i].dtype = 0
            arr[i].ndim = w.ndim
            for j, s in enumerate(w.shape):
                arr[i].shape[j] = s
            arr[i].data = w.ctypes.data
        check(self.lib.pb_create(C.byref(self.ctx), device, self.BAND, arr, len(keep), C.byref(fc), C.sizeof(fc)))

    def infer_sequence(self, frames: np.ndarray, scale: float = 0.75, iters: int = 12, backward: bool = False,
                       want_flow: bool = True, want_rgb: bool = True, out_flow: Optional[np.ndarray] = None,
                       out_rgb: Optional[np.ndarray] = None):
        """frames uint8 [F,H,W,3] -> (flow f32 [F-1,dirs,sh,sw,2] | None, rgb u8 [F-1,dirs,sh,sw,3] | None, maxdisp [F-1,dirs]).
        out_flow / out_rgb: caller-owned result arrays (page-locked ones are written by the copy engines directly)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        F, H, W, ch = frames.shape
        assert ch == 3 and F >= 2
        sh, sw = flow_out_size(H, W, scale)
        d = 2 if backward else 1
        flow = (out_flow if out_flow is not None else np.empty((F - 1, d, sh, sw, 2), np.float32)) if want_flow else None
        rgb = (out_rgb if out_rgb is not None else np.empty((F - 1, d, sh, sw, 3), np.uint8)) if want_rgb else None
        assert flow is None or (flow.dtype == np.float32 and flow.shape == (F - 1, d, sh, sw, 2) and flow.flags.c_contiguous)
        assert rgb is None or (rgb.dtype == np.uint8 and rgb.shape == (F - 1, d, sh, sw, 3) and rgb.flags.c_contiguous)
        mx = np.empty((F - 1, d), np.float32)
        check(self.lib.pb_flow_infer_sequence(self.ctx, _ptr(frames), F, H, W, C.c_float(scale), iters, int(backward),
                                              _ptr(flow), _ptr(rgb), _ptr(mx)))
        return flow, rgb, mx

    def submit_sequence(self, frames: np.ndarray, scale: float = 0.75, iters: int = 12, backward: bool = False,
                        out_flow: Optional[np.ndarray] = None, out_rgb: Optional[np.ndarray] = None, out_max: Optional[np.ndarray] = None):
        """Asynchronous infer_sequence (pb_flow_submit_sequence): page-locked arrays only (see DepthAnything.submit_batch); `wait()` returns
        (flow, rgb, maxdisp)."""
        F, H, W, ch = frames.shape
        assert ch == 3 and F >= 2 and frames.dtype == np.uint8 and frames.flags.c_contiguous
        sh, sw = flow_out_size(H, W, scale)
        d = 2 if backward else 1
        for a, dt, shp in ((out_flow, np.float32, (F - 1, d, sh, sw, 2)), (out_rgb, np.uint8, (F - 1, d, sh, sw, 3)), (out_max, np.float32, (F - 1, d))):
            assert a is None or (a.dtype == dt and a.shape == shp and a.flags.c_contiguous)
        check(self.lib.pb_flow_submit_sequence(self.ctx, _ptr(frames), F, H, W, C.c_float(scale), iters, int(backward), _ptr(out_flow), _ptr(out_rgb), _ptr(out_max)))
        self._hold((out_flow, out_rgb, out_max), frames, out_flow, out_rgb, out_max)

    def infer_sequence_masks(self, frames: np.ndarray, scale: float = 0.75, iters: int = 12, alpha_1: float = 0.05,
                             alpha_2: float = 0.5, want_flow: bool = True, want_rgb: bool = True, out_flow: Optional[np.ndarray] = None,
                             out_rgb: Optional[np.ndarray] = None, out_mask: Optional[np.ndarray] = None):
        """Both directions plus the forward/backward consistency masks (bands/flow_raft.py:58-64):
        -> (flow | None, rgb | None, maxdisp [F-1,2], mask bool [F-1,2,sh,sw]).  The same chunked three-stage pipeline as infer_sequence;
        out_flow / out_rgb / out_mask (uint8): caller-owned result arrays, page-locked ones are written by the copy engines directly."""
        frames = np.ascontiguousarray(frames, np.uint8)
        F, H, W, ch = frames.shape
        assert ch == 3 and F >= 2
        sh, sw = flow_out_size(H, W, scale)
        flow = (out_flow if out_flow is not None else np.empty((F - 1, 2, sh, sw, 2), np.float32)) if want_flow else None
        rgb = (out_rgb if out_rgb is not None else np.empty((F - 1, 2, sh, sw, 3), np.uint8)) if want_rgb else None
        mx = np.empty((F - 1, 2), np.float32)
        mask = out_mask if out_mask is not None else np.empty((F - 1, 2, sh, sw), np.uint8)
        assert flow is None or (flow.dtype == np.float32 and flow.shape == (F - 1, 2, sh, sw, 2) and flow.flags.c_contiguous)
        assert rgb is None or (rgb.dtype == np.uint8 and rgb.shape == (F - 1, 2, sh, sw, 3) and rgb.flags.c_contiguous)
        assert mask.dtype == np.uint8 and mask.shape == (F - 1, 2, sh, sw) and mask.flags.c_contiguous
        check(self.lib.pb_flow_infer_sequence_masks(self.ctx, _ptr(frames), F, H, W, C.c_float(scale), iters,
                                                    C.c_float(alpha_1), C.c_float(alpha_2), _ptr(flow), _ptr(rgb), _ptr(mx),
                                                    _ptr(mask)))
        return flow, rgb, mx, mask.view(np.bool_)

    def fwdbwd_mask(self, flows: np.ndarray, alpha_1: float = 0.05, alpha_2: float = 0.5) -> np.ndarray:
        """flows f32 [n,2,sh,sw,2] (forward, backward) -> bool [n,2,sh,sw] (bands/common/flow.py:28-40)."""
        flows = np.ascontiguousarray(flows, np.float32)
        n, d, sh, sw, two = flows.shape
        assert d == 2 and two == 2
        mask = np.empty((n, 2, sh, sw), np.uint8)
        check(self.lib.pb_flow_fwdbwd_mask(self.ctx, _ptr(flows), n, sh, sw, C.c_float(alpha_1), C.c_float(alpha_2), _ptr(mask)))
        return mask.view(np.bool_)

    def infer_sequence_dev(self, frames_ptr: int, F: int, H: int, W: int, scale: float, iters: int, backward: bool,
                           flow_ptr: int = 0, rgb_ptr: int = 0, max_ptr: int = 0):
        v = lambda p: C.c_void_p(p) if p else None
        check(self.lib.pb_flow_infer_sequence_dev(self.ctx, v(frames_ptr), F, H, W, C.c_float(scale), iters, int(backward),
                                                  v(flow_ptr), v(rgb_ptr), v(max_ptr)))

    def stage(self, name: str, cap: int = 1 << 26) -> np.ndarray:
        out = np.empty(cap, np.float32)
        shape = (C.c_int64 * 4)()
        n = check(self.lib.pb_flow_get_stage(self.ctx, name.encode(), _ptr(out), cap, shape))
        dims = [int(s) for s in shape]
        while len(dims) > 1 and dims[-1] == 1:
            dims.pop()
        return out[:n].reshape(dims).copy()


class FlowGMFlow(FlowRaft):
    """GMFlow optical-flow band on one GPU (bands/flow_gmflow.py:42-118 init_model / infer; the model at the band's default flags).

    weights: reference checkpoint naming (backbone.*, transformer.layers.N.*, feature_flow_attn.*, upsampler.*).  Same calls as FlowRaft
    (`iters` is ignored: GMFlow is not iterative; frames are padded to multiples of 16).  stage(): "feat", "block0", "tfeat" as
    [frames, tokens, 128], "flow_match", "flow_prop" as [pairs * dirs, tokens, 2]."""
    BAND = b"flow_gmflow"

    def set_inference_size(self, size=None):
        """--inference_size H W of the band (reference flow_gmflow.py:76-100): run the network on a bilinear (align_corners) resize of the
        scaled frame to (H, W), multiples of 16, and resize the flow back; None / (0, 0) = off (pad to /16, the default)."""
        h, w = (0, 0) if not size else (int(size[0]), int(size[1]))
        check(self.lib.pb_flow_set_inference_size(self.ctx, h, w))


def _mask_cfg(cfg: MaskCfg, max_batch: int, precision: int = 0) -> "_lib.pb_mask_cfg":
    return _lib.pb_mask_cfg((C.c_int32 * 4)(*cfg.blocks), cfg.scale_long, cfg.scale_short, cfg.num_classes, cfg.feat_channels,
                            cfg.stacked_convs, (C.c_int32 * 5)(*cfg.num_grids), (C.c_int32 * 5)(*cfg.strides),
                            cfg.mask_feat_channels, cfg.mask_out_channels, cfg.nms_pre, cfg.max_per_img, cfg.score_thr,
                            cfg.mask_thr, cfg.filter_thr, cfg.sigma, max_batch, precision)


def mask_net_size(cfg: MaskCfg | str, H: int, W: int) -> Tuple[int, int, int, int]:
    """(nh, nw, Hp, Wp): keep-ratio rescale to (scale_long, scale_short), then pad to a multiple of 32."""
    cfg = MASK_CFGS[cfg] if isinstance(cfg, str) else cfg
    c = _mask_cfg(cfg, 1)
    v = [C.c_int() for _ in range(4)]
    check(_lib.load().pb_mask_net_size(C.byref(c), H, W, *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


SDF_NTAB = 4130     # squared distances 0 .. 4128 are the ones getSDF's remap does not saturate on either side; entry 4129 = saturated


def sdf_tables(n_tab: int = SDF_NTAB):
    """(tab_out, tab_in): the byte `masks[..., 1] = sdf * 255 -> astype(uint8)` of the reference's getSDF (bands/mask_mmdet.py:64-69,
    150-152) for a pixel outside / inside the mask whose squared Euclidean distance to the other side is i - evaluated with the
    reference's own float64 numpy expression, so the device only has to deliver the exact integer i."""
    d = np.sqrt(np.arange(n_tab, dtype=np.float64))
    d[-1] = 1.0e6                                            # "at least this far": saturated on both sides
    tabs = []
    for sdf in (d, -d):
        v = (sdf + 127.0) / 255.0
        v = (v - 0.25) * 2.0
        g = (1.0 - np.clip(v, 0.0, 1.0)) * 255
        tabs.append(np.ascontiguousarray(g.astype(np.uint8)))
    return tabs[0], tabs[1]


class MaskMMDet(_Ctx):
    """SOLOv2 instance-mask band on one GPU (bands/mask_mmdet.py:36-39 init_model, :131-154 per-frame body).

    weights: mmdet state_dict naming (backbone.*, neck.*, mask_head.*) -> float32 ndarray.
    """

    def __init__(self, weights: Dict[str, np.ndarray], cfg: MaskCfg | str = "r101", device: int = 0, max_batch: int = 4,
                 precision: Optional[int] = None):
        super().__init__()
        self.cfg = MASK_CFGS[cfg] if isinstance(cfg, str) else cfg
        self.precision = _lib.default_precision() if precision is None else int(precision)
        c = _mask_cfg(self.cfg, max_batch, self.precision)
        keep = {k: _f32(v) for k, v in weights.items() if np.asarray(v).dtype.kind == "f"}
        arr = (_lib.pb_tensor * len(keep))()
        for i, (name, w) in enumerate(keep.items()):
            arr[i].name = name.encode()
            arr[i].dtype = 0
            arr[i].ndim = w.ndim
            for j, s in enumerate(w.shape):
                arr[i].shape[j] = s
            arr[i].data = w.ctypes.data
        check(self.lib.pb_create(C.byref(self.ctx), device, b"mask_mmdet", arr, len(keep), C.byref(c), C.sizeof(c)))
        self._hw = None

    def infer_batch(self, frames: np.ndarray, confidence: float = 0.5, keep_classes=None, out: Optional[np.ndarray] = None) -> np.ndarray:
        """frames uint8 [n,H,W,3] RGB -> uint8 [n,H,W,3] accumulated masks of the kept classes.  Calls of more than max_batch frames are
        pipelined (chunk i's id images return while chunk i + 1 runs); `out`: a caller-owned result array (a page-locked one is written by
        the copy engine directly)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        n, H, W, ch = frames.shape
        assert ch == 3
        if out is None:
            out = np.empty_like(frames)
        assert out.dtype == np.uint8 and out.shape == frames.shape and out.flags.c_contiguous
        ids = None if keep_classes is None else np.ascontiguousarray(keep_classes, np.int32)
        check(self.lib.pb_mask_infer_batch(self.ctx, _ptr(frames), n, H, W, C.c_float(confidence), _ptr(ids),
                                           0 if ids is None else len(ids), _ptr(out)))
        self._hw = (H, W)
        return out

    def set_sdf(self, on: bool = True):
        """--sdf of the band (reference mask_mmdet.py:64-69,150-152): every following infer_batch* writes the clamped signed distance
        field of the id image into its green channel, on the device.  The byte is tabulated here with the reference's float64
        expression (sdf_tables) and looked up by the exact squared distance the library computes."""
        if not on:
            check(self.lib.pb_mask_set_sdf(self.ctx, None, None, 0))
            return
        to, ti = sdf_tables()
        check(self.lib.pb_mask_set_sdf(self.ctx, _ptr(to), _ptr(ti), len(to)))

    def sdf_green(self, masks: np.ndarray) -> np.ndarray:
        """id images uint8 [n,H,W,3] -> the same with the SDF in the green channel (needs set_sdf)."""
        out = np.ascontiguousarray(masks, np.uint8).copy()
        n, H, W, ch = out.shape
        assert ch == 3
        check(self.lib.pb_mask_sdf_green(self.ctx, _ptr(out), n, H, W))
        return out

    def infer_batch_dev(self, frames_ptr: int, n: int, H: int, W: int, confidence: float, keep_classes, out_ptr: int):
        ids = None if keep_classes is None else np.ascontiguousarray(keep_classes, np.int32)
        check(self.lib.pb_mask_infer_batch_dev(self.ctx, C.c_void_p(frames_ptr), n, H, W, C.c_float(confidence), _ptr(ids),
                                               0 if ids is None else len(ids), C.c_void_p(out_ptr)))
        self._hw = (H, W)

    def instances(self, frame: int, with_masks: bool = False):
        """(scores, labels, masks bool [k,H,W] | None, candidate_count) of frame `frame` of the last call."""
        cap = self.cfg.max_per_img
        sc, lb, cand = np.empty(cap, np.float32), np.empty(cap, np.int32), C.c_int32()
        masks = None
        if with_masks:
            H, W = self._hw
            masks = np.empty((cap, H, W), np.uint8)
        k = check(self.lib.pb_mask_get_instances(self.ctx, frame, cap, _ptr(sc), _ptr(lb), _ptr(masks), C.byref(cand)))
        return sc[:k].copy(), lb[:k].copy(), (masks[:k].view(np.bool_) if with_masks else None), cand.value

    def stage(self, name: str, cap: int = 1 << 27) -> np.ndarray:
        out = np.empty(cap, np.float32)
        shape = (C.c_int64 * 4)()
        n = check(self.lib.pb_mask_get_stage(self.ctx, name.encode(), _ptr(out), cap, shape))
        return out[:n].reshape([int(s) for s in shape]).copy()
