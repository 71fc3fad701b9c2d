// C ABI of libprisma_bands.so (include/prisma_bands.h).
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <deque>
#include <mutex>
#include <vector>

#include "engine.h"
#include "mask_engine.h"
#include "gmflow_engine.h"
#include "raft_engine.h"

static thread_local char g_err[1024] = "";

void pb_cubic_taps(int src, int dst, int *idx, float *wt);   // engine.hip

void pb_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// pinned staging + device slots of the host-pointer entry points (pb_depth_infer_batch)
struct HostPipe {
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_h2d[2] = {}, ev_comp[2] = {}, ev_d2h[2] = {};
    void *h_in[2] = {}, *h_depth[2] = {}, *h_rgb[2] = {}, *h_mm[2] = {};
    void *d_in[2] = {}, *d_depth[2] = {}, *d_rgb[2] = {}, *d_mm[2] = {};
    size_t cap_in = 0, cap_d = 0, cap_r = 0, cap_m = 0;
    unsigned seq = 0;               // chunks enqueued so far: chunk i of a call takes slot (seq + i) & 1, so consecutive one-chunk calls alternate slots
    // device slots always; the pinned staging halves only for callers whose own buffers are pageable (`host`)
    int grow(void **h, void **d, size_t &cap, size_t need, bool host = true) {
        if (need <= cap && (!host || h[0] || !need)) return 0;
        const size_t want = need > cap ? need : cap;
        for (int i = 0; i < 2; ++i) {
            if (need > cap) {
                if (h[i]) PB_HIP(hipHostFree(h[i]));
                if (d[i]) PB_HIP(hipFree(d[i]));
                h[i] = d[i] = nullptr;
                PB_HIP(hipMalloc(&d[i], want));
            }
            if (host && !h[i]) PB_HIP(hipHostMalloc(&h[i], want, hipHostMallocDefault));
        }
        cap = want;
        return 0;
    }
    int ensure(size_t in_b, size_t d_b, size_t r_b, size_t m_b, bool host_in = true, bool host_d = true, bool host_r = true) {
        if (!s_in) {
            PB_HIP(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
            PB_HIP(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
            for (int i = 0; i < 2; ++i) {
                PB_HIP(hipEventCreateWithFlags(&ev_h2d[i], hipEventDisableTiming));
                PB_HIP(hipEventCreateWithFlags(&ev_comp[i], hipEventDisableTiming));
                PB_HIP(hipEventCreateWithFlags(&ev_d2h[i], hipEventDisableTiming));
            }
        }
        int r;
        if ((r = grow(h_in, d_in, cap_in, in_b, host_in)) || (r = grow(h_depth, d_depth, cap_d, d_b, host_d)) ||
            (r = grow(h_rgb, d_rgb, cap_r, r_b, host_r)) || (r = grow(h_mm, d_mm, cap_m, m_b)))
            return r;
        return 0;
    }
    void release() {
        if (s_in) { hipStreamSynchronize(s_in); hipStreamSynchronize(s_out); }
        void **hs[] = {h_in, h_depth, h_rgb, h_mm}, **ds[] = {d_in, d_depth, d_rgb, d_mm};
        for (int k = 0; k < 4; ++k)
            for (int i = 0; i < 2; ++i) {
                if (hs[k][i]) hipHostFree(hs[k][i]);
                if (ds[k][i]) hipFree(ds[k][i]);
            }
        for (int i = 0; i < 2; ++i) {
            if (ev_h2d[i]) hipEventDestroy(ev_h2d[i]);
            if (ev_comp[i]) hipEventDestroy(ev_comp[i]);
            if (ev_d2h[i]) hipEventDestroy(ev_d2h[i]);
        }
        if (s_in) { hipStreamDestroy(s_in); hipStreamDestroy(s_out); }
    }
};

// The caller's buffer is page-locked host memory (hipHostMalloc / hipHostRegister / torch pin_memory): the copy engines can address it
// directly, so the host-pointer entry points skip their own pinned staging copies (a 1080p clip of 32 frames is 199 MB each way: ~20 ms of
// one core per memcpy, which the pipeline can only hide while a LATER chunk computes)
static bool pb_is_pinned(const void *p, size_t bytes = 1) {
    if (!p) return false;
    // both ends of the range must be page-locked (a caller may hand a pointer into the middle of a registered region whose end lies before ours)
    for (const char *q : {(const char *)p, (const char *)p + (bytes ? bytes - 1 : 0)}) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, q) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (at.type != hipMemoryTypeHost) return false;
    }
    return true;
}

// Error exits of the host-pointer pipelines: with page-locked caller buffers the copy engines read and write CALLER memory asynchronously on the
// pipeline's two copy streams, so no path may return - with or without an error - while a copy or a kernel that feeds one is still in flight:
// the caller is free to release or reuse its buffers the moment the call returns.  Declared after the streams exist; its destructor drains them.
struct PipeDrain {
    hipStream_t a, b, c;
    ~PipeDrain() {
        if (a) (void)hipStreamSynchronize(a);
        if (b) (void)hipStreamSynchronize(b);
        if (c) (void)hipStreamSynchronize(c);
    }
};

// One host-pointer call = several engine infer() calls (one per chunk): the per-launch timer is reset ONCE per API call and accumulates over the
// chunks, so pb_get_kernel_stats describes the whole call as it does for a device-pointer call (stage snapshots - pb_*_get_stage -
// still describe the last chunk).  Restores the caller's accumulate setting.
struct TimerSpan {
    KernelTimer &t;
    bool saved;
    explicit TimerSpan(KernelTimer &timer) : t(timer), saved(timer.accumulate) { t.reset(); t.accumulate = true; }
    ~TimerSpan() { t.accumulate = saved; }
};

// pb_flow_infer_sequence: the same three-stage pipeline over chunks of frame pairs (a chunk = its pairs' frames + one halo frame)
struct FlowPipe {
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_h2d[2] = {}, ev_comp[2] = {}, ev_d2h[2] = {};
    void *h[5][2] = {}, *d[5][2] = {};        // 0 frames in, 1 flow out, 2 rgb out, 3 max displacement out, 4 consistency masks out
    size_t cap[5] = {};
    unsigned seq = 0;               // chunks enqueued so far (slot of chunk i of a call: (seq + i) & 1)
    int grow(int k, size_t need, bool host) {
        if (need <= cap[k] && (!host || h[k][0])) return 0;
        const size_t want = need > cap[k] ? need : cap[k];
        for (int i = 0; i < 2; ++i) {
            if (need > cap[k]) {
                if (d[k][i]) PB_HIP(hipFree(d[k][i]));
                d[k][i] = nullptr;
                PB_HIP(hipMalloc(&d[k][i], want));
                if (h[k][i]) { PB_HIP(hipHostFree(h[k][i])); h[k][i] = nullptr; }
            }
            if (host && !h[k][i]) PB_HIP(hipHostMalloc(&h[k][i], want, hipHostMallocDefault));
        }
        cap[k] = want;
        return 0;
    }
    int ensure_streams() {
        if (s_in) return 0;
        PB_HIP(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
        PB_HIP(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            PB_HIP(hipEventCreateWithFlags(&ev_h2d[i], hipEventDisableTiming));
            PB_HIP(hipEventCreateWithFlags(&ev_comp[i], hipEventDisableTiming));
            PB_HIP(hipEventCreateWithFlags(&ev_d2h[i], hipEventDisableTiming));
        }
        return 0;
    }
    void release() {
        if (s_in) { hipStreamSynchronize(s_in); hipStreamSynchronize(s_out); }
        for (int k = 0; k < 5; ++k)
            for (int i = 0; i < 2; ++i) {
                if (h[k][i]) hipHostFree(h[k][i]);
                if (d[k][i]) hipFree(d[k][i]);
            }
        for (int i = 0; i < 2; ++i) {
            if (ev_h2d[i]) hipEventDestroy(ev_h2d[i]);
            if (ev_comp[i]) hipEventDestroy(ev_comp[i]);
            if (ev_d2h[i]) hipEventDestroy(ev_d2h[i]);
        }
        if (s_in) { hipStreamDestroy(s_in); hipStreamDestroy(s_out); }
    }
};

// pb_mask_infer_batch: the engine walks the batch in chunks of max_batch frames itself and takes whole-batch device pointers, so the pipeline is
// whole-batch device buffers + one event per chunk: every chunk's H2D is enqueued up front on s_in, chunk i's kernels wait for event i, and its id
// images leave on s_out as soon as its last launch is enqueued (MaskEngine::chunk_begin / chunk_end).  Buffers and streams stay on the ctx.
struct MaskPipe {
    hipStream_t s_in = nullptr, s_out = nullptr;
    void *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;
    size_t cap_d = 0, cap_hi = 0, cap_ho = 0;
    std::vector<hipEvent_t> ev_in, ev_done;
    int ensure(size_t bytes, bool host_in, bool host_out, int chunks) {
        if (!s_in) {
            PB_HIP(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
            PB_HIP(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
        }
        if (bytes > cap_d) {
            if (d_in) PB_HIP(hipFree(d_in));
            if (d_out) PB_HIP(hipFree(d_out));
            d_in = d_out = nullptr; cap_d = 0;
            PB_HIP(hipMalloc(&d_in, bytes));
            PB_HIP(hipMalloc(&d_out, bytes));
            cap_d = bytes;
        }
        if (host_in && bytes > cap_hi) {
            if (h_in) PB_HIP(hipHostFree(h_in));
            h_in = nullptr; cap_hi = 0;
            PB_HIP(hipHostMalloc(&h_in, bytes, hipHostMallocDefault));
            cap_hi = bytes;
        }
        if (host_out && bytes > cap_ho) {
            if (h_out) PB_HIP(hipHostFree(h_out));
            h_out = nullptr; cap_ho = 0;
            PB_HIP(hipHostMalloc(&h_out, bytes, hipHostMallocDefault));
            cap_ho = bytes;
        }
        while ((int)ev_in.size() < chunks) {
            hipEvent_t a, b;
            PB_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
            PB_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
            ev_in.push_back(a); ev_done.push_back(b);
        }
        return 0;
    }
    void release() {
        if (s_in) { hipStreamSynchronize(s_in); hipStreamSynchronize(s_out); }
        if (d_in) hipFree(d_in);
        if (d_out) hipFree(d_out);
        if (h_in) hipHostFree(h_in);
        if (h_out) hipHostFree(h_out);
        for (auto e : ev_in) hipEventDestroy(e);
        for (auto e : ev_done) hipEventDestroy(e);
        if (s_in) { hipStreamDestroy(s_in); hipStreamDestroy(s_out); }
    }
};

struct pb_ctx {
    int device = 0;
    DepthEngine *depth = nullptr;
    RaftEngine *raft = nullptr;
    MaskEngine *mask = nullptr;
    hipStream_t stream = nullptr;   // == depth->stream when a band is loaded
    f16 *zero = nullptr;
    bool own_stream = false;
    int gemm_tile = TILE_AUTO, conv_tile = TILE_AUTO;
    int op_splitk = 0;              // pb_set_option("op_splitk", 1): pb_op_gemm / pb_op_conv2d lend launch_gemm a split-K workspace (gemm.h splitk; op-level tests)
    int host_chunk = 0;             // pb_set_option("host_chunk"): frames (depth) / frame pairs (flow) per chunk of the host-pointer pipelines, 0 = default
    HostPipe pipe;
    FlowPipe fpipe;
    MaskPipe mpipe;
    std::deque<hipEvent_t> pending;     // completion events of pb_*_submit_* calls not yet waited for (pb_wait pops the oldest)
    // pb_comm_init: RCCL communicator of the ranks (one process per GPU) for pb_gather_scalars
    void *comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    float *comm_buf = nullptr;
    size_t comm_cap = 0;
    // pb_depth_encode_still: its own stream and buffers, kept for the life of the ctx (the band calls it from its sink thread for
    // every --subpath frame while the main thread's batches run on `stream`; a hipMalloc / hipFree pair per frame synchronised the device)
    hipStream_t still_stream = nullptr;
    char *still_buf = nullptr;
    size_t still_cap = 0;
    std::mutex still_mu;
};

namespace {
struct DevMem {
    void *p = nullptr;
    ~DevMem() { if (p) hipFree(p); }
    int alloc(size_t bytes) {
        PB_HIP(hipMalloc(&p, bytes < 256 ? 256 : bytes));
        PB_HIP(hipMemset(p, 0, bytes < 256 ? 256 : bytes));
        PB_HIP(hipDeviceSynchronize());     // memset runs on the null stream; kernels use the ctx stream
        return 0;
    }
    template <class T> T *as() { return (T *)p; }
};
#define PB_TRY(expr) do { int _r = (expr); if (_r) return _r; } while (0)
}  // namespace

extern "C" {

const char *pb_last_error(void) { return g_err; }
int pb_version(void) { return 100; }
int pb_abi_version(void) { return PB_ABI_VERSION; }
int pb_struct_size(int which) {
    switch (which) {
        case 0: return (int)sizeof(pb_tensor);
        case 1: return (int)sizeof(pb_depth_cfg);
        case 2: return (int)sizeof(pb_flow_cfg);
        case 3: return (int)sizeof(pb_mask_cfg);
        case 4: return (int)sizeof(pb_kernel_stat);
        case 5: return (int)sizeof(pb_comm_id);
        default: return -1;
    }
}

int pb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pb_create(pb_ctx **out, int device_id, const char *band, const pb_tensor *weights, int n_weights, const void *cfg,
              size_t cfg_bytes) {
    PB_CHECK(out && band, PB_ERR_ARG, "pb_create: null argument");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    PB_CHECK(e == hipSuccess && ndev > 0, PB_ERR_DEVICE, "no HIP device available (%s); this library has no CPU path",
             e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    PB_CHECK(device_id >= 0 && device_id < ndev, PB_ERR_ARG, "device %d out of range (%d devices)", device_id, ndev);
    PB_HIP(hipSetDevice(device_id));
    pb_ctx *c = new pb_ctx();
    c->device = device_id;
    if (!strcmp(band, "depth_anything")) {
        if (!cfg || cfg_bytes != sizeof(pb_depth_cfg) || !weights || n_weights <= 0) {
            delete c;
            PB_CHECK(false, PB_ERR_ARG, "depth_anything: needs a pb_depth_cfg (%zu bytes, got %zu) and weights",
                     sizeof(pb_depth_cfg), cfg_bytes);
        }
        c->depth = new DepthEngine(device_id, *(const pb_depth_cfg *)cfg);
        int r = c->depth->load(weights, n_weights);
        if (r) {
            delete c->depth;
            delete c;
            return r;
        }
        c->stream = c->depth->stream;
        c->zero = (f16 *)c->depth->zero_page();
    } else if (!strcmp(band, "flow_raft") || !strcmp(band, "flow_gmflow")) {
        const bool gm = band[5] == 'g';
        if (!weights || n_weights <= 0 || (cfg && cfg_bytes != sizeof(pb_flow_cfg))) {
            delete c;
            PB_CHECK(false, PB_ERR_ARG, "%s: needs weights (and cfg = NULL or a pb_flow_cfg of %zu bytes, got %zu)", band, sizeof(pb_flow_cfg), cfg_bytes);
        }
        const int prec = cfg ? ((const pb_flow_cfg *)cfg)->precision : PB_PREC_F16;
        if (prec != PB_PREC_F16 && prec != PB_PREC_SPLIT) {
            delete c;
            PB_CHECK(false, PB_ERR_ARG, "%s: precision %d unknown", band, prec);
        }
        c->raft = gm ? new GmflowEngine(device_id) : new RaftEngine(device_id);      // GmflowEngine overrides load / infer / get_stage
        c->raft->split_w_ = prec == PB_PREC_SPLIT;
        {
            const char *mx = getenv("PB_MX");
            c->raft->mx_ = !gm && c->raft->split_w_ && !(mx && mx[0] == '0');
        }
        int r = c->raft->load(weights, n_weights);
        if (r) {
            delete c->raft;
            delete c;
            return r;
        }
        c->stream = c->raft->stream;
        c->zero = (f16 *)c->raft->zero_page();
    } else if (!strcmp(band, "mask_mmdet")) {
        if (!cfg || cfg_bytes != sizeof(pb_mask_cfg) || !weights || n_weights <= 0) {
            delete c;
            PB_CHECK(false, PB_ERR_ARG, "mask_mmdet: needs a pb_mask_cfg (%zu bytes, got %zu) and weights", sizeof(pb_mask_cfg),
                     cfg_bytes);
        }
        c->mask = new MaskEngine(device_id, *(const pb_mask_cfg *)cfg);
        int r = c->mask->load(weights, n_weights);
        if (r) {
            delete c->mask;
            delete c;
            return r;
        }
        c->stream = c->mask->stream;
        c->zero = (f16 *)c->mask->zero_page();
    } else if (!strcmp(band, "ops")) {
        // kernel-level context for the parity tests: a stream and a zero page, no model
        hipError_t e1 = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        void *z = nullptr;
        hipError_t e2 = hipMalloc(&z, 4096);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            delete c;
            PB_CHECK(false, PB_ERR_DEVICE, "ops context: stream/alloc failed");
        }
        hipMemset(z, 0, 4096);
        hipDeviceSynchronize();
        c->zero = (f16 *)z;
        c->own_stream = true;
    } else {
        delete c;
        PB_CHECK(false, PB_ERR_ARG, "unknown band '%s' (depth_anything | flow_raft | flow_gmflow | mask_mmdet | ops)", band);
    }
    *out = c;
    return 0;
}

// ---- RCCL (the per-frame scalar all-gather of SURVEY 8(e)) -------------------------------------------------
// librccl is opened lazily and privately (dlopen, RTLD_LOCAL): the library carries no link-time dependency on it and does
// not interpose the copy a host framework (torch.distributed's "nccl" backend) may have loaded.
namespace {
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, pb_comm_id, int) = nullptr;      // ncclUniqueId is passed by value: 128 bytes
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load() {
        if (h) return true;
        for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return false;
        GetUniqueId = (decltype(GetUniqueId))dlsym(h, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(h, "ncclCommInitRank");
        AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        return GetUniqueId && CommInitRank && AllGather && CommDestroy && GetErrorString;
    }
} g_rccl;
}  // namespace

int pb_comm_unique_id(pb_comm_id *id_out) {
    PB_CHECK(id_out, PB_ERR_ARG, "comm_unique_id: null");
    PB_CHECK(g_rccl.load(), PB_ERR_DEVICE, "librccl could not be loaded: %s", dlerror());
    const int rc = g_rccl.GetUniqueId(id_out);
    PB_CHECK(rc == 0, PB_ERR_DEVICE, "ncclGetUniqueId: %s", g_rccl.GetErrorString(rc));
    return 0;
}

int pb_comm_init(pb_ctx *c, const pb_comm_id *id, int rank, int world) {
    PB_CHECK(c && id && world >= 1 && rank >= 0 && rank < world, PB_ERR_ARG, "comm_init: bad arguments");
    PB_CHECK(!c->comm, PB_ERR_STATE, "comm_init: the ctx already has a communicator");
    PB_CHECK(g_rccl.load(), PB_ERR_DEVICE, "librccl could not be loaded: %s", dlerror());
    PB_HIP(hipSetDevice(c->device));
    const int rc = g_rccl.CommInitRank(&c->comm, world, *id, rank);
    PB_CHECK(rc == 0, PB_ERR_DEVICE, "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(rc));
    c->comm_rank = rank; c->comm_world = world;
    return 0;
}

int pb_gather_scalars(pb_ctx *c, const float *local, int n_local, float *global) {
    PB_CHECK(c && c->comm && local && global && n_local > 0, PB_ERR_ARG, "gather_scalars: needs pb_comm_init and non-empty buffers");
    PB_HIP(hipSetDevice(c->device));
    const size_t need = (size_t)(c->comm_world + 1) * n_local;
    if (need > c->comm_cap) {
        if (c->comm_buf) PB_HIP(hipFree(c->comm_buf));
        c->comm_buf = nullptr; c->comm_cap = 0;
        PB_HIP(hipMalloc((void **)&c->comm_buf, need * 4));
        c->comm_cap = need;
    }
    float *send = c->comm_buf, *recv = c->comm_buf + n_local;
    PB_HIP(hipMemcpyAsync(send, local, (size_t)n_local * 4, hipMemcpyHostToDevice, c->stream));
    const int rc = g_rccl.AllGather(send, recv, (size_t)n_local, 7 /* ncclFloat32 */, c->comm, c->stream);
    PB_CHECK(rc == 0, PB_ERR_DEVICE, "ncclAllGather: %s", g_rccl.GetErrorString(rc));
    PB_HIP(hipMemcpyAsync(global, recv, (size_t)c->comm_world * n_local * 4, hipMemcpyDeviceToHost, c->stream));
    PB_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

void pb_destroy(pb_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->comm) { hipStreamSynchronize(c->stream); g_rccl.CommDestroy(c->comm); }
    if (c->comm_buf) hipFree(c->comm_buf);
    if (c->still_stream) { hipStreamSynchronize(c->still_stream); hipStreamDestroy(c->still_stream); }
    if (c->still_buf) hipFree(c->still_buf);
    for (hipEvent_t e : c->pending) { hipEventSynchronize(e); hipEventDestroy(e); }
    c->pipe.release();
    c->fpipe.release();
    c->mpipe.release();
    if (c->depth) delete c->depth;
    if (c->raft) delete c->raft;
    if (c->mask) delete c->mask;
    if (c->own_stream) {
        hipStreamSynchronize(c->stream);
        hipFree(c->zero);
        hipStreamDestroy(c->stream);
    }
    delete c;
}

int pb_sync(pb_ctx *c) {
    PB_CHECK(c, PB_ERR_ARG, "null ctx");
    PB_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int pb_depth_infer_batch_dev(pb_ctx *c, const uint8_t *frames, int n, int H, int W, float *depth_out, uint8_t *rgb_out,
                             float *min_out, float *max_out, int flip) {
    PB_CHECK(c && c->depth, PB_ERR_STATE, "ctx has no depth_anything band");
    return c->depth->infer(frames, n, H, W, depth_out, rgb_out, min_out, max_out, flip);
}

// Host-pointer variant: a three-stage pipeline over chunks of max_batch frames.  Stage 1 copies the caller's frames
// into a pinned staging buffer and DMAs them to HBM on a copy stream, stage 2 is the band on the ctx stream, stage 3
// DMAs the results into pinned memory on a second copy stream and hands them to the caller's arrays.  Two slots per
// stage, ordered by events, so the PCIe traffic of chunks i+1 and i-1 overlaps the compute of chunk i.
// `submit`: the asynchronous form (pb_depth_submit_batch) - every caller buffer must be page-locked, nothing in the loop waits on the host (slot
// reuse is ordered by events on the streams: the H2D into a slot waits for the compute that last read it, the compute into a slot for the D2H
// that last emptied it), the min / max floats are DMAed straight into the caller's arrays, and the call returns after the enqueue with a
// completion event queued for pb_wait().  The blocking form keeps its host-side finish() for pageable buffers and drains on every exit.
static int depth_host_pipeline(pb_ctx *c, const uint8_t *frames, int n, int H, int W, float *depth_out, uint8_t *rgb_out, float *min_out,
                               float *max_out, int flip, bool submit) {
    PB_HIP(hipSetDevice(c->device));
    const size_t px = (size_t)H * W;
    // frames per chunk: pb_set_option("host_chunk"), else max_batch.  (Round 6 first cut calls of >= 16 frames in two so that one half's copies ran
    // under the other half's kernels: on the bench clip the 16-frame launches' longer tails cost more than the hidden copies gave -
    // profiles/r06j_pcie_entry_points.txt; copies are hidden ACROSS calls by the asynchronous form instead.)
    const int mb = c->depth->max_batch();
    const int cap = std::min(n, c->host_chunk > 0 ? std::min(c->host_chunk, mb) : mb);
    HostPipe &hp = c->pipe;
    const size_t in_b = (size_t)cap * px * 3, d_b = depth_out ? (size_t)cap * px * 4 : 0, r_b = rgb_out ? in_b : 0,
                 m_b = (size_t)cap * 8;
    // page-locked caller buffers are read / written by the copy engines directly (no staging memcpy on this thread)
    const bool pin_in = pb_is_pinned(frames, (size_t)n * px * 3), pin_d = pb_is_pinned(depth_out, (size_t)n * px * 4),
               pin_r = pb_is_pinned(rgb_out, (size_t)n * px * 3);
    const bool pin_mm = (!min_out || pb_is_pinned(min_out, (size_t)n * 4)) && (!max_out || pb_is_pinned(max_out, (size_t)n * 4));
    if (submit) PB_CHECK(pin_in && (!depth_out || pin_d) && (!rgb_out || pin_r) && pin_mm, PB_ERR_ARG,
                         "depth submit: every buffer of an asynchronous call must be page-locked host memory (hipHostMalloc / hipHostRegister / torch pin_memory)");
    PB_TRY(hp.ensure(in_b, d_b, r_b, m_b, !pin_in, !pin_d, !pin_r));
    PipeDrain drain{submit ? nullptr : hp.s_in, submit ? nullptr : c->stream, submit ? nullptr : hp.s_out};      // blocking form: no exit leaves a copy into caller memory in flight
    struct SubmitGuard {            // asynchronous form: an error exit drains too (the caller gets an error, not a half-enqueued call)
        hipStream_t a, b, c; bool armed;
        ~SubmitGuard() { if (armed) { (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b); (void)hipStreamSynchronize(c); } }
    } guard{hp.s_in, c->stream, hp.s_out, submit};
    TimerSpan span(c->depth->timer);
    const int chunks = (n + cap - 1) / cap;
    const unsigned base = hp.seq;
    hp.seq += (unsigned)chunks;
    auto finish = [&](int i) -> int {           // results of chunk i: pinned -> caller
        const int slot = (int)((base + i) & 1), s0 = i * cap, m = std::min(cap, n - s0);
        PB_HIP(hipEventSynchronize(hp.ev_d2h[slot]));
        if (depth_out && !pin_d) memcpy(depth_out + (size_t)s0 * px, hp.h_depth[slot], (size_t)m * px * 4);
        if (rgb_out && !pin_r) memcpy(rgb_out + (size_t)s0 * px * 3, hp.h_rgb[slot], (size_t)m * px * 3);
        const float *mm = (const float *)hp.h_mm[slot];
        if (min_out) memcpy(min_out + s0, mm, (size_t)m * 4);
        if (max_out) memcpy(max_out + s0, mm + cap, (size_t)m * 4);
        return 0;
    };
    for (int i = 0; i < chunks; ++i) {
        const int slot = (int)((base + i) & 1), s0 = i * cap, m = std::min(cap, n - s0);
        if (!submit && i >= 2) PB_TRY(finish(i - 2));       // frees this slot's pinned output and (through ev_d2h) its device buffers
        PB_HIP(hipStreamWaitEvent(hp.s_in, hp.ev_comp[slot], 0));      // the compute that last read this input slot: this call's chunk i - 2 or an earlier submission's (also in the blocking form: a submission may still be in flight)
        const void *src = frames + (size_t)s0 * px * 3;
        if (!pin_in) { memcpy(hp.h_in[slot], src, (size_t)m * px * 3); src = hp.h_in[slot]; }
        PB_HIP(hipMemcpyAsync(hp.d_in[slot], src, (size_t)m * px * 3, hipMemcpyHostToDevice, hp.s_in));
        PB_HIP(hipEventRecord(hp.ev_h2d[slot], hp.s_in));
        PB_HIP(hipStreamWaitEvent(c->stream, hp.ev_h2d[slot], 0));
        PB_HIP(hipStreamWaitEvent(c->stream, hp.ev_d2h[slot], 0));     // the D2H that last emptied this output slot
        float *mn = (float *)hp.d_mm[slot], *mx = mn + cap;
        PB_TRY(c->depth->infer((const uint8_t *)hp.d_in[slot], m, H, W, (float *)(d_b ? hp.d_depth[slot] : nullptr),
                               (uint8_t *)(r_b ? hp.d_rgb[slot] : nullptr), mn, mx, flip));
        PB_HIP(hipEventRecord(hp.ev_comp[slot], c->stream));
        PB_HIP(hipStreamWaitEvent(hp.s_out, hp.ev_comp[slot], 0));
        if (d_b) PB_HIP(hipMemcpyAsync(pin_d ? (void *)(depth_out + (size_t)s0 * px) : hp.h_depth[slot], hp.d_depth[slot], (size_t)m * px * 4, hipMemcpyDeviceToHost, hp.s_out));
        if (r_b) PB_HIP(hipMemcpyAsync(pin_r ? (void *)(rgb_out + (size_t)s0 * px * 3) : hp.h_rgb[slot], hp.d_rgb[slot], (size_t)m * px * 3, hipMemcpyDeviceToHost, hp.s_out));
        if (submit) {
            if (min_out) PB_HIP(hipMemcpyAsync(min_out + s0, mn, (size_t)m * 4, hipMemcpyDeviceToHost, hp.s_out));
            if (max_out) PB_HIP(hipMemcpyAsync(max_out + s0, mx, (size_t)m * 4, hipMemcpyDeviceToHost, hp.s_out));
        } else {
            PB_HIP(hipMemcpyAsync(hp.h_mm[slot], hp.d_mm[slot], m_b, hipMemcpyDeviceToHost, hp.s_out));
        }
        PB_HIP(hipEventRecord(hp.ev_d2h[slot], hp.s_out));
    }
    if (submit) {
        hipEvent_t done;
        PB_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        PB_HIP(hipEventRecord(done, hp.s_out));               // s_out runs its copies in order: the last chunk's results are the last thing it does
        c->pending.push_back(done);
        guard.armed = false;
        return 0;
    }
    for (int i = std::max(0, chunks - 2); i < chunks; ++i) PB_TRY(finish(i));
    return 0;
}

int pb_depth_infer_batch(pb_ctx *c, const uint8_t *frames, int n, int H, int W, float *depth_out, uint8_t *rgb_out,
                         float *min_out, float *max_out, int flip) {
    PB_CHECK(c && c->depth, PB_ERR_STATE, "ctx has no depth_anything band");
    PB_CHECK(frames && n > 0 && H > 0 && W > 0, PB_ERR_ARG, "infer: bad arguments");
    return depth_host_pipeline(c, frames, n, H, W, depth_out, rgb_out, min_out, max_out, flip, false);
}

int pb_depth_submit_batch(pb_ctx *c, const uint8_t *frames, int n, int H, int W, float *depth_out, uint8_t *rgb_out,
                          float *min_out, float *max_out, int flip) {
    PB_CHECK(c && c->depth, PB_ERR_STATE, "ctx has no depth_anything band");
    PB_CHECK(frames && n > 0 && H > 0 && W > 0, PB_ERR_ARG, "submit: bad arguments");
    return depth_host_pipeline(c, frames, n, H, W, depth_out, rgb_out, min_out, max_out, flip, true);
}

int pb_wait(pb_ctx *c) {
    PB_CHECK(c, PB_ERR_ARG, "null ctx");
    PB_CHECK(!c->pending.empty(), PB_ERR_STATE, "pb_wait: no submission is outstanding on this ctx");
    PB_HIP(hipSetDevice(c->device));
    hipEvent_t done = c->pending.front();
    c->pending.pop_front();
    const hipError_t e = hipEventSynchronize(done);
    (void)hipEventDestroy(done);
    PB_CHECK(e == hipSuccess, PB_ERR_DEVICE, "pb_wait: %s", hipGetErrorString(e));
    return 0;
}

int64_t pb_depth_get_stage(pb_ctx *c, const char *name, float *out, int64_t cap, int64_t shape_out[4]) {
    PB_CHECK(c && c->depth && name && out && shape_out, PB_ERR_ARG, "get_stage: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    return c->depth->get_stage(name, out, cap, shape_out);
}

int pb_set_profiling(pb_ctx *c, int enabled) {
    PB_CHECK(c && (c->depth || c->raft || c->mask), PB_ERR_STATE, "ctx has no band");
    KernelTimer *t = c->depth ? &c->depth->timer : (c->raft ? &c->raft->timer : &c->mask->timer);
    t->enabled = (enabled & 1) != 0;
    t->accumulate = (enabled & 4) != 0;
    t->clear();
    if (c->depth) c->depth->debug = (enabled & 2) != 0;
    if (c->raft) c->raft->debug = (enabled & 2) != 0;
    if (c->mask) c->mask->debug = (enabled & 2) != 0;
    return 0;
}

int pb_flow_out_size(int H, int W, float scale, int *sh, int *sw) {
    PB_CHECK(H > 0 && W > 0 && scale > 0.f && sh && sw, PB_ERR_ARG, "flow_out_size: bad arguments");
    RaftEngine::out_size(H, W, scale, sh, sw);
    return 0;
}

int pb_flow_infer_sequence_dev(pb_ctx *c, const uint8_t *frames, int F, int H, int W, float scale, int iters, int backward,
                               float *flow_out, uint8_t *rgb_out, float *maxdisp_out) {
    PB_CHECK(c && c->raft, PB_ERR_STATE, "ctx has no flow_raft band");
    return c->raft->infer(frames, F, H, W, scale, iters, backward, flow_out, rgb_out, maxdisp_out);
}

// Host-pointer variants (reference loop bands/flow_raft.py:98-113: a decoded frame pair per iteration; :63-64 the consistency masks of --mask):
// a three-stage pipeline over chunks of PB_FLOW_HOST_PAIRS (16) frame pairs - frames of chunk i + 1 to HBM on a copy stream, chunk i on the ctx
// stream, results of chunk i - 1 to the host on a second copy stream; two slots per stage ordered by events, device / pinned buffers kept on the
// ctx.  A chunk carries one halo frame, so the frame it shares with its neighbour is encoded twice (1 / 16 of the encoders' work); every pair's
// flow (and mask: both directions of a pair are in the same chunk) is that of one whole-sequence call - pairs do not interact.  Page-locked
// caller buffers are used directly.  mask_out != NULL: both directions + the forward / backward consistency masks (pb_flow_infer_sequence_masks).
static int flow_host_pipeline(pb_ctx *c, const uint8_t *frames, int F, int H, int W, float scale, int iters, int backward,
                              float *flow_out, uint8_t *rgb_out, float *maxdisp_out, uint8_t *mask_out, float alpha1, float alpha2, bool submit = false) {
    PB_HIP(hipSetDevice(c->device));
    int sh, sw;
    RaftEngine::out_size(H, W, scale, &sh, &sw);
    const int dirs = backward ? 2 : 1, pairs = F - 1;
    // (default 32 pairs since round 6 - was 16: a chunk's launches are the same kernels on fewer rows, and 16-pair chunks cost the bench clip more in
    // launch tails and re-encoded halo frames than their hidden copies gave; profiles/r06j_pcie_entry_points.txt)
    static const int env_cp = pb_env_int("PB_FLOW_HOST_PAIRS", 32);
    int cp = c->host_chunk > 0 ? c->host_chunk : (env_cp > 0 ? env_cp : 32);
    if (pairs <= cp + cp / 4) cp = pairs;                       // a short tail is not worth a chunk of its own
    const int chunks = (pairs + cp - 1) / cp;
    const size_t fpx = (size_t)H * W * 3, px = (size_t)sh * sw, nd = (size_t)pairs * dirs;
    FlowPipe &fp = c->fpipe;
    const bool pin_in = pb_is_pinned(frames, (size_t)F * fpx), pin_f = pb_is_pinned(flow_out, nd * px * 8), pin_r = pb_is_pinned(rgb_out, nd * px * 3),
               pin_k = pb_is_pinned(mask_out, nd * px);
    PB_TRY(fp.ensure_streams());
    // `submit`: the asynchronous form (pb_flow_submit_sequence; depth_host_pipeline above has the rules)
    if (submit) PB_CHECK(pin_in && (!flow_out || pin_f) && (!rgb_out || pin_r) && (!mask_out || pin_k) && (!maxdisp_out || pb_is_pinned(maxdisp_out, nd * 4)), PB_ERR_ARG,
                         "flow submit: every buffer of an asynchronous call must be page-locked host memory (hipHostMalloc / hipHostRegister / torch pin_memory)");
    PipeDrain drain{submit ? nullptr : fp.s_in, submit ? nullptr : c->stream, submit ? nullptr : fp.s_out};      // blocking form: no exit leaves a copy into caller memory in flight
    struct SubmitGuard {
        hipStream_t a, b, c; bool armed;
        ~SubmitGuard() { if (armed) { (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b); (void)hipStreamSynchronize(c); } }
    } guard{fp.s_in, c->stream, fp.s_out, submit};
    TimerSpan span(c->raft->timer);
    PB_TRY(fp.grow(0, (size_t)(cp + 1) * fpx, !pin_in));
    if (flow_out) PB_TRY(fp.grow(1, (size_t)cp * dirs * px * 8, !pin_f));
    if (rgb_out) PB_TRY(fp.grow(2, (size_t)cp * dirs * px * 3, !pin_r));
    PB_TRY(fp.grow(3, (size_t)cp * dirs * 4 + 256, true));
    if (mask_out) PB_TRY(fp.grow(4, (size_t)cp * dirs * px, !pin_k));
    const unsigned base = fp.seq;
    fp.seq += (unsigned)chunks;
    auto finish = [&](int i) -> int {           // results of chunk i: pinned staging -> caller (page-locked caller buffers were written directly)
        const int slot = (int)((base + i) & 1), p0 = i * cp, m = std::min(cp, pairs - p0);
        PB_HIP(hipEventSynchronize(fp.ev_d2h[slot]));
        if (flow_out && !pin_f) memcpy(flow_out + (size_t)p0 * dirs * px * 2, fp.h[1][slot], (size_t)m * dirs * px * 8);
        if (rgb_out && !pin_r) memcpy(rgb_out + (size_t)p0 * dirs * px * 3, fp.h[2][slot], (size_t)m * dirs * px * 3);
        if (maxdisp_out) memcpy(maxdisp_out + (size_t)p0 * dirs, fp.h[3][slot], (size_t)m * dirs * 4);
        if (mask_out && !pin_k) memcpy(mask_out + (size_t)p0 * dirs * px, fp.h[4][slot], (size_t)m * dirs * px);
        return 0;
    };
    for (int i = 0; i < chunks; ++i) {
        const int slot = (int)((base + i) & 1), p0 = i * cp, m = std::min(cp, pairs - p0);
        if (!submit && i >= 2) PB_TRY(finish(i - 2));       // frees this slot's staging and (through ev_d2h) its device buffers
        PB_HIP(hipStreamWaitEvent(fp.s_in, fp.ev_comp[slot], 0));      // the compute that last read this input slot (both forms: a submission may still be in flight)
        const void *src = frames + (size_t)p0 * fpx;
        if (!pin_in) { memcpy(fp.h[0][slot], src, (size_t)(m + 1) * fpx); src = fp.h[0][slot]; }
        PB_HIP(hipMemcpyAsync(fp.d[0][slot], src, (size_t)(m + 1) * fpx, hipMemcpyHostToDevice, fp.s_in));
        PB_HIP(hipEventRecord(fp.ev_h2d[slot], fp.s_in));
        PB_HIP(hipStreamWaitEvent(c->stream, fp.ev_h2d[slot], 0));
        PB_HIP(hipStreamWaitEvent(c->stream, fp.ev_d2h[slot], 0));     // the D2H that last emptied this output slot
        PB_TRY(c->raft->infer((const uint8_t *)fp.d[0][slot], m + 1, H, W, scale, iters, backward, (float *)(flow_out ? fp.d[1][slot] : nullptr),
                              (uint8_t *)(rgb_out ? fp.d[2][slot] : nullptr), (float *)fp.d[3][slot], (uint8_t *)(mask_out ? fp.d[4][slot] : nullptr),
                              alpha1, alpha2));
        PB_HIP(hipEventRecord(fp.ev_comp[slot], c->stream));
        PB_HIP(hipStreamWaitEvent(fp.s_out, fp.ev_comp[slot], 0));
        if (flow_out) PB_HIP(hipMemcpyAsync(pin_f ? (void *)(flow_out + (size_t)p0 * dirs * px * 2) : fp.h[1][slot], fp.d[1][slot], (size_t)m * dirs * px * 8, hipMemcpyDeviceToHost, fp.s_out));
        if (rgb_out) PB_HIP(hipMemcpyAsync(pin_r ? (void *)(rgb_out + (size_t)p0 * dirs * px * 3) : fp.h[2][slot], fp.d[2][slot], (size_t)m * dirs * px * 3, hipMemcpyDeviceToHost, fp.s_out));
        if (submit) { if (maxdisp_out) PB_HIP(hipMemcpyAsync(maxdisp_out + (size_t)p0 * dirs, fp.d[3][slot], (size_t)m * dirs * 4, hipMemcpyDeviceToHost, fp.s_out)); }
        else PB_HIP(hipMemcpyAsync(fp.h[3][slot], fp.d[3][slot], (size_t)m * dirs * 4, hipMemcpyDeviceToHost, fp.s_out));
        if (mask_out) PB_HIP(hipMemcpyAsync(pin_k ? (void *)(mask_out + (size_t)p0 * dirs * px) : fp.h[4][slot], fp.d[4][slot], (size_t)m * dirs * px, hipMemcpyDeviceToHost, fp.s_out));
        PB_HIP(hipEventRecord(fp.ev_d2h[slot], fp.s_out));
    }
    if (submit) {
        hipEvent_t done;
        PB_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        PB_HIP(hipEventRecord(done, fp.s_out));
        c->pending.push_back(done);
        guard.armed = false;
        return 0;
    }
    for (int i = std::max(0, chunks - 2); i < chunks; ++i) PB_TRY(finish(i));
    return 0;
}

int pb_flow_submit_sequence(pb_ctx *c, const uint8_t *frames, int F, int H, int W, float scale, int iters, int backward,
                            float *flow_out, uint8_t *rgb_out, float *maxdisp_out) {
    PB_CHECK(c && c->raft, PB_ERR_STATE, "ctx has no flow_raft band");
    PB_CHECK(frames && F >= 2 && H > 0 && W > 0, PB_ERR_ARG, "flow submit: bad arguments");
    return flow_host_pipeline(c, frames, F, H, W, scale, iters, backward, flow_out, rgb_out, maxdisp_out, nullptr, 0.05f, 0.5f, true);
}

int pb_flow_infer_sequence(pb_ctx *c, const uint8_t *frames, int F, int H, int W, float scale, int iters, int backward,
                           float *flow_out, uint8_t *rgb_out, float *maxdisp_out) {
    PB_CHECK(c && c->raft, PB_ERR_STATE, "ctx has no flow_raft band");
    PB_CHECK(frames && F >= 2 && H > 0 && W > 0, PB_ERR_ARG, "flow infer: bad arguments");
    return flow_host_pipeline(c, frames, F, H, W, scale, iters, backward, flow_out, rgb_out, maxdisp_out, nullptr, 0.05f, 0.5f);
}

int pb_flow_infer_sequence_masks_dev(pb_ctx *c, const uint8_t *frames, int F, int H, int W, float scale, int iters,
                                     float alpha1, float alpha2, float *flow_out, uint8_t *rgb_out, float *maxdisp_out,
                                     uint8_t *mask_out) {
    PB_CHECK(c && c->raft, PB_ERR_STATE, "ctx has no flow_raft band");
    PB_CHECK(mask_out, PB_ERR_ARG, "flow masks: mask_out is NULL");
    return c->raft->infer(frames, F, H, W, scale, iters, 1, flow_out, rgb_out, maxdisp_out, mask_out, alpha1, alpha2);
}

int pb_flow_infer_sequence_masks(pb_ctx *c, const uint8_t *frames, int F, int H, int W, float scale, int iters, float alpha1,
                                 float alpha2, float *flow_out, uint8_t *rgb_out, float *maxdisp_out, uint8_t *mask_out) {
    PB_CHECK(c && c->raft, PB_ERR_STATE, "ctx has no flow_raft band");
    PB_CHECK(frames && F >= 2 && H > 0 && W > 0 && mask_out, PB_ERR_ARG, "flow masks: bad arguments");
    return flow_host_pipeline(c, frames, F, H, W, scale, iters, 1, flow_out, rgb_out, maxdisp_out, mask_out, alpha1, alpha2);
}

int pb_flow_fwdbwd_mask(pb_ctx *c, const float *flows, int n, int sh, int sw, float alpha1, float alpha2, uint8_t *mask_out) {
    PB_CHECK(c && flows && mask_out && n > 0 && sh > 0 && sw > 0, PB_ERR_ARG, "fwdbwd_mask: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    const size_t px = (size_t)n * 2 * sh * sw;
    DevMem dO, dK;
    PB_TRY(dO.alloc(px * 8));
    PB_TRY(dK.alloc(px));
    PB_HIP(hipMemcpy(dO.p, flows, px * 8, hipMemcpyHostToDevice));
    PB_TRY(launch_fwdbwd_mask(c->stream, dO.as<float>(), n, sh, sw, alpha1, alpha2, dK.as<uint8_t>()));
    PB_HIP(hipStreamSynchronize(c->stream));
    PB_HIP(hipMemcpy(mask_out, dK.p, px, hipMemcpyDeviceToHost));
    return 0;
}

int pb_mask_net_size(const pb_mask_cfg *cfg, int H, int W, int *nh, int *nw, int *Hp, int *Wp) {
    PB_CHECK(cfg && H > 0 && W > 0 && nh && nw && Hp && Wp, PB_ERR_ARG, "mask_net_size: bad arguments");
    MaskEngine::net_size(*cfg, H, W, nh, nw, Hp, Wp);
    return 0;
}

int pb_mask_infer_batch_dev(pb_ctx *c, const uint8_t *frames, int n, int H, int W, float confidence, const int32_t *keep,
                            int n_keep, uint8_t *mask_out) {
    PB_CHECK(c && c->mask, PB_ERR_STATE, "ctx has no mask_mmdet band");
    return c->mask->infer(frames, n, H, W, confidence, keep, n_keep, mask_out);
}

// Host-pointer variant (reference loop bands/mask_mmdet.py:131-154: a decoded frame per iteration): frames of every chunk go to HBM on a copy
// stream up front, chunk i's kernels wait for chunk i's copy, and its id images (with the --sdf channel when tables are set) return on a second
// copy stream while chunk i + 1 runs; device buffers, pinned staging and streams are kept on the ctx (rounds 1-5: two hipMalloc / hipFree pairs
// and two exposed whole-batch copies per call).  Page-locked caller buffers are used directly.
int pb_mask_infer_batch(pb_ctx *c, const uint8_t *frames, int n, int H, int W, float confidence, const int32_t *keep,
                        int n_keep, uint8_t *mask_out) {
    PB_CHECK(c && c->mask, PB_ERR_STATE, "ctx has no mask_mmdet band");
    PB_CHECK(frames && mask_out && n > 0 && H > 0 && W > 0, PB_ERR_ARG, "mask infer: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    const size_t fb = (size_t)H * W * 3, bytes = (size_t)n * fb;
    const int cf = c->mask->frames_per_chunk(), chunks = (n + cf - 1) / cf;
    const bool pin_in = pb_is_pinned(frames, bytes), pin_out = pb_is_pinned(mask_out, bytes);
    MaskPipe &mp = c->mpipe;
    PB_TRY(mp.ensure(bytes, !pin_in, !pin_out, chunks));
    PipeDrain drain{mp.s_in, c->stream, mp.s_out};              // no exit, error or not, leaves a copy into caller memory in flight
    for (int i = 0; i < chunks; ++i) {
        const size_t off = (size_t)i * cf * fb, len = (size_t)std::min(cf, n - i * cf) * fb;
        const void *src = frames + off;
        if (!pin_in) { memcpy((char *)mp.h_in + off, src, len); src = (char *)mp.h_in + off; }
        PB_HIP(hipMemcpyAsync((char *)mp.d_in + off, src, len, hipMemcpyHostToDevice, mp.s_in));
        PB_HIP(hipEventRecord(mp.ev_in[i], mp.s_in));
    }
    hipStream_t st = c->stream;
    char *dst = pin_out ? (char *)mask_out : (char *)mp.h_out;
    c->mask->chunk_begin = [&mp, st, cf](int first, int) -> int {
        PB_HIP(hipStreamWaitEvent(st, mp.ev_in[first / cf], 0));
        return 0;
    };
    c->mask->chunk_end = [&mp, st, cf, fb, dst](int first, int m) -> int {
        const size_t off = (size_t)first * fb;
        PB_HIP(hipEventRecord(mp.ev_done[first / cf], st));
        PB_HIP(hipStreamWaitEvent(mp.s_out, mp.ev_done[first / cf], 0));
        PB_HIP(hipMemcpyAsync(dst + off, (char *)mp.d_out + off, (size_t)m * fb, hipMemcpyDeviceToHost, mp.s_out));
        return 0;
    };
    const int r = c->mask->infer((const uint8_t *)mp.d_in, n, H, W, confidence, keep, n_keep, (uint8_t *)mp.d_out);
    c->mask->chunk_begin = nullptr; c->mask->chunk_end = nullptr;     // (an argument error returns before infer() consumes them)
    if (r) return r;
    PB_HIP(hipStreamSynchronize(mp.s_out));
    if (!pin_out) memcpy(mask_out, mp.h_out, bytes);
    return 0;
}

int pb_mask_set_sdf(pb_ctx *c, const uint8_t *tab_out, const uint8_t *tab_in, int n_tab) {
    PB_CHECK(c && c->mask, PB_ERR_STATE, "ctx has no mask_mmdet band");
    PB_CHECK(n_tab <= 0 || (tab_out && tab_in && n_tab >= 2 && n_tab <= 4226), PB_ERR_ARG, "mask set_sdf: bad tables (n_tab = %d)", n_tab);
    return c->mask->set_sdf(tab_out, tab_in, n_tab);
}

int pb_mask_sdf_green_dev(pb_ctx *c, uint8_t *masks, int n, int H, int W) {
    PB_CHECK(c && c->mask, PB_ERR_STATE, "ctx has no mask_mmdet band");
    return c->mask->sdf_green(masks, n, H, W);
}

int pb_mask_sdf_green(pb_ctx *c, uint8_t *masks, int n, int H, int W) {
    PB_CHECK(c && c->mask, PB_ERR_STATE, "ctx has no mask_mmdet band");
    PB_CHECK(masks && n > 0 && H > 0 && W > 0, PB_ERR_ARG, "mask sdf: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    const size_t bytes = (size_t)n * H * W * 3;
    DevMem d;
    PB_TRY(d.alloc(bytes));
    PB_HIP(hipMemcpy(d.p, masks, bytes, hipMemcpyHostToDevice));
    PB_TRY(c->mask->sdf_green(d.as<uint8_t>(), n, H, W));
    PB_HIP(hipStreamSynchronize(c->stream));
    PB_HIP(hipMemcpy(masks, d.p, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int pb_mask_get_instances(pb_ctx *c, int frame, int cap, float *scores_out, int32_t *labels_out, uint8_t *masks_out,
                          int32_t *candidates_out) {
    PB_CHECK(c && c->mask, PB_ERR_STATE, "ctx has no mask_mmdet band");
    const auto &res = c->mask->results();
    PB_CHECK(frame >= 0 && frame < (int)res.size(), PB_ERR_ARG, "mask instances: frame %d of %zu", frame, res.size());
    const auto &r = res[frame];
    const int cnt = std::min<int>(cap, (int)r.scores.size());
    for (int i = 0; i < cnt; ++i) {
        if (scores_out) scores_out[i] = r.scores[i];
        if (labels_out) labels_out[i] = r.labels[i];
    }
    if (masks_out && cnt > 0) {
        PB_CHECK(!r.masks.empty(), PB_ERR_STATE, "instance masks are kept only with pb_set_profiling(ctx, 2)");
        memcpy(masks_out, r.masks.data(), r.masks.size() / r.scores.size() * cnt);
    }
    if (candidates_out) *candidates_out = r.candidates;
    return (int)r.scores.size();
}

int64_t pb_mask_get_stage(pb_ctx *c, const char *name, float *out, int64_t cap, int64_t shape_out[4]) {
    PB_CHECK(c && c->mask && name && out && shape_out, PB_ERR_ARG, "mask get_stage: bad arguments");
    return c->mask->get_stage(name, out, cap, shape_out);
}

int pb_flow_set_inference_size(pb_ctx *c, int h, int w) {
    PB_CHECK(c && c->raft, PB_ERR_STATE, "ctx has no flow band");
    GmflowEngine *g = dynamic_cast<GmflowEngine *>(c->raft);
    PB_CHECK(g, PB_ERR_STATE, "--inference_size is a flow_gmflow option");
    return g->set_inference_size(h, w);
}

int64_t pb_flow_get_stage(pb_ctx *c, const char *name, float *out, int64_t cap, int64_t shape_out[4]) {
    PB_CHECK(c && c->raft && name && out && shape_out, PB_ERR_ARG, "flow get_stage: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    return c->raft->get_stage(name, out, cap, shape_out);
}

int pb_set_option(pb_ctx *c, const char *key, int value) {
    PB_CHECK(c && key, PB_ERR_ARG, "set_option: bad arguments");
    int *g = c->depth ? &c->depth->gemm_tile : &c->gemm_tile;
    int *v = c->depth ? &c->depth->conv_tile : (c->raft ? &c->raft->conv_tile : (c->mask ? &c->mask->conv_tile : &c->conv_tile));
    if (!strcmp(key, "gemm_tile")) *g = value;
    else if (!strcmp(key, "conv_tile")) *v = value;
    else if (!strcmp(key, "host_chunk")) c->host_chunk = value;
    else if (!strcmp(key, "op_splitk")) c->op_splitk = value;
    else if (!strcmp(key, "tile_n96")) pb_gemm_set_n96(value);         // process-wide (gemm.h): A/B of the 128 x 96 tile inside one test process
    else PB_CHECK(false, PB_ERR_ARG, "unknown option '%s'", key);
    return 0;
}

int pb_get_kernel_stats(pb_ctx *c, pb_kernel_stat *out, int cap) {
    PB_CHECK(c && (c->depth || c->raft || c->mask) && out, PB_ERR_STATE, "ctx has no band");
    return c->depth ? c->depth->stats(out, cap) : (c->raft ? c->raft->stats(out, cap) : c->mask->stats(out, cap));
}

int pb_dev_alloc(pb_ctx *c, void **ptr, size_t bytes) {
    PB_CHECK(c && ptr, PB_ERR_ARG, "dev_alloc: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    hipError_t e = hipMalloc(ptr, bytes);
    PB_CHECK(e == hipSuccess, PB_ERR_MEMORY, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return 0;
}
int pb_dev_free(pb_ctx *c, void *ptr) {
    PB_CHECK(c, PB_ERR_ARG, "null ctx");
    PB_HIP(hipSetDevice(c->device));
    PB_HIP(hipFree(ptr));
    return 0;
}
int pb_memcpy_h2d(pb_ctx *c, void *dst, const void *src, size_t bytes) {
    PB_CHECK(c, PB_ERR_ARG, "null ctx");
    PB_HIP(hipSetDevice(c->device));
    PB_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
int pb_memcpy_d2h(pb_ctx *c, void *dst, const void *src, size_t bytes) {
    PB_CHECK(c, PB_ERR_ARG, "null ctx");
    PB_HIP(hipSetDevice(c->device));
    PB_HIP(hipStreamSynchronize(c->stream));
    PB_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

// ---- single-kernel entry points ---------------------------------------------------------------
int pb_op_gemm(pb_ctx *c, const float *A, const float *W, const float *bias, float *C, int M, int N, int K, int act,
               int tile) {
    PB_CHECK(c && A && W && C && M > 0 && N > 0 && K > 0 && N % 8 == 0, PB_ERR_ARG, "op_gemm: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    const int Kp = (int)round_up(K, 64), Np = (int)round_up(N, 256);
    const int64_t Mp = round_up(M, 256);
    DevMem a32, w32, b32, a16, w16, c16, c32;
    PB_TRY(a32.alloc((size_t)M * K * 4)); PB_TRY(w32.alloc((size_t)N * K * 4));
    PB_TRY(a16.alloc((size_t)Mp * Kp * 2)); PB_TRY(w16.alloc((size_t)Np * Kp * 2));
    PB_TRY(c16.alloc((size_t)Mp * N * 2)); PB_TRY(c32.alloc((size_t)M * N * 4));
    PB_HIP(hipMemcpy(a32.p, A, (size_t)M * K * 4, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(w32.p, W, (size_t)N * K * 4, hipMemcpyHostToDevice));
    if (bias) {
        PB_TRY(b32.alloc((size_t)N * 4));
        PB_HIP(hipMemcpy(b32.p, bias, (size_t)N * 4, hipMemcpyHostToDevice));
    }
    PB_TRY(launch_f32_to_f16(c->stream, a32.as<float>(), a16.as<f16>(), M, K, Kp));
    PB_TRY(launch_f32_to_f16(c->stream, w32.as<float>(), w16.as<f16>(), N, K, Kp));
    GemmArgs g;
    g.A = a16.as<f16>(); g.lda = Kp; g.W = w16.as<f16>(); g.K = Kp; g.M = M; g.N = N;
    g.bias = b32.as<float>(); g.out = c16.as<f16>(); g.ldo = N; g.act = act; g.zero = c->zero;
    DevMem skw;
    if (c->op_splitk) { PB_TRY(skw.alloc((size_t)512 * 128 * 128 * 4)); g.sk_ws = skw.as<float>(); g.sk_cap = (int64_t)512 * 128 * 128; }
    PB_TRY(launch_gemm(c->stream, A_DENSE, EPI_STD, tile, g));
    PB_TRY(launch_f16_to_f32(c->stream, c16.as<f16>(), c32.as<float>(), M, N, N));
    PB_HIP(hipStreamSynchronize(c->stream));
    PB_HIP(hipMemcpy(C, c32.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    return 0;
}

int pb_op_corr_volume(pb_ctx *c, const float *A, int M, const float *W, int N, int ldo, int guard_rows, float *out) {
    PB_CHECK(c && A && W && out && M > 0 && N > 0 && N % 8 == 0 && ldo >= N && guard_rows >= 0, PB_ERR_ARG, "op_corr_volume: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    const int64_t rows = (int64_t)M + guard_rows, Np = round_up(N, 64);
    DevMem a32, w32, a16, w16, o16, o32;
    PB_TRY(a32.alloc((size_t)M * 256 * 4)); PB_TRY(w32.alloc((size_t)N * 256 * 4));
    PB_TRY(a16.alloc((size_t)M * 256 * 2)); PB_TRY(w16.alloc((size_t)Np * 256 * 2));
    PB_TRY(o16.alloc((size_t)rows * ldo * 2)); PB_TRY(o32.alloc((size_t)rows * ldo * 4));
    PB_HIP(hipMemcpy(a32.p, A, (size_t)M * 256 * 4, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(w32.p, W, (size_t)N * 256 * 4, hipMemcpyHostToDevice));
    PB_HIP(hipMemsetAsync(w16.p, 0, (size_t)Np * 256 * 2, c->stream));
    PB_HIP(hipMemsetAsync(o16.p, 0x7e, (size_t)rows * ldo * 2, c->stream));          // 0x7e7e: a NaN in fp16
    PB_TRY(launch_f32_to_f16(c->stream, a32.as<float>(), a16.as<f16>(), M, 256, 256));
    PB_TRY(launch_f32_to_f16(c->stream, w32.as<float>(), w16.as<f16>(), N, 256, 256));
    PB_TRY(launch_corr_volume(c->stream, a16.as<f16>(), M, w16.as<f16>(), N, (int)Np, o16.as<f16>(), ldo));
    PB_TRY(launch_f16_to_f32(c->stream, o16.as<f16>(), o32.as<float>(), rows, ldo, ldo));
    PB_HIP(hipStreamSynchronize(c->stream));
    PB_HIP(hipMemcpy(out, o32.p, (size_t)rows * ldo * 4, hipMemcpyDeviceToHost));
    return 0;
}

int pb_op_gemm_bench(pb_ctx *c, int M, int N, int K, int tile, int epi, int iters, double *ms_out) {
    PB_CHECK(c && M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 8 == 0 && iters > 0 && ms_out, PB_ERR_ARG, "gemm_bench: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    DevMem a, w, o, r, b;
    const int64_t Mp = round_up(M, 256), Np = round_up(N, 256);
    PB_TRY(a.alloc((size_t)Mp * K * 2)); PB_TRY(w.alloc((size_t)Np * K * 2)); PB_TRY(o.alloc((size_t)Mp * N * 2));
    PB_TRY(r.alloc((size_t)Mp * N * 4)); PB_TRY(b.alloc((size_t)Np * 4));
    PB_TRY(launch_fill_random_f16(c->stream, a.as<f16>(), (int64_t)M * K, 1u, 1.f));
    PB_TRY(launch_fill_random_f16(c->stream, w.as<f16>(), (int64_t)N * K, 2u, 0.05f));
    GemmArgs g;
    g.A = a.as<f16>(); g.lda = K; g.W = w.as<f16>(); g.K = K; g.M = M; g.N = N; g.zero = c->zero; g.bias = b.as<float>();
    int e = EPI_STD, amode = A_DENSE;
    if (epi == 2) { e = EPI_RESID; g.resid = r.as<float>(); g.ldr = N; g.gamma = b.as<float>(); }
    else { g.out = o.as<f16>(); g.ldo = N; g.act = epi == 1 ? ACT_GELU : ACT_NONE; }
    if (epi >= 10) {                    // implicit-GEMM convolution over a [M / 18360, 102, 180, C] map (the RAFT update block's grid at 1080p x 0.75):
        const int taps = epi == 12 ? 5 : 9;         // 10: 3 x 3 tap-major, 11: 3 x 3 slice-major, 12: 1 x 5 tap-major; K = taps * C
        PB_CHECK(M % 18360 == 0 && K % (taps * 64) == 0, PB_ERR_ARG, "gemm_bench: conv modes need M = B * 102 * 180 and K = taps * C");
        amode = A_CONV;
        g.cH = g.cOH = 102; g.cW = g.cOW = 180; g.cC = K / taps; g.cStride = 1;
        if (epi == 12) { g.cKW = 5; g.cPad = 0; g.cPadX = 2; g.cKH = 1; }
        else { g.cKW = 3; g.cPad = 1; g.cKH = 3; g.cTapInner = epi == 11; }
    }
    g.ablate = pb_env_int("PB_GEMM_ABL", 0);       // timing-only epilogue ablations (wrong results): this tool op only, never an engine launch
    hipEvent_t e0, e1;
    PB_HIP(hipEventCreate(&e0)); PB_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) PB_TRY(launch_gemm(c->stream, amode, e, tile, g));
    if (const char *dump = getenv("PB_GEMM_DBG")) {      // per-block stamps of one launch -> binary file
        const int nblk = (int)((Mp / 256) * (Np / 256));
        DevMem d;
        PB_TRY(d.alloc((size_t)nblk * 64));
        g.dbg = d.as<long long>();
        PB_TRY(launch_gemm(c->stream, amode, e, tile, g));
        PB_HIP(hipStreamSynchronize(c->stream));
        std::vector<long long> h((size_t)nblk * 8);
        PB_HIP(hipMemcpy(h.data(), d.p, h.size() * 8, hipMemcpyDeviceToHost));
        if (FILE *f = fopen(dump, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        g.dbg = nullptr;
    }
    PB_HIP(hipEventRecord(e0, c->stream));
    for (int i = 0; i < iters; ++i) PB_TRY(launch_gemm(c->stream, amode, e, tile, g));
    PB_HIP(hipEventRecord(e1, c->stream));
    PB_HIP(hipStreamSynchronize(c->stream));
    float ms = 0;
    PB_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    *ms_out = ms / iters;
    return 0;
}

int pb_op_attention_bench(pb_ctx *c, int B, int heads, int N, int variant, int iters, double *ms_out) {
    PB_CHECK(c && B > 0 && heads > 0 && N > 0 && iters > 0 && ms_out, PB_ERR_ARG, "attention_bench: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    const int ntp = (int)round_up(N, 16), D = heads * 64;
    const size_t n = (size_t)B * heads * ntp * 64;
    DevMem dq, dk, dv, dout;
    PB_TRY(dq.alloc(n * 2 + 32768)); PB_TRY(dk.alloc(n * 2 + 32768)); PB_TRY(dv.alloc(n * 2 + 32768));
    PB_TRY(dout.alloc((size_t)B * ntp * D * 2));
    PB_TRY(launch_fill_random_f16(c->stream, dq.as<f16>(), (int64_t)n, 11u, 3.f * PB_QSCALE));
    PB_TRY(launch_fill_random_f16(c->stream, dk.as<f16>(), (int64_t)n, 12u, 3.f));
    PB_TRY(launch_fill_random_f16(c->stream, dv.as<f16>(), (int64_t)n, 13u, 1.f));
    hipEvent_t e0, e1;
    PB_HIP(hipEventCreate(&e0)); PB_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) PB_TRY(launch_attention(c->stream, dq.as<f16>(), dk.as<f16>(), dv.as<f16>(), dout.as<f16>(), B, heads, ntp, N, D, variant));
    PB_HIP(hipEventRecord(e0, c->stream));
    for (int i = 0; i < iters; ++i) PB_TRY(launch_attention(c->stream, dq.as<f16>(), dk.as<f16>(), dv.as<f16>(), dout.as<f16>(), B, heads, ntp, N, D, variant));
    PB_HIP(hipEventRecord(e1, c->stream));
    PB_HIP(hipStreamSynchronize(c->stream));
    float ms = 0;
    PB_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    *ms_out = ms / iters;
    return 0;
}

int pb_op_layernorm(pb_ctx *c, const float *x, const float *g, const float *b, float *y, int rows, int D) {
    PB_CHECK(c && x && g && b && y && rows > 0, PB_ERR_ARG, "op_layernorm: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    DevMem dx, dg, db, dy, dy32;
    PB_TRY(dx.alloc((size_t)rows * D * 4)); PB_TRY(dg.alloc((size_t)D * 4)); PB_TRY(db.alloc((size_t)D * 4));
    PB_TRY(dy.alloc((size_t)rows * D * 2)); PB_TRY(dy32.alloc((size_t)rows * D * 4));
    PB_HIP(hipMemcpy(dx.p, x, (size_t)rows * D * 4, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(dg.p, g, (size_t)D * 4, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(db.p, b, (size_t)D * 4, hipMemcpyHostToDevice));
    PB_TRY(launch_layernorm(c->stream, dx.as<float>(), dg.as<float>(), db.as<float>(), dy.as<f16>(), 1, rows, rows, D,
                            1e-6f, 0));
    PB_TRY(launch_f16_to_f32(c->stream, dy.as<f16>(), dy32.as<float>(), rows, D, D));
    PB_HIP(hipStreamSynchronize(c->stream));
    PB_HIP(hipMemcpy(y, dy32.p, (size_t)rows * D * 4, hipMemcpyDeviceToHost));
    return 0;
}

int pb_op_attention(pb_ctx *c, const float *q, const float *k, const float *v, float *o, int B, int heads, int N) {
    PB_CHECK(c && q && k && v && o && B > 0 && heads > 0 && N > 0, PB_ERR_ARG, "op_attention: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    const int ntp = (int)round_up(N, 16), D = heads * 64;
    const size_t bh = (size_t)B * heads;
    // host-side relayout to the engine's Q / K / Vt buffers (q pre-scaled by 64^-0.5 like the qkv epilogue)
    std::vector<f16> hq(bh * ntp * 64, (f16)0.f), hk(bh * ntp * 64, (f16)0.f), hv(bh * 64 * ntp, (f16)0.f);
    for (size_t i = 0; i < bh; ++i)
        for (int t = 0; t < N; ++t)
            for (int d = 0; d < 64; ++d) {
                const size_t s = (i * N + t) * 64 + d;
                hq[(i * ntp + t) * 64 + d] = (f16)(q[s] * PB_QSCALE);
                hk[(i * ntp + t) * 64 + d] = (f16)k[s];
                hv[(i * 64 + d) * ntp + t] = (f16)v[s];
            }
    DevMem dq, dk, dv, dout, dout32;
    const size_t slack = 32768;
    PB_TRY(dq.alloc(hq.size() * 2 + slack)); PB_TRY(dk.alloc(hk.size() * 2 + slack)); PB_TRY(dv.alloc(hv.size() * 2 + slack));
    PB_TRY(dout.alloc((size_t)B * ntp * D * 2)); PB_TRY(dout32.alloc((size_t)B * ntp * D * 4));
    PB_HIP(hipMemcpy(dq.p, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(dk.p, hk.data(), hk.size() * 2, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(dv.p, hv.data(), hv.size() * 2, hipMemcpyHostToDevice));
    PB_TRY(launch_attention(c->stream, dq.as<f16>(), dk.as<f16>(), dv.as<f16>(), dout.as<f16>(), B, heads, ntp, N, D));
    PB_TRY(launch_f16_to_f32(c->stream, dout.as<f16>(), dout32.as<float>(), (int64_t)B * ntp, D, D));
    PB_HIP(hipStreamSynchronize(c->stream));
    std::vector<float> ho((size_t)B * ntp * D);
    PB_HIP(hipMemcpy(ho.data(), dout32.p, ho.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < heads; ++h)
            for (int t = 0; t < N; ++t)
                for (int d = 0; d < 64; ++d)
                    o[(((size_t)b * heads + h) * N + t) * 64 + d] = ho[((size_t)b * ntp + t) * D + h * 64 + d];
    return 0;
}

int pb_op_attention128(pb_ctx *c, const float *q, const float *k, const float *v, const int8_t *region, float *o, int B, int L) {
    PB_CHECK(c && q && k && v && o && B > 0 && L > 0, PB_ERR_ARG, "op_attention128: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    const int ldv = (int)round_up(L, 32);
    const size_t n = (size_t)B * L * 128;
    std::vector<f16> hq(n), hk(n), hv((size_t)B * 128 * ldv, (f16)0.f);
    for (size_t i = 0; i < n; ++i) { hq[i] = (f16)q[i]; hk[i] = (f16)k[i]; }
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < L; ++t)
            for (int d = 0; d < 128; ++d) hv[((size_t)b * 128 + d) * ldv + t] = (f16)v[((size_t)b * L + t) * 128 + d];
    DevMem dq, dk, dv, dr, dout;
    PB_TRY(dq.alloc(n * 2)); PB_TRY(dk.alloc(n * 2)); PB_TRY(dv.alloc(hv.size() * 2)); PB_TRY(dout.alloc(n * 4));
    PB_HIP(hipMemcpy(dq.p, hq.data(), n * 2, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(dk.p, hk.data(), n * 2, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(dv.p, hv.data(), hv.size() * 2, hipMemcpyHostToDevice));
    if (region) {
        PB_TRY(dr.alloc((size_t)B * L));
        PB_HIP(hipMemcpy(dr.p, region, (size_t)B * L, hipMemcpyHostToDevice));
    }
    PB_TRY(launch_attention128(c->stream, dq.as<f16>(), dk.as<f16>(), dv.as<f16>(), region ? dr.as<int8_t>() : nullptr, dout.as<float>(), B, L, ldv));
    PB_HIP(hipStreamSynchronize(c->stream));
    PB_HIP(hipMemcpy(o, dout.p, n * 4, hipMemcpyDeviceToHost));
    return 0;
}

int pb_op_attention128_split(pb_ctx *c, const float *q, const float *k, const float *v, const int8_t *region, int nreg, float *o, int B, int L,
                             int vcols, int kxor) {
    PB_CHECK(c && q && k && v && o && B > 0 && L > 0 && (vcols == 128 || vcols == 32), PB_ERR_ARG, "op_attention128_split: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    const int ldv = (int)round_up(L, 32);
    const size_t n = (size_t)B * L * 128;
    std::vector<f16> hq(2 * n), hk(2 * n), hv((size_t)B * 2 * vcols * ldv, (f16)0.f);
    auto split = [](float x, f16 &hi, f16 &lo) { hi = (f16)x; lo = (f16)(x - (float)hi); };
    for (size_t r = 0; r < (size_t)B * L; ++r)
        for (int d = 0; d < 128; ++d) {
            split(q[r * 128 + d], hq[r * 256 + d], hq[r * 256 + 128 + d]);
            split(k[r * 128 + d], hk[r * 256 + d], hk[r * 256 + 128 + d]);
        }
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < L; ++t)
            for (int d = 0; d < vcols; ++d)
                split(v[((size_t)b * L + t) * vcols + d], hv[(((size_t)b * 2 + 0) * vcols + d) * ldv + t], hv[(((size_t)b * 2 + 1) * vcols + d) * ldv + t]);
    DevMem dq, dk, dv, dr, dout;
    PB_TRY(dq.alloc(2 * n * 2)); PB_TRY(dk.alloc(2 * n * 2)); PB_TRY(dv.alloc(hv.size() * 2)); PB_TRY(dout.alloc((size_t)B * L * vcols * 4));
    PB_HIP(hipMemcpy(dq.p, hq.data(), 2 * n * 2, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(dk.p, hk.data(), 2 * n * 2, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(dv.p, hv.data(), hv.size() * 2, hipMemcpyHostToDevice));
    if (region) {
        PB_TRY(dr.alloc((size_t)nreg * L));
        PB_HIP(hipMemcpy(dr.p, region, (size_t)nreg * L, hipMemcpyHostToDevice));
    }
    Attn128Args a;
    a.Q = dq.as<f16>(); a.K = dk.as<f16>(); a.Vt = dv.as<f16>(); a.region = region ? dr.as<int8_t>() : nullptr; a.nreg = nreg;
    a.O = dout.as<float>(); a.B = B; a.L = L; a.ldv = ldv; a.split = 1; a.vcols = vcols; a.kxor = kxor;
    PB_TRY(launch_attention128x(c->stream, a));
    PB_HIP(hipStreamSynchronize(c->stream));
    PB_HIP(hipMemcpy(o, dout.p, (size_t)B * L * vcols * 4, hipMemcpyDeviceToHost));
    return 0;
}

int pb_op_conv2d(pb_ctx *c, const float *x, const float *w, const float *bias, float *y, int B, int Ci, int H, int W,
                 int Co, int ks, int stride, int pad, int relu_in, int relu_out) {
    PB_CHECK(c && x && w && y && Co % 8 == 0 && (ks == 1 || ks == 3), PB_ERR_ARG, "op_conv2d: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    const int cip = (int)round_up(Ci, 64), cop = (int)round_up(Co, 64), K = ks * ks * cip;
    const int OH = (H + 2 * pad - ks) / stride + 1, OW = (W + 2 * pad - ks) / stride + 1;
    std::vector<f16> hw((size_t)round_up(Co, 256) * K, (f16)0.f);
    for (int o = 0; o < Co; ++o)
        for (int ci = 0; ci < Ci; ++ci)
            for (int t = 0; t < ks * ks; ++t)
                hw[(size_t)o * K + t * cip + ci] = (f16)w[((size_t)o * Ci + ci) * ks * ks + t];
    DevMem dx32, dx, dw, db, dy, dy32;
    PB_TRY(dx32.alloc((size_t)B * Ci * H * W * 4)); PB_TRY(dx.alloc((size_t)round_up((int64_t)B * H * W, 256) * cip * 2));
    PB_TRY(dw.alloc(hw.size() * 2)); PB_TRY(dy.alloc((size_t)round_up((int64_t)B * OH * OW, 256) * cop * 2));
    PB_TRY(dy32.alloc((size_t)B * Co * OH * OW * 4));
    PB_HIP(hipMemcpy(dx32.p, x, (size_t)B * Ci * H * W * 4, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(dw.p, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    if (bias) {
        PB_TRY(db.alloc((size_t)Co * 4));
        PB_HIP(hipMemcpy(db.p, bias, (size_t)Co * 4, hipMemcpyHostToDevice));
    }
    PB_TRY(launch_nchw_f32_to_nhwc_f16(c->stream, dx32.as<float>(), dx.as<f16>(), B, Ci, H, W, cip, relu_in));
    GemmArgs g;
    g.A = dx.as<f16>(); g.W = dw.as<f16>(); g.K = K; g.M = B * OH * OW; g.N = Co;
    g.cH = H; g.cW = W; g.cC = cip; g.cOH = OH; g.cOW = OW; g.cKW = ks; g.cStride = stride; g.cPad = pad;
    g.zero = c->zero; g.bias = db.as<float>(); g.out = dy.as<f16>(); g.ldo = cop; g.act = relu_out ? ACT_RELU : ACT_NONE;
    DevMem skw;
    if (c->op_splitk) { PB_TRY(skw.alloc((size_t)512 * 128 * 128 * 4)); g.sk_ws = skw.as<float>(); g.sk_cap = (int64_t)512 * 128 * 128; }
    PB_TRY(launch_gemm(c->stream, A_CONV, EPI_STD, c->conv_tile ? c->conv_tile : TILE_128, g));
    PB_TRY(launch_nhwc_f16_to_nchw_f32(c->stream, dy.as<f16>(), dy32.as<float>(), B, Co, OH, OW, cop));
    PB_HIP(hipStreamSynchronize(c->stream));
    PB_HIP(hipMemcpy(y, dy32.p, (size_t)B * Co * OH * OW * 4, hipMemcpyDeviceToHost));
    return 0;
}

int pb_op_bilinear(pb_ctx *c, const float *x, float *y, int B, int C, int H, int W, int OH, int OW, int align) {
    PB_CHECK(c && x && y && C % 8 == 0, PB_ERR_ARG, "op_bilinear: bad arguments (C %% 8)");
    PB_HIP(hipSetDevice(c->device));
    DevMem dx32, dx, dy, dy32;
    PB_TRY(dx32.alloc((size_t)B * C * H * W * 4)); PB_TRY(dx.alloc((size_t)B * C * H * W * 2));
    PB_TRY(dy.alloc((size_t)B * C * OH * OW * 2)); PB_TRY(dy32.alloc((size_t)B * C * OH * OW * 4));
    PB_HIP(hipMemcpy(dx32.p, x, (size_t)B * C * H * W * 4, hipMemcpyHostToDevice));
    PB_TRY(launch_nchw_f32_to_nhwc_f16(c->stream, dx32.as<float>(), dx.as<f16>(), B, C, H, W, C, 0));
    PB_TRY(launch_bilinear_nhwc(c->stream, dx.as<f16>(), dy.as<f16>(), B, H, W, OH, OW, C, C, align));
    PB_TRY(launch_nhwc_f16_to_nchw_f32(c->stream, dy.as<f16>(), dy32.as<float>(), B, C, OH, OW, C));
    PB_HIP(hipStreamSynchronize(c->stream));
    PB_HIP(hipMemcpy(y, dy32.p, (size_t)B * C * OH * OW * 4, hipMemcpyDeviceToHost));
    return 0;
}

int pb_op_preprocess(pb_ctx *c, const uint8_t *frame, int H, int W, float *out, int net_h, int net_w) {
    PB_CHECK(c && frame && out, PB_ERR_ARG, "op_preprocess: bad arguments");
    int nh, nw;
    PB_TRY(pb_depth_net_size(H, W, &nh, &nw));
    PB_CHECK(nh == net_h && nw == net_w, PB_ERR_ARG, "op_preprocess: net size is %dx%d, caller passed %dx%d", nh, nw,
             net_h, net_w);
    PB_HIP(hipSetDevice(c->device));
    std::vector<int> xi((size_t)nw * 4), yi((size_t)nh * 4);
    std::vector<float> xw((size_t)nw * 4), yw((size_t)nh * 4);
    pb_cubic_taps(W, nw, xi.data(), xw.data());
    pb_cubic_taps(H, nh, yi.data(), yw.data());
    DevMem df, dxi, dxw, dyi, dyw, dout;
    PB_TRY(df.alloc((size_t)H * W * 3)); PB_TRY(dxi.alloc(xi.size() * 4)); PB_TRY(dxw.alloc(xw.size() * 4));
    PB_TRY(dyi.alloc(yi.size() * 4)); PB_TRY(dyw.alloc(yw.size() * 4)); PB_TRY(dout.alloc((size_t)3 * nh * nw * 4));
    PB_HIP(hipMemcpy(df.p, frame, (size_t)H * W * 3, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(dxi.p, xi.data(), xi.size() * 4, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(dxw.p, xw.data(), xw.size() * 4, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(dyi.p, yi.data(), yi.size() * 4, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(dyw.p, yw.data(), yw.size() * 4, hipMemcpyHostToDevice));
    PB_TRY(launch_preprocess(c->stream, df.as<uint8_t>(), 1, H, W, nh, nw, dxi.as<int>(), dxw.as<float>(), dyi.as<int>(),
                             dyw.as<float>(), nullptr, 640, dout.as<float>()));
    PB_HIP(hipStreamSynchronize(c->stream));
    PB_HIP(hipMemcpy(out, dout.p, (size_t)3 * nh * nw * 4, hipMemcpyDeviceToHost));
    return 0;
}

int pb_op_encode_depth(pb_ctx *c, const float *depth, int n, int H, int W, int flip, uint8_t *rgb, float *mn, float *mx) {
    PB_CHECK(c && depth && n > 0, PB_ERR_ARG, "op_encode_depth: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    const size_t px = (size_t)H * W;
    DevMem dd, dr, dm, dmn;
    PB_TRY(dd.alloc(n * px * 4)); PB_TRY(dr.alloc(n * px * 3)); PB_TRY(dm.alloc((size_t)n * 8)); PB_TRY(dmn.alloc((size_t)n * 8));
    PB_HIP(hipMemcpy(dd.p, depth, n * px * 4, hipMemcpyHostToDevice));
    PB_TRY(launch_init_minmax(c->stream, dm.as<unsigned>(), n));
    PB_TRY(launch_minmax_only(c->stream, dd.as<float>(), n, (int64_t)px, dm.as<unsigned>()));
    PB_TRY(launch_heat_encode(c->stream, dd.as<float>(), n, H, W, dm.as<unsigned>(), flip, dr.as<uint8_t>(),
                              dmn.as<float>(), dmn.as<float>() + n));
    PB_HIP(hipStreamSynchronize(c->stream));
    if (rgb) PB_HIP(hipMemcpy(rgb, dr.p, n * px * 3, hipMemcpyDeviceToHost));
    if (mn) PB_HIP(hipMemcpy(mn, dmn.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (mx) PB_HIP(hipMemcpy(mx, dmn.as<float>() + n, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

int pb_depth_encode_still(pb_ctx *c, const float *depth, int H, int W, int flip, int encode_range, uint8_t *rgb_out, float *min_out,
                          float *max_out) {
    PB_CHECK(c && depth && rgb_out && H > 0 && W > 0, PB_ERR_ARG, "depth_encode_still: bad arguments");
    PB_HIP(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lock(c->still_mu);
    const size_t px = (size_t)H * W;
    // [depth f32 | rgb u8 x 3 | quantised u8 | 64 bytes of min / max words], each part 256-byte aligned
    const size_t o_rgb = round_up(px * 4, 256), o_q = o_rgb + round_up(px * 3, 256), o_mm = o_q + round_up(px, 256), need = o_mm + 256;
    if (!c->still_stream) PB_HIP(hipStreamCreateWithFlags(&c->still_stream, hipStreamNonBlocking));
    if (need > c->still_cap) {
        if (c->still_buf) { PB_HIP(hipStreamSynchronize(c->still_stream)); PB_HIP(hipFree(c->still_buf)); c->still_buf = nullptr; c->still_cap = 0; }
        PB_HIP(hipMalloc((void **)&c->still_buf, need));
        c->still_cap = need;
    }
    hipStream_t st = c->still_stream;
    PB_HIP(hipMemsetAsync(c->still_buf + o_mm, 0, 256, st));
    PB_HIP(hipMemcpyAsync(c->still_buf, depth, px * 4, hipMemcpyHostToDevice, st));
    unsigned *mm = (unsigned *)(c->still_buf + o_mm);
    PB_TRY(launch_still_encode(st, (const float *)c->still_buf, H, W, mm, mm + 2, (uint8_t *)(c->still_buf + o_q), flip, encode_range,
                               (uint8_t *)(c->still_buf + o_rgb), (float *)(mm + 4)));
    float mnmx[2];
    PB_HIP(hipMemcpyAsync(rgb_out, c->still_buf + o_rgb, px * 3, hipMemcpyDeviceToHost, st));
    PB_HIP(hipMemcpyAsync(mnmx, mm + 4, 8, hipMemcpyDeviceToHost, st));
    PB_HIP(hipStreamSynchronize(st));
    if (min_out) *min_out = mnmx[0];
    if (max_out) *max_out = mnmx[1];
    return 0;
}

}  // extern "C"
