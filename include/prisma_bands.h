/*
 * prisma_bands.h - C ABI of libprisma_bands.so (MI355X / gfx950 "bands" engine).
 *
 * The reference (patriciogonzalezvivo/prisma) has no native boundary: a band is a Python
 * script exposing init_model()/infer() and a per-frame loop.  This header is the boundary a
 * maintainer would bind from those scripts (ctypes stub: INTEGRATION.md).  Each entry point
 * cites the reference code it replaces; paths are relative to the reference checkout.
 *
 * Conventions
 *   - every function returns 0 on success or a negative pb_status; pb_last_error() holds the
 *     message of the last failure on the calling thread.
 *   - the caller owns every host buffer; the library owns device memory, streams and events
 *     inside pb_ctx.  One pb_ctx per GPU per process; a ctx is not re-entrant.
 *   - no CPU fallback: without a usable HIP device pb_create fails (PB_ERR_DEVICE).
 */
#ifndef PRISMA_BANDS_H
#define PRISMA_BANDS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pb_ctx pb_ctx;

typedef enum {
    PB_OK = 0,
    PB_ERR_ARG = -1,      /* bad argument / shape / missing weight            */
    PB_ERR_DEVICE = -2,   /* no HIP device, HIP call failed                   */
    PB_ERR_MEMORY = -3,   /* device or host allocation failed                 */
    PB_ERR_STATE = -4     /* call sequence error (e.g. wrong band for ctx)    */
} pb_status;

typedef enum { PB_F32 = 0, PB_F16 = 1, PB_U8 = 2, PB_I32 = 3 } pb_dtype;

/* One named weight tensor, reference state_dict naming
 * (Depth-Anything: `pretrained.*`, `depth_head.*` - bands/d_anything/dpt.py:139-166;
 *  RAFT: `fnet.*`, `cnet.*`, `update_block.*` - bands/raft/raft.py:24-58, without the
 *  `module.` prefix that bands/flow_raft.py:42-44 strips).  data is host memory, float32,
 *  C-contiguous; it is consumed (packed + uploaded) during pb_create and not referenced after. */
typedef struct {
    const char *name;
    int32_t dtype;        /* pb_dtype, PB_F32 only today */
    int32_t ndim;
    int64_t shape[6];
    const void *data;
} pb_tensor;

/* Depth-Anything geometry (bands/depth_anything.py:257-264 `--encoder`; the HF config that
 * DepthAnything.from_pretrained reads: encoder / features / out_channels). */
typedef struct {
    int32_t embed_dim;        /* 384 / 768 / 1024                       */
    int32_t depth;            /* 12 / 12 / 24 transformer blocks        */
    int32_t heads;            /* 6 / 12 / 16 (head_dim is always 64)    */
    int32_t features;         /* DPT head width 64 / 128 / 256          */
    int32_t out_channels[4];  /* DPT reassemble widths                  */
    int32_t pos_grid;         /* 37 (pos_embed rows = 1 + 37*37)        */
    int32_t max_batch;        /* frames per launch the arena is sized for (>= 1) */
    int32_t metric;           /* 1: ZoeDepth metric head on top (`--metric indoor|outdoor`, ViT-L only): weights also
                               * carry conv2.*, seed_bin_regressor.*, seed_projector.*, projectors.*, attractors.*,
                               * conditional_log_binomial.mlp.* (ZoeDepth state dict, `core.core.` prefix stripped);
                               * the network input is 392 x 518, depth_out is metric depth, use flip = 0 */
    int32_t precision;        /* pb_precision: 0 = one fp16 MFMA pass per GEMM, 1 = split-fp16 (see pb_precision)        */
} pb_depth_cfg;

/* Arithmetic of the GEMMs / convolutions (everything else - LayerNorm, softmax, residual streams, GRU state, norms'
 * statistics - is fp32 in both modes).
 *   PB_PREC_F16   : operands rounded to fp16 once, fp32 accumulation.  Against the fp32 reference: relative L2 < 1e-3,
 *                   max-norm error up to 1.6e-3 of the output range on the 24-block ViT-L.
 *   PB_PREC_SPLIT : the operands the error budget is dominated by are kept as hi + lo fp16 pairs and the products
 *                   a_hi w_hi + a_lo w_hi + a_hi w_lo accumulate in the same fp32 MFMA accumulators (2-3 passes over K):
 *                   max-norm and L2 error < 1e-3 on every reference vector - the mode the parity tests assert 1e-3 in. */
typedef enum { PB_PREC_F16 = 0, PB_PREC_SPLIT = 1 } pb_precision;

/* flow_raft options (cfg of pb_create; NULL = all zero). */
typedef struct {
    int32_t precision;        /* pb_precision */
} pb_flow_cfg;

/* SOLOv2 geometry and test_cfg for band = "mask_mmdet" (the values live in the mmdet config the reference downloads,
 * models/solov2_r101_fpn_3x_coco.py -> _base_ solov2_r50_fpn_1x_coco.py; bands/mask_mmdet.py:26-27). */
typedef struct {
    int32_t blocks[4];           /* ResNet bottleneck counts: 3,4,23,3 (R-101) / 3,4,6,3 (R-50)                 */
    int32_t scale_long;          /* test pipeline img_scale = (1333, 800), keep_ratio                            */
    int32_t scale_short;
    int32_t num_classes;         /* 80                                                                            */
    int32_t feat_channels;       /* 512: kernel / class branch width                                              */
    int32_t stacked_convs;       /* 4                                                                             */
    int32_t num_grids[5];        /* 40, 36, 24, 16, 12                                                            */
    int32_t strides[5];          /* 8, 8, 16, 32, 32 (area filter: mask area > stride)                            */
    int32_t mask_feat_channels;  /* 128                                                                           */
    int32_t mask_out_channels;   /* 256 (= dynamic kernel length)                                                 */
    int32_t nms_pre;             /* 500                                                                           */
    int32_t max_per_img;         /* 100                                                                           */
    float score_thr;             /* 0.1                                                                           */
    float mask_thr;              /* 0.5                                                                           */
    float filter_thr;            /* 0.05                                                                          */
    float sigma;                 /* 2.0 (gaussian Matrix NMS)                                                     */
    int32_t max_batch;           /* frames per backbone launch the arena is sized for (>= 1)                      */
    int32_t precision;           /* pb_precision                                                                  */
} pb_mask_cfg;

const char *pb_last_error(void);
int pb_version(void);
/* ABI guard: bumped whenever a public struct's layout or an entry point's signature changes.  A binding compares
 * pb_abi_version() with the PB_ABI_VERSION it was written against and pb_struct_size(which) with its own sizeof before the first
 * call (prisma_amd/_lib.py load(); integration/depth_anything_stub.py) - a stale binding fails at load, not by reading shifted
 * fields.  which: 0 pb_tensor, 1 pb_depth_cfg, 2 pb_flow_cfg, 3 pb_mask_cfg, 4 pb_kernel_stat, 5 pb_comm_id; -1 for others. */
#define PB_ABI_VERSION 3
int pb_abi_version(void);
int pb_struct_size(int which);
/* Number of visible HIP devices (0 on a CPU-only box; never fails). */
int pb_device_count(void);

/* init_model(): bands/depth_anything.py:48-76 (+ bands/d_anything/dpt.py:139-171).
 * band = "depth_anything" | "flow_raft" | "flow_gmflow" | "mask_mmdet" (the sections below).  Packs weights to fp16 MFMA layouts
 * and uploads them. */
int pb_create(pb_ctx **out, int device_id, const char *band, const pb_tensor *weights,
              int n_weights, const void *cfg, size_t cfg_bytes);
void pb_destroy(pb_ctx *ctx);

/* infer() + the per-frame post-process of process_video():
 * bands/depth_anything.py:100-143 and :215-221 (min/max, normalise, flip, heat_to_rgb).
 *   frames   : n x H x W x 3 uint8 RGB (decoder layout)            [host]
 *   depth_out: n x H x W float32 relative depth, or NULL           [host]
 *   rgb_out  : n x H x W x 3 uint8 heat-encoded frame, or NULL     [host]
 *   min_out / max_out: n floats each, or NULL                      [host]
 *   flip     : 1 for the relative model (bands/depth_anything.py:188) */
int pb_depth_infer_batch(pb_ctx *ctx, const uint8_t *frames, int n, int H, int W,
                         float *depth_out, uint8_t *rgb_out, float *min_out, float *max_out, int flip);

/* Same contract with every pointer in device memory of ctx's GPU (frames already resident
 * in HBM; used by bench.py and by pipelines that decode on the GPU).  Asynchronous on the
 * ctx stream; pb_sync() waits for it. */
int pb_depth_infer_batch_dev(pb_ctx *ctx, const uint8_t *frames, int n, int H, int W,
                             float *depth_out, uint8_t *rgb_out, float *min_out, float *max_out, int flip);
int pb_sync(pb_ctx *ctx);
/* Asynchronous host-pointer calls (round 6).  pb_depth_submit_batch / pb_flow_submit_sequence take the arguments of pb_depth_infer_batch /
 * pb_flow_infer_sequence, enqueue the whole three-stage pipeline (H2D, band, D2H of every chunk) and return; pb_wait(ctx) blocks until the
 * OLDEST submission of the ctx not yet waited for has its results in host memory (PB_ERR_STATE when none is outstanding).  Every buffer
 * handed to a submit - frames and each non-NULL output, the per-frame scalars included - must be page-locked host memory and stay valid and
 * untouched until the matching pb_wait returns.  Two submissions of one ctx may be in flight: the second one's uploads run under the first
 * one's kernels, the first one's downloads under the second one's, so a caller that streams clips (bands/depth_anything.py:203-225,
 * bands/flow_raft.py:98-113: the reference's frame loops) hides every copy.  Results are those of the blocking calls, bit for bit. */
int pb_depth_submit_batch(pb_ctx *ctx, const uint8_t *frames, int n, int H, int W,
                          float *depth_out, uint8_t *rgb_out, float *min_out, float *max_out, int flip);
int pb_flow_submit_sequence(pb_ctx *ctx, const uint8_t *frames, int F, int H, int W, float scale, int iters, int backward,
                            float *flow_out, uint8_t *rgb_out, float *maxdisp_out);
int pb_wait(pb_ctx *ctx);
/* Several bands at once: contexts share nothing, every *_dev entry point returns after the enqueue, so a caller enqueues on two or three
 * contexts and then pb_sync()s each (prisma_amd.engine.run_concurrently; replaces the reference's strictly sequential band order,
 * process.py:205-290, where the bands of one video are independent).  Results are those of running the bands one after the other. */

/* Multi-GPU: one process and one pb_ctx per GPU; frames shard by rank and never cross GPUs.  The only
 * exchange is the all-gather of the per-frame scalars the CSV files need in frame order - depth (min, max), flow max
 * displacement: 4-12 bytes per frame - over RCCL (xGMI inside a node).  Rank 0 calls pb_comm_unique_id and hands the 128
 * bytes to the other ranks by any host channel (a file, torch.distributed's store); every rank then calls pb_comm_init on
 * its ctx (collective), and pb_gather_scalars (collective): `local` n_local floats of this rank -> `global`
 * [world x n_local] on every rank, rank-major.  Ranks with fewer frames pad to a common n_local.  Replaces the implicit
 * single-process ordering of bands/depth_anything.py:215-238 and bands/flow_raft.py:138-141. */
typedef struct { char internal[128]; } pb_comm_id;       /* = ncclUniqueId */
int pb_comm_unique_id(pb_comm_id *id_out);
int pb_comm_init(pb_ctx *ctx, const pb_comm_id *id, int rank, int world);
int pb_gather_scalars(pb_ctx *ctx, const float *local, int n_local, float *global);

/* Still-image / `--subpath` post-process: write_depth(heatmap=True, encode_range) of bands/common/io.py:138-172 as called from
 * bands/depth_anything.py:176-180,221-225, with encode.py:73-95 (float_to_edge, saturation) and :141-146 (float_to_rgb) -
 * normalise by the map's own min / max, flip, heat ramp, Sobel-edge magnitude of the 8-bit map in the saturation, min / max packed
 * as 24-bit fixed point of [0, 1000] into pixels (0, 0) and (0, 1), uint8 truncation.  Bytes equal the reference's
 * (tests/golden/write_depth.npz; cv2.Sobel's ksize-1 taps restated).  Works on any ctx.
 *   depth   : H x W float32 [host]        rgb_out : H x W x 3 uint8 RGB (what the PNG holds) [host]
 *   min_out / max_out: the map's min / max, or NULL */
int pb_depth_encode_still(pb_ctx *ctx, const float *depth, int H, int W, int flip, int encode_range, uint8_t *rgb_out,
                          float *min_out, float *max_out);

/* Network input size for an H x W frame: keep-aspect lower-bound resize to 518, each side a
 * multiple of 14 (bands/d_anything/util/transform.py:100-166). */
int pb_depth_net_size(int H, int W, int *net_h, int *net_w);

/* flow_raft band (band = "flow_raft", cfg = NULL; weights: fnet.*, cnet.*, update_block.*).
 * Replaces bands/flow_raft.py:98-113 (frame loop: cv2.resize(fx=fy=scale, INTER_CUBIC), [prev,curr] /
 * [curr,prev] batch), :51-62 infer (InputPadder, RAFT(iters, test_mode=True), unpad) and
 * bands/common/flow.py:64-88 write_flow -> encode.py:98-126 process_flow.
 *   frames     : F x H x W x 3 uint8 RGB, consecutive frames of one clip (F >= 2)
 *   outputs    : for pair i = (frame i, frame i+1) and direction d (0 = forward i -> i+1, 1 = backward when
 *                `backward` != 0), index i*dirs + d:
 *     flow_out   [F-1, dirs, sh, sw, 2] float32 (u, v) pixels at the scaled resolution, or NULL
 *     rgb_out    [F-1, dirs, sh, sw, 3] uint8 process_flow encoding, or NULL
 *     maxdisp_out[F-1, dirs] float32 max displacement (the band's CSV value), or NULL
 *   (sh, sw) = pb_flow_out_size(H, W, scale). */
int pb_flow_out_size(int H, int W, float scale, int *sh, int *sw);
int pb_flow_infer_sequence(pb_ctx *ctx, const uint8_t *frames, int F, int H, int W, float scale, int iters, int backward,
                           float *flow_out, uint8_t *rgb_out, float *maxdisp_out);
int pb_flow_infer_sequence_dev(pb_ctx *ctx, const uint8_t *frames, int F, int H, int W, float scale, int iters, int backward,
                               float *flow_out, uint8_t *rgb_out, float *maxdisp_out);
/* Forward/backward consistency masks.  Replaces bands/common/flow.py:19-40 compute_fwdbwd_mask as
 * called from bands/flow_raft.py:63-64: the opposite flow is sampled at p + f(p) with cv2.remap's INTER_LINEAR /
 * BORDER_CONSTANT arithmetic, and mask = |f + f'| < alpha1 (|f| + |f'|) + alpha2 (reference defaults 0.05, 0.5).
 *   pb_flow_infer_sequence_masks*: both directions are always computed (dirs = 2); flow_out / rgb_out / maxdisp_out
 *     as above (any may be NULL), mask_out [F-1, 2, sh, sw] bytes of 0 / 1 (index 0 = forward mask, 1 = backward).
 *   pb_flow_fwdbwd_mask: the mask step alone on host flows [n, 2, sh, sw, 2].
 * The host-pointer variant is the same chunked three-stage pipeline as pb_flow_infer_sequence (both directions of a pair are in one
 * chunk; page-locked caller buffers are used directly); every host-pointer pipeline drains its streams before it returns, error or not. */
int pb_flow_infer_sequence_masks(pb_ctx *ctx, const uint8_t *frames, int F, int H, int W, float scale, int iters,
                                 float alpha1, float alpha2, float *flow_out, uint8_t *rgb_out, float *maxdisp_out,
                                 uint8_t *mask_out);
int pb_flow_infer_sequence_masks_dev(pb_ctx *ctx, const uint8_t *frames, int F, int H, int W, float scale, int iters,
                                     float alpha1, float alpha2, float *flow_out, uint8_t *rgb_out, float *maxdisp_out,
                                     uint8_t *mask_out);
int pb_flow_fwdbwd_mask(pb_ctx *ctx, const float *flows, int n, int sh, int sw, float alpha1, float alpha2,
                        uint8_t *mask_out);
/* flow_gmflow band (band = "flow_gmflow", cfg = pb_flow_cfg or NULL; weights: backbone.*, transformer.*, feature_flow_attn.*,
 * upsampler.* of the GMFlow state dict).  Same entry points and output contract as flow_raft above (pb_flow_infer_sequence*,
 * pb_flow_out_size; `iters` is ignored): replaces bands/flow_gmflow.py:60-118 infer (cv2.resize by --scale, InputPadder(16),
 * GMFlow(attn_splits_list=[2], corr_radius_list=[-1], prop_radius_list=[-1], pred_bidir_flow per `backward`)) ->
 * bands/gmflow/gmflow.py:12-170 (shared instance-norm encoder, position encoding, 6 transformer blocks with 2 x 2 shifted windows,
 * global matching, self-attention propagation, convex upsampling) and the same write_flow / process_flow encode.  Only the band's
 * default configuration is built (1 scale, 128 channels, 6 layers, 1 head, ffn x 4); the band script rejects other GMFlow flags. */
/* Stages of the last flow call: "fmap" [F,256,h/8,w/8], "flow_lo" [pairs*dirs, h/8*w/8, 2]; flow_gmflow (token-major fp32, shape
 * [n, tokens, channels, 1]): "feat" [F, h/8*w/8, 128] (encoder output), "block0" / "tfeat" [2 pairs, tokens, 128] (after the first / last
 * transformer block; both images of every pair), "flow_match" / "flow_prop" [pairs*dirs, tokens, 2]. */
int64_t pb_flow_get_stage(pb_ctx *ctx, const char *name, float *out, int64_t cap, int64_t shape_out[4]);
/* flow_gmflow --inference_size (reference bands/flow_gmflow.py:76-100): with (h, w) > 0 - multiples of 16 - the network runs on
 * F.interpolate(bilinear, align_corners = True) of the (scaled) frame to h x w instead of on the frame padded to /16, and the flow is
 * resized back the same way with u * W' / w, v * H' / h.  (0, 0) turns it off.  flow_gmflow contexts only. */
int pb_flow_set_inference_size(pb_ctx *ctx, int h, int w);

/* mask_mmdet band (band = "mask_mmdet", cfg = pb_mask_cfg; weights: backbone.*, neck.*, mask_head.* in mmdet's
 * state_dict naming).  Replaces the per-frame body of bands/mask_mmdet.py:131-154: inference_detector
 * (mmdet/apis/inference.py:99-162: Resize keep_ratio (1333, 800) -> Normalize -> Pad 32 -> SOLOv2 forward ->
 * get_results / Matrix NMS -> format_results) and the accumulation of :43-61,139-147.
 *   frames      : n x H x W x 3 uint8 RGB (what decord hands the band; the reference swaps to BGR for mmdet and
 *                 Normalize(to_rgb) swaps back, so means / stds apply in RGB order)
 *   confidence  : --confidence; an instance is drawn when its class is kept and score > 0.5 and > confidence
 *   keep_classes: class ids (model.CLASSES order) the band keeps, mask_mmdet.py:30; NULL keeps every class
 *   mask_out    : n x H x W x 3 uint8: per pixel (255 * number of drawn instances covering it) mod 256 in all
 *                 three channels - `masks.astype(np.uint8)` of the reference's float64 sum.
 * pb_mask_get_instances: what format_results held for frame `frame` of the last call, score-descending: up to cap
 *   scores / labels; returns the count.  masks_out (optional, [count, H, W] bytes of 0 / 1) needs
 *   pb_set_profiling(ctx, 2) before the infer call.
 * pb_mask_net_size: resized (nh, nw) and padded (Hp, Wp) network input for an H x W frame.
 * The host-pointer variant (reference loop bands/mask_mmdet.py:131-154) pipelines over the engine's chunks of max_batch frames: every
 * chunk's frames go to the device up front on a copy stream, chunk i's id images return on a second one while chunk i + 1 runs. */
int pb_mask_infer_batch(pb_ctx *ctx, const uint8_t *frames, int n, int H, int W, float confidence,
                        const int32_t *keep_classes, int n_keep, uint8_t *mask_out);
int pb_mask_infer_batch_dev(pb_ctx *ctx, const uint8_t *frames, int n, int H, int W, float confidence,
                            const int32_t *keep_classes, int n_keep, uint8_t *mask_out);
int pb_mask_get_instances(pb_ctx *ctx, int frame, int cap, float *scores_out, int32_t *labels_out, uint8_t *masks_out,
                          int32_t *candidates_out);
int pb_mask_net_size(const pb_mask_cfg *cfg, int H, int W, int *nh, int *nw, int *Hp, int *Wp);
/* --sdf (process.py passes it on every run, /root/reference/process.py:46-48,207; bands/mask_mmdet.py:64-69 getSDF, :150-152): the
 * clamped signed distance field of the id image in its GREEN channel.  getSDF's byte is a function of the side of the mask a pixel is
 * on and of its (integer) squared Euclidean distance n to the other side; the host tabulates it with the reference's own float64
 * expression - tab_out[i] / tab_in[i] for n = i outside / inside the mask, the last entry (index n_tab - 1) for every n >= n_tab - 1
 * (the remap saturates at sdf >= 64.25 and <= -63.25, so n_tab = 4130 holds every distinct byte; n_tab <= 4226) - and the library
 * computes the exact n on the device (mask_kernels.hip sdf_*_kernel).
 *   pb_mask_set_sdf : n_tab > 0 turns the green channel on for every following pb_mask_infer_batch* call of the ctx, 0 turns it off
 *   pb_mask_sdf_green(_dev): the same pass on id images the caller holds ([n, H, W, 3] uint8, in place; host / device pointer) */
int pb_mask_set_sdf(pb_ctx *ctx, const uint8_t *tab_out, const uint8_t *tab_in, int n_tab);
int pb_mask_sdf_green(pb_ctx *ctx, uint8_t *masks, int n, int H, int W);
int pb_mask_sdf_green_dev(pb_ctx *ctx, uint8_t *masks, int n, int H, int W);
/* Stages of the last mask call as float32 NCHW: "input", "c2".."c5", "p2".."p6", "mask_feats",
 * "kernel_pred<l>", "cls_logit<l>" (l = 0..4). */
int64_t pb_mask_get_stage(pb_ctx *ctx, const char *name, float *out, int64_t cap, int64_t shape_out[4]);

/* Debug/parity: copy a named intermediate of the last pb_depth_infer_batch* call to the host
 * as float32 in the reference's layout ([n, C, h, w] for maps, [n, tokens, D] for tokens).
 * Names: "tokens", "block<i>", "feat<i>", "layer<i>_rn", "path<i>", "output_conv1", "net_depth".
 * shape_out receives up to 4 dims; returns the element count or a negative pb_status.  shape_out[0] is the number of frames the
 * stage holds: the whole call for the ViT's stages; for the DPT head's stages the LAST head chunk (the trailing shape_out[0] frames of
 * the call) when the call was large enough for the head to run in several chunks (split precision: more than 16 frames of 1080p). */
int64_t pb_depth_get_stage(pb_ctx *ctx, const char *name, float *out, int64_t cap, int64_t shape_out[4]);

/* Device memory helpers so a Python host without torch can keep frames resident. */
int pb_dev_alloc(pb_ctx *ctx, void **ptr, size_t bytes);
int pb_dev_free(pb_ctx *ctx, void *ptr);
int pb_memcpy_h2d(pb_ctx *ctx, void *dst, const void *src, size_t bytes);
int pb_memcpy_d2h(pb_ctx *ctx, void *dst, const void *src, size_t bytes);

/* Timing of the kernels launched by the last infer call, measured with HIP events on the ctx
 * stream: fills up to cap entries; returns the count.  name points into static storage. */
typedef struct {
    const char *name;     /* kernel family, e.g. "gemm_f16", "attention", "conv3x3"      */
    double ms;            /* summed event time of that family in the last call            */
    double flops;         /* algorithmic FLOPs those launches performed (2 M N K of the layers) */
    double exec_flops;    /* MFMA work actually issued, in fp16-pass equivalents: x1.5 ... x3 for the split layers (PB_PREC_SPLIT; an e4m3 pass counts 0.5) */
    double bytes;         /* algorithmic HBM bytes (compulsory traffic) of those launches */
    int32_t launches;
} pb_kernel_stat;
/* enabled: bit 0 = time every kernel launch with HIP events on the ctx stream, bit 1 = keep debug stages / instance
 * masks, bit 2 = accumulate the timings over successive infer calls (pb_get_kernel_stats then reports the sums since
 * this call) instead of restarting at every infer call. */
int pb_set_profiling(pb_ctx *ctx, int enabled);
/* Tuning / A-B switches: "gemm_tile", "conv_tile" = 0 auto, 1 128x128, 2 256x256 ping-pong, 3 256x32, 9 256x64,
 * 12 128x96 (convolutions with N <= 96) (prisma_amd/csrc/gemm.h; the other round-1 variants were measured slower and
 * removed); "tile_n96" (process-wide) = 0: convolutions with 64 < N <= 96 stay on the 128x128 tile, 1: auto gives them
 * the 128x96 tile (same bytes), 2 (default): ... and the packed-channel K axis where the weights carry one (RAFT encoder
 * stage 2: another summation order); "host_chunk", "op_splitk": see INTEGRATION.md. */
int pb_set_option(pb_ctx *ctx, const char *key, int value);
int pb_get_kernel_stats(pb_ctx *ctx, pb_kernel_stat *out, int cap);

/* ---- single-kernel entry points (host buffers) used by the -m gpu parity tests ---------- */
/* C[M,N] = act(A[M,K] @ W[N,K]^T + bias) with fp16 operands, fp32 accumulate (MFMA).       */
int pb_op_gemm(pb_ctx *ctx, const float *A, const float *W, const float *bias, float *C,
               int M, int N, int K, int act /*0 none,1 relu,2 gelu*/, int tile /*0 auto,1 128x128,2 256x256*/);
/* Kernel micro-benchmark on device-resident uniform[-1,1) fp16 data: mean ms per launch over iters.
 * epi: 0 fp16 store, 1 bias+GELU fp16 store, 2 LayerScale + fp32 residual read-modify-write. */
int pb_op_gemm_bench(pb_ctx *ctx, int M, int N, int K, int tile, int epi, int iters, double *ms_out);
/* The flow band's all-pairs correlation kernel on its own (bands/raft/corr.py:52-60, without the 1/sqrt(256) factor the band folds into the
 * features): out[m, n] = fp16(sum_k fp16(A[m, k]) fp16(W[n, k])), A [M, 256], W [N, 256] fp32 on the host, N % 8 == 0, ldo >= N.
 * `out` holds (M + guard_rows) x ldo floats; everything the kernel must not touch - columns N..ldo-1 and the guard rows - is preset to NaN. */
int pb_op_corr_volume(pb_ctx *ctx, const float *A, int M, const float *W, int N, int ldo, int guard_rows, float *out);
/* LayerNorm over the last dim, eps 1e-6 (vision_transformer.py:95). */
/* times `iters` launches of the fused attention kernel on random Q, K, V (variant 0 = default); ms per launch */
int pb_op_attention_bench(pb_ctx *ctx, int B, int heads, int N, int variant, int iters, double *ms_out);
int pb_op_layernorm(pb_ctx *ctx, const float *x, const float *g, const float *b, float *y, int rows, int D);
/* softmax(q k^T * 64^-0.5) v per (batch, head); q,k,v,o: [B, heads, N, 64] float32
 * (dinov2/layers/attention.py:49-62). */
int pb_op_attention(pb_ctx *ctx, const float *q, const float *k, const float *v, float *o,
                    int B, int heads, int N);
/* NCHW float32 conv2d via NHWC fp16 implicit GEMM: x [B,Ci,H,W], w [Co,Ci,kh,kw]. */
/* single-head attention over 128-wide heads, q / k / v / o [B, L, 128]; region [B, L] or NULL: a key whose region id differs from the
 * query's gets -100 on its logit (bands/gmflow/transformer.py:8-15, 18-44, 47-101) - a building block of the flow_gmflow band */
int pb_op_attention128(pb_ctx *ctx, const float *q, const float *k, const float *v, const int8_t *region, float *o, int B, int L);
/* the same attention in the flow_gmflow band's split precision (q, k, v and the probabilities as hi + lo fp16 pairs: three MFMA passes
 * for the scores and for P V); v / o [B, L, vcols] with vcols 128 or 32 (coordinates / flow padded to 32 columns); region [nreg, L],
 * batch element b uses row b % nreg; keys and values of batch element b are those of b ^ kxor (cross attention between a pair's frames) */
int pb_op_attention128_split(pb_ctx *ctx, const float *q, const float *k, const float *v, const int8_t *region, int nreg, float *o,
                             int B, int L, int vcols, int kxor);
int pb_op_conv2d(pb_ctx *ctx, const float *x, const float *w, const float *bias, float *y,
                 int B, int Ci, int H, int W, int Co, int ksize, int stride, int pad, int relu_in, int relu_out);
/* bilinear resize NCHW float32, align_corners 0/1 (torch F.interpolate semantics). */
int pb_op_bilinear(pb_ctx *ctx, const float *x, float *y, int B, int C, int H, int W, int OH, int OW,
                   int align_corners);
/* uint8 frame -> normalised network input [3, net_h, net_w] float32
 * (bands/depth_anything.py:122-126). */
int pb_op_preprocess(pb_ctx *ctx, const uint8_t *frame, int H, int W, float *out, int net_h, int net_w);
/* float32 depth [n,H,W] -> heat RGB + min/max (bands/depth_anything.py:215-221). */
int pb_op_encode_depth(pb_ctx *ctx, const float *depth, int n, int H, int W, int flip,
                       uint8_t *rgb, float *mn, float *mx);

#ifdef __cplusplus
}
#endif
#endif /* PRISMA_BANDS_H */
