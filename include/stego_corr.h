/*
 * stego_corr.h  --  C ABI of libstego_corr.so: the MI355X (gfx950) implementation of
 * STEGO's feature-correspondence distillation loss.
 *
 * The reference (mhamilton723/STEGO) has no FFI/plugin interface: its boundary for this
 * path is the Python surface of src/modules.py.  Each entry point below therefore names
 * the reference function(s) it replaces (file:line relative to the reference repo); the
 * Python host layer (stego_amd/modules.py) re-creates the reference classes on top of
 * these calls and INTEGRATION.md shows the binding a maintainer of the reference adds.
 *
 * Conventions
 *   - all tensors are device (HBM) pointers to float32 unless stated; index data is int64;
 *   - inputs are read-only, outputs are fully overwritten; nothing is allocated, freed or
 *     synchronised inside a call: work is enqueued on `stream` (a hipStream_t) and the
 *     caller owns ordering; the only scratch is the caller-provided workspace;
 *   - every function returns STEGO_OK (0) or an error code (never throws / aborts);
 *     stego_error_string() renders a code.  Shape/argument errors are detected on the
 *     host before anything is enqueued.
 */
#ifndef STEGO_CORR_H
#define STEGO_CORR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STEGO_ABI_VERSION 7   /* 2: + stego_corr_workspace_prepare, stego_corr_fwd_prepared, stego_corr_fwd_launches, stego_finish_draws, stego_debug_set; K <= 128
                                * 3: StegoCorrDesc.flags (STEGO_FLAG_SHARED_DEVICE is per call, no longer a process-wide knob)
                                * 4: StegoHeadDesc.tokens_amax, stego_tokens_from_cache, stego_ref_dropout_masks, stego_ref_draws_indirect, stego_corr_workspace_prepare_now
                                * 5: stego_corr_event_counters
                                * 6: StegoVitDesc.precision (STEGO_VIT_F16X3: the backbone in the fp32 class)
                                * 7: stego_sample, stego_sample_bwd, stego_rowsum, stego_loss_pointwise_fwd / _bwd, stego_sample_panels, stego_dense_corr_panels, stego_sample_bwd_rows; stego_corr_fwd / _bwd take S = 12 .. 16 */

enum {
    STEGO_OK = 0,
    STEGO_ERR_NULL = 1,         /* a required pointer is NULL                               */
    STEGO_ERR_SHAPE = 2,        /* a dimension is <= 0 or inconsistent                      */
    STEGO_ERR_UNSUPPORTED = 3,  /* valid request this build has no kernel for (see limits)  */
    STEGO_ERR_WORKSPACE = 4,    /* workspace_bytes < stego_corr_workspace_bytes()           */
    STEGO_ERR_ALIGN = 5,        /* a pointer is not 4-byte aligned                          */
    STEGO_ERR_HIP = 1000        /* STEGO_ERR_HIP + hipError_t of a failed launch            */
};

/* Arithmetic used for the channel contraction (the einsum of modules.py:283-284). */
enum {
    STEGO_PREC_F32 = 0,      /* v_mfma_f32_32x32x2_f32: fp32 products, fp32 accumulate (runs at the VALU rate) */
    STEGO_PREC_F16X3 = 1     /* RECOMMENDED: every fp32 operand split into fp16 hi+lo (22 mantissa bits after a   */
                             /* per-point power-of-two prescale), hi*hi + hi*lo + lo*hi on the fp16 matrix cores, */
                             /* fp32 accumulate: measured error on the loss equals the F32 mode (2.6e-8 mean abs, */
                             /* bounds on adversarial inputs in tests/test_parity_gpu.py).  Feature correlation in */
                             /* every forward path; in the single-launch forward also the code correlation, and    */
                             /* in the backward the two code GEMMs (the three-launch forward keeps the code        */
                             /* correlation in exact fp32)                                                         */
};

/* hipStream_t without dragging the HIP headers into C callers. */
typedef void* stego_stream_t;

/* A float32 [N, C, H, W] map with arbitrary element strides (NCHW-contiguous, or the
 * channels-last strided view DinoFeaturizer produces at modules.py:97). */
typedef struct StegoMap {
    const float* data;
    int64_t stride_n, stride_c, stride_h, stride_w;   /* in elements */
} StegoMap;

/* Problem description = the cfg keys ContrastiveCorrelationLoss reads
 * (modules.py:330-387; configs/train_config.yml:41-64) + tensor shapes. */
typedef struct StegoCorrDesc {
    int32_t B;                 /* local batch: orig_feats.shape[0]            (modules.py:355) */
    int32_t C;                 /* feature channels (384 ViT-S, 768 ViT-B)                      */
    int32_t K;                 /* code channels = cfg.dim                                      */
    int32_t H, W;              /* feature-map height / width (28 at 224^2/8, 40 at 320^2/8)    */
    int32_t S;                 /* cfg.feature_samples; S*S sample points per image             */
    int32_t n_neg;             /* cfg.neg_samples                                              */
    int32_t pointwise;         /* cfg.pointwise   (modules.py:330-333)                         */
    int32_t zero_clamp;        /* cfg.zero_clamp  -> clamp min 0.0 else -9999.0 (:337-340)     */
    int32_t stabalize;         /* cfg.stabalize   -> clamp max 0.8               (:342-345)    */
    float pos_intra_shift;     /* cfg.pos_intra_shift (:376)                                   */
    float pos_inter_shift;     /* cfg.pos_inter_shift (:378)                                   */
    float neg_inter_shift;     /* cfg.neg_inter_shift (:387)                                   */
    int32_t precision;         /* STEGO_PREC_*                                                 */
    int32_t flags;             /* STEGO_FLAG_* (0 = defaults)                                  */
} StegoCorrDesc;

/* StegoCorrDesc.flags */
enum {
    STEGO_FLAG_SHARED_DEVICE = 1   /* other kernels (the gradient all-reduce of a data-parallel job overlapping the next step) run on the
                                    * device while this call does: the fused forward launches one workgroup per tile only, instead of one
                                    * per compute unit, so that it never waits for a compute unit somebody else holds.  Same results,
                                    * bit for bit; ~2 us slower on an otherwise idle device.  Per call: part of the descriptor. */
};

/* Limits of this build: S*S <= 128 (S <= 11): K (cfg.dim; the reference ships 70) <= 128 on the fused path (any parity,
 * channels-last maps with C = 192 / 384 / 768, i.e. what DinoFeaturizer emits) and <= 72 elsewhere (other layouts / widths,
 * helper()); 128 < S*S <= 256 (S = 12 .. 16, ABI 7): K <= 128, any C and layout - the same entry points run the multi-launch kernels of
 * csrc/corr_wide.hip (stego_corr_fwd_launches says 8; split-fp16 products in both precision modes; bitwise repeatable like S <= 11 unless a
 * pixel of a map receives more than 64 sample taps or the map has more than 4096 pixels); every per-image element offset < 2^31.
 * Anything else returns STEGO_ERR_UNSUPPORTED.  For K > 80 the backward's
 * GEMMs are split-fp16 products in both precision modes (their fp32 operand images no longer fit LDS).
 * Determinism: every kernel sums in a fixed order (bitwise repeatable results), with ONE exception: the backward of maps
 * wider than 64 pixels (no BASELINE config) takes a band fallback whose fp32 summation order follows arrival order; repeated
 * runs agree to <= 1e-6 of the largest gradient (tests/test_parity_gpu.py). */

int stego_abi_version(void);
const char* stego_error_string(int code);

/* Measurement / ablation knobs for tools (never needed by a caller of the product path).  The library reads
 * STEGO_DEBUG, STEGO_DEBUG_SAMPLE, STEGO_DEBUG_BWD, STEGO_DEBUG_VIT, STEGO_DEBUG_KNN, STEGO_FWD_VARIANT from the
 * environment ONCE, when it is loaded (no getenv in any call); this overrides knob `which` (0..5 in that order).
 * Knob 6, STEGO_SHARED_DEVICE: tools only since ABI 3 (the deployment setting is StegoCorrDesc.flags); a value > 8 still forces
 * that many phase-1 owner workgroups for measurements.  The knobs are atomics: flipping one while another thread launches is safe. */
int stego_debug_set(int32_t which, int32_t value);

/* Test hook: n_workgroups workgroups of 256 threads holding lds_bytes of LDS each spin for `microseconds` on `stream` - a
 * stand-in for a foreign kernel (a collective) that occupies compute units while the loss runs (tests/test_parity_gpu.py). */
int stego_debug_occupy(int32_t n_workgroups, int32_t lds_bytes, int32_t microseconds, stego_stream_t stream);

/* Buffer sizes (bytes; depend only on the descriptor; 0 for an invalid/unsupported descriptor).
 *   workspace : scratch of one forward call (sampled operand images + per-tile partial sums)
 *   saved_ctx : what the forward leaves for the backward of the CODE side (normalised sampled codes,
 *               their norms and the bilinear tap tables); pass NULL to the forward when no backward
 *               will follow (the data then lives in the workspace). */
size_t stego_corr_workspace_bytes(const StegoCorrDesc* desc);          /* forward scratch  */
size_t stego_corr_saved_ctx_bytes(const StegoCorrDesc* desc);          /* forward -> backward context */
size_t stego_corr_bwd_workspace_bytes(const StegoCorrDesc* desc);      /* backward scratch */
size_t stego_corr_helper_workspace_bytes(const StegoCorrDesc* desc);
size_t stego_corr_helper_saved_ctx_bytes(const StegoCorrDesc* desc);
size_t stego_corr_helper_bwd_workspace_bytes(const StegoCorrDesc* desc);

/*
 * Forward of ContrastiveCorrelationLoss.forward  (modules.py:349-398) with the RNG draws made by the
 * caller in the reference's order (coords1 :366, coords2 :367, super_perm x n_neg :383).
 *
 * Channels-last maps of the ViT widths (C = 192 / 384 / 768) with B <= the device's compute units (256) take the FUSED path:
 * ONE kernel launch (corr_fused_kernel) in which every workgroup owns one (pair-set, image) tile; when there are more tiles
 * ((2 + n_neg) * B) than compute units, the grid is a sequence of windows of whole pair-sets (floor(CUs / B) pair-sets each: the
 * rendezvous of a pair-set never spans two windows, so a window that is resident never waits for one that is not):
 *   - the anchor sets (@coords1) are sampled (sample, :287-288: border, align_corners) and L2-normalised
 *     (norm, :275-276) once, by the workgroups of the XCD that caches their image, and handed to the tiles that
 *     need them through write-through stores + a counter (no kernel boundary);
 *   - the positive / negative sets (@coords2, orig[perm]) are gathered straight into an LDS ring by a team of
 *     gather waves while a team of MFMA waves runs both correlation tensors (tensor_correlation, :283-284);
 *   - the pointwise row centring (:332), clamp*(fd-shift) (:337-345) and the batch-global old_mean (:331,:333: a
 *     rendezvous of the 32 tiles of a pair-set inside the launch) finish in the same workgroup;
 *   - the last workgroup to finish writes the two .mean()s (:393,:395) and the saved means.
 * Anything else (NCHW maps, other widths, odd K) takes three launches: sample_norm_kernel, corr_tile_kernel,
 * corr_finalize_kernel - same results.
 *
 * Workspace and hand-off words.  The fused path keeps a few hundred bytes of counters in the workspace that must be
 * ZERO when a launch starts; every launch leaves them zero again.  stego_corr_fwd() zeroes them itself (one extra
 * memset node in front of the kernel, ~5 us): it accepts any workspace memory.  A caller that keeps its workspace
 * calls stego_corr_workspace_prepare() once after allocating it (and again after a launch that FAILED), and then
 * stego_corr_fwd_prepared(): same arguments and results, one launch.  A workspace serves one call at a time
 * (stream order is enough).
 *
 *   feats, feats_pos : [B,C,H,W]   (orig_feats, orig_feats_pos; never differentiated)
 *   code, code_pos   : [B,K,H,W]   (orig_code, orig_code_pos)
 *   coords1, coords2 : [B,S,S,2] contiguous, values in [-1,1] (x=width first)
 *   perms            : int64 [n_neg,B] contiguous, values in [0,B)  (may be NULL iff n_neg==0)
 * outputs (contiguous):
 *   loss_means       : [3] = { pos_intra_loss.mean(), pos_inter_loss.mean(), neg_inter_loss.mean() }   (ABI 2: three floats -
 *                      the third is torch.cat(negative losses).mean(), what train_segmentation.py:176 computes; 0 when n_neg == 0)
 *   pos_intra_cd, pos_inter_cd : [B,S,S,S,S]
 *   neg_inter_loss, neg_inter_cd : [n_neg*B,S,S,S,S]   (torch.cat over negatives, :390-391)
 *   saved_w          : optional [(2+n_neg)*B, S^4]: (fd_centred - shift) per pair-set with the clamp pass-mask
 *                      1[min <= cd <= max] in the mantissa LSB (opaque to callers), and
 *   saved_mean       : optional [2+n_neg]: old_mean per pair-set; both or neither;
 *   saved_ctx        : optional, stego_corr_saved_ctx_bytes(); with saved_w/saved_mean it is all the
 *                      backward needs (the feature side is no_grad in the reference, :326).
 */
int stego_corr_fwd(const StegoCorrDesc* desc,
                   const StegoMap* feats, const StegoMap* feats_pos,
                   const StegoMap* code, const StegoMap* code_pos,
                   const float* coords1, const float* coords2, const int64_t* perms,
                   float* loss_means,
                   float* pos_intra_cd, float* pos_inter_cd,
                   float* neg_inter_loss, float* neg_inter_cd,
                   float* saved_w, float* saved_mean, void* saved_ctx,
                   void* workspace, size_t workspace_bytes, stego_stream_t stream);

/* The RNG draws of ContrastiveCorrelationLoss.forward as the generator emits them -> what the loss consumes, one launch:
 *   coords1/2 = u1/2 * 2 - 1                          (torch.rand(...) * 2 - 1, modules.py:366-367; bit-identical)
 *   perms[n]  = super_perm fix-up of raw_perms[n]     (perm[perm == arange] += 1; perm % B, modules.py:307-311, :383)
 * u1, u2: n_coord floats each; raw_perms: n_neg (<= 16) HOST-array of device pointers to int64 [B]; perms: int64 [n_neg, B]. */
int stego_finish_draws(const float* u1, const float* u2, int64_t n_coord, const int64_t* const* raw_perms, int32_t n_neg,
                       int32_t B, float* coords1, float* coords2, int64_t* perms, stego_stream_t stream);

/* The SAME draws from ONE launch: torch.rand(n_coord) x 2 (* 2 - 1) and torch.randperm(B) x n_neg (+ the super_perm fix-up) exactly
 * as PyTorch-ROCm's device generator produces them from the state (seed, offset) - Philox-4x32-10 with ATen's counter layout,
 * random_ keys, stable sort on the low key bits, reshuffle of duplicate-key islands (csrc/draws.hip restates
 * aten/src/ATen/native/cuda/{DistributionTemplates.h, Randperm.cu, Randperm.cuh}).  The caller reads (seed, offset) from the torch
 * generator (initial_seed(), get_offset()) and afterwards sets its offset to offset + stego_ref_draws_advance(...): the random stream of
 * a training run is the reference's, call for call.  `variant`: bit 0 = the distribution kernels' grid is sized per `unroll` elements,
 * bit 1 = the uniform conversion is an fma, bit 2 = randperm's keys come from the 32-bit flavour of random_ - properties of the
 * installed torch build; stego_amd/modules.py picks the variant that
 * reproduces the real torch calls (checked once per process) and keeps the torch calls if none does.  offset % 4 == 0, B <= 2048. */
int stego_ref_draws(uint64_t seed, uint64_t offset, int32_t variant, int64_t n_coord, int32_t n_neg, int32_t B, float* coords1,
                    float* coords2, int64_t* perms, stego_stream_t stream);
uint64_t stego_ref_draws_advance(int64_t n_coord, int32_t n_neg, int32_t B, int32_t variant);
/* The same launch with the generator state read ON THE DEVICE when the kernel runs: seed = *seed_ptr, offset = *offset_ptr +
 * offset_intragraph - ATen's graph-safe form (at::PhiloxCudaState as CUDAGeneratorImpl::philox_cuda_state(increment) returns it while a
 * stream is being captured: CUDAGraph::replay refreshes the two words before every replay, so every replay of a captured step draws the
 * generator's next numbers, as the captured torch calls would).  The caller obtains the triple from torch and registers the increment
 * stego_ref_draws_advance(...) with the generator in the same call (stego_amd/csrc/torch_glue_ext.cpp: the only torch C++ in the
 * tree, plumbing like the stream handle). */
int stego_ref_draws_indirect(const int64_t* seed_ptr, const int64_t* offset_ptr, uint64_t offset_intragraph, int32_t variant,
                             int64_t n_coord, int32_t n_neg, int32_t B, float* coords1, float* coords2, int64_t* perms,
                             stego_stream_t stream);

/* OPT-IN alternative to the torch draws (cfg.fast_draws): the same DISTRIBUTIONS as modules.py:366-367, 382-385 - coords
 * uniform on torch.rand's 2^-24 lattice, times 2 minus 1; one uniformly random permutation of [0, B) per negative followed by
 * the super_perm fix-up - from one kernel with its own counter-based generator (Philox-4x32-10) keyed by the 64 random bits at
 * `seed` (device memory; the caller draws them from the torch generator).  NOT the reference's random stream.
 * coords1/2: n_coord floats each; perms: int64 [n_neg, B]. */
int stego_fast_draws(const int64_t* seed, int64_t n_coord, int32_t n_neg, int32_t B, float* coords1, float* coords2,
                     int64_t* perms, stego_stream_t stream);

/* Kernel launches the forward needs for these maps: 1 = the fused path, 3 = sample / tile / finalize, 7 or 8 = S > 11 (csrc/corr_wide.hip);
 * < 0: -error code. */
int stego_corr_fwd_launches(const StegoCorrDesc* desc, const StegoMap* feats, const StegoMap* feats_pos,
                            const StegoMap* code, const StegoMap* code_pos);
int stego_corr_workspace_prepare(const StegoCorrDesc* desc, void* workspace, size_t workspace_bytes, stego_stream_t stream);
/* DEVICE address of two 32-bit event words inside a forward workspace, cumulative since it was prepared: [0] tiles of the single-launch
 * forward that gave up waiting for their anchor set and sampled it themselves, [1] negative tiles whose old_mean rendezvous timed out
 * and were finished by the launch's last workgroup.  Both are zero in normal operation; on a device shared with other kernels (or
 * partitioned) they say why a launch was ~30 us slower - results are the same either way.  No launch, no synchronisation: the caller
 * copies the words when it wants them.  NULL on a bad descriptor / workspace. */
const uint32_t* stego_corr_event_counters(const StegoCorrDesc* desc, const void* workspace, size_t workspace_bytes);
/* The same on a stream of the library's own, returning when the workspace IS prepared - without calling any HIP synchronisation API
 * (it watches a pinned flag word that a one-thread kernel sets behind the memset), so it is legal while the calling thread captures
 * a graph on another stream: there stego_corr_workspace_prepare(capture stream) would become a memset node that every replay repeats in
 * front of the forward.  The side stream and the flag are created by the first call on a device, which must not happen during a
 * capture: a host that captures calls stego_corr_workspace_prepare_now(NULL, NULL, 0) once at start-up (stego_amd does when it loads
 * the library).  This is the one entry point that allocates (once per device) and blocks. */
int stego_corr_workspace_prepare_now(const StegoCorrDesc* desc, void* workspace, size_t workspace_bytes);
int stego_corr_fwd_prepared(const StegoCorrDesc* desc,
                   const StegoMap* feats, const StegoMap* feats_pos,
                   const StegoMap* code, const StegoMap* code_pos,
                   const float* coords1, const float* coords2, const int64_t* perms,
                   float* loss_means,
                   float* pos_intra_cd, float* pos_inter_cd,
                   float* neg_inter_loss, float* neg_inter_cd,
                   float* saved_w, float* saved_mean, void* saved_ctx,
                   void* workspace, size_t workspace_bytes, stego_stream_t stream);

/*
 * Measurement hook: exactly stego_corr_fwd_prepared, run `iters` times with HIP events recorded on `stream` around each launch;
 * ms_kernels[3] (host) receives the mean durations in milliseconds after synchronising: fused path { 0, corr_fused_kernel, 0 },
 * three-launch path { sample_norm_kernel, corr_tile_kernel, corr_finalize_kernel }.  bench.py derives roofline.achieved from these.
 */
int stego_corr_fwd_profile(const StegoCorrDesc* desc,
                           const StegoMap* feats, const StegoMap* feats_pos,
                           const StegoMap* code, const StegoMap* code_pos,
                           const float* coords1, const float* coords2, const int64_t* perms,
                           float* loss_means,
                           float* pos_intra_cd, float* pos_inter_cd,
                           float* neg_inter_loss, float* neg_inter_cd,
                           float* saved_w, float* saved_mean, void* saved_ctx,
                           void* workspace, size_t workspace_bytes, stego_stream_t stream,
                           int32_t iters, float* ms_kernels);

/*
 * Backward of the above w.r.t. orig_code / orig_code_pos (what autograd derives through
 * modules.py:335-347,369-391: clamp mask, the two code GEMM adjoints, normalize backward, the
 * adjoint of the bilinear sampling incl. the orig_code[perm] gather of :385).  Two launches, no
 * global atomics.  Everything about the inputs comes from the forward's saved_w / saved_mean /
 * saved_ctx; only perms is passed again (the cd outputs are accepted for ABI stability and may be NULL: the clamp
 * mask travels inside saved_w).
 *
 *   g_intra, g_inter : device scalars, upstream of loss_means[0], loss_means[1] (NULL -> 0)
 *   g_neg_loss       : upstream of neg_inter_loss; g_neg_loss_stride = 1 -> dense
 *                      [n_neg*B,S^4], 0 -> one broadcast device scalar (the per-element value .mean() feeds),
 *                      -1 -> one device scalar that is the upstream of loss_means[2] (the kernel spreads it over the
 *                      n_neg*B*S^4 elements); NULL -> zero
 *   g_intra_cd, g_inter_cd, g_neg_cd : optional dense upstreams of the cd outputs (NULL -> 0)
 *   d_code, d_code_pos : OUT, channels-last dense [B,H,W,K] (i.e. grad.permute(0,2,3,1)), overwritten.
 *   workspace        : stego_corr_bwd_workspace_bytes()
 */
int stego_corr_bwd(const StegoCorrDesc* desc,
                   const int64_t* perms,
                   const float* saved_w, const float* saved_mean, const void* saved_ctx,
                   const float* pos_intra_cd, const float* pos_inter_cd, const float* neg_inter_cd,
                   const float* g_intra, const float* g_inter,
                   const float* g_neg_loss, int32_t g_neg_loss_stride,
                   const float* g_intra_cd, const float* g_inter_cd, const float* g_neg_cd,
                   float* d_code, float* d_code_pos,
                   void* workspace, size_t workspace_bytes, stego_stream_t stream);

/*
 * ContrastiveCorrelationLoss.helper (modules.py:325-347) on ALREADY SAMPLED tensors:
 *   f1,f2 : [N,C,S1,S2]   c1,c2 : [N,K,S1,S2]   (desc->B = N, desc->H = S1, desc->W = S2,
 *   desc->S is ignored, S1*S2 <= 128);  shift = desc->pos_intra_shift.
 * outputs: loss, cd : [N,S1,S2,S1,S2]; saved_w/saved_mean/saved_ctx as above with one pair-set
 * (sizes from the stego_corr_helper_* functions).
 */
int stego_corr_helper_fwd(const StegoCorrDesc* desc,
                          const StegoMap* f1, const StegoMap* f2,
                          const StegoMap* c1, const StegoMap* c2,
                          float* loss, float* cd, float* saved_w, float* saved_mean, void* saved_ctx,
                          void* workspace, size_t workspace_bytes, stego_stream_t stream);

/* Backward of helper w.r.t. c1, c2 (dense upstreams g_loss / g_cd, either may be NULL).
 * d_c1, d_c2: OUT channels-last dense [N,S1,S2,K], overwritten. */
int stego_corr_helper_bwd(const StegoCorrDesc* desc,
                          const float* saved_w, const float* saved_mean, const void* saved_ctx, const float* cd,
                          const float* g_loss, const float* g_cd,
                          float* d_c1, float* d_c2,
                          void* workspace, size_t workspace_bytes, stego_stream_t stream);

/*
 * All-pairs cosine top-k of precompute_knns.py:86-96 (`topk(einsum("nf,mf->nm", blk, X), 30)[1]`, all row blocks at
 * once) without materialising the [N,N] similarity matrix.
 *   X        : [N, D] fp32, row stride ldx elements (device)
 *   normalize: != 0 -> rows are L2-normalised first (F.normalize, eps 1e-12: precompute_knns.py:19)
 *   q_begin, q_count : the query rows to answer, q_begin a multiple of 128 (row-sharding across GPUs: every rank
 *                      holds the whole X and answers its own slice)
 *   out_idx  : OUT int64 [q_count, k], neighbours by descending similarity (rank 0 = the row itself unless a
 *              duplicate row ties with it; torch.topk leaves tie order unspecified, so does this)
 *   out_sims : OUT fp32 [q_count, k] or NULL
 * Limits: 1 <= k <= 32, k <= N < 2^31.  Contraction arithmetic: split-fp16 (3 MFMAs, fp32 accumulate), ~2e-7 abs.
 */
size_t stego_knn_workspace_bytes(int64_t N, int32_t D, int32_t k, int64_t q_count);
int stego_knn_topk(const float* X, int64_t N, int32_t D, int64_t ldx, int32_t k, int32_t normalize,
                   int64_t q_begin, int64_t q_count, int64_t* out_idx, float* out_sims,
                   void* workspace, size_t workspace_bytes, stego_stream_t stream);

/*
 * Dense feature correspondence, tensor_correlation() of modules.py:283-284:
 *   out[n,h,w,i,j] = sum_c a[n,c,h,w] * b[n,c,i,j]        a: [B,C,H1,W1]   b: [B,C,H2,W2]   out: [B,H1,W1,H2,W2] contiguous
 * normalize != 0 applies norm() (:275-276, F.normalize over C with eps 1e-10) to both maps first, which is how
 * plot_dino_correspondence.py:39-58 and plot_pr_curves.py:108-121 call it.  Forward only (those callers are
 * inference); products as fp16 hi+lo splits on the matrix cores, fp32 accumulate (fp32-grade, see STEGO_PREC_F16X3).
 */
size_t stego_dense_corr_workspace_bytes(int32_t B, int32_t C, int32_t H1, int32_t W1, int32_t H2, int32_t W2);
int stego_dense_corr(const StegoMap* a, const StegoMap* b, int32_t B, int32_t C, int32_t H1, int32_t W1, int32_t H2,
                     int32_t W2, int32_t normalize, float* out, void* workspace, size_t workspace_bytes,
                     stego_stream_t stream);

/* ---- sample() with an image index (reference: sample(), src/modules.py:287-288, as ContrastiveCorrelationLoss.forward calls it on
 * orig_feats[perm] / orig_code[perm], :384-385 - without the permuted copy of the maps).
 *   out[n][p][c] (channels-last rows, p = h * S + w) = bilinear sample (align_corners = True, border padding) of map[index[n]] (index
 *   NULL: map[n]) at coords[n % n_coords][w][h][:] ([..., 0] = x, [..., 1] = y in [-1, 1]).
 * stego_sample_bwd adds the adjoint into d_map (same shape and strides as the forward's map; the caller zeroes it): fp32 atomic adds,
 * so the summation order is not fixed.  Any strides; 16-byte loads when the map is channels-last.  Used by the loss for shapes the fused
 * kernels do not take (dim > 128: stego_amd.modules.ContrastiveCorrelationLoss.generic_forward). */
int stego_sample(const StegoMap* map, const int64_t* index, int32_t N, int32_t C, int32_t H, int32_t W, const float* coords,
                 int32_t n_coords, int32_t S, float* out, stego_stream_t stream);
int stego_sample_bwd(const float* g_out, const StegoMap* d_map, const int64_t* index, int32_t N, int32_t C, int32_t H, int32_t W,
                     const float* coords, int32_t n_coords, int32_t S, stego_stream_t stream);

/* ---- the sampled points as prepared OPERANDS of the dense-correspondence kernel (the loss for feature_samples 12 .. 16 / dim > 128: what
 * the reference computes as tensor_correlation(norm(sample(t[perm], coords)), ...), src/modules.py:328,335,384-385, without the fp32 rows
 * of the sampled features ever existing).  An operand image = [128-point block][64-channel chunk][hi | lo][128 rows][72 fp16] - the row's
 * values times a power-of-two scale, split into fp16 hi + lo - stego_panel_image_bytes(C, P) bytes per image, with a row-scale array of
 * ceil(P / 128) * 128 floats per image beside it; the caller allocates both (16-byte aligned) for as many images as it samples.
 *   stego_sample_panels: image n of the set = the S * S points of map[index[n]] (NULL: map[n]) at coords[n % n_coords], L2-normalised over
 *     the channels when `normalize` (eps 1e-10, F.normalize).  rows_out (optional): the normalised fp32 rows [N][P][C]; inv_out (optional):
 *     1 / max(|row|, eps) [N][P] - what the backward of norm() needs for tensors that carry a gradient.
 *   stego_dense_corr_panels: out[n][i][j] = <A image n % images_a, point i ; B image n, point j> (M x Ncols per pair, fp32, contiguous):
 *     one set of anchors against the second operands of several pair-sets.  The two operand sets may be the same buffer.  rowsum (optional,
 *     [N][M]): sum_j out[n][i][j], what the pointwise shift of helper() (src/modules.py:332) needs of fd.
 *   stego_sample_bwd_rows: the adjoint of the sampling for the gradient g_rows [N][P][C] of those rows, ADDED into d_map (fp32 atomics, as
 *     stego_sample_bwd); with rows_n / inv (what stego_sample_panels wrote) g_rows is the gradient of the NORMALISED rows and the backward of
 *     norm() is applied first: d row = inv * (g - rows_n <rows_n, g>), inv * g where the row's norm was below eps. */
size_t stego_panel_image_bytes(int32_t C, int32_t P);
int stego_sample_bwd_rows(const float* g_rows, const float* rows_n, const float* inv, const StegoMap* d_map, const int64_t* index, int32_t N,
                          int32_t C, int32_t H, int32_t W, const float* coords, int32_t n_coords, int32_t S, stego_stream_t stream);
int stego_sample_panels(const StegoMap* map, const int64_t* index, int32_t N, int32_t C, int32_t H, int32_t W, const float* coords,
                        int32_t n_coords, int32_t S, int32_t normalize, void* panels, float* row_scale, float* rows_out, float* inv_out,
                        stego_stream_t stream);
int stego_dense_corr_panels(const void* panels_a, const float* row_scale_a, int32_t images_a, const void* panels_b, const float* row_scale_b,
                            int32_t N, int32_t C, int32_t M, int32_t Ncols, float* out, float* rowsum, stego_stream_t stream);

/* ---- the elementwise part of helper() (src/modules.py:330-345) over the correlation tensors of ALL pair-sets at once, for shapes the
 * fused kernels do not take: fd, cd = [n_sets][B][P][P] contiguous (set 0 intra, 1 inter, 2.. negatives; shift[min(set, 2)]).
 *   stego_rowsum: out[r] = sum_j x[r][j] (rows x P contiguous).  The caller reduces rowsum to the per-set old_mean = sum / (B P P).
 *   _fwd: loss = -clamp(cd, clamp_min, clamp_max) * ((fd - rowsum / P) + old_mean[set] - shift) (pointwise = 0: fd - shift); writes the
 *         loss of the negative sets to neg_loss [n_sets - 2][B][P][P] and the row sums of the loss of every set to loss_rowsum [n_sets][B][P].
 *   _bwd: g_cd = -((fd - rowsum / P) + old_mean - shift) * 1[clamp_min <= cd <= clamp_max] * (g_sums[set] + (set >= 2 ? g_neg_bcast[0] +
 *         g_neg_loss : 0)); g_neg_loss (dense), g_neg_bcast (ONE device float: an expanded scalar upstream) and g_sums (device [n_sets])
 *         are the upstreams of the two outputs, each may be NULL. */
int stego_rowsum(const float* x, int64_t rows, int32_t P, float* out, stego_stream_t stream);
int stego_loss_pointwise_fwd(const float* fd, const float* cd, const float* rowsum, const float* old_mean, int32_t n_sets, int32_t B, int32_t P,
                             const float shift[3], float clamp_min, float clamp_max, int32_t pointwise, float* neg_loss, float* loss_rowsum,
                             stego_stream_t stream);
int stego_loss_pointwise_bwd(const float* fd, const float* cd, const float* rowsum, const float* old_mean, int32_t n_sets, int32_t B, int32_t P,
                             const float shift[3], float clamp_min, float clamp_max, int32_t pointwise, const float* g_neg_loss,
                             const float* g_neg_bcast, const float* g_sums, float* g_cd, stego_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* STEGO_CORR_H */
