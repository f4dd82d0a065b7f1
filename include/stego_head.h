/*
 * stego_head.h - C ABI of the segmentation head of DinoFeaturizer (the producer of `code`, the tensor the
 * correspondence loss differentiates), exported by the same libstego_corr.so.
 *
 * Replaces, forward and backward, src/modules.py:108-116 of the reference:
 *     code = self.cluster1(self.dropout(image_feat))                       # Conv2d(C, K, 1x1)               :70-72
 *     code += self.cluster2(self.dropout(image_feat))                      # Conv2d(C, C) -> ReLU -> Conv2d(C, K)   :74-78
 *     return self.dropout(image_feat), code                                # nn.Dropout2d(p = .1)            :33, :116
 * on the channels-last token matrix the frozen backbone emits (modules.py:97: feat[:, 1:, :] viewed as [B, C, h, w]).  The 1x1
 * convolutions are GEMMs over the tokens; nn.Dropout2d is a per-(image, channel) scale: the three masks are inputs (the host draws
 * them with the very torch calls F.dropout2d makes, so a seeded run consumes the generator like the reference), applied to the
 * token operand while it is staged - the dropped-out copies of the feature map are never written.  The backbone is frozen
 * (modules.py:30-31), so the backward produces the six parameter gradients only.
 *
 * Arithmetic: every product is an fp32 operand pair split into fp16 hi + lo (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, fp32
 * accumulate) - the fp32-class scheme of the loss kernels (STEGO_PREC_F16X3); biases, ReLU and sums in fp32.
 *
 * Conventions as in stego_corr.h: device pointers, nothing allocated / freed / synchronised, work enqueued on `stream`, STEGO_OK or
 * an error code.
 */
#ifndef STEGO_HEAD_H
#define STEGO_HEAD_H

#include "stego_corr.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct StegoHeadDesc {
    int32_t B;            /* images                                                                          */
    int32_t HW;           /* tokens per image (h * w)                                                        */
    int32_t C;            /* feature channels, a multiple of 32 (384 ViT-S, 768 ViT-B)                       */
    int32_t K;            /* code channels = cfg.dim, <= 128                                                 */
    int32_t nonlinear;    /* cfg.projection_type == "nonlinear": cluster2 exists (modules.py:74-78)          */
    int64_t tok_stride;   /* elements between consecutive tokens of an image (C for the backbone's output)   */
    int64_t img_stride;   /* elements between images ((1 + HW) * C: the class token is skipped by the caller) */
    const uint32_t* tokens_amax;   /* ABI 4, optional: device word holding the float bits of max |tokens| over exactly the elements the
                           * forward reads (stego_tokens_from_cache writes it while it produces the tokens); NULL = the forward makes its
                           * own pass over the tokens for it (77 MB at 2B = 64) */
} StegoHeadDesc;

/* Forward.
 *   tokens            : image_feat as tokens: element (b, t, c) at tokens[b * img_stride + t * tok_stride + c]
 *   mask1, mask2, mask3 : [B, C] channel scales of the three nn.Dropout2d draws (0 or 1 / (1 - p)), in the reference's order:
 *                       cluster1's input, cluster2's input, the returned feature map; NULL = no dropout (eval / cfg.dropout off)
 *   w1 [K, C], b1 [K]  : cluster1[0];  w21 [C, C], b21 [C] : cluster2[0];  w22 [K, C], b22 [K] : cluster2[2]  (Conv2d weights
 *                       viewed as [out, in]; the cluster2 pointers are ignored unless desc->nonlinear)
 * outputs
 *   code              : [B, HW, K] dense (= the [B, K, h, w] code map channels-last, the layout the loss kernels gather from)
 *   feats_out         : [B, HW, C] dense = tokens * mask3, or NULL (then nothing is written: the caller returns the tokens)
 *   saved_h           : [B * HW * C + 8] floats: relu(cluster2[0](dropout(x))) dense [B, HW, C], kept for the backward, followed by
 *                       8 words of operand scales (the largest magnitudes of x, the weights and H: every fp32 operand is multiplied
 *                       by a power of two before it is split into fp16 halves, so that small values keep 22 bits); NULL when no
 *                       backward follows (with desc->nonlinear H then lives in `workspace`)
 *   workspace         : stego_head_fwd_workspace_bytes() (scale words, the weights as fp16 hi / lo planes, H; with saved_h given the
 *                       H part - B * HW * C floats rounded up to 256 bytes, the last - may be left out) */
size_t stego_head_fwd_workspace_bytes(const StegoHeadDesc* desc);
int stego_head_fwd(const StegoHeadDesc* desc, const float* tokens, const float* mask1, const float* mask2, const float* mask3,
                   const float* w1, const float* b1, const float* w21, const float* b21, const float* w22, const float* b22,
                   float* code, float* feats_out, float* saved_h, void* workspace, size_t workspace_bytes, stego_stream_t stream);

/* Backward w.r.t. the six parameters (what autograd derives through modules.py:108-112; the token side gets no gradient).
 *   d_code            : [B, HW, K] dense upstream of `code` (what stego_corr_bwd writes, channels-last)
 *   tokens, mask1, mask2, saved_h, w22 : as in the forward
 * outputs (overwritten): dw1 [K, C], db1 [K], dw21 [C, C], db21 [C], dw22 [K, C], db22 [K]  (cluster2's are untouched unless nonlinear)
 *   workspace         : stego_head_bwd_workspace_bytes() */
size_t stego_head_bwd_workspace_bytes(const StegoHeadDesc* desc);
int stego_head_bwd(const StegoHeadDesc* desc, const float* tokens, const float* mask1, const float* mask2, const float* saved_h,
                   const float* w22, const float* d_code, float* dw1, float* db1, float* dw21, float* db21, float* dw22,
                   float* db22, void* workspace, size_t workspace_bytes, stego_stream_t stream);

/* Tokens of a cached backbone (stego_amd TokenCache: the frozen backbone's output for a fixed-crop dataset, kept in HBM as fp16) for
 * the dataset items `index`: out[i] = float(table[index[i]]), [n][ntok][D] dense, in one pass that also leaves the largest magnitude
 * of the rows >= skip_rows of every item (skip_rows = 1: the class token, which the head does not read) in *amax_bits - the word
 * StegoHeadDesc.tokens_amax takes.  D a multiple of 8; amax_bits may be NULL. */
int stego_tokens_from_cache(const void* table_f16, const int64_t* index, int32_t n, int32_t ntok, int32_t D, int32_t skip_rows,
                            float* out, uint32_t* amax_bits, stego_stream_t stream);

/* The dropout masks themselves, as torch draws them: nn.Dropout2d (modules.py:33, :109-114) = F.dropout2d -> ATen feature dropout:
 *     noise = x.new_empty(B, C, 1, 1).bernoulli_(1 - p).div_(1 - p)
 * per call.  n_masks consecutive calls from the device generator's state (seed, offset) - or, with seed_ptr / offset_ptr set, from the
 * graph-safe state read on the device (offset = *offset_ptr + offset; see stego_ref_draws_indirect) - in ONE launch, bit for bit: Philox
 * counter layout and grid policy of ATen's distribution template, curand_uniform4's conversion, value < (float)(1 - p), times the
 * float 1 / (1 - p).  masks: [n_masks][numel] floats (numel = B * C).  The caller advances the generator by
 * stego_ref_dropout_masks_advance().  `variant` bits 0 / 1 as for stego_ref_draws; the host checks the result against the real torch
 * calls once per process and size and keeps the torch calls if no variant matches. */
int stego_ref_dropout_masks(uint64_t seed, uint64_t offset, const int64_t* seed_ptr, const int64_t* offset_ptr, int32_t variant,
                            int32_t n_masks, int64_t numel, float keep_prob, float* masks, stego_stream_t stream);
uint64_t stego_ref_dropout_masks_advance(int64_t numel, int32_t n_masks, int32_t variant);

#ifdef __cplusplus
}
#endif
#endif /* STEGO_HEAD_H */
