/*
 * stego_vit.h - C ABI of the frozen DINO ViT backbone forward (the step right before the
 * correspondence-loss hot path), exported by the same libstego_corr.so.
 *
 * Replaces, for inference on a frozen backbone:
 *   src/dino/vision_transformer.py:225-237  VisionTransformer.get_intermediate_feat(x, n=1)  (feat only)
 *   src/dino/vision_transformer.py:195-205  prepare_tokens  (patch embed conv :123-133, cls token, pos embed)
 *   src/dino/vision_transformer.py:69-130   Attention / Mlp / Block  (qkv bias, softmax(QK^T/sqrt(d))V, proj, GELU MLP)
 * as called by DinoFeaturizer.forward (src/modules.py:83-107) and precompute_knns.get_feats (:15-21).
 *
 * Arithmetic: residual stream, LayerNorm statistics, softmax statistics and every accumulation are fp32; GEMM and
 * attention operands go to the fp16 matrix cores - in precision STEGO_VIT_F16X3 as hi + lo pairs with three MFMAs per product
 * (x = hi + lo to 2^-22: the fp32 class the reference's fp32 torch model computes in, and what the loss kernels and the
 * segmentation head of this library use), in STEGO_VIT_F16 as plain fp16 operands (2 - 3 x faster, error at the level of
 * torch's fp16 autocast).  Measured deviations from the fp32 / fp64 torch model: DESIGN.md 4.9 / tests/test_vit_native.py.
 * The attention maps and the qkv tensor the reference materialises at every block (:232-236) are never built.
 *
 * Conventions as in stego_corr.h: device pointers, nothing allocated / freed / synchronised, work enqueued on
 * `stream`, return STEGO_OK or an error code.
 */
#ifndef STEGO_VIT_H
#define STEGO_VIT_H

#include "stego_corr.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct StegoVitDesc {
    int32_t B;        /* images per call                                                     */
    int32_t H, W;     /* image size, multiples of `patch`                                    */
    int32_t patch;    /* 8 or 16                     (vision_transformer.py:123-133)         */
    int32_t D;        /* embed dim, multiple of 64:  192 / 384 / 768 (vit_tiny/small/base)   */
    int32_t depth;    /* blocks                                                              */
    int32_t heads;    /* D / heads must be 64                                                */
    int32_t hidden;   /* MLP hidden width (4 * D), multiple of 64                            */
    int32_t precision;/* STEGO_VIT_F16 | STEGO_VIT_F16X3 (ABI 6); weights packed for one precision   */
                      /* are read by forwards of the same precision only                            */
} StegoVitDesc;

enum { STEGO_VIT_F16 = 0, STEGO_VIT_F16X3 = 1 };

/* Number of fp32 parameter tensors stego_vit_pack_weights() takes: 4 + 12 * depth + 2, in this order
 *   patch_embed.proj.weight [D, 3*patch*patch]   patch_embed.proj.bias [D]   cls_token [D]
 *   pos_embed [1 + (H/patch)*(W/patch), D]   (already interpolated to this H, W: vision_transformer.py:171-193)
 *   per block: norm1.weight, norm1.bias, attn.qkv.weight [3D, D], attn.qkv.bias [3D], attn.proj.weight [D, D],
 *              attn.proj.bias, norm2.weight, norm2.bias, mlp.fc1.weight [hidden, D], mlp.fc1.bias,
 *              mlp.fc2.weight [D, hidden], mlp.fc2.bias
 *   norm.weight, norm.bias
 * all contiguous fp32 on the device (the layouts of the DINO checkpoints / nn.Linear). */
int32_t stego_vit_param_count(const StegoVitDesc* d);

/* Packs the weights once into the layout the kernels read (fp16 operand panels + fp32 vectors). */
size_t stego_vit_weights_bytes(const StegoVitDesc* d);
int stego_vit_pack_weights(const StegoVitDesc* d, const float* const* params, int32_t n_params, void* packed,
                           size_t packed_bytes, stego_stream_t stream);

/* tokens_out: OUT fp32 [B, 1 + hw, D] = norm(x) after the last block (the `feat[0]` of get_intermediate_feat(n=1));
 * row 0 of every image is the class token (DinoFeaturizer return_class_feat), rows 1.. are the patch tokens, i.e.
 * tokens_out[:, 1:, :] viewed as [B, h, w, D] is the channels-last feature map the loss kernels read directly.
 * img: fp32 [B, 3, H, W] contiguous. */
size_t stego_vit_workspace_bytes(const StegoVitDesc* d);
int stego_vit_forward(const StegoVitDesc* d, const void* packed, const float* img, float* tokens_out, void* workspace,
                      size_t workspace_bytes, stego_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* STEGO_VIT_H */
