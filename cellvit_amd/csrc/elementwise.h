// Bandwidth-bound helper kernels of the forward pass (layout changes, LayerNorm, heads).
#pragma once
#include "common.h"

namespace cva {

// LayerNorm over the last dim (eps inside sqrt, biased variance): nn.LayerNorm(eps=1e-6) of both
// encoders (cellvit.py:99, 559) and LayerNorm2d of the SAM neck on NHWC data (SAM/utils.py:38-50).
// in: fp32 rows [M, C] at stride ld_in; out: T or fp32 [M, C] contiguous.
template <typename T>
int launch_layernorm(const float* in, long ld_in, const float* gamma, const float* beta, void* out,
                     int out_f32, int M, int C, float eps, hipStream_t stream);
// fp16 engine: x_io[M, C] (fp32, row stride ld) += delta[M, C] (fp16); out[M, C] (fp16, may alias delta) = LayerNorm(x_io)
int launch_layernorm_add(float* x_io, long ld, const void* delta, const float* gamma, const float* beta, void* out,
                         int M, int C, float eps, hipStream_t stream);

// fp8 engine: LayerNorm whose result is the OCP MX-fp8 A operand of the next linear layer: out8 [M, C] e4m3 bytes + E8M0
// block scales in the fragment order of gemm8.hip (sc_a: A-side image, sc_w: W-side image or null).  delta_f16 != null:
// x_io += delta first (written back), as launch_layernorm_add.  M % 256 == 0, C % 128 == 0.
int launch_layernorm_mx8(float* x_io, long ld, const void* delta_f16, const float* gamma, const float* beta, void* out8, void* sc_a,
                         void* sc_w, int M, int C, float eps, hipStream_t stream);

// x fp32 NCHW [B,3,H,W] -> patch matrix [B*(H/16)*(W/16), 768] of T, k = c*256 + ky*16 + kx
// (the flattening of Conv2d(3, D, 16, 16).weight — vits_histo.py:273-280, image_encoder.py:418-426).
// Raw-tile input (F0 fused): x uint8 NHWC [B,H,W,3]; every consumer of the image evaluates the reference's inference
// transform (ToTensor + Normalize, cell_detection.py:214-227) on the fly: (u8 / 255 - mean[c]) / std[c] in fp32.
struct InputU8 { const uint8_t* x; float mean[3]; float stdv[3]; };
// u8 == nullptr: x is the normalised fp32 NCHW batch; otherwise x is ignored and the tile is read through *u8.
template <typename T> int launch_patchify(const float* x, const InputU8* u8, void* out, int B, int H, int W, hipStream_t stream);

// x fp32 NCHW [B,3,H,W] -> NHWC [B,H,W,8] of T, channels 3..7 zero (decoder0 input, cellvit.py:182,241).
template <typename T> int launch_nchw3_to_nhwc8(const float* x, const InputU8* u8, void* out, int B, int H, int W, int CP, hipStream_t stream);

// argmax over the channels of an fp32 NCHW map -> u8 [B, H*W] (first maximum, as torch.argmax; cellvit.py:366-374)
// fp16 packed 3x3 filter [Cout][tap][Ctot] -> [Cout][Ctot/64][tap][64] (K order of the implicit-GEMM convolution); Ctot % 64 == 0
int launch_conv_w_kmajor(const void* in, void* out, int Cout, int Ctot, hipStream_t stream);

int launch_argmax_nchw(const float* x, uint8_t* out, int B, int C, long hw, hipStream_t stream);
// u8 NHWC -> normalised fp32 NCHW [B,3,H*W] (the tensor the reference's DataLoader hands to model.forward)
int launch_normalize_u8(const InputU8& u8, float* out, int B, long hw, hipStream_t stream);
// cell-token pooling (cell_detection.py:396-409): out[rec_off[b] + slot, :] = mean of tokens_nhwc[b, r0:r1, c0:c1, :] for
// record slot < n_recs[b] of tile b (records: stride rec_stride bytes, leading int32 id, rmin, cmin, rmax, cmax)
int launch_pool_tokens(const float* tokens_nhwc, const void* recs, int rec_stride, int max_inst, const int32_t* n_recs,
                       const int64_t* rec_off, int B, int max_n, int gh, int gw, int D, int patch, float* out, hipStream_t stream);

// fp32 token rows -> T rows, optionally dropping a leading cls row per image (cellvit.py:186-189).
// in: [B, rpi_in, C] (rpi_in = ntok incl. cls); out: [B, ntok_out, C] with ntok_out = rpi_in - skip.
template <typename T>
int launch_cast_tokens(const float* in, void* out, int B, int rpi_in, int skip, int C, hipStream_t stream);

// fp32 -> T elementwise copy
template <typename T> int launch_cast(const float* in, void* out, long n, hipStream_t stream);

// ViT cls row: out[b*ntok + 0, :] = cls + pos[0]   (vits_histo.py:408-413)
int launch_cls_rows(const float* cls, const float* pos0, float* tokens, int B, int ntok, int C, hipStream_t stream);

// mean over rows per image: in fp32 [B, R, C] -> out fp32 [B, C]   (utils.py:231-233)
int launch_mean_rows(const float* in, float* out, int B, int R, int C, hipStream_t stream);

// 1x1 output conv of a decoder branch (cellvit.py:309-315) fused with the NHWC->NCHW permute:
// feat T [B*H*W, 64] · Wt[n_out, 64] + b -> logits fp32 [B, n_out, H*W].
// argmax_out (u8 [B*H*W]) optional: argmax over the first n_arg channels (== argmax of softmax,
// cellvit.py:369-374 — softmax is monotone, ties resolve to the lowest index as torch.argmax).
template <typename T>
int launch_head1x1(const void* feat, const float* Wt, const float* bias, float* logits, uint8_t* argmax_out,
                   int n_arg, long npix_per_img, int B, int n_out, hipStream_t stream);

}  // namespace cva
