// Tiled MFMA contraction  C[M,N] = A[M,K] · W[N,K]^T  with fused prologue gathers and epilogues.
// One kernel serves every dense contraction on the CellViT hot path:
//   * nn.Linear layers of both encoders (QKV / proj / fc1 / fc2 / heads)      — A linear
//   * patch-embed conv k16 s16 (after patchify)                               — A linear
//   * Conv2d 3x3 p1 of the U-Net decoder as an im2col-FREE implicit GEMM      — A gathered from
//     one or two NHWC sources (the skip ‖ upsampled concat is never materialised)
//   * ConvTranspose2d k2 s2 as one GEMM with a pixel-shuffle scatter epilogue
#pragma once
#include "common.h"

namespace cva {

enum : int { A_LINEAR = 0, A_CONV3 = 1 };
enum : int { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2 };
enum : int { OUT_LINEAR = 0, OUT_QKV = 1, OUT_CONVT = 2 };

struct GemmParams {
    // ---- problem ----
    int M, N, K;            // K = logical contraction length (multiple of the 16-B piece)
    const void* A;          // A_LINEAR: [rows, lda] of T.  A_CONV3: source 1 (NHWC, C1 channels)
    const void* A2;         // A_CONV3: source 2 (NHWC, C2 channels) or null
    const void* W;          // [N, ldw] of T, K-contiguous (nn.Linear layout), zero padded to ldw
    int lda, ldw;
    // A_LINEAR row remap: arow = m + (m / a_rpi) * a_extra + a_off   (a_rpi == 0: identity)
    int a_rpi, a_extra, a_off;
    // A_CONV3 geometry: M = B*H*W output pixels, K = 9*(C1+C2)
    int H, Wd, C1, C2;
    // ---- epilogue ----
    const float* bias;      // [N] or null
    int act;
    const float* res;       // fp32 residual [*, ldres] or null; row = res_mod ? m % res_mod : orow
    int ldres, res_mod;
    int out_mode;
    int out_f32;            // 1: store fp32, 0: store T
    void* out;              // OUT_LINEAR: [rows, ldc];  OUT_CONVT: NHWC [B, 2H, 2W, N/4]
    int ldc;
    // OUT_LINEAR row remap: orow = m + (m / o_rpi) * o_extra + o_off (o_rpi == 0: identity)
    int o_rpi, o_extra, o_off;
    // OUT_QKV: scatter q,k -> [S*heads, L, hd], v -> V^T [S*heads, hd, Lp]
    void* q_out; void* k_out; void* vt_out;
    int D, hd, heads, ntok, L, Lp;   // ntok tokens per image (incl. cls for ViT)
    int n_off;                       // OUT_QKV: column n of this launch is column n + n_off of the fused qkv projection
    int win, gw, gh, nwx, nwy;       // win > 0: window partition of the gh x gw token grid
    const void* zero;                // >= 16 B of zeros in device memory (filled in by launch_gemm)
    // conv3x3 halo kernel only: fused 1x1 output head (cellvit.py:309-315) on the ReLU output; the 64-channel
    // activation itself is then not written (out may be null).  logits fp32 NCHW [B, nout, H, W], argmax u8 [B, H, W].
    const float* head_W; const float* head_b; float* head_logits; uint8_t* head_argmax; int head_nout, head_narg;
    int epi_vec;                     // 1: LDS-staged 16-byte epilogue stores (set by launch_gemm)
    int dbg;                         // experiment switches (CVA_GEMM_DBG): 1 no staging, 2 no LDS reads, 4 no L2 prefetch
};

template <typename T> int launch_gemm(const GemmParams& p, int a_mode, hipStream_t stream);

// gemm8.hip: 256x256x64 8-phase fp16 kernel (A_LINEAR only); launch_gemm routes to it when the shape qualifies.
bool gemm8_supported(const GemmParams& p, int a_mode, size_t elem_size);
int launch_gemm8(const GemmParams& p, hipStream_t stream);

// >= 256 B of zeros in device memory (DMA source for out-of-range pieces)
void* gemm_zero_page();

// conv.hip: halo-tiled direct 3x3 convolution (fp16); GemmParams as for A_CONV3.  Returns -1 if the layer does not fit.
int launch_conv3x3_halo(const GemmParams& p, int batch, hipStream_t stream);

}  // namespace cva
