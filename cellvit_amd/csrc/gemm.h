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
enum : int { OUT_LINEAR = 0, OUT_QKV = 1, OUT_CONVT = 2, OUT_MX8 = 3 };   // OUT_MX8: MX-fp8 rows + block scales (fp8 engine)

// ---- OCP MX-fp8 operands of the fp8 engine (gemm8.hip, F8 = 1) --------------------------------------------------------
// Elements: e4m3 (max 448), one E8M0 scale (2^(byte - 127)) per 32 consecutive K elements of a row.  Data [rows, K] bytes,
// row-major.  Scales live in 1-KiB blocks, one per (256-row tile, 128-element K tile) = 256 rows x 4 K blocks, at
// ((row / 256) * (K / 128) + k / 128) * 1024 + off(row % 256, (k % 128) / 32), where off is the order in which the MFMA
// lanes of gemm8_kernel want them (one ds_read_b32 = the scales of the four fragments of a sub-tile for lane (g, li)):
//   A side (operand in the "A" LDS tile, natural row order): fragment row = wr*128 + mh*64 + mi*16 + li
//   W side (operand in the "W" LDS tile, rows permuted for the transposed epilogue): source row = wc*64 + q*16 + nj*4 + r,
//          read by lane li = q*4 + r of fragment nj
__host__ __device__ inline int mx8_scale_off_a(int rr, int g) {
    return (((((rr >> 7) * 2 + ((rr >> 6) & 1)) * 4 + g) * 16 + (rr & 15)) << 2) | ((rr >> 4) & 3);
}
__host__ __device__ inline int mx8_scale_off_w(int rr, int g) {
    return (((((rr >> 6) * 4 + g) * 16) + (((rr >> 4) & 3) * 4 + (rr & 3))) << 2) | ((rr >> 2) & 3);
}
__host__ __device__ inline long mx8_scale_index(long row, int k, int K, bool w_side) {
    const long blk = (row >> 8) * (K >> 7) + (k >> 7);
    const int rr = (int)(row & 255), g = (k & 127) >> 5;
    return blk * 1024 + (w_side ? mx8_scale_off_w(rr, g) : mx8_scale_off_a(rr, g));
}

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
    int conv_kmajor;                 // filter K order: 0 = (tap, channel) as packed by pack_conv3, 1 = (64-channel chunk, tap, channel)
    int conv_wshift, conv_cshift;    // implicit 3x3 convolution on the 8-phase kernel: log2(W), log2((C1 + C2) / 64) (set by launch_gemm8_conv3)
    // launch_gemm8_deconv (ConvTranspose2d k2 s2 followed by Conv2d 3x3 as ONE contraction over the low-resolution input, see gemm8.hip):
    // bias of the output pixels on the image border, [9][N/4] indexed (row case * 3 + column case), case 0 = first, 1 = interior,
    // 2 = last row / column of the 2H x 2W output; `bias` holds the interior case replicated per output parity ([N]).
    const float* comp_bias;
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
    // OUT_QKV: scatter q,k -> [S*heads, L, hd], v -> V^T [S*heads, hd, Lp]  (v_rm != 0: v ROW-major [S*heads, L, hd] like k, into vt_out —
    // the layout the attention kernels with a transposing LDS read consume, attention.h AttnParams::v_rm)
    void* q_out; void* k_out; void* vt_out;
    int v_rm;
    int D, hd, heads, ntok, L, Lp;   // ntok tokens per image (incl. cls for ViT)
    int n_off;                       // OUT_QKV: column n of this launch is column n + n_off of the fused qkv projection
    int win, gw, gh, nwx, nwy;       // win > 0: window partition of the gh x gw token grid
    const void* zero;                // >= 16 B of zeros in device memory (filled in by launch_gemm)
    // conv3x3 halo kernel only: fused 1x1 output head (cellvit.py:309-315) on the ReLU output; the 64-channel
    // activation itself is then not written (out may be null).  logits fp32 NCHW [B, nout, H, W], argmax u8 [B, H, W].
    const float* head_W; const float* head_b; float* head_logits; uint8_t* head_argmax; int head_nout, head_narg;
    // fp8 engine (launch_gemm8_f8): A / W are e4m3 bytes, lda / ldw in elements (= bytes)
    const void* a_scale;             // activation scales, A-side layout (mx8_scale_index(.., false))
    const void* a_scale_w;           // the same scales in the W-side layout (needed by the swapped V tiles of OUT_QKV), or null
    const void* w_scale;             // weight scales; tiles that run swapped (qkv rows >= 2D) are packed in the A-side layout
    void* out_scale;                 // OUT_MX8: scales of the fp8 output (A-side layout for a consumer GEMM with K = N)
    // 8-phase kernel on shapes that are not multiples of its 256 x 256 tile (ViT-S: M = B * 4097 rows, N = 384 / 1152 columns): the caller
    // pads — M and N are the padded extents (operand rows exist up to M; W / bias are zero padded to N), rows >= m_valid and columns
    // >= n_valid are computed and dropped in the epilogue.  0 = no guard.  n_valid a multiple of 16; OUT_QKV: n_valid == 3 * D.
    int m_valid, n_valid;
    int epi_vec;                     // 1: LDS-staged 16-byte epilogue stores (set by launch_gemm)
    int dbg;                         // experiment switches (CVA_GEMM_DBG): 1 no staging, 2 no LDS reads, 4 no L2 prefetch
    int stagger;                     // experiment (ablation builds, CVA_GEMM_STAGGER): start delay spread in units of 10 ns; > 0 per XCD, < 0 per workgroup
    // 8-phase kernel, phase-shifted tile walk (gemm8.hip): fp32 scratch of 256 x 256 accumulators per workgroup, or null = every
    // workgroup walks whole tiles in lockstep.  Filled in by launch_gemm8 for the launches that profit.
    float* park;
};

template <typename T> int launch_gemm(const GemmParams& p, int a_mode, hipStream_t stream);

// gemm8.hip: 256x256x64 8-phase fp16 kernel (A_LINEAR only); launch_gemm routes to it when the shape qualifies.
bool gemm8_supported(const GemmParams& p, int a_mode, size_t elem_size);
int launch_gemm8(const GemmParams& p, hipStream_t stream);
#ifdef CVA_ABLATION
// experiments/gemm4.hip (experiment flavour of the library only): the same tile with ONE wave per SIMD (4 waves x 128 x 128 accumulators);
// shapes as gemm8_supported, out modes per gemm4_takes.  sched: slot placement variant.
bool gemm4_takes(const GemmParams& p);
int launch_gemm4(const GemmParams& p, int sched, hipStream_t stream);
#endif
// deconv.hip: ConvTranspose2d(k2, s2) o Conv2d 3x3 as a halo-tiled direct convolution (4 waves = the four output parities), for the
// composed stages with Cout < 256 that launch_gemm8_deconv cannot take.  Same GemmParams contract as launch_gemm8_deconv (+ p.zero);
// returns -1 when the layer does not fit.
bool deconv_halo4_supported(const GemmParams& p);
int launch_deconv_halo4(const GemmParams& p, int batch, hipStream_t stream);
// fp8 engine: the same kernel on MX-fp8 operands; out_mode OUT_LINEAR (fp32 / fp16 out, optional residual), OUT_QKV or
// OUT_MX8.  Returns hipErrorInvalidValue when the shape does not qualify (there is no fallback kernel for fp8 operands).
bool gemm8_f8_supported(const GemmParams& p);
int launch_gemm8_f8(const GemmParams& p, hipStream_t stream);

// >= 256 B of zeros in device memory (DMA source for out-of-range pieces)
void* gemm_zero_page();

// gemm8.hip: implicit 3x3 convolution on the 8-phase kernel (Cout a multiple of 256, 64-channel K steps, power-of-two image
// sides); GemmParams as for A_CONV3.  Returns -1 if the layer does not fit.
bool gemm8_conv3_supported(const GemmParams& p);
int launch_gemm8_conv3(const GemmParams& p, hipStream_t stream);

// gemm8.hip: ConvTranspose2d(k2, s2) -> Conv2d(3x3, pad 1) (-> ReLU) composed into one contraction over the INPUT pixels: M = B*H*W,
// N = 4*Cout (n = (py*2 + px)*Cout + co, the output parity), K = 4*C1 (the 2x2 input pixels an output parity sees), W packed
// [N][(64-channel chunk, ty-py, tx-px, channel)], OUT_CONVT scatter.  Cout a multiple of 256.  Returns -1 if the layer does not fit.
bool gemm8_deconv_supported(const GemmParams& p);
int launch_gemm8_deconv(const GemmParams& p, hipStream_t stream);

// conv.hip: halo-tiled direct 3x3 convolution (fp16); GemmParams as for A_CONV3.  Returns -1 if the layer does not fit.
int launch_conv3x3_halo(const GemmParams& p, int batch, hipStream_t stream);

}  // namespace cva
