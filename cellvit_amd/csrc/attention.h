// Flash-style multi-head attention for both CellViT encoders, plus the decomposed relative
// position terms of the SAM encoder.
//   ViT-S  : softmax((q·kᵀ)·hd^-½)·v                         vits_histo.py:174-185
//   SAM    : softmax((q·hd^-½)·kᵀ + rel_h + rel_w)·v          SAM/image_encoder.py:235-257, 354-392
// Layouts (written by the QKV GEMM epilogue, gemm.h OUT_QKV):
//   Q, K : [S*heads, L, hd]   V^T : [S*heads, hd, Lp]          S = images (global) or windows
#pragma once
#include "common.h"
#include "gemm.h"

namespace cva {

struct AttnParams {
    const void* Q; const void* K; const void* Vt;   // T
    const float* relh;      // [S*heads, L, KH] fp32 or null
    const float* relw;      // [S*heads, L, KW]
    const float* tab_h;     // fused path (attention2): raw tables [2*KH-1, hd] / [2*KW-1, hd] fp32, or null
    const float* tab_w;
    void* out;              // T [tokens, D]
    int S, heads, L, Lp, hd, D;
    int nk;                 // number of keys (== L)
    int KH, KW;             // key grid (rel-pos only); key = kh*KW + kw
    float scale;
    // output row mapping (inverse of the QKV scatter)
    int ntok;               // tokens per image
    int win, gw, gh, nwx, nwy;
    void* win_prep;         // >= 32 KiB device scratch for attention_win.hip (rel-pos operands of the layer), or null
    int dbg;                // experiment switches (attention_win.hip, CVA_ATTNW_DBG)
    // fp8 engine, proj on the block-scaled MFMA (BASELINE.json configs[4]): the attention output leaves the kernel as MX-fp8 rows instead
    // of fp16 — out8 e4m3 [tokens, K8], K8 = 96 * heads: head h owns columns [96 h, 96 h + 96) = its hd = 80 values + 16 zeros (written
    // once with the buffer), so that no 32-element scale block straddles two heads, i.e. two workgroups; out8_scale = the E8M0 scale image
    // in the A-side fragment order of gemm8_kernel (mx8_scale_index(row, k, K8, false)).  `out` is then not written.  hd 80 only;
    // attn2_kernel and attnwp_kernel (attn_takes_out8).
    void* out8; void* out8_scale; int K8;
    int v_rm;               // `Vt` holds V ROW-major [S*heads, L, hd] (same layout as K): the kernels read the PV operand through the
                            // transposing LDS read ds_read_b64_tr_b16 (fp16, hd 80: attn2_kernel and attnwp_kernel; see attn_takes_vrm)
};

struct RelPosParams {
    const void* Q;          // T [S*heads, L, hd]
    const float* tab_h;     // [2*KH-1, hd] fp32 (already resized to the key grid)
    const float* tab_w;     // [2*KW-1, hd]
    float* relh; float* relw;
    int SH, L, hd, KH, KW;  // query grid == key grid (self attention): q = qy*KW + qx
};

struct PadKVParams {        // window mode: keys/values of zero-padded tokens are the qkv biases
    void* K; void* Vt;      // T
    const float* qkv_bias;  // [3*D]
    int B, heads, hd, D, L, Lp, win, gw, gh, nwx, nwy;
    int v_rm;               // Vt is V row-major [S*heads, L, hd]
};

// Row-major V (v_rm): true when the kernel the dispatcher of cellvit_abi.hip chooses for this layer runs (and pays) in the v_rm form —
// fp16, hd 80 (LDS row pitch 160 B = 40 dwords: the 8 keys a half-wave of ds_read_b64_tr_b16 touches fall on 8 distinct 8-bank groups),
// on the persistent single-pass kernel for short key sequences (the SAM windows: tables present, 192 < nk <= 208, >= 64 (window, head)
// items).  attn2_kernel has the form too but is slower in it: global layers keep V^T (production).  A per-LAYER property: the qkv
// projection of a layer writes the layout its attention kernel reads; the engine decides once per geometry for all window blocks
// (their pre-padded per-block V buffers are filled accordingly).
bool attn_takes_vrm(const AttnParams& p, size_t elem_size);
// MX-fp8 output (out8): true when the kernel the dispatcher chooses for this layer has the fp8 epilogue (fp16 storage, hd 80)
bool attn_takes_out8(const AttnParams& p);

#if defined(__HIPCC__)
// Epilogue of one query row of one head as MX-fp8 (see AttnParams::out8).  The lane holds O[query][d = n*16 + g*4 + r], n = 0 .. 4
// (transposed flash attention: the four lanes g = 0..3 with the same li own one query); scale block b = columns [32 b, 32 b + 32) of the
// head = fragments n = 2b, 2b+1 of the four lanes (block 2: fragment 4 + the 16 pad zeros).  OCP MX as everywhere in the fp8 engine: shared
// exponent floor(log2 amax) - 8, elements saturated to +-448, round to nearest even (v_cvt_pk_fp8_f32).  All four lanes of a query must call it.
__device__ __forceinline__ void attn_store_mx8(const AttnParams& p, const f32x4 (&o)[5], const float inv, const long row, const int h, const int g) {
    unsigned char* d8 = reinterpret_cast<unsigned char*>(p.out8) + row * (long)p.K8 + h * 96 + g * 4;
    unsigned char* sc = reinterpret_cast<unsigned char*>(p.out8_scale);
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        float amax = 0.f;
#pragma unroll
        for (int n = 2 * b; n < 2 * b + 2 && n < 5; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) amax = fmaxf(amax, fabsf(o[n][r] * inv));
        amax = fmaxf(amax, __shfl_xor(amax, 16));
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        const int ex = (int)((__float_as_uint(amax) >> 23) & 0xff);
        int sbyte = ex - 8; sbyte = sbyte < 0 ? 0 : sbyte;
        const float qs = __uint_as_float((unsigned)(254 - sbyte) << 23) * inv;          // 2^-(sbyte - 127) / l
#pragma unroll
        for (int n = 2 * b; n < 2 * b + 2 && n < 5; ++n) {
            const float x0 = __builtin_amdgcn_fmed3f(o[n][0] * qs, -448.f, 448.f), x1 = __builtin_amdgcn_fmed3f(o[n][1] * qs, -448.f, 448.f);
            const float x2 = __builtin_amdgcn_fmed3f(o[n][2] * qs, -448.f, 448.f), x3 = __builtin_amdgcn_fmed3f(o[n][3] * qs, -448.f, 448.f);
            int pk = 0;
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(x0, x1, pk, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(x2, x3, pk, true);
            *reinterpret_cast<int*>(d8 + n * 16) = pk;
        }
        if (g == 0) sc[mx8_scale_index(row, h * 96 + 32 * b, p.K8, false)] = (unsigned char)sbyte;
    }
}
#endif
template <typename T> int launch_attention(const AttnParams& p, hipStream_t stream);
// v2 (attention2.hip): fused rel-pos; returns -1 when the geometry is not covered (caller uses v1 + launch_relpos)
template <typename T> int launch_attention2(const AttnParams& p, hipStream_t stream);
// attention_win.hip: single-pass kernel for <= 208 keys (SAM windows), fp16 only; -1 when not covered
int launch_attention_win(const AttnParams& p, hipStream_t stream);
template <typename T> int launch_relpos(const RelPosParams& p, hipStream_t stream);
template <typename T> int launch_pad_kv(const PadKVParams& p, hipStream_t stream);

}  // namespace cva
