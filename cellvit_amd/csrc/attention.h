// Flash-style multi-head attention for both CellViT encoders, plus the decomposed relative
// position terms of the SAM encoder.
//   ViT-S  : softmax((q·kᵀ)·hd^-½)·v                         vits_histo.py:174-185
//   SAM    : softmax((q·hd^-½)·kᵀ + rel_h + rel_w)·v          SAM/image_encoder.py:235-257, 354-392
// Layouts (written by the QKV GEMM epilogue, gemm.h OUT_QKV):
//   Q, K : [S*heads, L, hd]   V^T : [S*heads, hd, Lp]          S = images (global) or windows
#pragma once
#include "common.h"

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
};

template <typename T> int launch_attention(const AttnParams& p, hipStream_t stream);
// v2 (attention2.hip): fused rel-pos; returns -1 when the geometry is not covered (caller uses v1 + launch_relpos)
template <typename T> int launch_attention2(const AttnParams& p, hipStream_t stream);
// attention3.hip: fp16 global attention (no windows, >= 256 keys, hd 64 / 80; rel-pos only in the key-tile-aligned form KW == 64):
// one 8-wave workgroup per CU, MFMA and softmax phases of the two wave groups in counter-phase, LDS-DMA staging.  -1 when not covered.
// Measured equal to attention2 (the two pipes of a SIMD do not overlap across waves): ablation builds only, production returns -1.
int launch_attention3(const AttnParams& p, hipStream_t stream);
// attention_win.hip: single-pass kernel for <= 208 keys (SAM windows), fp16 only; -1 when not covered
int launch_attention_win(const AttnParams& p, hipStream_t stream);
template <typename T> int launch_relpos(const RelPosParams& p, hipStream_t stream);
template <typename T> int launch_pad_kv(const PadKVParams& p, hipStream_t stream);

}  // namespace cva
