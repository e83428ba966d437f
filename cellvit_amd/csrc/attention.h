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

// Row-major V (v_rm): true when the kernels the dispatcher of cellvit_abi.hip would choose for this layer geometry both exist in the
// v_rm form — fp16, hd 80 (LDS row pitch 160 B = 40 dwords: the 8 keys a half-wave of ds_read_b64_tr_b16 touches fall on 8 distinct
// 8-bank groups), windows on the persistent single-pass kernel (tables present, 192 < nk <= 208, >= 64 (window, head) items), global
// blocks on attn2_kernel.  The decision is taken once per geometry (cv_set_geometry) for ALL blocks of the encoder.
bool attn_takes_vrm(const AttnParams& p, size_t elem_size);
template <typename T> int launch_attention(const AttnParams& p, hipStream_t stream);
// v2 (attention2.hip): fused rel-pos; returns -1 when the geometry is not covered (caller uses v1 + launch_relpos)
template <typename T> int launch_attention2(const AttnParams& p, hipStream_t stream);
// attention_win.hip: single-pass kernel for <= 208 keys (SAM windows), fp16 only; -1 when not covered
int launch_attention_win(const AttnParams& p, hipStream_t stream);
template <typename T> int launch_relpos(const RelPosParams& p, hipStream_t stream);
template <typename T> int launch_pad_kv(const PadKVParams& p, hipStream_t stream);

}  // namespace cva
