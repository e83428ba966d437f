// Flash-style attention on MFMA for gfx950 — see attention.h.
//
// One workgroup = 4 waves = 64 query rows of one (sequence, head); each wave owns 16 rows.
// Keys/values stream through LDS in tiles of 64 keys; S = Q·Kᵀ and O += P·V are 16x16x32 MFMAs,
// the online-softmax statistics are fp32 and reduced with wave shuffles inside each 16-lane row
// group (the reference's autocast keeps softmax in fp32 too).  The N² score matrix is never
// materialised (the reference builds 16x4096x4096 fp32 per global block, image_encoder.py:244-251).
#include "attention.h"

namespace cva {

namespace {

constexpr int QT = 64, KT = 64, NT = 256;
constexpr float LOG2E = 1.4426950408889634f;

template <typename T, int HD>
__global__ __launch_bounds__(NT) void attn_kernel(const AttnParams p) {
    using TR = Traits<T>;
    constexpr int PE = TR::PIECE;
    constexpr int HDP = (HD + 31) / 32 * 32;       // contraction length padded to the MMA step
    constexpr int NKS = HDP / 32, ND = HD / 16;
    constexpr int PK = lds_pitch<T>(HDP), PV = lds_pitch<T>(KT), PP = lds_pitch<T>(KT);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Ks = reinterpret_cast<T*>(smem_raw);                 // [KT][PK]
    T* Vts = Ks + KT * PK;                                  // [HD][PV]
    T* Ps = Vts + HD * PV;                                  // [QT][PP]
    float* relh_s = reinterpret_cast<float*>(Ps + QT * PP); // [QT][KH]
    float* relw_s = relh_s + QT * p.KH;                     // [QT][KW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * QT;
    const int sh = blockIdx.y;
    const bool has_rel = p.relh != nullptr;

    const T* __restrict__ Qg = reinterpret_cast<const T*>(p.Q) + (long)sh * p.L * HD;
    const T* __restrict__ Kg = reinterpret_cast<const T*>(p.K) + (long)sh * p.L * HD;
    const T* __restrict__ Vg = reinterpret_cast<const T*>(p.Vt) + (long)sh * HD * p.Lp;

    if (HDP > HD) {   // zero the contraction padding of the K tile once
        for (int i = tid; i < KT * (HDP - HD); i += NT) {
            const int r = i / (HDP - HD), c = i - r * (HDP - HD);
            Ks[r * PK + HD + c] = TR::from_float(0.f);
        }
    }
    if (has_rel) {
        for (int i = tid; i < QT * p.KH; i += NT) {
            const int q = i / p.KH, k = i - q * p.KH;
            relh_s[i] = (q0 + q < p.L) ? p.relh[((long)sh * p.L + q0 + q) * p.KH + k] : 0.f;
        }
        for (int i = tid; i < QT * p.KW; i += NT) {
            const int q = i / p.KW, k = i - q * p.KW;
            relw_s[i] = (q0 + q < p.L) ? p.relw[((long)sh * p.L + q0 + q) * p.KW + k] : 0.f;
        }
    }

    // Q fragments stay in registers for the whole key loop.
    typename TR::Frag qf[NKS];
    {
        const int row = q0 + wave * 16 + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d0 = ks * 32 + (lane >> 4) * 8;
            qf[ks] = (row < p.L && d0 < HD) ? TR::load_frag(Qg + (long)row * HD + d0) : TR::zero_frag();
        }
    }

    f32x4 o[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n) o[n] = (f32x4)(0.f);
    float m_run[4], l_run[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }

    const int ntiles = (p.nk + KT - 1) / KT;
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();   // previous tile fully consumed (also orders the prologue LDS writes)
        {   // K tile: rows = keys
            constexpr int PPR = HD / PE;
            for (int i = tid; i < KT * PPR; i += NT) {
                const int r = i / PPR, c = i - r * PPR;
                const int key = kt * KT + r;
                const Piece v = key < p.nk ? load_piece(Kg + (long)key * HD + c * PE) : zero_piece();
                store_piece(Ks + r * PK + c * PE, v);
            }
        }
        {   // V^T tile: rows = head-dim, cols = keys
            constexpr int PPR = KT / PE;
            for (int i = tid; i < HD * PPR; i += NT) {
                const int d = i / PPR, c = i - d * PPR;
                const int key0 = kt * KT + c * PE;
                Piece v = load_piece(Vg + (long)d * p.Lp + key0);
                if (key0 + PE > p.nk) {   // never let stale bytes past the last key meet P = 0
                    T* e = reinterpret_cast<T*>(&v);
#pragma unroll
                    for (int j = 0; j < PE; ++j) if (key0 + j >= p.nk) e[j] = TR::from_float(0.f);
                }
                store_piece(Vts + d * PV + c * PE, v);
            }
        }
        __syncthreads();

        // ---- S = Q Kᵀ  (4 blocks of 16 keys) ----
        f32x4 s[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            s[nb] = (f32x4)(0.f);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const typename TR::Frag kf =
                    TR::load_frag(Ks + (nb * 16 + (lane & 15)) * PK + ks * 32 + (lane >> 4) * 8);
                TR::mma(qf[ks], kf, s[nb]);
            }
        }
        // ---- scale, relative-position terms, key mask ----
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int key = kt * KT + nb * 16 + (lane & 15);
            const bool kok = key < p.nk;
            int kh = 0, kw = 0;
            if (has_rel && kok) { kh = key / p.KW; kw = key - kh * p.KW; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = s[nb][r] * p.scale;
                if (has_rel) {
                    const int ql = wave * 16 + (lane >> 4) * 4 + r;
                    v += relh_s[ql * p.KH + kh] + relw_s[ql * p.KW + kw];
                }
                v = kok ? v : -INFINITY;
                s[nb][r] = v;
                mx[r] = fmaxf(mx[r], v);
            }
        }
        float alpha[4], rs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = mx[r];
            v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2));
            v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 8));
            const float mn = fmaxf(m_run[r], v);
            alpha[r] = exp2f((m_run[r] - mn) * LOG2E);
            m_run[r] = mn;
            rs[r] = 0.f;
        }
        // ---- P = exp(S - m): to LDS in the A-operand layout of the P·V MMA ----
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = exp2f((s[nb][r] - m_run[r]) * LOG2E);
                rs[r] += pv;
                Ps[(wave * 16 + (lane >> 4) * 4 + r) * PP + nb * 16 + (lane & 15)] = TR::from_float(pv);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = rs[r];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            l_run[r] = l_run[r] * alpha[r] + v;
        }
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[n][r] *= alpha[r];
        __syncthreads();   // P visible to the lanes that read it as MMA fragments

        // ---- O += P V ----
#pragma unroll
        for (int ks = 0; ks < KT / 32; ++ks) {
            const typename TR::Frag pf =
                TR::load_frag(Ps + (wave * 16 + (lane & 15)) * PP + ks * 32 + (lane >> 4) * 8);
#pragma unroll
            for (int n = 0; n < ND; ++n) {
                const typename TR::Frag vf =
                    TR::load_frag(Vts + (n * 16 + (lane & 15)) * PV + ks * 32 + (lane >> 4) * 8);
                TR::mma(pf, vf, o[n]);
            }
        }
    }

    // ---- normalise and scatter back to token order ----
    const int s_idx = sh / p.heads, h = sh - s_idx * p.heads;
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qg = q0 + wave * 16 + (lane >> 4) * 4 + r;
        if (qg >= p.L) continue;
        long row;
        if (p.win > 0) {
            const int nw = p.nwx * p.nwy;
            const int b = s_idx / nw, w = s_idx - b * nw;
            const int wy = w / p.nwx, wx = w - wy * p.nwx;
            const int py = qg / p.win, px = qg - py * p.win;
            const int gy = wy * p.win + py, gx = wx * p.win + px;
            if (gy >= p.gh || gx >= p.gw) continue;      // padded query: discarded (image_encoder.py:316-317)
            row = (long)b * p.ntok + gy * p.gw + gx;
        } else {
            row = (long)s_idx * p.ntok + qg;
        }
        const float inv = 1.0f / l_run[r];
#pragma unroll
        for (int n = 0; n < ND; ++n)
            out[row * p.D + h * HD + n * 16 + (lane & 15)] = TR::from_float(o[n][r] * inv);
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void relpos_kernel(const RelPosParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* Qs = reinterpret_cast<float*>(smem_raw);   // [64][hd]
    const int tid = threadIdx.x;
    const int q0 = blockIdx.x * 64, sh = blockIdx.y;
    const T* __restrict__ Qg = reinterpret_cast<const T*>(p.Q) + (long)sh * p.L * p.hd;
    for (int i = tid; i < 64 * p.hd; i += NT) {
        const int q = i / p.hd;
        Qs[i] = (q0 + q < p.L) ? Traits<T>::to_float(Qg[(long)(q0) * p.hd + i]) : 0.f;
    }
    __syncthreads();
    const int nk = p.KH + p.KW;
    for (int o = tid; o < 64 * nk; o += NT) {
        const int q = o / nk, k = o - q * nk;
        const int qg = q0 + q;
        if (qg >= p.L) continue;
        const int qy = qg / p.KW, qx = qg - qy * p.KW;
        const float* __restrict__ tab;
        if (k < p.KH) tab = p.tab_h + (long)(qy - k + p.KH - 1) * p.hd;          // image_encoder.py:347-351
        else tab = p.tab_w + (long)(qx - (k - p.KH) + p.KW - 1) * p.hd;
        float acc = 0.f;
        for (int d = 0; d < p.hd; ++d) acc = fmaf(Qs[q * p.hd + d], tab[d], acc);
        if (k < p.KH) p.relh[((long)sh * p.L + qg) * p.KH + k] = acc;
        else p.relw[((long)sh * p.L + qg) * p.KW + (k - p.KH)] = acc;
    }
}

// Only the zero-padded positions of the edge windows are touched: blockIdx.y enumerates (image, edge window),
// the block walks that window's padded positions x channels.
template <typename T>
__global__ void pad_kv_kernel(const PadKVParams p) {
    const int nw = p.nwy * p.nwx;
    const int b = blockIdx.y / nw, w = blockIdx.y - b * nw;
    const int wy = w / p.nwx, wx = w - wy * p.nwx;
    const int vy = min(p.win, p.gh - wy * p.win), vx = min(p.win, p.gw - wx * p.win);   // valid rows / cols of this window
    if (vy == p.win && vx == p.win) return;                                            // interior window: nothing padded
    const int s = blockIdx.y;
    const int npad = p.L - vy * vx;
    const long total = (long)npad * p.D;
    const int head_pad = vy * (p.win - vx);
    auto pad_pos = [&](int k) {                      // k-th padded position of the window, row-major
        // rows 0..vy-1 have (win - vx) padded columns each, rows vy.. are fully padded
        int py, px;
        if (k < head_pad) { py = k / (p.win - vx); px = vx + (k - py * (p.win - vx)); }
        else { k -= head_pad; py = vy + k / p.win; px = k - (k / p.win) * p.win; }
        return py * p.win + px;
    };
    // K [s, h, pos, d]: channel fastest (contiguous runs of hd elements per position)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % p.D);
        const int pos = pad_pos((int)(i / p.D));
        const int h = c / p.hd, d = c - h * p.hd;
        reinterpret_cast<T*>(p.K)[(((long)s * p.heads + h) * p.L + pos) * p.hd + d] =
            Traits<T>::from_float(p.qkv_bias[p.D + c]);
        if (p.v_rm)      // V row-major: the same element of the V image
            reinterpret_cast<T*>(p.Vt)[(((long)s * p.heads + h) * p.L + pos) * p.hd + d] = Traits<T>::from_float(p.qkv_bias[2 * p.D + c]);
    }
    // V^T [s, h, d, pos]: position fastest (the padded positions of one row are a few contiguous runs)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i / npad);
        const int pos = pad_pos((int)(i - (long)c * npad));
        const int h = c / p.hd, d = c - h * p.hd;
        if (p.v_rm) continue;
        reinterpret_cast<T*>(p.Vt)[(((long)s * p.heads + h) * p.hd + d) * p.Lp + pos] =
            Traits<T>::from_float(p.qkv_bias[2 * p.D + c]);
    }
}

template <typename T, int HD>
size_t attn_lds_bytes(const AttnParams& p) {
    constexpr int HDP = (HD + 31) / 32 * 32;
    size_t b = (size_t)(KT * lds_pitch<T>(HDP) + HD * lds_pitch<T>(KT) + QT * lds_pitch<T>(KT)) * sizeof(T);
    if (p.relh) b += (size_t)QT * (p.KH + p.KW) * sizeof(float);
    return b;
}

template <typename T, int HD>
int launch_attn_hd(const AttnParams& p, hipStream_t stream) {
    const size_t lds = attn_lds_bytes<T, HD>(p);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<T, HD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((p.L + QT - 1) / QT, p.S * p.heads);
    hipLaunchKernelGGL((attn_kernel<T, HD>), grid, dim3(NT), lds, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

template <typename T>
int launch_attention(const AttnParams& p, hipStream_t stream) {
    if (p.relh && (p.KH > 64 || p.KW > 64)) return (int)hipErrorInvalidValue;
    switch (p.hd) {
        case 64: return launch_attn_hd<T, 64>(p, stream);
        case 80: return launch_attn_hd<T, 80>(p, stream);
        default: return (int)hipErrorInvalidValue;
    }
}

template <typename T>
int launch_relpos(const RelPosParams& p, hipStream_t stream) {
    dim3 grid((p.L + 63) / 64, p.SH);
    hipLaunchKernelGGL((relpos_kernel<T>), grid, dim3(NT), (size_t)64 * p.hd * sizeof(float), stream, p);
    return (int)hipGetLastError();
}

template <typename T>
int launch_pad_kv(const PadKVParams& p, hipStream_t stream) {
    const int max_pad = p.L - 1;
    const int bx = (int)(((long)max_pad * p.D + 255) / 256 > 64 ? 64 : ((long)max_pad * p.D + 255) / 256);
    hipLaunchKernelGGL((pad_kv_kernel<T>), dim3(bx, p.B * p.nwy * p.nwx), dim3(256), 0, stream, p);
    return (int)hipGetLastError();
}

template int launch_attention<half_t>(const AttnParams&, hipStream_t);
template int launch_attention<float>(const AttnParams&, hipStream_t);
template int launch_relpos<half_t>(const RelPosParams&, hipStream_t);
template int launch_relpos<float>(const RelPosParams&, hipStream_t);
template int launch_pad_kv<half_t>(const PadKVParams&, hipStream_t);
template int launch_pad_kv<float>(const PadKVParams&, hipStream_t);

}  // namespace cva
