// Bandwidth-bound helper kernels — see elementwise.h.
#include "elementwise.h"
#include "gemm.h"

#include <stdlib.h>

namespace cva {

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One wave per row, row held in registers (C <= 64*4*MAXV), two-pass mean / variance in fp32.
// (min 8 waves per SIMD: the row lives in 20 registers; without the bound the compiler took 128 VGPRs = half the
//  occupancy, and this HBM-bound kernel ran at 3.8 instead of ~5 TB/s)
template <typename T, int MAXV, int NT = 0>
__global__ __launch_bounds__(256, MAXV <= 5 ? 8 : 4) void layernorm_kernel(const float* __restrict__ in, long ld_in,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, void* out_, int out_f32,
                                                        int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* __restrict__ x = in + (long)row * ld_in;
    const int nv = C >> 2;   // float4 count
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * 64 + lane;
        if (idx < nv) {
            v[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + idx * 4)) : *reinterpret_cast<const f32x4*>(x + idx * 4);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        } else v[i] = (f32x4)(0.f);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * 64 + lane;
        if (idx < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * 64 + lane;
        if (idx < nv) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + idx * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + idx * 4);
            f32x4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
            if (out_f32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out_) + (long)row * C + idx * 4) = y;
            } else {
                T* o = reinterpret_cast<T*>(out_) + (long)row * C + idx * 4;
                if constexpr (sizeof(T) == 2) {      // one 8-byte store per lane (a wave writes 512 contiguous bytes)
                    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
                    const half4_t h = {(half_t)y[0], (half_t)y[1], (half_t)y[2], (half_t)y[3]};
                    if (NT == 1) __builtin_nontemporal_store(h, reinterpret_cast<half4_t*>(o)); else *reinterpret_cast<half4_t*>(o) = h;
                } else {
                    *reinterpret_cast<f32x4*>(o) = y;
                }
            }
        }
    }
}

// Residual add fused into the LayerNorm that consumes it (fp16 engine): x <- x + delta (delta = the fp16 output of the
// projection GEMM, bias included), the updated fp32 row goes back to the residual stream and its normalised fp16 image to
// `out`.  The GEMM then has a plain fp16 epilogue: its residual epilogue re-read and re-wrote the fp32 row in a burst at
// the end of every tile (all CUs together, +147 us on a 234 us proj launch), here the same bytes stream at the kernel's
// HBM rate.  `out` may alias `delta` (a lane reads its delta elements before it writes the same elements of out).
template <int MAXV>
__global__ __launch_bounds__(256, MAXV <= 5 ? 8 : 4) void layernorm_add_kernel(float* x_io, long ld, const half_t* delta,
                                                                                const float* __restrict__ gamma,
                                                                                const float* __restrict__ beta, half_t* out,
                                                                                int M, int C, float eps) {
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float* x = x_io + (long)row * ld;
    const half_t* dl = delta + (long)row * C;
    const int nv = C >> 2;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * 64 + lane;
        if (idx < nv) {
            const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + idx * 4));
            const half4_t d = __builtin_nontemporal_load(reinterpret_cast<const half4_t*>(dl + idx * 4));
            v[i] = a + f32x4{(float)d[0], (float)d[1], (float)d[2], (float)d[3]};
            __builtin_nontemporal_store(v[i], reinterpret_cast<f32x4*>(x + idx * 4));
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        } else v[i] = (f32x4)(0.f);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * 64 + lane;
        if (idx < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * 64 + lane;
        if (idx < nv) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + idx * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + idx * 4);
            half4_t h;
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = (half_t)((v[i][j] - mean) * rstd * g[j] + b[j]);
            __builtin_nontemporal_store(h, reinterpret_cast<half4_t*>(out + (long)row * C + idx * 4));
        }
    }
}

// LayerNorm with an OCP MX-fp8 result (fp8 engine): the normalised row leaves as e4m3 bytes + one E8M0 scale per 32
// columns — the A operand of the next linear layer (gemm8.hip, F8).  A 32-column block is the 4 values of 8 consecutive
// lanes.  HAS_DELTA: the residual add of the fp16 engine's layernorm_add_kernel (x += delta, written back) rides along.
// Scale bytes are stored in the consumer's fragment order (mx8_scale_index): A-side image always, W-side image as well
// when sc_w != null (the qkv projection runs its V tiles with the operands exchanged).
template <int MAXV, bool HAS_DELTA>
__global__ __launch_bounds__(256, MAXV <= 5 ? 8 : 4) void layernorm_mx8_kernel(float* x_io, long ld, const half_t* delta,
                                                                                const float* __restrict__ gamma,
                                                                                const float* __restrict__ beta,
                                                                                unsigned char* __restrict__ out8,
                                                                                unsigned char* __restrict__ sc_a,
                                                                                unsigned char* __restrict__ sc_w, int M, int C,
                                                                                float eps) {
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float* x = x_io + (long)row * ld;
    const int nv = C >> 2;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * 64 + lane;
        if (idx < nv) {
            v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + idx * 4));
            if (HAS_DELTA) {
                const half4_t d = __builtin_nontemporal_load(reinterpret_cast<const half4_t*>(delta + (long)row * C + idx * 4));
                v[i] = v[i] + f32x4{(float)d[0], (float)d[1], (float)d[2], (float)d[3]};
                __builtin_nontemporal_store(v[i], reinterpret_cast<f32x4*>(x + idx * 4));
            }
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        } else v[i] = (f32x4)(0.f);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * 64 + lane;
        if (idx < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = i * 64 + lane;          // nv is a multiple of 8 (C % 32 == 0): a block's 8 lanes are all in or all out
        float y[4] = {0.f, 0.f, 0.f, 0.f};
        if (idx < nv) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + idx * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + idx * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
        }
        float amax = fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3])));
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        amax = fmaxf(amax, __shfl_xor(amax, 2));
        amax = fmaxf(amax, __shfl_xor(amax, 4));
        if (idx < nv) {
            const int ex = (int)((__float_as_uint(amax) >> 23) & 0xff);
            int sbyte = ex - 8; sbyte = sbyte < 0 ? 0 : sbyte;                  // E8M0: 2^(sbyte - 127) = 2^(floor(log2 amax) - 8)
            const float inv = __uint_as_float((unsigned)(254 - sbyte) << 23);
            int pk = 0;
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(y[0] * inv, -448.f, 448.f),
                                                 __builtin_amdgcn_fmed3f(y[1] * inv, -448.f, 448.f), pk, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(y[2] * inv, -448.f, 448.f),
                                                 __builtin_amdgcn_fmed3f(y[3] * inv, -448.f, 448.f), pk, true);
            *reinterpret_cast<int*>(out8 + (long)row * C + idx * 4) = pk;
            if ((lane & 7) == 0) {
                const int k = idx * 4;
                sc_a[mx8_scale_index(row, k, C, false)] = (unsigned char)sbyte;
                if (sc_w) sc_w[mx8_scale_index(row, k, C, true)] = (unsigned char)sbyte;
            }
        }
    }
}

// Inference transform of the reference CLI (T.ToTensor + T.Normalize, cell_detection.py:214-227) evaluated on the fly
// when the input is the raw uint8 HWC tile: v = (u8 / 255 - mean[c]) / std[c], fp32, the same three roundings as torch
// (true division, -ffp-contract=off).
struct U8Norm { float mean[3]; float stdv[3]; };
__device__ __forceinline__ float u8_norm(uint8_t v, float mean, float stdv) { return ((float)v / 255.0f - mean) / stdv; }

template <typename T, bool U8>
__global__ void patchify_kernel(const float* __restrict__ x, const uint8_t* __restrict__ x8, U8Norm nm, T* __restrict__ out,
                                int B, int H, int W) {
    // one thread = 4 consecutive kx of one (token, c, ky): 16-B fp32 read (or 4 bytes at stride 3), 4 elements written
    const int gw = W >> 4, gh = H >> 4;
    const long total = (long)B * gh * gw * 192;   // 768 / 4
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % 192);
        const long tok = i / 192;
        const int k = k4 * 4;
        const int c = k >> 8, ky = (k >> 4) & 15, kx = k & 15;
        const int b = (int)(tok / (gh * gw));
        const int t = (int)(tok - (long)b * gh * gw);
        const int ty = t / gw, tx = t - ty * gw;
        f32x4 v;
        if constexpr (U8) {
            const uint8_t* s8 = x8 + (((long)b * H + ty * 16 + ky) * W + tx * 16 + kx) * 3 + c;
            const float mu = nm.mean[c], sd = nm.stdv[c];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = u8_norm(s8[j * 3], mu, sd);
        } else {
            v = *reinterpret_cast<const f32x4*>(x + (((long)b * 3 + c) * H + ty * 16 + ky) * W + tx * 16 + kx);
        }
        T* o = out + tok * 768 + k;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = Traits<T>::from_float(v[j]);
    }
}

// NCHW fp32 image (3 channels) -> NHWC of T with the channel count padded to CP (8: implicit-GEMM path, 32: the halo
// convolution kernel, which wants whole 32-channel chunks); one thread per (pixel, 8-channel group)
template <typename T, bool U8>
__global__ void nhwc8_kernel(const float* __restrict__ x, const uint8_t* __restrict__ x8, U8Norm nm, T* __restrict__ out, int B,
                             int H, int W, int CP) {
    const long hw = (long)H * W;
    const int groups = CP >> 3;
    const long total = (long)B * hw * groups;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / groups;
        const int gq = (int)(i - pix * groups);
        const long b = pix / hw, r = pix - b * hw;
        T v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = Traits<T>::from_float(0.f);
        if (gq == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                v[c] = Traits<T>::from_float(U8 ? u8_norm(x8[pix * 3 + c], nm.mean[c], nm.stdv[c]) : x[(b * 3 + c) * hw + r]);
        }
        T* o = out + pix * CP + gq * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = v[c];
    }
}

template <typename T>
__global__ void cast_tokens_kernel(const float* __restrict__ in, T* __restrict__ out, int B, int rpi_in, int skip,
                                   int C) {
    const int rpo = rpi_in - skip;
    const long total = (long)B * rpo * (C >> 2);
    const int c4n = C >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        const long row = i / c4n;
        const long b = row / rpo, t = row - b * rpo;
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + ((b * rpi_in + t + skip) * C + c4 * 4));
        T* o = out + row * C + c4 * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = Traits<T>::from_float(v[j]);
    }
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ in, T* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = Traits<T>::from_float(in[i]);
}

__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos0,
                                float* __restrict__ tokens, int B, int ntok, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    tokens[(long)b * ntok * C + c] = cls[c] + pos0[c];
}

__global__ void mean_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int R, int C) {
    // block = (b, 64-channel group); 256 threads = 4 row lanes x 64 channels
    __shared__ float red[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    // 8 independent partial sums per thread keep 8 loads in flight (the single dependent chain made this kernel
    // latency-bound: 0.14 TB/s on 64 workgroups); fixed summation order -> deterministic
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    if (c < C) {
        const float* src = in + (long)b * R * C + c;
        int r = rl;
        for (; r + 28 < R; r += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += src[(long)(r + 4 * u) * C];
        }
        for (; r < R; r += 4) acc[0] += src[(long)r * C];
    }
    const float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < C) out[(long)b * C + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                                   (red[2][threadIdx.x] + red[3][threadIdx.x])) / (float)R;
}

template <typename T, int NOUT>
__global__ __launch_bounds__(256) void head1x1_kernel(const T* __restrict__ feat, const float* __restrict__ Wt,
                                                      const float* __restrict__ bias, float* __restrict__ logits,
                                                      uint8_t* __restrict__ amax, int n_arg, long npix, long total) {
    __shared__ float ws[NOUT * 64 + NOUT];
    for (int i = threadIdx.x; i < NOUT * 64; i += 256) ws[i] = Wt[i];
    if (threadIdx.x < NOUT) ws[NOUT * 64 + threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    constexpr int PE = Traits<T>::PIECE;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float acc[NOUT];
#pragma unroll
        for (int n = 0; n < NOUT; ++n) acc[n] = ws[NOUT * 64 + n];
        const T* f = feat + i * 64;
#pragma unroll 1
        for (int pc = 0; pc < 64 / PE; ++pc) {
            const Piece pz = load_piece(f + pc * PE);
            const T* e = reinterpret_cast<const T*>(&pz);
#pragma unroll
            for (int j = 0; j < PE; ++j) {
                const float v = Traits<T>::to_float(e[j]);
#pragma unroll
                for (int n = 0; n < NOUT; ++n) acc[n] = fmaf(v, ws[n * 64 + pc * PE + j], acc[n]);
            }
        }
        const long b = i / npix, r = i - b * npix;
#pragma unroll
        for (int n = 0; n < NOUT; ++n) logits[(b * NOUT + n) * npix + r] = acc[n];
        if (amax) {
            int best = 0; float bv = acc[0];
#pragma unroll
            for (int n = 1; n < NOUT; ++n) if (n < n_arg && acc[n] > bv) { bv = acc[n]; best = n; }
            amax[i] = (uint8_t)best;
        }
    }
}

// argmax over the channel dim of an NCHW fp32 map (first maximum, as torch.argmax): the nuclei_binary_map /
// nuclei_type_map reduction of calculate_instance_map (cellvit.py:366-374) for maps that did not come with argmax planes
__global__ void argmax_nchw_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int C, long hw, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / hw, r = i - b * hw;
        const float* p = x + b * C * hw + r;
        float bv = p[0]; int best = 0;
        for (int c = 1; c < C; ++c) { const float v = p[c * hw]; if (v > bv) { bv = v; best = c; } }
        out[i] = (uint8_t)best;
    }
}

// u8 HWC -> normalised fp32 NCHW (what the reference hands to model.forward); parity helper for the fused input path
__global__ void normalize_u8_kernel(const uint8_t* __restrict__ x8, U8Norm nm, float* __restrict__ out, long hw, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / hw, r = i - b * hw;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(b * 3 + c) * hw + r] = u8_norm(x8[i * 3 + c], nm.mean[c], nm.stdv[c]);
    }
}

// Cell-token pooling (cell_detection.py:396-409): per instance record, mean of the encoder tokens under the bbox / patch
// window [floor(rmin/p), ceil(rmax/p)) x [floor(cmin/p), ceil(cmax/p)), indices cast to uint8 as the reference does.
// One workgroup per record; rows of the window are summed in raster order (the reference's (H W) order), then / count.
struct PoolRec { int32_t id, rmin, cmin, rmax, cmax; };
__global__ __launch_bounds__(256) void pool_tokens_kernel(const float* __restrict__ tok, const unsigned char* __restrict__ recs,
                                                          int rec_stride, int max_inst, const int32_t* __restrict__ n_recs,
                                                          const int64_t* __restrict__ rec_off, int gh, int gw, int D, int patch,
                                                          float* __restrict__ out) {
    const int b = blockIdx.y, slot = blockIdx.x;
    if (slot >= n_recs[b] || slot >= max_inst) return;
    const PoolRec r = *reinterpret_cast<const PoolRec*>(recs + ((long)b * max_inst + slot) * rec_stride);
    const int r0 = (r.rmin / patch) & 255, c0 = (r.cmin / patch) & 255;                       // floor (bbox >= 0), uint8 cast
    const int r1 = ((r.rmax + patch - 1) / patch) & 255, c1 = ((r.cmax + patch - 1) / patch) & 255;   // ceil
    const int rr1 = r1 < gh ? r1 : gh, cc1 = c1 < gw ? c1 : gw;                                 // python slicing clamps
    const int cnt = (rr1 > r0 ? rr1 - r0 : 0) * (cc1 > c0 ? cc1 - c0 : 0);
    float* o = out + (rec_off[b] + slot) * D;
    const float* base = tok + (long)b * gh * gw * D;
    for (int d = threadIdx.x; d < D; d += 256) {
        float s = 0.f;
        for (int y = r0; y < rr1; ++y)
            for (int x = c0; x < cc1; ++x) s += base[((long)y * gw + x) * D + d];
        o[d] = cnt > 0 ? s / (float)cnt : __builtin_nanf("");                              // torch.mean of an empty slice is nan
    }
}

inline int grid_for(long total, int per_block = 256, int cap = 16384) {
    long g = (total + per_block - 1) / per_block;
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

template <typename T>
int launch_layernorm(const float* in, long ld_in, const float* gamma, const float* beta, void* out, int out_f32,
                     int M, int C, float eps, hipStream_t stream) {
    if (C % 4 != 0 || C > 64 * 4 * 8) return (int)hipErrorInvalidValue;
    const dim3 grid((M + 3) / 4), block(256);
    if (C <= 64 * 4 * 2)
        hipLaunchKernelGGL((layernorm_kernel<T, 2>), grid, block, 0, stream, in, ld_in, gamma, beta, out, out_f32, M, C, eps);
    else if (C <= 64 * 4 * 5) {
        static const int nt = cva_env_int("CVA_LN", 1);   // 1 (default): nontemporal loads / stores
        if (nt == 2) hipLaunchKernelGGL((layernorm_kernel<T, 5, 2>), grid, block, 0, stream, in, ld_in, gamma, beta, out, out_f32, M, C, eps);
        else if (nt == 1) hipLaunchKernelGGL((layernorm_kernel<T, 5, 1>), grid, block, 0, stream, in, ld_in, gamma, beta, out, out_f32, M, C, eps);
        else hipLaunchKernelGGL((layernorm_kernel<T, 5>), grid, block, 0, stream, in, ld_in, gamma, beta, out, out_f32, M, C, eps);
    }
    else
        hipLaunchKernelGGL((layernorm_kernel<T, 8>), grid, block, 0, stream, in, ld_in, gamma, beta, out, out_f32, M, C, eps);
    return (int)hipGetLastError();
}

int launch_layernorm_add(float* x_io, long ld, const void* delta, const float* gamma, const float* beta, void* out,
                         int M, int C, float eps, hipStream_t stream) {
    if (C % 4 != 0 || C > 64 * 4 * 8) return (int)hipErrorInvalidValue;
    const dim3 grid((M + 3) / 4), block(256);
    const half_t* d = reinterpret_cast<const half_t*>(delta);
    half_t* o = reinterpret_cast<half_t*>(out);
    if (C <= 64 * 4 * 2) hipLaunchKernelGGL((layernorm_add_kernel<2>), grid, block, 0, stream, x_io, ld, d, gamma, beta, o, M, C, eps);
    else if (C <= 64 * 4 * 5) hipLaunchKernelGGL((layernorm_add_kernel<5>), grid, block, 0, stream, x_io, ld, d, gamma, beta, o, M, C, eps);
    else hipLaunchKernelGGL((layernorm_add_kernel<8>), grid, block, 0, stream, x_io, ld, d, gamma, beta, o, M, C, eps);
    return (int)hipGetLastError();
}

static U8Norm make_norm(const InputU8* u8) {
    U8Norm nm{};
    if (u8) for (int c = 0; c < 3; ++c) { nm.mean[c] = u8->mean[c]; nm.stdv[c] = u8->stdv[c]; }
    return nm;
}

int launch_layernorm_mx8(float* x_io, long ld, const void* delta_f16, const float* gamma, const float* beta, void* out8, void* sc_a,
                         void* sc_w, int M, int C, float eps, hipStream_t stream) {
    if (C % 128 != 0 || C > 64 * 4 * 8 || M % 256 != 0) return (int)hipErrorInvalidValue;
    const dim3 grid((M + 3) / 4), block(256);
    const half_t* d = reinterpret_cast<const half_t*>(delta_f16);
    unsigned char *o = reinterpret_cast<unsigned char*>(out8), *a = reinterpret_cast<unsigned char*>(sc_a), *w = reinterpret_cast<unsigned char*>(sc_w);
#define CVA_LN8(MV) do { if (d) hipLaunchKernelGGL((layernorm_mx8_kernel<MV, true>), grid, block, 0, stream, x_io, ld, d, gamma, beta, o, a, w, M, C, eps); \
                         else hipLaunchKernelGGL((layernorm_mx8_kernel<MV, false>), grid, block, 0, stream, x_io, ld, d, gamma, beta, o, a, w, M, C, eps); } while (0)
    if (C <= 64 * 4 * 2) CVA_LN8(2); else if (C <= 64 * 4 * 5) CVA_LN8(5); else CVA_LN8(8);
#undef CVA_LN8
    return (int)hipGetLastError();
}

template <typename T>
int launch_patchify(const float* x, const InputU8* u8, void* out, int B, int H, int W, hipStream_t stream) {
    const long total = (long)B * (H / 16) * (W / 16) * 192;
    const U8Norm nm = make_norm(u8);
    if (u8) hipLaunchKernelGGL((patchify_kernel<T, true>), dim3(grid_for(total)), dim3(256), 0, stream, nullptr, u8->x, nm,
                               reinterpret_cast<T*>(out), B, H, W);
    else hipLaunchKernelGGL((patchify_kernel<T, false>), dim3(grid_for(total)), dim3(256), 0, stream, x, nullptr, nm,
                            reinterpret_cast<T*>(out), B, H, W);
    return (int)hipGetLastError();
}

template <typename T>
int launch_nchw3_to_nhwc8(const float* x, const InputU8* u8, void* out, int B, int H, int W, int CP, hipStream_t stream) {
    if (CP < 8 || (CP & 7)) return (int)hipErrorInvalidValue;
    const U8Norm nm = make_norm(u8);
    const dim3 grid(grid_for((long)B * H * W * (CP >> 3)));
    if (u8) hipLaunchKernelGGL((nhwc8_kernel<T, true>), grid, dim3(256), 0, stream, nullptr, u8->x, nm, reinterpret_cast<T*>(out), B, H, W, CP);
    else hipLaunchKernelGGL((nhwc8_kernel<T, false>), grid, dim3(256), 0, stream, x, nullptr, nm, reinterpret_cast<T*>(out), B, H, W, CP);
    return (int)hipGetLastError();
}

int launch_argmax_nchw(const float* x, uint8_t* out, int B, int C, long hw, hipStream_t stream) {
    if (C < 1 || C > 255) return (int)hipErrorInvalidValue;
    const long total = (long)B * hw;
    hipLaunchKernelGGL(argmax_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, out, C, hw, total);
    return (int)hipGetLastError();
}

// Packed 3x3 filter [Cout][tap][Ctot] (fp16) -> [Cout][Ctot / 64][tap][64]: the K order of the implicit-GEMM convolution
// (gemm8.hip, conv_kmajor), 16 bytes per thread.
__global__ void conv_w_kmajor_kernel(const Piece* __restrict__ in, Piece* __restrict__ out, int Ctot, long total) {
    const int pc = Ctot / 8;                                  // 16-byte pieces per tap
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long co = i / (9L * pc);
        const int r = (int)(i - co * 9L * pc);                // position inside the output row
        const int chunk = r / 72, q = r - chunk * 72, tap = q >> 3, c8 = q & 7;
        out[i] = in[co * 9L * pc + (long)tap * pc + chunk * 8 + c8];
    }
}

int launch_conv_w_kmajor(const void* in, void* out, int Cout, int Ctot, hipStream_t stream) {
    if (Ctot % 64) return (int)hipErrorInvalidValue;
    const long total = (long)Cout * 9 * (Ctot / 8);
    hipLaunchKernelGGL(conv_w_kmajor_kernel, dim3(grid_for(total)), dim3(256), 0, stream, reinterpret_cast<const Piece*>(in),
                       reinterpret_cast<Piece*>(out), Ctot, total);
    return (int)hipGetLastError();
}

int launch_normalize_u8(const InputU8& u8, float* out, int B, long hw, hipStream_t stream) {
    const long total = (long)B * hw;
    hipLaunchKernelGGL(normalize_u8_kernel, dim3(grid_for(total)), dim3(256), 0, stream, u8.x, make_norm(&u8), out, hw, total);
    return (int)hipGetLastError();
}

int launch_pool_tokens(const float* tokens_nhwc, const void* recs, int rec_stride, int max_inst, const int32_t* n_recs,
                       const int64_t* rec_off, int B, int max_n, int gh, int gw, int D, int patch, float* out, hipStream_t stream) {
    if (max_n <= 0 || B <= 0) return 0;
    hipLaunchKernelGGL(pool_tokens_kernel, dim3(max_n, B), dim3(256), 0, stream, tokens_nhwc,
                       reinterpret_cast<const unsigned char*>(recs), rec_stride, max_inst, n_recs, rec_off, gh, gw, D, patch, out);
    return (int)hipGetLastError();
}

template <typename T>
int launch_cast_tokens(const float* in, void* out, int B, int rpi_in, int skip, int C, hipStream_t stream) {
    const long total = (long)B * (rpi_in - skip) * (C / 4);
    hipLaunchKernelGGL((cast_tokens_kernel<T>), dim3(grid_for(total)), dim3(256), 0, stream, in,
                       reinterpret_cast<T*>(out), B, rpi_in, skip, C);
    return (int)hipGetLastError();
}

template <typename T>
int launch_cast(const float* in, void* out, long n, hipStream_t stream) {
    hipLaunchKernelGGL((cast_kernel<T>), dim3(grid_for(n)), dim3(256), 0, stream, in, reinterpret_cast<T*>(out), n);
    return (int)hipGetLastError();
}

int launch_cls_rows(const float* cls, const float* pos0, float* tokens, int B, int ntok, int C, hipStream_t stream) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3((B * C + 255) / 256), dim3(256), 0, stream, cls, pos0, tokens, B, ntok, C);
    return (int)hipGetLastError();
}

int launch_mean_rows(const float* in, float* out, int B, int R, int C, hipStream_t stream) {
    hipLaunchKernelGGL(mean_rows_kernel, dim3((C + 63) / 64, B), dim3(256), 0, stream, in, out, B, R, C);
    return (int)hipGetLastError();
}

template <typename T>
int launch_head1x1(const void* feat, const float* Wt, const float* bias, float* logits, uint8_t* argmax_out,
                   int n_arg, long npix, int B, int n_out, hipStream_t stream) {
    const long total = npix * B;
    const dim3 grid(grid_for(total, 256, 8192)), block(256);
    const T* f = reinterpret_cast<const T*>(feat);
#define CVA_HEAD(N) case N: hipLaunchKernelGGL((head1x1_kernel<T, N>), grid, block, 0, stream, f, Wt, bias, logits, \
                                               argmax_out, n_arg, npix, total); break;
    switch (n_out) {
        CVA_HEAD(2) CVA_HEAD(3) CVA_HEAD(4) CVA_HEAD(5) CVA_HEAD(6) CVA_HEAD(7) CVA_HEAD(8) CVA_HEAD(9) CVA_HEAD(10)
        default: return (int)hipErrorInvalidValue;
    }
#undef CVA_HEAD
    return (int)hipGetLastError();
}

#define CVA_INST(T)                                                                                               \
    template int launch_layernorm<T>(const float*, long, const float*, const float*, void*, int, int, int, float, \
                                     hipStream_t);                                                                \
    template int launch_patchify<T>(const float*, const InputU8*, void*, int, int, int, hipStream_t);            \
    template int launch_nchw3_to_nhwc8<T>(const float*, const InputU8*, void*, int, int, int, int, hipStream_t); \
    template int launch_cast_tokens<T>(const float*, void*, int, int, int, int, hipStream_t);                     \
    template int launch_cast<T>(const float*, void*, long, hipStream_t);                                          \
    template int launch_head1x1<T>(const void*, const float*, const float*, float*, uint8_t*, int, long, int, int, \
                                   hipStream_t);
CVA_INST(half_t)
CVA_INST(float)
#undef CVA_INST

}  // namespace cva
