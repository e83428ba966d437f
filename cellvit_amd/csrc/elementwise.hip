// Bandwidth-bound helper kernels — see elementwise.h.
#include "elementwise.h"

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

template <typename T>
__global__ void patchify_kernel(const float* __restrict__ x, T* __restrict__ out, int B, int H, int W) {
    // one thread = 4 consecutive kx of one (token, c, ky): 16-B fp32 read, 4 elements written
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
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long)b * 3 + c) * H + ty * 16 + ky) * W + tx * 16 + kx);
        T* o = out + tok * 768 + k;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = Traits<T>::from_float(v[j]);
    }
}

// NCHW fp32 image (3 channels) -> NHWC of T with the channel count padded to CP (8: implicit-GEMM path, 32: the halo
// convolution kernel, which wants whole 32-channel chunks); one thread per (pixel, 8-channel group)
template <typename T>
__global__ void nhwc8_kernel(const float* __restrict__ x, T* __restrict__ out, int B, int H, int W, int CP) {
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
            for (int c = 0; c < 3; ++c) v[c] = Traits<T>::from_float(x[(b * 3 + c) * hw + r]);
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
        static const int nt = [] { const char* e = getenv("CVA_LN"); return e ? atoi(e) : 1; }();   // 1 (default): nontemporal loads / stores
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

template <typename T>
int launch_patchify(const float* x, void* out, int B, int H, int W, hipStream_t stream) {
    const long total = (long)B * (H / 16) * (W / 16) * 192;
    hipLaunchKernelGGL((patchify_kernel<T>), dim3(grid_for(total)), dim3(256), 0, stream, x,
                       reinterpret_cast<T*>(out), B, H, W);
    return (int)hipGetLastError();
}

template <typename T>
int launch_nchw3_to_nhwc8(const float* x, void* out, int B, int H, int W, int CP, hipStream_t stream) {
    if (CP < 8 || (CP & 7)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((nhwc8_kernel<T>), dim3(grid_for((long)B * H * W * (CP >> 3))), dim3(256), 0, stream, x,
                       reinterpret_cast<T*>(out), B, H, W, CP);
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
    template int launch_patchify<T>(const float*, void*, int, int, int, hipStream_t);                             \
    template int launch_nchw3_to_nhwc8<T>(const float*, void*, int, int, int, int, hipStream_t);                       \
    template int launch_cast_tokens<T>(const float*, void*, int, int, int, int, hipStream_t);                     \
    template int launch_cast<T>(const float*, void*, long, hipStream_t);                                          \
    template int launch_head1x1<T>(const void*, const float*, const float*, float*, uint8_t*, int, long, int, int, \
                                   hipStream_t);
CVA_INST(half_t)
CVA_INST(float)
#undef CVA_INST

}  // namespace cva
