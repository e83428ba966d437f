// On-device HoVer-Net post-processing for gfx950: the reference's cv2 / scipy / skimage chain
// (post_proc_cellvit.py:155-249 + the per-instance loop :95-151) as HIP kernels, bit-exact against
// oracle/postproc_ref.c on identical input maps.
//
// All stages are HBM-bound byte/integer/fp64 passes over the tile (algorithmic traffic 14.7 MB per
// 1024x1024 tile) except the marker-controlled watershed, which is an ORDERED flood:
// it decomposes exactly per 4-connected component of the nucleus mask, so one wavefront runs the
// sequential (value, age, index) priority flood of one component with its frontier pool in LDS while
// thousands of components proceed concurrently (persistent waves + atomic dequeue).
//
// Floating point: this file must be compiled with -ffp-contract=off; the only fused operations are
// the explicit fmaf()/fma() that restate OpenCV's convertTo (see oracle/postproc_ref.c).
#include "postproc.h"
#include "common.h"

#include <float.h>

#include <algorithm>
#include <vector>

namespace cva {

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------------------------------------
// connected components: lock-free union-find, root = smallest (raster-first) pixel index
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int uf_find(int* L, int i) {
    volatile int* V = L;
    int p;
    while ((p = V[i]) != i) i = p;
    return i;
}

__device__ __forceinline__ void uf_union(int* L, int a, int b) {
    for (;;) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }   // hook the larger root under the smaller
        const int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}

// L[i] = i where src is foreground (src != 0) xor invert, else -1
__global__ void k_cc_init(const uint8_t* __restrict__ src, int invert, int* __restrict__ L, int N) {
    const long base = (long)blockIdx.y * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const bool fg = (src[base + i] != 0) != (invert != 0);
        L[base + i] = fg ? i : -1;
    }
}

__global__ void k_cc_merge(int* __restrict__ Lb, int H, int W) {
    const int N = H * W;
    int* L = Lb + (long)blockIdx.y * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        if (L[i] < 0) continue;
        const int y = i / W, x = i - y * W;
        if (x > 0 && L[i - 1] >= 0) uf_union(L, i, i - 1);
        if (y > 0 && L[i - W] >= 0) uf_union(L, i, i - W);
    }
}

// Fast path for W % 64 == 0 (a wave = 64 consecutive pixels of one row): horizontal runs are linked
// without atomics (every foreground pixel points at the first pixel of its run inside the 64-pixel chunk),
// and only the first pixel of every vertical overlap segment / chunk seam does a union.
__global__ void k_cc_init_runs(const uint8_t* __restrict__ src, int invert, int* __restrict__ L, int N) {
    const long base = (long)blockIdx.y * N;
    const int lane = threadIdx.x & 63;
    for (int i0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63; i0 < N; i0 += gridDim.x * blockDim.x) {
        const int i = i0 + lane;
        const bool fg = (src[base + i] != 0) != (invert != 0);
        const unsigned long long m = __ballot(fg);
        int v = -1;
        if (fg) {
            const unsigned long long zeros_below = ~m & ((1ull << lane) - 1ull);
            const int start = zeros_below ? 64 - __clzll(zeros_below) : 0;
            v = i0 + start;
        }
        L[base + i] = v;
    }
}

// (the foreground bits of the left / upper-left neighbours come from the wave's ballots of its own row chunk and of the chunk above — two loads
//  per pixel instead of four; only lane 0 looks across the chunk seam.  Signs never change while roots are being hooked, so the masks are stable.)
__global__ void k_cc_merge_runs(int* __restrict__ Lb, int H, int W) {
    const int N = H * W, lane = threadIdx.x & 63;
    int* L = Lb + (long)blockIdx.y * N;
    for (int i0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63; i0 < N; i0 += gridDim.x * blockDim.x) {
        const int i = i0 + lane;
        const int y = i0 / W, xc = i0 - y * W;                               // (wave-uniform: the chunk's row and first column)
        const bool fg = L[i] >= 0;
        const bool up = y > 0 && L[i - W] >= 0;
        const unsigned long long m = __ballot(fg), um = __ballot(up);
        if (m == 0ull) continue;
        const bool seam_left = xc > 0 && L[i0 - 1] >= 0, seam_upleft = xc > 0 && y > 0 && L[i0 - W - 1] >= 0;
        if (!fg) continue;
        const bool left = lane ? ((m >> (lane - 1)) & 1ull) != 0 : seam_left;
        const bool upleft = lane ? ((um >> (lane - 1)) & 1ull) != 0 : seam_upleft;
        if (left && lane == 0) uf_union(L, i, i - 1);                        // run continues across a chunk seam
        if (up && !(left && upleft)) uf_union(L, i, i - W);
    }
}

// Flatten the forest; roots (L[i] == i: stable during this pass) also initialise the per-component accumulators of the
// kernels that follow — only root slots of those planes are ever read, so no whole-plane hipMemset is needed:
//   INIT 1: csize[root] = 0, bb planes (y0, y1, x0, x1)[root] = (+big, -1, +big, -1)   (k_comp_stats)
//   INIT 2: flag[root] = 0                                                            (k_border_flag / k_fill)
template <int INIT>
__global__ void k_cc_flatten(int* __restrict__ Lb, int N, int* __restrict__ csize, int* __restrict__ bb, int* __restrict__ flag) {
    const long base = (long)blockIdx.y * N;
    int* L = Lb + base;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int v = L[i];
        if (v < 0) continue;
        if (v == i) {
            if (INIT == 1) {
                csize[base + i] = 0;
                int* y0 = bb + base * 4;
                y0[i] = 0x7f7f7f7f; y0[(long)N + i] = -1; y0[2l * N + i] = 0x7f7f7f7f; y0[3l * N + i] = -1;
            } else if (INIT == 2) {
                flag[base + i] = 0;
            }
        } else {
            L[i] = uf_find(L, i);
        }
    }
}

// zero the few per-tile scalars of a run (component counter, flood queue head, marker count, overflow cursor)
__global__ void k_init_small(int* __restrict__ counters, unsigned long long* __restrict__ ovf_cursor, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4 * B) counters[i] = 0;
    if (i < B) ovf_cursor[i] = 0ull;
}

// ---- P1: component sizes / bounding boxes of the raw binary mask ----
__global__ void k_comp_stats(const int* __restrict__ Lb, int* __restrict__ csize, int* __restrict__ bb, int H, int W) {
    const int N = H * W;
    const long base = (long)blockIdx.y * N;
    const int* L = Lb + base;
    int* cs = csize + base;
    int* y0 = bb + base * 4; int* y1 = y0 + N; int* x0 = y1 + N; int* x1 = x0 + N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int r = L[i];
        if (r < 0) continue;
        const int y = i / W, x = i - y * W;
        if (x > 0 && L[i - 1] >= 0) continue;          // one thread per horizontal run
        int len = 1;
        while (x + len < W && L[i + len] >= 0) ++len;
        atomicAdd(&cs[r], len);
        atomicMin(&y0[r], y); atomicMax(&y1[r], y); atomicMin(&x0[r], x); atomicMax(&x1[r], x + len - 1);
    }
}

// Run walks without loops, for W % 64 == 0 (a wave = 64 consecutive pixels of one row, as in k_cc_init_runs): a run's length inside the chunk
// comes out of the wave's ballot mask, so the lane at a run's start needs no dependent loads — the kernels above let that lane walk its run
// pixel by pixel (a load per step, the other 63 lanes idle).  A run that crosses a chunk seam is accounted as two pieces; every accumulator
// is an integer sum / min / max, so the result is the same.  (profiles/r04_r_postproc_alone_kernel_stats.csv: k_comp_stats 814 us,
// k_marker_ids 610 us, k_inst_stats 1596 us per 64 tiles before.)
__device__ __forceinline__ int run_len_from(unsigned long long m, int lane) {     // bit `lane` of m is set: consecutive set bits from there
    const unsigned long long z = ~(m >> lane);
    return z ? __ffsll((long long)z) - 1 : 64 - lane;
}

__global__ void k_comp_stats_runs(const int* __restrict__ Lb, int* __restrict__ csize, int* __restrict__ bb, int H, int W) {
    const int N = H * W, lane = threadIdx.x & 63;
    const long base = (long)blockIdx.y * N;
    const int* L = Lb + base;
    int* cs = csize + base;
    int* y0 = bb + base * 4; int* y1 = y0 + N; int* x0 = y1 + N; int* x1 = x0 + N;
    for (int i0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63; i0 < N; i0 += gridDim.x * blockDim.x) {
        const int i = i0 + lane;
        const int r = L[i];
        const unsigned long long m = __ballot(r >= 0);
        if (r >= 0 && (lane == 0 || !((m >> (lane - 1)) & 1ull))) {
            const int len = run_len_from(m, lane);
            const int y = i / W, x = i - y * W;
            atomicAdd(&cs[r], len);
            atomicMin(&y0[r], y); atomicMax(&y1[r], y); atomicMin(&x0[r], x); atomicMax(&x1[r], x + len - 1);
        }
    }
}

// blb = fg && size >= 10 (hard-wired, post_proc:182); surviving roots are queued for the flood.  The flood of a component is a
// sequential chain (one pop per pixel), so the batch ends when the LARGEST components end: components of >= FLOOD_BIG pixels go to the
// front of the tile's list (filled from slot 0 up), the rest to the back (filled from the last slot down), and the flood's waves are
// dispatched tile-interleaved — every tile's long chains start in the first moments (longest-processing-time-first, approximately).
// The order in which components are flooded has no influence on the result (they do not interact).
constexpr int FLOOD_BIG = 1024;
__global__ void k_blb_finalize(const int* __restrict__ Lb, const int* __restrict__ csize, uint8_t* __restrict__ blb,
                               int* __restrict__ comp_list, int* __restrict__ small_count, int* __restrict__ big_count, int N,
                               int list_cap) {
    const long base = (long)blockIdx.y * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int r = Lb[base + i];
        const int sz = r >= 0 ? csize[base + r] : 0;
        const bool keep = sz >= 10;
        blb[base + i] = keep ? 1 : 0;
        if (keep && r == i) {                         // (>= 10 pixels per component: at most N / 10 entries, list_cap = N / 10 + 16)
            if (sz >= FLOOD_BIG) comp_list[(long)blockIdx.y * list_cap + atomicAdd(&big_count[blockIdx.y], 1)] = i;
            else comp_list[(long)blockIdx.y * list_cap + (list_cap - 1 - atomicAdd(&small_count[blockIdx.y], 1))] = i;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// min / max reductions (exact, order independent) -> cv2.normalize parameters
// ------------------------------------------------------------------------------------------------
constexpr int RED_BLOCKS = 128;

template <typename T>
__device__ __forceinline__ void block_minmax(T& mn, T& mx) {
    __shared__ double smn[NT], smx[NT];
    smn[threadIdx.x] = (double)mn; smx[threadIdx.x] = (double)mx;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            if (smn[threadIdx.x + s] < smn[threadIdx.x]) smn[threadIdx.x] = smn[threadIdx.x + s];
            if (smx[threadIdx.x + s] > smx[threadIdx.x]) smx[threadIdx.x] = smx[threadIdx.x + s];
        }
        __syncthreads();
    }
    mn = (T)smn[0]; mx = (T)smx[0];
    __syncthreads();
}

// planes: src + (tile*planes + p) * N ; partial: [tile][plane][RED_BLOCKS][2]
template <typename T>
__global__ void k_minmax_partial(const T* __restrict__ src, int planes, int N, double* __restrict__ partial) {
    const int tile = blockIdx.z, p = blockIdx.y;
    const T* a = src + ((long)tile * planes + p) * N;
    T mn = a[0], mx = a[0];
    for (int i = blockIdx.x * NT + threadIdx.x; i < N; i += RED_BLOCKS * NT) {
        const T v = a[i];
        if (v < mn) mn = v;
        if (v > mx) mx = v;
    }
    block_minmax<T>(mn, mx);
    if (threadIdx.x == 0) {
        double* o = partial + (((long)tile * planes + p) * RED_BLOCKS + blockIdx.x) * 2;
        o[0] = (double)mn; o[1] = (double)mx;
    }
}

// cv2.normalize(NORM_MINMAX, 0..1, CV_32F): scale/shift rounded to float (see oracle minmax_params)
__global__ void k_minmax_final(const double* __restrict__ partial, int planes, double* __restrict__ params) {
    const int tile = blockIdx.y, p = blockIdx.x;
    const double* a = partial + ((long)tile * planes + p) * RED_BLOCKS * 2;
    double mn = a[0], mx = a[1];
    for (int i = threadIdx.x; i < RED_BLOCKS; i += NT) {
        if (a[2 * i] < mn) mn = a[2 * i];
        if (a[2 * i + 1] > mx) mx = a[2 * i + 1];
    }
    block_minmax<double>(mn, mx);
    if (threadIdx.x == 0) {
        double s = (mx - mn > DBL_EPSILON) ? 1.0 / (mx - mn) : 0.0;
        s = (double)(float)s;
        const double sh = (double)((float)0.0f - (float)(mn * s));
        double* o = params + ((long)tile * planes + p) * 2;
        o[0] = s; o[1] = sh;
    }
}

// ------------------------------------------------------------------------------------------------
// separable Sobel (cv2.Sobel CV_64F, BORDER_REFLECT_101), 3x3 blur, combine
// ------------------------------------------------------------------------------------------------
struct SobelTaps { double d[21]; double s[21]; int ksize; };

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * (n - 1) - p; }
    return p;
}

// row pass on the min-max-normalised HV maps (normalisation fused: fmaf(src, a, b))
__global__ void k_sobel_row(const float* __restrict__ hv, const double* __restrict__ params, const SobelTaps taps,
                            double* __restrict__ tmp_h, double* __restrict__ tmp_v, int H, int W) {
    const int N = H * W, tile = blockIdx.y;
    const float* h = hv + (long)tile * 2 * N;
    const float* v = h + N;
    const float ah = (float)params[(tile * 2 + 0) * 2], bh = (float)params[(tile * 2 + 0) * 2 + 1];
    const float av = (float)params[(tile * 2 + 1) * 2], bv = (float)params[(tile * 2 + 1) * 2 + 1];
    const int r = taps.ksize / 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const float* hr = h + (long)y * W;
        const float* vr = v + (long)y * W;
        int xx = reflect101(x - r, W);
        double sh = taps.d[0] * (double)fmaf(hr[xx], ah, bh);
        double sv = taps.s[0] * (double)fmaf(vr[xx], av, bv);
        for (int j = 1; j < taps.ksize; ++j) {
            xx = reflect101(x - r + j, W);
            sh += taps.d[j] * (double)fmaf(hr[xx], ah, bh);
            sv += taps.s[j] * (double)fmaf(vr[xx], av, bv);
        }
        tmp_h[(long)tile * N + i] = sh;
        tmp_v[(long)tile * N + i] = sv;
    }
}

// column pass: sobel_h uses the symmetric smoothing kernel, sobel_v the anti-symmetric derivative kernel
// sob: [tile][2][N]
__global__ void k_sobel_col(const double* __restrict__ tmp_h, const double* __restrict__ tmp_v, const SobelTaps taps,
                            double* __restrict__ sob, int H, int W) {
    const int N = H * W, tile = blockIdx.y;
    const double* th = tmp_h + (long)tile * N;
    const double* tv = tmp_v + (long)tile * N;
    const int r = taps.ksize / 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        double sh = taps.s[r] * th[i];
        double sv = 0.0;
        for (int j = 1; j <= r; ++j) {
            const long up = (long)reflect101(y + j, H) * W + x, dn = (long)reflect101(y - j, H) * W + x;
            sh += taps.s[r + j] * (th[up] + th[dn]);
            sv += taps.d[r + j] * (tv[up] - tv[dn]);
        }
        sob[((long)tile * 2 + 0) * N + i] = sh;
        sob[((long)tile * 2 + 1) * N + i] = sv;
    }
}

// Four outputs per thread off one register window (W, H multiples of 4).  The two kernels above spend their time on addresses — a
// reflect101 loop and 64-bit index arithmetic per tap, 42 / 40 loads per pixel — not on the 84 / 62 fp64 operations (round-4 profile of the chain
// alone, profiles/r04_r_postproc_alone_kernel_stats.csv: 1161 + 1567 us per 64 tiles for 2.7 GB of traffic).  Here a thread loads the
// KS + 3 source values its four outputs share ONCE, interior threads without any border arithmetic; every output is the same chain of
// roundings as above (same products, same order of additions), so the results are bit-identical.
template <int KS>
__global__ void k_sobel_row4(const float* __restrict__ hv, const double* __restrict__ params, const SobelTaps taps,
                             double* __restrict__ tmp_h, double* __restrict__ tmp_v, int H, int W) {
    constexpr int R = KS / 2, A = (R + 3) & ~3, N4 = (A + 4 + R + 3) / 4;      // window [x0 - R, x0 + 3 + R] inside N4 aligned float4s from x0 - A
    const int N = H * W, tile = blockIdx.y, Wq = W >> 2;
    const float* h = hv + (long)tile * 2 * N;
    const float* v = h + N;
    const float ah = (float)params[(tile * 2 + 0) * 2], bh = (float)params[(tile * 2 + 0) * 2 + 1];
    const float av = (float)params[(tile * 2 + 1) * 2], bv = (float)params[(tile * 2 + 1) * 2 + 1];
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < (N >> 2); q += gridDim.x * blockDim.x) {
        const int y = q / Wq, x0 = (q - y * Wq) << 2;
        const bool interior = x0 - R >= 0 && x0 + 3 + R < W;
        const bool wide = x0 - A >= 0 && x0 - A + 4 * N4 <= W;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const float* row = (pl ? v : h) + (long)y * W;
            const float a = pl ? av : ah, b = pl ? bv : bh;
            double win[KS + 3];
            if (wide) {                 // 16-byte loads from the aligned column below x0 - R: consecutive lanes, consecutive float4s
                float buf[4 * N4];
#pragma unroll
                for (int k = 0; k < N4; ++k) {
                    const float4 f = *reinterpret_cast<const float4*>(row + x0 - A + 4 * k);
                    buf[4 * k] = f.x; buf[4 * k + 1] = f.y; buf[4 * k + 2] = f.z; buf[4 * k + 3] = f.w;
                }
#pragma unroll
                for (int k = 0; k < KS + 3; ++k) win[k] = (double)fmaf(buf[k + A - R], a, b);
            } else {
#pragma unroll
                for (int k = 0; k < KS + 3; ++k) {
                    const int xx = interior ? x0 - R + k : reflect101(x0 - R + k, W);
                    win[k] = (double)fmaf(row[xx], a, b);
                }
            }
            double o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double acc = (pl ? taps.s[0] : taps.d[0]) * win[u];
#pragma unroll
                for (int j = 1; j < KS; ++j) acc += (pl ? taps.s[j] : taps.d[j]) * win[u + j];
                o[u] = acc;
            }
            double* dst = (pl ? tmp_v : tmp_h) + (long)tile * N + (long)y * W + x0;
            *reinterpret_cast<double2*>(dst) = make_double2(o[0], o[1]);
            *reinterpret_cast<double2*>(dst + 2) = make_double2(o[2], o[3]);
        }
    }
}

template <int KS>
__global__ void k_sobel_col4(const double* __restrict__ tmp_h, const double* __restrict__ tmp_v, const SobelTaps taps,
                             double* __restrict__ sob, int H, int W) {
    constexpr int R = KS / 2;
    const int N = H * W, tile = blockIdx.y;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < (N >> 2); q += gridDim.x * blockDim.x) {
        const int yb = q / W, x = q - yb * W, y0 = yb << 2;              // consecutive threads: consecutive columns of four rows
        const bool interior = y0 - R >= 0 && y0 + 3 + R < H;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const double* src = (pl ? tmp_v : tmp_h) + (long)tile * N + x;
            double win[KS + 3];
#pragma unroll
            for (int k = 0; k < KS + 3; ++k) {
                const int yy = interior ? y0 - R + k : reflect101(y0 - R + k, H);
                win[k] = src[(long)yy * W];
            }
            double* dst = sob + ((long)tile * 2 + pl) * N + (long)y0 * W + x;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double acc = pl ? 0.0 : taps.s[R] * win[u + R];
#pragma unroll
                for (int j = 1; j <= R; ++j) {
                    // (up = row y + j, dn = row y - j, as above)
                    if (pl) acc += taps.d[R + j] * (win[u + R + j] - win[u + R - j]);
                    else acc += taps.s[R + j] * (win[u + R + j] + win[u + R - j]);
                }
                dst[(long)u * W] = acc;
            }
        }
    }
}

// post_proc:208-240: renormalise, 1 - x (float32), max, -(1 - blb) (-> float64), clip, dist0, marker seed
__global__ void k_combine(const double* __restrict__ sob, const double* __restrict__ params,
                          const uint8_t* __restrict__ blb, double* __restrict__ d0, uint8_t* __restrict__ mk, int N) {
    const int tile = blockIdx.y;
    const double sch = params[(tile * 2 + 0) * 2], shh = params[(tile * 2 + 0) * 2 + 1];
    const double scv = params[(tile * 2 + 1) * 2], shv = params[(tile * 2 + 1) * 2 + 1];
    const long base = (long)tile * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const float hn = (float)fma(sob[((long)tile * 2 + 0) * N + i], sch, shh);
        const float vn = (float)fma(sob[((long)tile * 2 + 1) * N + i], scv, shv);
        const float a = 1.0f - hn, b = 1.0f - vn;
        const float m = a > b ? a : b;
        const int bl = blb[base + i];
        double o = (double)m - (double)(1 - bl);
        if (o < 0) o = 0;
        d0[base + i] = (1.0 - o) * (double)bl;
        const int ob = o >= 0.4 ? 1 : 0;
        int mm = bl - ob; if (mm < 0) mm = 0;
        mk[base + i] = (uint8_t)mm;
    }
}

// dist = -GaussianBlur(d0, (3,3), 0): rows then columns, b*0.5 + (a + c)*0.25 each (post_proc:235)
__global__ void k_blur_neg(const double* __restrict__ d0, double* __restrict__ dist, int H, int W) {
    const int N = H * W;
    const double* s = d0 + (long)blockIdx.y * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const int xl = reflect101(x - 1, W), xr = reflect101(x + 1, W);
        double t[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const long row = (long)reflect101(y - 1 + k, H) * W;
            t[k] = s[row + x] * 0.5 + (s[row + xl] + s[row + xr]) * 0.25;
        }
        dist[(long)blockIdx.y * N + i] = -(t[1] * 0.5 + (t[0] + t[2]) * 0.25);
    }
}

// ------------------------------------------------------------------------------------------------
// binary_fill_holes + 5x5 ellipse opening
// ------------------------------------------------------------------------------------------------
__global__ void k_border_flag(const int* __restrict__ Lb, int* __restrict__ flag, int H, int W) {
    const int N = H * W;
    const long base = (long)blockIdx.y * N;
    const int per = 2 * (H + W);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < per; k += gridDim.x * blockDim.x) {
        int i;
        if (k < W) i = k;
        else if (k < 2 * W) i = (H - 1) * W + (k - W);
        else if (k < 2 * W + H) i = (k - 2 * W) * W;
        else i = (k - 2 * W - H) * W + W - 1;
        const int r = Lb[base + i];
        if (r >= 0) flag[base + r] = 1;
    }
}

__global__ void k_fill(const uint8_t* __restrict__ mk, const int* __restrict__ Lb, const int* __restrict__ flag,
                       uint8_t* __restrict__ out, int N) {
    const long base = (long)blockIdx.y * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int r = Lb[base + i];   // >= 0 on background pixels of mk
        out[base + i] = (mk[base + i] || (r >= 0 && !flag[base + r])) ? 1 : 0;
    }
}

// rows 00100 / 11111 / 11111 / 11111 / 00100; pixels outside the image are ignored (cv2 default border)
template <bool ERODE>
__global__ void k_morph5(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W) {
    const int N = H * W;
    const uint8_t* s = in + (long)blockIdx.y * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        bool v = ERODE;
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            const int half = (dy == -2 || dy == 2) ? 0 : 2;
            for (int dx = -half; dx <= half; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                const bool p = s[yy * W + xx] != 0;
                if (ERODE) v = v && p; else v = v || p;
            }
        }
        out[(long)blockIdx.y * N + i] = v ? 1 : 0;
    }
}

// The same structuring element on FOUR pixels per thread: the planes hold 0 / 1 bytes, so erosion / dilation of packed bytes is a bitwise
// AND / OR of byte-shifted words (v_alignbyte_b32 over the left / centre / right words of a row); 11 word loads per 4 pixels instead of 68 byte
// loads behind 17 border tests each (620 us per 64 tiles for 134 MB, profiles/r04_r_postproc_alone_kernel_stats.csv).  W a multiple of 4.
template <bool ERODE>
__global__ void k_morph5_w4(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W) {
    const int N = H * W, Ww = W >> 2;
    const unsigned* s = reinterpret_cast<const unsigned*>(in + (long)blockIdx.y * N);
    unsigned* o = reinterpret_cast<unsigned*>(out + (long)blockIdx.y * N);
    constexpr unsigned neutral = ERODE ? 0x01010101u : 0u;         // pixels outside the image are ignored
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (N >> 2); i += gridDim.x * blockDim.x) {
        const int y = i / Ww, xw = i - y * Ww;
        unsigned acc = neutral;
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            const unsigned* r = s + (long)yy * Ww;
            const unsigned w1 = r[xw];
            unsigned v = w1;
            if (dy != -2 && dy != 2) {
                const unsigned w0 = xw > 0 ? r[xw - 1] : neutral, w2 = xw + 1 < Ww ? r[xw + 1] : neutral;
                const unsigned m2 = __builtin_amdgcn_alignbyte(w1, w0, 2), m1 = __builtin_amdgcn_alignbyte(w1, w0, 3);
                const unsigned p1 = __builtin_amdgcn_alignbyte(w2, w1, 1), p2 = __builtin_amdgcn_alignbyte(w2, w1, 2);
                v = ERODE ? (w1 & m2 & m1 & p1 & p2) : (w1 | m2 | m1 | p1 | p2);
            }
            acc = ERODE ? (acc & v) : (acc | v);
        }
        o[i] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// raster-order ids: exclusive scan of the root flags (scipy.ndimage.label numbering)
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_ELEMS = 2048;   // per block

__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
    __shared__ int s[NT];
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < NT; o <<= 1) {
        const int t = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    const int incl = s[threadIdx.x];
    if (total) *total = s[NT - 1];
    __syncthreads();
    return incl - v;
}

__global__ void k_scan_partial(const int* __restrict__ Lb, int N, int* __restrict__ bsum, int nblk) {
    const int* L = Lb + (long)blockIdx.y * N;
    const int b0 = blockIdx.x * SCAN_ELEMS + threadIdx.x * 8;
    int c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int i = b0 + j; if (i < N && L[i] == i) ++c; }
    int tot;
    block_exclusive_scan(c, &tot);
    if (threadIdx.x == 0) bsum[(long)blockIdx.y * nblk + blockIdx.x] = tot;
}

__global__ void k_scan_blocks(int* __restrict__ bsum, int nblk, int* __restrict__ total_out) {   // one block per tile, nblk <= NT*8
    int* b = bsum + (long)blockIdx.x * nblk;
    int v[8], c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int i = threadIdx.x * 8 + j; v[j] = i < nblk ? b[i] : 0; c += v[j]; }
    int tot;
    int run = block_exclusive_scan(c, &tot);
    if (threadIdx.x == 0) total_out[blockIdx.x] = tot;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int i = threadIdx.x * 8 + j; if (i < nblk) b[i] = run; run += v[j]; }
}

// rank[root] = 1-based raster order of the component
__global__ void k_scan_apply(const int* __restrict__ Lb, int N, const int* __restrict__ bsum, int nblk,
                             int* __restrict__ rank) {
    const int* L = Lb + (long)blockIdx.y * N;
    const int b0 = blockIdx.x * SCAN_ELEMS + threadIdx.x * 8;
    int c = 0;
    bool isr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int i = b0 + j; isr[j] = i < N && L[i] == i; c += isr[j]; }
    int run = block_exclusive_scan(c, nullptr) + bsum[(long)blockIdx.y * nblk + blockIdx.x];
#pragma unroll
    for (int j = 0; j < 8; ++j) if (isr[j]) rank[(long)blockIdx.y * N + b0 + j] = ++run;
}

__global__ void k_marker_ids(const int* __restrict__ Lb, const int* __restrict__ rank, int* __restrict__ marker,
                             int* __restrict__ msize, int N, int W, int max_ids) {
    const long base = (long)blockIdx.y * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int r = Lb[base + i];
        int id = 0;
        if (r >= 0) {
            id = rank[base + r];
            const int x = i % W;
            if (id <= max_ids && !(x > 0 && Lb[base + i - 1] >= 0)) {     // one size update per horizontal run
                int len = 1;
                while (x + len < W && Lb[base + i + len] >= 0) ++len;
                atomicAdd(&msize[(long)blockIdx.y * (max_ids + 1) + id], len);
            }
        }
        marker[base + i] = id;
    }
}

// remove_small_objects(marker, object_size) (no relabel) and inst = markers * mask (skimage _validate_inputs)
__global__ void k_marker_ids_runs(const int* __restrict__ Lb, const int* __restrict__ rank, int* __restrict__ marker,
                                  int* __restrict__ msize, int N, int W, int max_ids) {      // W % 64 == 0, see run_len_from
    const long base = (long)blockIdx.y * N;
    const int lane = threadIdx.x & 63;
    for (int i0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63; i0 < N; i0 += gridDim.x * blockDim.x) {
        const int i = i0 + lane;
        const int r = Lb[base + i];
        const unsigned long long m = __ballot(r >= 0);
        int id = 0;
        if (r >= 0) {
            id = rank[base + r];
            if (id <= max_ids && (lane == 0 || !((m >> (lane - 1)) & 1ull)))
                atomicAdd(&msize[(long)blockIdx.y * (max_ids + 1) + id], run_len_from(m, lane));
        }
        marker[base + i] = id;
    }
}

__global__ void k_marker_filter(int* __restrict__ marker, const int* __restrict__ msize, int object_size,
                                const uint8_t* __restrict__ blb, int* __restrict__ inst, int N, int max_ids) {
    const long base = (long)blockIdx.y * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        int id = marker[base + i];
        if (id > max_ids || (id && msize[(long)blockIdx.y * (max_ids + 1) + id] < object_size)) id = 0;
        marker[base + i] = id;
        inst[base + i] = blb[base + i] ? id : -1;       // -1 = not in the mask (reset to 0 by k_inst_stats)
    }
}

// ------------------------------------------------------------------------------------------------
// marker-controlled watershed: one wave = one mask component, exact (value, age, index) order
// ------------------------------------------------------------------------------------------------
#ifndef CVA_POOL_LDS
#define CVA_POOL_LDS 512
#endif
#ifndef CVA_BITMAP_WORDS
#define CVA_BITMAP_WORDS 1024
#endif
constexpr int POOL_LDS = CVA_POOL_LDS;             // frontier entries kept in LDS (20 B each); larger pools continue in the global arena
constexpr int BITMAP_WORDS = CVA_BITMAP_WORDS;     // 32 * words pixels of bounding box keep their flood state in LDS
typedef unsigned long long u64;

struct FloodParams {
    const double* dist; const uint8_t* blb; int* inst;       // [B][N]
    const int* root1; const int* bb; const int* csize;      // component labels, bboxes, pixel counts of the mask
    const int* comp_list; const int* comp_count; const int* big_count; int list_cap;   // comp_count = the small ones (from the back)
    int* queue_head;                                          // [B] dequeue cursors
    u64* ovf_hi; u64* ovf_lo; int* ovf_lab; u64* ovf_cursor;  // overflow arena [B][N]
    int H, W, B;
    int dbg;
};

// Pool entries are 128-bit keys: hi = order-preserving image of the f64 distance value, lo = age << 42 | pixel
// index << 20 (| pool slot, OR-ed in while scanning).  Lexicographic (hi, lo) order == (value, age, index) order of
// the oracle; (age, index) is unique per entry so the slot bits never decide.
__device__ __forceinline__ u64 sortable_f64(double v) {
    if (v == 0.0) v = -0.0;                                    // -0.0 == +0.0 for the reference's comparisons
    const u64 b = (u64)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
struct Key2 { u64 hi, lo; };
__device__ __forceinline__ bool key_less(const Key2& a, const Key2& b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }

template <int CTRL, int ROWMASK>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
    int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xf, false);
    return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ void min_step(Key2& k) {
    Key2 o; o.hi = dpp_u64<CTRL, ROWMASK>(k.hi); o.lo = dpp_u64<CTRL, ROWMASK>(k.lo);
    if (key_less(o, k)) k = o;
}
__device__ __forceinline__ u64 readlane63(u64 v) {
    return ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63) << 32) |
           (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
}
// 64-bit steps of the same reduction (round 5): the minimum over the VALUE halves alone costs half the instructions of the 128-bit one, and the
// value decides almost always — (age, index) only break exact ties of the f64 distance
template <int CTRL, int ROWMASK>
__device__ __forceinline__ void min_step64(u64& k) {
    const u64 o = dpp_u64<CTRL, ROWMASK>(k);
    if (o < k) k = o;
}
__device__ __forceinline__ u64 wave_min_u64(u64 k) {
    min_step64<0xb1, 0xf>(k);
    min_step64<0x4e, 0xf>(k);
    min_step64<0x114, 0xf>(k);
    min_step64<0x118, 0xf>(k);
    min_step64<0x142, 0xa>(k);
    min_step64<0x143, 0xc>(k);
    return readlane63(k);
}
__device__ __forceinline__ Key2 wave_min_key128(Key2 k);
// wave-wide minimum of (hi, lo): minimum of hi first; one lane holds it -> its lo; several (an exact tie of values) -> the 128-bit reduction
// over the tied lanes.  Same result as the 128-bit reduction over all lanes.
__device__ __forceinline__ Key2 wave_min_key(Key2 k) {
    const u64 mh = wave_min_u64(k.hi);
    const u64 tied = __ballot(k.hi == mh);
    Key2 r;
    if (__popcll(tied) == 1) {
        const int w = __ffsll((long long)tied) - 1;
        r.hi = mh;
        r.lo = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k.lo >> 32), w) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k.lo, w);
        return r;
    }
    if (k.hi != mh) { k.hi = ~0ull; k.lo = ~0ull; }
    return wave_min_key128(k);
}
// wave-wide minimum in VALU data-parallel primitives (6 DPP steps, no LDS round trips)
__device__ __forceinline__ Key2 wave_min_key128(Key2 k) {
    min_step<0xb1, 0xf>(k);     // quad_perm [1,0,3,2]
    min_step<0x4e, 0xf>(k);     // quad_perm [2,3,0,1]
    min_step<0x114, 0xf>(k);    // row_shr:4
    min_step<0x118, 0xf>(k);    // row_shr:8
    min_step<0x142, 0xa>(k);    // row_bcast:15
    min_step<0x143, 0xc>(k);    // row_bcast:31
    Key2 r; r.hi = readlane63(k.hi); r.lo = readlane63(k.lo);
    return r;
}

// One workgroup = ONE wavefront, so LDS operations execute in program order and the fences below only need wavefront
// scope (compiler ordering).  A workgroup-scope release would also drain the fire-and-forget global stores of the labels
// (s_waitcnt vmcnt(0)) on every pop — measured: that wait, not the pool scan, was the per-pop cost.
__global__ __launch_bounds__(64) void k_flood(const FloodParams p) {
    __shared__ u64 s_hi[POOL_LDS];
    __shared__ u64 s_lo[POOL_LDS];
    __shared__ int s_lab[POOL_LDS];
    __shared__ unsigned s_bits[BITMAP_WORDS];      // claimable (unlabeled mask) pixels of the component's bbox
    const int lane = threadIdx.x;
    const int N = p.H * p.W, W = p.W, H = p.H;
    const int wshift = (W & (W - 1)) == 0 ? __ffs(W) - 1 : -1;
    {
        const int tile = blockIdx.x;                // tile-interleaved dispatch: wave slot blockIdx.y of every tile before slot + 1 of any
        const long base = (long)tile * N;
        const double* dist = p.dist + base;
        int* inst = p.inst + base;
        const int* root1 = p.root1 + base;
        const int* y0a = p.bb + base * 4; const int* y1a = y0a + N; const int* x0a = y1a + N; const int* x1a = x0a + N;
        u64* ohi = p.ovf_hi + base; u64* olo = p.ovf_lo + base; int* olab = p.ovf_lab + base;
        const int nbig = p.big_count[tile];
        int ncomp = nbig + p.comp_count[tile];
        if (ncomp > p.list_cap) ncomp = p.list_cap;
        for (;;) {
            int ci = 0;
            if (lane == 0) ci = atomicAdd(&p.queue_head[tile], 1);
            ci = __shfl(ci, 0);
            if (ci >= ncomp) break;
            const int root = p.comp_list[(long)tile * p.list_cap + (ci < nbig ? ci : p.list_cap - 1 - (ci - nbig))];
            const int by0 = y0a[root], by1 = y1a[root], bx0 = x0a[root], bx1 = x1a[root];
            const int bw = bx1 - bx0 + 1, barea = bw * (by1 - by0 + 1);
            const int carea = p.csize[base + root];   // pool entries never exceed the component's pixel count
            int n = 0;                 // pool size (wave uniform)
            long ovf_base = -1;        // lazily reserved slice of the overflow arena
            unsigned age = 0;
            const bool use_bits = barea <= BITMAP_WORDS * 32;   // else: state lives in global memory (atomicCAS)

            auto put = [&](int slot, u64 hi, u64 lo, int lab) {
                if (slot < POOL_LDS) { s_hi[slot] = hi; s_lo[slot] = lo; s_lab[slot] = lab; }
                else { const long o = ovf_base + (slot - POOL_LDS); ohi[o] = hi; olo[o] = lo; olab[o] = lab; }
            };
            // append the entries of the active lanes in lane order (wave-uniform bookkeeping)
            auto append = [&](bool have, double v, unsigned a, int idx, int lab) {
                const u64 m = __ballot(have);
                const int cnt = __popcll(m);
                if (cnt == 0) return;
                if (n + cnt > POOL_LDS && ovf_base < 0) {
                    u64 o = 0;
                    if (lane == 0) o = atomicAdd(&p.ovf_cursor[tile], (u64)carea);   // sum over components <= N
                    ovf_base = (long)__shfl((long long)o, 0);
                }
                if (have) {
                    const int rank = __popcll(m & ((1ull << lane) - 1ull));
                    put(n + rank, sortable_f64(v), ((u64)a << 42) | ((u64)(unsigned)idx << 20), lab);
                }
                n += cnt;
            };

            // ---- seed: marker pixels of this component that touch an unlabeled mask pixel (age 0).
            // Interior marker pixels never push anything when popped, so dropping them is exact.
            for (int t0 = 0; t0 < barea; t0 += 64) {
                const int t = t0 + lane;
                bool have = false; double v = 0; int idx = 0, lab = 0;
                if (t < barea) {
                    const int yy = by0 + t / bw, xx = bx0 + (t - (t / bw) * bw);
                    idx = yy * W + xx;
                    if (root1[idx] == root && (lab = inst[idx]) > 0) {
                        const bool open = (yy > 0 && inst[idx - W] == 0) || (xx > 0 && inst[idx - 1] == 0) ||
                                          (xx + 1 < W && inst[idx + 1] == 0) || (yy + 1 < H && inst[idx + W] == 0);
                        if (open) { have = true; v = dist[idx]; }
                    }
                }
                if (use_bits) {   // 64 consecutive bbox pixels -> two bitmap words, no atomics
                    const bool claimable = t < barea && root1[idx] == root && lab == 0 && inst[idx] == 0;
                    const u64 cm = __ballot(claimable);
                    if (lane == 0) s_bits[t0 >> 5] = (unsigned)cm;
                    if (lane == 32) s_bits[(t0 >> 5) + 1] = (unsigned)(cm >> 32);
                }
                append(have, v, 0u, idx, lab);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

            // ---- ordered flood ----
            // The entries pushed by a pop are appended one iteration LATE: their distance values are requested with the
            // claim, and the next iteration first scans the existing pool (LDS latency + compares) before it touches
            // them — the global-memory round trip of dist[q] overlaps the scan instead of preceding it.  The deferred
            // entries take part in that iteration's minimum from registers, so the pop order is unchanged.
            bool pend = false; double pend_v = 0; int pend_q = 0, pend_lab = 0; unsigned pend_age = 0;
            int npend = 0;
            while (n > 0 || npend > 0) {
                Key2 best; best.hi = ~0ull; best.lo = ~0ull;
                for (int sl = lane; sl < n; sl += 64) {
                    Key2 k;
                    if (sl < POOL_LDS) { k.hi = s_hi[sl]; k.lo = s_lo[sl]; }
                    else { const long o = ovf_base + (sl - POOL_LDS); k.hi = ohi[o]; k.lo = olo[o]; }
                    k.lo |= (u64)sl;
                    if (key_less(k, best)) best = k;
                }
                if (npend > 0) {
                    const u64 pm = __ballot(pend);
                    const int slot = n + __popcll(pm & ((1ull << lane) - 1ull));
                    append(pend, pend_v, pend_age, pend_q, pend_lab);      // n += npend
                    if (pend) {
                        Key2 k; k.hi = sortable_f64(pend_v);
                        k.lo = ((u64)pend_age << 42) | ((u64)(unsigned)pend_q << 20) | (u64)slot;
                        if (key_less(k, best)) best = k;
                    }
                    npend = 0; pend = false;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                }
                best = wave_min_key(best);
                // winner: identical on all lanes
                const int ps = (int)(best.lo & 0xFFFFFull), pidx = (int)((best.lo >> 20) & 0x3FFFFFull);
                int plab;
                if (ps < POOL_LDS) plab = s_lab[ps]; else plab = olab[ovf_base + (ps - POOL_LDS)];
                // remove: move the last entry into the hole
                const int last = n - 1;
                if (ps != last && lane == 0) {
                    u64 lh, ll; int lb;
                    if (last < POOL_LDS) { lh = s_hi[last]; ll = s_lo[last]; lb = s_lab[last]; }
                    else { const long o = ovf_base + (last - POOL_LDS); lh = ohi[o]; ll = olo[o]; lb = olab[o]; }
                    put(ps, lh, ll, lb);
                }
                n = last;
                // neighbours in skimage order: -W, -1, +1, +W ; claim with a coherent CAS
                const int py = wshift >= 0 ? (pidx >> wshift) : pidx / W, px = pidx - py * W;      // (the division is on the pop's dependency chain)
                bool have = false; int q = 0; double qv = 0;
                if (lane < 4) {
                    const int dy = lane == 0 ? -1 : (lane == 3 ? 1 : 0);
                    const int dx = lane == 1 ? -1 : (lane == 2 ? 1 : 0);
                    const int yy = py + dy, xx = px + dx;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                        q = yy * W + xx;
                        qv = dist[q];                                     // consumed after the NEXT iteration's pool scan
#ifdef CVA_ABLATION      // timing experiment (wrong flood order): what the dependent global load costs per pop (CVA_PP_DBG & 1)
                        if (p.dbg & 1) qv = (double)(q & 1023);
#endif
                        if (use_bits) {
                            if (yy >= by0 && yy <= by1 && xx >= bx0 && xx <= bx1) {
                                const int loc = (yy - by0) * bw + (xx - bx0);
                                const unsigned bit = 1u << (loc & 31);
                                have = (atomicAnd(&s_bits[loc >> 5], ~bit) & bit) != 0;   // LDS test-and-clear
                                if (have) inst[q] = plab;                 // fire-and-forget: nobody reads it back
                            }
                        } else {
                            have = atomicCAS(&inst[q], 0, plab) == 0;     // 0 = unlabeled mask pixel (-1 = outside)
                        }
                    }
                }
                const u64 m = __ballot(have);
                pend = have; pend_v = qv; pend_q = q; pend_lab = plab;
                pend_age = age + 1u + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                npend = __popcll(m);
                age += (unsigned)npend;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the hole fill precedes the next scan
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// per-instance records (post_proc:95-151)
// ------------------------------------------------------------------------------------------------
struct StatArrays {       // all [B][max_ids + 1] unless noted
    int* cnt; unsigned long long* sx; unsigned long long* sy;
    int* rmin; int* rmax; int* cmin; int* cmax; int* first; unsigned* hist /* [..][8] */; int* has_zero /* [B] */;
};

// Reset the per-id accumulators of the ids a tile can actually use (1 .. nmark[tile], the number of marker components)
// instead of memsetting all max_ids + 1 slots of eleven arrays: ~10^3 ids per tile against 65537 slots.
__global__ void k_stats_init(StatArrays st, int* __restrict__ msize, const int* __restrict__ nmark, int max_ids) {
    const int tile = blockIdx.y;
    const long sb = (long)tile * (max_ids + 1);
    const int hi = min(max_ids, nmark[tile]);
    if (blockIdx.x == 0 && threadIdx.x == 0) st.has_zero[tile] = 0;
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id <= hi; id += gridDim.x * blockDim.x) {
        msize[sb + id] = 0;
        st.cnt[sb + id] = 0; st.sx[sb + id] = 0ull; st.sy[sb + id] = 0ull;
        st.rmin[sb + id] = 0x7f7f7f7f; st.cmin[sb + id] = 0x7f7f7f7f; st.first[sb + id] = 0x7f7f7f7f;
        st.rmax[sb + id] = -1; st.cmax[sb + id] = -1;
#pragma unroll
        for (int k = 0; k < 8; ++k) st.hist[(sb + id) * 8 + k] = 0u;
    }
}

__global__ void k_inst_stats(int* __restrict__ inst, const uint8_t* __restrict__ type, StatArrays st, int H, int W,
                             int max_ids, int nr_types) {
    const int N = H * W, tile = blockIdx.y;
    const long base = (long)tile * N, sb = (long)tile * (max_ids + 1);
    bool zero_seen = false;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int id = inst[base + i];
        if (id <= 0) { zero_seen = true; if (id < 0) inst[base + i] = 0; continue; }
        if (id > max_ids) continue;
        const int y = i / W, x = i - y * W;
        if (x > 0 && inst[base + i - 1] == id) continue;     // one thread per horizontal run of the instance
        int len = 0;
        unsigned h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        do {
            if (nr_types > 0) {
                int t = type[base + i + len]; if (t > 7) t = 7;
#pragma unroll
                for (int k = 0; k < 8; ++k) h[k] += (t == k);
            }
            ++len;
        } while (x + len < W && inst[base + i + len] == id);
        atomicAdd(&st.cnt[sb + id], len);
        atomicAdd(&st.sx[sb + id], (unsigned long long)len * x + (unsigned long long)len * (len - 1) / 2);
        atomicAdd(&st.sy[sb + id], (unsigned long long)len * y);
        atomicMin(&st.rmin[sb + id], y); atomicMax(&st.rmax[sb + id], y);
        atomicMin(&st.cmin[sb + id], x); atomicMax(&st.cmax[sb + id], x + len - 1);
        atomicMin(&st.first[sb + id], i);
#pragma unroll
        for (int k = 0; k < 8; ++k) if (h[k]) atomicAdd(&st.hist[(sb + id) * 8 + k], h[k]);
    }
    if (__any(zero_seen) && (threadIdx.x & 63) == 0) st.has_zero[tile] = 1;
}

__global__ void k_inst_stats_runs(int* __restrict__ inst, const uint8_t* __restrict__ type, StatArrays st, int H, int W,
                                  int max_ids, int nr_types) {                                 // W % 64 == 0, see run_len_from
    const int N = H * W, tile = blockIdx.y, lane = threadIdx.x & 63;
    const long base = (long)tile * N, sb = (long)tile * (max_ids + 1);
    bool zero_seen = false;
    for (int i0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63; i0 < N; i0 += gridDim.x * blockDim.x) {
        const int i = i0 + lane;
        const int id = inst[base + i];
        if (id <= 0) { zero_seen = true; if (id < 0) inst[base + i] = 0; }
        const int id_left = __shfl_up(id, 1);
        const bool valid = id > 0 && id <= max_ids;
        const unsigned long long bm = __ballot(lane == 0 || id != id_left);        // first pixels of the chunk's runs (any id)
        unsigned long long tm[8];
        if (nr_types > 0) {
            int t = type[base + i]; if (t > 7) t = 7;
#pragma unroll
            for (int k = 0; k < 8; ++k) tm[k] = __ballot(t == k);
        }
        if (valid && ((bm >> lane) & 1ull)) {
            const unsigned long long above = bm & ~((2ull << lane) - 1ull);
            const int len = (above ? __ffsll((long long)above) - 1 : 64) - lane;
            const unsigned long long range = (len == 64 ? ~0ull : ((1ull << len) - 1ull)) << lane;
            const int y = i / W, x = i - y * W;
            atomicAdd(&st.cnt[sb + id], len);
            atomicAdd(&st.sx[sb + id], (unsigned long long)len * x + (unsigned long long)len * (len - 1) / 2);
            atomicAdd(&st.sy[sb + id], (unsigned long long)len * y);
            atomicMin(&st.rmin[sb + id], y); atomicMax(&st.rmax[sb + id], y);
            atomicMin(&st.cmin[sb + id], x); atomicMax(&st.cmax[sb + id], x + len - 1);
            atomicMin(&st.first[sb + id], i);
            if (nr_types > 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned c = (unsigned)__popcll(tm[k] & range);
                    if (c) atomicAdd(&st.hist[(sb + id) * 8 + k], c);
                }
            }
        }
    }
    if (__any(zero_seen) && (threadIdx.x & 63) == 0) st.has_zero[tile] = 1;
}

// one block per tile: ascending-id compaction (== np.unique order) + record arithmetic
__global__ void k_inst_records(StatArrays st, InstanceRec* __restrict__ recs, int* __restrict__ n_recs, int max_ids,
                               int max_inst, int nr_types, const int* __restrict__ nmark) {
    const int tile = blockIdx.x;
    const long sb = (long)tile * (max_ids + 1);
    __shared__ int s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const bool drop_first = st.has_zero[tile] == 0;   // np.unique(pred_inst)[1:] drops the smallest VALUE (quirk 1)
    const int id_hi = min(max_ids, nmark[tile]);      // ids are ranks of marker components
    for (int c0 = 1; c0 <= id_hi; c0 += NT) {
        const int id = c0 + threadIdx.x;
        const bool live = id <= id_hi && st.cnt[sb + id] > 0;
        int tot;
        const int pos = block_exclusive_scan(live ? 1 : 0, &tot) + s_base;
        __syncthreads();
        if (live) {
            const int slot = pos - (drop_first ? 1 : 0);
            if (slot >= 0 && slot < max_inst) {
                InstanceRec r;
                const int n = st.cnt[sb + id];
                r.id = id; r.rmin = st.rmin[sb + id]; r.cmin = st.cmin[sb + id];
                r.rmax = st.rmax[sb + id] + 1; r.cmax = st.cmax[sb + id] + 1; r.npix = n; r._pad = 0;
                const double m00 = (double)n;
                const double m10 = (double)((long long)st.sx[sb + id] - (long long)n * r.cmin);
                const double m01 = (double)((long long)st.sy[sb + id] - (long long)n * r.rmin);
                r.cx = m10 / m00 + (double)r.cmin;
                r.cy = m01 / m00 + (double)r.rmin;
                r.contour_off = st.first[sb + id];   // temporarily: raster-first pixel (contour start)
                r.contour_len = 0;
                r.type = 0; r.type_prob = 0.0;
                if (nr_types > 0) {
                    const unsigned* h = st.hist + (sb + id) * 8;
                    int best = -1, second = -1, present = 0;
                    for (int t = 0; t < nr_types && t < 8; ++t) { if (!h[t]) continue; ++present; if (best < 0 || h[t] > h[best]) best = t; }
                    for (int t = 0; t < nr_types && t < 8; ++t) { if (!h[t] || t == best) continue; if (second < 0 || h[t] > h[second]) second = t; }
                    int ty = best;
                    if (ty == 0 && present > 1) ty = second;
                    r.type = ty;
                    r.type_prob = (double)h[ty] / ((double)n + 1.0e-6);
                }
                recs[(long)tile * max_inst + slot] = r;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int n = s_base - (drop_first && s_base > 0 ? 1 : 0);
        n_recs[tile] = n;      // the TRUE count: n > max_inst tells the caller that records were dropped (capacity overflow)
    }
}

// ---- type vote for MORE than 8 classes (post_proc_cellvit.py:132-148 / :300-318: np.unique counts, sorted by count descending —
// a stable sort, so equal counts keep ascending type order — background replaced by the runner-up).  The 8-bin histogram of
// k_inst_stats is filled once per window of 8 types [t_lo, t_lo + 8), and a running (best, second, number of present types) per
// id is merged window by window in ascending type order: `>` keeps the LOWER type on equal counts, as the stable sort does.
struct VoteArrays { int* best_t; unsigned* best_c; int* second_t; unsigned* second_c; int* present; };

__global__ void k_vote_reset(VoteArrays v, unsigned* __restrict__ hist, const int* __restrict__ nmark, int max_ids, int first_window) {
    const int tile = blockIdx.y;
    const long sb = (long)tile * (max_ids + 1);
    const int hi = min(max_ids, nmark[tile]);
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id <= hi; id += gridDim.x * blockDim.x) {
#pragma unroll
        for (int k = 0; k < 8; ++k) hist[(sb + id) * 8 + k] = 0u;
        if (first_window) { v.best_t[sb + id] = -1; v.best_c[sb + id] = 0u; v.second_t[sb + id] = -1; v.second_c[sb + id] = 0u; v.present[sb + id] = 0; }
    }
}

// per horizontal run of an instance (as k_inst_stats): counts of the types inside the window
__global__ void k_type_hist_window(const int* __restrict__ inst, const uint8_t* __restrict__ type, unsigned* __restrict__ hist,
                                   int H, int W, int max_ids, int t_lo) {
    const int N = H * W, tile = blockIdx.y;
    const long base = (long)tile * N, sb = (long)tile * (max_ids + 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int id = inst[base + i];
        if (id <= 0 || id > max_ids) continue;
        const int x = i % W;
        if (x > 0 && inst[base + i - 1] == id) continue;
        int len = 0;
        unsigned h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        do {
            const int t = (int)type[base + i + len] - t_lo;
#pragma unroll
            for (int k = 0; k < 8; ++k) h[k] += (t == k);
            ++len;
        } while (x + len < W && inst[base + i + len] == id);
#pragma unroll
        for (int k = 0; k < 8; ++k) if (h[k]) atomicAdd(&hist[(sb + id) * 8 + k], h[k]);
    }
}

__global__ void k_vote_merge(VoteArrays v, const unsigned* __restrict__ hist, const int* __restrict__ nmark, int max_ids, int t_lo, int nr_types) {
    const int tile = blockIdx.y;
    const long sb = (long)tile * (max_ids + 1);
    const int hi = min(max_ids, nmark[tile]);
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id <= hi; id += gridDim.x * blockDim.x) {
        int bt = v.best_t[sb + id], st_ = v.second_t[sb + id], pr = v.present[sb + id];
        unsigned bc = v.best_c[sb + id], sc = v.second_c[sb + id];
        for (int k = 0; k < 8 && t_lo + k < nr_types; ++k) {
            const unsigned c = hist[(sb + id) * 8 + k];
            if (!c) continue;
            ++pr;
            if (c > bc) { st_ = bt; sc = bc; bt = t_lo + k; bc = c; }
            else if (c > sc) { st_ = t_lo + k; sc = c; }
        }
        v.best_t[sb + id] = bt; v.best_c[sb + id] = bc; v.second_t[sb + id] = st_; v.second_c[sb + id] = sc; v.present[sb + id] = pr;
    }
}

__global__ void k_vote_apply(VoteArrays v, InstanceRec* __restrict__ recs, const int* __restrict__ n_recs, int max_ids, int max_inst) {
    const int tile = blockIdx.y;
    const long sb = (long)tile * (max_ids + 1);
    const int nr = min(n_recs[tile], max_inst);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nr; k += gridDim.x * blockDim.x) {
        InstanceRec* r = &recs[(long)tile * max_inst + k];
        const int id = r->id;
        int ty = v.best_t[sb + id];
        unsigned c = v.best_c[sb + id];
        if (ty == 0 && v.present[sb + id] > 1) { ty = v.second_t[sb + id]; c = v.second_c[sb + id]; }
        r->type = ty < 0 ? 0 : ty;
        r->type_prob = (double)c / ((double)r->npix + 1.0e-6);
    }
}

// Suzuki-Abe outer border + CHAIN_APPROX_SIMPLE (OpenCV icvFetchContour), see oracle trace_contour
__device__ int trace_contour(const int* __restrict__ inst, int H, int W, int id, int x0, int y0, int* __restrict__ pts) {
    const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};
#define CVA_ON(xx, yy) ((xx) >= 0 && (xx) < W && (yy) >= 0 && (yy) < H && inst[(yy) * W + (xx)] == id)
    int n = 0, s = 4, x1, y1;
    do { s = (s - 1) & 7; x1 = x0 + DX[s]; y1 = y0 + DY[s]; } while (!CVA_ON(x1, y1) && s != 4);
    if (s == 4) { if (pts) { pts[0] = x0; pts[1] = y0; } return 1; }
    int x3 = x0, y3 = y0, prev_s = s ^ 4;
    for (;;) {
        int x4, y4;
        for (;;) { ++s; x4 = x3 + DX[s & 7]; y4 = y3 + DY[s & 7]; if (CVA_ON(x4, y4)) break; }
        s &= 7;
        if (s != prev_s) { if (pts) { pts[2 * n] = x3; pts[2 * n + 1] = y3; } ++n; prev_s = s; }
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) break;
        x3 = x4; y3 = y4; s = (s + 4) & 7;
    }
#undef CVA_ON
    return n;
}

__global__ void k_contour_count(const int* __restrict__ inst, InstanceRec* __restrict__ recs, const int* __restrict__ n_recs,
                                int H, int W, int max_inst) {
    const int tile = blockIdx.y, N = H * W;
    const int nr = min(n_recs[tile], max_inst);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nr; k += gridDim.x * blockDim.x) {
        InstanceRec* r = &recs[(long)tile * max_inst + k];
        const int f = r->contour_off;
        r->contour_len = trace_contour(inst + (long)tile * N, H, W, r->id, f % W, f / W, nullptr);
        r->_pad = f;
    }
}

__global__ void k_contour_offsets(InstanceRec* __restrict__ recs, const int* __restrict__ n_recs, int* __restrict__ n_pts,
                                  int max_inst) {
    const int tile = blockIdx.x;
    const int n = min(n_recs[tile], max_inst);
    __shared__ int s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += NT) {
        const int k = c0 + threadIdx.x;
        const int len = k < n ? recs[(long)tile * max_inst + k].contour_len : 0;
        int tot;
        const int off = block_exclusive_scan(len, &tot) + s_base;
        __syncthreads();
        if (k < n) recs[(long)tile * max_inst + k].contour_off = off;
        __syncthreads();
        if (threadIdx.x == 0) s_base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) n_pts[tile] = s_base;
}

__global__ void k_contour_write(const int* __restrict__ inst, InstanceRec* __restrict__ recs, const int* __restrict__ n_recs,
                                int* __restrict__ contours, int H, int W, int max_inst, int max_pts) {
    const int tile = blockIdx.y, N = H * W;
    const int nr = min(n_recs[tile], max_inst);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nr; k += gridDim.x * blockDim.x) {
        InstanceRec* r = &recs[(long)tile * max_inst + k];
        const int f = r->_pad;
        r->_pad = 0;
        if (r->contour_off + r->contour_len <= max_pts)
            trace_contour(inst + (long)tile * N, H, W, r->id, f % W, f / W, contours + ((long)tile * max_pts + r->contour_off) * 2);
    }
}

void sobel_taps(int ksize, SobelTaps* t) {   // cv2.getDerivKernels / getSobelKernels (integer recurrences)
    auto gen = [&](int order, double* out) {
        std::vector<long long> k(ksize + 1, 0);
        k[0] = 1;
        for (int i = 0; i < ksize - order - 1; ++i) {
            long long oldv = k[0];
            for (int j = 1; j <= ksize; ++j) { const long long nv = k[j] + k[j - 1]; k[j - 1] = oldv; oldv = nv; }
        }
        for (int i = 0; i < order; ++i) {
            long long oldv = -k[0];
            for (int j = 1; j <= ksize; ++j) { const long long nv = k[j - 1] - k[j]; k[j - 1] = oldv; oldv = nv; }
        }
        for (int j = 0; j < 21; ++j) out[j] = j < ksize ? (double)k[j] : 0.0;
    };
    t->ksize = ksize;
    gen(1, t->d);
    gen(0, t->s);
}

}  // namespace

// ================================================================================================
struct PostprocWorkspace {
    PostprocDims d;
    std::vector<void*> pool;
    size_t bytes = 0;
    int *L1 = nullptr, *L2 = nullptr, *csize = nullptr, *bb = nullptr, *flag = nullptr, *rank = nullptr, *marker = nullptr,
        *msize = nullptr, *comp_list = nullptr, *counters = nullptr, *bsum = nullptr;
    uint8_t *blb = nullptr, *mk = nullptr, *mk2 = nullptr;
    double *partial = nullptr, *params_hv = nullptr, *params_sob = nullptr, *tmp_h = nullptr, *tmp_v = nullptr, *sob = nullptr,
           *d0 = nullptr, *dist = nullptr, *ovf_v = nullptr;
    unsigned long long* ovf_lo = nullptr; int* ovf_lab = nullptr;
    unsigned long long* ovf_cursor = nullptr;
    StatArrays st{};
    VoteArrays vote{};               // allocated on the first call with more than 8 nucleus classes
    int list_cap = 0, nblk = 0;
};

// type vote over windows of 8 classes (nr_types > 8): overrides the type / type_prob fields k_inst_records wrote
static int vote_wide(PostprocWorkspace* w, const int32_t* inst, const uint8_t* type, int B, int nr_types, InstanceRec* recs,
                     const int32_t* n_recs, const int* nmark, hipStream_t st) {
    const PostprocDims& d = w->d;
    const size_t S = (size_t)d.B * (size_t)(d.max_ids + 1);
    if (!w->vote.best_t) {
        void* p[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < 5; ++k) {
            if (hipMalloc(&p[k], S * 4) != hipSuccess) return 1;
            w->pool.push_back(p[k]); w->bytes += S * 4;
        }
        w->vote.best_t = (int*)p[0]; w->vote.best_c = (unsigned*)p[1]; w->vote.second_t = (int*)p[2]; w->vote.second_c = (unsigned*)p[3];
        w->vote.present = (int*)p[4];
    }
    const int H = d.H, W = d.W, N = H * W;
    const dim3 grid(std::min((N + NT - 1) / NT, 2048), B), blk(NT), igrid(std::min((d.max_ids + NT) / NT, 32), B);
    for (int t_lo = 0; t_lo < nr_types; t_lo += 8) {
        hipLaunchKernelGGL(k_vote_reset, igrid, blk, 0, st, w->vote, w->st.hist, nmark, d.max_ids, t_lo == 0 ? 1 : 0);
        hipLaunchKernelGGL(k_type_hist_window, grid, blk, 0, st, inst, type, w->st.hist, H, W, d.max_ids, t_lo);
        hipLaunchKernelGGL(k_vote_merge, igrid, blk, 0, st, w->vote, w->st.hist, nmark, d.max_ids, t_lo, nr_types);
    }
    const dim3 cgrid((d.max_inst + NT - 1) / NT > 64 ? 64 : (d.max_inst + NT - 1) / NT, B);
    hipLaunchKernelGGL(k_vote_apply, cgrid, blk, 0, st, w->vote, recs, n_recs, d.max_ids, d.max_inst);
    return 0;
}

int pp_workspace_create(const PostprocDims& d, PostprocWorkspace** out) {
    PostprocWorkspace* w = new PostprocWorkspace();
    w->d = d;
    const size_t N = (size_t)d.H * d.W, B = d.B;
    auto A = [&](void** p, size_t bytes) -> bool {
        if (hipMalloc(p, bytes ? bytes : 16) != hipSuccess) return false;
        w->pool.push_back(*p); w->bytes += bytes;
        return true;
    };
    w->list_cap = (int)(N / 10 + 16);
    w->nblk = (int)((N + SCAN_ELEMS - 1) / SCAN_ELEMS);
    bool ok = true;
    ok = ok && A((void**)&w->L1, B * N * 4) && A((void**)&w->L2, B * N * 4) && A((void**)&w->csize, B * N * 4) &&
         A((void**)&w->bb, B * N * 16) && A((void**)&w->flag, B * N * 4) && A((void**)&w->rank, B * N * 4) &&
         A((void**)&w->marker, B * N * 4) && A((void**)&w->msize, B * (size_t)(d.max_ids + 1) * 4) &&
         A((void**)&w->comp_list, B * (size_t)w->list_cap * 4) && A((void**)&w->counters, B * 4 * 4) &&
         A((void**)&w->bsum, B * (size_t)w->nblk * 4) && A((void**)&w->blb, B * N) && A((void**)&w->mk, B * N) &&
         A((void**)&w->mk2, B * N) && A((void**)&w->partial, B * 2 * RED_BLOCKS * 2 * 8) && A((void**)&w->params_hv, B * 4 * 8) &&
         A((void**)&w->params_sob, B * 4 * 8) && A((void**)&w->tmp_h, B * N * 8) && A((void**)&w->tmp_v, B * N * 8) &&
         A((void**)&w->sob, B * 2 * N * 8) && A((void**)&w->d0, B * N * 8) && A((void**)&w->dist, B * N * 8) &&
         A((void**)&w->ovf_v, B * N * 8) && A((void**)&w->ovf_lo, B * N * 8) &&
         A((void**)&w->ovf_lab, B * N * 4) && A((void**)&w->ovf_cursor, B * 8);
    const size_t S = B * (size_t)(d.max_ids + 1);
    ok = ok && A((void**)&w->st.cnt, S * 4) && A((void**)&w->st.sx, S * 8) && A((void**)&w->st.sy, S * 8) &&
         A((void**)&w->st.rmin, S * 4) && A((void**)&w->st.rmax, S * 4) && A((void**)&w->st.cmin, S * 4) &&
         A((void**)&w->st.cmax, S * 4) && A((void**)&w->st.first, S * 4) && A((void**)&w->st.hist, S * 8 * 4) &&
         A((void**)&w->st.has_zero, B * 4);
    if (!ok || w->nblk > NT * 8) { pp_workspace_destroy(w); return 1; }
    *out = w;
    return 0;
}

void pp_workspace_destroy(PostprocWorkspace* w) {
    if (!w) return;
    for (void* p : w->pool) (void)hipFree(p);
    delete w;
}

size_t pp_workspace_bytes(const PostprocWorkspace* w) { return w->bytes; }
const double* pp_dbg_dist(const PostprocWorkspace* w) { return w->dist; }
const int32_t* pp_dbg_marker(const PostprocWorkspace* w) { return w->marker; }
const uint8_t* pp_dbg_blb_u8(const PostprocWorkspace* w) { return w->blb; }
const int32_t* pp_dbg_blb(const PostprocWorkspace*) { return nullptr; }

int pp_run(PostprocWorkspace* w, const uint8_t* bin, const uint8_t* type, const float* hv, int B, int object_size, int ksize,
           int nr_types, int32_t* inst_out, InstanceRec* recs, int32_t* n_recs, int32_t* contours, int32_t* n_pts,
           hipStream_t st) {
    const PostprocDims& d = w->d;
    if (B > d.B || B <= 0 || (ksize != 21 && ksize != 11)) return 1;
    const int H = d.H, W = d.W, N = H * W;
    const int gx = std::min((N + NT - 1) / NT, 2048);
    const dim3 grid(gx, B), blk(NT);
    SobelTaps taps;
    sobel_taps(ksize, &taps);
    // No hipMemset anywhere in this chain: accumulator planes are initialised at the component roots by the flatten pass
    // that precedes their first use (k_cc_flatten<INIT>), per-id arrays for the live ids only (k_stats_init).
    // ---- P1: mask, 4-connected components, small-object removal (hard-wired 10) ----
    hipLaunchKernelGGL(k_init_small, dim3((4 * B + NT - 1) / NT), blk, 0, st, w->counters, w->ovf_cursor, B);
    const bool runs = (W % 64) == 0;
    auto cc = [&](const uint8_t* src, int invert, int* L, int init) {
        if (runs) {
            hipLaunchKernelGGL(k_cc_init_runs, grid, blk, 0, st, src, invert, L, N);
            hipLaunchKernelGGL(k_cc_merge_runs, grid, blk, 0, st, L, H, W);
        } else {
            hipLaunchKernelGGL(k_cc_init, grid, blk, 0, st, src, invert, L, N);
            hipLaunchKernelGGL(k_cc_merge, grid, blk, 0, st, L, H, W);
        }
        if (init == 1) hipLaunchKernelGGL(k_cc_flatten<1>, grid, blk, 0, st, L, N, w->csize, w->bb, w->flag);
        else if (init == 2) hipLaunchKernelGGL(k_cc_flatten<2>, grid, blk, 0, st, L, N, w->csize, w->bb, w->flag);
        else hipLaunchKernelGGL(k_cc_flatten<0>, grid, blk, 0, st, L, N, w->csize, w->bb, w->flag);
    };
    cc(bin, 0, w->L1, 1);
    if (runs) hipLaunchKernelGGL(k_comp_stats_runs, grid, blk, 0, st, w->L1, w->csize, w->bb, H, W);
    else hipLaunchKernelGGL(k_comp_stats, grid, blk, 0, st, w->L1, w->csize, w->bb, H, W);
    int* comp_count = w->counters;            // [B]
    int* queue_head = w->counters + B;        // [B]
    int* big_count = w->counters + 3 * B;     // [B]
    hipLaunchKernelGGL(k_blb_finalize, grid, blk, 0, st, w->L1, w->csize, w->blb, w->comp_list, comp_count, big_count, N, w->list_cap);
    // ---- P2/P3: min-max normalise (fused) + separable Sobel in fp64 ----
    hipLaunchKernelGGL((k_minmax_partial<float>), dim3(RED_BLOCKS, 2, B), blk, 0, st, hv, 2, N, w->partial);
    hipLaunchKernelGGL(k_minmax_final, dim3(2, B), blk, 0, st, w->partial, 2, w->params_hv);
    const bool quad = (W % 4) == 0 && (H % 4) == 0 && W >= 32 && H >= 32;      // four outputs per thread (see k_sobel_row4)
    const dim3 gridq(std::min((N / 4 + NT - 1) / NT, 2048), B);
    if (quad && ksize == 21) {
        hipLaunchKernelGGL((k_sobel_row4<21>), gridq, blk, 0, st, hv, w->params_hv, taps, w->tmp_h, w->tmp_v, H, W);
        hipLaunchKernelGGL((k_sobel_col4<21>), gridq, blk, 0, st, w->tmp_h, w->tmp_v, taps, w->sob, H, W);
    } else if (quad) {
        hipLaunchKernelGGL((k_sobel_row4<11>), gridq, blk, 0, st, hv, w->params_hv, taps, w->tmp_h, w->tmp_v, H, W);
        hipLaunchKernelGGL((k_sobel_col4<11>), gridq, blk, 0, st, w->tmp_h, w->tmp_v, taps, w->sob, H, W);
    } else {
        hipLaunchKernelGGL(k_sobel_row, grid, blk, 0, st, hv, w->params_hv, taps, w->tmp_h, w->tmp_v, H, W);
        hipLaunchKernelGGL(k_sobel_col, grid, blk, 0, st, w->tmp_h, w->tmp_v, taps, w->sob, H, W);
    }
    hipLaunchKernelGGL((k_minmax_partial<double>), dim3(RED_BLOCKS, 2, B), blk, 0, st, w->sob, 2, N, w->partial);
    hipLaunchKernelGGL(k_minmax_final, dim3(2, B), blk, 0, st, w->partial, 2, w->params_sob);
    // ---- P4: combine, blur ----
    hipLaunchKernelGGL(k_combine, grid, blk, 0, st, w->sob, w->params_sob, w->blb, w->d0, w->mk, N);
    hipLaunchKernelGGL(k_blur_neg, grid, blk, 0, st, w->d0, w->dist, H, W);
    // ---- P5: fill holes (background components not touching the border), open, label, size filter ----
    cc(w->mk, 1, w->L2, 2);
    hipLaunchKernelGGL(k_border_flag, dim3((2 * (H + W) + NT - 1) / NT, B), blk, 0, st, w->L2, w->flag, H, W);
    hipLaunchKernelGGL(k_fill, grid, blk, 0, st, w->mk, w->L2, w->flag, w->mk2, N);
    if (quad) {
        hipLaunchKernelGGL((k_morph5_w4<true>), gridq, blk, 0, st, w->mk2, w->mk, H, W);
        hipLaunchKernelGGL((k_morph5_w4<false>), gridq, blk, 0, st, w->mk, w->mk2, H, W);
    } else {
        hipLaunchKernelGGL((k_morph5<true>), grid, blk, 0, st, w->mk2, w->mk, H, W);
        hipLaunchKernelGGL((k_morph5<false>), grid, blk, 0, st, w->mk, w->mk2, H, W);
    }
    cc(w->mk2, 0, w->L2, 0);
    hipLaunchKernelGGL(k_scan_partial, dim3(w->nblk, B), blk, 0, st, w->L2, N, w->bsum, w->nblk);
    int* nmark = w->counters + 2 * B;         // [B]
    hipLaunchKernelGGL(k_scan_blocks, dim3(B), blk, 0, st, w->bsum, w->nblk, nmark);
    hipLaunchKernelGGL(k_scan_apply, dim3(w->nblk, B), blk, 0, st, w->L2, N, w->bsum, w->nblk, w->rank);
    hipLaunchKernelGGL(k_stats_init, dim3(std::min((d.max_ids + NT) / NT, 32), B), blk, 0, st, w->st, w->msize, nmark, d.max_ids);
    if (runs) hipLaunchKernelGGL(k_marker_ids_runs, grid, blk, 0, st, w->L2, w->rank, w->marker, w->msize, N, W, d.max_ids);
    else hipLaunchKernelGGL(k_marker_ids, grid, blk, 0, st, w->L2, w->rank, w->marker, w->msize, N, W, d.max_ids);
    hipLaunchKernelGGL(k_marker_filter, grid, blk, 0, st, w->marker, w->msize, object_size, w->blb, inst_out, N, d.max_ids);
    // ---- P6: ordered flood ----
    FloodParams fp{};
    fp.dist = w->dist; fp.blb = w->blb; fp.inst = inst_out; fp.root1 = w->L1; fp.bb = w->bb; fp.csize = w->csize;
    fp.comp_list = w->comp_list; fp.comp_count = comp_count; fp.big_count = big_count; fp.list_cap = w->list_cap; fp.queue_head = queue_head;
    fp.ovf_hi = reinterpret_cast<unsigned long long*>(w->ovf_v); fp.ovf_lo = w->ovf_lo; fp.ovf_lab = w->ovf_lab; fp.ovf_cursor = w->ovf_cursor;
    fp.H = H; fp.W = W; fp.B = B;
    { static const int dbg = cva_env_int("CVA_PP_DBG", 0); fp.dbg = dbg; }   // ablation builds only (common.h)
    hipLaunchKernelGGL(k_flood, dim3(B, 1024), dim3(64), 0, st, fp);
    // ---- P7/P8: per-instance records + contours ----
    if (runs) hipLaunchKernelGGL(k_inst_stats_runs, grid, blk, 0, st, inst_out, type, w->st, H, W, d.max_ids, nr_types);
    else hipLaunchKernelGGL(k_inst_stats, grid, blk, 0, st, inst_out, type, w->st, H, W, d.max_ids, nr_types);
    hipLaunchKernelGGL(k_inst_records, dim3(B), blk, 0, st, w->st, recs, n_recs, d.max_ids, d.max_inst, nr_types, nmark);
    if (nr_types > 8 && vote_wide(w, inst_out, type, B, nr_types, recs, n_recs, nmark, st)) return 1;
    const dim3 cgrid((d.max_inst + 63) / 64 > 64 ? 64 : (d.max_inst + 63) / 64, B);
    hipLaunchKernelGGL(k_contour_count, cgrid, dim3(64), 0, st, inst_out, recs, n_recs, H, W, d.max_inst);
    hipLaunchKernelGGL(k_contour_offsets, dim3(B), blk, 0, st, recs, n_recs, n_pts, d.max_inst);
    if (contours)
        hipLaunchKernelGGL(k_contour_write, cgrid, dim3(64), 0, st, inst_out, recs, n_recs, contours, H, W, d.max_inst, d.max_pts);
    return (int)hipGetLastError();
}

__global__ void k_fill_counts(int* __restrict__ dst, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}

// Records + contours of GIVEN instance maps (post_proc_cellvit.py:252-330, `calculate_instances`, the ground-truth side of
// the evaluation callers): the P7/P8 tail of pp_run on caller-supplied ids.  Ids above max_ids have no accumulator
// slot and are not reported; ids <= 0 are background (negative ones are zeroed in place like pp_run does).
int pp_records(PostprocWorkspace* w, int32_t* inst_io, const uint8_t* type, int B, int nr_types, InstanceRec* recs,
               int32_t* n_recs, int32_t* contours, int32_t* n_pts, hipStream_t st) {
    const PostprocDims& d = w->d;
    if (B > d.B || B <= 0) return 1;
    const int H = d.H, W = d.W, N = H * W;
    const dim3 grid(std::min((N + NT - 1) / NT, 2048), B), blk(NT);
    int* nmark = w->counters + 2 * B;         // [B]: here simply "every id slot may be live"
    hipLaunchKernelGGL(k_fill_counts, dim3((B + NT - 1) / NT), blk, 0, st, nmark, B, d.max_ids);
    hipLaunchKernelGGL(k_stats_init, dim3(std::min((d.max_ids + NT) / NT, 32), B), blk, 0, st, w->st, w->msize, nmark, d.max_ids);
    if ((W % 64) == 0) hipLaunchKernelGGL(k_inst_stats_runs, grid, blk, 0, st, inst_io, type, w->st, H, W, d.max_ids, nr_types);
    else hipLaunchKernelGGL(k_inst_stats, grid, blk, 0, st, inst_io, type, w->st, H, W, d.max_ids, nr_types);
    hipLaunchKernelGGL(k_inst_records, dim3(B), blk, 0, st, w->st, recs, n_recs, d.max_ids, d.max_inst, nr_types, nmark);
    if (nr_types > 8 && vote_wide(w, inst_io, type, B, nr_types, recs, n_recs, nmark, st)) return 1;
    const dim3 cgrid((d.max_inst + 63) / 64 > 64 ? 64 : (d.max_inst + 63) / 64, B);
    hipLaunchKernelGGL(k_contour_count, cgrid, dim3(64), 0, st, inst_io, recs, n_recs, H, W, d.max_inst);
    hipLaunchKernelGGL(k_contour_offsets, dim3(B), blk, 0, st, recs, n_recs, n_pts, d.max_inst);
    if (contours)
        hipLaunchKernelGGL(k_contour_write, cgrid, dim3(64), 0, st, inst_io, recs, n_recs, contours, H, W, d.max_inst, d.max_pts);
    return (int)hipGetLastError();
}

}  // namespace cva
