// MFMA tile contraction for gfx950 — see gemm.h for the role of each mode.
//
// Geometry: 256 threads = 4 waves (2 x 2), block tile 128 x 128, wave tile 64 x 64 = 4 x 4
// fragments of v_mfma_f32_16x16x32_f16 (or 8 x v_mfma_f32_16x16x4_f32 in the fp32 parity path).
// K is consumed in 128-byte rows (64 halves / 32 floats) staged through LDS with a +32 B row
// pitch (bank-conflict-free ds_read_b128 fragment reads); the next K tile is prefetched into
// registers while the current one is multiplied.
#include "gemm.h"
#include "gemm_epilogue.h"

#include <stdlib.h>

namespace cva {

namespace {

constexpr int BM = 128, BN = 128, NT = 256;

using namespace epi;

template <typename T, int AMODE, int OMODE>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(const GemmParams p) {
    using TR = Traits<T>;
    constexpr int BK = TR::BK, PE = TR::PIECE, PITCH = lds_pitch<T>(BK);
    constexpr int PPR = BK / PE;            // 16-B pieces per tile row (8)
    constexpr int RPP = NT / PPR;           // rows covered per load pass (32)
    constexpr int APASS = BM / RPP, BPASS = BN / RPP;
    constexpr int KSTEPS = BK / 32;

    __shared__ __attribute__((aligned(16))) T smem[(BM + BN) * PITCH];
    T* As = smem;
    T* Bs = smem + BM * PITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    int tm_, tn_;
    if (p.dbg & 16) { const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n); tm_ = bid / tiles_n; tn_ = bid % tiles_n; }
    else tile_coords(xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm_, tn_);
    const int m0 = tm_ * BM, n0 = tn_ * BN;

    const int pc = tid % PPR, pr = tid / PPR;
    const T* __restrict__ Ap = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ A2p = reinterpret_cast<const T*>(p.A2);
    const T* __restrict__ Wp = reinterpret_cast<const T*>(p.W);

    // ---- per-thread row descriptors of the A and W tiles (fixed over the K loop) ----
    long a_off[APASS];      // A_LINEAR: element offset of the row; A_CONV3: unused
    int a_b[APASS], a_y[APASS], a_x[APASS];
    bool a_ok[APASS];
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int m = m0 + pr + RPP * i;
        a_ok[i] = m < p.M;
        if (AMODE == A_LINEAR) {
            long row = m;
            if (p.a_rpi > 0) row = (long)m + (long)(m / p.a_rpi) * p.a_extra + p.a_off;
            a_off[i] = row * (long)p.lda;
            a_b[i] = a_y[i] = a_x[i] = 0;
        } else {
            const int hw = p.H * p.Wd;
            const int b = m / hw, r = m - b * hw;
            a_b[i] = b; a_y[i] = r / p.Wd; a_x[i] = r - a_y[i] * p.Wd;
            a_off[i] = 0;
        }
    }
    long w_off[BPASS];
    bool w_ok[BPASS];
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
        const int n = n0 + pr + RPP * i;
        w_ok[i] = n < p.N;
        w_off[i] = (long)n * p.ldw;
    }
    const int ctot = p.C1 + p.C2;
    const int ppt = (AMODE == A_CONV3) ? ctot / PE : 1;   // pieces per filter tap

    Piece ra[APASS], rb[BPASS];
    auto fetch = [&](int k0) {
        const int k = k0 + pc * PE;
        if (AMODE == A_LINEAR) {
            const bool kok = k < p.K;
#pragma unroll
            for (int i = 0; i < APASS; ++i)
                ra[i] = (a_ok[i] && kok) ? load_piece(Ap + a_off[i] + k) : zero_piece();
        } else {
            const int q = k / PE;
            const int tap = q / ppt;
            const int c = (q - tap * ppt) * PE;
            const bool kok = tap < 9;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const bool second = c >= p.C1;
            const T* __restrict__ src = second ? A2p : Ap;
            const int cs = second ? p.C2 : p.C1;
            const int cc = second ? c - p.C1 : c;
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                const int yy = a_y[i] + dy, xx = a_x[i] + dx;
                const bool ok = kok && a_ok[i] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
                ra[i] = ok ? load_piece(src + ((long)(a_b[i] * p.H + yy) * p.Wd + xx) * cs + cc) : zero_piece();
            }
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) rb[i] = w_ok[i] ? load_piece(Wp + w_off[i] + k) : zero_piece();
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4)(0.f);

    const int nk = (p.K + BK - 1) / BK;
    fetch(0);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < APASS; ++i) store_piece(As + (pr + RPP * i) * PITCH + pc * PE, ra[i]);
#pragma unroll
        for (int i = 0; i < BPASS; ++i) store_piece(Bs + (pr + RPP * i) * PITCH + pc * PE, rb[i]);
        __syncthreads();
        if (kt + 1 < nk) fetch((kt + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            typename TR::Frag a[4], b[4];
            const int ko = ks * 32 + (lane >> 4) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = TR::load_frag(As + (wm * 64 + i * 16 + (lane & 15)) * PITCH + ko);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = TR::load_frag(Bs + (wn * 64 + j * 16 + (lane & 15)) * PITCH + ko);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) TR::mma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }

    if (p.epi_vec) {
        float* st = reinterpret_cast<float*>(smem) + wave * (16 * 68);
        gemm_epilogue_lds<T, OMODE, 4, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, st, lane);
    } else {
        gemm_epilogue<T, OMODE>(p, acc, m0 + wm * 64 + (lane >> 4) * 4, n0 + wn * 64 + (lane & 15));
    }
}


// ================================================================================================
// v2: direct-to-LDS staging (global_load_lds dwordx4), XOR-swizzled 128-byte rows, double-buffered.
//
// Each 1-KiB wave-instruction lands 8 tile rows x 128 B linearly in LDS (destination = wave-uniform
// base + lane*16).  The bank-conflict fix therefore goes on the SOURCE address: lane l fetches the
// logical 16-B piece (l&7) ^ ((row>>1)&7) of its row, and fragment reads apply the same involution.
// With 128-B rows two rows share one 256-B bank row, so XOR-ing with (row>>1)&7 spreads the 16 rows
// of every ds_read_b128 lane group over all 16 slots.  The permutation stays inside
// one 128-B line: global coalescing is unchanged.  Out-of-range rows / K tail read a zero page.
// One barrier per K tile; the next tile's DMA is in flight while the current one is multiplied.
// ================================================================================================
template <typename T, int AMODE, int OMODE, int WMW, int WNW, int MI, int NJ>
__global__ __launch_bounds__(WMW * WNW * 64) void gemm_glds_kernel(const GemmParams p) {
    using TR = Traits<T>;
    constexpr int BK = TR::BK, PE = TR::PIECE;
    constexpr int TBM = WMW * MI * 16, TBN = WNW * NJ * 16, NTH = WMW * WNW * 64, NW = NTH / 64;
    constexpr int ROWB = 128;                         // bytes per tile row
    constexpr int A_CHUNKS = TBM / 8, B_CHUNKS = TBN / 8;   // 1-KiB chunks (8 rows each)
    constexpr int A_PER_WAVE = A_CHUNKS / NW, B_PER_WAVE = B_CHUNKS / NW;
    static_assert(A_CHUNKS % NW == 0 && B_CHUNKS % NW == 0, "tile rows must split evenly over the waves");
    constexpr int KSTEPS = BK / 32;
    constexpr int BUF_BYTES = (TBM + TBN) * ROWB;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WNW, wn = wave - wm * WNW;
    const int tiles_n = (p.N + TBN - 1) / TBN, tiles_m = (p.M + TBM - 1) / TBM;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * TBM, n0 = tn * TBN;

    const T* __restrict__ Ap = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ A2p = reinterpret_cast<const T*>(p.A2);
    const T* __restrict__ Wp = reinterpret_cast<const T*>(p.W);
    const T* __restrict__ Zp = reinterpret_cast<const T*>(p.zero);

    // ---- per-lane descriptors of the rows this lane stages (fixed over the K loop) ----
    const int lrow = lane >> 3;                  // row inside the 8-row chunk
    const int lpc = lane & 7;                    // physical 16-B piece inside the row
    long a_off[A_PER_WAVE]; int a_b[A_PER_WAVE], a_y[A_PER_WAVE], a_x[A_PER_WAVE], a_lp[A_PER_WAVE];
    bool a_ok[A_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
        const int row = (wave * A_PER_WAVE + i) * 8 + lrow;
        const int m = m0 + row;
        a_ok[i] = m < p.M;
        a_lp[i] = lpc ^ ((row >> 1) & 7);        // logical piece this lane must fetch
        if (AMODE == A_LINEAR) {
            long r = m;
            if (p.a_rpi > 0) r = (long)m + (long)(m / p.a_rpi) * p.a_extra + p.a_off;
            a_off[i] = r * (long)p.lda;
            a_b[i] = a_y[i] = a_x[i] = 0;
        } else {
            const int hw = p.H * p.Wd;
            const int b = m / hw, r = m - b * hw;
            a_b[i] = b; a_y[i] = r / p.Wd; a_x[i] = r - a_y[i] * p.Wd; a_off[i] = 0;
        }
    }
    long w_off[B_PER_WAVE]; int w_lp[B_PER_WAVE]; bool w_ok[B_PER_WAVE];
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) {
        const int row = (wave * B_PER_WAVE + i) * 8 + lrow;
        const int n = n0 + row;
        w_ok[i] = n < p.N;
        w_lp[i] = lpc ^ ((row >> 1) & 7);
        w_off[i] = (long)n * p.ldw;
    }
    const int ctot = p.C1 + p.C2;
    const int ppt = (AMODE == A_CONV3) ? ctot / PE : 1;

    auto stage = [&](int kt, int buf) {
        unsigned char* sA = smem_raw + buf * BUF_BYTES;
        unsigned char* sB = sA + TBM * ROWB;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < A_PER_WAVE; ++i) {
            const T* src = Zp;
            const int k = k0 + a_lp[i] * PE;
            if (AMODE == A_LINEAR) {
                if (a_ok[i] && k < p.K) src = Ap + a_off[i] + k;
            } else {
                const int q = k / PE;
                const int tap = q / ppt;
                const int c = (q - tap * ppt) * PE;
                const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                const int yy = a_y[i] + dy, xx = a_x[i] + dx;
                if (tap < 9 && a_ok[i] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd) {
                    const bool second = c >= p.C1;
                    const T* base = second ? A2p : Ap;
                    const int cs = second ? p.C2 : p.C1;
                    src = base + ((long)(a_b[i] * p.H + yy) * p.Wd + xx) * cs + (second ? c - p.C1 : c);
                }
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sA + (wave * A_PER_WAVE + i) * 1024),
                                             16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_PER_WAVE; ++i) {
            const T* src = w_ok[i] ? Wp + w_off[i] + k0 + w_lp[i] * PE : Zp;   // W rows are zero padded to ldw
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sB + (wave * B_PER_WAVE + i) * 1024),
                                             16, 0, 0);
        }
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4)(0.f);

    // fragment read addressing: row r, logical piece lp -> byte r*128 + ((lp ^ ((r>>1)&7)) << 4)
    const int g = lane >> 4;
    const int a_row0 = wm * (MI * 16) + (lane & 15), b_row0 = wn * (NJ * 16) + (lane & 15);

    auto load_frag = [&](const unsigned char* base, int row, int ks) -> typename TR::Frag {
        const int sw = (row >> 1) & 7;
        if constexpr (sizeof(T) == 2) {
            const int lp = ks * 4 + g;
            return TR::load_frag(reinterpret_cast<const T*>(base + row * ROWB + ((lp ^ sw) << 4)));
        } else {
            typename TR::Frag f;
            const int lp = 2 * g;    // BK = 32 floats: one k-step spans the whole row
            const f32x4 lo = *reinterpret_cast<const f32x4*>(base + row * ROWB + ((lp ^ sw) << 4));
            const f32x4 hi = *reinterpret_cast<const f32x4*>(base + row * ROWB + (((lp + 1) ^ sw) << 4));
            f.v[0] = lo[0]; f.v[1] = lo[1]; f.v[2] = lo[2]; f.v[3] = lo[3];
            f.v[4] = hi[0]; f.v[5] = hi[1]; f.v[6] = hi[2]; f.v[7] = hi[3];
            return f;
        }
    };

    // ---- L2 prefetch: one 4-byte load per thread touches every 128-B line of tile kt+2 (A rows then W rows),
    // issued AFTER the DMA of tile kt+1 and left in flight across the barrier (counted vmcnt).  The DMA of the
    // following iteration then hits L2, so the per-iteration wait is an L2 round trip instead of an HBM/MALL one.
    constexpr bool L2PF = (AMODE == A_LINEAR) && (TBM + TBN) == NTH;    // exactly one line per thread
    unsigned pf_dummy = 0;
    const T* pf_row = Zp;
    bool pf_ok = false;
    if (L2PF) {
        if (tid < TBM) {
            const int m = m0 + tid;
            if (m < p.M) {
                long r = m;
                if (p.a_rpi > 0) r = (long)m + (long)(m / p.a_rpi) * p.a_extra + p.a_off;
                pf_row = Ap + r * (long)p.lda; pf_ok = true;
            }
        } else {
            const int n = n0 + (tid - TBM);
            if (n < p.N) { pf_row = Wp + (long)n * p.ldw; pf_ok = true; }
        }
    }

    const int nk = (p.K + BK - 1) / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk && !(p.dbg & 1)) stage(kt + 1, buf ^ 1);
        const bool do_pf = L2PF && (kt + 2 < nk) && !(p.dbg & 4);          // block-uniform
        if (do_pf) {
            const T* src = pf_ok ? pf_row + (long)(kt + 2) * BK : Zp;
            asm volatile("global_load_dword %0, %1, off" : "+v"(pf_dummy) : "v"(src) : "memory");
        }
        const unsigned char* sA = smem_raw + buf * BUF_BYTES;
        const unsigned char* sB = sA + TBM * ROWB;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            typename TR::Frag b[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = load_frag(sB, b_row0 + j * 16, ks);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const typename TR::Frag a = load_frag(sA, a_row0 + i * 16, ks);
#pragma unroll
                for (int j = 0; j < NJ; ++j) TR::mma(a, b[j], acc[i][j]);
            }
        }
        if (L2PF) {
            // tile kt+1 has landed (all older VMEM ops retire in order); the prefetch may stay in flight
            if (do_pf) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_dummy) :: "memory");
    if (p.epi_vec && NJ == 4) {
        float* st = reinterpret_cast<float*>(smem_raw) + wave * (16 * 68);
        gemm_epilogue_lds<T, OMODE, MI, NJ>(p, acc, m0 + wm * (MI * 16), n0 + wn * (NJ * 16), st, lane);
    } else {
        gemm_epilogue<T, OMODE, MI, NJ>(p, acc, m0 + wm * (MI * 16) + (lane >> 4) * 4, n0 + wn * (NJ * 16) + (lane & 15));
    }
}

}  // namespace

namespace {
void* zero_page() {   // 256 B of zeros: the DMA source of out-of-range pieces
    static void* z = nullptr;
    if (!z) { if (hipMalloc(&z, 256) != hipSuccess) return nullptr; (void)hipMemset(z, 0, 256); }
    return z;
}
int gemm_variant() {   // CVA_GEMM=8 (default): 8-phase 256x256 where the shape qualifies, else 1; 1: register-staged 128x128;
                       // 2: glds 128x128; 3: glds 256x128; 4: glds 256x256 (experiments)
    static int v = -1;
    if (v < 0) { const char* e = getenv("CVA_GEMM"); v = e ? atoi(e) : 8; if (v != 8 && (v < 1 || v > 4)) v = 8; }
    return v;
}
template <typename T, int AMODE, int OMODE, int WMW, int WNW, int MI, int NJ>
int launch_glds(const GemmParams& p, hipStream_t stream) {
    constexpr int TBM = WMW * MI * 16, TBN = WNW * NJ * 16;
    const int tiles = ((p.M + TBM - 1) / TBM) * ((p.N + TBN - 1) / TBN);
    const size_t lds = 2 * (size_t)(TBM + TBN) * 128;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<T, AMODE, OMODE, WMW, WNW, MI, NJ>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return (int)hipGetLastError();
        attr = true;
    }
    hipLaunchKernelGGL((gemm_glds_kernel<T, AMODE, OMODE, WMW, WNW, MI, NJ>), dim3(tiles), dim3(WMW * WNW * 64), lds, stream, p);
    return (int)hipGetLastError();
}
template <typename T, int WMW, int WNW, int MI, int NJ>
int dispatch_glds(const GemmParams& p, int a_mode, hipStream_t stream) {
    if (a_mode == A_CONV3) {
        if (p.out_mode != OUT_LINEAR) return (int)hipErrorInvalidValue;
        return launch_glds<T, A_CONV3, OUT_LINEAR, WMW, WNW, MI, NJ>(p, stream);
    }
    if (p.out_mode == OUT_LINEAR) return launch_glds<T, A_LINEAR, OUT_LINEAR, WMW, WNW, MI, NJ>(p, stream);
    if (p.out_mode == OUT_QKV) return launch_glds<T, A_LINEAR, OUT_QKV, WMW, WNW, MI, NJ>(p, stream);
    return launch_glds<T, A_LINEAR, OUT_CONVT, WMW, WNW, MI, NJ>(p, stream);
}
}  // namespace

void* gemm_zero_page() { return zero_page(); }

template <typename T>
int launch_gemm(const GemmParams& p_in, int a_mode, hipStream_t stream) {
    GemmParams p = p_in;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("CVA_GEMM_DBG"); dbg = e ? atoi(e) : 0; } p.dbg = dbg; }
    if (((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) <= 0) return 0;
    {   // vectorised (LDS-staged) epilogue preconditions
        static int epi = -1;
        if (epi < 0) { const char* e = getenv("CVA_EPI"); epi = e ? atoi(e) : 1; }
        const size_t es = sizeof(T);
        bool ok = epi != 0 && p.N % 8 == 0;
        auto al16 = [](const void* q) { return ((size_t)q & 15) == 0; };
        if (p.out_mode == OUT_LINEAR) {
            ok = ok && al16(p.out) && ((size_t)p.ldc * (p.out_f32 ? 4 : es)) % 16 == 0;
            if (p.res) ok = ok && al16(p.res) && p.ldres % 4 == 0;
        } else if (p.out_mode == OUT_QKV) {
            ok = ok && p.hd % 8 == 0 && p.D % 64 == 0 && al16(p.q_out) && al16(p.k_out) && al16(p.vt_out) && p.Lp % 8 == 0;
        } else {
            ok = ok && (p.N / 4) % 8 == 0 && al16(p.out);
        }
        p.epi_vec = ok ? 1 : 0;
    }
    const int variant = gemm_variant();
    if (variant == 8) {
        const long t256 = (long)(p.M / 256) * (p.N / 256);
        if (t256 >= 192 && gemm8_supported(p, a_mode, sizeof(T))) return launch_gemm8(p, stream);   // >= 75 % of the 256 CUs
    } else if (variant >= 2) {
        p.zero = zero_page();
        if (!p.zero) return (int)hipErrorOutOfMemory;
        // DMA pieces are whole 16-B units: the linear A operand needs 16-B aligned rows
        const bool aligned = a_mode == A_CONV3 || ((size_t)p.lda * sizeof(T)) % 16 == 0;
        if (aligned) {
            // big tiles only pay when the grid still fills the chip (256 CUs) about twice over
            const long t256x256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
            const long t256x128 = (long)((p.M + 255) / 256) * ((p.N + 127) / 128);
            if (variant == 4 && p.N >= 256 && t256x256 >= 256) return dispatch_glds<T, 2, 4, 8, 4>(p, a_mode, stream);
            if (variant >= 3 && p.N <= 64 && (p.M + 255) / 256 >= 512) return dispatch_glds<T, 4, 1, 4, 4>(p, a_mode, stream);
            if (variant >= 3 && t256x128 >= 512) return dispatch_glds<T, 4, 2, 4, 4>(p, a_mode, stream);
            if (variant == 2) return dispatch_glds<T, 2, 2, 4, 4>(p, a_mode, stream);
        }
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const dim3 g(tiles), b(NT);
    if (a_mode == A_CONV3) {
        if (p.out_mode != OUT_LINEAR) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL((gemm_kernel<T, A_CONV3, OUT_LINEAR>), g, b, 0, stream, p);
    } else if (p.out_mode == OUT_LINEAR) {
        hipLaunchKernelGGL((gemm_kernel<T, A_LINEAR, OUT_LINEAR>), g, b, 0, stream, p);
    } else if (p.out_mode == OUT_QKV) {
        hipLaunchKernelGGL((gemm_kernel<T, A_LINEAR, OUT_QKV>), g, b, 0, stream, p);
    } else {
        hipLaunchKernelGGL((gemm_kernel<T, A_LINEAR, OUT_CONVT>), g, b, 0, stream, p);
    }
    return (int)hipGetLastError();
}

template int launch_gemm<half_t>(const GemmParams&, int, hipStream_t);
template int launch_gemm<float>(const GemmParams&, int, hipStream_t);

}  // namespace cva
