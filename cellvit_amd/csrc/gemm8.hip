// 256 x 256 x 64 fp16 MFMA contraction with an 8-phase, two-K-tiles-per-iteration schedule (gfx950).
//
// 512 threads = 8 waves laid out 2 (M) x 4 (N); every wave owns a 128 x 64 block of C = 8 x 4
// accumulator fragments (128 VGPRs) and walks it one 64 x 32 quadrant (16 MFMAs) per phase.
//
// LDS (128 KiB, one workgroup per CU): two K-tile buffers E / O, each holding the A tile (256 rows x
// 128 B) and the W tile (256 rows x 128 B).  Tiles are written by the direct-to-LDS DMA
// (global_load_lds, 1 KiB = 8 rows per wave instruction, 4 instructions per wave and tile) and are
// swizzled on the SOURCE side exactly as in gemm.hip: lane l fetches logical 16-B piece
// (l&7) ^ ((row>>1)&7) of its row, fragment reads apply the same involution (bank-conflict free).
//
// Phase p = { ds_read the operand sub-tile the NEXT phase needs; issue one tile of DMA; barrier;
//             16 MFMAs at raised priority; barrier }.  The two M wave groups run one barrier apart, so
// one group's LDS reads / DMA issue overlap the other group's MFMAs on the same SIMDs, and every
// fragment read has a whole phase to land before its MFMAs issue (counted lgkmcnt, placed by the compiler).
// Register roles: A0 / A1 hold the A sub-tiles (rows 0-63 / 64-127 of the wave block), X / Y the W sub-tiles;
// X and Y swap roles every K tile so that the first quadrant of the next tile can be fetched one phase early.
//
//   phase  MFMA quadrant  operands   reads issued (for the next phase)        DMA issued           counted wait
//   1      (0,0)          A0, X      Y  <- E.W sub 1
//   2      (0,1)          A0, Y      A1 <- E.A sub 1
//   3      (1,1)          A1, Y      - (balanced schedule: first half of A0)  E.W <- tile kt+2     vmcnt(4): O complete
//   4      (1,0)          A1, X      A0 <- O.A sub 0, Y <- O.W sub 0          E.A <- tile kt+2
//   5      (0,0)          A0, Y      X  <- O.W sub 1
//   6      (0,1)          A0, X      A1 <- O.A sub 1
//   7      (1,1)          A1, X      - (balanced schedule: first half of A0)  O.W <- tile kt+3     vmcnt(4): E complete
//   8      (1,0)          A1, Y      A0 <- E.A sub 0, X <- E.W sub 0          O.A <- tile kt+3
//
// (DMA columns of the table: the schedule of the implicit-GEMM convolutions and of the fp8 variant.  The fp16 linear / qkv launches issue the same sixteen pieces
//  per wave and iteration TWO per phase, into the sub-tile that became free two phases earlier — table and measurements at the K loop, `BAL`.)
// Hazard rules the table obeys (barrier epochs, with the half-phase stagger between the wave groups):
//   WAR  a tile is re-staged >= 2 phases after its last ds_read (E.W: 1 -> 3, E.A: 2 -> 4, O.W: 5 -> 7, O.A: 6 -> 8);
//   RAW  a staged tile is read >= 1 phase after the counted vmcnt that retires it (O: wait in 3, read from 4;
//        E: wait in 7, read from 8).  vmcnt never drops to 0 inside the loop: one tile of DMA (4 loads per
//        lane) is always left in flight across the barriers.
// (Measured: mixing ordinary VGPR loads, e.g. an L2 prefetch touch, into the same vmcnt stream breaks the counted
//  waits — LDS-DMA loads and VGPR loads do not retire in order with respect to each other.  Giving the DMA to one
//  wave group and L2-prefetch touches to the other is correct but 4-8 % SLOWER than no prefetch: the stalls at the
//  counted waits are not HBM misses that a touch could turn into L2 hits.)
//
// Phase-shifted tile walk (OUT_LINEAR, p.park != null).  One persistent workgroup per CU and equal tiles mean that all 256 CUs reach
// their epilogues in the same instant: the stores (and fc2's fp32 residual reads) of a whole round of tiles hit HBM as one burst while
// every matrix pipe idles, then HBM idles while everybody computes.  Measured with a start-delay experiment (profiles/r04_a_*): spreading
// the workgroups' epilogues over a tile period is worth 6-7 % on fc2 (512 KB of epilogue traffic per tile), 4.5 % on proj, nothing on
// fc1 (its epilogue is GELU-VALU bound) — but a start delay pays a tile of idle tail.  Instead the workgroups of XCD x start INSIDE their
// first tile, at K tile phi = x * nk / 8 (even): they accumulate K tiles [phi, nk) of that tile, PARK the fp32 accumulators in a private
// scratch (256 KB per workgroup, written and later read by the same lanes — no cross-workgroup traffic, no flags), walk their
// remaining tiles as usual, and finish with K tiles [0, phi) of the first tile on top of the parked partial sums.  Every XCD's tile
// boundaries are thereby shifted by x / 8 of a tile period; the total number of K tiles per workgroup is unchanged, nobody idles.
// The workgroups of one XCD stay in step with each other (they share A / W panels through that XCD's L2).
//
// F8 = 1: the same schedule on OCP MX-fp8 operands (e4m3 elements, one E8M0 scale per 32 K elements: BASELINE.json
// configs[4]).  A tile row is still 128 bytes = 128 K elements = ONE v_mfma_scale_f32_16x16x128_f8f6f4 step (8 passes, twice
// the FLOPs per cycle of the fp16 instruction), so the LDS image, the DMA, the number of fragment reads and the phase
// table carry over unchanged; what differs:
//   * the K order of the instruction (measured, tools/probes/probe_mfma_scale.hip): lane (g, li) supplies row li's bytes
//     K[16g, 16g+16) in its first four VGPRs and K[64+16g, 64+16g+16) in the last four — the SAME two 16-byte pieces (g, g + 4)
//     the fp16 path reads for its two k steps, so fragment addresses and the XOR swizzle are shared — while the E8M0 scale of
//     K block j = K[32j, 32j+32) is taken from the scale VGPR of the lanes with g = j (byte chosen by op_sel);
//   * the scales of a K tile (256 rows x 4 blocks = 1 KiB per operand) ride in with the tile's A stage as ONE extra
//     global_load_lds_dword per wave and are stored in global memory already in the order the fragment reads want them:
//     ONE ds_read_b32 hands a lane the scales of the four fragments of a sub-tile, selected per MFMA by op_sel
//     (layouts: mx8_scale_off_a / mx8_scale_off_w in gemm.h, written by the producers / the weight packer).
#include <map>
#include <mutex>
#include <utility>

#include "gemm.h"
#include "gemm_epilogue.h"
#include "gemm8_epi.h"

namespace cva {

namespace {

using namespace epi;
using namespace g8;


#define G8_BAR()                                   \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        __builtin_amdgcn_s_barrier();              \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)

#define G8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// The lane id recomputed from the execution mask (two instructions, NO live register) and opaque to the optimiser: everything a tile boundary
// derives from the lane (DMA row offsets, bias offsets, epilogue addresses) is loop invariant, so the compiler hoists it out of the persistent
// tile loop, keeps it live across the K loop — which needs all 256 VGPRs — and spills it; every reload at a tile boundary is a scratch load +
// s_waitcnt vmcnt(0), i.e. a full round trip behind the next tile's prologue DMA or the epilogue's stores.  With this (and the opaque divisors
// of gemm8_epi.h) every instantiation has ScratchSize 0: epilogue 6.6 -> 6.2 us per tile, fc1 -1.6 %, qkv -1.0 % (profiles/r04_m_gemm_timeline.txt,
// r04_no_gemm8_spills_lean_epilogue_ab.txt).
__device__ __forceinline__ int g8_fresh_lane() {
    int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
}
#ifdef CVA_ABLATION      // experiment: per-tile timeline of wave 0 / wave 4 of the first 8 workgroups (CVA_GEMM_DBG & 32768), written to p.park
#define G8_STAMP(slot)                                                                                                  \
    do { if ((p.dbg & 32768) && !(p.dbg & 262144) && p.park && blockIdx.x < 8 && (wave & 3) == 0 && lane == 0 && item < 24) \
             reinterpret_cast<long long*>(p.park)[((blockIdx.x * 2 + (wave >> 2)) * 24 + item) * 8 + (slot)] = (long long)wall_clock64(); } while (0)
// CVA_GEMM_DBG & 262144 (with 32768): instead of the tile timeline, the shader clock (s_memtime, core cycles) at the top of each of the eight phases of
// the K loop's third iteration — how long each phase of the schedule really takes
#define G8_PSTAMP(slot)                                                                                                 \
    do { if ((p.dbg & 262144) && kt == 4 && p.park && blockIdx.x < 8 && (wave & 3) == 0 && lane == 0 && item < 24)       \
             reinterpret_cast<long long*>(p.park)[((blockIdx.x * 2 + (wave >> 2)) * 24 + item) * 8 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define G8_STAMP(slot) do { } while (0)
#define G8_PSTAMP(slot) do { } while (0)
#endif

// 16 MFMAs of quadrant (MH, NH): C[MH*4+mi][NH*2+nj] += A[mi][ks] * W[nj][ks]
#define G8_MMQ(n, a, b, MH, NH)                                                                             \
    do {                                                                                                  \
        G8_WAIT(n, a, b);                                                                                 \
        __builtin_amdgcn_s_setprio(1);                                                                    \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                  \
            _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                              \
                _Pragma("unroll") for (int nj = 0; nj < 2; ++nj)                                          \
                    acc[(MH) * 4 + mi][(NH) * 2 + nj] =                                                   \
                        TRANS ? __builtin_amdgcn_mfma_f32_16x16x32_f16(b[nj][ks], a[mi][ks],              \
                                                                       acc[(MH) * 4 + mi][(NH) * 2 + nj], 0, 0, 0) \
                              : __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi][ks], b[nj][ks],              \
                                                                       acc[(MH) * 4 + mi][(NH) * 2 + nj], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                    \
    } while (0)

// F8: the 8 MFMAs of quadrant (MH, NH) of one K tile: C^T[nj][mi] += W[nj] . A[mi]^T with the block scales of both operands.
// a / b: per-fragment 32-byte operands (8 VGPRs, must be one register tuple — which is why the fp8 fragments are ordinary
// compiler-visible LDS loads and not inline-asm reads: two asm-defined 16-byte halves would have to be copied together);
// sa / sb: one dword holding the four fragment scales, selected by op_sel.
#define G8_MM8(a, b, sa, sb, MH, NH, mi, nj)                                                                     \
    acc[(MH) * 4 + (mi)][(NH) * 2 + (nj)] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                    \
        b[nj], a[mi], acc[(MH) * 4 + (mi)][(NH) * 2 + (nj)], 0, 0, (NH) * 2 + (nj), sb, (mi), sa)
// (the two empty asm statements pin the MFMAs between this phase's barriers: register-only instructions are otherwise free
//  to move across "memory"-clobbering asm and sched_barrier, and the compiler was seen to collect a whole iteration's
//  MFMAs behind all of its loads — every fragment live at once, 240 spilled VGPRs)
#define G8_MMQ8(a, b, sa, sb, MH, NH)                                                                            \
    do {                                                                                                         \
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(sa), "+v"(sb)); \
        __builtin_amdgcn_s_setprio(1);                                                                           \
        G8_MM8(a, b, sa, sb, MH, NH, 0, 0); G8_MM8(a, b, sa, sb, MH, NH, 0, 1);                                  \
        G8_MM8(a, b, sa, sb, MH, NH, 1, 0); G8_MM8(a, b, sa, sb, MH, NH, 1, 1);                                  \
        G8_MM8(a, b, sa, sb, MH, NH, 2, 0); G8_MM8(a, b, sa, sb, MH, NH, 2, 1);                                  \
        G8_MM8(a, b, sa, sb, MH, NH, 3, 0); G8_MM8(a, b, sa, sb, MH, NH, 3, 1);                                  \
        __builtin_amdgcn_s_setprio(0);                                                                           \
        asm volatile("" : "+v"(acc[(MH) * 4 + 0][(NH) * 2]), "+v"(acc[(MH) * 4 + 0][(NH) * 2 + 1]),               \
                          "+v"(acc[(MH) * 4 + 1][(NH) * 2]), "+v"(acc[(MH) * 4 + 1][(NH) * 2 + 1]),               \
                          "+v"(acc[(MH) * 4 + 2][(NH) * 2]), "+v"(acc[(MH) * 4 + 2][(NH) * 2 + 1]),               \
                          "+v"(acc[(MH) * 4 + 3][(NH) * 2]), "+v"(acc[(MH) * 4 + 3][(NH) * 2 + 1]));              \
    } while (0)
// The fp8 path's barrier is inline asm: the compiler then does not treat it as a point where its own (tracked) LDS loads
// must have returned, and places a counted s_waitcnt lgkmcnt(n) in front of the first MFMA that consumes a fragment instead
// — the next phase's reads stay in flight across the barrier, as the hand-counted waits of the fp16 path arrange.
#define G8_BAR8()                                  \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        asm volatile("s_barrier" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)

template <int OMODE, int TRANS, int ABL, int F8, int CV3>
__global__ __launch_bounds__(G8_NT) void gemm8_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    static_assert(!F8 || (TRANS == 1 && ABL == 0), "the fp8 variant exists for the direct (transposed) epilogues only");
    static_assert(!CV3 || ((OMODE == OUT_LINEAR || OMODE == OUT_CONVT) && TRANS == 1 && !F8), "implicit 3x3 convolution: fp16, direct epilogues");
    constexpr bool COMP = CV3 && OMODE == OUT_CONVT;    // ConvTranspose2d k2 s2 o Conv2d 3x3 composed (see launch_gemm8_deconv)
    constexpr int ESZ = F8 ? 1 : 2;                 // bytes per operand element; a tile row is 128 bytes either way
#ifndef CVA_G8_BAL_QKV
#define CVA_G8_BAL_QKV 1
#endif
    // balanced DMA schedule of the K loop (two pieces per wave and phase, see the loop): the fp16 linear / qkv launches.  The implicit-GEMM convolutions keep
    // four pieces in phases 3, 4, 7, 8: their A stage carries a descriptor / tap set-up per call, which the split doubles (measured +2 %).
#ifndef CVA_G8_BAL_F8
#define CVA_G8_BAL_F8 1
#endif
    constexpr bool BAL = F8 ? (CVA_G8_BAL_F8 != 0) : (!CV3 && (OMODE != OUT_QKV || CVA_G8_BAL_QKV));
    constexpr int KTE = 128 / ESZ;                  // K elements per tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_n = p.N / G8_BN, tiles_m = p.M / G8_BM, ntiles = tiles_m * tiles_n;
    const int nk = p.K / KTE;                       // even, >= 2 (host)

    // ---- per-tile state.  (m0, n0, swap) of the tile being computed / stored; the DMA row offsets and bases below
    // always describe the tile whose DMA is issued NEXT (they are advanced to tile t+1 before tile t's epilogue).
    const int lrow = lane >> 3, lpc = lane & 7;
    // Lane offsets of the DMA are 32-bit and relative to the TILE's first row (64-bit wave-uniform base per tile), so the
    // operands themselves may be larger than 2 GiB (hidden activations of > 48 tiles at 1024^2).
    auto a_row = [&](int m) -> long {     // physical activation row of logical row m
        long r = m;
        if (p.a_rpi > 0) r = (long)m + (long)(m / p.a_rpi) * p.a_extra + p.a_off;
        return r;
    };
    unsigned a_voff[4], w_voff[4];
    const unsigned char* Ab;
    const unsigned char* Wb;
    // CV3 (implicit 3x3 convolution over NHWC pixels, see the header): tap-validity bits of the lane's four rows (9 bits each,
    // two rows per register) and the tile's two buffer bases (source 1 / source 2 of a channel concat), both pointing at the
    // top-left tap of the tile's first pixel so that every tap offset is non-negative.
    unsigned cmask[2] = {0u, 0u};
    unsigned long long cbase1 = 0, cbase2 = 0;
    // COMP: output parity of the tile's columns, the tile's first pixel inside its image, and per lane row four border flags
    // (bit 0: y == 0, 1: y == H-1, 2: x == 0, 3: x == W-1; row i in bits 4i..4i+3) from which every tap's validity follows
    int comp_py = 0, comp_px = 0, comp_pim = 0;
    unsigned eflags = 0u;
    // F8: this wave's 256-byte share of the tile's scale block pair.  Waves 0-3 fetch the A-side block (scales of the
    // operand that sits in the "A" LDS tile), waves 4-7 the W-side block; block (row tile, K tile) is 1 KiB.
    const unsigned char* Sb = nullptr;
    const unsigned sc_voff = (unsigned)((wave & 3) * 256 + lane * 4);
    // K direction of the tile whose DMA is issued next: a workgroup's consecutive tiles share their A panel (same tile
    // row, next columns), so every other tile walks K BACKWARDS — the panel's most recently streamed K slices are
    // still in L2 when the next tile starts from that end.
    int kstart = 0, kstep = 1;
    // Phase-shifted walk (see the header): the workgroup's tiles as a sequence of ITEMS.  phi == 0: item i = tile i, whole.
    // phi > 0: item 0 = K tiles [phi, nk) of tile 0 (parked), items 1 .. ntl-1 = tiles 1 .. ntl-1, item ntl = K tiles [0, phi) of tile 0.
    constexpr bool PSHIFT = OMODE == OUT_LINEAR && TRANS == 1 && !F8 && !CV3 && ABL == 0;
    const int ntl = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    int phi = 0;
    if (PSHIFT && p.park && ntl >= 2 && !(p.dbg & 32768)) phi = ((((int)blockIdx.x & 7) * nk) >> 3) & ~1;
    const int nitems = ntl + (phi ? 1 : 0);
    int kcnt_next = nk;                 // K tiles of the item whose DMA is issued next (even, >= 2)
    // Which 8-row pieces of a K tile this wave stages (four of the A tile, four of the W tile).  fp16: two pieces of EACH sub-tile the fragment reads
    // distinguish — A sub-tile MH = rows wr*128 + MH*64 .. +64 of both wave rows, W sub-tile NH = rows wc*64 + NH*32 .. +32 of the four wave columns —
    // i = 0, 1 in sub-tile 0, i = 2, 3 in sub-tile 1, so that every wave can issue two pieces of whichever sub-tile has just become free (the
    // balanced DMA schedule of the K loop, BAL).  The other kernels keep 32 contiguous rows per wave.
    auto a_piece_row = [&](int i) -> int {
        if (!BAL) return wave * 32 + i * 8;
        const int q = 2 * wave + (i & 1);
        return (q >> 3) * 128 + (q & 7) * 8 + (i >> 1) * 64;
    };
    auto w_piece_row = [&](int i) -> int {
        if (!BAL) return wave * 32 + i * 8;
        const int q = 2 * wave + (i & 1);
        return (q >> 2) * 64 + (q & 3) * 8 + (i >> 1) * 32;
    };
    auto tile_setup = [&](int item, int& m0, int& n0, bool& swap) {
        const int fl = g8_fresh_lane();
        const int lrow = fl >> 3, lpc = fl & 7;       // (shadow the kernel-scope copies: recomputed per tile, see g8_fresh_lane)
        int tm, tn;
        const int tile = (int)blockIdx.x + ((phi && item == ntl) ? 0 : item) * (int)gridDim.x;
        tile_coords(xcd_remap(tile, ntiles), tiles_m, tiles_n, tm, tn);
        m0 = tm * G8_BM; n0 = tn * G8_BN;
        const bool rev = (item & 1) && !(p.dbg & 128);
        kstart = rev ? nk - 1 : 0; kstep = rev ? -1 : 1; kcnt_next = nk;
        if (phi && item == 0) { kstart = phi; kstep = 1; kcnt_next = nk - phi; }
        if (phi && item == ntl) { kstart = 0; kstep = 1; kcnt_next = phi; }
        // v columns of the fused qkv projection: exchange the operands (see epilogue8_vt); block-uniform
        swap = OMODE == OUT_QKV && TRANS == 1 && !p.v_rm && (n0 + p.n_off) >= 2 * p.D;
        if (COMP) {
            const int pim = m0 & (p.H * p.Wd - 1);
            const int par = n0 / (p.N >> 2);
            comp_py = par >> 1; comp_px = par & 1; comp_pim = pim;
            eflags = 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = a_piece_row(i) + lrow, wrow = w_piece_row(i) + lrow;         // LDS rows of the A / W tile this lane fills
                const int lp = lpc ^ ((row >> 1) & 7), wlp = lpc ^ ((wrow >> 1) & 7);
                const int prow = (wrow & ~63) | (((wrow >> 2) & 3) << 4) | (((wrow >> 4) & 3) << 2) | (wrow & 3);
                const int pp = pim + row, y = pp >> p.conv_wshift, x = pp & (p.Wd - 1);
                eflags |= ((y == 0 ? 1u : 0u) | (y == p.H - 1 ? 2u : 0u) | (x == 0 ? 4u : 0u) | (x == p.Wd - 1 ? 8u : 0u)) << (4 * i);
                a_voff[i] = (unsigned)(row * p.C1 * 2) + lp * 16;
                w_voff[i] = (unsigned)((long)prow * p.ldw * 2) + wlp * 16;
            }
            // input pixels: base at the top-left 3x3 tap of the tile's first pixel.  Second source (the skip connection at the
            // OUTPUT resolution, 2H x 2W): input pixel m = (b*H + y)*W + x maps to output pixel index 4*m - 2*x of parity (0, 0);
            // base at that pixel's top-left tap for the tile's first pixel.
            const long origin = ((long)m0 - p.Wd - 1) * p.C1 * 2;
            cbase1 = (unsigned long long)(reinterpret_cast<const unsigned char*>(p.A) + origin);
            const long origin2 = (4L * m0 - 2L * (pim & (p.Wd - 1)) - 2L * p.Wd - 1) * p.C2 * 2;
            cbase2 = p.A2 ? (unsigned long long)(reinterpret_cast<const unsigned char*>(p.A2) + origin2) : cbase1;
            Ab = nullptr;
            Wb = reinterpret_cast<const unsigned char*>(p.W) + (long)n0 * p.ldw * 2;
            return;
        }
        if (CV3) {
            const int pim = m0 & (p.H * p.Wd - 1);          // first pixel of the tile inside its image (H*W a power of two)
            cmask[0] = cmask[1] = 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = a_piece_row(i) + lrow, wrow = w_piece_row(i) + lrow;
                const int lp = lpc ^ ((row >> 1) & 7), wlp = lpc ^ ((wrow >> 1) & 7);
                const int prow = (wrow & ~63) | (((wrow >> 2) & 3) << 4) | (((wrow >> 4) & 3) << 2) | (wrow & 3);
                const int pp = pim + row, y = pp >> p.conv_wshift, x = pp & (p.Wd - 1);
                // tap t = ky*3 + kx reads pixel (y + ky - 1, x + kx - 1): bit t set = inside the image
                const unsigned mk = (y > 0 ? 0x1FFu : 0x1F8u) & (y < p.H - 1 ? 0x1FFu : 0x03Fu) & (x > 0 ? 0x1FFu : 0x1B6u) &
                                    (x < p.Wd - 1 ? 0x1FFu : 0x0DBu);
                cmask[i >> 1] |= mk << ((i & 1) * 9);
                a_voff[i] = (unsigned)(row * p.C1 * 2) + lp * 16;
                w_voff[i] = (unsigned)((long)prow * p.ldw * 2) + wlp * 16;
            }
            const long origin = ((long)m0 - p.Wd - 1) * p.C1 * 2;          // may lie before the tensor: only masked lanes would touch it
            cbase1 = (unsigned long long)(reinterpret_cast<const unsigned char*>(p.A) + origin);
            cbase2 = p.A2 ? (unsigned long long)(reinterpret_cast<const unsigned char*>(p.A2) + origin) : cbase1;
            Ab = nullptr;
            Wb = reinterpret_cast<const unsigned char*>(p.W) + (long)n0 * p.ldw * 2;
            return;
        }
        // (opaque: the 64-bit row pitches are loop invariant and would otherwise be kept across the K loop — in VGPRs, the SGPR file being
        //  full — and spilled, like the lane id)
        const long lda_b = (long)g8::opaque_s(p.lda) * ESZ, ldw_b = (long)g8::opaque_s(p.ldw) * ESZ;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = a_piece_row(i) + lrow, wrow = w_piece_row(i) + lrow;             // LDS rows of the A / W tile this lane fills
            const int lp = lpc ^ ((row >> 1) & 7), wlp = lpc ^ ((wrow >> 1) & 7);
            // TRANS: LDS row wc*64 + j*16 + g*4 + r of the "W" tile holds source row wc*64 + g*16 + j*4 + r, so that a
            // lane's 16 accumulator values per output row are 16 consecutive columns (see epilogue8_direct)
            const int prow = TRANS == 1 ? ((wrow & ~63) | (((wrow >> 2) & 3) << 4) | (((wrow >> 4) & 3) << 2) | (wrow & 3)) : wrow;
            const long ar0 = a_row(m0);
            if (!swap) {
                a_voff[i] = (unsigned)((a_row(m0 + row) - ar0) * lda_b) + lp * 16;
                w_voff[i] = (unsigned)((long)prow * ldw_b) + wlp * 16;
            } else {
                a_voff[i] = (unsigned)((long)row * ldw_b) + lp * 16;
                w_voff[i] = (unsigned)((a_row(m0 + prow) - ar0) * lda_b) + wlp * 16;
            }
        }
        {
            const unsigned char* abase = reinterpret_cast<const unsigned char*>(p.A) + a_row(m0) * lda_b;
            const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.W) + (long)n0 * ldw_b;
            Ab = swap ? wbase : abase;
            Wb = swap ? abase : wbase;
        }
        if (F8) {
            // scale images: activations  A-side layout p.a_scale (p.a_scale_w = W-side layout, for the swapped V tiles);
            //               weights      p.w_scale, packed per 256-row tile in the layout of the side the tile runs on
            const bool a_side = wave < 4;
            const unsigned char* act = reinterpret_cast<const unsigned char*>(a_side ? p.a_scale : p.a_scale_w);   // image of the side they land on
            const unsigned char* wgt = reinterpret_cast<const unsigned char*>(p.w_scale);
            const bool take_act = a_side == !swap;          // A side holds activations unless swapped
            Sb = take_act ? act + (long)(m0 / G8_BM) * nk * 1024 : wgt + (long)(n0 / G8_BN) * nk * 1024;
        }
    };

    // Direct-to-LDS DMA, written as inline asm to pin the `v_off, s[base:base+1]` addressing form (the builtin lets the
    // optimiser keep per-lane 64-bit pointers in VGPRs: 16 registers this kernel does not have).  M0 = LDS byte address
    // of the 1-KiB destination (wave-uniform); no other code in this kernel depends on M0.
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem8;
#define G8_DMA(voff, base, ldsaddr)                                                                         \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), \
                 "s"(ldsaddr) : "memory")
#define G8_DMA4(voff, base, ldsaddr)                                                                        \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(base), \
                 "s"(ldsaddr) : "memory")
    // CV3: the A tile of K step k = (tap, 64-channel chunk) is 256 pixels x 64 channels of the tap-shifted image — one
    // buffer-addressed LDS-DMA per 8 rows whose per-lane offset is the pixel's own row offset, the tap / chunk displacement is
    // the instruction's scalar offset, and lanes whose tap falls outside the image carry an offset beyond the descriptor's
    // range: the hardware then writes ZEROS into their LDS slots (measured, tools/probes/probe_buffer_lds_oob.hip) — the
    // convolution's zero padding costs one select per piece and no memory traffic.
#define G8_BDMA(voff, desc, soff, ldsaddr)                                                                   \
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(desc), \
                 "s"(soff), "s"(ldsaddr) : "memory")
    auto stage_a_conv = [&](int buf, int kt, int i0, int i1) {
        const int kidx = kstart + kt * kstep;
        if (COMP) {
            // K steps [0, C1/16): the composed part, K order (64-channel chunk, 2x2 input pixel) — output parity (py, px) sees input
            // pixels (y + py - 1 + s, x + px - 1 + t) = 3x3 tap (py + s, px + t) of the input image.  K steps beyond: the 3x3 taps of
            // the second source at the output resolution, K order (64-channel chunk, tap); output pixel (2y + py, 2x + px).
            const int nkc = p.C1 >> 4;
            unsigned sm, soff; unsigned long long b; bool first;
            if (kidx < nkc) {
                const int t4 = kidx & 3, ty = comp_py + (t4 >> 1), tx = comp_px + (t4 & 1);
                sm = (ty == 0 ? 1u : 0u) | (ty == 2 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == 2 ? 8u : 0u);
                soff = (unsigned)(((ty * p.Wd + tx) * p.C1 + (kidx >> 2) * G8_BK) * 2);
                b = cbase1; first = true;
            } else {
                const int k2 = kidx - nkc, chunk = (k2 * 7282) >> 16, tap = k2 - chunk * 9;      // k2 / 9, exact for k2 < 4096
                const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
                sm = (ky == 0 && comp_py == 0 ? 1u : 0u) | (ky == 2 && comp_py == 1 ? 2u : 0u) |
                     (kx == 0 && comp_px == 0 ? 4u : 0u) | (kx == 2 && comp_px == 1 ? 8u : 0u);
                soff = (unsigned)((((comp_py + ky) * 2 * p.Wd + comp_px + kx) * p.C2 + chunk * G8_BK) * 2);
                b = cbase2; first = false;
            }
            sm = __builtin_amdgcn_readfirstlane(sm); soff = __builtin_amdgcn_readfirstlane(soff);
            i32x4_t d;
            d[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
            d[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));
            d[2] = 0x40000000;
            d[3] = 0x00020000;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + buf * G8_TILE + wave * 4096);
            const int x0 = comp_pim & (p.Wd - 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < i0 || i >= i1) continue;
                unsigned vo = a_voff[i];
                if (!first) {         // block-uniform: the lane's pixel at the output resolution, relative to the tile's first
                    int row = a_piece_row(i) + lrow;
                    asm volatile("" : "+v"(row));      // recomputed per K step (a few VALU ops): hoisted, the four offsets cost registers this kernel does not have
                    const int x = (comp_pim + row) & (p.Wd - 1);
                    vo = (unsigned)((4 * row - 2 * (x - x0)) * p.C2 * 2) + ((lpc ^ ((row >> 1) & 7)) << 4);
                }
                if ((eflags >> (4 * i)) & sm) vo = 0x80000000u;        // tap outside the image: the DMA writes zeros
                G8_BDMA(vo, d, soff, dst + i * 1024);
            }
            return;
        }
        int tap, ch;
        if (p.conv_kmajor) {          // K order (64-channel chunk, tap): the nine taps of a chunk re-read the same L2 lines back to back
            const int chunk = (kidx * 7282) >> 16;        // kidx / 9, exact for kidx < 4096
            tap = __builtin_amdgcn_readfirstlane(kidx - chunk * 9);
            ch = chunk * G8_BK;
        } else {                      // K order (tap, channel): the packed filter layout of pack_conv3
            tap = __builtin_amdgcn_readfirstlane(kidx >> p.conv_cshift);
            ch = (kidx & ((1 << p.conv_cshift) - 1)) * G8_BK;
        }
        const bool second = ch >= p.C1;                               // block-uniform
        const int ty = (tap * 11) >> 5, tx = tap - ty * 3;
        const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(((ty * p.Wd + tx) * p.C1 + (second ? ch - p.C1 : ch)) * 2));
        const unsigned long long b = second ? cbase2 : cbase1;
        i32x4_t d;
        d[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        d[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));
        d[2] = 0x40000000;                                            // num_records: valid offsets are tile-relative (< 2^30)
        d[3] = 0x00020000;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + buf * G8_TILE + wave * 4096);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < i0 || i >= i1) continue;
            const unsigned ok = (cmask[i >> 1] >> ((i & 1) * 9 + tap)) & 1u;
            const unsigned vo = ok ? a_voff[i] : 0x80000000u;
            G8_BDMA(vo, d, soff, dst + i * 1024);
        }
    };
    auto stage_a = [&](int buf, int kt, int i0 = 0, int i1 = 4) {      // pieces [i0, i1) of the wave's four (see a_piece_row)
        if (CV3) { stage_a_conv(buf, kt, i0, i1); return; }
        if (F8 && !BAL) {      // the K tile's scale blocks first (oldest load of the stage: every counted wait that covers the tile covers them)
            const unsigned char* sbase = uniform_ptr(Sb + (long)(kstart + kt * kstep) * 1024);
            const unsigned sdst = __builtin_amdgcn_readfirstlane(lds0 + G8_SC + buf * 2048 + wave * 256);
            G8_DMA4(sc_voff, sbase, sdst);
        }
        const unsigned char* base = uniform_ptr(Ab + (long)(kstart + kt * kstep) * (G8_BK * 2));
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + buf * G8_TILE + a_piece_row(0) * 128);      // pieces 1, 2, 3 at fixed distances
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i >= i0 && i < i1) G8_DMA(a_voff[i], base, dst + (i & 1) * 1024 + (i >> 1) * (BAL ? 8192 : 2048));
    };
    auto stage_sc = [&](int buf, int kt) {         // fp8, balanced schedule: the K tile's scale blocks as a piece of their own
        const unsigned char* sbase = uniform_ptr(Sb + (long)(kstart + kt * kstep) * 1024);
        const unsigned sdst = __builtin_amdgcn_readfirstlane(lds0 + G8_SC + buf * 2048 + wave * 256);
        G8_DMA4(sc_voff, sbase, sdst);
    };
    auto stage_w = [&](int buf, int kt, int i0 = 0, int i1 = 4) {
        const unsigned char* base = uniform_ptr(Wb + (long)(kstart + kt * kstep) * (G8_BK * 2));
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + G8_WOFF + buf * G8_TILE + w_piece_row(0) * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i >= i0 && i < i1) G8_DMA(w_voff[i], base, dst + (i & 1) * 1024 + (i >> 1) * (BAL ? 4096 : 2048));
    };

    // ---- fragment read offsets: row r (r & 15 == lane & 15), logical piece ks*4 + g -> byte r*128 + ((lp ^ ((r>>1)&7)) << 4)
    // (fp8: the same two pieces g and g + 4 form the lane's 32-byte operand of the K = 128 instruction, see the file header)
    const int g = lane >> 4, li = lane & 15;
    const int off0 = li * 128 + ((g ^ ((li >> 1) & 7)) << 4);
    const int d1 = 64 - 2 * (off0 & 64);                          // offset of the second piece (g + 4): off ^ 64
    const unsigned a_ad0 = lds0 + (wr * 128) * 128 + off0, a_ad1 = a_ad0 + d1;              // + buf*32K + mh*8K + mi*2K
    const unsigned w_ad0 = lds0 + G8_WOFF + (wc * 64) * 128 + off0, w_ad1 = w_ad0 + d1;     // + buf*32K + nh*4K + nj*2K

    // Fragment reads are raw ds_read_b128 (the compiler does not track them): every consumer is preceded by an
    // explicit counted s_waitcnt lgkmcnt(n) that names the fragments as operands (G8_WAIT_*).
    half8_t A0[4][2], A1[4][2], X[2][2], Y[2][2];
#define G8_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define G8_RD_A(a, BUF, MH)                                                                     \
    do {                                                                                        \
        G8_DSR(a[0][0], a_ad0, (BUF) * G8_TILE + (MH) * 8192 + 0 * 2048);                       \
        G8_DSR(a[0][1], a_ad1, (BUF) * G8_TILE + (MH) * 8192 + 0 * 2048);                       \
        G8_DSR(a[1][0], a_ad0, (BUF) * G8_TILE + (MH) * 8192 + 1 * 2048);                       \
        G8_DSR(a[1][1], a_ad1, (BUF) * G8_TILE + (MH) * 8192 + 1 * 2048);                       \
        G8_DSR(a[2][0], a_ad0, (BUF) * G8_TILE + (MH) * 8192 + 2 * 2048);                       \
        G8_DSR(a[2][1], a_ad1, (BUF) * G8_TILE + (MH) * 8192 + 2 * 2048);                       \
        G8_DSR(a[3][0], a_ad0, (BUF) * G8_TILE + (MH) * 8192 + 3 * 2048);                       \
        G8_DSR(a[3][1], a_ad1, (BUF) * G8_TILE + (MH) * 8192 + 3 * 2048);                       \
    } while (0)
// the two halves of G8_RD_A (fragment rows 0, 1 | 2, 3 of the sub-tile): the balanced schedule reads them in consecutive phases
#define G8_RD_A_LO(a, BUF, MH)                                                                  \
    do {                                                                                        \
        G8_DSR(a[0][0], a_ad0, (BUF) * G8_TILE + (MH) * 8192 + 0 * 2048);                       \
        G8_DSR(a[0][1], a_ad1, (BUF) * G8_TILE + (MH) * 8192 + 0 * 2048);                       \
        G8_DSR(a[1][0], a_ad0, (BUF) * G8_TILE + (MH) * 8192 + 1 * 2048);                       \
        G8_DSR(a[1][1], a_ad1, (BUF) * G8_TILE + (MH) * 8192 + 1 * 2048);                       \
    } while (0)
#define G8_RD_A_HI(a, BUF, MH)                                                                  \
    do {                                                                                        \
        G8_DSR(a[2][0], a_ad0, (BUF) * G8_TILE + (MH) * 8192 + 2 * 2048);                       \
        G8_DSR(a[2][1], a_ad1, (BUF) * G8_TILE + (MH) * 8192 + 2 * 2048);                       \
        G8_DSR(a[3][0], a_ad0, (BUF) * G8_TILE + (MH) * 8192 + 3 * 2048);                       \
        G8_DSR(a[3][1], a_ad1, (BUF) * G8_TILE + (MH) * 8192 + 3 * 2048);                       \
    } while (0)
#define G8_RD_W(b, BUF, NH)                                                                     \
    do {                                                                                        \
        G8_DSR(b[0][0], w_ad0, (BUF) * G8_TILE + (NH) * 4096 + 0 * 2048);                       \
        G8_DSR(b[0][1], w_ad1, (BUF) * G8_TILE + (NH) * 4096 + 0 * 2048);                       \
        G8_DSR(b[1][0], w_ad0, (BUF) * G8_TILE + (NH) * 4096 + 1 * 2048);                       \
        G8_DSR(b[1][1], w_ad1, (BUF) * G8_TILE + (NH) * 4096 + 1 * 2048);                       \
    } while (0)
// wait until at most n LDS reads are outstanding; the fragments about to be consumed are tied to the wait
#define G8_WAIT(n, a, b)                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                      \
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]),      \
                   "+v"(a[3][0]), "+v"(a[3][1]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1])       \
                 :: "memory")

    // ---- F8: fragments (32 bytes = pieces 2g, 2g+1 of a row) and scale dwords as compiler-visible LDS loads
    typedef const __attribute__((address_space(3))) unsigned char* lds_cp;
    const lds_cp lds_base = (lds_cp)(__attribute__((address_space(3))) void*)smem8;
    const lds_cp fa0 = lds_base + (a_ad0 - lds0), fw0 = lds_base + (w_ad0 - lds0);      // second half at + d1
    const lds_cp sa_p = lds_base + G8_SC + (wr * 2) * 256 + lane * 4;                    // + buf*2048 + mh*256
    const lds_cp sw_p = lds_base + G8_SC + 1024 + wc * 256 + lane * 4;                   // + buf*2048
    i32x8_t F0[4], F1[4], FX[2], FY[2];
    int sA0 = 0, sA1 = 0, sWE = 0, sWO = 0;
    auto ld32 = [&](lds_cp q) -> i32x8_t {
        const i32x4_t lo = *reinterpret_cast<const __attribute__((address_space(3))) i32x4_t*>(q);
        const i32x4_t hi = *reinterpret_cast<const __attribute__((address_space(3))) i32x4_t*>(q + d1);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
#define G8F_RD_A(f, sc, BUF, MH)                                                                               \
    do {                                                                                                       \
        _Pragma("unroll") for (int mi_ = 0; mi_ < 4; ++mi_) f[mi_] = ld32(fa0 + (BUF) * G8_TILE + (MH) * 8192 + mi_ * 2048); \
        sc = *reinterpret_cast<const __attribute__((address_space(3))) int*>(sa_p + (BUF) * 2048 + (MH) * 256); \
    } while (0)
#define G8F_RD_W(f, BUF, NH)                                                                                   \
    do { _Pragma("unroll") for (int nj_ = 0; nj_ < 2; ++nj_) f[nj_] = ld32(fw0 + (BUF) * G8_TILE + (NH) * 4096 + nj_ * 2048); } while (0)
#define G8F_RD_SW(sc, BUF) sc = *reinterpret_cast<const __attribute__((address_space(3))) int*>(sw_p + (BUF) * 2048)

    constexpr bool no_dma = ABL & 1, no_rd = ABL & 2, no_epi = ABL & 4;   // experiment instantiations (CVA_GEMM_DBG), ABL = 0 in production

    // E <- tile 0, O <- tile 1 of the K loop (16 DMA loads per lane); wave 0 first fetches the tile's 256 bias values
    // into LDS slot `slot` (the oldest load of the group, so the counted waits below cover it): the epilogue then needs
    // no VMEM load, which would otherwise have to wait for the whole in-order DMA queue.
    auto stage_prologue = [&](int n0_, bool swap_, int slot) {
        if (TRANS && p.bias && wave == 0) {
            (void)swap_;
            const unsigned char* src = uniform_ptr(reinterpret_cast<const unsigned char*>(p.bias + n0_));
            const unsigned boff = (unsigned)g8_fresh_lane() * 16u;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + G8_BIAS + slot * 1024);
            G8_DMA(boff, src, dst);
        }
        if (!BAL) {
            stage_a(0, 0);
            stage_w(0, 0);
            stage_w(1, 1);
            if (!F8) stage_a(1, 1);  // (fp8 schedule: O.A is staged in phase 1 of every iteration, the first included)
        } else if (F8) {             // fp8, balanced: E whole (scale blocks first), O's sub-tiles 0 — phases 1 / 2 of every iteration stage the rest of O
            stage_sc(0, 0); stage_a(0, 0); stage_w(0, 0);
            stage_a(1, 1, 0, 2); stage_w(1, 1, 0, 2);
        } else {                     // fp16: E whole, O without its A sub-tile 1 (phase 1 of every iteration stages that, the first included): 14 pieces
            stage_a(0, 0, 0, 2); stage_w(0, 0, 0, 2); stage_w(0, 0, 2, 4); stage_a(0, 0, 2, 4);
            stage_a(1, 1, 0, 2); stage_w(1, 1, 0, 2); stage_w(1, 1, 2, 4);
        }
    };

    // ---- persistent loop over output tiles: the DMA of tile t+1's first two K tiles is issued BEFORE tile t's
    // epilogue, so its latency (and the epilogue's store drain) overlap instead of adding up
    if (TRANS && (OMODE == OUT_LINEAR || OMODE == OUT_MX8) && p.act == ACT_GELU) {     // (read only in epilogues: many barriers later)
        gelu_fill_lut(reinterpret_cast<float*>(smem8 + G8_LUT), threadIdx.x, G8_NT);
    }
    int m0, n0; bool swap;
    int item = 0;
    int slot = 0;
#ifdef CVA_ABLATION
    if (p.stagger) {       // experiment: de-phase the workgroups' epilogues (stores of all CUs otherwise hit HBM in the same instant)
        const long spread = p.stagger > 0 ? p.stagger : -p.stagger;
        const long target = p.stagger > 0 ? spread * (blockIdx.x & 7) / 8 : spread * (long)(((blockIdx.x & 7) * 32 + (blockIdx.x >> 3)) & 255) / 256;
        const long t0 = (long)wall_clock64();
        while ((long)wall_clock64() - t0 < target) __builtin_amdgcn_s_sleep(16);
    }
#endif
    tile_setup(item, m0, n0, swap);
    stage_prologue(n0, swap, slot);
    for (; item < nitems; ++item, slot ^= 1) {
        const int nk_it = kcnt_next;                // K tiles of THIS item (tile_setup moves on to the next item before the epilogue)
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4)(0.f);

        if constexpr (F8) {
            // fp8 schedule: a fragment set is read at the START of the phase that first consumes it (three sets = 64 VGPRs
            // live instead of four: the 32-byte operands must be register tuples, so the reads are compiler-visible loads and
            // the budget is the allocator's); the exposed LDS latency hides behind the other wave group's MFMAs (the groups
            // run one barrier apart).  Reads one phase later -> every re-stage one phase later than in the fp16 table:
            //   phase  quadrant  reads at phase start                         DMA issued              counted wait
            //   1      (0,0)     A0 <- E.A s0 (+scales), X <- E.W s0 (+sc)    O.A <- tile kt+1
            //   2      (0,1)     Y  <- E.W s1
            //   3      (1,1)     A1 <- E.A s1 (+scales)
            //   4      (1,0)     -                                            E.W <- tile kt+2        vmcnt(4): O complete
            //   5      (0,0)     A0 <- O.A s0 (+scales), Y <- O.W s0 (+sc)    E.A <- tile kt+2
            //   6      (0,1)     X  <- O.W s1
            //   7      (1,1)     A1 <- O.A s1 (+scales)
            //   8      (1,0)     -                                            O.W <- tile kt+3        vmcnt(4): E complete
            //   (the prologue of an output tile stages E.A, E.W and O.W: phase 1 is the same in every iteration)
            // WAR (re-stage >= 2 phases after the last read): E.W 2 -> 4, E.A 3 -> 5, O.W 6 -> 8, O.A 7 -> 1'; the scale images
            // travel with the A stages (E: last read 3 -> 5, O: 7 -> 1').  RAW (read >= 1 phase after the retiring wait): O 4 -> 5, E 8 -> 1'.
            G8_VMCNT(4);                            // E (tile + its scale blocks) has landed; four pieces of O may still be in flight
            G8_BAR8();
            if (wr == 1) G8_BAR8();                 // stagger the second wave group by one barrier
            if constexpr (BAL) {
            // Balanced DMA (as in the fp16 loop below, shifted by the one phase the fp8 reads are later): two pieces per wave in every phase, the scale
            // blocks with the W sub-tile 1 pieces; waits = what has been issued since the piece that is needed:
            //   phase  issues                              waits for (read in)                      vmcnt
            //   1      O.W s1 + O scales <- kt+1 (3)       -
            //   2      O.A s1 <- kt+1 (2)                  E.A s1 (3)                               9
            //   3      E.A s0 <- kt+2 (2)                  -
            //   4      E.W s0 <- kt+2 (2)                  O.A s0, O.W s0, O scales (5)             6
            //   5      E.W s1 + E scales <- kt+2 (3)       -
            //   6      E.A s1 <- kt+2 (2)                  O.A s1 (7)                               9
            //   7      O.A s0 <- kt+3 (2)                  -
            //   8      O.W s0 <- kt+3 (2)                  E.A s0, E.W s0, E.W s1, E scales (1', 2') 6
            for (int kt = 0; kt < nk_it; kt += 2) {
                const bool more = kt + 2 < nk_it;   // block-uniform
                // ---- phase 1
                G8F_RD_A(F0, sA0, 0, 0); G8F_RD_W(FX, 0, 0); G8F_RD_SW(sWE, 0);
                stage_w(1, kt + 1, 2, 4); stage_sc(1, kt + 1);
                G8_BAR8(); G8_MMQ8(F0, FX, sA0, sWE, 0, 0); G8_BAR8();
                // ---- phase 2
                G8F_RD_W(FY, 0, 1);
                stage_a(1, kt + 1, 2, 4); G8_VMCNT(9);
                G8_BAR8(); G8_MMQ8(F0, FY, sA0, sWE, 0, 1); G8_BAR8();
                // ---- phase 3
                G8F_RD_A(F1, sA1, 0, 1);
                if (more) stage_a(0, kt + 2, 0, 2);
                G8_BAR8(); G8_MMQ8(F1, FY, sA1, sWE, 1, 1); G8_BAR8();
                // ---- phase 4
                if (more) { stage_w(0, kt + 2, 0, 2); G8_VMCNT(6); } else { G8_VMCNT(2); }
                G8_BAR8(); G8_MMQ8(F1, FX, sA1, sWE, 1, 0); G8_BAR8();
                // ---- phase 5
                G8F_RD_A(F0, sA0, 1, 0); G8F_RD_W(FY, 1, 0); G8F_RD_SW(sWO, 1);
                if (more) { stage_w(0, kt + 2, 2, 4); stage_sc(0, kt + 2); }
                G8_BAR8(); G8_MMQ8(F0, FY, sA0, sWO, 0, 0); G8_BAR8();
                // ---- phase 6
                G8F_RD_W(FX, 1, 1);
                if (more) { stage_a(0, kt + 2, 2, 4); G8_VMCNT(9); } else { G8_VMCNT(0); }
                G8_BAR8(); G8_MMQ8(F0, FX, sA0, sWO, 0, 1); G8_BAR8();
                // ---- phase 7
                G8F_RD_A(F1, sA1, 1, 1);
                if (more) stage_a(1, kt + 3, 0, 2);
                G8_BAR8(); G8_MMQ8(F1, FX, sA1, sWO, 1, 1); G8_BAR8();
                // ---- phase 8
                if (more) { stage_w(1, kt + 3, 0, 2); G8_VMCNT(6); }
                G8_BAR8(); G8_MMQ8(F1, FY, sA1, sWO, 1, 0); G8_BAR8();
            }
            } else {
            for (int kt = 0; kt < nk_it; kt += 2) {
                const bool more = kt + 2 < nk_it;   // block-uniform
                // ---- phase 1
                G8F_RD_A(F0, sA0, 0, 0); G8F_RD_W(FX, 0, 0); G8F_RD_SW(sWE, 0);
                stage_a(1, kt + 1);
                G8_BAR8(); G8_MMQ8(F0, FX, sA0, sWE, 0, 0); G8_BAR8();
                // ---- phase 2
                G8F_RD_W(FY, 0, 1);
                G8_BAR8(); G8_MMQ8(F0, FY, sA0, sWE, 0, 1); G8_BAR8();
                // ---- phase 3
                G8F_RD_A(F1, sA1, 0, 1);
                G8_BAR8(); G8_MMQ8(F1, FY, sA1, sWE, 1, 1); G8_BAR8();
                // ---- phase 4
                if (more) { stage_w(0, kt + 2); G8_VMCNT(4); } else { G8_VMCNT(0); }
                G8_BAR8(); G8_MMQ8(F1, FX, sA1, sWE, 1, 0); G8_BAR8();
                // ---- phase 5
                G8F_RD_A(F0, sA0, 1, 0); G8F_RD_W(FY, 1, 0); G8F_RD_SW(sWO, 1);
                if (more) stage_a(0, kt + 2);
                G8_BAR8(); G8_MMQ8(F0, FY, sA0, sWO, 0, 0); G8_BAR8();
                // ---- phase 6
                G8F_RD_W(FX, 1, 1);
                G8_BAR8(); G8_MMQ8(F0, FX, sA0, sWO, 0, 1); G8_BAR8();
                // ---- phase 7
                G8F_RD_A(F1, sA1, 1, 1);
                G8_BAR8(); G8_MMQ8(F1, FX, sA1, sWO, 1, 1); G8_BAR8();
                // ---- phase 8
                if (more) { stage_w(1, kt + 3); G8_VMCNT(4); }
                G8_BAR8(); G8_MMQ8(F1, FY, sA1, sWO, 1, 0); G8_BAR8();
            }
            }
        } else {
        if constexpr (BAL) {
        G8_STAMP(0);
        G8_VMCNT(6);                                // E has landed (O's six pieces may still be in flight); older epilogue stores have drained
        G8_STAMP(1);
        G8_BAR();
        if (!no_rd) { G8_RD_A(A0, 0, 0); G8_RD_W(X, 0, 0); }
        if (wr == 1) G8_BAR();                      // stagger the second wave group by one barrier

        // DMA schedule (fp16 linear / qkv): the sixteen pieces of a wave and iteration spread over ALL phases, 2 / 2 / 3 / 1 per half iteration (the third piece of phases
        // 3 / 7, which read nothing, is taken from phases 4 / 8, which read twelve fragments), each into the sub-tile that became free two phases earlier, each
        // waited for one phase before its first read — the wait counts the pieces issued since ("what the last four phases issued may be in flight"):
        //   phase  issues                         (free since)  waits for (read in)     vmcnt |  phase  issues                         waits for (read in)      vmcnt
        //   1      O.A sub 1 <- kt+1 (2)          (8)           E.A sub 1 (2)           8     |  5      E.A sub 1 <- kt+2 (2)          O.A sub 1 (6)            8
        //   2      E.A sub 0 <- kt+2 (2)          (2)           O.A sub 0 (3, 4)        8     |  6      O.A sub 0 <- kt+3 (2)          E.A sub 0 (7, 8)         8
        //   3      E.W sub 0 (2), sub 1 (1)       (2, 3)        O.W sub 0 (4)           9     |  7      O.W sub 0 (2), sub 1 (1)       E.W sub 0 (8)            9
        //   4      E.W sub 1 (1)                  (3)           O.W sub 1 (5)           8     |  8      O.W sub 1 (1)                  E.W sub 1 (1)            8
        // Fragment reads (round 5): phases 4 / 8 used to issue twelve (A0: 8, W: 4) and phases 3 / 7 none — twelve 1-KiB reads of eight waves are 384 cycles of the
        // LDS's 256 B/clk inside a 512-cycle phase, next to the DMA's writes: 619 core cycles against 540 (profiles/r04_x_gemm8_phase_cycles.txt).  A0's registers are
        // free from phase 3 on (its MFMAs run in phases 1, 2), so the first half of that read (fragment rows 0, 1) moves to phases 3 / 7: 4 / 8 / 4 / 8 reads per phase;
        // the sub-tile it reads is retired one phase earlier for that (the waits of phases 2 / 6: two pieces issued since).
        // Measured (profiles/r04_x_gemm8_phase_cycles.txt): with four pieces in each of phases 3, 4, 7, 8 those phases took 700 - 900 core cycles against 520 - 540 for
        // the phases without DMA (2 x 256 matrix-pipe cycles) — sixteen 1-KiB pieces of four waves in one slot are 256 cycles of the CU's 64-B/clk vector-memory path
        // alone; 2 / 2 / 2 / 2: 550 / 655 for phases 3 / 4; 2 / 2 / 3 / 1: 540 / 619.  In the last iteration (nothing left to stage) the waits count down what is in flight.
        for (int kt = 0; kt < nk_it; kt += 2) {
            const bool more = kt + 2 < nk_it;       // block-uniform
            const bool dm = more && !no_dma;
            // ---- phase 1
            G8_PSTAMP(0);
            if (!no_rd) G8_RD_W(Y, 0, 1);
            if (!no_dma) stage_a(1, kt + 1, 2, 4);
            G8_VMCNT(8);
            G8_BAR(); G8_MMQ(4, A0, X, 0, 0); G8_BAR();
            // ---- phase 2
            G8_PSTAMP(1);
            if (!no_rd) G8_RD_A(A1, 0, 1);
            if (dm) { stage_a(0, kt + 2, 0, 2); G8_VMCNT(8); } else { G8_VMCNT(6); }      // O.A sub 0 has landed: its first half is read in phase 3
            G8_BAR(); G8_MMQ(8, A0, Y, 0, 1); G8_BAR();
            // ---- phase 3
            G8_PSTAMP(2);
            if (!no_rd) G8_RD_A_LO(A0, 1, 0);
            if (dm) { stage_w(0, kt + 2, 0, 3); G8_VMCNT(9); } else { G8_VMCNT(4); }
            G8_BAR(); G8_MMQ(4, A1, Y, 1, 1); G8_BAR();
            // ---- phase 4
            G8_PSTAMP(3);
            if (!no_rd) { G8_RD_A_HI(A0, 1, 0); G8_RD_W(Y, 1, 0); }
            if (dm) { stage_w(0, kt + 2, 3, 4); G8_VMCNT(8); } else { G8_VMCNT(2); }
            G8_BAR(); G8_MMQ(12, A1, X, 1, 0); G8_BAR();
            // ---- phase 5
            G8_PSTAMP(4);
            if (!no_rd) G8_RD_W(X, 1, 1);
            if (dm) { stage_a(0, kt + 2, 2, 4); G8_VMCNT(8); } else { G8_VMCNT(0); }
            G8_BAR(); G8_MMQ(4, A0, Y, 0, 0); G8_BAR();
            // ---- phase 6
            G8_PSTAMP(5);
            if (!no_rd) G8_RD_A(A1, 1, 1);
            if (dm) { stage_a(1, kt + 3, 0, 2); G8_VMCNT(8); }                            // E.A sub 0 (tile kt + 2) has landed: first half read in phase 7
            G8_BAR(); G8_MMQ(8, A0, X, 0, 1); G8_BAR();
            // ---- phase 7
            G8_PSTAMP(6);
            if (!no_rd) G8_RD_A_LO(A0, 0, 0);               // (after the last tile: a harmless read of stale data)
            if (dm) { stage_w(1, kt + 3, 0, 3); G8_VMCNT(9); }
            G8_BAR(); G8_MMQ(4, A1, X, 1, 1); G8_BAR();
            // ---- phase 8
            G8_PSTAMP(7);
            if (!no_rd) { G8_RD_A_HI(A0, 0, 0); G8_RD_W(X, 0, 0); }
            if (dm) { stage_w(1, kt + 3, 3, 4); G8_VMCNT(8); }
            G8_BAR(); G8_MMQ(12, A1, Y, 1, 0); G8_BAR();
        }
        } else {     // four pieces per wave in phases 3, 4, 7, 8 (the header's table): the implicit-GEMM convolutions
        G8_STAMP(0);
        G8_VMCNT(8);                                // E has landed (O may still be in flight); older epilogue stores have drained
        G8_STAMP(1);
        G8_BAR();
        if (!no_rd) { G8_RD_A(A0, 0, 0); G8_RD_W(X, 0, 0); }
        if (wr == 1) G8_BAR();                      // stagger the second wave group by one barrier

        for (int kt = 0; kt < nk_it; kt += 2) {
            const bool more = kt + 2 < nk_it;       // block-uniform
            // ---- phase 1
            if (!no_rd) G8_RD_W(Y, 0, 1);
            G8_BAR(); G8_MMQ(4, A0, X, 0, 0); G8_BAR();
            // ---- phase 2
            if (!no_rd) G8_RD_A(A1, 0, 1);
            G8_BAR(); G8_MMQ(8, A0, Y, 0, 1); G8_BAR();
            // ---- phase 3
            if (more && !no_dma) { stage_w(0, kt + 2); G8_VMCNT(4); } else { G8_VMCNT(0); }
            G8_BAR(); G8_MMQ(0, A1, Y, 1, 1); G8_BAR();
            // ---- phase 4
            if (!no_rd) { G8_RD_A(A0, 1, 0); G8_RD_W(Y, 1, 0); }
            if (more && !no_dma) stage_a(0, kt + 2);
            G8_BAR(); G8_MMQ(12, A1, X, 1, 0); G8_BAR();
            // ---- phase 5
            if (!no_rd) G8_RD_W(X, 1, 1);
            G8_BAR(); G8_MMQ(4, A0, Y, 0, 0); G8_BAR();
            // ---- phase 6
            if (!no_rd) G8_RD_A(A1, 1, 1);
            G8_BAR(); G8_MMQ(8, A0, X, 0, 1); G8_BAR();
            // ---- phase 7
            if (more && !no_dma) { stage_w(1, kt + 3); G8_VMCNT(4); } else { G8_VMCNT(0); }
            G8_BAR(); G8_MMQ(0, A1, X, 1, 1); G8_BAR();
            // ---- phase 8
            if (!no_rd) { G8_RD_A(A0, 0, 0); G8_RD_W(X, 0, 0); }     // (after the last tile: a harmless read of stale data)
            if (more && !no_dma) stage_a(1, kt + 3);
            G8_BAR(); G8_MMQ(12, A1, Y, 1, 0); G8_BAR();
        }
        }
        }
        G8_STAMP(2);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (wr == 0) G8_BAR();                      // re-align the wave groups: every LDS read has retired
        G8_STAMP(3);

        const int em0 = m0, en0 = n0; const bool eswap = swap;
        const bool has_next = item + 1 < nitems;
        const bool park_it = PSHIFT && phi && item == 0, unpark_it = PSHIFT && phi && item == ntl;
        // bias of this lane's outputs from the tile's LDS slot (staged with the tile's first DMA group)
        float bv[16];
        const int elane = g8_fresh_lane();          // the epilogue's own copy of the lane id (see g8_fresh_lane)
        if (TRANS) {
            const float* bs = reinterpret_cast<const float*>(smem8 + G8_BIAS + slot * 1024);
            if (!eswap) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 b4 = (f32x4)(0.f);
                    if (p.bias) b4 = *reinterpret_cast<const f32x4*>(bs + wc * 64 + (elane >> 4) * 16 + q * 4);
                    bv[q * 4 + 0] = b4[0]; bv[q * 4 + 1] = b4[1]; bv[q * 4 + 2] = b4[2]; bv[q * 4 + 3] = b4[3];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) { bv[i] = p.bias ? bs[wr * 128 + i * 16 + (elane & 15)] : 0.f; bv[8 + i] = 0.f; }
            }
        }
        // Epilogues that read a residual issue the next tile's DMA after their last load (VMEM loads retire in order: a
        // load issued behind the DMA group could only be consumed once the whole group had landed).
        const bool dma_first = TRANS && has_next && !(OMODE == OUT_LINEAR && p.res) && !(p.dbg & 16);
        if (dma_first) {                            // direct epilogues do not touch LDS: start the next tile's DMA first
            tile_setup(item + 1, m0, n0, swap);
            stage_prologue(n0, swap, slot ^ 1);
        }
        if constexpr (PSHIFT) {
            // lane-linear scratch image [wave][fragment][lane] of f32x4: every instruction moves 1 KiB; written and read by the same lanes
            if (park_it) {                          // K tiles [phi, nk) of the first tile: keep the partial sums, no epilogue
                const int lo = g8_fresh_lane() * 4; // (opaque: the 32 addresses below are loop invariant and would be hoisted out of the item loop)
                float* pk = p.park + ((size_t)blockIdx.x * 256 + (size_t)wave * 32) * 256 + lo;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(acc[i][j], reinterpret_cast<f32x4*>(pk + (i * 4 + j) * 256));
                if (has_next && !dma_first) { tile_setup(item + 1, m0, n0, swap); stage_prologue(n0, swap, slot ^ 1); }
                continue;
            }
        }
        // last item of a phase-shifted walk (no DMA in flight): K tiles [0, phi) + the parked partial sums, which the epilogue adds
        // to its per-row temporaries (the accumulators themselves are never modified outside the K loop: a conditional update of
        // all 128 of them made the register allocator keep two copies — 72 spilled VGPRs around EVERY epilogue)
        const float* parked = nullptr;
        if constexpr (PSHIFT) {
            if (unpark_it) {
                const int lo = g8_fresh_lane() * 4;
                parked = p.park + ((size_t)blockIdx.x * 256 + (size_t)wave * 32) * 256 + lo;
            }
        }
        G8_STAMP(4);
        if (no_epi) {      // experiment: keep the accumulators live, one store per lane
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            reinterpret_cast<half_t*>(p.out)[(long)(em0 + wr * 128 + (lane >> 4)) * p.ldc + en0 + wc * 64 + (lane & 15)] = (half_t)t;
            if (has_next && !dma_first) { tile_setup(item + 1, m0, n0, swap); stage_prologue(n0, swap, slot ^ 1); }
        } else if (TRANS) {
            if (eswap) epilogue8_vt(p, acc, bv, en0 + wr * 128, em0 + wc * 64, elane);
            else epilogue8_direct<OMODE>(p, acc, bv, em0 + wr * 128, en0 + wc * 64, elane, reinterpret_cast<const float*>(smem8 + G8_LUT), parked);
            G8_STAMP(5);
            if (has_next && !dma_first) {
                tile_setup(item + 1, m0, n0, swap);
                stage_prologue(n0, swap, slot ^ 1);
            }
        } else {
            float* st = reinterpret_cast<float*>(smem8) + wave * (16 * 68);
            gemm_epilogue_lds<half_t, OMODE, 8, 4>(p, acc, em0 + wr * 128, en0 + wc * 64, st, lane);
            if (has_next) {                         // staged epilogue used LDS: fence it before the next tile's DMA
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                G8_BAR();
                tile_setup(item + 1, m0, n0, swap);
                stage_prologue(n0, swap, slot ^ 1);
            }
        }
    }
}

// Scratch of the phase-shifted walk: 256 KB per workgroup, one buffer per stream (launches of one stream are ordered; the forward
// runs all its GEMMs on one stream).  Allocated on first use, kept for the life of the process.
}  // namespace
float* park_scratch(hipStream_t stream, int grid) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::pair<float*, int>> ws;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto& e = ws[{dev, stream}];
    if (e.second < grid) {
        if (e.first) (void)hipFree(e.first);
        e.first = nullptr; e.second = 0;
        void* q = nullptr;
        if (hipMalloc(&q, (size_t)grid * 256 * 256 * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        e.first = reinterpret_cast<float*>(q); e.second = grid;
    }
    return e.first;
}
namespace {

#ifdef CVA_ABLATION
}  // namespace
}  // namespace cva
// experiment (ablation builds only, not part of the C ABI): copy the timeline stamps of the last stamped launch on the null stream to the host
extern "C" int cv_dbg_gemm_stamps(long long* dst, int n) {
    float* src = cva::park_scratch(nullptr, 1);
    if (!src || hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpy(dst, src, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
namespace cva {
namespace {
#endif

template <int OMODE, int TRANS, int ABL, int F8 = 0, int CV3 = 0>
int launch8(const GemmParams& p, hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8_kernel<OMODE, TRANS, ABL, F8, CV3>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return (int)hipGetLastError();
        attr = true;
    }
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
            n_cu = 256;
    }
    const int tiles = (p.M / G8_BM) * (p.N / G8_BN);
    const int grid = (tiles < n_cu || (p.dbg & 32)) ? tiles : n_cu;   // persistent: one workgroup per CU walks tiles grid-stride
    if (OMODE == OUT_LINEAR && TRANS == 1 && !F8 && !CV3 && ABL == 0) {
        // Phase-shifted walk (file header) where the epilogue moves many bytes per tile: fp32 output / fp32 residual (fc2: 512 KB per
        // tile).  Same-call A/B (profiles/r04_b_gemm_phase_shift.txt, M = 262144): fc2 3255 -> 3204 us (-1.6 %), proj 930 -> 932 (0),
        // fc1 3159 -> 3181 (+0.7 %: its GELU epilogue is VALU bound, the park / unpark traffic is pure cost) — so fc2-like launches only.
        // Needs >= 2 tiles per workgroup and >= 8 K tiles.
        GemmParams q = p;
        static const int mode = cva_env_int("CVA_GEMM_PHASE", -1);       // ablation builds: 0 off, 1 every OUT_LINEAR launch, -1 heuristic
        const int nk = p.K / G8_BK;
        bool want = p.out_f32 || p.res;
        if (mode == 0) want = false; else if (mode == 1) want = true;
        q.park = ((want || (p.dbg & 32768)) && tiles >= 2 * grid && nk >= 8 && grid % 8 == 0) ? park_scratch(stream, grid) : nullptr;
        hipLaunchKernelGGL((gemm8_kernel<OMODE, TRANS, ABL, F8, CV3>), dim3(grid), dim3(G8_NT), G8_LDS, stream, q);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL((gemm8_kernel<OMODE, TRANS, ABL, F8, CV3>), dim3(grid), dim3(G8_NT), G8_LDS, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

bool gemm8_supported(const GemmParams& p, int a_mode, size_t elem_size) {
    if (elem_size != 2 || a_mode != A_LINEAR || !p.epi_vec) return false;
    if (p.M % G8_BM || p.N % G8_BN || p.K % (2 * G8_BK) || p.K < 2 * G8_BK) return false;
    if (((size_t)p.A & 15) || ((size_t)p.W & 15) || (p.lda % 8) || (p.ldw % 8) || p.ldw < p.K) return false;
    // (32-bit lane offsets are relative to a tile's first row: 256 rows, plus the few rows an a_rpi remap can insert)
    if ((256L + (p.a_rpi > 0 ? (256L / p.a_rpi + 1) * p.a_extra : 0)) * p.lda * 2 >= (1L << 31) || 256L * p.ldw * 2 >= (1L << 31)) return false;
    if (p.n_valid && (p.n_valid % 16 || p.n_valid > p.N || p.N - p.n_valid >= G8_BN || p.out_mode == OUT_CONVT)) return false;
    if (p.m_valid && (p.m_valid > p.M || p.M - p.m_valid >= G8_BM)) return false;
    // fused qkv projection: the v columns (operands exchanged, V^T epilogue) must start on a column tile
    if (p.out_mode == OUT_QKV && ((2 * p.D) % G8_BN || p.hd % 16 || (p.n_valid ? p.n_valid : p.N) != 3 * p.D || p.n_off)) return false;
    return true;
}

// Implicit 3x3 convolution (stride 1, zero padding 1) on the 8-phase kernel: M = B*H*W NHWC pixels, N = Cout, K = 9 * (C1 + C2)
// with k = tap * Ctot + c (the packed filter layout of pack_conv3).  Shapes it takes: Cout a multiple of 256, 64-channel K
// steps that never straddle the two sources, power-of-two image sides (a 256-pixel tile never straddles two images).
static int ilog2_exact(long v) { int s = 0; while ((1L << s) < v) ++s; return (1L << s) == v ? s : -1; }
bool gemm8_conv3_supported(const GemmParams& p) {
    const int Ctot = p.C1 + p.C2;
    if (p.out_mode != OUT_LINEAR || p.out_f32 || p.res || p.head_W || !p.A || !p.W || !p.out) return false;
    if (p.N % G8_BN || p.M % G8_BM || p.C1 % G8_BK || (p.C2 && (p.C2 != p.C1 || !p.A2))) return false;
    if (ilog2_exact(Ctot / G8_BK) < 1 || ilog2_exact(p.Wd) < 0 || ilog2_exact((long)p.H * p.Wd) < 8) return false;
    if (p.K != 9 * Ctot || p.ldw != p.K || p.ldc % 8 || ((size_t)p.A & 15) || ((size_t)p.A2 & 15) || ((size_t)p.W & 15)) return false;
    if ((258L + 2L * p.Wd) * p.C1 * 2 >= (1L << 30) || 256L * p.ldw * 2 >= (1L << 31)) return false;
    return true;
}

int launch_gemm8_conv3(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    if (!gemm8_conv3_supported(p)) return -1;
    p.dbg = 0; p.epi_vec = 1; p.a_rpi = 0; p.o_rpi = 0;
    { static const int km = cva_env_int("CVA_CONV_KMAJOR", -1); if (km >= 0) p.conv_kmajor = km; }   // ablation builds only (timing A/B)
    p.conv_wshift = ilog2_exact(p.Wd);
    p.conv_cshift = ilog2_exact((p.C1 + p.C2) / G8_BK);
    return launch8<OUT_LINEAR, 1, 0, 0, 1>(p, stream);
}

// ConvTranspose2d(k2, s2) followed by Conv2d(3x3, pad 1): output pixel (2y + py, 2x + px) depends on the 2 x 2 input pixels
// (y + py - 1 + s, x + px - 1 + t), s, t in {0, 1}, through weights composed on the host (cellvit_abi.hip, pack_deconv_comp) — one
// contraction of K = 4 * Cin per output parity over the LOW-resolution pixels instead of K = Cin (transposed convolution) plus
// K = 9 * Cout at four times the pixels, and the up-sampled intermediate never exists.  The A tile of K step (chunk, s, t) is the
// 3x3-convolution tap (py + s, px + t) of the same machinery (stage_a_conv); a 256-column tile lies inside one parity.
// With a second source A2 (NHWC [B, 2H, 2W, C2]: the skip connection the up-sampled map is concatenated with, cellvit.py:236-242) the
// same accumulators also take the 3x3 convolution of that source — K grows by 9 * C2 steps whose A tiles are the tap-shifted
// OUTPUT-resolution pixels (2y + py, 2x + px) of the tile's input pixels (a stride-2 gather: per-lane offsets, as before).
bool gemm8_deconv_supported(const GemmParams& p) {
    if (p.out_mode != OUT_CONVT || p.out_f32 || p.res || p.head_W || !p.A || !p.W || !p.out || !p.bias || !p.comp_bias) return false;
    if ((p.A2 != nullptr) != (p.C2 > 0) || p.C2 % G8_BK) return false;
    if (p.N % 4 || (p.N / 4) % G8_BN || p.M % G8_BM || p.C1 % G8_BK || p.C1 < G8_BK) return false;
    if (ilog2_exact(p.Wd) < 0 || ilog2_exact((long)p.H * p.Wd) < 8) return false;
    const int K = 4 * p.C1 + 9 * p.C2;
    if (p.K != K || p.ldw != K || (K / G8_BK) % 2 || 9 * p.C2 / G8_BK >= 4096) return false;
    if (((size_t)p.A & 15) || ((size_t)p.A2 & 15) || ((size_t)p.W & 15) || ((size_t)p.out & 15) || ((size_t)p.comp_bias & 15)) return false;
    if ((258L + 2L * p.Wd) * p.C1 * 2 >= (1L << 30) || (1026L + 4L * p.Wd) * p.C2 * 2 >= (1L << 30) || 256L * p.ldw * 2 >= (1L << 31)) return false;
    return true;
}

int launch_gemm8_deconv(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    if (!gemm8_deconv_supported(p)) return -1;
    p.dbg = 0; p.epi_vec = 1; p.a_rpi = 0; p.o_rpi = 0; p.conv_kmajor = 1;
    p.conv_wshift = ilog2_exact(p.Wd);
    p.conv_cshift = 0;
    return launch8<OUT_CONVT, 1, 0, 0, 1>(p, stream);
}

bool gemm8_f8_supported(const GemmParams& p) {
    if (p.M % G8_BM || p.N % G8_BN || p.K % 256 || p.K < 256) return false;             // K tiles of 128 elements, two per iteration
    if (((size_t)p.A & 15) || ((size_t)p.W & 15) || (p.lda % 16) || (p.ldw % 16) || p.lda < p.K || p.ldw < p.K) return false;
    if (p.a_rpi || !p.a_scale || !p.w_scale || p.v_rm) return false;      // (fp8 V tiles run swapped: V^T only)
    if (((size_t)p.a_scale & 3) || ((size_t)p.w_scale & 3)) return false;
    if (256L * p.lda >= (1L << 31) || 256L * p.ldw >= (1L << 31)) return false;               // 32-bit lane offsets inside a tile
    if (p.out_mode == OUT_QKV && (p.D % G8_BN || p.hd % 16 || p.N != 3 * p.D || p.n_off || !p.a_scale_w)) return false;
    if (p.out_mode == OUT_MX8 && (!p.out || !p.out_scale || p.ldc % 16 || p.N % 128)) return false;
    if (p.out_mode == OUT_CONVT) return false;
    return true;
}

int launch_gemm8_f8(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    p.dbg = 0; p.epi_vec = 1;
    if (!gemm8_f8_supported(p)) return (int)hipErrorInvalidValue;
    if (p.out_mode == OUT_LINEAR) return launch8<OUT_LINEAR, 1, 0, 1>(p, stream);
    if (p.out_mode == OUT_QKV) return launch8<OUT_QKV, 1, 0, 1>(p, stream);
    return launch8<OUT_MX8, 1, 0, 1>(p, stream);
}

int launch_gemm8(const GemmParams& p, hipStream_t stream) {
#ifdef CVA_ABLATION
    {   // experiment (ablation builds): CVA_GEMM4 = 10 + s runs the one-wave-per-SIMD kernel of gemm4.hip with slot placement s, 30 + a its
        // work-skipping instantiations.  Measured slower than this kernel (profiles/r03_exp_gemm4.txt), so production never routes there.
        static const int g4 = cva_env_int("CVA_GEMM4", 0);
        if (g4 >= 10 && !(p.dbg & 7) && gemm4_takes(p)) return launch_gemm4(p, g4 - 10, stream);
    }
#endif
    if (p.out_mode == OUT_LINEAR) {
#ifdef CVA_ABLATION      // work-skipping instantiations exist in ablation builds only (p.dbg is 0 otherwise)
        switch (p.dbg & 7) {
            case 4: return launch8<OUT_LINEAR, 1, 4>(p, stream);
            case 7: return launch8<OUT_LINEAR, 1, 7>(p, stream);
            case 1: return launch8<OUT_LINEAR, 1, 1>(p, stream);
            case 2: return launch8<OUT_LINEAR, 1, 2>(p, stream);
            case 3: return launch8<OUT_LINEAR, 1, 3>(p, stream);
            case 5: return launch8<OUT_LINEAR, 0, 0>(p, stream);      // LDS-staged epilogue, for A/B
            default: break;
        }
#endif
        return launch8<OUT_LINEAR, 1, 0>(p, stream);
    }
    if (p.out_mode == OUT_QKV) return launch8<OUT_QKV, 1, 0>(p, stream);
    if ((p.N >> 2) % 16 == 0 && !p.out_f32 && !(p.dbg & 2048)) return launch8<OUT_CONVT, 1, 0>(p, stream);   // direct 32-byte runs
    return launch8<OUT_CONVT, 0, 0>(p, stream);
}

}  // namespace cva
