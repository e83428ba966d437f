// 256 x 128 x 32 fp16 MFMA contraction, TWO independent workgroups per CU (gfx950) — an EXPERIMENT, compiled into the ablation
// flavour only (python -m cellvit_amd.build --ablation; CVA_GEMM2 = 1).  Measured 15-25 % slower than gemm8.hip in all three
// variants that were tried (profiles/r03_exp_gemm2.txt); results are correct (full-matrix checks, no races).
//
// Why: the additive ablations of the 256 x 256 kernels (gemm8.hip: 8 waves in barrier-separated phases; gemm4.hip: one wave per
// SIMD) say that the matrix pipe idles while a workgroup does anything else — operand DMA issue (the waves queue for the CU's
// one address unit), the epilogue (store-issue bound, 6-12 us per tile), barrier hand-overs.  Inside ONE workgroup none of it can
// be overlapped without a barrier protocol that costs as much as it hides (gemm8.hip), and a per-CU stagger of the workgroups
// was measured slower (the lock step is what makes the shared panels hit in L2).  Two UNSYNCHRONISED workgroups on a CU overlap
// each other's non-MFMA time for free (same idea as conv3x3_halo4_kernel): while one stores its tile or waits for its DMA, the
// other one's MFMAs own the pipe.
//
// Workgroup: 256 threads = 4 waves, 2 (M) x 2 (N); tile 256 x 128, wave tile 128 x 64 = 8 x 4 fragments of
// v_mfma_f32_16x16x32_f16 (128 accumulator VGPRs, as gemm8.hip) — so the direct epilogues of gemm8_epi.h apply unchanged.
// K is walked in steps of 32 (= one MFMA K) through a ring of THREE 24-KiB stages (A: 16 fragment blocks of 1 KiB, W: 8): 72 KiB
// per workgroup + bias + the GELU table = 78 KiB, two workgroups fit the CU's 160 KiB.  A fragment block is 16 rows x 64 B, written
// by ONE LDS-DMA instruction (lane l: row l >> 2, slot l & 3) with a source-side XOR swizzle of the four 16-byte slots of a row
// that makes the fragment reads (ds_read_b128: lane (g, li) <- row li, piece g) bank-conflict free.  A wave issues 6 of the 24
// blocks of a stage.
//
// (GELU: the x * Phi(x) table of gemm8_epi.h at half the resolution, h = 1/64: |dPhi| < 7.6e-6, still two orders below half an
// fp16 ulp of the result — the table has to fit twice into a CU.)
//
// Step t (one barrier):  s_waitcnt vmcnt(6)   this wave's blocks of stage t+1 have landed (those of t+2 stay in flight)
//                        s_barrier            => everyone's have, and everyone has finished reading stage t (fragments of step t
//                                                were fetched during step t-1)
//                        DMA stage t+3 -> ring slot t % 3 (just freed), 32 MFMAs on fragment set t & 1, interleaved with the 12
//                        fragment reads of step t+1 -> set (t+1) & 1
// The K loop runs over ALL of a workgroup's tiles as one stream (the last steps of a tile stage the first steps of the next
// one), as gemm4.hip does.  Operand traffic L2 -> LDS is 1.5 x that of a 256 x 256 tile; the fabric-side footprint of an XCD
// (64 workgroups = 8 x 8 tiles = 2048 x 1024 outputs) is the same as gemm8.hip's 8 x 4 tiles of 256 x 256.
#include "gemm.h"
#include "gemm_epilogue.h"
#include "gemm8_epi.h"

namespace cva {

namespace {

using namespace epi;
using namespace g8;

constexpr int G2_BM = 256, G2_BN = 128, G2_BK = 32, G2_NT = 256;
constexpr int G2_BLK = 1024;                      // one fragment block: 16 rows x 64 B
constexpr int G2_STAGE = 24 * G2_BLK;             // 16 A blocks + 8 W blocks
constexpr int G2_BIAS = 3 * G2_STAGE;             // two 1-KiB slots (one LDS-DMA instruction each; 128 floats used)
constexpr int G2_LUT = G2_BIAS + 2048;            // GELU: Phi(x) at x = -8 + i/64, i = 0 .. 1024 (fp32)
constexpr int G2_LUTN = 1024;
constexpr int G2_LDS = G2_LUT + (G2_LUTN + 1) * 4 + 12;

#define G2_SB() __builtin_amdgcn_sched_barrier(0)
#define G2_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int OMODE>
__global__ __launch_bounds__(G2_NT, 2) void gemm2_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = p.N / G2_BN, tiles_m = p.M / G2_BM, ntiles = tiles_m * tiles_n;
    const int nk = p.K / G2_BK;                     // >= 3 (host)

    // ---- DMA state of the tile whose K steps are staged NEXT.  Block b of a stage: b < 16 -> A rows 16b .. 16b+15, else W rows
    // 16(b-16) ..; this wave issues blocks wave + 4q, q = 0 .. 5 (q < 4: A, q >= 4: W).  Lane l fetches (row l & 15, piece l >> 4).
    // DMA lane l -> row l >> 2, 16-byte LDS slot l & 3 of the row's 64 B: four ADJACENT lanes fetch one row segment (coalesced; with
    // the lane-linear order of the first version — lane -> (row l & 15, piece l >> 4) — every 16-lane quantum of an instruction touched
    // 16 different lines: measured 35 % slower than gemm8.hip).  LDS slot pp of row r holds logical piece pp ^ SWZ[(r >> 2) & 3]:
    // with SWZ = {0, 2, 3, 1} the 16 lanes of every ds_read_b128 service group hit 16 different bank quads.
    const int swz_tab = 0x1320;                     // SWZ[k] = (swz_tab >> 4k) & 3
    const int drow = lane >> 2, dpc = (lane & 3) ^ ((swz_tab >> (((drow >> 2) & 3) * 4)) & 3);
    unsigned voff[6];
    const unsigned char* Ab;
    const unsigned char* Wb;
    auto a_row = [&](int m) -> long {
        long r = m;
        if (p.a_rpi > 0) r = (long)m + (long)(m / p.a_rpi) * p.a_extra + p.a_off;
        return r;
    };
    auto tile_setup = [&](int tile, int& m0, int& n0) {
        int tm, tn;
        tile_coords(xcd_remap(tile, ntiles), tiles_m, tiles_n, tm, tn);
        m0 = tm * G2_BM; n0 = tn * G2_BN;
        const long ar0 = a_row(m0);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int b = wave + 4 * q;
            if (q < 4) {
                voff[q] = (unsigned)((a_row(m0 + b * 16 + drow) - ar0) * (long)p.lda * 2) + dpc * 16;
            } else {
                const int row = (b - 16) * 16 + drow;        // LDS row of the W tile; it holds source row prow (see gemm8.hip: a lane's 16
                const int prow = (row & ~63) | (((row >> 2) & 3) << 4) | (((row >> 4) & 3) << 2) | (row & 3);   // accumulator values = 16 consecutive columns)
                voff[q] = (unsigned)((long)prow * p.ldw * 2) + dpc * 16;
            }
        }
        Ab = reinterpret_cast<const unsigned char*>(p.A) + ar0 * (long)p.lda * 2;
        Wb = reinterpret_cast<const unsigned char*>(p.W) + (long)n0 * p.ldw * 2;
    };

    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem2;
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + wave * G2_BLK);      // block `wave` of ring slot 0
    // stage K step `kt` of the current DMA tile into the ring slot at byte offset `soff` (6 LDS-DMA instructions)
    auto stage = [&](unsigned soff, int kt) {
        const unsigned char* ab = uniform_ptr(Ab + (long)kt * (G2_BK * 2));
        const unsigned char* wb = uniform_ptr(Wb + (long)kt * (G2_BK * 2));
        const unsigned d0 = __builtin_amdgcn_readfirstlane(lds_wave + soff);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const unsigned dst = d0 + q * 4 * G2_BLK;
            if (q < 4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff[q]), "s"(ab), "s"(dst) : "memory");
            else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff[q]), "s"(wb), "s"(dst) : "memory");
        }
    };
    auto stage_bias = [&](int n0_, int slot) {      // 128 floats = 512 B: lanes 0-31 of wave 0 (lanes 32-63 write the same values into the slot's unused half)
        if (p.bias && wave == 0) {
            const unsigned char* src = uniform_ptr(reinterpret_cast<const unsigned char*>(p.bias + n0_));
            const unsigned boff = (lane & 31) * 16;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + G2_BIAS + slot * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(boff), "s"(src), "s"(dst) : "memory");
        }
    };

    // fragment read addresses: lane (g, li) reads logical piece g of row li = LDS slot g ^ SWZ[(li >> 2) & 3]; A blocks wr*8 + i, W blocks 16 + wc*4 + j
    const int fg = lane >> 4, fli = lane & 15;
    const unsigned rd0 = lds0 + fli * 64 + ((fg ^ ((swz_tab >> (((fli >> 2) & 3) * 4)) & 3)) << 4);
    const unsigned a_rd = rd0 + (wr * 8) * G2_BLK, w_rd = rd0 + (16 + wc * 4) * G2_BLK;      // + slot * G2_STAGE + i * 1 KiB

    if (OMODE == OUT_LINEAR && p.act == ACT_GELU) {               // (read only in epilogues: many barriers later)
        float* lut = reinterpret_cast<float*>(smem2 + G2_LUT);
        for (int i = threadIdx.x; i <= G2_LUTN; i += G2_NT) lut[i] = 0.5f * (1.0f + erff((-8.0f + (float)i * (1.0f / 64.f)) * 0.70710678118654752f));
    }
    // crude de-phasing of the two workgroups of a CU (the second half of the grid is dispatched into the second slots): the
    // late half starts ~8 us later, about one epilogue, so that the two never store at the same time.  Performance only.
    if (blockIdx.x >= (gridDim.x >> 1)) {
        __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);          // 2 x 127 x 64 cycles
    }

    half8_t FA[2][8], FW[2][4];
#define G2_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define G2_RD_W(SET, j) G2_DSR(FW[SET][j], w_nx, (j) * G2_BLK)
#define G2_RD_A(SET, i) G2_DSR(FA[SET][i], a_nx, (i) * G2_BLK)
#define G2_WAIT_SET(SET)                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                \
                 : "+v"(FA[SET][0]), "+v"(FA[SET][1]), "+v"(FA[SET][2]), "+v"(FA[SET][3]), "+v"(FA[SET][4]),           \
                   "+v"(FA[SET][5]), "+v"(FA[SET][6]), "+v"(FA[SET][7]), "+v"(FW[SET][0]), "+v"(FW[SET][1]),           \
                   "+v"(FW[SET][2]), "+v"(FW[SET][3])                                                                  \
                 :: "memory")
#define G2_MM(SET, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FW[SET][j], FA[SET][i], acc[i][j], 0, 0, 0)
// DMA block q of the step being staged (issued between the MFMA rows: one load per row instead of a burst of six per wave —
// 24 per workgroup — behind the barrier, where they queue for the CU's address unit)
#define G2_DMA(q, base)                                                                                                \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff[q]), "s"(base), "s"(d0_ + (q) * 4 * G2_BLK) : "memory")
// one row of a step: A fragment i x the 4 W fragments, then one slot (fragment reads of the next step)
#define G2_ROW(SET, i, OP)                                                                                             \
    do { G2_MM(SET, i, 0); G2_MM(SET, i, 1); G2_MM(SET, i, 2); G2_MM(SET, i, 3); OP; G2_SB(); } while (0)
// One step on fragment set SET: the wait + barrier, the DMA of step t+3 into the slot this step frees, 32 MFMAs with the 12
// fragment reads of step t+1 (from the NEXT ring slot) in the slots of rows 0-5; then the ring offsets rotate.
#define G2_STEP(SET)                                                                                                   \
    do {                                                                                                               \
        G2_VMCNT(6); G2_SB();                                                                                          \
        __builtin_amdgcn_s_barrier(); G2_SB();                                                                         \
        const int kd_ = dtile < ntiles ? dkt : 0;   /* (past the last tile: step 0 of the last tile again: 6 loads per step) */ \
        const unsigned char* ab_ = uniform_ptr(Ab + (long)kd_ * (G2_BK * 2));                                          \
        const unsigned char* wb_ = uniform_ptr(Wb + (long)kd_ * (G2_BK * 2));                                          \
        const unsigned d0_ = __builtin_amdgcn_readfirstlane(lds_wave + s_cur);                                         \
        const unsigned a_nx = a_rd + s_nxt, w_nx = w_rd + s_nxt;                                                       \
        G2_WAIT_SET(SET); G2_SB();                                                                                     \
        G2_ROW(SET, 0, G2_RD_W(SET ^ 1, 0); G2_RD_W(SET ^ 1, 1); G2_DMA(0, ab_));                                       \
        G2_ROW(SET, 1, G2_RD_W(SET ^ 1, 2); G2_RD_W(SET ^ 1, 3); G2_DMA(1, ab_));                                       \
        G2_ROW(SET, 2, G2_RD_A(SET ^ 1, 0); G2_RD_A(SET ^ 1, 1); G2_DMA(2, ab_));                                       \
        G2_ROW(SET, 3, G2_RD_A(SET ^ 1, 2); G2_RD_A(SET ^ 1, 3); G2_DMA(3, ab_));                                       \
        G2_ROW(SET, 4, G2_RD_A(SET ^ 1, 4); G2_RD_A(SET ^ 1, 5); G2_DMA(4, wb_));                                       \
        G2_ROW(SET, 5, G2_RD_A(SET ^ 1, 6); G2_RD_A(SET ^ 1, 7); G2_DMA(5, wb_));                                       \
        G2_ROW(SET, 6, (void)0);                                                                                       \
        G2_ROW(SET, 7, (void)0);                                                                                       \
        dma_cursor_move();                                                                                             \
        { const unsigned t_ = s_cur; s_cur = s_nxt; s_nxt = s_nn; s_nn = t_; }                                         \
    } while (0)

    // ---- the K-step stream over all tiles of this workgroup.  Ring slot of step t: byte offsets (s_cur, s_nxt, s_nn) rotate every
    // step; fragment set = step parity (nk is even, so every tile starts on set 0).  The DMA cursor (dtile, dkt) runs three steps
    // ahead of the MFMAs and crosses into the next tile during a tile's last three steps.
    int tile = blockIdx.x;
    int dtile = tile, dkt = 0;
    int dm0, dn0;
    int bslot = 0;                                  // bias slot of the tile the DMA cursor is in
    tile_setup(dtile, dm0, dn0);
    stage_bias(dn0, bslot);
    auto dma_cursor_move = [&]() {                  // after a step's 6 loads have been issued
        if (++dkt == nk) {
            dkt = 0;
            if (dtile < ntiles) dtile += gridDim.x;
            if (dtile < ntiles) { bslot ^= 1; tile_setup(dtile, dm0, dn0); stage_bias(dn0, bslot); }
        }
    };
    auto dma_advance = [&](unsigned soff) {         // stage the cursor's K step into the slot at `soff`, move the cursor (prologue)
        stage(soff, dtile < ntiles ? dkt : 0);      // (past the last tile: step 0 of the last tile again — every step issues 6 loads)
        dma_cursor_move();
    };
    int m0 = dm0, n0 = dn0;                         // the tile being ACCUMULATED
    unsigned s_cur = 0, s_nxt = G2_STAGE, s_nn = 2 * G2_STAGE;
    dma_advance(0); dma_advance(G2_STAGE); dma_advance(2 * G2_STAGE);
    G2_VMCNT(12);                                   // step 0 (and the first bias) has landed
    G2_SB(); __builtin_amdgcn_s_barrier(); G2_SB();
    {
        const unsigned a_nx = a_rd, w_nx = w_rd;
        G2_RD_W(0, 0); G2_RD_W(0, 1); G2_RD_W(0, 2); G2_RD_W(0, 3);
        G2_RD_A(0, 0); G2_RD_A(0, 1); G2_RD_A(0, 2); G2_RD_A(0, 3);
        G2_RD_A(0, 4); G2_RD_A(0, 5); G2_RD_A(0, 6); G2_RD_A(0, 7);
    }

    int ebslot = 0;
    for (; tile < ntiles; tile += gridDim.x, ebslot ^= 1) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4)(0.f);
        const int em0 = m0, en0 = n0;
        for (int kt = 0; kt < nk; kt += 2) {
            G2_STEP(0);
            G2_STEP(1);
        }
        // (the fragments of the next tile's step 0 are in set 0 already; the cursor moved into that tile during the last three steps)
        m0 = dm0; n0 = dn0;

        float bv[16];
        {
            const float* bs = reinterpret_cast<const float*>(smem2 + G2_BIAS + ebslot * 1024);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 b4 = (f32x4)(0.f);
                if (p.bias) b4 = *reinterpret_cast<const f32x4*>(bs + wc * 64 + (lane >> 4) * 16 + q * 4);
                bv[q * 4 + 0] = b4[0]; bv[q * 4 + 1] = b4[1]; bv[q * 4 + 2] = b4[2]; bv[q * 4 + 3] = b4[3];
            }
        }
        epilogue8_direct<OMODE, 64>(p, acc, bv, em0 + wr * 128, en0 + wc * 64, lane, reinterpret_cast<const float*>(smem2 + G2_LUT));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the stray fetches behind the last tile
}

template <int OMODE>
int launch2(const GemmParams& p, hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<OMODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                G2_LDS) != hipSuccess)
            return (int)hipGetLastError();
        attr = true;
    }
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
            n_cu = 256;
    }
    const int tiles = (p.M / G2_BM) * (p.N / G2_BN);
    const int grid = tiles < 2 * n_cu ? tiles : 2 * n_cu;          // persistent: two workgroups per CU walk the tiles grid-stride
    hipLaunchKernelGGL((gemm2_kernel<OMODE>), dim3(grid), dim3(G2_NT), G2_LDS, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

#ifndef CVA_ABLATION
bool gemm2_supported(const GemmParams&) { return false; }
int launch_gemm2(const GemmParams&, hipStream_t) { return -1; }
#else
bool gemm2_supported(const GemmParams& p) {
    if (p.out_mode != OUT_LINEAR) return false;
    if (p.M % G2_BM || p.N % G2_BN || p.K % (2 * G2_BK) || p.K < 4 * G2_BK) return false;
    if (((size_t)p.A & 15) || ((size_t)p.W & 15) || (p.lda % 8) || (p.ldw % 8) || p.ldw < p.K) return false;
    if ((256L + (p.a_rpi > 0 ? (256L / p.a_rpi + 1) * p.a_extra : 0)) * p.lda * 2 >= (1L << 31) || 128L * p.ldw * 2 >= (1L << 31)) return false;
    return true;
}

int launch_gemm2(const GemmParams& p, hipStream_t stream) { return launch2<OUT_LINEAR>(p, stream); }
#endif

}  // namespace cva
