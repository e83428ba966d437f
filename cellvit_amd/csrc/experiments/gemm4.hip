// 256 x 256 x 64 fp16 MFMA contraction with ONE wave per SIMD (gfx950) — an EXPERIMENT, compiled into the ablation flavour only
// (python -m cellvit_amd.build --ablation; CVA_GEMM4 = 10 + slot placement).  It is the organisation of the vendor library's kernel
// for these shapes (4 waves x 128 x 128 accumulators in AGPRs); built to see whether it beats the 8-wave / 8-phase kernel of
// gemm8.hip.  It does not (same box, same call, profiles/r03_exp_gemm4.txt: fc1 900 vs 965, fc2 948 vs 963, proj 830 vs 825
// TFLOP/s), and its additive ablation says why: MFMAs + the one barrier per K tile alone run at 1717 TFLOP/s, but with ONE wave
// per SIMD nothing covers the wave's own stalls — operand DMA issue costs 17-20 % (four waves issue their LDS-DMA pieces at the
// same moment and queue for the CU's one address unit), fragment reads 5 %, and the epilogue is store-issue bound at HALF the
// rate of eight storing waves (24 % of the fc1 launch).  Results are bit-identical to gemm8.hip (tests/test_gpu_gemm8.py ran on it).
//
// 256 threads = 4 waves laid out 2 (M) x 2 (N); every wave owns a 128 x 128 block of C = 8 x 8 accumulator fragments of
// v_mfma_f32_16x16x32_f16 = 256 accumulator registers (the AGPR half of the unified 512-entry file), the other half holds
// two sets of operand fragments (2 x (8 A + 8 W) x 4 VGPRs = 128) and the addressing.  With one wave per SIMD the matrix
// pipe is fed by ONE in-order instruction stream: the 128 MFMAs of a K tile issue back to back and everything else — the
// fragment reads of the next K step, the LDS-DMA of the K tile after next — is slotted between them, at most one LDS read
// and one DMA per MFMA pair (an MFMA occupies the pipe for 16 cycles = 4 issue slots).  The 8-wave kernel of gemm8.hip
// alternates two waves per SIMD through eight barrier-separated phases per two K tiles instead; its MFMA segments are 16
// instructions long and every hand-over between the two waves costs pipe time (measured there: MFMAs + barriers alone run
// at 75 % of the pipe).  Here a K tile has ONE barrier.
//
// LDS (as gemm8.hip): two K-tile stages, [S0.A][S1.A][S0.W][S1.W], 256 rows x 128 B each, written by direct-to-LDS DMA
// (global_load_lds_dwordx4: 8 rows per wave instruction; a wave stages rows 64w .. 64w+63 of both operands, 16 instructions
// per K tile) with the source-side XOR swizzle piece ^= (row >> 1) & 7; fragment reads apply the same involution.
//
// Schedule of K tile kt (stage s = kt & 1), two K steps of 32:
//   step A   64 MFMAs on fragment set 0;  slots: 16 ds_read_b128 -> set 1 (K step 1 of stage s)
//   step B   s_waitcnt vmcnt(0) (this wave's share of tile kt+1 has landed), lgkmcnt(0), s_barrier
//            (=> every wave's share has landed AND every wave has finished reading stage s)
//            64 MFMAs on set 1;  slots: 16 ds_read_b128 -> set 0 (K step 0 of tile kt+1, stage s^1),
//                                       16 DMA pieces: tile kt+2 -> stage s
// The DMA of tile kt+2 is issued during step B of tile kt and is waited for at step B of tile kt+1: one K tile
// (128 MFMAs = 2048 matrix-pipe cycles) of latency budget.  Fragment reads are raw ds_read_b128 (inline asm, not tracked by
// the compiler) retired by s_waitcnt lgkmcnt(0) at the step boundary, whose asm statement names the fragments as operands.
//
// Accumulators are TRANSPOSED (C^T = mfma(W, A)) and the W rows are permuted at DMA time exactly as in gemm8.hip, so the
// direct epilogues of gemm8_epi.h apply unchanged to each 128 x 64 half of the wave block; the v columns of the fused qkv
// projection run with the operands exchanged.  Persistent: one workgroup per CU walks the tile list grid-stride as ONE stream
// of K tiles (see the kernel): the last K iteration of a tile stages the next tile's first two K tiles.
#include "../gemm.h"
#include "../gemm_epilogue.h"
#include "../gemm8_epi.h"

namespace cva {

namespace {

using namespace epi;
using namespace g8;

constexpr int G4_NT = 256;

#define G4_SB() __builtin_amdgcn_sched_barrier(0)
#define G4_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// ---- slot operations (compile-time indices: the LDS offsets are instruction immediates) -------------------------------
// fragment read R of a K step: R = 0..7 -> W fragment R, R = 8..15 -> A fragment R-8; KS selects the 16-byte piece pair
template <int SET, int BUF, int R>
__device__ __forceinline__ void g4_read(half8_t (&FA)[2][8], half8_t (&FW)[2][8], const unsigned a_ad, const unsigned w_ad) {
    if constexpr (R < 8)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(FW[SET][R]) : "v"(w_ad), "n"(BUF * G8_TILE + R * 2048));
    else
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(FA[SET][R - 8]) : "v"(a_ad), "n"(BUF * G8_TILE + (R - 8) * 2048));
}

// DMA piece P of a K tile into stage BUF: P even -> A piece P/2, P odd -> W piece P/2 (8 rows = 1 KiB each)
template <int BUF, int P>
__device__ __forceinline__ void g4_dma(const unsigned (&a_voff)[8], const unsigned (&w_voff)[8], const unsigned char* a_base,
                                       const unsigned char* w_base, const unsigned lds_wave) {
    constexpr int i = P >> 1;
    if constexpr ((P & 1) == 0) {
        const unsigned dst = lds_wave + BUF * G8_TILE + i * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(a_voff[i]), "s"(a_base), "s"(dst) : "memory");
    } else {
        const unsigned dst = lds_wave + G8_WOFF + BUF * G8_TILE + i * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(w_voff[i]), "s"(w_base), "s"(dst) : "memory");
    }
}

// The MFMAs are inline asm with the accumulator as ONE tied AGPR operand: with the builtin the register allocator (all 256
// AGPRs live) rotated accumulator tuples through VGPR copies on the loop back edge — hundreds of v_accvgpr moves per K tile.
// What the compiler's hazard recogniser no longer sees is covered by hand: the accumulators are first read (epilogue) behind
// G4_MFMA_DRAIN; an accumulator is touched again 64 MFMAs later; operand fragments come from LDS behind counted waits.
#define G4_MM(SET, i, j)                                                                                               \
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[(j) >> 2][i][(j) & 3]) : "v"(FW[SET][j]), "v"(FA[SET][i]))
#define G4_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 7" ::: "memory")     // > the 11 wait states an 8-pass MFMA result needs

// wait until the LDS reads of fragment set SET have returned; the fragments are tied to the wait
#define G4_WAIT_SET(SET)                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                \
                 : "+v"(FA[SET][0]), "+v"(FA[SET][1]), "+v"(FA[SET][2]), "+v"(FA[SET][3]), "+v"(FA[SET][4]),           \
                   "+v"(FA[SET][5]), "+v"(FA[SET][6]), "+v"(FA[SET][7]), "+v"(FW[SET][0]), "+v"(FW[SET][1]),           \
                   "+v"(FW[SET][2]), "+v"(FW[SET][3]), "+v"(FW[SET][4]), "+v"(FW[SET][5]), "+v"(FW[SET][6]),           \
                   "+v"(FW[SET][7])                                                                                    \
                 :: "memory")

// One row (A fragment ROW x the 8 W fragments) of a K step: four MFMA pairs, each followed by one slot.
//   KIND 0 (step A): slots 0..15 of the step read set 1 from stage BUF.
//   KIND 1 (step B): reads of set 0 from stage BUF^1 and the DMA of the tile after next into stage BUF, placed by SCHED:
//     SCHED 0  slots 0..15 read, slots 16..31 DMA         SCHED 1  even slots read, odd slots DMA
//     SCHED 2  slots 0..15 read AND DMA                   SCHED 3  slots 0..15 DMA, slots 16..31 read
template <int KIND, int SCHED, int BUF, int ROW, int SLOT, bool RD, bool DMA>
__device__ __forceinline__ void g4_slot(half8_t (&FA)[2][8], half8_t (&FW)[2][8], const unsigned (&a_ad)[2], const unsigned (&w_ad)[2],
                                        const unsigned (&a_voff)[8], const unsigned (&w_voff)[8], const unsigned char* a_base,
                                        const unsigned char* w_base, const unsigned lds_wave, const unsigned char* pa_base,
                                        const unsigned char* pw_base) {
    constexpr int s = ROW * 4 + SLOT;
    if constexpr (KIND == 0) {
        if constexpr (s < 16 && RD) g4_read<1, BUF, s>(FA, FW, a_ad[1], w_ad[1]);
        // SCHED 4: the second half of the PREVIOUS step B's K tile (pieces 8..15, stage BUF ^ 1) is issued here, in slots 16..31
        if constexpr (SCHED == 4 && s >= 16 && (s & 1) == 0 && DMA) g4_dma<BUF ^ 1, 8 + ((s - 16) >> 1)>(a_voff, w_voff, pa_base, pw_base, lds_wave);
    } else if constexpr (SCHED == 4) {
        if constexpr ((s & 1) == 0 && RD) g4_read<0, BUF ^ 1, (s >> 1)>(FA, FW, a_ad[0], w_ad[0]);
        if constexpr ((s & 3) == 1 && DMA) g4_dma<BUF, (s >> 2)>(a_voff, w_voff, a_base, w_base, lds_wave);      // pieces 0..7
    } else {
        constexpr int r = SCHED == 0 ? (s < 16 ? s : -1) : SCHED == 1 ? ((s & 1) == 0 ? s >> 1 : -1) : SCHED == 2 ? (s < 16 ? s : -1)
                                                                                                    : (s >= 16 ? s - 16 : -1);
        constexpr int d = SCHED == 0 ? (s >= 16 ? s - 16 : -1) : SCHED == 1 ? ((s & 1) == 1 ? s >> 1 : -1) : SCHED == 2 ? (s < 16 ? s : -1)
                                                                                                     : (s < 16 ? s : -1);
        if constexpr (r >= 0 && RD) g4_read<0, BUF ^ 1, r>(FA, FW, a_ad[0], w_ad[0]);
        if constexpr (d >= 0 && DMA) g4_dma<BUF, d>(a_voff, w_voff, a_base, w_base, lds_wave);
    }
}

#define G4_ROW(KIND, SET, BUF, ROW, RD, DMA)                                                                                    \
    do {                                                                                                               \
        G4_MM(SET, ROW, 0); G4_MM(SET, ROW, 1);                                                                        \
        g4_slot<KIND, SCHED, BUF, ROW, 0, RD, DMA>(FA, FW, a_ad, w_ad, a_voff, w_voff, a_base, w_base, lds_wave, pa_base, pw_base); G4_SB(); \
        G4_MM(SET, ROW, 2); G4_MM(SET, ROW, 3);                                                                        \
        g4_slot<KIND, SCHED, BUF, ROW, 1, RD, DMA>(FA, FW, a_ad, w_ad, a_voff, w_voff, a_base, w_base, lds_wave, pa_base, pw_base); G4_SB(); \
        G4_MM(SET, ROW, 4); G4_MM(SET, ROW, 5);                                                                        \
        g4_slot<KIND, SCHED, BUF, ROW, 2, RD, DMA>(FA, FW, a_ad, w_ad, a_voff, w_voff, a_base, w_base, lds_wave, pa_base, pw_base); G4_SB(); \
        G4_MM(SET, ROW, 6); G4_MM(SET, ROW, 7);                                                                        \
        g4_slot<KIND, SCHED, BUF, ROW, 3, RD, DMA>(FA, FW, a_ad, w_ad, a_voff, w_voff, a_base, w_base, lds_wave, pa_base, pw_base); G4_SB(); \
    } while (0)

#define G4_STEP(KIND, SET, BUF, RD, DMA)                                                                                        \
    do {                                                                                                               \
        G4_ROW(KIND, SET, BUF, 0, RD, DMA); G4_ROW(KIND, SET, BUF, 1, RD, DMA); G4_ROW(KIND, SET, BUF, 2, RD, DMA); G4_ROW(KIND, SET, BUF, 3, RD, DMA);    \
        G4_ROW(KIND, SET, BUF, 4, RD, DMA); G4_ROW(KIND, SET, BUF, 5, RD, DMA); G4_ROW(KIND, SET, BUF, 6, RD, DMA); G4_ROW(KIND, SET, BUF, 7, RD, DMA);    \
    } while (0)

template <int OMODE, int SCHED, int ABL>
__global__ __launch_bounds__(G4_NT) void gemm4_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem4[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = p.N / G8_BN, tiles_m = p.M / G8_BM, ntiles = tiles_m * tiles_n;
    const int nk = p.K / G8_BK;                     // even, >= 2 (host)

    // ---- DMA state: always describes the output tile whose K tiles are staged NEXT
    const int lrow = lane >> 3, lpc = lane & 7;
    auto a_row = [&](int m) -> long {
        long r = m;
        if (p.a_rpi > 0) r = (long)m + (long)(m / p.a_rpi) * p.a_extra + p.a_off;
        return r;
    };
    unsigned a_voff[8], w_voff[8];
    const unsigned char* Ab;
    const unsigned char* Wb;
    int kstart = 0, kstep = 1;
    auto tile_setup = [&](int tile, int& m0, int& n0, bool& swap) {
        int tm, tn;
        tile_coords(xcd_remap(tile, ntiles), tiles_m, tiles_n, tm, tn);
        m0 = tm * G8_BM; n0 = tn * G8_BN;
        // consecutive tiles of a workgroup share their A panel: every other tile walks K backwards (its latest slices are in L2)
        const bool rev = (tile / (int)gridDim.x) & 1;
        kstart = rev ? nk - 1 : 0; kstep = rev ? -1 : 1;
        swap = OMODE == OUT_QKV && !p.v_rm && (n0 + p.n_off) >= 2 * p.D;          // v columns: operands exchanged (block-uniform)
        const long ar0 = a_row(m0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = wave * 64 + i * 8 + lrow;
            const int lp = lpc ^ ((row >> 1) & 7);
            // LDS row wq*64 + j*16 + g*4 + r of the "W" tile holds source row wq*64 + g*16 + j*4 + r: a lane's 16 accumulator
            // values per output row are then 16 consecutive columns (epilogue8_direct)
            const int prow = (row & ~63) | (((row >> 2) & 3) << 4) | (((row >> 4) & 3) << 2) | (row & 3);
            if (!swap) {
                a_voff[i] = (unsigned)((a_row(m0 + row) - ar0) * (long)p.lda * 2) + lp * 16;
                w_voff[i] = (unsigned)((long)prow * p.ldw * 2) + lp * 16;
            } else {
                a_voff[i] = (unsigned)((long)row * p.ldw * 2) + lp * 16;
                w_voff[i] = (unsigned)((a_row(m0 + prow) - ar0) * (long)p.lda * 2) + lp * 16;
            }
        }
        const unsigned char* abase = reinterpret_cast<const unsigned char*>(p.A) + ar0 * (long)p.lda * 2;
        const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.W) + (long)n0 * p.ldw * 2;
        Ab = swap ? wbase : abase;
        Wb = swap ? abase : wbase;
    };

    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem4;
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds0 + wave * 8192);      // this wave's 64 rows of a stage
    auto k_base = [&](const unsigned char* b, int kt) { return uniform_ptr(b + (long)(kstart + kt * kstep) * (G8_BK * 2)); };
    // wave 0 fetches a tile's 256 bias values into LDS slot `slot` (an LDS-DMA like the operands: covered by the same waits)
    auto stage_bias = [&](int n0_, int slot) {
        if (p.bias && wave == 0) {
            const unsigned char* src = uniform_ptr(reinterpret_cast<const unsigned char*>(p.bias + n0_));
            const unsigned boff = lane * 16;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + G8_BIAS + slot * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(boff), "s"(src), "s"(dst) : "memory");
        }
    };

    // ---- fragment read addresses: row r (r & 15 == lane & 15), logical piece ks*4 + g -> byte r*128 + ((piece ^ ((r>>1)&7)) << 4)
    const int g = lane >> 4, li = lane & 15;
    const int off0 = li * 128 + ((g ^ ((li >> 1) & 7)) << 4);
    const int d1 = 64 - 2 * (off0 & 64);                          // second K step: piece g + 4 = offset ^ 64
    unsigned a_ad[2], w_ad[2];
    a_ad[0] = lds0 + (wr * 128) * 128 + off0; a_ad[1] = a_ad[0] + d1;                 // + stage*32K + i*2K
    w_ad[0] = lds0 + G8_WOFF + (wc * 128) * 128 + off0; w_ad[1] = w_ad[0] + d1;       // + stage*32K + j*2K

    if ((OMODE == OUT_LINEAR) && p.act == ACT_GELU) {             // (read only in epilogues: many barriers later)
        gelu_fill_lut(reinterpret_cast<float*>(smem4 + G8_LUT), threadIdx.x, G4_NT);
    }

    // ---- the K-tile stream.  The workgroup's output tiles form ONE sequence of K tiles: the last K iteration of an output
    // tile stages K tiles 0 and 1 of the NEXT output tile and reads its first fragments, so the loop body below is the only
    // code that contains MFMAs, fragment reads or operand DMA — no prologue per tile, no peeled tail, and the next tile's
    // operands are in flight during the epilogue.  (After the workgroup's last tile the "next" tile is that tile again: two
    // K tiles and 16 fragments are fetched into nothing, once per workgroup.)
    half8_t FA[2][8], FW[2][8];
    const unsigned char* pa_carry = nullptr; const unsigned char* pw_carry = nullptr;   // SCHED 4: bases of the K tile the previous step B began
    int m0, n0; bool swap;
    int tile = blockIdx.x;
    int slot = 0;
    tile_setup(tile, m0, n0, swap);
    stage_bias(n0, slot);
    {
        const unsigned char* a_base = k_base(Ab, 0); const unsigned char* w_base = k_base(Wb, 0);
#define G4_D(P) g4_dma<0, P>(a_voff, w_voff, a_base, w_base, lds_wave)
        G4_D(0); G4_D(1); G4_D(2); G4_D(3); G4_D(4); G4_D(5); G4_D(6); G4_D(7);
        G4_D(8); G4_D(9); G4_D(10); G4_D(11); G4_D(12); G4_D(13); G4_D(14); G4_D(15);
#undef G4_D
    }
    {
        const unsigned char* a_base = k_base(Ab, 1); const unsigned char* w_base = k_base(Wb, 1);
#define G4_D(P) g4_dma<1, P>(a_voff, w_voff, a_base, w_base, lds_wave)
        G4_D(0); G4_D(1); G4_D(2); G4_D(3); G4_D(4); G4_D(5); G4_D(6); G4_D(7);
        G4_D(8); G4_D(9); G4_D(10); G4_D(11); G4_D(12); G4_D(13); G4_D(14); G4_D(15);
#undef G4_D
    }
    pa_carry = k_base(Ab, 1); pw_carry = k_base(Wb, 1);   // (SCHED 4 re-issues the second half of K tile 1 in its first step A: harmless)
    G4_VMCNT(16);                                   // K tile 0 (and the bias) has landed; tile 1 may still be in flight
    G4_SB(); __builtin_amdgcn_s_barrier(); G4_SB();
    for (; tile < ntiles; tile += gridDim.x, slot ^= 1) {
        const int em0 = m0, en0 = n0; const bool eswap = swap;        // the tile being accumulated (m0 / n0 / swap move on to the next)
        const bool has_next = tile + (int)gridDim.x < ntiles;
        f32x4 acc[2][8][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4)(0.f);
        // fragments of K step 0 (stage 0 holds this tile's K tile 0: staged by the previous tile's last iteration, or above).  The
        // uniform loop body already fetched them at the end of the previous tile; they are fetched AGAIN here so that no fragment
        // is live across the epilogue (64 VGPRs the epilogue needs: with them live the compiler spilled accumulators to scratch).
        {
#define G4_R0(R) g4_read<0, 0, R>(FA, FW, a_ad[0], w_ad[0])
            G4_R0(0); G4_R0(1); G4_R0(2); G4_R0(3); G4_R0(4); G4_R0(5); G4_R0(6); G4_R0(7);
            G4_R0(8); G4_R0(9); G4_R0(10); G4_R0(11); G4_R0(12); G4_R0(13); G4_R0(14); G4_R0(15);
#undef G4_R0
        }

// one K tile in stage BUF: step A on set 0, the barrier, step B on set 1 (a_base / w_base: the K tile step B stages)
#define G4_TILE(BUF, KT2)                                                                                              \
    do {                                                                                                               \
        const unsigned char* a_base = k_base(Ab, KT2);                                                                 \
        const unsigned char* w_base = k_base(Wb, KT2);                                                                 \
        const unsigned char* pa_base = pa_carry; const unsigned char* pw_base = pw_carry;                              \
        pa_carry = a_base; pw_carry = w_base;                                                                          \
        G4_WAIT_SET(0); G4_SB();                                                                                       \
        G4_STEP(0, 0, BUF, !(ABL & 2), !(ABL & 1));                                                                    \
        G4_VMCNT(0);                                                                                                   \
        G4_WAIT_SET(1); G4_SB();                                                                                       \
        __builtin_amdgcn_s_barrier(); G4_SB();                                                                         \
        G4_STEP(1, 1, BUF, !(ABL & 2), !(ABL & 1));                                                                    \
    } while (0)
        for (int kt = 0; kt < nk; kt += 2) {
            int kn = kt + 2;                        // K tile index (of the tile the DMA state describes) staged by this iteration
            if (kn >= nk) {                         // last iteration of this output tile: the stream moves on to the next tile
                tile_setup(has_next ? tile + (int)gridDim.x : tile, m0, n0, swap);
                if (has_next) stage_bias(n0, slot ^ 1);
                kn = 0;
            }
            asm volatile("s_nop 3" ::: "memory");   // (accumulators zeroed by VALU writes just before the first iteration)
            G4_TILE(0, kn);
            G4_TILE(1, kn + 1);
            G4_MFMA_DRAIN();                        // whatever follows (back edge, epilogue, compiler copies) may read an accumulator
        }
#undef G4_TILE

        float bv[2][16];
        {
            const float* bs = reinterpret_cast<const float*>(smem4 + G8_BIAS + slot * 1024);
            if (!eswap) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 b4 = (f32x4)(0.f);
                        if (p.bias) b4 = *reinterpret_cast<const f32x4*>(bs + wc * 128 + h * 64 + g * 16 + q * 4);
                        bv[h][q * 4 + 0] = b4[0]; bv[h][q * 4 + 1] = b4[1]; bv[h][q * 4 + 2] = b4[2]; bv[h][q * 4 + 3] = b4[3];
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    bv[0][i] = p.bias ? bs[wr * 128 + i * 16 + li] : 0.f; bv[0][8 + i] = 0.f;
                    bv[1][i] = bv[0][i]; bv[1][8 + i] = 0.f;
                }
            }
        }
        const float* lut = reinterpret_cast<const float*>(smem4 + G8_LUT);
        if constexpr (ABL & 4) {      // experiment: keep the accumulators live, one store per lane
            float t = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) t += acc[h][i][j][0] + acc[h][i][j][1] + acc[h][i][j][2] + acc[h][i][j][3];
            reinterpret_cast<half_t*>(p.out)[(long)(em0 + wr * 128 + g) * p.ldc + en0 + wc * 128 + li] = (half_t)(t + bv[0][0] + lut[0]);
        } else if (eswap) {
            epilogue8_vt(p, acc[0], bv[0], en0 + wr * 128, em0 + wc * 128, lane);
            epilogue8_vt(p, acc[1], bv[1], en0 + wr * 128, em0 + wc * 128 + 64, lane);
        } else {
            epilogue8_direct<OMODE>(p, acc[0], bv[0], em0 + wr * 128, en0 + wc * 128, lane, lut);
            epilogue8_direct<OMODE>(p, acc[1], bv[1], em0 + wr * 128, en0 + wc * 128 + 64, lane, lut);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the stray fetches behind the last tile
}

template <int OMODE, int SCHED, int ABL = 0>
int launch4(const GemmParams& p, hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel<OMODE, SCHED, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
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
    const int grid = tiles < n_cu ? tiles : n_cu;          // persistent: one workgroup per CU walks tiles grid-stride
    hipLaunchKernelGGL((gemm4_kernel<OMODE, SCHED, ABL>), dim3(grid), dim3(G4_NT), G8_LDS, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

#ifndef CVA_ABLATION
#error "experiments/gemm4.hip belongs to the experiment flavour of the library only (python -m cellvit_amd.build --ablation)"
#else
// Shapes: those of gemm8_supported (the caller checks it); out modes OUT_LINEAR, OUT_QKV and OUT_CONVT with 32-byte runs.
bool gemm4_takes(const GemmParams& p) {
    if (p.out_mode == OUT_LINEAR || p.out_mode == OUT_QKV) return true;
    return p.out_mode == OUT_CONVT && (p.N >> 2) % 16 == 0 && !p.out_f32;
}

int launch_gemm4(const GemmParams& p, int sched, hipStream_t stream) {
#ifdef CVA_ABLATION      // slot-placement variants exist in ablation builds only (timing A/B)
    if (p.out_mode == OUT_LINEAR) {
        switch (sched) {
            case 0: return launch4<OUT_LINEAR, 0>(p, stream);
            case 2: return launch4<OUT_LINEAR, 2>(p, stream);
            case 3: return launch4<OUT_LINEAR, 3>(p, stream);
            case 4: return launch4<OUT_LINEAR, 4>(p, stream);      // DMA spread over both K steps (linear shapes without row remap only)
            case 21: return launch4<OUT_LINEAR, 1, 1>(p, stream);      // work-skipping (wrong results): no operand DMA in the loop,
            case 22: return launch4<OUT_LINEAR, 1, 2>(p, stream);      // no fragment reads,
            case 23: return launch4<OUT_LINEAR, 1, 3>(p, stream);      // neither,
            case 24: return launch4<OUT_LINEAR, 1, 4>(p, stream);      // no epilogue,
            case 27: return launch4<OUT_LINEAR, 1, 7>(p, stream);      // MFMAs + barriers only
            default: break;
        }
    }
#endif
    (void)sched;
    if (p.out_mode == OUT_LINEAR) return launch4<OUT_LINEAR, 1>(p, stream);
    if (p.out_mode == OUT_QKV) return launch4<OUT_QKV, 1>(p, stream);
    return launch4<OUT_CONVT, 1>(p, stream);
}

#endif

}  // namespace cva
