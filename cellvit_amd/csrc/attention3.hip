// Attention v3 for gfx950 (fp16, global blocks): the transposed flash attention of attention2.hip re-organised around the two
// pipes of a SIMD.  A key tile costs a wave 44 MFMAs (704 matrix-pipe cycles) and ~1000 VALU cycles of softmax; in attention2 the
// two waves of a SIMD (from two independent 4-wave workgroups) ran both in sequence with nothing to keep them apart, and a SIMD
// spent ~4850 cycles per key tile for 1408 cycles of MFMA work.  Here:
//
//   * ONE 8-wave workgroup per CU, 256 queries (32 per wave) of one (sequence, head); waves 0-3 (group A) and 4-7 (group B) share
//     each SIMD pairwise and run the same loop ONE BARRIER APART:
//         iteration t:   M(t): O^T += V^T(t-1) P^T(t-1);  S^T(t) = K(t) Q^T      (MFMA only, raised priority)
//                        barrier
//                        V(t): online softmax of S^T(t) -> P^T(t)                  (VALU only)
//                        barrier
//     so while group A is in M, group B is in V and vice versa — each pipe always has exactly one wave of the SIMD asking for it.
//   * K and V^T tiles arrive by LDS-DMA into two slots each, issued a full iteration ahead by the group that has just passed an
//     "even" barrier (below); no register staging, no LDS stores, no second barrier for the staging.
//         even barrier E(t) (global order h = 2t): K(t), V^T(t-1) have landed; issue K(t+1), V^T(t).
//         group A meets E(t) right before M(t); group B, one barrier behind, right before V(t-1).
//     K(t) and V^T(t-1) are read by A in h = 2t and by B in h = 2t + 1; their slots are re-issued at E(t+1), h = 2t + 2.
//   * K rows keep the padded pitch of attention2 (conflict-free ds_read_b128); pad columns and keys >= nk are written as ZEROS by
//     the DMA itself (buffer addressing, offsets beyond num_records).  V^T rows are 128 bytes (64 keys) with the 16-byte pieces
//     XOR-swizzled by (row >> 1) & 7: the 8-byte fragment halves of 32 lanes then cover all 64 banks exactly once.
//
// Same arithmetic as attention2 (S^T = K Q^T, P^T straight from the S^T registers, lazy running maximum, decomposed rel-pos with the
// kw term in registers and one kh scalar per key tile): SAM/image_encoder.py:235-257, 354-392; vits_histo.py:174-185.
//
// MEASURED (profiles/r03_exp_attention3.txt): bit-identical results, but NOT faster than attention2 (2.48 vs 2.40 ms per 16-tile launch).
// The ablation instantiations below say why: a wave issuing only MFMAs and a wave issuing only softmax VALU work on the same SIMD take
// as long TOGETHER (1070 us) as one after the other (540 + 590 us) — on this part the two pipes do not overlap across waves at these
// issue rates, so the counter-phase buys nothing and the kernel's time is the SUM of its MFMA and VALU issue cycles either way.
// Production therefore stays on attention2; this file is compiled into the ablation flavour only (CVA_ATTN=5 selects it).
#include "attention.h"

#ifndef CVA_ABLATION
namespace cva { int launch_attention3(const AttnParams&, hipStream_t) { return -1; } }
#else

namespace cva {

namespace {

constexpr int KT3 = 64, NT3 = 512, QW3 = 32, QT3 = 256;
constexpr float LOG2E3 = 1.4426950408889634f;
constexpr int RC3 = 128 + 8;                       // relcat row pitch (halves): KH + KW <= 128

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

#define A3_BAR() asm volatile("s_barrier" ::: "memory")
#define A3_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// LDS-DMA (inline asm pins the addressing forms; M0 = LDS byte address of the 1-KiB destination, wave-uniform)
#define A3_DMA(voff, base, ldsaddr)                                                                       \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), \
                 "s"(ldsaddr) : "memory")
#define A3_BDMA(voff, desc, soff, ldsaddr)                                                                   \
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(desc), \
                 "s"(soff), "s"(ldsaddr) : "memory")

template <int HD, int BIAS>      // BIAS 0: none; 2: decomposed rel-pos with KW == 64 == key tile
__global__ __launch_bounds__(NT3, 1) void attn3_kernel(const AttnParams p) {
    using TR = Traits<half_t>;
    using Frag = TR::Frag;
    constexpr int HDP = (HD + 31) / 32 * 32, NKS = HDP / 32, ND = HD / 16;
    constexpr int PK = lds_pitch<half_t>(HDP), PKP = PK / 8;          // K row pitch in halves / in 16-byte pieces
    constexpr int KSLOT = KT3 * PK * 2, VSLOT = HD * 128;            // bytes per slot
    constexpr int NKI = KT3 * PKP / 64, NVI = HD * 8 / 64;           // DMA instructions (64 lanes x 16 B) per tile
    static_assert(NKI <= 16 && NVI <= 16, "two DMA instructions per wave and operand");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    half_t* Ks = reinterpret_cast<half_t*>(smem3);                              // [2][KT3][PK]
    unsigned char* Vs = smem3 + 2 * KSLOT;                                      // [2][HD][128 B], swizzled pieces
    half_t* Rc = reinterpret_cast<half_t*>(smem3 + 2 * KSLOT + 2 * VSLOT);      // BIAS 2: [QT3][RC3]
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem3;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // timing experiments (wrong results): 1 no softmax, 2 no MFMA phases, 4 both groups in the same phase, 8 no DMA in the loop, 16 no rel-pos prologue, 32 no main loop,
                         // 64 waves 0-3 run only the MFMA phases, waves 4-7 only the softmax (do the two pipes of a SIMD overlap across waves?)
    const int dbg = p.dbg;
    const int grp = (dbg & 4) ? 0 : wave >> 2;
    const int g = lane >> 4, li = lane & 15;
    const int lin = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int qblk = lin % gridDim.x;
    const int q0 = qblk * QT3 + wave * QW3;
    const int sh = lin / gridDim.x;

    const half_t* __restrict__ Qg = reinterpret_cast<const half_t*>(p.Q) + (long)sh * p.L * HD;
    const half_t* __restrict__ Kg = reinterpret_cast<const half_t*>(p.K) + (long)sh * p.L * HD;
    const half_t* __restrict__ Vg = reinterpret_cast<const half_t*>(p.Vt) + (long)sh * HD * p.Lp;
    const int ntiles = (p.nk + KT3 - 1) / KT3;

    // ---- DMA lane maps.  Instruction u (wave w issues u = w and u = w + 8) fills LDS pieces u*64 .. u*64+63 of the slot.
    unsigned kvo[2], vvo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int u = wave + j * 8;
        const int q = u * 64 + lane;
        const int r = q / PKP, c = q - r * PKP;                    // K: row (key of the tile), 16-byte column
        kvo[j] = (u < NKI && c < HD / 8) ? (unsigned)((r * HD + c * 8) * 2) : 0x80000000u;      // pad columns: zeros
        const int d = q >> 3, jp = q & 7;                          // V^T: row d, physical piece jp holds source piece jp ^ sw(d)
        vvo[j] = (unsigned)(((long)d * p.Lp + ((jp ^ ((d >> 1) & 7)) * 8)) * 2);
    }
    i32x4_t kdesc;
    {
        const unsigned long long b = (unsigned long long)Kg;
        kdesc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        kdesc[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));
        kdesc[2] = p.nk * HD * 2;                                   // keys >= nk: zeros
        kdesc[3] = 0x00020000;
    }
    auto issue_k = [&](int t) {
        if (t >= ntiles) return;
        const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(t * KT3 * HD * 2));
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (t & 1) * KSLOT + wave * 1024);
        A3_BDMA(kvo[0], kdesc, soff, dst);
        if (wave + 8 < NKI) A3_BDMA(kvo[1], kdesc, soff, dst + 8 * 1024);
    };
    auto issue_v = [&](int t) {
        if (t >= ntiles) return;
        const unsigned long long b = (unsigned long long)(Vg + (long)t * KT3);
        const unsigned long long base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                        (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + 2 * KSLOT + (t & 1) * VSLOT + wave * 1024);
        if (wave < NVI) A3_DMA(vvo[0], base, dst);
        if (wave + 8 < NVI) A3_DMA(vvo[1], base, dst + 8 * 1024);
    };
    issue_k(0);

    // ---- Q fragments (B operand of S^T): lane -> query li of block qb, head-dim slice g*8.. ----
    Frag qf[2][NKS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int row = q0 + qb * 16 + li;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d0 = ks * 32 + g * 8;
            qf[qb][ks] = (row < p.L && d0 < HD) ? TR::load_frag(Qg + (long)row * HD + d0) : TR::zero_frag();
        }
    }

    // ---- decomposed rel-pos: relcat[q][kh] = q . tab_h[qy - kh + KH - 1] / scale ; relcat[q][KH + kw] likewise ----
    const float inv_scale = 1.0f / p.scale;
    if (BIAS == 2 && !(dbg & 16)) {
        half_t* myrc = Rc + (wave * QW3) * RC3;
        for (int i = lane; i < QW3 * RC3 / 8; i += 64) store_piece(myrc + i * 8, zero_piece());
#pragma unroll 1
        for (int tbl = 0; tbl < 2; ++tbl) {
            const float* __restrict__ tab = tbl == 0 ? p.tab_h : p.tab_w;
            const int Ksz = tbl == 0 ? p.KH : p.KW;
            const int off = tbl == 0 ? 0 : p.KH;
            const int nj = 2 * Ksz - 1;
#pragma unroll 1
            for (int jb = 0; jb * 16 < nj; ++jb) {
                Frag tf[NKS];
                const int j = jb * 16 + li;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const int d0 = ks * 32 + g * 8;
                    tf[ks] = TR::zero_frag();
                    if (j < nj && d0 < HD) {
                        const f32x4 t0 = *reinterpret_cast<const f32x4*>(tab + (long)j * HD + d0);
                        const f32x4 t1 = *reinterpret_cast<const f32x4*>(tab + (long)j * HD + d0 + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { tf[ks].v[e] = (half_t)t0[e]; tf[ks].v[4 + e] = (half_t)t1[e]; }
                    }
                }
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    f32x4 acc = (f32x4)(0.f);
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) TR::mma(tf[ks], qf[qb][ks], acc);
                    // acc[r] = q(li) . tab[jb*16 + g*4 + r]
                    const int q = q0 + qb * 16 + li;
                    if (q < p.L) {
                        const int qy = q / p.KW, qx = q - qy * p.KW;
                        const int c = tbl == 0 ? qy : qx;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int jj = jb * 16 + g * 4 + r;
                            const int kk = c - jj + Ksz - 1;          // image_encoder.py:347-351
                            if (jj < nj && kk >= 0 && kk < Ksz) myrc[(qb * 16 + li) * RC3 + off + kk] = (half_t)(acc[r] * inv_scale);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();          // (the relcat rows of a wave are its own, but the barrier also orders them against the reads below)

    const float c1 = p.scale * LOG2E3;                 // relcat holds bias / scale, so bias * log2e = relcat * c1
    f32x4 bw[2][4];                                    // BIAS 2: tile-invariant kw terms
    if (BIAS == 2) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    bw[qb][kb][r] = (float)Rc[(wave * QW3 + qb * 16 + li) * RC3 + p.KH + kb * 16 + g * 4 + r] * c1;
    }

    f32x4 o[2][ND];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int n = 0; n < ND; ++n) o[qb][n] = (f32x4)(0.f);
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    f32x4 s[2][4];
    Frag pf[2][2];
    const bool wave_active = q0 < p.L;                 // waves whose 32 queries are all padding only help staging

    // V^T fragment halves: row d = n*16 + li, logical 8-byte halves (2m)*4 + g and (2m+1)*4 + g of the 16 of a row
    const int vsw = (li >> 1) & 7;                     // (d >> 1) & 7 with d = n*16 + li
    const unsigned v_row = (unsigned)li * 128u + (unsigned)(g & 1) * 8u;
    auto v_off = [&](int n, int m, int half) -> unsigned {      // byte offset inside a slot
        const int piece = (((2 * m + half) * 4 + g) >> 1) ^ vsw;
        return (unsigned)n * 2048u + v_row + (unsigned)piece * 16u;
    };

    auto pv = [&](int t, int nkb) {                    // O^T += V^T(t) P^T(t)
        const unsigned char* vs = Vs + (t & 1) * VSLOT;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (2 * m >= nkb) continue;                // all P of these 32 keys are exactly 0
#pragma unroll
            for (int n = 0; n < ND; ++n) {
                const half4_t a = *reinterpret_cast<const half4_t*>(vs + v_off(n, m, 0));
                const half4_t b = *reinterpret_cast<const half4_t*>(vs + v_off(n, m, 1));
                Frag vf;
                vf.v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
                TR::mma(vf, pf[0][m], o[0][n]);
                TR::mma(vf, pf[1][m], o[1][n]);
            }
        }
    };
    auto qk = [&](int t, int nkb) {                    // S^T(t) = K(t) Q^T  (k-step outermost: dependent MFMAs are 8 apart)
        const half_t* ks_ = Ks + (t & 1) * (KT3 * PK);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) { s[0][kb] = (f32x4)(0.f); s[1][kb] = (f32x4)(0.f); }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                if (kb >= nkb) continue;               // wave-uniform: masked to -inf in the softmax
                const Frag kf = TR::load_frag(ks_ + (kb * 16 + li) * PK + ks * 32 + g * 8);
                TR::mma(kf, qf[0][ks], s[0][kb]);
                TR::mma(kf, qf[1][ks], s[1][kb]);
            }
        }
    };
    // ---- M phase of a full tile pair, software-pipelined by hand.  In counter-phase the wave runs its MFMAs ALONE on the SIMD (its
    // partner is in the softmax), so nothing hides a fragment read the compiler placed right in front of its consumer — measured: a
    // lone wave kept the matrix pipe 43 % busy with compiler-scheduled reads.  The reads are raw ds_read instructions issued one group
    // (10 PV MFMAs / 8 QK MFMAs) ahead, with counted waits that name the fragments they release:
    //     V(m=0) V(m=1) | wait V0 | PV m=0 | K(ks=0) | wait V1 | PV m=1 | K(ks=1) | wait K0 | QK ks=0 | K(ks=2) | wait K1 | QK ks=1 | wait K2 | QK ks=2
    const unsigned k_lane = lds0 + (unsigned)(li * PK + g * 8) * 2u;                               // + slot + kb*16*PK*2 + ks*64
    unsigned v_lane[2][2];                                                                          // + slot + n*2048
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) v_lane[m][hf] = lds0 + 2 * KSLOT + v_off(0, m, hf);
#define A3_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define A3_DSR64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define A3_RD_V(M)                                                                      \
    do {                                                                                \
        _Pragma("unroll") for (int n_ = 0; n_ < ND; ++n_) {                             \
            if (n_ == 0) { A3_DSR64(vlo[M][0], va[M][0], 0 * 2048); A3_DSR64(vhi[M][0], va[M][1], 0 * 2048); }      \
            if (n_ == 1) { A3_DSR64(vlo[M][1], va[M][0], 1 * 2048); A3_DSR64(vhi[M][1], va[M][1], 1 * 2048); }      \
            if (n_ == 2) { A3_DSR64(vlo[M][2], va[M][0], 2 * 2048); A3_DSR64(vhi[M][2], va[M][1], 2 * 2048); }      \
            if (n_ == 3) { A3_DSR64(vlo[M][3], va[M][0], 3 * 2048); A3_DSR64(vhi[M][3], va[M][1], 3 * 2048); }      \
            if (n_ == 4) { A3_DSR64(vlo[M][4], va[M][0], 4 * 2048); A3_DSR64(vhi[M][4], va[M][1], 4 * 2048); }      \
        }                                                                               \
    } while (0)
#define A3_RD_K(KS)                                                                     \
    do {                                                                                \
        A3_DSR128(kfr[KS][0].v, ka, 0 * 16 * PK * 2 + (KS) * 64); A3_DSR128(kfr[KS][1].v, ka, 1 * 16 * PK * 2 + (KS) * 64); \
        A3_DSR128(kfr[KS][2].v, ka, 2 * 16 * PK * 2 + (KS) * 64); A3_DSR128(kfr[KS][3].v, ka, 3 * 16 * PK * 2 + (KS) * 64); \
    } while (0)
#define A3_WAIT_V(N, M)                                                                                                        \
    do {                                                                                                                       \
        if (ND == 5) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(vlo[M][0]), "+v"(vlo[M][1]), "+v"(vlo[M][2]), "+v"(vlo[M][3]), "+v"(vlo[M][ND - 1]), \
                                   "+v"(vhi[M][0]), "+v"(vhi[M][1]), "+v"(vhi[M][2]), "+v"(vhi[M][3]), "+v"(vhi[M][ND - 1]));                     \
        else asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(vlo[M][0]), "+v"(vlo[M][1]), "+v"(vlo[M][2]), "+v"(vlo[M][3]),                        \
                          "+v"(vhi[M][0]), "+v"(vhi[M][1]), "+v"(vhi[M][2]), "+v"(vhi[M][3]));                                                    \
    } while (0)
#define A3_WAIT_K(N, KS) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(kfr[KS][0].v), "+v"(kfr[KS][1].v), "+v"(kfr[KS][2].v), "+v"(kfr[KS][3].v))
    auto m_phase = [&](int t, bool with_pv) {
        half4_t vlo[2][5], vhi[2][5];
        Frag kfr[3][4];
        const unsigned ka = k_lane + (unsigned)(t & 1) * KSLOT;
        unsigned va[2][2];
        const unsigned vslot = (unsigned)((t + 1) & 1) * VSLOT;           // V^T(t - 1)
#pragma unroll
        for (int m = 0; m < 2; ++m) { va[m][0] = v_lane[m][0] + vslot; va[m][1] = v_lane[m][1] + vslot; }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) { s[0][kb] = (f32x4)(0.f); s[1][kb] = (f32x4)(0.f); }
        auto pv_m = [&](int m) {
#pragma unroll
            for (int n = 0; n < ND; ++n) {
                Frag vf;
                vf.v = __builtin_shufflevector(vlo[m][n], vhi[m][n], 0, 1, 2, 3, 4, 5, 6, 7);
                TR::mma(vf, pf[0][m], o[0][n]);
                TR::mma(vf, pf[1][m], o[1][n]);
            }
        };
        auto qk_ks = [&](int ks) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) { TR::mma(kfr[ks][kb], qf[0][ks], s[0][kb]); TR::mma(kfr[ks][kb], qf[1][ks], s[1][kb]); }
        };
        if (with_pv) {
            A3_RD_V(0); A3_RD_V(1);
            if (ND == 5) A3_WAIT_V(10, 0); else A3_WAIT_V(8, 0);
            pv_m(0);
            A3_RD_K(0);
            A3_WAIT_V(4, 1);
            pv_m(1);
        } else {
            A3_RD_K(0);
        }
        A3_RD_K(1);
        A3_WAIT_K(4, 0);
        qk_ks(0);
        if (NKS == 3) {
            A3_RD_K(NKS - 1);
            A3_WAIT_K(4, 1);
            qk_ks(1);
            A3_WAIT_K(0, NKS - 1);
            qk_ks(NKS - 1);
        } else {
            A3_WAIT_K(0, 1);
            qk_ks(1);
        }
    };

    auto nkb_of = [&](int t) -> int { return t + 1 < ntiles ? 4 : min(4, (p.nk - t * KT3 + 15) / 16); };

    // ---- online softmax per query, LAZY reference maximum (log2 domain), as attention2; the element-wise part in packed fp32
    // (v_pk_fma / v_pk_add: two elements per VALU slot) — the softmax is the longer of the two phases ----
    auto softmax = [&](int t) {
        constexpr float LAZY_TAU = 8.0f;
        const bool ragged = (t + 1 == ntiles) && (p.nk & (KT3 - 1)) != 0;     // only the last tile can hold invalid keys
        float tmax[2], bh[2];
        const f32x4 c4 = {c1, c1, c1, c1};
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            bh[qb] = 0.f;
            if (BIAS == 2) bh[qb] = (float)Rc[(wave * QW3 + qb * 16 + li) * RC3 + t] * c1;      // kh == key tile
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                f32x4 v = BIAS == 2 ? __builtin_elementwise_fma(s[qb][kb], c4, bw[qb][kb]) : s[qb][kb] * c4;
                if (ragged) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const int key = t * KT3 + kb * 16 + g * 4 + r; v[r] = key < p.nk ? v[r] : -INFINITY; }
                }
                s[qb][kb] = v;
                mx = fmaxf(fmaxf(mx, v[0]), v[1]);
                mx = fmaxf(fmaxf(mx, v[2]), v[3]);
            }
            tmax[qb] = mx + bh[qb];
        }
        if (__any((tmax[0] > m_run[0] + LAZY_TAU) || (tmax[1] > m_run[1] + LAZY_TAU))) {      // wave-uniform, rare after the first tiles
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float mx = tmax[qb];
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float mn = fmaxf(m_run[qb], mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - mn);   // first tile: 2^(-inf) = 0 on zero accumulators
                m_run[qb] = mn;
                l_run[qb] *= alpha;
#pragma unroll
                for (int n = 0; n < ND; ++n) o[qb][n] *= alpha;
            }
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float shift = bh[qb] - m_run[qb];
            const f32x4 sh4 = {shift, shift, shift, shift};
            f32x4 rs = (f32x4)(0.f);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const f32x4 a = s[qb][kb] + sh4;
                f32x4 e;
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(a[r]);
                rs += e;
                // contraction slot (m, g*8 + j): j < 4 -> key block 2m, reg j ; j >= 4 -> key block 2m+1, reg j-4
                const half4_t h4 = __builtin_convertvector(e, half4_t);
#pragma unroll
                for (int r = 0; r < 4; ++r) pf[qb][kb >> 1].v[(kb & 1) * 4 + r] = h4[r];
            }
            l_run[qb] += (rs[0] + rs[1]) + (rs[2] + rs[3]);                 // per-lane partial row sum
        }
    };

    // ---- E(0): K(0) has landed; issue K(1), V^T(0).  Group B then drops one barrier behind group A. ----
    A3_VMCNT0();
    A3_BAR();
    issue_k(1); issue_v(0);
    if (grp == 1) A3_BAR();
    if (dbg & 1) { pf[0][0] = pf[0][1] = pf[1][0] = pf[1][1] = qf[0][0]; l_run[0] = l_run[1] = 1.f; }
    if (dbg & 2) { for (int kb = 0; kb < 4; ++kb) { s[0][kb] = (f32x4)(0.f); s[1][kb] = (f32x4)(0.f); } }

    for (int t = 0; t < ((dbg & 32) ? 0 : ntiles); ++t) {
        // ---- M(t)
        if (wave_active && !(dbg & 2) && !((dbg & 64) && wave >= 4)) {
            __builtin_amdgcn_s_setprio(1);
            if (nkb_of(t) == 4) m_phase(t, t > 0);
            else { if (t > 0) pv(t - 1, 4); qk(t, nkb_of(t)); }      // ragged last key tile: compiler-scheduled reads
            __builtin_amdgcn_s_setprio(0);
        }
        if (grp == 1) A3_VMCNT0();
        A3_BAR();
        if (grp == 1 && !(dbg & 8)) { issue_k(t + 2); issue_v(t + 1); }          // group B: this is E(t+1)
        // ---- V(t)
        if (wave_active && !(dbg & 1) && !((dbg & 64) && wave < 4)) softmax(t);
        if (grp == 0) A3_VMCNT0();
        A3_BAR();
        if (grp == 0 && !(dbg & 8)) { issue_k(t + 2); issue_v(t + 1); }          // group A: this is E(t+1)
    }
    if (wave_active) pv(ntiles - 1, nkb_of(ntiles - 1));
    if (grp == 0 && !(dbg & 4)) A3_BAR();                          // pairs with group B's last barrier

    // ---- normalise; lane holds O[query li of qb][d = n*16 + g*4 + r] ----
    const int s_idx = sh / p.heads, h = sh - s_idx * p.heads;
    half_t* __restrict__ out = reinterpret_cast<half_t*>(p.out);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float lsum = l_run[qb];                                // combine the four lanes' partial row sums (all lanes active)
        lsum += __shfl_xor(lsum, 16);
        lsum += __shfl_xor(lsum, 32);
        const int qg = q0 + qb * 16 + li;
        if (qg >= p.L) continue;
        const long row = (long)s_idx * p.ntok + qg;
        const float inv = 1.0f / lsum;
#pragma unroll
        for (int n = 0; n < ND; ++n) {
            half4_t v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (half_t)(o[qb][n][r] * inv);
            *reinterpret_cast<half4_t*>(out + row * p.D + h * HD + n * 16 + g * 4) = v;
        }
    }
}

template <int HD, int BIAS>
int launch_attn3_impl(const AttnParams& p, hipStream_t stream) {
    constexpr int HDP = (HD + 31) / 32 * 32;
    const size_t lds = (size_t)2 * KT3 * lds_pitch<half_t>(HDP) * 2 + (size_t)2 * HD * 128 + (BIAS == 2 ? (size_t)QT3 * RC3 * 2 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_kernel<HD, BIAS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((p.L + QT3 - 1) / QT3, p.S * p.heads);
    hipLaunchKernelGGL((attn3_kernel<HD, BIAS>), grid, dim3(NT3), lds, stream, p);
    return (int)hipGetLastError();
}

template <int HD>
int launch_attn3_hd(const AttnParams& p, hipStream_t stream) {
    if (!p.tab_h) return launch_attn3_impl<HD, 0>(p, stream);
    if (p.KW == KT3 && p.KH <= 64 && p.nk == p.KH * p.KW) return launch_attn3_impl<HD, 2>(p, stream);
    return -1;
}

}  // namespace

// fp16 global attention (no window partition) on >= 256 keys; -1 when the geometry is not covered (caller: attention2)
int launch_attention3(const AttnParams& p_in, hipStream_t stream) {
    AttnParams p = p_in;
    { static const int dbg3 = cva_env_int("CVA_ATTN3_DBG", 0); p.dbg = dbg3; }      // ablation builds only
    if (p.win > 0 || p.nk < 256 || p.nk != p.L || (p.Lp & 63) || p.Lp < p.nk) return -1;
    if (((size_t)p.K & 15) || ((size_t)p.Vt & 15) || (long)p.hd * p.Lp * 2 >= (1L << 31) || (long)p.nk * p.hd * 2 >= (1L << 31)) return -1;
    switch (p.hd) {
        case 64: return launch_attn3_hd<64>(p, stream);
        case 80: return launch_attn3_hd<80>(p, stream);
        default: return -1;
    }
}

}  // namespace cva

#endif  // CVA_ABLATION
