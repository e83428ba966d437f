// Halo-tiled direct 3x3 convolution on MFMA for gfx950 (fp16 production path of the U-Net decoder).
//
// Conv2d 3x3 p1 + folded bias/BatchNorm + ReLU on NHWC activations (reference: Conv2DBlock / Deconv2DBlock,
// models/segmentation/cell_segmentation/utils.py:29-40, 73-83), optionally over the channel concat of two
// sources (cellvit.py:236-242) which is never materialised.
//
// The implicit-GEMM formulation (gemm.hip, A_CONV3) re-fetches every input pixel nine times through L2.
// Here a workgroup stages, per 32-channel chunk, the (16+2) x (32+2) input halo ONCE into LDS together with
// the nine 64 x 32 filter taps, and the nine taps are then pure LDS -> MFMA work:
//   staged bytes per chunk 76 KB for 18.9 MFLOP  ->  ~250 FLOP per staged byte (implicit GEMM: 64).
// Staging is direct-to-LDS DMA (global_load_lds_dwordx4); a pixel / filter row is 64 bytes = 4 pieces and the
// bank-conflict fix is the source-side XOR  piece ^= ((row >> 2) & 1) << 1  (conflict-free for ANY base pixel,
// which matters because the nine taps read the halo at nine different alignments).
// 512 threads = 8 waves; wave w owns image rows 2w, 2w+1 of the 16 x 32 pixel tile (64 pixels) x 64 output
// channels = 4 x 4 fragments of v_mfma_f32_16x16x32_f16.  Epilogue: LDS transpose -> 16-byte NHWC stores (8-wave kernel);
// the 4-wave kernel of the 512^2 / 1024^2 layers stores straight from the accumulators (cout_of_row).
#include <stdlib.h>

#include "common.h"
#include "gemm.h"

namespace cva {

namespace {

constexpr int TH = 16, TW = 32, HW_ = TW + 2, HH_ = TH + 2;
constexpr int HALO_PIX = HH_ * HW_;                  // 612
constexpr int HALO_INSTR = (HALO_PIX + 15) / 16;     // 39 one-KiB DMA instructions
constexpr int HALO_BYTES = HALO_INSTR * 1024;        // 39936
constexpr int W_INSTR = 9 * 4;                       // 9 taps x 64 rows / 16 rows per instruction
constexpr int W_BYTES = W_INSTR * 1024;              // 36864
constexpr int CONV_LDS = HALO_BYTES + W_BYTES;       // 76800 per stage; two stages = 153600 B, one workgroup per CU
constexpr int NTH = 512, NWAVE = 8;
constexpr int MAX_H = (HALO_INSTR + NWAVE - 1) / NWAVE;   // 5
constexpr int MAX_W = (W_INSTR + NWAVE - 1) / NWAVE;      // 5

__device__ __forceinline__ int swz(int row) { return ((row >> 2) & 1) << 1; }

// ABL (ablation builds only, CVA_CONV_DBG): 1 = stage the filter taps of the first chunk only, 2 = the halo of the first
// chunk only, 4 = no MFMAs — timing experiments, results are wrong by construction.
template <int MINW, int ABL = 0>
__global__ __launch_bounds__(NTH, MINW) void conv3x3_halo_kernel(const GemmParams p) {
    using TR = Traits<half_t>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sH = smem;
    unsigned char* sW = smem + HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int H = p.H, W = p.Wd;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    // 1-D grid, XCD-aware: every XCD gets a contiguous range of (spatial tile, output-channel block) pairs with the channel
    // block fastest, so the channel blocks of one spatial tile run together on one XCD and share its L2 copy of the
    // input halo (measured before: FETCH_SIZE 1.5 GB per launch, the input re-read once per 64 output channels)
    const int ncb = (p.N + 63) / 64;
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int cb = t % ncb; t /= ncb;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int n0 = cb * 64;
    const int ctot = p.C1 + p.C2;

    const half_t* __restrict__ S1 = reinterpret_cast<const half_t*>(p.A);
    const half_t* __restrict__ S2 = reinterpret_cast<const half_t*>(p.A2);
    const half_t* __restrict__ Wp = reinterpret_cast<const half_t*>(p.W);
    const half_t* __restrict__ Zp = reinterpret_cast<const half_t*>(p.zero);

    // ---- per-lane staging descriptors (fixed over the channel loop) ----
    int h_pix[MAX_H];   // pixel index * 4 + logical piece, or -1 (zero page)
#pragma unroll
    for (int i = 0; i < MAX_H; ++i) {
        const int k = wave + NWAVE * i;                 // DMA instruction index
        const int hp = k * 16 + (lane >> 2);            // halo pixel
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool ok = k < HALO_INSTR && hp < HALO_PIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
        h_pix[i] = ok ? ((((b * H + gy) * W + gx) << 2) | ((lane & 3) ^ swz(hp))) : -1;
    }
    int w_row[MAX_W];   // element offset of (row, tap) + logical piece * 8, or -1
#pragma unroll
    for (int i = 0; i < MAX_W; ++i) {
        const int k = wave + NWAVE * i;
        const int tap = k >> 2, n = (k & 3) * 16 + (lane >> 2);
        const bool ok = k < W_INSTR && (n0 + n) < p.N;
        w_row[i] = ok ? (n0 + n) * p.ldw + tap * ctot + ((lane & 3) ^ swz(n)) * 8 : -1;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4)(0.f);

    // fragment addressing.  Halo pixel hp holds its four 16-byte pieces at byte hp*64 + ((piece ^ swz(hp)) << 4); a
    // lane reads piece g of pixel a_hp[i] + (dy*HW_ + dx).  Filter row n of tap t sits at t*4096 + n*64 + swizzled piece.
    // (address form as in conv3x3_halo4_kernel below: a register per fragment + the read's immediate offset + ONE swizzle bit taken from a
    //  per-lane mask indexed by the displacement class E = (2 * row + dx) mod 8 — two instructions per read instead of six)
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem;
    unsigned a_base[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        a_base[i] = lds0 + (unsigned)(((2 * wave + (i >> 1) + 1) * HW_ + ((i & 1) * 16 + li + 1) - (HW_ + 1)) * 64) + ((unsigned)(g & 1) << 4);
    unsigned a_u5 = 0u;
    {
        const int r8 = ((2 * wave + 1) * HW_ + li + 1) & 7;
#pragma unroll
        for (int e = 0; e < 8; ++e) a_u5 |= (unsigned)((((r8 + e) >> 2) & 1) ^ ((g >> 1) & 1)) << (5 + e);
    }
    unsigned b_ad[4];                                // LDS byte address of W fragment j, tap 0, buffer 0
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int n = j * 16 + li; b_ad[j] = lds0 + HALO_BYTES + n * 64 + ((g ^ swz(n)) << 4); }

    // Fragment reads are raw ds_read_b128 issued ONE TAP AHEAD of their MFMAs (register double buffer) and retired by
    // counted lgkmcnt waits: the LDS latency hides behind the 16 MFMAs of the previous tap instead of being exposed
    // after every fragment (the compiler's own schedule waited lgkmcnt(0) every 4 MFMAs).
    half8_t Af[2][4], Bf[2][4];
#define CV_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define CV_WAIT(n, S)                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                    \
                 : "+v"(Af[S][0]), "+v"(Af[S][1]), "+v"(Af[S][2]), "+v"(Af[S][3]), "+v"(Bf[S][0]), "+v"(Bf[S][1]), \
                   "+v"(Bf[S][2]), "+v"(Bf[S][3])                                                               \
                 :: "memory")
#define CV_RD_A8(i, TAP, S)                                                                                     \
    do {                                                                                                        \
        constexpr int e_ = (2 * (((i) >> 1) + (TAP) / 3 - 1) + ((TAP) % 3 - 1) + 16) & 7;                       \
        unsigned ad;                                                                                            \
        asm volatile("v_lshrrev_b32 %0, %1, %2\n\tv_and_or_b32 %0, %0, 32, %3" : "=&v"(ad) : "n"(e_), "v"(a_u5), "v"(a_cur[i])); \
        CV_DSR(Af[S][i], ad, (((TAP) / 3 - 1) * HW_ + ((TAP) % 3 - 1) + HW_ + 1) * 64);                         \
    } while (0)
    // reads of tap TAP into register set S from the stage buffer a_cur / b_cur point into
#define CV_READ_TAP(TAP, S, bbase)                                                                              \
    do {                                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) CV_DSR(Bf[S][j], b_cur[j], (TAP) * 4096);                 \
        CV_RD_A8(0, TAP, S); CV_RD_A8(1, TAP, S); CV_RD_A8(2, TAP, S); CV_RD_A8(3, TAP, S);                     \
    } while (0)
#define CV_MMA_TAP(S)                                                                                           \
    do {                                                                                                        \
        if (!(ABL & 4))                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Af[S][i], Bf[S][j], acc[i][j], 0, 0, 0);     \
    } while (0)
#define CV_STEP(TAP, bbase)                                                                                     \
    do {                                                                                                        \
        if ((TAP) < 8) { CV_READ_TAP(((TAP) + 1) % 9, ((TAP) + 1) & 1, bbase); CV_WAIT(8, (TAP) & 1); }         \
        else { CV_WAIT(0, (TAP) & 1); }                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        CV_MMA_TAP((TAP) & 1);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    } while (0)

    // double-buffered: the DMA of chunk ch+1 is in flight while chunk ch is multiplied (one barrier per chunk)
    auto stage = [&](int ch, int buf) {
        const int c0 = ch * 32;
        const bool second = c0 >= p.C1;
        const half_t* __restrict__ src = second ? S2 : S1;
        const int cs = second ? p.C2 : p.C1;
        const int cc = second ? c0 - p.C1 : c0;
        unsigned char* dH = sH + buf * CONV_LDS;
        unsigned char* dW = sW + buf * CONV_LDS;
#pragma unroll
        for (int i = 0; i < MAX_H; ++i) {
            const int k = wave + NWAVE * i;
            if (k < HALO_INSTR && !((ABL & 2) && ch > 0)) {                     // wave-uniform
                const half_t* s = h_pix[i] >= 0 ? src + (long)(h_pix[i] >> 2) * cs + cc + (h_pix[i] & 3) * 8 : Zp;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                                 (__attribute__((address_space(3))) void*)(dH + k * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < MAX_W; ++i) {
            const int k = wave + NWAVE * i;
            if (k < W_INSTR && !((ABL & 1) && ch > 0)) {
                const half_t* s = w_row[i] >= 0 ? Wp + w_row[i] + c0 : Zp;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                                 (__attribute__((address_space(3))) void*)(dW + k * 1024), 16, 0, 0);
            }
        }
    };
    const int nchunks = ctot / 32;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        const unsigned bbase = buf * CONV_LDS;
        unsigned a_cur[4], b_cur[4];                     // this chunk's stage buffer (opaque: the 36 tap addresses must not be hoisted)
#pragma unroll
        for (int i = 0; i < 4; ++i) { a_cur[i] = a_base[i] + bbase; b_cur[i] = b_ad[i] + bbase; asm volatile("" : "+v"(a_cur[i]), "+v"(b_cur[i])); }
        CV_READ_TAP(0, 0, bbase);                        // first tap of the chunk (its latency is exposed once per chunk)
        if (ch + 1 < nchunks) stage(ch + 1, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        CV_STEP(0, bbase); CV_STEP(1, bbase); CV_STEP(2, bbase);
        CV_STEP(3, bbase); CV_STEP(4, bbase); CV_STEP(5, bbase);
        CV_STEP(6, bbase); CV_STEP(7, bbase); CV_STEP(8, bbase);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    __syncthreads();

    // ---- epilogue: bias (+ReLU), 16-pixel slabs through LDS, 16-byte NHWC stores ----
    float* st = reinterpret_cast<float*>(smem) + wave * (16 * 68);
    const bool fuse = p.head_W != nullptr;
    float* hw = reinterpret_cast<float*>(smem + 36864);       // head weights [nout][64] + bias [nout]
    if (fuse) {
        for (int i = tid; i < p.head_nout * 65; i += NTH)
            hw[i] = i < p.head_nout * 64 ? p.head_W[i] : p.head_b[i - p.head_nout * 64];
        __syncthreads();
    }
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int n = n0 + j * 16 + li; bv[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r] + bv[j];
                if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
                st[(g * 4 + r) * 68 + j * 16 + li] = v;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int y = y0 + 2 * wave + (i >> 1);
        const int xb = x0 + (i & 1) * 16;
        if (fuse) {
            // 1x1 head on the slab: lane = (pixel rr, quarter of the 64 channels); quad reduction
            const int rr = lane >> 2, part = lane & 3;
            float ah[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) ah[n] = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float v = st[rr * 68 + part * 16 + c];
#pragma unroll
                for (int n = 0; n < 8; ++n) if (n < p.head_nout) ah[n] = fmaf(v, hw[n * 64 + part * 16 + c], ah[n]);
            }
#pragma unroll
            for (int n = 0; n < 8; ++n) { ah[n] += __shfl_xor(ah[n], 1); ah[n] += __shfl_xor(ah[n], 2); }
            const int x = xb + rr;
            if (part == 0 && y < H && x < W) {
                const long hwp = (long)H * W, pix = (long)y * W + x;
                int best = 0; float bvv = 0.f;
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    if (n < p.head_nout) {
                        const float v = ah[n] + hw[p.head_nout * 64 + n];
                        p.head_logits[((long)b * p.head_nout + n) * hwp + pix] = v;
                        if (n == 0) bvv = v; else if (n < p.head_narg && v > bvv) { bvv = v; best = n; }
                    }
                }
                if (p.head_argmax) p.head_argmax[(long)b * hwp + pix] = (uint8_t)best;
            }
        } else if (y < H) {
            if (p.out_f32) {
                float* out = reinterpret_cast<float*>(p.out);
                for (int rr = lane >> 4; rr < 16; rr += 4) {
                    const int x = xb + rr, n = n0 + (lane & 15) * 4;
                    if (x < W && n < p.N) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(st + rr * 68 + (lane & 15) * 4);
                        *reinterpret_cast<f32x4*>(out + (((long)b * H + y) * W + x) * p.ldc + n) = v;
                    }
                }
            } else {
                half_t* out = reinterpret_cast<half_t*>(p.out);
                for (int rr = lane >> 3; rr < 16; rr += 8) {
                    const int x = xb + rr, n = n0 + (lane & 7) * 8;
                    if (x < W && n < p.N) {
                        const f32x4 lo = *reinterpret_cast<const f32x4*>(st + rr * 68 + (lane & 7) * 8);
                        const f32x4 hi = *reinterpret_cast<const f32x4*>(st + rr * 68 + (lane & 7) * 8 + 4);
                        half8_t o;
                        o[0] = (half_t)lo[0]; o[1] = (half_t)lo[1]; o[2] = (half_t)lo[2]; o[3] = (half_t)lo[3];
                        o[4] = (half_t)hi[0]; o[5] = (half_t)hi[1]; o[6] = (half_t)hi[2]; o[7] = (half_t)hi[3];
                        *reinterpret_cast<half8_t*>(out + (((long)b * H + y) * W + x) * p.ldc + n) = o;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Two-workgroups-per-CU variant for the layers with few channel chunks (the 64- and 128-channel layers at 512^2 / 1024^2: 2-8
// chunks per tile).  With the kernel above ONE workgroup fills the CU (two 76.8-KB stages), so everything that is not MFMA
// work — the address set-up, the first stage's DMA latency, the epilogue and the workgroup turn-over — is exposed: measured
// 6.5 us + 1.6 us per chunk per workgroup against 2.2 us of MFMA work per chunk.  Here a workgroup is 4 waves with ONE stage
// (76.8 KB), a wave owns 4 image rows = 128 pixels x 64 output channels (8 x 4 fragments, 128 accumulator VGPRs, 12 fragment
// reads per 32 MFMAs instead of 8 per 16), and two workgroups share a CU: they run unsynchronised, so one's DMA waits,
// prologue and epilogue overlap the other's MFMAs — ping-pong at workgroup granularity, no extra barriers.
// Direct epilogue.  The MFMAs run with the operands exchanged (C^T = W . X^T: lane (g, li) of fragment (i, j) holds pixel li of slab i and
// the fragment's filter rows 4g .. 4g+3), and the 64 filter rows of the block are permuted when they are staged: LDS row j*16 + 4g + r
// carries output channel (j >> 1)*32 + g*8 + (j & 1)*4 + r.  A lane then owns, per pixel, channels [g*8, g*8+8) and [32 + g*8, 32 + g*8+8):
// two 16-byte NHWC stores per slab with the four lanes of a pixel covering 64 contiguous bytes each — and exactly the two B operands
// (pixel li, K chunk g) of the fused 1x1 head's MFMAs.  No LDS round trip: the former epilogue wrote every slab to LDS with 16
// ds_write_b16 per lane and read it back transposed; for the 64-channel layers at 1024^2 the kernel's non-MFMA skeleton was 482 of 775 us
// (profiles/r02_exp_conv_halo4_ablation.txt) — measured gain 2 - 4 % per layer (profiles/r04_p_conv_direct_epilogue_ab.txt): most of that
// skeleton is the layers' HBM traffic, not the epilogue's instructions.
__device__ __forceinline__ int cout_of_row(int n) { return ((n >> 5) << 5) + (((n >> 2) & 3) << 3) + (((n >> 4) & 1) << 2) + (n & 3); }

constexpr int NTH4 = 256, NWAVE4 = 4;
constexpr int MAX_H4 = (HALO_INSTR + NWAVE4 - 1) / NWAVE4;   // 10
constexpr int MAX_W4 = (W_INSTR + NWAVE4 - 1) / NWAVE4;      // 9

template <int ABL = 0>      // ABL as in conv3x3_halo_kernel (ablation builds only)
__global__ __launch_bounds__(NTH4, 2) void conv3x3_halo4_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sH = smem;
    unsigned char* sW = smem + HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int H = p.H, W = p.Wd;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ncb = (p.N + 63) / 64;
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int cb = t % ncb; t /= ncb;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int n0 = cb * 64;
    const int ctot = p.C1 + p.C2;

    const half_t* __restrict__ S1 = reinterpret_cast<const half_t*>(p.A);
    const half_t* __restrict__ S2 = reinterpret_cast<const half_t*>(p.A2);
    const half_t* __restrict__ Wp = reinterpret_cast<const half_t*>(p.W);
    const half_t* __restrict__ Zp = reinterpret_cast<const half_t*>(p.zero);

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4)(0.f);

    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem;
    // A-fragment addresses.  Halo pixel hp keeps piece q at byte hp*64 + ((q ^ swz(hp)) << 4), swz(hp) = 2 * bit 2 of hp, so a lane's
    // address for (fragment i, tap) is  [hp0(i) - 35]*64 + ((g & 1) << 4)  +  (tap displacement + 35)*64  with bit 5 = g's bit 1 ^ bit 2 of hp.
    // The first term is a register per fragment, the second the read's immediate offset, and bit 2 of hp = hp0(0) + 34*k + 16*(i&1) + dx
    // depends on the lane only through hp0(0) mod 8: the eight possible answers (E = (2k + dx) mod 8) are one per-lane bit mask, so a read
    // costs a shift and an and-or instead of six address instructions (measured: 4.2 VALU instructions per MFMA in this kernel, and VALU
    // work beyond two instructions per MFMA gap is not hidden — profiles/r03_probe_mfma_valu_overlap.txt).
    unsigned a_base[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        a_base[i] = lds0 + (unsigned)(((4 * wave + (i >> 1) + 1) * HW_ + ((i & 1) * 16 + li + 1) - (HW_ + 1)) * 64) + ((unsigned)(g & 1) << 4);
    unsigned a_u5 = 0u;                              // bit 5 + E: bit 1 of the physical piece for displacement class E
    {
        const int r8 = ((4 * wave + 1) * HW_ + li + 1) & 7;
#pragma unroll
        for (int e = 0; e < 8; ++e) a_u5 |= (unsigned)((((r8 + e) >> 2) & 1) ^ ((g >> 1) & 1)) << (5 + e);
    }
    unsigned b_ad[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int n = j * 16 + li; b_ad[j] = lds0 + HALO_BYTES + n * 64 + ((g ^ swz(n)) << 4); }

    // Fragment pipeline, half a tap ahead (64 fragment VGPRs): a tap is two halves of 16 MFMAs (fragments 0-3 / 4-7 of A against
    // the four B fragments).  While the SECOND half of tap t multiplies (A_hi, B[t&1]), the reads of tap t+1's first half are in
    // flight (A_lo is free again, B goes to the other B set); while the FIRST half of tap t+1 multiplies, the reads of its A_hi
    // are in flight.  Every read has 16 MFMAs (256 matrix-pipe cycles) to land; only a chunk's first half-tap is exposed.
    half8_t Af[8], Bf[2][4];
#define CV4_WAIT(n, S)                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                     \
                 : "+v"(Af[0]), "+v"(Af[1]), "+v"(Af[2]), "+v"(Af[3]), "+v"(Af[4]), "+v"(Af[5]), "+v"(Af[6]),    \
                   "+v"(Af[7]), "+v"(Bf[S][0]), "+v"(Bf[S][1]), "+v"(Bf[S][2]), "+v"(Bf[S][3])                   \
                 :: "memory")
#define CV4_RD_A(i, TAP)                                                                                         \
    do {                                                                                                         \
        constexpr int doff_ = ((TAP) / 3 - 1) * HW_ + ((TAP) % 3 - 1);                                           \
        constexpr int e_ = (2 * (((i) >> 1) + (TAP) / 3 - 1) + ((TAP) % 3 - 1) + 16) & 7;                        \
        unsigned ad;    /* asm: two instructions per read, neither hoisted nor shared across taps (72 addresses would not fit) */ \
        asm volatile("v_lshrrev_b32 %0, %1, %2\n\tv_and_or_b32 %0, %0, 32, %3" : "=&v"(ad) : "n"(e_), "v"(a_u5), "v"(a_base[i]));  \
        CV_DSR(Af[i], ad, (doff_ + HW_ + 1) * 64);                                                               \
    } while (0)
#define CV4_RD_LO(TAP)                                                                                           \
    do {                                                                                                         \
        CV4_RD_A(0, TAP); CV4_RD_A(1, TAP); CV4_RD_A(2, TAP); CV4_RD_A(3, TAP);                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) CV_DSR(Bf[(TAP) & 1][j], b_ad[j], (TAP) * 4096);           \
    } while (0)
#define CV4_RD_HI(TAP) do { CV4_RD_A(4, TAP); CV4_RD_A(5, TAP); CV4_RD_A(6, TAP); CV4_RD_A(7, TAP); } while (0)
#define CV4_MMA(I0, S)                                                                                           \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if (!(ABL & 4))                                                                                          \
        _Pragma("unroll") for (int i = (I0); i < (I0) + 4; ++i)                                                  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Bf[S][j], Af[i], acc[i][j], 0, 0, 0);         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)
// in flight on entry: LO(TAP) [8 reads].  LAST: no prefetch of the next tap (the stage is about to be overwritten).
#define CV4_STEP(TAP, LAST)                                                                                      \
    do {                                                                                                         \
        CV4_RD_HI(TAP);                                                                                          \
        CV4_WAIT(4, (TAP) & 1);                       /* LO(TAP) landed, HI(TAP) in flight */                    \
        CV4_MMA(0, (TAP) & 1);                                                                                   \
        if (!(LAST)) { CV4_RD_LO((TAP) + 1); CV4_WAIT(8, (TAP) & 1); } else { CV4_WAIT(0, (TAP) & 1); }          \
        CV4_MMA(4, (TAP) & 1);                                                                                   \
    } while (0)

    // DMA descriptors are recomputed per stage (a few dozen VALU instructions per chunk) instead of living in 19 VGPRs
    auto stage = [&](int ch) {
        const int c0 = ch * 32;
        const bool second = c0 >= p.C1;
        const half_t* __restrict__ src = second ? S2 : S1;
        const int cs = second ? p.C2 : p.C1;
        const int cc = second ? c0 - p.C1 : c0;
#pragma unroll
        for (int i = 0; i < MAX_H4; ++i) {
            const int k = wave + NWAVE4 * i;
            if (k < HALO_INSTR && !((ABL & 2) && ch > 0)) {                     // wave-uniform
                const int hp = k * 16 + (lane >> 2);
                const int hy = hp / HW_, hx = hp - hy * HW_;
                const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                const bool ok = hp < HALO_PIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
                const half_t* s = ok ? src + (long)((b * H + gy) * W + gx) * cs + cc + ((lane & 3) ^ swz(hp)) * 8 : Zp;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                                 (__attribute__((address_space(3))) void*)(sH + k * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < MAX_W4; ++i) {
            const int k = wave + NWAVE4 * i;
            if (k < W_INSTR && !((ABL & 1) && ch > 0)) {
                const int tap = k >> 2, n = (k & 3) * 16 + (lane >> 2);
                const int nc = cout_of_row(n);                                  // LDS row n holds the taps of output channel n0 + nc
                const half_t* s = (n0 + nc) < p.N ? Wp + (long)(n0 + nc) * p.ldw + tap * ctot + ((lane & 3) ^ swz(n)) * 8 + c0 : Zp;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                                 (__attribute__((address_space(3))) void*)(sW + k * 1024), 16, 0, 0);
            }
        }
    };
    const int nchunks = ctot / 32;
    for (int ch = 0; ch < nchunks; ++ch) {
        stage(ch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // (opaque per chunk: keeps the 72 tap addresses from being hoisted out of the loop into registers)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a_base[i]));
        CV4_RD_LO(0);
        CV4_STEP(0, 0); CV4_STEP(1, 0); CV4_STEP(2, 0);
        CV4_STEP(3, 0); CV4_STEP(4, 0); CV4_STEP(5, 0);
        CV4_STEP(6, 0); CV4_STEP(7, 0); CV4_STEP(8, 1);
        __syncthreads();                                  // every wave has finished reading the stage
    }

    // ---- epilogue: direct (see cout_of_row), eight 16-pixel slabs per wave ----
    const bool fuse = p.head_W != nullptr;
    // Fused 1x1 head (64 -> head_nout <= 8 channels, + argmax): two MFMAs per 16-pixel slab on the fp16 activations —
    // C^T[out n][pixel] = W[n][:] . act[pixel][:], lane (g, li) then owns pixel li and outputs g*4 .. g*4+3 — instead of
    // 16 * nout fp32 FMAs per lane and slab (the six-output type head cost 3.5 ms more per step than the two-output heads).
    // Like the reference under autocast, the head multiplies fp16 activations by fp16 weights with fp32 accumulation.
    half8_t wf[2]; f32x4 hb4 = (f32x4)(0.f);
    wf[0] = (half8_t)(0); wf[1] = (half8_t)(0);
    if (fuse) {
        if (li < p.head_nout) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) wf[ks][e] = (half_t)p.head_W[li * 64 + ks * 32 + g * 8 + e];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) if (g * 4 + r < p.head_nout) hb4[r] = p.head_b[g * 4 + r];
    }
    float bv[16];                                   // bias of the lane's channels: fragment j -> n0 + (j >> 1)*32 + g*8 + (j & 1)*4 + r
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + (j >> 1) * 32 + g * 8 + (j & 1) * 4 + r;
            bv[j * 4 + r] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
        }
    const bool relu = p.act == ACT_RELU;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int y = y0 + 4 * wave + (i >> 1);
        const int x = x0 + (i & 1) * 16 + li;
        const bool valid = y < H && x < W;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = acc[i][j][r] + bv[j * 4 + r];
                v[j * 4 + r] = relu ? fmaxf(t, 0.f) : t;
            }
        if (p.out_f32 && !fuse) {
            float* o = reinterpret_cast<float*>(p.out) + (((long)b * H + y) * W + x) * p.ldc + n0 + g * 8;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (valid && n0 + q * 32 + g * 8 < p.N) {
                    *reinterpret_cast<f32x4*>(o + q * 32) = (f32x4){v[q * 8 + 0], v[q * 8 + 1], v[q * 8 + 2], v[q * 8 + 3]};
                    *reinterpret_cast<f32x4*>(o + q * 32 + 4) = (f32x4){v[q * 8 + 4], v[q * 8 + 5], v[q * 8 + 6], v[q * 8 + 7]};
                }
            continue;
        }
        half8_t hq[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) hq[q][e] = (half_t)v[q * 8 + e];
        if (!fuse) {
            half_t* o = reinterpret_cast<half_t*>(p.out) + (((long)b * H + y) * W + x) * p.ldc + n0 + g * 8;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (valid && n0 + q * 32 + g * 8 < p.N) *reinterpret_cast<half8_t*>(o + q * 32) = hq[q];
            continue;
        }
        f32x4 ha = hb4;
        ha = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], hq[0], ha, 0, 0, 0);
        ha = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1], hq[1], ha, 0, 0, 0);
        const long hwp = (long)H * W, pix = (long)y * W + x;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (valid && g * 4 + r < p.head_nout) p.head_logits[((long)b * p.head_nout + g * 4 + r) * hwp + pix] = ha[r];
        // first maximum over outputs 0 .. head_narg-1 (strict >, ascending n): local over r, then lanes g = 0 | 1
        float bvv = ha[0]; int best = g * 4;
        if (g != 0 && !(g * 4 < p.head_narg)) bvv = -INFINITY;
#pragma unroll
        for (int r = 1; r < 4; ++r)
            if (g * 4 + r < p.head_narg && ha[r] > bvv) { bvv = ha[r]; best = g * 4 + r; }
        const float ov = __shfl_xor(bvv, 16);
        const int oi = __shfl_xor(best, 16);
        if (p.head_narg > 4 && ov > bvv) best = oi;
        if (g == 0 && valid && p.head_argmax) p.head_argmax[(long)b * hwp + pix] = (uint8_t)best;
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Persistent variant with the filter RESIDENT in LDS, for the layers whose whole filter fits next to a halo ring: Cin = 32 / 64 -> 64 output
// channels, i.e. the full-resolution 64-channel layers (decoder0.1 and the last convolution of every branch, with the fused 1x1 head).
// Round 5.  In conv3x3_halo4_kernel above every workgroup stages, per 16 x 32-pixel tile, 39.9 KB of halo AND 36.9 KB of filter taps per
// 32-channel chunk — the same taps for all 2048 tiles of an image — and then waits for them (one stage per workgroup, two workgroups per CU
// hiding each other's waits): those layers ran at 2 - 3.4 TB/s and 30 - 35 % of the MFMA rate, far from both roofs (profiles/r04_p_*).
// Here ONE workgroup of 8 waves per CU walks the tiles grid-stride; the filter (36.9 KB per chunk) is staged once per workgroup; the halo of
// the next chunk — of this tile or of the workgroup's next tile — is in flight in the other of two 39.9-KB slots while the nine taps of the
// current chunk multiply, so LDS-DMA traffic per tile falls from 153 KB to 80 KB and no tile starts with an exposed round trip.
// Wave w owns image rows 2w, 2w+1 of the tile (64 pixels x 64 output channels), fragment pipeline one tap ahead as in conv3x3_halo_kernel;
// operands exchanged + permuted filter rows as in conv3x3_halo4_kernel, so the epilogue (bias, ReLU, 16-byte NHWC stores or the fused head's
// two MFMAs per slab + argmax) runs from the accumulators.  The epilogue's stores are never waited for: each wave retires its DMA pieces
// (vmcnt(0)) at the END of a chunk's taps, before the stores are issued; they drain under the next chunk.
template <int NCH>
__global__ __launch_bounds__(NTH, 2) void conv3x3_res_kernel(const GemmParams p, const int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sW = smem;                            // [NCH][9 taps][64 rows][64 B]
    unsigned char* sH = smem + NCH * W_BYTES;            // two halo slots

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int H = p.H, W = p.Wd;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ctot = p.C1;                               // single source (host)

    const half_t* __restrict__ S1 = reinterpret_cast<const half_t*>(p.A);
    const half_t* __restrict__ Wp = reinterpret_cast<const half_t*>(p.W);
    const half_t* __restrict__ Zp = reinterpret_cast<const half_t*>(p.zero);

    // ---- the filter, once: LDS row n of a tap holds output channel cout_of_row(n) (direct epilogue, see conv3x3_halo4_kernel)
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int i = 0; i < MAX_W; ++i) {
            const int k = wave + NWAVE * i;
            if (k < W_INSTR) {                           // wave-uniform
                const int tap = k >> 2, n = (k & 3) * 16 + (lane >> 2);
                const half_t* s = Wp + (long)cout_of_row(n) * p.ldw + tap * ctot + ((lane & 3) ^ swz(n)) * 8 + c * 32;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                                 (__attribute__((address_space(3))) void*)(sW + c * W_BYTES + k * 1024), 16, 0, 0);
            }
        }

    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem;
    unsigned a_base[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        a_base[i] = lds0 + (unsigned)(NCH * W_BYTES) +
                    (unsigned)(((2 * wave + (i >> 1) + 1) * HW_ + ((i & 1) * 16 + li + 1) - (HW_ + 1)) * 64) + ((unsigned)(g & 1) << 4);
    unsigned a_u5 = 0u;
    {
        const int r8 = ((2 * wave + 1) * HW_ + li + 1) & 7;
#pragma unroll
        for (int e = 0; e < 8; ++e) a_u5 |= (unsigned)((((r8 + e) >> 2) & 1) ^ ((g >> 1) & 1)) << (5 + e);
    }
    unsigned b_ad[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int n = j * 16 + li; b_ad[j] = lds0 + n * 64 + ((g ^ swz(n)) << 4); }

    // tile id of the workgroup's it-th tile: every XCD walks a contiguous range (neighbouring tiles share halo columns / rows through its L2)
    auto tile_of = [&](int it, int& b, int& y0, int& x0) {
        int t = xcd_remap((int)blockIdx.x + it * (int)gridDim.x, ntiles);
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y; b = t / tiles_y;
        y0 = ty * TH; x0 = tx * TW;
    };
    // halo of chunk c of tile (b, y0, x0) -> slot (descriptors recomputed per stage: a few dozen VALU instructions per chunk)
    auto stage_halo = [&](int b, int y0, int x0, int c, int slot) {
#pragma unroll
        for (int i = 0; i < MAX_H; ++i) {
            const int k = wave + NWAVE * i;
            if (k < HALO_INSTR) {                        // wave-uniform
                const int hp = k * 16 + (lane >> 2);
                const int hy = hp / HW_, hx = hp - hy * HW_;
                const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                const bool ok = hp < HALO_PIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
                const half_t* s = ok ? S1 + (long)((b * H + gy) * W + gx) * ctot + c * 32 + ((lane & 3) ^ swz(hp)) * 8 : Zp;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                                 (__attribute__((address_space(3))) void*)(sH + slot * HALO_BYTES + k * 1024), 16, 0, 0);
            }
        }
    };

    // ---- epilogue constants
    const bool fuse = p.head_W != nullptr;
    half8_t wf[2]; f32x4 hb4 = (f32x4)(0.f);
    wf[0] = (half8_t)(0); wf[1] = (half8_t)(0);
    if (fuse) {
        if (li < p.head_nout) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) wf[ks][e] = (half_t)p.head_W[li * 64 + ks * 32 + g * 8 + e];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) if (g * 4 + r < p.head_nout) hb4[r] = p.head_b[g * 4 + r];
    }
    float bv[16];                                   // bias of the lane's channels: fragment j -> (j >> 1)*32 + g*8 + (j & 1)*4 + r
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j * 4 + r] = p.bias ? p.bias[(j >> 1) * 32 + g * 8 + (j & 1) * 4 + r] : 0.f;
    const bool relu = p.act == ACT_RELU;

    half8_t Af[2][4], Bf[2][4];
#define CR_MMA_TAP(S)                                                                                           \
    do {                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Bf[S][j], Af[S][i], acc[i][j], 0, 0, 0);     \
    } while (0)
#define CR_STEP(TAP)                                                                                            \
    do {                                                                                                        \
        if ((TAP) < 8) { CV_READ_TAP(((TAP) + 1) % 9, ((TAP) + 1) & 1, 0); CV_WAIT(8, (TAP) & 1); }             \
        else { CV_WAIT(0, (TAP) & 1); }                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        CR_MMA_TAP((TAP) & 1);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    } while (0)

    int it = 0;
    int b, y0, x0;
    tile_of(0, b, y0, x0);
    stage_halo(b, y0, x0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int slot = 0;
    for (;;) {
        const bool has_next = (int)blockIdx.x + (it + 1) * (int)gridDim.x < ntiles;      // block-uniform
        int nb = 0, ny0 = 0, nx0 = 0;
        if (has_next) tile_of(it + 1, nb, ny0, nx0);
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4)(0.f);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            __syncthreads();                             // this chunk's halo (and, the first time, the filter) has landed for every wave;
                                                         // every wave has finished reading the other slot
            unsigned a_cur[4], b_cur[4];                 // (opaque: the 36 tap addresses must not be hoisted)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a_cur[i] = a_base[i] + (unsigned)slot * HALO_BYTES; b_cur[i] = b_ad[i] + c * W_BYTES;
                asm volatile("" : "+v"(a_cur[i]), "+v"(b_cur[i]));
            }
            CV_READ_TAP(0, 0, 0);
            if (c + 1 < NCH) stage_halo(b, y0, x0, c + 1, slot ^ 1);
            else if (has_next) stage_halo(nb, ny0, nx0, 0, slot ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            CR_STEP(0); CR_STEP(1); CR_STEP(2); CR_STEP(3); CR_STEP(4); CR_STEP(5); CR_STEP(6); CR_STEP(7); CR_STEP(8);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the next halo (they had nine taps to land)
            slot ^= 1;
        }
        // ---- epilogue of tile (b, y0, x0) from the accumulators; its stores drain under the next chunk
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = y0 + 2 * wave + (i >> 1);
            const int x = x0 + (i & 1) * 16 + li;
            const bool valid = y < H && x < W;
            float v[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t = acc[i][j][r] + bv[j * 4 + r];
                    v[j * 4 + r] = relu ? fmaxf(t, 0.f) : t;
                }
            half8_t hq[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) hq[q][e] = (half_t)v[q * 8 + e];
            if (!fuse) {
                half_t* o = reinterpret_cast<half_t*>(p.out) + (((long)b * H + y) * W + x) * p.ldc + g * 8;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (valid) *reinterpret_cast<half8_t*>(o + q * 32) = hq[q];
                continue;
            }
            f32x4 ha = hb4;
            ha = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], hq[0], ha, 0, 0, 0);
            ha = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1], hq[1], ha, 0, 0, 0);
            const long hwp = (long)H * W, pix = (long)y * W + x;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (valid && g * 4 + r < p.head_nout) p.head_logits[((long)b * p.head_nout + g * 4 + r) * hwp + pix] = ha[r];
            float bvv = ha[0]; int best = g * 4;
            if (g != 0 && !(g * 4 < p.head_narg)) bvv = -INFINITY;
#pragma unroll
            for (int r = 1; r < 4; ++r)
                if (g * 4 + r < p.head_narg && ha[r] > bvv) { bvv = ha[r]; best = g * 4 + r; }
            const float ov = __shfl_xor(bvv, 16);
            const int oi = __shfl_xor(best, 16);
            if (p.head_narg > 4 && ov > bvv) best = oi;
            if (g == 0 && valid && p.head_argmax) p.head_argmax[(long)b * hwp + pix] = (uint8_t)best;
        }
        if (!has_next) break;
        ++it; b = nb; y0 = ny0; x0 = nx0;
    }
#undef CR_STEP
#undef CR_MMA_TAP
}

}  // namespace

// Returns -1 when the layer does not fit this kernel (caller uses the implicit-GEMM path).
int launch_conv3x3_halo(const GemmParams& p, int batch, hipStream_t stream) {
    if (p.C1 % 32 != 0 || p.C2 % 32 != 0 || p.N % 8 != 0 || p.ldc % 8 != 0 || !p.zero) return -1;
    if (p.head_W) { if (p.N != 64 || p.head_nout < 1 || p.head_nout > 8 || !p.head_logits) return -1; }
    else if (((size_t)p.out & 15) != 0 || !p.out) return -1;
    static bool attr8 = false;
    if (!attr8) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) return (int)hipGetLastError();
        attr8 = true;
    }
    const int tiles = batch * ((p.H + TH - 1) / TH) * ((p.Wd + TW - 1) / TW);
    // Cin 32 / 64 -> 64 channels, one source, fp16 output or the fused head: the persistent kernel with the filter resident in LDS
    static const int res_on = cva_env_int("CVA_CONV_RES", 1);
    if (res_on && p.N == 64 && p.C2 == 0 && (p.C1 == 32 || p.C1 == 64) && !p.out_f32 && p.ldc == 64 && tiles >= 64) {
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
                n_cu = 256;
        }
        const int grid_r = tiles < n_cu ? tiles : n_cu;
        if (p.C1 == 64) {
            static bool a2 = false;
            if (!a2) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_res_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return (int)hipGetLastError(); a2 = true; }
            hipLaunchKernelGGL(conv3x3_res_kernel<2>, dim3(grid_r), dim3(NTH), 2 * W_BYTES + 2 * HALO_BYTES, stream, p, tiles);
        } else {
            static bool a1 = false;
            if (!a1) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_res_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return (int)hipGetLastError(); a1 = true; }
            hipLaunchKernelGGL(conv3x3_res_kernel<1>, dim3(grid_r), dim3(NTH), W_BYTES + 2 * HALO_BYTES, stream, p, tiles);
        }
        return (int)hipGetLastError();
    }
    const dim3 grid(tiles * ((p.N + 63) / 64));
    // few channel chunks per tile: the 4-wave single-stage kernel, two workgroups per CU (see conv3x3_halo4_kernel)
    static const int halo4_max_chunks = cva_env_int("CVA_CONV_HALO4", 8);
    if ((p.C1 + p.C2) / 32 <= halo4_max_chunks) {
        static bool attr4 = false;
        if (!attr4) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo4_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    CONV_LDS) != hipSuccess) return (int)hipGetLastError();
            attr4 = true;
        }
#ifdef CVA_ABLATION
        static const int one_wg = cva_env_int("CVA_CONV_HALO4_ONE", 0);     // experiment: claim the whole LDS -> one workgroup per CU
        if (one_wg) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo4_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL(conv3x3_halo4_kernel<0>, grid, dim3(NTH4), 2 * CONV_LDS, stream, p);
            return (int)hipGetLastError();
        }
        static const int dbg4 = cva_env_int("CVA_CONV_DBG", 0);
        if (dbg4) {
#define CVA_CONV4_ABL(A) do { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo4_kernel<A>), hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS); \
                              hipLaunchKernelGGL(conv3x3_halo4_kernel<A>, grid, dim3(NTH4), CONV_LDS, stream, p); return (int)hipGetLastError(); } while (0)
            switch (dbg4) { case 1: CVA_CONV4_ABL(1); case 2: CVA_CONV4_ABL(2); case 3: CVA_CONV4_ABL(3); case 4: CVA_CONV4_ABL(4); case 7: CVA_CONV4_ABL(7); default: break; }
#undef CVA_CONV4_ABL
        }
#endif
        hipLaunchKernelGGL(conv3x3_halo4_kernel<0>, grid, dim3(NTH4), CONV_LDS, stream, p);
        return (int)hipGetLastError();
    }
#ifdef CVA_ABLATION
    static const int dbg = cva_env_int("CVA_CONV_DBG", 0);
    if (dbg) {
        static bool attr = false;
#define CVA_CONV_ABL(A) do { if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<2, A>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
                             hipLaunchKernelGGL((conv3x3_halo_kernel<2, A>), grid, dim3(NTH), 2 * CONV_LDS, stream, p); return (int)hipGetLastError(); } while (0)
        switch (dbg) { case 1: CVA_CONV_ABL(1); case 2: CVA_CONV_ABL(2); case 3: CVA_CONV_ABL(3); case 4: CVA_CONV_ABL(4); case 7: CVA_CONV_ABL(7); default: break; }
#undef CVA_CONV_ABL
    }
#endif
    hipLaunchKernelGGL(conv3x3_halo_kernel<2>, grid, dim3(NTH), 2 * CONV_LDS, stream, p);
    return (int)hipGetLastError();
}

}  // namespace cva
