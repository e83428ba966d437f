// ConvTranspose2d(k2, s2) o Conv2d(3x3, p1) + folded BatchNorm + ReLU as ONE halo-tiled direct convolution, for the full-resolution
// up-sampling stages of the decoder (Cout = 128 / 64 at 512^2 / 1024^2: `decoder2_upsampler.2 -> decoder1_upsampler.0`,
// `decoder1_upsampler.2 -> decoder0_header.0`, cellvit.py:285-315, and the shared `Deconv2DBlock`s with Cout < 256, utils.py:46-86) — the
// stages the 256-column tiles of gemm8.hip's composed launch cannot take (a tile there must lie inside one output parity).  gfx950, fp16.
//
// Output pixel (2y + py, 2x + px) of the up-sampled resolution reads the up-sampled map at its 3x3 neighbourhood, i.e. the INPUT pixels
// (y + py - 1 + s, x + px - 1 + t), s, t in {0, 1}, through weights composed on the host per parity (cellvit_abi.hip, pack_deconv_comp:
// Wc[par][co][(chunk, s, t, ch)], then the 3x3 filter of the skip half [(chunk, tap, ch)]): a 2x2-tap convolution of the input per parity
// plus the ordinary 3x3 convolution of the skip source (cat([skip, up]), cellvit.py:236-242) at the output resolution.  The up-sampled
// tensor (0.6 GB per 1024^2 tile over the three branches, written and read back) and its seven launches never exist, and the up-sampled
// half of the convolution costs 4 * Cin instead of 9 * Cup MACs per output (-11 %).
//
// One workgroup = 4 waves = a 16 x 32-pixel tile of the OUTPUT x 64 output channels; wave w owns the tile's 128 pixels of parity
// (py, px) = (w >> 1, w & 1): eight slabs of 16 pixels (slab r = output row y0 + 2r + py, lane column li = output column x0 + 2 li + px) x 4
// fragments of 16 channels = 128 accumulator VGPRs — conv3x3_halo4_kernel's register budget, two workgroups per CU covering each other's
// staging waits.  Stages (one at a time, 76 KB):
//   input chunk (32 channels of z): the (8 + 2) x (16 + 2) input pixels under the tile (11.5 KB) + 4 parities x 4 taps x 64 rows of composed
//     filter (64 KB); a wave multiplies its parity's four taps: 4 x 32 MFMAs;
//   skip chunk (32 channels of the skip source): the (16 + 2) x (32 + 2) halo stored PARITY-SPLIT — four planes of 9 x 17 pixels, plane
//     (hy & 1, hx & 1) — so that the 16 pixels of a slab, which lie two apart in the image, are 16 CONSECUTIVE pixels of one plane for every
//     tap (the 64-byte pixel rows and their XOR swizzle are conv.hip's: conflict-free for any 16 consecutive pixels) + the nine 64 x 32 taps
//     (36 KB, the same for all parities): 9 x 32 MFMAs per wave.
// The parity is a compile-time constant of the tap loops (four copies of the loop body, one per wave), so every fragment address is ONE of
// eight per-lane registers (the swizzle bit for the displacement class D mod 8) + an immediate: no address arithmetic beside the MFMAs.
// Direct epilogue as in conv3x3_halo4_kernel (operands exchanged, filter rows permuted: a lane owns channels [8g, 8g + 8) and [32 + 8g, ..)
// of its pixel); the transposed convolution's bias reaches an output pixel through the taps inside the image only: border pixels take their
// bias from the nine-case table of pack_deconv_comp.
#include "common.h"
#include "gemm.h"

namespace cva {

namespace {

constexpr int DTH = 16, DTW = 32;                              // output tile
constexpr int DZH = DTH / 2 + 2, DZW = DTW / 2 + 2;            // input halo 10 x 18
constexpr int DZ_PIX = DZH * DZW;                              // 180
constexpr int DZ_INSTR = (DZ_PIX + 15) / 16;                   // 12 one-KiB DMA instructions
constexpr int DZ_BYTES = DZ_INSTR * 1024;                      // 12288
constexpr int DWZ_INSTR = 16 * 4;                              // 4 parities x 4 taps x 64 rows / 16 rows per instruction
constexpr int DWZ_BYTES = DWZ_INSTR * 1024;                    // 65536
constexpr int DPW = DTW / 2 + 1, DPH = DTH / 2 + 1;            // a parity plane of the skip halo: 9 rows x 17 pixels
constexpr int DPS = DPH * DPW;                                 // 153 pixels per plane
constexpr int DS_PIX = 4 * DPS;                                // 612
constexpr int DS_INSTR = (DS_PIX + 15) / 16;                   // 39
constexpr int DS_BYTES = DS_INSTR * 1024;                      // 39936
constexpr int DWS_INSTR = 9 * 4;
constexpr int DWS_BYTES = DWS_INSTR * 1024;                    // 36864
constexpr int DECONV_LDS = (DZ_BYTES + DWZ_BYTES) > (DS_BYTES + DWS_BYTES) ? (DZ_BYTES + DWZ_BYTES) : (DS_BYTES + DWS_BYTES);   // 77824: two per CU
constexpr int DNT = 256, DNW = 4;

__device__ __forceinline__ int dswz(int row) { return ((row >> 2) & 1) << 1; }
// LDS filter row n of a 64-row block carries output channel dcout(n) (see conv.hip, cout_of_row: the direct epilogue's permutation)
__device__ __forceinline__ int dcout(int n) { return ((n >> 5) << 5) + (((n >> 2) & 3) << 3) + (((n >> 4) & 1) << 2) + (n & 3); }

#define DH_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define DH_WAIT(n, S)                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                     \
                 : "+v"(Af[0]), "+v"(Af[1]), "+v"(Af[2]), "+v"(Af[3]), "+v"(Af[4]), "+v"(Af[5]), "+v"(Af[6]),    \
                   "+v"(Af[7]), "+v"(Bf[S][0]), "+v"(Bf[S][1]), "+v"(Bf[S][2]), "+v"(Bf[S][3])                   \
                 :: "memory")
// A fragment of slab i at halo displacement D (pixels): the per-lane address register of displacement class D mod 8 + the immediate
#define DH_RD_A(i, D) do { constexpr int d_ = (D); DH_DSR(Af[i], adE[d_ & 7], d_ * 64); } while (0)
#define DH_MMA(I0, S)                                                                                            \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        _Pragma("unroll") for (int i = (I0); i < (I0) + 4; ++i)                                                  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Bf[S][j], Af[i], acc[i][j], 0, 0, 0);         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)

// ---- input-chunk taps: tap T4 = s * 2 + t reads input halo pixel (r + PY + s, li + PX + t); filter block (parity, T4) at T4 * 4096 past bz
#define ZD(i, T4) (((i) + PY + ((T4) >> 1)) * DZW + PX + ((T4) & 1))
#define Z_RD_LO(T4)                                                                                              \
    do {                                                                                                         \
        DH_RD_A(0, ZD(0, T4)); DH_RD_A(1, ZD(1, T4)); DH_RD_A(2, ZD(2, T4)); DH_RD_A(3, ZD(3, T4));              \
        DH_DSR(Bf[(T4) & 1][0], bz[0], (T4) * 4096); DH_DSR(Bf[(T4) & 1][1], bz[1], (T4) * 4096);                \
        DH_DSR(Bf[(T4) & 1][2], bz[2], (T4) * 4096); DH_DSR(Bf[(T4) & 1][3], bz[3], (T4) * 4096);                \
    } while (0)
#define Z_RD_HI(T4) do { DH_RD_A(4, ZD(4, T4)); DH_RD_A(5, ZD(5, T4)); DH_RD_A(6, ZD(6, T4)); DH_RD_A(7, ZD(7, T4)); } while (0)
// in flight on entry: LO(T4) [8 reads].  Half a tap ahead, as conv3x3_halo4_kernel: every read has 16 MFMAs to land.
#define Z_STEP(T4, LAST)                                                                                         \
    do {                                                                                                         \
        Z_RD_HI(T4);                                                                                             \
        DH_WAIT(4, (T4) & 1);                                                                                    \
        DH_MMA(0, (T4) & 1);                                                                                     \
        if (!(LAST)) { Z_RD_LO((T4) + 1); DH_WAIT(8, (T4) & 1); } else { DH_WAIT(0, (T4) & 1); }                 \
        DH_MMA(4, (T4) & 1);                                                                                     \
    } while (0)

// ---- skip-chunk taps: tap (ky, kx) reads halo pixel (2r + PY + ky, 2 li + PX + kx) = plane ((PY + ky) & 1, (PX + kx) & 1), row r + ((PY + ky) >> 1),
// column li + ((PX + kx) >> 1) of the parity-split halo
#define SD(i, TAP) (((((PY + (TAP) / 3) & 1) * 2 + ((PX + (TAP) % 3) & 1)) * DPS) + ((i) + ((PY + (TAP) / 3) >> 1)) * DPW + ((PX + (TAP) % 3) >> 1))
#define S_RD_LO(TAP)                                                                                             \
    do {                                                                                                         \
        DH_RD_A(0, SD(0, TAP)); DH_RD_A(1, SD(1, TAP)); DH_RD_A(2, SD(2, TAP)); DH_RD_A(3, SD(3, TAP));          \
        DH_DSR(Bf[(TAP) & 1][0], bs[0], (TAP) * 4096); DH_DSR(Bf[(TAP) & 1][1], bs[1], (TAP) * 4096);            \
        DH_DSR(Bf[(TAP) & 1][2], bs[2], (TAP) * 4096); DH_DSR(Bf[(TAP) & 1][3], bs[3], (TAP) * 4096);            \
    } while (0)
#define S_RD_HI(TAP) do { DH_RD_A(4, SD(4, TAP)); DH_RD_A(5, SD(5, TAP)); DH_RD_A(6, SD(6, TAP)); DH_RD_A(7, SD(7, TAP)); } while (0)
#define S_STEP(TAP, LAST)                                                                                        \
    do {                                                                                                         \
        S_RD_HI(TAP);                                                                                            \
        DH_WAIT(4, (TAP) & 1);                                                                                   \
        DH_MMA(0, (TAP) & 1);                                                                                    \
        if (!(LAST)) { S_RD_LO((TAP) + 1); DH_WAIT(8, (TAP) & 1); } else { DH_WAIT(0, (TAP) & 1); }              \
        DH_MMA(4, (TAP) & 1);                                                                                    \
    } while (0)

// GemmParams as for launch_gemm8_deconv: A = input NHWC [B, H, W, C1], A2 = skip NHWC [B, 2H, 2W, C2] or null, W = composed filter
// [4 * Cout][4 * C1 + 9 * C2] (ldw = K), N = 4 * Cout, bias = interior bias (Cout values), comp_bias = the nine-case table [9][Cout],
// out = NHWC [B, 2H, 2W, Cout] fp16, H / Wd = INPUT height / width.
// The whole per-wave program with the wave's parity as a compile-time constant (the kernel below branches ONCE, wave-uniformly, into one of the four
// instances: with the branch inside the chunk loops the 128 accumulators met in PHI copies at every join and spilled).  All four instances execute
// the same sequence of barriers.
template <int PY, int PX>
__device__ __forceinline__ void deconv_halo4_body(const GemmParams& p, unsigned char* dsm) {
    constexpr int wave = PY * 2 + PX;
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = lane >> 4, li = lane & 15;
    const int H = p.H, W = p.Wd, H2 = 2 * H, W2 = 2 * W;
    const int cout = p.N >> 2;
    const int tiles_x = (W2 + DTW - 1) / DTW, tiles_y = (H2 + DTH - 1) / DTH;
    const int ncb = cout / 64;
    // XCD-aware, channel block fastest: the channel blocks of one spatial tile run together on one XCD and share its L2 copy of the halos
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int cb = t % ncb; t /= ncb;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int b = t / tiles_y;
    const int y0 = ty * DTH, x0 = tx * DTW;            // output pixel of the tile's corner
    const int yl0 = y0 >> 1, xl0 = x0 >> 1;            // input pixel under it
    const int n0 = cb * 64;

    const half_t* __restrict__ Zs = reinterpret_cast<const half_t*>(p.A);
    const half_t* __restrict__ Ss = reinterpret_cast<const half_t*>(p.A2);
    const half_t* __restrict__ Wp = reinterpret_cast<const half_t*>(p.W);
    const half_t* __restrict__ Zp = reinterpret_cast<const half_t*>(p.zero);

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4)(0.f);

    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)dsm;
    // Halo pixel hp keeps piece q at byte hp * 64 + ((q ^ dswz(hp)) << 4), dswz(hp) = 2 * bit 2 of hp.  A lane reads piece g of pixel li + D: bit 0 of the
    // physical piece is g's, bit 1 = (g >> 1) ^ bit 2 of (li + D), which depends on D through D mod 8 only — eight address registers, one per class.
    unsigned adE[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
        adE[e] = lds0 + (unsigned)(li * 64) + ((unsigned)(g & 1) << 4) + ((unsigned)(((((li & 7) + e) >> 2) & 1) ^ ((g >> 1) & 1)) << 5);
    unsigned bz[4], bs[4];                           // filter fragment j: input stage (this wave's parity), skip stage
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = j * 16 + li;
        const unsigned rowoff = (unsigned)(n * 64 + ((g ^ dswz(n)) << 4));
        bz[j] = lds0 + DZ_BYTES + (unsigned)wave * 16384u + rowoff;
        bs[j] = lds0 + DS_BYTES + rowoff;
    }

    // ---- staging (descriptors recomputed per stage: a few dozen VALU instructions per chunk instead of live registers)
    // (the lane id is made opaque per call: everything a stage derives from it is loop invariant, and hoisted out of the chunk loops it would
    //  be ~40 address pairs live next to 192 accumulator / fragment registers — spilled)
    auto stage_z = [&](int ch) {
        const int c0 = ch * 32;
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
#pragma unroll
        for (int i = 0; i < (DZ_INSTR + DNW - 1) / DNW; ++i) {
            const int k = wave + DNW * i;
            if (k < DZ_INSTR) {                                                       // wave-uniform
                const int hp = k * 16 + (lane >> 2);
                const int zy = hp / DZW, zx = hp - zy * DZW;
                const int gy = yl0 - 1 + zy, gx = xl0 - 1 + zx;
                const bool ok = hp < DZ_PIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
                const half_t* s = ok ? Zs + (long)((b * H + gy) * W + gx) * p.C1 + c0 + ((lane & 3) ^ dswz(hp)) * 8 : Zp;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                                 (__attribute__((address_space(3))) void*)(dsm + k * 1024), 16, 0, 0);
            }
        }
        // composed filter: row (par, co), K offset (64-channel chunk, tap, channel) -> this 32-channel chunk's half of the 64-channel block
        const int koff = (c0 >> 6) * 256 + (c0 & 32);
#pragma unroll
        for (int i = 0; i < DWZ_INSTR / DNW; ++i) {
            const int k = wave + DNW * i;
            const int blk = k >> 2, n = (k & 3) * 16 + (lane >> 2);
            const int row = (blk >> 2) * cout + n0 + dcout(n);
            const half_t* s = Wp + (long)row * p.ldw + koff + (blk & 3) * 64 + ((lane & 3) ^ dswz(n)) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                             (__attribute__((address_space(3))) void*)(dsm + DZ_BYTES + k * 1024), 16, 0, 0);
        }
    };
    auto stage_s = [&](int ch) {
        const int c0 = ch * 32;
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
#pragma unroll
        for (int i = 0; i < (DS_INSTR + DNW - 1) / DNW; ++i) {
            const int k = wave + DNW * i;
            if (k < DS_INSTR) {
                const int q = k * 16 + (lane >> 2);                                   // LDS pixel slot: plane, row, column
                const int pl = q / DPS, rem = q - pl * DPS;
                const int row = rem / DPW, col = rem - row * DPW;
                const int gy = y0 - 1 + 2 * row + (pl >> 1), gx = x0 - 1 + 2 * col + (pl & 1);
                const bool ok = q < DS_PIX && gy >= 0 && gy < H2 && gx >= 0 && gx < W2;
                const half_t* s = ok ? Ss + (long)((b * H2 + gy) * W2 + gx) * p.C2 + c0 + ((lane & 3) ^ dswz(q)) * 8 : Zp;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                                 (__attribute__((address_space(3))) void*)(dsm + k * 1024), 16, 0, 0);
            }
        }
        const int koff = 4 * p.C1 + (c0 >> 6) * 576 + (c0 & 32);                      // the skip half: (64-channel chunk, tap, channel) behind the input part
#pragma unroll
        for (int i = 0; i < DWS_INSTR / DNW; ++i) {
            const int k = wave + DNW * i;
            const int tap = k >> 2, n = (k & 3) * 16 + (lane >> 2);
            const half_t* s = Wp + (long)(n0 + dcout(n)) * p.ldw + koff + tap * 64 + ((lane & 3) ^ dswz(n)) * 8;   // (the same rows in every parity: parity 0's)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                             (__attribute__((address_space(3))) void*)(dsm + DS_BYTES + k * 1024), 16, 0, 0);
        }
    };

    half8_t Af[8], Bf[2][4];
    const int nzc = p.C1 / 32, nsc = p.C2 / 32;
    for (int ch = 0; ch < nzc; ++ch) {
        stage_z(ch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        Z_RD_LO(0);
        Z_STEP(0, 0); Z_STEP(1, 0); Z_STEP(2, 0); Z_STEP(3, 1);
        __syncthreads();                                  // every wave has finished reading the stage
    }
    for (int ch = 0; ch < nsc; ++ch) {
        stage_s(ch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        S_RD_LO(0);
        S_STEP(0, 0); S_STEP(1, 0); S_STEP(2, 0);
        S_STEP(3, 0); S_STEP(4, 0); S_STEP(5, 0);
        S_STEP(6, 0); S_STEP(7, 0); S_STEP(8, 1);
        __syncthreads();
    }

    // ---- epilogue: direct (see dcout): fragment j, register r of lane (g, li) = channel n0 + (j >> 1) * 32 + g * 8 + (j & 1) * 4 + r of pixel li of slab i
    constexpr int py = PY, px = PX;
    float bv[16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j * 4 + r] = p.bias[n0 + (j >> 1) * 32 + g * 8 + (j & 1) * 4 + r];
    const bool relu = p.act == ACT_RELU;
    const int X = x0 + 2 * li + px;
    const int cc = X == 0 ? 0 : (X == W2 - 1 ? 2 : 1);
    half_t* __restrict__ outp = reinterpret_cast<half_t*>(p.out);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int Y = y0 + 2 * i + py;
        const int rc = Y == 0 ? 0 : (Y == H2 - 1 ? 2 : 1);
        const bool valid = Y < H2 && X < W2;
        float v[16];
        if (rc != 1 || cc != 1) {                         // border pixel: the taps outside the image carry no share of the transposed convolution's bias
            const float* tb = p.comp_bias + (long)(rc * 3 + cc) * cout + n0 + g * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(tb + (j >> 1) * 32 + (j & 1) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float a = acc[i][j][r] + t4[r]; v[j * 4 + r] = relu ? fmaxf(a, 0.f) : a; }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float a = acc[i][j][r] + bv[j * 4 + r]; v[j * 4 + r] = relu ? fmaxf(a, 0.f) : a; }
        }
        half_t* o = outp + (((long)b * H2 + Y) * W2 + X) * p.ldc + n0 + g * 8;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            half8_t hq;
#pragma unroll
            for (int e = 0; e < 8; ++e) hq[e] = (half_t)v[q * 8 + e];
            if (valid) *reinterpret_cast<half8_t*>(o + q * 32) = hq;
        }
    }
}

__global__ __launch_bounds__(DNT, 2) void deconv_halo4_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    switch (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6)) {          // wave-uniform
        case 0: deconv_halo4_body<0, 0>(p, dsm); break;
        case 1: deconv_halo4_body<0, 1>(p, dsm); break;
        case 2: deconv_halo4_body<1, 0>(p, dsm); break;
        default: deconv_halo4_body<1, 1>(p, dsm); break;
    }
}

}  // namespace

// Shapes: Cout a multiple of 64, input channels a multiple of 64 (the composed filter's 64-channel K blocks), skip channels 0 or a multiple of 64.
bool deconv_halo4_supported(const GemmParams& p) {
    if (p.out_mode != OUT_CONVT || p.out_f32 || p.res || p.head_W || !p.A || !p.W || !p.out || !p.bias || !p.comp_bias || !p.zero) return false;
    if ((p.A2 != nullptr) != (p.C2 > 0) || p.C2 % 64 || p.C1 % 64 || p.C1 < 64) return false;
    if (p.N % 4 || (p.N / 4) % 64 || p.K != 4 * p.C1 + 9 * p.C2 || p.ldw != p.K) return false;
    if (((size_t)p.A & 15) || ((size_t)p.A2 & 15) || ((size_t)p.W & 15) || ((size_t)p.out & 15) || ((size_t)p.comp_bias & 15)) return false;
    if ((long)p.H * p.Wd * 4 >= (1L << 31)) return false;
    return true;
}

int launch_deconv_halo4(const GemmParams& p_in, int batch, hipStream_t stream) {
    GemmParams p = p_in;
    if (!deconv_halo4_supported(p)) return -1;
    p.ldc = p.N / 4;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&deconv_halo4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DECONV_LDS) != hipSuccess)
            return (int)hipGetLastError();
        attr = true;
    }
    const int tiles = batch * ((2 * p.H + DTH - 1) / DTH) * ((2 * p.Wd + DTW - 1) / DTW);
    hipLaunchKernelGGL(deconv_halo4_kernel, dim3(tiles * (p.N / 4 / 64)), dim3(DNT), DECONV_LDS, stream, p);
    return (int)hipGetLastError();
}

}  // namespace cva
