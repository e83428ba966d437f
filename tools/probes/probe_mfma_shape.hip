// Which fp16 MFMA shape does a gfx950 SIMD sustain more FLOP/s on under the part's power budget — v_mfma_f32_16x16x32_f16 or
// v_mfma_f32_32x32x16_f16?  Register-only loops in the register blocking of gemm8.hip: a wave owns a 128 x 64 block of C (128 accumulator VGPRs)
// and per "K tile" of 64 multiplies 8 A fragments + 4 W fragments (16 bytes per lane each, 48 VGPRs) into it:
//   SMALL: 64 x v_mfma_f32_16x16x32_f16 (8 x 4 accumulators of 4 VGPRs; A[mi][ks] x W[nj][ks])
//   BIG  : 32 x v_mfma_f32_32x32x16_f16 (4 x 2 accumulators of 16 VGPRs; A[mb][ks] x W[nb][ks], ks = 0..3)
// Both are 64 * 16384 = 32 * 32768 FLOP = 1024 matrix-pipe cycles per wave.  Operands are RANDOM fp16 (DATA = 1) or zeros (DATA = 0): the clock the
// part sustains depends on the toggling.  One workgroup of 256 / 512 threads per CU (1 / 2 waves per SIMD), 256 workgroups.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_mfma_shape.hip -o tools/probes/_bin/probe_mfma_shape && tools/probes/_bin/probe_mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// MX-fp8 (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 operands, E8M0 block scales = 2^0): the same 128 x 64 wave block, one K tile of 128 =
// 8 x 4 instructions of 65536 FLOP (32 matrix-pipe cycles each at the 5 PFLOP/s peak): 4 A fragments + 2 W fragments of 32 bytes per lane
// per 64 x 32 quadrant, as gemm8.hip's F8 loop holds them.
__global__ __launch_bounds__(512) void k8(int iters, const i32x8* __restrict__ src, float* out, long long* clk) {
    i32x8 A[8], W[4];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) A[i] = src[(size_t)t * 12 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) W[i] = src[(size_t)t * 12 + 8 + i];
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4)(0.f);
    const int sc = 0x7f7f7f7f;
    const long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int mh = (q == 0 || q == 1) ? 0 : 1, nh = (q == 0 || q == 3) ? 0 : 1;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
                    acc[mh * 4 + mi][nh * 2 + nj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(W[nh * 2 + nj], A[mh * 4 + mi], acc[mh * 4 + mi][nh * 2 + nj], 0, 0, 0, sc, 0, sc);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(A[i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(W[i]));
    }
    const long long c1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[t] = r;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

void run8(int threads, int iters, const i32x8* src, float* out, long long* clk, const char* data) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k8, dim3(256), dim3(threads), 0, 0, 64, src, out, clk);
    hipDeviceSynchronize();
    float best = 1e30f; long long cyc = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k8, dim3(256), dim3(threads), 0, 0, iters, src, out, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; long long c[256]; hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost); cyc = c[0]; }
    }
    const double flop = (double)iters * 32.0 * 65536.0 * (threads / 64) * 256.0;
    const double us = best * 1e3;
    printf("  %-10s  waves/SIMD=%d data=%-6s: %8.0f us  %7.1f TFLOP/s  s_memtime %lld ticks (%.1f per MFMA and SIMD-wave, tick rate %.0f MHz)\n",
           "mx8 K=128", threads / 256, data, us, flop / us * 1e-6, cyc, (double)cyc / ((double)iters * 32 * (threads / 256)), (double)cyc / us);
}

template <int BIG>
__global__ __launch_bounds__(512) void k(int iters, const half8* __restrict__ src, float* out, long long* clk) {
    half8 A[8], W[4];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) A[i] = src[(size_t)t * 12 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) W[i] = src[(size_t)t * 12 + 8 + i];
    f32x4 acc[8][4];
    f32x16 big[4][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4)(0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) big[i][j] = (f32x16)(0.f);
    const long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (BIG) {
            // two quadrants of 64 x 32 per "phase pair": (mb pair, nb) x 4 k steps = 8 MFMAs per quadrant
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mh = (q == 0 || q == 1) ? 0 : 1, nb = (q == 0 || q == 3) ? 0 : 1;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
                        big[mh * 2 + mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[nb * 2 + (ks & 1)], A[(mh * 2 + mb) * 2 + (ks >> 1)], big[mh * 2 + mb][nb], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mh = (q == 0 || q == 1) ? 0 : 1, nh = (q == 0 || q == 3) ? 0 : 1;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int nj = 0; nj < 2; ++nj)
                            acc[mh * 4 + mi][nh * 2 + nj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[nh * 2 + nj], A[(mi * 2 + ks) & 7], acc[mh * 4 + mi][nh * 2 + nj], 0, 0, 0);
            }
        }
        // keep the operands opaque (no hoisting / CSE across iterations), no instructions
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(A[i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(W[i]));
    }
    const long long c1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    if (BIG) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) r += big[i][j][e];
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    }
    out[t] = r;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

template <int BIG>
void run(int threads, int iters, const half8* src, float* out, long long* clk, const char* data) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BIG>), dim3(256), dim3(threads), 0, 0, 64, src, out, clk);
    hipDeviceSynchronize();
    float best = 1e30f; long long cyc = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<BIG>), dim3(256), dim3(threads), 0, 0, iters, src, out, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; long long c[256]; hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost); cyc = c[0]; }
    }
    const double flop = (double)iters * 64.0 * 16384.0 * (threads / 64) * 256.0;
    const double us = best * 1e3;
    printf("  %-10s %s waves/SIMD=%d data=%-6s: %8.0f us  %7.1f TFLOP/s  s_memtime %lld ticks (%.1f per %s MFMA and SIMD-wave, tick rate %.0f MHz)\n",
           BIG ? "32x32x16" : "16x16x32", "", threads / 256, data, us, flop / us * 1e-6, cyc,
           (double)cyc / ((double)iters * (BIG ? 32 : 64) * (threads / 256)), BIG ? "32x32x16" : "16x16x32", (double)cyc / us);
}

int main() {
    const size_t n = (size_t)256 * 512 * 12;
    std::vector<_Float16> h(n * 8);
    srand(1234);
    for (auto& v : h) v = (_Float16)(((float)rand() / (float)RAND_MAX) * 2.f - 1.f);
    half8 *src, *zero; float* out; long long* clk;
    hipMalloc(&src, n * 16); hipMalloc(&zero, n * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
    hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice); hipMemset(zero, 0, n * 16);
    // fp8 operands: random e4m3 bytes without the NaN encodings (0x7f / 0xff), 32 bytes per lane and fragment
    const size_t n8 = (size_t)256 * 512 * 12;
    std::vector<unsigned char> h8(n8 * 32);
    for (auto& v : h8) { unsigned char b = (unsigned char)(rand() & 0xff); if ((b & 0x7f) == 0x7f) b &= 0xfe; v = b; }
    i32x8 *src8, *zero8;
    hipMalloc(&src8, n8 * 32); hipMalloc(&zero8, n8 * 32);
    hipMemcpy(src8, h8.data(), n8 * 32, hipMemcpyHostToDevice); hipMemset(zero8, 0, n8 * 32);
    const int it = 20000;
    for (int pass = 0; pass < 2; ++pass) {
        printf("pass %d (%d iterations of one 128x64x64 wave-tile step = 1024 matrix-pipe cycles per wave):\n", pass, it);
        for (int threads = 256; threads <= 512; threads += 256) {
            run<0>(threads, it, src, out, clk, "random");
            run<1>(threads, it, src, out, clk, "random");
            run<0>(threads, it, zero, out, clk, "zeros");
            run<1>(threads, it, zero, out, clk, "zeros");
            run8(threads, it / 2, src8, out, clk, "random");
            run8(threads, it / 2, zero8, out, clk, "zeros");
        }
    }
    return 0;
}
