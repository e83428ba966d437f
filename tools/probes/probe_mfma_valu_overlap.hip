// Do the matrix pipe and the vector ALU of a gfx950 SIMD overlap?  One workgroup of 256 or 512 threads per CU (1 or 2 waves per SIMD), every
// wave loops over register-only work:
//   M  : 16 independent v_mfma_f32_16x16x32_f16 per iteration (BIG = 1: 8 v_mfma_f32_32x32x16_f16, the same matrix-pipe time)
//   V  : NV independent VALU instructions (PKD = 1: v_pk_fma_f32, 0: v_fma_f32) (+ NE v_exp_f32) per iteration
//   MV : both in ONE wave, interleaved 1 MFMA : NV/16 VALU
//   M|V: waves 0-3 run M, waves 4-7 run V (two waves per SIMD, one of each kind)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_mfma_valu_overlap.hip -o /tmp/probe_ov && /tmp/probe_ov
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MM(i) do { if (BIG) { if ((i) < 8) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(big[(i) & 7]) : "v"(a), "v"(b)); } \
                   else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b)); } while (0)
#define PK(i) do { if (PKD) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c0), "v"(c1)); \
                   else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(e2[i]) : "v"(c0[0]), "v"(c1[0])); } while (0)
#define EX(i) asm volatile("v_exp_f32 %0, %0" : "+v"(e[i]))

template <int MODE, int NV, int NE, int PKD, int BIG>   // MODE 0: M, 1: V, 2: MV interleaved in one wave, 3: waves 0-3 M / waves 4-7 V
__global__ void k(int iters, float* out) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[16]; f32x2 x[16]; float e[16], e2[16]; f32x16 big[8];
    for (int i = 0; i < 8; ++i) big[i] = (f32x16)(0.f);
    for (int i = 0; i < 16; ++i) e2[i] = 0.25f + i;
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * j); }
    for (int i = 0; i < 16; ++i) { acc[i] = (f32x4)(0.f); x[i] = (f32x2)(0.5f + i); e[i] = -1.0f - i; }
    const f32x2 c0 = {0.999f, 1.001f}, c1 = {0.001f, -0.001f};
    const bool doM = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool doV = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (BIG) {          // 8 long MFMAs: the i-th pair of short slots shares one
                    if (!(i & 1)) MM(i >> 1);
                } else MM(i);
#pragma unroll
                for (int v = 0; v < NV / 16; ++v) PK((i * (NV / 16) + v) & 15);
                if (i < NE) EX(i);
            }
        } else {
            if (doM) {
#pragma unroll
                for (int i = 0; i < 16; ++i) MM(i);
            }
            if (doV) {
#pragma unroll
                for (int v = 0; v < NV; ++v) PK(v & 15);
#pragma unroll
                for (int i = 0; i < NE; ++i) EX(i);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][3] + x[i][0] + x[i][1] + e[i] + e2[i] + big[i & 7][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE, int NV, int NE, int PKD, int BIG>
float run(int threads, int iters, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV, NE, PKD, BIG>), dim3(256), dim3(threads), 0, 0, 16, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV, NE, PKD, BIG>), dim3(256), dim3(threads), 0, 0, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

template <int NV, int NE, int PKD, int BIG>
void suite(float* out) {
    const int it = 20000;
    printf("NV=%d %s + NE=%d v_exp_f32 per %s (%d iterations, us):\n", NV, PKD ? "v_pk_fma_f32" : "v_fma_f32", NE, BIG ? "8 x 32x32x16" : "16 x 16x16x32", it);
    printf("  1 wave/SIMD : M %.0f  V %.0f  MV(one wave, interleaved) %.0f\n", run<0, NV, NE, PKD, BIG>(256, it, out), run<1, NV, NE, PKD, BIG>(256, it, out), run<2, NV, NE, PKD, BIG>(256, it, out));
    printf("  2 waves/SIMD: M %.0f  V %.0f  MV %.0f  M|V(waves 0-3 M, waves 4-7 V) %.0f\n", run<0, NV, NE, PKD, BIG>(512, it, out), run<1, NV, NE, PKD, BIG>(512, it, out),
           run<2, NV, NE, PKD, BIG>(512, it, out), run<3, NV, NE, PKD, BIG>(512, it, out));
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    suite<16, 0, 1, 0>(out); suite<48, 0, 1, 0>(out); suite<32, 16, 1, 0>(out);
    suite<16, 0, 0, 0>(out); suite<32, 0, 0, 0>(out); suite<48, 0, 0, 0>(out); suite<64, 0, 0, 0>(out); suite<32, 16, 0, 0>(out);
    suite<16, 0, 0, 1>(out); suite<32, 0, 0, 1>(out); suite<48, 0, 0, 1>(out); suite<64, 0, 0, 1>(out); suite<32, 16, 0, 1>(out); suite<48, 0, 1, 1>(out);
    printf("(16 MFMAs = 256 matrix-pipe cycles at 16 cycles each; a v_pk_fma_f32 / v_exp_f32 wave64 issue is 4 / 4-16 cycles)\n");
    return 0;
}
