// What exactly does ds_read_b64_tr_b16 return on gfx950?  (the transposing LDS read the attention kernels would use to consume a
// ROW-major V tile [key][d] as the A operand of O^T = V^T . P^T — VERDICT r03 item 2.)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_ds_read_tr.hip -o /tmp/probe_tr && /tmp/probe_tr
// LDS holds u16 element e at byte 2e (value = e).  Experiment 1: every lane passes a per-lane address, the four returned 16-bit
// values of every lane are printed -> which lane's address supplies which element.  Experiment 2: the layout the PV step wants —
// lane (g, i) of a 16-lane group passes the address of key 4g + i/4, elements d0 + 4*(i%4) .. +3 of a [key][PITCH] tile — and the
// check that lane (g, li) then holds V[4g + r][d0 + li], r = 0..3 (the transposed 4 x 16 block).  Experiment 3: timing of the read
// for row pitches 160 B (hd 80, unpadded), 128 B (hd 64), 144 B, 176 B: cycles per instruction (bank-conflict classes).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned short u16;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void k_probe(const unsigned* addr_in, unsigned* out) {
    __shared__ __attribute__((aligned(16))) u16 lds[8192];
    const int l = threadIdx.x;
    for (int i = l; i < 8192; i += 64) lds[i] = (u16)i;
    __syncthreads();
    const unsigned base = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds;
    const unsigned a = base + addr_in[l];
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    out[l * 2] = r[0]; out[l * 2 + 1] = r[1];
}

template <int PITCH>
__global__ void k_time(unsigned* out, int iters) {
    __shared__ __attribute__((aligned(16))) u16 lds[16384];
    const int l = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (u16)i;
    __syncthreads();
    const unsigned base = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds;
    const int g = l >> 4, i = l & 15;
    const unsigned a = base + (4 * g + (i >> 2)) * PITCH + (i & 3) * 8;
    u32x2 acc = {0u, 0u};
    const long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        u32x2 r0, r1, r2, r3;
        asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:32\n\tds_read_b64_tr_b16 %2, %4 offset:64\n\t"
                     "ds_read_b64_tr_b16 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a) : "memory");
        acc[0] += r0[0] + r1[0] + r2[0] + r3[0]; acc[1] += r0[1] + r1[1] + r2[1] + r3[1];
    }
    const long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = (unsigned)(t1 - t0); out[1] = acc[0] + acc[1]; }
}

int main() {
    unsigned *d_addr, *d_out;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 4096);
    unsigned addr[64], out[128];
    // ---- experiment 1: lane l passes address l * 64 bytes (element 32 l): who gets what?
    for (int l = 0; l < 64; ++l) addr[l] = l * 64;
    hipMemcpy(d_addr, addr, sizeof(addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost);
    printf("exp1: lane l passes byte address 64*l (element 32*l).  returned elements (e0 e1 e2 e3), as (source lane : element offset)\n");
    for (int l = 0; l < 64; ++l) {
        const unsigned e[4] = {out[2 * l] & 0xffff, out[2 * l] >> 16, out[2 * l + 1] & 0xffff, out[2 * l + 1] >> 16};
        printf("  lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf("  (%2u:%u)", e[j] / 32, e[j] % 32);
        printf("\n");
    }
    // ---- experiment 2: the PV layout, pitch 160 B (hd = 80), d0 = 16
    const int PITCH = 160, d0 = 16;
    for (int l = 0; l < 64; ++l) { const int g = l >> 4, i = l & 15; addr[l] = (4 * g + (i >> 2)) * PITCH + (d0 + 4 * (i & 3)) * 2; }
    hipMemcpy(d_addr, addr, sizeof(addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, li = l & 15;
        const unsigned e[4] = {out[2 * l] & 0xffff, out[2 * l] >> 16, out[2 * l + 1] & 0xffff, out[2 * l + 1] >> 16};
        for (int r = 0; r < 4; ++r) {
            const unsigned want = (unsigned)(((4 * g + r) * PITCH) / 2 + d0 + li);      // V[4g + r][d0 + li]
            if (e[r] != want) { if (bad < 8) printf("  exp2 mismatch lane %d r %d: got element %u want %u\n", l, r, e[r], want); ++bad; }
        }
    }
    printf("exp2 (lane (g,i) -> key 4g + i/4, d0 + 4(i%%4); expect lane (g,li) = V[4g+r][d0+li]): %s (%d mismatches)\n", bad ? "DIFFERENT" : "AS EXPECTED", bad);
    // ---- experiment 3: timing per pitch
    const int iters = 4096;
#define TIME(P)                                                                                         \
    do {                                                                                                \
        hipLaunchKernelGGL(k_time<P>, dim3(1), dim3(64), 0, 0, d_out, iters);                           \
        hipMemcpy(out, d_out, 8, hipMemcpyDeviceToHost);                                                \
        printf("exp3: pitch %3d B, one wave : %.1f cycles per ds_read_b64_tr_b16\n", P, (double)out[0] / iters / 4);  \
        hipLaunchKernelGGL(k_time<P>, dim3(1), dim3(256), 0, 0, d_out, iters);                          \
        hipMemcpy(out, d_out, 8, hipMemcpyDeviceToHost);                                                \
        printf("exp3: pitch %3d B, four waves: %.1f cycles per ds_read_b64_tr_b16 (wave 0's clock)\n", P, (double)out[0] / iters / 4);  \
    } while (0)
    TIME(128); TIME(144); TIME(160); TIME(176); TIME(192);
    return 0;
}
