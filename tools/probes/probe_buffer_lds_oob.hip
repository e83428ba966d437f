// What does an LDS-DMA buffer load (`buffer_load_dwordx4 ... offen lds`) write for lanes whose offset fails the descriptor's
// range check on gfx950 — zeros, or nothing?  And is the SGPR offset part of the range check?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_buffer_lds_oob.hip -o /tmp/probe_oob && /tmp/probe_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const unsigned char* src, unsigned num_records, const unsigned* voff_in, unsigned soff, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[256];
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) lds[i] = 0xABABABABu;
    __syncthreads();
    const unsigned long b = (unsigned long)src;
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu));
    d[1] = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
    d[2] = __builtin_amdgcn_readfirstlane((int)num_records);
    d[3] = 0x00020000;
    const unsigned voff = voff_in[l];
    const unsigned dst = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds;
    const unsigned so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_waitcnt vmcnt(0)"
                 :: "v"(voff), "s"(d), "s"(so), "s"(dst) : "memory");
    __syncthreads();
    for (int i = l; i < 256; i += 64) out[i] = lds[i];
}

int main() {
    const int N = 4096;
    unsigned char* hs = new unsigned char[N];
    for (int i = 0; i < N; ++i) hs[i] = (unsigned char)(1 + i % 200);
    unsigned char* ds; unsigned *dv, *dout; unsigned hv[64], ho[256];
    hipMalloc(&ds, N); hipMalloc(&dv, 256); hipMalloc(&dout, 1024);
    hipMemcpy(ds, hs, N, hipMemcpyHostToDevice);
    struct { const char* name; unsigned nrec; unsigned soff; } cases[] = {
        {"A: nrec 2048, soff 0", 2048, 0}, {"B: nrec 2048, soff 1024 (voff < nrec <= voff + soff for lanes 64..)", 2048, 1024}};
    for (auto& c : cases) {
        for (int l = 0; l < 64; ++l) hv[l] = l * 16;
        hv[3] = 0xFFFFFFF0u; hv[10] = 0x80000000u; hv[20] = 2048; hv[21] = 2040; hv[40] = 4096 + 16;
        hipMemcpy(dv, hv, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, ds, c.nrec, dv, c.soff, dout);
        hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
        printf("%s\n", c.name);
        for (int l : {0, 2, 3, 10, 20, 21, 40, 63}) {
            unsigned exp0 = 0; unsigned o = hv[l] + c.soff;
            if (o + 4 <= (unsigned)N) memcpy(&exp0, hs + o, 4);
            printf("  lane %2d voff %08x: lds = %08x %08x %08x %08x   (source bytes there: %08x)\n", l, hv[l], ho[l * 4], ho[l * 4 + 1],
                   ho[l * 4 + 2], ho[l * 4 + 3], exp0);
        }
    }
    return 0;
}
