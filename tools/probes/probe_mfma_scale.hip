// Empirical semantics of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_mfma_scale.hip -o /tmp/probe && /tmp/probe
// Answers: which row/col a lane's operand belongs to, whether lane (g = l>>4) holds K block g for both operands, which
// lane's scale VGPR byte (op_sel) scales which K block of which row.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OA, int OB>
__global__ void k(const unsigned char* a, const unsigned char* b, const int* sa, const int* sb, float* c) {
    const int l = threadIdx.x;
    i32x8 A, B;
    memcpy(&A, a + l * 32, 32);
    memcpy(&B, b + l * 32, 32);
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, 0, 0, OA, sa[l], OB, sb[l]);
    // C layout: col = lane & 15, row = (lane >> 4) * 4 + r
    for (int r = 0; r < 4; ++r) c[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}

static unsigned char ha[64 * 32], hb[64 * 32];
static int hsa[64], hsb[64];
static float hc[256];
template <int OA, int OB> static void run() {
    unsigned char *da, *db; int *dsa, *dsb; float* dc;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dsa, sizeof(hsa)); hipMalloc(&dsb, sizeof(hsb)); hipMalloc(&dc, sizeof(hc));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, sizeof(hsa), hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof(hsb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<OA, OB>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
    hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dc);
}
static void fill(unsigned char* p, unsigned char v) { memset(p, v, 64 * 32); }
static void scales(int* s, int v) { for (int i = 0; i < 64; ++i) s[i] = v; }
static const int S1 = 0x7f7f7f7f;   // all four bytes 2^0

int main() {
    // T1: all ones -> 128 everywhere
    fill(ha, 0x38); fill(hb, 0x38); scales(hsa, S1); scales(hsb, S1);
    run<0, 0>();
    printf("T1 all-ones: C[0][0]=%g C[5][9]=%g C[15][15]=%g (expect 128)\n", hc[0], hc[5 * 16 + 9], hc[255]);
    // T2: one-hot A element (lane L, byte p) with B all ones: which output row lights up (value 1 across the row)
    for (int L : {0, 1, 15, 16, 17, 33, 63}) {
        fill(ha, 0); fill(hb, 0x38); ha[L * 32 + 5] = 0x38; run<0, 0>();
        int row = -1, cnt = 0; for (int i = 0; i < 256; ++i) if (hc[i] != 0) { ++cnt; row = i / 16; }
        printf("T2 A one-hot lane %2d byte 5: %d nonzero outputs, row %d (C[row][0]=%g)\n", L, cnt, row, row >= 0 ? hc[row * 16] : 0.f);
    }
    // T3: same for B (A all ones): which output column
    for (int L : {0, 1, 15, 16, 33, 63}) {
        fill(ha, 0x38); fill(hb, 0); hb[L * 32 + 7] = 0x38; run<0, 0>();
        int col = -1, cnt = 0; for (int i = 0; i < 256; ++i) if (hc[i] != 0) { ++cnt; col = i % 16; }
        printf("T3 B one-hot lane %2d byte 7: %d nonzero outputs, col %d\n", L, cnt, col);
    }
    // T4: K correspondence: A one-hot (lane ga*16+3, byte pa), B one-hot (lane gb*16+4, byte pb): nonzero iff same k
    printf("T4 k-match (A lane g*16+3 byte p) x (B lane g'*16+4 byte p'):\n");
    for (int ga = 0; ga < 4; ++ga) for (int pa : {0, 9, 31})
        for (int gb = 0; gb < 4; ++gb) for (int pb : {0, 9, 31}) {
            fill(ha, 0); fill(hb, 0); ha[(ga * 16 + 3) * 32 + pa] = 0x38; hb[(gb * 16 + 4) * 32 + pb] = 0x38; run<0, 0>();
            if (hc[3 * 16 + 4] != 0) printf("   g=%d p=%2d  matches  g'=%d p'=%2d  (C[3][4]=%g)\n", ga, pa, gb, pb, hc[3 * 16 + 4]);
        }
    // T5: whose scale_a byte scales which (row, k block): all ones, then lane L's scale_a byte `by` := 2^1
    printf("T5 scale_a (op_sel 0): set lane L byte 0 to x2 -> change of row sums (C[row][0] - 128):\n");
    for (int L : {0, 3, 16, 19, 35, 51}) {
        fill(ha, 0x38); fill(hb, 0x38); scales(hsa, S1); scales(hsb, S1); hsa[L] = (S1 & ~0xff) | 0x80; run<0, 0>();
        printf("   L=%2d:", L); for (int r = 0; r < 16; ++r) if (hc[r * 16] != 128) printf(" row %d: %+g", r, hc[r * 16] - 128); printf("\n");
    }
    printf("T6 scale_b (op_sel 0): set lane L byte 0 to x2 -> change of C[0][col] - 128:\n");
    for (int L : {0, 3, 16, 19, 35, 51}) {
        fill(ha, 0x38); fill(hb, 0x38); scales(hsa, S1); scales(hsb, S1); hsb[L] = (S1 & ~0xff) | 0x80; run<0, 0>();
        printf("   L=%2d:", L); for (int cc = 0; cc < 16; ++cc) if (hc[cc] != 128) printf(" col %d: %+g", cc, hc[cc] - 128); printf("\n");
    }
    // T7: op_sel = byte select?  lane 3 scale_a bytes (x1, x2, x4, x8): row 3 sum with op_sel 0..3
    fill(ha, 0x38); fill(hb, 0x38); scales(hsa, S1); scales(hsb, S1);
    for (int L = 0; L < 64; ++L) hsa[L] = 0x827f8180 | 0;     // bytes: b0 = 0x80 (x2), b1 = 0x81 (x4), b2 = 0x7f (x1), b3 = 0x82 (x8)
    run<0, 0>(); printf("T7 op_sel_a=0: C[3][0]=%g (x2 -> 256)\n", hc[3 * 16]);
    run<1, 0>(); printf("T7 op_sel_a=1: C[3][0]=%g (x4 -> 512)\n", hc[3 * 16]);
    run<2, 0>(); printf("T7 op_sel_a=2: C[3][0]=%g (x1 -> 128)\n", hc[3 * 16]);
    run<3, 0>(); printf("T7 op_sel_a=3: C[3][0]=%g (x8 -> 1024)\n", hc[3 * 16]);
    for (int L = 0; L < 64; ++L) { hsa[L] = S1; hsb[L] = 0x827f8180; }
    run<0, 1>(); printf("T7 op_sel_b=1: C[0][3]=%g (x4 -> 512)\n", hc[3]);
    run<0, 3>(); printf("T7 op_sel_b=3: C[0][3]=%g (x8 -> 1024)\n", hc[3]);
    // T8: which lane's scale_a multiplies element (lane La, byte p) of A?  B all ones, unit scales except one lane x2.
    for (int La : {3, 19, 35, 51}) for (int pb : {5, 20}) {
        printf("T8 A one-hot (lane %2d byte %2d): scaled by scale_a of lane", La, pb);
        for (int Ls = 3; Ls < 64; Ls += 16) {
            fill(ha, 0); fill(hb, 0x38); ha[La * 32 + pb] = 0x38; scales(hsa, S1); scales(hsb, S1); hsa[Ls] = (S1 & ~0xff) | 0x80;
            run<0, 0>();
            if (hc[3 * 16] == 2.f) printf(" %d", Ls);
        }
        printf("\n");
    }
    return 0;
}
