// Shared device/host helpers for the cellvit_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace cva {

typedef _Float16 half_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

// Experiment switches (kernel-variant A/B selectors and *_DBG ablations that SKIP parts of a kernel) exist only in a
// library built with -DCVA_ABLATION (python -m cellvit_amd.build --ablation).  The production library ignores the
// environment altogether: a timed region can not be one variable away from skipping work.
#ifdef CVA_ABLATION
inline int cva_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
constexpr int CVA_ABLATION_BUILD = 1;
#else
inline int cva_env_int(const char*, int dflt) { return dflt; }
constexpr int CVA_ABLATION_BUILD = 0;
#endif

// ---------------------------------------------------------------------------------------------
// Storage-type traits.  T = half_t is the production path (fp16 storage, fp32 MFMA accumulate,
// mirrors torch.autocast(float16) of the reference, cell_detection.py:314-316); T = float is the
// exact-fp32 parity path (v_mfma_f32_16x16x4_f32 == k-ordered fmaf chain).
// A "piece" is one 16-byte global/LDS transaction: 8 halves or 4 floats.
// A "frag" is the per-lane A/B operand of one 16x16x32 MMA step: 8 consecutive K elements.
// ---------------------------------------------------------------------------------------------
template <typename T> struct Traits;

template <> struct Traits<half_t> {
    static constexpr int PIECE = 8;          // elements per 16 B
    static constexpr int BK = 64;            // K elements per LDS tile row (128 B)
    struct Frag { half8_t v; };
    static __device__ __forceinline__ Frag load_frag(const half_t* p) {   // 16-byte aligned
        Frag f; f.v = *reinterpret_cast<const half8_t*>(p); return f;
    }
    static __device__ __forceinline__ Frag zero_frag() { Frag f; f.v = (half8_t)(0); return f; }
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c, 0, 0, 0);
    }
    static __device__ __forceinline__ half_t from_float(float x) { return (half_t)x; }
    static __device__ __forceinline__ float to_float(half_t x) { return (float)x; }
};

template <> struct Traits<float> {
    static constexpr int PIECE = 4;
    static constexpr int BK = 32;            // 128 B rows as well
    struct Frag { float v[8]; };
    static __device__ __forceinline__ Frag load_frag(const float* p) {    // 16-byte aligned
        Frag f;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
        f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
        f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
        return f;
    }
    static __device__ __forceinline__ Frag zero_frag() {
        Frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f.v[j] = 0.f;
        return f;
    }
    // Lane l holds K elements (l>>4)*8 + j, j = 0..7, for BOTH operands, so the j-th
    // 16x16x4 MFMA contracts k in {j, 8+j, 16+j, 24+j}: the union over j is the full 32-deep
    // step and every product a*b is formed exactly once in fp32.
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x4& c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
    }
    static __device__ __forceinline__ float from_float(float x) { return x; }
    static __device__ __forceinline__ float to_float(float x) { return x; }
};

// LDS row pitch (in elements) of a [rows][KW] tile whose rows are KW elements wide: row bytes
// + 32 B.  With row bytes a multiple of 64 the pitch/16 is ≡ 2 (mod 4), which makes the four
// 16-lane groups of a ds_read_b128 fragment read (rows l&15, 16-B column l>>4) bank-conflict free.
template <typename T> __host__ __device__ constexpr int lds_pitch(int kw) { return kw + 32 / (int)sizeof(T); }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

struct alignas(16) Piece { uint32_t w[4]; };
__device__ __forceinline__ Piece zero_piece() { Piece p; p.w[0] = p.w[1] = p.w[2] = p.w[3] = 0u; return p; }
__device__ __forceinline__ Piece load_piece(const void* p) { return *reinterpret_cast<const Piece*>(p); }
__device__ __forceinline__ void store_piece(void* p, const Piece& v) { *reinterpret_cast<Piece*>(p) = v; }

// XCD-aware remap of a linear workgroup id: hardware round-robins consecutive ids over the 8 XCDs,
// so give each XCD a contiguous chunk of the logical tile space (bijective for any count).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace cva

#define CVA_CHECK_HIP(expr)                                                                 \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) { cva_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, \
                                              hipGetErrorString(_e)); return CV_ERR_HIP; }  \
    } while (0)
