// On-device HoVer-Net post-processing (post_proc_cellvit.py:67-249) — host-side interface.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cva {

// Same field order as the C-ABI struct cv_instance (include/cellvit_amd.h).
struct InstanceRec {
    int32_t id;
    int32_t rmin, cmin, rmax, cmax;   // max exclusive (tools.py:24-34)
    int32_t npix;
    int32_t type;
    int32_t contour_off, contour_len;
    int32_t _pad;
    double cx, cy;
    double type_prob;
};

struct PostprocWorkspace;   // opaque, owned by the pp handle

struct PostprocDims {
    int B, H, W;
    int max_inst;    // record slots per tile
    int max_pts;     // contour points per tile
    int max_ids;     // upper bound of marker ids per tile
};

int pp_workspace_create(const PostprocDims& d, PostprocWorkspace** out);
void pp_workspace_destroy(PostprocWorkspace* ws);
size_t pp_workspace_bytes(const PostprocWorkspace* ws);

// Enqueue the whole chain for B tiles.  bin/type: u8 [B,H,W]; hv: f32 [B,2,H,W];
// inst_out: i32 [B,H,W]; recs: [B,max_inst]; n_recs/n_pts: i32 [B] (device); contours: i32 [B,max_pts,2].
int pp_run(PostprocWorkspace* ws, const uint8_t* bin, const uint8_t* type, const float* hv, int B, int object_size,
           int ksize, int nr_types, int32_t* inst_out, InstanceRec* recs, int32_t* n_recs, int32_t* contours,
           int32_t* n_pts, hipStream_t stream);

// Records + contours of caller-supplied instance maps (i32 [B,H,W], modified in place: negative ids -> 0), the P7/P8 tail only.
int pp_records(PostprocWorkspace* ws, int32_t* inst_io, const uint8_t* type, int B, int nr_types, InstanceRec* recs,
               int32_t* n_recs, int32_t* contours, int32_t* n_pts, hipStream_t stream);

// debug taps of the last run (device pointers, valid until the next run)
const int32_t* pp_dbg_blb(const PostprocWorkspace* ws);     // u8 promoted? no: int32 not stored; see .hip
const double* pp_dbg_dist(const PostprocWorkspace* ws);
const int32_t* pp_dbg_marker(const PostprocWorkspace* ws);
const uint8_t* pp_dbg_blb_u8(const PostprocWorkspace* ws);

}  // namespace cva
