/* cellvit_amd.h — C ABI of the MI355X-native CellViT inference hot path.
 *
 * The reference (TIO-IKIM/CellViT) is pure Python and has NO FFI / plugin layer: its seam is the
 * nn.Module API.  Each entry point below cites the reference interface it replaces; the Python shim
 * (cellvit_amd/model.py) and any other host (C, C++, cgo, JNI ...) bind exactly these symbols.
 *
 * Conventions
 *   - every function returns an int status (CV_OK == 0); cv_last_error() gives the message.
 *   - the CALLER owns all input/output device buffers (e.g. torch tensors); the library borrows the
 *     pointers for the duration of the call and owns only its packed weights and workspace.
 *   - work is enqueued on the caller's HIP stream (`stream` is a hipStream_t passed as void*;
 *     NULL = default stream).  No hidden synchronisation except where a host-visible value is
 *     returned (cv_postproc_* result counts, cv_debug_read).
 *   - one handle per (process, device); a handle is not thread safe.
 */
#ifndef CELLVIT_AMD_H
#define CELLVIT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    CV_OK = 0,
    CV_ERR_INVALID = 1,      /* bad argument                                   -> ValueError          */
    CV_ERR_HIP = 2,          /* HIP runtime failure                             -> RuntimeError        */
    CV_ERR_STATE = 3,        /* call order (not finalized, geometry not set)    -> RuntimeError        */
    CV_ERR_SHAPE = 4,        /* H or W not divisible by the patch size          -> AssertionError      */
                             /*   (reference: cellvit.py:170-175, 603-608)                             */
    CV_ERR_UNSUPPORTED = 5,  /* unknown arch / magnification                    -> NotImplementedError */
                             /*   (reference: cellvit.py:530, post_proc_cellvit.py:61-62)              */
    CV_ERR_MISSING_WEIGHT = 6/* state_dict key absent / wrong shape             -> RuntimeError        */
};

enum { CV_ARCH_VIT = 0, CV_ARCH_SAM = 1 };
enum { CV_DTYPE_F16 = 0, CV_DTYPE_F32 = 1,     /* storage type of activations/weights; MFMA accumulates fp32 */
       CV_DTYPE_F8 = 2 };                      /* the fp16 engine with OCP MX-fp8 (e4m3 + E8M0 block scales, CDNA4
                                                  v_mfma_scale_f32_16x16x128_f8f6f4) qkv / fc1 / fc2 contractions in the
                                                  SAM encoders; attention core, proj and the decoder stay fp16
                                                  (BASELINE.json configs[4]).  cv_create only; the op-level entry points
                                                  take CV_DTYPE_F16 / CV_DTYPE_F32.                                     */

typedef struct cv_handle cv_handle;

/* Model hyper-parameters.  Replaces the constructor arguments of
 *   CellViT(...)      models/segmentation/cell_segmentation/cellvit.py:57-75
 *   CellViT256(...)   cellvit.py:444-479      CellViTSAM(...)  cellvit.py:514-572, 646-665          */
typedef struct cv_config {
    int32_t arch;                 /* CV_ARCH_VIT | CV_ARCH_SAM */
    int32_t embed_dim, depth, num_heads, mlp_ratio;
    int32_t extract_layers[4];    /* 1-based block indices of the skip connections */
    int32_t num_nuclei_classes, num_tissue_classes;
    int32_t regression_loss;      /* 1: binary branch has 2 extra channels (cellvit.py:133-141) */
    int32_t patch_size;           /* 16 */
    int32_t window_size;          /* SAM: 14 */
    int32_t n_global;             /* SAM: number of global-attention blocks */
    int32_t global_attn_indexes[8];
    int32_t neck_chans;           /* SAM: 256 */
    int32_t compute_dtype;        /* CV_DTYPE_F16 (production) | CV_DTYPE_F32 (parity/debug) | CV_DTYPE_F8 */
} cv_config;

/* Device buffers for the outputs of one forward call (fp32, contiguous, caller-allocated; NULL = skip).
 * Shapes follow the dict returned by CellViT.forward (cellvit.py:160-169).                            */
typedef struct cv_outputs {
    float* tissue_types;          /* [B, num_tissue_classes]; num_tissue_classes == 0 (head = nn.Identity, vits_histo.py:359-362,
                                     cellvit.py:568-572): the pooled embedding, [B, embed_dim] (ViT) / [B, neck_chans] (SAM) */
    float* nuclei_binary_map;     /* [B, 2, H, W]                                  */
    float* hv_map;                /* [B, 2, H, W]                                  */
    float* nuclei_type_map;       /* [B, num_nuclei_classes, H, W]                 */
    float* regression_map;        /* [B, 2, H, W] if regression_loss               */
    float* tokens_nhwc;           /* [B, H/16, W/16, D]; the shim returns .permute(0,3,1,2) == z4 */
    uint8_t* binary_argmax;       /* [B, H, W] argmax over nuclei_binary_map channels (cellvit.py:372) */
    uint8_t* type_argmax;         /* [B, H, W] argmax over nuclei_type_map channels   (cellvit.py:369) */
} cv_outputs;

const char* cv_last_error(void);
/* 1 if the library was built with -DCVA_ABLATION (experiment switches CVA_* honoured, incl. work-skipping *_DBG
 * instantiations); 0 for the production build, which ignores the environment.  bench.py refuses ablation builds.   */
int cv_build_is_ablation(void);
/* The compiler flags this library was built with (cellvit_amd/build.py records them); bench.py prints them in its line. */
const char* cv_build_flags(void);

/* nn.Module construction — cellvit.py:57-151 / 514-572. */
int cv_create(const cv_config* cfg, cv_handle** out);
int cv_destroy(cv_handle* h);

/* model.load_state_dict(ckpt["model_state_dict"]) — cell_detection.py:137, key names as produced by
 * the reference modules (base_trainer.py:229-245).  host_ptr: contiguous fp32 (dtype 1) or int64
 * (dtype 2, only num_batches_tracked) host data.  Unknown keys are rejected.                          */
int cv_load_weight(cv_handle* h, const char* ref_key, const void* host_ptr, int dtype,
                   const int64_t* shape, int ndim);

/* model.eval().to(device) — cell_detection.py:138-140: fold BatchNorm+bias into the conv weights,
 * repack ConvTranspose2d as 4 pointwise GEMMs, cast to the compute dtype, upload.                     */
int cv_finalize(cv_handle* h);

/* Fix the input geometry (allocates the workspace).  The reference derives these per call from
 * x.shape; a fixed geometry lets all launches be enqueued without host round trips.                   */
int cv_set_geometry(cv_handle* h, int max_batch, int H, int W);

/* Input-size dependent tables, fp32 host data, computed by the caller with the reference's own
 * formulas (the shim uses the identical torch ops):
 *   "pos_table"      [ntok, D]   ViT: interpolate_pos_encoding (vits_histo.py:377-402), cls row first
 *                                 SAM: pos_embed[:, :gh, :gw, :] (cell_segmentation/utils.py:222-224)
 *   "rel_h.<i>"      [2*KH-1, hd]  get_rel_pos resize of block i (SAM/image_encoder.py:333-344)
 *   "rel_w.<i>"      [2*KW-1, hd]                                                                     */
int cv_set_derived(cv_handle* h, const char* name, const float* host_ptr, const int64_t* shape, int ndim);

/* CellViT.forward(x, retrieve_tokens) — cellvit.py:153-210 (ViT), :586-644 (SAM).
 * x_dev: fp32 NCHW [B,3,H,W] normalised tile batch on the device.                                     */
int cv_forward(cv_handle* h, const float* x_dev, int B, int H, int W, const cv_outputs* out, void* stream);

/* The same forward on the RAW tile: x_u8 uint8 NHWC [B,H,W,3] on the device.  The reference's inference transform
 * (T.ToTensor + T.Normalize(mean, std), cell_detection.py:214-227: (u8 / 255 - mean[c]) / std[c] in fp32) is evaluated
 * inside the two kernels that read the image (patch matrix, decoder-0 NHWC loader); no normalised copy reaches HBM.     */
int cv_forward_u8(cv_handle* h, const uint8_t* x_u8, const float* mean3, const float* std3, int B, int H, int W,
                  const cv_outputs* out, void* stream);

/* Debug taps (synchronises): copy a named intermediate of the LAST forward to host memory as fp32.
 * names: "tokens0", "block<i>", "z<1..4>", "skip<0..3>".  Returns the element count in *n_out.        */
int cv_set_debug(cv_handle* h, int enable);
/* Engine choices of the current geometry (0 before cv_set_geometry): bit 0 = the window blocks keep V row-major (qkv epilogue + window
 * attention kernel with the transposing LDS read), bit 1 = fp8 engine with proj on MX-fp8 (the attention kernels emit MX-fp8 rows).   */
int cv_geometry_flags(const cv_handle* h);
/* Run-time engine options (production builds; must precede cv_set_geometry — a later call returns CV_ERR_STATE):
 *   "fp8_proj" = 0 : the fp8 engine keeps attn.proj on fp16 (qkv / fc1 / fc2 on MX-fp8 only: the tighter accuracy bounds of
 *                    tests/test_gpu_fp8.py's second leg); 1 (default): proj on MX-fp8 as well where the geometry allows it.
 * No counterpart in the reference (its only reduced precision is torch.autocast fp16, cell_detection.py:314-316).            */
int cv_set_option(cv_handle* h, const char* name, int value);
int cv_debug_read(cv_handle* h, const char* name, float* host_dst, size_t capacity, size_t* n_out);

/* Stage events of the forward, for a caller that runs other device work (the post-processing of the PREVIOUS batch, cell_detection.py:306-421
 * has the two back to back) on a second stream: the encoder's GEMMs are persistent whole-CU workgroups that a co-running latency-bound chain
 * only time-slices with, the full-resolution decoder stages are short two-per-CU workgroups it interleaves with.
 *   stage 0: arm (create the events; every later forward records them on its stream);
 *   stage 1: `stream` waits until the most recent forward has finished its encoder (shared skip decoders start);
 *   stage 2: ... until its first branch has reached the full-resolution stages.
 * CV_ERR_STATE if no forward has recorded the stage yet.  No counterpart in the reference (single stream).                              */
int cv_stream_wait_stage(cv_handle* h, int stage, void* stream);

/* Live per-kernel-class timing (HIP events recorded on the launch stream around every launch of the
 * class while enabled).  Classes: 0 linear GEMM (fp16), 1 QKV GEMM (fp16), 2 conv3x3, 3 convT2x2, 4 attention,
 * 5 MX-fp8 GEMMs of the fp8 engine (qkv, fc1, fc2).
 * cv_profile_collect synchronises, fills three arrays of 6 (ms, launches, algorithmic FLOPs) and resets. */
int cv_profile_enable(cv_handle* h, int on);
int cv_profile_collect(cv_handle* h, double* total_ms, int64_t* launches, double* flops);

/* ---- single-operator entry points (the -m gpu parity tests drive the kernels through these) ---- */
/* out[M,N] = act(A[M,K] · W[N,K]^T + bias) (+ residual); A, W device buffers of `dtype`.              */
int cv_op_linear(int dtype, const void* A, const void* W, const float* bias, const float* residual,
                 void* out, int out_f32, int M, int N, int K, int act, void* stream);
int cv_op_layernorm(int dtype, const float* x, const float* gamma, const float* beta, void* out,
                    int out_f32, int M, int C, float eps, void* stream);
/* fp16 engine: x_io[M,C] (fp32) += delta[M,C] (fp16, the projection's output); out[M,C] (fp16, may alias delta) =
 * LayerNorm(x_io).  The residual add of vits_histo.py:236 / image_encoder.py:182 fused into the norm that follows it.   */
int cv_op_layernorm_add(float* x_io, const void* delta_f16, const float* gamma, const float* beta, void* out_f16,
                        int M, int C, float eps, void* stream);
/* NHWC 3x3 conv (pad 1) over the channel concat of src1 (C1) and src2 (C2, may be NULL/0);
 * Wk: [Cout, 9*(C1+C2)] of `dtype`, k = tap*(C1+C2) + c; bias fp32 [Cout] or NULL.                     */
int cv_op_conv3x3(int dtype, const void* src1, int C1, const void* src2, int C2, const void* Wk,
                  const float* bias, void* out, int out_f32, int B, int H, int W, int Cout, int relu,
                  void* stream);
/* NHWC ConvTranspose2d k2 s2: Wk [4*Cout, Cin] (n = (dy*2+dx)*Cout + co), bias4 fp32 [4*Cout].         */
int cv_op_convT2x2(int dtype, const void* src, const void* Wk, const float* bias4, void* out,
                   int B, int H, int W, int Cin, int Cout, void* stream);
/* ConvTranspose2d k2 s2 -> Conv2d 3x3 pad 1 -> BatchNorm2d (eval) -> ReLU as ONE launch on the fp16 engine: the two linear maps are
 * composed into a contraction over the INPUT pixels.  Cs == 0: Deconv2DBlock (models/segmentation/cell_segmentation/
 * utils.py:46-86; dropout is identity in eval).  Cs > 0: the decoder stages of cellvit.py:255-304, where the convolution runs on
 * torch.cat([skip, up-sampled], dim=1) (cellvit.py:236-242) — `skip` fp16 NHWC [B, 2H, 2W, Cs] on the device.
 * Weights are HOST fp32 arrays in the reference's layouts (wt [Cin, Cup, 2, 2], w3 [Cout, Cs + Cup, 3, 3]); src fp16 NHWC [B, H, W, Cin],
 * out fp16 NHWC [B, 2H, 2W, Cout] on the device.  Cout % 256 == 0, Cin % 64 == 0, Cs % 64 == 0, (4 Cin + 9 Cs) / 64 even, power-of-two
 * H, W with H*W >= 256, else CV_ERR_UNSUPPORTED.  Synchronises the stream (the composed weights are temporary).                          */
int cv_op_deconv_block(const float* wt, const float* bt, const float* w3, const float* b3, const float* bn_weight,
                       const float* bn_bias, const float* bn_mean, const float* bn_var, const void* src, const void* skip,
                       void* out, int B, int H, int W, int Cin, int Cup, int Cs, int Cout, void* stream);
/* One attention layer on token rows x[B*ntok, D] (already normalised): qkv GEMM + scatter, optional
 * window partition (win > 0, zero-padded tokens) and decomposed rel-pos (tab_h/tab_w != NULL).
 * out: `dtype` [B*ntok, D] = softmax(...)·v re-assembled in token order (before the output proj).     */
int cv_op_attention(int dtype, const void* x, const void* Wqkv, const float* bqkv, const float* tab_h,
                    const float* tab_w, void* out, int B, int gh, int gw, int has_cls, int heads,
                    int D, int win, void* stream);

/* ---- fp8 engine (CV_DTYPE_F8), single operators.  MX-fp8 tensor = e4m3 bytes [rows, K] row-major + one E8M0 scale byte
 * (2^(b - 127)) per 32 K elements.  Scale bytes live in 1-KiB blocks per (256-row tile, 128-element K tile) in the order the
 * MFMA lanes read them ("A-side" / "W-side" images, cellvit_amd/csrc/gemm.h: mx8_scale_index).                            */
/* Host-side reference quantiser (what cv_finalize applies to the qkv / fc1 / fc2 weights): layout 0 = A-side image,
 * 1 = W-side image, 2 = plain row-major [rows, K/32] (for checking against other implementations).                       */
int cv_mx8_quantize_host(const float* x, int rows, int K, int layout, uint8_t* data, uint8_t* scales);
/* out[M,N] = act(A8 . W8^T + bias) (+ residual fp32).  A scales: A-side image; W scales: W-side image.  out_kind 0: fp16,
 * 1: fp32, 2: MX-fp8 (bytes to `out`, A-side scale image for a consumer with K = N to `out_scale`).  M, N % 256, K % 256.   */
int cv_op_linear_mx8(const void* A8, const void* a_scale_a, const void* a_scale_w, const void* W8, const void* w_scale,
                     const float* bias, const float* residual, void* out, int out_kind, void* out_scale, int M, int N, int K,
                     int act, void* stream);
/* LayerNorm(x_io (+= delta_f16 when given, written back)) -> MX-fp8 rows; scale_w (W-side image) may be NULL.                */
int cv_op_layernorm_mx8(float* x_io, const void* delta_f16, const float* gamma, const float* beta, void* out8, void* scale_a,
                        void* scale_w, int M, int C, float eps, void* stream);
/* cv_op_attention with the fused qkv projection on MX-fp8 operands (x8 + both scale images from cv_op_layernorm_mx8;
 * Wqkv8 with rows >= 2D packed in the A-side image, the others W-side); no cls token.  out: fp16 [B*gh*gw, D].              */
int cv_op_attention_mx8(const void* x8, const void* scale_a, const void* scale_w, const void* Wqkv8, const void* wqkv_scale,
                        const float* bqkv, const float* tab_h, const float* tab_w, void* out, int B, int gh, int gw, int heads,
                        int D, int win, void* stream);

/* The fp16 attention layer (as cv_op_attention, no cls token, hd = 80) with the MX-fp8 row epilogue that feeds the fp8 engine's proj:
 * out8 e4m3 [B*gh*gw, 96 * heads] — head h owns columns [96 h, 96 h + 96) = its 80 values + 16 zeros, so that no 32-element scale block
 * straddles two heads; the caller zero-fills it (pad columns are never written) — and out8_scale, the E8M0 scale image in the A-side order
 * of the 8-phase kernel (3 * heads scales per row).  Replaces the fp16 `attn_out` of SAM/image_encoder.py:255-258 on that engine.      */
int cv_op_attention_rows_mx8(const void* x, const void* Wqkv, const float* bqkv, const float* tab_h, const float* tab_w, void* out8,
                             void* out8_scale, int B, int gh, int gw, int heads, int D, int win, void* stream);

/* argmax over dim 1 of an fp32 NCHW map -> u8 [B,H,W], first maximum (torch.argmax of cellvit.py:366-374).           */
int cv_op_argmax_nchw(const float* x, uint8_t* out, int B, int C, int H, int W, void* stream);
/* u8 NHWC [B,H,W,3] -> fp32 NCHW [B,3,H,W], the inference transform of cell_detection.py:214-227 as a stand-alone op.   */
int cv_op_normalize_u8(const uint8_t* x_u8, const float* mean3, const float* std3, float* out, int B, int H, int W,
                       void* stream);

/* ---- per-tile instance post-processing ------------------------------------------------------------
 * Replaces DetectionCellPostProcessor.post_process_cell_segmentation + __proc_np_hv
 *   (cell_segmentation/utils/post_proc_cellvit.py:67-153, 155-249) and the per-sample glue of
 *   CellViT.calculate_instance_map (cellvit.py:351-383), batched and fully on the device.            */
typedef struct cv_instance {
    int32_t id;                        /* instance id == surviving marker id (ids are not compacted)  */
    int32_t rmin, cmin, rmax, cmax;    /* bbox [[rmin,cmin],[rmax,cmax]], max exclusive (tools.py:24-34) */
    int32_t npix;
    int32_t type;                      /* majority nucleus type, background replaced by the runner-up  */
    int32_t contour_off, contour_len;  /* into the tile's (x, y) int32 contour array; len < 3 => the
                                          reference drops the instance from its dict (post_proc:113-116) */
    int32_t reserved;
    double cx, cy;                     /* centroid (x, y), tile coordinates                             */
    double type_prob;                  /* votes(type) / (npix + 1e-6)  (post_proc:149)                   */
} cv_instance;

typedef struct cv_pp cv_pp;
/* max_inst record slots and max_pts contour points per tile; nr_types <= 256 (u8 type planes; more than 8 classes vote in windows of 8). */
int cv_pp_create(int max_batch, int H, int W, int max_inst, int max_pts, cv_pp** out);
int cv_pp_destroy(cv_pp* pp);
/* Device inputs: bin_argmax / type_argmax u8 [B,H,W] (cellvit.py:369-374), hv f32 [B,2,H,W].
 * Device outputs: inst_map i32 [B,H,W]; recs [B,max_inst]; n_recs, n_pts i32 [B];
 * contours i32 [B,max_pts,2] (may be NULL).  magnification 40 | 20 else CV_ERR_UNSUPPORTED.            */
int cv_pp_run(cv_pp* pp, const uint8_t* bin_argmax, const uint8_t* type_argmax, const float* hv, int B,
              int magnification, int nr_types, int32_t* inst_map, cv_instance* recs, int32_t* n_recs,
              int32_t* contours, int32_t* n_pts, void* stream);
/* Same with explicit (object_size, Sobel ksize in {21, 11}) — DetectionCellPostProcessor(gt=True)
 * uses (100, 21) (post_proc_cellvit.py:63-65).                                                         */
int cv_pp_run_params(cv_pp* pp, const uint8_t* bin_argmax, const uint8_t* type_argmax, const float* hv, int B,
                     int object_size, int ksize, int nr_types, int32_t* inst_map, cv_instance* recs,
                     int32_t* n_recs, int32_t* contours, int32_t* n_pts, void* stream);
/* Records + contours of GIVEN instance maps: replaces calculate_instances (post_proc_cellvit.py:252-330, the
 * ground-truth side of the evaluation callers, inference_cellvit_experiment_pannuke.py:744-746).  inst_map i32 [B,H,W]
 * (device; ids <= 0 are background, negative ones are zeroed in place; ids above H*W/16 have no slot and are not
 * reported), type_map u8 [B,H,W] (argmax of the one-hot type map; NULL with nr_types 0).  Outputs as cv_pp_run.       */
int cv_pp_records(cv_pp* pp, int32_t* inst_map, const uint8_t* type_map, int B, int nr_types, cv_instance* recs,
                  int32_t* n_recs, int32_t* contours, int32_t* n_pts, void* stream);
/* Slide-level de-duplication of margin cells — replaces the shapely STRtree queries + polygon intersections of
 * CellPostProcessor._remove_overlap (cell_segmentation/inference/cell_detection.py:676-767) for the cells of one slide.
 * All pointers are DEVICE pointers except n_pairs_host.  bbox i32 [n,4] = (rmin, cmin, rmax, cmax) in slide coordinates,
 * ct_off i64 [n+1] = contour offsets (points) into ct_xy i32 [*,2] = (x, y) contour points in slide coordinates, extent =
 * HOST i32 [4] (min row, min col, max row, max col over all boxes).  Outputs: pairs i32 [cap,2] = candidate pairs (i < j) whose
 * boxes overlap strictly (order unspecified), inter f64 [cap] = EXACT area of the intersection of the two contour polygons
 * (even-odd interiors; -1 where a fixed per-thread capacity was exceeded: evaluate that pair on the host), area f64 [n] = polygon
 * areas.  Synchronises the stream (the pair count is returned to the host); CV_ERR_SHAPE when more than cap pairs exist —
 * n_pairs_host then holds the required capacity.                                                                                 */
int cv_stitch_overlaps(const int32_t* bbox, const int64_t* ct_off, const int32_t* ct_xy, int n, const int32_t* extent,
                       int32_t* pairs, double* inter, double* area, int cap, int32_t* n_pairs_host, void* stream);
/* Invalid (self-touching) contour rings of the same function (:689-704: `if not poly.is_valid` -> `poly.buffer(0)` -> the part of the
 * largest area).  cv_stitch_ring_flags — DEVICE pointers: flags u8 [n] = 1 where ring i (ct_off / ct_xy as above) is not simple
 * (two non-adjacent edges share a point, consecutive edges fold back, a vertex repeats).  cv_stitch_repair_rings — HOST pointers,
 * host code: the rings with flags[i] != 0 (flags NULL: every ring is examined) whose lattice chain visits a lattice point twice
 * are replaced by their largest simple lobe; out_off i64 [n+1], out_xy i32 [ct_off[n], 2] receive all rings; n_repaired may be NULL. */
int cv_stitch_ring_flags(const int64_t* ct_off, const int32_t* ct_xy, int n, uint8_t* flags, void* stream);
int cv_stitch_repair_rings(const int64_t* ct_off, const int32_t* ct_xy, int n, const uint8_t* flags, int64_t* out_off,
                           int32_t* out_xy, int32_t* n_repaired);
/* The greedy rounds of the same function (:706-767) over a pair list — HOST pointers, host code: per round cells are visited in
 * index order, a visited cell collects its live, not yet visited overlap partners and the largest of them (the first of equal
 * areas, np.argmax :743-746) survives in its place (the cell itself when it has none); stops after a round without overlaps or max_rounds (reference: 20).  overlap[k] != 0 marks
 * pairs overlapping by more than 1 % of either area; alive u8 [n] in/out; overlaps_out i32 [max_rounds] per-round counts.       */
int cv_stitch_select(const int32_t* pairs, const uint8_t* overlap, int n_pairs, const double* area, uint8_t* alive, int n,
                     int max_rounds, int32_t* rounds_out, int32_t* overlaps_out);
/* Writer of `cells.json` / `cell_detection.json` (cell_detection.py:438-457) from packed HOST arrays — host code, no device.
 * header = the rendered members '"wsi_metadata": ..., "processed_patches": ..., "type_map": ...' (without braces); per cell k:
 * bbox i64 [n,4] and centroid f64 [n,2] in slide coordinates, contour points ct_xy i64 [*,2] with offsets ct_off i64 [n+1],
 * type_prob, type, patch_rc i32 [n,2] = (row, col), status, offset_global i64 [n,2], edge u8 [n], edge_pos u8 [n,4] =
 * [top, right, down, left] (get_cell_position, :787-817; edge_patches follow from get_edge_patch, :877-902).
 * detection_only != 0 writes {bbox, centroid, type} per cell (the other arrays may be NULL).  Same JSON document as
 * json.dump of the reference's dicts (keys, order, values; doubles with 17 significant digits), one cell per line.           */
int cv_write_cells_json(const char* path, const char* header, int detection_only, int n, const int64_t* bbox,
                        const double* centroid, const int64_t* ct_off, const int64_t* ct_xy, const double* type_prob,
                        const int32_t* type, const int32_t* patch_rc, const int32_t* status, const int64_t* offset_global,
                        const uint8_t* edge, const uint8_t* edge_pos);
/* The optional geojson pair of the inference CLI (cell_segmentation/inference/cell_detection.py:538-597, `convert_geojson`, with
 * cell_segmentation/datamodel/template_geojson.py:9-52): a JSON list of one Feature per nucleus type present, ascending type order, whose geometry
 * collects the cells of that type in slide order — polygons != 0: MultiPolygon of the closed contour rings (ring + its first point, coordinates as
 * floats), else MultiPoint of the centroids.  feat_type int32 [n_feat]; feat_head[i] / feat_tail[i]: the caller-rendered text around the coordinate
 * list of feature i ('{"type": "Feature", "id": "...", "geometry": {"type": "MultiPolygon", "coordinates": [' and ']}, "properties": {...}}').
 * Arrays as for cv_write_cells_json (centroid may be NULL for polygons, the contour arrays for points).  Same document as json.dump of the
 * reference's list after parsing (the reference writes indent=2 and random ids).                                                      */
int cv_write_geojson(const char* path, int polygons, int n, const double* centroid, const int64_t* ct_off, const int64_t* ct_xy,
                     const int32_t* type, int n_feat, const int32_t* feat_type, const char* const* feat_head, const char* const* feat_tail);

/* Streaming slide tail (round 5; the reference's writers run after the whole tile loop, cell_detection.py:423-475).
 * cv_render_cells: the cells of ONE finished batch rendered once, one JSON object per cell (the text cv_write_cells_json emits between its
 *   ",\n" separators), into an opaque buffer with per-cell offsets; arguments as cv_write_cells_json.  Host code, releases nothing global.
 * cv_textbuf_compact: the kept cells (keep[k] != 0, NULL = all) of a buffer joined by ",\n" into dst; returns the byte count (also when
 *   dst is NULL / cap too small: nothing written) — the writer concatenates the batches' chunks in slide order between the file's header
 *   and footer: byte for byte cv_write_cells_json of the kept cells.
 * cv_write_rows: kept rows (row_bytes each) of n_chunks host arrays, in order, written at file_offset of an existing file with pwrite on
 *   several threads; *crc_out = CRC-32 (zip) of the bytes, *rows_out = rows written.  Fills the tensor holes of a cells.pt archive that
 *   torch.save wrote under torch.serialization.skip_data.                                                                              */
typedef struct cv_textbuf cv_textbuf;
int cv_render_cells(int detection_only, int n, const int64_t* bbox, const double* centroid, const int64_t* ct_off, const int64_t* ct_xy,
                    const double* type_prob, const int32_t* type, const int32_t* patch_rc, const int32_t* status,
                    const int64_t* offset_global, const uint8_t* edge, const uint8_t* edge_pos, cv_textbuf** out);
int64_t cv_textbuf_compact(const cv_textbuf* b, const uint8_t* keep, char* dst, int64_t cap);
void cv_textbuf_free(cv_textbuf* b);
int cv_write_rows(const char* path, int64_t file_offset, int64_t row_bytes, int n_chunks, const void* const* chunk_ptr,
                  const int64_t* chunk_rows, const uint8_t* const* keep, uint32_t* crc_out, int64_t* rows_out);
/* Cell-token pooling of the inference CLI (cell_detection.py:396-409) on the device record arrays of cv_pp_run:
 * out[rec_offset[b] + i, :] = mean over tokens_nhwc[b, floor(rmin/p):ceil(rmax/p), floor(cmin/p):ceil(cmax/p), :] for
 * record i < n_recs[b] (indices cast to uint8 as the reference does).  rec_offset: int64 [B] device (exclusive prefix
 * of n_recs, or any layout the caller wants), max_n = max_b n_recs[b] (grid size).  out fp32 [sum n_recs, D].          */
int cv_pool_tokens(const float* tokens_nhwc, int B, int gh, int gw, int D, int patch_size, const cv_instance* recs,
                   int max_inst, const int32_t* n_recs, const int64_t* rec_offset, int max_n, float* out, void* stream);
/* Debug taps of the last run (synchronises): "dist" f64 [B,H,W], "marker" i32 [B,H,W], "blb" u8 [B,H,W]. */
int cv_pp_debug_read(cv_pp* pp, const char* name, void* host_dst, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* CELLVIT_AMD_H */
