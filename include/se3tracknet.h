/* se3tracknet.h -- C ABI of the MI355X (gfx950) se(3)-TrackNet per-frame inference hot path.
 *
 * The reference (wenbowen123/iros20-6d-pose-tracking) has no FFI: its boundary is the Python
 * class predict.py:127 `Tracker` plus three file formats.  Every entry point below replaces one
 * reference call site on the path `Tracker.on_track` (predict.py:217-296); the Python mirror in
 * iros20-6d-pose-tracking_amd/ binds them with ctypes.
 *
 * Conventions
 *   - plain C types only; device pointers are caller-owned (e.g. torch tensors' data_ptr());
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - every se3tn_* compute call (preprocess, crop_raw, infer, render, render_frame, fill_depth) is stream-ordered and
 *     asynchronous: no hipMalloc, no hipDeviceSynchronize, no host<->device copy of results => hipGraph-capturable.
 *     All device memory is allocated by the init-time calls: se3tn_create (activations, 176x176 z-buffer),
 *     se3tn_upload_weights / se3tn_bind_weights / se3tn_set_winograd (Winograd planes), se3tn_set_precision (f16x3 split
 *     panels), se3tn_mesh_create / _set_texture, and se3tn_reserve (full-frame z-buffer + fill_depth scratch).  The two
 *     full-frame calls grow their scratch on a first un-reserved call with a larger frame (draining the device first);
 *     inside a stream capture they refuse with SE3TN_E_STATE instead -- call se3tn_reserve at start-up;
 *   - return value: 0 = ok, >0 = a hipError_t, <0 = SE3TN_E_*; se3tn_last_error() has the text;
 *   - one context per (process, device); a context is thread-compatible, not thread-safe, and its compute calls share one set of
 *     activation workspaces: issue them on ONE stream, or order the streams yourself (events / synchronisation) -- two se3tn_infer of
 *     the same context in flight on two streams corrupt each other.  Throughput over several streams = one context per stream on a
 *     shared weight blob (se3tn_bind_weights; the Python mirror's PipelinedEngine).
 */
#ifndef SE3TRACKNET_H
#define SE3TRACKNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SE3TN_OK 0
#define SE3TN_E_ARG (-1)      /* bad argument (NULL, n > max_batch, ...)            */
#define SE3TN_E_STATE (-2)    /* call order (weights not bound, ...)                */
#define SE3TN_E_SHAPE (-3)    /* tensor shape does not match Se3TrackNet.state_dict */
#define SE3TN_E_KEY (-4)      /* unknown / missing state_dict key                   */
#define SE3TN_E_DEVICE (-5)   /* not a gfx950 device                                */

#define SE3TN_RES 176         /* dataset_info.yml `resolution` (predict.py:130)     */

/* input tensor layouts accepted by se3tn_infer */
#define SE3TN_NCHW 0 /* float32 [N,4,176,176], channels R,G,B,D: the reference operator
                        boundary `self.model(dataA, dataB)` predict.py:270-271            */
#define SE3TN_NHWC 1 /* float32 [N,176,176,4]: what se3tn_preprocess writes               */

typedef struct se3tn_ctx se3tn_ctx;

const char* se3tn_version(void);
const char* se3tn_last_error(void);

/* ---- life cycle ------------------------------------------------------------------------- */
/* Replaces `Se3TrackNet(image_size).cuda().eval()` (predict.py:155-158): selects `device`,
 * checks it is gfx950, allocates the activation workspace for up to `max_batch` pairs. */
int se3tn_create(int device, int max_batch, se3tn_ctx** out);
void se3tn_destroy(se3tn_ctx* ctx);
int se3tn_max_batch(const se3tn_ctx* ctx);
/* Start-up reservation of everything the stream-ordered calls would otherwise have to allocate on first use: the
 * z-buffer of se3tn_render_frame and the scratch images of se3tn_fill_depth for camera frames of up to H x W pixels
 * (Tracker.__init__ knows them from dataset_info['camera'], predict.py:147-148), and, if weights are bound, the Winograd
 * planes / f16x3 split panels of the modes currently selected.  Synchronises the device when it has to re-allocate. */
int se3tn_reserve(se3tn_ctx* ctx, int H, int W);

/* ---- weights: `model.load_state_dict(checkpoint['state_dict'])` (predict.py:151-156) ------ */
/* Hand over one float32 entry of Se3TrackNet.state_dict() (host memory, contiguous, the key and
 * shape exactly as torch saved them; int64 `num_batches_tracked` entries are not passed). */
int se3tn_set_tensor(se3tn_ctx* ctx, const char* key, const float* host_data,
                     const int64_t* shape, int ndim);
/* After all 106 tensors: fold BatchNorm (eval mode, eps=1e-5, float64 math) into the
 * convolutions and pack into the device layout (host side).  Fails with SE3TN_E_KEY if a
 * tensor is missing. */
int se3tn_pack_weights(se3tn_ctx* ctx);
size_t se3tn_packed_bytes(const se3tn_ctx* ctx);     /* size of the packed blob          */
const void* se3tn_packed_host(const se3tn_ctx* ctx); /* host pointer, valid after pack   */
/* Single-GPU: library-owned device copy of the packed blob. */
int se3tn_upload_weights(se3tn_ctx* ctx, void* stream);
/* Multi-GPU: use a caller-owned device blob in packed format (rank 0 packs, RCCL broadcasts
 * the bytes, every rank binds its copy).  The blob must outlive the context.  The blob holds the BN-folded float32
 * panels only (54 MB); the Winograd planes and the f16x3 split panels are derived from it on each rank's device. */
int se3tn_bind_weights(se3tn_ctx* ctx, const void* device_blob, size_t bytes);

/* ---- per-dataset constants --------------------------------------------------------------- */
/* mean.npy / std.npy: float64[8] = A(R,G,B,D), B(R,G,B,D) (predict.py:657-658). */
int se3tn_set_normalization(se3tn_ctx* ctx, const double mean[8], const double std[8]);
/* Arithmetic of the convolutions:
 *   SE3TN_PREC_F32   (default) exact float32 MFMA everywhere;
 *   SE3TN_PREC_F16X3 operands split into f16 hi + f16 lo (22 significant bits), products formed as
 *                    hi*hi + hi*lo + lo*hi on the f16 matrix cores with float32 accumulation:
 *                    float32-class error (measured ~1e-6 on the outputs) at 16/3 the MFMA rate.
 *                    Activations of those layers must stay inside the f16 range (|x| < 65000);
 *                    se3tn_overflow reports (and clears) a violation -- rerun in SE3TN_PREC_F32 then. */
#define SE3TN_PREC_F32 0
#define SE3TN_PREC_F16X3 1
/* Init-time call: the first selection of SE3TN_PREC_F16X3 (and a later change of weights while it is selected) derives the
 * split-f16 panels + per-cout power-of-two scales from the bound blob on the device (54 MB, allocated once). */
int se3tn_set_precision(se3tn_ctx* ctx, int mode);
int se3tn_overflow(se3tn_ctx* ctx, int* flag); /* synchronises */
/* Test hooks for that derivation: the device copy (NULL until derived) and its size, and the host statement of the same
 * arithmetic applied to a packed blob in host memory (se3tn_packed_host) -- the two must agree bit for bit. */
size_t se3tn_split_weights_bytes(const se3tn_ctx* ctx);
const void* se3tn_split_weights_device(const se3tn_ctx* ctx);
int se3tn_split_weights_host(const void* packed_blob_host, size_t blob_bytes, void* out_split, size_t out_bytes);
/* Algorithm of the stride-1 256/512-channel convolutions (AB2.*, trans|rot conv2.*) in
 * SE3TN_PREC_F32: batches of n >= min_batch pairs run them as Winograd F(tile x tile, 3x3), tile = 2 | 4 | 6 | 6_4 | AUTO
 * (0 keeps the current tile) -- float32 MFMA GEMMs on (tile+2)^2 transformed planes, 2.25x / 4x / 5.06x fewer
 * multiplies per output; smaller batches and min_batch = 0 use the direct implicit-GEMM kernels.
 * All are float32 arithmetic and differ by rounding only (rms error of one layer relative to its
 * largest activation: direct 5e-8, tile 2 1.8e-7, tile 4 7e-7, tile 6 1.8e-6; end to end the logits move by
 * 0.5 / 0.6 / 1.4e-6 against the direct kernels) -- the same freedom cuDNN takes under
 * torch.backends.cudnn.benchmark = True in the reference (predict.py:78).  Tile 4 runs a whole residual block as one
 * fused launch sequence (in-transform, GEMM, [out | in] mid transform through LDS, GEMM, out-transform | fused avg-pool + FC tail);
 * tile 6 (64 planes: at batch 64 exactly two 128 x 128 GEMM tiles per workgroup slot, 21 % fewer multiplies than tile 4) pays from
 * 14 pairs on.  Where the rounding goes (batched 30-degree closed loop, 64 tracks, every pair against the CPU oracle,
 * scripts/rounding_by_block.py): max |d logit| 4.9e-6 with tile 4 everywhere, 6.0e-6 with tile 6 on the 256-channel block only,
 * 1.3e-5 with tile 6 on the 512-channel heads as well -- the heads' products go straight into the average pool + FC.
 * SE3TN_WINOGRAD_TILE_6_4 = tile 6 for the 256-channel block, tile 4 for the heads.
 * SE3TN_WINOGRAD_TILE_AUTO (the default): tile 4 below SE3TN_WINOGRAD_TILE6_MIN_BATCH pairs; from there tile 6 for the 256-channel
 * block, and for the heads tile 6 while rot_normalizer (se3tn_set_normalizers) <= SE3TN_WINOGRAD_HEADS_TILE6_MAX_ROT, else tile 4: the
 * rotation logits' rounding reaches the composed pose multiplied by rot_normalizer (tolerance 1e-5; with the reference's default
 * 5 degrees the pose moves by < 1.2e-6 either way, with YCBInEOAT's 30 degrees by 3.1e-6 with tile 4 heads, 6.4e-6 with tile 6).
 * SE3TN_PREC_F16X3 has F(4x4) only and uses it whatever is selected. */
#define SE3TN_WINOGRAD_TILE_AUTO 46
#define SE3TN_WINOGRAD_TILE_6_4 64
#define SE3TN_WINOGRAD_TILE6_MIN_BATCH 14
#define SE3TN_WINOGRAD_HEADS_TILE6_MAX_ROT 0.2   /* radians (11.5 degrees) */
#define SE3TN_WINOGRAD_DEFAULT_MIN_BATCH 6
#define SE3TN_WINOGRAD_DEFAULT_TILE SE3TN_WINOGRAD_TILE_AUTO
int se3tn_set_winograd(se3tn_ctx* ctx, int min_batch, int tile);
int se3tn_get_winograd(const se3tn_ctx* ctx, int* min_batch, int* tile);
/* Algorithm of the 64-channel trunk (convA2 / convB2 / convB3, 44 x 44 maps) in SE3TN_PREC_F32: batches of n >= min_batch pairs MAY
 * run its four launches as a FUSED Winograd F(2x2, 3x3) kernel -- input transform, the 16 per-frequency products on the f32 matrix
 * cores and the output transform inside one workgroup per image quadrant, nothing spilled to memory; 2.25x fewer multiplies and
 * exactly 2 / 1 rounds of workgroups on the 256 CUs at batch 64.  A launch takes that kernel when its 4 n groups workgroups fill
 * rounds of the device's CUs to >= min_fill_percent (n >= 34; the grouped A|B launches from n = 18: measured crossovers,
 * scripts/trunk_sweep.sh, profiles/r03f_trunk_sweep.txt); otherwise, and always
 * with min_batch = 0, the direct implicit-GEMM kernels run.  Float32 arithmetic in a different association (rounding as tile 2
 * above: the logits move by < 1.1e-6). */
#define SE3TN_TRUNK_WINOGRAD_DEFAULT_MIN_BATCH 8
#define SE3TN_TRUNK_WINOGRAD_DEFAULT_MIN_FILL 55   /* percent of the last round of workgroups; 0 = every launch with n >= min_batch */
int se3tn_set_trunk_winograd(se3tn_ctx* ctx, int min_batch, int min_fill_percent);
int se3tn_get_trunk_winograd(const se3tn_ctx* ctx, int* min_batch, int* min_fill_percent);
/* Rounding of OffsetDepth's `depth -= pose[2,3]*1000` (data_augmentation.py:134-144: a float64 scalar subtracted from a float32 array):
 *   SE3TN_OFFSET_RULE_NUMPY1 (default) what every NumPy the reference can run on does (it pins Python 3.6 => NumPy <= 1.19,
 *                            docker/dockerfile:28; `np.float`, Utils.py:307, stops importing at 1.24): value-based casting, the
 *                            scalar becomes float32 and the subtraction is ONE float32 operation;
 *   SE3TN_OFFSET_RULE_NUMPY2 NEP 50 (NumPy >= 2, were the reference ported to it): float64 subtraction, rounded to float32 once.
 * The two differ by <= 1 ulp(f32) on the offset depth and only when z * 1000 is not a float32 (tests/golden/preprocess_numpy1.npz
 * was produced by the reference's own classes under NumPy 1.26.4, preprocess.npz under NumPy 2.2; the kernel matches each bit for bit). */
#define SE3TN_OFFSET_RULE_NUMPY1 0
#define SE3TN_OFFSET_RULE_NUMPY2 1
int se3tn_set_offset_rule(se3tn_ctx* ctx, int rule);
int se3tn_get_offset_rule(const se3tn_ctx* ctx);
/* trans_normalizer / rot_normalizer of Tracker.__init__ (predict.py:128). */
int se3tn_set_normalizers(se3tn_ctx* ctx, double trans_normalizer, double rot_normalizer);

/* ---- pre-processing: crop_bbox + OffsetDepth + NormalizeChannels + ToTensor --------------- */
/* One 176x176 crop (Utils.py:320-359, data_augmentation.py:124-189).  For the rendered image A
 * pass the 176x176 render itself with window (0,0,176,176). */
typedef struct se3tn_crop {
  const uint8_t* rgb;    /* device, [H,W,3] uint8 RGB                                     */
  const uint16_t* depth; /* device, [H,W] uint16 millimetres                              */
  int32_t H, W;          /* frame size                                                    */
  int32_t left, top, right, bottom; /* crop window = min/max of compute_bbox's (v,u) corners;
                                       may leave the frame (zero padding, Utils.py:330-342) */
  double z_offset_mm;    /* poseA[2,3]*1000 (data_augmentation.py:137-140; both A and B use
                            poseA, :130-131); sign handled as the reference does            */
  int32_t stats;         /* 0: mean[:4]/std[:4] (image A)   1: mean[4:]/std[4:] (image B)  */
  int32_t _pad;
} se3tn_crop;
/* `crops` is a HOST array of n descriptors (copied into kernel arguments, no sync);
 * out_nhwc: device float32 [n,176,176,4] -- or one of the context's own input buffers
 * (se3tn_input_buffer), which are stored zero-bordered and consumed in place by se3tn_infer. */
int se3tn_preprocess(se3tn_ctx* ctx, const se3tn_crop* crops, int n, float* out_nhwc,
                     void* stream);
/* crop_bbox (Utils.py:320-359) ALONE: the zero-padded crop + cv2.resize(..., INTER_NEAREST) of one window,
 * as raw images -- what Tracker.render_window returns for a full-frame renderer (predict.py:209-213) and
 * what a caller of Tracker.dataset.processData feeds it.  crop->z_offset_mm / stats are ignored.
 * Outputs (device): rgb uint8 [176,176,3], depth uint16 [176,176]. */
int se3tn_crop_raw(se3tn_ctx* ctx, const se3tn_crop* crop, uint8_t* rgb_out, uint16_t* depth_out, void* stream);
/* the context's own input buffers, which = 0 (A) / 1 (B).  Opaque layout ([max_batch,182,182,4],
 * 3-pixel zero border): only pass them to se3tn_preprocess (as out) and se3tn_infer (as A / B). */
float* se3tn_input_buffer(se3tn_ctx* ctx, int which);

/* ---- the network + pose update ------------------------------------------------------------ */
/* Replaces `self.model(dataA,dataB)` (predict.py:270-271 -> se3_tracknet.py:81-112) and, when
 * poseA/poseB are given, `TrackDataset.processPredict` (datasets.py:159-175) for each pair.
 *   A, B     device float32, layout per `layout`, n pairs (A or B may be the context's own
 *            input buffers)
 *   trans,rot device float32 [n,3] (tanh outputs, as prediction['trans'/'rot']); may be NULL
 *   poseA    device float64 [n,16] row-major 4x4 object-in-camera (metres); may be NULL
 *   poseB    device float64 [n,16]: t_B = trans*tn + t_A,  R_B = Rodrigues(rot*rn) . R_A     */
int se3tn_infer(se3tn_ctx* ctx, const float* A, const float* B, int n, int layout, float* trans,
                float* rot, const double* poseA, double* poseB, void* stream);
/* hipGraph replay of se3tn_infer (off by default).  When on, the second se3tn_infer with an argument
 * set (pointers, n, layout, modes) captures its launches on `stream` (which must not be the null stream)
 * and later calls replay the graph: for the launch-bound batch-1 tracking step.  The input / output /
 * pose buffers must therefore be persistent device buffers whose CONTENTS change between calls. */
int se3tn_enable_graphs(se3tn_ctx* ctx, int on);
/* output['feature'] (se3_tracknet.py:96): device float32 [n,256,22,22] NCHW, valid after infer */
int se3tn_get_feature(se3tn_ctx* ctx, int n, float* feature_nchw, void* stream);
/* pre-tanh FC outputs of the last se3tn_infer: device float32 [max_batch,6] (trans, rot) */
const float* se3tn_logits(se3tn_ctx* ctx);

/* ---- rendered image A: HIP rasteriser replacing the reference's OpenGL renderer ---------------- */
/* vispy_renderer.py:47-178 (VispyRenderer) as called by Tracker.render_window (predict.py:193-215).
 * Mesh = what the reference reads from the .ply: float32 vertices [V,3] (metres), unit vertex normals
 * [V,3], vertex colours [V,3] already divided by 255, int32 triangles [F,3] (host pointers, copied). */
typedef struct se3tn_mesh se3tn_mesh;
int se3tn_mesh_create(se3tn_ctx* ctx, const float* verts, const float* normals, const float* colors01, int V,
                      const int32_t* faces, int F, se3tn_mesh** out);
void se3tn_mesh_destroy(se3tn_mesh* mesh);
/* ob_in_cam: row-major 4x4 object pose in the OpenCV camera; K row-major 3x3; window = {left, top,
 * right, bottom} as render_window derives it from compute_bbox(..., scale=(1000,-1000,1000))
 * (predict.py:201-206: u range, and the range of the FLIPPED row coordinate cy - fy y/z).
 * Outputs (device): rgb uint8 [176,176,3], depth uint16 [176,176] millimetres, 0 = background. */
int se3tn_render(se3tn_ctx* ctx, se3tn_mesh* mesh, const double ob_in_cam[16], const double K[9],
                 const int32_t window[4], uint8_t* rgb, uint16_t* depth, void* stream);
/* Batches of 1-5 pairs -- the regime Tracker.on_track runs in -- take kernels of their own: the four 64 -> 64 trunk convs without a
 * K split or a reduction launch (conv64_small.hip) and the 128 .. 512-channel convs as one round of 128-pixel x 32-cout x
 * channel-slice workgroups with a layer-fixed slice count (conv_slices_small.hip), the stem + max-pool in one launch of
 * 16-pool-pixel tiles (stem_pool_small.hip), and a tail that adds the last conv's partial-sum slices itself (tail_parts_kernel).
 * float32, same tolerances, other summation orders than the kernels larger batches take.  Every kernel of the family works image by
 * image with a layer-fixed summation order: for every n <= 5 a pair has the same bits alone or in any batch (round 6; from 6 pairs the
 * algorithms change -- Winograd thresholds at 6 / 14 pairs -- and the last bits of a pair's result depend on the batch it travels in,
 * as in the reference under cuDNN's per-shape algorithm choice, predict.py:78).  on = 0 keeps every batch size on the general
 * kernels (default 1; SE3TN_SMALL_KERNELS=0 in the environment of se3tn_create does the same). */
int se3tn_set_small_kernels(se3tn_ctx* ctx, int on);
int se3tn_get_small_kernels(const se3tn_ctx* ctx);
/* Rasterisation rule.  OpenGL leaves sub-pixel precision, interpolation arithmetic and float -> unorm rounding to the
 * implementation; image A is defined here as what the reference's VispyRenderer produces on the conformant software GL the
 * goldens were rendered on (tests/golden/gl_swiftshader*.npz), reproduced byte for byte: window coordinates snapped to
 * 1 / 2^sub_bits pixel (4 = that implementation = the GL minimum GL_SUBPIXEL_BITS, default; 8 = what desktop GPUs report),
 * integer top-left coverage, that implementation's plane-equation arithmetic and 16-bit unorm conversion.  The depth read-back
 * (vispy_renderer.py:163-169) mixes a float32 array with float64 scalars: it follows se3tn_set_offset_rule's NumPy generation.
 * sub_bits = 8 has NO golden: no desktop GL is reachable from the build / test boxes, so that setting is the same arithmetic
 * with a finer snap and is pinned by nothing but the rule tests (oracle/pin_gl.py pins it on first contact with such a GL). */
int se3tn_set_raster_rule(se3tn_ctx* ctx, int sub_bits);
int se3tn_get_raster_rule(const se3tn_ctx* ctx);

/* The reference's second renderer, offscreen_renderer.py:48-83 (pyrender; dataset_info['renderer'] == 'pyrenderer',
 * predict.py:161-164, textured .obj models): a FULL W x H camera frame, ambient light only, depth = camera z.
 * se3tn_mesh_set_texture attaches the material: per-vertex texture coordinates uv [V,2] (OBJ convention, v up),
 * an RGB uint8 image [th,tw,3] (row 0 = top; NULL: the vertex colours are the base colour) and the mtl's Kd
 * (NULL: 1,1,1); host pointers, copied; the mip pyramid is built here.  se3tn_render_frame writes device
 * rgb uint8 [H,W,3] and depth uint16 [H,W] millimetres ((depth * 1000).astype(uint16), predict.py:211); feed
 * them to se3tn_preprocess / se3tn_crop_raw with the plain compute_bbox window exactly like a camera frame
 * (predict.py:209-213).  Z-buffer: se3tn_reserve(ctx, H, W) at start-up (else grown on the first call, see Conventions). */
int se3tn_mesh_set_texture(se3tn_mesh* mesh, const float* uv, const uint8_t* rgb, int tw, int th, const float kd[3]);
int se3tn_render_frame(se3tn_ctx* ctx, se3tn_mesh* mesh, const double ob_in_cam[16], const double K[9], int W, int H,
                       uint8_t* rgb, uint16_t* depth, void* stream);

/* ---- one frame of Tracker.on_track in ONE call -------------------------------------------------------- */
/* predict.py:217-296 for samples == 1, with the model mesh rendered by this library: compute_bbox (host float64) -> image A
 * (se3tn_render at the y-flipped window, :193-208) -> crop + normalise of image A and of the camera frame (ONE launch) -> network ->
 * pose update -> read-back.  rgb / depth are HOST pointers to the camera frame (uint8 [H,W,3] RGB, uint16 [H,W] millimetres):
 * only the rows / columns the crop window covers are staged (pinned) and uploaded, in one copy together with the pose.
 * rgbA_dev / depthA_dev: optional device buffers (uint8 [176,176,3], uint16 [176,176]) that receive image A (NULL: internal).
 * Outputs are host memory: pose_out = the 4x4 float64 estimate (row-major); trans_out / rot_out [3] (may be NULL) the network's
 * tanh outputs; bbox_vu [4][2] (may be NULL) compute_bbox's corners.  SYNCHRONOUS on `stream` (as predict.py:275-276 is); the first
 * call (or a larger frame) allocates the staging buffers.  Same arithmetic as se3tn_render + se3tn_preprocess x2 + se3tn_infer. */
int se3tn_on_track(se3tn_ctx* ctx, se3tn_mesh* mesh, const double prev_pose[16], const double K[9], double object_width_mm,
                   const uint8_t* rgb, const uint16_t* depth, int H, int W, uint8_t* rgbA_dev, uint16_t* depthA_dev,
                   double pose_out[16], float trans_out[3], float rot_out[3], int32_t bbox_vu[8], void* stream);

/* ---- n tracks in ONE call ----------------------------------------------------------------------------- */
/* The loop body of predict.py:217-296 for n INDEPENDENT (pose, camera frame) pairs of the same object -- several sequences /
 * cameras / hypotheses advanced together (frames of one track are serial: this is where a batch comes from in deployment;
 * Tracker.on_track_batch).  Per pair exactly se3tn_on_track's arithmetic: compute_bbox (host float64), image A of all n poses in
 * FOUR rasteriser launches (grid.y = pose), the crop windows of the n frames staged through pinned memory in one copy, both crops of
 * every pair, the network on n pairs, the pose update.  prev_poses [n,16] row-major; rgb / depth: n HOST pointers to uint8 [H,W,3] /
 * uint16 [H,W] frames of one size (the same pointer may repeat).  rgbA_dev / depthA_dev: optional device buffers
 * [n,176,176,3] / [n,176,176] that receive the images A (NULL: internal).  Outputs (host): pose_out [n,16], trans_out / rot_out
 * [n,3] (may be NULL), bbox_vu [n,4,2] (may be NULL).  n <= max_batch of se3tn_create.  SYNCHRONOUS on `stream`; the first call
 * (a larger n, mesh or frame) allocates. */
int se3tn_on_track_batch(se3tn_ctx* ctx, se3tn_mesh* mesh, int n, const double* prev_poses, const double K[9], double object_width_mm,
                         const uint8_t* const* rgb, const uint16_t* const* depth, int H, int W, uint8_t* rgbA_dev,
                         uint16_t* depthA_dev, double* pose_out, float* trans_out, float* rot_out, int32_t* bbox_vu, void* stream);

/* ---- live-camera front end: depth hole filling ---------------------------------------------------- */
/* Utils.py:455-514 `fill_depth` as predict_ros.py:38-41 applies it to every depth frame before on_track:
 *     depth = fill_depth(depth_mm / 1e3, max_depth, extrapolate, blur_type);  out_mm = (depth * 1000).astype(uint16)
 * depth_mm: device uint16 [H,W] millimetres; out_mm: device uint16 [H,W] and / or out_m: device float32 [H,W] metres
 * (either may be NULL).  blur: the reference's blur_type ('bilateral' is its default; anything else but 'gaussian'
 * skips the blur).  Scratch images: se3tn_reserve(ctx, H, W) at start-up (else grown on the first call, see Conventions).
 * A context's scratch is shared by its calls: use one stream per context for se3tn_fill_depth / se3tn_render*. */
#define SE3TN_BLUR_NONE 0
#define SE3TN_BLUR_BILATERAL 1
#define SE3TN_BLUR_GAUSSIAN 2
int se3tn_fill_depth(se3tn_ctx* ctx, const uint16_t* depth_mm, int H, int W, double max_depth_m, int extrapolate,
                     int blur, uint16_t* out_mm, float* out_m, void* stream);

/* ---- host-side pieces of the path (pure CPU, float64, as the reference computes them) ----- */
/* Utils.py:302-316 compute_bbox with scale (1000,1000,1000): pose row-major 4x4 (metres), K
 * row-major 3x3, width in mm; out_vu[8] = 4 x (v,u) int32, np.round (half-to-even). */
int se3tn_compute_bbox(const double pose[16], const double K[9], double object_width_mm,
                       int32_t out_vu[8]);
/* datasets.py:159-175 processPredict on the host. */
int se3tn_pose_update_host(const double poseA[16], const float trans[3], const float rot[3],
                           double trans_normalizer, double rot_normalizer, double poseB[16]);

/* ---- introspection for tests / profiling --------------------------------------------------- */
/* Device pointer + geometry of an internal NHWC activation buffer after se3tn_infer.
 * names: "inA" "inB" [n,182,182,4] (3-pixel zero border), "stem" [n,88,88,128]; the conv activations carry a one-pixel
 * zero border: "pool" "t64" "q64" [n,46,46,128] (channels 0-63 branch A, 64-127 branch B),
 * "ab" "ab_t" [n,24,24,256], "head" "head_t" [n,13,13,1024] (0-511 trans, 512-1023 rot).
 * dims = {H, W, C} as stored (borders included). */
int se3tn_debug_buffer(se3tn_ctx* ctx, const char* name, const float** ptr, int32_t dims[3]);
/* The fused Winograd blocks (batches of n >= the se3tn_set_winograd threshold) keep the activation between a
 * residual block's two convolutions in LDS and reduce the heads' last activation in registers: "ab_t", "head_t"
 * and "head" are then NOT written.  on != 0 makes those kernels store them as well (tests, feature inspection);
 * results are bit-identical either way.  At 1-5 pairs the batch-1 kernel family never writes the un-pooled "stem" map nor the
 * final "head" map (the tail adds the last conv's partial sums itself): on != 0 selects, for those two stages, the general kernels
 * that do write them -- same tolerances, another summation order (not the same bits as the default at 1-5 pairs). */
int se3tn_keep_intermediates(se3tn_ctx* ctx, int on);
/* stream-ordered device-to-device copy (lets a ctypes host wrap the raw pointers above into its
 * own tensors without a second HIP binding) */
int se3tn_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
/* Timing hooks used by bench.py.  se3tn_profile_enable(ctx, slots) (slots <= 64, 0 = off) makes
 * every following se3tn_infer record hipEvents around each of its launches on `stream`, into event
 * set (call index % slots) -- no synchronisation is added to the timed loop.  Reading a slot
 * synchronises on its last event.  conv_ms = time inside the dominant kernel family (the 3x3
 * f32-MFMA convolutions, `conv_launches` launches), total_ms = all launches of that infer. */
int se3tn_profile_enable(se3tn_ctx* ctx, int slots);
int se3tn_profile_read(se3tn_ctx* ctx, int slot, float* conv_ms, int* conv_launches, float* total_ms);
/* Per-launch breakdown of one slot: fills up to `cap` names / milliseconds, returns the number of
 * launches (0 on error, see se3tn_last_error). */
int se3tn_profile_launches(se3tn_ctx* ctx, int slot, int cap, const char** names, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* SE3TRACKNET_H */
