// C ABI of the hot path (include/se3tracknet.h).  Host orchestration only: every kernel lives in
// the .hip files.  One context per (process, device).
#include <cmath>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "se3tn_internal.h"
#include "weights.h"

using namespace se3tn;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
static int hipfail(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return (int)e;
}
#define HIPCHK(x)                                  \
  do {                                             \
    hipError_t _e = (x);                           \
    if (_e != hipSuccess) return hipfail(_e, #x);  \
  } while (0)

enum { MAX_LAUNCHES = 24, MAX_SLOTS = 64 };

// one captured se3tn_infer: every argument that is baked into the kernel launches
struct GraphKey {
  const void *A, *B, *trans, *rot, *poseA, *poseB, *blob;
  int n, layout, prec, wino_min_batch, wino_tile, keep, wino64, wino64_fill;
  int small_kernels, splitk_fused, wino_fuse, gemmp, tail_parts;   // every switch that selects a kernel family (a graph captured under another
                                                       // setting must not be replayed)
  double tn, rn;
  bool operator==(const GraphKey& o) const { return std::memcmp(this, &o, sizeof(GraphKey)) == 0; }
};
struct GraphEntry {
  GraphKey key;
  hipGraphExec_t exec = nullptr;  // null: seen once (run eagerly, which also sets the kernels' LDS attributes)
  bool fast = false;
  const float* head_final = nullptr;
};

struct se3tn_ctx {
  int device = -1, max_batch = 0;
  TensorMap tensors;
  std::vector<float> packed;
  float* blob_owned = nullptr;
  const float* blob = nullptr;
  BlobLayout L;
  // activations (NHWC float32)
  float *inA = nullptr, *inB = nullptr;         // [mb,182,182,4] (3-pixel zero border)
  float* stem = nullptr;                        // [mb,88,88,128]
  // zero-bordered conv activations (se3tn_internal.h: ConvArgs)
  float *pool = nullptr, *t64 = nullptr, *q64 = nullptr;  // [mb,46,46,128]
  float *ab = nullptr, *ab_t = nullptr;         // [mb,24,24,256]
  float *head = nullptr, *head_t = nullptr;     // [mb,13,13,1024]
  float* head_f = nullptr;                      // f16x3 mode: float32 output of the last head conv (the
                                                // in-place residual update cannot change format)
  const float* head_final = nullptr;            // what the tail of the last infer read
  float* logits = nullptr;                      // [mb,6]
  float* fcpart = nullptr;                      // [mb,2,<=32,3] partial FC dot products (fused Winograd tail: 8 slices per head; tail_kernel 8; tail_parts_kernel 32)
  bool keep_intermediates = false;              // se3tn_keep_intermediates: fused blocks also store ab_t / head_t / head
  int auto_tile_override[2] = {0, 0};           // SE3TN_WINOGRAD_AUTO_TILE_AB2 / _HEADS = 4 | 6: what AUTO picks per block (rounding studies)
  int trunk_kernel = 1;                         // SE3TN_TRUNK_KERNEL = 2: the register-V trunk experiment (only in -DSE3TN_TRUNK_REGV=1 builds)
  int gemmp = -1;                               // SE3TN_WINO_GEMMP = 0 | 1: never | always the persistent 128 x 256 Winograd GEMM (default: by tile count)
  bool wino_fuse = true;                        // whole residual blocks as one fused launch sequence (SE3TN_WINOGRAD_FUSE=0: conv by conv)
  float* part = nullptr;                        // split-K partial sums (small-batch latency path)
  int* splitk_sem = nullptr;                    // [2 x SE3TN_SPLITK_MAX_TILES] arrival / seen counters of the fused split-K reduction (zero between launches)
  int tail_parts_ch = 16;                       // SE3TN_TAIL_PARTS=2: 32 workgroups x 32 channels per pair instead of 64 x 16 (developer A/B)
  bool tail_parts = true;                       // SE3TN_TAIL_PARTS=0 (developer switch): batch 1-5 keeps conv_reduce + tail_kernel after the last head conv
  bool small_kernels = true;                    // SE3TN_SMALL_KERNELS=0 (developer switch): batch 1-5 through the split-K kernels only
  bool splitk_fused = false;                    // SE3TN_SPLITK_FUSED=1 (developer switch): the reduction inside the split-K launch -- bitwise the same results,
                                                // but SLOWER on this chip (363 vs 268 us per batch-1 forward: EXPERIMENTS item 41), so off
  size_t part_bytes = 0;
  // Winograd F(m x m,3x3) path (wino_mfma.hip) of convAB2.* and trans|rot conv2.* at n >= wino_min_batch
  int wino_min_batch = SE3TN_WINOGRAD_DEFAULT_MIN_BATCH;  // 0 = never
  int wino_tile = SE3TN_WINOGRAD_DEFAULT_TILE;            // m = 2 | 4 | 6 | SE3TN_WINOGRAD_TILE_AUTO
  int wino_tile_derived = 0;                              // the m wino_u was derived for
  float* wino_u[4] = {nullptr, nullptr, nullptr, nullptr};  // U = G g G^T of LAB2_1, LAB2_2, LH2_1, LH2_2 (tile 2 | 4: wino_tile_derived)
  float* wino_u6[4] = {nullptr, nullptr, nullptr, nullptr}; // the F(6x6) planes of the same four convs (tile 6 | AUTO only)
  const float* wino6_blob = nullptr;                        // the blob wino_u6 was derived from
  float *wino_v = nullptr, *wino_m = nullptr;   // [g][16][T][C] input tiles / per-frequency products
  const float* wino_blob = nullptr;             // the blob wino_u was derived from
  // fused F(2x2) path of the 64-channel trunk (wino64_fused.hip) at n >= wino64_min_batch (0 = never)
  int wino64_min_batch = SE3TN_TRUNK_WINOGRAD_DEFAULT_MIN_BATCH;
  int wino64_min_fill = SE3TN_TRUNK_WINOGRAD_DEFAULT_MIN_FILL;
  int num_cus = 256;   // compute units of the device (the fused trunk kernel runs one workgroup per CU: it pays in full rounds only)
  float* wino64_u[4] = {nullptr, nullptr, nullptr, nullptr};   // F(2x2) planes of L64_1..L64_4 ([group][chunk 2][16][64][32])
  const float* wino64_blob = nullptr;
  SplitLayout SL;                               // f16x3 mode: split panels in split_w (derived from the blob, not part of it)
  float* split_w = nullptr;
  const float* split_blob = nullptr;            // the blob split_w was derived from
  float* wino_us[4] = {nullptr, nullptr, nullptr, nullptr};   // f16x3 Winograd blocks: U as split rows ...
  float* wino_usc[4] = {nullptr, nullptr, nullptr, nullptr};  // ... and their per-(group, frequency, cout) 2^-k
  const float* wino_us_blob = nullptr;          // the blob / tile the split planes were derived for
  int wino_us_tile = 0;
  int prec = SE3TN_PREC_F32;                    // se3tn_set_precision
  int offset_rule = SE3TN_OFFSET_RULE_NUMPY1;   // se3tn_set_offset_rule
  int raster_sub_bits = 4;                      // se3tn_set_raster_rule: sub-pixel bits of the rasteriser's window coordinates
  // se3tn_on_track: pinned host staging [pose 128 B | frame window rgb | depth], its device mirror, device outputs and their pinned copy
  uint8_t* trk_host = nullptr; uint8_t* trk_dev = nullptr; size_t trk_bytes = 0;
  uint8_t* trk_out_dev = nullptr; uint8_t* trk_out_host = nullptr;   // ONE mapped pinned block: device address | host address
  // se3tn_on_track_batch: instance table (pinned host + device), z-buffers, image A stack, staging, outputs -- grown at first use / larger n
  int tb_cap = 0;
  size_t tb_stage_bytes = 0;
  RasterInstance *tb_inst_host = nullptr, *tb_inst_dev = nullptr;
  unsigned long long* tb_zbuf = nullptr;
  uint8_t *tb_rgbA = nullptr, *tb_stage_host = nullptr, *tb_stage_dev = nullptr, *tb_out_host = nullptr, *tb_out_dev = nullptr;
  uint16_t* tb_depthA = nullptr;
  bool rearm_counters = false;                  // a failed launch sequence: clear tail_arrive / splitk_sem before the next one
  int* tail_arrive = nullptr;                   // [max_batch] arrival counters of tail_kernel's 16 workgroups per pair (zero between launches)
  int* tail_flag = nullptr; int tail_seq = 0;   // set around se3tn_on_track's infer: the tail kernel stores tail_seq to this (mapped) word
  hipStream_t trk_copy_stream = nullptr; hipEvent_t trk_copy_event = nullptr;
  uint8_t* trk_rgbA = nullptr; uint16_t* trk_depthA = nullptr;
  int in_split[2] = {0, 0};                     // pixel format currently held by inA / inB
  bool last_fast = false;                       // the last infer ran the f16x3 kernels (ab is split rows)
  int* overflow = nullptr;                      // device flag: a split-row store left the f16 range
  unsigned long long* zbuf = nullptr;           // rasteriser z-buffer keys [zbuf_px] (176*176, grown by se3tn_render_frame)
  size_t zbuf_px = 0;
  float* fd_buf = nullptr;                      // se3tn_fill_depth scratch: 3 images + minmax[2] + lut[4098]
  size_t fd_pixels = 0;
  bool use_graphs = false;                      // se3tn_enable_graphs
  std::vector<GraphEntry> graphs;
  double mean[8], stdv[8];
  bool have_norm = false;
  double tn = 0.03, rn = 5.0 * 3.14159265358979323846 / 180.0;
  // profiling
  bool prof = false;
  int slots = 0;          // number of event sets (one per profiled se3tn_infer, used round-robin)
  long long infer_count = 0;
  hipEvent_t* ev = nullptr;  // points at evs[slot]
  hipEvent_t evs[MAX_SLOTS][MAX_LAUNCHES + 1];
  int n_launch = 0;
  int slot_launches[MAX_SLOTS];
  const char* names[MAX_LAUNCHES];
  bool is_conv[MAX_LAUNCHES];
};

struct se3tn_mesh {
  float *verts = nullptr, *normals = nullptr, *colors = nullptr;
  int* faces = nullptr;
  float4* vpost = nullptr;  // [V] clip positions, [V] snapped window coordinates (raster_vertex_kernel)
  int4* vsnap = nullptr;
  int* big = nullptr;     // [1 + F] queue of large triangles (raster_queue_kernel)
  int* clipq = nullptr;   // [1 + F] queue of triangles that cross the frustum (raster_queue_kernel)
  int V = 0, F = 0;
  // batched rasteriser scratch (se3tn_on_track_batch): batch_cap instances of vpost / vsnap / big / clipq
  int batch_cap = 0;
  float4* b_vpost = nullptr;
  int4* b_vsnap = nullptr;
  int *b_big = nullptr, *b_clipq = nullptr;
  // pyrender-style material (se3tn_mesh_set_texture): uv per vertex, RGB mip pyramid, Kd
  float* uv = nullptr;
  uint8_t* tex = nullptr;
  int tw = 0, th = 0, tlevels = 0;
  unsigned tex_off[16] = {};
  float kd[3] = {1.f, 1.f, 1.f};
};

// Init-time entry points that allocate or launch (plane derivation, workspace growth) run on the CONTEXT's device whatever device the
// calling thread has current (two contexts on two GPUs in one process: ADVICE r3), and leave the caller's current device as it was.
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    if (device < 0) return;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) err = hipSetDevice(device);
    else prev = -1;   // nothing to restore
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// ---- Winograd workspaces ----------------------------------------------------------------------------
static const ConvId kWinoConvs[4] = {LAB2_1, LAB2_2, LH2_1, LH2_2};
static int wino_slot(ConvId id) {
  for (int i = 0; i < 4; ++i)
    if (kWinoConvs[i] == id) return i;
  return -1;
}
// V / M: the largest layer shape over m = 2 | 4 | 6, floats per image: (m=2) 16 x 36 tiles x 1024 ch | 16 x 121 x 256
// (m = 6: 64 frequencies x 4 | 16 tiles: smaller)
static size_t wino_ws_floats(int max_batch) {
  const size_t head = (size_t)16 * 36 * 1024, ab = (size_t)16 * 121 * 256;
  return (size_t)max_batch * (head > ab ? head : ab);
}
// The tile the 256 / 512-channel blocks of a batch of n pairs run with.  F(6x6) exists in float32 only (SE3TN_PREC_F16X3 keeps F(4x4)
// and its fused head block on split operands) and pays from 14 pairs on (scripts/tile_sweep.sh: 0.511 vs 0.462 ms at n = 6, 0.701 vs
// 0.718 at n = 16, 1.76 vs 1.86 at n = 64): SE3TN_WINOGRAD_TILE_AUTO switches there.  With tile 6 | AUTO selected both U sets are
// resident: the F(4x4) planes in wino_u, the F(6x6) planes in wino_u6.
// which: 0 = the 256-channel block (convAB2), 1 = the 512-channel heads
static int tile_for(const se3tn_ctx* c, int n, int which) {
  const int t = c->wino_tile;
  if (t == 2 || t == 4) return t;
  if (c->prec == SE3TN_PREC_F16X3) return 4;
  if (t == 6) return 6;
  if (t == SE3TN_WINOGRAD_TILE_6_4) return which == 0 ? 6 : 4;
  if (c->auto_tile_override[which]) return c->auto_tile_override[which];   // developer switch: SE3TN_WINOGRAD_AUTO_TILE_AB2 / _HEADS
  if (n < SE3TN_WINOGRAD_TILE6_MIN_BATCH) return 4;
  // AUTO: F(6x6) where its rounding is cheap.  The heads' products feed the average pool + FC directly: F(6x6) there doubles the
  // logits' rounding error (1.3e-5 vs 6.0e-6 in the batched 30-degree closed loop), and the rotation logits reach the composed pose
  // multiplied by rot_normalizer -- with a large normaliser (YCBInEOAT's 30 degrees, predict.py:586) the heads stay on F(4x4)
  return (which == 0 || c->rn <= SE3TN_WINOGRAD_HEADS_TILE6_MAX_ROT) ? 6 : 4;
}
static const ConvId kWino64Convs[4] = {L64_1, L64_2, L64_3, L64_4};
static int wino64_prepare(se3tn_ctx* c, hipStream_t st) {
  if (c->wino64_min_batch <= 0 || c->max_batch < c->wino64_min_batch || !c->blob || c->wino64_blob == c->blob) return SE3TN_OK;
  for (int i = 0; i < 4; ++i) {
    const Conv3& s = conv_specs()[kWino64Convs[i]];
    const size_t per_g = (size_t)16 * s.cin * s.cout;
    if (!c->wino64_u[i]) HIPCHK(hipMalloc((void**)&c->wino64_u[i], s.groups * per_g * sizeof(float)));
    for (int g = 0; g < s.groups; ++g)
      HIPCHK(launch_wino_weights(c->blob + c->L.conv_w[kWino64Convs[i]] + (size_t)g * conv3_words(s.cin, s.cout),
                                 c->wino64_u[i] + g * per_g, s.cin, s.cout, 2, st));
  }
  HIPCHK(hipStreamSynchronize(st));  // init-time
  c->wino64_blob = c->blob;
  return SE3TN_OK;
}

static int wino_prepare(se3tn_ctx* c, hipStream_t st) {
  DeviceGuard dg(c->device);
  HIPCHK(dg.err);
  if (int rc = wino64_prepare(c, st)) return rc;
  if (c->wino_min_batch <= 0 || c->max_batch < c->wino_min_batch) return SE3TN_OK;
  if (!c->wino_v) {
    HIPCHK(hipMalloc((void**)&c->wino_v, wino_ws_floats(c->max_batch) * sizeof(float)));
    HIPCHK(hipMalloc((void**)&c->wino_m, wino_ws_floats(c->max_batch) * sizeof(float)));
    for (int i = 0; i < 4; ++i) {
      const Conv3& s = conv_specs()[kWinoConvs[i]];
      HIPCHK(hipMalloc((void**)&c->wino_u[i], (size_t)s.groups * 36 * s.cin * s.cout * sizeof(float)));
    }
  }
  const int tile = c->wino_tile == 2 ? 2 : 4;   // the planes in wino_u (tile 6 | AUTO keep the F(4x4) set there as well)
  if (c->blob && (c->wino_tile == 6 || c->wino_tile == SE3TN_WINOGRAD_TILE_AUTO || c->wino_tile == SE3TN_WINOGRAD_TILE_6_4) &&
      c->wino6_blob != c->blob) {
    for (int i = 0; i < 4; ++i) {
      const Conv3& s = conv_specs()[kWinoConvs[i]];
      if (!c->wino_u6[i]) HIPCHK(hipMalloc((void**)&c->wino_u6[i], (size_t)s.groups * 64 * s.cin * s.cout * sizeof(float)));
      for (int g = 0; g < s.groups; ++g)
        HIPCHK(launch_wino_weights(c->blob + c->L.conv_w[kWinoConvs[i]] + (size_t)g * conv3_words(s.cin, s.cout),
                                   c->wino_u6[i] + (size_t)g * 64 * s.cin * s.cout, s.cin, s.cout, 6, st));
    }
    HIPCHK(hipStreamSynchronize(st));  // init-time
    c->wino6_blob = c->blob;
  }
  if (c->blob && (c->wino_blob != c->blob || c->wino_tile_derived != tile)) {
    const bool new_blob = c->wino_blob != c->blob;
    const int nf = (tile + 2) * (tile + 2);
    for (int i = 0; i < 4; ++i) {
      const Conv3& s = conv_specs()[kWinoConvs[i]];
      for (int g = 0; g < s.groups; ++g)
        HIPCHK(launch_wino_weights(c->blob + c->L.conv_w[kWinoConvs[i]] + (size_t)g * conv3_words(s.cin, s.cout),
                                   c->wino_u[i] + (size_t)g * nf * s.cin * s.cout, s.cin, s.cout, tile, st));
    }
    HIPCHK(hipStreamSynchronize(st));  // init-time
    c->wino_blob = c->blob;
    c->wino_tile_derived = tile;
    if (new_blob) c->wino_us_blob = nullptr;   // the f16x3 split planes (always of the F(4x4) U) follow the weights
  }
  return SE3TN_OK;
}

// f16x3 mode: the split-f16 panels + per-cout scales, derived on the device from the bound float32 blob (init time: called
// from se3tn_set_precision / se3tn_upload_weights / se3tn_bind_weights, never from a stream-ordered compute call)
static int split_prepare(se3tn_ctx* c, hipStream_t st) {
  if (c->device < 0 || c->prec != SE3TN_PREC_F16X3 || !c->blob) return SE3TN_OK;
  DeviceGuard dg(c->device);
  HIPCHK(dg.err);
  if (c->split_blob != c->blob) {
    if (!c->split_w) HIPCHK(hipMalloc((void**)&c->split_w, c->SL.total * sizeof(float)));
    HIPCHK(launch_split_weights(c->blob, c->L, c->split_w, c->SL, st));
    HIPCHK(hipStreamSynchronize(st));  // init-time
    c->split_blob = c->blob;
  }
  // the fused F(4x4) blocks in f16x3 mode: split rows of U = G g G^T (derived by wino_prepare, which runs first)
  if (c->wino_u[0] && c->wino_blob == c->blob && c->wino_tile_derived == 4 &&
      (c->wino_us_blob != c->blob || c->wino_us_tile != 4)) {
    for (int i = 0; i < 4; ++i) {
      const Conv3& s = conv_specs()[kWinoConvs[i]];
      const size_t per_g = (size_t)36 * s.cin * s.cout;
      if (!c->wino_us[i]) {
        HIPCHK(hipMalloc((void**)&c->wino_us[i], s.groups * per_g * sizeof(float)));
        HIPCHK(hipMalloc((void**)&c->wino_usc[i], (size_t)s.groups * 36 * s.cout * sizeof(float)));
      }
      for (int g = 0; g < s.groups; ++g)
        HIPCHK(launch_split_wino_u(c->wino_u[i] + g * per_g, c->wino_us[i] + g * per_g, c->wino_usc[i] + (size_t)g * 36 * s.cout, s.cin,
                                   s.cout, 36, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    c->wino_us_blob = c->blob;
    c->wino_us_tile = 4;
  }
  return SE3TN_OK;
}

// scratch of se3tn_fill_depth (3 float images + min/max + table) and the rasteriser's z-buffer: grown here, at start-up
// (se3tn_reserve) or on the first call with a larger frame.  Growing frees the old buffer, so the whole DEVICE is drained
// first: another stream of this context may still be reading it.
static int reserve_fill_depth(se3tn_ctx* c, size_t px) {
  if (px <= c->fd_pixels) return SE3TN_OK;
  HIPCHK(hipDeviceSynchronize());
  if (c->fd_buf) HIPCHK(hipFree(c->fd_buf));
  c->fd_buf = nullptr; c->fd_pixels = 0;
  HIPCHK(hipMalloc((void**)&c->fd_buf, (3 * px + 2 + 4098 + 2) * sizeof(float)));
  c->fd_pixels = px;
  return SE3TN_OK;
}
static int reserve_zbuf(se3tn_ctx* c, size_t px) {
  if (px <= c->zbuf_px) return SE3TN_OK;
  HIPCHK(hipDeviceSynchronize());
  if (c->zbuf) HIPCHK(hipFree(c->zbuf));
  c->zbuf = nullptr; c->zbuf_px = 0;
  HIPCHK(hipMalloc((void**)&c->zbuf, sizeof(unsigned long long) * px));
  c->zbuf_px = px;
  return SE3TN_OK;
}
static bool stream_is_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}

extern "C" {

const char* se3tn_version(void) { return "se3tracknet-gfx950 0.5.0 (blob v7)"; }
const char* se3tn_last_error(void) { return g_err.c_str(); }

int se3tn_create(int device, int max_batch, se3tn_ctx** out) {
  if (!out || max_batch < 1) return fail(SE3TN_E_ARG, "se3tn_create: bad argument");
  se3tn_ctx* c = new se3tn_ctx();
  c->device = device;
  c->max_batch = max_batch;
  c->L = blob_layout();
  c->SL = split_layout();
  if (const char* e = std::getenv("SE3TN_WINOGRAD_AUTO_TILE_AB2")) c->auto_tile_override[0] = (std::atoi(e) == 4 || std::atoi(e) == 6) ? std::atoi(e) : 0;
  if (const char* e = std::getenv("SE3TN_WINOGRAD_AUTO_TILE_HEADS")) c->auto_tile_override[1] = (std::atoi(e) == 4 || std::atoi(e) == 6) ? std::atoi(e) : 0;
  if (const char* e = std::getenv("SE3TN_TRUNK_KERNEL")) c->trunk_kernel = std::atoi(e) == 2 ? 2 : 1;
  if (const char* e = std::getenv("SE3TN_WINO_GEMMP")) c->gemmp = std::atoi(e) != 0 ? 1 : 0;
  if (const char* e = std::getenv("SE3TN_WINOGRAD_FUSE")) c->wino_fuse = std::atoi(e) != 0;      // developer A/B switch (and the tests'
                                                                                                  // bit-equality check of the two forms)
  if (const char* e = std::getenv("SE3TN_TRUNK_WINOGRAD")) c->wino64_min_batch = std::atoi(e);   // developer A/B switch
  if (const char* e = std::getenv("SE3TN_TRUNK_WINOGRAD_FILL")) c->wino64_min_fill = std::atoi(e);   // (scripts/trunk_sweep.sh)
  for (int i = 0; i < 8; ++i) { c->mean[i] = 0.0; c->stdv[i] = 1.0; }
  if (device >= 0) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { delete c; return hipfail(e, "hipGetDeviceProperties"); }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      std::string a = prop.gcnArchName;
      delete c;
      return fail(SE3TN_E_DEVICE, "device is " + a + ", this library is built for gfx950 only");
    }
    if (prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount;
    e = hipSetDevice(device);
    if (e != hipSuccess) { delete c; return hipfail(e, "hipSetDevice"); }
    const size_t mb = (size_t)max_batch;
    auto padded = [&](int s, int ch) { return (mb * (s + 2) * (s + 2) + PAD_SLACK_PX) * ch; };
    struct { float** p; size_t words; bool zero; } bufs[] = {
        {&c->inA, (mb * IN_P + IN_SLACK_ROWS) * IN_P * 4, true},
        {&c->inB, (mb * IN_P + IN_SLACK_ROWS) * IN_P * 4, true},
        {&c->stem, mb * S1 * S1 * 128, false}, {&c->pool, padded(S2, 128), true},
        {&c->t64, padded(S2, 128), true},      {&c->q64, padded(S2, 128), true},
        {&c->ab, padded(S3, 256), true},       {&c->ab_t, padded(S3, 256), true},
        {&c->head, padded(S4, 1024), true},    {&c->head_t, padded(S4, 1024), true},
        {&c->head_f, padded(S4, 1024), true},
        {&c->logits, mb * 6, true},           {&c->fcpart, mb * 192, true}};
    for (auto& b : bufs) {
      e = hipMalloc((void**)b.p, b.words * sizeof(float));
      if (e != hipSuccess) { se3tn_destroy(c); return hipfail(e, "hipMalloc(workspace)"); }
      // the one-pixel borders are written here once and never again (kernels store interiors only)
      if (b.zero) e = hipMemset(*b.p, 0, b.words * sizeof(float));
      if (e != hipSuccess) { se3tn_destroy(c); return hipfail(e, "hipMemset(workspace)"); }
    }
    // split-K workspace: 16 slices of the widest layer (1024 couts x 121 px) up to batch 16
    c->part_bytes = (size_t)16 * 1024 * 121 * 16 * sizeof(float);
    e = hipMalloc((void**)&c->part, c->part_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&c->splitk_sem, sizeof(int) * 2 * SE3TN_SPLITK_MAX_TILES);
    if (e == hipSuccess) e = hipMemset(c->splitk_sem, 0, sizeof(int) * 2 * SE3TN_SPLITK_MAX_TILES);
    if (e == hipSuccess) e = hipMalloc((void**)&c->tail_arrive, sizeof(int) * c->max_batch);
    if (e == hipSuccess) e = hipMemset(c->tail_arrive, 0, sizeof(int) * c->max_batch);
    if (const char* sf = std::getenv("SE3TN_SPLITK_FUSED")) c->splitk_fused = std::atoi(sf) != 0;
    if (const char* sk = std::getenv("SE3TN_SMALL_KERNELS")) c->small_kernels = std::atoi(sk) != 0;
    if (const char* tp = std::getenv("SE3TN_TAIL_PARTS")) { c->tail_parts = std::atoi(tp) != 0; c->tail_parts_ch = std::atoi(tp) == 2 ? 32 : 16; }
    if (e != hipSuccess) { se3tn_destroy(c); return hipfail(e, "hipMalloc(split-K workspace)"); }
    e = hipMalloc((void**)&c->zbuf, sizeof(unsigned long long) * RES * RES);
    if (e != hipSuccess) { se3tn_destroy(c); return hipfail(e, "hipMalloc(zbuf)"); }
    c->zbuf_px = (size_t)RES * RES;
    e = hipMalloc((void**)&c->overflow, sizeof(int));
    if (e == hipSuccess) e = hipMemset(c->overflow, 0, sizeof(int));
    if (e != hipSuccess) { se3tn_destroy(c); return hipfail(e, "hipMalloc(overflow flag)"); }
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { se3tn_destroy(c); return hipfail(e, "hipDeviceSynchronize"); }
  }
  *out = c;
  return SE3TN_OK;
}

void se3tn_destroy(se3tn_ctx* c) {
  if (!c) return;
  if (c->device >= 0) {
    float* bufs[] = {c->inA, c->inB, c->stem, c->pool, c->t64, c->q64, c->ab, c->ab_t, c->head,
                     c->head_t, c->head_f, c->logits, c->fcpart, c->part, (float*)c->splitk_sem, (float*)c->tail_arrive, c->blob_owned, c->split_w, c->wino_v, c->wino_m,
                     c->wino_u[0], c->wino_u[1], c->wino_u[2], c->wino_u[3], c->wino_u6[0], c->wino_u6[1], c->wino_u6[2], c->wino_u6[3],
                     c->wino_us[0], c->wino_us[1], c->wino_us[2], c->wino_us[3], c->wino_usc[0], c->wino_usc[1], c->wino_usc[2],
                     c->wino_usc[3], c->wino64_u[0], c->wino64_u[1], c->wino64_u[2], c->wino64_u[3]};
    for (float* b : bufs)
      if (b) (void)hipFree(b);
    for (auto& g : c->graphs)
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (c->overflow) (void)hipFree(c->overflow);
    if (c->trk_host) (void)hipHostFree(c->trk_host);
    if (c->trk_out_host) (void)hipHostFree(c->trk_out_host);
    if (c->trk_copy_event) (void)hipEventDestroy(c->trk_copy_event);
    if (c->trk_copy_stream) (void)hipStreamDestroy(c->trk_copy_stream);
    for (void* b : {(void*)c->trk_dev, (void*)c->trk_rgbA, (void*)c->trk_depthA})
      if (b) (void)hipFree(b);
    if (c->zbuf) (void)hipFree(c->zbuf);
    for (void* b : {(void*)c->tb_inst_host, (void*)c->tb_out_host, (void*)c->tb_stage_host})
      if (b) (void)hipHostFree(b);
    for (void* b : {(void*)c->tb_inst_dev, (void*)c->tb_zbuf, (void*)c->tb_rgbA, (void*)c->tb_depthA, (void*)c->tb_out_dev, (void*)c->tb_stage_dev})
      if (b) (void)hipFree(b);
    if (c->fd_buf) (void)hipFree(c->fd_buf);
    for (int s = 0; s < c->slots; ++s)
      for (auto& e : c->evs[s]) (void)hipEventDestroy(e);
  }
  delete c;
}

int se3tn_max_batch(const se3tn_ctx* c) { return c ? c->max_batch : 0; }

int se3tn_set_tensor(se3tn_ctx* c, const char* key, const float* data, const int64_t* shape, int ndim) {
  if (!c || !key || !data || !shape || ndim < 1 || ndim > 4) return fail(SE3TN_E_ARG, "se3tn_set_tensor: bad argument");
  for (const ExpectedTensor& e : expected_tensors()) {
    if (e.key != key) continue;
    if ((int)e.shape.size() != ndim) return fail(SE3TN_E_SHAPE, std::string("rank mismatch for ") + key);
    size_t count = 1;
    for (int i = 0; i < ndim; ++i) {
      if (shape[i] != e.shape[i]) return fail(SE3TN_E_SHAPE, std::string("shape mismatch for ") + key);
      count *= (size_t)shape[i];
    }
    StoredTensor& t = c->tensors[key];
    t.shape.assign(shape, shape + ndim);
    t.data.assign(data, data + count);
    return SE3TN_OK;
  }
  return fail(SE3TN_E_KEY, std::string("unexpected state_dict key: ") + key);
}

int se3tn_pack_weights(se3tn_ctx* c) {
  if (!c) return fail(SE3TN_E_ARG, "null ctx");
  std::string err = pack_blob(c->tensors, c->packed);
  if (!err.empty()) return fail(SE3TN_E_KEY, err);
  c->tensors.clear();
  return SE3TN_OK;
}

size_t se3tn_packed_bytes(const se3tn_ctx* c) { return c ? c->L.total * sizeof(float) : 0; }
const void* se3tn_packed_host(const se3tn_ctx* c) { return (c && !c->packed.empty()) ? c->packed.data() : nullptr; }

int se3tn_upload_weights(se3tn_ctx* c, void* stream) {
  if (!c || c->device < 0) return fail(SE3TN_E_ARG, "se3tn_upload_weights: no device context");
  if (c->packed.empty()) return fail(SE3TN_E_STATE, "se3tn_upload_weights: call se3tn_pack_weights first");
  DeviceGuard dg(c->device);   // init-time entry point: runs on the context's device, leaves the caller's current device alone
  HIPCHK(dg.err);
  if (!c->blob_owned) HIPCHK(hipMalloc((void**)&c->blob_owned, c->L.total * sizeof(float)));
  HIPCHK(hipMemcpyAsync(c->blob_owned, c->packed.data(), c->L.total * sizeof(float), hipMemcpyHostToDevice,
                        (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));  // init-time only: the host vector may go away
  c->blob = c->blob_owned;
  c->wino_blob = nullptr;  // same address, new contents: every derived plane set follows the weights
  c->wino6_blob = nullptr;
  c->wino64_blob = nullptr;
  c->split_blob = nullptr;
  c->wino_us_blob = nullptr;
  if (int rc = wino_prepare(c, (hipStream_t)stream)) return rc;
  return split_prepare(c, (hipStream_t)stream);
}

int se3tn_bind_weights(se3tn_ctx* c, const void* device_blob, size_t bytes) {
  if (!c || c->device < 0 || !device_blob) return fail(SE3TN_E_ARG, "se3tn_bind_weights: bad argument");
  if (bytes != c->L.total * sizeof(float)) return fail(SE3TN_E_SHAPE, "se3tn_bind_weights: blob size mismatch");
  DeviceGuard dg(c->device);
  HIPCHK(dg.err);
  uint32_t hdr[4];
  HIPCHK(hipMemcpy(hdr, device_blob, sizeof(hdr), hipMemcpyDeviceToHost));  // init-time validation
  if (hdr[0] != BLOB_MAGIC || hdr[1] != BLOB_VERSION || hdr[2] != (uint32_t)c->L.total)
    return fail(SE3TN_E_SHAPE, "se3tn_bind_weights: bad blob header");
  c->blob = (const float*)device_blob;
  c->wino_blob = nullptr;
  c->wino6_blob = nullptr;
  c->wino64_blob = nullptr;
  c->split_blob = nullptr;
  c->wino_us_blob = nullptr;
  if (int rc = wino_prepare(c, nullptr)) return rc;
  return split_prepare(c, nullptr);
}

int se3tn_set_winograd(se3tn_ctx* c, int min_batch, int tile) {
  if (!c || min_batch < 0 || (tile != 0 && tile != 2 && tile != 4 && tile != 6 && tile != SE3TN_WINOGRAD_TILE_AUTO && tile != SE3TN_WINOGRAD_TILE_6_4))
    return fail(SE3TN_E_ARG, "se3tn_set_winograd: min_batch >= 0 and tile in {0, 2, 4, 6, SE3TN_WINOGRAD_TILE_6_4, SE3TN_WINOGRAD_TILE_AUTO}");
  c->wino_min_batch = min_batch;
  if (tile) c->wino_tile = tile;
  if (c->device < 0) return SE3TN_OK;
  if (int rc = wino_prepare(c, nullptr)) return rc;
  return split_prepare(c, nullptr);
}

// The fused trunk kernel occupies a CU per workgroup (126 KB of LDS) for ~52 us whatever the batch: it beats the direct kernels when
// its 4 n groups workgroups fill the rounds of the CUs to more than half (measured, scripts/trunk_sweep.sh -> profiles/r03f_trunk_sweep.txt:
// n = 64 -> 2 | 1 rounds, 108 | 56 us vs 155 | 80 us direct; n = 20 grouped = 0.63 round 53 vs 80; but n = 16 grouped and n = 32 single
// = half a round 52 vs 45; the direct kernels step from one round of their own tiles to two at n = 17 grouped / n = 34 single).
static bool wino64_pays(const se3tn_ctx* c, int n, int groups) {
  if (c->wino64_min_batch <= 0 || n < c->wino64_min_batch) return false;
  const long wgs = 4L * n * groups, rounds = (wgs + c->num_cus - 1) / c->num_cus;
  return 100 * wgs >= (long)c->wino64_min_fill * rounds * c->num_cus;   // the rounds are at least min_fill % full
}

int se3tn_set_trunk_winograd(se3tn_ctx* c, int min_batch, int min_fill_percent) {
  if (!c || min_batch < 0 || min_fill_percent < 0 || min_fill_percent > 100)
    return fail(SE3TN_E_ARG, "se3tn_set_trunk_winograd: min_batch >= 0, 0 <= min_fill_percent <= 100");
  c->wino64_min_batch = min_batch;
  c->wino64_min_fill = min_fill_percent;
  if (c->device < 0) return SE3TN_OK;
  return wino_prepare(c, nullptr);
}

int se3tn_get_trunk_winograd(const se3tn_ctx* c, int* min_batch, int* min_fill_percent) {
  if (!c || !min_batch || !min_fill_percent) return fail(SE3TN_E_ARG, "se3tn_get_trunk_winograd: bad argument");
  *min_batch = c->wino64_min_batch;
  *min_fill_percent = c->wino64_min_fill;
  return SE3TN_OK;
}

int se3tn_get_winograd(const se3tn_ctx* c, int* min_batch, int* tile) {
  if (!c || !min_batch || !tile) return fail(SE3TN_E_ARG, "se3tn_get_winograd: bad argument");
  *min_batch = c->wino_min_batch;
  *tile = c->wino_tile;
  return SE3TN_OK;
}

int se3tn_set_normalization(se3tn_ctx* c, const double mean[8], const double stdv[8]) {
  if (!c || !mean || !stdv) return fail(SE3TN_E_ARG, "se3tn_set_normalization: bad argument");
  std::memcpy(c->mean, mean, sizeof(c->mean));
  std::memcpy(c->stdv, stdv, sizeof(c->stdv));
  c->have_norm = true;
  return SE3TN_OK;
}

int se3tn_set_precision(se3tn_ctx* c, int mode) {
  if (!c || (mode != SE3TN_PREC_F32 && mode != SE3TN_PREC_F16X3)) return fail(SE3TN_E_ARG, "se3tn_set_precision: bad mode");
  c->prec = mode;
  // first selection of f16x3 (or new weights since): derive the split panels from the bound blob -- here, not in se3tn_infer
  // (after U: the split rows of the F(4x4) planes come from wino_u)
  if (c->device >= 0) {
    if (int rc = wino_prepare(c, nullptr)) return rc;
  }
  return split_prepare(c, nullptr);
}

int se3tn_reserve(se3tn_ctx* c, int H, int W) {
  if (!c || c->device < 0 || H < 1 || W < 1) return fail(SE3TN_E_ARG, "se3tn_reserve: bad argument");
  DeviceGuard dg(c->device);
  HIPCHK(dg.err);
  if (int rc = reserve_fill_depth(c, (size_t)H * W)) return rc;
  if (int rc = reserve_zbuf(c, (size_t)H * W)) return rc;
  if (int rc = wino_prepare(c, nullptr)) return rc;
  return split_prepare(c, nullptr);
}

size_t se3tn_split_weights_bytes(const se3tn_ctx* c) { return c ? c->SL.total * sizeof(float) : 0; }
const void* se3tn_split_weights_device(const se3tn_ctx* c) { return (c && c->split_blob) ? c->split_w : nullptr; }
int se3tn_split_weights_host(const void* packed_blob_host, size_t blob_bytes, void* out_split, size_t out_bytes) {
  const BlobLayout L = blob_layout();
  const SplitLayout S = split_layout();
  if (!packed_blob_host || !out_split || blob_bytes != L.total * sizeof(float) || out_bytes != S.total * sizeof(float))
    return fail(SE3TN_E_ARG, "se3tn_split_weights_host: bad argument");
  split_blob_host((const float*)packed_blob_host, (float*)out_split);
  return SE3TN_OK;
}

int se3tn_keep_intermediates(se3tn_ctx* c, int on) {
  if (!c) return fail(SE3TN_E_ARG, "null ctx");
  c->keep_intermediates = on != 0;
  return SE3TN_OK;
}

int se3tn_overflow(se3tn_ctx* c, int* flag) {
  if (!c || c->device < 0 || !flag) return fail(SE3TN_E_ARG, "se3tn_overflow: bad argument");
  HIPCHK(hipMemcpy(flag, c->overflow, sizeof(int), hipMemcpyDeviceToHost));
  if (*flag) HIPCHK(hipMemset(c->overflow, 0, sizeof(int)));
  return SE3TN_OK;
}

int se3tn_set_offset_rule(se3tn_ctx* c, int rule) {
  if (!c || (rule != SE3TN_OFFSET_RULE_NUMPY1 && rule != SE3TN_OFFSET_RULE_NUMPY2)) return fail(SE3TN_E_ARG, "se3tn_set_offset_rule: bad rule");
  c->offset_rule = rule;
  return SE3TN_OK;
}
int se3tn_get_offset_rule(const se3tn_ctx* c) { return c ? c->offset_rule : -1; }

int se3tn_set_normalizers(se3tn_ctx* c, double tn, double rn) {
  if (!c) return fail(SE3TN_E_ARG, "null ctx");
  c->tn = tn;
  c->rn = rn;
  return SE3TN_OK;
}

float* se3tn_input_buffer(se3tn_ctx* c, int which) { return !c ? nullptr : (which == 0 ? c->inA : c->inB); }

int se3tn_preprocess(se3tn_ctx* c, const se3tn_crop* crops, int n, float* out, void* stream) {
  if (!c || c->device < 0 || !crops || !out || n < 0) return fail(SE3TN_E_ARG, "se3tn_preprocess: bad argument");
  if (!c->have_norm) return fail(SE3TN_E_STATE, "se3tn_preprocess: call se3tn_set_normalization first");
  // the context's own input buffers hold max_batch images: more crops would run past them
  if ((out == c->inA || out == c->inB) && n > c->max_batch)
    return fail(SE3TN_E_ARG, "se3tn_preprocess: n > max_batch for the context's input buffer");
  for (int i = 0; i < n; ++i) {
    const se3tn_crop& k = crops[i];
    if (!k.rgb || !k.depth || k.H < 1 || k.W < 1 || k.right <= k.left || k.bottom <= k.top || (k.stats & ~1))
      return fail(SE3TN_E_ARG, "se3tn_preprocess: bad crop descriptor " + std::to_string(i));
  }
  CropArgs a;
  std::memcpy(a.mean, c->mean, sizeof(a.mean));
  std::memcpy(a.stdv, c->stdv, sizeof(a.stdv));
  for (int i0 = 0; i0 < n; i0 += CropArgs::MAX) {
    a.n = (n - i0 < CropArgs::MAX) ? n - i0 : CropArgs::MAX;
    std::memcpy(a.c, crops + i0, sizeof(se3tn_crop) * a.n);
    // the context's own input buffers are zero-bordered [n,182,182,4]; anything else is plain NHWC
    a.padded = (out == c->inA || out == c->inB) ? 1 : 0;
    // the context's input buffers hold split pixels in f16x3 mode (consumed by the f16 stem)
    a.split = (a.padded && c->prec == SE3TN_PREC_F16X3) ? 1 : 0;
    a.overflow = c->overflow;
    a.offset_rule = c->offset_rule;
    if (a.padded) c->in_split[out == c->inA ? 0 : 1] = a.split;
    a.out = out + (size_t)i0 * (a.padded ? IN_P * IN_P : RES * RES) * 4;
    a.out2 = nullptr; a.n_first = a.n;
    HIPCHK(launch_preprocess(a, (hipStream_t)stream));
  }
  return SE3TN_OK;
}

int se3tn_crop_raw(se3tn_ctx* c, const se3tn_crop* k, uint8_t* rgb_out, uint16_t* depth_out, void* stream) {
  if (!c || c->device < 0 || !k || !rgb_out || !depth_out) return fail(SE3TN_E_ARG, "se3tn_crop_raw: bad argument");
  if (!k->rgb || !k->depth || k->H < 1 || k->W < 1 || k->right <= k->left || k->bottom <= k->top)
    return fail(SE3TN_E_ARG, "se3tn_crop_raw: bad crop descriptor");
  HIPCHK(launch_crop_raw(*k, rgb_out, depth_out, (hipStream_t)stream));
  return SE3TN_OK;
}

static int prof_mark(se3tn_ctx* c, hipStream_t st, const char* name, bool is_conv) {
  if (!c->prof) return 0;
  if (c->n_launch >= MAX_LAUNCHES) return 0;
  c->names[c->n_launch] = name;
  c->is_conv[c->n_launch] = is_conv;
  ++c->n_launch;
  return (int)hipEventRecord(c->ev[c->n_launch], st);
}

// Profile names say which ALGORITHM a launch took ("convAB2.conv1 [F(6x6)]"): the tests assert from them that the path they mean to
// check is the one that ran.  Interned once per (name, tile); the strings live as long as the library.
static const char* algo_name(const char* name, int tile, bool fused_block = false) {
  static std::mutex mu;
  static std::map<std::pair<std::string, int>, std::string> names;
  std::lock_guard<std::mutex> lk(mu);
  auto key = std::make_pair(std::string(name), tile * 2 + (fused_block ? 1 : 0));
  auto it = names.find(key);
  if (it == names.end()) {
    const std::string t = std::to_string(tile);
    it = names.emplace(key, std::string(name) + " [F(" + t + "x" + t + ")]" + (fused_block ? " fused block" : "")).first;
  }
  return it->second.c_str();
}

static int infer_launch(se3tn_ctx* c, const float* A, const float* B, int n, int layout, float* trans, float* rot,
                        const double* poseA, double* poseB, void* stream);

int se3tn_enable_graphs(se3tn_ctx* c, int on) {
  if (!c || c->device < 0) return fail(SE3TN_E_ARG, "se3tn_enable_graphs: no device context");
  c->use_graphs = on != 0;
  return SE3TN_OK;
}

// With graphs enabled, the ~25 dependent launches of one se3tn_infer are captured once per argument
// set (second call with identical arguments; the first runs eagerly) and replayed with one
// hipGraphLaunch afterwards: the batch-1 tracking step is launch-bound (kernels of 10-30 us).
static int infer_graph_or_launch(se3tn_ctx* c, const float* A, const float* B, int n, int layout, float* trans, float* rot,
                                 const double* poseA, double* poseB, void* stream);

int se3tn_infer(se3tn_ctx* c, const float* A, const float* B, int n, int layout, float* trans, float* rot,
                const double* poseA, double* poseB, void* stream) {
  if (c && c->rearm_counters && c->device >= 0 && !stream_is_capturing((hipStream_t)stream)) {
    // a launch sequence that failed half-way can leave the tail's arrival counters / the split-K semaphores non-zero, and every
    // later frame would then find no "last" workgroup: clear them, stream-ordered, before the next frame
    if (c->tail_arrive) HIPCHK(hipMemsetAsync(c->tail_arrive, 0, sizeof(int) * c->max_batch, (hipStream_t)stream));
    if (c->splitk_sem) HIPCHK(hipMemsetAsync(c->splitk_sem, 0, sizeof(int) * 2 * SE3TN_SPLITK_MAX_TILES, (hipStream_t)stream));
    c->rearm_counters = false;
  }
  const int rc = infer_graph_or_launch(c, A, B, n, layout, trans, rot, poseA, poseB, stream);
  if (rc != SE3TN_OK && rc != SE3TN_E_ARG && rc != SE3TN_E_STATE && c) c->rearm_counters = true;
  return rc;
}

static int infer_graph_or_launch(se3tn_ctx* c, const float* A, const float* B, int n, int layout, float* trans, float* rot,
                                 const double* poseA, double* poseB, void* stream) {
  if (!c || !c->use_graphs || c->prof || stream == nullptr)  // the null stream cannot be captured
    return infer_launch(c, A, B, n, layout, trans, rot, poseA, poseB, stream);
  GraphKey key;
  std::memset(&key, 0, sizeof(key));
  key.A = A; key.B = B; key.trans = trans; key.rot = rot; key.poseA = poseA; key.poseB = poseB; key.blob = c->blob;
  key.n = n; key.layout = layout; key.prec = c->prec;
  key.wino_min_batch = c->wino_min_batch; key.wino_tile = c->wino_tile; key.keep = c->keep_intermediates ? 1 : 0;
  key.wino64 = c->wino64_min_batch; key.wino64_fill = c->wino64_min_fill; key.tn = c->tn; key.rn = c->rn;
  key.small_kernels = c->small_kernels ? 1 : 0; key.splitk_fused = c->splitk_fused ? 1 : 0; key.wino_fuse = c->wino_fuse ? 1 : 0;
  key.gemmp = c->gemmp; key.tail_parts = c->tail_parts ? c->tail_parts_ch : 0;
  hipStream_t st = (hipStream_t)stream;
  for (auto& g : c->graphs) {
    if (!(g.key == key)) continue;
    if (g.exec) {
      const int want_split = c->prec == SE3TN_PREC_F16X3 ? 1 : 0;
      if ((A == c->inA && c->in_split[0] != want_split) || (B == c->inB && c->in_split[1] != want_split))
        return fail(SE3TN_E_STATE, "se3tn_infer: the input buffers were filled under a different precision mode");
      HIPCHK(hipGraphLaunch(g.exec, st));
      c->last_fast = g.fast;
      c->head_final = g.head_final;
      return SE3TN_OK;
    }
    // second sighting: capture
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = infer_launch(c, A, B, n, layout, trans, rot, poseA, poseB, stream);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc != SE3TN_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return hipfail(e, "hipStreamEndCapture");
    const hipError_t e2 = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e2 != hipSuccess) { g.exec = nullptr; return hipfail(e2, "hipGraphInstantiate"); }
    g.fast = c->last_fast;
    g.head_final = c->head_final;
    HIPCHK(hipGraphLaunch(g.exec, st));
    return SE3TN_OK;
  }
  if (c->graphs.size() >= 16) {  // bounded cache: drop the oldest entry
    if (c->graphs.front().exec) (void)hipGraphExecDestroy(c->graphs.front().exec);
    c->graphs.erase(c->graphs.begin());
  }
  GraphEntry ge;
  ge.key = key;
  c->graphs.push_back(ge);
  return infer_launch(c, A, B, n, layout, trans, rot, poseA, poseB, stream);
}

static int infer_launch(se3tn_ctx* c, const float* A, const float* B, int n, int layout, float* trans, float* rot,
                        const double* poseA, double* poseB, void* stream) {
  if (!c || c->device < 0 || !A || !B) return fail(SE3TN_E_ARG, "se3tn_infer: bad argument");
  if (n < 1 || n > c->max_batch) return fail(SE3TN_E_ARG, "se3tn_infer: n outside [1, max_batch]");
  if (!c->blob) return fail(SE3TN_E_STATE, "se3tn_infer: weights not uploaded/bound");
  if ((poseA == nullptr) != (poseB == nullptr)) return fail(SE3TN_E_ARG, "se3tn_infer: poseA and poseB go together");
  if (layout != SE3TN_NCHW && layout != SE3TN_NHWC) return fail(SE3TN_E_ARG, "se3tn_infer: bad layout");
  hipStream_t st = (hipStream_t)stream;
  const float* W = c->blob;
  const BlobLayout& L = c->L;
  c->n_launch = 0;
  int slot = 0;
  if (c->prof) {
    slot = (int)(c->infer_count++ % c->slots);
    c->ev = c->evs[slot];
    HIPCHK(hipEventRecord(c->ev[0], st));
  }

  const int want_split = c->prec == SE3TN_PREC_F16X3 ? 1 : 0;
  if (A != c->inA || B != c->inB) {  // external tensors: copy into the zero-bordered input buffers
    if (A != c->inA) {
      HIPCHK(launch_to_padded_input(A, c->inA, n, layout == SE3TN_NCHW, want_split, c->overflow, st));
      c->in_split[0] = want_split;
    }
    if (B != c->inB) {
      HIPCHK(launch_to_padded_input(B, c->inB, n, layout == SE3TN_NCHW, want_split, c->overflow, st));
      c->in_split[1] = want_split;
    }
    HIPCHK((hipError_t)prof_mark(c, st, "to_padded_input", false));
    A = c->inA;
    B = c->inB;
  }
  if (c->in_split[0] != want_split || c->in_split[1] != want_split)
    return fail(SE3TN_E_STATE, "se3tn_infer: the input buffers were filled under a different precision mode "
                               "(call se3tn_set_precision before se3tn_preprocess)");
  if (want_split && c->split_blob != c->blob)
    return fail(SE3TN_E_STATE, "se3tn_infer: f16x3 split panels not derived (se3tn_set_precision after the weights are bound)");
  const float* WS = c->split_w;
  const SplitLayout& SL = c->SL;
  const bool fast = c->prec == SE3TN_PREC_F16X3;  // both the big-tile and the split-K kernels
  c->last_fast = fast;
  // batch 1-2 (the per-frame regime): stem + max-pool of both branches in ONE launch of 16-pool-pixel tiles; the 88 x 88 x 128 stem
  // map is not stored, so a caller who asked for the intermediates gets the batch-64 pair
  const bool small_stem = c->small_kernels && !want_split && !c->keep_intermediates && n <= SE3TN_STEM_SMALL_MAX_N;
  if (small_stem) {
    HIPCHK(launch_stem_pool_small(A, B, W + L.stem_w, W + L.stem_b, c->pool, n, st));
    HIPCHK((hipError_t)prof_mark(c, st, "stem7x7 + maxpool [small tiles]", false));
  } else {
    if (want_split)
      HIPCHK(launch_stem(A, B, WS + SL.stem_ws, W + L.stem_b, WS + SL.stem_sc, c->stem, n, st));
    else
      HIPCHK(launch_stem(A, B, W + L.stem_w, W + L.stem_b, nullptr, c->stem, n, st));
    HIPCHK((hipError_t)prof_mark(c, st, "stem7x7_mfma", false));
#if !defined(SE3TN_ABLATE_POOL)   // (timing-only variant build: what the step costs WITHOUT the pool pass -- EXPERIMENTS item 43)
    HIPCHK(launch_maxpool(c->stem, c->pool, n, fast ? 1 : 0, st));
#endif
    HIPCHK((hipError_t)prof_mark(c, st, "maxpool3x3s2", false));
  }

  bool skip_reduce = false;   // set around the last head conv when tail_parts_kernel consumes its partial sums
  auto conv = [&](ConvId id, const float* in, int in_ld, int in_gs, const float* res, int res_ld, int res_gs,
                  float* out, int out_ld, int out_gs, int hin, int stride, int epi, const char* name) -> int {
    const Conv3& s = conv_specs()[id];
    const int ws = wino_slot(id);
    const int wtile = tile_for(c, n, ws >= 2 ? 1 : 0);
    const bool u_ready = wtile == 6 ? (c->wino_u6[0] && c->wino6_blob == c->blob) : (c->wino_tile_derived == wtile);
    if (ws >= 0 && !fast && c->wino_min_batch > 0 && n >= c->wino_min_batch && c->wino_v && u_ready) {
      WinoArgs w{};
      w.in = in; w.U = wtile == 6 ? c->wino_u6[ws] : c->wino_u[ws]; w.bias = W + L.conv_b[id]; w.res = res; w.out = out;
      w.V = c->wino_v; w.Mw = c->wino_m;
      w.in_ld = in_ld; w.res_ld = res_ld; w.out_ld = out_ld;
      w.m = wtile; w.nf = (w.m + 2) * (w.m + 2);
      w.H = hin; w.W = hin; w.th = (hin + w.m - 1) / w.m; w.tw = w.th;
      w.n = n; w.T = n * w.th * w.tw;
      w.C = s.cin; w.Cout = s.cout; w.groups = s.groups;
      w.in_gs = in_gs; w.res_gs = res_gs; w.out_gs = out_gs; w.bias_gs = s.cout;
      w.u_gs = (long long)w.nf * s.cin * s.cout;
      w.num_cus = c->num_cus; w.gemmp = c->gemmp;
      hipError_t e = launch_wino_conv(w, epi, st);
      if (e != hipSuccess) return hipfail(e, name);
      return prof_mark(c, st, c->prof ? algo_name(name, wtile) : name, true);
    }
    if (id <= L64_4 && !fast && wino64_pays(c, n, s.groups) && c->wino64_blob == c->blob && stride == 1 && hin == S2 && epi != 2) {
      const int slot64 = (int)id - (int)L64_1;
      hipError_t e = launch_wino64(in, in_ld, in_gs, c->wino64_u[slot64], (long long)16 * s.cin * s.cout, W + L.conv_b[id], s.cout, res,
                                   res_ld, res_gs, out, out_ld, out_gs, n, s.groups, epi, c->trunk_kernel, st);
      if (e != hipSuccess) return hipfail(e, name);
      static const char* const fused_names[4] = {"conv64 A2.conv1|B2.conv1 [fused F(2x2)]", "conv64 A2.conv2|B2.conv2 [fused F(2x2)]",
                                                 "conv64 B3.conv1 [fused F(2x2)]", "conv64 B3.conv2 [fused F(2x2)]"};
      return prof_mark(c, st, fused_names[slot64], true);   // (the profile says which algorithm a launch took)
    }
    ConvArgs a{};
    a.in = in; a.w = W + L.conv_w[id]; a.bias = W + L.conv_b[id];
    if (fast) {
      a.fast = 1;
      a.overflow = c->overflow;
      a.w = WS + SL.conv_ws[id];
      a.wscale = WS + SL.conv_sc[id];
    } a.res = res; a.out = out; a.part = c->part; a.part_bytes = c->part_bytes;
    a.sem = c->splitk_fused ? c->splitk_sem : nullptr;
    a.small_ok = c->small_kernels ? 1 : 0;
    a.skip_reduce = skip_reduce ? 1 : 0;
    a.in_ld = in_ld; a.res_ld = res_ld; a.out_ld = out_ld;
    a.H = hin; a.W = hin; a.Ho = (hin - 1) / stride + 1; a.Wo = a.Ho;
    a.M = n * a.Ho * a.Wo;
    a.groups = s.groups;
    a.in_gs = in_gs; a.res_gs = res_gs; a.out_gs = out_gs; a.bias_gs = s.cout;
    a.w_gs = (long long)conv3_words(s.cin, s.cout);
    hipError_t e = launch_conv3x3(a, s.cin, s.cout, stride, epi, st);
    if (e != hipSuccess) return hipfail(e, name);
    return prof_mark(c, st, name, true);
  };
  int rc;
  // 64-channel trunk: pool = a0|b0 ; t64 scratch ; q64 = a1|b1 -> a1|b2 (== torch.cat((a,b),1))
  if ((rc = conv(L64_1, c->pool, 128, 64, nullptr, 0, 0, c->t64, 128, 64, S2, 1, 0, "conv64 A2.conv1|B2.conv1"))) return rc;
  if ((rc = conv(L64_2, c->t64, 128, 64, c->pool, 128, 64, c->q64, 128, 64, S2, 1, 1, "conv64 A2.conv2|B2.conv2"))) return rc;
  if ((rc = conv(L64_3, c->q64 + 64, 128, 0, nullptr, 0, 0, c->t64 + 64, 128, 0, S2, 1, 0, "conv64 B3.conv1"))) return rc;
  if ((rc = conv(L64_4, c->t64 + 64, 128, 0, c->q64 + 64, 128, 0, c->q64 + 64, 128, 0, S2, 1, 1, "conv64 B3.conv2"))) return rc;
  if ((rc = conv(LAB1, c->q64, 128, 0, nullptr, 0, 0, c->ab, 256, 0, S2, 2, 2, "convAB1 s2"))) return rc;
  // ResnetBasicBlocks of 256 / 512 channels: at n >= wino_min_batch with F(4x4) or F(6x6) the whole block runs through
  // launch_wino_block (conv1's out-transform fused with conv2's in-transform; the heads' last out-transform fused
  // with avg-pool + FC + tanh); otherwise conv by conv (direct / split-K / F(2x2) kernels)
  auto block_ok = [&](int which) -> bool {
    const int bt = tile_for(c, n, which);
    const bool planes_ready = bt == 6 ? (!fast && c->wino_u6[0] && c->wino6_blob == c->blob) : (bt == 4 && c->wino_tile_derived == 4);
    return c->wino_fuse && c->wino_min_batch > 0 && n >= c->wino_min_batch && c->wino_v && planes_ready &&
           (!fast || (c->wino_us_blob == c->blob && c->wino_us_tile == 4));
  };
  struct MarkCtx { se3tn_ctx* c; hipStream_t st; const char* name; };
  auto mark_fn = [](void* p) -> int { MarkCtx* m = (MarkCtx*)p; return prof_mark(m->c, m->st, m->name, true); };
  auto block = [&](ConvId id1, ConvId id2, float* io, float* mid, int ld, int gs, int hin, const TailArgs* tl,
                   const char* name1, const char* name2) -> int {
    const Conv3& s = conv_specs()[id1];
    const int btile = tile_for(c, n, tl ? 1 : 0);
    WinoArgs w{};
    float* const* Uset = btile == 6 ? c->wino_u6 : c->wino_u;
    w.in = io; w.U = Uset[wino_slot(id1)]; w.bias = W + L.conv_b[id1]; w.res = nullptr; w.out = mid;
    w.V = c->wino_v; w.Mw = c->wino_m;
    w.in_ld = ld; w.res_ld = ld; w.out_ld = ld;
    w.m = btile; w.nf = (btile + 2) * (btile + 2);
    w.H = hin; w.W = hin; w.th = (hin + btile - 1) / btile; w.tw = w.th;
    w.n = n; w.T = n * w.th * w.tw;
    w.C = s.cin; w.Cout = s.cout; w.groups = s.groups;
    w.in_gs = gs; w.res_gs = gs; w.out_gs = gs; w.bias_gs = s.cout;
    w.u_gs = (long long)w.nf * s.cin * s.cout;
    w.num_cus = c->num_cus; w.gemmp = c->gemmp;
    const float *U2 = Uset[wino_slot(id2)], *usc2 = nullptr;
    float* keep2 = (tl && c->keep_intermediates) ? io : nullptr;
    if (fast) {   // f16x3: split-row activations / V / U, float32 M
      w.split = 1;
      w.U = c->wino_us[wino_slot(id1)]; w.uscale = c->wino_usc[wino_slot(id1)];
      U2 = c->wino_us[wino_slot(id2)]; usc2 = c->wino_usc[wino_slot(id2)];
      w.overflow = c->overflow;
      if (keep2) keep2 = c->head_f;   // the kept head activation is float32 (head itself holds split rows)
    }
    MarkCtx mc{c, st, c->prof ? algo_name(name1, btile, true) : name1};
    hipError_t e = launch_wino_block(w, U2, usc2, W + L.conv_b[id2], io, c->keep_intermediates ? 1 : 0, keep2, tl, st, mark_fn, &mc);
    if (e != hipSuccess) return hipfail(e, name2);
    return prof_mark(c, st, c->prof ? algo_name(name2, btile, true) : name2, true);
  };
  // f16x3 mode: the batched GEMMs of the 256-channel block are bandwidth-bound at the f16 matrix rate (41 FLOP per byte of V / M
  // traffic) and lose to the direct f16x3 kernels (0.247 vs 0.220 ms at batch 64); the 512-channel heads win (0.349 vs 0.427 ms)
  if (block_ok(0) && !fast) {
    if ((rc = block(LAB2_1, LAB2_2, c->ab, c->ab_t, 256, 0, S3, nullptr, "convAB2.conv1", "convAB2.conv2"))) return rc;
  } else {
    if ((rc = conv(LAB2_1, c->ab, 256, 0, nullptr, 0, 0, c->ab_t, 256, 0, S3, 1, 0, "convAB2.conv1"))) return rc;
    if ((rc = conv(LAB2_2, c->ab_t, 256, 0, c->ab, 256, 0, c->ab, 256, 0, S3, 1, 1, "convAB2.conv2"))) return rc;
  }
  if ((rc = conv(LH1, c->ab, 256, 0, nullptr, 0, 0, c->head, 1024, 0, S3, 2, 2, "trans|rot conv1 s2"))) return rc;
  if (block_ok(1)) {
    TailArgs tl{W + L.fc_w, W + L.fc_b, c->logits, c->fcpart, trans, rot, poseA, poseB, c->tn, c->rn};
    if ((rc = block(LH2_1, LH2_2, c->head, c->head_t, 1024, 512, S4, &tl, "trans|rot conv2.conv1",
                    "trans|rot conv2.conv2 + avgpool+fc+tanh+pose"))) return rc;
    c->head_final = fast ? c->head_f : c->head;
  } else {
    if ((rc = conv(LH2_1, c->head, 1024, 512, nullptr, 0, 0, c->head_t, 1024, 512, S4, 1, 0, "trans|rot conv2.conv1"))) return rc;
    float* head_out = fast ? c->head_f : c->head;
    // batch 1-5 (conv_slices_small: 8 partial-sum slices per output): the tail adds the slices itself -- no conv_reduce launch, the
    // final head map is not written (se3tn_keep_intermediates keeps the old sequence).  Same predicate as launch_conv3x3's.
    const bool parts_tail = !fast && c->small_kernels && c->tail_parts && !c->keep_intermediates && c->part && n <= SE3TN_SLICES_SMALL_MAX_N &&
                            conv_slices_small_count(512, 1, S4) == 8 &&
                            (size_t)8 * 2 * n * S4 * S4 * 512 * sizeof(float) <= c->part_bytes && !c->splitk_fused;
    skip_reduce = parts_tail;
    rc = conv(LH2_2, c->head_t, 1024, 512, c->head, 1024, 512, head_out, 1024, 512, S4, 1, 1, "trans|rot conv2.conv2");
    skip_reduce = false;
    if (rc) return rc;
    if (parts_tail) {
      c->head_final = nullptr;     // (not materialised in this configuration)
      const int M = n * S4 * S4;
      HIPCHK(launch_tail_parts(c->part, 8, (size_t)2 * M * 512, M, W + L.conv_b[LH2_2], c->head, 1024, W + L.fc_w, W + L.fc_b, c->logits, trans,
                               rot, poseA, poseB, c->tn, c->rn, n, st, c->fcpart, c->tail_arrive, c->tail_flag, c->tail_seq, c->tail_parts_ch));
      HIPCHK((hipError_t)prof_mark(c, st, "tail slices+avgpool+fc+tanh+pose", false));
    } else {
      c->head_final = head_out;
      HIPCHK(launch_tail(head_out, W + L.fc_w, W + L.fc_b, c->logits, trans, rot, poseA, poseB, c->tn, c->rn, n, st, c->fcpart, c->tail_arrive, c->tail_flag, c->tail_seq));
      HIPCHK((hipError_t)prof_mark(c, st, "tail avgpool+fc+tanh+pose", false));
    }
  }
  if (c->prof) c->slot_launches[slot] = c->n_launch;
  return SE3TN_OK;
}

int se3tn_get_feature(se3tn_ctx* c, int n, float* feature_nchw, void* stream) {
  if (!c || c->device < 0 || !feature_nchw || n < 1 || n > c->max_batch) return fail(SE3TN_E_ARG, "se3tn_get_feature: bad argument");
  HIPCHK(launch_padded_nhwc_to_nchw(c->ab, feature_nchw, n, S3, S3, 256, c->last_fast ? 1 : 0, (hipStream_t)stream));
  return SE3TN_OK;
}

const float* se3tn_logits(se3tn_ctx* c) { return c ? c->logits : nullptr; }

int se3tn_debug_buffer(se3tn_ctx* c, const char* name, const float** ptr, int32_t dims[3]) {
  if (!c || !name || !ptr || !dims) return fail(SE3TN_E_ARG, "se3tn_debug_buffer: bad argument");
  struct { const char* n; const float* p; int h, w, ch; } t[] = {
      {"inA", c->inA, IN_P, IN_P, 4},    {"inB", c->inB, IN_P, IN_P, 4},     {"stem", c->stem, S1, S1, 128},
      {"pool", c->pool, S2 + 2, S2 + 2, 128},    {"t64", c->t64, S2 + 2, S2 + 2, 128},
      {"q64", c->q64, S2 + 2, S2 + 2, 128},      {"ab", c->ab, S3 + 2, S3 + 2, 256},
      {"ab_t", c->ab_t, S3 + 2, S3 + 2, 256},    {"head", c->head_final ? c->head_final : c->head, S4 + 2, S4 + 2, 1024},
      {"head_t", c->head_t, S4 + 2, S4 + 2, 1024}};
  for (auto& e : t)
    if (std::strcmp(e.n, name) == 0) {
      *ptr = e.p; dims[0] = e.h; dims[1] = e.w; dims[2] = e.ch;
      return SE3TN_OK;
    }
  return fail(SE3TN_E_KEY, std::string("unknown buffer ") + name);
}

// ---- rasteriser: replaces VispyRenderer (vispy_renderer.py:47-178) on the per-frame path ---------
int se3tn_mesh_create(se3tn_ctx* c, const float* verts, const float* normals, const float* colors01, int V,
                      const int32_t* faces, int F, se3tn_mesh** out) {
  if (!c || c->device < 0 || !verts || !normals || !colors01 || !faces || V < 3 || F < 1 || !out)
    return fail(SE3TN_E_ARG, "se3tn_mesh_create: bad argument");
  for (int i = 0; i < 3 * F; ++i)
    if (faces[i] < 0 || faces[i] >= V) return fail(SE3TN_E_ARG, "se3tn_mesh_create: face index out of range");
  se3tn_mesh* m = new se3tn_mesh();
  m->V = V; m->F = F;
  struct { void** p; const void* src; size_t bytes; } up[] = {
      {(void**)&m->verts, verts, sizeof(float) * 3 * V},     {(void**)&m->normals, normals, sizeof(float) * 3 * V},
      {(void**)&m->colors, colors01, sizeof(float) * 3 * V}, {(void**)&m->faces, faces, sizeof(int) * 3 * F},
      {(void**)&m->vpost, nullptr, sizeof(float4) * V},      {(void**)&m->vsnap, nullptr, sizeof(int4) * V},
      {(void**)&m->big, nullptr, sizeof(int) * (1 + (size_t)F)}, {(void**)&m->clipq, nullptr, sizeof(int) * (1 + (size_t)F)}};
  for (auto& u : up) {
    hipError_t e = hipMalloc(u.p, u.bytes);
    if (e == hipSuccess && u.src) e = hipMemcpy(*u.p, u.src, u.bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { se3tn_mesh_destroy(m); return hipfail(e, "se3tn_mesh_create"); }
  }
  *out = m;
  return SE3TN_OK;
}

int se3tn_mesh_set_texture(se3tn_mesh* m, const float* uv, const uint8_t* rgb, int tw, int th, const float kd[3]) {
  if (!m || (rgb && (!uv || tw < 1 || th < 1))) return fail(SE3TN_E_ARG, "se3tn_mesh_set_texture: bad argument");
  if (kd) { m->kd[0] = kd[0]; m->kd[1] = kd[1]; m->kd[2] = kd[2]; }
  if (m->uv || m->tex) (void)hipDeviceSynchronize();   // a render may still be reading the old material
  if (m->uv) { (void)hipFree(m->uv); m->uv = nullptr; }
  if (m->tex) { (void)hipFree(m->tex); m->tex = nullptr; }
  m->tlevels = 0;
  if (!rgb) return SE3TN_OK;
  // mip pyramid: 2x2 box filter per level (what glGenerateMipmap implementations do), levels back to back
  std::vector<uint8_t> pyr(rgb, rgb + (size_t)tw * th * 3);
  int w = tw, h = th, levels = 1;
  size_t off = 0;
  m->tex_off[0] = 0;
  while ((w > 1 || h > 1) && levels < 16) {
    const int nw = w > 1 ? w / 2 : 1, nh = h > 1 ? h / 2 : 1;
    const size_t noff = off + (size_t)w * h * 3;
    pyr.resize(noff + (size_t)nw * nh * 3);
    const uint8_t* src = pyr.data() + off;
    uint8_t* dst = pyr.data() + noff;
    for (int y = 0; y < nh; ++y)
      for (int x = 0; x < nw; ++x)
        for (int ch = 0; ch < 3; ++ch) {
          const int x0 = 2 * x < w ? 2 * x : w - 1, x1 = 2 * x + 1 < w ? 2 * x + 1 : w - 1;
          const int y0 = 2 * y < h ? 2 * y : h - 1, y1 = 2 * y + 1 < h ? 2 * y + 1 : h - 1;
          const int sum = src[((size_t)y0 * w + x0) * 3 + ch] + src[((size_t)y0 * w + x1) * 3 + ch] +
                          src[((size_t)y1 * w + x0) * 3 + ch] + src[((size_t)y1 * w + x1) * 3 + ch];
          dst[((size_t)y * nw + x) * 3 + ch] = (uint8_t)((sum + 2) >> 2);
        }
    m->tex_off[levels] = (unsigned)noff;
    off = noff; w = nw; h = nh; ++levels;
  }
  // both buffers exist and are filled before either is published: the resolve kernel reads uv whenever tex is set
  uint8_t* d_tex = nullptr;
  float* d_uv = nullptr;
  hipError_t e = hipMalloc((void**)&d_tex, pyr.size());
  if (e == hipSuccess) e = hipMemcpy(d_tex, pyr.data(), pyr.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&d_uv, sizeof(float) * 2 * m->V);
  if (e == hipSuccess) e = hipMemcpy(d_uv, uv, sizeof(float) * 2 * m->V, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (d_tex) (void)hipFree(d_tex);
    if (d_uv) (void)hipFree(d_uv);
    return hipfail(e, "se3tn_mesh_set_texture");
  }
  m->tex = d_tex; m->uv = d_uv;
  m->tw = tw; m->th = th; m->tlevels = levels;
  return SE3TN_OK;
}

void se3tn_mesh_destroy(se3tn_mesh* m) {
  if (!m) return;
  void* bufs[] = {m->verts, m->normals, m->colors, m->faces, m->vpost, m->vsnap, m->big, m->clipq, m->uv, m->tex,
                  m->b_vpost, m->b_vsnap, m->b_big, m->b_clipq};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  delete m;
}

// what both renderers share
static void raster_common(RasterArgs& a, se3tn_ctx* c, se3tn_mesh* m, uint8_t* rgb, uint16_t* depth) {
  a.verts = m->verts; a.normals = m->normals; a.colors = m->colors; a.faces = m->faces; a.vpost = m->vpost; a.vsnap = m->vsnap;
  a.zbuf = c->zbuf; a.big = m->big; a.clipq = m->clipq; a.rgb = rgb; a.depth = depth; a.V = m->V; a.F = m->F;
  a.numpy_rule = c->offset_rule; a.sub_bits = c->raster_sub_bits;
}
// proj * view as the vertex shader evaluates it (vispy_renderer.py:92): per column of the result, products accumulated left to
// right in float32, never fused (volatile keeps the host compiler from contracting or re-associating)
static void raster_pv(RasterArgs& a, const float P[4][4], const float V[4][4]) {
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 4; ++r) {
      volatile float acc = P[r][0] * V[0][j];
      for (int k = 1; k < 4; ++k) {
        volatile float prod = P[r][k] * V[k][j];
        acc = acc + prod;
      }
      a.PV[4 * r + j] = acc;
    }
}

int se3tn_set_raster_rule(se3tn_ctx* c, int sub_bits) {
  if (!c || (sub_bits != 4 && sub_bits != 8)) return fail(SE3TN_E_ARG, "se3tn_set_raster_rule: sub-pixel bits must be 4 or 8");
  c->raster_sub_bits = sub_bits;
  return SE3TN_OK;
}
int se3tn_get_raster_rule(const se3tn_ctx* c) { return c ? c->raster_sub_bits : -1; }

int se3tn_set_small_kernels(se3tn_ctx* c, int on) {
  if (!c) return fail(SE3TN_E_ARG, "null ctx");
  c->small_kernels = on != 0;
  return SE3TN_OK;
}
int se3tn_get_small_kernels(const se3tn_ctx* c) { return c ? (c->small_kernels ? 1 : 0) : -1; }

// The float32 uniforms of the reference's VispyRenderer for one pose / window: a.PV, a.light, a.dA, a.dB
// (false: singular pose)
static bool vispy_uniforms(RasterArgs& a, const double ob_in_cam[16], const double K[9], const int32_t window[4]) {
  // The float32 uniforms the reference uploads, formed as it forms them (float64 numpy, then the cast of the upload):
  // update_cam_mat (vispy_renderer.py:135-150): ortho (rounded to float32 FIRST, :146) . proj in float64 ...
  const double n = R_NEAR_D, f = R_FAR_D;
  const double left = window[0], top = window[1], right = window[2], bottom = window[3];
  const double o00 = (double)(float)(2.0 / (right - left)), o03 = (double)(float)(-(right + left) / (right - left));
  const double o11 = (double)(float)(2.0 / (top - bottom)), o13 = (double)(float)(-(top + bottom) / (top - bottom));
  const double o22 = (double)(float)(-2.0 / (f - n)), o23 = (double)(float)(-(f + n) / (f - n));
  double OP[4][4] = {{o00 * K[0], 0.0, o00 * -K[2] + o03 * -1.0, 0.0},
                     {0.0, o11 * K[4], o11 * -K[5] + o13 * -1.0, 0.0},
                     {0.0, 0.0, o22 * (n + f) + o23 * -1.0, o22 * (n * f)},
                     {0.0, 0.0, -1.0, 0.0}};
  a.dA = OP[2][2]; a.dB = OP[2][3];   // projection_matrix[2,2], [3,2] of the stored TRANSPOSE (:149, :163-164)
  // ... view = ob2cam_gl = inv(glcam_in_cvcam) . ob_in_cam = diag(1,-1,-1,1) . ob_in_cam (predict.py:202-207), exact in float64
  float P32[4][4], V32[4][4];
  const double sgn[4] = {1.0, -1.0, -1.0, 1.0};
  for (int r = 0; r < 4; ++r)
    for (int q = 0; q < 4; ++q) {
      P32[r][q] = (float)OP[r][q];
      V32[r][q] = (float)(sgn[r] * ob_in_cam[4 * r + q]);
    }
  raster_pv(a, P32, V32);
  // light_direction = (inv(ob2cam_gl^T) . [0, 0.1, -0.9, 1])[:3] (vispy_renderer.py:172) in float64.  With G = [R' t'; 0 1]:
  // inv(G^T) = [inv(R'^T) 0; * 1], so the first three components are inv(R')^T . (0, 0.1, -0.9) -- a true inverse (adjugate /
  // determinant), not R' itself: poses composed over many frames are orthonormal only to float32 rounding.
  double R[3][3], adjT[3][3];
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) R[r][q] = sgn[r] * ob_in_cam[4 * r + q];
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) {   // cofactor (r, q) = entry (r, q) of adj(R)^T = det(R) inv(R)^T
      const int r1 = (r + 1) % 3, r2 = (r + 2) % 3, q1 = (q + 1) % 3, q2 = (q + 2) % 3;
      adjT[r][q] = R[r1][q1] * R[r2][q2] - R[r1][q2] * R[r2][q1];
    }
  const double det = R[0][0] * adjT[0][0] + R[0][1] * adjT[0][1] + R[0][2] * adjT[0][2];
  if (det == 0.0) return false;
  const double l[3] = {0.0, 0.1, -0.9};
  for (int r = 0; r < 3; ++r) a.light[r] = (float)((adjT[r][0] * l[0] + adjT[r][1] * l[1] + adjT[r][2] * l[2]) / det);
  return true;
}

int se3tn_render(se3tn_ctx* c, se3tn_mesh* m, const double ob_in_cam[16], const double K[9], const int32_t window[4],
                 uint8_t* rgb, uint16_t* depth, void* stream) {
  if (!c || c->device < 0 || !m || !ob_in_cam || !K || !window || !rgb || !depth)
    return fail(SE3TN_E_ARG, "se3tn_render: bad argument");
  if (window[2] <= window[0] || window[3] <= window[1]) return fail(SE3TN_E_ARG, "se3tn_render: empty window");
  RasterArgs a{};
  raster_common(a, c, m, rgb, depth);
  a.rw = RES; a.rh = RES; a.mode = 0;
  if (!vispy_uniforms(a, ob_in_cam, K, window)) return fail(SE3TN_E_ARG, "se3tn_render: singular pose");
  HIPCHK(launch_raster(a, (hipStream_t)stream));
  return SE3TN_OK;
}

// np.round: round half to even
static double round_half_even(double x) { return std::nearbyint(x); }

// Utils.py:302-316 compute_bbox with scale (1000, sy, 1000) -> (left, top, right, bottom) = min / max of the (u, v) corners
// Returns false when a corner is not a finite value inside the int32 range (z == 0, NaN, a pose at the camera centre): the
// float -> int cast would be undefined behaviour.
static bool bbox_window(const double pose[16], const double K[9], double width, double sy, int32_t win[4], int32_t vu[8]) {
  const double x = pose[3] * 1000, y = pose[7] * sy, z = pose[11] * 1000, off = width / 2;
  const double px[4] = {x - off, x - off, x + off, x + off};
  const double py[4] = {y - off, y + off, y - off, y + off};
  int32_t umin = INT32_MAX, umax = INT32_MIN, vmin = INT32_MAX, vmax = INT32_MIN;
  for (int i = 0; i < 4; ++i) {
    const double uf = round_half_even(px[i] * K[0] / z + K[2]), vf = round_half_even(py[i] * K[4] / z + K[5]);
    if (!(std::fabs(uf) < 1.0e9) || !(std::fabs(vf) < 1.0e9)) return false;   // (also false for NaN)
    const int32_t u = (int32_t)uf, v = (int32_t)vf;
    if (vu) { vu[2 * i] = v; vu[2 * i + 1] = u; }
    umin = u < umin ? u : umin; umax = u > umax ? u : umax;
    vmin = v < vmin ? v : vmin; vmax = v > vmax ? v : vmax;
  }
  win[0] = umin; win[1] = vmin; win[2] = umax; win[3] = vmax;
  return true;
}

int se3tn_on_track(se3tn_ctx* c, se3tn_mesh* m, const double prev_pose[16], const double K[9], double object_width_mm,
                   const uint8_t* rgb, const uint16_t* depth, int H, int W, uint8_t* rgbA_dev, uint16_t* depthA_dev,
                   double pose_out[16], float trans_out[3], float rot_out[3], int32_t bbox_vu[8], void* stream) {
  if (!c || c->device < 0 || !m || !prev_pose || !K || !rgb || !depth || H < 1 || W < 1 || !pose_out || !(object_width_mm > 0))
    return fail(SE3TN_E_ARG, "se3tn_on_track: bad argument");
  if (!c->have_norm) return fail(SE3TN_E_STATE, "se3tn_on_track: call se3tn_set_normalization first");
  hipStream_t st = (hipStream_t)stream;
  static const bool trace = std::getenv("SE3TN_TRACK_TRACE") != nullptr;   // developer switch: host-side timeline of the call
  static double acc[6] = {0, 0, 0, 0, 0, 0};
  static int ncall = 0;
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = trace ? now() : 0.0;
  // start-up allocations (first call / larger frame): pinned staging for a whole frame, its device mirror, outputs
  const size_t need = 256 + (size_t)H * W * 5 + 64;
  if (need > c->trk_bytes) {
    DeviceGuard dg(c->device);                     //  staging buffers belong to the context's device, whatever the caller's current one
    if (dg.err != hipSuccess) return hipfail(dg.err, "hipSetDevice");
    HIPCHK(hipDeviceSynchronize());
    if (c->trk_host) HIPCHK(hipHostFree(c->trk_host));
    if (c->trk_dev) HIPCHK(hipFree(c->trk_dev));
    c->trk_host = nullptr; c->trk_dev = nullptr; c->trk_bytes = 0;
    HIPCHK(hipHostMalloc((void**)&c->trk_host, need, hipHostMallocDefault));
    HIPCHK(hipMalloc((void**)&c->trk_dev, need));
    c->trk_bytes = need;
  }
  if (!c->trk_out_host) {
    DeviceGuard dg(c->device);
    if (dg.err != hipSuccess) return hipfail(dg.err, "hipSetDevice");
    // pose | trans | rot | completion word in MAPPED pinned host memory: the tail kernel writes them across the bus itself and the
    // host polls the word -- no device-to-host copy, no stream wake-up on the per-frame critical path
    HIPCHK(hipHostMalloc((void**)&c->trk_out_host, 256, hipHostMallocMapped));
    std::memset(c->trk_out_host, 0, 256);
    HIPCHK(hipHostGetDevicePointer((void**)&c->trk_out_dev, c->trk_out_host, 0));
    HIPCHK(hipMalloc((void**)&c->trk_rgbA, (size_t)RES * RES * 3));
    HIPCHK(hipMalloc((void**)&c->trk_depthA, (size_t)RES * RES * 2));
    HIPCHK(hipStreamCreateWithFlags(&c->trk_copy_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->trk_copy_event, hipEventDisableTiming));
  }
  // predict.py:231-235: bbox of the previous pose (host float64, round half to even) -> crop window of image B;
  // :201-206: the same with the y axis flipped -> the renderer's window
  int32_t winB[4], winA[4], vu[8];
  // an object at or behind the camera plane has no crop window (the reference's compute_bbox divides by z and its crop comes out
  // empty or mirrored; predict.py never gets there): refuse instead of running the frame
  if (!(prev_pose[11] > 0) || !bbox_window(prev_pose, K, object_width_mm, 1000.0, winB, vu) ||
      !bbox_window(prev_pose, K, object_width_mm, -1000.0, winA, nullptr))
    return fail(SE3TN_E_ARG, "se3tn_on_track: pose is not in front of the camera (z <= 0 or not finite)");
  if (winB[2] <= winB[0] || winB[3] <= winB[1]) return fail(SE3TN_E_ARG, "se3tn_on_track: empty crop window (pose behind the camera?)");
  // image A first: its four launches run on the device while the host stages the frame
  uint8_t* rA = rgbA_dev ? rgbA_dev : c->trk_rgbA;
  uint16_t* dA = depthA_dev ? depthA_dev : c->trk_depthA;
  if (int rc = se3tn_render(c, m, prev_pose, K, winA, rA, dA, stream)) return rc;
  const double t1 = trace ? now() : 0.0;
  // only the part of the frame the window covers travels: rows / columns [y0, y1) x [x0, x1) into pinned memory, ONE copy
  // with the pose in front, on a copy stream of its own (beside the rasteriser, not behind it).  crop_bbox's canvas is zero
  // outside the frame (Utils.py:327-342): outside this sub-image too.
  int x0 = winB[0] > 0 ? winB[0] : 0, x1 = winB[2] < W ? winB[2] : W;
  int y0 = winB[1] > 0 ? winB[1] : 0, y1 = winB[3] < H ? winB[3] : H;
  int sw = x1 - x0, sh = y1 - y0;
  uint8_t* hp = c->trk_host;
  std::memcpy(hp, prev_pose, 128);
  uint8_t* h_rgb = hp + 256;
  const bool miss = sw <= 0 || sh <= 0;   // the window misses the frame: a 1 x 1 zero sub-image it does not touch
  if (miss) { sw = sh = 1; x0 = winB[0] - 8; y0 = winB[1] - 8; }
  const size_t d_off = 256 + (((size_t)sw * sh * 3 + 63) & ~(size_t)63);
  if (miss) {
    std::memset(h_rgb, 0, 3);
    std::memset(hp + d_off, 0, 2);
  } else {
    for (int y = 0; y < sh; ++y) {
      std::memcpy(h_rgb + (size_t)y * sw * 3, rgb + ((size_t)(y0 + y) * W + x0) * 3, (size_t)sw * 3);
      std::memcpy(hp + d_off + (size_t)y * sw * 2, depth + (size_t)(y0 + y) * W + x0, (size_t)sw * 2);
    }
  }
  const size_t bytes = d_off + (size_t)sw * sh * 2;
  HIPCHK(hipMemcpyAsync(c->trk_dev, hp, bytes, hipMemcpyHostToDevice, c->trk_copy_stream));
  HIPCHK(hipEventRecord(c->trk_copy_event, c->trk_copy_stream));
  HIPCHK(hipStreamWaitEvent(st, c->trk_copy_event, 0));
  const double t2 = trace ? now() : 0.0;
  // image A and image B in ONE preprocess launch (data_augmentation.py:124-189 for both, with poseA's z)
  CropArgs a;
  std::memcpy(a.mean, c->mean, sizeof(a.mean));
  std::memcpy(a.stdv, c->stdv, sizeof(a.stdv));
  const double z_mm = prev_pose[11] * 1000;
  se3tn_crop& ca = a.c[0];
  ca.rgb = rA; ca.depth = dA; ca.H = RES; ca.W = RES; ca.left = 0; ca.top = 0; ca.right = RES; ca.bottom = RES;
  ca.z_offset_mm = z_mm; ca.stats = 0; ca._pad = 0;
  se3tn_crop& cb = a.c[1];
  cb.rgb = c->trk_dev + 256; cb.depth = (const uint16_t*)(c->trk_dev + d_off); cb.H = sh; cb.W = sw;
  cb.left = winB[0] - x0; cb.top = winB[1] - y0; cb.right = winB[2] - x0; cb.bottom = winB[3] - y0;
  cb.z_offset_mm = z_mm; cb.stats = 1; cb._pad = 0;
  a.n = 2; a.n_first = 1; a.out = c->inA; a.out2 = c->inB; a.padded = 1;
  a.split = c->prec == SE3TN_PREC_F16X3 ? 1 : 0;
  a.overflow = c->overflow; a.offset_rule = c->offset_rule;
  c->in_split[0] = c->in_split[1] = a.split;
  if (const hipError_t e = launch_preprocess(a, st)) { (void)hipStreamSynchronize(c->trk_copy_stream); return hipfail(e, "launch_preprocess"); }
  const double t3 = trace ? now() : 0.0;
  float* trans_d = (float*)(c->trk_out_dev + 128);
  float* rot_d = (float*)(c->trk_out_dev + 144);
  int* flag_d = (int*)(c->trk_out_dev + 192);
  volatile int* flag_h = (volatile int*)(c->trk_out_host + 192);
  const bool poll = !c->use_graphs && !c->prof;   // (a captured graph would replay a stale sequence number)
  const int seq = ++c->tail_seq == 0 ? ++c->tail_seq : c->tail_seq;
  c->tail_flag = poll ? flag_d : nullptr;
  const int rc_inf = se3tn_infer(c, c->inA, c->inB, 1, SE3TN_NHWC, trans_d, rot_d, (const double*)c->trk_dev, (double*)c->trk_out_dev, stream);
  c->tail_flag = nullptr;
  if (rc_inf) {   // the staged frame is still travelling: the next call would overwrite the pinned buffer under the copy
    (void)hipStreamSynchronize(c->trk_copy_stream);
    return rc_inf;
  }
  const double t4 = trace ? now() : 0.0;
  bool seen = false;
  if (poll) {   // 5 ms of polling covers every healthy frame; anything slower (or a fault) falls through to the stream wait
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(5);
    for (;;) {
      for (int it = 0; it < 256 && !seen; ++it) seen = __atomic_load_n((const int*)flag_h, __ATOMIC_ACQUIRE) == seq;
      if (seen || std::chrono::steady_clock::now() > t_end) break;
    }
  }
  if (!seen) HIPCHK(hipStreamSynchronize(st));
  if (trace) {
    const double t5 = now();
    const double d[6] = {t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0};
    for (int i = 0; i < 6; ++i) acc[i] += d[i];
    if (++ncall % 200 == 0) {
      std::fprintf(stderr, "se3tn_on_track host timeline (mean of 200, us): bbox+render enqueue %.1f | stage+H2D enqueue %.1f | preprocess enqueue %.1f | "
                   "infer enqueue %.1f | wait for the tail's word %.1f | total %.1f\n", acc[0] / 200, acc[1] / 200, acc[2] / 200, acc[3] / 200, acc[4] / 200, acc[5] / 200);
      for (double& v : acc) v = 0;
    }
  }
  std::memcpy(pose_out, c->trk_out_host, 128);
  if (trans_out) std::memcpy(trans_out, c->trk_out_host + 128, 12);
  if (rot_out) std::memcpy(rot_out, c->trk_out_host + 144, 12);
  if (bbox_vu) std::memcpy(bbox_vu, vu, sizeof(vu));
  return SE3TN_OK;
}

// ---- n tracks per call -------------------------------------------------------------------------------------------------------
// start-up allocations of se3tn_on_track_batch (first call, a larger n, a larger mesh or frame)
static int reserve_track_batch(se3tn_ctx* c, se3tn_mesh* m, int n, size_t stage_bytes) {
  if (n <= c->tb_cap && n <= m->batch_cap && stage_bytes <= c->tb_stage_bytes) return SE3TN_OK;
  DeviceGuard dg(c->device);
  if (dg.err != hipSuccess) return hipfail(dg.err, "hipSetDevice");
  HIPCHK(hipDeviceSynchronize());
  if (n > c->tb_cap) {
    if (c->tb_inst_host) HIPCHK(hipHostFree(c->tb_inst_host));
    if (c->tb_out_host) HIPCHK(hipHostFree(c->tb_out_host));
    for (void* b : {(void*)c->tb_inst_dev, (void*)c->tb_zbuf, (void*)c->tb_rgbA, (void*)c->tb_depthA, (void*)c->tb_out_dev})
      if (b) HIPCHK(hipFree(b));
    c->tb_inst_host = nullptr; c->tb_out_host = nullptr; c->tb_inst_dev = nullptr; c->tb_zbuf = nullptr; c->tb_rgbA = nullptr;
    c->tb_depthA = nullptr; c->tb_out_dev = nullptr; c->tb_cap = 0;
    const int cap = n > c->max_batch ? n : c->max_batch;
    HIPCHK(hipHostMalloc((void**)&c->tb_inst_host, sizeof(RasterInstance) * cap, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&c->tb_out_host, (size_t)cap * 160, hipHostMallocDefault));
    HIPCHK(hipMalloc((void**)&c->tb_inst_dev, sizeof(RasterInstance) * cap));
    HIPCHK(hipMalloc((void**)&c->tb_zbuf, sizeof(unsigned long long) * RES * RES * cap));
    HIPCHK(hipMalloc((void**)&c->tb_rgbA, (size_t)RES * RES * 3 * cap));
    HIPCHK(hipMalloc((void**)&c->tb_depthA, (size_t)RES * RES * 2 * cap));
    HIPCHK(hipMalloc((void**)&c->tb_out_dev, (size_t)cap * 160));
    c->tb_cap = cap;
  }
  if (n > m->batch_cap) {
    for (void* b : {(void*)m->b_vpost, (void*)m->b_vsnap, (void*)m->b_big, (void*)m->b_clipq})
      if (b) HIPCHK(hipFree(b));
    m->b_vpost = nullptr; m->b_vsnap = nullptr; m->b_big = nullptr; m->b_clipq = nullptr; m->batch_cap = 0;
    const int cap = n > c->max_batch ? n : c->max_batch;
    HIPCHK(hipMalloc((void**)&m->b_vpost, sizeof(float4) * (size_t)m->V * cap));
    HIPCHK(hipMalloc((void**)&m->b_vsnap, sizeof(int4) * (size_t)m->V * cap));
    HIPCHK(hipMalloc((void**)&m->b_big, sizeof(int) * (size_t)(1 + m->F) * cap));
    HIPCHK(hipMalloc((void**)&m->b_clipq, sizeof(int) * (size_t)(1 + m->F) * cap));
    m->batch_cap = cap;
  }
  if (stage_bytes > c->tb_stage_bytes) {
    if (c->tb_stage_host) HIPCHK(hipHostFree(c->tb_stage_host));
    if (c->tb_stage_dev) HIPCHK(hipFree(c->tb_stage_dev));
    c->tb_stage_host = nullptr; c->tb_stage_dev = nullptr; c->tb_stage_bytes = 0;
    HIPCHK(hipHostMalloc((void**)&c->tb_stage_host, stage_bytes, hipHostMallocDefault));
    HIPCHK(hipMalloc((void**)&c->tb_stage_dev, stage_bytes));
    c->tb_stage_bytes = stage_bytes;
  }
  if (!c->trk_copy_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->trk_copy_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->trk_copy_event, hipEventDisableTiming));
  }
  return SE3TN_OK;
}

int se3tn_on_track_batch(se3tn_ctx* c, se3tn_mesh* m, int n, const double* prev_poses, const double K[9], double object_width_mm,
                         const uint8_t* const* rgb, const uint16_t* const* depth, int H, int W, uint8_t* rgbA_dev, uint16_t* depthA_dev,
                         double* pose_out, float* trans_out, float* rot_out, int32_t* bbox_vu, void* stream) {
  if (!c || c->device < 0 || !m || !prev_poses || !K || !rgb || !depth || H < 1 || W < 1 || !pose_out || !(object_width_mm > 0))
    return fail(SE3TN_E_ARG, "se3tn_on_track_batch: bad argument");
  if (n < 1 || n > c->max_batch) return fail(SE3TN_E_ARG, "se3tn_on_track_batch: n outside [1, max_batch]");
  if (!c->have_norm) return fail(SE3TN_E_STATE, "se3tn_on_track_batch: call se3tn_set_normalization first");
  if (stream_is_capturing((hipStream_t)stream)) return fail(SE3TN_E_STATE, "se3tn_on_track_batch: synchronous call, not capturable");
  hipStream_t st = (hipStream_t)stream;
  static const bool trace = std::getenv("SE3TN_TRACK_TRACE") != nullptr;   // developer switch: host-side timeline of the call
  static double tb_acc[6] = {0, 0, 0, 0, 0, 0};
  static int tb_calls = 0;
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = trace ? now() : 0.0;
  // pass 1 (host float64): the windows of every pair (predict.py:231-235 / :201-206) and the size of the staged sub-images
  std::vector<int32_t> win(8 * (size_t)n), vu(8 * (size_t)n);
  std::vector<size_t> off_rgb(n), off_d(n);
  std::vector<int> geo(4 * (size_t)n);        // x0, y0, sw, sh of the staged sub-image
  size_t bytes = (((size_t)n * 128) + 255) & ~(size_t)255;   // the poses in front
  for (int i = 0; i < n; ++i) {
    const double* P = prev_poses + 16 * (size_t)i;
    int32_t* wB = &win[8 * (size_t)i];
    int32_t* wA = wB + 4;
    if (!rgb[i] || !depth[i]) return fail(SE3TN_E_ARG, "se3tn_on_track_batch: null frame pointer");
    if (!(P[11] > 0) || !bbox_window(P, K, object_width_mm, 1000.0, wB, &vu[8 * (size_t)i]) || !bbox_window(P, K, object_width_mm, -1000.0, wA, nullptr))
      return fail(SE3TN_E_ARG, "se3tn_on_track_batch: a pose is not in front of the camera (z <= 0 or not finite)");
    if (wB[2] <= wB[0] || wB[3] <= wB[1]) return fail(SE3TN_E_ARG, "se3tn_on_track_batch: empty crop window");
    int x0 = wB[0] > 0 ? wB[0] : 0, x1 = wB[2] < W ? wB[2] : W;
    int y0 = wB[1] > 0 ? wB[1] : 0, y1 = wB[3] < H ? wB[3] : H;
    int sw = x1 - x0, sh = y1 - y0;
    if (sw <= 0 || sh <= 0) { sw = sh = -1; x0 = wB[0] - 8; y0 = wB[1] - 8; }   // the window misses the frame: a 1 x 1 zero sub-image
    geo[4 * i] = x0; geo[4 * i + 1] = y0; geo[4 * i + 2] = sw; geo[4 * i + 3] = sh;
    const size_t px = sw > 0 ? (size_t)sw * sh : 1;
    off_rgb[i] = bytes; bytes += (px * 3 + 63) & ~(size_t)63;
    off_d[i] = bytes;   bytes += (px * 2 + 63) & ~(size_t)63;
  }
  if (int rc = reserve_track_batch(c, m, n, bytes)) return rc;
  // image A of all n poses: the instance table goes up on the launch stream, then FOUR launches (grid.y = pose)
  uint8_t* rA = rgbA_dev ? rgbA_dev : c->tb_rgbA;
  uint16_t* dA = depthA_dev ? depthA_dev : c->tb_depthA;
  RasterArgs ra{};
  raster_common(ra, c, m, rA, dA);
  ra.rw = RES; ra.rh = RES; ra.mode = 0;
  ra.vpost = m->b_vpost; ra.vsnap = m->b_vsnap; ra.big = m->b_big; ra.clipq = m->b_clipq; ra.zbuf = c->tb_zbuf;
  for (int i = 0; i < n; ++i) {
    RasterArgs one{};
    if (!vispy_uniforms(one, prev_poses + 16 * (size_t)i, K, &win[8 * (size_t)i + 4])) return fail(SE3TN_E_ARG, "se3tn_on_track_batch: singular pose");
    RasterInstance& I = c->tb_inst_host[i];
    std::memcpy(I.PV, one.PV, sizeof(I.PV));
    std::memcpy(I.light, one.light, sizeof(I.light));
    I._pad = 0.f; I.dA = one.dA; I.dB = one.dB;
  }
  HIPCHK(hipMemcpyAsync(c->tb_inst_dev, c->tb_inst_host, sizeof(RasterInstance) * n, hipMemcpyHostToDevice, st));
  ra.inst = c->tb_inst_dev;
  HIPCHK(launch_raster(ra, st, n));
  const double t1 = trace ? now() : 0.0;
  // the frames' windows: staged through pinned memory while the rasteriser runs, ONE copy with the poses in front
  uint8_t* hp = c->tb_stage_host;
  std::memcpy(hp, prev_poses, (size_t)n * 128);
  auto stage_pairs = [&](int i0, int i1) {
    for (int i = i0; i < i1; ++i) {
      const int x0 = geo[4 * i], y0 = geo[4 * i + 1], sw = geo[4 * i + 2], sh = geo[4 * i + 3];
      if (sw < 0) { std::memset(hp + off_rgb[i], 0, 3); std::memset(hp + off_d[i], 0, 2); continue; }
      for (int y = 0; y < sh; ++y) {
        std::memcpy(hp + off_rgb[i] + (size_t)y * sw * 3, rgb[i] + ((size_t)(y0 + y) * W + x0) * 3, (size_t)sw * 3);
        std::memcpy(hp + off_d[i] + (size_t)y * sw * 2, depth[i] + (size_t)(y0 + y) * W + x0, (size_t)sw * 2);
      }
    }
  };
  // above ~8 MB the row-wise gather is worth helper threads (measured, profiles/r06_tracker_batch_staging.txt: 13.5 MB at 64 tracks 0.34-0.43 ms
  // on one core, 0.25 ms on four; 4.5 MB at 21 tracks 0.10 ms on one core, 0.15 ms on four -- the thread start costs more than it
  // saves); SE3TN_STAGE_THREADS = 1 .. 8 (default 4) sets the number of threads incl. the caller's
  static const int stage_threads = [] {
    const char* e = std::getenv("SE3TN_STAGE_THREADS");
    const int v = e ? std::atoi(e) : 4;
    return v < 1 ? 1 : (v > 8 ? 8 : v);
  }();
  const int nth = bytes > ((size_t)8 << 20) ? (stage_threads < n ? stage_threads : n) : 1;
  if (nth > 1) {
    std::vector<std::thread> helpers;
    for (int k = 1; k < nth; ++k) helpers.emplace_back(stage_pairs, (int)((long long)n * k / nth), (int)((long long)n * (k + 1) / nth));
    stage_pairs(0, n / nth);
    for (auto& h : helpers) h.join();
  } else {
    stage_pairs(0, n);
  }
  const double t2 = trace ? now() : 0.0;
  HIPCHK(hipMemcpyAsync(c->tb_stage_dev, hp, bytes, hipMemcpyHostToDevice, c->trk_copy_stream));
  HIPCHK(hipEventRecord(c->trk_copy_event, c->trk_copy_stream));
  HIPCHK(hipStreamWaitEvent(st, c->trk_copy_event, 0));
  // both crops of every pair (data_augmentation.py:124-189 with poseA's z), 64 descriptors per launch
  std::vector<se3tn_crop> crops(2 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    const double z_mm = prev_poses[16 * (size_t)i + 11] * 1000;
    se3tn_crop& ca = crops[i];
    ca.rgb = rA + (size_t)i * RES * RES * 3; ca.depth = dA + (size_t)i * RES * RES; ca.H = RES; ca.W = RES;
    ca.left = 0; ca.top = 0; ca.right = RES; ca.bottom = RES; ca.z_offset_mm = z_mm; ca.stats = 0; ca._pad = 0;
    se3tn_crop& cb = crops[(size_t)n + i];
    const int x0 = geo[4 * i], y0 = geo[4 * i + 1], sw = geo[4 * i + 2], sh = geo[4 * i + 3];
    const int32_t* wB = &win[8 * (size_t)i];
    cb.rgb = c->tb_stage_dev + off_rgb[i]; cb.depth = (const uint16_t*)(c->tb_stage_dev + off_d[i]);
    cb.H = sw < 0 ? 1 : sh; cb.W = sw < 0 ? 1 : sw;
    cb.left = wB[0] - x0; cb.top = wB[1] - y0; cb.right = wB[2] - x0; cb.bottom = wB[3] - y0;
    cb.z_offset_mm = z_mm; cb.stats = 1; cb._pad = 0;
  }
  int rc = se3tn_preprocess(c, crops.data(), n, c->inA, stream);
  if (rc == SE3TN_OK) rc = se3tn_preprocess(c, crops.data() + n, n, c->inB, stream);
  float* trans_d = (float*)(c->tb_out_dev + (size_t)n * 128);
  float* rot_d = trans_d + 3 * (size_t)n;
  if (rc == SE3TN_OK)
    rc = se3tn_infer(c, c->inA, c->inB, n, SE3TN_NHWC, trans_d, rot_d, (const double*)c->tb_stage_dev, (double*)c->tb_out_dev, stream);
  if (rc != SE3TN_OK) { (void)hipStreamSynchronize(c->trk_copy_stream); return rc; }
  const double t3 = trace ? now() : 0.0;
  HIPCHK(hipMemcpyAsync(c->tb_out_host, c->tb_out_dev, (size_t)n * 152, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (trace) {
    const double t4 = now();
    const double d[5] = {t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0};
    for (int i = 0; i < 5; ++i) tb_acc[i] += d[i];
    if (++tb_calls % 50 == 0) {
      std::fprintf(stderr, "se3tn_on_track_batch host timeline (n = %d, mean of 50, us): bboxes + uniforms + render enqueue %.1f | staging %.1f MB %.1f | "
                   "upload + crops + infer enqueue %.1f | wait for the device %.1f | total %.1f\n", n, tb_acc[0] / 50, bytes / 1e6, tb_acc[1] / 50,
                   tb_acc[2] / 50, tb_acc[3] / 50, tb_acc[4] / 50);
      for (double& v : tb_acc) v = 0;
    }
  }
  std::memcpy(pose_out, c->tb_out_host, (size_t)n * 128);
  if (trans_out) std::memcpy(trans_out, c->tb_out_host + (size_t)n * 128, (size_t)n * 12);
  if (rot_out) std::memcpy(rot_out, c->tb_out_host + (size_t)n * 140, (size_t)n * 12);
  if (bbox_vu) std::memcpy(bbox_vu, vu.data(), sizeof(int32_t) * 8 * (size_t)n);
  return SE3TN_OK;
}

int se3tn_fill_depth(se3tn_ctx* c, const uint16_t* depth_mm, int H, int W, double max_depth_m, int extrapolate, int blur,
                     uint16_t* out_mm, float* out_m, void* stream) {
  if (!c || c->device < 0 || !depth_mm || H < 1 || W < 1 || (!out_mm && !out_m) || blur < 0 || blur > 2)
    return fail(SE3TN_E_ARG, "se3tn_fill_depth: bad argument");
  const size_t px = (size_t)H * W;
  if (px > c->fd_pixels) {   // not reserved (se3tn_reserve): grow now -- impossible inside a stream capture
    if (stream_is_capturing((hipStream_t)stream))
      return fail(SE3TN_E_STATE, "se3tn_fill_depth: scratch too small for this frame inside a stream capture: call se3tn_reserve(ctx, H, W) first");
    if (int rc = reserve_fill_depth(c, px)) return rc;
  }
  FillDepthArgs a{};
  a.depth_mm = depth_mm; a.H = H; a.W = W; a.max_depth = max_depth_m; a.extrapolate = extrapolate; a.blur = blur;
  a.sigma_color = 1.5; a.sigma_space = 2.0;   // Utils.py:505 cv2.bilateralFilter(depth, 5, 1.5, 2.0)
  a.buf0 = c->fd_buf; a.buf1 = c->fd_buf + px; a.buf2 = c->fd_buf + 2 * px;
  a.minmax = reinterpret_cast<unsigned*>(c->fd_buf + 3 * px);
  a.lut = c->fd_buf + 3 * px + 2;
  a.out_mm = out_mm; a.out_m = out_m;
  HIPCHK(launch_fill_depth(a, (hipStream_t)stream));
  return SE3TN_OK;
}

int se3tn_render_frame(se3tn_ctx* c, se3tn_mesh* m, const double ob_in_cam[16], const double K[9], int W, int H, uint8_t* rgb,
                       uint16_t* depth, void* stream) {
  if (!c || c->device < 0 || !m || !ob_in_cam || !K || !rgb || !depth || W < 1 || H < 1)
    return fail(SE3TN_E_ARG, "se3tn_render_frame: bad argument");
  const size_t px = (size_t)W * H;
  if (px > c->zbuf_px) {   // not reserved (se3tn_reserve): grow now -- impossible inside a stream capture
    if (stream_is_capturing((hipStream_t)stream))
      return fail(SE3TN_E_STATE, "se3tn_render_frame: z-buffer too small for this frame inside a stream capture: call se3tn_reserve(ctx, H, W) first");
    if (int rc = reserve_zbuf(c, px)) return rc;
  }
  if (H > 2048) return fail(SE3TN_E_ARG, "se3tn_render_frame: frames of more than 2048 rows are not supported");
  RasterArgs a{};
  raster_common(a, c, m, rgb, depth);
  a.rw = W; a.rh = H; a.mode = 1;
  a.uv = m->uv; a.tex = m->tex; a.tw = m->tw; a.th = m->th; a.tlevels = m->tlevels;
  for (int i = 0; i < 16; ++i) a.tex_off[i] = m->tex_off[i];
  a.kd[0] = m->kd[0]; a.kd[1] = m->kd[1]; a.kd[2] = m->kd[2];
  // pyrender's IntrinsicsCamera projection at W x H (GL clip space, y up) times cvcam_in_glcam . ob_in_cam, formed in float64 and
  // uploaded as ONE float32 matrix; the vertex shader multiplies it with the position
  const double n = R_NEAR_D, f = R_FAR_D;
  double P[4][4] = {{2.0 * K[0] / W, 0.0, 1.0 - 2.0 * K[2] / W, 0.0},
                    {0.0, 2.0 * K[4] / H, 2.0 * K[5] / H - 1.0, 0.0},
                    {0.0, 0.0, (f + n) / (n - f), 2.0 * f * n / (n - f)},
                    {0.0, 0.0, -1.0, 0.0}};
  const double sgn[4] = {1.0, -1.0, -1.0, 1.0};
  for (int r = 0; r < 4; ++r)
    for (int q = 0; q < 4; ++q) {
      double acc = 0.0;
      for (int k = 0; k < 4; ++k) acc += P[r][k] * (sgn[k] * ob_in_cam[4 * k + q]);
      a.PV[4 * r + q] = (float)acc;
    }
  HIPCHK(launch_raster(a, (hipStream_t)stream));
  return SE3TN_OK;
}

int se3tn_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
  if (!dst || !src) return fail(SE3TN_E_ARG, "se3tn_memcpy_d2d: null pointer");
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return SE3TN_OK;
}

int se3tn_profile_enable(se3tn_ctx* c, int slots) {
  if (!c || c->device < 0 || slots < 0 || slots > MAX_SLOTS) return fail(SE3TN_E_ARG, "se3tn_profile_enable: bad argument");
  for (int s = c->slots; s < slots; ++s) {
    for (auto& e : c->evs[s]) HIPCHK(hipEventCreate(&e));
    c->slot_launches[s] = 0;
    c->slots = s + 1;
  }
  c->prof = slots > 0;
  c->infer_count = 0;
  for (int s = 0; s < c->slots; ++s) c->slot_launches[s] = 0;
  return SE3TN_OK;
}

int se3tn_profile_read(se3tn_ctx* c, int slot, float* conv_ms, int* conv_launches, float* total_ms) {
  if (!c || slot < 0 || slot >= c->slots || c->slot_launches[slot] == 0)
    return fail(SE3TN_E_STATE, "se3tn_profile_read: nothing recorded in this slot");
  const int nl = c->slot_launches[slot];
  hipEvent_t* ev = c->evs[slot];
  HIPCHK(hipEventSynchronize(ev[nl]));
  float conv = 0.f, tot = 0.f;
  int nconv = 0;
  for (int i = 0; i < nl; ++i) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
    tot += ms;
    if (c->is_conv[i]) { conv += ms; ++nconv; }
  }
  if (conv_ms) *conv_ms = conv;
  if (conv_launches) *conv_launches = nconv;
  if (total_ms) *total_ms = tot;
  return SE3TN_OK;
}

int se3tn_profile_launches(se3tn_ctx* c, int slot, int cap, const char** names, float* ms) {
  if (!c || slot < 0 || slot >= c->slots || c->slot_launches[slot] == 0) {
    fail(SE3TN_E_STATE, "se3tn_profile_launches: nothing recorded in this slot");
    return 0;
  }
  const int nl = c->slot_launches[slot];
  hipEvent_t* ev = c->evs[slot];
  if (hipEventSynchronize(ev[nl]) != hipSuccess) return 0;
  int k = nl < cap ? nl : cap;
  for (int i = 0; i < k; ++i) {
    if (names) names[i] = c->names[i];
    if (ms && hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]) != hipSuccess) return 0;
  }
  return k;
}

// ---- host-side float64 pieces -----------------------------------------------------------------

int se3tn_compute_bbox(const double pose[16], const double K[9], double width, int32_t out_vu[8]) {
  if (!pose || !K || !out_vu) return fail(SE3TN_E_ARG, "se3tn_compute_bbox: bad argument");
  // Utils.py:302-316 with scale=(1000,1000,1000)
  const double x = pose[3] * 1000, y = pose[7] * 1000, z = pose[11] * 1000, off = width / 2;
  const double px[4] = {x - off, x - off, x + off, x + off};
  const double py[4] = {y - off, y + off, y - off, y + off};
  for (int i = 0; i < 4; ++i) {
    const double u = px[i] * K[0] / z + K[2];
    const double v = py[i] * K[4] / z + K[5];
    out_vu[2 * i + 0] = (int32_t)round_half_even(v);
    out_vu[2 * i + 1] = (int32_t)round_half_even(u);
  }
  return SE3TN_OK;
}

int se3tn_pose_update_host(const double A[16], const float trans[3], const float rot[3], double tn, double rn,
                           double B[16]) {
  if (!A || !trans || !rot || !B) return fail(SE3TN_E_ARG, "se3tn_pose_update_host: bad argument");
  const float tf = (float)tn, rf = (float)rn;
  const volatile float t0 = trans[0] * tf, t1 = trans[1] * tf, t2 = trans[2] * tf;
  const volatile float r0 = rot[0] * rf, r1 = rot[1] * rf, r2 = rot[2] * rf;
  const double rx = r0, ry = r1, rz = r2;
  double R[9];
  const double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
  if (theta < 2.220446049250313e-16) {
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  } else {
    const double c = std::cos(theta), s = std::sin(theta), c1 = 1.0 - c, it = 1.0 / theta;
    const double x = rx * it, y = ry * it, z = rz * it;
    const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    const double rxm[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int i = 0; i < 9; ++i) R[i] = c * ((i % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[i] + s * rxm[i];
  }
  for (int i = 0; i < 9; ++i) R[i] = (double)(float)R[i];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += R[i * 3 + k] * A[k * 4 + j];
      B[i * 4 + j] = acc;
    }
  B[3] = (double)t0 + A[3];
  B[7] = (double)t1 + A[7];
  B[11] = (double)t2 + A[11];
  B[12] = B[13] = B[14] = 0.0;
  B[15] = 1.0;
  return SE3TN_OK;
}

}  // extern "C"
