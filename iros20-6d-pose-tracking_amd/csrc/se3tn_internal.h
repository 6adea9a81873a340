// Internal definitions shared by the host-side weight packer and the gfx950 kernels.
// Network geometry follows se3_tracknet.py:52-112 of the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/se3tracknet.h"

namespace se3tn {

constexpr int RES = SE3TN_RES;  // 176
constexpr int S1 = 88;          // after the 7x7 s2 stems
constexpr int S2 = 44;          // after maxpool 3x3 s2
constexpr int S3 = 22;          // after convAB1 (s2)
constexpr int S4 = 11;          // after {trans,rot}_conv1 (s2)

// ---------------------------------------------------------------------------------------------
// Packed weight blob (float32 words).  One contiguous allocation so that multi-GPU start-up is a
// single RCCL broadcast.  3x3 convolutions are stored as the MFMA "A" operand panels
//   [chunk = Cin/32][tap = r*3+s][Cout][32]        (BN folded, see weights.cpp)
// i.e. for every K-step (chunk,tap) a [Cout][32] row-major tile whose rows are the K-contiguous
// runs a lane reads with one ds_read_b128.  Stems: [Cout = 64][204], k = pair*8 + half*4 + c
// (tap pairing: weights.cpp pack_stem / stem7x7_mfma.hip).
// ---------------------------------------------------------------------------------------------
struct Conv3 {
  int cin, cout, groups;  // weights of `groups` independent convs stored back to back
};
constexpr size_t conv3_words(int cin, int cout) { return (size_t)9 * cin * cout; }

enum ConvId {
  L64_1 = 0,  // convA2.conv1 | convB2.conv1     (2 groups, 64->64)
  L64_2,      // convA2.conv2 | convB2.conv2     (2 groups)
  L64_3,      // convB3.conv1
  L64_4,      // convB3.conv2
  LAB1,       // convAB1 128->256 s2
  LAB2_1,     // convAB2.conv1 256->256
  LAB2_2,     // convAB2.conv2
  LH1,        // trans_conv1 | rot_conv1 fused along Cout: 256->1024 s2
  LH2_1,      // trans_conv2.conv1 | rot_conv2.conv1   (2 groups, 512->512)
  LH2_2,      // trans_conv2.conv2 | rot_conv2.conv2   (2 groups)
  NUM_CONV3
};

struct BlobLayout {
  // offsets in float32 words
  size_t stem_w;   // [2 branches][64][204]
  size_t stem_b;   // [2][64]
  size_t conv_w[NUM_CONV3];
  size_t conv_b[NUM_CONV3];
  size_t fc_w;     // [2 heads][3][512]
  size_t fc_b;     // [2][4] (3 used)
  size_t total;    // words, incl. the 64-word header
};

constexpr int HEADER_WORDS = 64;
constexpr uint32_t BLOB_MAGIC = 0x53453354u;  // 'SE3T'
constexpr uint32_t BLOB_VERSION = 7;  // 7: float32 panels only (54 MB); the f16x3 split panels are derived on the device

// f16x3 mode: the same panels as "split rows" (32 x f16 hi | 32 x f16 lo of w * 2^k per cout) and the per-cout 2^-k.
// NOT part of the blob (since v7): a context-owned device buffer derived from the bound blob by launch_split_weights when
// SE3TN_PREC_F16X3 is first selected -- like the Winograd planes, so the RCCL broadcast carries the float32 panels only.
struct SplitLayout {
  size_t stem_ws;  // [2][64][204] words, each 16-byte (pair, half) entry = 4 f16 hi | 4 f16 lo
  size_t stem_sc;  // [2][64] per-cout 2^-k
  size_t conv_ws[NUM_CONV3];
  size_t conv_sc[NUM_CONV3];
  size_t total;    // words
};

inline const Conv3* conv_specs() {
  static const Conv3 s[NUM_CONV3] = {
      {64, 64, 2},   {64, 64, 2},   {64, 64, 1},   {64, 64, 1},  {128, 256, 1},
      {256, 256, 1}, {256, 256, 1}, {256, 1024, 1}, {512, 512, 2}, {512, 512, 2}};
  return s;
}

inline BlobLayout blob_layout() {
  BlobLayout L{};
  size_t o = HEADER_WORDS;
  L.stem_w = o; o += (size_t)2 * 64 * 204;
  L.stem_b = o; o += 2 * 64;
  const Conv3* s = conv_specs();
  for (int i = 0; i < NUM_CONV3; ++i) {
    L.conv_w[i] = o; o += conv3_words(s[i].cin, s[i].cout) * s[i].groups;
    L.conv_b[i] = o; o += (size_t)s[i].cout * s[i].groups;
  }
  L.fc_w = o; o += 2 * 3 * 512;
  L.fc_b = o; o += 2 * 4;
  o = (o + 63) & ~(size_t)63;
  L.total = o;
  return L;
}

inline SplitLayout split_layout() {
  SplitLayout L{};
  size_t o = 0;
  L.stem_ws = o; o += (size_t)2 * 64 * 204;
  L.stem_sc = o; o += 2 * 64;
  const Conv3* s = conv_specs();
  for (int i = 0; i < NUM_CONV3; ++i) {
    L.conv_ws[i] = o; o += conv3_words(s[i].cin, s[i].cout) * s[i].groups;
    L.conv_sc[i] = o; o += (size_t)s[i].cout * s[i].groups;
  }
  o = (o + 63) & ~(size_t)63;
  L.total = o;
  return L;
}

// Exponent k of the exact per-cout power-of-two weight scaling of the f16x3 mode: the largest |w| of a cout row lands in
// [2^10, 2^11) (k = floor(10 - log2(mx)), evaluated on the float's own exponent so that host and device agree bit for bit).
__host__ __device__ inline int split_exponent(float mx) {
  if (!(mx > 0.f)) return 0;
  int e;
  const float m = frexpf(mx, &e);  // mx = m 2^e, m in [0.5, 1)
  int k = (m == 0.5f) ? 11 - e : 10 - e;
  if (k > 40) k = 40;
  if (k < -20) k = -20;
  return k;
}

// ---------------------------------------------------------------------------------------------
// kernel argument blocks
// ---------------------------------------------------------------------------------------------
// All conv activations are NHWC with a one-pixel zero border: a tensor of interior size H x W is
// stored as [n][H+2][W+2][ld] (+ PAD_SLACK_PX pixels of slack after the last image, the slab DMA
// rounds its run up to 8 pixels).
constexpr int PAD_SLACK_PX = 8;
// The network input (one 16-byte RGBD pixel) is stored with a 3-pixel zero border: [n,182,182,4]
// (+ IN_SLACK_ROWS rows of slack: the stem's slab DMA always fetches 13 rows).
constexpr int IN_PAD = 3;
constexpr int IN_P = RES + 2 * IN_PAD;  // 182
constexpr int IN_SLACK_ROWS = 6;
constexpr int SE3TN_SPLITK_MAX_TILES = 1024;   // output tiles (m tile x group x n tile) of a split-K launch with the fused reduction
struct ConvArgs {
  const float* in;    // padded NHWC, `in_ld` floats per pixel; channel offset already applied
  const float* w;     // packed panels of group 0
  const float* bias;  // folded bias of group 0
  const float* res;   // residual (geometry of out) or nullptr
  float* out;         // padded NHWC
  int in_ld, res_ld, out_ld;
  int H, W, Ho, Wo;   // input / output INTERIOR spatial size
  int M;              // n * Ho * Wo  (GEMM rows = output pixels)
  int tiles_n;        // Cout / BN
  int groups;         // independent convolutions in this launch
  int in_gs, res_gs, out_gs, bias_gs;  // per-group strides (floats)
  long long w_gs;
  // small-batch split-K path (conv3x3_splitk_kernel): partial-sum workspace, 0 slices = not used
  float* part;
  size_t part_bytes;
  int slices;
  // fused reduction of the split-K path: sem[tile] counts the slices of an output tile that have stored their partial sums,
  // sem[SE3TN_SPLITK_MAX_TILES + tile] the reduce workgroups that have seen it complete (both are zero between launches).
  // nullptr = separate conv_reduce_kernel launch.  epi / outf / resf: the epilogue the in-kernel reduction applies
  int* sem;
  int epi, outf, resf, rpt;   // rpt: reduce workgroups per output tile
  int skip_reduce;            // conv_slices_small only: the caller consumes the partial sums itself (tail_parts_kernel): no conv_reduce launch
  int small_ok;               // the batch-1-5 kernel family (conv64_small, conv_slices_small) may be used (SE3TN_SMALL_KERNELS=0 switches them off)
  // f16x3 mode: per-cout power-of-two weight scale (acc * wscale = true sum), overflow flag,
  // fast = 0 (f32 everywhere) | 1 | 2 (see launch_conv3x3)
  const float* wscale;
  int* overflow;
  int fast;
};

// Winograd F(m x m,3x3) path of the stride-1 256/512-channel convs at large batch (wino_mfma.hip)
struct WinoArgs {
  const float* in;    // padded NHWC input, channel offset of group 0 applied
  const float* U;     // transformed weights of group 0: [chunk][nf][Cout][32]
  const float* bias;
  const float* res;   // residual or nullptr
  float* out;         // padded NHWC
  float* V;           // workspace [groups][nf][T][C]     (transformed input tiles)
  float* Mw;          // workspace [groups][nf][T][Cout]  (per-frequency products)
  int in_ld, res_ld, out_ld;
  int H, W;           // interior size (input == output, stride 1)
  int m, nf;          // output tile edge (2 | 4), nf = (m+2)^2 frequencies
  int th, tw;         // m x m output tiles per image: ceil(H/m), ceil(W/m)
  int n, T;           // images, T = n * th * tw
  int C, Cout, groups;
  int in_gs, res_gs, out_gs, bias_gs;
  long long u_gs;     // floats per group of U
  int num_cus;        // compute units of the device (0: 256); sizes the persistent GEMM grid
  int gemmp;          // persistent 128 x 256 GEMM: -1 = when its tiles fill every CU twice, 0 = never, 1 = whenever the shape allows
  // f16x3 mode of the fused F(4x4) blocks (split != 0): in / res / out are split-row tensors, V is written as split rows, U points
  // at the split planes and uscale[g][f][Cout] holds their exact power-of-two scales (undone in the GEMM epilogue, M is float32)
  int split;
  const float* uscale;
  int* overflow;
};

// outputs of the fused out-transform + avg-pool + FC + tanh (wino_tail_kernel)
struct TailArgs {
  const float* fc_w;  // [2 heads][3][512]
  const float* fc_b;  // [2][4]
  float* logits;      // [n,6] pre-tanh
  float* fcpart;      // [n][2 heads][8 channel slices][3] partial FC dot products
  float* trans;       // [n,3] or nullptr
  float* rot;         // [n,3] or nullptr
  const double* poseA;  // [n,16] or nullptr
  double* poseB;
  double tn, rn;
};

struct CropArgs {  // one launch handles up to MAX crops
  static constexpr int MAX = 64;  // 64 x 56 B + constants = 3.75 KB of kernel arguments (limit 4 KB, asserted below)
  se3tn_crop c[MAX];
  double mean[8], stdv[8];
  float* out;   // plain [n,176,176,4], or (padded != 0) the interior of [n,182,182,4]
  float* out2;  // crops n_first.. go here, numbered from 0 (image A and image B of a frame in ONE launch: se3tn_on_track); else unused
  int n_first;  // (n when out2 is unused)
  int n;
  int padded;
  int split;    // f16x3 mode: a pixel is stored as 4 x f16 hi | 4 x f16 lo (same 16 bytes)
  int offset_rule;  // SE3TN_OFFSET_RULE_*: how `depth -= z` rounds (NumPy 1.x: one float32 operation; NumPy 2: float64, rounded once)
  int* overflow;
};

static_assert(sizeof(CropArgs) <= 4096, "CropArgs travels as kernel arguments: HIP's limit is 4 KB");

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies per DEVICE: a launcher remembers which device
// ordinals it has already raised the limit on (several contexts on different GPUs in one process).
struct PerDeviceOnce {
  bool done[64] = {};
  // returns the flag of the calling thread's current device (nullptr if the ordinal cannot be read)
  bool* current() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    return &done[dev];
  }
};

// launchers (defined in the .hip files)
// f16x3 split panels + per-cout scales of every conv / both stems from the bound float32 blob (kernels_misc.hip)
hipError_t launch_split_weights(const float* blob, const BlobLayout& L, float* split, const SplitLayout& S, hipStream_t st);
// split rows + per-(frequency, cout) scales of one group's Winograd planes U [chunk][nf][cout][32] (f16x3 Winograd blocks)
hipError_t launch_split_wino_u(const float* U, float* Us, float* scale, int cin, int cout, int nf, hipStream_t st);
// NCHW [n,4,176,176] (nchw != 0) or plain NHWC [n,176,176,4] -> interior of the padded [n,182,182,4]
hipError_t launch_to_padded_input(const float* in, float* out, int n, int nchw, int split, int* overflow,
                                  hipStream_t st);
hipError_t launch_preprocess(const CropArgs& a, hipStream_t st);
hipError_t launch_crop_raw(const se3tn_crop& c, uint8_t* rgb_out, uint16_t* depth_out, hipStream_t st);
// wscale != nullptr selects the f16x3 stem (split pixels, split weights w)
hipError_t launch_stem(const float* inA, const float* inB, const float* w, const float* bias,
                       const float* wscale, float* out, int n, hipStream_t st);
// split_out != 0: write split rows (f16 hi | f16 lo) for the f16x3 mode
hipError_t launch_maxpool(const float* in, float* out, int n, int split_out, hipStream_t st);
// conv3x3: cin/cout/stride select the instantiation; epi: 0 bias+relu, 1 bias+res+relu, 2 bias+selu
hipError_t launch_conv3x3(const ConvArgs& a, int cin, int cout, int stride, int epi, hipStream_t st);
// Winograd F(m x m,3x3): U = G g G^T (float64, rounded once) from the packed direct weights
// [chunk][9][cout][32] -> [chunk][(m+2)^2][cout][32]; launch_wino_conv = input transform + nf x groups
// batched MFMA GEMMs + output transform with the conv epilogue (epi 0 | 1)
hipError_t launch_wino_weights(const float* packed, float* U, int cin, int cout, int m, hipStream_t st);
hipError_t launch_wino_conv(const WinoArgs& a, int epi, hipStream_t st);
// fused Winograd F(2x2,3x3) of a 64 -> 64 channel convolution on the 44 x 44 maps (wino64_fused.hip); U: F(2x2) planes of launch_wino_weights
hipError_t launch_wino64(const float* in, int in_ld, int in_gs, const float* U, long long u_gs, const float* bias, int bias_gs,
                         const float* res, int res_ld, int res_gs, float* out, int out_ld, int out_gs, int n, int groups, int epi,
                         int variant, hipStream_t st);
// a whole residual block on the Winograd F(4x4) path with the fused mid / tail transforms (wino_mfma.hip);
// mark_after_mid: optional profiling hook called between conv1 and conv2 (returns non-zero on error)
hipError_t launch_wino_block(const WinoArgs& c1, const float* U2, const float* uscale2, const float* bias2, float* out2, int keep_mid,
                             float* keep_out2, const TailArgs* tl, hipStream_t st, int mark_after_mid(void*), void* mark_ctx);
// stem + max-pool of both branches in one launch at batch 1-2 (stem_pool_small.hip)
#ifndef SE3TN_STEM_SMALL_MAX_N
#define SE3TN_STEM_SMALL_MAX_N 5
#endif
hipError_t launch_stem_pool_small(const float* inA, const float* inB, const float* w, const float* bias, float* pool, int n, hipStream_t st);
// the 64 -> 64 trunk convs at batch 1-5 without a K split (conv64_small.hip)
hipError_t launch_conv64_small(const ConvArgs& a, int n, int epi, hipStream_t st);
// the 128 .. 512-channel convs at batch 1-5: 128 pixels x 32 couts x one channel slice with all nine taps per workgroup
// (conv_slices_small.hip); partial sums for conv_reduce_kernel
#ifndef SE3TN_SLICES_SMALL_MAX_N
#define SE3TN_SLICES_SMALL_MAX_N 5   // up to this many pairs the 128 .. 512-channel convs take conv_slices_small_kernel (0: never)
#endif
int conv_slices_small_count(int cin, int stride, int H);
hipError_t launch_conv_slices_small(const ConvArgs& a, int cin, int stride, hipStream_t st);
hipError_t launch_tail(const float* head, const float* fc_w, const float* fc_b, float* logits,
                       float* trans, float* rot, const double* poseA, double* poseB, double tn,
                       double rn, int n, hipStream_t st, float* fcpart, int* arrive, int* done_flag = nullptr, int done_seq = 0);
// the same fed by the 8 partial-sum slices of the last head conv at batch 1-5 (no conv_reduce launch in between); fcpart [n][64][3]
hipError_t launch_tail_parts(const float* part, int slices, size_t slice_stride, int M, const float* bias, const float* res, int res_ld,
                             const float* fc_w, const float* fc_b, float* logits, float* trans, float* rot, const double* poseA,
                             double* poseB, double tn, double rn, int n, hipStream_t st, float* fcpart, int* arrive, int* done_flag = nullptr,
                             int done_seq = 0, int ch = 16);
// padded [n,h+2,w+2,c] NHWC interior -> [n,c,h,w]
// split != 0: the source holds split rows (32 f16 hi | 32 f16 lo per 32-channel chunk)
hipError_t launch_padded_nhwc_to_nchw(const float* in, float* out, int n, int h, int w, int c, int split,
                                      hipStream_t st);

// rasteriser (raster.hip)
constexpr double R_NEAR_D = 0.1, R_FAR_D = 2.0;   // vispy_renderer.py:139-140, offscreen_renderer.py znear / zfar
// per-instance uniforms of a BATCHED rasteriser launch (grid.y = instance: n poses of one mesh in four launches, se3tn_on_track_batch)
struct RasterInstance {
  float PV[16];
  float light[3];
  float _pad;
  double dA, dB;
};
struct RasterArgs {
  const RasterInstance* inst;  // nullptr: one instance, uniforms below.  Otherwise instance b = blockIdx.y takes PV / light / dA / dB from
                               // inst[b] and its scratch / outputs at b x (V | 1 + F | rw rh) elements behind the pointers below
  const float* verts;    // [V,3] object space
  const float* normals;  // [V,3]
  const float* colors;   // [V,3] in [0,1]
  const int* faces;      // [F,3]
  float4* vpost;         // [V] scratch: post-transform clip position (x, y, (z + w) / 2, w)
  int4* vsnap;           // [V] scratch: window X, Y in 1 / 2^sub_bits pixel, bits of z / w, bits of 1 / w
  unsigned long long* zbuf;  // [rw*rh] scratch: (sortable z << 32 | triangle)
  int* big;              // [1 + F] queue of triangles with large bounding boxes (big[0] = count)
  int* clipq;            // [1 + F] queue of triangles that cross the frustum
  uint8_t* rgb;          // out [rh,rw,3]
  uint16_t* depth;       // out [rh,rw] millimetres
  float PV[16];          // the shader's proj * view (mode 1: u_pv), row-major
  float light[3];        // light_direction in object space
  double dA, dB;         // mode 0: projection_matrix[2,2], [3,2] as vispy_renderer.py:163-164 reads them (float64)
  int numpy_rule;        // SE3TN_OFFSET_RULE_*: the NumPy generation's scalar casting in vispy_renderer.py:165-169
  int sub_bits;          // sub-pixel bits of the window coordinates (4 | 8)
  int V, F;
  int rw, rh;            // output resolution (176 x 176 for the Vispy-style window, the camera frame in pyrender mode)
  int mode;              // 0: VispyRenderer (Lambert shader, window crop)   1: pyrender (ambient only, full frame)
  const float* uv;       // mode 1: [V,2] texture coordinates or nullptr
  const uint8_t* tex;    // mode 1: RGB uint8 mip pyramid or nullptr (then the vertex colours are the base colour)
  int tw, th, tlevels;
  unsigned tex_off[16];  // byte offset of every mip level
  float kd[3];           // base colour factor (mtl Kd)
};
hipError_t launch_raster(const RasterArgs& a, hipStream_t st, int instances = 1);

// depth hole filling (depth_fill.hip)
struct FillDepthArgs {
  const uint16_t* depth_mm;  // [H,W]
  int H, W;
  double max_depth;          // metres
  int extrapolate;
  int blur;                  // SE3TN_BLUR_*
  double sigma_color, sigma_space;
  float *buf0, *buf1, *buf2; // [H,W] float32 scratch
  unsigned* minmax;          // [2]
  float* lut;                // [4098]
  uint16_t* out_mm;          // [H,W] or nullptr
  float* out_m;              // [H,W] or nullptr
};
hipError_t launch_fill_depth(const FillDepthArgs& a, hipStream_t st);

// host-side packer (weights.cpp)
struct HostTensor {
  const float* data;
  int64_t shape[4];
  int ndim;
};

}  // namespace se3tn
