// 3x3 convolution (pad 1, stride 1|2) + folded-BN bias + {ReLU | residual+ReLU | SELU} as an
// im2col-free implicit GEMM on the gfx950 exact-f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces every nn.Conv2d(k=3)+BatchNorm2d(+SELU/ReLU/+identity) of Se3TrackNet
// (se3_tracknet.py:59-76 via network_modules.py:59-66 ConvBNReLU, :86-120 ResnetBasicBlock).
//
// GEMM view: rows = output pixels (n,ho,wo) flattened over the batch, cols = Cout,
// K = (chunk of 32 input channels) x (tap r,s).  K is walked chunk-major / tap-minor so the 9
// shifted reads of one 32-channel slab of the input tile happen in 9 consecutive K-steps and are
// served by L1/L2 (no im2col buffer ever exists).
//
// MFMA operand roles are SWAPPED w.r.t. the textbook GEMM: A-operand = weight rows
// (i = cout), B-operand = pixel rows (j = pixel).  The accumulator then holds, per lane, ONE pixel
// and 4 runs of 4 consecutive couts  ->  the epilogue is float4 bias / residual / store traffic
// (16 x dwordx4 per wave instead of 64 x dword).
//
// Tile: BM = WM*PT*32 pixels x BN = WN*CT*32 couts per 256-thread workgroup (4 waves, one per SIMD,
// 2 workgroups per CU).  Per K-step: [BM][32] pixel tile + [BN][32] weight tile staged through
// registers into double-buffered LDS (rows padded to 36 floats: conflict-free ds_read_b128),
// one barrier per K-step, next tile's global loads in flight under the current tile's MFMAs.
#include "se3tn_internal.h"

#ifndef SE3TN_ABLATE
#define SE3TN_ABLATE 0  // timing ablations only (wrong results): 1 = no DMA, 2 = no barrier
#endif

namespace se3tn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LDK = 36;  // floats per LDS row: 32 + 4 pad -> row stride 144 B (9 x 16 B, odd)

constexpr float SELU_ALPHA = 1.6732632423543772848170429916717f;
constexpr float SELU_SCALE = 1.0507009873554804934193349852946f;

__device__ __forceinline__ float selu_f(float v) {
  return v > 0.f ? SELU_SCALE * v : (SELU_SCALE * SELU_ALPHA) * expm1f(v);
}

// EPI: 0 = bias+ReLU, 1 = bias+residual+ReLU, 2 = bias+SELU
template <int EPI>
__device__ __forceinline__ float4 apply_epilogue(float4 v, const float4 b, const float* res) {
  v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  if (EPI == 1) {
    const float4 r = *reinterpret_cast<const float4*>(res);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  if (EPI == 2) {
    v.x = selu_f(v.x); v.y = selu_f(v.y); v.z = selu_f(v.z); v.w = selu_f(v.w);
  } else {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  }
  return v;
}

template <int CIN, int STRIDE, int WM, int WN, int PT, int CT, int EPI>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(const ConvArgs a) {
  constexpr int BM = WM * PT * 32, BN = WN * CT * 32;
  constexpr int NCH = CIN / 32, KT = NCH * 9;
  constexpr int PR = BM / 32, WR = BN / 32;  // rows staged per thread (pixels / weights)
  constexpr int BUF = (BM + BN) * LDK;       // floats per LDS buffer
  static_assert(WM * WN == 4, "4 waves per workgroup");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, hh = lane >> 5;

  // workgroup -> (pixel tile, weight panel).  Consecutive workgroup ids land on consecutive XCDs
  // (id % 8); panels = groups*tiles_n divides 8, so every XCD's L2 only ever sees panels
  // congruent to its id: a weight panel (<= 2.4 MB) stays resident in that XCD's 4 MB L2.
  const int panels = a.groups * a.tiles_n;
  const int p = blockIdx.x % panels, mt = blockIdx.x / panels;
  const int g = p / a.tiles_n, nt = p % a.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;

  const float* __restrict__ in = a.in + (size_t)g * a.in_gs;
  const float* __restrict__ wgt = a.w + (size_t)g * a.w_gs;

  // ---- per-thread staging geometry: thread t owns 16-byte column c4 of rows r0 + 32*j ----------
  const int c4 = tid & 7, r0 = tid >> 3;
  int rowoff[PR];
  unsigned rowmask[PR];
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int j = 0; j < PR; ++j) {
    const int m = m0 + r0 + 32 * j;
    unsigned mask = 0;
    int off = 0;
    if (m < a.M) {
      const int n = m / HoWo, rem = m - n * HoWo;
      const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
      const int hi0 = ho * STRIDE - 1, wi0 = wo * STRIDE - 1;
      off = ((n * a.H + hi0) * a.W + wi0) * a.in_ld + c4 * 4;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          if ((unsigned)(hi0 + r) < (unsigned)a.H && (unsigned)(wi0 + s) < (unsigned)a.W)
            mask |= 1u << (r * 3 + s);
    }
    rowoff[j] = off;
    rowmask[j] = mask;
  }

  // Staging registers and the three phases of a K-step are written out as macros (not lambdas /
  // runtime-indexed arrays) so that the prefetched tile provably stays in VGPRs.
  static_assert(PR == 4 && (WR == 2 || WR == 4), "staging code below is written for BM=128, BN=64|128");
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;  // named scalars: hipcc keeps arrays pinned by a
                                                   // scheduling barrier in scratch memory
  rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define LOAD_PIX(J, DST)                                                                            \
  DST = *reinterpret_cast<const float4*>(((rowmask[J] >> tap_) & 1u) ? in + rowoff[J] + toff_ : a.zeros);
#define LOAD_TILE(CH, TAP)                                                                         \
  {                                                                                                \
    const int tap_ = (TAP);                                                                        \
    const int r_ = tap_ / 3, s_ = tap_ - r_ * 3;                                                   \
    const int toff_ = (r_ * a.W + s_) * a.in_ld + (CH) * 32;                                       \
    /* out-of-image taps read a 16-byte zero line instead of branching around the load */         \
    LOAD_PIX(0, ra0) LOAD_PIX(1, ra1) LOAD_PIX(2, ra2) LOAD_PIX(3, ra3)                            \
    const float* wt_ = wgt + ((size_t)((CH) * 9 + tap_) * (a.tiles_n * BN) + n0) * 32 + tid * 4;   \
    rb0 = *reinterpret_cast<const float4*>(wt_);                                                   \
    rb1 = *reinterpret_cast<const float4*>(wt_ + 1024);                                            \
    if constexpr (WR == 4) {                                                                       \
      rb2 = *reinterpret_cast<const float4*>(wt_ + 2048);                                          \
      rb3 = *reinterpret_cast<const float4*>(wt_ + 3072);                                          \
    }                                                                                              \
  }
#define STORE_TILE(BUFI)                                                                           \
  {                                                                                                \
    float* dst_ = smem + (BUFI) * BUF + r0 * LDK + c4 * 4;                                         \
    *reinterpret_cast<float4*>(dst_) = ra0;                                                        \
    *reinterpret_cast<float4*>(dst_ + 32 * LDK) = ra1;                                             \
    *reinterpret_cast<float4*>(dst_ + 64 * LDK) = ra2;                                             \
    *reinterpret_cast<float4*>(dst_ + 96 * LDK) = ra3;                                             \
    *reinterpret_cast<float4*>(dst_ + BM * LDK) = rb0;                                             \
    *reinterpret_cast<float4*>(dst_ + (BM + 32) * LDK) = rb1;                                      \
    if constexpr (WR == 4) {                                                                       \
      *reinterpret_cast<float4*>(dst_ + (BM + 64) * LDK) = rb2;                                    \
      *reinterpret_cast<float4*>(dst_ + (BM + 96) * LDK) = rb3;                                    \
    }                                                                                              \
  }
#define MMA_GROUP(KG)                                                                              \
  {                                                                                                \
    float4 pv_[PT], wv_[CT];                                                                       \
    _Pragma("unroll") for (int i = 0; i < PT; ++i)                                                 \
        pv_[i] = *reinterpret_cast<const float4*>(pP + i * 32 * LDK + (KG) * 8);                   \
    _Pragma("unroll") for (int j = 0; j < CT; ++j)                                                 \
        wv_[j] = *reinterpret_cast<const float4*>(pW + j * 32 * LDK + (KG) * 8);                   \
    _Pragma("unroll") for (int j = 0; j < CT; ++j) _Pragma("unroll") for (int i = 0; i < PT; ++i)  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[j].x, pv_[i].x, acc[i][j], 0, 0, 0);  \
    _Pragma("unroll") for (int j = 0; j < CT; ++j) _Pragma("unroll") for (int i = 0; i < PT; ++i)  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[j].y, pv_[i].y, acc[i][j], 0, 0, 0);  \
    _Pragma("unroll") for (int j = 0; j < CT; ++j) _Pragma("unroll") for (int i = 0; i < PT; ++i)  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[j].z, pv_[i].z, acc[i][j], 0, 0, 0);  \
    _Pragma("unroll") for (int j = 0; j < CT; ++j) _Pragma("unroll") for (int i = 0; i < PT; ++i)  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[j].w, pv_[i].w, acc[i][j], 0, 0, 0);  \
  }

  f32x16 acc[PT][CT];
#pragma unroll
  for (int i = 0; i < PT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  LOAD_TILE(0, 0)
  STORE_TILE(0)
  __syncthreads();

  int ch = 0, tap = 0;
  for (int kt = 0; kt < KT - 1; ++kt) {
    const int buf = kt & 1;
    if (++tap == 9) { tap = 0; ++ch; }
    LOAD_TILE(ch, tap)
    // keep all global loads of the next tile in flight under this tile's MFMAs: without the pin
    // hipcc sinks the weight loads to just before their ds_write and eats the L2 latency
    __builtin_amdgcn_sched_barrier(0);
    const float* pP = smem + buf * BUF + (wm * PT * 32 + l31) * LDK + hh * 4;
    const float* pW = smem + buf * BUF + (BM + wn * CT * 32 + l31) * LDK + hh * 4;
    MMA_GROUP(0)
    MMA_GROUP(1)
    // the other LDS buffer has been free since the barrier that ended the previous K-step: write
    // the prefetched tile half-way, so the ds_writes retire under the remaining MFMAs
#ifdef SE3TN_PIN_STORE
    __builtin_amdgcn_sched_barrier(0);
#endif
    STORE_TILE(buf ^ 1)
#ifdef SE3TN_PIN_STORE
    __builtin_amdgcn_sched_barrier(0);
#endif
    MMA_GROUP(2)
    MMA_GROUP(3)
    __syncthreads();
  }
  {
    const int buf = (KT - 1) & 1;
    const float* pP = smem + buf * BUF + (wm * PT * 32 + l31) * LDK + hh * 4;
    const float* pW = smem + buf * BUF + (BM + wn * CT * 32 + l31) * LDK + hh * 4;
    MMA_GROUP(0)
    MMA_GROUP(1)
    MMA_GROUP(2)
    MMA_GROUP(3)
  }
#undef LOAD_TILE
#undef LOAD_PIX
#undef STORE_TILE
#undef MMA_GROUP

  // ---- epilogue: lane holds pixel (l31) x couts {8q + 4hh + 0..3}, q = 0..3, per 32x32 tile ----
  const float* __restrict__ bias = a.bias + (size_t)g * a.bias_gs;
  const float* __restrict__ res = (EPI == 1) ? a.res + (size_t)g * a.res_gs : nullptr;
  float* __restrict__ out = a.out + (size_t)g * a.out_gs;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int m = m0 + (wm * PT + i) * 32 + l31;
    if (m >= a.M) continue;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = n0 + (wn * CT + j) * 32 + q * 8 + hh * 4;
        float4 v = make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2],
                               acc[i][j][4 * q + 3]);
        const float4 b = *reinterpret_cast<const float4*>(bias + c);
        v = apply_epilogue<EPI>(v, b, (EPI == 1) ? res + (size_t)m * a.res_ld + c : nullptr);
        *reinterpret_cast<float4*>(out + (size_t)m * a.out_ld + c) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Variant B: the same GEMM with the operand tiles DMA'd straight into LDS
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass, the wait for the next tile sits at
// the END of the K-step under 4096 cycles of MFMA).  The LDS image of a DMA is lane-linear
// (wave-uniform base + lane*16 B), so rows are unpadded 128-byte runs and bank conflicts are avoided
// by an XOR swizzle applied on the SOURCE side (which 16-byte channel slot a lane fetches) and again
// on the ds_read_b128 address:  physical slot = logical slot ^ ((row >> 1) & 7).
// ------------------------------------------------------------------------------------------------
// The DMA is issued from inline asm on purpose: for the builtin form hipcc (ROCm 7.2) cannot prove
// that the DMA's LDS destination (the other buffer) does not alias the ds_reads of the current
// buffer and inserts `s_waitcnt vmcnt(0)` in front of the first ds_read of every K-step, which
// serialises the pipeline.  An asm load is invisible to its waitcnt bookkeeping; the one wait this
// kernel needs (`vmcnt(0)` before the barrier that ends the K-step) is written by hand.
// M0 = wave-uniform LDS byte address; lane l lands at M0 + 16 l.
__device__ __forceinline__ void glds16(const float* g, unsigned lds_byte_addr) {
#if (SE3TN_ABLATE & 4)
  asm volatile("" ::"v"(g), "s"(lds_byte_addr));  // keep the address arithmetic, drop the DMA
  return;
#endif
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g), "s"(lds_byte_addr)
      : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const float* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) float*)p;
}

template <int CIN, int STRIDE, int WM, int WN, int PT, int CT, int EPI, int MINW>
__global__ __launch_bounds__(256, MINW) void conv3x3_glds_kernel(const ConvArgs a) {
  constexpr int BM = WM * PT * 32, BN = WN * CT * 32;
  constexpr int NCH = CIN / 32, KT = NCH * 9;
  constexpr int PR = BM / 32, WR = BN / 32;
  constexpr int BUF = (BM + BN) * 32;  // floats per LDS buffer, rows of 32 floats, no padding
  static_assert(WM * WN == 4, "4 waves per workgroup");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, hh = lane >> 5;

  const int panels = a.groups * a.tiles_n;
  const int p = blockIdx.x % panels, mt = blockIdx.x / panels;
  const int g = p / a.tiles_n, nt = p % a.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;
  const float* __restrict__ in = a.in + (size_t)g * a.in_gs;
  const float* __restrict__ wgt = a.w + (size_t)g * a.w_gs;

  // staging: thread t fills LDS slot (t & 7) of rows (t >> 3) + 32 j; with the swizzle that slot holds
  // channel block c4 = (t & 7) ^ ((row >> 1) & 7)   (32 j does not change (row >> 1) & 7)
  const int r0 = tid >> 3;
  const int c4 = (tid & 7) ^ ((r0 >> 1) & 7);
  int rowoff[PR];
  unsigned rowmask[PR];
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int j = 0; j < PR; ++j) {
    const int m = m0 + r0 + 32 * j;
    unsigned mask = 0;
    int off = 0;
    if (m < a.M) {
      const int n = m / HoWo, rem = m - n * HoWo;
      const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
      const int hi0 = ho * STRIDE - 1, wi0 = wo * STRIDE - 1;
      off = ((n * a.H + hi0) * a.W + wi0) * a.in_ld + c4 * 4;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          if ((unsigned)(hi0 + r) < (unsigned)a.H && (unsigned)(wi0 + s) < (unsigned)a.W)
            mask |= 1u << (r * 3 + s);
    }
    rowoff[j] = off;
    rowmask[j] = mask;
  }
  const int woff = r0 * 32 + c4 * 4;  // within a [BN][32] weight tile
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  const int wid_u = __builtin_amdgcn_readfirstlane(wid);

#define ISSUE_TILE(CH, TAP, BUFI)                                                                  \
  {                                                                                                \
    const int tap_ = (TAP);                                                                        \
    const int r_ = tap_ / 3, s_ = tap_ - r_ * 3;                                                   \
    const int toff_ = (r_ * a.W + s_) * a.in_ld + (CH) * 32;                                       \
    /* this wave's 1 KB chunks: rows 32 j + 8 wid .. +7 */                                           \
    const unsigned lb_ = lds_base + (unsigned)(((BUFI) * BUF + wid_u * 256) * 4);                  \
    _Pragma("unroll") for (int j = 0; j < PR; ++j)                                                 \
        glds16(((rowmask[j] >> tap_) & 1u) ? in + rowoff[j] + toff_ : a.zeros, lb_ + j * 4096);    \
    const float* wt_ = wgt + ((size_t)((CH) * 9 + tap_) * (a.tiles_n * BN) + n0) * 32 + woff;     \
    _Pragma("unroll") for (int j = 0; j < WR; ++j) glds16(wt_ + j * 1024, lb_ + (BM * 128) + j * 4096); \
  }

  // fragment reads: lane (l31, hh) reads logical slot 2 kg + hh of row (tile base + l31); the
  // tile bases are multiples of 32 rows, so the swizzle term only depends on l31
  const int X = (l31 >> 1) & 7;
  const int lo = (hh ^ (X & 1)) * 4, xk = X >> 1;
  const int fo0 = ((0 ^ xk) << 3) + lo, fo1 = ((1 ^ xk) << 3) + lo, fo2 = ((2 ^ xk) << 3) + lo,
            fo3 = ((3 ^ xk) << 3) + lo;

  f32x16 acc[PT][CT];
#pragma unroll
  for (int i = 0; i < PT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#define MMA_GROUP_S(FO)                                                                            \
  {                                                                                                \
    float4 pv_[PT], wv_[CT];                                                                       \
    _Pragma("unroll") for (int i = 0; i < PT; ++i)                                                 \
        pv_[i] = *reinterpret_cast<const float4*>(pP + i * 1024 + (FO));                           \
    _Pragma("unroll") for (int j = 0; j < CT; ++j)                                                 \
        wv_[j] = *reinterpret_cast<const float4*>(pW + j * 1024 + (FO));                           \
    _Pragma("unroll") for (int j = 0; j < CT; ++j) _Pragma("unroll") for (int i = 0; i < PT; ++i)  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[j].x, pv_[i].x, acc[i][j], 0, 0, 0);  \
    _Pragma("unroll") for (int j = 0; j < CT; ++j) _Pragma("unroll") for (int i = 0; i < PT; ++i)  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[j].y, pv_[i].y, acc[i][j], 0, 0, 0);  \
    _Pragma("unroll") for (int j = 0; j < CT; ++j) _Pragma("unroll") for (int i = 0; i < PT; ++i)  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[j].z, pv_[i].z, acc[i][j], 0, 0, 0);  \
    _Pragma("unroll") for (int j = 0; j < CT; ++j) _Pragma("unroll") for (int i = 0; i < PT; ++i)  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[j].w, pv_[i].w, acc[i][j], 0, 0, 0);  \
  }

  ISSUE_TILE(0, 0, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int ch = 0, tap = 0;
  for (int kt = 0; kt < KT - 1; ++kt) {
    const int buf = kt & 1;
    if (++tap == 9) { tap = 0; ++ch; }
#if !(SE3TN_ABLATE & 1)
    ISSUE_TILE(ch, tap, buf ^ 1)
#endif
    const float* pP = smem + buf * BUF + (wm * PT * 32 + l31) * 32;
    const float* pW = smem + buf * BUF + (BM + wn * CT * 32 + l31) * 32;
    MMA_GROUP_S(fo0)
    MMA_GROUP_S(fo1)
    MMA_GROUP_S(fo2)
    MMA_GROUP_S(fo3)
    // the DMA issued at the top of this K-step has had the whole MFMA phase to land
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !(SE3TN_ABLATE & 2)
    __syncthreads();
#endif
  }
  {
    const int buf = (KT - 1) & 1;
    const float* pP = smem + buf * BUF + (wm * PT * 32 + l31) * 32;
    const float* pW = smem + buf * BUF + (BM + wn * CT * 32 + l31) * 32;
    MMA_GROUP_S(fo0)
    MMA_GROUP_S(fo1)
    MMA_GROUP_S(fo2)
    MMA_GROUP_S(fo3)
  }
#undef ISSUE_TILE
#undef MMA_GROUP_S

  const float* __restrict__ bias = a.bias + (size_t)g * a.bias_gs;
  const float* __restrict__ res = (EPI == 1) ? a.res + (size_t)g * a.res_gs : nullptr;
  float* __restrict__ out = a.out + (size_t)g * a.out_gs;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int m = m0 + (wm * PT + i) * 32 + l31;
    if (m >= a.M) continue;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = n0 + (wn * CT + j) * 32 + q * 8 + hh * 4;
        float4 v = make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2],
                               acc[i][j][4 * q + 3]);
        const float4 b = *reinterpret_cast<const float4*>(bias + c);
        v = apply_epilogue<EPI>(v, b, (EPI == 1) ? res + (size_t)m * a.res_ld + c : nullptr);
        *reinterpret_cast<float4*>(out + (size_t)m * a.out_ld + c) = v;
      }
    }
  }
}

template <int CIN, int STRIDE, int WM, int WN, int PT, int CT, int EPI, int MINW>
static hipError_t launch_glds(ConvArgs a, hipStream_t st) {
  constexpr int BM = WM * PT * 32, BN = WN * CT * 32;
  constexpr size_t lds = (size_t)2 * (BM + BN) * 32 * sizeof(float);
  auto kern = conv3x3_glds_kernel<CIN, STRIDE, WM, WN, PT, CT, EPI, MINW>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int tiles_m = (a.M + BM - 1) / BM;
  hipLaunchKernelGGL(kern, dim3(tiles_m * a.tiles_n * a.groups), dim3(256), lds, st, a);
  return hipGetLastError();
}

template <int CIN, int STRIDE, int WM, int WN, int PT, int CT, int EPI>
static hipError_t launch_one(ConvArgs a, hipStream_t st) {
  constexpr int BM = WM * PT * 32, BN = WN * CT * 32;
  constexpr size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
  auto kern = conv3x3_mfma_kernel<CIN, STRIDE, WM, WN, PT, CT, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int tiles_m = (a.M + BM - 1) / BM;
  const dim3 grid(tiles_m * a.tiles_n * a.groups);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  return hipGetLastError();
}

// Tile shapes.  Cout=64 layers: 128 pixels x 64 couts (4 waves stacked along pixels);
// everything else: 128 x 128 (2x2 waves, 2x2 32x32 tiles per wave).
hipError_t launch_conv3x3(const ConvArgs& a0, int cin, int cout, int stride, int epi, hipStream_t st) {
  ConvArgs a = a0;
#ifdef SE3TN_GLDS
  if (cin == 64 && cout == 64 && stride == 1) {
    a.tiles_n = 1;
    if (epi == 0) return launch_glds<64, 1, 4, 1, 1, 2, 0, 3>(a, st);
    if (epi == 1) return launch_glds<64, 1, 4, 1, 1, 2, 1, 3>(a, st);
  }
  a.tiles_n = cout / 128;
  if (cin == 128 && stride == 2 && epi == 2) return launch_glds<128, 2, 2, 2, 2, 2, 2, 2>(a, st);
  if (cin == 256 && stride == 1 && epi == 0) return launch_glds<256, 1, 2, 2, 2, 2, 0, 2>(a, st);
  if (cin == 256 && stride == 1 && epi == 1) return launch_glds<256, 1, 2, 2, 2, 2, 1, 2>(a, st);
  if (cin == 256 && stride == 2 && epi == 2) return launch_glds<256, 2, 2, 2, 2, 2, 2, 2>(a, st);
  if (cin == 512 && stride == 1 && epi == 0) return launch_glds<512, 1, 2, 2, 2, 2, 0, 2>(a, st);
  if (cin == 512 && stride == 1 && epi == 1) return launch_glds<512, 1, 2, 2, 2, 2, 1, 2>(a, st);
  return hipErrorInvalidValue;
#endif
  if (cin == 64 && cout == 64 && stride == 1) {
    a.tiles_n = 1;
    if (epi == 0) return launch_one<64, 1, 4, 1, 1, 2, 0>(a, st);
    if (epi == 1) return launch_one<64, 1, 4, 1, 1, 2, 1>(a, st);
  }
  a.tiles_n = cout / 128;
  if (cin == 128 && stride == 2 && epi == 2) return launch_one<128, 2, 2, 2, 2, 2, 2>(a, st);
  if (cin == 256 && stride == 1 && epi == 0) return launch_one<256, 1, 2, 2, 2, 2, 0>(a, st);
  if (cin == 256 && stride == 1 && epi == 1) return launch_one<256, 1, 2, 2, 2, 2, 1>(a, st);
  if (cin == 256 && stride == 2 && epi == 2) return launch_one<256, 2, 2, 2, 2, 2, 2>(a, st);
  if (cin == 512 && stride == 1 && epi == 0) return launch_one<512, 1, 2, 2, 2, 2, 0>(a, st);
  if (cin == 512 && stride == 1 && epi == 1) return launch_one<512, 1, 2, 2, 2, 2, 1>(a, st);
  return hipErrorInvalidValue;
}

}  // namespace se3tn
