// Stem: Conv2d(4->64, k=7, s=2, p=3) + folded BN + SELU for both branches (convA1 / convB1,
// se3_tracknet.py:57,61 -> network_modules.py:59-66), then MaxPool2d(3, s=2, p=1)
// (se3_tracknet.py:58,62) as a separate streaming kernel.
//
// stem7x7_slab_kernel: exact-f32 MFMA implicit GEMM (rows = output pixels, K = 49 taps x 4
// channels, N = 64), persistent: 256 workgroups (128 per branch), 8 waves each, loop over tiles of
// 256 consecutive output pixels of one image.
//  * the network input is stored with a 3-pixel zero border ([n,182,182,4], one pixel = 16 bytes):
//    tap (r,s) of output (ho,wo) is padded input pixel (2ho+r, 2wo+s) -- no bounds logic;
//  * per tile the 13 padded input rows it touches are ONE contiguous 37 KB run: DMA'd
//    (global_load_lds, purely linear, 4 pieces per M0 setup) into a double-buffered LDS slab while
//    the previous tile computes; one barrier per TILE;
//  * the branch's whole folded weight matrix [64][204] (k = pair*8 + half*4 + c, row padded to 816
//    bytes: conflict-free ds_read_b128) is DMA'd into LDS once per workgroup;
//  * a pixel is one float4 = one tap's 4 channels.  v_mfma_f32_32x32x2_f32 takes k from lanes 0-31
//    and k+1 from lanes 32-63, so the two half-waves read two DIFFERENT taps and 4 MFMAs consume
//    the pair.  Taps are paired so that the half-wave term folds into three lane-constant base
//    addresses (in-row pairs (r,2j)|(r,2j+1); column-6 pairs (r,6)|(r+1,6); (6,6)|zero weights):
//    every fragment read of the inner loop is base + compile-time immediate, no VALU at all.
// Output: [n,88,88,128] NHWC, branch A in channels 0-63, branch B in 64-127.
#include "se3tn_internal.h"

#ifndef SE3TN_STEM_ABLATE
#define SE3TN_STEM_ABLATE 0  // timing ablations only: 1 = no output stores, 2 = no slab DMA
#endif

namespace se3tn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
constexpr float SELU_ALPHA = 1.6732632423543772848170429916717f;
constexpr float SELU_SCALE = 1.0507009873554804934193349852946f;
__device__ __forceinline__ float selu_s(float v) {
  // exp via v_exp_f32: absolute error ~1e-7 on a value of magnitude <= 1.76 (f32-roundoff class)
  return v > 0.f ? SELU_SCALE * v : (SELU_SCALE * SELU_ALPHA) * (__expf(v) - 1.f);
}

constexpr int IP = RES + 6;                    // 182: padded input rows / columns
constexpr int TILE = 256;                      // output pixels per tile
constexpr int TILES_PER_IMAGE = (S1 * S1 + TILE - 1) / TILE;  // 31 (the last one holds 64 pixels)
constexpr int SLAB_ROWS = 13;                  // 2 * (4 output rows - 1) + 7
constexpr int SLAB_PIECES = (SLAB_ROWS * IP * 16 + 1023) / 1024;  // 37 x 1 KB
constexpr int SLAB_BYTES = SLAB_PIECES * 1024;
constexpr int WROW = 204;                      // floats per cout row in LDS / in the blob
constexpr int WBYTES = 64 * WROW * 4;          // 52,224 = 51 x 1 KB

struct StemArgs {
  const float* in[2];  // [n,182,182,4] per branch (zero border of 3)
  const float* w;      // [2][64][204]
  const float* bias;   // [2][64]
  const float* wscale; // F16X3: [2][64] per-cout 2^-k
  float* out;          // [n,88,88,128]
  int n;
};

// F16X3 = 1: the same kernel on the f16 matrix cores.  A pixel is 4 x f16 hi | 4 x f16 lo (the same 16
// bytes), a weight entry (pair, half-wave) is w_hi(4) | w_lo(4).  With B = [x_hi | x_lo] as loaded and
// A1 = [w_hi | w_hi], A2 = [w_lo | 0] built in registers, two v_mfma_f32_32x32x16_f16 per tap pair give
// w_hi x_hi + w_hi x_lo + w_lo x_hi for both taps (vs four f32 MFMAs of twice the cycles each).

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
// lane l: LDS[lds + IMM + 16 l] <- global[base + 16 l + IMM]
template <int IMM>
__device__ __forceinline__ void dma1k(const float* base, unsigned voff, unsigned lds) {
  const unsigned long long b_ = (unsigned long long)base;
  const unsigned blo_ = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)b_);
  const unsigned bhi_ = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b_ >> 32));
  const unsigned long long sb_ = ((unsigned long long)bhi_ << 32) | (unsigned long long)blo_;
  const unsigned lds_ = __builtin_amdgcn_readfirstlane(lds);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:%4\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(lds_), "s"(sb_), "i"(IMM)
      : "memory");
}

template <int F16X3>
__global__ __launch_bounds__(512, 2) void stem7x7_slab_kernel(const StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [weights][slab 0][slab 1]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int nblk = gridDim.x >> 1;
  const int br = blockIdx.x / nblk, bid = blockIdx.x - br * nblk;
  const float* __restrict__ in = a.in[br];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const unsigned lvoff = lane * 16;
  const int ntiles = a.n * TILES_PER_IMAGE;

  // slab of tile t: padded input rows [2 h0, 2 h0 + 13) of image t / 31, h0 = first output row
  auto issue_slab = [&](int t, int buf) {
#if (SE3TN_STEM_ABLATE & 2)
    return;  // timing ablation: no slab DMA
#endif
    const int img = t / TILES_PER_IMAGE, h0 = ((t - img * TILES_PER_IMAGE) * TILE) / S1;
    const float* src = in + ((size_t)img * IP * IP + (size_t)(2 * h0) * IP) * 4;
    const unsigned dst = lds0 + WBYTES + buf * SLAB_BYTES;
    // pieces 4 wid .. 4 wid + 3 with one M0 setup, then pieces 32 + wid for wid < 5
    const float* s4 = src + wid * 1024;
    dma1k<0>(s4, lvoff, dst + wid * 4096);
    dma1k<1024>(s4, lvoff, dst + wid * 4096);
    dma1k<2048>(s4, lvoff, dst + wid * 4096);
    dma1k<3072>(s4, lvoff, dst + wid * 4096);
    if (wid < SLAB_PIECES - 32) dma1k<0>(src + (32 + wid) * 256, lvoff, dst + (32 + wid) * 1024);
  };

  // weights of this branch: 51 pieces
  {
    const float* wsrc = a.w + (size_t)br * 64 * WROW;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int pc = wid + 8 * j;
      if (pc < WBYTES / 1024) dma1k<0>(wsrc + pc * 256, lvoff, lds0 + pc * 1024);
    }
  }
  if (bid < ntiles) issue_slab(bid, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const unsigned char* wl = smem + (l31 * WROW + hh * 4) * 4;  // + ct * 32 rows + pair * 32 bytes
  // lane-constant bias values (8 x float4), loaded once: an epilogue that loads them per tile waits
  // an L2 round trip per load
  float4 bias_r[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bias_r[j][q] = *reinterpret_cast<const float4*>(a.bias + br * 64 + j * 32 + q * 8 + hh * 4);
  float4 wsc_r[2][4];
  if (F16X3) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        wsc_r[j][q] = *reinterpret_cast<const float4*>(a.wscale + br * 64 + j * 32 + q * 8 + hh * 4);
  }

  auto epilogue = [&](const f32x16 (&acc)[2], float* __restrict__ out) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = j * 32 + q * 8 + hh * 4;
        const float4 b = bias_r[j][q];
        float4 w = make_float4(1.f, 1.f, 1.f, 1.f);
        if (F16X3) w = wsc_r[j][q];
        float4 v;
        v.x = selu_s(acc[j][4 * q + 0] * w.x + b.x);
        v.y = selu_s(acc[j][4 * q + 1] * w.y + b.y);
        v.z = selu_s(acc[j][4 * q + 2] * w.z + b.z);
        v.w = selu_s(acc[j][4 * q + 3] * w.w + b.w);
#if (SE3TN_STEM_ABLATE & 1)
        asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));  // timing ablation: no stores
#else
        *reinterpret_cast<float4*>(out + c) = v;
#endif
      }
  };

  // The 8 waves of the workgroup run in lock-step (one barrier per tile), and waves w and w+4
  // share a SIMD.  If both did [MFMA][epilogue] the matrix pipe would idle during every epilogue, so
  // waves 4-7 run one tile behind on the epilogue: [epilogue of the previous tile][MFMA], which puts
  // each SIMD's two waves in complementary phases.
  const bool deferred = wid >= 4;
  f32x16 held[2];
  float* held_out = nullptr;

  int it = 0;
  for (int t = bid; t < ntiles; t += nblk, ++it) {
    const int buf = it & 1;
    if (t + nblk < ntiles) issue_slab(t + nblk, buf ^ 1);
    if (deferred && held_out) epilogue(held, held_out);

    const int img = t / TILES_PER_IMAGE, p0 = (t - img * TILES_PER_IMAGE) * TILE;
    const int h0 = p0 / S1;
    const int p = p0 + wid * 32 + l31;         // output pixel of this lane within the image
    const bool ok = p < S1 * S1;
    const int pc = ok ? p : S1 * S1 - 1;
    const int ho = pc / S1, wo = pc - ho * S1;
    const unsigned char* sl = smem + WBYTES + buf * SLAB_BYTES + ((2 * (ho - h0)) * IP + 2 * wo) * 16;
    const unsigned char* slA = sl + hh * 16;        // in-row pairs: half-wave 1 reads tap s+1
    const unsigned char* slB = sl + hh * IP * 16;   // column-6 pairs: half-wave 1 reads row r+1

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // fragments of pair k+1 are fetched before the 8 MFMAs of pair k are issued (hipcc otherwise
    // re-uses one register set and exposes an LDS round trip per pair); all addresses are
    // base + compile-time immediate
#define STEM_LOAD(F, K)                                                                              \
    {                                                                                               \
      constexpr int k_ = (K);                                                                       \
      const unsigned char* px_ = k_ < 21 ? slA + ((k_ / 3) * IP + 2 * (k_ % 3)) * 16                 \
                                 : k_ < 24 ? slB + (2 * (k_ - 21) * IP + 6) * 16                     \
                                           : sl + (6 * IP + 6) * 16;                                \
      F##p = *reinterpret_cast<const float4*>(px_);                                                 \
      F##w0 = *reinterpret_cast<const float4*>(wl + k_ * 32);                                       \
      F##w1 = *reinterpret_cast<const float4*>(wl + 32 * WROW * 4 + k_ * 32);                       \
    }
#define STEM_MMA16(W, P, J)                                                                         \
    {                                                                                               \
      const uint4v wu_ = __builtin_bit_cast(uint4v, W);                                             \
      const uint4v a1_ = {wu_[0], wu_[1], wu_[0], wu_[1]};                                          \
      const uint4v a2_ = {wu_[2], wu_[3], 0u, 0u};                                                  \
      const half8 pb_ = __builtin_bit_cast(half8, P);                                               \
      acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a1_), pb_, acc[J], 0, 0, 0); \
      acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a2_), pb_, acc[J], 0, 0, 0); \
    }
#define STEM_MMA(F)                                                                                 \
    if (F16X3) {                                                                                    \
      STEM_MMA16(F##w0, F##p, 0)                                                                    \
      STEM_MMA16(F##w1, F##p, 1)                                                                    \
    } else {                                                                                        \
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(F##w0.x, F##p.x, acc[0], 0, 0, 0);              \
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(F##w1.x, F##p.x, acc[1], 0, 0, 0);              \
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(F##w0.y, F##p.y, acc[0], 0, 0, 0);              \
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(F##w1.y, F##p.y, acc[1], 0, 0, 0);              \
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(F##w0.z, F##p.z, acc[0], 0, 0, 0);              \
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(F##w1.z, F##p.z, acc[1], 0, 0, 0);              \
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(F##w0.w, F##p.w, acc[0], 0, 0, 0);              \
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(F##w1.w, F##p.w, acc[1], 0, 0, 0);              \
    }
#define STEM_FENCE __builtin_amdgcn_sched_barrier(0);  /* keeps the next pair's reads ahead of the MFMAs */
#define STEM_STEP2(K)   /* pairs K (set A) and K+1 (set B); set A of pair K is already loaded */     \
    STEM_LOAD(fB, (K) + 1) STEM_FENCE STEM_MMA(fA) STEM_LOAD(fA, (K) + 2) STEM_FENCE STEM_MMA(fB)
    float4 fAp, fAw0, fAw1, fBp, fBw0, fBw1;
    STEM_LOAD(fA, 0)
    STEM_STEP2(0) STEM_STEP2(2) STEM_STEP2(4) STEM_STEP2(6) STEM_STEP2(8) STEM_STEP2(10)
    STEM_STEP2(12) STEM_STEP2(14) STEM_STEP2(16) STEM_STEP2(18) STEM_STEP2(20)
    STEM_LOAD(fB, 23) STEM_FENCE STEM_MMA(fA) STEM_LOAD(fA, 24) STEM_FENCE STEM_MMA(fB)
    STEM_MMA(fA)
#undef STEM_FENCE
#undef STEM_STEP2
#undef STEM_MMA
#undef STEM_MMA16
#undef STEM_LOAD

    float* out = ok ? a.out + ((size_t)img * S1 * S1 + p) * 128 + br * 64 : nullptr;
    if (deferred) {
      held[0] = acc[0]; held[1] = acc[1];
      held_out = out;
    } else if (out) {
      epilogue(acc, out);
    }
    // next tile's slab has had this tile's 200 MFMAs per wave to land
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (deferred && held_out) epilogue(held, held_out);
}

hipError_t launch_stem(const float* inA, const float* inB, const float* w, const float* bias,
                       const float* wscale, float* out, int n, hipStream_t st) {
  constexpr size_t lds = WBYTES + 2 * SLAB_BYTES;  // 128,000 B
  static PerDeviceOnce attr;
  bool* done = attr.current();
  if (!done || !*done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem7x7_slab_kernel<0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem7x7_slab_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (done) *done = true;
  }
  StemArgs a;
  a.in[0] = inA; a.in[1] = inB; a.w = w; a.bias = bias; a.wscale = wscale; a.out = out; a.n = n;
  const int ntiles = n * TILES_PER_IMAGE;
  const int per_branch = ntiles < 128 ? ntiles : 128;  // one workgroup per CU, half the chip per branch
  if (wscale) hipLaunchKernelGGL(stem7x7_slab_kernel<1>, dim3(2 * per_branch), dim3(512), lds, st, a);
  else hipLaunchKernelGGL(stem7x7_slab_kernel<0>, dim3(2 * per_branch), dim3(512), lds, st, a);
  return hipGetLastError();
}

// MaxPool2d(kernel 3, stride 2, padding 1) on [n,88,88,128] -> the interior of the zero-bordered
// [n,46,46,128] tensor the 64-channel convs read; the pool's own padding is -inf (never wins).
// A streaming pass: 254 MB in, 68 MB out at batch 64, bound by what the memory system gives a pass that reads every byte once
// (scripts/probes/hbm_stream.hip: 54 us for this shape, a write of 256 MB then a 4 : 1 read-back).  So every input element is
// loaded by ONE thread (+ 1 / 8 of the columns and one halo row per strip twice): a thread owns 4 neighbouring output
// columns x 4 channels (9 input columns), takes the horizontal maxima of an input row in registers and walks down a strip
// of output rows, carrying the odd row it shares with the next output row.  One workgroup = one (image, strip): the 11 column
// groups x 32 channel quads; the shared columns meet in that CU's L1.  max is exact: the order of the comparisons is free.
constexpr int POOL_QG = S2 / 4;                 // 11 groups of 4 output columns
constexpr int POOL_THREADS = POOL_QG * 32;      // 352
static_assert(S1 == 2 * S2 && S2 % 4 == 0, "pool geometry");

__device__ __forceinline__ float4 max4(const float4 a, const float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

// horizontal 3-maxima of input row y for the thread's 4 outputs (input columns x0 .. x0 + 8, x0 = -1 for the first group)
__device__ __forceinline__ void pool_row(const float* __restrict__ src, int y, int x0, bool left, float4 hm[4]) {
  const float* p = src + (size_t)y * S1 * 128;
  float4 h[9];
  h[0] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  if (left) h[0] = *reinterpret_cast<const float4*>(p + x0 * 128);
#pragma unroll
  for (int j = 1; j < 9; ++j) h[j] = *reinterpret_cast<const float4*>(p + (x0 + j) * 128);
#pragma unroll
  for (int k = 0; k < 4; ++k) hm[k] = max4(max4(h[2 * k], h[2 * k + 1]), h[2 * k + 2]);
}

__global__ __launch_bounds__(POOL_THREADS) void maxpool3x3s2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                     int n, int rows, int split_out) {
  // workgroup -> (image, strip).  Consecutive workgroup ids land on different XCDs (id % 8); strips s and s + 1 share
  // one input row, so an XCD gets whole images (image = 8 k + xcd) and the shared row comes from its own L2
  const int strips = S2 / rows;
  int blk = blockIdx.x;
  const int full = n / 8 * 8;
  if (blk < full * strips) {
    const int xcd = blk & 7, seq = blk >> 3;
    blk = ((seq / strips) * 8 + xcd) * strips + seq % strips;
  }
  const int img = blk / strips, strip = blk - img * strips;
  const int c4 = threadIdx.x & 31, g = threadIdx.x >> 5;
  const float* src = in + (size_t)img * S1 * S1 * 128 + c4 * 4;
  const int x0 = 8 * g - 1;
  const bool left = g > 0;
  const int po0 = strip * rows;
  float4 carry[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) carry[k] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  if (po0 > 0) pool_row(src, 2 * po0 - 1, x0, left, carry);
  for (int r = 0; r < rows; ++r) {
    const int po = po0 + r;
    float4 a[4], b[4];
    pool_row(src, 2 * po, x0, left, a);
    pool_row(src, 2 * po + 1, x0, left, b);
    float* dst = out + ((size_t)(img * (S2 + 2) + po + 1) * (S2 + 2) + 4 * g + 1) * 128;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 m = max4(max4(carry[k], a[k]), b[k]);
      carry[k] = b[k];
      if (split_out) {  // f16x3 mode: 32-channel chunk = 32 f16 hi | 32 f16 lo (SELU output is bounded below, finite)
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        const int c = c4 * 4;
        unsigned char* p_ = reinterpret_cast<unsigned char*>(dst + k * 128) + (c >> 5) * 128 + (c & 31) * 2;
        const float f[4] = {m.x, m.y, m.z, m.w};
        half4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = (_Float16)f[e]; l[e] = (_Float16)(f[e] - (float)h[e]); }
        *reinterpret_cast<half4*>(p_) = h;
        *reinterpret_cast<half4*>(p_ + 64) = l;
      } else {
        *reinterpret_cast<float4*>(dst + k * 128 + c4 * 4) = m;
      }
    }
  }
}

// Output rows per strip: the longest strips (fewest halo rows, longest streams) that still give every one of the 256 CUs a
// workgroup.  Measured at batch 64 (profiles/EXPERIMENTS.md item 37): 11 rows = 256 workgroups 51.9 us; 4 rows = 704 workgroups
// (2.75 per CU) 59.7; 22 rows = 128 workgroups 77.1; 2 / 1 rows 66 / 65; the one-pixel-per-thread kernel before it 70.6.
// SE3TN_POOL_ROWS (developer switch, read once) forces a divisor of 44.
static int pool_rows(int n) {
  static const int forced = [] { const char* e = getenv("SE3TN_POOL_ROWS"); return e ? atoi(e) : 0; }();
  if (forced > 0 && S2 % forced == 0) return forced;
  const int divs[] = {44, 22, 11, 4, 2, 1};
  for (int r : divs)
    if (n * (S2 / r) >= 256) return r;
  return 1;
}

hipError_t launch_maxpool(const float* in, float* out, int n, int split_out, hipStream_t st) {
  const int rows = pool_rows(n);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(n * (S2 / rows)), dim3(POOL_THREADS), 0, st, in, out, n, rows, split_out);
  return hipGetLastError();
}

}  // namespace se3tn
