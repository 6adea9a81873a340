// Fused Winograd F(2x2,3x3) for the 64 -> 64 channel 3x3 stride-1 convolutions of the trunk (convA2 / convB2 / convB3 of
// Se3TrackNet, se3_tracknet.py:59-64 via network_modules.py:86-120) on the 44 x 44 maps, for large batches.
//
// Why a different kernel.  The direct 64-channel kernels (conv3x3_slab_kernel<64,...>) sit at 0.70 of the f32 MFMA peak and cannot
// move: K = 576 per tile and 968 / 484 tiles on 256 workgroups = 3.78 / 1.89 rounds (profiles/EXPERIMENTS.md).  The plain Winograd route
// (transform pass -> batched GEMM -> transform pass, wino_mfma.hip) does not pay at 64 channels: the V / M planes are 4x the activation
// and the 16 GEMMs have K = 64 -- the passes alone cost what the direct kernel costs.  This kernel keeps EVERYTHING on the chip:
//
//   workgroup = one image x one 22 x 22-pixel quadrant x one group = 11 x 11 Winograd tiles (121 of 128 MFMA rows) x 64 couts
//               => 4 workgroups per image and group: 512 (A2|B2) or 256 (B3) workgroups = exactly 2 / 1 rounds on 256 CUs
//   per 32-channel chunk: the 24 x 24-pixel input patch is DMA'd into LDS once (73.7 KB)
//   per (chunk, frequency f = 4 i + j), 32 steps per tile:
//       V_f = (B^T d B)_ij  of the 121 tiles x 32 channels, computed from the patch by the VALU straight into an LDS operand tile
//             (B^T rows have two +-1 entries: V is a signed sum of 4 patch pixels)                          [double-buffered, 2 x 16 KB]
//       U_f chunk [64 couts x 32] DMA'd from the pre-transformed weights (U = G g G^T, float64 -> float32 once)   [double-buffered, 2 x 8 KB]
//       M_f = V_f U_f on v_mfma_f32_32x32x2_f32: 8 waves = 4 row blocks x 2 cout blocks, one 32 x 32 block each
//       Y[y][x] += A^T[y][i] A^T[x][j] M_f   in registers (A^T entries are 0 / +-1): the 16 M_f never exist at the same time
//   the step is software-pipelined inside every wave (MFMAs of step s | fold of step s - 1 | operand tile of step s + 1) and the 16
//   frequencies are unrolled, so that every B^T / A^T entry is a compile-time 0 / +-1: vector work is NOT hidden behind MFMAs on
//   gfx950 (profiles/r03_probe_mfma_valu.txt), the instruction count of a step is what the kernel's time follows (profiles/EXPERIMENTS.md items 23-24)
//   at the end Y is staged through LDS: + bias (+ residual), ReLU, pixel-major float4 stores of the 2 x 2 outputs per tile.
//
// MFMA work: 16 x 2 x (128 x 64 x 32) MACs per workgroup = 2.13x less than direct (incl. the 7 unused rows).  Float32 throughout; rounding as
// F(2x2) in wino_mfma.hip (1.5e-7 of the layer's largest activation).  Operand tiles use the same 128-byte rows, XOR swizzle and
// fragment layout as conv3x3_mfma.hip / wino_mfma.hip.
#include "mfma_common.h"

#ifndef W64_KNOCKOUT
#define W64_KNOCKOUT 0   // timing knock-outs (wrong results): bit 0 = no window reads from the LDS patch;
                         // wino64_regv_kernel: bit 1 = no U loads after the first, bit 2 = no window reads
#endif

namespace se3tn {

namespace {
constexpr int W64_TILES = 11;                 // Winograd tiles per quadrant edge
constexpr int W64_PATCH = 2 * W64_TILES + 2;  // 24 input pixels per edge
constexpr int W64_PATCH_FLOATS = W64_PATCH * W64_PATCH * 32;   // 18,432
constexpr int W64_V_FLOATS = 128 * 32, W64_U_FLOATS = 64 * 32;
// (U tiles fetched three steps ahead into four buffers with counted vmcnt waits + raw s_barrier: measured SLOWER, 0.152 vs 0.141 ms for the
// grouped launch -- as everywhere in this library the K-loop does not wait for data, it loses issue slots to the DMA)
constexpr int W64_UBUFS = 2;
constexpr int W64_LDS_FLOATS = W64_PATCH_FLOATS + 2 * W64_V_FLOATS + W64_UBUFS * W64_U_FLOATS;   // 30,720 floats = 122,880 B
constexpr int W64_YLDS_FLOATS = W64_TILES * W64_TILES * (4 * 64 + 4);                          // epilogue staging: 31,460 floats = 125,840 B
constexpr size_t W64_LDS_BYTES = sizeof(float) * (W64_YLDS_FLOATS > W64_LDS_FLOATS ? W64_YLDS_FLOATS : W64_LDS_FLOATS);
}  // namespace

// B^T row i of F(2x2): two non-zero entries (r0, +-1), (r1, +-1):  i=0: d0 - d2;  1: d1 + d2;  2: -d1 + d2;  3: d1 - d3
__device__ __forceinline__ void bt_row(int i, int& r0, int& r1, float& c0, float& c1) {
  r0 = i == 0 ? 0 : 1;
  r1 = i == 3 ? 3 : 2;
  c0 = i == 2 ? -1.f : 1.f;
  c1 = (i == 0 || i == 3) ? -1.f : 1.f;
}
// A^T column i of F(2x2): A^T = [[1,1,1,0],[0,1,-1,-1]]
__device__ __forceinline__ void at_col(int i, float& a0, float& a1) {
  a0 = i == 3 ? 0.f : 1.f;
  a1 = i == 0 ? 0.f : (i == 1 ? 1.f : -1.f);
}

// Vector adds with compile-time signs, as inline asm: (1) asm volatile stays where it is written -- plain C++ adds the compiler sinks
// below the step's barrier, all of them; (2) the signs ride on the operands' neg modifiers, no extra instruction.
// (Packed v_pk_add_f32 / v_pk_fma_f32 were measured and dropped: on gfx950 a packed float32 instruction costs exactly two scalar ones,
// profiles/r03_probe_mfma_valu.txt, and the library stays free of packed-f32 code, profiles/EXPERIMENTS.md items 13.)
template <bool NA, bool NB>
__device__ __forceinline__ float add_pm(const float a, const float b) {   // (+|-) a (+|-) b
  float r;
  if constexpr (!NA && !NB) asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (NA && !NB) asm volatile("v_sub_f32 %0, %2, %1" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (!NA && NB) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else asm volatile("v_sub_f32 %0, -%1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// y += K m for a compile-time K in {0, +1, -1}
template <int K>
__device__ __forceinline__ void acc_pm(float& y, const float m) {
  if constexpr (K > 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(y) : "v"(m));
  else if constexpr (K < 0) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(y) : "v"(m));
}
// the Toom-Cook matrices of F(2x2, 3x3) as compile-time tables
struct W64Tab {
  // B^T row i = sgn0 d[r0] + sgn1 d[r1]:  0: d0 - d2;  1: d1 + d2;  2: -d1 + d2;  3: d1 - d3
  static constexpr int bt_r0(int i) { return i == 0 ? 0 : 1; }
  static constexpr int bt_r1(int i) { return i == 3 ? 3 : 2; }
  static constexpr bool bt_n0(int i) { return i == 2; }
  static constexpr bool bt_n1(int i) { return i == 0 || i == 3; }
  // A^T = [[1,1,1,0],[0,1,-1,-1]]: entry (row y, column i)
  static constexpr int at(int y, int i) { return y == 0 ? (i == 3 ? 0 : 1) : (i == 0 ? 0 : (i == 1 ? 1 : -1)); }
  // float offset inside the LDS patch of window pixel (r, s): LDS column of patch column s is (s >> 1) + 12 (s & 1)
  static constexpr int off(int r, int s) { return (r * W64_PATCH + (s >> 1) + 12 * (s & 1)) * 32; }
};

struct Wino64Args {
  const float* in;    // padded NHWC [n][46][46][in_ld], channel offset of group 0 applied
  const float* U;     // group 0: [chunk 2][f 16][cout 64][32]
  const float* bias;  // group 0: [64]
  const float* res;   // residual (geometry of out) or nullptr
  float* out;         // padded NHWC
  int in_ld, res_ld, out_ld;
  int in_gs, res_gs, out_gs, bias_gs;
  long long u_gs;
  int n, groups;
};

template <int EPI>
__global__ __launch_bounds__(512) void wino64_fused_kernel(const Wino64Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;
  float* Vb = smem + W64_PATCH_FLOATS;
  float* Ub = Vb + 2 * W64_V_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = wid >> 1, cbk = wid & 1;      // this wave's 32 x 32 block: tiles 32 rb .. 32 rb + 31, couts 32 cbk .. 32 cbk + 31
  const int l31 = lane & 31, hh = lane >> 5;

  const int g = blockIdx.x % a.groups, rest = blockIdx.x / a.groups;
  const int quad = rest & 3, img = rest >> 2;
  const int y0 = (quad >> 1) * 2 * W64_TILES, x0 = (quad & 1) * 2 * W64_TILES;   // padded coordinates of the patch origin
  constexpr int HP = 46;
  const float* __restrict__ in = a.in + (size_t)g * a.in_gs + ((size_t)(img * HP + y0) * HP + x0) * a.in_ld;
  const float* __restrict__ Ug = a.U + (size_t)g * a.u_gs;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

  // ---- DMA geometry ------------------------------------------------------------------------------------------------------
  // patch piece id = tid + 512 j (j < 9): pixel id >> 3 (row-major in the 24 x 24 patch), 16-byte slot id & 7 -> LDS linear
  // U tile: thread -> row tid >> 3 (cout), LDS slot tid & 7 holds channel block (tid & 7) ^ ((row >> 1) & 7)
  const int ur = tid >> 3;
  const unsigned uvoff = (unsigned)((ur * 32 + (((tid & 7) ^ ((ur >> 1) & 7)) << 2)) * 4);

#define W64_ISSUE_PATCH(CH)                                                                                     \
  {                                                                                                            \
    const float* pb_ = in + (CH) * 32;                                                                          \
    _Pragma("unroll") for (int j_ = 0; j_ < 9; ++j_) {   /* 4608 pieces = 9 x 512 */                             \
      const int id_ = tid + 512 * j_, px_ = id_ >> 3, py_ = px_ / W64_PATCH, pxx_ = px_ - py_ * W64_PATCH;      \
      const int xs_ = pxx_ < 12 ? 2 * pxx_ : 2 * pxx_ - 23;   /* LDS column pxx_ holds patch column xs_ */        \
      glds16<0>(pb_, (unsigned)(((py_ * HP + xs_) * a.in_ld + (id_ & 7) * 4) * 4),                              \
                lds0 + (unsigned)((wid * 64 + 512 * j_) * 16));                                                 \
    }                                                                                                          \
  }
#define W64_ISSUE_U(CH, F, BUFI)                                                                                \
  glds16<0>(Ug + (size_t)((CH) * 16 + (F)) * W64_U_FLOATS, uvoff,                                              \
            lds0 + (unsigned)((W64_PATCH_FLOATS + 2 * W64_V_FLOATS + (BUFI) * W64_U_FLOATS + wid * 256) * 4));

  // ---- input transform tasks: (tile t, channel block cb) = id >> 3, id & 7 for id = tid, tid + 512.  Branch-free: the 56 ids past
  // tile 120 read tile 120's window and write operand rows 121..127, which the MFMA computes and nobody stores.
  int tpix[2], tdst[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int tt = (tid + 512 * j) >> 3, tcb = tid & 7;
    const int ttr = tt < W64_TILES * W64_TILES ? tt : W64_TILES * W64_TILES - 1;
    const int tty = ttr / W64_TILES, ttx = ttr - tty * W64_TILES;
    tpix[j] = ((2 * tty) * W64_PATCH + ttx) * 32 + tcb * 4;                 // float index of window pixel (0,0), this channel block
    tdst[j] = tt * 32 + ((tcb ^ ((tt >> 1) & 7)) << 2);                       // swizzled operand row t
  }
  // offsets / coefficients of frequency F's B^T entries (uniform: scalar registers)
#define W64_BT(F)                                                                                               \
  int r0_, r1_, s0_, s1_;                                                                                      \
  float cr0_, cr1_, cs0_, cs1_;                                                                                \
  bt_row((F) >> 2, r0_, r1_, cr0_, cr1_);                                                                      \
  bt_row((F) & 3, s0_, s1_, cs0_, cs1_);                                                                       \
  const int q0_ = (s0_ >> 1) + 12 * (s0_ & 1), q1_ = (s1_ >> 1) + 12 * (s1_ & 1);   /* LDS columns */           \
  const int o00_ = (r0_ * W64_PATCH + q0_) * 32, o01_ = (r0_ * W64_PATCH + q1_) * 32;                          \
  const int o10_ = (r1_ * W64_PATCH + q0_) * 32, o11_ = (r1_ * W64_PATCH + q1_) * 32;                          \
  const float c00_ = cr0_ * cs0_, c01_ = cr0_ * cs1_, c10_ = cr1_ * cs0_, c11_ = cr1_ * cs1_;
#define W64_TR_LOAD(D)                                                                                          \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                            \
    D[j_][0] = *reinterpret_cast<const float4*>(patch + tpix[j_] + o00_);                                       \
    D[j_][1] = *reinterpret_cast<const float4*>(patch + tpix[j_] + o01_);                                       \
    D[j_][2] = *reinterpret_cast<const float4*>(patch + tpix[j_] + o10_);                                       \
    D[j_][3] = *reinterpret_cast<const float4*>(patch + tpix[j_] + o11_);                                       \
  }
#define W64_TR_STORE(D, VBUF)                                                                                   \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                            \
    float4 v_;                                                                                                 \
    v_.x = c00_ * D[j_][0].x + c01_ * D[j_][1].x + c10_ * D[j_][2].x + c11_ * D[j_][3].x;                       \
    v_.y = c00_ * D[j_][0].y + c01_ * D[j_][1].y + c10_ * D[j_][2].y + c11_ * D[j_][3].y;                       \
    v_.z = c00_ * D[j_][0].z + c01_ * D[j_][1].z + c10_ * D[j_][2].z + c11_ * D[j_][3].z;                       \
    v_.w = c00_ * D[j_][0].w + c01_ * D[j_][1].w + c10_ * D[j_][2].w + c11_ * D[j_][3].w;                       \
    *reinterpret_cast<float4*>(Vb + (VBUF) * W64_V_FLOATS + tdst[j_]) = v_;                                     \
  }
#define W64_TRANSFORM(F, VBUF)                                                                                  \
  {                                                                                                            \
    W64_BT(F)                                                                                                  \
    float4 d_[2][4];                                                                                           \
    W64_TR_LOAD(d_)                                                                                            \
    W64_TR_STORE(d_, VBUF)                                                                                     \
  }

  // ---- fragment offsets of the four 8-k groups of a step (as wino_gemm_kernel) ------------------------------------------------------
  const int X = (l31 >> 1) & 7;
  const int lo = (hh ^ (X & 1)) * 4, xk = X >> 1;
  const int fo0 = ((0 ^ xk) << 3) + lo, fo1 = ((1 ^ xk) << 3) + lo, fo2 = ((2 ^ xk) << 3) + lo, fo3 = ((3 ^ xk) << 3) + lo;

  float Y[16][4];   // Y[e][o]: accumulator element e (cout 8 (e >> 2) + 4 hh + (e & 3)) of output pixel o = 2 y + x of the tile
#pragma unroll
  for (int e = 0; e < 16; ++e)
#pragma unroll
    for (int o = 0; o < 4; ++o) Y[e][o] = 0.f;

  // ---- prologue ------------------------------------------------------------------------------------------------------------
  W64_ISSUE_PATCH(0)
  W64_ISSUE_U(0, 0, 0)
  wait_dma_and_barrier();
  W64_TRANSFORM(0, 0)
  __syncthreads();

  // 32 steps = 2 chunks x 16 frequencies, software-pipelined inside every wave: while the matrix pipe works through the 16 MFMAs of
  // step s (one 32 x 32 block over the step's 32 channels), the vector ALU folds the products of step s - 1 into Y and builds this
  // thread's two pieces of the operand tile of step s + 1.  The three are independent inside a step, the step is one basic block,
  // and scheduling fences keep "1 MFMA, then 4 fold FMAs + 2-3 transform ops" sixteen times in that order.
  // (Knock-out measurements of the un-pipelined kernel -- matrix phase, then fold, then transform: the grouped launch takes 132 us;
  // 78 us without the MFMAs, 109 us without the transform, 123 us without the fold: nothing overlapped, all waves of a workgroup
  // meet at the step barrier in the same phase.)
  f32x16 accA, accB;
#pragma unroll
  for (int e = 0; e < 16; ++e) accB[e] = 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define W64_FOLD(PREV, F)                                                                                       \
  {                                                                                                            \
    float ay0_, ay1_, ax0_, ax1_;                                                                              \
    at_col((F) >> 2, ay0_, ay1_);                                                                              \
    at_col((F) & 3, ax0_, ax1_);                                                                               \
    const float k00_ = ay0_ * ax0_, k01_ = ay0_ * ax1_, k10_ = ay1_ * ax0_, k11_ = ay1_ * ax1_;                \
    _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) {                                                        \
      const float m_ = PREV[e_];                                                                               \
      Y[e_][0] += k00_ * m_; Y[e_][1] += k01_ * m_; Y[e_][2] += k10_ * m_; Y[e_][3] += k11_ * m_;              \
    }                                                                                                          \
  }
#define W64_MFMA(ACC)                                                                                        \
  ACC = q_ == 0 ? __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[0], pv_[0], zero16, 0, 0, 0)                         \
                : __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[q_], pv_[q_], ACC, 0, 0, 0);
  // One step with the frequency as a compile-time constant: the B^T / A^T entries (0, +-1) become adds / subtracts with neg modifiers
  // or vanish (36 of the 64 fold terms are non-zero), the window offsets become ds_read immediates, nothing is computed per step.
#define W64_STEP(F, ACC, PREV)                                                                                  \
  {                                                                                                            \
    constexpr int f_ = (F), buf = f_ & 1, fn_ = (f_ + 1) & 15, fp_ = (f_ + 15) & 15;                            \
    const bool last_ = ch == 1 && f_ == 15;                                                                    \
    if (!last_) W64_ISSUE_U(f_ == 15 ? 1 : ch, fn_, buf ^ 1)                                                    \
    if (f_ == 15 && ch == 0) W64_ISSUE_PATCH(1) /* the patch buffer is free: the transform of step 15 ran during step 14 */ \
    {                                                                                                          \
      const float* pP = Vb + buf * W64_V_FLOATS + (rb * 32 + l31) * 32;                                        \
      const float* pW = Ub + buf * W64_U_FLOATS + (cbk * 32 + l31) * 32;                                       \
      float pv_[16], wv_[16];                                                                                  \
      float d_[2][4][4], v_[2][4];   /* [task][window pixel | -][channel] */                                    \
      *reinterpret_cast<float4*>(pv_ + 0) = *reinterpret_cast<const float4*>(pP + fo0);                         \
      *reinterpret_cast<float4*>(wv_ + 0) = *reinterpret_cast<const float4*>(pW + fo0);                         \
      *reinterpret_cast<float4*>(pv_ + 4) = *reinterpret_cast<const float4*>(pP + fo1);                         \
      *reinterpret_cast<float4*>(wv_ + 4) = *reinterpret_cast<const float4*>(pW + fo1);                         \
      *reinterpret_cast<float4*>(pv_ + 8) = *reinterpret_cast<const float4*>(pP + fo2);                         \
      *reinterpret_cast<float4*>(wv_ + 8) = *reinterpret_cast<const float4*>(pW + fo2);                         \
      *reinterpret_cast<float4*>(pv_ + 12) = *reinterpret_cast<const float4*>(pP + fo3);                        \
      *reinterpret_cast<float4*>(wv_ + 12) = *reinterpret_cast<const float4*>(pW + fo3);                        \
      /* window pixels of frequency fn_ = (i, j): rows bt_r0/1(i), columns bt_r0/1(j) */                        \
      /* (step 15 of chunk 0 transforms a patch that is being replaced and the last step one nobody reads: harmless, branch-free) */ \
      constexpr int i_ = fn_ >> 2, j_ = fn_ & 3;                                                               \
      constexpr int oo_[4] = {W64Tab::off(W64Tab::bt_r0(i_), W64Tab::bt_r0(j_)), W64Tab::off(W64Tab::bt_r0(i_), W64Tab::bt_r1(j_)), \
                              W64Tab::off(W64Tab::bt_r1(i_), W64Tab::bt_r0(j_)), W64Tab::off(W64Tab::bt_r1(i_), W64Tab::bt_r1(j_))}; \
      constexpr bool nn_[4] = {W64Tab::bt_n0(i_) != W64Tab::bt_n0(j_), W64Tab::bt_n0(i_) != W64Tab::bt_n1(j_),  \
                               W64Tab::bt_n1(i_) != W64Tab::bt_n0(j_), W64Tab::bt_n1(i_) != W64Tab::bt_n1(j_)}; \
      _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_)                                                          \
        _Pragma("unroll") for (int w_ = 0; w_ < 4; ++w_) {                                                      \
          if (W64_KNOCKOUT & 1) *reinterpret_cast<float4*>(d_[t_][w_]) = make_float4(1.f, 2.f, 3.f, 4.f); /* timing only */ \
          else *reinterpret_cast<float4*>(d_[t_][w_]) = *reinterpret_cast<const float4*>(patch + tpix[t_] + oo_[w_]); \
        }                                                                                                      \
      constexpr int pi_ = fp_ >> 2, pj_ = fp_ & 3;   /* fold: products of the previous step (zeros before the first) */ \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {                                                       \
        W64_MFMA(ACC)                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (q_ < 8) {   /* first half of the matrix phase: fold (no LDS data needed) */                          \
          _Pragma("unroll") for (int e_ = 2 * q_; e_ < 2 * q_ + 2; ++e_) {                                      \
            const float m_ = PREV[e_];                                                                         \
            acc_pm<W64Tab::at(0, pi_) * W64Tab::at(0, pj_)>(Y[e_][0], m_);                                     \
            acc_pm<W64Tab::at(0, pi_) * W64Tab::at(1, pj_)>(Y[e_][1], m_);                                     \
            acc_pm<W64Tab::at(1, pi_) * W64Tab::at(0, pj_)>(Y[e_][2], m_);                                     \
            acc_pm<W64Tab::at(1, pi_) * W64Tab::at(1, pj_)>(Y[e_][3], m_);                                     \
          }                                                                                                    \
        } else {        /* then the operand tile of the next step (its window reads have landed by now) */       \
          const int t_ = (q_ - 8) >> 2, x_ = (q_ - 8) & 3;                                                     \
          v_[t_][x_] = add_pm<false, false>(add_pm<nn_[0], nn_[1]>(d_[t_][0][x_], d_[t_][1][x_]),              \
                                            add_pm<nn_[2], nn_[3]>(d_[t_][2][x_], d_[t_][3][x_]));             \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
      }                                                                                                        \
      *reinterpret_cast<float4*>(Vb + (buf ^ 1) * W64_V_FLOATS + tdst[0]) = *reinterpret_cast<float4*>(v_[0]); \
      *reinterpret_cast<float4*>(Vb + (buf ^ 1) * W64_V_FLOATS + tdst[1]) = *reinterpret_cast<float4*>(v_[1]); \
    }                                                                                                          \
    if (!last_) wait_dma_and_barrier();                                                                        \
    if (f_ == 15 && ch == 0) { /* chunk boundary: the second patch has landed (waited above): only now can its first tile be built */ \
      W64_TRANSFORM(0, 0)                                                                                      \
      __syncthreads();                                                                                         \
    }                                                                                                          \
  }
  for (int ch = 0; ch < 2; ++ch) {
    W64_STEP(0, accA, accB) W64_STEP(1, accB, accA) W64_STEP(2, accA, accB) W64_STEP(3, accB, accA)
    W64_STEP(4, accA, accB) W64_STEP(5, accB, accA) W64_STEP(6, accA, accB) W64_STEP(7, accB, accA)
    W64_STEP(8, accA, accB) W64_STEP(9, accB, accA) W64_STEP(10, accA, accB) W64_STEP(11, accB, accA)
    W64_STEP(12, accA, accB) W64_STEP(13, accB, accA) W64_STEP(14, accA, accB) W64_STEP(15, accB, accA)
  }
  W64_FOLD(accB, 15)    // step 31
#undef W64_STEP
#undef W64_MFMA
#undef W64_FOLD
#undef W64_ISSUE_PATCH
#undef W64_ISSUE_U
#undef W64_TRANSFORM
#undef W64_TR_LOAD
#undef W64_TR_STORE
#undef W64_BT

  // ---- epilogue.  A lane holds tile t x 16 couts x 4 pixels: stored straight from the registers that is 32 contiguous bytes per pixel and
  // lane pair.  So Y goes through the now idle LDS -- [tile][pixel o][64 couts], tile rows padded to 260 floats so that neighbouring
  // lanes hit different banks -- and comes back pixel-major: 16 lanes x float4 = the 256 contiguous bytes of one pixel's 64 couts, four
  // pixels per wave instruction; bias, residual (read the same way) and ReLU on the way out.  (Worth 1-3 us per launch.  The stores
  // themselves cost 5-14 us per round of workgroups -- knock-out -- because a full round ends at once and writes 32 MB with every CU
  // idle; a persistent variant that issues the next unit's first DMA before storing measured no better: 0.109 / 0.117 ms either way.)
  constexpr int YS = 4 * 64 + 4;
  __syncthreads();   // all operand tiles are consumed
  {
    const int t = rb * 32 + l31;
    if (t < W64_TILES * W64_TILES) {
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(smem + t * YS + o * 64 + cbk * 32 + q * 8 + hh * 4) =
              make_float4(Y[4 * q + 0][o], Y[4 * q + 1][o], Y[4 * q + 2][o], Y[4 * q + 3][o]);
    }
  }
  const float* __restrict__ bias = a.bias + (size_t)g * a.bias_gs;
  const float* __restrict__ res = (EPI == 1) ? a.res + (size_t)g * a.res_gs : nullptr;
  float* __restrict__ out = a.out + (size_t)g * a.out_gs;
  const int c4 = (tid & 15) * 4;
  // (The residual reads of the <1> instances requested before this barrier instead of inside the store loop -- -DW64_RES_HOIST=1 of
  // round 4 -- were measured in round 5: 38,162 / 37,983 against 38,153 / 38,098 pairs/s, no gain; the variant is gone.)
  __syncthreads();
  const float4 b = *reinterpret_cast<const float4*>(bias + c4);
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int p = (k * 512 + tid) >> 4;          // (tile, pixel) index: 484 of the 512 are real
    const int t = p >> 2, o = p & 3;
    if (t < W64_TILES * W64_TILES) {
      const int ty = t / W64_TILES, tx = t - ty * W64_TILES;
      const int oy = y0 + 2 * ty + (o >> 1) + 1, ox = x0 + 2 * tx + (o & 1) + 1;   // padded coordinates of the output pixel
      const size_t pix = (size_t)(img * HP + oy) * HP + ox;
      float4 v = *reinterpret_cast<const float4*>(smem + t * YS + o * 64 + c4);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      if (EPI == 1) {
        const float4 r = *reinterpret_cast<const float4*>(res + pix * a.res_ld + c4);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      *reinterpret_cast<float4*>(out + pix * a.out_ld + c4) = v;
    }
  }
}

#ifndef SE3TN_TRUNK_REGV
#define SE3TN_TRUNK_REGV 0
#endif
#if SE3TN_TRUNK_REGV
// =================================================================================================================================
// wino64_regv_kernel -- the same fused F(2x2,3x3) convolution WITHOUT a barrier in its main loop (round-4 EXPERIMENT, compiled only with
// -DSE3TN_TRUNK_REGV=1 and selected by SE3TN_TRUNK_KERNEL=2: parity tests green, SLOWER than wino64_fused_kernel -- 137 / 146 / 71 / 76 us
// against 110 / 116 / 56 / 60; knock-outs: without its U loads 101 / 108 / 53 / 56, without window reads as well 96 / 104 / 51 / 55 = this
// algorithm's floor (MFMA + vector adds + prologue / epilogue), which the shipped kernel is within 12-14 % of.  profiles/EXPERIMENTS.md item 32).
// What the first form pays per step besides its MFMAs and its vector adds is the workgroup-wide rendezvous: V_f goes through LDS (every
// thread builds two pieces of a tile all eight waves read), U_f arrives by LDS-DMA, so all waves meet at a barrier 32 times, wait for each
// other's skew, then all issue their fragment reads at once (measured: ~20 % of a step is neither matrix nor vector time).  Here:
//   * a wave owns 16 tiles x all 64 couts and runs v_mfma_f32_16x16x4_f32 (4 cout blocks = 4 independent accumulators): the B operand of a
//     lane is V of ITS tile for 8 channels -- exactly what that lane's two float4 window transforms produce.  V never leaves registers;
//   * the A operand (U_f: 64 couts x 32 channels = 8 KB per step) is read straight from global memory by every wave (8 x float4 per lane
//     per step, one step ahead): the 8 KB tile is shared by the eight waves through the L1 and by all workgroups through L2 (256 KB per
//     chunk for the whole launch);
//   * both 32-channel chunk patches are resident in LDS (2 x 73.7 KB), XOR-swizzled by LDS column (slot = channel block ^ (column & 7)) so
//     that the 16 lanes of a ds_read_b128 group -- 16 neighbouring tiles, two channel blocks -- cover all 64 banks;
//   * so after the prologue the only workgroup barrier is the one before chunk 1's patch is first read; the waves drift apart and the two
//     waves of a SIMD fill each other's vector phases with MFMAs by themselves.
// Arithmetic: the same products and sums per output as wino64_fused_kernel in the same frequency order; the k order inside a 32-channel
// chunk differs (4 (jq + 4 h) + c instead of ascending), so results agree to float32 rounding, not bitwise.
// =================================================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int EPI>
__global__ __launch_bounds__(512) void wino64_regv_kernel(const Wino64Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [chunk 2][24][24][8 slots][4]; the epilogue's staging afterwards
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, jq = lane >> 4;   // tile within the wave's 16 | k quarter = channel blocks jq, jq + 4

  const int g = blockIdx.x % a.groups, rest = blockIdx.x / a.groups;
  const int quad = rest & 3, img = rest >> 2;
  const int y0 = (quad >> 1) * 2 * W64_TILES, x0 = (quad & 1) * 2 * W64_TILES;
  constexpr int HP = 46;
  const float* __restrict__ in = a.in + (size_t)g * a.in_gs + ((size_t)(img * HP + y0) * HP + x0) * a.in_ld;
  const float* __restrict__ Ug = a.U + (size_t)g * a.u_gs;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

  // patch piece id = tid + 512 j (j < 9): LDS pixel id >> 3 (row-major 24 x 24, LDS column lc holds patch column lc < 12 ? 2 lc : 2 lc - 23),
  // LDS slot id & 7 holds channel block (id & 7) ^ (lc & 7)
#define W64R_ISSUE_PATCH(CH)                                                                                    \
  {                                                                                                            \
    const float* pb_ = in + (CH) * 32;                                                                          \
    _Pragma("unroll") for (int j_ = 0; j_ < 9; ++j_) {                                                          \
      const int id_ = tid + 512 * j_, px_ = id_ >> 3, py_ = px_ / W64_PATCH, lc_ = px_ - py_ * W64_PATCH;       \
      const int xs_ = lc_ < 12 ? 2 * lc_ : 2 * lc_ - 23;                                                        \
      glds16<0>(pb_, (unsigned)(((py_ * HP + xs_) * a.in_ld + (((id_ & 7) ^ (lc_ & 7)) << 2)) * 4),             \
                lds0 + (unsigned)(((CH) * W64_PATCH_FLOATS) * 4 + (wid * 64 + 512 * j_) * 16));                 \
    }                                                                                                          \
  }

  // this lane's tile (the 7 rows past tile 120 repeat tile 120: computed, never stored) and its window bases: float offsets of window
  // column s (0..3), channel-block half h, row 0; a window row adds r x 768 floats (24 pixels x 8 slots x 4)
  const int tt = wid * 16 + r16;
  const int ttr = tt < W64_TILES * W64_TILES ? tt : W64_TILES * W64_TILES - 1;
  const int tty = ttr / W64_TILES, ttx = ttr - tty * W64_TILES;
  int pb[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int lc = (s & 1) ? 12 + ttx + (s >> 1) : ttx + (s >> 1);
#pragma unroll
    for (int h = 0; h < 2; ++h) pb[s][h] = ((2 * tty * W64_PATCH + lc) * 8 + ((jq + 4 * h) ^ (lc & 7))) * 4;
  }
  const float* __restrict__ ul = Ug + r16 * 32 + jq * 4;   // + ((chunk 16 + f) 64 + 16 b) 32 + 16 h
#define W64R_LOAD_U(DST, CH, F)                                                                                 \
  _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_) _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)             \
      DST[b_][h_] = *reinterpret_cast<const float4*>(ul + ((CH) * 16 + (F)) * W64_U_FLOATS + b_ * 512 + h_ * 16);

  float Y[16][4];   // Y[e][o]: accumulator element e = 4 b + reg (cout 16 b + 4 jq + reg) of output pixel o = 2 y + x of the tile
#pragma unroll
  for (int e = 0; e < 16; ++e)
#pragma unroll
    for (int o = 0; o < 4; ++o) Y[e][o] = 0.f;

  float4 Ua[4][2], Ub[4][2];
  float Va[2][4], Vb[2][4];
  f32x4 accA[4], accB[4];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < 4; ++b) accB[b] = zero4;

  W64R_ISSUE_PATCH(0)
  W64R_LOAD_U(Ua, 0, 0)
  wait_dma_and_barrier();
  W64R_ISSUE_PATCH(1)   // lands under the first steps; every wave waits for its own pieces, the barrier before step 15 for the others'
  {   // V of step 0: frequency (0, 0) = (d0 - d2) x (d0 - d2)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float4 d00 = *reinterpret_cast<const float4*>(smem + pb[0][h]), d02 = *reinterpret_cast<const float4*>(smem + pb[2][h]);
      const float4 d20 = *reinterpret_cast<const float4*>(smem + pb[0][h] + 2 * 768), d22 = *reinterpret_cast<const float4*>(smem + pb[2][h] + 2 * 768);
      Va[h][0] = (d00.x - d02.x) - (d20.x - d22.x); Va[h][1] = (d00.y - d02.y) - (d20.y - d22.y);
      Va[h][2] = (d00.z - d02.z) - (d20.z - d22.z); Va[h][3] = (d00.w - d02.w) - (d20.w - d22.w);
    }
  }

  // One step, frequency compile-time: [U of the next step: 8 global float4] [window of the next step: 8 ds_read_b128]
  // 32 x { MFMA ; fold of the previous step's products (first 16) | transform of the next step's V (next 8) }
#define W64R_STEP(F, UC, UN, VC, VN, ACC, PREV)                                                                 \
  {                                                                                                            \
    constexpr int f_ = (F), fn_ = (f_ + 1) & 15, fp_ = (f_ + 15) & 15;                                          \
    const bool last_ = ch == 1 && f_ == 15;                                                                    \
    const int chn_ = f_ == 15 ? 1 : ch;                                                                        \
    if (f_ == 15 && ch == 0) wait_dma_and_barrier();   /* chunk 1's patch: every wave's pieces have landed */     \
    if (!last_ && !(W64_KNOCKOUT & 2)) W64R_LOAD_U(UN, chn_, fn_)                                               \
    if (W64_KNOCKOUT & 2) { _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_) { UN[b_][0] = UC[b_][1]; UN[b_][1] = UC[b_][0]; } } /* timing only */ \
    constexpr int i_ = fn_ >> 2, j_ = fn_ & 3;                                                                 \
    constexpr int ra_ = W64Tab::bt_r0(i_) * 768, rb_ = W64Tab::bt_r1(i_) * 768;                                \
    constexpr int sa_ = W64Tab::bt_r0(j_), sb_ = W64Tab::bt_r1(j_);                                            \
    constexpr bool nn_[4] = {W64Tab::bt_n0(i_) != W64Tab::bt_n0(j_), W64Tab::bt_n0(i_) != W64Tab::bt_n1(j_),    \
                             W64Tab::bt_n1(i_) != W64Tab::bt_n0(j_), W64Tab::bt_n1(i_) != W64Tab::bt_n1(j_)};   \
    const float* pn_ = smem + chn_ * W64_PATCH_FLOATS;                                                          \
    float d_[2][4][4];                                                                                         \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                          \
      if (W64_KNOCKOUT & 4) { _Pragma("unroll") for (int w_ = 0; w_ < 4; ++w_) *reinterpret_cast<float4*>(d_[h_][w_]) = make_float4(1.f, 2.f, 3.f, 4.f); continue; } \
      *reinterpret_cast<float4*>(d_[h_][0]) = *reinterpret_cast<const float4*>(pn_ + pb[sa_][h_] + ra_);        \
      *reinterpret_cast<float4*>(d_[h_][1]) = *reinterpret_cast<const float4*>(pn_ + pb[sb_][h_] + ra_);        \
      *reinterpret_cast<float4*>(d_[h_][2]) = *reinterpret_cast<const float4*>(pn_ + pb[sa_][h_] + rb_);        \
      *reinterpret_cast<float4*>(d_[h_][3]) = *reinterpret_cast<const float4*>(pn_ + pb[sb_][h_] + rb_);        \
    }                                                                                                          \
    constexpr int pi_ = fp_ >> 2, pj_ = fp_ & 3;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    _Pragma("unroll") for (int m_ = 0; m_ < 32; ++m_) {                                                         \
      const int q_ = m_ >> 2, b_ = m_ & 3, h_ = q_ >> 2, k_ = q_ & 3;                                          \
      const float ua_ = k_ == 0 ? UC[b_][h_].x : k_ == 1 ? UC[b_][h_].y : k_ == 2 ? UC[b_][h_].z : UC[b_][h_].w; \
      ACC[b_] = q_ == 0 ? __builtin_amdgcn_mfma_f32_16x16x4f32(ua_, VC[h_][k_], zero4, 0, 0, 0)                 \
                        : __builtin_amdgcn_mfma_f32_16x16x4f32(ua_, VC[h_][k_], ACC[b_], 0, 0, 0);              \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      if (m_ < 16) {   /* fold element m_ of the previous step's products */                                     \
        const float m__ = PREV[m_ >> 2][m_ & 3];                                                                \
        acc_pm<W64Tab::at(0, pi_) * W64Tab::at(0, pj_)>(Y[m_][0], m__);                                        \
        acc_pm<W64Tab::at(0, pi_) * W64Tab::at(1, pj_)>(Y[m_][1], m__);                                        \
        acc_pm<W64Tab::at(1, pi_) * W64Tab::at(0, pj_)>(Y[m_][2], m__);                                        \
        acc_pm<W64Tab::at(1, pi_) * W64Tab::at(1, pj_)>(Y[m_][3], m__);                                        \
      } else if (m_ < 24) {   /* the next step's V: 8 values (2 channel blocks x 4), its window reads have landed */ \
        const int t_ = (m_ - 16) >> 2, x_ = (m_ - 16) & 3;                                                     \
        VN[t_][x_] = add_pm<false, false>(add_pm<nn_[0], nn_[1]>(d_[t_][0][x_], d_[t_][1][x_]),                \
                                          add_pm<nn_[2], nn_[3]>(d_[t_][2][x_], d_[t_][3][x_]));               \
      }                                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
    }                                                                                                          \
  }
  for (int ch = 0; ch < 2; ++ch) {
    W64R_STEP(0, Ua, Ub, Va, Vb, accA, accB) W64R_STEP(1, Ub, Ua, Vb, Va, accB, accA)
    W64R_STEP(2, Ua, Ub, Va, Vb, accA, accB) W64R_STEP(3, Ub, Ua, Vb, Va, accB, accA)
    W64R_STEP(4, Ua, Ub, Va, Vb, accA, accB) W64R_STEP(5, Ub, Ua, Vb, Va, accB, accA)
    W64R_STEP(6, Ua, Ub, Va, Vb, accA, accB) W64R_STEP(7, Ub, Ua, Vb, Va, accB, accA)
    W64R_STEP(8, Ua, Ub, Va, Vb, accA, accB) W64R_STEP(9, Ub, Ua, Vb, Va, accB, accA)
    W64R_STEP(10, Ua, Ub, Va, Vb, accA, accB) W64R_STEP(11, Ub, Ua, Vb, Va, accB, accA)
    W64R_STEP(12, Ua, Ub, Va, Vb, accA, accB) W64R_STEP(13, Ub, Ua, Vb, Va, accB, accA)
    W64R_STEP(14, Ua, Ub, Va, Vb, accA, accB) W64R_STEP(15, Ub, Ua, Vb, Va, accB, accA)
  }
  {   // fold of step 31 (frequency 15 = (3, 3)): A^T column 3 = (0, -1) x (0, -1): only Y[.][3] += m
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float m = accB[e >> 2][e & 3];
      acc_pm<W64Tab::at(0, 3) * W64Tab::at(0, 3)>(Y[e][0], m);
      acc_pm<W64Tab::at(0, 3) * W64Tab::at(1, 3)>(Y[e][1], m);
      acc_pm<W64Tab::at(1, 3) * W64Tab::at(0, 3)>(Y[e][2], m);
      acc_pm<W64Tab::at(1, 3) * W64Tab::at(1, 3)>(Y[e][3], m);
    }
  }
#undef W64R_STEP
#undef W64R_LOAD_U
#undef W64R_ISSUE_PATCH

  // ---- epilogue: as wino64_fused_kernel (Y through LDS, pixel-major float4 stores); a lane holds tile tt x couts 16 b + 4 jq + 0..3
  constexpr int YS = 4 * 64 + 4;
  __syncthreads();   // every wave is done with the patches
  if (tt < W64_TILES * W64_TILES) {
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        *reinterpret_cast<float4*>(smem + tt * YS + o * 64 + b * 16 + jq * 4) =
            make_float4(Y[4 * b + 0][o], Y[4 * b + 1][o], Y[4 * b + 2][o], Y[4 * b + 3][o]);
  }
  __syncthreads();
  const float* __restrict__ bias = a.bias + (size_t)g * a.bias_gs;
  const float* __restrict__ res = (EPI == 1) ? a.res + (size_t)g * a.res_gs : nullptr;
  float* __restrict__ out = a.out + (size_t)g * a.out_gs;
  const int c4 = (tid & 15) * 4;
  const float4 bq = *reinterpret_cast<const float4*>(bias + c4);
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int p = (k * 512 + tid) >> 4;
    const int t = p >> 2, o = p & 3;
    if (t < W64_TILES * W64_TILES) {
      const int ty = t / W64_TILES, tx = t - ty * W64_TILES;
      const int oy = y0 + 2 * ty + (o >> 1) + 1, ox = x0 + 2 * tx + (o & 1) + 1;
      const size_t pix = (size_t)(img * HP + oy) * HP + ox;
      float4 v = *reinterpret_cast<const float4*>(smem + t * YS + o * 64 + c4);
      v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
      if (EPI == 1) {
        const float4 r = *reinterpret_cast<const float4*>(res + pix * a.res_ld + c4);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      *reinterpret_cast<float4*>(out + pix * a.out_ld + c4) = v;
    }
  }
}

#endif  // SE3TN_TRUNK_REGV

// in / res / out: padded NHWC tensors of the 44 x 44 maps; U: F(2x2) planes [chunk][16][64][32] per group (launch_wino_weights, m = 2)
hipError_t launch_wino64(const float* in, int in_ld, int in_gs, const float* U, long long u_gs, const float* bias, int bias_gs,
                         const float* res, int res_ld, int res_gs, float* out, int out_ld, int out_gs, int n, int groups, int epi,
                         int variant, hipStream_t st) {
  Wino64Args a{};
  a.in = in; a.U = U; a.bias = bias; a.res = res; a.out = out;
  a.in_ld = in_ld; a.res_ld = res_ld; a.out_ld = out_ld;
  a.in_gs = in_gs; a.res_gs = res_gs; a.out_gs = out_gs; a.bias_gs = bias_gs; a.u_gs = u_gs;
  a.n = n; a.groups = groups;
  // variant 1: wino64_fused_kernel (V through LDS, barrier per step); 2: wino64_regv_kernel (V in registers, no barrier in the loop)
  constexpr size_t lds2 = sizeof(float) * 2 * W64_PATCH_FLOATS;   // 147,456 B >= the epilogue's 125,840
  static_assert(lds2 >= sizeof(float) * W64_YLDS_FLOATS, "the epilogue staging must fit the patch buffers");
  static PerDeviceOnce attr[4];
#if !SE3TN_TRUNK_REGV
  variant = 1;   // the register-V experiment is not part of the default build
#endif
  const void* kern = epi == 1 ? reinterpret_cast<const void*>(wino64_fused_kernel<1>) : reinterpret_cast<const void*>(wino64_fused_kernel<0>);
#if SE3TN_TRUNK_REGV
  if (variant == 2) kern = epi == 1 ? reinterpret_cast<const void*>(wino64_regv_kernel<1>) : reinterpret_cast<const void*>(wino64_regv_kernel<0>);
#endif
  const size_t lds = variant == 2 ? lds2 : W64_LDS_BYTES;
  bool* done = attr[(variant == 2 ? 2 : 0) + (epi == 1 ? 1 : 0)].current();
  if (!done || !*done) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (done) *done = true;
  }
  const dim3 grid(n * 4 * groups);
#if SE3TN_TRUNK_REGV
  if (variant == 2) {
    if (epi == 1) hipLaunchKernelGGL(wino64_regv_kernel<1>, grid, dim3(512), lds, st, a);
    else hipLaunchKernelGGL(wino64_regv_kernel<0>, grid, dim3(512), lds, st, a);
    return hipGetLastError();
  }
#endif
  if (epi == 1) hipLaunchKernelGGL(wino64_fused_kernel<1>, grid, dim3(512), lds, st, a);
  else hipLaunchKernelGGL(wino64_fused_kernel<0>, grid, dim3(512), lds, st, a);
  return hipGetLastError();
}

}  // namespace se3tn
