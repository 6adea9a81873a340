// Winograd F(m x m, 3x3), m = 2 | 4, form of the stride-1 3x3 convolutions with 256 / 512 input
// channels (convAB2.conv1/.conv2 and trans|rot conv2.conv1/.conv2 of Se3TrackNet, se3_tracknet.py:68-76
// via network_modules.py:86-120 ResnetBasicBlock) for large batches.  Still float32 end to end, still on
// v_mfma_f32_32x32x2_f32 -- but (m+2)^2 multiplies per m x m output tile instead of 9 m^2:
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A            (Lavin & Gray 2016, cross-correlation form;
//                                                        Toom-Cook matrices of the points listed below)
//     m = 2: points 0, 1, -1, inf         16 products / 4 outputs   (2.25x fewer than direct)
//     m = 4: points 0, 1, -1, 1/2, -2, inf  36 products / 16 outputs (4x fewer)
//
// with the channel sum done per "frequency" f = (m+2) i + j as a GEMM  M_f[T x Cout] = V_f[T x C] U_f[C x Cout].
// A 22x22 map tiles as 11x11 (m=2) or 6x6 (m=4, 24 rows computed, 2 dropped); 11x11 as 6x6 / 3x3.
// Net MFMA work vs direct at batch 64: m=2 1.89-2.25x less, m=4 3.36x less.
// Three launches per convolution:
//   wino_input_kernel   d ((m+2)^2 windows of the zero-bordered NHWC tensor) -> V[g][f][T][C]
//   wino_gemm_kernel    (m+2)^2 x groups independent GEMMs; {128|96} x 128 tiles, 4 waves, 2 workgroups
//                       per CU, operands LDS-DMA'd with the XOR swizzle / fragment layout of conv3x3_mfma.hip
//   wino_output_kernel  A^T M A + folded-BN bias (+ residual) + ReLU -> interior of the padded output
// U = G g G^T is derived on the device in float64 (rounded once) from the packed direct weights when the
// blob is uploaded / bound (wino_weight_kernel): the blob format and the host packer do not change.
// The transform passes are pure HBM / Infinity-Cache streams.
//
// Rounding: the transforms amplify float32 rounding (measured rms error of one 512-channel layer vs a
// float64 convolution, relative to the layer's largest activation: direct 5e-8, m=2 1.5e-7, m=4 6e-7);
// after the four layers that use it the logits move by ~1e-6, two orders inside the 1e-4 tolerance
// (tests/test_gpu_parity.py::test_winograd_*).
#include "mfma_common.h"
#include "pose_device.h"

#ifndef WINO4_VEC
#define WINO4_VEC 2  // channels per thread of the F(4x4) transform kernels (2: 8-byte, 4: 16-byte accesses; measured equal)
#endif
#ifndef WINO_MID_CS_AB
#define WINO_MID_CS_AB 32  // channels per workgroup of wino_mid_kernel on the 22 x 22 maps (LDS 26 x 26 x CS floats)
#endif
#ifndef WINO_MID_CS_H
#define WINO_MID_CS_H 32   // ... on the 11 x 11 maps (LDS 14 x 14 x CS floats)
#endif

#ifdef SE3TN_GEMMP_TRACE
// diagnostic build only (scripts/gemmp_trace.py): waves of workgroup 0 of the last wino_gemmp_kernel launch stamp every K-step
// [wave 8][k-step 64][5 stamps]: step start | DMA issued | MFMAs issued | vmcnt(0) passed | barrier passed   (s_memtime, shader clock)
__device__ unsigned long long se3tn_gemmp_trace[8 * 64 * 5];
extern "C" int se3tn_debug_gemmp_trace(void* host_out, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(se3tn_gemmp_trace), bytes, 0, hipMemcpyDeviceToHost);
}
#define GP_STAMP(K, I) if (blockIdx.x == 0 && lane == 0 && (K) < 64) ::se3tn_gemmp_trace[(wid * 64 + (K)) * 5 + (I)] = __builtin_amdgcn_s_memtime();
#else
#define GP_STAMP(K, I)
#endif
#ifdef SE3TN_WG_TRACE
// diagnostic build only (scripts/wg_trace.py): every workgroup of wino_gemm_kernel records where and when it ran
__device__ unsigned long long se3tn_wg_trace[8 * 4096];
extern "C" int se3tn_debug_wg_trace(void* host_out, size_t bytes, int clear) {
  if (clear) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(se3tn_wg_trace)) != hipSuccess) return 1;
    return (int)hipMemset(p, 0, sizeof(unsigned long long) * 8 * 4096);
  }
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(se3tn_wg_trace), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

namespace se3tn {

// ---- Toom-Cook matrices ------------------------------------------------------------------------------
template <int M>
__host__ __device__ __forceinline__ constexpr float wino_bt(int i, int j) {  // B^T [(M+2) x (M+2)]
  constexpr float t2[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
  constexpr float t4[6][6] = {{1, -1.5f, -2, 1.5f, 1, 0}, {0, -1, 0.5f, 2.5f, 1, 0}, {0, 1, -2.5f, 0.5f, 1, 0},
                              {0, -2, -1, 2, 1, 0},       {0, 0.5f, -1, -0.5f, 1, 0}, {0, 1, -1.5f, -2, 1.5f, 1}};
  // F(6x6): points {0, 1, -1, 2, -2, 1/2, -1/2, inf} (scripts/study_winograd_rounding.py derives all three tables in exact
  // rationals and reproduces t4 entry by entry; tests/test_winograd_matrices.py checks the identity on the tables parsed from here)
  constexpr float t6[8][8] = {{-1, 0, 5.25f, 0, -5.25f, 0, 1, 0},        {0, 1, 1, -4.25f, -4.25f, 1, 1, 0},
                              {0, -1, 1, 4.25f, -4.25f, -1, 1, 0},       {0, 0.5f, 0.25f, -2.5f, -1.25f, 2, 1, 0},
                              {0, -0.5f, 0.25f, 2.5f, -1.25f, -2, 1, 0}, {0, 2, 4, -2.5f, -5, 0.5f, 1, 0},
                              {0, -2, 4, 2.5f, -5, -0.5f, 1, 0},         {0, -1, 0, 5.25f, 0, -5.25f, 0, 1}};
  return M == 2 ? t2[i & 3][j & 3] : M == 4 ? t4[i % 6][j % 6] : t6[i & 7][j & 7];
}
template <int M>
__host__ __device__ __forceinline__ constexpr float wino_at(int i, int j) {  // A^T [M x (M+2)]
  constexpr float t2[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
  constexpr float t4[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 0.5f, -2, 0}, {0, 1, 1, 0.25f, 4, 0}, {0, 1, -1, 0.125f, -8, 1}};
  constexpr float t6[6][8] = {{1, 1, 1, 1, 1, 1, 1, 0},
                              {0, 1, -1, 2, -2, 0.5f, -0.5f, 0},
                              {0, 1, 1, 4, 4, 0.25f, 0.25f, 0},
                              {0, 1, -1, 8, -8, 0.125f, -0.125f, 0},
                              {0, 1, 1, 16, 16, 0.0625f, 0.0625f, 0},
                              {0, 1, -1, 32, -32, 0.03125f, -0.03125f, 1}};
  return M == 2 ? t2[i & 1][j & 3] : M == 4 ? t4[i & 3][j % 6] : t6[i % 6][j & 7];
}
template <int M>
__host__ __device__ __forceinline__ constexpr double wino_g(int i, int j) {  // G [(M+2) x 3]
  constexpr double t2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  constexpr double t4[6][3] = {{1, 0, 0},
                               {1.0 / 3, 1.0 / 3, 1.0 / 3},
                               {-1.0 / 3, 1.0 / 3, -1.0 / 3},
                               {-16.0 / 15, -8.0 / 15, -4.0 / 15},
                               {1.0 / 15, -2.0 / 15, 4.0 / 15},
                               {0, 0, 1}};
  constexpr double t6[8][3] = {{-1, 0, 0},
                               {-2.0 / 9, -2.0 / 9, -2.0 / 9},
                               {-2.0 / 9, 2.0 / 9, -2.0 / 9},
                               {1.0 / 90, 1.0 / 45, 2.0 / 45},
                               {1.0 / 90, -1.0 / 45, 2.0 / 45},
                               {32.0 / 45, 16.0 / 45, 8.0 / 45},
                               {32.0 / 45, -16.0 / 45, 8.0 / 45},
                               {0, 0, 1}};
  return M == 2 ? t2[i & 3][j] : M == 4 ? t4[i % 6][j] : t6[i & 7][j];
}

template <int VEC> struct VecT;
template <> struct VecT<4> { typedef float4 type; };
template <> struct VecT<2> { typedef float2 type; };

// acc += c * x with the multiplications by 0 / +-1 folded away (c is a compile-time constant after unrolling)
__device__ __forceinline__ void axpy(float& acc, float c, float x, bool& first) {
  if (c == 0.f) return;
  const float t = c == 1.f ? x : c == -1.f ? -x : c * x;
  acc = first ? t : acc + t;
  first = false;
}

// ---- f16x3 mode (SP != 0): activations / V as split rows (mfma_common.h), VEC consecutive channels c..c+VEC-1 --------
// element (row, c) of a tensor with `ld` floats per row: hi halves at byte (row ld) 4 + (c >> 5) 128 + (c & 31) 2, lo 64 B on
template <int VEC>
__device__ __forceinline__ void load_split_vec(const float* base, size_t row, int ld, int c, float (&o)[VEC]) {
  const unsigned char* p_ = reinterpret_cast<const unsigned char*>(base) + split_byte_off(row, ld, c);
  typedef _Float16 hv __attribute__((ext_vector_type(VEC)));
  const hv h = *reinterpret_cast<const hv*>(p_);
  const hv l = *reinterpret_cast<const hv*>(p_ + 64);
#pragma unroll
  for (int e = 0; e < VEC; ++e) o[e] = (float)h[e] + (float)l[e];
}
template <int VEC>
__device__ __forceinline__ bool store_split_vec(float* base, size_t row, int ld, int c, const float (&v)[VEC]) {
  unsigned char* p_ = reinterpret_cast<unsigned char*>(base) + split_byte_off(row, ld, c);
  typedef _Float16 hv __attribute__((ext_vector_type(VEC)));
  hv h, l;
  bool bad = false;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    h[e] = (_Float16)v[e];
    l[e] = (_Float16)(v[e] - (float)h[e]);
    bad |= !(fabsf(v[e]) <= 65000.f);
  }
  *reinterpret_cast<hv*>(p_) = h;
  *reinterpret_cast<hv*>(p_ + 64) = l;
  return bad;
}

// decode a flat thread index into (tile, vector of channels) and the tile into (image, ty, tx)
struct TileId { int t, cv, n, ty, tx; };
__device__ __forceinline__ TileId tile_of(int idx, int cvn, int th, int tw) {
  TileId r;
  r.cv = idx % cvn; r.t = idx / cvn;
  const int tpi = th * tw;
  r.n = r.t / tpi;
  const int rem = r.t - r.n * tpi;
  r.ty = rem / tw; r.tx = rem - r.ty * tw;
  return r;
}

// one thread: (group, tile t, VEC channels)
template <int M, int VEC, int SP = 0>
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoArgs a) {
  constexpr int N = M + 2;
  typedef typename VecT<VEC>::type vec;
  const int cvn = a.C / VEC;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.T * cvn) return;
  const int g = blockIdx.y;
  const TileId id = tile_of(idx, cvn, a.th, a.tw);
  const int Hp = a.H + 2, Wp = a.W + 2;
  const float* __restrict__ src = a.in + (size_t)g * a.in_gs + (SP ? 0 : id.cv * VEC);
  // padded rows M ty .. M ty + M + 1 = input rows M ty - 1 .. M ty + M; windows of the last tile row /
  // column may reach past the border: those entries only feed dropped outputs
  float bt[N][N][VEC];  // B^T d
  if constexpr (M == 6) {
    // 8 x 8 windows: d and B^T d together would be 256 registers per thread at VEC = 2, so the window is read one COLUMN at a time
    // (the row transform of column s needs column s only); same operations in the same order as the generic form below
    static_assert(SP == 0, "F(6x6) is a float32 path");
#pragma unroll
    for (int s = 0; s < N; ++s) {
      float dc[N][VEC];
      const int X = M * id.tx + s;
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const int Y = M * id.ty + r;
        vec v;
        if (Y < Hp && X < Wp) v = *reinterpret_cast<const vec*>(src + (size_t)((id.n * Hp + Y) * Wp + X) * a.in_ld);
        else __builtin_memset(&v, 0, sizeof(v));
        __builtin_memcpy(dc[r], &v, sizeof(v));
      }
#pragma unroll
      for (int i = 0; i < N; ++i)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float acc = 0.f;
          bool first = true;
#pragma unroll
          for (int r = 0; r < N; ++r) axpy(acc, wino_bt<M>(i, r), dc[r][e], first);
          bt[i][s][e] = acc;
        }
    }
  } else {
  float d[N][N][VEC];
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int s = 0; s < N; ++s) {
      const int Y = M * id.ty + r, X = M * id.tx + s;
      if (SP) {
        if (Y < Hp && X < Wp) load_split_vec<VEC>(src, (size_t)((id.n * Hp + Y) * Wp + X), a.in_ld, id.cv * VEC, d[r][s]);
        else for (int e = 0; e < VEC; ++e) d[r][s][e] = 0.f;
        continue;
      }
      vec v;
      if (Y < Hp && X < Wp) v = *reinterpret_cast<const vec*>(src + (size_t)((id.n * Hp + Y) * Wp + X) * a.in_ld);
      else __builtin_memset(&v, 0, sizeof(v));
      __builtin_memcpy(d[r][s], &v, sizeof(v));
    }
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int s = 0; s < N; ++s)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float acc = 0.f;
        bool first = true;
#pragma unroll
        for (int r = 0; r < N; ++r) axpy(acc, wino_bt<M>(i, r), d[r][s][e], first);
        bt[i][s][e] = acc;
      }
  }
  float* __restrict__ dst = a.V + ((size_t)g * a.nf * a.T + id.t) * a.C + (SP ? 0 : id.cv * VEC);
  const size_t fs = (size_t)a.T * a.C;  // floats per frequency plane
  bool bad = false;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float acc = 0.f;
        bool first = true;
#pragma unroll
        for (int s = 0; s < N; ++s) axpy(acc, wino_bt<M>(j, s), bt[i][s][e], first);
        o[e] = acc;
      }
      if (SP) {   // V plane f as split rows of C floats: the GEMM's f16 hi | lo operand
        bad |= store_split_vec<VEC>(dst + (size_t)(N * i + j) * fs, 0, a.C, id.cv * VEC, o);
        continue;
      }
      vec v;
      __builtin_memcpy(&v, o, sizeof(v));
      *reinterpret_cast<vec*>(dst + (size_t)(N * i + j) * fs) = v;
    }
  if (SP && bad) atomicOr(a.overflow, 1);
}

// one thread: (group, tile t, VEC couts)
template <int M, int VEC, int EPI, int SP = 0>
__global__ __launch_bounds__(256) void wino_output_kernel(const WinoArgs a) {
  constexpr int N = M + 2;
  typedef typename VecT<VEC>::type vec;
  const int cvn = a.Cout / VEC;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.T * cvn) return;
  const int g = blockIdx.y;
  const TileId id = tile_of(idx, cvn, a.th, a.tw);
  const int Hp = a.H + 2, Wp = a.W + 2;
  const float* __restrict__ src = a.Mw + ((size_t)g * a.nf * a.T + id.t) * a.Cout + id.cv * VEC;
  const size_t fs = (size_t)a.T * a.Cout;
  // The residual of the whole tile is requested BEFORE the planes: inside the per-pixel loop below (a branch and a store per
  // pixel) each of the M x M reads waited out its own memory latency, one after the other (EXPERIMENTS item 38).  Pixels past
  // the map edge read a clamped (valid) address and are dropped with their outputs.
  float rr[M][M][VEC];
  if constexpr (EPI == 1 && SP == 0) {
    const float* __restrict__ res0 = a.res + (size_t)g * a.res_gs + id.cv * VEC;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j < M; ++j) {
        int oy = M * id.ty + i, ox = M * id.tx + j;
        oy = oy < a.H ? oy : a.H - 1; ox = ox < a.W ? ox : a.W - 1;
        const vec v = *reinterpret_cast<const vec*>(res0 + (size_t)((id.n * Hp + oy + 1) * Wp + ox + 1) * a.res_ld);
        __builtin_memcpy(rr[i][j], &v, sizeof(v));
      }
  }
  float u[M][N][VEC];  // A^T m, one column s at a time
#pragma unroll
  for (int s = 0; s < N; ++s) {
    float m[N][VEC];
#pragma unroll
    for (int r = 0; r < N; ++r) {
      const vec v = *reinterpret_cast<const vec*>(src + (size_t)(N * r + s) * fs);
      __builtin_memcpy(m[r], &v, sizeof(v));
    }
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float acc = 0.f;
        bool first = true;
#pragma unroll
        for (int r = 0; r < N; ++r) axpy(acc, wino_at<M>(i, r), m[r][e], first);
        u[i][s][e] = acc;
      }
  }
  float b[VEC];
  {
    const vec v = *reinterpret_cast<const vec*>(a.bias + (size_t)g * a.bias_gs + id.cv * VEC);
    __builtin_memcpy(b, &v, sizeof(v));
  }
  const float* __restrict__ res = (EPI == 1) ? a.res + (size_t)g * a.res_gs + (SP ? 0 : id.cv * VEC) : nullptr;
  float* __restrict__ out = a.out + (size_t)g * a.out_gs + (SP ? 0 : id.cv * VEC);
  bool bad = false;
  // every read of this thread has landed before the first store: inside the per-pixel branches the compiler cannot count the
  // stores in flight, so a later first use of a loaded register would become `s_waitcnt vmcnt(0)` = wait for the previous store
  if constexpr (SP == 0) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int oy = M * id.ty + i, ox = M * id.tx + j;
      if (oy >= a.H || ox >= a.W) continue;
      const size_t pix = (size_t)((id.n * Hp + oy + 1) * Wp + ox + 1);
      float r[VEC];
      if (EPI == 1) {
        if (SP) {
          load_split_vec<VEC>(res, pix, a.res_ld, id.cv * VEC, r);
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) r[e] = rr[i][j][e];
        }
      }
      float o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float acc = 0.f;
        bool first = true;
#pragma unroll
        for (int s = 0; s < N; ++s) axpy(acc, wino_at<M>(j, s), u[i][s][e], first);
        acc += b[e];
        if (EPI == 1) acc += r[e];
        o[e] = fmaxf(acc, 0.f);
      }
      if (SP) {
        bad |= store_split_vec<VEC>(out, pix, a.out_ld, id.cv * VEC, o);
        continue;
      }
      vec v;
      __builtin_memcpy(&v, o, sizeof(v));
      *reinterpret_cast<vec*>(out + pix * a.out_ld) = v;
    }
  if (SP && bad) atomicOr(a.overflow, 1);
}

// packed [chunk][9][cout][32] -> U [chunk][(M+2)^2][cout][32], one thread per (chunk, cout, k)
template <int M>
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ packed, float* __restrict__ U,
                                                          int cout, int total) {
  constexpr int N = M + 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int k = idx & 31, co = (idx >> 5) % cout, ch = idx / (32 * cout);
  double g[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) g[r][s] = (double)packed[((size_t)(ch * 9 + r * 3 + s) * cout + co) * 32 + k];
  double gg[N][3];  // G g
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int s = 0; s < 3; ++s) gg[i][s] = wino_g<M>(i, 0) * g[0][s] + wino_g<M>(i, 1) * g[1][s] + wino_g<M>(i, 2) * g[2][s];
  const size_t fs = (size_t)cout * 32;
  float* dst = U + ((size_t)ch * N * N * cout + co) * 32 + k;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j)
      dst[(size_t)(N * i + j) * fs] = (float)(gg[i][0] * wino_g<M>(j, 0) + gg[i][1] * wino_g<M>(j, 1) + gg[i][2] * wino_g<M>(j, 2));
}

// =================================================================================================
// M_b[T x Cout] = V_b[T x CIN] * U_b, b = nf g + f.  (WM PT 32) rows x 128 couts per workgroup of 4 waves
// (WM x WN, wave tile PT x CT blocks of 32 x 32), K walked in 32-channel chunks, double-buffered LDS-DMA
// (one barrier per chunk), raw accumulators stored.  Workgroup id -> (cout panel, row tile, b) with
// panel = id % panels: like the direct kernels an XCD (id % 8) only touches its own weight panel of b.
// =================================================================================================
template <int CIN, int WM, int WN, int PT, int CT, int MM = MM_F32>
__global__ __launch_bounds__(256, 2) void wino_gemm_kernel(const WinoArgs a) {
  constexpr int BM = WM * PT * 32, BN = WN * CT * 32;
  constexpr int NCH = CIN / 32;
  constexpr int BUF = (BM + BN) * 32;
  static_assert(WM * WN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, hh = lane >> 5;

  const int panels = a.Cout / BN, mtiles = (a.T + BM - 1) / BM;
  const int nt = blockIdx.x % panels, rest = blockIdx.x / panels;
  const int mt = rest % mtiles, b = rest / mtiles;
  const int g = b / a.nf, f = b - g * a.nf;
  const int m0 = mt * BM, n0 = nt * BN;
  const float* __restrict__ Vb = a.V + (size_t)b * a.T * CIN;
  const float* __restrict__ Ub = a.U + (size_t)g * a.u_gs + ((size_t)f * a.Cout + n0) * 32;
  const int mlast = a.T - 1;

  // staging: thread t fills LDS slot (t & 7) of rows (t >> 3) + 32 j with channel block (t & 7) ^ ((row >> 1) & 7)
  const int r0 = tid >> 3;
  const int c4 = (tid & 7) ^ ((r0 >> 1) & 7);
  unsigned pvoff[BM / 32];
#pragma unroll
  for (int j = 0; j < BM / 32; ++j) pvoff[j] = (unsigned)((min(m0 + r0 + 32 * j, mlast) * CIN + c4 * 4) * 4);
  const unsigned wvoff = (unsigned)((r0 * 32 + c4 * 4) * 4);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

#define ISSUE_TILE(CH, BUFI)                                                                         \
  {                                                                                                  \
    const float* pb_ = Vb + (CH) * 32;                                                               \
    const unsigned lb_ = lds0 + (unsigned)(((BUFI) * BUF + wid * 256) * 4);                          \
    _Pragma("unroll") for (int j_ = 0; j_ < BM / 32; ++j_) glds16<0>(pb_, pvoff[j_], lb_ + j_ * 4096); \
    const float* tb_ = Ub + (size_t)(CH) * a.nf * a.Cout * 32;                                       \
    _Pragma("unroll") for (int j_ = 0; j_ < BN / 32; ++j_)                                           \
        glds16<0>(tb_ + 1024 * j_, wvoff, lb_ + BM * 128 + j_ * 4096);                               \
  }

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

#ifdef SE3TN_WG_TRACE
  unsigned long long t_start_ = 0;
  if (tid == 0 && blockIdx.x < 4096) {
    t_start_ = __builtin_amdgcn_s_memrealtime();
    unsigned hw_, xcc_;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
    ::se3tn_wg_trace[blockIdx.x * 8 + 0] = 1ull + blockIdx.x;
    ::se3tn_wg_trace[blockIdx.x * 8 + 1] = hw_;
    ::se3tn_wg_trace[blockIdx.x * 8 + 2] = xcc_;
    ::se3tn_wg_trace[blockIdx.x * 8 + 3] = t_start_;
  }
#endif
#ifdef SE3TN_WINO_STAGGER
  // experiment: the two workgroups of a CU start together and, tiles being equal, stay in lock step (both in their
  // prologue, both in their epilogue): delay one of them by about half a tile
  {
#if SE3TN_WINO_STAGGER == 1
    const bool late = blockIdx.x >= 256 && blockIdx.x < 512;
#else
    const bool late = blockIdx.x < 512 && (blockIdx.x & 1);
#endif
    if (late)
      for (int i_ = 0; i_ < (CIN == 256 ? 3 : 6); ++i_) __builtin_amdgcn_s_sleep(127);
  }
#endif
  ISSUE_TILE(0, 0)
  wait_dma_and_barrier();

  for (int ch = 0; ch < NCH; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < NCH) ISSUE_TILE(ch + 1, buf ^ 1)
    const float* pP = smem + buf * BUF + (wm * PT * 32 + l31) * 32;
    const float* pW = smem + buf * BUF + (BM + wn * CT * 32 + l31) * 32;
#define FOG(G) ((G) == 0 ? fo0 : (G) == 1 ? fo1 : (G) == 2 ? fo2 : fo3)
#define PXF(G) *reinterpret_cast<const float4*>(pP + i * 1024 + FOG(G))
#define WTF(G) *reinterpret_cast<const float4*>(pW + j * 1024 + FOG(G))
    if (MM == MM_F16X3) {   // V and U are split rows (f16 hi | lo): hi*hi + hi*lo + lo*hi on the f16 matrix cores
      SE3TN_MMA_SPLIT(PT, CT, PXF, WTF)
    } else {
      SE3TN_MMA_GROUP(PT, CT, PXF(0), WTF(0))
      SE3TN_MMA_GROUP(PT, CT, PXF(1), WTF(1))
      SE3TN_MMA_GROUP(PT, CT, PXF(2), WTF(2))
      SE3TN_MMA_GROUP(PT, CT, PXF(3), WTF(3))
    }
#undef PXF
#undef WTF
#undef FOG
    if (ch + 1 < NCH) wait_dma_and_barrier();
  }
#undef ISSUE_TILE

#ifdef SE3TN_WG_TRACE
  if (tid == 0 && blockIdx.x < 4096) ::se3tn_wg_trace[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime();   // K-loop done
#endif
  // lane holds row l31 x couts {8 q + 4 hh + 0..3} of each 32 x 32 block
  float* __restrict__ Mb = a.Mw + (size_t)b * a.T * a.Cout;
  // MM_F16X3: the accumulators carry U's exact per-(frequency, cout) power-of-two scale: undo it here, M stays float32
  const float* __restrict__ wsc = (MM == MM_F16X3) ? a.uscale + ((size_t)g * a.nf + f) * a.Cout : nullptr;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int m = m0 + (wm * PT + i) * 32 + l31;
    if (m > mlast) continue;
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = n0 + (wn * CT + j) * 32 + q * 8 + hh * 4;
        float4 v = make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        if (MM == MM_F16X3) {
          const float4 w = *reinterpret_cast<const float4*>(wsc + c);
          v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w;
        }
        *reinterpret_cast<float4*>(Mb + (size_t)m * a.Cout + c) = v;
      }
  }
#ifdef SE3TN_WG_TRACE
  if (tid == 0 && blockIdx.x < 4096) ::se3tn_wg_trace[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();   // stores issued
#endif
}

// -------------------------------------------------------------------------------------------------
// wino_gemm8_kernel: the same 96 x 128 tile, LDS image and DMA as wino_gemm_kernel<CIN, 1, 4, 3, 1>, but EIGHT waves per
// workgroup: waves 0-3 and 4-7 take the two 16-channel halves of every 32-channel K-step (intra-workgroup split-K), and the
// two half sums are added through LDS at the end of the tile (fixed order: low half + high half).  Why: a WG-level trace of
// the 4-wave kernel (scripts/wg_trace.py, profiles/r03_wg_trace.txt) shows that with 6.75 tiles per CU a CU spends ~21 % of
// the launch with ONE resident workgroup = one wave per SIMD, which cannot keep the matrix pipe busy on its own (no
// second wave to issue under its ds_reads / barrier / DMA wait).  With 8 waves a lone workgroup still has two waves per
// SIMD, and two co-resident workgroups have four.
template <int CIN>
__global__ __launch_bounds__(512, 2) void wino_gemm8_kernel(const WinoArgs a) {
  constexpr int PT = 3, BM = 96, BN = 128;
  constexpr int NCH = CIN / 32;
  constexpr int BUF = (BM + BN) * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid & 3, kh = wid >> 2;
  const int l31 = lane & 31, hh = lane >> 5;

  const int panels = a.Cout / BN, mtiles = (a.T + BM - 1) / BM;
  const int nt = blockIdx.x % panels, rest = blockIdx.x / panels;
  const int mt = rest % mtiles, b = rest / mtiles;
  const int g = b / a.nf, f = b - g * a.nf;
  const int m0 = mt * BM, n0 = nt * BN;
  const float* __restrict__ Vb = a.V + (size_t)b * a.T * CIN;
  const float* __restrict__ Ub = a.U + (size_t)g * a.u_gs + ((size_t)f * a.Cout + n0) * 32;
  const int mlast = a.T - 1;

  // staging: thread t fills LDS slot (t & 7) of row (t >> 3) + 64 j with channel block (t & 7) ^ ((row >> 1) & 7);
  // V rows 0..95: pass 0 all eight waves, pass 1 waves 0-3 (rows 64..95); U rows 0..127: two passes
  const int r0 = tid >> 3;
  const int c4 = (tid & 7) ^ ((r0 >> 1) & 7);
  const unsigned pv0 = (unsigned)((min(m0 + r0, mlast) * CIN + c4 * 4) * 4);
  const unsigned pv1 = (unsigned)((min(m0 + r0 + 64, mlast) * CIN + c4 * 4) * 4);
  const unsigned wvoff = (unsigned)((r0 * 32 + c4 * 4) * 4);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

#define ISSUE_TILE8(CH, BUFI)                                                                        \
  {                                                                                                  \
    const float* pb_ = Vb + (CH) * 32;                                                               \
    const unsigned lb_ = lds0 + (unsigned)(((BUFI) * BUF + wid * 256) * 4);                          \
    glds16<0>(pb_, pv0, lb_);                                                                        \
    if (wid < 4) glds16<0>(pb_, pv1, lb_ + 8192);                                                    \
    const float* tb_ = Ub + (size_t)(CH) * a.nf * a.Cout * 32;                                       \
    glds16<0>(tb_, wvoff, lb_ + BM * 128);                                                           \
    glds16<0>(tb_ + 2048, wvoff, lb_ + BM * 128 + 8192);                                             \
  }

  const int X = (l31 >> 1) & 7;
  const int lo = (hh ^ (X & 1)) * 4, xk = X >> 1;
  // this wave's two 8-k groups of a K-step: 2 kh, 2 kh + 1
  const int foA = (((2 * kh) ^ xk) << 3) + lo, foB = (((2 * kh + 1) ^ xk) << 3) + lo;

  f32x16 acc[PT][1];
#pragma unroll
  for (int i = 0; i < PT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][0][e] = 0.f;

  ISSUE_TILE8(0, 0)
  wait_dma_and_barrier();

  for (int ch = 0; ch < NCH; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < NCH) ISSUE_TILE8(ch + 1, buf ^ 1)
    const float* pP = smem + buf * BUF + l31 * 32;
    const float* pW = smem + buf * BUF + (BM + wn * 32 + l31) * 32;
#define PXF8(FO) *reinterpret_cast<const float4*>(pP + i * 1024 + (FO))
#define WTF8(FO) *reinterpret_cast<const float4*>(pW + j * 1024 + (FO))
    SE3TN_MMA_GROUP(PT, 1, PXF8(foA), WTF8(foA))
    SE3TN_MMA_GROUP(PT, 1, PXF8(foB), WTF8(foB))
#undef PXF8
#undef WTF8
    if (ch + 1 < NCH) wait_dma_and_barrier();
  }
#undef ISSUE_TILE8

  // the two K halves meet in LDS (the operand buffers are free after this barrier): float4 slots [wn][i][q][lane]
  __syncthreads();
  float4* xch = reinterpret_cast<float4*>(smem);
  if (kh == 1) {
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        xch[((wn * PT + i) * 4 + q) * 64 + lane] =
            make_float4(acc[i][0][4 * q + 0], acc[i][0][4 * q + 1], acc[i][0][4 * q + 2], acc[i][0][4 * q + 3]);
  }
  __syncthreads();
  if (kh == 1) return;
  // lane holds row l31 x couts {8 q + 4 hh + 0..3} of each 32 x 32 block
  float* __restrict__ Mb = a.Mw + (size_t)b * a.T * a.Cout;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int m = m0 + i * 32 + l31;
    if (m > mlast) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 h = xch[((wn * PT + i) * 4 + q) * 64 + lane];
      const int c = n0 + wn * 32 + q * 8 + hh * 4;
      *reinterpret_cast<float4*>(Mb + (size_t)m * a.Cout + c) =
          make_float4(acc[i][0][4 * q + 0] + h.x, acc[i][0][4 * q + 1] + h.y, acc[i][0][4 * q + 2] + h.z, acc[i][0][4 * q + 3] + h.w);
    }
  }
}


// -------------------------------------------------------------------------------------------------
// wino_gemmp_kernel ("p" = persistent): 128 rows x 256 couts per tile, EIGHT waves (2 x 4, each 64 x 64 = 2 x 2 blocks), one
// workgroup per CU that walks tiles v = w, w + G, ... without leaving its K-loop: the first chunk of the NEXT tile is DMA'd during the
// last K-step of the current one, the accumulators are stored (fire and forget) between two K-steps, so a tile boundary costs the
// store issue only -- no prologue, no drained pipeline, whatever the K length (K = 256 is 8 K-steps per tile).
// Why this shape: with F(6x6) a batch-64 launch is exactly 512 of these tiles = 2.0 per CU for BOTH layer shapes
// (1024 x 256 x 64 planes and 2 x 256 x 512 x 64), a V row tile is fetched once for all its couts (AB2) and the per-K-step DMA is
// 48 KB per 128 x 256 x 32 MACs where two co-resident 128 x 128 workgroups move 64 KB.
// Tile order: virtual tile v runs on XCD v % 8 (observed dispatch: workgroup w on XCD w % 8; speed only); the tiles of one plane
// b = (group, frequency) are consecutive slots of ONE XCD, so U_b and V_b cross that XCD's L2 once.
// Same fragment layout, swizzle and k order as wino_gemm_kernel: bit-identical M.
// -------------------------------------------------------------------------------------------------
// (Measured and not kept -- profiles/r04_gemmp_trace.txt, EXPERIMENTS item 33, source in commit c7ec287: 256 x 256 tiles with 4 x 2 waves
// of 64 x 128; fragments of the next 8-k group requested a group early; the DMA pieces spread between the MFMA groups; s_setprio
// alternation between the two waves of a SIMD: all the same speed or slower.)
template <int CIN>
__global__ __launch_bounds__(512, 2) void wino_gemmp_kernel(const WinoArgs a, int total_tiles) {
  constexpr int BM = 128, BN = 256;
  constexpr int WN = 4, PT = 2, CT = 2;   // 2 x 4 waves, wave tile 64 rows x 64 couts
  constexpr int VP = BM / 64;             // DMA pieces per thread for the V tile (U: 4)
  constexpr int NCH = CIN / 32;
  constexpr int BUF = (BM + BN) * 32;
  static_assert(NCH % 2 == 0, "the double buffer's parity must be the same at every tile start");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, hh = lane >> 5;
  const int panels = a.Cout / BN, mtiles = (a.T + BM - 1) / BM, tpp = panels * mtiles;
  const int mlast = a.T - 1;

  // staging: thread t fills LDS slot (t & 7) of rows (t >> 3) + 64 j with channel block (t & 7) ^ ((row >> 1) & 7)
  const int r0 = tid >> 3;
  const int c4 = (tid & 7) ^ ((r0 >> 1) & 7);
  const unsigned wvoff = (unsigned)((r0 * 32 + c4 * 4) * 4);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

  // virtual tile -> (plane b, row tile, cout panel)
  auto decode = [&](int v, int& b, int& m0, int& n0) {
    const int xcd = v & 7, s = v >> 3;
    const int q = s / tpp, r = s - q * tpp;
    b = q * 8 + xcd;
    n0 = (r % panels) * BN;
    m0 = (r / panels) * BM;
  };
  const float* Vb;
  const float* Ub;
  unsigned pvoff[VP];
  auto set_tile = [&](int b, int m0, int n0) {
    const int g = b / a.nf, f = b - g * a.nf;
    Vb = a.V + (size_t)b * a.T * CIN;
    Ub = a.U + (size_t)g * a.u_gs + ((size_t)f * a.Cout + n0) * 32;
#pragma unroll
    for (int j = 0; j < VP; ++j) pvoff[j] = (unsigned)((min(m0 + r0 + 64 * j, mlast) * CIN + c4 * 4) * 4);
  };
#define ISSUE_TILEP(CH, BUFI)                                                                        \
  {                                                                                                  \
    const float* pb_ = Vb + (CH) * 32;                                                               \
    const unsigned lb_ = lds0 + (unsigned)(((BUFI) * BUF + wid * 256) * 4);                          \
    _Pragma("unroll") for (int j_ = 0; j_ < VP; ++j_) glds16<0>(pb_, pvoff[j_], lb_ + j_ * 8192);     \
    const float* tb_ = Ub + (size_t)(CH) * a.nf * a.Cout * 32;                                       \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) glds16<0>(tb_ + 2048 * j_, wvoff, lb_ + BM * 128 + j_ * 8192); \
  }

  const int X = (l31 >> 1) & 7;
  const int lo = (hh ^ (X & 1)) * 4, xk = X >> 1;
  const int fo0 = ((0 ^ xk) << 3) + lo, fo1 = ((1 ^ xk) << 3) + lo, fo2 = ((2 ^ xk) << 3) + lo, fo3 = ((3 ^ xk) << 3) + lo;

  f32x16 acc[PT][CT];
#pragma unroll
  for (int i = 0; i < PT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int v = blockIdx.x;
  if (v >= total_tiles) return;
  int b, m0, n0;
  decode(v, b, m0, n0);
  set_tile(b, m0, n0);
  ISSUE_TILEP(0, 0)
  wait_dma_and_barrier();

  int kstep_ = 0;
  (void)kstep_;
  while (true) {
    const int vn = v + gridDim.x;
    const bool more = vn < total_tiles;
    const int cb = b, cm0 = m0, cn0 = n0;   // the tile whose accumulators are being formed
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      const int buf = ch & 1;
      GP_STAMP(kstep_, 0)
      // (where the six DMA pieces of a wave are issued does not matter: all at the step start, as here, or spread between the MFMA
      // groups, a K-step takes 9,690 cycles for 8,192 of matrix work; nor does alternating s_setprio between the two waves of a
      // SIMD help -- profiles/r04_gemmp_trace.txt, EXPERIMENTS item 33)
      if (ch + 1 < NCH) {
        ISSUE_TILEP(ch + 1, buf ^ 1)
      } else if (more) {   // the next tile's first chunk rides under this tile's last K-step
        decode(vn, b, m0, n0);
        set_tile(b, m0, n0);
        ISSUE_TILEP(0, buf ^ 1)
      }
      GP_STAMP(kstep_, 1)
      const float* pP = smem + buf * BUF + (wm * PT * 32 + l31) * 32;
      const float* pW = smem + buf * BUF + (BM + wn * CT * 32 + l31) * 32;
#define FOG(G) ((G) == 0 ? fo0 : (G) == 1 ? fo1 : (G) == 2 ? fo2 : fo3)
#define PXF(G) *reinterpret_cast<const float4*>(pP + i * 1024 + FOG(G))
#define WTF(G) *reinterpret_cast<const float4*>(pW + j * 1024 + FOG(G))
      SE3TN_MMA_GROUP(PT, CT, PXF(0), WTF(0))
      SE3TN_MMA_GROUP(PT, CT, PXF(1), WTF(1))
      SE3TN_MMA_GROUP(PT, CT, PXF(2), WTF(2))
      SE3TN_MMA_GROUP(PT, CT, PXF(3), WTF(3))
#undef PXF
#undef WTF
#undef FOG
      GP_STAMP(kstep_, 2)
      if (ch + 1 < NCH || more) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GP_STAMP(kstep_, 3)
        __syncthreads();
      }
      GP_STAMP(kstep_, 4)
      ++kstep_;
    }
    // lane holds row l31 x couts {8 q + 4 hh + 0..3} of each 32 x 32 block; the stores drain under the next tile's first K-step
    float* __restrict__ Mb = a.Mw + (size_t)cb * a.T * a.Cout;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int m = cm0 + (wm * PT + i) * 32 + l31;
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = cn0 + (wn * CT + j) * 32 + q * 8 + hh * 4;
          if (m <= mlast)
            *reinterpret_cast<float4*>(Mb + (size_t)m * a.Cout + c) =
                make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = 0.f;
        }
    }
    if (!more) break;
    v = vn;
  }
#undef ISSUE_TILEP
}

// =================================================================================================
// Block-level fusions of the F(4x4) path (a ResnetBasicBlock = conv1+BN+ReLU, conv2+BN, +x, ReLU;
// network_modules.py:103-120).  Both remove a full write + read of an activation tensor between two
// HBM-bound transform passes:
//   wino_mid_kernel   out-transform of conv1 (A^T M A + bias, ReLU) and in-transform of conv2 (B^T d B) in one
//                     pass: one workgroup = one image x CS channels, the 4TH x 4TH activation lives in LDS
//                     (zero border included) between the two phases; the activation tensor itself is only
//                     written on request (keep != nullptr: tests / se3tn_debug_buffer);
//   wino_tail_kernel  out-transform of the heads' last conv (+ bias + residual, ReLU) with AdaptiveAvgPool2d(1),
//                     Linear(512,3) and Tanh (se3_tracknet.py:72-73,77-78,100-109): one workgroup = one image x
//                     one head, the 11 x 11 x 512 activation is reduced in registers and never stored
//                     (again unless asked for).  The float64 pose update follows in pose_update_kernel.
// =================================================================================================
template <> struct VecT<1> { typedef float type; };

// M = 4 | 6 (the output tile edge), TH x TH tiles per image, CS channels per workgroup, VEC channels per thread.
// F(6x6): 22 x 22 maps = 4 x 4 tiles, 11 x 11 = 2 x 2; the padded activation the two phases share has the same 26 x 26 / 14 x 14
// pixels as F(4x4)'s (M TH + 2).  Same operations in the same order as wino_output_kernel<M> followed by wino_input_kernel<M>:
// the fused and the conv-by-conv form give identical bits (tests/test_gpu_parity.py).
template <int M, int TH, int CS, int VEC, int SP = 0>
__global__ __launch_bounds__(((TH * TH * (CS / VEC) + 63) / 64) * 64) void wino_mid_kernel(const WinoArgs a, float* __restrict__ keep) {
  constexpr int N = M + 2, HP = M * TH + 2, NT = TH * TH, CP = CS / VEC, ITEMS = NT * CP;
  typedef typename VecT<VEC>::type vec;
  static_assert(SP == 0 || (M == 4 && VEC == 2), "the f16x3 blocks are F(4x4) with channel pairs");
  extern __shared__ __attribute__((aligned(16))) float act[];  // [HP][HP][CS], padded coordinates
  const int n = blockIdx.x, c0 = blockIdx.y * CS, g = blockIdx.z;
  const int tid = threadIdx.x;
  for (int i = tid; i < HP * HP * CS / 4; i += blockDim.x) reinterpret_cast<float4*>(act)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const int tile = tid / CP, cp = tid - tile * CP;
  const int ty = tile / TH, tx = tile - ty * TH;
  const bool active = tid < ITEMS;
  const int t = n * NT + tile;
  const int Hp = a.H + 2, Wp = a.W + 2;
  if (active) {
    const float* __restrict__ src = a.Mw + ((size_t)g * a.nf * a.T + t) * a.Cout + c0 + cp * VEC;
    const size_t fs = (size_t)a.T * a.Cout;
    float u[M][N][VEC];
#pragma unroll
    for (int s = 0; s < N; ++s) {
      float m[N][VEC];
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const vec v = *reinterpret_cast<const vec*>(src + (size_t)(N * r + s) * fs);
        __builtin_memcpy(m[r], &v, sizeof(v));
      }
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float acc = 0.f;
          bool first = true;
#pragma unroll
          for (int r = 0; r < N; ++r) axpy(acc, wino_at<M>(i, r), m[r][e], first);
          u[i][s][e] = acc;
        }
    }
    float b[VEC];
    {
      const vec v = *reinterpret_cast<const vec*>(a.bias + (size_t)g * a.bias_gs + c0 + cp * VEC);
      __builtin_memcpy(b, &v, sizeof(v));
    }
    float* __restrict__ kp = keep ? keep + (size_t)g * a.out_gs + c0 + cp * VEC : nullptr;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const int oy = M * ty + i, ox = M * tx + j;
        if (oy >= a.H || ox >= a.W) continue;
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float acc = 0.f;
          bool first = true;
#pragma unroll
          for (int s = 0; s < N; ++s) axpy(acc, wino_at<M>(j, s), u[i][s][e], first);
          acc += b[e];
          o[e] = fmaxf(acc, 0.f);
        }
        vec v;
        __builtin_memcpy(&v, o, sizeof(v));
        *reinterpret_cast<vec*>(act + ((oy + 1) * HP + ox + 1) * CS + cp * VEC) = v;
        if (kp) *reinterpret_cast<vec*>(kp + (size_t)((n * Hp + oy + 1) * Wp + ox + 1) * a.out_ld) = v;
      }
  }
  __syncthreads();
  if (active) {
    float bt[N][N][VEC];  // B^T d, one column s at a time
#pragma unroll
    for (int s = 0; s < N; ++s) {
      float d[N][VEC];
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const vec v = *reinterpret_cast<const vec*>(act + ((M * ty + r) * HP + M * tx + s) * CS + cp * VEC);
        __builtin_memcpy(d[r], &v, sizeof(v));
      }
#pragma unroll
      for (int i = 0; i < N; ++i)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float acc = 0.f;
          bool first = true;
#pragma unroll
          for (int r = 0; r < N; ++r) axpy(acc, wino_bt<M>(i, r), d[r][e], first);
          bt[i][s][e] = acc;
        }
    }
    float* __restrict__ dst = a.V + ((size_t)g * a.nf * a.T + t) * a.C + (SP ? 0 : c0 + cp * VEC);
    const size_t fs = (size_t)a.T * a.C;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = 0; j < N; ++j) {
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float acc = 0.f;
          bool first = true;
#pragma unroll
          for (int s = 0; s < N; ++s) axpy(acc, wino_bt<M>(j, s), bt[i][s][e], first);
          o[e] = acc;
        }
        if constexpr (SP != 0) {
          bad |= store_split_vec<VEC>(dst + (size_t)(N * i + j) * fs, 0, a.C, c0 + cp * VEC, o);
        } else {
          vec v;
          __builtin_memcpy(&v, o, sizeof(v));
          *reinterpret_cast<vec*>(dst + (size_t)(N * i + j) * fs) = v;
        }
      }
    if (SP && bad) atomicOr(a.overflow, 1);
  }
}

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// grid (n, 512 / CS channel slices, 2 heads); thread = (tile, channel pair) like wino_output_kernel, so the pass has the
// same parallelism; the per-channel sums over the map are reduced through LDS in a fixed order, each workgroup
// contributes the partial dot products of its CS channels with the head's three FC rows:
// fcpart[n][head][slice][3].  fc_finish_kernel adds the slices (fixed order), the bias, applies tanh and the pose update.
template <int M, int TH, int CS, int SP = 0>
__global__ __launch_bounds__(((TH * TH * (CS / 2) + 63) / 64) * 64) void wino_tail_kernel(const WinoArgs a, float* __restrict__ keep,
                                                                                         const float* __restrict__ fc_w,
                                                                                         float* __restrict__ fcpart) {
  constexpr int N = M + 2, NT = TH * TH, CP = CS / 2, ITEMS = NT * CP;
  static_assert(SP == 0 || M == 4, "the f16x3 blocks are F(4x4)");
  static_assert(CP == 32, "the FC partials are reduced inside one half-wave");
  __shared__ float red[NT][CP][2];
  const int n = blockIdx.x, sl = blockIdx.y, g = blockIdx.z, tid = threadIdx.x;
  const int tile = tid / CP, cp = tid - tile * CP;
  const int c = sl * CS + cp * 2;   // channel within the head
  const int Hp = a.H + 2, Wp = a.W + 2;
  if (tid < ITEMS) {
    const int ty = tile / TH, tx = tile - ty * TH;
    const size_t fs = (size_t)a.T * a.Cout;
    const float* __restrict__ src = a.Mw + ((size_t)g * a.nf * a.T + n * NT + tile) * a.Cout + c;
    const float* __restrict__ res = a.res + (size_t)g * a.res_gs + (SP ? 0 : c);
    float* __restrict__ kp = keep ? keep + (size_t)g * a.out_gs + c : nullptr;   // always float32 (f16x3 mode: head_f)
    const float2 b = *reinterpret_cast<const float2*>(a.bias + (size_t)g * a.bias_gs + c);
    // the tile's residual, requested before the planes (see wino_output_kernel); clamped address past the map edge
    float2 rr[M][M];
    if constexpr (SP == 0) {
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) {
          int oy = M * ty + i, ox = M * tx + j;
          oy = oy < a.H ? oy : a.H - 1; ox = ox < a.W ? ox : a.W - 1;
          rr[i][j] = *reinterpret_cast<const float2*>(res + (size_t)((n * Hp + oy + 1) * Wp + ox + 1) * a.res_ld);
        }
    }
    float u[M][N][2];
#pragma unroll
    for (int s = 0; s < N; ++s) {
      float m[N][2];
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const float2 v = *reinterpret_cast<const float2*>(src + (size_t)(N * r + s) * fs);
        m[r][0] = v.x; m[r][1] = v.y;
      }
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float acc = 0.f;
          bool first = true;
#pragma unroll
          for (int r = 0; r < N; ++r) axpy(acc, wino_at<M>(i, r), m[r][e], first);
          u[i][s][e] = acc;
        }
    }
    float sum[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const int oy = M * ty + i, ox = M * tx + j;
        if (oy >= a.H || ox >= a.W) continue;
        const size_t pix = (size_t)((n * Hp + oy + 1) * Wp + ox + 1);
        float2 r;
        if (SP) {
          float rr[2];
          load_split_vec<2>(res, pix, a.res_ld, c, rr);
          r = make_float2(rr[0], rr[1]);
        } else {
          r = rr[i][j];
        }
        float o[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float acc = 0.f;
          bool first = true;
#pragma unroll
          for (int s = 0; s < N; ++s) axpy(acc, wino_at<M>(j, s), u[i][s][e], first);
          o[e] = fmaxf(acc + (e ? b.y : b.x) + (e ? r.y : r.x), 0.f);
          sum[e] += o[e];
        }
        if (kp) *reinterpret_cast<float2*>(kp + pix * a.out_ld) = make_float2(o[0], o[1]);
      }
    red[tile][cp][0] = sum[0];
    red[tile][cp][1] = sum[1];
  }
  __syncthreads();
  if (tid < CP) {   // lanes 0-31 of wave 0: channel pair tid
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) { s0 += red[t][tid][0]; s1 += red[t][tid][1]; }
    const float inv = (float)(a.H * a.W);
    s0 /= inv; s1 /= inv;
    float acc[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      const float2 w = *reinterpret_cast<const float2*>(fc_w + (g * 3 + o) * 512 + sl * CS + tid * 2);
      float v = s0 * w.x + s1 * w.y;
#pragma unroll
      for (int sh = 16; sh > 0; sh >>= 1) v += __shfl_xor(v, sh, 64);   // lanes 0-31 only exchange among themselves
      acc[o] = v;
    }
    if (tid == 0) {
      float* dst = fcpart + ((size_t)(n * 2 + g) * (512 / CS) + sl) * 3;
      dst[0] = acc[0]; dst[1] = acc[1]; dst[2] = acc[2];
    }
  }
}

// 6 threads per pair (one per output): logit = sum of the slices' partial dot products (fixed order) + bias; tanh;
// the pair's first thread then composes the pose.  10 pairs per 64-thread workgroup.
__global__ __launch_bounds__(64) void fc_finish_kernel(const float* __restrict__ fcpart, int slices, const TailArgs tl,
                                                        const double* __restrict__ poseA, double* __restrict__ poseB, double tn,
                                                        double rn, int n) {
  __shared__ float y[60];
  const int t = threadIdx.x, li = t / 6, k = t - li * 6;
  const int i = blockIdx.x * 10 + li;
  const bool ok = t < 60 && i < n;
  if (ok) {
    const int g = k / 3, o = k - g * 3;
    const float* p = fcpart + ((size_t)(i * 2 + g) * slices) * 3 + o;
    float lg = 0.f;
    for (int s = 0; s < slices; ++s) lg += p[s * 3];
    lg += tl.fc_b[g * 4 + o];
    const float v = tanhf(lg);
    y[t] = v;
    tl.logits[i * 6 + k] = lg;
    float* dst = g == 0 ? tl.trans : tl.rot;
    if (dst) dst[i * 3 + o] = v;
  }
  __syncthreads();
  if (ok && k == 0 && poseA) pose_compose(y + li * 6, poseA + (size_t)i * 16, poseB + (size_t)i * 16, tn, rn);
}

// ---- launchers ---------------------------------------------------------------------------------
hipError_t launch_wino_weights(const float* packed, float* U, int cin, int cout, int m, hipStream_t st) {
  const int total = cin * cout;
  if (m == 2) hipLaunchKernelGGL(wino_weight_kernel<2>, dim3((total + 255) / 256), dim3(256), 0, st, packed, U, cout, total);
  else if (m == 4) hipLaunchKernelGGL(wino_weight_kernel<4>, dim3((total + 255) / 256), dim3(256), 0, st, packed, U, cout, total);
  else if (m == 6) hipLaunchKernelGGL(wino_weight_kernel<6>, dim3((total + 255) / 256), dim3(256), 0, st, packed, U, cout, total);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

template <int CIN, int WM, int WN, int PT, int CT, int MM = MM_F32>
static hipError_t launch_gemm(const WinoArgs& a, hipStream_t st) {
  static PerDeviceOnce attr;
  auto kern = wino_gemm_kernel<CIN, WM, WN, PT, CT, MM>;
  constexpr int BM = WM * PT * 32, BN = WN * CT * 32;
  const size_t lds = 2 * (BM + BN) * 32 * sizeof(float);
  bool* done = attr.current();
  if (!done || !*done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (done) *done = true;
  }
  const int grid = (a.Cout / BN) * ((a.T + BM - 1) / BM) * a.groups * a.nf;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}

template <int CIN>
static hipError_t launch_gemm8(const WinoArgs& a, hipStream_t st) {
  static PerDeviceOnce attr;
  auto kern = wino_gemm8_kernel<CIN>;
  const size_t lds = 2 * (96 + 128) * 32 * sizeof(float);
  bool* done = attr.current();
  if (!done || !*done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (done) *done = true;
  }
  const int grid = (a.Cout / 128) * ((a.T + 95) / 96) * a.groups * a.nf;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
  return hipGetLastError();
}

template <int CIN>
static hipError_t launch_gemmp(const WinoArgs& a, int total_tiles, int grid, hipStream_t st) {
  static PerDeviceOnce attr;
  auto kern = wino_gemmp_kernel<CIN>;
  const size_t lds = 2 * (128 + 256) * 32 * sizeof(float);
  bool* done = attr.current();
  if (!done || !*done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (done) *done = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a, total_tiles);
  return hipGetLastError();
}

#ifndef SE3TN_WINO_GEMMP
#define SE3TN_WINO_GEMMP 1   // 0: never take the persistent 128 x 256 kernel
#endif
#ifndef SE3TN_WINO_GEMM8
#define SE3TN_WINO_GEMM8 0   // 1: 96-row tiles run as 8-wave workgroups with intra-workgroup split-K (wino_gemm8_kernel)
#endif
// 128- or 96-row tiles: whichever leaves the shorter per-CU queue (in 32 x 32 x K blocks) on 256 CUs
template <int CIN>
static hipError_t launch_gemm_auto(const WinoArgs& a, hipStream_t st) {
  const long long per_b = (long long)(a.Cout / 128) * a.groups * a.nf;
  const long long q128 = ((((a.T + 127) / 128) * per_b + 255) / 256) * 16;
  const long long q96 = ((((a.T + 95) / 96) * per_b + 255) / 256) * 12;
  if (a.split) return q96 < q128 ? launch_gemm<CIN, 1, 4, 3, 1, MM_F16X3>(a, st) : launch_gemm<CIN, 2, 2, 2, 2, MM_F16X3>(a, st);
  // persistent 128 x 256 tiles, one 8-wave workgroup per CU: when they fill every CU at least twice (F(6x6) at batch 64: exactly 2.0)
  // and the plane count lets the XCD-local tile order cover them (groups nf % 8 == 0: 36 F(4x4) planes do not).
  // a.gemmp: -1 = that rule, 0 = never, 1 = whenever the shape allows (SE3TN_WINO_GEMMP at se3tn_create: A/B runs and the tests'
  // bit-equality check of the two kernels on ragged shapes)
  if (SE3TN_WINO_GEMMP && a.gemmp != 0 && a.Cout % 256 == 0 && (a.groups * a.nf) % 8 == 0) {
    const int cus = a.num_cus > 0 ? a.num_cus : 256;
    const int tiles = (a.Cout / 256) * ((a.T + 127) / 128) * a.groups * a.nf;
    // the automatic rule takes it for the plane sets it was measured on (EXPERIMENTS item 27): F(6x6).  The two-group F(4x4) heads
    // (72 planes, T = 576 at batch 64 = 4.5 row tiles) also pass the divisibility test above and would trade their exact-fit 96-row
    // tiling for it unmeasured: they keep launch_gemm unless gemmp == 1 forces it.
    const bool measured_shape = a.nf == 64;
    if (a.gemmp == 1 || (tiles >= 2 * cus && measured_shape)) return launch_gemmp<CIN>(a, tiles, tiles < cus ? tiles : (cus / 8) * 8, st);
  }
  if (SE3TN_WINO_GEMM8 && q96 < q128) return launch_gemm8<CIN>(a, st);
  return q96 < q128 ? launch_gemm<CIN, 1, 4, 3, 1>(a, st) : launch_gemm<CIN, 2, 2, 2, 2>(a, st);
}

template <int M, int VEC>
static hipError_t launch_transformed(const WinoArgs& a, int epi, hipStream_t st) {
  hipLaunchKernelGGL((wino_input_kernel<M, VEC>), dim3((a.T * (a.C / VEC) + 255) / 256, a.groups), dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = a.C == 256 ? launch_gemm_auto<256>(a, st) : launch_gemm_auto<512>(a, st);
  if (e != hipSuccess) return e;
  const dim3 og((a.T * (a.Cout / VEC) + 255) / 256, a.groups);
  if (epi == 0) hipLaunchKernelGGL((wino_output_kernel<M, VEC, 0>), og, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((wino_output_kernel<M, VEC, 1>), og, dim3(256), 0, st, a);
  return hipGetLastError();
}


// One ResnetBasicBlock on the F(4x4) path: in-transform, GEMM(U1), [out|in] mid transform, GEMM(U2), then either
// the out-transform with the residual epilogue or (tl != nullptr, the heads' last block) the fused
// out-transform + avg-pool + FC + tanh.  c1 describes conv1 (in = block input, out = the intermediate activation
// buffer, only written if keep_mid); conv2 reads the same V / Mw workspaces, residual = c1.in, output = out2.
template <int M, int TH, int CS, int VEC, int SP = 0>
static hipError_t launch_mid(const WinoArgs& a, float* keep, hipStream_t st) {
  constexpr int HP = M * TH + 2, THREADS = ((TH * TH * (CS / VEC) + 63) / 64) * 64;
  constexpr size_t lds = (size_t)HP * HP * CS * sizeof(float);
  static PerDeviceOnce attr;
  auto kern = wino_mid_kernel<M, TH, CS, VEC, SP>;
  bool* done = attr.current();
  if (!done || !*done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (done) *done = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.n, a.C / CS, a.groups), dim3(THREADS), lds, st, a, keep);
  return hipGetLastError();
}

#ifndef WINO6_MID_VEC
#define WINO6_MID_VEC 1    // channels per thread of the F(6x6) mid transform: 1 = 512 threads on the 22 x 22 maps (8 waves per CU beside
#endif                     // 86 KB of LDS), 2 = 256
#ifndef WINO6_MID_CS_AB
#define WINO6_MID_CS_AB 32 // ... on the 22 x 22 maps (LDS 26 x 26 x CS floats: 86 KB at 32 = one workgroup per CU; 16 -> three of 256 threads: measured equal)
#endif
#ifndef WINO6_MID_CS_H
#define WINO6_MID_CS_H 64  // channels per workgroup of the F(6x6) mid transform on the 11 x 11 maps (LDS 14 x 14 x CS floats)
#endif

hipError_t launch_wino_block(const WinoArgs& c1, const float* U2, const float* uscale2, const float* bias2, float* out2, int keep_mid,
                             float* keep_out2, const TailArgs* tl, hipStream_t st, int mark_after_mid(void*), void* mark_ctx) {
  if ((c1.m != 4 && c1.m != 6) || c1.nf != (c1.m + 2) * (c1.m + 2) || c1.C != c1.Cout || (c1.C != 256 && c1.C != 512))
    return hipErrorInvalidValue;
  const int th_ab = c1.m == 4 ? 6 : 4, th_h = c1.m == 4 ? 3 : 2;
  if (!((c1.th == th_ab && c1.C == 256) || (c1.th == th_h && c1.C == 512)) || c1.tw != c1.th) return hipErrorInvalidValue;
  if (c1.m == 6 && c1.split) return hipErrorInvalidValue;   // F(6x6) is a float32 path
  const bool ab = c1.C == 256;
  if (c1.m == 6) {
    hipLaunchKernelGGL((wino_input_kernel<6, 2>), dim3((c1.T * (c1.C / 2) + 255) / 256, c1.groups), dim3(256), 0, st, c1);
  } else {
    const dim3 ig((c1.T * (c1.C / WINO4_VEC) + 255) / 256, c1.groups);
    if (c1.split) hipLaunchKernelGGL((wino_input_kernel<4, WINO4_VEC, 1>), ig, dim3(256), 0, st, c1);
    else hipLaunchKernelGGL((wino_input_kernel<4, WINO4_VEC>), ig, dim3(256), 0, st, c1);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = ab ? launch_gemm_auto<256>(c1, st) : launch_gemm_auto<512>(c1, st);
  if (e != hipSuccess) return e;
  float* keep1 = keep_mid ? c1.out : nullptr;
  if (c1.m == 6) e = ab ? launch_mid<6, 4, WINO6_MID_CS_AB, WINO6_MID_VEC>(c1, keep1, st) : launch_mid<6, 2, WINO6_MID_CS_H, WINO6_MID_VEC>(c1, keep1, st);
  else if (c1.split) e = ab ? launch_mid<4, 6, WINO_MID_CS_AB, 2, 1>(c1, keep1, st) : launch_mid<4, 3, WINO_MID_CS_H, 2, 1>(c1, keep1, st);
  else e = ab ? launch_mid<4, 6, WINO_MID_CS_AB, 2>(c1, keep1, st) : launch_mid<4, 3, WINO_MID_CS_H, 2>(c1, keep1, st);
  if (e != hipSuccess) return e;
  if (mark_after_mid && mark_after_mid(mark_ctx)) return hipErrorUnknown;
  WinoArgs c2 = c1;
  c2.U = U2; c2.uscale = uscale2; c2.bias = bias2; c2.res = c1.in; c2.res_ld = c1.in_ld; c2.res_gs = c1.in_gs; c2.out = out2;
  e = ab ? launch_gemm_auto<256>(c2, st) : launch_gemm_auto<512>(c2, st);
  if (e != hipSuccess) return e;
  if (tl) {
    if (ab || c1.groups != 2 || c1.Cout != 512) return hipErrorInvalidValue;
    constexpr int CS = 64;
    const dim3 tg(c1.n, 512 / CS, 2);
    if (c1.m == 6) hipLaunchKernelGGL((wino_tail_kernel<6, 2, CS>), tg, dim3(((4 * (CS / 2) + 63) / 64) * 64), 0, st, c2, keep_out2, tl->fc_w, tl->fcpart);
    else if (c1.split) hipLaunchKernelGGL((wino_tail_kernel<4, 3, CS, 1>), tg, dim3(((9 * (CS / 2) + 63) / 64) * 64), 0, st, c2, keep_out2, tl->fc_w, tl->fcpart);
    else hipLaunchKernelGGL((wino_tail_kernel<4, 3, CS>), tg, dim3(((9 * (CS / 2) + 63) / 64) * 64), 0, st, c2, keep_out2, tl->fc_w, tl->fcpart);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fc_finish_kernel, dim3((c1.n + 9) / 10), dim3(64), 0, st, tl->fcpart, 512 / CS, *tl, tl->poseA, tl->poseB,
                       tl->tn, tl->rn, c1.n);
  } else if (c1.m == 6) {
    hipLaunchKernelGGL((wino_output_kernel<6, 2, 1>), dim3((c2.T * (c2.Cout / 2) + 255) / 256, c2.groups), dim3(256), 0, st, c2);
  } else {
    const dim3 og((c2.T * (c2.Cout / WINO4_VEC) + 255) / 256, c2.groups);
    if (c1.split) hipLaunchKernelGGL((wino_output_kernel<4, WINO4_VEC, 1, 1>), og, dim3(256), 0, st, c2);
    else hipLaunchKernelGGL((wino_output_kernel<4, WINO4_VEC, 1>), og, dim3(256), 0, st, c2);
  }
  return hipGetLastError();
}

hipError_t launch_wino_conv(const WinoArgs& a, int epi, hipStream_t st) {
  if ((a.C != 256 && a.C != 512) || a.Cout % 128 != 0 || (epi != 0 && epi != 1)) return hipErrorInvalidValue;
  if (a.m == 2 && a.nf == 16) return launch_transformed<2, 4>(a, epi, st);
  if (a.m == 4 && a.nf == 36) return launch_transformed<4, WINO4_VEC>(a, epi, st);
  if (a.m == 6 && a.nf == 64 && !a.split) return launch_transformed<6, 2>(a, epi, st);
  return hipErrorInvalidValue;
}

}  // namespace se3tn
