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

namespace se3tn {

// ---- Toom-Cook matrices ------------------------------------------------------------------------------
template <int M>
__host__ __device__ __forceinline__ constexpr float wino_bt(int i, int j) {  // B^T [(M+2) x (M+2)]
  constexpr float t2[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
  constexpr float t4[6][6] = {{1, -1.5f, -2, 1.5f, 1, 0}, {0, -1, 0.5f, 2.5f, 1, 0}, {0, 1, -2.5f, 0.5f, 1, 0},
                              {0, -2, -1, 2, 1, 0},       {0, 0.5f, -1, -0.5f, 1, 0}, {0, 1, -1.5f, -2, 1.5f, 1}};
  return M == 2 ? t2[i & 3][j & 3] : t4[i][j];
}
template <int M>
__host__ __device__ __forceinline__ constexpr float wino_at(int i, int j) {  // A^T [M x (M+2)]
  constexpr float t2[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
  constexpr float t4[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 0.5f, -2, 0}, {0, 1, 1, 0.25f, 4, 0}, {0, 1, -1, 0.125f, -8, 1}};
  return M == 2 ? t2[i & 1][j & 3] : t4[i][j];
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
  return M == 2 ? t2[i & 3][j] : t4[i][j];
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
template <int M, int VEC>
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoArgs a) {
  constexpr int N = M + 2;
  typedef typename VecT<VEC>::type vec;
  const int cvn = a.C / VEC;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.T * cvn) return;
  const int g = blockIdx.y;
  const TileId id = tile_of(idx, cvn, a.th, a.tw);
  const int Hp = a.H + 2, Wp = a.W + 2;
  const float* __restrict__ src = a.in + (size_t)g * a.in_gs + id.cv * VEC;
  // padded rows M ty .. M ty + M + 1 = input rows M ty - 1 .. M ty + M; windows of the last tile row /
  // column may reach past the border: those entries only feed dropped outputs
  float d[N][N][VEC];
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int s = 0; s < N; ++s) {
      const int Y = M * id.ty + r, X = M * id.tx + s;
      vec v;
      if (Y < Hp && X < Wp) v = *reinterpret_cast<const vec*>(src + (size_t)((id.n * Hp + Y) * Wp + X) * a.in_ld);
      else __builtin_memset(&v, 0, sizeof(v));
      __builtin_memcpy(d[r][s], &v, sizeof(v));
    }
  float bt[N][N][VEC];  // B^T d
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
  float* __restrict__ dst = a.V + ((size_t)g * a.nf * a.T + id.t) * a.C + id.cv * VEC;
  const size_t fs = (size_t)a.T * a.C;  // floats per frequency plane
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
      vec v;
      __builtin_memcpy(&v, o, sizeof(v));
      *reinterpret_cast<vec*>(dst + (size_t)(N * i + j) * fs) = v;
    }
}

// one thread: (group, tile t, VEC couts)
template <int M, int VEC, int EPI>
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
  const float* __restrict__ res = (EPI == 1) ? a.res + (size_t)g * a.res_gs + id.cv * VEC : nullptr;
  float* __restrict__ out = a.out + (size_t)g * a.out_gs + id.cv * VEC;
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int oy = M * id.ty + i, ox = M * id.tx + j;
      if (oy >= a.H || ox >= a.W) continue;
      const size_t pix = (size_t)((id.n * Hp + oy + 1) * Wp + ox + 1);
      float r[VEC];
      if (EPI == 1) {
        const vec v = *reinterpret_cast<const vec*>(res + pix * a.res_ld);
        __builtin_memcpy(r, &v, sizeof(v));
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
      vec v;
      __builtin_memcpy(&v, o, sizeof(v));
      *reinterpret_cast<vec*>(out + pix * a.out_ld) = v;
    }
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
template <int CIN, int WM, int WN, int PT, int CT>
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
    SE3TN_MMA_GROUP(PT, CT, PXF(0), WTF(0))
    SE3TN_MMA_GROUP(PT, CT, PXF(1), WTF(1))
    SE3TN_MMA_GROUP(PT, CT, PXF(2), WTF(2))
    SE3TN_MMA_GROUP(PT, CT, PXF(3), WTF(3))
#undef PXF
#undef WTF
#undef FOG
    if (ch + 1 < NCH) wait_dma_and_barrier();
  }
#undef ISSUE_TILE

  // lane holds row l31 x couts {8 q + 4 hh + 0..3} of each 32 x 32 block
  float* __restrict__ Mb = a.Mw + (size_t)b * a.T * a.Cout;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int m = m0 + (wm * PT + i) * 32 + l31;
    if (m > mlast) continue;
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = n0 + (wn * CT + j) * 32 + q * 8 + hh * 4;
        *reinterpret_cast<float4*>(Mb + (size_t)m * a.Cout + c) =
            make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
      }
  }
}

// ---- launchers ---------------------------------------------------------------------------------
hipError_t launch_wino_weights(const float* packed, float* U, int cin, int cout, int m, hipStream_t st) {
  const int total = cin * cout;
  if (m == 2) hipLaunchKernelGGL(wino_weight_kernel<2>, dim3((total + 255) / 256), dim3(256), 0, st, packed, U, cout, total);
  else if (m == 4) hipLaunchKernelGGL(wino_weight_kernel<4>, dim3((total + 255) / 256), dim3(256), 0, st, packed, U, cout, total);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

template <int CIN, int WM, int WN, int PT, int CT>
static hipError_t launch_gemm(const WinoArgs& a, hipStream_t st) {
  static PerDeviceOnce attr;
  auto kern = wino_gemm_kernel<CIN, WM, WN, PT, CT>;
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

// 128- or 96-row tiles: whichever leaves the shorter per-CU queue (in 32 x 32 x K blocks) on 256 CUs
template <int CIN>
static hipError_t launch_gemm_auto(const WinoArgs& a, hipStream_t st) {
  const long long per_b = (long long)(a.Cout / 128) * a.groups * a.nf;
  const long long q128 = ((((a.T + 127) / 128) * per_b + 255) / 256) * 16;
  const long long q96 = ((((a.T + 95) / 96) * per_b + 255) / 256) * 12;
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

hipError_t launch_wino_conv(const WinoArgs& a, int epi, hipStream_t st) {
  if ((a.C != 256 && a.C != 512) || a.Cout % 128 != 0 || (epi != 0 && epi != 1)) return hipErrorInvalidValue;
  if (a.m == 2 && a.nf == 16) return launch_transformed<2, 4>(a, epi, st);
  if (a.m == 4 && a.nf == 36) return launch_transformed<4, 2>(a, epi, st);
  return hipErrorInvalidValue;
}

}  // namespace se3tn
