// Winograd F(2x2,3x3) form of the stride-1 3x3 convolutions with 256 / 512 input channels
// (convAB2.conv1/.conv2 and trans|rot conv2.conv1/.conv2 of Se3TrackNet, se3_tracknet.py:68-76 via
// network_modules.py:86-120 ResnetBasicBlock) for large batches.  Still float32 end to end, still on
// v_mfma_f32_32x32x2_f32 -- but 16 multiplies per 2x2 output tile instead of 36:
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A            (Lavin & Gray 2016, cross-correlation form)
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
//     A^T = [1 1 1 0; 0 1 -1 -1]
//
// with the channel sum done per frequency f = 4 r + s as a GEMM  M_f[T x Cout] = V_f[T x C] U_f[C x Cout].
// 11x11 maps tile as 6x6 (the 12th row / column is computed and dropped: 1.89x fewer MFMA flops than
// direct), 22x22 as 11x11 (2.25x).  Three launches per convolution:
//   wino_input_kernel   d (4x4 windows of the zero-bordered NHWC tensor) -> V[g][f][T][C]
//   wino_gemm_kernel    16 x groups independent GEMMs, 128 x 128 tiles, 4 waves, 2 workgroups per CU,
//                       operands LDS-DMA'd with the same XOR swizzle / fragment layout as conv3x3_mfma.hip
//   wino_output_kernel  A^T M A + folded-BN bias (+ residual) + ReLU -> interior of the padded output
// U = G g G^T is derived on the device in float64 from the packed direct weights when the blob is
// uploaded / bound (wino_weight_kernel), so the blob format and the host packer do not change.
// The transform passes are pure HBM streams (V and M are 4x the activation they come from); they cost
// ~0.15 ms of the ~0.42 ms a 512-channel layer takes at batch 64 (direct: 0.57 ms).
#include "mfma_common.h"

namespace se3tn {

__device__ __forceinline__ float4 add4(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(const float4 a, const float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// one thread: (group, tile t, 4 channels)
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoArgs a) {
  const int c4n = a.C >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.T * c4n) return;
  const int g = blockIdx.y;
  const int c4 = idx % c4n, t = idx / c4n;
  const int tpi = a.th * a.tw;
  const int n = t / tpi, rem = t - n * tpi;
  const int ty = rem / a.tw, tx = rem - ty * a.tw;
  const int Hp = a.H + 2, Wp = a.W + 2;
  const float* __restrict__ src = a.in + (size_t)g * a.in_gs + c4 * 4;
  // padded rows 2 ty .. 2 ty + 3 = input rows 2 ty - 1 .. 2 ty + 2; for odd H the last window reaches one
  // row / column past the border: it only feeds the dropped 12th output row / column
  float4 d[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int Y = 2 * ty + r, X = 2 * tx + s;
      d[r][s] = (Y < Hp && X < Wp) ? *reinterpret_cast<const float4*>(src + (size_t)((n * Hp + Y) * Wp + X) * a.in_ld)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  float4 bt[4][4];  // B^T d
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bt[0][s] = sub4(d[0][s], d[2][s]);
    bt[1][s] = add4(d[1][s], d[2][s]);
    bt[2][s] = sub4(d[2][s], d[1][s]);
    bt[3][s] = sub4(d[1][s], d[3][s]);
  }
  float* __restrict__ dst = a.V + ((size_t)g * 16 * a.T + t) * a.C + c4 * 4;
  const size_t fs = (size_t)a.T * a.C;  // floats per frequency plane
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    *reinterpret_cast<float4*>(dst + (4 * r + 0) * fs) = sub4(bt[r][0], bt[r][2]);
    *reinterpret_cast<float4*>(dst + (4 * r + 1) * fs) = add4(bt[r][1], bt[r][2]);
    *reinterpret_cast<float4*>(dst + (4 * r + 2) * fs) = sub4(bt[r][2], bt[r][1]);
    *reinterpret_cast<float4*>(dst + (4 * r + 3) * fs) = sub4(bt[r][1], bt[r][3]);
  }
}

// one thread: (group, tile t, 4 couts)
template <int EPI>
__global__ __launch_bounds__(256) void wino_output_kernel(const WinoArgs a) {
  const int c4n = a.Cout >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.T * c4n) return;
  const int g = blockIdx.y;
  const int c4 = idx % c4n, t = idx / c4n;
  const int tpi = a.th * a.tw;
  const int n = t / tpi, rem = t - n * tpi;
  const int ty = rem / a.tw, tx = rem - ty * a.tw;
  const int Hp = a.H + 2, Wp = a.W + 2;
  const float* __restrict__ src = a.Mw + ((size_t)g * 16 * a.T + t) * a.Cout + c4 * 4;
  const size_t fs = (size_t)a.T * a.Cout;
  float4 u[2][4];  // A^T m
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float4 m0 = *reinterpret_cast<const float4*>(src + (0 + s) * fs);
    const float4 m1 = *reinterpret_cast<const float4*>(src + (4 + s) * fs);
    const float4 m2 = *reinterpret_cast<const float4*>(src + (8 + s) * fs);
    const float4 m3 = *reinterpret_cast<const float4*>(src + (12 + s) * fs);
    u[0][s] = add4(add4(m0, m1), m2);
    u[1][s] = sub4(sub4(m1, m2), m3);
  }
  const float4 b = *reinterpret_cast<const float4*>(a.bias + (size_t)g * a.bias_gs + c4 * 4);
  const float* __restrict__ res = (EPI == 1) ? a.res + (size_t)g * a.res_gs + c4 * 4 : nullptr;
  float* __restrict__ out = a.out + (size_t)g * a.out_gs + c4 * 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float4 y[2] = {add4(add4(u[i][0], u[i][1]), u[i][2]), sub4(sub4(u[i][1], u[i][2]), u[i][3])};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int oy = 2 * ty + i, ox = 2 * tx + j;
      if (oy >= a.H || ox >= a.W) continue;
      const size_t pix = (size_t)((n * Hp + oy + 1) * Wp + ox + 1);
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (EPI == 1) r = *reinterpret_cast<const float4*>(res + pix * a.res_ld);
      *reinterpret_cast<float4*>(out + pix * a.out_ld) = apply_epilogue<EPI>(y[j], b, r);
    }
  }
}

// packed [chunk][9][cout][32] -> U [chunk][16][cout][32], one thread per (chunk, cout, k)
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ packed, float* __restrict__ U,
                                                          int cout, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int k = idx & 31, co = (idx >> 5) % cout, ch = idx / (32 * cout);
  double g[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) g[r][s] = (double)packed[((size_t)(ch * 9 + r * 3 + s) * cout + co) * 32 + k];
  double gg[4][3];  // G g
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    gg[0][s] = g[0][s];
    gg[1][s] = 0.5 * (g[0][s] + g[1][s] + g[2][s]);
    gg[2][s] = 0.5 * (g[0][s] - g[1][s] + g[2][s]);
    gg[3][s] = g[2][s];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double u0 = gg[r][0], u1 = 0.5 * (gg[r][0] + gg[r][1] + gg[r][2]),
                 u2 = 0.5 * (gg[r][0] - gg[r][1] + gg[r][2]), u3 = gg[r][2];
    float* dst = U + ((size_t)(ch * 16 + 4 * r) * cout + co) * 32 + k;
    const size_t fs = (size_t)cout * 32;
    dst[0] = (float)u0; dst[fs] = (float)u1; dst[2 * fs] = (float)u2; dst[3 * fs] = (float)u3;
  }
}

// =================================================================================================
// M_b[T x Cout] = V_b[T x CIN] * U_b, b = 16 g + f.  128 rows x 128 couts per workgroup, K walked in
// 32-channel chunks, double-buffered LDS-DMA (one barrier per chunk), raw accumulators stored.
// Workgroup id -> (cout panel, row tile, b) with panel = id % panels: like the direct kernels an XCD
// (id % 8) only ever touches its own weight panel of the current b.
// =================================================================================================
template <int CIN>
__global__ __launch_bounds__(256, 2) void wino_gemm_kernel(const WinoArgs a) {
  constexpr int BM = 128, BN = 128, PT = 2, CT = 2;
  constexpr int NCH = CIN / 32;
  constexpr int BUF = (BM + BN) * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, hh = lane >> 5;

  const int panels = a.Cout / BN, mtiles = (a.T + BM - 1) / BM;
  const int nt = blockIdx.x % panels, rest = blockIdx.x / panels;
  const int mt = rest % mtiles, b = rest / mtiles;
  const int g = b >> 4, f = b & 15;
  const int m0 = mt * BM, n0 = nt * BN;
  const float* __restrict__ Vb = a.V + (size_t)b * a.T * CIN;
  const float* __restrict__ Ub = a.U + (size_t)g * a.u_gs + ((size_t)f * a.Cout + n0) * 32;
  const int mlast = a.T - 1;

  // staging: thread t fills LDS slot (t & 7) of rows (t >> 3) + 32 j with channel block (t & 7) ^ ((row >> 1) & 7)
  const int r0 = tid >> 3;
  const int c4 = (tid & 7) ^ ((r0 >> 1) & 7);
  unsigned pvoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) pvoff[j] = (unsigned)((min(m0 + r0 + 32 * j, mlast) * CIN + c4 * 4) * 4);
  const unsigned wvoff = (unsigned)((r0 * 32 + c4 * 4) * 4);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

#define ISSUE_TILE(CH, BUFI)                                                                         \
  {                                                                                                  \
    const float* pb_ = Vb + (CH) * 32;                                                               \
    const unsigned lb_ = lds0 + (unsigned)(((BUFI) * BUF + wid * 256) * 4);                          \
    glds16<0>(pb_, pvoff[0], lb_);                                                                   \
    glds16<0>(pb_, pvoff[1], lb_ + 4096);                                                            \
    glds16<0>(pb_, pvoff[2], lb_ + 8192);                                                            \
    glds16<0>(pb_, pvoff[3], lb_ + 12288);                                                           \
    const float* tb_ = Ub + (size_t)(CH) * 16 * a.Cout * 32;                                         \
    glds16<0>(tb_, wvoff, lb_ + BM * 128);                                                           \
    glds16<0>(tb_ + 1024, wvoff, lb_ + BM * 128 + 4096);                                             \
    glds16<0>(tb_ + 2048, wvoff, lb_ + BM * 128 + 8192);                                             \
    glds16<0>(tb_ + 3072, wvoff, lb_ + BM * 128 + 12288);                                            \
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

  // lane holds row l31 x couts {8 q + 4 hh + 0..3} of each 32 x 32 tile
  float* __restrict__ Mb = a.Mw + (size_t)b * a.T * a.Cout;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int m = m0 + (wm * PT + i) * 32 + l31;
    if (m > mlast) continue;
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = n0 + wn * CT * 32 + j * 32 + q * 8 + hh * 4;
        *reinterpret_cast<float4*>(Mb + (size_t)m * a.Cout + c) =
            make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
      }
  }
}

// ---- launchers ---------------------------------------------------------------------------------
hipError_t launch_wino_weights(const float* packed, float* U, int cin, int cout, hipStream_t st) {
  const int total = cin * cout;
  hipLaunchKernelGGL(wino_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, st, packed, U, cout, total);
  return hipGetLastError();
}

template <int CIN>
static hipError_t launch_gemm(const WinoArgs& a, hipStream_t st) {
  static bool attr = false;
  auto kern = wino_gemm_kernel<CIN>;
  const size_t lds = 2 * (128 + 128) * 32 * sizeof(float);
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr = true;
  }
  const int grid = (a.Cout / 128) * ((a.T + 127) / 128) * a.groups * 16;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_wino_conv(const WinoArgs& a, int epi, hipStream_t st) {
  if ((a.C != 256 && a.C != 512) || a.Cout % 128 != 0 || (epi != 0 && epi != 1)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(wino_input_kernel, dim3((a.T * (a.C >> 2) + 255) / 256, a.groups), dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = a.C == 256 ? launch_gemm<256>(a, st) : launch_gemm<512>(a, st);
  if (e != hipSuccess) return e;
  const dim3 og((a.T * (a.Cout >> 2) + 255) / 256, a.groups);
  if (epi == 0) hipLaunchKernelGGL(wino_output_kernel<0>, og, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(wino_output_kernel<1>, og, dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace se3tn
