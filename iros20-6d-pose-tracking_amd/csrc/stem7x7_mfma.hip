// Stem: Conv2d(4->64, k=7, s=2, p=3) + folded BN + SELU for both branches (convA1 / convB1,
// se3_tracknet.py:57,61 -> network_modules.py:59-66), then MaxPool2d(3, s=2, p=1)
// (se3_tracknet.py:58,62) as a separate streaming kernel.
//
// Same exact-f32 MFMA implicit GEMM as conv3x3_mfma.hip with a different gather: the NHWC input
// has 4 channels = one 16-byte pixel, a K-step is one filter ROW r (7 taps x 4 channels = 28
// k's, k = s*4+c); 7 K-steps.  Thread t stages tap s = t&7 (s = 7 is an idle slot) of rows
// (t>>3)+32j: consecutive lanes read consecutive pixels (contiguous 112-byte runs).
// Output: [n,88,88,128] NHWC, branch A in channels 0-63, branch B in 64-127.
#include "se3tn_internal.h"

namespace se3tn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDK = 36;
constexpr float SELU_ALPHA = 1.6732632423543772848170429916717f;
constexpr float SELU_SCALE = 1.0507009873554804934193349852946f;
__device__ __forceinline__ float selu_s(float v) {
  return v > 0.f ? SELU_SCALE * v : (SELU_SCALE * SELU_ALPHA) * expm1f(v);
}

struct StemArgs {
  const float* in[2];  // [n,176,176,4] per branch
  const float* w;      // [2][7][64][32]
  const float* bias;   // [2][64]
  float* out;          // [n,88,88,128]
  int M;               // n*88*88
};

__global__ __launch_bounds__(256, 2) void stem7x7_mfma_kernel(const StemArgs a) {
  constexpr int BM = 128, BN = 64, PR = 4, WR = 2, CT = 2;
  constexpr int BUF = (BM + BN) * LDK;
  __shared__ __attribute__((aligned(16))) float smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int br = blockIdx.y;
  const int m0 = blockIdx.x * BM;
  const float* __restrict__ in = a.in[br];
  const float* __restrict__ wgt = a.w + (size_t)br * 7 * 64 * 32;

  const int s = tid & 7, r0 = tid >> 3;
  int rowoff[PR], hi0[PR];
  bool wok[PR];
#pragma unroll
  for (int j = 0; j < PR; ++j) {
    const int m = m0 + r0 + 32 * j;
    rowoff[j] = 0; hi0[j] = -1000; wok[j] = false;
    if (m < a.M) {
      const int n = m / (S1 * S1), rem = m - n * (S1 * S1);
      const int ho = rem / S1, wo = rem - ho * S1;
      const int h0 = 2 * ho - 3, w0 = 2 * wo - 3 + s;
      hi0[j] = h0;
      wok[j] = (s < 7) && ((unsigned)w0 < (unsigned)RES);
      rowoff[j] = ((n * RES + h0) * RES + w0) * 4;
    }
  }

  float4 ra[PR], rb[WR];
  auto load_tile = [&](int r) {
#pragma unroll
    for (int j = 0; j < PR; ++j) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (wok[j] && (unsigned)(hi0[j] + r) < (unsigned)RES)
        v = *reinterpret_cast<const float4*>(in + rowoff[j] + r * (RES * 4));
      ra[j] = v;
    }
    const float* wt = wgt + (size_t)r * (64 * 32) + tid * 4;
#pragma unroll
    for (int j = 0; j < WR; ++j) rb[j] = *reinterpret_cast<const float4*>(wt + j * 1024);
  };
  auto store_tile = [&](int buf) {
    float* dst = smem + buf * BUF + r0 * LDK + s * 4;
#pragma unroll
    for (int j = 0; j < PR; ++j) *reinterpret_cast<float4*>(dst + (32 * j) * LDK) = ra[j];
#pragma unroll
    for (int j = 0; j < WR; ++j) *reinterpret_cast<float4*>(dst + (BM + 32 * j) * LDK) = rb[j];
  };

  f32x16 acc[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int r = 0; r < 7; ++r) {
    const int buf = r & 1;
    if (r + 1 < 7) load_tile(r + 1);
    const float* pP = smem + buf * BUF + (wid * 32 + l31) * LDK + hh * 2;
    const float* pW = smem + buf * BUF + (BM + l31) * LDK + hh * 2;
#pragma unroll
    for (int kg = 0; kg < 7; ++kg) {  // 4 k's per group: lanes 0-31 take k, k+1; lanes 32-63 k+2, k+3
      const float2 pv = *reinterpret_cast<const float2*>(pP + kg * 4);
      float2 wv[CT];
#pragma unroll
      for (int j = 0; j < CT; ++j) wv[j] = *reinterpret_cast<const float2*>(pW + j * 32 * LDK + kg * 4);
#pragma unroll
      for (int j = 0; j < CT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j].x, pv.x, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < CT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j].y, pv.y, acc[j], 0, 0, 0);
    }
    if (r + 1 < 7) store_tile(buf ^ 1);
    __syncthreads();
  }

  const int m = m0 + wid * 32 + l31;
  if (m >= a.M) return;
  const float* __restrict__ bias = a.bias + br * 64;
  float* __restrict__ out = a.out + (size_t)m * 128 + br * 64;
#pragma unroll
  for (int j = 0; j < CT; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = j * 32 + q * 8 + hh * 4;
      const float4 b = *reinterpret_cast<const float4*>(bias + c);
      float4 v;
      v.x = selu_s(acc[j][4 * q + 0] + b.x);
      v.y = selu_s(acc[j][4 * q + 1] + b.y);
      v.z = selu_s(acc[j][4 * q + 2] + b.z);
      v.w = selu_s(acc[j][4 * q + 3] + b.w);
      *reinterpret_cast<float4*>(out + c) = v;
    }
}

hipError_t launch_stem(const float* inA, const float* inB, const float* w, const float* bias,
                       float* out, int n, hipStream_t st) {
  StemArgs a;
  a.in[0] = inA; a.in[1] = inB; a.w = w; a.bias = bias; a.out = out;
  a.M = n * S1 * S1;
  const dim3 grid((a.M + 127) / 128, 2);
  hipLaunchKernelGGL(stem7x7_mfma_kernel, grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

// MaxPool2d(kernel 3, stride 2, padding 1) on [n,88,88,128] -> the interior of the zero-bordered
// [n,46,46,128] tensor the 64-channel convs read; the pool's own padding is -inf (never wins), so
// its input keeps explicit bounds.  One thread per (output pixel, 4 channels).
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ in,
                                                            float* __restrict__ out, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c4 = idx & 31;
  const int pix = idx >> 5;
  const int n = pix / (S2 * S2), rem = pix - n * (S2 * S2);
  const int po = rem / S2, qo = rem - po * S2;
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int y = 2 * po + dy;
    if ((unsigned)y >= (unsigned)S1) continue;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int x = 2 * qo + dx;
      if ((unsigned)x >= (unsigned)S1) continue;
      const float4 v = *reinterpret_cast<const float4*>(in + ((size_t)(n * S1 + y) * S1 + x) * 128 + c4 * 4);
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
  }
  *reinterpret_cast<float4*>(out + ((size_t)(n * (S2 + 2) + po + 1) * (S2 + 2) + qo + 1) * 128 + c4 * 4) = m;
}

hipError_t launch_maxpool(const float* in, float* out, int n, hipStream_t st) {
  const int total = n * S2 * S2 * 32;
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((total + 255) / 256), dim3(256), 0, st, in, out, total);
  return hipGetLastError();
}

}  // namespace se3tn
