// convAB1 (128 -> 256, stride 2), convAB2.conv1 / .conv2 (256 -> 256) and trans|rot conv1 (256 -> 1024, stride 2) at batch 1-2
// (se3_tracknet.py:64-76; the regime Tracker.on_track runs in) WITHOUT a K split: no partial sums, no reduction launch.
//
// At one pair these layers have 484 / 484 / 121 output pixels.  The split-K path cuts K over 12-24 workgroups per 128 x 128 tile
// and pays for it with a partial-sum round trip of 12-24 MB and a second launch (19-25 us per layer, EXPERIMENTS item 46).  Here a
// workgroup owns a 4 x 4 tile of outputs x 64 couts and walks the WHOLE K dimension:
//   * its input patch (6 x 6 or, stride 2, 9 x 9 pixels x all channels: 37-83 KB) is resident in LDS;
//   * the weights of its 64-cout slice stream through a ring of D 8-KB slots, one K-step (32 channels x one tap: 64 rows of the
//     packed panels as they are) per slot, D - 1 steps ahead of the matrix cores: one LDS-DMA instruction per thread per K-step, a
//     counted s_waitcnt + ONE barrier per K-step (the barrier that publishes K-step k also frees the slot of K-step k - 1);
//   * 8 waves: wave w = cout block w & 3 (16 couts) on the K-steps of parity w >> 2 -- the two halves of a block's sum meet in LDS at
//     the end (a fixed order: even steps + odd steps), v_mfma_f32_16x16x4_f32 as in conv64_small.hip (A = weights, B = pixels: a
//     lane ends with one pixel x 4 consecutive couts);
//   * the map edge: tile origins are clamped (4t -> min(4t, HO - 4)) so that every patch lies inside the stored zero-bordered input;
//     a clamped tile stores only the rows / columns it owns (the in-place residual of convAB2.conv2 must see each input once).
// 144 workgroups per pair (36 tiles x 4 slices; 9 x 16 for the head conv).  Float32 only; f16x3 keeps the split-K kernels.
#include "mfma_common.h"

namespace se3tn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// s_waitcnt vmcnt(n) for a wave-uniform n in [0, 12]
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<1>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 3: wait_vmcnt<3>(); break;
    case 4: wait_vmcnt<4>(); break;
    case 5: wait_vmcnt<5>(); break;
    case 6: wait_vmcnt<6>(); break;
    case 7: wait_vmcnt<7>(); break;
    case 8: wait_vmcnt<8>(); break;
    case 9: wait_vmcnt<9>(); break;
    case 10: wait_vmcnt<10>(); break;
    case 11: wait_vmcnt<11>(); break;
    default: wait_vmcnt<12>(); break;
  }
}

// CIN input channels, STRIDE 1 | 2, HO output rows = columns, D ring slots, EPI 0 bias + ReLU | 1 + residual | 2 bias + SELU
template <int CIN, int STRIDE, int HO, int D, int EPI>
__global__ __launch_bounds__(512, 1) void conv_small_stream_kernel(const ConvArgs a, int n, int slices) {
  constexpr int PW = 3 * STRIDE + 3;                 // patch edge: 6 | 9 input pixels
  constexpr int PP = PW * PW;
  constexpr int NCH = CIN / 32, KT = NCH * 9;
  constexpr int PATCH_FLOATS = PP * CIN;
  constexpr int SLOT_FLOATS = 64 * 32;
  constexpr int TT = (HO + 3) / 4;                   // tiles per row / column
  constexpr int HI = STRIDE == 1 ? HO : 2 * HO;      // input interior rows
  static_assert(D >= 3 && D <= 14, "ring depth");
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [patch][D slots]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wid & 3, par = wid >> 2;
  const int b = blockIdx.x;
  const int sl = b % slices, tile = (b / slices) % (TT * TT), img = (b / (slices * TT * TT)) % n, g = b / (slices * TT * TT * n);
  const int ty = tile / TT, tx = tile - ty * TT;
  const int oy = min(4 * ty, HO - 4), ox = min(4 * tx, HO - 4);
  const int Wp = HI + 2;
  const int cout = slices * 64, n0 = sl * 64;
  const float* __restrict__ in = a.in + (size_t)g * a.in_gs + ((size_t)(img * Wp + STRIDE * oy) * Wp + STRIDE * ox) * a.in_ld;
  const float* __restrict__ wgt = a.w + (size_t)g * a.w_gs + (size_t)n0 * 32;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

  // ---- the patch: PP pixels x CIN / 4 16-byte columns; physical column (within a 32-channel chunk) = logical ^ (pixel & 7)
  constexpr int PSLOTS = PP * (CIN / 4);
#pragma unroll
  for (int j = 0; j < (PSLOTS + 511) / 512; ++j) {
    const int slot = j * 512 + tid;
    if (slot < PSLOTS) {
      const int pp = slot / (CIN / 4), pc = slot - pp * (CIN / 4);
      const int col = (pc & ~7) | ((pc & 7) ^ (pp & 7));
      const int py = pp / PW, px = pp - py * PW;
      glds16<0>(in, (unsigned)(((py * Wp + px) * a.in_ld + col * 4) * 4), lds0 + (unsigned)((j * 512 + wid * 64) * 16));
    }
  }
  // ---- weights: K-step ks = rows n0..n0+63 of panel (chunk, tap); thread -> row tid >> 3, column (tid & 7) ^ ((row >> 1) & 7)
  const int r0 = tid >> 3;
  const unsigned wvoff = (unsigned)((r0 * 32 + (((tid & 7) ^ ((r0 >> 1) & 7)) * 4)) * 4);
  const unsigned ring0 = lds0 + (unsigned)(PATCH_FLOATS * 4 + wid * 1024);
  auto issue = [&](int ks) {
    glds16<0>(wgt + (size_t)ks * cout * 32, wvoff, ring0 + (unsigned)((ks % D) * SLOT_FLOATS * 4));
  };
#pragma unroll
  for (int ks = 0; ks < D - 1; ++ks) issue(ks);

  // fragment addresses
  const int pix = lane & 15, q = lane >> 4;
  const int row = cb * 16 + pix, rsw = (row >> 1) & 7;
  const int woff0 = row * 32 + ((q ^ rsw) << 2), woff1 = row * 32 + (((4 + q) ^ rsw) << 2);
  const int pbase = (STRIDE * (pix >> 2)) * PW + STRIDE * (pix & 3);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};

  int ch = 0, tap = 0;
#pragma unroll 1
  for (int ks = 0; ks < KT; ++ks) {
    // K-step ks has landed once at most the (D - 2) younger steps (fewer near the end) are outstanding
    const int younger = min(D - 2, KT - 1 - ks);
    if (younger == D - 2) wait_vmcnt<D - 2>();
    else wait_vmcnt_dyn(younger);
    __syncthreads();
    if (ks + D - 1 < KT) issue(ks + D - 1);             // into the slot K-step ks - 1 has just released
    if ((ks & 1) == par) {
      const int r = tap / 3, s = tap - r * 3;
      const int pp = pbase + r * PW + s;
      const float* px = smem + pp * CIN + ch * 32;
      const float* wt = smem + PATCH_FLOATS + (ks % D) * SLOT_FLOATS;
      const int sw = pp & 7;
      const float4 x0 = *reinterpret_cast<const float4*>(px + ((q ^ sw) << 2));
      const float4 x1 = *reinterpret_cast<const float4*>(px + (((4 + q) ^ sw) << 2));
      const float4 w0 = *reinterpret_cast<const float4*>(wt + woff0);
      const float4 w1 = *reinterpret_cast<const float4*>(wt + woff1);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, x0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, x1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, x0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, x1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, x0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, x1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, x0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, x1.w, acc1, 0, 0, 0);
    }
    if (++tap == 9) { tap = 0; ++ch; }
  }

  // ---- the two K-parities of a cout block meet: odd steps' sums through LDS (the patch area is free now), even + odd
  __syncthreads();
  float4 v = make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
  float4* xch = reinterpret_cast<float4*>(smem);
  if (par == 1) xch[cb * 64 + lane] = v;
  __syncthreads();
  if (par == 1) return;
  {
    const float4 u = xch[cb * 64 + lane];
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  // ---- epilogue: pixel (lane & 15) of the tile, couts n0 + 16 cb + 4 (lane >> 4) .. + 3; a clamped tile stores what it owns
  const int y = oy + (pix >> 2), x = ox + (pix & 3);
  if (y < 4 * ty || x < 4 * tx) return;
  const int c = n0 + cb * 16 + q * 4;
  const size_t opix = ((size_t)img * (HO + 2) + y + 1) * (HO + 2) + x + 1;
  const float4 bias = *reinterpret_cast<const float4*>(a.bias + (size_t)g * a.bias_gs + c);
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (EPI == 1) r = *reinterpret_cast<const float4*>(a.res + (size_t)g * a.res_gs + opix * a.res_ld + c);
  v = apply_epilogue<EPI>(v, bias, r);
  *reinterpret_cast<float4*>(a.out + (size_t)g * a.out_gs + opix * a.out_ld + c) = v;
}

template <int CIN, int STRIDE, int HO, int D, int EPI>
static hipError_t launch_one(const ConvArgs& a, int n, int cout, hipStream_t st) {
  constexpr int PW = 3 * STRIDE + 3;
  constexpr size_t lds = ((size_t)PW * PW * CIN + (size_t)D * 64 * 32) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = conv_small_stream_kernel<CIN, STRIDE, HO, D, EPI>;
  static PerDeviceOnce once;
  bool* done = once.current();
  if (!(done && *done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (done) *done = true;
  }
  constexpr int TT = (HO + 3) / 4;
  const int slices = cout / 64;
  hipLaunchKernelGGL(kern, dim3(a.groups * n * TT * TT * slices), dim3(512), lds, st, a, n, slices);
  return hipGetLastError();
}

// returns hipErrorNotSupported for a shape this family does not cover (the caller falls back to the split-K path)
hipError_t launch_conv_small_stream(const ConvArgs& a, int n, int cin, int cout, int stride, int epi, hipStream_t st) {
  if (cin == 128 && cout == 256 && stride == 2 && a.H == 44 && epi == 2) return launch_one<128, 2, 22, 12, 2>(a, n, cout, st);
  if (cin == 256 && cout == 256 && stride == 1 && a.H == 22 && epi == 0) return launch_one<256, 1, 22, 12, 0>(a, n, cout, st);
  if (cin == 256 && cout == 256 && stride == 1 && a.H == 22 && epi == 1) return launch_one<256, 1, 22, 12, 1>(a, n, cout, st);
  if (cin == 256 && cout == 1024 && stride == 2 && a.H == 22 && epi == 2) return launch_one<256, 2, 11, 9, 2>(a, n, cout, st);
  return hipErrorNotSupported;
}

}  // namespace se3tn
