// The 64 -> 64, 3x3, stride-1 convs of the trunk (ResnetBasicBlock convA2 / convB2 / convB3, network_modules.py:86-120) at batch
// 1-5: the regime Tracker.on_track runs in (predict.py:416: one pair per frame, frames serial).
//
// At one pair a trunk conv is 1,936 pixels x 64 couts x K = 576: 0.14 GFLOP, 1.8 us of the matrix cores -- and the split-K path
// (conv3x3_splitk_kernel + conv_reduce_kernel) took 15-17 us for it: two launches, a 6 MB partial-sum round trip, three latency-bound
// K-steps per workgroup (EXPERIMENTS item 44).  This kernel needs neither a K split nor a reduction: 16-pixel tiles (4 x 4 outputs) x
// all 64 couts give 121 tiles per image and branch = 242 workgroups for the A|B pair, one per CU, each holding the WHOLE weight tensor
// of its branch (18 K-steps x 8 KB = 144 KB, the packed panels as they are) and its 6 x 6 x 64 input patch (9 KB) in LDS.
//   * matrix instruction: v_mfma_f32_16x16x4_f32 (exact float32, like the 32x32x2 form elsewhere); A operand = weights, B operand =
//     pixels, so a lane ends with ONE pixel x 4 consecutive couts (float4 bias / residual / store);  wave w = couts 16w..16w+15;
//   * LDS-DMA: the patch and four weight tiles are requested before the first K-step, tile ks + 4 from between the MFMAs of K-step
//     ks; K-step ks starts after `s_waitcnt vmcnt(what was requested after its tile)` + barrier, its operands are read one K-step
//     ahead of their MFMAs;
//   * a 16-byte LDS read feeds four MFMAs: the lane quarter q holds k = 4q..4q+3 of a 16-channel group, MFMA m takes element m of both
//     operands (a permutation of k inside the group -- the same on both sides);
//   * LDS images are XOR-swizzled like the other kernels' (weights: 16-byte column ^ ((row >> 1) & 7), applied on the DMA source
//     address; patch: column ^ (patch pixel & 7)): a fragment read puts at most two lanes of a 16-lane group on one bank group;
//   * two accumulators (the two 16-channel groups of a K-step) are added at the end: 144 dependent MFMAs would wait on each other;
//   * stored padding: the input carries its zero border, the patch is read without bounds logic; only interiors are stored.
// Float32 only; the f16x3 mode keeps the split-K kernels.  Results differ from the split-K path in the last bits (another summation
// order), are bitwise reproducible, and do not depend on the batch (every n uses the same tiles).
#include "mfma_common.h"

namespace se3tn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C64_PATCH_FLOATS = 36 * 64;            // 6 x 6 pixels x 64 channels
constexpr int C64_TILE_FLOATS = 64 * 32;             // one K-step of weights: 64 couts x 32 channels
constexpr int C64_KSTEPS = 18;                       // 2 channel chunks x 9 taps
constexpr size_t C64_LDS_BYTES = (size_t)(C64_PATCH_FLOATS + C64_KSTEPS * C64_TILE_FLOATS) * sizeof(float);   // 156,672

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct C64Frag {
  const float* smem;
  int wrow_off[2];    // weight fragment: float offset inside a K-step tile for 16-channel group 0 / 1
  int pix, q;         // lane's pixel (0..15) and quarter (0..3)
};

// DMA schedule: the patch and the first C64_AHEAD weight tiles are requested before the first K-step; K-step ks requests tile
// ks + C64_AHEAD between its MFMAs (the issue of an LDS-DMA instruction blocks the wave for ~35 cycles while the four waves share the
// address pipe: 153 KB issued up front kept the matrix pipe idle for 1.9 us of a 4 us kernel -- scripts/small_trace.py, EXPERIMENTS
// item 51).  c64_outstanding(s) = DMA instructions issued AFTER tile s's two at the moment K-step s synchronises (in the body of
// K-step s - 1, before that body's own request): what `s_waitcnt vmcnt` may leave in flight.
constexpr int C64_AHEAD = 4;
__host__ __device__ constexpr int c64_tiles_issued(int s) { return s == 0 ? C64_AHEAD : (s + C64_AHEAD - 1 < C64_KSTEPS ? s + C64_AHEAD - 1 : C64_KSTEPS); }
__host__ __device__ constexpr int c64_outstanding(int s) { return 2 * (c64_tiles_issued(s) - (s + 1)); }
__host__ __device__ constexpr bool c64_schedule_ok() {
  for (int s = 0; s < C64_KSTEPS; ++s)
    if (c64_outstanding(s) < 0) return false;
  return true;
}
static_assert(c64_schedule_ok(), "a K-step would wait for a weight tile that has not been requested");

struct C64Dma {        // this thread's part of a weight tile: 16 bytes of rows tid >> 3 and 32 + (tid >> 3)
  const float* wgt;
  unsigned wvoff, wl;
};
template <int KS>
__device__ __forceinline__ void c64_request_tile(const C64Dma& d) {
  glds16<0>(d.wgt + (size_t)KS * C64_TILE_FLOATS, d.wvoff, d.wl + (unsigned)(KS * C64_TILE_FLOATS * 4));
  glds16<0>(d.wgt + (size_t)KS * C64_TILE_FLOATS + 1024, d.wvoff, d.wl + (unsigned)(KS * C64_TILE_FLOATS * 4 + 4096));
}

struct C64Ops {
  float4 x0, x1, w0, w1;
};

template <int KS>
__device__ __forceinline__ void c64_sync() {
  wait_vm<c64_outstanding(KS)>();
  __syncthreads();
}

template <int KS>
__device__ __forceinline__ void c64_load(const C64Frag& f, C64Ops& o) {
  constexpr int ch = KS / 9, tap = KS % 9, r = tap / 3, s = tap % 3;
  const int pp = ((f.pix >> 2) + r) * 6 + (f.pix & 3) + s;            // patch pixel of this lane's output pixel under tap (r, s)
  const float* px = f.smem + pp * 64 + ch * 32;
  const float* wt = f.smem + C64_PATCH_FLOATS + KS * C64_TILE_FLOATS;
  const int sw = pp & 7;
  o.x0 = *reinterpret_cast<const float4*>(px + ((f.q ^ sw) << 2));
  o.x1 = *reinterpret_cast<const float4*>(px + (((4 + f.q) ^ sw) << 2));
  o.w0 = *reinterpret_cast<const float4*>(wt + f.wrow_off[0]);
  o.w1 = *reinterpret_cast<const float4*>(wt + f.wrow_off[1]);
}

template <int HALF>
__device__ __forceinline__ void c64_mma(const C64Ops& o, f32x4& acc0, f32x4& acc1) {
  if (HALF == 0) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.w0.x, o.x0.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.w1.x, o.x1.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.w0.y, o.x0.y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.w1.y, o.x1.y, acc1, 0, 0, 0);
  } else {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.w0.z, o.x0.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.w1.z, o.x1.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.w0.w, o.x0.w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.w1.w, o.x1.w, acc1, 0, 0, 0);
  }
}

// K-step KS: wait + barrier for tile KS + 1, its operands read, THEN the MFMAs of KS with the request for tile KS + C64_AHEAD in
// their middle.  sched_barrier pins that order (the scheduler would sink the reads to their use).
template <int KS>
struct C64Run {
  static __device__ __forceinline__ void go(const C64Frag& f, const C64Dma& d, C64Ops& o0, C64Ops& o1, f32x4& a0, f32x4& a1) {
    C64Run<KS - 1>::go(f, d, o0, o1, a0, a1);
    if constexpr (KS + 1 < C64_KSTEPS) {
      c64_sync<KS + 1>();
      c64_load<KS + 1>(f, (KS & 1) ? o0 : o1);
    }
    __builtin_amdgcn_sched_barrier(0);
    c64_mma<0>((KS & 1) ? o1 : o0, a0, a1);
    if constexpr (KS + C64_AHEAD < C64_KSTEPS) c64_request_tile<KS + C64_AHEAD>(d);
    c64_mma<1>((KS & 1) ? o1 : o0, a0, a1);
    __builtin_amdgcn_sched_barrier(0);
  }
};
template <>
struct C64Run<-1> {
  static __device__ __forceinline__ void go(const C64Frag& f, const C64Dma&, C64Ops& o0, C64Ops&, f32x4&, f32x4&) {
    c64_sync<0>();
    c64_load<0>(f, o0);
  }
};

// grid: (121 tiles, n images, groups) workgroups of 256 threads (no division by a run-time value in the index arithmetic: this
// kernel lives for 4 us); EPI 0 = bias + ReLU, 1 = bias + residual + ReLU
template <int EPI>
__global__ __launch_bounds__(256, 1) void conv64_small_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x, img = blockIdx.y, g = blockIdx.z;
  const int ty = tile / 11, tx = tile - ty * 11;
  constexpr int Wp = S2 + 2;                            // 46
  const float* __restrict__ in = a.in + (size_t)g * a.in_gs + ((size_t)(img * Wp + ty * 4) * Wp + tx * 4) * a.in_ld;   // patch origin (padded)
  const float* __restrict__ wgt = a.w + (size_t)g * a.w_gs;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

  // ---- request everything: the patch (576 16-byte slots: pixel = slot >> 4, physical column = slot & 15) ...
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (j == 2 && wid != 0) break;                      // slots 512..575: the first wave only
    const int slot = j * 256 + tid;
    const int pp = slot >> 4, pc = slot & 15;
    const int col = (pc & 8) | ((pc & 7) ^ (pp & 7));   // logical 16-byte column that belongs in this physical slot
    const int py = pp / 6, pxx = pp - py * 6;
    const unsigned voff = (unsigned)(((py * Wp + pxx) * a.in_ld + col * 4) * 4);
    glds16<0>(in, voff, lds0 + (unsigned)((j * 256 + wid * 64) * 16));
  }
  // ... then the first weight tiles (the packed panels [chunk][tap][64 couts][32]: 8 KB each, row r, column (tid & 7) ^ ((r >> 1) & 7));
  // the rest are requested from inside the K-steps
  C64Dma dma;
  {
    const int r0 = tid >> 3;
    const int c4 = (tid & 7) ^ ((r0 >> 1) & 7);
    dma.wgt = wgt;
    dma.wvoff = (unsigned)((r0 * 32 + c4 * 4) * 4);
    dma.wl = lds0 + (unsigned)(C64_PATCH_FLOATS * 4 + wid * 1024);
  }
  c64_request_tile<0>(dma); c64_request_tile<1>(dma); c64_request_tile<2>(dma); c64_request_tile<3>(dma);
  static_assert(C64_AHEAD == 4, "the prologue requests tiles 0 .. C64_AHEAD - 1");

  C64Frag f;
  f.smem = smem;
  f.pix = lane & 15;
  f.q = lane >> 4;
  {
    const int row = wid * 16 + (lane & 15), sw = (row >> 1) & 7;
    f.wrow_off[0] = row * 32 + ((f.q ^ sw) << 2);
    f.wrow_off[1] = row * 32 + (((4 + f.q) ^ sw) << 2);
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  C64Ops o0, o1;
  C64Run<C64_KSTEPS - 1>::go(f, dma, o0, o1, acc0, acc1);

  // ---- epilogue: this lane = pixel (lane & 15), couts 16 wid + 4 (lane >> 4) .. + 3
  const int c = wid * 16 + (lane >> 4) * 4;
  const int ho = ty * 4 + (f.pix >> 2), wo = tx * 4 + (f.pix & 3);
  const size_t opix = ((size_t)img * Wp + ho + 1) * Wp + wo + 1;
  float4 v = make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
  const float4 bias = *reinterpret_cast<const float4*>(a.bias + (size_t)g * a.bias_gs + c);
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (EPI == 1) r = *reinterpret_cast<const float4*>(a.res + (size_t)g * a.res_gs + opix * a.res_ld + c);
  v = apply_epilogue<EPI>(v, bias, r);
  *reinterpret_cast<float4*>(a.out + (size_t)g * a.out_gs + opix * a.out_ld + c) = v;
}

hipError_t launch_conv64_small(const ConvArgs& a, int n, int epi, hipStream_t st) {
  static PerDeviceOnce attr0, attr1;
  auto k0 = conv64_small_kernel<0>;
  auto k1 = conv64_small_kernel<1>;
  bool* done = (epi == 1 ? attr1 : attr0).current();
  if (!(done && *done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(epi == 1 ? k1 : k0), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)C64_LDS_BYTES);
    if (e != hipSuccess) return e;
    if (done) *done = true;
  }
  const dim3 grid(121, n, a.groups);
  if (epi == 1) hipLaunchKernelGGL(k1, grid, dim3(256), C64_LDS_BYTES, st, a);
  else hipLaunchKernelGGL(k0, grid, dim3(256), C64_LDS_BYTES, st, a);
  return hipGetLastError();
}

}  // namespace se3tn
