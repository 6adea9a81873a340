// Stem + max-pool of both branches at batch 1-2 in ONE launch: Conv2d(4 -> 64, k = 7, s = 2, p = 3) + folded BN + SELU, then
// MaxPool2d(3, s = 2, p = 1) (se3_tracknet.py:57-58, 61-62; network_modules.py:59-66) -- the regime Tracker.on_track runs in.
//
// The batch-64 pair (stem7x7_slab_kernel: persistent 256-pixel tiles; maxpool3x3s2_kernel: a streaming pass over the 88 x 88 x 128
// map) occupies 62 of the 256 compute units at one pair and took 20 + 7 us there (EXPERIMENTS item 45).  Here a workgroup owns a
// 4 x 4 tile of POOL outputs of one branch: 121 tiles per image x 2 branches = 242 workgroups, one per CU.  It computes the 9 x 9
// stem outputs under its pool windows (27 % more than its share: the one-pixel halo), applies bias + SELU, keeps them in LDS and
// pools them from there: the 88 x 88 x 128 stem map is neither written nor read (4 MB each way per pair).
//   * v_mfma_f32_16x16x4_f32: k = 4 is exactly ONE tap's four channels = one 16-byte input pixel.  A operand = weights (wave w =
//     couts 16w..16w+15, the 49 taps' values of its lane live in registers for the whole kernel), B operand = pixels: one 4-byte
//     LDS read per MFMA, at a compile-time offset from the lane's patch address (the 23 x 23 input patch is stored as it lies);
//   * the packed stem weights are used as they are ([64][204]: slot e = tap (pair e / 2, half e % 2) of stem7x7_mfma.hip's pairing);
//   * two pixel blocks in flight (independent accumulators), 6 blocks of 16 cover the 81 stem pixels;
//   * edge tiles start their 9 x 9 window at stem row / column 0 instead of -1 (what lies outside the map is -inf for the pool: it
//     is simply not looked at), so every patch lies inside the stored (zero-bordered) input.
// Float32 only (f16x3 keeps the batch-64 pair); with se3tn_keep_intermediates the batch-64 pair runs, because it stores the map.
#include "mfma_common.h"

namespace se3tn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SP_IP = RES + 6;          // 182: padded input rows / columns
constexpr int SP_PW = 23;               // patch edge: 2 * 8 + 7
constexpr int SP_PATCH_BYTES = ((SP_PW * SP_PW * 16 + 1023) / 1024) * 1024;   // 9,216
constexpr int SP_WROW = 204;

struct StemPoolArgs {
  const float* in[2];   // [n,182,182,4] per branch
  const float* w;       // [2][64][204]
  const float* bias;    // [2][64]
  float* pool;          // [n,46,46,128] zero-bordered, branch A in channels 0-63, B in 64-127
  int n;
};

// tap (r, s) of weight slot e (weights.cpp: pack_stem)
__host__ __device__ constexpr int sp_tap_r(int e) { return e < 42 ? (e >> 1) / 3 : (e < 48 ? 2 * ((e >> 1) - 21) + (e & 1) : 6); }
__host__ __device__ constexpr int sp_tap_s(int e) { return e < 42 ? 2 * ((e >> 1) % 3) + (e & 1) : 6; }

// the 49 taps' B operands of two pixel blocks: one 4-byte LDS read each, at a compile-time offset from the lane's patch address.
// ALL of a block pair's reads are issued before its first MFMA (and the next pair's while this one computes): issued next to their
// use, each pair of MFMAs waited for an LDS round trip (EXPERIMENTS item 50: 14.2 us -> see there)
struct SpOperands {
  float x0[49], x1[49];
};
__device__ __forceinline__ void sp_load(SpOperands& o, const float* p0, const float* p1) {
#pragma unroll
  for (int e = 0; e < 49; ++e) {
    const int off = (sp_tap_r(e) * SP_PW + sp_tap_s(e)) * 4;
    o.x0[e] = p0[off];
    o.x1[e] = p1[off];
  }
}
// the MFMAs of one block pair, with the NEXT pair's operand reads interleaved one tap at a time and the PREVIOUS pair's bias + SELU
// (eight expm1f per lane: ~0.9 us of vector work per pair) spread over the first eight taps -- vector instructions issue while
// the matrix pipe works.  sched_barrier pins the order: left to itself the scheduler sinks every read to its use.
template <bool PREFETCH, bool EPI>
__device__ __forceinline__ void sp_mma(const float (&w)[49], const SpOperands& o, SpOperands& nxt, const float* n0, const float* n1,
                                       f32x4& a0, f32x4& a1, const f32x4& p0, const f32x4& p1, const float4& bias, float (&ov)[8]) {
  const float bv[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
  for (int e = 0; e < 49; ++e) {
    if (PREFETCH) {
      const int off = (sp_tap_r(e) * SP_PW + sp_tap_s(e)) * 4;
      nxt.x0[e] = n0[off];
      nxt.x1[e] = n1[off];
    }
    if (EPI && e < 8) ov[e] = selu_f((e < 4 ? p0[e & 3] : p1[e & 3]) + bv[e & 3]);
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], o.x0[e], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], o.x1[e], a1, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(256) void stem_pool_small_kernel(const StemPoolArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SP_PATCH_BYTES + 81 * 64 * 4];
  float* patch = reinterpret_cast<float*>(smem);
  float* stile = reinterpret_cast<float*>(smem + SP_PATCH_BYTES);     // [81 stem pixels][64 couts]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x, img = blockIdx.y, br = blockIdx.z;
  const int ti = tile / 11, tj = tile - ti * 11;
  const int R0 = ti == 0 ? 0 : 8 * ti - 1, C0 = tj == 0 ? 0 : 8 * tj - 1;   // first stem row / column of the 9 x 9 window
  const float* __restrict__ in = a.in[br] + ((size_t)img * SP_IP * SP_IP + (size_t)(2 * R0) * SP_IP + 2 * C0) * 4;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(patch));

  // ---- the 23 x 23 input patch (529 pixels of 16 bytes), as it lies
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int slot = j * 256 + tid;
    if (slot < SP_PW * SP_PW) {
      const int py = slot / SP_PW, px = slot - py * SP_PW;
      glds16<0>(in, (unsigned)((py * SP_IP + px) * 16), lds0 + (unsigned)((j * 256 + wid * 64) * 16));
    }
  }
  // ---- this lane's weights: cout 16 wid + (lane & 15), channel lane >> 4 of the 49 taps
  float w[49];
  {
    const float* wp = a.w + ((size_t)br * 64 + wid * 16 + (lane & 15)) * SP_WROW + (lane >> 4);
#pragma unroll
    for (int e = 0; e < 49; ++e) w[e] = wp[e * 4];
  }
  const int c = wid * 16 + (lane >> 4) * 4;              // the four couts this lane ends with
  const float4 bias = *reinterpret_cast<const float4*>(a.bias + br * 64 + c);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- 6 blocks of 16 stem pixels, two at a time; the operands of pair i + 1 are read while pair i computes
  const int kk = lane >> 4;
  auto patch_ptr = [&](int blk) -> const float* {
    const int p = min(blk * 16 + (lane & 15), 80);
    return patch + ((2 * (p / 9)) * SP_PW + 2 * (p % 9)) * 4 + kk;
  };
  SpOperands ops[2];
  sp_load(ops[0], patch_ptr(0), patch_ptr(1));
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc[3][2];
  float ov[8];
  auto store_pair = [&](int it) {             // bias + SELU of pair `it` is in ov: the two pixels' four couts each
    const int q0 = it * 32 + (lane & 15), q1 = q0 + 16;
    if (q0 < 81) *reinterpret_cast<float4*>(stile + q0 * 64 + c) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    if (q1 < 81) *reinterpret_cast<float4*>(stile + q1 * 64 + c) = make_float4(ov[4], ov[5], ov[6], ov[7]);
  };
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int pb = 2 * it;
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[it][0][e] = 0.f; acc[it][1][e] = 0.f; }
    if (it == 0)
      sp_mma<true, false>(w, ops[0], ops[1], patch_ptr(2), patch_ptr(3), acc[0][0], acc[0][1], acc[0][0], acc[0][1], bias, ov);
    else if (it == 1)
      sp_mma<true, true>(w, ops[1], ops[0], patch_ptr(pb + 2), patch_ptr(pb + 3), acc[1][0], acc[1][1], acc[0][0], acc[0][1], bias, ov);
    else
      sp_mma<false, true>(w, ops[0], ops[0], nullptr, nullptr, acc[2][0], acc[2][1], acc[1][0], acc[1][1], bias, ov);
    if (it > 0) store_pair(it - 1);
  }
  {
    const float bv[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) ov[e] = selu_f((e < 4 ? acc[2][0][e & 3] : acc[2][1][e & 3]) + bv[e & 3]);
    store_pair(2);
  }
  __syncthreads();

  // ---- MaxPool2d(3, 2, 1) of the tile: thread = (pool pixel, 4 couts); rows / columns outside the 88 x 88 map do not exist
  {
    const int pp = tid >> 4, c4 = (tid & 15) * 4;
    const int i = 4 * ti + (pp >> 2), j = 4 * tj + (pp & 3);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int sy = 2 * i + dy;
      if (sy < 0 || sy >= S1) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int sx = 2 * j + dx;
        if (sx < 0 || sx >= S1) continue;
        const float4 v = *reinterpret_cast<const float4*>(stile + ((sy - R0) * 9 + (sx - C0)) * 64 + c4);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    *reinterpret_cast<float4*>(a.pool + (((size_t)img * (S2 + 2) + i + 1) * (S2 + 2) + j + 1) * 128 + br * 64 + c4) = m;
  }
}

hipError_t launch_stem_pool_small(const float* inA, const float* inB, const float* w, const float* bias, float* pool, int n,
                                  hipStream_t st) {
  StemPoolArgs a;
  a.in[0] = inA; a.in[1] = inB; a.w = w; a.bias = bias; a.pool = pool; a.n = n;
  hipLaunchKernelGGL(stem_pool_small_kernel, dim3(121, n, 2), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace se3tn
