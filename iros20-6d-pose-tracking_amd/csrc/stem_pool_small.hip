// Stem + max-pool of both branches at batch 1-2 in ONE launch: Conv2d(4 -> 64, k = 7, s = 2, p = 3) + folded BN + SELU, then
// MaxPool2d(3, s = 2, p = 1) (se3_tracknet.py:57-58, 61-62; network_modules.py:59-66) -- the regime Tracker.on_track runs in.
//
// The batch-64 pair (stem7x7_slab_kernel: persistent 256-pixel tiles; maxpool3x3s2_kernel: a streaming pass over the 88 x 88 x 128
// map) occupies 62 of the 256 compute units at one pair and took 20 + 7 us there (EXPERIMENTS item 45).  Here a workgroup owns a
// 4 x 4 tile of POOL outputs of one branch: 121 tiles per image x 2 branches = 242 workgroups, one per CU.  It computes the 9 x 9
// stem outputs under its pool windows (27 % more than its share: the one-pixel halo), applies bias + SELU, keeps them in LDS and
// pools them from there: the 88 x 88 x 128 stem map is neither written nor read (4 MB each way per pair).
//   * v_mfma_f32_16x16x4_f32 with k = FOUR TAPS of one channel: weight slots 4 g .. 4 g + 3 (13 groups cover the 49 taps; the three
//     slots past the last tap carry zeros).  Lane (i, q) of the A operand holds cout i, slot 4 g + q: its four channels are ONE
//     16-byte load from the packed row ([64][204]: slot e = tap (pair e / 2, half e % 2) of stem7x7_mfma.hip's pairing, used as it
//     is); lane (j, q) of the B operand reads pixel j under THAT slot's tap: one 16-byte LDS read = the pixel's four channels.
//     MFMA c of the group takes component c of both.  13 loads per lane for the weights and 13 LDS reads per 16-pixel block instead
//     of 49 + 49 four-byte ones: the first version spent 3.5 of its 14 us issuing loads (scripts/small_trace.py, EXPERIMENTS item 53);
//   * two pixel blocks in flight (independent accumulators), 6 blocks of 16 cover the 81 stem pixels;
//   * edge tiles start their 9 x 9 window at stem row / column 0 instead of -1 (what lies outside the map is -inf for the pool: it
//     is simply not looked at), so every patch lies inside the stored (zero-bordered) input.
// Float32 only (f16x3 keeps the batch-64 pair); with se3tn_keep_intermediates the batch-64 pair runs, because it stores the map.
#include "mfma_common.h"

namespace se3tn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#if defined(SE3TN_SMALL_TRACE)   // developer build: phase stamps (100 MHz) of every workgroup, read back by scripts/small_trace.py
static __device__ unsigned long long g_sp_trace[512][8];
#define SP_TRACE(P)                                                                                      \
  {                                                                                                      \
    const unsigned lb_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                 \
    if (threadIdx.x == 0 && lb_ < 512) g_sp_trace[lb_][P] = wall_clock64();                              \
  }
#else
#define SP_TRACE(P)
#endif

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

constexpr int SP_GROUPS = 13;     // groups of four weight slots: 52 slots, 49 taps
// float offset (inside the patch) of slot e's tap; slots 49 .. 51 do not exist: offset 0, their weights are forced to zero
__host__ __device__ constexpr int sp_slot_off(int e) { return e < 49 ? (sp_tap_r(e) * SP_PW + sp_tap_s(e)) * 4 : 0; }

// SELU with a short expm1: on (-0.5, 0] the Taylor series to v^9 (truncation < 3e-10 relative), below exp(v) - 1 through the
// hardware exponential (the difference of 1 is at least 0.39 there: no cancellation).  ~17 vector instructions instead of the ~50
// of expm1f -- vector work does not hide under this wave's own MFMAs (measured: the block pairs took MFMA time + SELU time), and
// a lane computes 24 of these.  Within 2e-7 relative of expm1f; used by this kernel only.
__device__ __forceinline__ float sp_selu(float v) {
  const float t = v * (1.f / 9.f);
  float p = fmaf(t, 1.f, 1.f);                       // 1 + v/9
  p = fmaf(p * v, 1.f / 8.f, 1.f);
  p = fmaf(p * v, 1.f / 7.f, 1.f);
  p = fmaf(p * v, 1.f / 6.f, 1.f);
  p = fmaf(p * v, 1.f / 5.f, 1.f);
  p = fmaf(p * v, 1.f / 4.f, 1.f);
  p = fmaf(p * v, 1.f / 3.f, 1.f);
  p = fmaf(p * v, 1.f / 2.f, 1.f);
  const float small = p * v;                         // v + v^2/2 + ... + v^9/9!
  const float big = __expf(v) - 1.f;
  const float em1 = v > -0.5f ? small : big;
  return v > 0.f ? SELU_SCALE * v : (SELU_SCALE * SELU_ALPHA) * em1;
}

// B operands of two 16-pixel blocks: per group one 16-byte read each, at this lane's slot offset
struct SpOperands {
  float4 x0[SP_GROUPS], x1[SP_GROUPS];
};
__device__ __forceinline__ void sp_load(SpOperands& o, const float* p0, const float* p1, const int (&off)[SP_GROUPS]) {
#pragma unroll
  for (int g = 0; g < SP_GROUPS; ++g) {
    o.x0[g] = *reinterpret_cast<const float4*>(p0 + off[g]);
    o.x1[g] = *reinterpret_cast<const float4*>(p1 + off[g]);
  }
}
// the MFMAs of one block pair (13 groups x 4 channels x 2 blocks), with the NEXT pair's operand reads interleaved one group at a
// time and the PREVIOUS pair's bias + SELU spread over the first eight groups.  sched_barrier pins the order: left to itself the
// scheduler sinks every read to its use.
template <bool PREFETCH, bool EPI>
__device__ __forceinline__ void sp_mma(const float4 (&w)[SP_GROUPS], const SpOperands& o, SpOperands& nxt, const float* n0, const float* n1,
                                       const int (&off)[SP_GROUPS], f32x4& a0, f32x4& a1, const f32x4& p0, const f32x4& p1,
                                       const float4& bias, float (&ov)[8]) {
  const float bv[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
  for (int g = 0; g < SP_GROUPS; ++g) {
    if (PREFETCH) {
      nxt.x0[g] = *reinterpret_cast<const float4*>(n0 + off[g]);
      nxt.x1[g] = *reinterpret_cast<const float4*>(n1 + off[g]);
    }
    if (EPI && g < 8) ov[g] = sp_selu((g < 4 ? p0[g & 3] : p1[g & 3]) + bv[g & 3]);
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g].x, o.x0[g].x, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g].x, o.x1[g].x, a1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g].y, o.x0[g].y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g].y, o.x1[g].y, a1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g].z, o.x0[g].z, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g].z, o.x1[g].z, a1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g].w, o.x0[g].w, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g].w, o.x1[g].w, a1, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(256) void stem_pool_small_kernel(const StemPoolArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SP_PATCH_BYTES + 81 * 64 * 4];
  float* patch = reinterpret_cast<float*>(smem);
  float* stile = reinterpret_cast<float*>(smem + SP_PATCH_BYTES);     // [81 stem pixels][64 couts]
  SP_TRACE(0)
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
  // ---- this lane's weights: cout 16 wid + (lane & 15), slots 4 g + (lane >> 4): 13 loads of 16 bytes (the slots past tap 48 are zero)
  const int kk = lane >> 4;
  float4 w[SP_GROUPS];
  {
    const float* wp = a.w + ((size_t)br * 64 + wid * 16 + (lane & 15)) * SP_WROW + kk * 4;
#pragma unroll
    for (int g = 0; g < SP_GROUPS - 1; ++g) w[g] = *reinterpret_cast<const float4*>(wp + g * 16);
    w[SP_GROUPS - 1] = kk == 0 ? *reinterpret_cast<const float4*>(wp + (SP_GROUPS - 1) * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // ... and its slots' tap offsets inside the patch
  int off[SP_GROUPS];
#pragma unroll
  for (int g = 0; g < SP_GROUPS; ++g)
    off[g] = kk == 0 ? sp_slot_off(4 * g) : kk == 1 ? sp_slot_off(4 * g + 1) : kk == 2 ? sp_slot_off(4 * g + 2) : sp_slot_off(4 * g + 3);
  const int c = wid * 16 + (lane >> 4) * 4;              // the four couts this lane ends with
  const float4 bias = *reinterpret_cast<const float4*>(a.bias + br * 64 + c);
  SP_TRACE(1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  SP_TRACE(2)

  // ---- 6 blocks of 16 stem pixels, two at a time; the operands of pair i + 1 are read while pair i computes
  auto patch_ptr = [&](int blk) -> const float* {
    const int p = min(blk * 16 + (lane & 15), 80);
    return patch + ((2 * (p / 9)) * SP_PW + 2 * (p % 9)) * 4;
  };
  SpOperands ops[2];
  sp_load(ops[0], patch_ptr(0), patch_ptr(1), off);
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
      sp_mma<true, false>(w, ops[0], ops[1], patch_ptr(2), patch_ptr(3), off, acc[0][0], acc[0][1], acc[0][0], acc[0][1], bias, ov);
    else if (it == 1)
      sp_mma<true, true>(w, ops[1], ops[0], patch_ptr(pb + 2), patch_ptr(pb + 3), off, acc[1][0], acc[1][1], acc[0][0], acc[0][1], bias, ov);
    else
      sp_mma<false, true>(w, ops[0], ops[0], nullptr, nullptr, off, acc[2][0], acc[2][1], acc[1][0], acc[1][1], bias, ov);
    if (it > 0) store_pair(it - 1);
#if defined(SE3TN_SMALL_TRACE)
    if (acc[it][0][0] == 1.2345e-30f) return;
    if (it == 0) { SP_TRACE(3) } else if (it == 1) { SP_TRACE(4) } else { SP_TRACE(5) }
#endif
  }
  {
    const float bv[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) ov[e] = sp_selu((e < 4 ? acc[2][0][e & 3] : acc[2][1][e & 3]) + bv[e & 3]);
    store_pair(2);
  }
  __syncthreads();
  SP_TRACE(6)

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
#if defined(SE3TN_SMALL_TRACE)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SP_TRACE(7)
#endif
}

#if defined(SE3TN_SMALL_TRACE)
extern "C" int se3tn_debug_trace_stem(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sp_trace), sizeof(unsigned long long) * 512 * 8);
}
#endif

hipError_t launch_stem_pool_small(const float* inA, const float* inB, const float* w, const float* bias, float* pool, int n,
                                  hipStream_t st) {
  StemPoolArgs a;
  a.in[0] = inA; a.in[1] = inB; a.w = w; a.bias = bias; a.pool = pool; a.n = n;
  hipLaunchKernelGGL(stem_pool_small_kernel, dim3(121, n, 2), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace se3tn
