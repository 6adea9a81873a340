// Probe: what does the ds_read_b128 -> v_mfma_f32_32x32x2_f32 stream of the conv / GEMM kernels sustain on gfx950,
// as a function of HOW the fragment loads are placed around the MFMAs, of the barrier, and of the workgroup shape?
// Build: hipcc -O3 --offload-arch=gfx950 scripts/probes/mfma_stream.hip -o scripts/probes/bin/mfma_stream
// Each variant: 512 workgroups x 256 threads (2 per CU) or 256 x 512 threads, 128 x 128 (256 x 128) tile per workgroup,
// PT = CT = 2 blocks of 32 x 32 per wave, K-steps of 32 channels out of a double-buffered LDS image (random data).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, C, 0, 0, 0)

__device__ __forceinline__ void glds16(const float* sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
  const unsigned long long b_ = (unsigned long long)sbase;
  const unsigned blo_ = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)b_);
  const unsigned bhi_ = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b_ >> 32));
  const unsigned long long sb_ = ((unsigned long long)bhi_ << 32) | (unsigned long long)blo_;
  const unsigned lds_ = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:0\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff_bytes), "s"(lds_), "s"(sb_) : "memory");
}

// V 7: product structure incl. the global->LDS DMA of the next K-step's operand tile (8 x 1 KB per wave), source =
//      the same 32 KB per workgroup every K-step (L2-resident);  V 8: source streams through `big` (HBM / Infinity Cache)
constexpr bool has_loader(int V) { return V == 12 || V == 13; }

template <int V, int NW>
__global__ __launch_bounds__((NW + (has_loader(V) ? 1 : 0)) * 64, 2) void probe(float* out, const float* seed, int niter,
                                                                             const float* big = nullptr, size_t big_floats = 0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int ROWS = (NW == 4 ? 256 : 384);          // pixel rows + weight rows per buffer
  constexpr int BUF = ROWS * 32;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < ((V == 9 || V == 10) ? 3 : 2) * BUF; i += blockDim.x) smem[i] = seed[i & 4095];
  __syncthreads();
  const int l31 = lane & 31, hh = lane >> 5;
  const int wm = NW == 4 ? (wid >> 1) : (wid >> 1), wn = wid & 1;
  const int X = (l31 >> 1) & 7, lo = (hh ^ (X & 1)) * 4, xk = X >> 1;
  int fo[4];
  for (int g = 0; g < 4; ++g) fo[g] = ((g ^ xk) << 3) + lo;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float4 rp[2] = {make_float4(1.f, 2.f, 3.f, 4.f), make_float4(0.5f, 0.25f, 2.f, 1.f)}, rw[2] = {rp[1], rp[0]};
  const int PROWS = NW == 4 ? 128 : 256;
  for (int kt = 0; kt < niter; ++kt) {
    const float* pP = smem + (kt & 1) * BUF + (wm * 64 + l31) * 32;
    const float* pW = smem + (kt & 1) * BUF + (PROWS + wn * 64 + l31) * 32;
#define PX(G, I) *reinterpret_cast<const float4*>(pP + (I) * 1024 + fo[G])
#define WT(G, J) *reinterpret_cast<const float4*>(pW + (J) * 1024 + fo[G])
#define GROUP(P0, P1, W0, W1)                                                                       \
    MF(W0.x, P0.x, acc[0][0]); MF(W0.x, P1.x, acc[1][0]); MF(W1.x, P0.x, acc[0][1]); MF(W1.x, P1.x, acc[1][1]); \
    MF(W0.y, P0.y, acc[0][0]); MF(W0.y, P1.y, acc[1][0]); MF(W1.y, P0.y, acc[0][1]); MF(W1.y, P1.y, acc[1][1]); \
    MF(W0.z, P0.z, acc[0][0]); MF(W0.z, P1.z, acc[1][0]); MF(W1.z, P0.z, acc[0][1]); MF(W1.z, P1.z, acc[1][1]); \
    MF(W0.w, P0.w, acc[0][0]); MF(W0.w, P1.w, acc[1][0]); MF(W1.w, P0.w, acc[0][1]); MF(W1.w, P1.w, acc[1][1]);
    if (V == 7 || V == 8) {
      const unsigned lds0 = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) float*)smem;
      const float* src = V == 7 ? big + (size_t)blockIdx.x * 8192
                                : big + ((size_t)blockIdx.x * 8192 + (size_t)kt * 8192 * gridDim.x) % (big_floats - 8192);
      constexpr int NJ = NW == 4 ? 8 : 6;    // 32 KB (128 + 128 rows) | 48 KB (256 + 128 rows) per K-step and workgroup
#pragma unroll
      for (int j = 0; j < NJ; ++j)           // piece (j, wid): 1 KB at byte offset j * 1024 * NW + wid * 1024 of the tile
        glds16(src + j * 256 * NW, (unsigned)(lane * 16 + wid * 1024),
               lds0 + (unsigned)((((kt + 1) & 1) * BUF + j * 256 * NW + wid * 256) * 4));
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 p0 = PX(g, 0), p1 = PX(g, 1), w0 = WT(g, 0), w1 = WT(g, 1);
        GROUP(p0, p1, w0, w1)
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      continue;
    }
    if (has_loader(V)) {                 // one extra wave issues ALL the DMA; the compute waves only read LDS and multiply
      constexpr int NJ = NW == 4 ? 8 : 6;
      if (wid == NW) {
        const unsigned lds0 = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) float*)smem;
        const float* src = V == 12 ? big + (size_t)blockIdx.x * 12288
                                   : big + ((size_t)blockIdx.x * 12288 + (size_t)kt * 12288 * gridDim.x) % (big_floats - 12288);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int w = 0; w < NW; ++w)
            glds16(src + j * 256 * NW, (unsigned)(lane * 16 + w * 1024),
                   lds0 + (unsigned)((((kt + 1) & 1) * BUF + j * 256 * NW + w * 256) * 4));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 p0 = PX(g, 0), p1 = PX(g, 1), w0 = WT(g, 0), w1 = WT(g, 1);
          GROUP(p0, p1, w0, w1)
        }
      }
      __syncthreads();
      continue;
    }
    if (V == 11 || V == 14) {            // classical staging: global_load_dwordx4 -> VGPRs -> ds_write_b128
      constexpr int NJ = NW == 4 ? 8 : 6;
      const float* src = V == 11 ? big + (size_t)blockIdx.x * 12288
                                 : big + ((size_t)blockIdx.x * 12288 + (size_t)kt * 12288 * gridDim.x) % (big_floats - 12288);
      float4 st[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) st[j] = *reinterpret_cast<const float4*>(src + j * 256 * NW + wid * 256 + lane * 4);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 p0 = PX(g, 0), p1 = PX(g, 1), w0 = WT(g, 0), w1 = WT(g, 1);
        GROUP(p0, p1, w0, w1)
      }
      float* dst = smem + ((kt + 1) & 1) * BUF;
#pragma unroll
      for (int j = 0; j < NJ; ++j) *reinterpret_cast<float4*>(dst + j * 256 * NW + wid * 256 + lane * 4) = st[j];
      __syncthreads();
      continue;
    }
    if (V == 9 || V == 10) {             // 9: L2-resident src, 10: streaming; 3 buffers, DMA issued TWO K-steps ahead
      const unsigned lds0 = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) float*)smem;
      const float* src = V == 9 ? big + (size_t)blockIdx.x * 12288
                                : big + ((size_t)blockIdx.x * 12288 + (size_t)kt * 12288 * gridDim.x) % (big_floats - 12288);
      constexpr int NJ = NW == 4 ? 8 : 6;
      const int b2 = (kt + 2) % 3, b0 = kt % 3;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        glds16(src + j * 256 * NW, (unsigned)(lane * 16 + wid * 1024), lds0 + (unsigned)((b2 * BUF + j * 256 * NW + wid * 256) * 4));
      const float* qP = smem + b0 * BUF + (wm * 64 + l31) * 32;
      const float* qW = smem + b0 * BUF + (PROWS + wn * 64 + l31) * 32;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 p0 = *reinterpret_cast<const float4*>(qP + fo[g]), p1 = *reinterpret_cast<const float4*>(qP + 1024 + fo[g]);
        const float4 w0 = *reinterpret_cast<const float4*>(qW + fo[g]), w1 = *reinterpret_cast<const float4*>(qW + 1024 + fo[g]);
        GROUP(p0, p1, w0, w1)
      }
      if (NW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      __syncthreads();
      continue;
    }
    if (V == 0) {                       // registers only
      for (int g = 0; g < 4; ++g) { GROUP(rp[0], rp[1], rw[0], rw[1]) }
    } else if (V == 1 || V == 4) {      // as the product kernels: load a group's fragments, then its 16 MFMAs
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 p0 = PX(g, 0), p1 = PX(g, 1), w0 = WT(g, 0), w1 = WT(g, 1);
        GROUP(p0, p1, w0, w1)
      }
    } else if (V == 2 || V == 5) {      // all 16 fragment loads first, then 64 MFMAs
      float4 p0[4], p1[4], w0[4], w1[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) { p0[g] = PX(g, 0); p1[g] = PX(g, 1); w0[g] = WT(g, 0); w1[g] = WT(g, 1); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 4; ++g) { GROUP(p0[g], p1[g], w0[g], w1[g]) }
    } else if (V == 3 || V == 6) {      // software pipeline: group g+1's loads are issued before group g's MFMAs
      float4 a0 = PX(0, 0), a1 = PX(0, 1), b0 = WT(0, 0), b1 = WT(0, 1);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 n0, n1, m0, m1;
        if (g < 3) { n0 = PX(g + 1, 0); n1 = PX(g + 1, 1); m0 = WT(g + 1, 0); m1 = WT(g + 1, 1); }
        __builtin_amdgcn_sched_barrier(0);
        GROUP(a0, a1, b0, b1)
        __builtin_amdgcn_sched_barrier(0);
        if (g < 3) { a0 = n0; a1 = n1; b0 = m0; b1 = m1; }
      }
    }
    if (V >= 4) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  if (wid < NW) out[blockIdx.x * NW * 64 + tid] = s;
}

template <int V, int NW>
static void run(const char* name, float* out, const float* seed) {
  const int niter = 2000, grid = NW == 4 ? 512 : 256;
  const size_t lds = (size_t)(NW == 4 ? 256 : 384) * 32 * 4 * ((V == 9 || V == 10) ? 3 : 2);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<V, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  float* big = nullptr;
  const size_t big_floats = (size_t)256 << 20 >> 2 << 2;   // 1 GiB / 4 ... see below
  if (V >= 7) hipMalloc(&big, (size_t)1 << 30);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<V, NW>), dim3(grid), dim3((NW + (has_loader(V) ? 1 : 0)) * 64), lds, 0, out, seed, niter, (const float*)big, (size_t)(1 << 28));
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double flop = (double)grid * NW * niter * 64 * 4096.0;
  if (big) hipFree(big);
  printf("%-58s %8.3f ms  %7.1f TF  (%.1f %% of 157.3)\n", name, best, flop / best / 1e9, flop / best / 1e9 / 1.573);
}

int main() {
  float *out, *seed;
  hipMalloc(&out, 512 * 512 * 4);
  std::vector<float> h(4096);
  srand(1);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  hipMalloc(&seed, 4096 * 4);
  hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  run<0, 4>("4 waves x2/CU  registers only", out, seed);
  run<1, 4>("4 waves x2/CU  per-group loads (product kernels)", out, seed);
  run<2, 4>("4 waves x2/CU  16 loads up front", out, seed);
  run<3, 4>("4 waves x2/CU  software-pipelined groups", out, seed);
  run<4, 4>("4 waves x2/CU  per-group loads + barrier per K-step", out, seed);
  run<5, 4>("4 waves x2/CU  16 loads up front + barrier", out, seed);
  run<6, 4>("4 waves x2/CU  software-pipelined + barrier", out, seed);
  run<7, 4>("4 waves x2/CU  product structure + DMA from L2-resident src", out, seed);
  run<8, 4>("4 waves x2/CU  product structure + DMA streaming 1 GiB", out, seed);
  run<12, 4>("4+1 waves x2/CU  loader wave issues all DMA, L2-resident src", out, seed);
  run<13, 4>("4+1 waves x2/CU  loader wave issues all DMA, streaming", out, seed);
  run<11, 4>("4 waves x2/CU  global_load -> VGPR -> ds_write, L2-resident", out, seed);
  run<14, 4>("4 waves x2/CU  global_load -> VGPR -> ds_write, streaming", out, seed);
  run<12, 8>("8+1 waves x1/CU  loader wave issues all DMA, L2-resident src", out, seed);
  run<13, 8>("8+1 waves x1/CU  loader wave issues all DMA, streaming", out, seed);
  run<11, 8>("8 waves x1/CU  global_load -> VGPR -> ds_write, L2-resident", out, seed);
  run<14, 8>("8 waves x1/CU  global_load -> VGPR -> ds_write, streaming", out, seed);
  run<0, 8>("8 waves x1/CU  registers only", out, seed);
  run<9, 8>("8 waves x1/CU  3 buffers, DMA 2 K-steps ahead, L2-resident src", out, seed);
  run<10, 8>("8 waves x1/CU  3 buffers, DMA 2 K-steps ahead, streaming", out, seed);
  run<1, 8>("8 waves x1/CU  per-group loads", out, seed);
  run<4, 8>("8 waves x1/CU  per-group loads + barrier per K-step", out, seed);
  run<5, 8>("8 waves x1/CU  16 loads up front + barrier", out, seed);
  run<6, 8>("8 waves x1/CU  software-pipelined + barrier", out, seed);
  run<7, 8>("8 waves x1/CU  product structure + DMA from L2-resident src", out, seed);
  run<8, 8>("8 waves x1/CU  product structure + DMA streaming 1 GiB", out, seed);
  return 0;
}
