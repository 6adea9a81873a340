// Hardware probe: how fast can ONE workgroup per compute unit fill its LDS from an L2-resident buffer (the start-up of the
// batch-1 kernels: conv64_small 153 KB, conv_slices_small 66-127 KB per workgroup)?
//   mode 0: empty kernel (launch floor)        mode 1: LDS-DMA, global_load_lds_dwordx4, everything issued up front
//   mode 2: LDS-DMA with the nt policy         mode 3: global_load_dwordx4 into registers, ds_write_b128 (12 loads in flight)
//   mode 4: like 1, every workgroup reads ITS OWN 144 KB (no sharing in L2: MALL / HBM stream)
// prints the kernel time (hipEvent, mean of 50 launches) and the per-CU / chip fill rate after subtracting mode 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int N = 36;                 // 16-byte slots per thread: 256 threads x 36 x 16 B = 144 KB
__device__ __forceinline__ void glds(const float* sbase, unsigned voff, unsigned lds, bool nt) {
  unsigned keep;
  if (nt)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds), "s"(sbase) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds), "s"(sbase) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* g, float* o, size_t wg_stride) {
  extern __shared__ __attribute__((aligned(16))) float s[];
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* src = g + (size_t)blockIdx.x * wg_stride;
  const unsigned long long c0 = wall_clock64();
  if (MODE == 1 || MODE == 2 || MODE == 4) {
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(const __attribute__((address_space(3))) float*)s);
#pragma unroll
    for (int j = 0; j < N; ++j) glds(src + j * 1024, (unsigned)(tid * 16), lds + (unsigned)((j * 256 + wid * 64) * 16), MODE == 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (MODE == 3) {
#pragma unroll
    for (int b = 0; b < N; b += 12) {
      float4 v[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) v[j] = *reinterpret_cast<const float4*>(src + (b + j) * 1024 + tid * 4);
#pragma unroll
      for (int j = 0; j < 12; ++j) *reinterpret_cast<float4*>(s + (b + j) * 1024 + tid * 4) = v[j];
    }
  }
  __syncthreads();
  const unsigned long long c1 = wall_clock64();
  if (tid == 0) o[blockIdx.x] = (float)(c1 - c0);                          // 100 MHz ticks from kernel entry to "LDS full"
  if (MODE != 0 && s[tid * 7] == 123456.f) o[blockIdx.x] = s[tid];       // (keeps the fill alive)
}
static float g_ticks_mean, g_ticks_max;
template <int MODE>
float run(const float* g, float* o, size_t stride, int grid) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 150 * 1024, 0, g, o, stride);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 150 * 1024, 0, g, o, stride);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  float h[256]; hipMemcpy(h, o, grid * 4, hipMemcpyDeviceToHost);
  g_ticks_mean = 0; g_ticks_max = 0;
  for (int i = 0; i < grid; ++i) { g_ticks_mean += h[i] / grid; if (h[i] > g_ticks_max) g_ticks_max = h[i]; }
  return ms / 200 * 1000.f;
}
int main() {
  const int grid = 256;
  const size_t words = (size_t)N * 1024;
  float *g, *o;
  hipMalloc(&g, words * 4 * grid); hipMalloc(&o, grid * 4);
  hipMemset(g, 0, words * 4 * grid);
  const double kb = words * 4 / 1024.0;
  const char* nm[] = {"empty kernel", "LDS-DMA, shared source (L2)", "LDS-DMA nt, shared source", "registers + ds_write, shared source", "LDS-DMA, private source (MALL/HBM)"};
  float t[5], tm[5], tx[5];
  t[0] = run<0>(g, o, 0, grid); tm[0] = g_ticks_mean; tx[0] = g_ticks_max;
  t[1] = run<1>(g, o, 0, grid); tm[1] = g_ticks_mean; tx[1] = g_ticks_max;
  t[2] = run<2>(g, o, 0, grid); tm[2] = g_ticks_mean; tx[2] = g_ticks_max;
  t[3] = run<3>(g, o, 0, grid); tm[3] = g_ticks_mean; tx[3] = g_ticks_max;
  t[4] = run<4>(g, o, words, grid); tm[4] = g_ticks_mean; tx[4] = g_ticks_max;
  for (int i = 0; i < 5; ++i)
    printf("%-38s %.2f us per launch (200 back to back); in-kernel entry -> LDS full: mean %.2f us, slowest workgroup %.2f us -> %.0f KB at %.1f GB/s per CU\n",
           nm[i], t[i], tm[i] / 100.f, tx[i] / 100.f, kb, i ? kb * 1024 / (tm[i] / 100.f * 1e3) : 0.0);
  return 0;
}
