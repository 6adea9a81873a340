// Hardware probe for VERDICT r5 #2(a): "consumer-side split-K reduction" at batch 1 -- the NEXT conv_slices_small (or the tail) sums
// the 4-8 partial slices of its producer (+ bias / activation) while staging its input patch through registers into LDS, instead of
// a conv_reduce_kernel launch in between.  What it would save per layer: one conv_reduce_kernel (4.4-4.8 us in the kernel trace) + one
// launch boundary (~1.8-2.4 us).  What it costs: every consumer workgroup re-reduces ITS patch, so each partial sum is read by
// (n-tiles of the consumer) x (halo overlap) workgroups instead of once.  This probe measures that cost: for each producer -> consumer
// edge of the batch-1 forward it runs the consumer's grid (one round of 128-256 workgroups of 256 threads) with
//   mode 0  the prologue as shipped: the patch [PP pixels x CH channels] DMA'd global -> LDS from the reduced activation
//   mode 1  the consumer-side prologue: SL partial patches read with global_load_dwordx4 (8 in flight per thread), summed in slice order,
//           bias + ReLU, ds_write_b128 into the same LDS image
// and prints kernel time (hipEvent, mean of 200 back-to-back launches; the buffers stay cache-resident: the OPTIMISTIC case for mode 1)
// and the in-kernel "entry -> patch complete" time from wall_clock64 (mean over the workgroups).
//   hipcc --offload-arch=gfx950 -O3 -o consumer_reduce consumer_reduce.hip && ./consumer_reduce
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Edge {
  const char* name;
  int PP, CH, SL;       // consumer patch pixels, channels per consumer slice, producer slices
  int M, COUT;          // producer output: pixels, channels (partial tensor [SL][M][COUT])
  int wgs;              // consumer workgroups per pair
  int cons_slices;      // consumer's own channel slices (channel offset = (wg % cons_slices) * CH)
};

__device__ __forceinline__ void glds16(const float* sbase, unsigned voff, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds), "s"(sbase) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const float* __restrict__ act, const float* __restrict__ part, const float* __restrict__ bias,
                                                float* out, int PP, int CH, int SL, int M, int COUT, int cons_slices) {
  extern __shared__ __attribute__((aligned(16))) float s[];
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sl = blockIdx.x % cons_slices, rest = blockIdx.x / cons_slices;
  const int choff = sl * CH;
  const int m0 = (rest * 37) % M;                       // where this workgroup's patch starts in the producer's map
  const int q = CH / 4;                                 // float4 per pixel
  const int slots = PP * q;
  const unsigned long long c0 = wall_clock64();
  if (MODE == 0) {
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(const __attribute__((address_space(3))) float*)s);
    for (int b = 0; b < slots; b += 256) {
      const int e = min(b + tid, slots - 1);
      const int p = e / q, c = (e - p * q) * 4;
      const int m = (m0 + p) % M;
      glds16(act, (unsigned)(((size_t)m * COUT + choff + c) * 4), lds + (unsigned)((b + wid * 64) * 16));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    const size_t stride = (size_t)M * COUT;
    for (int b = 0; b < slots; b += 256) {
      const int e = min(b + tid, slots - 1);
      const int p = e / q, c = (e - p * q) * 4;
      const int m = (m0 + p) % M;
      const float* src = part + (size_t)m * COUT + choff + c;
      float4 v = *reinterpret_cast<const float4*>(src);
#pragma unroll 8
      for (int k = 1; k < SL; ++k) {
        const float4 u = *reinterpret_cast<const float4*>(src + k * stride);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      const float4 bb = *reinterpret_cast<const float4*>(bias + choff + c);
      v.x = fmaxf(v.x + bb.x, 0.f); v.y = fmaxf(v.y + bb.y, 0.f); v.z = fmaxf(v.z + bb.z, 0.f); v.w = fmaxf(v.w + bb.w, 0.f);
      *reinterpret_cast<float4*>(s + (size_t)(b + tid) * 4) = v;
    }
  }
  __syncthreads();
  const unsigned long long c1 = wall_clock64();
  if (tid == 0) out[blockIdx.x] = (float)(c1 - c0);
  if (s[(tid * 7) % (slots * 4)] == 123456.f) out[blockIdx.x] = s[tid];      // keeps the fill alive
}

template <int MODE>
static void run(const Edge& e, const float* act, const float* part, const float* bias, float* out, float& us, float& inner_us) {
  const size_t lds = ((size_t)e.PP * e.CH * 4 + 256 * 16 + 1023) & ~(size_t)1023;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 10; ++i)
    hipLaunchKernelGGL(probe<MODE>, dim3(e.wgs), dim3(256), lds, 0, act, part, bias, out, e.PP, e.CH, e.SL, e.M, e.COUT, e.cons_slices);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < 200; ++i)
    hipLaunchKernelGGL(probe<MODE>, dim3(e.wgs), dim3(256), lds, 0, act, part, bias, out, e.PP, e.CH, e.SL, e.M, e.COUT, e.cons_slices);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  us = ms / 200 * 1000.f;
  std::vector<float> h(e.wgs);
  hipMemcpy(h.data(), out, sizeof(float) * e.wgs, hipMemcpyDeviceToHost);
  double t = 0; for (float v : h) t += v;
  inner_us = (float)(t / e.wgs / 100.0);               // wall_clock64: 100 MHz
}

__global__ void empty_kernel(float* o) { if (threadIdx.x == 1234567) o[0] = 1.f; }

int main() {
  // consumer kernels of conv_slices_small.hip: <CHUNKS, STRIDE, PPMAX, WO, SLICES, NT>; producer partial tensors [SL][M][COUT]
  const Edge edges[] = {
      {"convAB1 (4 slices) -> convAB2.conv1 <1,1,216,22,8,8>", 216, 32, 4, 484, 256, 256, 8},
      {"convAB2.conv1 (8) -> convAB2.conv2 <1,1,216,22,8,8>", 216, 32, 8, 484, 256, 256, 8},
      {"convAB2.conv2 (8) -> trans|rot conv1 <1,2,552,11,8,32>", 552, 32, 8, 484, 256, 256, 8},
      {"trans|rot conv1 (8) -> conv2.conv1 <2,1,169,11,8,16>", 169, 64, 8, 121, 1024, 256, 8},
      {"conv2.conv1 (8) -> conv2.conv2 <2,1,169,11,8,16>", 169, 64, 8, 121, 1024, 256, 8},
      {"conv2.conv2 (8) -> tail (16 workgroups x 64 channels x 121 pixels)", 121, 64, 8, 121, 1024, 16, 16},
      {"conv2.conv2 (8) -> tail as 64 workgroups x 16 channels", 121, 16, 8, 121, 1024, 64, 64},
  };
  float *act, *part, *bias, *out;
  const size_t maxpart = (size_t)8 * 484 * 1024;
  hipMalloc(&act, sizeof(float) * 484 * 1024); hipMalloc(&part, sizeof(float) * maxpart); hipMalloc(&bias, sizeof(float) * 1024);
  hipMalloc(&out, sizeof(float) * 1024);
  hipMemset(act, 0, sizeof(float) * 484 * 1024); hipMemset(part, 0, sizeof(float) * maxpart); hipMemset(bias, 0, sizeof(float) * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, out);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, out);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("empty kernel, 256 workgroups, back to back: %.2f us per launch\n", ms / 200 * 1000.f);
  printf("%-72s %10s %10s %12s %12s %14s\n", "edge (producer -> consumer)", "DMA us", "reduce us", "DMA inner", "reduce inner", "partial MB read");
  for (const Edge& e : edges) {
    float u0, i0, u1, i1;
    run<0>(e, act, part, bias, out, u0, i0);
    run<1>(e, act, part, bias, out, u1, i1);
    printf("%-72s %10.2f %10.2f %12.2f %12.2f %14.1f\n", e.name, u0, u1, i0, i1, (double)e.wgs * e.PP * e.CH * e.SL * 4 / 1e6);
  }
  return 0;
}
