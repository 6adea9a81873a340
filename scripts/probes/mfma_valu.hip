// Probe: does vector-ALU work placed between the MFMAs of a wave run in the shadow of the matrix pipe?
// 256 workgroups x 8 waves (2 per SIMD, as wino64_fused_kernel), every wave: ITER x { 16 x [ v_mfma_f32_32x32x2_f32 on ONE accumulator
// (a dependent chain, as a 32 x 32 block per wave has), then NV independent v_fma_f32 ] }.
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/mfma_valu.hip -o scripts/probes/bin/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NV, int NACC, bool MFMA, int KIND = 0>   // KIND 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_pk_add_f32, 3: v_add_f32
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int c = 0; c < NACC; ++c)
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  float y[8];
  for (int i = 0; i < 8; ++i) y[i] = i;
  f32x2 y2[8], a2 = {a, a}, b2 = {b, b};
  for (int i = 0; i < 8; ++i) y2[i] = f32x2{(float)i, (float)i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (MFMA) acc[q % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q % NACC], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(y[v & 7]) : "v"(a), "v"(b));
        else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y2[v & 7]) : "v"(a2), "v"(b2));
        else if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y2[v & 7]) : "v"(a2));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(y[v & 7]) : "v"(a));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int c = 0; c < NACC; ++c)
    for (int e = 0; e < 16; ++e) s += acc[c][e];
  for (int i = 0; i < 8; ++i) s += y[i] + y2[i].x + y2[i].y;
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

// Specialised waves: waves 0-3 of a workgroup (one per SIMD) issue only MFMAs, waves 4-7 (the other wave of each SIMD) only VALU work
// (NV v_fma_f32 per MFMA slot of the partner wave): does the vector work of ONE wave hide behind the matrix work of ANOTHER?
template <int NV, int WHO>   // WHO: 1 = MFMA waves only (others exit), 2 = VALU waves only, 3 = both
__global__ __launch_bounds__(512) void ksplit(float* out, int iters) {
  const int wid = threadIdx.x >> 6;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  float y[8];
  for (int i = 0; i < 8; ++i) y[i] = i;
  if (wid < 4) {
    if (WHO & 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  } else if (WHO & 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(y[v & 7]) : "v"(a), "v"(b));
      }
    }
  }
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc[e];
  for (int i = 0; i < 8; ++i) s += y[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NV, int WHO>
void run_split(const char* name, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((ksplit<NV, WHO>), dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-60s %8.3f ms   %6.1f clk per slot (one MFMA of the matrix wave + NV VALU of the vector wave)\n", name, best,
         best * 1e-3 * 2.4e9 / (iters * 16.0));
}

template <int NV, int NACC, bool MFMA, int KIND = 0>
void run(const char* name, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NACC, MFMA, KIND>), dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double mf = MFMA ? 256.0 * 8 * iters * 16 * 4096.0 : 0.0;
  const double clk_per_group = best * 1e-3 * 2.4e9 / (iters * 16.0);   // per (MFMA + NV VALU) slot of one wave pair, at 2.4 GHz
  printf("%-44s %8.3f ms  %7.1f TF   %6.1f clk per slot (2 waves/SIMD, 2.4 GHz nominal)\n", name, best, mf / best * 1e-9, clk_per_group);
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<0, 1, true>("MFMA chain only", out);
  run<2, 1, true>("MFMA chain + 2 VALU each", out);
  run<4, 1, true>("MFMA chain + 4 VALU each", out);
  run<8, 1, true>("MFMA chain + 8 VALU each", out);
  run<12, 1, true>("MFMA chain + 12 VALU each", out);
  run<16, 1, true>("MFMA chain + 16 VALU each", out);
  run<8, 2, true>("2 MFMA chains + 8 VALU each", out);
  run<8, 1, true, 1>("MFMA chain + 8 v_pk_fma_f32 each", out);
  run<8, 1, true, 2>("MFMA chain + 8 v_pk_add_f32 each", out);
  run<8, 1, true, 3>("MFMA chain + 8 v_add_f32 each", out);
  run<4, 1, true, 1>("MFMA chain + 4 v_pk_fma_f32 each", out);
  run<8, 1, false, 1>("8 v_pk_fma_f32 only", out);
  run<8, 1, false, 2>("8 v_pk_add_f32 only", out);
  run<8, 1, false, 3>("8 v_add_f32 only", out);
  run<4, 1, false>("4 VALU only", out);
  run<8, 1, false>("8 VALU only", out);
  run<16, 1, false>("16 VALU only", out);
  printf("\nspecialised waves (per SIMD: one wave issues only MFMAs, the other only v_fma_f32)\n");
  run_split<16, 1>("matrix waves alone (1 wave per SIMD, MFMA chain)", out);
  run_split<16, 2>("vector waves alone, 16 VALU per slot", out);
  run_split<16, 3>("both: matrix waves + vector waves with 16 VALU per slot", out);
  run_split<8, 2>("vector waves alone, 8 VALU per slot", out);
  run_split<8, 3>("both: matrix waves + vector waves with 8 VALU per slot", out);
  run_split<24, 2>("vector waves alone, 24 VALU per slot", out);
  run_split<24, 3>("both: matrix waves + vector waves with 24 VALU per slot", out);
  return 0;
}
