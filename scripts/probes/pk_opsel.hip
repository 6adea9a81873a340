// Stand-alone characterisation of profiles/EXPERIMENTS.md items 13: packed-float32 VALU instructions with an `op_sel` swizzle
// give wrong results in a wave while waves of ANOTHER kernel on the same CU issue f16 matrix instructions.
//   victims    : valu<FORM>: 4096 x `v_pk_add_f32 a, a, v <FORM>` on lane-dependent small integers (exact in float32)
//   co-runners : spin<KIND>: register-only loops of one instruction class
// Build: hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -o variants/libpk_opsel.so scripts/probes/pk_opsel.hip
// Run:   python scripts/probes/pk_opsel.py   (on the GPU box)
#include <hip/hip_runtime.h>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// FORM: 0 plain | 1 op_sel:[0,1] op_sel_hi:[1,0] (src1 halves swapped) | 2 op_sel_hi:[1,0] (src1.lo broadcast)
//       3 op_sel:[1,0] op_sel_hi:[0,1] (src0 halves swapped) | 4 op_sel:[0,1] (src1.hi broadcast)
//       5 v_pk_fma_f32 with src1 halves swapped | 6 v_pk_fma_f32 plain
template <int FORM>
__global__ __launch_bounds__(256) void valu(float2* __restrict__ out, int iters) {
  const int t = threadIdx.x;
  f2 a = {0.f, 0.f};
  f2 v = {(float)(t + 1), (float)(2 * t + 1)};
  f2 one = {1.f, 1.f};
  for (int i = 0; i < iters; ++i) {
    if (FORM == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(v));
    if (FORM == 1) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a) : "v"(v));
    if (FORM == 2) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a) : "v"(v));
    if (FORM == 3) {   // a = swap(a) + v  (two of them restore the order)
      asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(a) : "v"(v));
    }
    if (FORM == 4) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(a) : "v"(v));
    if (FORM == 5) asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(a) : "v"(v), "v"(one));   // a += 1 * swap(v)
    if (FORM == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,1]" : "+v"(a) : "v"(v), "v"(one));   // plain fma
  }
  out[blockIdx.x * 256 + t] = make_float2(a[0], a[1]);
}

// KIND: 0 v_mfma_f32_32x32x16_f16 | 1 v_mfma_f32_32x32x2_f32 | 2 v_pk_fma_f16 (VALU) | 3 v_mfma_f32_16x16x32_f16 | 4 v_mfma_f32_32x32x16_bf16
//       5 v_mfma_f32_32x32x8_f16 (the pre-gfx950 f16 instruction)
template <int KIND>
__global__ __launch_bounds__(256) void spin(float* __restrict__ sink, int iters) {
  f32x16 acc = {0};
  float acc4[4] = {0, 0, 0, 0};
  half8 ha, hb;
  for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)(0.001f * (threadIdx.x + e)); hb[e] = (_Float16)(0.002f * e); }
  half2v p = {(_Float16)1.0f, (_Float16)0.5f}, q = {(_Float16)0.25f, (_Float16)0.125f}, r = {(_Float16)0.f, (_Float16)0.f};
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
    if (KIND == 1) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(0.001f * i, 0.5f, acc, 0, 0, 0);
    if (KIND == 2) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(r) : "v"(p), "v"(q));
    if (KIND == 3) {
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      f32x4 c = {acc4[0], acc4[1], acc4[2], acc4[3]};
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c, 0, 0, 0);
      acc4[0] = c[0]; acc4[1] = c[1]; acc4[2] = c[2]; acc4[3] = c[3];
    }
    if (KIND == 5) {
      typedef _Float16 half4 __attribute__((ext_vector_type(4)));
      half4 a4 = {ha[0], ha[1], ha[2], ha[3]}, b4 = {hb[0], hb[1], hb[2], hb[3]};
      acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc, 0, 0, 0);
    }
    if (KIND == 4) {
      typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, ha), __builtin_bit_cast(bf8, hb), acc, 0, 0, 0);
    }
  }
  float s = acc4[0] + acc4[1] + acc4[2] + acc4[3] + (float)r[0] + (float)r[1];
  for (int e = 0; e < 16; ++e) s += acc[e];
  if (s == 12345.678f) sink[0] = s;     // keeps the loop alive
}

extern "C" int victim(int form, float* out, int blocks, int iters, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  float2* o = reinterpret_cast<float2*>(out);
  switch (form) {
    case 0: hipLaunchKernelGGL(valu<0>, dim3(blocks), dim3(256), 0, st, o, iters); break;
    case 1: hipLaunchKernelGGL(valu<1>, dim3(blocks), dim3(256), 0, st, o, iters); break;
    case 2: hipLaunchKernelGGL(valu<2>, dim3(blocks), dim3(256), 0, st, o, iters); break;
    case 3: hipLaunchKernelGGL(valu<3>, dim3(blocks), dim3(256), 0, st, o, iters); break;
    case 4: hipLaunchKernelGGL(valu<4>, dim3(blocks), dim3(256), 0, st, o, iters); break;
    case 5: hipLaunchKernelGGL(valu<5>, dim3(blocks), dim3(256), 0, st, o, iters); break;
    case 6: hipLaunchKernelGGL(valu<6>, dim3(blocks), dim3(256), 0, st, o, iters); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
extern "C" int corunner(int kind, float* sink, int blocks, int iters, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (kind) {
    case 0: hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
    case 1: hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
    case 2: hipLaunchKernelGGL(spin<2>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
    case 3: hipLaunchKernelGGL(spin<3>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
    case 4: hipLaunchKernelGGL(spin<4>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
    default: hipLaunchKernelGGL(spin<5>, dim3(blocks), dim3(256), 0, st, sink, iters); break;
  }
  return (int)hipGetLastError();
}
