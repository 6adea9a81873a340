// Hardware probe: where does `global_load_lds_dwordx4 v, s[base] offset:N` put its data?
// (does the instruction offset apply to the global address, the LDS address, or both)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* g, float* o) {
  __shared__ __attribute__((aligned(16))) float s[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) s[i] = -1.f;
  __syncthreads();
  unsigned lds = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) float*)s;
  unsigned voff = threadIdx.x * 16;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:2048\n\ts_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(voff), "s"(lds), "s"(g) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) o[i] = s[i];
}
int main() {
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (float)i;
  float *g, *o;
  hipMalloc(&g, 4096 * 4); hipMalloc(&o, 2048 * 4);
  hipMemcpy(g, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o);
  std::vector<float> r(2048);
  hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
  int first = -1;
  for (int i = 0; i < 2048; ++i) if (r[i] >= 0 && first < 0) first = i;
  printf("first written LDS float index = %d (value %.0f); lds[0]=%.0f lds[512]=%.0f lds[767]=%.0f\n", first,
         first >= 0 ? r[first] : -1.f, r[0], r[512], r[767]);
  printf("=> LDS offset applied: %s ; global offset applied: %s\n", first == 512 ? "yes" : "no",
         (first >= 0 && r[first] == 512.f) ? "yes" : "no");
  return 0;
}
