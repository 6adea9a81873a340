// Hardware probe: the effective core clock a SHORT kernel sees, by launch regime.  One wave per SIMD issues 2048 dependent-free
// v_mfma_f32_32x32x2_f32 (64 cycles each = 131,072 core cycles); wall_clock64() (constant 100 MHz) brackets them inside the kernel.
//   regime A: launches back to back        regime B: host synchronises after every launch
//   regime C: B + the host sleeps 200 us   regime D: B + sleeps 2 ms      (a tracker's frame loop is B..C)
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* o, int reps) {
  f32x16 a0 = {0}, a1 = {0};
  const float x = (float)threadIdx.x;
  const unsigned long long c0 = wall_clock64();
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 1.f, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 2.f, a1, 0, 0, 0);
    }
  }
  const unsigned long long c1 = wall_clock64();
  if (threadIdx.x == 0) o[blockIdx.x] = (float)(c1 - c0);
  if (a0[0] + a1[3] == 12345.f) o[0] = a0[5];
}
static double regime(float* o, int reps, int mode, int launches) {
  double acc = 0; int cnt = 0;
  for (int i = 0; i < launches; ++i) {
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, o, reps);
    if (mode >= 1) hipDeviceSynchronize();
    if (mode == 2) usleep(200);
    if (mode == 3) usleep(2000);
    if (mode >= 1 && i >= launches / 2) { float h[256]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost); for (float v : h) { acc += v; ++cnt; } }
  }
  hipDeviceSynchronize();
  if (mode == 0) { float h[256]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost); for (float v : h) { acc += v; ++cnt; } }
  return acc / cnt;   // mean 100 MHz ticks
}
int main() {
  float* o; hipMalloc(&o, 1024);
  const char* nm[] = {"A back to back", "B sync per launch", "C sync + 200 us sleep", "D sync + 2 ms sleep"};
  for (int reps : {16, 128}) {      // 256 / 2048 MFMAs per wave: ~7 us / ~55 us at 2.4 GHz
    const double cycles = (double)reps * 16 * 64;
    for (int m = 0; m < 4; ++m) {
      const double ticks = regime(o, reps, m, m == 3 ? 200 : 2000);
      printf("%4d MFMAs/wave, %-22s: %.2f us in-kernel -> effective core clock %.0f MHz\n", reps * 16, nm[m], ticks / 100.0, cycles / (ticks / 100.0));
    }
  }
  return 0;
}
