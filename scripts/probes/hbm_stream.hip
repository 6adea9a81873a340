// Probe: what do plain streaming passes get from the MI355X memory system?  The zero-MFMA passes of a batch-64 step (max-pool:
// 275 MB fetched + 62 MB written in 71 us; preprocess; the Winograd transform passes) run at 4.2-4.75 TB/s: this probe measures
// the rates a read-only, a write-only, a copy and a pool-shaped (4 : 1 read : write) pass reach on the same box, at working
// sets below and above the 256 MB Infinity Cache, so that those figures can be read against the machine and not against the
// 8 TB/s data-sheet peak.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/probes/hbm_stream.hip -o scripts/probes/bin/hbm_stream
// Run:   scripts/probes/bin/hbm_stream            (prints one line per pass x size; ~5 s)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every thread moves float4s at a grid stride: the access pattern of the product's streaming kernels
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ in, float4* __restrict__ sink, size_t n) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 v = in[i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (acc.x == 12345.678f) sink[0] = acc;   // never true: keeps the loads
}

__global__ __launch_bounds__(256) void write_kernel(float4* __restrict__ out, size_t n) {
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

// 4 reads : 1 write, the max-pool's ratio (unique bytes): out[i] = max of in[4i .. 4i+3]
__global__ __launch_bounds__(256) void reduce4_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t nout) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nout; i += (size_t)gridDim.x * 256) {
    const size_t blk = i >> 6, l = i & 63;   // a wave reads 4 consecutive 1 KB runs
    const float4* p = in + (blk * 4) * 64 + l;
    const float4 a = p[0], b = p[64], c = p[128], d = p[192];
    float4 m;
    m.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x)); m.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
    m.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z)); m.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
    out[i] = m;
  }
}

template <class F>
static double time_us(F launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int r = 0; r < reps + 2; ++r) {
    CHECK(hipEventRecord(e0, 0));
    launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  const size_t MB = 1 << 20;
  const size_t cap = 2048 * MB;
  float4 *a, *b;
  CHECK(hipMalloc(&a, cap)); CHECK(hipMalloc(&b, cap));
  CHECK(hipMemset(a, 0, cap)); CHECK(hipMemset(b, 0, cap));
  CHECK(hipDeviceSynchronize());
  const size_t sizes[] = {32 * MB, 64 * MB, 128 * MB, 256 * MB, 512 * MB, 1024 * MB, 2048 * MB};
  const int grids[] = {2048, 8192, 32768};
  printf("%-22s %8s %8s %10s %10s\n", "pass", "MB", "grid", "us", "TB/s");
  for (size_t bytes : sizes) {
    const size_t n = bytes / 16;
    for (int g : grids) {
      double us = time_us([&] { hipLaunchKernelGGL(read_kernel, dim3(g), dim3(256), 0, 0, a, b, n); }, 7);
      printf("%-22s %8zu %8d %10.1f %10.2f\n", "read", bytes / MB, g, us, bytes / us * 1e-6);
      us = time_us([&] { hipLaunchKernelGGL(write_kernel, dim3(g), dim3(256), 0, 0, b, n); }, 7);
      printf("%-22s %8zu %8d %10.1f %10.2f\n", "write", bytes / MB, g, us, bytes / us * 1e-6);
      us = time_us([&] { hipLaunchKernelGGL(copy_kernel, dim3(g), dim3(256), 0, 0, a, b, n / 2); }, 7);
      printf("%-22s %8zu %8d %10.1f %10.2f\n", "copy (r + w bytes)", bytes / MB, g, us, bytes / us * 1e-6);
      us = time_us([&] { hipLaunchKernelGGL(reduce4_kernel, dim3(g), dim3(256), 0, 0, a, b, n / 5); }, 7);
      printf("%-22s %8zu %8d %10.1f %10.2f\n", "4 reads : 1 write", bytes / MB, g, us, bytes / us * 1e-6);
    }
  }
  // producer -> consumer through the Infinity Cache, the stem -> pool situation: a pass writes X MB, the next pass reads them
  for (size_t bytes : {64 * MB, 128 * MB, 256 * MB, 512 * MB}) {
    const size_t n = bytes / 16;
    double us = time_us([&] {
      hipLaunchKernelGGL(write_kernel, dim3(8192), dim3(256), 0, 0, a, n);
      hipLaunchKernelGGL(reduce4_kernel, dim3(8192), dim3(256), 0, 0, a, b, n / 4);
    }, 7);
    double usw = time_us([&] { hipLaunchKernelGGL(write_kernel, dim3(8192), dim3(256), 0, 0, a, n); }, 7);
    printf("%-22s %8zu %8d %10.1f %10.2f   (read-back pass alone: %.1f us, %.2f TB/s on r + w bytes)\n", "write then 4r:1w", bytes / MB, 8192,
           us, (bytes * 2.25) / us * 1e-6, us - usw, (bytes * 1.25) / (us - usw) * 1e-6);
  }
  return 0;
}
