// Probe for DESIGN.md section 7 item 13: a reader that sums 169 rows of a CONSTANT buffer (nobody writes it) with
// three wait disciplines, launched on its own stream while another stream runs the library's kernels.
//   pool_counted : `for p: s += row[p]`          -> 13 loads in flight, counted s_waitcnt vmcnt(12..0), each followed
//                                                    at once by the add that consumes the row (the old tail_kernel loop)
//   pool_drained : 13 loads, s_waitcnt vmcnt(0), 13 adds (the shipped tail_kernel loop)
//   pool_b64     : as pool_counted with 8-byte loads (two per row)
// Build: hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -o variants/libvmcnt_probe.so scripts/probes/vmcnt_race.hip
// Run:   python scripts/probes/vmcnt_race.py      (on the GPU box)
#include <hip/hip_runtime.h>

constexpr int PP = 169;

__global__ __launch_bounds__(256) void pool_counted(const float* __restrict__ x, float4* __restrict__ out) {
  const int n = blockIdx.x, t = threadIdx.x;
  const float* src = x + (size_t)n * PP * 1024 + t * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 13
  for (int p = 0; p < PP; ++p) {
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)p * 1024);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  out[n * 256 + t] = s;
}

__global__ __launch_bounds__(256) void pool_b64(const float* __restrict__ x, float4* __restrict__ out) {
  const int n = blockIdx.x, t = threadIdx.x;
  const float* src = x + (size_t)n * PP * 1024 + t * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 13
  for (int p = 0; p < PP; ++p) {
    const float2 a = *reinterpret_cast<const float2*>(src + (size_t)p * 1024);
    const float2 b = *reinterpret_cast<const float2*>(src + (size_t)p * 1024 + 2);
    s.x += a.x; s.y += a.y; s.z += b.x; s.w += b.y;
  }
  out[n * 256 + t] = s;
}

__global__ __launch_bounds__(256) void pool_drained(const float* __restrict__ x, float4* __restrict__ out) {
  const int n = blockIdx.x, t = threadIdx.x;
  const float* src = x + (size_t)n * PP * 1024 + t * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
  for (int p0 = 0; p0 < PP; p0 += 13) {
    float4 v[13];
#pragma unroll
    for (int q = 0; q < 13; ++q) v[q] = *reinterpret_cast<const float4*>(src + (size_t)(p0 + q) * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 13; ++q) asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q].z), "+v"(v[q].w));
#pragma unroll
    for (int q = 0; q < 13; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
  }
  out[n * 256 + t] = s;
}

// Writer with the store pattern of the conv epilogue (store_tiles): workgroup = 256 rows x 128 channels, 8 waves as
// 4 x 2, wave = 64 rows x 64 channels, lane (l31, hh) stores float4 at row l31 (+32), channels j*32 + q*8 + hh*4.
// Copies `master` to `x`: the reader that follows in the stream must see exactly `master`.
__global__ __launch_bounds__(512) void writer_tiles(const float* __restrict__ master, float* __restrict__ x, int rows) {
  const int panels = 8;                              // 1024 channels / 128
  const int p = blockIdx.x % panels, mt = blockIdx.x / panels;
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hh = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = mt * 256 + (wm * 2 + i) * 32 + l31;
    if (row >= rows) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t off = (size_t)row * 1024 + p * 128 + wn * 64 + j * 32 + q * 8 + hh * 4;
        *reinterpret_cast<float4*>(x + off) = *reinterpret_cast<const float4*>(master + off);
      }
  }
}

extern "C" int probe_write(const float* master, float* x, int rows, void* stream) {
  const int tiles = (rows + 255) / 256;
  hipLaunchKernelGGL(writer_tiles, dim3(tiles * 8), dim3(512), 0, (hipStream_t)stream, master, x, rows);
  return (int)hipGetLastError();
}

extern "C" int probe_launch(int which, const float* x, float* out, int n, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  float4* o = reinterpret_cast<float4*>(out);
  if (which == 0) hipLaunchKernelGGL(pool_counted, dim3(n), dim3(256), 0, st, x, o);
  else if (which == 1) hipLaunchKernelGGL(pool_drained, dim3(n), dim3(256), 0, st, x, o);
  else hipLaunchKernelGGL(pool_b64, dim3(n), dim3(256), 0, st, x, o);
  return (int)hipGetLastError();
}
