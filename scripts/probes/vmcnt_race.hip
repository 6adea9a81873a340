// Probe for profiles/EXPERIMENTS.md items 13: a reader that sums 169 rows of a CONSTANT buffer (nobody writes it) with
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

// The library's tail_kernel as first written (counted loop, wave-shuffle + LDS reduction, FC, tanh), pose update left out.
__device__ __forceinline__ float wave_sum_(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int DRAIN, int DUMP>
__global__ __launch_bounds__(256) void tail_copy(const float* __restrict__ head, const float* __restrict__ fc_w,
                                                 const float* __restrict__ fc_b, float* __restrict__ logits,
                                                 float4* __restrict__ pooled, float* __restrict__ wavepart) {
  __shared__ float part[4][3];
  const int n = blockIdx.x, t = threadIdx.x;
  const float* src = head + (size_t)n * PP * 1024 + t * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (DRAIN) {
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
  } else {
#pragma unroll 13
    for (int p = 0; p < PP; ++p) {
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)p * 1024);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  if (DUMP) pooled[n * 256 + t] = s;
  const float inv = 121.f;
  s.x /= inv; s.y /= inv; s.z /= inv; s.w /= inv;
  const int hd = t >> 7, cl = (t & 127) * 4;
  float acc[3];
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    const float4 w = *reinterpret_cast<const float4*>(fc_w + (hd * 3 + o) * 512 + cl);
    acc[o] = wave_sum_(s.x * w.x + s.y * w.y + s.z * w.z + s.w * w.w);
  }
  if ((t & 63) == 0) {
    part[t >> 6][0] = acc[0]; part[t >> 6][1] = acc[1]; part[t >> 6][2] = acc[2];
    if (DUMP) { wavepart[(n * 4 + (t >> 6)) * 3 + 0] = acc[0]; wavepart[(n * 4 + (t >> 6)) * 3 + 1] = acc[1]; wavepart[(n * 4 + (t >> 6)) * 3 + 2] = acc[2]; }
  }
  __syncthreads();
  if (t < 6) {
    const int h = t / 3, o = t - h * 3;
    logits[n * 6 + t] = part[2 * h][o] + part[2 * h + 1][o] + fc_b[h * 4 + o];
  }
}

extern "C" int probe_tail(int drain, const float* head, const float* fc_w, const float* fc_b, float* logits, float* pooled,
                          float* wavepart, int n, void* stream) {
  float4* pl = reinterpret_cast<float4*>(pooled);
  hipStream_t st = (hipStream_t)stream;
  if (drain == 1) hipLaunchKernelGGL((tail_copy<1, 0>), dim3(n), dim3(256), 0, st, head, fc_w, fc_b, logits, pl, wavepart);
  else if (drain == 0) hipLaunchKernelGGL((tail_copy<0, 0>), dim3(n), dim3(256), 0, st, head, fc_w, fc_b, logits, pl, wavepart);
  else hipLaunchKernelGGL((tail_copy<0, 1>), dim3(n), dim3(256), 0, st, head, fc_w, fc_b, logits, pl, wavepart);
  return (int)hipGetLastError();
}

// No memory at all in the loop: 4096 swizzled packed adds of lane-dependent constants; the exact result is known.
template <int SWZ>
__global__ __launch_bounds__(256) void valu_only(float4* __restrict__ out) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int t = threadIdx.x;
  f2 a = {0.f, 0.f};
  f2 v = {(float)(t + 1), (float)(2 * t + 1)};       // small integers: every partial sum is exact in float32
  for (int i = 0; i < 4096; ++i) {
    if (SWZ) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a) : "v"(v));
    else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(v));
  }
  out[blockIdx.x * 256 + t] = SWZ ? make_float4(a[1], a[0], 0.f, 0.f) : make_float4(a[0], a[1], 0.f, 0.f);
}

// pool_counted + a token use of LDS (72 bytes, one barrier): does merely owning an LDS allocation matter?
__global__ __launch_bounds__(256) void pool_counted_lds(const float* __restrict__ x, float4* __restrict__ out) {
  __shared__ float token[18];
  const int n = blockIdx.x, t = threadIdx.x;
  const float* src = x + (size_t)n * PP * 1024 + t * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 13
  for (int p = 0; p < PP; ++p) {
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)p * 1024);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (t < 18) token[t] = 0.f;
  __syncthreads();
  s.x += token[t % 18];
  out[n * 256 + t] = s;
}

// pool_counted + the arithmetic that follows the loop in tail_kernel (division, FC dot products, wave shuffles), no LDS:
// the pooled sums are still written out, the FC results go to a side buffer
__global__ __launch_bounds__(256) void pool_counted_fc(const float* __restrict__ x, float4* __restrict__ out,
                                                       const float* __restrict__ fc_w, float* __restrict__ side) {
  const int n = blockIdx.x, t = threadIdx.x;
  const float* src = x + (size_t)n * PP * 1024 + t * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 13
  for (int p = 0; p < PP; ++p) {
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)p * 1024);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  out[n * 256 + t] = s;
  const float inv = 121.f;
  s.x /= inv; s.y /= inv; s.z /= inv; s.w /= inv;
  const int hd = t >> 7, cl = (t & 127) * 4;
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    const float4 w = *reinterpret_cast<const float4*>(fc_w + (hd * 3 + o) * 512 + cl);
    const float a = wave_sum_(s.x * w.x + s.y * w.y + s.z * w.z + s.w * w.w);
    if ((t & 63) == 0) side[(n * 4 + (t >> 6)) * 3 + o] = a;
  }
}

// The counted-wait loop written in assembly so that the ONLY difference between the two instantiations is the form of the
// first packed add: 6 rows (12 eight-byte loads) in flight, `s_waitcnt vmcnt(N)` then at once the two adds of the row.
// SWZ = 1: the (x, y) accumulator lives swapped in its register pair and is added with op_sel:[0,1] op_sel_hi:[1,0]
// (what hipcc chose in the old tail_kernel); SWZ = 0: plain v_pk_add_f32 for both pairs.
template <int SWZ>
__global__ __launch_bounds__(256) void pool_counted_pk(const float* __restrict__ x, float4* __restrict__ out) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int n = blockIdx.x, t = threadIdx.x;
  const float* src = x + (size_t)n * PP * 1024 + t * 4;
  f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#define ROW(K, CNT)                                                                                   \
  "s_waitcnt vmcnt(" #CNT ")\n\t"                                                                     \
  "v_pk_add_f32 %[a01], %[a01], %[x" #K "]" SWZSTR "\n\t"                                              \
  "v_pk_add_f32 %[a23], %[a23], %[z" #K "]\n\t"
#define LOADS                                                                                         \
  "global_load_dwordx2 %[x0], %[p0], off\n\tglobal_load_dwordx2 %[z0], %[p0], off offset:8\n\t"       \
  "global_load_dwordx2 %[x1], %[p1], off\n\tglobal_load_dwordx2 %[z1], %[p1], off offset:8\n\t"       \
  "global_load_dwordx2 %[x2], %[p2], off\n\tglobal_load_dwordx2 %[z2], %[p2], off offset:8\n\t"       \
  "global_load_dwordx2 %[x3], %[p3], off\n\tglobal_load_dwordx2 %[z3], %[p3], off offset:8\n\t"       \
  "global_load_dwordx2 %[x4], %[p4], off\n\tglobal_load_dwordx2 %[z4], %[p4], off offset:8\n\t"       \
  "global_load_dwordx2 %[x5], %[p5], off\n\tglobal_load_dwordx2 %[z5], %[p5], off offset:8\n\t"
#define OPERANDS                                                                                      \
  : [a01] "+v"(a01), [a23] "+v"(a23), [x0] "=&v"(x0), [z0] "=&v"(z0), [x1] "=&v"(x1), [z1] "=&v"(z1), [x2] "=&v"(x2),      \
    [z2] "=&v"(z2), [x3] "=&v"(x3), [z3] "=&v"(z3), [x4] "=&v"(x4), [z4] "=&v"(z4), [x5] "=&v"(x5), [z5] "=&v"(z5)          \
  : [p0] "v"(p + 0 * 1024), [p1] "v"(p + 1 * 1024), [p2] "v"(p + 2 * 1024), [p3] "v"(p + 3 * 1024), [p4] "v"(p + 4 * 1024), \
    [p5] "v"(p + 5 * 1024)                                                                            \
  : "memory"
  for (int p0 = 0; p0 + 6 <= PP; p0 += 6) {       // 28 batches = rows 0..167
    const float* p = src + (size_t)p0 * 1024;
    f2 x0, z0, x1, z1, x2, z2, x3, z3, x4, z4, x5, z5;
    if (SWZ == 2) {       // swizzled adds after a FULL drain
#define SWZSTR " op_sel:[0,1] op_sel_hi:[1,0]"
      asm volatile(LOADS ROW(0, 0) ROW(1, 0) ROW(2, 0) ROW(3, 0) ROW(4, 0) ROW(5, 0) OPERANDS);
#undef SWZSTR
    } else if (SWZ) {
#define SWZSTR " op_sel:[0,1] op_sel_hi:[1,0]"
      asm volatile(LOADS ROW(0, 10) ROW(1, 8) ROW(2, 6) ROW(3, 4) ROW(4, 2) ROW(5, 0) OPERANDS);
#undef SWZSTR
    } else {
#define SWZSTR ""
      asm volatile(LOADS ROW(0, 10) ROW(1, 8) ROW(2, 6) ROW(3, 4) ROW(4, 2) ROW(5, 0) OPERANDS);
#undef SWZSTR
    }
  }
#undef ROW
#undef LOADS
#undef OPERANDS
  {                                                // row 168
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)168 * 1024);
    if (SWZ) { a01[0] += v.y; a01[1] += v.x; } else { a01[0] += v.x; a01[1] += v.y; }
    a23[0] += v.z; a23[1] += v.w;
  }
  out[n * 256 + t] = SWZ ? make_float4(a01[1], a01[0], a23[0], a23[1]) : make_float4(a01[0], a01[1], a23[0], a23[1]);
}

extern "C" int probe_launch2(int which, const float* x, float* out, const float* fc_w, float* side, int n, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  float4* o = reinterpret_cast<float4*>(out);
  if (which == 3) hipLaunchKernelGGL(pool_counted_lds, dim3(n), dim3(256), 0, st, x, o);
  else if (which == 5) hipLaunchKernelGGL(pool_counted_pk<0>, dim3(n), dim3(256), 0, st, x, o);
  else if (which == 6) hipLaunchKernelGGL(pool_counted_pk<1>, dim3(n), dim3(256), 0, st, x, o);
  else if (which == 7) hipLaunchKernelGGL(pool_counted_pk<2>, dim3(n), dim3(256), 0, st, x, o);
  else if (which == 8) hipLaunchKernelGGL(valu_only<0>, dim3(n), dim3(256), 0, st, o);
  else if (which == 9) hipLaunchKernelGGL(valu_only<1>, dim3(n), dim3(256), 0, st, o);
  else hipLaunchKernelGGL(pool_counted_fc, dim3(n), dim3(256), 0, st, x, o, fc_w, side);
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
