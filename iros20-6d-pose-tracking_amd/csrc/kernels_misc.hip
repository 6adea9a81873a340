// Small HBM-bound kernels either side of the convolution stack:
//   preprocess   crop_bbox + OffsetDepth + NormalizeChannels + ToTensor  (Utils.py:320-359,
//                data_augmentation.py:124-189)  -> NHWC float4 pixels
//   to_padded_input  adapter for the reference operator boundary (NCHW, predict.py:267-271) and for
//                plain NHWC inputs: copies into the zero-bordered [n,182,182,4] layout the stem reads
//   tail         AdaptiveAvgPool2d(1) + Linear(512,3) + Tanh for both heads
//                (se3_tracknet.py:72-73,77-78,100-109) + TrackDataset.processPredict
//                (datasets.py:159-175): t_B = trans*tn + t_A, R_B = Rodrigues(rot*rn) . R_A
//   padded_nhwc_to_nchw  output['feature'] in the reference's layout
#include "se3tn_internal.h"
#include "pose_device.h"

namespace se3tn {

// f16x3 mode: a network-input pixel (R,G,B,D) is stored as 4 x f16 hi | 4 x f16 lo in the same 16 bytes
__device__ __forceinline__ float4 split_pixel(float4 v, bool& bad) {
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  const float f[4] = {v.x, v.y, v.z, v.w};
  half8 h;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = (_Float16)f[e];
    h[4 + e] = (_Float16)(f[e] - (float)h[e]);
    bad |= !(fabsf(f[e]) <= 65000.f);
  }
  return __builtin_bit_cast(float4, h);
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void to_padded_input_kernel(const float* __restrict__ in,
                                                               float* __restrict__ out, int total, int nchw,
                                                               int split, int* overflow) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // (n, y, x)
  if (idx >= total) return;
  constexpr int HW = RES * RES;
  const int n = idx / HW, p = idx - n * HW;
  const int y = p / RES, x = p - y * RES;
  float4 v;
  if (nchw) {
    const float* src = in + (size_t)n * 4 * HW + p;
    v.x = src[0]; v.y = src[HW]; v.z = src[2 * HW]; v.w = src[3 * HW];
  } else {
    v = *reinterpret_cast<const float4*>(in + (size_t)idx * 4);
  }
  if (split) {
    bool bad = false;
    v = split_pixel(v, bad);
    if (bad) atomicOr(overflow, 1);
  }
  *reinterpret_cast<float4*>(out + (((size_t)n * IN_P + y + IN_PAD) * IN_P + x + IN_PAD) * 4) = v;
}

hipError_t launch_to_padded_input(const float* in, float* out, int n, int nchw, int split, int* overflow,
                                  hipStream_t st) {
  const int total = n * RES * RES;
  hipLaunchKernelGGL(to_padded_input_kernel, dim3((total + 255) / 256), dim3(256), 0, st, in, out, total, nchw,
                     split, overflow);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// One thread per output pixel.  Index rule of cv2.resize(INTER_NEAREST) (OpenCV resizeNN):
//   sx = min(floor(x * (1.0 / ((double)dst / src))), src - 1)      evaluated in float64,
// the crop canvas is zero outside the frame (Utils.py:327-342).  The arithmetic mirrors what NumPy does in the
// reference: the depth offset (data_augmentation.py:137-140) as ONE float32 operation with the scalar cast to float32 first
// (SE3TN_OFFSET_RULE_NUMPY1: value-based casting, every NumPy the reference runs on) or in float64 rounded once (NUMPY2, NEP 50);
// (x - mean)/std in f64 then stored as f32 (:160-164, :182-187: array operands, float64 under both).
__global__ __launch_bounds__(256) void preprocess_kernel(const CropArgs a) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= RES * RES) return;
  const int i = blockIdx.y;
  const se3tn_crop& c = a.c[i];
  const int y = p / RES, x = p - y * RES;
  const int cw = c.right - c.left, chh = c.bottom - c.top;
  const double ifx = 1.0 / ((double)RES / (double)cw);
  const double ify = 1.0 / ((double)RES / (double)chh);
  int sx = (int)floor((double)x * ifx); sx = sx < cw - 1 ? sx : cw - 1;
  int sy = (int)floor((double)y * ify); sy = sy < chh - 1 ? sy : chh - 1;
  const int fx = c.left + sx, fy = c.top + sy;
  float r = 0.f, g = 0.f, b = 0.f, d = 0.f;
  if ((unsigned)fx < (unsigned)c.W && (unsigned)fy < (unsigned)c.H) {
    const size_t q = (size_t)fy * c.W + fx;
    const uint8_t* px = c.rgb + q * 3;
    r = (float)px[0]; g = (float)px[1]; b = (float)px[2];
    d = (float)c.depth[q];
  }
  const bool invalid = (d <= 100.f) || (d >= 2000.f);
  const double z = c.z_offset_mm;
  if (a.offset_rule == SE3TN_OFFSET_RULE_NUMPY1) {
    const float zf = (float)z;   // np.float64 scalar -> float32 (round to nearest even), then a float32 add / subtract
    d = (z < 0.0) ? d + zf : d - zf;
  } else {
    d = (z < 0.0) ? (float)((double)d + z) : (float)((double)d - z);
  }
  if (invalid) d = 2000.f;
  const double* mean = a.mean + 4 * c.stats;
  const double* sd = a.stdv + 4 * c.stats;
  float4 o;
  o.x = (float)(((double)r - mean[0]) / sd[0]);
  o.y = (float)(((double)g - mean[1]) / sd[1]);
  o.z = (float)(((double)b - mean[2]) / sd[2]);
  o.w = (float)(((double)d - mean[3]) / sd[3]);
  float* base = i < a.n_first ? a.out : a.out2;
  const int io = i < a.n_first ? i : i - a.n_first;
  float* dst = a.padded ? base + (((size_t)io * IN_P + y + IN_PAD) * IN_P + x + IN_PAD) * 4
                        : base + ((size_t)io * RES * RES + p) * 4;
  if (a.split) {
    bool bad = false;
    o = split_pixel(o, bad);
    if (bad) atomicOr(a.overflow, 1);
  }
  *reinterpret_cast<float4*>(dst) = o;
}

hipError_t launch_preprocess(const CropArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(preprocess_kernel, dim3((RES * RES + 255) / 256, a.n), dim3(256), 0, st, a);
  return hipGetLastError();
}

// crop_bbox alone (Utils.py:320-359): same window / index rule as preprocess_kernel, raw integer outputs
__global__ __launch_bounds__(256) void crop_raw_kernel(const se3tn_crop c, uint8_t* __restrict__ rgb_out,
                                                        uint16_t* __restrict__ depth_out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= RES * RES) return;
  const int y = p / RES, x = p - y * RES;
  const int cw = c.right - c.left, chh = c.bottom - c.top;
  const double ifx = 1.0 / ((double)RES / (double)cw);
  const double ify = 1.0 / ((double)RES / (double)chh);
  int sx = (int)floor((double)x * ifx); sx = sx < cw - 1 ? sx : cw - 1;
  int sy = (int)floor((double)y * ify); sy = sy < chh - 1 ? sy : chh - 1;
  const int fx = c.left + sx, fy = c.top + sy;
  uint8_t r = 0, g = 0, b = 0;
  uint16_t d = 0;
  if ((unsigned)fx < (unsigned)c.W && (unsigned)fy < (unsigned)c.H) {
    const size_t q = (size_t)fy * c.W + fx;
    r = c.rgb[q * 3]; g = c.rgb[q * 3 + 1]; b = c.rgb[q * 3 + 2];
    d = c.depth[q];
  }
  rgb_out[p * 3] = r; rgb_out[p * 3 + 1] = g; rgb_out[p * 3 + 2] = b;
  depth_out[p] = d;
}

hipError_t launch_crop_raw(const se3tn_crop& c, uint8_t* rgb_out, uint16_t* depth_out, hipStream_t st) {
  hipLaunchKernelGGL(crop_raw_kernel, dim3((RES * RES + 255) / 256), dim3(256), 0, st, c, rgb_out, depth_out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Tail: one 256-thread workgroup per pair.  head = zero-bordered [n,13,13,1024] (trans 0-511 | rot
// 512-1023): summing all 169 rows equals summing the 121 interior pixels.
// Thread t averages channels 4t..4t+3 (coalesced 4 KB rows), multiplies by
// its 3x4 slice of the head's FC matrix; wave-shuffle + LDS reduction over the head's 128 threads.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// AdaptiveAvgPool2d(1) + Linear(512, 3) + tanh of both heads + the float64 pose update (se3_tracknet.py:100-110, datasets.py:159-175)
// for the configurations whose last conv stores the head map (small batches; large ones fuse this into wino_tail_kernel).
// 16 workgroups per pair, one per 64-channel slice: a single workgroup reading the whole 692 KB map is bound by what ONE compute
// unit can pull (11.8 us at batch 1, however its loop is arranged: EXPERIMENTS item 47).  Thread (pg, col) sums pixels pg, pg + 16, ...
// of channels 4 col .. 4 col + 3 (11 loads, all in flight); the 16 pixel groups, then the 16 columns' partial dot products, are added
// in a fixed order; the slice's three partial logits go to fcpart[pair][head][slice][3] with device scope, and the LAST of the
// 16 workgroups to arrive (a counter per pair, re-armed by that workgroup) adds the slices in slice order, applies bias + tanh and
// composes the pose -- one launch, bitwise reproducible whatever the arrival order.
__global__ __launch_bounds__(256) void tail_kernel(const float* __restrict__ head, const float* __restrict__ fc_w,
                                                   const float* __restrict__ fc_b, float* __restrict__ logits,
                                                   float* __restrict__ trans, float* __restrict__ rot,
                                                   const double* __restrict__ poseA, double* __restrict__ poseB, double tn, double rn,
                                                   float* __restrict__ fcpart, int* __restrict__ arrive, int* done_flag, int done_seq) {
  __shared__ float4 psum[16][16];
  __shared__ float dots[16][3];
  __shared__ int last;
  const int i = blockIdx.x >> 4, sl16 = blockIdx.x & 15;    // pair, 64-channel slice (0-7 trans head, 8-15 rot head)
  const int t = threadIdx.x, col = t & 15, pg = t >> 4;
  constexpr int PP = (S4 + 2) * (S4 + 2);
  const float* src = head + (size_t)i * PP * 1024 + sl16 * 64 + col * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < (PP + 15) / 16; ++k) {
    const int p = pg + 16 * k;
    if (p < PP) {
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)p * 1024);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  psum[pg][col] = s;
  __syncthreads();
  if (t < 16) {
    float4 m = psum[0][t];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 v = psum[k][t]; m.x += v.x; m.y += v.y; m.z += v.z; m.w += v.w; }
    const float inv = (float)(S4 * S4);
    m.x /= inv; m.y /= inv; m.z /= inv; m.w /= inv;
    const int hd = sl16 >> 3, cl = (sl16 & 7) * 64 + t * 4;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      const float4 w = *reinterpret_cast<const float4*>(fc_w + (hd * 3 + o) * 512 + cl);
      dots[t][o] = m.x * w.x + m.y * w.y + m.z * w.z + m.w * w.w;
    }
  }
  __syncthreads();
  if (t == 0) {
    float* mine = fcpart + ((size_t)i * 16 + sl16) * 3;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float d = dots[0][o];
#pragma unroll
      for (int k = 1; k < 16; ++k) d += dots[k][o];
      __hip_atomic_store(mine + o, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // written through: another XCD reads it below
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = __hip_atomic_fetch_add(arrive + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 15;
    if (last) __hip_atomic_store(arrive + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last) return;                                   // (uniform: `last` is a shared word)
  __shared__ float outv[6];
  if (t < 6) {
    const int h = t / 3, o = t - h * 3;
    const float* p = fcpart + ((size_t)i * 16 + h * 8) * 3 + o;
    float lg = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) lg += __hip_atomic_load(p + k * 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lg += fc_b[h * 4 + o];
    const float y = tanhf(lg);
    logits[i * 6 + t] = lg;
    outv[t] = y;
    if (h == 0) { if (trans) trans[i * 3 + o] = y; }
    else        { if (rot) rot[i * 3 + o] = y; }
    if (done_flag) __threadfence_system();   // (se3tn_on_track: the outputs live in mapped host memory)
  }
  if (poseA == nullptr) return;
  __syncthreads();
  if (t == 0) {
    pose_compose(outv, poseA + (size_t)i * 16, poseB + (size_t)i * 16, tn, rn);
    if (done_flag && i == 0) {   // the host polls this word instead of waiting for the stream (one frame, n == 1)
      __threadfence_system();
      __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

hipError_t launch_tail(const float* head, const float* fc_w, const float* fc_b, float* logits,
                       float* trans, float* rot, const double* poseA, double* poseB, double tn,
                       double rn, int n, hipStream_t st, float* fcpart, int* arrive, int* done_flag, int done_seq) {
  hipLaunchKernelGGL(tail_kernel, dim3(n * 16), dim3(256), 0, st, head, fc_w, fc_b, logits, trans, rot,
                     poseA, poseB, tn, rn, fcpart, arrive, done_flag, done_seq);
  return hipGetLastError();
}

// The same tail fed by the PARTIAL SUMS of the last head conv (conv_slices_small at batch 1-5: part[slice][head][n x 121][512], 8 slices)
// instead of its reduced output map: this kernel adds the slices in slice order, the folded bias, the residual (the block's input,
// network_modules.py:118-120) and applies the ReLU per pixel, THEN pools -- the conv_reduce_kernel launch between the conv and the tail
// (4.4 us + a launch boundary) is gone and the 44 KB-per-pair head map is never written.  Measured before building
// (scripts/probes/consumer_reduce.hip, profiles/r06_probe_consumer_reduce.txt): with the tail's 16 workgroups of 64 channels the eight-fold
// reads cost 7.7 us (16 compute units cannot pull 4 MB fast enough); as 64 workgroups of 16 channels 2.2 us.  So: 64 workgroups per
// pair (2 heads x 32 slices of 16 channels); thread (pg, col) = pixels pg and pg + 64 x channels 4 col .. 4 col + 3 (16 + 2 loads, all in
// flight); pixel groups, then columns, added in a fixed order; fcpart[pair][head][32][3]; the last arriver finishes as above.
template <int CH, int SLICES>      // channels per workgroup (16 | 32), partial-sum slices
__global__ __launch_bounds__(256) void tail_parts_kernel(const float* __restrict__ part, size_t slice_stride, int M,
                                                         const float* __restrict__ bias, const float* __restrict__ res, int res_ld,
                                                         const float* __restrict__ fc_w, const float* __restrict__ fc_b,
                                                         float* __restrict__ logits, float* __restrict__ trans, float* __restrict__ rot,
                                                         const double* __restrict__ poseA, double* __restrict__ poseB, double tn, double rn,
                                                         float* __restrict__ fcpart, int* __restrict__ arrive, int* done_flag, int done_seq) {
  constexpr int COLS = CH / 4, PGS = 256 / COLS, NSL = 512 / CH, WGS = 2 * NSL;   // float4 columns, pixel groups, slices per head
  constexpr int HW = S4 * S4, PPT = (HW + PGS - 1) / PGS;                         // pixels per thread: 2 | 4
  __shared__ float4 psum[PGS][COLS];
  __shared__ float4 qsum[4][COLS];
  __shared__ float dots[COLS][3];
  __shared__ int last;
  const int i = blockIdx.x / WGS, slw = blockIdx.x - i * WGS;     // pair, CH-channel slice (first half: trans head, second half: rot head)
  const int hd = slw / NSL, c0 = (slw - hd * NSL) * CH;
  const int t = threadIdx.x, col = t % COLS, pg = t / COLS;
  const int c = c0 + col * 4;
  const float4 b = *reinterpret_cast<const float4*>(bias + hd * 512 + c);
  // every load of this thread first (PPT x (SLICES + 1), all in flight), then the sums in slice order
  float4 v[PPT][SLICES], r[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = min(pg + PGS * k, HW - 1);
    const float* src = part + ((size_t)hd * M + (size_t)i * HW + p) * 512 + c;
#pragma unroll
    for (int q = 0; q < SLICES; ++q) v[k][q] = *reinterpret_cast<const float4*>(src + q * slice_stride);
    const int py = p / S4, px = p - py * S4;
    const size_t opix = ((size_t)i * (S4 + 2) + py + 1) * (S4 + 2) + px + 1;
    r[k] = *reinterpret_cast<const float4*>(res + opix * res_ld + hd * 512 + c);
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    if (pg + PGS * k < HW) {
      float4 a = v[k][0];
#pragma unroll
      for (int q = 1; q < SLICES; ++q) { a.x += v[k][q].x; a.y += v[k][q].y; a.z += v[k][q].z; a.w += v[k][q].w; }   // conv_reduce_kernel's order
      a.x = fmaxf((a.x + b.x) + r[k].x, 0.f); a.y = fmaxf((a.y + b.y) + r[k].y, 0.f);     // bias, then residual, then ReLU
      a.z = fmaxf((a.z + b.z) + r[k].z, 0.f); a.w = fmaxf((a.w + b.w) + r[k].w, 0.f);     // (apply_epilogue<1>)
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
  }
  psum[pg][col] = s;
  __syncthreads();
  if (t < 4 * COLS) {                                        // a quarter of the pixel groups per thread, then the 4 quarters: fixed order
    const int cc = t % COLS, qd = t / COLS;
    float4 m = psum[qd * (PGS / 4)][cc];
#pragma unroll
    for (int k = 1; k < PGS / 4; ++k) { const float4 u = psum[qd * (PGS / 4) + k][cc]; m.x += u.x; m.y += u.y; m.z += u.z; m.w += u.w; }
    qsum[qd][cc] = m;
  }
  __syncthreads();
  if (t < COLS) {
    float4 m = qsum[0][t];
#pragma unroll
    for (int k = 1; k < 4; ++k) { const float4 u = qsum[k][t]; m.x += u.x; m.y += u.y; m.z += u.z; m.w += u.w; }
    const float inv = (float)HW;
    m.x /= inv; m.y /= inv; m.z /= inv; m.w /= inv;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      const float4 w = *reinterpret_cast<const float4*>(fc_w + (hd * 3 + o) * 512 + c0 + t * 4);
      dots[t][o] = m.x * w.x + m.y * w.y + m.z * w.z + m.w * w.w;
    }
  }
  __syncthreads();
  if (t == 0) {
    float* mine = fcpart + ((size_t)i * WGS + slw) * 3;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float d = dots[0][o];
#pragma unroll
      for (int k = 1; k < COLS; ++k) d += dots[k][o];
      __hip_atomic_store(mine + o, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // written through: another XCD reads it below
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = __hip_atomic_fetch_add(arrive + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == WGS - 1;
    if (last) __hip_atomic_store(arrive + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last) return;                                   // (uniform: `last` is a shared word)
  __shared__ float outv[6];
  if (t < 6) {
    const int h = t / 3, o = t - h * 3;
    const float* p = fcpart + ((size_t)i * WGS + h * NSL) * 3 + o;
    float pl[NSL];
#pragma unroll
    for (int k = 0; k < NSL; ++k) pl[k] = __hip_atomic_load(p + k * 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float lg = 0.f;
#pragma unroll
    for (int k = 0; k < NSL; ++k) lg += pl[k];           // slice order
    lg += fc_b[h * 4 + o];
    const float y = tanhf(lg);
    logits[i * 6 + t] = lg;
    outv[t] = y;
    if (h == 0) { if (trans) trans[i * 3 + o] = y; }
    else        { if (rot) rot[i * 3 + o] = y; }
    if (done_flag) __threadfence_system();   // (se3tn_on_track: the outputs live in mapped host memory)
  }
  if (poseA == nullptr) return;
  __syncthreads();
  if (t == 0) {
    pose_compose(outv, poseA + (size_t)i * 16, poseB + (size_t)i * 16, tn, rn);
    if (done_flag && i == 0) {
      __threadfence_system();
      __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

hipError_t launch_tail_parts(const float* part, int slices, size_t slice_stride, int M, const float* bias, const float* res, int res_ld,
                             const float* fc_w, const float* fc_b, float* logits, float* trans, float* rot, const double* poseA,
                             double* poseB, double tn, double rn, int n, hipStream_t st, float* fcpart, int* arrive, int* done_flag,
                             int done_seq, int ch) {
  if (slices != 8 || (ch != 16 && ch != 32)) return hipErrorInvalidValue;
  if (ch == 16)
    hipLaunchKernelGGL((tail_parts_kernel<16, 8>), dim3(n * 64), dim3(256), 0, st, part, slice_stride, M, bias, res, res_ld, fc_w, fc_b,
                       logits, trans, rot, poseA, poseB, tn, rn, fcpart, arrive, done_flag, done_seq);
  else
    hipLaunchKernelGGL((tail_parts_kernel<32, 8>), dim3(n * 32), dim3(256), 0, st, part, slice_stride, M, bias, res, res_ld, fc_w, fc_b,
                       logits, trans, rot, poseA, poseB, tn, rn, fcpart, arrive, done_flag, done_seq);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void padded_nhwc_to_nchw_kernel(const float* __restrict__ in,
                                                                   float* __restrict__ out, int h, int w,
                                                                   int c, int total, int split) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // output index (n, c, y, x)
  if (idx >= total) return;
  const int x = idx % w, t1 = idx / w;
  const int y = t1 % h, t2 = t1 / h;
  const int ch = t2 % c, n = t2 / c;
  const size_t pix = (size_t)(n * (h + 2) + y + 1) * (w + 2) + x + 1;
  if (split) {  // f16x3 mode: value = hi + lo
    const _Float16* row = reinterpret_cast<const _Float16*>(in + pix * c + (ch >> 5) * 32);
    out[idx] = (float)row[ch & 31] + (float)row[32 + (ch & 31)];
  } else {
    out[idx] = in[pix * c + ch];
  }
}

hipError_t launch_padded_nhwc_to_nchw(const float* in, float* out, int n, int h, int w, int c, int split,
                                      hipStream_t st) {
  const int total = n * h * w * c;
  hipLaunchKernelGGL(padded_nhwc_to_nchw_kernel, dim3((total + 255) / 256), dim3(256), 0, st, in, out, h, w, c, total, split);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// f16x3 mode: split panels of one packed weight matrix, derived on the device (init time; the blob carries float32 only).
// One workgroup per cout row: max |w| over the row -> exact power-of-two scale (split_exponent) -> hi | lo f16 parts.
// rows x [cout][32] layout of the 3x3 panels (row_words = 32, rows = chunks x 9 taps, hi at +0, lo at +32 halves), or the
// stem's [64][204] (one "row" of 200 used words per cout, 16-byte entries 4 hi | 4 lo).  weights.cpp: split_blob_host is the
// host statement of the same arithmetic (checked bit for bit in tests/test_gpu_parity.py).
// wino != 0: Winograd planes U [chunk][nf = wino][cout][32], one workgroup per (frequency f = blockIdx.y, cout): its rows are the
// chunks only (stride nf cout 32), so that every frequency gets its own scale (the planes differ by up to ~2 orders of magnitude)
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, _Float16* __restrict__ ws,
                                                             float* __restrict__ sc, int rows, int cout, int stem, int wino) {
  __shared__ float red[256];
  const int o = blockIdx.x, t = threadIdx.x;
  const int per_row = stem ? 200 : 32;
  const size_t row_stride = stem ? 0 : wino ? (size_t)wino * cout * 32 : (size_t)cout * 32;   // words between rows of one cout
  const size_t base = stem ? (size_t)o * 204 : wino ? ((size_t)blockIdx.y * cout + o) * 32 : (size_t)o * 32;
  if (wino) sc += (size_t)blockIdx.y * cout;
  float mx = 0.f;
  for (int i = t; i < rows * per_row; i += 256) {
    const int r = i / per_row, ci = i - r * per_row;
    mx = fmaxf(mx, fabsf(w[base + r * row_stride + ci]));
  }
  red[t] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) red[t] = fmaxf(red[t], red[t + s]);
    __syncthreads();
  }
  const int k = split_exponent(red[0]);
  const float scl = ldexpf(1.0f, k);
  if (t == 0) sc[o] = ldexpf(1.0f, -k);
  for (int i = t; i < rows * per_row; i += 256) {
    const int r = i / per_row, ci = i - r * per_row;
    const size_t word = base + r * row_stride + ci;
    const float v = w[word] * scl;                                   // exact (power of two)
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    // 3x3 panels: a 32-word row = 32 hi | 32 lo halves;   stem: a 4-word entry = 4 hi | 4 lo halves
    const size_t h = stem ? ((size_t)o * 204 + (ci & ~3)) * 2 + (ci & 3) : (base + r * row_stride) * 2 + ci;
    ws[h] = hi;
    ws[h + (stem ? 4 : 32)] = lo;
  }
}

hipError_t launch_split_weights(const float* blob, const BlobLayout& L, float* split, const SplitLayout& S, hipStream_t st) {
  hipError_t e = hipMemsetAsync(split, 0, S.total * sizeof(float), st);
  if (e != hipSuccess) return e;
  const Conv3* spec = conv_specs();
  for (int id = 0; id < NUM_CONV3; ++id)
    for (int g = 0; g < spec[id].groups; ++g) {
      const size_t gw = conv3_words(spec[id].cin, spec[id].cout) * g;
      hipLaunchKernelGGL(split_weights_kernel, dim3(spec[id].cout), dim3(256), 0, st, blob + L.conv_w[id] + gw,
                         reinterpret_cast<_Float16*>(split + S.conv_ws[id] + gw), split + S.conv_sc[id] + (size_t)spec[id].cout * g,
                         spec[id].cin / 32 * 9, spec[id].cout, 0, 0);
    }
  for (int br = 0; br < 2; ++br)
    hipLaunchKernelGGL(split_weights_kernel, dim3(64), dim3(256), 0, st, blob + L.stem_w + (size_t)br * 64 * 204,
                       reinterpret_cast<_Float16*>(split + S.stem_ws + (size_t)br * 64 * 204), split + S.stem_sc + br * 64, 1, 64, 1, 0);
  return hipGetLastError();
}

hipError_t launch_split_wino_u(const float* U, float* Us, float* scale, int cin, int cout, int nf, hipStream_t st) {
  hipLaunchKernelGGL(split_weights_kernel, dim3(cout, nf), dim3(256), 0, st, U, reinterpret_cast<_Float16*>(Us), scale, cin / 32, cout,
                     0, nf);
  return hipGetLastError();
}

}  // namespace se3tn
