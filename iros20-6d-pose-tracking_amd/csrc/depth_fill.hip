// Depth hole filling of the live-camera front end: Utils.py:455-514 `fill_depth` as predict_ros.py:38-41 calls it
//     depth = fill_depth(depth_mm / 1e3, max_depth=2.0, extrapolate=False);  depth_mm' = (depth * 1000).astype(uint16)
// (SURVEY.md 8f rank 4).  The reference runs it as a chain of OpenCV calls on a float32 image of INVERTED depth
// (valid pixels become max_depth - d, holes stay 0, so grey dilation = "take the nearest surface"):
//     cv2.dilate(diamond 5x5) -> cv2.morphologyEx(CLOSE, 5x5) -> holes := cv2.dilate(7x7)
//     [-> extrapolate to the top of the image -> holes := cv2.dilate(31x31)]
//     -> cv2.medianBlur(5) -> cv2.bilateralFilter(5, 1.5, 2.0) | cv2.GaussianBlur((5,5), 0) -> invert back.
// Here every step is one streaming kernel over the H x W image (one thread per pixel, the image is 1.2 MB and
// lives in L2); the OpenCV conventions restated (OpenCV is not available offline -- see DESIGN.md section 4):
//   * dilate / erode: flat structuring element anchored at its centre, pixels outside the image are ignored
//     (BORDER_CONSTANT with morphologyDefaultBorderValue = -inf / +inf);
//   * medianBlur (float32, ksize 5): exact median of the 25 taps, BORDER_REPLICATE;
//   * bilateralFilter (float32): BORDER_REFLECT_101, taps within the radius-2 disc (12 + centre), colour weight from
//     a 4096-bin table of exp(-d^2 / (2 sigma_c^2)) over the image's [min, max] range with linear interpolation
//     (bilateralFilter_32f), centre tap weight 1;
//   * GaussianBlur((5,5), sigma 0): the fixed 5-tap kernel [1 4 6 4 1] / 16, rows then columns, BORDER_REFLECT_101.
// Selections (max / min / median) are exact, so the chain is bit-exact against the CPU oracle up to the median;
// the two blurs are float32 sums whose association follows the scalar OpenCV loops (SIMD builds of OpenCV differ in
// the last ulp from each other as well).
#include <cmath>

#include "se3tn_internal.h"

namespace se3tn {

struct BilateralTaps { float w[12]; };

__device__ __forceinline__ int reflect101(int p, int n) {
  if (p < 0) p = -p;
  if (p >= n) p = 2 * n - 2 - p;
  return p < 0 ? 0 : (p >= n ? n - 1 : p);
}

// uint16 millimetres -> float32 inverted metres (Utils.py:459-471)
__global__ __launch_bounds__(256) void fd_prepare_kernel(const uint16_t* __restrict__ mm, float* __restrict__ out, int total,
                                                          float max_depth) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float d = (float)((double)mm[i] / 1e3);          // predict_ros.py:40 `depth/1e3` is float64, fill_depth casts to float32
  if (d > 0.1f) d = max_depth - d;
  out[i] = d;
}

// SHAPE 0: full (2R+1)^2 square, 1: the 5x5 diamond of Utils.py:460-467 (|dx| + |dy| <= 2).  OP 0: dilate (max), 1: erode (min).
// FILL != 0: out = (in < 0.1) ? dilated : in   (Utils.py:483-485, :497-499)
template <int OP, int SHAPE, int R, int FILL>
__global__ __launch_bounds__(256) void fd_morph_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  const float centre = in[i];
  if (FILL && !(centre < 0.1f)) { out[i] = centre; return; }
  float v = OP == 0 ? -INFINITY : INFINITY;
  for (int dy = -R; dy <= R; ++dy) {
    const int yy = y + dy;
    if ((unsigned)yy >= (unsigned)H) continue;
    const int span = SHAPE == 1 ? R - (dy < 0 ? -dy : dy) : R;
    for (int dx = -span; dx <= span; ++dx) {
      const int xx = x + dx;
      if ((unsigned)xx >= (unsigned)W) continue;
      const float t = in[yy * W + xx];
      v = OP == 0 ? fmaxf(v, t) : fminf(v, t);
    }
  }
  out[i] = v;
}

// Utils.py:488-494: every column is filled from the top of the image down to its first valid pixel with that pixel's
// value (np.argmax of an all-False column is 0: nothing changes)
__global__ __launch_bounds__(256) void fd_extrapolate_kernel(float* __restrict__ d, int H, int W) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  int top = 0;
  for (int y = 0; y < H; ++y)
    if (d[y * W + x] > 0.1f) { top = y; break; }
  const float v = d[top * W + x];
  for (int y = 0; y < top; ++y) d[y * W + x] = v;
}

// cv2.medianBlur(float32, 5): exact median of 25 taps, BORDER_REPLICATE
__global__ __launch_bounds__(256) void fd_median5_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  float v[25];
#pragma unroll
  for (int dy = -2; dy <= 2; ++dy) {
    const int yy = min(max(y + dy, 0), H - 1);
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      const int xx = min(max(x + dx, 0), W - 1);
      v[(dy + 2) * 5 + dx + 2] = in[yy * W + xx];
    }
  }
  // partial selection sort up to the 13th smallest (fully unrolled: the array stays in registers)
#pragma unroll
  for (int a = 0; a < 13; ++a) {
#pragma unroll
    for (int b = a + 1; b < 25; ++b) {
      const float lo = fminf(v[a], v[b]), hi = fmaxf(v[a], v[b]);
      v[a] = lo; v[b] = hi;
    }
  }
  out[i] = v[12];
}

// min / max of the image as order-preserving unsigned keys (any sign), mm[0] = min key, mm[1] = max key
__device__ __forceinline__ unsigned f32_key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__global__ __launch_bounds__(256) void fd_minmax_kernel(const float* __restrict__ in, int total, unsigned* __restrict__ mm) {
  unsigned lo = 0xffffffffu, hi = 0u;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const unsigned k = f32_key(in[i]);
    lo = min(lo, k); hi = max(hi, k);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, (unsigned)__shfl_xor((int)lo, o, 64));
    hi = max(hi, (unsigned)__shfl_xor((int)hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], lo); atomicMax(&mm[1], hi); }   // integer atomics: order-independent
}

// the two blurs keep multiply and add separate (as the scalar OpenCV loops and the numpy oracle do)
#pragma clang fp contract(off)

constexpr int BIL_BINS = 1 << 12;

// expLUT of bilateralFilter_32f: lut[i] = exp((i / scale_index)^2 * gauss_color_coeff), zeros once it underflowed
__global__ __launch_bounds__(256) void fd_bilateral_lut_kernel(const unsigned* __restrict__ mm, float* __restrict__ lut,
                                                               double gauss_color_coeff) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= BIL_BINS + 2) return;
  const float len = (float)((double)key_f32(mm[1]) - (double)key_f32(mm[0]));
  const float scale_index = (float)BIL_BINS / len;
  const double val = (double)i / (double)scale_index;
  lut[i] = (float)exp(val * val * gauss_color_coeff);   // (a float that underflowed to 0 stays 0 for all larger i)
}

__global__ __launch_bounds__(256) void fd_bilateral5_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                            const unsigned* __restrict__ mm, const float* __restrict__ lut,
                                                            const BilateralTaps taps) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const float vmin = key_f32(mm[0]), vmax = key_f32(mm[1]);
  const float val0 = in[i];
  if (fabs((double)vmin - (double)vmax) < 1.1920928955078125e-07) { out[i] = val0; return; }   // src.copyTo(dst)
  const float scale_index = (float)BIL_BINS / (float)((double)vmax - (double)vmin);
  const int y = i / W, x = i - y * W;
  float wsum = 1.f, sum = val0;
  // taps in OpenCV's order: i (rows) outer, j (columns) inner, r <= radius, centre skipped; the space weights
  // (float)exp(r * r * gauss_space_coeff), r = sqrt(i^2 + j^2) in double, come from the host
  int k = 0;
#pragma unroll
  for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      if ((dy == 0 && dx == 0) || dy * dy + dx * dx > 4) continue;
      const float sw = taps.w[k++];
      const float val = in[reflect101(y + dy, H) * W + reflect101(x + dx, W)];
      float alpha = fabsf(val - val0) * scale_index;
      const int idx = (int)floorf(alpha);
      alpha -= (float)idx;
      const float w = sw * (lut[idx] + alpha * (lut[idx + 1] - lut[idx]));
      sum += val * w;
      wsum += w;
    }
  out[i] = sum / wsum;
}

// cv2.GaussianBlur(depth, (5,5), 0) restricted to the valid pixels (Utils.py:509-512): separable [1 4 6 4 1]/16
template <int VERTICAL>
__global__ __launch_bounds__(256) void fd_gauss5_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  const float k[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
  float s = 0.f;
#pragma unroll
  for (int t = -2; t <= 2; ++t) {
    const float v = VERTICAL ? in[reflect101(y + t, H) * W + x] : in[y * W + reflect101(x + t, W)];
    s += v * k[t + 2];
  }
  out[i] = s;
}
__global__ __launch_bounds__(256) void fd_select_valid_kernel(const float* __restrict__ depth, const float* __restrict__ blurred,
                                                               float* __restrict__ out, int total) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float d = depth[i];
  out[i] = d > 0.1f ? blurred[i] : d;
}

// invert back (Utils.py:515-517) and (depth * 1000).astype(np.uint16) (predict_ros.py:41); optional float32 metres
__global__ __launch_bounds__(256) void fd_finish_kernel(const float* __restrict__ in, uint16_t* __restrict__ out_mm,
                                                         float* __restrict__ out_m, int total, float max_depth) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float d = in[i];
  if (d > 0.1f) d = max_depth - d;
  if (out_m) out_m[i] = d;
  if (out_mm) {
    // (depth * 1000).astype(np.uint16) as NumPy evaluates it on x86-64: truncate to int32, keep the low 16 bits.  A region
    // beyond max_depth stays NEGATIVE through dilate / close / fill / median (inverted depth max_depth - d < 0 is "empty"
    // and only its rim is filled), so e.g. -0.5 m -> -500 -> 65036, not 0.  Either value is "invalid" for OffsetDepth
    // (<= 100 or >= 2000 -> 2000, data_augmentation.py:136), but the uint16 frame is the reference's bit for bit.
    // NumPy leaves an out-of-range float -> uint16 cast undefined; this is what NumPy 2.2 (the build that made the goldens) and the
    // 1.x series do on x86-64 (cvttss2si, then the low 16 bits).  Values a 32-bit int cannot hold and NaN are made explicit here
    // instead of relying on the device's own conversion: cvttss2si returns the "integer indefinite" 0x80000000 for them -> 0.
    const float mm = d * 1000.f;
    const int iv = (mm >= -2147483648.f && mm < 2147483648.f) ? (int)mm : (int)0x80000000u;
    out_mm[i] = (uint16_t)(unsigned)iv;
  }
}

hipError_t launch_fill_depth(const FillDepthArgs& a, hipStream_t st) {
  const int total = a.H * a.W, grid = (total + 255) / 256;
  float *p = a.buf0, *q = a.buf1;
  auto swap = [&]() { float* t = p; p = q; q = t; };
  const float md = (float)a.max_depth;
  hipLaunchKernelGGL(fd_prepare_kernel, dim3(grid), dim3(256), 0, st, a.depth_mm, p, total, md);
  hipLaunchKernelGGL((fd_morph_kernel<0, 1, 2, 0>), dim3(grid), dim3(256), 0, st, p, q, a.H, a.W); swap();   // dilate, diamond
  hipLaunchKernelGGL((fd_morph_kernel<0, 0, 2, 0>), dim3(grid), dim3(256), 0, st, p, q, a.H, a.W); swap();   // close = dilate 5x5
  hipLaunchKernelGGL((fd_morph_kernel<1, 0, 2, 0>), dim3(grid), dim3(256), 0, st, p, q, a.H, a.W); swap();   //         then erode 5x5
  hipLaunchKernelGGL((fd_morph_kernel<0, 0, 3, 1>), dim3(grid), dim3(256), 0, st, p, q, a.H, a.W); swap();   // holes := dilate 7x7
  if (a.extrapolate) {
    hipLaunchKernelGGL(fd_extrapolate_kernel, dim3((a.W + 255) / 256), dim3(256), 0, st, p, a.H, a.W);
    hipLaunchKernelGGL((fd_morph_kernel<0, 0, 15, 1>), dim3(grid), dim3(256), 0, st, p, q, a.H, a.W); swap();  // holes := dilate 31x31
  }
  hipLaunchKernelGGL(fd_median5_kernel, dim3(grid), dim3(256), 0, st, p, q, a.H, a.W); swap();
  if (a.blur == 1) {         // bilateral (the reference's default)
    hipError_t e = hipMemsetAsync(a.minmax, 0xff, sizeof(unsigned), st);              // min key := 0xffffffff
    if (e == hipSuccess) e = hipMemsetAsync(a.minmax + 1, 0, sizeof(unsigned), st);   // max key := 0
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fd_minmax_kernel, dim3(64), dim3(256), 0, st, p, total, a.minmax);
    hipLaunchKernelGGL(fd_bilateral_lut_kernel, dim3((BIL_BINS + 2 + 255) / 256), dim3(256), 0, st, a.minmax, a.lut,
                       -0.5 / (a.sigma_color * a.sigma_color));
    BilateralTaps taps;
    {
      const double gauss_space_coeff = -0.5 / (a.sigma_space * a.sigma_space);
      int k = 0;
      for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx) {
          const double r = std::sqrt((double)dy * dy + (double)dx * dx);
          if (r > 2 || (dy == 0 && dx == 0)) continue;
          taps.w[k++] = (float)std::exp(r * r * gauss_space_coeff);
        }
    }
    hipLaunchKernelGGL(fd_bilateral5_kernel, dim3(grid), dim3(256), 0, st, p, q, a.H, a.W, a.minmax, a.lut, taps);
    swap();
  } else if (a.blur == 2) {  // gaussian, valid pixels only
    hipLaunchKernelGGL((fd_gauss5_kernel<0>), dim3(grid), dim3(256), 0, st, p, q, a.H, a.W);
    hipLaunchKernelGGL((fd_gauss5_kernel<1>), dim3(grid), dim3(256), 0, st, q, a.buf2, a.H, a.W);
    hipLaunchKernelGGL(fd_select_valid_kernel, dim3(grid), dim3(256), 0, st, p, a.buf2, q, total);
    swap();
  }
  hipLaunchKernelGGL(fd_finish_kernel, dim3(grid), dim3(256), 0, st, p, a.out_mm, a.out_m, total, md);
  return hipGetLastError();
}

}  // namespace se3tn
