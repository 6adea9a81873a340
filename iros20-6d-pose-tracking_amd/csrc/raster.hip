// HIP rasteriser for the "rendered image A" of Tracker.render_window (predict.py:193-215), replacing
// the reference's OpenGL renderer vispy_renderer.py:47-178.  It writes
// rgbA (uint8 176x176x3) and depthA (uint16 mm) on the device, i.e. straight into the input of the
// preprocessing kernel -- no GL context, no glReadPixels, no host round trip.
//
// What is restated from the reference (parity unpinned: a GL driver's sub-pixel snapping and
// depth-buffer format are not observable offline):
//   * vertex shader (:78-98): gl_Position = proj * view * vec4(pos,1) with
//       view = diag(1,-1,-1,1) * ob_in_cv_cam,  proj = ortho(left,right,bottom,top) * K-projection
//       (update_cam_mat :135-150, near 0.1 m, far 2.0 m) -- here folded analytically:
//       X = fx x/z + cx,  Y = cy - fy y/z  (the window is given in these coordinates),  w = z;
//   * fixed function: viewport 176x176, pixel centres at +0.5, top-left fill rule, depth test LESS
//     against a cleared 1.0 (:153-156), depth affine in window space, other varyings
//     perspective-correct.  NO face culling: the reference calls gloo.set_cull_face('back') (:155),
//     which only selects glCullFace -- GL_CULL_FACE is never enabled (set_state(depth_test=True) only),
//     and it could not be: the window's y axis is flipped by ortho(top<bottom), so "front" faces are
//     clockwise here.  Both windings are rasterised and the depth test keeps the nearest surface;
//   * fragment shader (:54-76): lightDir = normalize(-light_direction - fragpos) (object space),
//       colour = clamp((0.4 max(dot(n, lightDir), 0) + 0.65) * vertexColour, 0, 1)  -> UNORM8;
//   * read-back (:160-169): rows bottom-up (=> top-down in the OpenCV image), distance recovered from
//     the depth buffer = camera z, background 0, depth_mm = uint16(z * 1000).
//
// Second mode (a.mode == 1): the reference's OTHER renderer, offscreen_renderer.py:48-83 (pyrender), selected by
// dataset_info['renderer'] == 'pyrenderer' (predict.py:161-164) for textured .obj models: a FULL camera frame
// (IntrinsicsCamera(fx,fy,cx,cy, znear 0.1, zfar 2.0): pixel (i, r) covers u in [i,i+1), v in [r,r+1)), the scene lit
// by ambient light [1,1,1] only -> fragment colour = base colour (Kd x texture, trilinear with box-filtered mip levels,
// REPEAT wrap; or the vertex colour), depth returned as linear camera z; Tracker.render_window then crops it with
// crop_bbox (predict.py:209-213).  Parity with pyrender's shader / a GL driver's texture filtering is unpinned.
//
// Clipping: fragments are depth-tested against the near / far planes per pixel, so a triangle that crosses the near plane in
// FRONT of the camera is cut exactly; a triangle with a vertex at or behind the camera plane (z <= 0) is dropped whole, not
// clipped -- it cannot occur for a tracked object (range 0.3-2 m, radius < 0.3 m).
//
// Three kernels (+ raster_big_kernel for triangles with large bounding boxes): vertices -> window space; one thread per triangle scatters (depth | triangle id) keys
// with 64-bit atomicMin (deterministic z-buffer, ties broken by triangle index); one thread per pixel
// re-derives the barycentrics of the winning triangle, interpolates and shades.
#include "se3tn_internal.h"

namespace se3tn {

constexpr float R_NEAR = 0.1f, R_FAR = 2.0f;
constexpr int RASTER_BIG_PX = 256;   // bounding boxes above this many pixels go to raster_big_kernel

__global__ __launch_bounds__(256) void raster_vertex_kernel(const RasterArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.V) return;
  const float px = a.verts[3 * i], py = a.verts[3 * i + 1], pz = a.verts[3 * i + 2];
  const float x = a.M[0] * px + a.M[1] * py + a.M[2] * pz + a.M[3];
  const float y = a.M[4] * px + a.M[5] * py + a.M[6] * pz + a.M[7];
  const float z = a.M[8] * px + a.M[9] * py + a.M[10] * pz + a.M[11];
  const float iw = 1.0f / z;  // clip w = z
  const float X = a.fx * x * iw + a.cx, Y = a.cy - a.fy * y * iw;
  const float xn = (2.f * X - a.right - a.left) / (a.right - a.left);
  const float yn = (2.f * Y - a.top - a.bottom) / (a.top - a.bottom);
  // z_ndc = -A + B / z with A = -(n+f)/(f-n), B = -2nf/(f-n)
  const float A = -(R_NEAR + R_FAR) / (R_FAR - R_NEAR), B = -2.f * R_NEAR * R_FAR / (R_FAR - R_NEAR);
  const float zn = -A + B * iw;
  a.vwin[i] = make_float4((xn + 1.f) * (a.rw * 0.5f), (yn + 1.f) * (a.rh * 0.5f), (zn + 1.f) * 0.5f, z > 0.f ? iw : -1.f);
}

__device__ __forceinline__ float edge_fn(float ax, float ay, float bx, float by, float px, float py) {
  return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}
// top-left rule for an edge a->b of a counter-clockwise triangle (y up): left edges go down, top edges
// are horizontal and go left
__device__ __forceinline__ bool top_left(float ax, float ay, float bx, float by) {
  const float dx = bx - ax, dy = by - ay;
  return dy < 0.f || (dy == 0.f && dx < 0.f);
}

__device__ __forceinline__ bool covers(const float4 v0, const float4 v1, const float4 v2, float area, float px,
                                       float py, float& l0, float& l1, float& l2) {
  const float e0 = edge_fn(v1.x, v1.y, v2.x, v2.y, px, py);
  const float e1 = edge_fn(v2.x, v2.y, v0.x, v0.y, px, py);
  const float e2 = edge_fn(v0.x, v0.y, v1.x, v1.y, px, py);
  const bool in0 = e0 > 0.f || (e0 == 0.f && top_left(v1.x, v1.y, v2.x, v2.y));
  const bool in1 = e1 > 0.f || (e1 == 0.f && top_left(v2.x, v2.y, v0.x, v0.y));
  const bool in2 = e2 > 0.f || (e2 == 0.f && top_left(v0.x, v0.y, v1.x, v1.y));
  const float ia = 1.0f / area;
  l0 = e0 * ia; l1 = e1 * ia; l2 = e2 * ia;
  return in0 && in1 && in2;
}

__global__ __launch_bounds__(256) void raster_triangle_kernel(const RasterArgs a) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.F) return;
  const float4 v0 = a.vwin[a.faces[3 * t]];
  float4 v1 = a.vwin[a.faces[3 * t + 1]], v2 = a.vwin[a.faces[3 * t + 2]];
  if (v0.w <= 0.f || v1.w <= 0.f || v2.w <= 0.f) return;  // behind the camera: not clipped, dropped
  float area = edge_fn(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y);
  if (area == 0.f || area != area) return;  // degenerate
  if (area < 0.f) { const float4 tmp = v1; v1 = v2; v2 = tmp; area = -area; }  // orient counter-clockwise
  const float xmin = fminf(v0.x, fminf(v1.x, v2.x)), xmax = fmaxf(v0.x, fmaxf(v1.x, v2.x));
  const float ymin = fminf(v0.y, fminf(v1.y, v2.y)), ymax = fmaxf(v0.y, fmaxf(v1.y, v2.y));
  const int i0 = max(0, (int)floorf(xmin - 0.5f)), i1 = min(a.rw - 1, (int)ceilf(xmax - 0.5f));
  const int j0 = max(0, (int)floorf(ymin - 0.5f)), j1 = min(a.rh - 1, (int)ceilf(ymax - 0.5f));
  if (i1 < i0 || j1 < j0) return;
  // a triangle that covers many pixels (coarse mesh, close-up) would serialise this thread: it is queued for
  // raster_big_kernel, where a whole wave walks its bounding box.  The z-buffer keys make the result independent of
  // the order in which the queue is filled.
  if (a.big && (i1 - i0 + 1) * (j1 - j0 + 1) > RASTER_BIG_PX) {
    const int k = atomicAdd(a.big, 1);
    a.big[1 + k] = t;
    return;
  }
  for (int j = j0; j <= j1; ++j)
    for (int i = i0; i <= i1; ++i) {
      float l0, l1, l2;
      if (!covers(v0, v1, v2, area, i + 0.5f, j + 0.5f, l0, l1, l2)) continue;
      const float zw = l0 * v0.z + l1 * v1.z + l2 * v2.z;
      if (!(zw >= 0.f && zw < 1.f)) continue;  // near / far planes; LESS against the cleared 1.0
      const unsigned long long key = ((unsigned long long)__float_as_uint(zw) << 32) | (unsigned)t;
      atomicMin(a.zbuf + j * a.rw + i, key);
    }
}

// one wave per queued triangle, lanes stride over the pixels of its bounding box
__global__ __launch_bounds__(256) void raster_big_kernel(const RasterArgs a) {
  const int nbig = a.big[0];
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
  for (int k = wave; k < nbig; k += nwaves) {
    const int t = a.big[1 + k];
    const float4 v0 = a.vwin[a.faces[3 * t]];
    float4 v1 = a.vwin[a.faces[3 * t + 1]], v2 = a.vwin[a.faces[3 * t + 2]];
    float area = edge_fn(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y);
    if (area < 0.f) { const float4 tmp = v1; v1 = v2; v2 = tmp; area = -area; }
    const float xmin = fminf(v0.x, fminf(v1.x, v2.x)), xmax = fmaxf(v0.x, fmaxf(v1.x, v2.x));
    const float ymin = fminf(v0.y, fminf(v1.y, v2.y)), ymax = fmaxf(v0.y, fmaxf(v1.y, v2.y));
    const int i0 = max(0, (int)floorf(xmin - 0.5f)), i1 = min(a.rw - 1, (int)ceilf(xmax - 0.5f));
    const int j0 = max(0, (int)floorf(ymin - 0.5f)), j1 = min(a.rh - 1, (int)ceilf(ymax - 0.5f));
    const int bw = i1 - i0 + 1, npx = bw * (j1 - j0 + 1);
    for (int q = lane; q < npx; q += 64) {
      const int j = j0 + q / bw, i = i0 + q % bw;
      float l0, l1, l2;
      if (!covers(v0, v1, v2, area, i + 0.5f, j + 0.5f, l0, l1, l2)) continue;
      const float zw = l0 * v0.z + l1 * v1.z + l2 * v2.z;
      if (!(zw >= 0.f && zw < 1.f)) continue;
      const unsigned long long key = ((unsigned long long)__float_as_uint(zw) << 32) | (unsigned)t;
      atomicMin(a.zbuf + j * a.rw + i, key);
    }
  }
}

// texel (x, y) of mip level l (RGB uint8, levels stored back to back, level l is max(tw >> l, 1) x max(th >> l, 1)); REPEAT wrap
__device__ __forceinline__ void sample_bilinear(const RasterArgs& a, int level, float u, float v, float* out) {
  const int w = max(a.tw >> level, 1), h = max(a.th >> level, 1);
  const uint8_t* base = a.tex + a.tex_off[level];
  // image row 0 is the TOP of the picture, v = 0 its bottom (OBJ / GL convention)
  const float x = u * w - 0.5f, y = (1.0f - v) * h - 0.5f;
  const float xf = floorf(x), yf = floorf(y);
  const float ax = x - xf, ay = y - yf;
  int x0 = (int)xf % w, y0 = (int)yf % h;
  if (x0 < 0) x0 += w;
  if (y0 < 0) y0 += h;
  const int x1 = (x0 + 1) % w, y1 = (y0 + 1) % h;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float t00 = base[((size_t)y0 * w + x0) * 3 + c], t10 = base[((size_t)y0 * w + x1) * 3 + c];
    const float t01 = base[((size_t)y1 * w + x0) * 3 + c], t11 = base[((size_t)y1 * w + x1) * 3 + c];
    out[c] = (t00 * (1.f - ax) + t10 * ax) * (1.f - ay) + (t01 * (1.f - ax) + t11 * ax) * ay;
  }
}

__global__ __launch_bounds__(256) void raster_resolve_kernel(const RasterArgs a) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.rw * a.rh) return;
  // GL window row j counts bottom-up, and glReadPixels returns rows in that order; the reference
  // reshapes the buffer as-is.  `bottom` is the LARGER Y = cy - fy y/z, i.e. the smaller OpenCV v, so
  // array row j is already top-down in the OpenCV image: output index = window index.
  const int j = p / a.rw, i = p - j * a.rw;
  const unsigned long long key = a.zbuf[p];
  uint8_t* rgb = a.rgb + (size_t)p * 3;
  if (key == ~0ull) {
    rgb[0] = 0; rgb[1] = 0; rgb[2] = 0;
    a.depth[p] = 0;
    return;
  }
  const int t = (int)(unsigned)key;
  const int f0 = a.faces[3 * t];
  int f1 = a.faces[3 * t + 1], f2 = a.faces[3 * t + 2];
  const float4 v0 = a.vwin[f0];
  float4 v1 = a.vwin[f1], v2 = a.vwin[f2];
  float area = edge_fn(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y);
  if (area < 0.f) {  // same orientation as the scatter pass
    const float4 tv = v1; v1 = v2; v2 = tv;
    const int tf = f1; f1 = f2; f2 = tf;
    area = -area;
  }
  float l0, l1, l2;
  covers(v0, v1, v2, area, i + 0.5f, j + 0.5f, l0, l1, l2);
  // perspective-correct weights
  const float q0 = l0 * v0.w, q1 = l1 * v1.w, q2 = l2 * v2.w;
  const float iq = 1.0f / (q0 + q1 + q2);
  const float b0 = q0 * iq, b1 = q1 * iq, b2 = q2 * iq;
  const float zw = __uint_as_float((unsigned)(key >> 32));
  const float A = -(R_NEAR + R_FAR) / (R_FAR - R_NEAR), B = -2.f * R_NEAR * R_FAR / (R_FAR - R_NEAR);
  const float dist = B / (zw * -2.0f + 1.0f - A) * -1.0f;   // linear depth recovered from the depth buffer == camera z
  if (a.mode == 1) {
    // pyrender, ambient light only: colour = Kd * (texture | vertex colour); no lighting term
    float col[3];
    if (a.tex) {
      // perspective-correct uv at this pixel and at its right / upper neighbours -> level of detail
      float uv[3][2];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        float m0, m1, m2;
        covers(v0, v1, v2, area, i + 0.5f + (s == 1 ? 1.f : 0.f), j + 0.5f + (s == 2 ? 1.f : 0.f), m0, m1, m2);
        const float r0 = m0 * v0.w, r1 = m1 * v1.w, r2 = m2 * v2.w, ir = 1.0f / (r0 + r1 + r2);
#pragma unroll
        for (int c = 0; c < 2; ++c)
          uv[s][c] = (r0 * a.uv[2 * f0 + c] + r1 * a.uv[2 * f1 + c] + r2 * a.uv[2 * f2 + c]) * ir;
      }
      const float dudx = (uv[1][0] - uv[0][0]) * a.tw, dvdx = (uv[1][1] - uv[0][1]) * a.th;
      const float dudy = (uv[2][0] - uv[0][0]) * a.tw, dvdy = (uv[2][1] - uv[0][1]) * a.th;
      const float rho = fmaxf(sqrtf(dudx * dudx + dvdx * dvdx), sqrtf(dudy * dudy + dvdy * dvdy));
      const float lod = fminf(fmaxf(log2f(fmaxf(rho, 1e-8f)), 0.f), (float)(a.tlevels - 1));
      const int l0i = (int)floorf(lod), l1i = min(l0i + 1, a.tlevels - 1);
      const float fl = lod - (float)l0i;
      float c0[3], c1[3];
      sample_bilinear(a, l0i, uv[0][0], uv[0][1], c0);
      sample_bilinear(a, l1i, uv[0][0], uv[0][1], c1);
#pragma unroll
      for (int c = 0; c < 3; ++c) col[c] = (c0[c] + fl * (c1[c] - c0[c])) * (1.0f / 255.0f);
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) col[c] = b0 * a.colors[3 * f0 + c] + b1 * a.colors[3 * f1 + c] + b2 * a.colors[3 * f2 + c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] = (uint8_t)(int)rintf(fminf(fmaxf(col[c] * a.kd[c], 0.f), 1.f) * 255.f);
    a.depth[p] = (uint16_t)(dist * 1000.f);      // (depth * 1000).astype(np.uint16), predict.py:211
    return;
  }
  float pos[3], nrm[3], col[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    pos[c] = b0 * a.verts[3 * f0 + c] + b1 * a.verts[3 * f1 + c] + b2 * a.verts[3 * f2 + c];
    nrm[c] = b0 * a.normals[3 * f0 + c] + b1 * a.normals[3 * f1 + c] + b2 * a.normals[3 * f2 + c];
    col[c] = b0 * a.colors[3 * f0 + c] + b1 * a.colors[3 * f1 + c] + b2 * a.colors[3 * f2 + c];
  }
  float lx = -a.light[0] - pos[0], ly = -a.light[1] - pos[1], lz = -a.light[2] - pos[2];
  const float il = rsqrtf(lx * lx + ly * ly + lz * lz);
  lx *= il; ly *= il; lz *= il;
  const float diff = 0.4f * fmaxf(nrm[0] * lx + nrm[1] * ly + nrm[2] * lz, 0.f) + 0.65f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = fminf(fmaxf(diff * col[c], 0.f), 1.f);
    rgb[c] = (uint8_t)(int)rintf(v * 255.f);
  }
  // distance = B / (zw * -2 + 1 - A) * -1  (vispy_renderer.py:164-169) == camera z
  a.depth[p] = (dist >= B / (A + 1.f)) ? (uint16_t)0 : (uint16_t)(dist * 1000.f);
}

hipError_t launch_raster(const RasterArgs& a, hipStream_t st) {
  hipError_t e = hipMemsetAsync(a.zbuf, 0xff, sizeof(unsigned long long) * a.rw * a.rh, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(raster_vertex_kernel, dim3((a.V + 255) / 256), dim3(256), 0, st, a);
  if (a.big) {
    e = hipMemsetAsync(a.big, 0, sizeof(int), st);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(raster_triangle_kernel, dim3((a.F + 255) / 256), dim3(256), 0, st, a);
  if (a.big) hipLaunchKernelGGL(raster_big_kernel, dim3(128), dim3(256), 0, st, a);
  hipLaunchKernelGGL(raster_resolve_kernel, dim3((a.rw * a.rh + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace se3tn
