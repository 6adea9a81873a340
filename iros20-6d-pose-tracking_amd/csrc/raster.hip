// HIP rasteriser for the "rendered image A" of Tracker.render_window (predict.py:193-215), replacing the reference's OpenGL
// renderer vispy_renderer.py:47-178.  It writes rgbA (uint8 176x176x3) and depthA (uint16 mm) on
// the device, i.e. straight into the input of the preprocessing kernel -- no GL context, no glReadPixels, no host round trip.
//
// OpenGL leaves sub-pixel precision, interpolation arithmetic and the float -> unorm conversion to the implementation, so "the
// reference's bytes" exist only relative to ONE implementation.  The committed goldens (tests/golden/gl_swiftshader*.npz) are the
// UNMODIFIED reference class running on SwiftShader 4.1 (OpenGL ES 3.0, conformant).  This file executes that implementation's
// fixed-function arithmetic operation by operation -- float32 without contraction (the Makefile compiles it with
// -ffp-contract=off), integers for coverage -- and the goldens are reproduced byte for byte (rgb and depth; tests/test_gl_swiftshader.py).
// The statement of the rules, each confirmed bit for bit against the live library, is oracle/ss_rules.py (test infrastructure):
//   vertex      clip = (P V) p, products accumulated left to right;  z' = (z + w) / 2;  rhw = 1 / w;
//               X = rint(X0 + (x rhw) Wh s), Y likewise: window coordinates in 1 / s pixel (s = 2^sub_bits; GL_SUBPIXEL_BITS = 4 there,
//               se3tn_set_raster_rule selects 4 or 8) with the pixel centres on multiples of s;  Z = z' rhw
//   coverage    exact integers on (X, Y): top-left rule in GL window coordinates = rows ceil(Ymin / s) <= y < ceil(Ymax / s), columns
//               ceil(xl(y)) <= x < ceil(xr(y)); no face culling (the reference selects glCullFace but never enables GL_CULL_FACE)
//   rotation    v0 <- the vertex with the largest clip w
//   depth       z = (C + (y - Y0/s) B) + (x - X0/s) A from the integer deltas, coordinates formed per 2x2 quad; LESS, in draw order
//               (here: 64-bit atomicMin on (z | triangle index) -- the same winner)
//   varyings    attribute / w planes from the matrix M, 1 / w interpolated, rcp = 1 / w + one Newton step, value = plane * rcp
//   clipping    Sutherland-Hodgman in clip space (near, far, left, right, top, bottom), polygon vertices snapped, outline walked
//   shader      vispy_renderer.py:54-76 operation by operation; colour -> UNORM8 through a truncated 16-bit value
//   read-back   rows bottom-up as glReadPixels returns them (the reference reshapes as-is), depth -> millimetres by
//               vispy_renderer.py:163-169 under the NumPy generation selected by se3tn_set_offset_rule (float32 | float64 scalars)
//
// Second mode (a.mode == 1): the reference's OTHER renderer, offscreen_renderer.py:48-83 (pyrender), selected by
// dataset_info['renderer'] == 'pyrenderer' (predict.py:161-164) for textured .obj models: a FULL camera frame
// (IntrinsicsCamera(fx,fy,cx,cy, znear 0.1, zfar 2.0)), the scene lit by ambient light [1,1,1] only -> fragment colour = base colour
// (Kd x texture, trilinear with box-filtered mip levels, REPEAT wrap; or the vertex colour), rows flipped on read-back, depth
// linearised to camera z.  Coverage, depth and vertex colours follow the same rules exactly; the texture FILTER is float32
// arithmetic here (the GL implementation filters in 16-bit fixed point): textured colours agree to a few / 255.
//
// Four launches: vertices -> clip / window space (+ z-buffer / queue clears); one thread per triangle scatters (z | triangle id) keys;
// raster_queue_kernel: large bounding boxes by one wave per triangle, triangles crossing the frustum clipped and their outline walked
// by one wave; one thread per pixel re-derives the plane equations of the winning triangle, interpolates and shades.
#include "se3tn_internal.h"

namespace se3tn {

constexpr int RASTER_BIG_PX = 256;   // bounding boxes above this many pixels go to the wave-per-triangle path of raster_queue_kernel
constexpr int CLIP_RIGHT = 1, CLIP_TOP = 2, CLIP_FAR = 4, CLIP_LEFT = 8, CLIP_BOTTOM = 16, CLIP_NEAR = 32;

// batched launch: this workgroup's instance (blockIdx.y) -- uniforms from the instance table, scratch and outputs at its offset
__device__ __forceinline__ RasterArgs raster_instance(const RasterArgs& a0) {
  RasterArgs a = a0;
  if (a0.inst != nullptr) {
    const size_t b = blockIdx.y;
    const RasterInstance* __restrict__ I = a0.inst + b;
#pragma unroll
    for (int k = 0; k < 16; ++k) a.PV[k] = I->PV[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) a.light[k] = I->light[k];
    a.dA = I->dA; a.dB = I->dB;
    const size_t px = (size_t)a0.rw * a0.rh;
    a.vpost += b * a0.V; a.vsnap += b * a0.V;
    a.zbuf += b * px;
    a.big += b * (size_t)(1 + a0.F); a.clipq += b * (size_t)(1 + a0.F);
    a.rgb += b * px * 3; a.depth += b * px;
  }
  return a;
}

__global__ __launch_bounds__(256) void raster_vertex_kernel(const RasterArgs a0) {
  const RasterArgs a = raster_instance(a0);
  const int i = blockIdx.x * 256 + threadIdx.x;
  // this launch also clears the z-buffer and the two queue counters (they are first touched by the NEXT launch)
  for (int p = i; p < a.rw * a.rh; p += gridDim.x * 256) a.zbuf[p] = ~0ull;
  if (i == 0) { a.big[0] = 0; a.clipq[0] = 0; }
  if (i >= a.V) return;
  const float px = a.verts[3 * i], py = a.verts[3 * i + 1], pz = a.verts[3 * i + 2];
  float c[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float acc = a.PV[4 * r] * px;
    acc = acc + a.PV[4 * r + 1] * py;
    acc = acc + a.PV[4 * r + 2] * pz;
    acc = acc + a.PV[4 * r + 3] * 1.0f;
    c[r] = acc;
  }
  const float w = c[3];
  const float zc = (c[2] + w) * 0.5f;
  const float rhw = 1.0f / (w == 0.f ? 1.0f : w);
  const float s = (float)(1 << a.sub_bits);
  const float Wx = (float)a.rw * 0.5f * s, Hx = (float)a.rh * 0.5f * s;
  const float X0 = Wx - 0.5f * s, Y0 = Hx - 0.5f * s;
  const float Xf = X0 + (c[0] * rhw) * Wx;
  const float Yf = Y0 + (c[1] * rhw) * Hx;
  const int X = fabsf(Xf) < 1.0e9f ? __float2int_rn(Xf) : INT_MIN;
  const int Y = fabsf(Yf) < 1.0e9f ? __float2int_rn(Yf) : INT_MIN;
  a.vpost[i] = make_float4(c[0], c[1], zc, w);
  a.vsnap[i] = make_int4(X, Y, __float_as_int(zc * rhw), __float_as_int(rhw));
}

__device__ __forceinline__ int clip_flags(const float4 p) {
  return (p.x > p.w ? CLIP_RIGHT : 0) | (p.y > p.w ? CLIP_TOP : 0) | (p.z > p.w ? CLIP_FAR : 0) | (p.x < -p.w ? CLIP_LEFT : 0) |
         (p.y < -p.w ? CLIP_BOTTOM : 0) | (p.z < 0.f ? CLIP_NEAR : 0);
}

// everything derived once per triangle
struct TriSetup {
  float dx, dy;          // X0 / s, Y0 / s of the rotated v0
  float zA, zB, zC;
  float M[3][3];         // rows: rotated vertices; columns: d/dx, d/dy, constant of (1/w-weighted) barycentric planes
  int rot[3];            // rotated order -> position in the triangle (0..2)
};

__device__ __forceinline__ void rotate_max_w(const float w0, const float w1, const float w2, int rot[3]) {
  const float wmax = fmaxf(fmaxf(w0, w1), w2);
  int r0 = 0, r1 = 1, r2 = 2;
  if (wmax == w1) { const int t = r0; r0 = r1; r1 = r2; r2 = t; }      // both conditions on the ORIGINAL w's
  if (wmax == w2) { const int t = r2; r2 = r1; r1 = r0; r0 = t; }
  rot[0] = r0; rot[1] = r1; rot[2] = r2;
}

template <bool Varyings>
__device__ __forceinline__ void tri_setup(const float4 post[3], const int4 snap[3], int sub_bits, TriSetup& s) {
  rotate_max_w(post[0].w, post[1].w, post[2].w, s.rot);
  const int4 s0 = snap[s.rot[0]], s1 = snap[s.rot[1]], s2 = snap[s.rot[2]];
  const float fs = (float)(1 << sub_bits), inv_s = sub_bits == 4 ? 0.0625f : 0.00390625f;
  s.dx = (float)s0.x * inv_s;
  s.dy = (float)s0.y * inv_s;
  const int X1 = s1.x - s0.x, Y1 = s1.y - s0.y, X2 = s2.x - s0.x, Y2 = s2.y - s0.y;
  const float fx1 = (float)X1, fy1 = (float)Y1, fx2 = (float)X2, fy2 = (float)Y2;
  const float z0 = __int_as_float(s0.z);
  const float z1 = __int_as_float(s1.z) - z0, z2 = __int_as_float(s2.z) - z0;
  const float D = 1.0f / (fx1 * fy2 - fx2 * fy1);
  s.zA = ((fy2 * z1 - fy1 * z2) * D) * fs;
  s.zB = ((fx1 * z2 - fx2 * z1) * D) * fs;
  s.zC = z0 * 1.0f + 0.0f;
  if (Varyings) {
    const float w1 = post[s.rot[1]].w, w2 = post[s.rot[2]].w;
    const float rhw0 = __int_as_float(s0.w);
    const float px1 = (w1 * inv_s) * fx1, py1 = (w1 * inv_s) * fy1;
    const float px2 = (w2 * inv_s) * fx2, py2 = (w2 * inv_s) * fy2;
    const float area = px1 * py2 - px2 * py1;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) s.M[r][c] = 0.f;
    s.M[0][2] = rhw0;
    if (area != 0.f) {
      const float Ai = 1.0f / area;
      const float Dm = Ai * rhw0;
      s.M[0][0] = (py1 * w2 - py2 * w1) * Dm;
      s.M[0][1] = (px2 * w1 - px1 * w2) * Dm;
      s.M[1][0] = py2 * Ai;
      s.M[1][1] = -px2 * Ai;
      s.M[2][0] = -py1 * Ai;
      s.M[2][1] = px1 * Ai;
    }
  }
}

// (x - X0/s) as the 2x2-quad loop forms it: float(even) + (odd - d)
__device__ __forceinline__ float quad_coord(int x, float d) { return (float)(x & ~1) + ((float)(x & 1) - d); }

__device__ __forceinline__ float plane_eval(float A, float B, float C, float xx, float yy) { return (C + yy * B) + xx * A; }

__device__ __forceinline__ unsigned sortable(float z) {
  const unsigned b = __float_as_uint(z);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unsortable(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__device__ __forceinline__ void depth_write(const RasterArgs& a, const TriSetup& s, int x, int y, int t) {
  const float z = plane_eval(s.zA, s.zB, s.zC, quad_coord(x, s.dx), quad_coord(y, s.dy));
  if (!(z < 1.0f)) return;   // GL_LESS against the cleared 1.0 (and NaN)
  const unsigned long long key = ((unsigned long long)sortable(z) << 32) | (unsigned)t;
  atomicMin(a.zbuf + (size_t)y * a.rw + x, key);
}

// integer edge functions of an unclipped triangle, orientation made positive
struct Edges {
  long long ax[3], ay[3], dx[3], dy[3];
  bool tie[3];
  int x0, x1, y0, y1;   // pixel bounding box [x0, x1) x [y0, y1)
};

__device__ __forceinline__ int ceil_shift(int v, int bits) { return (v + (1 << bits) - 1) >> bits; }

__device__ __forceinline__ bool edges_setup(const RasterArgs& a, const int4 snap[3], bool d, Edges& e) {
  int X[3] = {snap[0].x, snap[1].x, snap[2].x}, Y[3] = {snap[0].y, snap[1].y, snap[2].y};
  const long long area2 = (long long)(X[1] - X[0]) * (Y[2] - Y[0]) - (long long)(Y[1] - Y[0]) * (X[2] - X[0]);
  // the outline is walked in the direction the FLOAT area selects (d); if rounding ever made the two disagree the span tables
  // come out inverted and nothing is drawn
  if (area2 == 0 || (area2 < 0) != d) return false;
  if (area2 < 0) { int t = X[1]; X[1] = X[2]; X[2] = t; t = Y[1]; Y[1] = Y[2]; Y[2] = t; }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int b = k == 2 ? 0 : k + 1;
    e.ax[k] = X[k]; e.ay[k] = Y[k];
    e.dx[k] = X[b] - X[k]; e.dy[k] = Y[b] - Y[k];
    e.tie[k] = e.dy[k] < 0 || (e.dy[k] == 0 && e.dx[k] > 0);
  }
  const int sb = a.sub_bits;
  e.x0 = max(ceil_shift(min(X[0], min(X[1], X[2])), sb), 0);
  e.x1 = min(ceil_shift(max(X[0], max(X[1], X[2])), sb), a.rw);
  e.y0 = max(ceil_shift(min(Y[0], min(Y[1], Y[2])), sb), 0);
  e.y1 = min(ceil_shift(max(Y[0], max(Y[1], Y[2])), sb), a.rh);
  return e.x0 < e.x1 && e.y0 < e.y1;
}

__device__ __forceinline__ bool edges_inside(const Edges& e, int x, int y, int sub_bits) {
  const long long px = (long long)x << sub_bits, py = (long long)y << sub_bits;
  bool in = true;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const long long E = e.dx[k] * (py - e.ay[k]) - e.dy[k] * (px - e.ax[k]);
    in = in && (E > 0 || (E == 0 && e.tie[k]));
  }
  return in;
}

// loads a triangle; returns 0: nothing to draw, 1: unclipped, 2: needs clipping.  d = outline direction
__device__ __forceinline__ int tri_load(const RasterArgs& a, int t, float4 post[3], int4 snap[3], bool& d) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int v = a.faces[3 * t + k];
    post[k] = a.vpost[v];
    snap[k] = a.vsnap[v];
  }
  const int f0 = clip_flags(post[0]), f1 = clip_flags(post[1]), f2 = clip_flags(post[2]);
  if (f0 & f1 & f2) return 0;
  const float x0 = (float)snap[0].x, x1 = (float)snap[1].x, x2 = (float)snap[2].x;
  const float y0 = (float)snap[0].y, y1 = (float)snap[1].y, y2 = (float)snap[2].y;
  float A = ((y2 - y0) * x1 + (y1 - y2) * x0) + (y0 - y1) * x2;
  if (A == 0.0f || A != A) return 0;
  if ((__float_as_int(post[0].w) ^ __float_as_int(post[1].w) ^ __float_as_int(post[2].w)) < 0) A = -A;
  d = A < 0.0f;
  return (f0 | f1 | f2) ? 2 : 1;
}

__global__ __launch_bounds__(256) void raster_triangle_kernel(const RasterArgs a0) {
  const RasterArgs a = raster_instance(a0);
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.F) return;
  float4 post[3];
  int4 snap[3];
  bool d;
  const int kind = tri_load(a, t, post, snap, d);
  if (kind == 0) return;
  if (kind == 2) {   // crosses the frustum: clipped and walked by a whole wave (raster_queue_kernel)
    const int k = atomicAdd(a.clipq, 1);
    a.clipq[1 + k] = t;
    return;
  }
  Edges e;
  if (!edges_setup(a, snap, d, e)) return;
  // a triangle that covers many pixels (coarse mesh, close-up) would serialise this thread: it is queued for
  // raster_queue_kernel, where a whole wave walks its bounding box.  The z-buffer keys make the result independent of
  // the order in which the queue is filled.
  if ((e.x1 - e.x0) * (e.y1 - e.y0) > RASTER_BIG_PX) {
    const int k = atomicAdd(a.big, 1);
    a.big[1 + k] = t;
    return;
  }
  TriSetup s;
  tri_setup<false>(post, snap, a.sub_bits, s);
  for (int y = e.y0; y < e.y1; ++y)
    for (int x = e.x0; x < e.x1; ++x)
      if (edges_inside(e, x, y, a.sub_bits)) depth_write(a, s, x, y, t);
}

// one wave per queued triangle, lanes stride over the pixels of its bounding box
constexpr int BIG_BLOCKS = 128, CLIP_BLOCKS = 64;   // raster_queue_kernel: blocks [0, 128) walk the big-triangle queue, the rest clip
__device__ __forceinline__ void raster_big_part(const RasterArgs& a) {
  const int nbig = a.big[0];
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (BIG_BLOCKS * 256) >> 6;
  for (int k = wave; k < nbig; k += nwaves) {
    const int t = a.big[1 + k];
    float4 post[3];
    int4 snap[3];
    bool d;
    if (tri_load(a, t, post, snap, d) != 1) continue;
    Edges e;
    if (!edges_setup(a, snap, d, e)) continue;
    TriSetup s;
    tri_setup<false>(post, snap, a.sub_bits, s);
    const int bw = e.x1 - e.x0, npx = bw * (e.y1 - e.y0);
    for (int q = lane; q < npx; q += 64) {
      const int y = e.y0 + q / bw, x = e.x0 + q % bw;
      if (edges_inside(e, x, y, a.sub_bits)) depth_write(a, s, x, y, t);
    }
  }
}

// ---- triangles that cross the frustum ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 clip_edge(const float4 Vi, const float4 Vj, float di, float dj) {
  const float D = 1.0f / (dj - di);
  return make_float4((dj * Vi.x - di * Vj.x) * D, (dj * Vi.y - di * Vj.y) * D, (dj * Vi.z - di * Vj.z) * D, (dj * Vi.w - di * Vj.w) * D);
}

__device__ __forceinline__ float plane_dist(const float4 v, int plane) {
  switch (plane) {
    case 0: return v.z;            // near
    case 1: return v.w - v.z;      // far
    case 2: return v.w + v.x;      // left
    case 3: return v.w - v.x;      // right
    case 4: return v.w - v.y;      // top
    default: return v.w + v.y;     // bottom
  }
}
__device__ __forceinline__ void plane_exact(float4& b, int plane) {
  switch (plane) {
    case 0: b.z = 0.f; break;
    case 1: b.z = b.w; break;
    case 2: b.x = -b.w; break;
    case 3: b.x = b.w; break;
    case 4: b.y = b.w; break;
    default: b.y = -b.w; break;
  }
}

constexpr int CLIP_MAX_ROWS = 2048;   // frame heights the outline tables hold (se3tn_render_frame checks)

// one wave (the first of a clip block of raster_queue_kernel) per queued triangle: lane 0 clips the triangle to a polygon and snaps it, the wave
// walks the polygon's edges IN ORDER (a later edge overwrites an earlier one on a shared row, as the span tables of the
// implementation do), then rasterises the rows of the outline
__global__ __launch_bounds__(256) void raster_queue_kernel(const RasterArgs a0) {
  const RasterArgs a = raster_instance(a0);
  __shared__ int tab[2][CLIP_MAX_ROWS];   // [0] left, [1] right
  __shared__ int PX[12], PY[12], pn;
  if (blockIdx.x < BIG_BLOCKS) {
    raster_big_part(a);
    return;
  }
  const int nq = a.clipq[0];
  if (nq == 0 || threadIdx.x >= 64) return;   // the clip blocks work with their first wave (barriers count live waves only)
  const int lane = threadIdx.x;
  const int sb = a.sub_bits;
  for (int q = blockIdx.x - BIG_BLOCKS; q < nq; q += CLIP_BLOCKS) {
    const int t = a.clipq[1 + q];
    float4 post[3];
    int4 snap[3];
    bool d;
    if (tri_load(a, t, post, snap, d) != 2) continue;   // (uniform over the wave)
    __syncthreads();
    if (lane == 0) {
      const int fo = clip_flags(post[0]) | clip_flags(post[1]) | clip_flags(post[2]);
      const int flag_of[6] = {CLIP_NEAR, CLIP_FAR, CLIP_LEFT, CLIP_RIGHT, CLIP_TOP, CLIP_BOTTOM};
      float4 P[2][12];
      int n = 3, cur = 0;
      P[0][0] = post[0]; P[0][1] = post[1]; P[0][2] = post[2];
      for (int pl = 0; pl < 6 && n >= 3; ++pl) {
        if (!(fo & flag_of[pl])) continue;
        int m = 0;
        for (int i = 0; i < n; ++i) {
          const int j = i == n - 1 ? 0 : i + 1;
          const float di = plane_dist(P[cur][i], pl), dj = plane_dist(P[cur][j], pl);
          if (di >= 0.f) {
            P[cur ^ 1][m++] = P[cur][i];
            if (dj < 0.f) {
              float4 b = clip_edge(P[cur][i], P[cur][j], di, dj);
              plane_exact(b, pl);
              P[cur ^ 1][m++] = b;
            }
          } else if (dj > 0.f) {
            float4 b = clip_edge(P[cur][j], P[cur][i], dj, di);
            plane_exact(b, pl);
            P[cur ^ 1][m++] = b;
          }
        }
        n = m;
        cur ^= 1;
      }
      if (n < 3) n = 0;
      const float s = (float)(1 << sb);
      const float Wx = (float)a.rw * 0.5f * s, Hx = (float)a.rh * 0.5f * s;
      const float X0 = Wx - 0.5f * s, Y0 = Hx - 0.5f * s;
      for (int i = 0; i < n; ++i) {
        const float4 v = P[cur][i];
        const float rhw = v.w != 0.f ? 1.0f / v.w : 1.0f;
        PX[i] = __float2int_rn(X0 + (v.x * rhw) * Wx);
        PY[i] = __float2int_rn(Y0 + (v.y * rhw) * Hx);
      }
      pn = n;
    }
    __syncthreads();
    const int n = pn;
    if (n < 3) continue;
    int ylo = INT_MAX, yhi = INT_MIN;
    for (int i = 0; i < n; ++i) { ylo = min(ylo, PY[i]); yhi = max(yhi, PY[i]); }
    const int ymin = max(ceil_shift(ylo, sb), 0), ymax = min(ceil_shift(yhi, sb), a.rh);
    for (int y = ymin + lane; y < ymax; y += 64) { tab[0][y] = 0; tab[1][y] = 0; }
    __syncthreads();
    const int di = d ? 1 : 0;
    for (int i = 0; i < n; ++i) {
      const int ia = (i + 1 - di) % n, ib = (i + di) % n;
      const int Xa = PX[ia], Ya = PY[ia], Xb = PX[ib], Yb = PY[ib];
      if (Ya != Yb) {
        const bool swap = Yb < Ya;
        const long long X1 = swap ? Xb : Xa, Y1 = swap ? Yb : Ya, X2 = swap ? Xa : Xb, Y2 = swap ? Ya : Yb;
        const int y1 = max(ceil_shift((int)Y1, sb), 0), y2 = min(ceil_shift((int)Y2, sb), a.rh);
        const long long DX = X2 - X1, DY = Y2 - Y1, den = DY << sb;
        for (int y = y1 + lane; y < y2; y += 64) {
          const long long num = X1 * DY + DX * (((long long)y << sb) - Y1);
          long long x = num / den;               // truncation towards zero ...
          if (num % den > 0) x += 1;             // ... made a ceiling
          tab[swap ? 1 : 0][y] = (int)min(max(x, 0ll), (long long)a.rw);
        }
      }
      __syncthreads();
    }
    TriSetup s;
    tri_setup<false>(post, snap, sb, s);
    for (int y = ymin; y < ymax; ++y) {
      const int xl = tab[0][y], xr = tab[1][y];
      for (int x = xl + lane; x < xr; x += 64) depth_write(a, s, x, y, t);
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

// float colour -> the byte an 8-bit unorm target stores: through a truncated 16-bit value
__device__ __forceinline__ uint8_t unorm8(float c) {
  c = fminf(fmaxf(c, 0.f), 1.f);
  const int c16 = (int)(c * 65535.0f);
  return (uint8_t)((c16 - (c16 >> 8) + 128) >> 8);
}

struct Interp {
  float xx, yy, rcp;
};
__device__ __forceinline__ Interp interp_at(const TriSetup& s, int x, int y) {
  Interp it;
  it.xx = quad_coord(x, s.dx);
  it.yy = quad_coord(y, s.dy);
  const float A = (s.M[0][0] + s.M[1][0]) + s.M[2][0], B = (s.M[0][1] + s.M[1][1]) + s.M[2][1], C = (s.M[0][2] + s.M[1][2]) + s.M[2][2];
  const float w = plane_eval(A, B, C, it.xx, it.yy);
  float rcp = 1.0f / w;
  rcp = (rcp + rcp) - (w * rcp) * rcp;
  it.rcp = rcp;
  return it;
}
// attribute values a0, a1, a2 of the ROTATED vertices
__device__ __forceinline__ float interp(const TriSetup& s, const Interp& it, float a0, float a1, float a2) {
  const float A = (a0 * s.M[0][0] + a1 * s.M[1][0]) + a2 * s.M[2][0];
  const float B = (a0 * s.M[0][1] + a1 * s.M[1][1]) + a2 * s.M[2][1];
  const float C = (a0 * s.M[0][2] + a1 * s.M[1][2]) + a2 * s.M[2][2];
  return plane_eval(A, B, C, it.xx, it.yy) * it.rcp;
}

__global__ __launch_bounds__(256) void raster_resolve_kernel(const RasterArgs a0) {
  const RasterArgs a = raster_instance(a0);
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.rw * a.rh) return;
  // GL window row j counts bottom-up, and glReadPixels returns rows in that order.  Mode 0: the reference reshapes the buffer
  // as-is (`bottom` is the LARGER Y = cy - fy y/z, i.e. the smaller OpenCV v, so array row j is already top-down in the OpenCV
  // image).  Mode 1: pyrender flips the rows on read-back.
  const int j = p / a.rw, i = p - j * a.rw;
  const size_t o = a.mode == 1 ? (size_t)(a.rh - 1 - j) * a.rw + i : (size_t)p;
  const unsigned long long key = a.zbuf[p];
  uint8_t* rgb = a.rgb + o * 3;
  if (key == ~0ull) {
    rgb[0] = 0; rgb[1] = 0; rgb[2] = 0;
    a.depth[o] = 0;
    return;
  }
  const int t = (int)(unsigned)key;
  const float zw = unsortable((unsigned)(key >> 32));
  int vid[3];
  float4 post[3];
  int4 snap[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    vid[k] = a.faces[3 * t + k];
    post[k] = a.vpost[vid[k]];
    snap[k] = a.vsnap[vid[k]];
  }
  TriSetup s;
  tri_setup<true>(post, snap, a.sub_bits, s);
  const int f0 = vid[s.rot[0]], f1 = vid[s.rot[1]], f2 = vid[s.rot[2]];
  const Interp it = interp_at(s, i, j);
  if (a.mode == 1) {
    // pyrender, ambient light only: colour = Kd * (texture | vertex colour); no lighting term
    float col[3];
    if (a.tex) {
      // uv at the three corners of this pixel's 2x2 quad on the winning triangle's planes -> level of detail
      float uv[3][2];
      const int xq = i & ~1, yq = j & ~1;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const Interp iq = interp_at(s, xq + (k == 1 ? 1 : 0), yq + (k == 2 ? 1 : 0));
#pragma unroll
        for (int c = 0; c < 2; ++c) uv[k][c] = interp(s, iq, a.uv[2 * f0 + c], a.uv[2 * f1 + c], a.uv[2 * f2 + c]);
      }
      const float u = interp(s, it, a.uv[2 * f0], a.uv[2 * f1], a.uv[2 * f2]);
      const float v = interp(s, it, a.uv[2 * f0 + 1], a.uv[2 * f1 + 1], a.uv[2 * f2 + 1]);
      const float dudx = (uv[1][0] - uv[0][0]) * a.tw, dvdx = (uv[1][1] - uv[0][1]) * a.th;
      const float dudy = (uv[2][0] - uv[0][0]) * a.tw, dvdy = (uv[2][1] - uv[0][1]) * a.th;
      const float rho = fmaxf(sqrtf(dudx * dudx + dvdx * dvdx), sqrtf(dudy * dudy + dvdy * dvdy));
      const float lod = fminf(fmaxf(log2f(fmaxf(rho, 1e-8f)), 0.f), (float)(a.tlevels - 1));
      const int l0i = (int)floorf(lod), l1i = min(l0i + 1, a.tlevels - 1);
      const float fl = lod - (float)l0i;
      float c0[3], c1[3];
      sample_bilinear(a, l0i, u, v, c0);
      sample_bilinear(a, l1i, u, v, c1);
#pragma unroll
      for (int c = 0; c < 3; ++c) col[c] = (c0[c] + fl * (c1[c] - c0[c])) * (1.0f / 255.0f);
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) col[c] = interp(s, it, a.colors[3 * f0 + c], a.colors[3 * f1 + c], a.colors[3 * f2 + c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] = unorm8(col[c] * a.kd[c]);
    // depth buffer -> camera z as pyrender linearises it (float32), then (depth * 1000).astype(np.uint16), predict.py:211
    const float z_ndc = zw * 2.0f - 1.0f;
    const float dist = (float)(2.0 * R_NEAR_D * R_FAR_D) / ((float)(R_FAR_D + R_NEAR_D) - z_ndc * (float)(R_FAR_D - R_NEAR_D));
    a.depth[o] = (uint16_t)(int)(dist * 1000.f);
    return;
  }
  float pos[3], nrm[3], col[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    pos[c] = interp(s, it, a.verts[3 * f0 + c], a.verts[3 * f1 + c], a.verts[3 * f2 + c]);
    nrm[c] = interp(s, it, a.normals[3 * f0 + c], a.normals[3 * f1 + c], a.normals[3 * f2 + c]);
    col[c] = interp(s, it, a.colors[3 * f0 + c], a.colors[3 * f1 + c], a.colors[3 * f2 + c]);
  }
  // vispy_renderer.py:54-76
  const float lx = -a.light[0] - pos[0], ly = -a.light[1] - pos[1], lz = -a.light[2] - pos[2];
  const float dot = (lx * lx + ly * ly) + lz * lz;
  const float rsq = 1.0f / sqrtf(dot);
  const float ndl = (nrm[0] * (lx * rsq) + nrm[1] * (ly * rsq)) + nrm[2] * (lz * rsq);
  const float light3 = 0.4f * fmaxf(ndl, 0.f) + 0.65f;
#pragma unroll
  for (int c = 0; c < 3; ++c) rgb[c] = unorm8(light3 * col[c]);
  // vispy_renderer.py:163-169: distance = B / (depth * -2.0 + 1.0 - A) * -1;  distance[distance >= B / (A + 1)] = 0;  (distance * 1000).astype(uint16)
  const float tz = zw * -2.0f + 1.0f;
  if (a.numpy_rule == SE3TN_OFFSET_RULE_NUMPY2) {       // float64 scalars promote the array (NEP 50)
    double dist = a.dB / ((double)tz - a.dA) * -1.0;
    if (dist >= a.dB / (a.dA + 1.0)) dist = 0.0;
    a.depth[o] = (uint16_t)(long long)(dist * 1000.0);
  } else {                                              // value-based casting: every operation float32
    float dist = ((float)a.dB / (tz - (float)a.dA)) * -1.0f;
    if (dist >= (float)(a.dB / (a.dA + 1.0))) dist = 0.f;
    a.depth[o] = (uint16_t)(int)(dist * 1000.f);
  }
}

hipError_t launch_raster(const RasterArgs& a, hipStream_t st, int instances) {
  // one thread per vertex; the z-buffer clear strides over the grid (at least 128 blocks, or one per 256 pixels if that is fewer)
  // instances > 1 (a.inst set): grid.y = instance, the same four launches for all poses
  const int vb = (a.V + 255) / 256, zb = (a.rw * a.rh + 255) / 256;
  const int clear_blocks = zb < 128 ? zb : 128;
  const unsigned ny = a.inst ? (unsigned)instances : 1u;
  hipLaunchKernelGGL(raster_vertex_kernel, dim3(vb > clear_blocks ? vb : clear_blocks, ny), dim3(256), 0, st, a);
  hipLaunchKernelGGL(raster_triangle_kernel, dim3((a.F + 255) / 256, ny), dim3(256), 0, st, a);
  hipLaunchKernelGGL(raster_queue_kernel, dim3(BIG_BLOCKS + CLIP_BLOCKS, ny), dim3(256), 0, st, a);
  hipLaunchKernelGGL(raster_resolve_kernel, dim3((a.rw * a.rh + 255) / 256, ny), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace se3tn
