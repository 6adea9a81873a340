// Device-side building blocks shared by the MFMA kernels (conv3x3_mfma.hip, wino_mfma.hip):
// LDS-DMA issue, the f32 / f16x3 MFMA K-step macros, split-row access, epilogue math.
#pragma once
#include "se3tn_internal.h"

#ifndef SE3TN_ABLATE
#define SE3TN_ABLATE 0  // timing ablations only (wrong results): 1 = no DMA, 2 = no barrier
#endif

namespace se3tn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// Arithmetic modes of the MFMA core (template parameter MM)
//   MM_F32   v_mfma_f32_32x32x2_f32 on float32 operands: bitwise an fmaf chain (default).
//   MM_F16X3 operands stored as "split rows": per (pixel | cout, 32-channel chunk) 128 bytes =
//            32 x f16 hi | 32 x f16 lo with x = hi + lo to 22 significant bits; the product is formed
//            as hi*hi + hi*lo + lo*hi with three v_mfma_f32_32x32x16_f16 (f32 accumulate): 16/3 = 5.3x
//            the f32 MFMA rate at f32-class error (the dropped lo*lo term is 2^-22 relative).
//            A split row has the SAME size, 16-byte slot structure and swizzle as a float32 row:
//            slots 0-3 = hi channels 0-31, slots 4-7 = lo, and slot 2 kb + hh (+4) is exactly the
//            8-half MFMA operand of lane-half hh for 16-channel block kb -- i.e. the four 8-k groups
//            of the f32 kernel ARE (hi kb0, hi kb1, lo kb0, lo kb1).
enum { MM_F32 = 0, MM_F16X3 = 1 };
// tensor element formats of the epilogue (template parameters OUTF / RESF)
enum { FMT_F32 = 0, FMT_SPLIT = 1 };

constexpr float SELU_ALPHA = 1.6732632423543772848170429916717f;
constexpr float SELU_SCALE = 1.0507009873554804934193349852946f;

__device__ __forceinline__ float selu_f(float v) {
  return v > 0.f ? SELU_SCALE * v : (SELU_SCALE * SELU_ALPHA) * expm1f(v);
}

// EPI: 0 = bias+ReLU, 1 = bias+residual+ReLU, 2 = bias+SELU
template <int EPI>
__device__ __forceinline__ float4 apply_epilogue(float4 v, const float4 b, const float4 r) {
  v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  if (EPI == 1) { v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
  if (EPI == 2) {
    v.x = selu_f(v.x); v.y = selu_f(v.y); v.z = selu_f(v.z); v.w = selu_f(v.w);
  } else {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  }
  return v;
}

// ---- LDS-DMA helpers ---------------------------------------------------------------------------
// global_load_lds_dwordx4 voff, s[base:base+1] offset:IMM
//   lane l:  LDS[M0 + IMM + 16 l] <- 16 bytes at (base + voff_l + IMM)     (IMM moves BOTH sides,
//   verified on hardware: scripts/probes/probe_glds.hip)
__device__ __forceinline__ unsigned lds_addr_of(const float* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) float*)p;
}
template <int IMM>
__device__ __forceinline__ void glds16(const float* sbase, unsigned voff_bytes, unsigned lds_byte_addr) {
#if (SE3TN_ABLATE & 1)
  return;
#endif
  // both scalar operands are wave-uniform by construction; readfirstlane makes that provable to
  // the compiler (an "s" constraint on a value it believes divergent does not assemble)
  const unsigned long long b_ = (unsigned long long)sbase;
  const unsigned blo_ = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)b_);  // (the builtin returns int)
  const unsigned bhi_ = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b_ >> 32));
  const unsigned long long sb_ = ((unsigned long long)bhi_ << 32) | (unsigned long long)blo_;
  const unsigned lds_ = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:%4\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff_bytes), "s"(lds_), "s"(sb_), "i"(IMM)
      : "memory");
}
__device__ __forceinline__ void wait_dma_and_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !(SE3TN_ABLATE & 2)
  __syncthreads();
#endif
}

// One 8-k group of a K-step: PTN x CTN x 4 MFMAs.  PVEXPR may use `i`, WVEXPR may use `j`.
#define SE3TN_MMA_GROUP(PTN, CTN, PVEXPR, WVEXPR)                                                     \
  {                                                                                                  \
    float4 pv_[PTN], wv_[CTN];                                                                       \
    _Pragma("unroll") for (int i = 0; i < PTN; ++i) pv_[i] = PVEXPR;                                 \
    _Pragma("unroll") for (int j = 0; j < CTN; ++j) wv_[j] = WVEXPR;                                 \
    _Pragma("unroll") for (int j = 0; j < CTN; ++j) _Pragma("unroll") for (int i = 0; i < PTN; ++i)  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[j].x, pv_[i].x, acc[i][j], 0, 0, 0);    \
    _Pragma("unroll") for (int j = 0; j < CTN; ++j) _Pragma("unroll") for (int i = 0; i < PTN; ++i)  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[j].y, pv_[i].y, acc[i][j], 0, 0, 0);    \
    _Pragma("unroll") for (int j = 0; j < CTN; ++j) _Pragma("unroll") for (int i = 0; i < PTN; ++i)  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[j].z, pv_[i].z, acc[i][j], 0, 0, 0);    \
    _Pragma("unroll") for (int j = 0; j < CTN; ++j) _Pragma("unroll") for (int i = 0; i < PTN; ++i)  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[j].w, pv_[i].w, acc[i][j], 0, 0, 0);    \
  }

// MM_F16X3: all four groups' fragments, then per 16-channel block kb: Whi*Phi + Whi*Plo + Wlo*Phi.
// PVEXPR(G) / WVEXPR(G): float4 fragment of group G (may use `i` / `j`).
#define SE3TN_MMA_SPLIT(PTN, CTN, PVEXPR, WVEXPR)                                                    \
  {                                                                                                  \
    half8 ph_[4][PTN], wh_[4][CTN];                                                                  \
    _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                               \
      _Pragma("unroll") for (int i = 0; i < PTN; ++i) {                                              \
        const float4 t_ = PVEXPR(gq);                                                                \
        ph_[gq][i] = __builtin_bit_cast(half8, t_);                                                  \
      }                                                                                              \
      _Pragma("unroll") for (int j = 0; j < CTN; ++j) {                                              \
        const float4 t_ = WVEXPR(gq);                                                                \
        wh_[gq][j] = __builtin_bit_cast(half8, t_);                                                  \
      }                                                                                              \
    }                                                                                                \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                               \
      _Pragma("unroll") for (int j = 0; j < CTN; ++j) _Pragma("unroll") for (int i = 0; i < PTN; ++i) \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_[kb][j], ph_[kb][i], acc[i][j], 0, 0, 0);     \
      _Pragma("unroll") for (int j = 0; j < CTN; ++j) _Pragma("unroll") for (int i = 0; i < PTN; ++i) \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_[kb][j], ph_[kb + 2][i], acc[i][j], 0, 0, 0); \
      _Pragma("unroll") for (int j = 0; j < CTN; ++j) _Pragma("unroll") for (int i = 0; i < PTN; ++i) \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_[kb + 2][j], ph_[kb][i], acc[i][j], 0, 0, 0); \
    }                                                                                                \
  }

// ---- split-row element access: 4 consecutive channels c..c+3 (c % 4 == 0) of pixel `pix` ---------
__device__ __forceinline__ size_t split_byte_off(size_t pix, int ld, int c) {
  return (pix * ld) * 4 + (size_t)(c >> 5) * 128 + (c & 31) * 2;
}
__device__ __forceinline__ float4 load_split4(const float* base, size_t pix, int ld, int c) {
  const unsigned char* p_ = reinterpret_cast<const unsigned char*>(base) + split_byte_off(pix, ld, c);
  const half4 h = *reinterpret_cast<const half4*>(p_);
  const half4 l = *reinterpret_cast<const half4*>(p_ + 64);
  return make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2],
                     (float)h[3] + (float)l[3]);
}
// returns true if a value is outside the f16 range (the caller raises the overflow flag)
__device__ __forceinline__ bool store_split4(float* base, size_t pix, int ld, int c, float4 v) {
  unsigned char* p_ = reinterpret_cast<unsigned char*>(base) + split_byte_off(pix, ld, c);
  half4 h, l;
  const float f[4] = {v.x, v.y, v.z, v.w};
  bool bad = false;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = (_Float16)f[e];
    l[e] = (_Float16)(f[e] - (float)h[e]);
    bad |= !(fabsf(f[e]) <= 65000.f);
  }
  *reinterpret_cast<half4*>(p_) = h;
  *reinterpret_cast<half4*>(p_ + 64) = l;
  return bad;
}

// padded-flat pixel index of interior pixel m (flattened over the batch) of an [n,H+2,W+2,*] tensor
__device__ __forceinline__ int padded_index(int m, int HW, int W) {
  const int n = m / HW, rem = m - n * HW;
  const int h = rem / W, w = rem - h * W;
  return (n * (HW / W + 2) + h + 1) * (W + 2) + w + 1;
}


}  // namespace se3tn
