// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the IndexTTS hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/indextts_hip.h"

// ITTS_DIAG = 1 builds libindextts_hip_diag.so (make diag): tuning overrides + in-kernel time stamps for the tools/
// scripts.  The product library has neither: no process-wide mutable state besides immutable tables.
#ifndef ITTS_DIAG
#define ITTS_DIAG 0
#endif
#if ITTS_DIAG
#include "../../include/indextts_hip_diag.h"
#endif
#ifndef ITTS_STAMPS
#define ITTS_STAMPS 0       // diagnostic build only (make diag): s_memtime stamps per workgroup, see tools/timeline_*.py
#endif
#if ITTS_STAMPS
namespace itts { extern unsigned long long* g_stamp_buf; extern unsigned long long* g_stamp_buf_sample; }
// one stamp: s_memtime of lane 0 of the workgroup into the local array st_[i] (declared by the kernel)
#define ITTS_STAMP_IF(cond, i)                                                                         \
  do {                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    if ((cond) && threadIdx.x == 0) {                                                                  \
      unsigned long long t_;                                                                           \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
      st_[i] = t_;                                                                                     \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  } while (0)
#else
#define ITTS_STAMP_IF(cond, i) do { } while (0)
#endif

namespace itts {

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;

// ---------------------------------------------------------------------------------------------------------------
// Element traits.  E = elements per 16-byte fragment chunk; KS = 4*E = K extent of one "k-step":
// one v_mfma_f32_16x16x32 (16-bit types) or four v_mfma_f32_16x16x4_f32 (fp32, exact fmaf-chain numerics).
// Fragment convention (both operands): lane l = (g<<4)|r holds 16 bytes = elements k0 + g*E + [0,E) of A row r
// (resp. B column r).  For fp32 the four elements feed four MFMAs, i.e. MFMA i consumes k = k0 + 4g + i; A and B use
// the same assignment so the product is the plain dot product over the 16 k of the step.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int E = 4, KS = 16, DT = ITTS_F32;
  typedef f32x4 frag;
  static __device__ __forceinline__ float to_f(float v) { return v; }
  static __device__ __forceinline__ float from_f(float v) { return v; }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
  }
};
template <>
struct Elem<bf16_t> {
  static constexpr int E = 8, KS = 32, DT = ITTS_BF16;
  typedef bf16x8 frag;
  static __device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }
  static __device__ __forceinline__ bf16_t from_f(float v) { return (bf16_t)v; }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Elem<f16_t> {
  static constexpr int E = 8, KS = 32, DT = ITTS_F16;
  typedef f16x8 frag;
  static __device__ __forceinline__ float to_f(f16_t v) { return (float)v; }
  static __device__ __forceinline__ f16_t from_f(float v) { return (f16_t)v; }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

template <typename F>
__device__ __forceinline__ F zero_frag() {
  F z;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(F) / sizeof(z[0])); ++i) z[i] = 0;
  return z;
}

// 16-byte global / LDS accesses through a POD carrier
struct __attribute__((aligned(16))) B16 {
  uint32_t w[4];
};
template <typename F>
__device__ __forceinline__ F ld16(const void* p) {
  B16 v = *reinterpret_cast<const B16*>(p);
  return __builtin_bit_cast(F, v);
}
// 16-byte load with the non-temporal cache policy: for bytes a launch reads exactly once (streamed weights, KV rows)
template <typename F>
__device__ __forceinline__ F ld16_nt(const void* p) {
  u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return __builtin_bit_cast(F, v);
}
template <typename F>
__device__ __forceinline__ void st16(void* p, F f) {
  *reinterpret_cast<B16*>(p) = __builtin_bit_cast(B16, f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Packed activation layout ("PA") of the decode step: X[M][K] of T stored as MFMA operand fragments, 1 KiB per (k-step,
// 16-row tile), so that the skinny GEMM reads an activation fragment as ONE contiguous 1-KiB wave-load (8 cache lines)
// instead of 16 rows x 64 bytes (16 half-used lines): the activations, re-read by every workgroup, are two thirds of a
// CU's load traffic in these GEMMs.   element (m, k) -> ((k / KS * MTP + m / 16) * 64 + (k % KS) / E * 16 + m % 16) * E
// + k % E,   MTP = ceil(M / 16) row tiles; rows up to 16 * MTP exist (padding rows are never stored by consumers).
template <typename T>
__device__ __forceinline__ int64_t pa_off(int m, int k, int mtp) {
  constexpr int E = Elem<T>::E, KS = Elem<T>::KS;
  return ((((int64_t)(k / KS) * mtp + (m >> 4)) * 64 + ((k % KS) / E) * 16 + (m & 15)) * E) + (k % E);
}

// KV cache addressing.  Contiguous form (tab == NULL): caches T [rows][H][smax][64], position j of cache row `row`.  Paged form:
// a pool T [blocks][H][bs][64] (bs = 1 << bs_log2 positions per block) and a block table int32 [rows][ITTS_KV_TAB]; position j
// of row `row` lives in block tab[row][(j >> bs_log2) % ITTS_KV_TAB] -- a RING over the position index, so a decode loop whose
// shared write position only ever grows keeps addressing a bounded table (a row's live window must stay under
// (ITTS_KV_TAB - 2) * bs positions).  Returns the ELEMENT offset of (row, head h, position j, dim 0).
__device__ __forceinline__ int64_t kv_elem_off(const int32_t* __restrict__ tab, int bs_log2, int row, int j, int H, int h, int smax) {
  if (tab == nullptr) return (((int64_t)row * H + h) * smax + j) * 64;
  const int blk = tab[row * ITTS_KV_TAB + ((j >> bs_log2) & (ITTS_KV_TAB - 1))];
  return ((((int64_t)blk * H + h) << bs_log2) + (j & ((1 << bs_log2) - 1))) * 64;
}

// transformers NewGELUActivation (gelu_new): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
__device__ __forceinline__ float gelu_new(float x) {
  const float k = 0.7978845608028654f;
  float u = k * (x + 0.044715f * x * x * x);
  // tanh(u) = 1 - 2/(1 + e^{2u}) on the hardware exp/rcp (saturates correctly at +-inf); ~1e-6 relative error
  float th = 1.0f - __fdividef(2.0f, 1.0f + __expf(2.0f * u));
  return 0.5f * x * (1.0f + th);
}

}  // namespace itts

// host-side helpers -------------------------------------------------------------------------------------------------
namespace itts {
void set_error(const char* fmt, ...);
int check_launch(const char* what);
}  // namespace itts

#define ITTS_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      itts::set_error(__VA_ARGS__);    \
      return ITTS_ERR_INVALID;         \
    }                                  \
  } while (0)
