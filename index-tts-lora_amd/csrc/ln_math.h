// LayerNorm of one fp32 row held one float4 per thread (threads with tid*4 >= D hold zeros and are inactive).
// Shared by ln_reduce_wide_kernel (its own launch) and by the reducer tail of the split-K skinny GEMM, so that the two
// forms of a decode step run the SAME instruction sequence on the same values: the tail form is bit-identical to the
// launch form by construction, not by luck of two compilations.
#pragma once
#include "common.h"

namespace itts {

// stat: LDS scratch, >= 2*2*16 floats, laid out [pass][mean|var][wave]
template <bool LN2>
__device__ __forceinline__ void wide_layernorm(f32x4& v, const f32x4& lw, const f32x4& lb, const f32x4& lw2, const f32x4& lb2,
                                               float* stat, int tid, int nw, int D, bool act) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int pass = 0; pass < (LN2 ? 2 : 1); ++pass) {
    float s = wave_sum(v[0] + v[1] + v[2] + v[3]);
    if (lane == 0) stat[(pass * 2 + 0) * 16 + wave] = s;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < nw; ++i) tot += stat[(pass * 2 + 0) * 16 + i];
    const float mean = tot / (float)D;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float d = v[e] - mean;
      q = fmaf(d, d, q);
    }
    q = wave_sum(act ? q : 0.f);
    if (lane == 0) stat[(pass * 2 + 1) * 16 + wave] = q;
    __syncthreads();
    float qt = 0.f;
    for (int i = 0; i < nw; ++i) qt += stat[(pass * 2 + 1) * 16 + i];
    const float rstd = rsqrtf(qt / (float)D + 1e-5f);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ww = pass == 0 ? lw[e] : lw2[e];
      float bb = pass == 0 ? lb[e] : lb2[e];
      v[e] = act ? (v[e] - mean) * rstd * ww + bb : 0.f;
    }
  }
}

template <typename T>
__device__ __forceinline__ void store_row4(T* dst, const f32x4& v) {
  if constexpr (sizeof(T) == 4) {
    st16(dst, v);
  } else {
    typedef T t4 __attribute__((ext_vector_type(4)));
    t4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f(v[e]);
    *reinterpret_cast<t4*>(dst) = o;
  }
}

}  // namespace itts
