// Bandwidth-bound kernels: anti-aliased SnakeBeta activation, LayerNorm, embedding step, tanh -> PCM16.
#include "common.h"

namespace itts {

struct Fir24 {
  float up[12];
  float down[12];
};

// -------------------------------------------------------------------------------------------------------------------
// Anti-aliased SnakeBeta, channels-last [B][T][C].
// Follows alias_free_torch/act.py:10-28: UpSample1d (resample.py:10-35) -> SnakeBeta (activations.py:63-122) ->
// DownSample1d/LowPassFilter1d (resample.py:38-48, filter.py:60-95); the fused form is the one of
// anti_alias_activation_cuda.cu:44-181, re-derived in polyphase form:
//   u[2q]   = 2 * sum_{d=-3..2} x[q+d] * up[5-2d]        u[2q+1] = 2 * sum_{d=-2..3} x[q+d] * up[6-2d]   (x index clamped)
//   s[m]    = u[m] + sin^2(u[m] * e^alpha) / (e^beta + 1e-9)
//   y[t]    = sum_{j<12} down[j] * s[clamp(2t + j - 5, 0, 2T-1)]
// One workgroup = TT output rows x CS channels of one batch element: x tile (TT+10 rows) and the 2TT+10 intermediate
// rows are staged in LDS as fp32, so every input element is read from HBM once (the CUDA original re-reads a 44-element
// window per thread from global memory).
// -------------------------------------------------------------------------------------------------------------------
constexpr int AA_TT = 32;
constexpr int AA_CS_MAX = 64;

template <typename T>
__global__ __launch_bounds__(256) void aa_snake_btc_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                            const float* __restrict__ alpha_log,
                                                            const float* __restrict__ beta_log, Fir24 f, int T_len, int C,
                                                            int CS) {
  __shared__ float xs[(AA_TT + 10) * AA_CS_MAX];
  __shared__ float ss[(2 * AA_TT + 10) * AA_CS_MAX];
  const int t0 = blockIdx.x * AA_TT;
  const int c0 = blockIdx.y * CS;
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  const T* xb = x + (int64_t)b * T_len * C;
  T* yb = y + (int64_t)b * T_len * C;
  const int cs = min(CS, C - c0);  // channels in this slab

  // phase 1: x tile, rows t0-5 .. t0+TT+4, replicate-clamped
  for (int idx = tid; idx < (AA_TT + 10) * cs; idx += 256) {
    int i = idx / cs, c = idx - i * cs;
    int row = min(max(t0 - 5 + i, 0), T_len - 1);
    xs[i * AA_CS_MAX + c] = Elem<T>::to_f(xb[(int64_t)row * C + c0 + c]);
  }
  __syncthreads();

  // phase 2: s[m] for m = 2*t0-5 .. 2*t0+2*TT+4
  for (int idx = tid; idx < (2 * AA_TT + 10) * cs; idx += 256) {
    int mi = idx / cs, c = idx - mi * cs;
    int m = min(max(2 * t0 - 5 + mi, 0), 2 * T_len - 1);
    int q = m >> 1;
    float u = 0.f;
    if (m & 1) {
#pragma unroll
      for (int d = -2; d <= 3; ++d) {
        int row = min(max(q + d, 0), T_len - 1) - (t0 - 5);
        u = fmaf(xs[row * AA_CS_MAX + c], f.up[6 - 2 * d], u);
      }
    } else {
#pragma unroll
      for (int d = -3; d <= 2; ++d) {
        int row = min(max(q + d, 0), T_len - 1) - (t0 - 5);
        u = fmaf(xs[row * AA_CS_MAX + c], f.up[5 - 2 * d], u);
      }
    }
    u *= 2.0f;
    float a = expf(alpha_log[c0 + c]);
    float ib = 1.0f / (expf(beta_log[c0 + c]) + 1e-9f);
    float sn = sinf(u * a);
    ss[mi * AA_CS_MAX + c] = u + ib * sn * sn;
  }
  __syncthreads();

  // phase 3: stride-2 low-pass
  for (int idx = tid; idx < AA_TT * cs; idx += 256) {
    int tt = idx / cs, c = idx - tt * cs;
    int t = t0 + tt;
    if (t >= T_len) break;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) acc = fmaf(f.down[j], ss[(2 * tt + j) * AA_CS_MAX + c], acc);
    yb[(int64_t)t * C + c0 + c] = Elem<T>::from_f(acc);
  }
}

// Reference-op layout [B][C][T] (drop-in for anti_alias_activation_cuda.forward): one thread per output sample.
template <typename T>
__global__ __launch_bounds__(256) void aa_snake_bct_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                            const float* __restrict__ alpha_log,
                                                            const float* __restrict__ beta_log, Fir24 f, int T_len, int C) {
  int64_t row = blockIdx.y;  // b*C + c
  int c = (int)(row % C);
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T_len) return;
  const T* xr = x + row * T_len;
  float a = expf(alpha_log[c]);
  float ib = 1.0f / (expf(beta_log[c]) + 1e-9f);
  float xw[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) xw[i] = Elem<T>::to_f(xr[min(max(t - 5 + i, 0), T_len - 1)]);
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    int m = min(max(2 * t + j - 5, 0), 2 * T_len - 1);
    int q = m >> 1;
    float u = 0.f;
    if (m & 1) {
#pragma unroll
      for (int d = -2; d <= 3; ++d) {
        int xi = min(max(q + d, 0), T_len - 1) - (t - 5);
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 11; ++i) v = (i == xi) ? xw[i] : v;
        u = fmaf(v, f.up[6 - 2 * d], u);
      }
    } else {
#pragma unroll
      for (int d = -3; d <= 2; ++d) {
        int xi = min(max(q + d, 0), T_len - 1) - (t - 5);
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 11; ++i) v = (i == xi) ? xw[i] : v;
        u = fmaf(v, f.up[5 - 2 * d], u);
      }
    }
    u *= 2.0f;
    float sn = sinf(u * a);
    acc = fmaf(f.down[j], u + ib * sn * sn, acc);
  }
  y[row * T_len + t] = Elem<T>::from_f(acc);
}

// -------------------------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, two-pass statistics in registers (D <= 4096, D % 4 == 0), optional second LayerNorm.
// -------------------------------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 16;  // float4 per lane

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                         const float* __restrict__ b, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, void* __restrict__ yv, int y_f32,
                                                         int M, int D) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* hr = h + (int64_t)row * D;
  int nv = D / 4;  // float4 count
  f32x4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int idx = lane + i * 64;
    if (idx < nv) {
      v[i] = ld16<f32x4>(hr + idx * 4);
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    } else {
      v[i] = f32x4{0, 0, 0, 0};
    }
  }
  float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int idx = lane + i * 64;
    if (idx < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float d = v[i][e] - mean;
        q = fmaf(d, d, q);
      }
    }
  }
  float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int idx = lane + i * 64;
    if (idx < nv) {
      f32x4 ww = ld16<f32x4>(w + idx * 4), bb = ld16<f32x4>(b + idx * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * rstd * ww[e] + bb[e];
    }
  }
  if (w2 != nullptr) {
    s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
      if (lane + i * 64 < nv) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    mean = wave_sum(s) / (float)D;
    q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
      if (lane + i * 64 < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float d = v[i][e] - mean;
          q = fmaf(d, d, q);
        }
      }
    rstd = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      int idx = lane + i * 64;
      if (idx < nv) {
        f32x4 ww = ld16<f32x4>(w2 + idx * 4), bb = ld16<f32x4>(b2 + idx * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * rstd * ww[e] + bb[e];
      }
    }
  }
  if (y_f32) {
    float* yr = (float*)yv + (int64_t)row * D;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      int idx = lane + i * 64;
      if (idx < nv) st16(yr + idx * 4, v[i]);
    }
  } else {
    T* yr = (T*)yv + (int64_t)row * D;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      int idx = lane + i * 64;
      if (idx < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) yr[idx * 4 + e] = Elem<T>::from_f(v[i][e]);
      }
    }
  }
}

__global__ __launch_bounds__(256) void embed_step_kernel(const int32_t* __restrict__ tokens, const float* __restrict__ table,
                                                          const float* __restrict__ pos_table,
                                                          const int32_t* __restrict__ step, int pos_add,
                                                          float* __restrict__ h, int D) {
  int b = blockIdx.x;
  int tok = tokens[b];
  int p = step[0] + pos_add;
  const float* e = table + (int64_t)tok * D;
  const float* pe = pos_table + (int64_t)p * D;
  for (int i = threadIdx.x; i < D; i += 256) h[(int64_t)b * D + i] = e[i] + pe[i];
}

template <typename T>
__global__ __launch_bounds__(256) void tanh_pcm_kernel(const T* __restrict__ x, float* __restrict__ wav,
                                                        int16_t* __restrict__ pcm, int64_t n, int apply_tanh) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    float v = Elem<T>::to_f(x[i]);
    if (apply_tanh) v = tanhf(v);
    if (wav) wav[i] = v;
    if (pcm) {
      float s = fminf(fmaxf(32767.0f * v, -32767.0f), 32767.0f);
      pcm[i] = (int16_t)s;  // truncation toward zero, as numpy astype(int16) (infer.py:911)
    }
  }
}

}  // namespace itts

using namespace itts;

extern "C" int itts_aa_snake_fwd(const void* x, void* y, const float* alpha_log, const float* beta_log,
                                 const float* up_filter12, const float* down_filter12, int B, int T, int C, int dtype,
                                 int layout, void* stream) {
  ITTS_REQUIRE(x && y && alpha_log && beta_log && up_filter12 && down_filter12, "itts_aa_snake_fwd: null pointer");
  ITTS_REQUIRE(B >= 0 && T >= 0 && C > 0, "itts_aa_snake_fwd: bad shape B=%d T=%d C=%d", B, T, C);
  if (B == 0 || T == 0) return ITTS_OK;
  Fir24 f;
  for (int i = 0; i < 12; ++i) {
    f.up[i] = up_filter12[i];
    f.down[i] = down_filter12[i];
  }
  hipStream_t s = (hipStream_t)stream;
  if (layout == 0) {
    int CS = C >= 64 ? 64 : C;
    if (C > 64 && C % 64 != 0) CS = (C % 48 == 0) ? 48 : ((C % 32 == 0) ? 32 : 64);
    dim3 grid((T + AA_TT - 1) / AA_TT, (C + CS - 1) / CS, B), block(256);
    ITTS_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "itts_aa_snake_fwd: grid too large");
    switch (dtype) {
      case ITTS_F32:
        hipLaunchKernelGGL(aa_snake_btc_kernel<float>, grid, block, 0, s, (const float*)x, (float*)y, alpha_log, beta_log, f, T, C, CS);
        break;
      case ITTS_BF16:
        hipLaunchKernelGGL(aa_snake_btc_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, (bf16_t*)y, alpha_log, beta_log, f, T, C, CS);
        break;
      case ITTS_F16:
        hipLaunchKernelGGL(aa_snake_btc_kernel<f16_t>, grid, block, 0, s, (const f16_t*)x, (f16_t*)y, alpha_log, beta_log, f, T, C, CS);
        break;
      default:
        ITTS_REQUIRE(false, "itts_aa_snake_fwd: unknown dtype %d", dtype);
    }
  } else {
    ITTS_REQUIRE((int64_t)B * C <= 65535, "itts_aa_snake_fwd: B*C too large for layout 1");
    dim3 grid((T + 255) / 256, B * C), block(256);
    switch (dtype) {
      case ITTS_F32:
        hipLaunchKernelGGL(aa_snake_bct_kernel<float>, grid, block, 0, s, (const float*)x, (float*)y, alpha_log, beta_log, f, T, C);
        break;
      case ITTS_BF16:
        hipLaunchKernelGGL(aa_snake_bct_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, (bf16_t*)y, alpha_log, beta_log, f, T, C);
        break;
      case ITTS_F16:
        hipLaunchKernelGGL(aa_snake_bct_kernel<f16_t>, grid, block, 0, s, (const f16_t*)x, (f16_t*)y, alpha_log, beta_log, f, T, C);
        break;
      default:
        ITTS_REQUIRE(false, "itts_aa_snake_fwd: unknown dtype %d", dtype);
    }
  }
  return check_launch("itts_aa_snake_fwd");
}

extern "C" int itts_layernorm(const float* h, const float* w, const float* b, const float* w2, const float* b2, void* y,
                              int y_f32, int M, int D, int dtype, void* stream) {
  ITTS_REQUIRE(h && w && b && y, "itts_layernorm: null pointer");
  ITTS_REQUIRE(D % 4 == 0 && D <= 4 * 64 * LN_MAXV && D > 0, "itts_layernorm: unsupported D=%d", D);
  if (M == 0) return ITTS_OK;
  dim3 grid((M + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case ITTS_F32:
      hipLaunchKernelGGL(layernorm_kernel<float>, grid, block, 0, s, h, w, b, w2, b2, y, y_f32, M, D);
      break;
    case ITTS_BF16:
      hipLaunchKernelGGL(layernorm_kernel<bf16_t>, grid, block, 0, s, h, w, b, w2, b2, y, y_f32, M, D);
      break;
    case ITTS_F16:
      hipLaunchKernelGGL(layernorm_kernel<f16_t>, grid, block, 0, s, h, w, b, w2, b2, y, y_f32, M, D);
      break;
    default:
      ITTS_REQUIRE(false, "itts_layernorm: unknown dtype %d", dtype);
  }
  return check_launch("itts_layernorm");
}

extern "C" int itts_embed_step(const int32_t* tokens, const float* table, const float* pos_table, const int32_t* step,
                               int pos_add, float* h, int B, int D, void* stream) {
  ITTS_REQUIRE(tokens && table && pos_table && step && h && B > 0 && D > 0, "itts_embed_step: bad arguments");
  hipLaunchKernelGGL(embed_step_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, tokens, table, pos_table, step, pos_add, h, D);
  return check_launch("itts_embed_step");
}

extern "C" int itts_tanh_pcm(const void* x, float* wav, int16_t* pcm, int64_t n, int dtype, int apply_tanh, void* stream) {
  ITTS_REQUIRE(x && n >= 0, "itts_tanh_pcm: bad arguments");
  if (n == 0) return ITTS_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  dim3 grid((unsigned)blocks), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case ITTS_F32:
      hipLaunchKernelGGL(tanh_pcm_kernel<float>, grid, block, 0, s, (const float*)x, wav, pcm, n, apply_tanh);
      break;
    case ITTS_BF16:
      hipLaunchKernelGGL(tanh_pcm_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, wav, pcm, n, apply_tanh);
      break;
    case ITTS_F16:
      hipLaunchKernelGGL(tanh_pcm_kernel<f16_t>, grid, block, 0, s, (const f16_t*)x, wav, pcm, n, apply_tanh);
      break;
    default:
      ITTS_REQUIRE(false, "itts_tanh_pcm: unknown dtype %d", dtype);
  }
  return check_launch("itts_tanh_pcm");
}
