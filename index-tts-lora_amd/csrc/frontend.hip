// Prompt front-end kernels: what the Conformer + Perceiver conditioner needs besides the GEMMs (which run on
// gemm_skinny / gemm_conv with the LayerNorms folded into their consumers).  One prompt is ~150 rows of 512 channels: every
// kernel here is a few microseconds of latency, not bandwidth -- the point is the launch COUNT (9 launches per Conformer
// block instead of ~45 library launches) and that no intermediate leaves the 16-bit packed operand layout the GEMMs read.
//   subsample_conv_kernel   Conv2d(1, C, 3, stride 2) + ReLU of Conv2dSubsampling2 -> the [T'][C*F'] operand of embed.out
//                           (indextts/gpt/conformer/subsampling.py:111-143)
//   mha_small_kernel        multi-head attention over a few hundred keys, optional relative-position term WITHOUT rel_shift:
//                           softmax(((q+u) k^T + (q+v) p^T) / sqrt(dk)) v   (conformer/attention.py RelPositionMultiHeadedAttention
//                           as driven by conformer_encoder.py:167-290; perceiver.py:271-312 without the position term)
//   glu_dwconv_ln_silu_kernel  GLU -> depthwise Conv1d(k) -> LayerNorm -> SiLU of the convolution module
//                           (conformer_encoder.py ConvolutionModule, :87-165)
//   rows_kernel             row-wise residual-stream operations: slab sum + bias, LayerNorm / l2-normalise, packed T copy
//   geglu_kernel            gelu(gate) * x of the Perceiver's feed-forward (perceiver.py:181-193)
#include "common.h"
#include <mutex>

namespace itts {

// ---------------------------------------------------------------------------------------------------------------
// Conv2d(1 -> C, 3 x 3, stride 2) + ReLU over mel [T][F] (time-major), written as the row-major T-typed operand
// y[t'][c * F2 + f'] of the following Linear(C * F2 -> d).  One workgroup per output row t'.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void subsample_conv_kernel(const float* __restrict__ mel, const float* __restrict__ w,
                                                             const float* __restrict__ b, T* __restrict__ y, int F, int C, int F2) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [3][F] mel rows | [C][9] weights | [C] bias
  float* mrow = sm;
  float* wl = sm + 3 * F;
  float* bl = wl + C * 9;
  const int t = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < 3 * F; i += 256) mrow[i] = mel[(int64_t)(2 * t) * F + i];
  for (int i = tid; i < C * 9; i += 256) wl[i] = w[i];
  for (int i = tid; i < C; i += 256) bl[i] = b[i];
  __syncthreads();
  const int n8 = C * F2 / 8;
  T* yr = y + (int64_t)t * C * F2;
  for (int it = tid; it < n8; it += 256) {
    typedef T t8 __attribute__((ext_vector_type(8)));
    t8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = it * 8 + e;
      const int c = idx / F2, f = idx - c * F2;
      const float* wc = wl + c * 9;
      float acc = bl[c];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc = fmaf(wc[i * 3 + j], mrow[i * F + 2 * f + j], acc);
      o[e] = Elem<T>::from_f(fmaxf(acc, 0.f));
    }
    *reinterpret_cast<t8*>(yr + it * 8) = o;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Small multi-head attention, head dim 64.  grid (ceil(Tq / 64), H), 4 waves, one 16-row query tile per wave; the keys are
// walked in chunks of KC rows staged in LDS (K, V and -- RELPOS -- the projected position table P of the head), 32 keys per
// step with an online softmax.  Everything runs transposed (keys / feature dims as MFMA rows, queries as columns): a lane owns
// ONE query row, so the running max / sum are lane scalars and the probabilities leave the score accumulators already in
// B-operand order for the second product (key slot (g, j) of a 32-key step = key 16 (j / 4) + 4 g + j % 4; the V^T fragments
// are fetched with the same slot order through the transposed LDS read).
// ---------------------------------------------------------------------------------------------------------------
struct MhaParams {
  int Tq, Tk, H;
  const void* q;
  const void* k;
  const void* v;
  int64_t qs, ks, vs;   // row strides (elements)
  const void* pos;      // [H][Tk][64] or NULL
  const float* bu;
  const float* bv;
  float scale;
  void* out;
  int out_mtp;
};

constexpr int MHA_KC = 192;   // keys per LDS chunk
constexpr int MHA_RS = 72;    // LDS row stride (elements): 144 bytes, an odd multiple of 16

typedef short mha_v4s __attribute__((__vector_size__(4 * sizeof(short))));

template <typename T, bool RELPOS>
__global__ __launch_bounds__(256) void mha_small_kernel(MhaParams p) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  extern __shared__ __attribute__((aligned(16))) char mha_lds[];
  T* Ki = reinterpret_cast<T*>(mha_lds);
  T* Vi = Ki + MHA_KC * MHA_RS;
  T* Pi = Vi + MHA_KC * MHA_RS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;
  const int h = blockIdx.y;
  const int q0 = ((int)blockIdx.x * 4 + wave) * 16;
  const bool active = q0 < p.Tq;

  // query fragments (B operand: column = query row r, k = feature dims 32 kk + 8 g .. + 7), with the two position biases added
  frag qu[2], qv[2];
  {
    const int row = min(q0 + r, p.Tq - 1);
    const T* qp = (const T*)p.q + (int64_t)row * p.qs + h * 64 + 8 * g;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const frag qf = ld16<frag>(qp + 32 * kk);
      if constexpr (RELPOS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float qe = EL::to_f(qf[e]);
          qu[kk][e] = EL::from_f(qe + p.bu[h * 64 + 32 * kk + 8 * g + e]);
          qv[kk][e] = EL::from_f(qe + p.bv[h * 64 + 32 * kk + 8 * g + e]);
        }
      } else {
        qu[kk] = qf;
        qv[kk] = qf;
      }
    }
  }
  f32x4 o[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  for (int c0 = 0; c0 < p.Tk; c0 += MHA_KC) {
    const int rows = min(MHA_KC, p.Tk - c0);
    const int rows32 = (rows + 31) & ~31;
    if (c0 > 0) __syncthreads();   // every wave is done with the previous chunk
    for (int idx = tid; idx < rows32 * 8; idx += 256) {
      const int row = idx >> 3, seg = idx & 7;
      const bool ok = row < rows;
      const int64_t kr = c0 + row;
      const frag z = zero_frag<frag>();
      st16(Ki + row * MHA_RS + seg * 8, ok ? ld16<frag>((const T*)p.k + kr * p.ks + h * 64 + seg * 8) : z);
      st16(Vi + row * MHA_RS + seg * 8, ok ? ld16<frag>((const T*)p.v + kr * p.vs + h * 64 + seg * 8) : z);
      if constexpr (RELPOS)
        st16(Pi + row * MHA_RS + seg * 8, ok ? ld16<frag>((const T*)p.pos + ((int64_t)h * p.Tk + kr) * 64 + seg * 8) : z);
    }
    __syncthreads();
    if (!active) continue;
    for (int kb = 0; kb < rows32; kb += 32) {
      f32x4 s[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const T* kr = Ki + (kb + 16 * j + r) * MHA_RS + 8 * g;
        s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        s[j] = EL::mma(ld16<frag>(kr), qu[0], s[j]);
        s[j] = EL::mma(ld16<frag>(kr + 32), qu[1], s[j]);
        if constexpr (RELPOS) {
          const T* pr = Pi + (kb + 16 * j + r) * MHA_RS + 8 * g;
          s[j] = EL::mma(ld16<frag>(pr), qv[0], s[j]);
          s[j] = EL::mma(ld16<frag>(pr + 32), qv[1], s[j]);
        }
      }
      // lane (g, r): s[j][e] = score of query r against key c0 + kb + 16 j + 4 g + e
      float mloc = -INFINITY;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = c0 + kb + 16 * j + 4 * g + e;
          s[j][e] = key < p.Tk ? s[j][e] * p.scale : -INFINITY;
          mloc = fmaxf(mloc, s[j][e]);
        }
      mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float m_new = fmaxf(m_run, mloc);       // finite: the first key of a step always exists
      const float alpha = __expf(m_run - m_new);
      float lsum = 0.f;
      frag pf;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = __expf(s[j][e] - m_new);
          const T pt = EL::from_f(pe);
          pf[4 * j + e] = pt;
          lsum += EL::to_f(pt);                      // the normaliser sums what the product multiplies
        }
      lsum += __shfl_xor(lsum, 16, 64);
      lsum += __shfl_xor(lsum, 32, 64);
      l_run = l_run * alpha + lsum;
      m_run = m_new;
      // V^T fragments: lane (g, r) needs feature dim 16 nt + r of keys kb + 4 g + {0..3} and kb + 16 + 4 g + {0..3}
      const int qq = r >> 2, pq = r & 3;
      typedef __attribute__((address_space(3))) mha_v4s* lptr;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const T* a = Vi + (kb + 4 * g + qq) * MHA_RS + 16 * nt + 4 * pq;
        const mha_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a));
        const mha_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a + 16 * MHA_RS));
        typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
        const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        o[nt] = o[nt] * alpha;
        o[nt] = EL::mma(__builtin_bit_cast(frag, both), pf, o[nt]);
      }
    }
  }
  if (!active || q0 + r >= p.Tq) return;
  const float inv = 1.0f / l_run;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    typedef T t4 __attribute__((ext_vector_type(4)));
    t4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = EL::from_f(o[nt][e] * inv);
    *reinterpret_cast<t4*>((T*)p.out + pa_off<T>(q0 + r, h * 64 + 16 * nt + 4 * g, p.out_mtp)) = ov;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Convolution module between the two pointwise convolutions: x [T][2C] (value | gate) -> GLU -> depthwise conv over time
// (zero "same" padding) -> LayerNorm over channels -> SiLU -> packed T operand [T][C] of pointwise_conv2.
// One workgroup per output row, C / 2 threads, two channels per thread.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red, int nw) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < nw; ++w) s += red[w];
  return s;
}

template <typename T, int KT>
__global__ __launch_bounds__(1024) void glu_dwconv_ln_silu_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ b, const float* __restrict__ lw,
                                                                  const float* __restrict__ lb, T* __restrict__ y, int Tn, int C,
                                                                  int mtp, float eps) {
  __shared__ float red[16];
  const int t = blockIdx.x, c = 2 * threadIdx.x;
  typedef T t2 __attribute__((ext_vector_type(2)));
  t2 av[KT], gv[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) {
    const int row = t + j - (KT - 1) / 2;
    const bool ok = row >= 0 && row < Tn;
    const T* xr = x + (int64_t)(ok ? row : 0) * 2 * C;
    av[j] = *reinterpret_cast<const t2*>(xr + c);
    gv[j] = *reinterpret_cast<const t2*>(xr + C + c);
    if (!ok) av[j] = t2{Elem<T>::from_f(0.f), Elem<T>::from_f(0.f)};
  }
  float acc[2] = {b[c], b[c + 1]};
#pragma unroll
  for (int j = 0; j < KT; ++j)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float gl = Elem<T>::to_f(av[j][e]) / (1.f + __expf(-Elem<T>::to_f(gv[j][e])));
      acc[e] = fmaf(gl, w[(c + e) * KT + j], acc[e]);
    }
  const int nw = (blockDim.x + 63) >> 6;
  const float mean = block_sum(acc[0] + acc[1], red, nw) / (float)C;
  const float d0 = acc[0] - mean, d1 = acc[1] - mean;
  const float var = block_sum(d0 * d0 + d1 * d1, red, nw) / (float)C;
  const float rstd = rsqrtf(var + eps);
  t2 o;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float v = (e == 0 ? d0 : d1) * rstd * lw[c + e] + lb[c + e];
    o[e] = Elem<T>::from_f(v / (1.f + __expf(-v)));
  }
  *reinterpret_cast<t2*>(y + pa_off<T>(t, c, mtp)) = o;
}

// ---------------------------------------------------------------------------------------------------------------
// Row operations on the fp32 residual stream: v = (x ? x[m] : 0) + bias + sum_s slab[s][m]; norm 1: LayerNorm(w, b);
// norm 2: v / max(|v|_2, 1e-12) * sqrt(D) * w (the Perceiver's RMSNorm).  Result to y (fp32, may alias x) and / or as a
// T-typed packed copy.  One workgroup per row, 4 consecutive columns per thread and pass.
// ---------------------------------------------------------------------------------------------------------------
struct RowsParams {
  int M, D;
  const float* x;
  const float* slab;
  int nslab;
  const float* bias;
  int norm;
  const float* w;
  const float* b;
  float eps;
  float* y;
  void* yp;
  int y_row0, y_mtp;
};

template <typename T>
__global__ __launch_bounds__(256) void rows_kernel(RowsParams p) {
  __shared__ float red[16];
  const int m = blockIdx.x, tid = threadIdx.x;
  constexpr int NP = 2;                          // D <= 2048
  f32x4 v[NP];
  const int D4 = p.D / 4;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c4 = tid + i * 256;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c4 < D4) {
      if (p.x != nullptr) v[i] = ld16<f32x4>(p.x + (int64_t)m * p.D + c4 * 4);
      if (p.bias != nullptr) v[i] += ld16<f32x4>(p.bias + c4 * 4);
      for (int s = 0; s < p.nslab; ++s) v[i] += ld16<f32x4>(p.slab + ((int64_t)s * p.M + m) * p.D + c4 * 4);
    }
  }
  if (p.norm != 0) {
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) s1 += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    const float mean = p.norm == 1 ? block_sum(s1, red, 4) / (float)p.D : 0.f;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (tid + i * 256 < D4)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[i][e] - mean;
          s2 = fmaf(d, d, s2);
        }
    s2 = block_sum(s2, red, 4);
    const float k = p.norm == 1 ? rsqrtf(s2 / (float)p.D + p.eps) : sqrtf((float)p.D) / fmaxf(sqrtf(s2), 1e-12f);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c4 = tid + i * 256;
      if (c4 < D4) {
        const f32x4 wv = ld16<f32x4>(p.w + c4 * 4);
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.norm == 1) bv = ld16<f32x4>(p.b + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * k * wv[e] + bv[e];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c4 = tid + i * 256;
    if (c4 < D4) {
      if (p.y != nullptr) st16(p.y + (int64_t)m * p.D + c4 * 4, v[i]);
      if (p.yp != nullptr) {
        typedef T t4 __attribute__((ext_vector_type(4)));
        t4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f(v[i][e]);
        *reinterpret_cast<t4*>((T*)p.yp + pa_off<T>(p.y_row0 + m, c4 * 4, p.y_mtp)) = o;
      }
    }
  }
}

// gelu(gate) * x (erf form, torch's default), h [M][2 Kp] (x | gate) -> packed T [M][Kp]
template <typename T>
__global__ __launch_bounds__(256) void geglu_kernel(const T* __restrict__ h, T* __restrict__ y, int M, int Kp, int mtp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int K4 = Kp / 4;
  if (i >= (int64_t)M * K4) return;
  const int m = (int)(i / K4), c = (int)(i - (int64_t)m * K4) * 4;
  typedef T t4 __attribute__((ext_vector_type(4)));
  const t4 xv = *reinterpret_cast<const t4*>(h + (int64_t)m * 2 * Kp + c);
  const t4 gv = *reinterpret_cast<const t4*>(h + (int64_t)m * 2 * Kp + Kp + c);
  t4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float gt = Elem<T>::to_f(gv[e]);
    o[e] = Elem<T>::from_f(0.5f * gt * (1.f + erff(gt * 0.70710678118654752f)) * Elem<T>::to_f(xv[e]));
  }
  *reinterpret_cast<t4*>(y + pa_off<T>(m, c, mtp)) = o;
}

}  // namespace itts

using namespace itts;

extern "C" int itts_subsample_conv(const float* mel, const float* w, const float* b, void* y, int T, int F, int C, int dtype,
                                   void* stream) {
  ITTS_REQUIRE(mel && w && b && y && T >= 3 && F >= 3 && C > 0, "itts_subsample_conv: bad arguments");
  ITTS_REQUIRE(dtype == ITTS_BF16 || dtype == ITTS_F16, "itts_subsample_conv: the front-end kernels are built for bf16 / f16");
  const int T2 = (T - 3) / 2 + 1, F2 = (F - 3) / 2 + 1;
  ITTS_REQUIRE((C * F2) % 8 == 0, "itts_subsample_conv: C * F' = %d must be a multiple of 8", C * F2);
  const size_t lds = (size_t)(3 * F + C * 10) * 4;
  ITTS_REQUIRE(lds <= 64 * 1024, "itts_subsample_conv: C = %d, F = %d do not fit the staging buffer", C, F);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == ITTS_BF16)
    hipLaunchKernelGGL(subsample_conv_kernel<bf16_t>, dim3(T2), dim3(256), lds, s, mel, w, b, (bf16_t*)y, F, C, F2);
  else
    hipLaunchKernelGGL(subsample_conv_kernel<f16_t>, dim3(T2), dim3(256), lds, s, mel, w, b, (f16_t*)y, F, C, F2);
  return check_launch("itts_subsample_conv");
}

extern "C" int itts_mha_small(const itts_mha_args* a, void* stream) {
  ITTS_REQUIRE(a && a->q && a->k && a->v && a->out, "itts_mha_small: null args");
  ITTS_REQUIRE(a->dtype == ITTS_BF16 || a->dtype == ITTS_F16, "itts_mha_small: built for bf16 / f16");
  ITTS_REQUIRE(a->Tq > 0 && a->Tk > 0 && a->H > 0 && a->H <= 65535, "itts_mha_small: bad shape Tq=%d Tk=%d H=%d", a->Tq, a->Tk, a->H);
  ITTS_REQUIRE(a->q_stride % 8 == 0 && a->k_stride % 8 == 0 && a->v_stride % 8 == 0 && a->q_stride >= a->H * 64 &&
                   a->k_stride >= a->H * 64 && a->v_stride >= a->H * 64,
               "itts_mha_small: row strides must be multiples of 8 elements and cover H * 64");
  ITTS_REQUIRE(a->out_mtp * 16 >= a->Tq, "itts_mha_small: out_mtp = %d row tiles do not cover Tq = %d", a->out_mtp, a->Tq);
  const bool rel = a->pos != nullptr;
  if (rel) ITTS_REQUIRE(a->bias_u && a->bias_v, "itts_mha_small: the relative-position form needs bias_u and bias_v");
  MhaParams p;
  p.Tq = a->Tq; p.Tk = a->Tk; p.H = a->H;
  p.q = a->q; p.k = a->k; p.v = a->v;
  p.qs = a->q_stride; p.ks = a->k_stride; p.vs = a->v_stride;
  p.pos = a->pos; p.bu = a->bias_u; p.bv = a->bias_v;
  p.scale = a->scale;
  p.out = a->out; p.out_mtp = a->out_mtp;
  const dim3 grid((a->Tq + 63) / 64, a->H), block(256);
  const size_t lds = (size_t)(rel ? 3 : 2) * MHA_KC * MHA_RS * 2;
  hipStream_t s = (hipStream_t)stream;
#define ITTS_MHA(T_, R_)                                                                                                 \
  do {                                                                                                                   \
    static std::once_flag attr_;   /* one-shot per instantiation, safe under concurrent first calls */                   \
    std::call_once(attr_, [] {                                                                                           \
      (void)hipFuncSetAttribute((const void*)mha_small_kernel<T_, R_>, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                3 * MHA_KC * MHA_RS * 2);                                                                \
    });                                                                                                                  \
    hipLaunchKernelGGL((mha_small_kernel<T_, R_>), grid, block, lds, s, p);                                              \
  } while (0)
  if (a->dtype == ITTS_BF16) {
    if (rel) ITTS_MHA(bf16_t, true);
    else ITTS_MHA(bf16_t, false);
  } else {
    if (rel) ITTS_MHA(f16_t, true);
    else ITTS_MHA(f16_t, false);
  }
#undef ITTS_MHA
  return check_launch("itts_mha_small");
}

extern "C" int itts_glu_dwconv_ln_silu(const void* x, const float* w, const float* b, const float* ln_w, const float* ln_b, void* y,
                                       int T, int C, int taps, int y_mtp, float eps, int dtype, void* stream) {
  ITTS_REQUIRE(x && w && b && ln_w && ln_b && y && T > 0, "itts_glu_dwconv_ln_silu: bad arguments");
  ITTS_REQUIRE(dtype == ITTS_BF16 || dtype == ITTS_F16, "itts_glu_dwconv_ln_silu: built for bf16 / f16");
  ITTS_REQUIRE(C % 128 == 0 && C <= 2048, "itts_glu_dwconv_ln_silu: C = %d must be a multiple of 128, at most 2048", C);
  ITTS_REQUIRE(taps == 15 || taps == 7 || taps == 31, "itts_glu_dwconv_ln_silu: taps = %d (7, 15 and 31 are built)", taps);
  ITTS_REQUIRE(y_mtp * 16 >= T, "itts_glu_dwconv_ln_silu: y_mtp = %d row tiles do not cover T = %d", y_mtp, T);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(T), block(C / 2);
  if (eps <= 0.f) eps = 1e-5f;
#define ITTS_DW(T_, K_)                                                                                                      \
  hipLaunchKernelGGL((glu_dwconv_ln_silu_kernel<T_, K_>), grid, block, 0, s, (const T_*)x, w, b, ln_w, ln_b, (T_*)y, T, C, y_mtp, eps)
  if (dtype == ITTS_BF16) {
    if (taps == 15) ITTS_DW(bf16_t, 15);
    else if (taps == 7) ITTS_DW(bf16_t, 7);
    else ITTS_DW(bf16_t, 31);
  } else {
    if (taps == 15) ITTS_DW(f16_t, 15);
    else if (taps == 7) ITTS_DW(f16_t, 7);
    else ITTS_DW(f16_t, 31);
  }
#undef ITTS_DW
  return check_launch("itts_glu_dwconv_ln_silu");
}

extern "C" int itts_rows(const itts_rows_args* a, void* stream) {
  ITTS_REQUIRE(a && a->M > 0 && a->D > 0 && a->D % 4 == 0 && a->D <= 2048, "itts_rows: bad shape (D %% 4 == 0, D <= 2048)");
  ITTS_REQUIRE(a->x != nullptr || a->nslab > 0, "itts_rows: neither x nor slabs");
  ITTS_REQUIRE(a->nslab >= 0 && (a->nslab == 0 || a->slab != nullptr), "itts_rows: slab is null");
  ITTS_REQUIRE(a->norm >= 0 && a->norm <= 2 && (a->norm == 0 || a->w != nullptr) && (a->norm != 1 || a->b != nullptr),
               "itts_rows: norm %d needs its weights", a->norm);
  ITTS_REQUIRE(a->y != nullptr || a->y_packed != nullptr, "itts_rows: no output");
  const int mtp = a->y_mtp > 0 ? a->y_mtp : (a->M + 15) / 16;
  if (a->y_packed != nullptr) {
    ITTS_REQUIRE(a->dtype == ITTS_BF16 || a->dtype == ITTS_F16, "itts_rows: the packed copy is bf16 / f16");
    ITTS_REQUIRE(a->D % 32 == 0 && a->y_row0 >= 0 && (a->y_row0 + a->M) <= mtp * 16, "itts_rows: packed copy: D %% 32, rows inside y_mtp tiles");
  }
  RowsParams p;
  p.M = a->M; p.D = a->D;
  p.x = a->x; p.slab = a->slab; p.nslab = a->nslab; p.bias = a->bias;
  p.norm = a->norm; p.w = a->w; p.b = a->b; p.eps = a->eps > 0.f ? a->eps : 1e-5f;
  p.y = a->y; p.yp = a->y_packed; p.y_row0 = a->y_row0; p.y_mtp = mtp;
  hipStream_t s = (hipStream_t)stream;
  if (a->dtype == ITTS_BF16) hipLaunchKernelGGL(rows_kernel<bf16_t>, dim3(a->M), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(rows_kernel<f16_t>, dim3(a->M), dim3(256), 0, s, p);
  return check_launch("itts_rows");
}

extern "C" int itts_geglu(const void* h, void* y, int M, int Kp, int y_mtp, int dtype, void* stream) {
  ITTS_REQUIRE(h && y && M > 0 && Kp > 0 && Kp % 32 == 0, "itts_geglu: bad arguments (Kp %% 32 == 0)");
  ITTS_REQUIRE(dtype == ITTS_BF16 || dtype == ITTS_F16, "itts_geglu: built for bf16 / f16");
  const int mtp = y_mtp > 0 ? y_mtp : (M + 15) / 16;
  ITTS_REQUIRE(mtp * 16 >= M, "itts_geglu: y_mtp does not cover M");
  const int64_t n = (int64_t)M * (Kp / 4);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (dtype == ITTS_BF16) hipLaunchKernelGGL(geglu_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)h, (bf16_t*)y, M, Kp, mtp);
  else hipLaunchKernelGGL(geglu_kernel<f16_t>, grid, block, 0, s, (const f16_t*)h, (f16_t*)y, M, Kp, mtp);
  return check_launch("itts_geglu");
}
