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

namespace itts {

// ---------------------------------------------------------------------------------------------------------------
// Conv2d(1 -> C, 3 x 3, stride 2) + ReLU over mel [T][F] (time-major), written as the row-major T-typed operand
// y[t'][c * F2 + f'] of the following Linear(C * F2 -> d).  One workgroup per output row t'; a LANE is one output frequency f'
// and keeps its 3 x 3 input window in registers, a wave walks its share of the channels with the nine weights and the bias as
// wave-uniform operands: nine FMAs per output; the wave's run of the row leaves through LDS as 16-byte vectors.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, bool STAGED>
__global__ __launch_bounds__(512) void subsample_conv_kernel(const float* __restrict__ mel, const float* __restrict__ w,
                                                             const float* __restrict__ b, T* __restrict__ y, int F, int C, int F2) {
  extern __shared__ __attribute__((aligned(16))) char sub_lds[];   // STAGED: the wave's run of the output row, written out as 16-byte vectors
  const int t = blockIdx.x, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NW = blockDim.x >> 6;
  const int cpw = (C + NW - 1) / NW;                 // channels per wave, a contiguous run
  const int c_begin = wave * cpw, c_end = min(C, c_begin + cpw);
  T* yr = y + (int64_t)t * C * F2;
  T* stage = reinterpret_cast<T*>(sub_lds) + (size_t)wave * cpw * F2;
  for (int f0 = 0; f0 < F2; f0 += 64) {
    const int f = f0 + lane;
    const bool ok = f < F2;
    float m[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) m[i * 3 + j] = ok ? mel[(int64_t)(2 * t + i) * F + 2 * f + j] : 0.f;
    // the nine weights and the bias of channel cg + lane sit in this lane's registers; channel by channel they are broadcast
    // with v_readlane (scalar loads per channel would serialise on their round trips)
    for (int cg = c_begin; cg < c_end; cg += 64) {
      const int mc = min(cg + lane, c_end - 1);
      float wr[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) wr[k] = w[mc * 9 + k];
      const float br = b[mc];
      const int nc = min(64, c_end - cg);
      for (int ci = 0; ci < nc; ++ci) {
        float acc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, br), ci));
#pragma unroll
        for (int k = 0; k < 9; ++k)
          acc = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wr[k]), ci)), m[k], acc);
        const T o = Elem<T>::from_f(fmaxf(acc, 0.f));
        if (ok) {
          if constexpr (STAGED) stage[(cg + ci - c_begin) * F2 + f] = o;
          else yr[(cg + ci) * F2 + f] = o;
        }
      }
    }
  }
  if constexpr (STAGED) {
    // one wave wrote this run and reads it back: the LDS pipe keeps a wave's accesses in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int n16 = (c_end - c_begin) * F2 / 8;
    T* dst = yr + (int64_t)c_begin * F2;
    for (int i = lane; i < n16; i += 64) st16(dst + i * 8, ld16<B16>(stage + i * 8));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Small multi-head attention, head dim 64.  grid (ceil(Tq / 16), H): one 16-row query tile per workgroup, whose 4 waves SPLIT
// THE KEYS (32-key steps dealt round-robin) and merge their partial softmax states through LDS at the end -- a prompt is ~150
// keys, so the key walk is the only loop long enough to cut, and 10 x 8 tiles x 4 waves fill the chip where 24 workgroups did not.
// Everything runs transposed (keys / feature dims as MFMA rows, queries as columns): a lane owns ONE query row, so the running
// max / sum are lane scalars and the probabilities leave the score accumulators already in B-operand order for the second
// product (key slot (g, j) of a 32-key step = key 16 (j / 4) + 4 g + j % 4).  K and the projected position table P are read
// straight from global memory in A-operand order; V goes through a wave-private LDS image and comes back transposed
// (ds_read_b64_tr_b16) with the same slot order.
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

constexpr int MHA_RS = 72;    // LDS row stride of the V image (elements): 144 bytes, an odd multiple of 16

typedef short mha_v4s __attribute__((__vector_size__(4 * sizeof(short))));

template <typename T, bool RELPOS>
__global__ __launch_bounds__(256) void mha_small_kernel(MhaParams p) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  __shared__ __attribute__((aligned(16))) T Vimg[4][32 * MHA_RS];
  __shared__ __attribute__((aligned(16))) float comb[4][4][64][4];
  __shared__ float ml[4][16][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;
  const int h = blockIdx.y;
  const int q0 = (int)blockIdx.x * 16;
  T* Vi = Vimg[wave];

  // query fragments (B operand: column = query row r, k = feature dims 32 kk + 8 g .. + 7), with the two position biases added
  frag qu[2], qv[2];
  {
    const int row = min(q0 + r, p.Tq - 1);
    const T* qp = (const T*)p.q + (int64_t)row * p.qs + h * 64 + 8 * g;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const frag qf = ld16<frag>(qp + 32 * kk);
      if constexpr (RELPOS) {
        const f32x4 u0 = ld16<f32x4>(p.bu + h * 64 + 32 * kk + 8 * g), u1 = ld16<f32x4>(p.bu + h * 64 + 32 * kk + 8 * g + 4);
        const f32x4 v0 = ld16<f32x4>(p.bv + h * 64 + 32 * kk + 8 * g), v1 = ld16<f32x4>(p.bv + h * 64 + 32 * kk + 8 * g + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float qe = EL::to_f(qf[e]);
          qu[kk][e] = EL::from_f(qe + (e < 4 ? u0[e & 3] : u1[e & 3]));
          qv[kk][e] = EL::from_f(qe + (e < 4 ? v0[e & 3] : v1[e & 3]));
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

  for (int kb = wave * 32; kb < p.Tk; kb += 128) {
    // ---- operand requests of this 32-key step
    frag kf[2][2], pf_[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t key = min(kb + 16 * j + r, p.Tk - 1);        // rows past the end are masked below
      const T* kr = (const T*)p.k + key * p.ks + h * 64 + 8 * g;
      kf[j][0] = ld16<frag>(kr);
      kf[j][1] = ld16<frag>(kr + 32);
      if constexpr (RELPOS) {
        const T* pr = (const T*)p.pos + ((int64_t)h * p.Tk + key) * 64 + 8 * g;
        pf_[j][0] = ld16<frag>(pr);
        pf_[j][1] = ld16<frag>(pr + 32);
      }
    }
    frag vrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = lane + 64 * i, row = idx >> 3, seg = idx & 7;
      vrow[i] = kb + row < p.Tk ? ld16<frag>((const T*)p.v + (int64_t)(kb + row) * p.vs + h * 64 + seg * 8) : zero_frag<frag>();
    }
    // ---- scores (transposed): lane (g, r): s[j][e] = query r against key kb + 16 j + 4 g + e
    f32x4 s[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      s[j] = EL::mma(kf[j][0], qu[0], s[j]);
      s[j] = EL::mma(kf[j][1], qu[1], s[j]);
      if constexpr (RELPOS) {
        s[j] = EL::mma(pf_[j][0], qv[0], s[j]);
        s[j] = EL::mma(pf_[j][1], qv[1], s[j]);
      }
    }
    float mloc = -INFINITY;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = kb + 16 * j + 4 * g + e;
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
        const T pt = EL::from_f(__expf(s[j][e] - m_new));
        pf[4 * j + e] = pt;
        lsum += EL::to_f(pt);                      // the normaliser sums what the product multiplies
      }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    l_run = l_run * alpha + lsum;
    m_run = m_new;
    // ---- V rows into the wave's image, back out transposed: lane (g, r) gets feature dim 16 nt + r of keys kb + 4 g + {0..3}
    // and kb + 16 + 4 g + {0..3}.  One wave writes and reads the image: the LDS pipe keeps a wave's accesses in order.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = lane + 64 * i;
      st16(Vi + (idx >> 3) * MHA_RS + (idx & 7) * 8, vrow[i]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int qq = r >> 2, pq = r & 3;
    typedef __attribute__((address_space(3))) mha_v4s* lptr;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const T* a = Vi + (4 * g + qq) * MHA_RS + 16 * nt + 4 * pq;
      const mha_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a));
      const mha_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a + 16 * MHA_RS));
      typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
      const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      o[nt] = o[nt] * alpha;
      o[nt] = EL::mma(__builtin_bit_cast(frag, both), pf, o[nt]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // ---- merge the four partial states: wave w finishes feature tile nt = w
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) st16(&comb[wave][nt][lane][0], o[nt]);
  if (g == 0) {
    ml[wave][r][0] = m_run;
    ml[wave][r][1] = l_run;
  }
  __syncthreads();
  float M = -INFINITY;
#pragma unroll
  for (int w = 0; w < 4; ++w) M = fmaxf(M, ml[w][r][0]);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float L = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float sc = __expf(ml[w][r][0] - M);     // a wave without keys: exp(-inf) = 0
    L = fmaf(sc, ml[w][r][1], L);
    acc += ld16<f32x4>(&comb[w][wave][lane][0]) * sc;
  }
  if (q0 + r >= p.Tq) return;
  const float inv = 1.0f / L;
  typedef T t4 __attribute__((ext_vector_type(4)));
  t4 ov;
#pragma unroll
  for (int e = 0; e < 4; ++e) ov[e] = EL::from_f(acc[e] * inv);
  *reinterpret_cast<t4*>((T*)p.out + pa_off<T>(q0 + r, h * 64 + 16 * wave + 4 * g, p.out_mtp)) = ov;
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
      for (int s0 = 0; s0 < p.nslab; s0 += 8) {     // eight requests in flight, summed in slab order
        f32x4 tv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          tv[j] = s0 + j < p.nslab ? ld16<f32x4>(p.slab + ((int64_t)(s0 + j) * p.M + m) * p.D + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i] += tv[j];
      }
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

// ---------------------------------------------------------------------------------------------------------------
// The GPT prompt rows of UnifiedVoice.prepare_gpt_inputs (indextts/gpt/model.py:606-667) in one launch: per batch row, the
// text ids without start / stop ids become  start | ids | stop, are embedded (token + position tables), follow the C
// conditioning latents and are pushed to the RIGHT end of the P = C + L + 2 positions (left padding = L - n zero rows);
// emb fp32 [B][P][D], mask int64 [B][P + 1] (0 on the padding, 1 elsewhere and on the start-mel slot), pad int32 [B].
// grid (ceil(P / 8), B): every workgroup compacts its row's ids again (a block scan over <= L ids) and writes 8 positions.
// ---------------------------------------------------------------------------------------------------------------
constexpr int PREFIX_MAXL = 2046;
__global__ __launch_bounds__(256) void prefix_rows_kernel(const int64_t* __restrict__ text, const float* __restrict__ conds, int Bc,
                                                          const float* __restrict__ temb, const float* __restrict__ tpos,
                                                          float* __restrict__ emb, int64_t* __restrict__ mask,
                                                          int32_t* __restrict__ pad_out, int L, int C, int D, int start_tok,
                                                          int stop_tok, int n_tok, int n_pos) {
  __shared__ int ctok[PREFIX_MAXL + 2];
  __shared__ int wsum[4];
  __shared__ int base_s;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t* tr = text + (int64_t)b * L;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int l0 = 0; l0 < L; l0 += 256) {           // stable compaction of the ids that are neither start nor stop
    const int l = l0 + tid;
    const int64_t tk = l < L ? tr[l] : (int64_t)stop_tok;
    const bool ok = l < L && tk != stop_tok && tk != start_tok;
    const unsigned long long bal = __ballot(ok);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (ok) ctok[1 + off + before] = (int)tk;
    __syncthreads();
    if (tid == 0) base_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  const int n = base_s;
  if (tid == 0) {
    ctok[0] = start_tok;
    ctok[n + 1] = stop_tok;
  }
  __syncthreads();
  const int P = C + L + 2, pad = L - n;
  if (blockIdx.x == 0) {
    for (int i = tid; i <= P; i += 256) mask[(int64_t)b * (P + 1) + i] = (i >= pad) ? 1 : 0;
    if (tid == 0) pad_out[b] = pad;
  }
  const int D4 = D / 4;
  for (int pi = 0; pi < 8; ++pi) {
    const int pos = (int)blockIdx.x * 8 + pi;
    if (pos >= P) break;
    const int src = pos - pad;
    float* dst = emb + ((int64_t)b * P + pos) * D;
    if (src < 0) {
      for (int i = tid; i < D4; i += 256) st16(dst + i * 4, f32x4{0.f, 0.f, 0.f, 0.f});
    } else if (src < C) {
      const float* cr = conds + ((int64_t)(Bc == 1 ? 0 : b) * C + src) * D;
      for (int i = tid; i < D4; i += 256) st16(dst + i * 4, ld16<f32x4>(cr + i * 4));
    } else {
      const int j = src - C;                        // 0 .. n + 1
      const int tk = min(max(ctok[j], 0), n_tok - 1);
      const float* er = temb + (int64_t)tk * D;
      const float* pr = tpos + (int64_t)min(j, n_pos - 1) * D;
      for (int i = tid; i < D4; i += 256) st16(dst + i * 4, ld16<f32x4>(er + i * 4) + ld16<f32x4>(pr + i * 4));
    }
  }
}

// ===============================================================================================================
// Speaker encoder (ECAPA-TDNN, indextts/BigVGAN/ECAPA_TDNN.py:470-581): what it needs besides the 1 x 1 convolutions,
// which run on gemm_skinny with the ReLU + affine epilogues.  Activations are T-typed in the packed operand layout
// ([frames][channels]; a run of 32-channel k-steps of a packed operand is itself a packed operand, so the three
// SE-Res2Net outputs are written straight into the [frames][1536] operand of the MFA convolution).
// ===============================================================================================================
__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// x fp32 [T][F] -> packed T [T][Kp]: column j * F + f = x[reflect(t + (j - (taps-1)/2) * dil)][f], zero past taps * F: the
// operand of the first TDNN block's k = 5 convolution (reflect "same" padding, nnet/CNN.py:430-488) as a plain GEMM.
template <typename T>
__global__ __launch_bounds__(256) void im2col_reflect_kernel(const float* __restrict__ x, T* __restrict__ y, int Tn, int F, int taps,
                                                             int dil, int Kp, int mtp) {
  const int t = blockIdx.x;
  for (int k8 = threadIdx.x; k8 < Kp / 8; k8 += 256) {
    typedef T t8 __attribute__((ext_vector_type(8)));
    t8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k8 * 8 + e, j = k / F, f = k - j * F;
      o[e] = Elem<T>::from_f(j < taps ? x[(int64_t)reflect_idx(t + (j - (taps - 1) / 2) * dil, Tn) * F + f] : 0.f);
    }
    *reinterpret_cast<t8*>(y + pa_off<T>(t, k8 * 8, mtp)) = o;
  }
}

// One step of a Res2Net block: chunk s of the block's first TDNN output (64 channels), plus the previous step's output,
// through a k = 3 dilated convolution (reflect padding) -> ReLU -> BatchNorm affine, written as chunk s of the block's
// concatenated output (ECAPA_TDNN.py Res2NetBlock).  y1 / cat: packed [T][C] operands (C = 64 * scale); grid = row tiles,
// 4 waves = the four 16-channel output tiles; the weights are the A operand (a lane ends with 4 consecutive output channels).
struct Res2Params {
  const void* y1;
  void* cat;
  const void* wp;      // packed [3 * 64][64]
  const float* bias;
  const float* scale;
  const float* shift;
  int T, mtp, s, dil, first;
};

template <typename T>
__global__ __launch_bounds__(256) void res2_step_kernel(Res2Params p) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
  const int t0 = (int)blockIdx.x * 16;
  const T* Y1 = (const T*)p.y1;
  T* CAT = (T*)p.cat;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  frag wf[6], xf[6];
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) {
    const int j = ks >> 1, kk = ks & 1;
    wf[ks] = ld16<frag>((const char*)p.wp + (((int64_t)wave * 6 + ks) * 64 + lane) * 16);
    const int row = reflect_idx(min(t0 + r, p.T - 1) + (j - 1) * p.dil, p.T);
    const int c0 = 64 * p.s + 32 * kk + 8 * g;
    const frag a = ld16<frag>(Y1 + pa_off<T>(row, c0, p.mtp));
    if (p.first) {
      xf[ks] = a;
    } else {
      const frag b = ld16<frag>(CAT + pa_off<T>(row, c0 - 64, p.mtp));
#pragma unroll
      for (int e = 0; e < 8; ++e) xf[ks][e] = EL::from_f(EL::to_f(a[e]) + EL::to_f(b[e]));
    }
  }
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) acc = EL::mma(wf[ks], xf[ks], acc);
  const int row = t0 + r, col = 16 * wave + 4 * g;
  if (row < p.T) {
    const f32x4 bs = ld16<f32x4>(p.bias + col), sc = ld16<f32x4>(p.scale + col), sh = ld16<f32x4>(p.shift + col);
    typedef T t4 __attribute__((ext_vector_type(4)));
    t4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = EL::from_f(fmaf(fmaxf(acc[e] + bs[e], 0.f), sc[e], sh[e]));
    *reinterpret_cast<t4*>(CAT + pa_off<T>(row, 64 * p.s + col, p.mtp)) = o;
  }
  if (p.first && tid < 128) {      // chunk 0 passes through unchanged
    const int rr = tid >> 3, c8 = (tid & 7) * 8;
    if (t0 + rr < p.T) st16(CAT + pa_off<T>(t0 + rr, c8, p.mtp), ld16<frag>(Y1 + pa_off<T>(t0 + rr, c8, p.mtp)));
  }
}

// Squeeze-and-excitation gate of a block: g = sigmoid(W2 relu(W1 mean_t(y) + b1) + b2), y packed [T][C]; W1 T [H][C], W2 T [C][H]
// row-major.  One workgroup of 1024 threads (300 frames x 512 channels: a few microseconds of one CU, one launch).
template <typename T>
__global__ __launch_bounds__(1024) void se_gate_kernel(const T* __restrict__ y, const T* __restrict__ w1, const float* __restrict__ b1,
                                                       const T* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ gate,
                                                       int Tn, int C, int H, int mtp) {
  extern __shared__ __attribute__((aligned(16))) float se_lds[];   // mean [C] | h [H]
  float* mean = se_lds;
  float* hid = mean + C;
  typedef typename Elem<T>::frag frag;
  const int tid = threadIdx.x;
  {
    // column sums: a wave walks the 1-KiB blocks of one 32-channel k-step (lane (g, r) = row r, channels 8 g .. 8 g + 7 of the
    // block: one contiguous wave-load per block), four blocks in flight, then folds the 16 rows of the lanes together
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
    for (int ks = wave; ks < C / 32; ks += 16) {
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int mt = 0; mt < mtp; mt += 4) {
        frag v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          v[j] = (mt + j < mtp && (mt + j) * 16 + r < Tn) ? ld16<frag>(y + (((int64_t)ks * mtp + mt + j) * 64 + lane) * 8) : zero_frag<frag>();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += Elem<T>::to_f(v[j][e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) a[e] += __shfl_xor(a[e], o, 64);
        if (r == 0) mean[ks * 32 + g * 8 + e] = a[e] / (float)Tn;
      }
    }
  }
  __syncthreads();
  for (int o = tid >> 3; o < H; o += 128) {          // 8 lanes per hidden unit; 8 weight requests in flight per lane
    const int part8 = tid & 7;
    float sum = 0.f;
    for (int i0 = part8 * 8; i0 < C; i0 += 512) {
      frag wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) wv[j] = i0 + 64 * j < C ? ld16<frag>(w1 + (int64_t)o * C + i0 + 64 * j) : zero_frag<frag>();
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (i0 + 64 * j < C)
#pragma unroll
          for (int e = 0; e < 8; ++e) sum = fmaf(Elem<T>::to_f(wv[j][e]), mean[i0 + 64 * j + e], sum);
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    sum += __shfl_xor(sum, 4, 64);
    if (part8 == 0) hid[o] = fmaxf(sum + b1[o], 0.f);
  }
  __syncthreads();
  for (int c = tid >> 1; c < C; c += 512) {          // 2 lanes per channel
    const int half = tid & 1;
    float sum = 0.f;
    for (int i0 = half * 8; i0 < H; i0 += 128) {
      frag wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) wv[j] = i0 + 16 * j < H ? ld16<frag>(w2 + (int64_t)c * H + i0 + 16 * j) : zero_frag<frag>();
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (i0 + 16 * j < H)
#pragma unroll
          for (int e = 0; e < 8; ++e) sum = fmaf(Elem<T>::to_f(wv[j][e]), hid[i0 + 16 * j + e], sum);
    }
    sum += __shfl_xor(sum, 1, 64);
    if (half == 0) gate[c] = 1.f / (1.f + __expf(-(sum + b2[c])));
  }
}

// out = gate[c] * y + res over packed [T][C] operands of identical geometry (the block's output: SE scaling + residual)
template <typename T>
__global__ __launch_bounds__(256) void scale_resid_kernel(const T* __restrict__ y, const T* __restrict__ res, const float* __restrict__ gate,
                                                          T* __restrict__ out, int nchunks, int mtp) {
  typedef typename Elem<T>::frag frag;
  const int q = (int)blockIdx.x * 256 + threadIdx.x;
  if (q >= nchunks) return;
  const int blk = q >> 6, lane = q & 63;
  const int ks = blk / mtp, col = ks * 32 + (lane >> 4) * 8;
  const frag a = ld16<frag>(y + (int64_t)q * 8), b = ld16<frag>(res + (int64_t)q * 8);
  const f32x4 g0 = ld16<f32x4>(gate + col), g1 = ld16<f32x4>(gate + col + 4);
  frag o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = Elem<T>::from_f(fmaf(e < 4 ? g0[e & 3] : g1[e & 3], Elem<T>::to_f(a[e]), Elem<T>::to_f(b[e])));
  st16(out + (int64_t)q * 8, o);
}

// Statistics over time of packed x [T][C], per channel, optionally weighted by softmax_t(logit[t][c]) (logit T row-major [T][C]):
//   w_t = softmax over t (or 1 / T);  m = sum_t w_t x_t;  s = sqrt(max(sum_t w_t (x_t - m)^2, 1e-12))
//   out (T [2 C]) = [m * scale[:C] + shift[:C] | s * scale[C:] + shift[C:]]   (scale / shift NULL = identity)
// = the global-context statistics and the attentive statistics pooling (+ BatchNorm) of ECAPA_TDNN.py:543-581.
// grid C / 64 workgroups of 1024 threads; up to 512 frames stay in registers over the passes.
template <typename T>
__global__ __launch_bounds__(1024) void col_stats_kernel(const T* __restrict__ x, const T* __restrict__ logit, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, T* __restrict__ out, int Tn, int C, int mtp) {
  // 64 channels = two 32-channel k-steps of the packed operand; wave w walks the 1-KiB blocks (k-step w & 1, row tiles w >> 1,
  // + 8, ...): lane (g, r) = row r, channels 8 g .. 8 g + 7 of a block, one contiguous wave-load per block
  __shared__ float red[8][64];
  __shared__ float tot[64];
  typedef typename Elem<T>::frag frag;
  constexpr int NR = 4;                               // blocks a lane keeps in registers (T <= 8 * 16 * NR = 512: one pass over memory)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
  const int ks2 = wave & 1, grp = wave >> 1;
  const int ks = (int)blockIdx.x * 2 + ks2;
  const int c0 = ks * 32 + 8 * g;                     // this lane's 8 channels
  const int cl = ks2 * 32 + 8 * g;                    // ... within the workgroup's 64
  auto reduce = [&](const float (&vin)[8], bool is_max) {   // over rows: the 16 r-lanes (shuffles), then the 8 waves of a k-step (LDS)
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = vin[e];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const float u = __shfl_xor(v[e], o, 64);
        v[e] = is_max ? fmaxf(v[e], u) : v[e] + u;
      }
    }
    __syncthreads();
    if (r == 0)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[grp][cl + e] = v[e];
    __syncthreads();
    if (tid < 64) {
      float a = red[0][tid];
#pragma unroll
      for (int i = 1; i < 8; ++i) a = is_max ? fmaxf(a, red[i][tid]) : a + red[i][tid];
      tot[tid] = a;
    }
    __syncthreads();
  };
  const bool wtd = logit != nullptr;
  const bool one_pass = mtp <= 8 * NR;
  frag xr[NR], lr[NR];
  auto row_of = [&](int mb, int j) { return (mb + grp + 8 * j) * 16 + r; };
  auto fetch = [&](int mb) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int mt = mb + grp + 8 * j, t = mt * 16 + r;
      xr[j] = t < Tn ? ld16<frag>(x + (((int64_t)ks * mtp + mt) * 64 + lane) * 8) : zero_frag<frag>();
      if (wtd) lr[j] = t < Tn ? ld16<frag>(logit + (int64_t)t * C + c0) : zero_frag<frag>();
    }
  };
  float mx[8], v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) mx[e] = 0.f;
  fetch(0);
  if (wtd) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = -INFINITY;
    for (int mb = 0; mb < mtp; mb += 8 * NR) {
      if (!one_pass) fetch(mb);
#pragma unroll
      for (int j = 0; j < NR; ++j)
        if (row_of(mb, j) < Tn)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], Elem<T>::to_f(lr[j][e]));
    }
    reduce(v, true);
#pragma unroll
    for (int e = 0; e < 8; ++e) mx[e] = tot[cl + e];
  }
  float se[8], sx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) se[e] = sx[e] = 0.f;
  for (int mb = 0; mb < mtp; mb += 8 * NR) {
    if (!one_pass) fetch(mb);
#pragma unroll
    for (int j = 0; j < NR; ++j)
      if (row_of(mb, j) < Tn)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float w = wtd ? __expf(Elem<T>::to_f(lr[j][e]) - mx[e]) : 1.f;
          se[e] += w;
          sx[e] = fmaf(w, Elem<T>::to_f(xr[j][e]), sx[e]);
        }
  }
  float den[8], mean[8];
  reduce(se, false);
#pragma unroll
  for (int e = 0; e < 8; ++e) den[e] = tot[cl + e];
  reduce(sx, false);
#pragma unroll
  for (int e = 0; e < 8; ++e) mean[e] = tot[cl + e] / den[e];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  for (int mb = 0; mb < mtp; mb += 8 * NR) {
    if (!one_pass) fetch(mb);
#pragma unroll
    for (int j = 0; j < NR; ++j)
      if (row_of(mb, j) < Tn)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float w = wtd ? __expf(Elem<T>::to_f(lr[j][e]) - mx[e]) : 1.f;
          const float d = Elem<T>::to_f(xr[j][e]) - mean[e];
          v[e] = fmaf(w * d, d, v[e]);
        }
  }
  reduce(v, false);
  if (grp == 0 && r == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e;
      float m = mean[e], sd = sqrtf(fmaxf(tot[cl + e] / den[e], 1e-12f));
      if (scale != nullptr) {
        m = fmaf(m, scale[c], shift[c]);
        sd = fmaf(sd, scale[C + c], shift[C + c]);
      }
      out[c] = Elem<T>::from_f(m);
      out[C + c] = Elem<T>::from_f(sd);
    }
  }
}

}  // namespace itts

using namespace itts;

extern "C" int itts_subsample_conv(const float* mel, const float* w, const float* b, void* y, int T, int F, int C, int dtype,
                                   void* stream) {
  ITTS_REQUIRE(mel && w && b && y && T >= 3 && F >= 3 && C > 0, "itts_subsample_conv: bad arguments");
  ITTS_REQUIRE(dtype == ITTS_BF16 || dtype == ITTS_F16, "itts_subsample_conv: the front-end kernels are built for bf16 / f16");
  const int T2 = (T - 3) / 2 + 1, F2 = (F - 3) / 2 + 1;
  hipStream_t s = (hipStream_t)stream;
  // staged form: every wave's run of cpw channels x F2 outputs is a whole number of 16-byte vectors and the row fits 64 KiB of LDS
  const int cpw = (C + 7) / 8;
  const bool staged = C % 8 == 0 && (cpw * F2) % 8 == 0 && (size_t)C * F2 * 2 <= 64 * 1024;
  const size_t lds = staged ? (size_t)C * F2 * 2 : 0;
  if (dtype == ITTS_BF16) {
    if (staged) hipLaunchKernelGGL((subsample_conv_kernel<bf16_t, true>), dim3(T2), dim3(512), lds, s, mel, w, b, (bf16_t*)y, F, C, F2);
    else hipLaunchKernelGGL((subsample_conv_kernel<bf16_t, false>), dim3(T2), dim3(512), 0, s, mel, w, b, (bf16_t*)y, F, C, F2);
  } else {
    if (staged) hipLaunchKernelGGL((subsample_conv_kernel<f16_t, true>), dim3(T2), dim3(512), lds, s, mel, w, b, (f16_t*)y, F, C, F2);
    else hipLaunchKernelGGL((subsample_conv_kernel<f16_t, false>), dim3(T2), dim3(512), 0, s, mel, w, b, (f16_t*)y, F, C, F2);
  }
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
  const dim3 grid((a->Tq + 15) / 16, a->H), block(256);
  hipStream_t s = (hipStream_t)stream;
#define ITTS_MHA(T_, R_) hipLaunchKernelGGL((mha_small_kernel<T_, R_>), grid, block, 0, s, p)
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

extern "C" int itts_prefix_rows(const int64_t* text, const float* conds, int conds_rows, const float* text_emb, const float* text_pos,
                                float* emb, int64_t* mask, int32_t* pad, int B, int L, int C, int D, int start_tok, int stop_tok,
                                int n_tok, int n_pos, void* stream) {
  ITTS_REQUIRE((text || L == 0) && (conds || C == 0) && text_emb && text_pos && emb && mask && pad, "itts_prefix_rows: null args");
  ITTS_REQUIRE(B > 0 && B <= 65535 && L >= 0 && L <= PREFIX_MAXL && C >= 0 && D > 0 && D % 4 == 0 && n_tok > 0 && n_pos > 0,
               "itts_prefix_rows: bad shape (L <= %d, D %% 4 == 0)", PREFIX_MAXL);
  ITTS_REQUIRE(conds_rows == 1 || conds_rows == B, "itts_prefix_rows: conds has 1 or B rows");
  const int P = C + L + 2;
  hipLaunchKernelGGL(prefix_rows_kernel, dim3((P + 7) / 8, B), dim3(256), 0, (hipStream_t)stream, text, conds, conds_rows, text_emb,
                     text_pos, emb, mask, pad, L, C, D, start_tok, stop_tok, n_tok, n_pos);
  return check_launch("itts_prefix_rows");
}

#define ITTS_FE_DTYPE(who) ITTS_REQUIRE(dtype == ITTS_BF16 || dtype == ITTS_F16, who ": built for bf16 / f16")

extern "C" int itts_im2col_reflect(const float* x, void* y, int T, int F, int taps, int dil, int Kp, int y_mtp, int dtype, void* stream) {
  ITTS_REQUIRE(x && y && T > 0 && F > 0 && taps > 0 && taps % 2 == 1 && dil > 0, "itts_im2col_reflect: bad arguments");
  ITTS_FE_DTYPE("itts_im2col_reflect");
  ITTS_REQUIRE(Kp % 32 == 0 && Kp >= taps * F && y_mtp * 16 >= T, "itts_im2col_reflect: Kp %% 32 == 0, Kp >= taps * F, y_mtp covers T");
  ITTS_REQUIRE((taps - 1) / 2 * dil < T, "itts_im2col_reflect: the reflect padding needs more than %d frames", (taps - 1) / 2 * dil);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == ITTS_BF16) hipLaunchKernelGGL(im2col_reflect_kernel<bf16_t>, dim3(T), dim3(256), 0, s, x, (bf16_t*)y, T, F, taps, dil, Kp, y_mtp);
  else hipLaunchKernelGGL(im2col_reflect_kernel<f16_t>, dim3(T), dim3(256), 0, s, x, (f16_t*)y, T, F, taps, dil, Kp, y_mtp);
  return check_launch("itts_im2col_reflect");
}

extern "C" int itts_res2_step(const void* y1, void* cat, const void* wp, const float* bias, const float* scale, const float* shift, int T,
                              int mtp, int chunk, int dil, int first, int dtype, void* stream) {
  ITTS_REQUIRE(y1 && cat && wp && bias && scale && shift && T > 0 && chunk >= 1 && dil > 0, "itts_res2_step: bad arguments");
  ITTS_FE_DTYPE("itts_res2_step");
  ITTS_REQUIRE(mtp * 16 >= T && dil < T, "itts_res2_step: mtp covers T, dilation < T");
  Res2Params p;
  p.y1 = y1; p.cat = cat; p.wp = wp; p.bias = bias; p.scale = scale; p.shift = shift;
  p.T = T; p.mtp = mtp; p.s = chunk; p.dil = dil; p.first = first != 0;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == ITTS_BF16) hipLaunchKernelGGL(res2_step_kernel<bf16_t>, dim3((T + 15) / 16), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(res2_step_kernel<f16_t>, dim3((T + 15) / 16), dim3(256), 0, s, p);
  return check_launch("itts_res2_step");
}

extern "C" int itts_se_gate(const void* y, const void* w1, const float* b1, const void* w2, const float* b2, float* gate, int T, int C,
                            int H, int mtp, int dtype, void* stream) {
  ITTS_REQUIRE(y && w1 && b1 && w2 && b2 && gate && T > 0, "itts_se_gate: bad arguments");
  ITTS_FE_DTYPE("itts_se_gate");
  ITTS_REQUIRE(C % 64 == 0 && C <= 1024 && H % 16 == 0 && H <= 512 && mtp * 16 >= T, "itts_se_gate: C %% 64, C <= 1024, H %% 16, H <= 512");
  const size_t lds = (size_t)(C + H) * 4;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == ITTS_BF16)
    hipLaunchKernelGGL(se_gate_kernel<bf16_t>, dim3(1), dim3(1024), lds, s, (const bf16_t*)y, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, gate, T, C, H, mtp);
  else
    hipLaunchKernelGGL(se_gate_kernel<f16_t>, dim3(1), dim3(1024), lds, s, (const f16_t*)y, (const f16_t*)w1, b1, (const f16_t*)w2, b2, gate, T, C, H, mtp);
  return check_launch("itts_se_gate");
}

extern "C" int itts_scale_resid(const void* y, const void* res, const float* gate, void* out, int T, int C, int mtp, int dtype, void* stream) {
  ITTS_REQUIRE(y && res && gate && out && T > 0 && C % 32 == 0 && mtp * 16 >= T, "itts_scale_resid: bad arguments");
  ITTS_FE_DTYPE("itts_scale_resid");
  const int nchunks = C / 32 * mtp * 64;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((nchunks + 255) / 256), block(256);
  if (dtype == ITTS_BF16) hipLaunchKernelGGL(scale_resid_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)y, (const bf16_t*)res, gate, (bf16_t*)out, nchunks, mtp);
  else hipLaunchKernelGGL(scale_resid_kernel<f16_t>, grid, block, 0, s, (const f16_t*)y, (const f16_t*)res, gate, (f16_t*)out, nchunks, mtp);
  return check_launch("itts_scale_resid");
}

extern "C" int itts_col_stats(const void* x, const void* logit, const float* scale, const float* shift, void* out, int T, int C, int mtp,
                              int dtype, void* stream) {
  ITTS_REQUIRE(x && out && T > 0 && C % 64 == 0 && mtp * 16 >= T, "itts_col_stats: bad arguments (C %% 64 == 0)");
  ITTS_REQUIRE((scale == nullptr) == (shift == nullptr), "itts_col_stats: scale and shift come together");
  ITTS_FE_DTYPE("itts_col_stats");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == ITTS_BF16)
    hipLaunchKernelGGL(col_stats_kernel<bf16_t>, dim3(C / 64), dim3(1024), 0, s, (const bf16_t*)x, (const bf16_t*)logit, scale, shift, (bf16_t*)out, T, C, mtp);
  else
    hipLaunchKernelGGL(col_stats_kernel<f16_t>, dim3(C / 64), dim3(1024), 0, s, (const f16_t*)x, (const f16_t*)logit, scale, shift, (f16_t*)out, T, C, mtp);
  return check_launch("itts_col_stats");
}
