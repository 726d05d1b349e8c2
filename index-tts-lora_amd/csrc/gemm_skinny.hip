// Skinny (decode-step) GEMM: Y[M<=32][N] = epi( pro(X)[M][K] @ W[K][N] + bias ).
//
// HBM-bound weight streaming: every 1-KiB packed weight block is read exactly once per launch, straight into the
// MFMA B-fragment registers (no LDS round trip: the block is not shared between waves), several blocks in flight per
// wave.  One workgroup owns one 16-column tile; its waves split the K range and their fp32 partial tiles are summed
// in a FIXED order through LDS (deterministic, no atomics).  LayerNorm prologues are computed directly in A-fragment
// layout (row statistics combined across the waves through a few hundred bytes of LDS), so the normalised activations
// never touch LDS or HBM.  Replaces the per-step Conv1D/LayerNorm/Linear calls of HF GPT2Block as driven by
// indextts/gpt/model.py:163-193.
#include "common.h"

namespace itts {

constexpr int SK_CH = 5;  // k-steps per register chunk

struct SkinnyParams {
  int M, N, K;
  const void* wp;
  const float* bias;
  int pro;
  const void* x;
  const float* h;
  const float *ln_w, *ln_b, *ln2_w, *ln2_b;
  int epi;
  void* y;
  float* yf;
  void* kcache;
  void* vcache;
  const int32_t* pos;
  int heads, smax;
};

// Sum `v` over the 4 lanes that share the same fragment row r (lanes r, r+16, r+32, r+48).
__device__ __forceinline__ float rowgroup_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

template <typename T, int MT, bool LN>
__global__ __launch_bounds__((LN && sizeof(T) == 2) ? 512 : 1024) void gemm_skinny_kernel(SkinnyParams p) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  constexpr int E = EL::E, KS = EL::KS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // smem layout: red[NW][MT][64][4] | stat[NW][MT*16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
  const int nt = blockIdx.x;
  const int g = lane >> 4, r = lane & 15;
  const int KT = p.K / KS;
  const int spw = (KT + NW - 1) / NW;
  const int s_begin = wave * spw;
  const int s_end = min(KT, s_begin + spw);
  float* red = smem;
  float* stat = smem + NW * MT * 256;

  const char* bp = (const char*)p.wp + ((int64_t)nt * KT * 64 + lane) * 16;

  frag bf[SK_CH];
#pragma unroll
  for (int i = 0; i < SK_CH; ++i) {
    int s = s_begin + i;
    bf[i] = (s < s_end) ? ld16<frag>(bp + (int64_t)s * 1024) : zero_frag<frag>();
  }

  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (LN) {
    // ---- LayerNorm prologue in fragment layout; the wave's whole K range is one chunk (host guarantees spw <= SK_CH)
    float xa[SK_CH][MT][E];
#pragma unroll
    for (int i = 0; i < SK_CH; ++i) {
      int s = s_begin + i;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        int row = mt * 16 + r;
        bool ok = (s < s_end) && (row < p.M);
        const float* src = p.h + (int64_t)row * p.K + s * KS + g * E;
#pragma unroll
        for (int e4 = 0; e4 < E / 4; ++e4) {
          f32x4 v = ok ? ld16<f32x4>(src + e4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) xa[i][mt][e4 * 4 + e] = v[e];
        }
      }
    }
    const int nln = (p.pro == ITTS_PRO_LN2) ? 2 : 1;
    for (int pass = 0; pass < nln; ++pass) {
      const float* lw = pass == 0 ? p.ln_w : p.ln2_w;
      const float* lb = pass == 0 ? p.ln_b : p.ln2_b;
      float mean[MT], rstd[MT];
      // mean
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < SK_CH; ++i)
#pragma unroll
          for (int e = 0; e < E; ++e) s1 += xa[i][mt][e];
        s1 = rowgroup_sum(s1);
        if (g == 0) stat[wave * (MT * 16) + mt * 16 + r] = s1;
      }
      __syncthreads();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float t = 0.f;
        for (int w = 0; w < NW; ++w) t += stat[w * (MT * 16) + mt * 16 + r];
        mean[mt] = t / (float)p.K;
      }
      __syncthreads();
      // variance (two-pass); positions outside [s_begin,s_end) hold zeros and must not contribute
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < SK_CH; ++i) {
          if (s_begin + i < s_end) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
              float d = xa[i][mt][e] - mean[mt];
              s2 = fmaf(d, d, s2);
            }
          }
        }
        s2 = rowgroup_sum(s2);
        if (g == 0) stat[wave * (MT * 16) + mt * 16 + r] = s2;
      }
      __syncthreads();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float t = 0.f;
        for (int w = 0; w < NW; ++w) t += stat[w * (MT * 16) + mt * 16 + r];
        rstd[mt] = rsqrtf(t / (float)p.K + 1e-5f);
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < SK_CH; ++i) {
        int s = s_begin + i;
        if (s < s_end) {
          float wv[E], bv[E];
#pragma unroll
          for (int e4 = 0; e4 < E / 4; ++e4) {
            f32x4 a = ld16<f32x4>(lw + s * KS + g * E + e4 * 4), b = ld16<f32x4>(lb + s * KS + g * E + e4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              wv[e4 * 4 + e] = a[e];
              bv[e4 * 4 + e] = b[e];
            }
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < E; ++e) xa[i][mt][e] = (xa[i][mt][e] - mean[mt]) * rstd[mt] * wv[e] + bv[e];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < SK_CH; ++i) {
      if (s_begin + i < s_end) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          frag af;
#pragma unroll
          for (int e = 0; e < E; ++e) af[e] = EL::from_f(xa[i][mt][e]);
          acc[mt] = EL::mma(af, bf[i], acc[mt]);
        }
      }
    }
  } else {
    // ---- A straight from global (T [M][K]); weight chunks double-buffered in registers
    const T* X = (const T*)p.x;
    for (int base = s_begin; base < s_end; base += SK_CH) {
      frag bn[SK_CH];
      const int nxt = base + SK_CH;
#pragma unroll
      for (int i = 0; i < SK_CH; ++i) {
        int s = nxt + i;
        bn[i] = (s < s_end) ? ld16<frag>(bp + (int64_t)s * 1024) : zero_frag<frag>();
      }
#pragma unroll
      for (int i = 0; i < SK_CH; ++i) {
        int s = base + i;
        if (s < s_end) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            int row = mt * 16 + r;
            frag af = (row < p.M) ? ld16<frag>(X + (int64_t)row * p.K + s * KS + g * E) : zero_frag<frag>();
            acc[mt] = EL::mma(af, bf[i], acc[mt]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < SK_CH; ++i) bf[i] = bn[i];
    }
  }

  // ---- cross-wave reduction, fixed order
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) st16(red + ((wave * MT + mt) * 64 + lane) * 4, acc[mt]);
  __syncthreads();
  for (int e = tid; e < MT * 256; e += blockDim.x) {
    int mt = e >> 8, rr = (e >> 4) & 15, c = e & 15;
    int src = ((rr >> 2) << 4) | c, j = rr & 3;
    float v = 0.f;
    for (int w = 0; w < NW; ++w) v += red[((w * MT + mt) * 64 + src) * 4 + j];
    int row = mt * 16 + rr, col = nt * 16 + c;
    if (row >= p.M || col >= p.N) continue;
    if (p.bias) v += p.bias[col];
    switch (p.epi) {
      case ITTS_EPI_STORE:
        ((T*)p.y)[(int64_t)row * p.N + col] = EL::from_f(v);
        break;
      case ITTS_EPI_GELU_STORE:
        ((T*)p.y)[(int64_t)row * p.N + col] = EL::from_f(gelu_new(v));
        break;
      case ITTS_EPI_RESID_F32:
        p.yf[(int64_t)row * p.N + col] += v;
        break;
      case ITTS_EPI_STORE_F32:
        p.yf[(int64_t)row * p.N + col] = v;
        break;
      case ITTS_EPI_QKV_CACHE: {
        int D = p.N / 3;
        if (col < D) {
          ((T*)p.y)[(int64_t)row * D + col] = EL::from_f(v);
        } else {
          int cc = col - D;
          T* cache = (T*)(cc < D ? p.kcache : p.vcache);
          if (cc >= D) cc -= D;
          int hh = cc >> 6, dd = cc & 63;
          int pos = p.pos[0];
          cache[(((int64_t)row * p.heads + hh) * p.smax + pos) * 64 + dd] = EL::from_f(v);
        }
      } break;
    }
  }
}

template <typename T, int MT>
static int launch_skinny(const SkinnyParams& p, int NW, hipStream_t s) {
  if (p.pro != ITTS_PRO_NONE && sizeof(T) == 2 && NW > 8) {
    set_error("itts_gemm_skinny: LayerNorm prologue supports at most 8 waves for 16-bit types");
    return ITTS_ERR_INVALID;
  }
  size_t lds = (size_t)NW * MT * 256 * 4 + (size_t)NW * MT * 16 * 4;
  int NT = (p.N + 15) / 16;
  if (p.pro != ITTS_PRO_NONE)
    hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, true>), dim3(NT), dim3(NW * 64), lds, s, p);
  else
    hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, false>), dim3(NT), dim3(NW * 64), lds, s, p);
  return check_launch("itts_gemm_skinny");
}

}  // namespace itts

using namespace itts;

extern "C" int itts_gemm_skinny(const itts_skinny_args* a, void* stream) {
  ITTS_REQUIRE(a && a->wp, "itts_gemm_skinny: null args");
  const int ks = a->dtype == ITTS_F32 ? 16 : 32;
  const size_t esz = a->dtype == ITTS_F32 ? 4 : 2;
  ITTS_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0 && a->K % ks == 0, "itts_gemm_skinny: bad shape M=%d N=%d K=%d (K %% %d != 0)",
               a->M, a->N, a->K, ks);
  ITTS_REQUIRE(a->pro == ITTS_PRO_NONE ? a->x != nullptr : (a->h && a->ln_w && a->ln_b), "itts_gemm_skinny: missing prologue input");
  ITTS_REQUIRE(a->pro != ITTS_PRO_LN2 || (a->ln2_w && a->ln2_b), "itts_gemm_skinny: missing second LayerNorm");
  if (a->epi == ITTS_EPI_QKV_CACHE)
    ITTS_REQUIRE(a->y && a->kcache && a->vcache && a->pos && a->N % 3 == 0 && a->N / 3 == a->heads * 64 && a->smax > 0,
                 "itts_gemm_skinny: bad QKV epilogue arguments");
  else if (a->epi == ITTS_EPI_RESID_F32 || a->epi == ITTS_EPI_STORE_F32)
    ITTS_REQUIRE(a->yf, "itts_gemm_skinny: yf is null");
  else
    ITTS_REQUIRE((a->epi == ITTS_EPI_STORE || a->epi == ITTS_EPI_GELU_STORE) && a->y, "itts_gemm_skinny: bad epilogue %d", a->epi);
  if (a->M == 0) return ITTS_OK;
  const int KT = a->K / ks;
  int NW;
  if (a->pro != ITTS_PRO_NONE) {
    NW = (KT + SK_CH - 1) / SK_CH;
    ITTS_REQUIRE(NW <= 16, "itts_gemm_skinny: K=%d too large for a LayerNorm prologue (max %d)", a->K, 16 * SK_CH * ks);
  } else {
    NW = (KT + 2 * SK_CH - 1) / (2 * SK_CH);  // ~10 k-steps per wave
    if (NW > 16) NW = 16;
  }
  if (NW < 1) NW = 1;
  hipStream_t s = (hipStream_t)stream;
  const int rows_per = (a->dtype == ITTS_F32) ? 16 : 32;
  for (int r0 = 0; r0 < a->M; r0 += rows_per) {
    SkinnyParams p;
    p.M = a->M - r0 < rows_per ? a->M - r0 : rows_per;
    p.N = a->N;
    p.K = a->K;
    p.wp = a->wp;
    p.bias = a->bias;
    p.pro = a->pro;
    p.x = a->x ? (const char*)a->x + (size_t)r0 * a->K * esz : nullptr;
    p.h = a->h ? a->h + (size_t)r0 * a->K : nullptr;
    p.ln_w = a->ln_w;
    p.ln_b = a->ln_b;
    p.ln2_w = a->ln2_w;
    p.ln2_b = a->ln2_b;
    p.epi = a->epi;
    const size_t ycols = a->epi == ITTS_EPI_QKV_CACHE ? (size_t)a->N / 3 : (size_t)a->N;
    p.y = a->y ? (char*)a->y + (size_t)r0 * ycols * esz : nullptr;
    p.yf = a->yf ? a->yf + (size_t)r0 * a->N : nullptr;
    const size_t crow = (size_t)a->heads * a->smax * 64 * esz;
    p.kcache = a->kcache ? (char*)a->kcache + (size_t)r0 * crow : nullptr;
    p.vcache = a->vcache ? (char*)a->vcache + (size_t)r0 * crow : nullptr;
    p.pos = a->pos;
    p.heads = a->heads;
    p.smax = a->smax;
    int rc;
    if (a->dtype == ITTS_F32) {
      rc = launch_skinny<float, 1>(p, NW, s);
    } else if (a->dtype == ITTS_BF16) {
      rc = p.M <= 16 ? launch_skinny<bf16_t, 1>(p, NW, s) : launch_skinny<bf16_t, 2>(p, NW, s);
    } else if (a->dtype == ITTS_F16) {
      rc = p.M <= 16 ? launch_skinny<f16_t, 1>(p, NW, s) : launch_skinny<f16_t, 2>(p, NW, s);
    } else {
      ITTS_REQUIRE(false, "itts_gemm_skinny: unknown dtype %d", a->dtype);
    }
    if (rc != ITTS_OK) return rc;
  }
  return ITTS_OK;
}
