// Skinny (decode-step) GEMM: Y[M<=32][N] = epi( X[M][K] @ W[K][N] + bias ).
//
// HBM-bound weight streaming, latency-first structure: every global load a wave needs (its 1-KiB packed weight blocks,
// straight into MFMA B-fragment registers, and its A fragments of the T-typed activations) is issued BEFORE the first
// use, so a launch costs one memory round trip plus the stream time.  A workgroup owns one 16-column tile and one slice
// of K (grid.y = split-K factor); its waves split that slice and their fp32 partial tiles are summed through LDS in a
// FIXED order.  With split-K > 1 each slice stores its partial tile into its own fp32 slab [ks][M][N]; the slabs are
// summed, again in a fixed order, by the consumer (itts_ln_reduce), so the result is deterministic and needs no atomics
// and no extra launch.  Replaces the per-step Conv1D/Linear calls of HF GPT2Block as driven by
// indextts/gpt/model.py:163-193.
#include "common.h"

namespace itts {

struct SkinnyParams {
  int M, N, K;
  const void* wp;
  const float* bias;
  const void* x;
  int epi;
  void* y;
  float* yf;
  void* kcache;
  void* vcache;
  const int32_t* pos;
  int heads, smax;
  int ksplit;
  int slab_rows;  // total rows of a slab (the caller's M), the stride between split-K slabs
};

// NTB = column tiles per workgroup.  More than one workgroup per CU does not overlap for this kernel (measured: 257
// column tiles cost a full second round), so shapes with more than 256 tiles give each workgroup several tiles instead.
template <typename T, int MT, int SPW, int NTB>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(SkinnyParams p) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  constexpr int E = EL::E, KS = EL::KS;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [NW][NTB][MT][64][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
  const int nt0 = blockIdx.x * NTB, ks = blockIdx.y;
  const int NTtot = (p.N + 15) / 16;
  const int g = lane >> 4, r = lane & 15;
  const int KT = p.K / KS;
  const int SB = (KT + p.ksplit - 1) / p.ksplit;  // k-steps per split slice
  const int b_begin = ks * SB, b_end = min(KT, b_begin + SB);
  const int spw = (b_end - b_begin + NW - 1) / NW;
  const int s_begin = b_begin + wave * spw;
  const int s_end = min(b_end, s_begin + spw);

  const char* bp = (const char*)p.wp + ((int64_t)nt0 * KT * 64 + lane) * 16;  // tile t of this workgroup: + t*KT*1024
  const T* X = (const T*)p.x;

  // epilogue operands are requested now, so that their latency overlaps the weight stream
  float bias_pre = 0.f;
  int pos_pre = 0;
  if (tid < NTB * MT * 256) {  // the element this thread handles first in the epilogue (e == tid)
    int col = (nt0 + tid / (MT * 256)) * 16 + (tid & 15);
    if (p.bias != nullptr && ks == 0 && col < p.N) bias_pre = p.bias[col];
    if (p.epi == ITTS_EPI_QKV_CACHE) pos_pre = p.pos[0];
  }

  f32x4 acc[NTB][MT];
#pragma unroll
  for (int t = 0; t < NTB; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int base = s_begin; base < s_end; base += SPW) {
    frag bf[NTB][SPW];
    frag af[SPW][MT];
#pragma unroll
    for (int t = 0; t < NTB; ++t)
#pragma unroll
      for (int i = 0; i < SPW; ++i) {
        int s = base + i;
        bf[t][i] = (s < s_end && nt0 + t < NTtot) ? ld16<frag>(bp + ((int64_t)t * KT + s) * 1024) : zero_frag<frag>();
      }
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
      int s = base + i;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        int row = mt * 16 + r;
        af[i][mt] = (s < s_end && row < p.M) ? ld16<frag>(X + (int64_t)row * p.K + s * KS + g * E) : zero_frag<frag>();
      }
    }
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
#pragma unroll
      for (int t = 0; t < NTB; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = EL::mma(af[i][mt], bf[t][i], acc[t][mt]);
    }
  }

  // ---- cross-wave reduction, fixed order
#pragma unroll
  for (int t = 0; t < NTB; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) st16(red + (((wave * NTB + t) * MT + mt) * 64 + lane) * 4, acc[t][mt]);
  __syncthreads();
  for (int e = tid; e < NTB * MT * 256; e += blockDim.x) {
    int t = e / (MT * 256), e1 = e - t * (MT * 256);
    int mt = e1 >> 8, rr = (e1 >> 4) & 15, c = e1 & 15;
    int row = mt * 16 + rr, col = (nt0 + t) * 16 + c;
    if (row >= p.M || col >= p.N) continue;
    float bs = (e == tid) ? bias_pre : ((p.bias != nullptr && ks == 0) ? p.bias[col] : 0.f);
    int src = ((rr >> 2) << 4) | c, j = rr & 3;
    float v = 0.f;
    for (int w = 0; w < NW; ++w) v += red[(((w * NTB + t) * MT + mt) * 64 + src) * 4 + j];
    v += bs;
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
      case ITTS_EPI_SLAB_F32:
        p.yf[((int64_t)ks * p.slab_rows + row) * p.N + col] = v;
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
          int pos = (e == tid) ? pos_pre : p.pos[0];
          cache[(((int64_t)row * p.heads + hh) * p.smax + pos) * 64 + dd] = EL::from_f(v);
        }
      } break;
    }
  }
}

template <typename T, int MT>
static int launch_skinny(const SkinnyParams& p, hipStream_t s) {
  constexpr int KS = Elem<T>::KS;
  const int KT = p.K / KS;
  const int SB = (KT + p.ksplit - 1) / p.ksplit;
  // waves per workgroup: ~5 k-steps per wave, at most 8 waves; more than 5 steps per wave -> 10-step register chunks
  int NW = (SB + 4) / 5;
  if (NW > 8) NW = 8;
  if (NW < 1) NW = 1;
  const int spw = (SB + NW - 1) / NW;
  const int NT = (p.N + 15) / 16;
  // keep the grid within one round of the 256 CUs
  int ntb = (NT * p.ksplit + 255) / 256;
  if (ntb > 3) ntb = 3;
  if (spw > 5 && ntb > 2) ntb = 2;  // register budget of the 10-step variant
  size_t lds = (size_t)NW * ntb * MT * 256 * 4;
  dim3 grid((NT + ntb - 1) / ntb, p.ksplit), block(NW * 64);
#define ITTS_SK(SPW_, NTB_) hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, SPW_, NTB_>), grid, block, lds, s, p)
  if (spw <= 5) {
    if (ntb == 1) ITTS_SK(5, 1); else if (ntb == 2) ITTS_SK(5, 2); else ITTS_SK(5, 3);
  } else {
    if (ntb == 1) ITTS_SK(10, 1); else ITTS_SK(10, 2);
  }
#undef ITTS_SK
  return check_launch("itts_gemm_skinny");
}

}  // namespace itts

using namespace itts;

extern "C" int itts_gemm_skinny(const itts_skinny_args* a, void* stream) {
  ITTS_REQUIRE(a && a->wp && a->x, "itts_gemm_skinny: null args");
  const int ks = a->dtype == ITTS_F32 ? 16 : 32;
  const size_t esz = a->dtype == ITTS_F32 ? 4 : 2;
  ITTS_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0 && a->K % ks == 0, "itts_gemm_skinny: bad shape M=%d N=%d K=%d (K %% %d != 0)",
               a->M, a->N, a->K, ks);
  const int ksplit = a->ksplit > 0 ? a->ksplit : 1;
  ITTS_REQUIRE(ksplit <= a->K / ks && ksplit <= 64, "itts_gemm_skinny: ksplit=%d too large", ksplit);
  ITTS_REQUIRE(ksplit == 1 || a->epi == ITTS_EPI_SLAB_F32, "itts_gemm_skinny: ksplit > 1 requires the slab epilogue");
  if (a->epi == ITTS_EPI_QKV_CACHE)
    ITTS_REQUIRE(a->y && a->kcache && a->vcache && a->pos && a->N % 3 == 0 && a->N / 3 == a->heads * 64 && a->smax > 0,
                 "itts_gemm_skinny: bad QKV epilogue arguments");
  else if (a->epi == ITTS_EPI_RESID_F32 || a->epi == ITTS_EPI_STORE_F32 || a->epi == ITTS_EPI_SLAB_F32)
    ITTS_REQUIRE(a->yf, "itts_gemm_skinny: yf is null");
  else
    ITTS_REQUIRE((a->epi == ITTS_EPI_STORE || a->epi == ITTS_EPI_GELU_STORE) && a->y, "itts_gemm_skinny: bad epilogue %d", a->epi);
  if (a->M == 0) return ITTS_OK;
  hipStream_t s = (hipStream_t)stream;
  const int rows_per = (a->dtype == ITTS_F32) ? 16 : 32;
  for (int r0 = 0; r0 < a->M; r0 += rows_per) {
    SkinnyParams p;
    p.M = a->M - r0 < rows_per ? a->M - r0 : rows_per;
    p.N = a->N;
    p.K = a->K;
    p.wp = a->wp;
    p.bias = a->bias;
    p.x = (const char*)a->x + (size_t)r0 * a->K * esz;
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
    p.ksplit = ksplit;
    p.slab_rows = a->M;
    int rc;
    if (a->dtype == ITTS_F32) {
      rc = launch_skinny<float, 1>(p, s);
    } else if (a->dtype == ITTS_BF16) {
      rc = p.M <= 16 ? launch_skinny<bf16_t, 1>(p, s) : launch_skinny<bf16_t, 2>(p, s);
    } else if (a->dtype == ITTS_F16) {
      rc = p.M <= 16 ? launch_skinny<f16_t, 1>(p, s) : launch_skinny<f16_t, 2>(p, s);
    } else {
      ITTS_REQUIRE(false, "itts_gemm_skinny: unknown dtype %d", a->dtype);
    }
    if (rc != ITTS_OK) return rc;
  }
  return ITTS_OK;
}
