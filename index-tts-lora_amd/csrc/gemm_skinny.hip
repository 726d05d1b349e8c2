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

// Build-time A/B switch: non-temporal policy for the once-read weight blocks.  Measured neutral in the token loop
// (1100.8 vs 1102 us per token; FC / FC2 -0.2..0.3 us, QKV / out-projection +0.2 us per launch), default policy kept.
#ifndef ITTS_NT_WEIGHTS
#define ITTS_NT_WEIGHTS 0
#endif

namespace itts {

template <typename F>
__device__ __forceinline__ F ldw(const void* p) {
  if constexpr (ITTS_NT_WEIGHTS) return ld16_nt<F>(p);
  else return ld16<F>(p);
}

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
  int slab_rows;
  float* ln_h;            // fused LayerNorm producer stage (see itts_skinny_args)
  const float* ln_slab;
  int ln_nslab;
  const float* ln_bias;
  const float* ln_w;
  const float* ln_b;
  int32_t* ln_counter;
  int32_t* ln_counter_prev;
  int lnf;  // A operand is the fp32 residual stream, normalised per row (LayerNorm without affine) on the fly  // total rows of a slab (the caller's M), the stride between split-K slabs
};

// NTB = column tiles per workgroup.  More than one workgroup per CU does not overlap for this kernel (measured: 257
// column tiles cost a full second round), so shapes with more than 256 tiles give each workgroup several tiles instead.
// LNF: the A operand is fp32 [M][K] and is LayerNorm-normalised on the fly ((x - mean) * rstd, eps 1e-5; the affine part
// is folded into the packed weights / bias by the caller).  Each wave computes exact two-pass statistics of its own K
// slice, the slices are merged across the waves with Chan's parallel-variance formula through LDS (one barrier), and the
// normalised values are rounded to T only then -- same numerics as LayerNorm in fp32 followed by a T-typed matmul.
// Requires ksplit == 1 and the wave's whole K slice in one register chunk.
// MAXT = threads per workgroup the register budget is sized for (512: up to 8 waves; 1024: up to 16 waves, which halves
// the loads queued per lane at the price of a 128-register cap).
// Producer stage of a fused launch: one row of x = LayerNorm(h + bias + slabs) per workgroup, numerics and association
// order of ln_reduce_wide_kernel.  The row is published for the GEMM workgroups of the SAME launch with write-through
// (sc1) stores, a drain of every storing wave, a workgroup barrier and one agent-scope counter add (the R1 recipe of the
// CDNA hand-off rules); the residual stream itself is only read by later launches and is stored normally.
template <typename T>
__device__ __forceinline__ void fused_ln_row(const SkinnyParams& p, float* red) {
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int D = p.K;
  const bool act = tid * 4 < D;
  const int o = act ? tid * 4 : 0;
  float* hr = p.ln_h + (int64_t)row * D;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 v = ld16<f32x4>(hr + o);
  const f32x4 lw = ld16<f32x4>(p.ln_w + o), lb = ld16<f32x4>(p.ln_b + o);
  {
    // all loads unconditional (absent operands alias the row itself and are discarded): one round trip, no branches
    const float* bsrc = p.ln_bias != nullptr ? p.ln_bias + o : hr + o;
    f32x4 bs = ld16<f32x4>(bsrc);
    f32x4 sl[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float* ssrc = i < p.ln_nslab ? p.ln_slab + ((int64_t)i * p.slab_rows + row) * D + o : hr + o;
      sl[i] = ld16<f32x4>(ssrc);
    }
    if (p.ln_nslab > 0) {
      v += (p.ln_bias != nullptr) ? bs : zero;
#pragma unroll
      for (int i = 0; i < 3; ++i) v += (i < p.ln_nslab) ? sl[i] : zero;
      if (act) st16(hr + o, v);
    }
  }
  if (!act) v = zero;
  float s = wave_sum(v[0] + v[1] + v[2] + v[3]);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  float tot = 0.f;
  for (int i = 0; i < nw; ++i) tot += red[i];
  const float mean = tot / (float)D;
  float q = 0.f;
  if (act) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float d = v[e] - mean;
      q = fmaf(d, d, q);
    }
  }
  q = wave_sum(q);
  if (lane == 0) red[32 + wave] = q;
  __syncthreads();
  float qt = 0.f;
  for (int i = 0; i < nw; ++i) qt += red[32 + i];
  const float rstd = rsqrtf(qt / (float)D + 1e-5f);
  if (act) {
    T* yr = (T*)const_cast<void*>(p.x) + (int64_t)row * D + o;
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (v[e] - mean) * rstd * lw[e] + lb[e];
    if constexpr (sizeof(T) == 4) {
      uint64_t lo = ((uint64_t)__float_as_uint(r[1]) << 32) | __float_as_uint(r[0]);
      uint64_t hi = ((uint64_t)__float_as_uint(r[3]) << 32) | __float_as_uint(r[2]);
      __hip_atomic_store((uint64_t*)yr, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((uint64_t*)yr + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      typedef T t4 __attribute__((ext_vector_type(4)));
      t4 ov = {Elem<T>::from_f(r[0]), Elem<T>::from_f(r[1]), Elem<T>::from_f(r[2]), Elem<T>::from_f(r[3])};
      __hip_atomic_store((uint64_t*)yr, __builtin_bit_cast(uint64_t, ov), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the workgroup signals
  __syncthreads();
  if (tid == 0) {
    if (row == 0 && p.ln_counter_prev != nullptr) *p.ln_counter_prev = 0;  // that launch has completed (stream order)
    __hip_atomic_fetch_add(p.ln_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <typename T, int MT, int SPW, int NTB, bool LNF, int MAXT = 512, bool FUSE = false>
__global__ __launch_bounds__(MAXT) void gemm_skinny_kernel(SkinnyParams p) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  constexpr int E = EL::E, KS = EL::KS;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [NW][NTB][MT][64][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
  if constexpr (FUSE) {
    if ((int)blockIdx.x < p.M) {
      fused_ln_row<T>(p, red);
      return;
    }
  }
  const int nt0 = (FUSE ? (int)blockIdx.x - p.M : (int)blockIdx.x) * NTB, ks = blockIdx.y;
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

  if constexpr (LNF) {
    const float* H = (const float*)p.x;
    float* stat = red + NW * NTB * MT * 256;  // [NW][MT*16][2] (mean, M2) of each wave's K slice
    frag bf[NTB][SPW];
    float xa[SPW][MT][E];
#pragma unroll
    for (int t = 0; t < NTB; ++t)
#pragma unroll
      for (int i = 0; i < SPW; ++i) {
        int s = s_begin + i;
        bf[t][i] = (s < s_end && nt0 + t < NTtot) ? ldw<frag>(bp + ((int64_t)t * KT + s) * 1024) : zero_frag<frag>();
      }
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
      int s = s_begin + i;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        int row = mt * 16 + r;
        bool ok = (s < s_end) && (row < p.M);
        const float* src = H + (int64_t)row * p.K + s * KS + g * E;
#pragma unroll
        for (int e4 = 0; e4 < E / 4; ++e4) {
          f32x4 v = ok ? ld16<f32x4>(src + e4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) xa[i][mt][e4 * 4 + e] = v[e];
        }
      }
    }
    const float n_w = (float)((s_end - s_begin) * KS);  // elements of a row in this wave's slice
    float mean_w[MT], m2_w[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float s1 = 0.f;
#pragma unroll
      for (int i = 0; i < SPW; ++i)
#pragma unroll
        for (int e = 0; e < E; ++e) s1 += xa[i][mt][e];  // slots past s_end hold zeros
      s1 += __shfl_xor(s1, 16, 64);
      s1 += __shfl_xor(s1, 32, 64);
      mean_w[mt] = n_w > 0.f ? s1 / n_w : 0.f;
      float s2 = 0.f;
#pragma unroll
      for (int i = 0; i < SPW; ++i)
        if (s_begin + i < s_end) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            float d = xa[i][mt][e] - mean_w[mt];
            s2 = fmaf(d, d, s2);
          }
        }
      s2 += __shfl_xor(s2, 16, 64);
      s2 += __shfl_xor(s2, 32, 64);
      m2_w[mt] = s2;
      if (g == 0) {
        stat[(wave * (MT * 16) + mt * 16 + r) * 2 + 0] = mean_w[mt];
        stat[(wave * (MT * 16) + mt * 16 + r) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    // merge the NW slices (all slices but possibly the last have spw*KS elements)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float tot = 0.f, msum = 0.f;
      for (int w = 0; w < NW; ++w) {
        int sb = b_begin + w * spw, se = min(b_end, sb + spw);
        float nw = (float)(max(se - sb, 0) * KS);
        msum += nw * stat[(w * (MT * 16) + mt * 16 + r) * 2];
        tot += nw;
      }
      float mean = msum / tot, m2 = 0.f;
      for (int w = 0; w < NW; ++w) {
        int sb = b_begin + w * spw, se = min(b_end, sb + spw);
        float nw = (float)(max(se - sb, 0) * KS);
        float d = stat[(w * (MT * 16) + mt * 16 + r) * 2] - mean;
        m2 += stat[(w * (MT * 16) + mt * 16 + r) * 2 + 1] + nw * d * d;
      }
      float rstd = rsqrtf(m2 / tot + 1e-5f);
#pragma unroll
      for (int i = 0; i < SPW; ++i) {
        frag af;
#pragma unroll
        for (int e = 0; e < E; ++e) af[e] = EL::from_f((xa[i][mt][e] - mean) * rstd);
        if (s_begin + i < s_end) {
#pragma unroll
          for (int t = 0; t < NTB; ++t) acc[t][mt] = EL::mma(af, bf[t][i], acc[t][mt]);
        }
      }
    }
  } else {
  for (int base = s_begin; base < s_end; base += SPW) {
    frag bf[NTB][SPW];
    frag af[SPW][MT];
#pragma unroll
    for (int t = 0; t < NTB; ++t)
#pragma unroll
      for (int i = 0; i < SPW; ++i) {
        int s = base + i;
        bf[t][i] = (s < s_end && nt0 + t < NTtot) ? ldw<frag>(bp + ((int64_t)t * KT + s) * 1024) : zero_frag<frag>();
      }
    if constexpr (FUSE) {
      // the weight blocks are in flight; now wait for the producer workgroups' rows: ONE lane polls the counter (relaxed),
      // ONE agent-scope acquire drops this CU's stale lines, the barrier releases the other waves to plain loads
      if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(p.ln_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.M) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 22)) break;  // never expected: the producers are the first workgroups of this launch
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
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

int g_tune_ntb = 0, g_tune_nw = 0;  // itts_debug_set(1|2, v): tuning overrides (0 = heuristic)

template <typename T, int MT>
static int launch_skinny(const SkinnyParams& p, hipStream_t s) {
  constexpr int KS = Elem<T>::KS;
  const int KT = p.K / KS;
  const int SB = (KT + p.ksplit - 1) / p.ksplit;
  // waves per workgroup: 8 whenever the slice has 8 k-steps (measured: the split-K 3 out-projection, 14 k-steps, takes
  // 3.8 us with 8 waves x 2 steps against 4.4 us with 3 waves x 5; 10-16 waves change nothing for any of the four GEMMs);
  // more than 5 steps per wave -> 10-step register chunks
  int NW = SB;
  if (NW > 8) NW = 8;
  if (NW < 1) NW = 1;
  if (g_tune_nw > 0 && !p.lnf) NW = g_tune_nw > 16 ? 16 : g_tune_nw;
  if (NW > 8 && (SB + NW - 1) / NW > 5) NW = 8;  // the 16-wave build only exists for 5-step register chunks
  const int spw = (SB + NW - 1) / NW;
  const int NT = (p.N + 15) / 16;
  // keep the grid within one round of the 256 CUs
  int ntb = (NT * p.ksplit + 255) / 256;
  if (g_tune_ntb > 0) ntb = g_tune_ntb;
  if (ntb > 3) ntb = 3;
  if (spw > 5 && ntb > 2) ntb = 2;  // register budget of the 10-step variant
  size_t lds = (size_t)NW * ntb * MT * 256 * 4 + (p.lnf ? (size_t)NW * MT * 16 * 2 * 4 : 0);
  dim3 grid((NT + ntb - 1) / ntb, p.ksplit), block(NW * 64);
  if (p.ln_h != nullptr) {
    // fused producer stage: M extra workgroups in front; one pass of the k-loop only (all weight blocks in flight at once)
    if (p.ksplit != 1 || p.lnf || spw > 5 || NW != 8 || p.K % 256 != 0 || p.K / 4 > NW * 64 || p.ln_nslab < 0 || p.ln_nslab > 3) {
      set_error("itts_gemm_skinny: the fused LayerNorm stage needs ksplit 1, K %% 256 == 0, K <= %d, <= 3 slabs", 8 * 5 * KS);
      return ITTS_ERR_INVALID;
    }
    grid.x += p.M;
#define ITTS_SKF(NTB_) hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, 5, NTB_, false, 512, true>), grid, block, lds, s, p)
    if (ntb == 1) ITTS_SKF(1); else if (ntb == 2) ITTS_SKF(2); else ITTS_SKF(3);
#undef ITTS_SKF
    return check_launch("itts_gemm_skinny");
  }
#define ITTS_SK(SPW_, NTB_, LNF_) hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, SPW_, NTB_, LNF_>), grid, block, lds, s, p)
  if (p.lnf) {
    if (spw > 5 || p.ksplit != 1) {
      set_error("itts_gemm_skinny: the LayerNorm-fused A operand needs K <= %d and ksplit == 1", 8 * 5 * KS);
      return ITTS_ERR_INVALID;
    }
    if (ntb == 1) ITTS_SK(5, 1, true); else if (ntb == 2) ITTS_SK(5, 2, true); else ITTS_SK(5, 3, true);
  } else if (spw <= 5 && NW > 8) {
#define ITTS_SK16(NTB_) hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, 5, NTB_, false, 1024>), grid, block, lds, s, p)
    if (ntb == 1) ITTS_SK16(1); else if (ntb == 2) ITTS_SK16(2); else ITTS_SK16(3);
#undef ITTS_SK16
  } else if (spw <= 5) {
    if (ntb == 1) ITTS_SK(5, 1, false); else if (ntb == 2) ITTS_SK(5, 2, false); else ITTS_SK(5, 3, false);
  } else {
    if (ntb == 1) ITTS_SK(10, 1, false); else ITTS_SK(10, 2, false);
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
  if (a->ln_h != nullptr)
    ITTS_REQUIRE(a->ln_w && a->ln_b && a->ln_counter && (a->ln_nslab == 0 || a->ln_slab) && !a->x_ln_f32 &&
                     a->M <= (a->dtype == ITTS_F32 ? 16 : 32),
                 "itts_gemm_skinny: bad fused-LayerNorm arguments (needs M <= 32 rows, 16 in fp32)");
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
    p.x = (const char*)a->x + (size_t)r0 * a->K * (a->x_ln_f32 ? 4 : esz);
    p.lnf = a->x_ln_f32 ? 1 : 0;
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
    p.ln_h = a->ln_h;
    p.ln_slab = a->ln_slab;
    p.ln_nslab = a->ln_nslab;
    p.ln_bias = a->ln_bias;
    p.ln_w = a->ln_w;
    p.ln_b = a->ln_b;
    p.ln_counter = a->ln_counter;
    p.ln_counter_prev = a->ln_counter_prev;
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

namespace itts { extern int g_conv_cfg; extern int g_attn_waves; }

extern "C" int itts_debug_set(int key, int value) {
  if (key == 1) itts::g_tune_ntb = value;
  else if (key == 2) itts::g_tune_nw = value;
  else if (key == 3) itts::g_conv_cfg = value;
  else if (key == 4) itts::g_attn_waves = (value == 8) ? 8 : 4;
  else return ITTS_ERR_INVALID;
  return ITTS_OK;
}
