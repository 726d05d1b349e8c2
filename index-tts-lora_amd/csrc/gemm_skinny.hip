// Skinny (decode-step) GEMM: Y[M<=96][N] = epi( X[M][K] @ W[K][N] + bias ).
//
// HBM-bound weight streaming, latency-first structure: every global load a wave needs (its 1-KiB packed weight blocks,
// straight into MFMA operand registers, and its fragments of the T-typed activations) is issued BEFORE the first use, so
// a launch costs one memory round trip plus the stream time.  A workgroup owns 1-3 16-column tiles and one slice of K
// (grid.y = split-K factor); its waves split that slice and their fp32 partial tiles are summed through LDS in a FIXED
// order.  The MFMAs run with the WEIGHT fragment as the A operand: the accumulator tile comes out transposed, a lane
// holds FOUR CONSECUTIVE OUTPUT COLUMNS of one batch row, and every epilogue access is 8 or 16 bytes wide.
// Up to 96 rows (6 row tiles) share one pass over the weights: batch 32 x 3 beams (the infer() default) or several pooled
// requests read the 966 MB of decoder weights ONCE per token.
//
// With split-K > 1 each slice stores its partial tile into its own fp32 slab [ks][M][N]; the next launch (itts_ln_reduce)
// sums the slabs in a fixed order.
//
// LayerNorm folded into the consumer (FOLD, round 4): the QKV and FC projections of a block consume LN(h).  With
//   LN(h) W + b = rstd (h (gamma . W) - mean c) + d,   c_j = sum_k gamma_k W_kj,   d_j = sum_k beta_k W_kj + b_j
// (gamma . W packed at load time, c and d fp32 vectors) the GEMM multiplies the RAW residual rows -- a T-typed packed copy
// of h that the producing launch's residual epilogue writes -- and needs the row statistics only in its EPILOGUE.  They
// come from the matrix pipe, off the fragments the wave holds anyway: sum h = ones x frag, sum h^2 = diagonal of frag x
// frag (the 16 x 16 Gram tile of the row tile), accumulated in fp32 over the wave's K share and reduced across the waves
// together with the partial tiles.  Every workgroup covers the whole K, so every workgroup has the full statistics of
// all its rows: no cross-workgroup hand-off, no [residual-reduce + LayerNorm] launch in front of the GEMM.  The
// producers (attention out-projection, FC2) run without split-K and add into the fp32 residual stream in their epilogue
// (one owner per element: deterministic) and also store the T-typed packed copy.  A block is 5 launches instead of 7.
// Replaces the per-step Conv1D/Linear (+ residual + LayerNorm) calls of HF GPT2Block as driven by
// indextts/gpt/model.py:163-193.
#include "common.h"
#include "ln_math.h"
#include <type_traits>

#ifndef ITTS_FOLD_ORDER
#define ITTS_FOLD_ORDER 0   // build-time A/B of the LayerNorm-folded form: 0 = activations requested first, statistics MFMAs under the weight
                            // stream; 1 = weights first, statistics MFMAs behind the main ones
#endif
#ifndef ITTS_NT_WEIGHTS
#define ITTS_NT_WEIGHTS 0   // build-time A/B: non-temporal policy for the once-read weight blocks (measured neutral)
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
  const int32_t* kv_tab;   // QKV epilogue into a paged cache: block table [rows][ITTS_KV_TAB], or NULL
  int kv_bs_log2;
  const float* cvec;       // FOLD: c_j = sum_k gamma_k W_kj (bias then holds d_j)
  float ln_eps;
  int32_t* bump;           // one device word this launch increments (it must not read it)
  int x_pa, y_pa;          // packed-activation layout for x / y
  int mtp, row0;           // row tiles of the WHOLE operand, first row of this launch (a multiple of 16)
  int y_mtp, y_row0;       // the same for a packed y (the rows may land inside a taller packed operand)
  const float* post_scale; // RELU_AFFINE epilogues: y = relu(v) * post_scale[n] + post_shift[n]
  const float* post_shift;
#if ITTS_STAMPS
  unsigned long long* stamps;
#endif
#if ITTS_DIAG
  int exp;   // diagnostic ablations: bit 0 = every activation fragment is k-step 0's (L1-resident), bit 1 = every weight block is block 0
#endif
};

#if ITTS_STAMPS
unsigned long long* g_stamp_buf = nullptr;
unsigned long long* g_stamp_buf_sample = nullptr;
#define ITTS_STAMP(i) ITTS_STAMP_IF(p.stamps != nullptr, i)
#define ITTS_STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define ITTS_STAMP(i) do { } while (0)
#define ITTS_STAMP_DRAIN() do { } while (0)
#endif

// 4 consecutive elements of a row; `nval` of them exist (N need not be a multiple of 4: the 8194-column head)
template <typename T>
__device__ __forceinline__ void store4(T* dst, const f32x4& v, int nval) {
  if (nval >= 4 && ((reinterpret_cast<uintptr_t>(dst) & (sizeof(T) * 4 - 1)) == 0)) {
    if constexpr (sizeof(T) == 4) {
      st16(dst, v);
    } else {
      typedef T t4 __attribute__((ext_vector_type(4)));
      t4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f(v[e]);
      *reinterpret_cast<t4*>(dst) = o;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < nval) dst[e] = Elem<T>::from_f(v[e]);
  }
}

__device__ __forceinline__ f32x4 load4f(const float* src, int nval) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (nval >= 4 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) return ld16<f32x4>(src);
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (e < nval) v[e] = src[e];
  return v;
}

template <typename F>
__device__ __forceinline__ F ones_frag() {
  F z;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(F) / sizeof(z[0])); ++i) z[i] = 1;
  return z;
}

// NTB = column tiles per workgroup (grids stay within one round of the 256 CUs: a 257th workgroup costs a full second
// round for this kernel), SPW = k-steps a wave keeps in registers per pass, MT = 16-row tiles per workgroup (grid.z walks
// the row tiles of the launch in groups of MT), FOLD = LayerNorm folded into this GEMM (see the head of the file),
// MAXW = most waves per workgroup (16: the register budget of a 1024-thread workgroup, 128 per lane).
template <typename T, int MT, int SPW, int NTB, bool FOLD, int MAXW>
__global__ __launch_bounds__(MAXW * 64) void gemm_skinny_kernel(SkinnyParams p) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  constexpr int E = EL::E, KS = EL::KS;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [NW][NTB][MT][64][4] | FOLD: [NW][MT][16][2]
#if ITTS_STAMPS
  unsigned long long st_[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) st_[i] = 0;
  unsigned long long rt0_ = 0;
  if (p.stamps != nullptr && threadIdx.x == 0) rt0_ = __builtin_amdgcn_s_memrealtime();
#endif
  ITTS_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
  const int nt0 = (int)blockIdx.x * NTB, ks = blockIdx.y;
  const int mt0 = (int)blockIdx.z * MT;       // first row tile of this workgroup inside the launch's rows
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

  // Epilogue operands of this wave's output units are requested now, in front of the weight stream: their latency overlaps it
  // and the epilogue issues no load of its own.  pre2 = the residual values the RESID epilogue adds to, or (FOLD) c.
  // UPRE units per wave cover every launch with >= 8 waves; launches with fewer waves (tiny K) finish in a second loop.
  constexpr int UPRE = (NTB * MT + 7) / 8;
  f32x4 bias_pre[UPRE], pre2[UPRE];
  int pos_pre = 0;
  {
    // range-checked loads: a null bias, another K slice, columns past N (the 8194-column head), rows past M read zeros -- no
    // branch, so no join at which the compiler would wait for this round trip before the weight requests go out
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias), 0, p.bias != nullptr ? p.N * 4 : 0, 0x00020000);
    const bool resid = !FOLD && p.epi == ITTS_EPI_RESID_F32;
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(
        FOLD ? (void*)const_cast<float*>(p.cvec) : (void*)p.yf, 0, FOLD ? p.N * 4 : (resid ? p.M * p.N * 4 : 0), 0x00020000);
#pragma unroll
    for (int ui = 0; ui < UPRE; ++ui) {
      const int u = wave + ui * NW;
      const int t = u / MT, mt = u - t * MT;
      const int col0 = (nt0 + t) * 16 + g * 4, row = (mt0 + mt) * 16 + r;
      const bool uok = u < NTB * MT && col0 < p.N;
      const unsigned boff = (uok && ks == 0) ? (unsigned)col0 * 4u : 0x80000000u;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        bias_pre[ui][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, boff + 4u * e, 0, 0));
      const unsigned o2 = !uok ? 0x80000000u : FOLD ? (unsigned)col0 * 4u : (row < p.M ? (unsigned)(row * p.N + col0) * 4u : 0x80000000u);
      pre2[ui] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2, o2, 0, 0));
    }
  }
  if (wave < NTB * MT && p.epi == ITTS_EPI_QKV_CACHE) pos_pre = p.pos[0];

  f32x4 acc[NTB][MT];
#pragma unroll
  for (int t = 0; t < NTB; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // FOLD: row statistics of the raw rows on the matrix pipe.  s1: ones x frag -> every lane (g, r) holds sum_k h[r][k] in all
  // four elements; s2: frag x frag -> lane (g, r) element e holds sum_k h[4g+e][k] h[r][k], the diagonal (g == r>>2, e == r&3)
  // is sum_k h[r][k]^2.  Exact products of the T-typed values, fp32 accumulation.
  f32x4 s1[FOLD ? MT : 1], s2[FOLD ? MT : 1];
#pragma unroll
  for (int mt = 0; mt < (FOLD ? MT : 1); ++mt) s1[mt] = s2[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const frag ones = ones_frag<frag>();

  auto x_frag = [&](int s, const int mt) -> frag {
    const int row = (mt0 + mt) * 16 + r;
#if ITTS_DIAG
    if (p.exp & 1) s = s < s_end ? 0 : s;
#endif
    if (p.x_pa)   // one contiguous 1-KiB block per (k-step, row tile); padding rows exist and are never stored
      return (s < s_end) ? ld16<frag>(X + (((int64_t)s * p.mtp + (p.row0 >> 4) + mt0 + mt) * 64 + lane) * E) : zero_frag<frag>();
    return (s < s_end && row < p.M) ? ld16<frag>(X + (int64_t)row * p.K + s * KS + g * E) : zero_frag<frag>();
  };
  auto w_frag = [&](int s, const int t) -> frag {
#if ITTS_DIAG
    if (p.exp & 2) s = s < s_end ? 0 : s;
#endif
    return (s < s_end && nt0 + t < NTtot) ? ldw<frag>(bp + ((int64_t)t * KT + s) * 1024) : zero_frag<frag>();
  };

  // One pass over SPW k-steps from `base`.  The FIRST pass always runs (a wave without a K share requests nothing and adds
  // zeros): every wave executes one straight line -- operand requests, one wait, MFMAs -- with no join in front of the
  // requests (round 3: together with the branch-free bias preload and the LDS-only barrier, 5.65 -> 5.29 us per launch
  // over a block's four GEMMs against the round-2 kernel on the same box, tools/probes/ab_r02_gemm.py).
  // paged KV cache (QKV epilogue): the block that holds the append position of each of this wave's output rows.  Requested
  // BEHIND the first pass's operand requests (it needs the position word, and nothing before the epilogue needs it).
  int blk_pre[UPRE];
#pragma unroll
  for (int ui = 0; ui < UPRE; ++ui) blk_pre[ui] = 0;
  auto table_requests = [&]() {
    const int32_t* tp = p.kv_tab != nullptr ? p.kv_tab : (const int32_t*)p.wp;   // a readable word either way: a select, no branch
#pragma unroll
    for (int ui = 0; ui < UPRE; ++ui) {
      const int u = wave + ui * NW;
      const int mt = u % MT;
      const int row = (mt0 + mt) * 16 + r;
      const int idx = (p.kv_tab != nullptr && row < p.M) ? (p.row0 + row) * ITTS_KV_TAB + ((pos_pre >> p.kv_bs_log2) & (ITTS_KV_TAB - 1)) : 0;
      blk_pre[ui] = tp[idx];
    }
  };
  auto k_pass = [&](const int base, auto first_tag) {
    constexpr bool FIRST_PASS = decltype(first_tag)::value;
    frag bf[NTB][SPW];
    if constexpr (MT <= 2) {
      frag af[SPW][MT];
      if constexpr (FOLD && ITTS_FOLD_ORDER == 0) {
        // activations FIRST (vmcnt retires in issue order): the statistics MFMAs below need only them and run while the
        // weight blocks, the long pole from HBM, are still in flight
#pragma unroll
        for (int i = 0; i < SPW; ++i)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) af[i][mt] = x_frag(base + i, mt);
      }
#pragma unroll
      for (int t = 0; t < NTB; ++t)
#pragma unroll
        for (int i = 0; i < SPW; ++i) bf[t][i] = w_frag(base + i, t);
      if constexpr (!FOLD || ITTS_FOLD_ORDER != 0) {
#pragma unroll
        for (int i = 0; i < SPW; ++i)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) af[i][mt] = x_frag(base + i, mt);
      }
      if constexpr (FIRST_PASS) table_requests();
      ITTS_STAMP(1);
#if ITTS_STAMPS
      ITTS_STAMP_DRAIN();
      ITTS_STAMP(2);
#endif
      auto stats = [&]() {
#pragma unroll
        for (int i = 0; i < SPW; ++i)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            s1[mt] = EL::mma(ones, af[i][mt], s1[mt]);
            s2[mt] = EL::mma(af[i][mt], af[i][mt], s2[mt]);
          }
      };
      if constexpr (FOLD && ITTS_FOLD_ORDER == 0) stats();
#pragma unroll
      for (int i = 0; i < SPW; ++i) {
#pragma unroll
        for (int t = 0; t < NTB; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[t][mt] = EL::mma(bf[t][i], af[i][mt], acc[t][mt]);  // weights = A operand
      }
      if constexpr (FOLD && ITTS_FOLD_ORDER != 0) stats();
    } else {
      // more than 32 rows: the activation fragments (L2-resident, shared by every workgroup) are fetched row tile by row
      // tile behind the weight blocks; the unrolled loop lets the loads of tile mt+1 fly under the MFMAs of tile mt
#pragma unroll
      for (int t = 0; t < NTB; ++t)
#pragma unroll
        for (int i = 0; i < SPW; ++i) bf[t][i] = w_frag(base + i, t);
      if constexpr (FIRST_PASS) table_requests();
      ITTS_STAMP(1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        frag af[SPW];
#pragma unroll
        for (int i = 0; i < SPW; ++i) af[i] = x_frag(base + i, mt);
        if constexpr (FOLD) {
#pragma unroll
          for (int i = 0; i < SPW; ++i) {
            s1[mt] = EL::mma(ones, af[i], s1[mt]);
            s2[mt] = EL::mma(af[i], af[i], s2[mt]);
          }
        }
#pragma unroll
        for (int i = 0; i < SPW; ++i)
#pragma unroll
          for (int t = 0; t < NTB; ++t) acc[t][mt] = EL::mma(bf[t][i], af[i], acc[t][mt]);
      }
      ITTS_STAMP(2);
    }
  };
  k_pass(s_begin, std::true_type{});
  for (int base = s_begin + SPW; base < s_end; base += SPW) k_pass(base, std::false_type{});
  ITTS_STAMP(3);

  // ---- cross-wave reduction, fixed order.  Lane (g, r) of a tile holds Y[row = mt*16 + r][col = tile*16 + 4g .. 4g+3].
#pragma unroll
  for (int t = 0; t < NTB; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) st16(red + (((wave * NTB + t) * MT + mt) * 64 + lane) * 4, acc[t][mt]);
  float* stat = red + NW * NTB * MT * 256;   // FOLD: [wave][mt][row 0..15] x {sum, sum of squares}
  if constexpr (FOLD) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int e = r & 3;
      const float d2 = e == 0 ? s2[mt][0] : e == 1 ? s2[mt][1] : e == 2 ? s2[mt][2] : s2[mt][3];
      if (g == (r >> 2)) {
        float* sp = stat + ((wave * MT + mt) * 16 + r) * 2;
        sp[0] = s1[mt][0];
        sp[1] = d2;
      }
    }
  }
  // LDS-only wait + raw barrier (__syncthreads() would also drain vmcnt)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  ITTS_STAMP(4);
  // one output unit (column tile t, row tile mt): sum the waves' partial tiles, add the bias, apply the epilogue
  auto unit = [&](const int u, const f32x4 bs, const f32x4 p2, const int blk) {
    const int t = u / MT, mt = u - t * MT;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < NW; ++w) v += ld16<f32x4>(red + (((w * NTB + t) * MT + mt) * 64 + lane) * 4);
    const int row = (mt0 + mt) * 16 + r, col0 = (nt0 + t) * 16 + g * 4;
    if constexpr (FOLD) {
      float S1 = 0.f, S2 = 0.f;
      for (int w = 0; w < NW; ++w) {
        const float* sp = stat + ((w * MT + mt) * 16 + r) * 2;
        S1 += sp[0];
        S2 += sp[1];
      }
      const float inv = 1.0f / (float)p.K;
      const float mean = S1 * inv;
      const float var = fmaxf(fmaf(-mean, mean, S2 * inv), 0.f);
      const float rstd = rsqrtf(var + p.ln_eps);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(rstd, fmaf(-mean, p2[e], v[e]), bs[e]);   // rstd (h W' - mean c) + d
    } else {
      v += bs;
    }
    if (row >= p.M || col0 >= p.N) return;
    const int nval = min(4, p.N - col0);
    switch (p.epi) {
      case ITTS_EPI_STORE:
        store4<T>((T*)p.y + (p.y_pa ? pa_off<T>(p.y_row0 + row, col0, p.y_mtp) : (int64_t)row * p.N + col0), v, nval);
        break;
      case ITTS_EPI_GELU_STORE: {
        f32x4 gv = {gelu_new(v[0]), gelu_new(v[1]), gelu_new(v[2]), gelu_new(v[3])};
        store4<T>((T*)p.y + (p.y_pa ? pa_off<T>(p.y_row0 + row, col0, p.y_mtp) : (int64_t)row * p.N + col0), gv, nval);
      } break;
      case ITTS_EPI_SILU_STORE: {
        f32x4 sv;
#pragma unroll
        for (int e = 0; e < 4; ++e) sv[e] = v[e] / (1.f + __expf(-v[e]));
        store4<T>((T*)p.y + (p.y_pa ? pa_off<T>(p.y_row0 + row, col0, p.y_mtp) : (int64_t)row * p.N + col0), sv, nval);
      } break;
      case ITTS_EPI_RELU_AFFINE_STORE:
      case ITTS_EPI_RELU_AFFINE_TANH_STORE: {
        // TDNN block of the speaker encoder: BatchNorm (eval: an affine map per channel) BEHIND the ReLU
        const f32x4 sc = load4f(p.post_scale + col0, nval), sh = load4f(p.post_shift + col0, nval);
        f32x4 rv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          rv[e] = fmaf(fmaxf(v[e], 0.f), sc[e], sh[e]);
          if (p.epi == ITTS_EPI_RELU_AFFINE_TANH_STORE) rv[e] = tanhf(rv[e]);
        }
        store4<T>((T*)p.y + (p.y_pa ? pa_off<T>(p.y_row0 + row, col0, p.y_mtp) : (int64_t)row * p.N + col0), rv, nval);
      } break;
      case ITTS_EPI_RESID_F32: {
        // residual stream update, one owner per element (no split-K): the old values were requested with the bias.  The
        // T-typed copy (p.y, optional) is what the next LayerNorm-folded GEMM multiplies.
        if constexpr (!FOLD) {
          const f32x4 nv = p2 + v;
          store4<float>(p.yf + (int64_t)row * p.N + col0, nv, nval);
          if (p.y != nullptr)
            store4<T>((T*)p.y + (p.y_pa ? pa_off<T>(p.y_row0 + row, col0, p.y_mtp) : (int64_t)row * p.N + col0), nv, nval);
        }
      } break;
      case ITTS_EPI_STORE_F32:
        store4<float>(p.yf + (int64_t)row * p.N + col0, v, nval);
        break;
      case ITTS_EPI_SLAB_F32:
        store4<float>(p.yf + ((int64_t)ks * p.slab_rows + row) * p.N + col0, v, nval);
        break;
      case ITTS_EPI_QKV_CACHE: {
        const int D = p.N / 3;   // a 4-column group never straddles q|k|v or a head (all multiples of 64)
        if (col0 < D) {
          store4<T>((T*)p.y + (int64_t)row * D + col0, v, nval);
        } else {
          int cc = col0 - D;
          T* cache = (T*)(cc < D ? p.kcache : p.vcache);
          if (cc >= D) cc -= D;
          const int hh = cc >> 6, dd = cc & 63;
          const int64_t at = p.kv_tab == nullptr
                                 ? (((int64_t)row * p.heads + hh) * p.smax + pos_pre) * 64
                                 : ((((int64_t)blk * p.heads + hh) << p.kv_bs_log2) + (pos_pre & ((1 << p.kv_bs_log2) - 1))) * 64;
          store4<T>(cache + at + dd, v, nval);
        }
      } break;
    }
  };
#pragma unroll
  for (int ui = 0; ui < UPRE; ++ui) {
    const int u = wave + ui * NW;
    if (u < NTB * MT) unit(u, bias_pre[ui], pre2[ui], blk_pre[ui]);
  }
  for (int u = wave + UPRE * NW; u < NTB * MT; u += NW) {   // fewer than 8 waves (tiny K): the remaining units
    f32x4 bs = {0.f, 0.f, 0.f, 0.f}, p2 = bs;
    const int t = u / MT, mt = u - t * MT;
    const int col0 = (nt0 + t) * 16 + g * 4, row = (mt0 + mt) * 16 + r;
    if (col0 < p.N) {
      if (p.bias != nullptr && ks == 0) bs = load4f(p.bias + col0, p.N - col0);
      if constexpr (FOLD) p2 = load4f(p.cvec + col0, p.N - col0);
      else if (p.epi == ITTS_EPI_RESID_F32 && row < p.M) p2 = load4f(p.yf + (int64_t)row * p.N + col0, p.N - col0);
    }
    int blk = 0;
    if (p.kv_tab != nullptr && row < p.M)
      blk = p.kv_tab[(p.row0 + row) * ITTS_KV_TAB + ((pos_pre >> p.kv_bs_log2) & (ITTS_KV_TAB - 1))];
    unit(u, bs, p2, blk);
  }
  // the loop-state word this launch advances (nothing in this launch reads it)
  if (p.bump != nullptr && tid == 0 && (blockIdx.x | blockIdx.y | blockIdx.z) == 0) p.bump[0] += 1;
  ITTS_STAMP(5);
#if ITTS_STAMPS
  if (p.stamps != nullptr && threadIdx.x == 0) {
    ITTS_STAMP_DRAIN();
    unsigned long long te_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te_)::"memory");
    unsigned xcc_;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
    unsigned long long* o_ = p.stamps + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16;
#pragma unroll
    for (int i = 0; i < 10; ++i) o_[i] = st_[i];
    o_[10] = te_;
    o_[11] = rt0_;
    o_[12] = __builtin_amdgcn_s_memrealtime();
    o_[13] = xcc_ & 0xF;
  }
#endif
}

// Launch geometry of one skinny GEMM (shared by the launcher and by tools through itts_skinny_plan)
struct SkinnyPlan {
  int NW, spw, ntb, gx, gy, gz, SPWc, MT;
  size_t lds;
};

#if ITTS_DIAG
int g_tune_ntb = 0, g_tune_nw = 0;  // diagnostic build: itts_debug_set(1|2, v) overrides (0 = heuristic)
int g_skinny_exp = 0;               // itts_debug_set(6, bits): load ablations of the skinny GEMM (SkinnyParams::exp)
#endif

// MTall = 16-row tiles of the launch; rows_per_wg (0 = all of them in every workgroup, else 16 / 32: the row tiles are dealt
// to grid.z -- more, lighter workgroups for the GEMMs that run WITHOUT split-K); wide: up to 16 waves per workgroup
template <typename T>
static SkinnyPlan plan_skinny(int N, int K, int ksplit, int MTall, int rows_per_wg, bool wide, bool fold) {
  constexpr int KS = Elem<T>::KS;
  SkinnyPlan q;
  int MT = MTall;
  if (rows_per_wg > 0 && rows_per_wg / 16 < MTall) MT = rows_per_wg / 16;
  const int gz = (MTall + MT - 1) / MT;
  const int KT = K / KS;
  const int SB = (KT + ksplit - 1) / ksplit;
  // waves per workgroup: 8 whenever the slice has 8 k-steps (measured: the split-K 3 out-projection, 14 k-steps, takes
  // 3.8 us with 8 waves x 2 steps against 4.4 us with 3 waves x 5; more than 8 waves change nothing there).  `wide`: 16 waves
  // when that keeps the slice to ONE pass of <= 10 k-steps per wave (a second pass is a second memory round trip)
  int NW = SB > 8 ? 8 : SB;
  if (wide && MT == 1 && SB > 80) NW = 16;
  if (NW < 1) NW = 1;
#if ITTS_DIAG
  if (g_tune_nw > 0) NW = g_tune_nw > 8 ? 8 : g_tune_nw;
#endif
  const int spw = (SB + NW - 1) / NW;
  const int NT = (N + 15) / 16;
  int ntb = (NT * ksplit * gz + 255) / 256;   // keep the grid within one round of the 256 CUs
#if ITTS_DIAG
  if (g_tune_ntb > 0) ntb = g_tune_ntb;
#endif
  const int SPWc = spw <= 5 ? 5 : 10;     // register-chunk variant
  const int ntb_max = (fold && MT <= 2 && SPWc == 5) ? 4 : 3;   // 4 tiles: the folded GEMMs of a many-row step with the rows dealt to grid.z
  if (ntb > ntb_max) ntb = ntb_max;
  if (SPWc == 10 && ntb > 2) ntb = 2;     // register budget of the 10-step variant
  if (MT > 2 && SPWc == 10) ntb = 1;      // 4-6 row tiles with 10-step chunks: accumulators + weight fragments
  if (NW == 16) ntb = 1;                  // 128 registers per lane
  if (fold && MT > 2 && ntb > 2) ntb = 2; // the statistics accumulators take 8 registers per row tile
  q.NW = NW;
  q.spw = spw;
  q.ntb = ntb;
  q.SPWc = SPWc;
  q.MT = MT;
  q.gx = (NT + ntb - 1) / ntb;
  q.gy = ksplit;
  q.gz = gz;
  q.lds = (size_t)NW * ntb * MT * 256 * 4 + (fold ? (size_t)NW * MT * 32 * 4 : 0);
  if (q.lds < 1024) q.lds = 1024;
  return q;
}

template <typename T, int MT, bool FOLD>
static void launch_skinny_mt(const SkinnyParams& p, const SkinnyPlan& q, hipStream_t s) {
  dim3 grid(q.gx, q.gy, q.gz), block(q.NW * 64);
#define ITTS_SK(SPW_, NTB_, MAXW_) hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, SPW_, NTB_, FOLD, MAXW_>), grid, block, q.lds, s, p)
  if constexpr (MT == 1 && !FOLD) {
    if (q.NW == 16) {
      ITTS_SK(10, 1, 16);
      return;
    }
  }
  if (q.SPWc == 5) {
    if (q.ntb == 1) ITTS_SK(5, 1, 8);
    else if (q.ntb == 2) ITTS_SK(5, 2, 8);
    else if (q.ntb == 3) {
      if constexpr (FOLD && MT > 2) ITTS_SK(5, 2, 8);   // (the plan never asks for 3 tiles there: register budget)
      else ITTS_SK(5, 3, 8);
    } else {
      if constexpr (FOLD && MT <= 2) ITTS_SK(5, 4, 8);
      else ITTS_SK(5, 1, 8);                            // (never planned)
    }
  } else {
    if (q.ntb == 1) ITTS_SK(10, 1, 8);
    else {
      if constexpr (MT <= 2) ITTS_SK(10, 2, 8);
      else ITTS_SK(10, 1, 8);
    }
  }
#undef ITTS_SK
}

template <typename T>
static int launch_skinny(const SkinnyParams& p, int rows_per_wg, bool wide, hipStream_t s) {
  const bool fold = p.cvec != nullptr;
  const int MTall = p.M <= 16 ? 1 : p.M <= 32 ? 2 : p.M <= 64 ? 4 : p.M <= 96 ? 6 : (p.M + 15) / 16;   // > 96: rows dealt to grid.z
  const SkinnyPlan q = plan_skinny<T>(p.N, p.K, p.ksplit, MTall, rows_per_wg, wide && !fold, fold);
  if constexpr (sizeof(T) == 4) {
    if (fold) {
      set_error("itts_gemm_skinny: the LayerNorm-folded form is built for bf16 / f16");
      return ITTS_ERR_INVALID;
    }
    launch_skinny_mt<T, 1, false>(p, q, s);
  } else {
#define ITTS_MT(MT_)                                          \
  do {                                                        \
    if (fold) launch_skinny_mt<T, MT_, true>(p, q, s);        \
    else launch_skinny_mt<T, MT_, false>(p, q, s);            \
  } while (0)
    if (q.MT == 1) ITTS_MT(1);
    else if (q.MT == 2) ITTS_MT(2);
    else if (q.MT == 4) ITTS_MT(4);
    else ITTS_MT(6);
#undef ITTS_MT
  }
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
    ITTS_REQUIRE(a->y && a->kcache && a->vcache && a->pos && a->N % 3 == 0 && a->N / 3 == a->heads * 64 &&
                     (a->kv_tab != nullptr ? (a->kv_bs == 16 || a->kv_bs == 32 || a->kv_bs == 64) : a->smax > 0),
                 "itts_gemm_skinny: bad QKV epilogue arguments (paged cache: kv_bs must be 16, 32 or 64)");
  else if (a->epi == ITTS_EPI_RESID_F32 || a->epi == ITTS_EPI_STORE_F32 || a->epi == ITTS_EPI_SLAB_F32)
    ITTS_REQUIRE(a->yf, "itts_gemm_skinny: yf is null");
  else
    ITTS_REQUIRE((a->epi == ITTS_EPI_STORE || a->epi == ITTS_EPI_GELU_STORE || a->epi == ITTS_EPI_SILU_STORE ||
                  a->epi == ITTS_EPI_RELU_AFFINE_STORE || a->epi == ITTS_EPI_RELU_AFFINE_TANH_STORE) && a->y,
                 "itts_gemm_skinny: bad epilogue %d", a->epi);
  if (a->epi == ITTS_EPI_RELU_AFFINE_STORE || a->epi == ITTS_EPI_RELU_AFFINE_TANH_STORE)
    ITTS_REQUIRE(a->post_scale && a->post_shift, "itts_gemm_skinny: the ReLU + affine epilogues need post_scale and post_shift");
  if (a->epi == ITTS_EPI_RESID_F32)
    ITTS_REQUIRE(a->N % 4 == 0 && (int64_t)a->M * a->N < (1ll << 29), "itts_gemm_skinny: the residual epilogue needs N %% 4 == 0");
  if (a->ln_c != nullptr)
    ITTS_REQUIRE(ksplit == 1 && a->bias && a->N % 4 == 0 && a->epi != ITTS_EPI_RESID_F32 && a->epi != ITTS_EPI_SLAB_F32 &&
                     a->dtype != ITTS_F32,
                 "itts_gemm_skinny: the LayerNorm-folded form needs ksplit 1, bias (= d), N %% 4 == 0, a storing epilogue, bf16 / f16");
  ITTS_REQUIRE(a->rows_per_wg == 0 || a->rows_per_wg == 16 || a->rows_per_wg == 32, "itts_gemm_skinny: rows_per_wg must be 0, 16 or 32");
  if (a->y_packed)
    ITTS_REQUIRE((a->epi == ITTS_EPI_STORE || a->epi == ITTS_EPI_GELU_STORE || a->epi == ITTS_EPI_SILU_STORE ||
                  a->epi == ITTS_EPI_RELU_AFFINE_STORE || a->epi == ITTS_EPI_RELU_AFFINE_TANH_STORE ||
                  a->epi == ITTS_EPI_RESID_F32) && a->N % ks == 0 && a->y,
                 "itts_gemm_skinny: a packed y needs the STORE / GELU_STORE / SILU_STORE / RESID_F32 epilogue and N %% %d == 0", ks);
  const int y_mtp = a->y_mtp > 0 ? a->y_mtp : (a->M + 15) / 16;
  ITTS_REQUIRE(a->y_row0 >= 0 && a->y_row0 % 16 == 0 && (!a->y_packed || a->y_row0 + a->M <= y_mtp * 16) &&
                   (a->y_packed || (a->y_row0 == 0 && a->y_mtp == 0)),
               "itts_gemm_skinny: y_row0 / y_mtp place the rows of a PACKED y inside a taller operand (y_row0 %% 16 == 0)");
  ITTS_REQUIRE(a->x_mtp == 0 || (a->x_packed && a->x_mtp * 16 >= a->M), "itts_gemm_skinny: x_mtp is for a packed x of at least M rows");
  if (a->M == 0) return ITTS_OK;
  // one launch covers 96 rows per workgroup; with rows_per_wg > 0 the row tiles are dealt to grid.z and any M runs as one launch
  const int rows_per = (a->dtype == ITTS_F32) ? 16 : (a->rows_per_wg > 0 ? (a->M > 96 ? a->M : 96) : 96);
  ITTS_REQUIRE(a->M <= 96 || a->rows_per_wg == 0 || (a->M + 15) / 16 / (a->rows_per_wg / 16) < 65535, "itts_gemm_skinny: too many rows");
  hipStream_t s = (hipStream_t)stream;
  for (int r0 = 0; r0 < a->M; r0 += rows_per) {
    SkinnyParams p;
    p.M = a->M - r0 < rows_per ? a->M - r0 : rows_per;
    p.N = a->N;
    p.K = a->K;
    p.wp = a->wp;
    p.bias = a->bias;
    p.x = a->x_packed ? (const char*)a->x : (const char*)a->x + (size_t)r0 * a->K * esz;
    p.x_pa = a->x_packed ? 1 : 0;
    p.y_pa = a->y_packed ? 1 : 0;
    p.mtp = a->x_mtp > 0 ? a->x_mtp : (a->M + 15) / 16;
    p.row0 = r0;
    p.y_mtp = y_mtp;
    p.post_scale = a->post_scale;
    p.post_shift = a->post_shift;
    p.y_row0 = r0 + a->y_row0;
    p.epi = a->epi;
    const size_t ycols = a->epi == ITTS_EPI_QKV_CACHE ? (size_t)a->N / 3 : (size_t)a->N;
    p.y = a->y ? (a->y_packed ? (char*)a->y : (char*)a->y + (size_t)r0 * ycols * esz) : nullptr;
    p.yf = a->yf ? a->yf + (size_t)r0 * a->N : nullptr;
    const size_t crow = (size_t)a->heads * a->smax * 64 * esz;
    const bool paged = a->kv_tab != nullptr;   // (the table row, not the cache pointer, carries the chunk's first row)
    p.kcache = a->kcache ? (char*)a->kcache + (paged ? 0 : (size_t)r0 * crow) : nullptr;
    p.vcache = a->vcache ? (char*)a->vcache + (paged ? 0 : (size_t)r0 * crow) : nullptr;
    p.kv_tab = a->kv_tab;
    p.kv_bs_log2 = a->kv_bs == 64 ? 6 : a->kv_bs == 32 ? 5 : 4;
    p.pos = a->pos;
    p.heads = a->heads;
    p.smax = a->smax;
    p.ksplit = ksplit;
    p.slab_rows = a->M;
    p.cvec = a->ln_c;
    p.ln_eps = a->ln_eps > 0.f ? a->ln_eps : 1e-5f;
    p.bump = r0 == 0 ? a->bump : nullptr;
#if ITTS_STAMPS
    p.stamps = g_stamp_buf;
#endif
#if ITTS_DIAG
    p.exp = g_skinny_exp;
#endif
    int rc;
    if (a->dtype == ITTS_F32) rc = launch_skinny<float>(p, a->rows_per_wg, a->wide_wg != 0, s);
    else if (a->dtype == ITTS_BF16) rc = launch_skinny<bf16_t>(p, a->rows_per_wg, a->wide_wg != 0, s);
    else if (a->dtype == ITTS_F16) rc = launch_skinny<f16_t>(p, a->rows_per_wg, a->wide_wg != 0, s);
    else ITTS_REQUIRE(false, "itts_gemm_skinny: unknown dtype %d", a->dtype);
    if (rc != ITTS_OK) return rc;
  }
  return ITTS_OK;
}

extern "C" int itts_skinny_plan(int dtype, int M, int N, int K, int ksplit, int rows_per_wg, int wide_wg, int fold, int* out8) {
  ITTS_REQUIRE(out8 && N > 0 && K > 0 && ksplit > 0 && M > 0, "itts_skinny_plan: bad arguments");
  const int MT = dtype == ITTS_F32 ? 1 : M <= 16 ? 1 : M <= 32 ? 2 : M <= 64 ? 4 : (M <= 96 || rows_per_wg == 0) ? 6 : (M + 15) / 16;
  const SkinnyPlan q = dtype == ITTS_F32 ? plan_skinny<float>(N, K, ksplit, 1, 0, false, false)
                                         : plan_skinny<bf16_t>(N, K, ksplit, MT, rows_per_wg, wide_wg != 0 && !fold, fold != 0);
  out8[0] = q.gx; out8[1] = q.gy; out8[2] = q.NW; out8[3] = q.ntb; out8[4] = q.spw; out8[5] = (int)q.lds; out8[6] = q.gz; out8[7] = q.MT;
  return ITTS_OK;
}

#if ITTS_DIAG
// ---- diagnostic build only (libindextts_hip_diag.so, include/indextts_hip_diag.h); absent from the product library
namespace itts { extern int g_conv_cfg; extern int g_attn_waves; extern int g_conv_exp; }

extern "C" int itts_debug_set(int key, int value) {
  if (key == 1) itts::g_tune_ntb = value;
  else if (key == 2) itts::g_tune_nw = value;
  else if (key == 3) itts::g_conv_cfg = value;
  else if (key == 4) itts::g_attn_waves = (value == 8) ? 8 : 4;
  else if (key == 5) itts::g_conv_exp = value;
  else if (key == 6) itts::g_skinny_exp = value;
  else return ITTS_ERR_INVALID;
  return ITTS_OK;
}

// every later itts_gemm_skinny launch writes 16 u64 per workgroup to `buf` (NULL switches it off)
extern "C" int itts_debug_stamps(void* buf) {
#if ITTS_STAMPS
  itts::g_stamp_buf = (unsigned long long*)buf;
  return ITTS_OK;
#else
  (void)buf;
  return ITTS_ERR_INVALID;
#endif
}

// the same for the tiled convolution kernel: 16 x u64 per workgroup (tools/timeline_conv.py)
namespace itts { extern unsigned long long* g_stamp_buf_conv; }
extern "C" int itts_debug_stamps_conv(void* buf) {
#if ITTS_STAMPS
  itts::g_stamp_buf_conv = (unsigned long long*)buf;
  return ITTS_OK;
#else
  (void)buf;
  return ITTS_ERR_INVALID;
#endif
}

// the same for itts_sample: 16 x u64 per batch row (see tools/timeline_sample.py for the stamp positions)
extern "C" int itts_debug_stamps_sample(void* buf) {
#if ITTS_STAMPS
  itts::g_stamp_buf_sample = (unsigned long long*)buf;
  return ITTS_OK;
#else
  (void)buf;
  return ITTS_ERR_INVALID;
#endif
}
#endif
