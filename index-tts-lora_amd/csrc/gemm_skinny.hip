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
// With split-K > 1 each slice stores its partial tile into its own fp32 slab [ks][M][N].  The slabs are then summed, in a
// fixed order, either by the next launch (itts_ln_reduce), or -- "reducer tail", tail_h != NULL -- inside this launch: a
// workgroup that has stored its slab signals (one add into a sharded arrival counter); workgroups 0..M-1 then wait until
// every workgroup of the launch has signalled and each turns one row into  h[row] += bias + slabs,
// y[row] = LayerNorm(h[row])  with the arithmetic of itts_ln_reduce (csrc/ln_math.h), so the consumer GEMM finds its
// normalised input ready at the kernel boundary and a transformer block needs 5 launches instead of 7.  Hand-off rules
// (CDNA inter-workgroup visibility): slab tiles are stored write-through (sc1), every storing wave drains (s_waitcnt
// vmcnt(0)), the workgroup meets at a barrier, ONE lane signals with an agent-scope atomic; a reducer's first wave polls
// the shards with sc1 loads, the workgroup meets at a barrier, and every slab byte is read with an sc1 load (never served
// from this CU's L1).  Only M workgroups ever wait and the others exit at once, so the wait cannot starve a workgroup
// that has not been dispatched yet (grids stay within one round of the CUs anyway).
// Replaces the per-step Conv1D/Linear (+ residual + LayerNorm) calls of HF GPT2Block as driven by
// indextts/gpt/model.py:163-193.
#include "common.h"
#include "ln_math.h"

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
  // reducer tail
  float* t_h;
  const float* t_bias;
  const float* t_w;
  const float* t_b;
  const float* t_w2;
  const float* t_b2;
  void* t_y;
  uint32_t* t_counter;
  const int32_t* t_epoch;
  int32_t* t_err;
  int t_acquire;
  int x_pa, y_pa, t_y_pa;  // packed-activation layout for x / y (STORE, GELU_STORE) / the tail's y
  int mtp, row0;           // row tiles of the WHOLE operand, first row of this launch (a multiple of 16)
#if ITTS_STAMPS
  unsigned long long* stamps;
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

constexpr int TAIL_SPIN_LIMIT = 1 << 22;  // ~seconds; a reducer that gives up sets *t_err (the host raises at its next sync)

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

// Reducer tail: row `row` of  h += bias + slabs ; y = LN(h)  (then LN2 when t_w2 != NULL) -- itts_ln_reduce's arithmetic.
// Two halves: the operands no other workgroup of this launch writes (residual row, bias, LayerNorm parameters) are
// requested BEFORE the reducer polls for its launch's signals; the slab rows are read after.
struct TailOps {
  f32x4 v, lw, lb, lw2, lb2, bs;
};

__device__ __forceinline__ TailOps tail_issue(const SkinnyParams& p, int row) {
  const int tid = threadIdx.x, D = p.N;
  const int o = (tid * 4 < D) ? tid * 4 : 0;
  const bool two = p.t_w2 != nullptr;
  TailOps t;
  t.v = ld16<f32x4>(p.t_h + (int64_t)row * D + o);
  t.lw = ld16<f32x4>(p.t_w + o);
  t.lb = ld16<f32x4>(p.t_b + o);
  t.lw2 = ld16<f32x4>((two ? p.t_w2 : p.t_w) + o);
  t.lb2 = ld16<f32x4>((two ? p.t_b2 : p.t_b) + o);
  t.bs = ld16<f32x4>((p.t_bias != nullptr ? p.t_bias : p.t_w) + o);
  return t;
}

template <typename T>
__device__ __forceinline__ void tail_finish(const SkinnyParams& p, float* lds, int row, const TailOps& t) {
  const int tid = threadIdx.x, nw = blockDim.x >> 6;
  const int D = p.N;
  const bool act = tid * 4 < D;
  const int o = act ? tid * 4 : 0;
  float* hr = p.t_h + (int64_t)row * D;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 v = t.v;
  f32x4 sl[4];
  {
    // sc1 loads: the slab bytes were stored write-through by other CUs during THIS launch; they must not come from L1
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        p.yf, 0, (int)((int64_t)p.ksplit * p.slab_rows * D * 4), 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned off = (i < p.ksplit && act) ? (unsigned)((((int64_t)i * p.slab_rows + row) * D + o) * 4) : 0xFFFFFFF0u;
      sl[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
    }
  }
  if (p.t_bias != nullptr) v += t.bs;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < p.ksplit) v += sl[i];   // same association order as itts_ln_reduce
  if (act) st16(hr + o, v);
  else v = zero;
  if (p.t_w2 != nullptr) wide_layernorm<true>(v, t.lw, t.lb, t.lw2, t.lb2, lds, tid, nw, D, act);
  else wide_layernorm<false>(v, t.lw, t.lb, t.lw2, t.lb2, lds, tid, nw, D, act);
  if (act) store_row4<T>((T*)p.t_y + (p.t_y_pa ? pa_off<T>(row, o, p.mtp) : (int64_t)row * D + o), v);
}

// NTB = column tiles per workgroup (grids stay within one round of the 256 CUs: a 257th workgroup costs a full second
// round for this kernel), SPW = k-steps a wave keeps in registers per pass, MT = 16-row tiles (M <= 16*MT).
template <typename T, int MT, int SPW, int NTB, bool TAIL>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(SkinnyParams p) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  constexpr int E = EL::E, KS = EL::KS;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [NW][NTB][MT][64][4]
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
  // and the epilogue issues no load of its own.
  // UPRE units per wave cover every launch with 8 waves; launches with fewer waves (tiny K) finish in a second loop.
  constexpr int UPRE = (NTB * MT + 7) / 8;
  f32x4 bias_pre[UPRE];
  int pos_pre = 0;
  unsigned epoch_pre = 0;
  {
    // range-checked dword loads: a null bias, another K slice, columns past N (the 8194-column head) read zeros -- no branch,
    // so no join at which the compiler would wait for this round trip before the weight requests go out
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias), 0, p.bias != nullptr ? p.N * 4 : 0, 0x00020000);
#pragma unroll
    for (int ui = 0; ui < UPRE; ++ui) {
      const int u = wave + ui * NW;
      const int col0 = (nt0 + u / MT) * 16 + g * 4;
      const unsigned boff = (u < NTB * MT && ks == 0) ? (unsigned)col0 * 4u : 0x80000000u;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        bias_pre[ui][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, boff + 4u * e, 0, 0));
    }
  }
  if (wave < NTB * MT && p.epi == ITTS_EPI_QKV_CACHE) pos_pre = p.pos[0];
  if constexpr (TAIL) epoch_pre = (unsigned)p.t_epoch[0];

  f32x4 acc[NTB][MT];
#pragma unroll
  for (int t = 0; t < NTB; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // One pass over SPW k-steps from `base`.  The FIRST pass always runs (a wave without a K share requests nothing and adds
  // zeros): every wave executes one straight line -- operand requests, one wait, MFMAs -- with no join in front of the
  // requests (round 3: together with the branch-free bias preload and the LDS-only barrier, 5.65 -> 5.29 us per launch
  // over a block's four GEMMs against the round-2 kernel on the same box, tools/probes/ab_r02_gemm.py).
  auto k_pass = [&](const int base) {
    frag bf[NTB][SPW];
#pragma unroll
    for (int t = 0; t < NTB; ++t)
#pragma unroll
      for (int i = 0; i < SPW; ++i) {
        int s = base + i;
        bf[t][i] = (s < s_end && nt0 + t < NTtot) ? ldw<frag>(bp + ((int64_t)t * KT + s) * 1024) : zero_frag<frag>();
      }
    if constexpr (MT <= 2) {
      frag af[SPW][MT];
#pragma unroll
      for (int i = 0; i < SPW; ++i) {
        int s = base + i;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          int row = mt * 16 + r;
          if (p.x_pa)   // one contiguous 1-KiB block per (k-step, row tile); padding rows exist and are never stored
            af[i][mt] = (s < s_end) ? ld16<frag>(X + (((int64_t)s * p.mtp + (p.row0 >> 4) + mt) * 64 + lane) * E) : zero_frag<frag>();
          else
            af[i][mt] = (s < s_end && row < p.M) ? ld16<frag>(X + (int64_t)row * p.K + s * KS + g * E) : zero_frag<frag>();
        }
      }
      ITTS_STAMP(1);
#if ITTS_STAMPS
      ITTS_STAMP_DRAIN();
      ITTS_STAMP(2);
#endif
#pragma unroll
      for (int i = 0; i < SPW; ++i) {
#pragma unroll
        for (int t = 0; t < NTB; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[t][mt] = EL::mma(bf[t][i], af[i][mt], acc[t][mt]);  // weights = A operand
      }
    } else {
      // more than 32 rows: the activation fragments (L2-resident, shared by every workgroup) are fetched row tile by row
      // tile behind the weight blocks; the unrolled loop lets the loads of tile mt+1 fly under the MFMAs of tile mt
      ITTS_STAMP(1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        frag af[SPW];
        const int row = mt * 16 + r;
#pragma unroll
        for (int i = 0; i < SPW; ++i) {
          int s = base + i;
          if (p.x_pa)
            af[i] = (s < s_end) ? ld16<frag>(X + (((int64_t)s * p.mtp + (p.row0 >> 4) + mt) * 64 + lane) * E) : zero_frag<frag>();
          else
            af[i] = (s < s_end && row < p.M) ? ld16<frag>(X + (int64_t)row * p.K + s * KS + g * E) : zero_frag<frag>();
        }
#pragma unroll
        for (int i = 0; i < SPW; ++i)
#pragma unroll
          for (int t = 0; t < NTB; ++t) acc[t][mt] = EL::mma(bf[t][i], af[i], acc[t][mt]);
      }
      ITTS_STAMP(2);
    }
  };
  k_pass(s_begin);
  for (int base = s_begin + SPW; base < s_end; base += SPW) k_pass(base);
  ITTS_STAMP(3);

  // ---- cross-wave reduction, fixed order.  Lane (g, r) of a tile holds Y[row = mt*16 + r][col = tile*16 + 4g .. 4g+3].
#pragma unroll
  for (int t = 0; t < NTB; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) st16(red + (((wave * NTB + t) * MT + mt) * 64 + lane) * 4, acc[t][mt]);
  // LDS-only wait + raw barrier (__syncthreads() would also drain vmcnt)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  ITTS_STAMP(4);
  const __amdgpu_buffer_rsrc_t rslab = __builtin_amdgcn_make_buffer_rsrc(
      p.yf, 0, TAIL ? (int)((int64_t)p.ksplit * p.slab_rows * p.N * 4) : 0, 0x00020000);
  // one output unit (column tile t, row tile mt): sum the waves' partial tiles, add the bias, apply the epilogue
  auto unit = [&](const int u, const f32x4 bs) {
    const int t = u / MT, mt = u - t * MT;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < NW; ++w) v += ld16<f32x4>(red + (((w * NTB + t) * MT + mt) * 64 + lane) * 4);
    const int row = mt * 16 + r, col0 = (nt0 + t) * 16 + g * 4;
    if (row >= p.M || col0 >= p.N) return;
    const int nval = min(4, p.N - col0);
    v += bs;
    switch (p.epi) {
      case ITTS_EPI_STORE:
        store4<T>((T*)p.y + (p.y_pa ? pa_off<T>(p.row0 + row, col0, p.mtp) : (int64_t)row * p.N + col0), v, nval);
        break;
      case ITTS_EPI_GELU_STORE: {
        f32x4 gv = {gelu_new(v[0]), gelu_new(v[1]), gelu_new(v[2]), gelu_new(v[3])};
        store4<T>((T*)p.y + (p.y_pa ? pa_off<T>(p.row0 + row, col0, p.mtp) : (int64_t)row * p.N + col0), gv, nval);
      } break;
      case ITTS_EPI_RESID_F32: {
        float* dst = p.yf + (int64_t)row * p.N + col0;
        f32x4 old = load4f(dst, nval);
        store4<float>(dst, old + v, nval);
      } break;
      case ITTS_EPI_STORE_F32:
        store4<float>(p.yf + (int64_t)row * p.N + col0, v, nval);
        break;
      case ITTS_EPI_SLAB_F32: {
        const int64_t eoff = ((int64_t)ks * p.slab_rows + row) * p.N + col0;
        if constexpr (TAIL) {
          // write-through store (aux 16 = sc1) through a wave-uniform descriptor: the tile leaves this XCD's L2 at once,
          // so no release fence is needed before the ticket (N % 4 == 0 is checked by the launcher)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rslab, (unsigned)(eoff * 4), 0, 16);
        } else {
          store4<float>(p.yf + eoff, v, nval);
        }
      } break;
      case ITTS_EPI_QKV_CACHE: {
        const int D = p.N / 3;   // a 4-column group never straddles q|k|v or a head (all multiples of 64)
        if (col0 < D) {
          store4<T>((T*)p.y + (int64_t)row * D + col0, v, nval);
        } else {
          int cc = col0 - D;
          T* cache = (T*)(cc < D ? p.kcache : p.vcache);
          if (cc >= D) cc -= D;
          const int hh = cc >> 6, dd = cc & 63;
          store4<T>(cache + (((int64_t)row * p.heads + hh) * p.smax + pos_pre) * 64 + dd, v, nval);
        }
      } break;
    }
  };
#pragma unroll
  for (int ui = 0; ui < UPRE; ++ui) {
    const int u = wave + ui * NW;
    if (u < NTB * MT) unit(u, bias_pre[ui]);
  }
  for (int u = wave + UPRE * NW; u < NTB * MT; u += NW) {   // fewer than 8 waves (tiny K): the remaining units
    f32x4 bs = {0.f, 0.f, 0.f, 0.f};
    const int col0 = (nt0 + u / MT) * 16 + g * 4;
    if (p.bias != nullptr && ks == 0 && col0 < p.N) bs = load4f(p.bias + col0, p.N - col0);
    unit(u, bs);
  }
  ITTS_STAMP(5);

  if constexpr (TAIL) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains its write-through stores ...
    __syncthreads();                                   // ... before the one lane that signals for all of them
    ITTS_STAMP(6);
    // Signal: one fire-and-forget agent-scope add into one of 8 counter shards (240 returning adds on ONE word serialise
    // at ~12 ns each at the memory side: 1.5 us median per workgroup, measured).  Reducers are STATIC: workgroup r < M owns
    // row r, so nobody needs the value the add returns, and a reducer can request its row's other operands before it polls.
    const unsigned total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (tid == 0) (void)__hip_atomic_fetch_add(p.t_counter + (lin & 7u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ITTS_STAMP(7);
    if (lin < (unsigned)p.M) {
      const TailOps ops = tail_issue(p, (int)lin);   // requested now, consumed after the wait
      __builtin_amdgcn_sched_barrier(0);
      if (wave == 0) {
        // lanes 0-7 each watch one shard: it must reach epoch * (workgroups of this launch that signal into it)
        const unsigned sh = lane & 7u;
        const unsigned want = epoch_pre * (total / 8u + (sh < (total & 7u) ? 1u : 0u));
        int spins = 0;
        for (;;) {
          const unsigned got = __hip_atomic_load(p.t_counter + sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__all((int)(got - want) >= 0)) {
            if (__any((int)(got - want) > 0) && lane == 0) atomicExch(p.t_err, 2);   // counter / epoch out of step
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          if (++spins > TAIL_SPIN_LIMIT) {
            if (lane == 0) atomicExch(p.t_err, 1);   // sticky: the host checks it at its next synchronisation and raises
            break;
          }
        }
        if (p.t_acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      ITTS_STAMP(8);
      tail_finish<T>(p, red, (int)lin, ops);
      ITTS_STAMP(9);
    }
  }
#if ITTS_STAMPS
  if (p.stamps != nullptr && threadIdx.x == 0) {
    ITTS_STAMP_DRAIN();
    unsigned long long te_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te_)::"memory");
    unsigned xcc_;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
    unsigned long long* o_ = p.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16;
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
  int NW, spw, ntb, gx, gy, SPWc;
  size_t lds;
};

#if ITTS_DIAG
int g_tune_ntb = 0, g_tune_nw = 0;  // diagnostic build: itts_debug_set(1|2, v) overrides (0 = heuristic)
#endif

template <typename T>
static SkinnyPlan plan_skinny(int N, int K, int ksplit, int MT) {
  constexpr int KS = Elem<T>::KS;
  SkinnyPlan q;
  const int KT = K / KS;
  const int SB = (KT + ksplit - 1) / ksplit;
  // waves per workgroup: 8 whenever the slice has 8 k-steps (measured: the split-K 3 out-projection, 14 k-steps, takes
  // 3.8 us with 8 waves x 2 steps against 4.4 us with 3 waves x 5; more than 8 waves change nothing)
  int NW = SB > 8 ? 8 : SB;
  if (NW < 1) NW = 1;
#if ITTS_DIAG
  if (g_tune_nw > 0) NW = g_tune_nw > 8 ? 8 : g_tune_nw;
#endif
  const int spw = (SB + NW - 1) / NW;
  const int NT = (N + 15) / 16;
  int ntb = (NT * ksplit + 255) / 256;   // keep the grid within one round of the 256 CUs
#if ITTS_DIAG
  if (g_tune_ntb > 0) ntb = g_tune_ntb;
#endif
  if (ntb > 3) ntb = 3;
  const int SPWc = spw <= 5 ? 5 : 10;     // register-chunk variant
  if (SPWc == 10 && ntb > 2) ntb = 2;     // register budget of the 10-step variant
  if (MT > 2 && SPWc == 10) ntb = 1;      // 4-6 row tiles with 10-step chunks: accumulators + weight fragments
  q.NW = NW;
  q.spw = spw;
  q.ntb = ntb;
  q.SPWc = SPWc;
  q.gx = (NT + ntb - 1) / ntb;
  q.gy = ksplit;
  q.lds = (size_t)NW * ntb * MT * 256 * 4;
  if (q.lds < 1024) q.lds = 1024;
  return q;
}

template <typename T, int MT>
static int launch_skinny(const SkinnyParams& p, hipStream_t s) {
  const SkinnyPlan q = plan_skinny<T>(p.N, p.K, p.ksplit, MT);
  dim3 grid(q.gx, q.gy), block(q.NW * 64);
  const bool tail = p.t_h != nullptr;
  if (tail) {
    if (p.epi != ITTS_EPI_SLAB_F32 || q.NW * 64 * 4 < p.N || p.N % 4 != 0 || p.ksplit > 4 || (int)(grid.x * grid.y) < p.M ||
        grid.x * grid.y > 256) {
      set_error("itts_gemm_skinny: the reducer tail needs the slab epilogue, N %% 4 == 0, N <= %d, ksplit <= 4 and M <= workgroups <= 256 "
                "(N=%d, workgroups=%d, M=%d)", q.NW * 256, p.N, (int)(grid.x * grid.y), p.M);
      return ITTS_ERR_INVALID;
    }
  }
#define ITTS_SK(SPW_, NTB_)                                                                                            \
  do {                                                                                                                 \
    if (tail) hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, SPW_, NTB_, true>), grid, block, q.lds, s, p);              \
    else hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, SPW_, NTB_, false>), grid, block, q.lds, s, p);                  \
  } while (0)
  if (q.SPWc == 5) {
    if (q.ntb == 1) ITTS_SK(5, 1);
    else if (q.ntb == 2) ITTS_SK(5, 2);
    else ITTS_SK(5, 3);
  } else {
    if (q.ntb == 1) ITTS_SK(10, 1);
    else {
      if constexpr (MT <= 2) ITTS_SK(10, 2);
      else ITTS_SK(10, 1);
    }
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
  const int rows_per = (a->dtype == ITTS_F32) ? 16 : 96;
  if (a->tail_h != nullptr)
    ITTS_REQUIRE(a->tail_w && a->tail_b && a->tail_y && a->tail_counter && a->tail_epoch && a->tail_err &&
                     (a->tail_w2 == nullptr) == (a->tail_b2 == nullptr) && a->M <= rows_per,
                 "itts_gemm_skinny: bad reducer-tail arguments (needs M <= %d rows in one launch)", rows_per);
  if (a->y_packed)
    ITTS_REQUIRE((a->epi == ITTS_EPI_STORE || a->epi == ITTS_EPI_GELU_STORE) && a->N % ks == 0,
                 "itts_gemm_skinny: a packed y needs the STORE / GELU_STORE epilogue and N %% %d == 0", ks);
  if (a->tail_y_packed) ITTS_REQUIRE(a->tail_h && a->N % ks == 0, "itts_gemm_skinny: packed tail_y needs a tail and N %% %d == 0", ks);
  if (a->M == 0) return ITTS_OK;
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
    p.t_y_pa = a->tail_y_packed ? 1 : 0;
    p.mtp = (a->M + 15) / 16;
    p.row0 = r0;
    p.epi = a->epi;
    const size_t ycols = a->epi == ITTS_EPI_QKV_CACHE ? (size_t)a->N / 3 : (size_t)a->N;
    p.y = a->y ? (a->y_packed ? (char*)a->y : (char*)a->y + (size_t)r0 * ycols * esz) : nullptr;
    p.yf = a->yf ? a->yf + (size_t)r0 * a->N : nullptr;
    const size_t crow = (size_t)a->heads * a->smax * 64 * esz;
    p.kcache = a->kcache ? (char*)a->kcache + (size_t)r0 * crow : nullptr;
    p.vcache = a->vcache ? (char*)a->vcache + (size_t)r0 * crow : nullptr;
    p.pos = a->pos;
    p.heads = a->heads;
    p.smax = a->smax;
    p.ksplit = ksplit;
    p.slab_rows = a->M;
    p.t_h = a->tail_h;
    p.t_bias = a->tail_bias;
    p.t_w = a->tail_w;
    p.t_b = a->tail_b;
    p.t_w2 = a->tail_w2;
    p.t_b2 = a->tail_b2;
    p.t_y = a->tail_y;
    p.t_counter = (uint32_t*)a->tail_counter;
    p.t_epoch = a->tail_epoch;
    p.t_err = a->tail_err;
    p.t_acquire = a->tail_acquire;
#if ITTS_STAMPS
    p.stamps = g_stamp_buf;
#endif
    int rc;
    if (a->dtype == ITTS_F32) {
      rc = launch_skinny<float, 1>(p, s);
    } else if (a->dtype == ITTS_BF16) {
      rc = p.M <= 16 ? launch_skinny<bf16_t, 1>(p, s) : p.M <= 32 ? launch_skinny<bf16_t, 2>(p, s)
           : p.M <= 64 ? launch_skinny<bf16_t, 4>(p, s) : launch_skinny<bf16_t, 6>(p, s);
    } else if (a->dtype == ITTS_F16) {
      rc = p.M <= 16 ? launch_skinny<f16_t, 1>(p, s) : p.M <= 32 ? launch_skinny<f16_t, 2>(p, s)
           : p.M <= 64 ? launch_skinny<f16_t, 4>(p, s) : launch_skinny<f16_t, 6>(p, s);
    } else {
      ITTS_REQUIRE(false, "itts_gemm_skinny: unknown dtype %d", a->dtype);
    }
    if (rc != ITTS_OK) return rc;
  }
  return ITTS_OK;
}

extern "C" int itts_skinny_plan(int dtype, int M, int N, int K, int ksplit, int* out6) {
  ITTS_REQUIRE(out6 && N > 0 && K > 0 && ksplit > 0 && M > 0, "itts_skinny_plan: bad arguments");
  const int MT = M <= 16 ? 1 : M <= 32 ? 2 : M <= 64 ? 4 : 6;
  const SkinnyPlan q = dtype == ITTS_F32 ? plan_skinny<float>(N, K, ksplit, 1) : plan_skinny<bf16_t>(N, K, ksplit, MT);
  out6[0] = q.gx; out6[1] = q.gy; out6[2] = q.NW; out6[3] = q.ntb; out6[4] = q.spw; out6[5] = (int)q.lds;
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
