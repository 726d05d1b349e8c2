// On-device token selection for the decode loop (no host synchronisation per step).
// Restates the transformers 4.44.2 processors reached from indextts/gpt/model.py:710-715 in their call order:
// repetition penalty -> temperature -> top-k -> top-p -> softmax -> draw | argmax, plus the EOS/pad bookkeeping of
// GenerationMixin (finished rows keep emitting pad = stop token).  One workgroup per batch row.
#include "common.h"

namespace itts {

constexpr int SM_MAXV = 8448;   // logits staged in LDS (33 x 256)
constexpr int SM_MAXC = 1024;   // candidate cap after top-k

struct SampleParams {
  const float* logits;
  int B, V, ldl;
  int32_t* tokens;
  int32_t* history;
  int hist_cap;
  int32_t* finished;
  int32_t* state;
  const int32_t* extra_ids;
  int n_extra;
  const int32_t* force_stop;
  float rep_penalty, temperature, top_p;
  int top_k, do_sample;
  uint32_t seed_lo, seed_hi;
  int stop_token;
  float* dbg_scores;
  int advance;
  const int32_t* row_step0;
#if ITTS_STAMPS
  unsigned long long* stamps;
#endif
};

__device__ __forceinline__ uint32_t fkey(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Philox4x32-10 (Salmon et al. 2011); returns the first output word.
__device__ __forceinline__ uint32_t philox_first(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c0;
}

__global__ __launch_bounds__(256) void sample_kernel(SampleParams p) {
  __shared__ float sv[SM_MAXV];
  __shared__ uint32_t flag[SM_MAXV / 32];
  __shared__ int hist[256];
  __shared__ float cs[SM_MAXC];
  __shared__ int ci[SM_MAXC];
  __shared__ __attribute__((aligned(16))) float ss[SM_MAXC];
  __shared__ int si[SM_MAXC];
  __shared__ float rv[4];
  __shared__ int ri[4];
  __shared__ int sh_n, sh_tok, sh_keep;
  __shared__ uint32_t sh_prefix;

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = p.state[0];                                                  // loop step: the Philox counter
  const int k = kg - (p.row_step0 != nullptr ? p.row_step0[b] : 0);           // the ROW's step (a refilled slot starts at 0)
  const int V = p.V;
  const float* lg = p.logits + (int64_t)b * p.ldl;
#if ITTS_STAMPS
  unsigned long long st_[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) st_[i] = 0;
  unsigned long long* const sbuf_ = p.stamps;
  if (sbuf_ != nullptr && threadIdx.x == 0) st_[14] = __builtin_amdgcn_s_memrealtime();
#define SSTAMP(i) ITTS_STAMP_IF(sbuf_ != nullptr, i)
#else
#define SSTAMP(i) do { } while (0)
#endif
  SSTAMP(0);
  const bool forced = p.finished[b] != 0 || (p.force_stop != nullptr && p.force_stop[b] >= 0 && p.force_stop[b] <= k);

  // the row's logits are requested first (registers): their latency overlaps the bitmap construction
  float lv[SM_MAXV / 256];
#pragma unroll
  for (int i = 0; i < SM_MAXV / 256; ++i) {
    int idx = tid + i * 256;
    lv[i] = idx < V ? lg[idx] : 0.f;
  }
  // ---- raw logits -> LDS, then the repetition penalty as a pass over the HISTORY (<= k + n_extra ids, a few per thread)
  //      instead of a membership test (and, for flagged lanes' waves, an IEEE division) on each of the 8194 scores: the
  //      bitmap only de-duplicates -- the first thread to set an id's bit applies the penalty to that id's score
  for (int i = tid; i < SM_MAXV / 32; i += 256) flag[i] = 0u;
  if (tid == 0) { sh_n = 0; sh_tok = p.stop_token; sh_keep = 0; }
#pragma unroll
  for (int i = 0; i < SM_MAXV / 256; ++i) {
    int idx = tid + i * 256;
    if (idx < V) sv[idx] = lv[i];
  }
  __syncthreads();
  SSTAMP(1);
  if (p.rep_penalty != 1.0f) {
    const int nh = min(k, p.hist_cap);
    for (int i = tid; i < p.n_extra + nh; i += 256) {
      const int id = i < p.n_extra ? p.extra_ids[i] : p.history[(int64_t)b * p.hist_cap + (i - p.n_extra)];
      if (id >= 0 && id < V) {
        const uint32_t bit = 1u << (id & 31);
        if (!(atomicOr(&flag[id >> 5], bit) & bit)) {
          const float v = sv[id];
          sv[id] = v < 0.f ? v * p.rep_penalty : v / p.rep_penalty;
        }
      }
    }
    __syncthreads();
  }
  const float inv_t = (p.do_sample && p.temperature != 1.0f) ? 1.0f / p.temperature : 1.0f;
#pragma unroll
  for (int i = 0; i < SM_MAXV / 256; ++i) {   // processed scores back into registers (independent LDS reads)
    int idx = tid + i * 256;
    lv[i] = idx < V ? sv[idx] : 0.f;
  }
  if (inv_t != 1.0f) {
#pragma unroll
    for (int i = 0; i < SM_MAXV / 256; ++i) {
      int idx = tid + i * 256;
      if (idx < V) {
        lv[i] = lv[i] / p.temperature;
        sv[idx] = lv[i];
      }
    }
  }
  __syncthreads();
  SSTAMP(2);

  if (!p.do_sample) {
    // ---- greedy: argmax, lowest id on ties
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
      float v = sv[i];
      if (v > bv) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(bv, o, 64);
      int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      float fv = rv[0];
      int fi = ri[0];
      for (int w = 1; w < 4; ++w)
        if (rv[w] > fv || (rv[w] == fv && ri[w] < fi)) { fv = rv[w]; fi = ri[w]; }
      sh_tok = fi;
    }
    if (p.dbg_scores)
      for (int i = tid; i < V; i += 256) p.dbg_scores[(int64_t)b * V + i] = sv[i];
    __syncthreads();
  } else {
    int kk = p.top_k > 0 ? min(p.top_k, V) : min(V, SM_MAXC);
    if (kk > SM_MAXC) kk = SM_MAXC;
    uint32_t keys[SM_MAXV / 256];
    uint32_t kmax = 0u;
#pragma unroll
    for (int i = 0; i < SM_MAXV / 256; ++i) {
      int idx = tid + i * 256;
      keys[i] = idx < V ? fkey(lv[i]) : 0u;  // key 0 is below every real float key
      kmax = max(kmax, keys[i]);
    }
    // ---- top-k threshold.  Fast path (k <= 256): the k-th largest of the 256 per-thread maxima is a LOWER bound of the
    //      k-th largest score (k scores are at least that large), so the candidates are the few scores above it and the
    //      exact k-th value falls out of their rank sort -- two barriers instead of 32 counting rounds.  If more than
    //      SM_MAXC scores pass the bound (heavily tied logits), or k > 256, the exact bitwise bisection below decides.
    uint32_t thr = 0u;
    bool exact = false;   // thr is the exact k-th largest key (bisection) rather than a lower bound
    if (kk <= 256) {
      uint32_t* tmax = reinterpret_cast<uint32_t*>(ss);   // ss is free until the rank sort
      tmax[tid] = kmax;
      __syncthreads();
      if (wave == 0) {
        // ONE wave finds the kk-th largest of the 256 maxima by bitwise bisection over its 4 values per lane: ballots and
        // scalar popcounts only, no barrier per round (a workgroup-wide round costs ~0.25 us of barrier + LDS turnaround)
        const u32x4 m4 = *reinterpret_cast<const u32x4*>(tmax + lane * 4);
        uint32_t t = 0u;
        for (int bit = 31; bit >= 0; --bit) {
          const uint32_t cand = t | (1u << bit);
          const int cnt = __popcll(__ballot(m4[0] >= cand)) + __popcll(__ballot(m4[1] >= cand)) +
                          __popcll(__ballot(m4[2] >= cand)) + __popcll(__ballot(m4[3] >= cand));
          if (cnt >= kk) t = cand;
        }
        if (lane == 0) sh_prefix = t;
      }
      __syncthreads();
      thr = sh_prefix;
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < SM_MAXV / 256; ++i) cnt += __popcll(__ballot(keys[i] >= thr));
      if (lane == 0) hist[wave] = cnt;
      __syncthreads();
      if (hist[0] + hist[1] + hist[2] + hist[3] > SM_MAXC) thr = 0u;   // too many ties above the bound: bisect exactly
      __syncthreads();
    }
    if (thr == 0u) {
      exact = true;
      // the k-th largest order-preserving key, built bit by bit (32 counting rounds over the register-resident keys; no
      // LDS atomics -- random logits share their exponent bits, which serialises a radix histogram on a handful of bins)
      for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = thr | (1u << bit);
        // wave-wide count without cross-lane shuffles: one ballot + scalar popcount per key slot
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < SM_MAXV / 256; ++i) cnt += __popcll(__ballot(keys[i] >= cand));
        int* slot = &hist[(bit & 1) * 4];
        if (lane == 0) slot[wave] = cnt;
        __syncthreads();
        if (slot[0] + slot[1] + slot[2] + slot[3] >= kk) thr = cand;
      }
    }
    __syncthreads();
    if (tid == 0) sh_prefix = thr;
    __syncthreads();
    SSTAMP(3);
    const uint32_t kth = sh_prefix;  // key of the k-th largest score; ties with it are kept (HF: scores < kth removed)
#pragma unroll
    for (int i = 0; i < SM_MAXV / 256; ++i) {   // the keys are still in registers: only the hits touch LDS
      if (keys[i] >= kth && tid + i * 256 < V) {
        int slot = atomicAdd(&sh_n, 1);
        if (slot < SM_MAXC) { cs[slot] = lv[i]; ci[slot] = tid + i * 256; }
      }
    }
    __syncthreads();
    SSTAMP(4);
    const int n = min(sh_n, SM_MAXC);
    // ---- rank sort: descending score, ascending id
    for (int i = tid; i < n; i += 256) {
      float v = cs[i];
      int id = ci[i];
      int rank = 0;
      for (int j = 0; j < n; ++j) {
        float w = cs[j];
        rank += (w > v || (w == v && ci[j] < id)) ? 1 : 0;
      }
      ss[rank] = v;
      si[rank] = id;
    }
    __syncthreads();
    if (!exact && n > kk) {
      // candidates above the lower bound, sorted: keep the k best plus everything tied with the k-th (HF removes
      // scores < k-th value only)
      if (tid == 0) {
        int keepk = kk;
        const float kv = ss[kk - 1];
        while (keepk < n && ss[keepk] == kv) ++keepk;
        sh_n = keepk;
      }
      __syncthreads();
    }
    SSTAMP(5);
    const int n2 = min(sh_n, SM_MAXC);
    // softmax numerators relative to the maximum, one per thread (cs is reused for them); the Philox draw is computed by
    // another wave meanwhile; lane 0 then only runs the order-sensitive fp32 running sums
    for (int i = tid; i < n2; i += 256) cs[i] = expf(ss[i] - ss[0]);
    if (tid == 64) {
      // key = launch argument + the 64-bit seed held in state[4..5] (lets a captured launch serve every seed)
      const uint64_t key = (((uint64_t)p.seed_hi << 32) | p.seed_lo) + (((uint64_t)(uint32_t)p.state[5] << 32) | (uint32_t)p.state[4]);
      uint32_t x = philox_first((uint32_t)b, (uint32_t)kg, 0u, 0u, (uint32_t)key, (uint32_t)(key >> 32));
      rv[0] = (float)(x >> 8) * (1.0f / 16777216.0f);
    }
    __syncthreads();
    if (wave == 0) {
      const int n = __builtin_amdgcn_readfirstlane(n2);
      const float u = rv[0];
      const float lim = 1.0f - p.top_p;
      if (n <= 64) {
        // the order-sensitive fp32 running sums of the reference, with the numerators in registers (lane i holds
        // candidate i) and v_readlane instead of one dependent LDS read per term; every lane computes the same scalars
        const int e_bits = __float_as_int(lane < n ? cs[lane] : 0.f);
        const int id = lane < n ? si[lane] : 0;
        auto term = [&](int i) { return __int_as_float(__builtin_amdgcn_readlane(e_bits, i)); };
        float total = 0.f;
        for (int i = n - 1; i >= 0; --i) total += term(i);
        int keep = n;
        if (p.top_p < 1.0f) {
          // HF TopPLogitsWarper: ascending cumulative probability <= 1 - top_p is removed, at least one token kept
          float cum = 0.f;
          for (int i = n - 1; i >= 1; --i) {
            cum += term(i) / total;
            if (cum <= lim) keep = i; else break;
          }
        }
        float tot2 = 0.f;
        for (int i = 0; i < keep; ++i) tot2 += term(i);
        const float dthr = u * tot2;
        float run = 0.f;
        int pick_i = keep - 1;
        for (int i = 0; i < keep; ++i) {
          run += term(i);
          if (run > dthr) { pick_i = i; break; }
        }
        const int pick = __builtin_amdgcn_readlane(id, pick_i);
        if (lane == 0) { sh_keep = keep; sh_tok = pick; }
      } else if (lane == 0) {
        float total = 0.f;
        for (int i = n - 1; i >= 0; --i) total += cs[i];
        int keep = n;
        if (p.top_p < 1.0f) {
          float cum = 0.f;
          for (int i = n - 1; i >= 1; --i) {
            cum += cs[i] / total;
            if (cum <= lim) keep = i; else break;
          }
        }
        sh_keep = keep;
        float tot2 = 0.f;
        for (int i = 0; i < keep; ++i) tot2 += cs[i];
        float dthr = u * tot2, run = 0.f;
        int pick = si[keep - 1];
        for (int i = 0; i < keep; ++i) {
          run += cs[i];
          if (run > dthr) { pick = si[i]; break; }
        }
        sh_tok = pick;
      }
    }
    __syncthreads();
    if (p.dbg_scores) {
      for (int i = tid; i < V; i += 256) p.dbg_scores[(int64_t)b * V + i] = -INFINITY;
      __syncthreads();
      for (int i = tid; i < sh_keep; i += 256) p.dbg_scores[(int64_t)b * V + si[i]] = ss[i];
    }
  }

  SSTAMP(6);
  // ---- bookkeeping
  if (tid == 0) {
    int tok = forced ? p.stop_token : sh_tok;
    p.tokens[b] = tok;
    if (k < p.hist_cap) p.history[(int64_t)b * p.hist_cap + k] = tok;
    if (tok == p.stop_token && p.finished[b] == 0) {
      p.finished[b] = 1;
      atomicAdd(&p.state[2], 1);
    }
    if (p.advance) {
      // legacy protocol (advance != 0): the last row of the launch advances the loop state after every workgroup has read
      // state[0] -- one device-wide fence and one returning atomic per row
      __threadfence();
      int done = atomicAdd(&p.state[3], 1);
      if (done == p.B - 1) {
        p.state[3] = 0;
        p.state[0] = kg + 1;
        p.state[1] = p.state[1] + 1;
      }
    }
  }
#if ITTS_STAMPS
  SSTAMP(7);
  if (sbuf_ != nullptr && threadIdx.x == 0) {
    st_[15] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < 16; ++i) sbuf_[(size_t)blockIdx.x * 16 + i] = st_[i];
  }
#endif
}

}  // namespace itts

using namespace itts;

extern "C" int itts_sample(const itts_sample_args* a, void* stream) {
  ITTS_REQUIRE(a && a->logits && a->tokens && a->history && a->finished && a->state, "itts_sample: null pointer");
  ITTS_REQUIRE(a->B > 0 && a->V > 0 && a->V <= SM_MAXV && a->ldl >= a->V, "itts_sample: bad shape B=%d V=%d (max %d)", a->B, a->V, SM_MAXV);
  ITTS_REQUIRE(a->rep_penalty > 0.f && a->temperature > 0.f, "itts_sample: rep_penalty/temperature must be positive");
  ITTS_REQUIRE(a->top_k <= SM_MAXC, "itts_sample: top_k=%d exceeds %d", a->top_k, SM_MAXC);
  // the candidate store holds SM_MAXC entries: sampling from the whole vocabulary (top-k disabled) would be truncated to
  // the 1024 best logits, silently -- refuse it instead (every caller on the infer.py path passes k = 30 or 50)
  ITTS_REQUIRE(!a->do_sample || a->top_k > 0, "itts_sample: do_sample needs 1 <= top_k <= %d (top_k=%d)", SM_MAXC, a->top_k);
  SampleParams p;
  p.logits = a->logits;
  p.B = a->B;
  p.V = a->V;
  p.ldl = a->ldl;
  p.tokens = a->tokens;
  p.history = a->history;
  p.hist_cap = a->hist_cap;
  p.finished = a->finished;
  p.state = a->state;
  p.extra_ids = a->extra_ids;
  p.n_extra = a->extra_ids ? a->n_extra : 0;
  p.force_stop = a->force_stop;
  p.rep_penalty = a->rep_penalty;
  p.temperature = a->temperature;
  p.top_p = a->top_p;
  p.top_k = a->top_k;
  p.do_sample = a->do_sample;
  p.seed_lo = (uint32_t)(a->seed & 0xFFFFFFFFull);
  p.seed_hi = (uint32_t)(a->seed >> 32);
  p.stop_token = a->stop_token;
  p.dbg_scores = a->dbg_scores;
  p.advance = a->no_advance ? 0 : 1;
  p.row_step0 = a->row_step0;
#if ITTS_STAMPS
  p.stamps = itts::g_stamp_buf_sample;
#endif
  hipLaunchKernelGGL(sample_kernel, dim3(a->B), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("itts_sample");
}
