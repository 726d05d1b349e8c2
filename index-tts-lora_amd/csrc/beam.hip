// Beam search / beam-sample step of the decode loop (num_beams > 1), on device, no host synchronisation per step.
// Restates, for the call at indextts/gpt/model.py:710-715, transformers 4.44.2 `GenerationMixin._beam_search` +
// `BeamSearchScorer.process` + `BeamHypotheses.add/is_done` (third-party; the restatement and its own random stream are
// specified in oracle/beam_ref.py, which these kernels match token for token).  Two launches per step: one workgroup per
// ROW runs the workgroup-wide machinery of the sampling kernel over that row's logits and leaves its kept candidates in a
// scratch array; one workgroup per BATCH ELEMENT then pools its beams' candidates, draws 2*num_beams of them (or takes
// the best), and one lane does the scorer bookkeeping.
// The KV cache follows through itts_beam_kv_rows (a row table; itts_beam_reorder_kv is the copying reference form).
#include "common.h"

namespace itts {

constexpr int BM_MAXV = 8448;   // logits staged in LDS (33 x 256)
constexpr int BM_MAXC = 1024;   // candidate cap per row after top-k, and of the pool
constexpr int BM_MAXB = 8;      // num_beams cap

struct BeamParams {
  const float* logits;
  int B, nb, V, ldl;
  int32_t* tokens;
  int32_t* src;
  float* beam_scores;
  int32_t* hist;       // [2][R][cap]
  int cap;
  float* hyp_score;
  int32_t* hyp_len;
  int32_t* hyp_tok;    // [B][nb][cap]
  int32_t* n_hyp;
  float* worst;
  int32_t* done;
  int32_t* state;
  const int32_t* extra_ids;
  int n_extra;
  float rep_penalty, temperature, top_p, length_penalty;
  int top_k, do_sample;
  uint32_t seed_lo, seed_hi;
  int eos;
  float* cand_s;       // [R][BM_MAXC] phase-1 output: candidate score + running beam score
  int32_t* cand_i;     // [R][BM_MAXC]                 beam * V + token
  int32_t* cand_n;     // [R]
};

__device__ __forceinline__ uint32_t bkey(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ uint32_t philox_first3(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t k0, uint32_t k1) {
  uint32_t c3 = 0u;
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

__device__ __forceinline__ float block_max(float v, float* scratch, int tid) {
  v = wave_max(v);
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  float r = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_sum(float v, float* scratch, int tid) {
  v = wave_sum(v);
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  float r = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
  __syncthreads();
  return r;
}

// ---- Phase 1: one workgroup per ROW (batch element x beam).  log-softmax -> repetition penalty -> temperature -> top-k ->
// top-p (min_tokens_to_keep = 2) of that row; its kept candidates, sorted (score desc, id asc), go to the scratch arrays as
// (row score + running beam score, beam * V + token).  Until round 2 the pooling workgroup did its beams' rows one after
// the other (32 workgroups busy, 181 us per token at 32 x 3); the arithmetic of a row is unchanged.
__global__ __launch_bounds__(256) void beam_rows_kernel(BeamParams p) {
  __shared__ uint32_t flag[BM_MAXV / 32];
  __shared__ float cs[BM_MAXC];   // row candidates (unsorted); later their softmax numerators
  __shared__ int ci[BM_MAXC];
  __shared__ float ss[BM_MAXC];   // row candidates (sorted)
  __shared__ int si[BM_MAXC];
  __shared__ float scratch[4];
  __shared__ int cnt4[8];
  __shared__ int sh_n, sh_keep;
  __shared__ uint32_t sh_thr;

  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = p.nb, V = p.V;
  const int b = row / nb, beam = row - b * nb;
  const int k = p.state[0];                 // step = tokens generated so far per row
  const int R = p.B * nb;
  const int32_t* hin = p.hist + (int64_t)(k & 1) * R * p.cap;
  if (p.done[b] != 0) {                     // finished batch element: nothing to pool
    if (tid == 0) p.cand_n[row] = 0;
    return;
  }
  const float* lg = p.logits + (int64_t)row * p.ldl;
  // ---- repetition-penalty membership bitmap of this row (fake prefix ids + its own generated tokens)
  for (int i = tid; i < BM_MAXV / 32; i += 256) flag[i] = 0u;
  if (tid == 0) sh_n = 0;
  __syncthreads();
  if (p.rep_penalty != 1.0f) {
    for (int i = tid; i < p.n_extra; i += 256) {
      int id = p.extra_ids[i];
      if (id >= 0 && id < V) atomicOr(&flag[id >> 5], 1u << (id & 31));
    }
    const int nh = min(k, p.cap);
    for (int i = tid; i < nh; i += 256) {
      int id = hin[(int64_t)row * p.cap + i];
      if (id >= 0 && id < V) atomicOr(&flag[id >> 5], 1u << (id & 31));
    }
  }
  // ---- log-softmax (logits register-resident: one global read, per-thread partial sums in index order as before)
  float lv[BM_MAXV / 256];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < BM_MAXV / 256; ++i) {
    const int idx = tid + i * 256;
    lv[i] = idx < V ? lg[idx] : -INFINITY;
    mx = fmaxf(mx, lv[i]);
  }
  mx = block_max(mx, scratch, tid);
  float se = 0.f;
#pragma unroll
  for (int i = 0; i < BM_MAXV / 256; ++i)
    if (tid + i * 256 < V) se += expf(lv[i] - mx);
  se = block_sum(se, scratch, tid);         // its barriers also order the bitmap
  const float lse = mx + logf(se);
  const bool warp = p.do_sample != 0;
  uint32_t keys[BM_MAXV / 256];
  uint32_t kmax = 0u;
#pragma unroll
  for (int i = 0; i < BM_MAXV / 256; ++i) {
    const int idx = tid + i * 256;
    float v = lv[i] - lse;
    if (idx < V) {
      if ((flag[idx >> 5] >> (idx & 31)) & 1u) v = v < 0.f ? v * p.rep_penalty : v / p.rep_penalty;
      if (warp && p.temperature != 1.0f) v = v / p.temperature;
    }
    lv[i] = v;
    keys[i] = idx < V ? bkey(v) : 0u;       // key 0 is below every real float key
    kmax = max(kmax, keys[i]);
  }
  // ---- k-th largest key (beam search: the row's top 2*nb suffice for the global top 2*nb).  As in the sampling kernel:
  //      the kk-th largest of the 256 per-thread maxima is a LOWER bound of the kk-th largest score, one wave finds it with
  //      ballots only, and the exact kk-th (ties kept) falls out of the rank sort of the few scores above the bound; the
  //      32-round bisection over all keys (a workgroup barrier per bit) is the fallback for heavily tied rows.
  int kk = warp ? (p.top_k > 0 ? max(p.top_k, 2) : BM_MAXC) : 2 * nb;
  kk = min(min(kk, V), BM_MAXC);
  uint32_t thr = 0u;
  bool exact = false;
  if (kk <= 256) {
    uint32_t* tmax = reinterpret_cast<uint32_t*>(ss);
    tmax[tid] = kmax;
    __syncthreads();
    if (wave == 0) {
      const u32x4 m4 = *reinterpret_cast<const u32x4*>(tmax + lane * 4);
      uint32_t t = 0u;
      for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = t | (1u << bit);
        const int cnt = __popcll(__ballot(m4[0] >= cand)) + __popcll(__ballot(m4[1] >= cand)) +
                        __popcll(__ballot(m4[2] >= cand)) + __popcll(__ballot(m4[3] >= cand));
        if (cnt >= kk) t = cand;
      }
      if (lane == 0) sh_thr = t;
    }
    __syncthreads();
    thr = sh_thr;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < BM_MAXV / 256; ++i) cnt += __popcll(__ballot(keys[i] >= thr));
    if (lane == 0) cnt4[wave] = cnt;
    __syncthreads();
    if (cnt4[0] + cnt4[1] + cnt4[2] + cnt4[3] > BM_MAXC) thr = 0u;   // too many ties above the bound: bisect exactly
    __syncthreads();
  }
  if (thr == 0u) {
    exact = true;
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t cand = thr | (1u << bit);
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < BM_MAXV / 256; ++i) cnt += __popcll(__ballot(keys[i] >= cand));
      int* slot = &cnt4[(bit & 1) * 4];
      if (lane == 0) slot[wave] = cnt;
      __syncthreads();
      if (slot[0] + slot[1] + slot[2] + slot[3] >= kk) thr = cand;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < BM_MAXV / 256; ++i) {   // the keys are still in registers: only the hits touch LDS
    if (keys[i] >= thr && tid + i * 256 < V) {
      int slot = atomicAdd(&sh_n, 1);
      if (slot < BM_MAXC) { cs[slot] = lv[i]; ci[slot] = tid + i * 256; }
    }
  }
  __syncthreads();
  int n = min(sh_n, BM_MAXC);
  for (int i = tid; i < n; i += 256) {   // rank sort: descending score, ascending id
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
  if (!exact && n > kk) {   // above the lower bound, sorted: the kk best plus everything tied with the kk-th
    if (tid == 0) {
      int keepk = kk;
      const float kv = ss[kk - 1];
      while (keepk < n && ss[keepk] == kv) ++keepk;
      sh_n = keepk;
    }
    __syncthreads();
    n = min(sh_n, BM_MAXC);
  }
  // softmax numerators in parallel (same values as the serial loop computed); lane 0 keeps the order-sensitive sums
  for (int i = tid; i < n; i += 256) cs[i] = expf(ss[i] - ss[0]);
  __syncthreads();
  if (tid == 0) {
    int keep = n;
    if (warp && p.top_p < 1.0f) {
      // TopPLogitsWarper with min_tokens_to_keep = 2: ascending cumulative probability <= 1 - top_p is removed
      float total = 0.f;
      for (int i = n - 1; i >= 0; --i) total += cs[i];
      float cum = 0.f;
      const float lim = 1.0f - p.top_p;
      for (int i = n - 1; i >= 2; --i) {
        cum += cs[i] / total;
        if (cum <= lim) keep = i; else break;
      }
    }
    if (!warp) keep = min(n, 2 * nb);
    sh_keep = keep;
    p.cand_n[row] = keep;
  }
  __syncthreads();
  const int keep = sh_keep;
  const float bsc = p.beam_scores[row];
  for (int i = tid; i < keep; i += 256) {
    p.cand_s[(int64_t)row * BM_MAXC + i] = ss[i] + bsc;
    p.cand_i[(int64_t)row * BM_MAXC + i] = beam * V + si[i];
  }
}

// ---- Phase 2: one workgroup per batch element pools its beams' candidates (in beam order, capped at BM_MAXC), draws
// 2*num_beams of them without replacement (or takes the best), runs BeamSearchScorer.process and moves the histories.
__global__ __launch_bounds__(256) void beam_step_kernel(BeamParams p) {
  __shared__ float cs[BM_MAXC];   // exp weights of the pool
  __shared__ int ci[BM_MAXC];     // alive flags of the pool
  __shared__ float ss[BM_MAXC];   // sorted pool scores
  __shared__ int si[BM_MAXC];     // sorted pool flat indices
  __shared__ float pool_s[BM_MAXC];
  __shared__ int pool_i[BM_MAXC];
  __shared__ int nx_tok[BM_MAXB], nx_src[BM_MAXB];       // next beams: token, source beam
  __shared__ int add_slot[BM_MAXB], add_beam[BM_MAXB];   // hypotheses closed this step
  __shared__ int sh_nadd;
  __shared__ int picks[2 * BM_MAXB];      // drawn / taken candidates (lane 0 bookkeeping; LDS keeps it out of scratch)
  __shared__ float nx_score[BM_MAXB];
  __shared__ float sh_u[2 * BM_MAXB];     // the draws' uniforms
  __shared__ float sh_hs[BM_MAXB];        // closed-hypothesis scores of this batch element
  __shared__ int sh_nh;
  __shared__ float sh_worst;

  const int b = blockIdx.x, tid = threadIdx.x;
  const int nb = p.nb, V = p.V;
  const int k = p.state[0];                 // step = tokens generated so far per row
  const int R = p.B * nb;
  const int32_t* hin = p.hist + (int64_t)(k & 1) * R * p.cap;
  int32_t* hout = p.hist + (int64_t)((k + 1) & 1) * R * p.cap;
  const bool was_done = p.done[b] != 0;
  if (tid == 0) sh_nadd = 0;
  __syncthreads();

  if (!was_done) {
    int np = 0;
    for (int beam = 0; beam < nb; ++beam) {
      const int row = b * nb + beam;
      const int keep = min(p.cand_n[row], BM_MAXC - np);
      for (int i = tid; i < keep; i += 256) {
        pool_s[np + i] = p.cand_s[(int64_t)row * BM_MAXC + i];
        pool_i[np + i] = p.cand_i[(int64_t)row * BM_MAXC + i];
      }
      np += keep;
    }
    __syncthreads();
    // ---- pool sorted by (score desc, flat index asc) -> ss / si
    for (int i = tid; i < np; i += 256) {
      float v = pool_s[i];
      int id = pool_i[i];
      int rank = 0;
      for (int j = 0; j < np; ++j) {
        float w = pool_s[j];
        rank += (w > v || (w == v && pool_i[j] < id)) ? 1 : 0;
      }
      ss[rank] = v;
      si[rank] = id;
    }
    __syncthreads();
    if (p.do_sample) {
      // exp weights in parallel (the serial loop computed the same values one by one); the draws' uniforms and the
      // batch element's hypothesis state (global loads the bookkeeping lane would otherwise wait for) meanwhile
      for (int j = tid; j < np; j += 256) { cs[j] = expf(ss[j] - ss[0]); ci[j] = 1; }
      if (tid >= 64 && tid < 64 + 2 * BM_MAXB) {
        const int i = tid - 64;
        // key = launch argument + the 64-bit seed held in state[4..5] (lets a captured launch serve every seed)
        const uint64_t key = (((uint64_t)p.seed_hi << 32) | p.seed_lo) + (((uint64_t)(uint32_t)p.state[5] << 32) | (uint32_t)p.state[4]);
        uint32_t x = philox_first3((uint32_t)b, (uint32_t)k, (uint32_t)i, (uint32_t)key, (uint32_t)(key >> 32));
        sh_u[i] = (float)(x >> 8) * (1.0f / 16777216.0f);
      }
    }
    if (tid >= 128 && tid < 128 + BM_MAXB) sh_hs[tid - 128] = (tid - 128) < nb ? p.hyp_score[b * nb + (tid - 128)] : 0.f;
    if (tid == 192) { sh_nh = p.n_hyp[b]; sh_worst = p.worst[b]; }
    __syncthreads();
    if (tid == 0) {
      int npick = 0;
      const int want = min(2 * nb, np);
      if (p.do_sample) {
        // Draws without replacement.  A drawn candidate's weight becomes 0: adding +0 leaves the fp32 running sums exactly
        // what "skip the dead entries" gave, and both passes are straight loops the compiler can pipeline (they were
        // branchy LDS chases: 60 us per step at 32 x 3).
        for (int i = 0; i < want; ++i) {
          float total = 0.f;
          for (int j = 0; j < np; ++j) total += cs[j];
          const float thr2 = sh_u[i] * total;
          float run = 0.f;
          int pick = -1;
          for (int j = 0; j < np; ++j) {
            run += cs[j];
            pick = (pick < 0 && run > thr2) ? j : pick;   // first j whose running sum passes: necessarily a live one
          }
          if (pick < 0) {                                  // rounding left the last running sum at or below u * total
            for (int j = np - 1; j >= 0; --j)
              if (ci[j]) { pick = j; break; }
          }
          ci[pick] = 0;
          cs[pick] = 0.f;
          picks[npick++] = pick;
        }
        // torch.sort(descending) of the drawn scores; stable in draw order (insertion sort)
        for (int i = 1; i < npick; ++i) {
          int pi = picks[i];
          int j = i - 1;
          while (j >= 0 && ss[picks[j]] < ss[pi]) { picks[j + 1] = picks[j]; --j; }
          picks[j + 1] = pi;
        }
      } else {
        for (int i = 0; i < want; ++i) picks[npick++] = i;
      }
      // ---- BeamSearchScorer.process for this batch element (hypothesis scores mirrored in LDS: sh_hs)
      const float gen = (float)(k + 1);                         // cur_len - decoder_prompt_len
      const float lp_div = p.length_penalty == 0.f ? 1.f : powf(gen, p.length_penalty);
      int nxt = 0, nh = sh_nh;
      float worst = sh_worst;
      float best = -INFINITY;
      for (int i = 0; i < npick; ++i) best = fmaxf(best, ss[picks[i]]);
      for (int rank = 0; rank < npick && nxt < nb; ++rank) {
        const int flat = si[picks[rank]];
        const int beam = flat / V, tok = flat - beam * V;
        const float sc = ss[picks[rank]];
        if (tok == p.eos) {
          if (rank >= nb) continue;
          const float hs = sc / lp_div;
          if (nh < nb || hs > worst) {
            int slot;
            if (nh < nb) {
              slot = nh++;
            } else {  // evict the worst (lowest score, earliest slot)
              slot = 0;
              for (int h = 1; h < nb; ++h)
                if (sh_hs[h] < sh_hs[slot]) slot = h;
            }
            sh_hs[slot] = hs;
            p.hyp_score[b * nb + slot] = hs;
            p.hyp_len[b * nb + slot] = k;
            add_slot[sh_nadd] = slot;
            add_beam[sh_nadd] = beam;
            ++sh_nadd;
            worst = sh_hs[0];
            for (int h = 1; h < nh; ++h) worst = fminf(worst, sh_hs[h]);
          }
        } else {
          nx_tok[nxt] = tok;
          nx_src[nxt] = beam;
          nx_score[nxt] = sc;
          ++nxt;
        }
      }
      // fewer live continuations than beams (HF raises): keep the batch element alive on copies of its first beam
      for (; nxt < nb; ++nxt) {
        nx_tok[nxt] = nxt > 0 ? nx_tok[0] : p.eos;
        nx_src[nxt] = nxt > 0 ? nx_src[0] : 0;
        nx_score[nxt] = -1e9f;
      }
      p.n_hyp[b] = nh;
      p.worst[b] = worst;
      bool done = false;
      if (nh >= nb) done = worst >= best / lp_div;
      if (done) {
        p.done[b] = 1;
        atomicAdd(&p.state[2], 1);
      }
      for (int i = 0; i < nb; ++i) {
        p.tokens[b * nb + i] = nx_tok[i];
        p.src[b * nb + i] = b * nb + nx_src[i];
        p.beam_scores[b * nb + i] = nx_score[i];
      }
    }
    __syncthreads();
    // ---- closed hypotheses: copy the source beam's tokens (in closing order: a later one may reuse a slot)
    const int nadd = sh_nadd;
    for (int a = 0; a < nadd; ++a) {
      const int32_t* srow = hin + (int64_t)(b * nb + add_beam[a]) * p.cap;
      int32_t* drow = p.hyp_tok + (int64_t)(b * nb + add_slot[a]) * p.cap;
      for (int i = tid; i < min(k, p.cap); i += 256) drow[i] = srow[i];
      __syncthreads();
    }
  } else {
    // finished batch element: pad token, beam index 0 of this batch element, score 0 (BeamSearchScorer.process, done branch)
    if (tid < nb) {
      nx_tok[tid] = p.eos;
      nx_src[tid] = 0;
      p.tokens[b * nb + tid] = p.eos;
      p.src[b * nb + tid] = b * nb;
      p.beam_scores[b * nb + tid] = 0.f;
    }
    __syncthreads();
  }
  // ---- token histories follow their beams
  for (int i = 0; i < nb; ++i) {
    const int32_t* srow = hin + (int64_t)(b * nb + nx_src[i]) * p.cap;
    int32_t* drow = hout + (int64_t)(b * nb + i) * p.cap;
    for (int j = tid; j < min(k, p.cap); j += 256) drow[j] = srow[j];
    if (tid == 0 && k < p.cap) drow[k] = nx_tok[i];
  }
  // ---- step bookkeeping by the last workgroup to arrive (every workgroup has read state[0] by then)
  if (tid == 0) {
    __threadfence();
    int arrived = atomicAdd(&p.state[3], 1);
    if (arrived == p.B - 1) {
      p.state[3] = 0;
      p.state[0] = k + 1;
      p.state[1] = p.state[1] + 1;
    }
  }
}

// KV rows follow their beams WITHOUT moving a byte of the cache (GPT2InferenceModel._reorder_cache, model.py:207-218, as
// a table permutation).  tbl[parity][r][j] = physical cache row that holds position j of logical row r.  Runs right after
// beam_step_kernel (state[0] = k + 1 tokens, state[1] = P = the position the next step appends): table k+1 is written from
// table k -- positions < P come from the source beam's row, position P lives in row r itself.  One workgroup per row.
__global__ __launch_bounds__(256) void beam_kv_rows_kernel(int32_t* __restrict__ tbl, const int32_t* __restrict__ src,
                                                            const int32_t* __restrict__ state, int R, int smax) {
  const int r = blockIdx.x, k1 = state[0], P = state[1];
  const int32_t* srow = tbl + ((int64_t)((k1 + 1) & 1) * R + src[r]) * smax;
  int32_t* drow = tbl + ((int64_t)(k1 & 1) * R + r) * smax;
  const int nvalid = min(P, smax);
  for (int j = threadIdx.x; j < nvalid; j += 256) drow[j] = srow[j];
  if (threadIdx.x == 0 && P < smax) drow[P] = r;
}

// Permute the KV rows of each batch element by src (row r <- row src[r]) for cache positions [0, state[1]).
// One workgroup per (head, batch element, layer x {K,V}); every thread moves whole 16-byte chunks of ALL num_beams
// rows (loads of all sources before the first store), so the permutation is done in place.
template <typename T>
__global__ __launch_bounds__(256) void beam_reorder_kv_kernel(T* kc, T* vc, const int32_t* __restrict__ src,
                                                              const int32_t* __restrict__ state, int B, int nb, int H,
                                                              int smax, int64_t layer_stride) {
  const int h = blockIdx.x, b = blockIdx.y, lz = blockIdx.z;
  const int layer = lz >> 1;
  T* base = ((lz & 1) ? vc : kc) + (int64_t)layer * layer_stride;
  int s[BM_MAXB];
  bool ident = true;
  for (int i = 0; i < nb; ++i) {
    s[i] = src[b * nb + i];
    ident = ident && (s[i] == b * nb + i);
  }
  if (ident) return;
  const int ctx = state[1];
  const int nchunk = ctx * 64 * (int)sizeof(T) / 16;
  for (int c = threadIdx.x; c < nchunk; c += 256) {
    u32x4 v[BM_MAXB];
#pragma unroll
    for (int i = 0; i < BM_MAXB; ++i)
      if (i < nb) v[i] = *reinterpret_cast<const u32x4*>((const char*)(base + ((int64_t)s[i] * H + h) * smax * 64) + (int64_t)c * 16);
#pragma unroll
    for (int i = 0; i < BM_MAXB; ++i)
      if (i < nb) *reinterpret_cast<u32x4*>((char*)(base + ((int64_t)(b * nb + i) * H + h) * smax * 64) + (int64_t)c * 16) = v[i];
  }
}

}  // namespace itts

using namespace itts;

extern "C" int itts_beam_step(const itts_beam_args* a, void* stream) {
  ITTS_REQUIRE(a && a->logits && a->tokens && a->src && a->beam_scores && a->hist && a->hyp_score && a->hyp_len && a->hyp_tok &&
                   a->n_hyp && a->worst && a->done && a->state && a->cand_scores && a->cand_ids && a->cand_n,
               "itts_beam_step: null pointer");
  ITTS_REQUIRE(a->B > 0 && a->num_beams >= 2 && a->num_beams <= BM_MAXB, "itts_beam_step: num_beams=%d outside 2..%d", a->num_beams,
               BM_MAXB);
  ITTS_REQUIRE(a->V > 0 && a->V <= BM_MAXV && a->ldl >= a->V, "itts_beam_step: bad vocabulary V=%d (max %d)", a->V, BM_MAXV);
  ITTS_REQUIRE(a->rep_penalty > 0.f && a->temperature > 0.f && a->hist_cap > 0, "itts_beam_step: bad parameters");
  ITTS_REQUIRE(a->top_k <= BM_MAXC / BM_MAXB, "itts_beam_step: top_k=%d exceeds %d", a->top_k, BM_MAXC / BM_MAXB);
  // the candidate pool of a batch element holds BM_MAXC entries for all its beams: beam-SAMPLE over the whole vocabulary
  // (top-k disabled) would clip later beams silently -- refuse it (infer.py passes top_k = 30)
  ITTS_REQUIRE(!a->do_sample || a->top_k > 0, "itts_beam_step: do_sample needs 1 <= top_k <= %d (top_k=%d)", BM_MAXC / BM_MAXB,
               a->top_k);
  BeamParams p;
  p.logits = a->logits;
  p.B = a->B;
  p.nb = a->num_beams;
  p.V = a->V;
  p.ldl = a->ldl;
  p.tokens = a->tokens;
  p.src = a->src;
  p.beam_scores = a->beam_scores;
  p.hist = a->hist;
  p.cap = a->hist_cap;
  p.hyp_score = a->hyp_score;
  p.hyp_len = a->hyp_len;
  p.hyp_tok = a->hyp_tok;
  p.n_hyp = a->n_hyp;
  p.worst = a->worst;
  p.done = a->done;
  p.state = a->state;
  p.extra_ids = a->extra_ids;
  p.n_extra = a->extra_ids ? a->n_extra : 0;
  p.rep_penalty = a->rep_penalty;
  p.temperature = a->temperature;
  p.top_p = a->top_p;
  p.length_penalty = a->length_penalty;
  p.top_k = a->top_k;
  p.do_sample = a->do_sample;
  p.seed_lo = (uint32_t)(a->seed & 0xFFFFFFFFull);
  p.seed_hi = (uint32_t)(a->seed >> 32);
  p.eos = a->eos_token;
  p.cand_s = a->cand_scores;
  p.cand_i = a->cand_ids;
  p.cand_n = a->cand_n;

  hipLaunchKernelGGL(beam_rows_kernel, dim3(a->B * a->num_beams), dim3(256), 0, (hipStream_t)stream, p);
  hipLaunchKernelGGL(beam_step_kernel, dim3(a->B), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("itts_beam_step");
}

extern "C" int itts_beam_kv_rows(int32_t* kv_rows, const int32_t* src, const int32_t* state, int rows, int smax, void* stream) {
  ITTS_REQUIRE(kv_rows && src && state && rows > 0 && smax > 0, "itts_beam_kv_rows: bad arguments");
  hipLaunchKernelGGL(beam_kv_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, kv_rows, src, state, rows, smax);
  return check_launch("itts_beam_kv_rows");
}

extern "C" int itts_beam_reorder_kv(void* kcache, void* vcache, const int32_t* src, const int32_t* state, int layers, int B,
                                    int num_beams, int heads, int smax, int64_t layer_stride, int dtype, void* stream) {
  ITTS_REQUIRE(kcache && vcache && src && state, "itts_beam_reorder_kv: null pointer");
  ITTS_REQUIRE(layers > 0 && B > 0 && num_beams >= 2 && num_beams <= BM_MAXB && heads > 0 && smax > 0,
               "itts_beam_reorder_kv: bad shape");
  dim3 grid(heads, B, layers * 2), block(256);
  ITTS_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "itts_beam_reorder_kv: grid too large");
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case ITTS_F32:
      hipLaunchKernelGGL(beam_reorder_kv_kernel<float>, grid, block, 0, s, (float*)kcache, (float*)vcache, src, state, B, num_beams,
                         heads, smax, layer_stride);
      break;
    case ITTS_BF16:
      hipLaunchKernelGGL(beam_reorder_kv_kernel<bf16_t>, grid, block, 0, s, (bf16_t*)kcache, (bf16_t*)vcache, src, state, B,
                         num_beams, heads, smax, layer_stride);
      break;
    case ITTS_F16:
      hipLaunchKernelGGL(beam_reorder_kv_kernel<f16_t>, grid, block, 0, s, (f16_t*)kcache, (f16_t*)vcache, src, state, B, num_beams,
                         heads, smax, layer_stride);
      break;
    default:
      ITTS_REQUIRE(false, "itts_beam_reorder_kv: unknown dtype %d", dtype);
  }
  return check_launch("itts_beam_reorder_kv");
}
