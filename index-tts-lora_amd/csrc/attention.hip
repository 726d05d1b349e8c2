// Attention kernels for the GPT-2 decoder (20 heads x 64, scale 1/8, causal, left-pad mask).
//   attn_decode : one query per (batch row, head) over the KV cache -- HBM-bound streaming of K and V.
//   attn_prefill: whole-sequence causal attention (prefill of the decode loop, teacher-forced latent pass).
// Semantics follow HF GPT2Attention as driven by indextts/gpt/model.py:169-182: scores = q.k/sqrt(64), keys visible
// iff causal and attention_mask == 1 (left padding, model.py:643-649); softmax in fp32.
#include "common.h"
#include <type_traits>

namespace itts {

constexpr int HD = 64;  // head dim

// ---------------------------------------------------------------------------------------------------------------------
// Decode: workgroup = (b, h), 4 waves.  Each lane owns a 16-byte slice of a key/value row (8 bf16 / 4 fp32 dims);
// LPR = 64/E lanes cover one row, a wave-load covers RPW = 64/LPR rows (1 KiB, contiguous).  Single pass with online
// softmax: for a chunk of CH row groups the K AND V fragments are all requested before the first use (one memory round
// trip per chunk; a typical 100-300 token context is one chunk), every lane keeps a running (max, sum, o[E]) for the
// rows it saw, and the partial states are merged across row groups (shuffles) and waves (LDS) at the end.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int AD_MAXCTX = 16384;   // validation bound on cache positions per row (nothing in the kernel is sized by it)
constexpr int AD_CH = 8;

__device__ __forceinline__ void softmax_merge(float& m, float& l, float m2, float l2, float& sa, float& sb) {
  float M = fmaxf(m, m2);
  sa = (m == -INFINITY) ? 0.f : __expf(m - M);
  sb = (m2 == -INFINITY) ? 0.f : __expf(m2 - M);
  l = l * sa + l2 * sb;
  m = M;
}

// IND: beam search -- key/value position j of logical row b lives in physical cache row kv_rows[parity][b][j] (a table
// that itts_beam_step permutes instead of copying cache rows); parity = *kv_step & 1.
// PAGED: the cache is a block pool behind a per-row block table (include/indextts_hip.h, "Paged KV cache"): lane t of every
// wave keeps table entry t of the row (64 entries, requested with the query), and a key's block id comes out of that
// register with one cross-lane read per K / V request -- no dependent memory access in front of the K / V stream.
template <typename T, int NWV, bool IND, bool PAGED>
__global__ __launch_bounds__(NWV * 64) void attn_decode_kernel(const T* __restrict__ q, const T* __restrict__ kc,
                                                           const T* __restrict__ vc, T* __restrict__ out,
                                                           const int32_t* __restrict__ pad, const int32_t* __restrict__ pos,
                                                           int H, int smax, int out_mtp, const int32_t* __restrict__ kv_rows,
                                                           const int32_t* __restrict__ kv_step, int rows_total,
                                                           const int32_t* __restrict__ skip_rows,
                                                           const int32_t* __restrict__ kv_share,
                                                           const int32_t* __restrict__ kv_tab, int bs_log2) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  constexpr int E = EL::E;
  constexpr int LPR = HD / E;        // lanes per row: 8 (16-bit) / 16 (fp32)

  constexpr int RPW = 64 / LPR;      // rows per wave-load: 8 / 4
  constexpr int CH = AD_CH * 4 / NWV;   // chunks per wave: a workgroup pass always covers 4 * RPW * AD_CH keys
  __shared__ float w_m[NWV], w_l[NWV];
  __shared__ float w_o[NWV][HD];
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int part = lane % LPR, rg = lane / LPR;
  // Two memory round trips, not four: the query fragment and the three device scalars (left padding, cache position, the
  // row's stop flag) are requested together -- nothing here depends on a loaded value -- and the K / V requests follow as
  // soon as the scalars are back; the query is converted only after those have been issued.  (The first version read the
  // stop flag, branched, read pad / pos, waited for q and only then requested K / V: in-kernel latency chain of ~4 trips.)
  const frag qv = ld16<frag>(q + ((int64_t)b * H + h) * HD + part * E);
  int tabv = 0, tab0v = 0;             // PAGED: this row's block table (entry = lane), and row 0's for the shared first keys
  if constexpr (PAGED) {
    tabv = kv_tab[b * ITTS_KV_TAB + lane];
    tab0v = kv_tab[lane];
  }
  const int bsm = (1 << bs_log2) - 1;
  const int32_t* skip_ptr = skip_rows != nullptr ? skip_rows + b : pos;   // a readable word either way: no branch around the load
  const int skip_raw = *skip_ptr;
  const int j0 = pad[b];
  // rows that already emitted their stop token keep decoding only formally (the sampler pads them with the stop token
  // whatever their logits are): their attention -- the one stage whose cost grows with the rows -- is left out: no keys, no
  // requests, no store.  A select, not an early return: a branch here makes the compiler sink the other scalar loads (and the
  // query load) below it, one more round trip each.
  const bool skipped = skip_rows != nullptr && skip_raw != 0;
  // kv_share (one word, (p0 << 8) | C, or NULL / 0): the first C keys of EVERY row ([pad_b, pad_b + C): a one-prompt batch's
  // conditioning latents) hold the same bytes as cache row 0's positions [p0, p0 + C) -- read them there: 32 rows x 20 heads
  // then hit one L2-resident copy instead of streaming 32 copies from HBM (13 % of the K / V bytes at config 3's contexts).
  const int32_t* share_ptr = kv_share != nullptr ? kv_share : pos;     // (a readable word either way, as for skip_rows)
  const int share_raw = *share_ptr;
  const int share_w = (kv_share != nullptr && !IND) ? share_raw : 0;
  const int shC = share_w & 255;
  int pos0 = pos[0];
  asm volatile("" : "+s"(pos0));              // (keeps the load here: the compiler would otherwise load it only for rows that need it)
  const int ctx = skipped ? j0 : pos0 + 1;    // keys [j0, ctx)
  const T* kb = kc + (PAGED ? 0 : ((int64_t)b * H + h) * smax * HD) + part * E;
  const T* vb = vc + (PAGED ? 0 : ((int64_t)b * H + h) * smax * HD) + part * E;
  const int64_t sh_off = ((int64_t)h * smax + (share_w >> 8)) * HD + part * E;   // row 0, head h, position p0
  const int32_t* tab = nullptr;
  if constexpr (IND) tab = kv_rows + ((int64_t)(kv_step[0] & 1) * rows_total + b) * smax;

  float qf[E];
  float m = -INFINITY, l = 0.f, o[E];
#pragma unroll
  for (int e = 0; e < E; ++e) o[e] = 0.f;

  // one pass over 4 * RPW * AD_CH keys; the FIRST pass (typical contexts need no other) converts the query behind its K / V
  // requests -- written as a separate instance so that the compiler cannot hoist that conversion, and with it the wait for
  // the query, in front of the requests
  // Key groups are cut from a position that is a multiple of RPW (the left padding rounded down; the rows in front of pad_b
  // are masked): a wave-load's RPW consecutive keys then never straddle a cache block (block sizes are multiples of RPW), so
  // in the paged form the block id of a request is WAVE-UNIFORM -- one v_readlane out of the table register per request, no
  // per-lane lookup in front of the K / V stream.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto key_pass = [&](const int base, auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    frag kf[CH], vf[CH];
    int prow[CH];
    // Positions past the context are CLAMPED onto its last key (same cache lines; their scores are forced to -inf below)
    // instead of being skipped: a conditional load is a branch, and a join makes the compiler drain the memory queue.
    if constexpr (IND) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {   // the rows of this chunk's keys, all requested before the first K/V load
        int j = min(base + (i * NWV + wave) * RPW + rg, ctx - 1);
        prow[i] = tab[j];
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      // first key of this wave-load (wave-uniform); groups past the context fall onto the group of its last key, and the rows
      // past it onto that key (same cache lines, one block)
      const int jf = min(base + (i * NWV + wave_u) * RPW, (ctx - 1) & ~(RPW - 1));
      int j = min(jf + rg, ctx - 1);
      int64_t ro = (int64_t)j * HD;
      if constexpr (IND) ro += (int64_t)(prow[i] - b) * H * smax * HD;   // same head, same position, another row
      if constexpr (PAGED) {
        const int blk = __builtin_amdgcn_readlane(tabv, (jf >> bs_log2) & (ITTS_KV_TAB - 1));
        ro = ((((int64_t)blk * H + h) << bs_log2) + (j & bsm)) * HD;
      }
      const T* kp = kb + ro;
      const T* vp = vb + ro;
      if constexpr (!IND) {
        const bool sh = (unsigned)(j - j0) < (unsigned)shC;   // a select, not a branch (also false for j < j0: skipped rows)
        int64_t so = sh_off + (int64_t)(j - j0) * HD;
        if constexpr (PAGED) {
          const int ps = (share_w >> 8) + (j - j0);            // the same key in row 0's window (its block: a per-lane lookup)
          const int blk0 = __shfl(tab0v, (ps >> bs_log2) & (ITTS_KV_TAB - 1), 64);
          so = ((((int64_t)blk0 * H + h) << bs_log2) + (ps & bsm)) * HD + part * E;
        }
        kp = sh ? kc + so : kp;
        vp = sh ? vc + so : vp;
      }
      kf[i] = ld16<frag>(kp);
      vf[i] = ld16<frag>(vp);
    }
    __builtin_amdgcn_sched_barrier(0);   // every K AND V request of the pass is out before anything waits
    if constexpr (FIRST) {
#pragma unroll
      for (int e = 0; e < E; ++e) qf[e] = EL::to_f(qv[e]) * 0.125f;
      __builtin_amdgcn_sched_barrier(0);
    }
    float sc[CH];
    float cmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) d = fmaf(qf[e], EL::to_f(kf[i][e]), d);
#pragma unroll
      for (int off = 1; off < LPR; off <<= 1) d += __shfl_xor(d, off, 64);
      int j = base + (i * NWV + wave) * RPW + rg;
      sc[i] = (j >= j0 && j < ctx) ? d : -INFINITY;
      cmax = fmaxf(cmax, sc[i]);
    }
    {
      // no branch on "this lane saw a key" (the V fragments would be sunk into it and requested a round trip late): a lane
      // without keys keeps m = -inf and adds zeros
      const float M = fmaxf(m, cmax);
      const float Ms = (M == -INFINITY) ? 0.f : M;
      const float corr = (m == -INFINITY) ? 0.f : __expf(m - Ms);
      l *= corr;
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] *= corr;
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        float pv = __expf(sc[i] - Ms);  // -inf -> 0
        l += pv;
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = fmaf(pv, EL::to_f(vf[i][e]), o[e]);
      }
      m = M;
    }
  };
  constexpr int PASS = 4 * RPW * AD_CH;
  if (j0 < ctx) {
    const int b0 = j0 & ~(RPW - 1);
    key_pass(b0, std::true_type{});
    for (int base = b0 + PASS; base < ctx; base += PASS) key_pass(base, std::false_type{});
  }
  // merge across the row groups of the wave (lanes that share `part`)
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
    float sa, sb;
    softmax_merge(m, l, m2, l2, sa, sb);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float o2 = __shfl_xor(o[e], off, 64);
      o[e] = o[e] * sa + o2 * sb;
    }
  }
  if (rg == 0) {
#pragma unroll
    for (int e = 0; e < E; ++e) w_o[wave][part * E + e] = o[e];
    if (part == 0) {
      w_m[wave] = m;
      w_l[wave] = l;
    }
  }
  __syncthreads();
  if (tid < HD && !skipped) {
    float M = w_m[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) M = fmaxf(M, w_m[w]);
    float L = 0.f, acc = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
      float sw = (w_m[w] == -INFINITY) ? 0.f : __expf(w_m[w] - M);
      L += w_l[w] * sw;
      acc += w_o[w][tid] * sw;
    }
    // out_mtp > 0: packed-activation layout (the out-projection GEMM's operand), else row-major [B][H*64]
    const int64_t o = out_mtp > 0 ? pa_off<T>(b, h * HD + tid, out_mtp) : ((int64_t)b * H + h) * HD + tid;
    out[o] = EL::from_f(L > 0.f ? acc / L : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Prefill / latent pass: flash-style causal attention on the matrix cores.  Workgroup = 4 waves = 64 consecutive queries
// of one (b, h); wave w owns 16 query rows.  Per 64-key tile: K rows and a TRANSPOSED V tile are staged in LDS, S = Q K^T
// (A = Q fragments kept in registers, B = K rows: both are 16-byte row-major reads), online softmax in the accumulator
// layout (row max / sum over the 16 lanes that hold a row's columns), P goes through a per-wave LDS scratch to become an
// A operand, O += P V with B = rows of V^T.  fp32 statistics; exact-fp32 MFMA in fp32 mode.  The workgroup whose query
// tile covers a key tile also writes those k/v rows to the cache.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_prefill_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                            T* __restrict__ kc, T* __restrict__ vc,
                                                            const int32_t* __restrict__ pad, int S, int H, int smax,
                                                            const int32_t* __restrict__ row_off,
                                                            const int32_t* __restrict__ cache_shift,
                                                            const T* __restrict__ pkc, const T* __restrict__ pvc,
                                                            const int32_t* __restrict__ pre_len,
                                                            const int32_t* __restrict__ pre_row,
                                                            const int32_t* __restrict__ pre_pos0, int pre_qkv,
                                                            T* wkc, T* wvc, const int32_t* __restrict__ w_row,
                                                            const int32_t* __restrict__ w_pos0,
                                                            const int32_t* __restrict__ kv_tab, int bs_log2) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  constexpr int E = EL::E, KS = EL::KS, NKS = HD / KS;
  constexpr int ES = (int)sizeof(T);
  constexpr int RS = HD * ES + 16;   // LDS row stride in bytes (64 elements + 16 B pad)
  constexpr int CPR = HD / E;        // 16-byte chunks per row
  __shared__ __attribute__((aligned(16))) unsigned char Ks[64 * RS];
  __shared__ __attribute__((aligned(16))) unsigned char Vt[64 * RS];
  __shared__ __attribute__((aligned(16))) unsigned char Ps[4][16 * RS];
  const int q0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int D = H * HD;
  // packed form (row_off != NULL): the rows of batch element b are [row_off[b], row_off[b+1]) of qkv / out, there is no
  // padding row at all, and cache row = cache_shift[b] + local index (the left-padded cache layout of the decode loop)
  const int rbeg = row_off ? row_off[b] : b * S;
  if (row_off) S = row_off[b + 1] - rbeg;
  if (q0 >= S) return;  // workgroup-uniform
  const int cshift = (row_off && cache_shift) ? cache_shift[b] : 0;
  const int j0 = (pad && !row_off) ? pad[b] : 0;
  const T* base = qkv + (int64_t)rbeg * 3 * D;
  // prefix form (pkc != NULL, packed rows only): the element's sequence is  pl cached keys | its S rows of qkv; only the S
  // rows are queries (query q sits at sequence position pl + q), and key j < pl is read from the cache row pre_row[b] at
  // position pre_pos0[b] + j.  Key tiles are cut from sequence position 0, exactly as if the prefix rows were part of qkv.
  // pre_qkv != 0: the prefix keys are ROWS OF qkv instead (rows pre_pos0[b] .. of the packed buffer: a block that several
  // elements share, computed once in this very pass); pkc / pvc then point at the k / v thirds of qkv.
  const int pl = pkc != nullptr ? pre_len[b] : 0;
  const int64_t pstr = pre_qkv ? (int64_t)3 * D : HD;
  const int64_t pbase = pkc == nullptr ? 0
                        : pre_qkv      ? (int64_t)pre_pos0[b] * 3 * D + h * HD
                                       : (((int64_t)pre_row[b] * H + h) * smax + pre_pos0[b]) * HD;
  // (paged cache: a cached prefix key is addressed through the block table, see kv_elem_off)
  const int p_row = (pkc != nullptr && !pre_qkv) ? pre_row[b] : 0, p_pos0 = (pkc != nullptr && !pre_qkv) ? pre_pos0[b] : 0;
  const int Stot = pl + S;
  // cache append of the prefix form: this workgroup's own 64 rows go to cache row w_row[b], positions w_pos0[b] + local row
  if (wkc != nullptr) {
    const int wr = w_row[b], wp0 = w_pos0[b];
    for (int ch = tid; ch < 64 * CPR; ch += 256) {
      const int kk = ch / CPR, dc = ch - kk * CPR;
      const int q = q0 + kk;
      if (q < S) {
        const T* src = base + (int64_t)q * 3 * D + D + h * HD + dc * E;
        const int64_t o = kv_elem_off(kv_tab, bs_log2, wr, wp0 + q, H, h, smax) + dc * E;
        st16(wkc + o, ld16<frag>(src));
        st16(wvc + o, ld16<frag>(src + D));
      }
    }
  }

  frag qf[NKS];
  {
    const int row = q0 + wave * 16 + c;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      qf[ks] = (row < S) ? ld16<frag>(base + (int64_t)row * 3 * D + h * HD + ks * KS + g * E) : zero_frag<frag>();
  }
  f32x4 O[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) O[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { m[j] = -INFINITY; l[j] = 0.f; }

  const int jend = min(Stot, pl + q0 + 64);
  for (int jt = (j0 / 64) * 64; jt < jend; jt += 64) {
    __syncthreads();
    // ---- stage K rows and V^T (and append to the cache when this workgroup owns the tile)
    for (int ch = tid; ch < 64 * CPR; ch += 256) {
      const int kk = ch / CPR, dc = ch - kk * CPR;
      const int key = jt + kk;
      frag kv = zero_frag<frag>(), vv = zero_frag<frag>();
      if (key < pl) {
        const int64_t po = (kv_tab != nullptr && !pre_qkv) ? kv_elem_off(kv_tab, bs_log2, p_row, p_pos0 + key, H, h, smax) + dc * E
                                                           : pbase + (int64_t)key * pstr + dc * E;
        kv = ld16<frag>(pkc + po);
        vv = ld16<frag>(pvc + po);
      } else if (key < Stot) {
        const T* src = base + (int64_t)(key - pl) * 3 * D + D + h * HD + dc * E;
        kv = ld16<frag>(src);
        vv = ld16<frag>(src + D);
        if (kc != nullptr && jt == q0) {
          const int64_t o = kv_elem_off(kv_tab, bs_log2, b, key + cshift, H, h, smax) + dc * E;
          st16(kc + o, kv);
          st16(vc + o, vv);
        }
      }
      st16(Ks + kk * RS + dc * E * ES, kv);
#pragma unroll
      for (int e = 0; e < E; ++e) *reinterpret_cast<T*>(Vt + (dc * E + e) * RS + kk * ES) = vv[e];
    }
    __syncthreads();
    if (jt + 64 <= j0) continue;  // tile entirely inside the left padding (block-uniform)
    // ---- S = Q K^T  (per wave: 16 queries x 64 keys)
    f32x4 sacc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      sacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        frag bf = ld16<frag>(Ks + (16 * n + c) * RS + (ks * KS + g * E) * ES);
        sacc[n] = EL::mma(qf[ks], bf, sacc[n]);
      }
    }
    // ---- online softmax; this lane holds rows 4g+j (j<4), columns 16n+c (n<4)
    float rmax[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = pl + q0 + wave * 16 + 4 * g + j;   // sequence position of the query
      float mx = -INFINITY;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int key = jt + 16 * n + c;
        const bool vis = (key <= q) && (key >= j0) && (key < Stot);
        const float v = vis ? sacc[n][j] * 0.125f : -INFINITY;
        sacc[n][j] = v;
        mx = fmaxf(mx, v);
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      rmax[j] = mx;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float mn = fmaxf(m[j], rmax[j]);
      const float corr = (m[j] == -INFINITY) ? 0.f : __expf(m[j] - mn);
      float rs = 0.f;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const float pv = (sacc[n][j] == -INFINITY) ? 0.f : __expf(sacc[n][j] - mn);
        sacc[n][j] = pv;
        rs += pv;
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) rs += __shfl_xor(rs, o, 64);
      l[j] = l[j] * corr + rs;
      m[j] = mn;
#pragma unroll
      for (int n = 0; n < 4; ++n) O[n][j] *= corr;
    }
    // ---- P: accumulator layout -> A-operand layout through the wave's LDS scratch
    unsigned char* ps = Ps[wave];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<T*>(ps + (4 * g + j) * RS + (16 * n + c) * ES) = EL::from_f(sacc[n][j]);
    // same wave wrote and reads: LDS operations of a wave complete in order, no barrier needed
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      frag af = ld16<frag>(ps + c * RS + (ks * KS + g * E) * ES);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        frag bf = ld16<frag>(Vt + (16 * n + c) * RS + (ks * KS + g * E) * ES);
        O[n] = EL::mma(af, bf, O[n]);
      }
    }
  }
  // ---- normalise and store: rows 4g+j, dims 16n+c
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = q0 + wave * 16 + 4 * g + j;
    if (q >= S) continue;
    const float inv = l[j] > 0.f ? 1.0f / l[j] : 0.f;
    T* orow = out + ((int64_t)rbeg + q) * D + h * HD;
#pragma unroll
    for (int n = 0; n < 4; ++n) orow[16 * n + c] = EL::from_f(O[n][j] * inv);
  }
}

}  // namespace itts

using namespace itts;

#if ITTS_DIAG
namespace itts { int g_attn_waves = 4; }  // diagnostic build: itts_debug_set(4, 4|8) waves per decode-attention workgroup
#else
namespace itts { constexpr int g_attn_waves = 4; }  // measured equal to 8; the 8-wave instantiation folds away
#endif

extern "C" int itts_attn_decode(const void* q, const void* kcache, const void* vcache, void* out, const int32_t* pad,
                                const int32_t* pos, int B, int H, int smax, int dtype, int out_packed, const int32_t* kv_rows,
                                const int32_t* kv_step, const int32_t* skip_rows, const int32_t* kv_share, const int32_t* kv_tab,
                                int kv_bs, void* stream) {
  const int out_mtp = out_packed ? (B + 15) / 16 : 0;
  ITTS_REQUIRE(q && kcache && vcache && out && pad && pos, "itts_attn_decode: null pointer");
  const bool paged = kv_tab != nullptr;
  ITTS_REQUIRE(B > 0 && H > 0 && (paged || (smax > 0 && smax <= AD_MAXCTX)), "itts_attn_decode: bad shape B=%d H=%d smax=%d (max %d)", B, H,
               smax, AD_MAXCTX);
  ITTS_REQUIRE(!paged || ((kv_bs == 16 || kv_bs == 32 || kv_bs == 64) && kv_rows == nullptr),
               "itts_attn_decode: a paged cache needs kv_bs in {16, 32, 64} and no beam row table");
  const int bs_log2 = kv_bs == 64 ? 6 : kv_bs == 32 ? 5 : 4;
  ITTS_REQUIRE((kv_rows == nullptr) == (kv_step == nullptr), "itts_attn_decode: pass both or neither of kv_rows / kv_step");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(H, B), block(g_attn_waves * 64);
  const bool ind = kv_rows != nullptr;
#define ITTS_AD(TT_, NW_, IND_, PG_)                                                                                        \
  hipLaunchKernelGGL((attn_decode_kernel<TT_, NW_, IND_, PG_>), grid, block, 0, s, (const TT_*)q, (const TT_*)kcache,      \
                     (const TT_*)vcache, (TT_*)out, pad, pos, H, smax, out_mtp, kv_rows, kv_step, B, skip_rows, kv_share, \
                     kv_tab, bs_log2)
#define ITTS_AD_T(TT_)                                                            \
  do {                                                                            \
    if (paged) { if (g_attn_waves == 8) ITTS_AD(TT_, 8, false, true); else ITTS_AD(TT_, 4, false, true); } \
    else if (g_attn_waves == 8) { if (ind) ITTS_AD(TT_, 8, true, false); else ITTS_AD(TT_, 8, false, false); } \
    else { if (ind) ITTS_AD(TT_, 4, true, false); else ITTS_AD(TT_, 4, false, false); }          \
  } while (0)
  switch (dtype) {
    case ITTS_F32:
      ITTS_AD_T(float);
      break;
    case ITTS_BF16:
      ITTS_AD_T(bf16_t);
      break;
    case ITTS_F16:
      ITTS_AD_T(f16_t);
      break;
    default:
      ITTS_REQUIRE(false, "itts_attn_decode: unknown dtype %d", dtype);
  }
#undef ITTS_AD_T
#undef ITTS_AD
  return check_launch("itts_attn_decode");
}

static int attn_prefill_impl(const void* qkv, void* out, void* kcache, void* vcache, const int32_t* pad, int B, int S, int H,
                             int smax, int dtype, const int32_t* row_off, const int32_t* cache_shift, void* stream,
                             const void* pkc = nullptr, const void* pvc = nullptr, const int32_t* pre_len = nullptr,
                             const int32_t* pre_row = nullptr, const int32_t* pre_pos0 = nullptr, int pre_qkv = 0,
                             void* wkc = nullptr, void* wvc = nullptr, const int32_t* w_row = nullptr,
                             const int32_t* w_pos0 = nullptr, const int32_t* kv_tab = nullptr, int kv_bs = 0) {
  ITTS_REQUIRE(qkv && out, "itts_attn_prefill: null pointer");
  ITTS_REQUIRE(kv_tab == nullptr || kv_bs == 16 || kv_bs == 32 || kv_bs == 64, "itts_attn_prefill: kv_bs must be 16, 32 or 64");
  const int bs_log2 = kv_bs == 64 ? 6 : kv_bs == 32 ? 5 : 4;
  if (kv_tab != nullptr && kcache != nullptr) smax = S > smax ? S : smax;   // (no per-row capacity in the paged form)
  ITTS_REQUIRE((kcache == nullptr) == (vcache == nullptr), "itts_attn_prefill: pass both caches or neither");
  ITTS_REQUIRE(B > 0 && S > 0 && H > 0 && (!kcache || S <= smax), "itts_attn_prefill: bad shape B=%d S=%d H=%d smax=%d", B, S, H, smax);
  ITTS_REQUIRE(B <= 65535 && H <= 65535, "itts_attn_prefill: grid too large");
  dim3 grid((S + 63) / 64, H, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case ITTS_F32:
      hipLaunchKernelGGL(attn_prefill_kernel<float>, grid, block, 0, s, (const float*)qkv, (float*)out, (float*)kcache, (float*)vcache, pad, S, H, smax, row_off, cache_shift,
                         (const float*)pkc, (const float*)pvc, pre_len, pre_row, pre_pos0, pre_qkv, (float*)wkc, (float*)wvc, w_row, w_pos0, kv_tab, bs_log2);
      break;
    case ITTS_BF16:
      hipLaunchKernelGGL(attn_prefill_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)qkv, (bf16_t*)out, (bf16_t*)kcache, (bf16_t*)vcache, pad, S, H, smax, row_off, cache_shift,
                         (const bf16_t*)pkc, (const bf16_t*)pvc, pre_len, pre_row, pre_pos0, pre_qkv, (bf16_t*)wkc, (bf16_t*)wvc, w_row, w_pos0, kv_tab, bs_log2);
      break;
    case ITTS_F16:
      hipLaunchKernelGGL(attn_prefill_kernel<f16_t>, grid, block, 0, s, (const f16_t*)qkv, (f16_t*)out, (f16_t*)kcache, (f16_t*)vcache, pad, S, H, smax, row_off, cache_shift,
                         (const f16_t*)pkc, (const f16_t*)pvc, pre_len, pre_row, pre_pos0, pre_qkv, (f16_t*)wkc, (f16_t*)wvc, w_row, w_pos0, kv_tab, bs_log2);
      break;
    default:
      ITTS_REQUIRE(false, "itts_attn_prefill: unknown dtype %d", dtype);
  }
  return check_launch("itts_attn_prefill");
}

extern "C" int itts_attn_prefill(const void* qkv, void* out, void* kcache, void* vcache, const int32_t* pad, int B, int S,
                                 int H, int smax, int dtype, void* stream) {
  return attn_prefill_impl(qkv, out, kcache, vcache, pad, B, S, H, smax, dtype, nullptr, nullptr, stream);
}

extern "C" int itts_attn_prefill_prefix(const void* qkv, void* out, const void* kcache, const void* vcache, const int32_t* row_off,
                                        const int32_t* pre_len, const int32_t* pre_row, const int32_t* pre_pos0, int B, int Smax,
                                        int H, int smax, int dtype, const int32_t* kv_tab, int kv_bs, void* stream) {
  ITTS_REQUIRE(row_off && kcache && vcache && pre_len && pre_row && pre_pos0, "itts_attn_prefill_prefix: null pointer");
  return attn_prefill_impl(qkv, out, nullptr, nullptr, nullptr, B, Smax, H, smax, dtype, row_off, nullptr, stream, kcache, vcache,
                           pre_len, pre_row, pre_pos0, 0, nullptr, nullptr, nullptr, nullptr, kv_tab, kv_bs);
}

extern "C" int itts_attn_prefill_shared(const void* qkv, void* out, void* kcache, void* vcache, const int32_t* row_off,
                                        const int32_t* pre_len, const int32_t* pre_row0, const int32_t* w_row,
                                        const int32_t* w_pos0, int E, int Smax, int H, int smax, int dtype, const int32_t* kv_tab,
                                        int kv_bs, void* stream) {
  ITTS_REQUIRE(qkv && row_off && pre_len && pre_row0 && (kcache == nullptr) == (vcache == nullptr) &&
                   (kcache == nullptr || (w_row && w_pos0)), "itts_attn_prefill_shared: null pointer");
  const int D = H * 64;
  const int es = dtype == ITTS_F32 ? 4 : 2;
  const char* q = (const char*)qkv;
  return attn_prefill_impl(qkv, out, nullptr, nullptr, nullptr, E, Smax, H, smax, dtype, row_off, nullptr, stream,
                           q + (int64_t)D * es, q + (int64_t)2 * D * es, pre_len, pre_len /* unused */, pre_row0, 1, kcache, vcache,
                           w_row, w_pos0, kv_tab, kv_bs);
}

extern "C" int itts_attn_prefill_packed(const void* qkv, void* out, void* kcache, void* vcache, const int32_t* row_off,
                                        const int32_t* cache_shift, int B, int Smax, int H, int smax, int dtype,
                                        const int32_t* kv_tab, int kv_bs, void* stream) {
  ITTS_REQUIRE(row_off, "itts_attn_prefill_packed: row_off is null");
  return attn_prefill_impl(qkv, out, kcache, vcache, nullptr, B, Smax, H, smax, dtype, row_off, cache_shift, stream, nullptr, nullptr,
                           nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, kv_tab, kv_bs);
}

// ---------------------------------------------------------------------------------------------------------------
// A one-prompt batch's shared rows (the conditioning latents) are computed once by the packed prefill and land in cache row 0;
// this copies their keys / values, all layers in one launch, to the same sequence positions of every other row:
//   row b, positions [pad[b], pad[b] + C)  <-  row 0, positions [p0, p0 + C)          (b = 1 .. B-1; 16 bytes per thread)
// Both cache forms (kv_tab NULL: contiguous rows).  Replaces two advanced-indexing gathers and two scatters of the torch form.
// ---------------------------------------------------------------------------------------------------------------
namespace itts {
__global__ __launch_bounds__(256) void kv_share_rows_kernel(char* __restrict__ kc, char* __restrict__ vc, int64_t layer_bytes, int H, int C,
                                                            int p0, const int32_t* __restrict__ pad, int smax,
                                                            const int32_t* __restrict__ kv_tab, int bs_log2, int es, int total) {
  const int idx = (int)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cpr = 64 * es / 16;                        // 16-byte chunks of a 64-element row: 8 (16-bit) / 16 (fp32)
  const int seg = idx % cpr, kv = (idx / cpr) & 1, rest = idx / (2 * cpr);
  const int c = rest % C, h = rest / C;
  const int b = (int)blockIdx.y + 1;
  char* base = (kv ? vc : kc) + (int64_t)blockIdx.z * layer_bytes;
  const int64_t so = kv_elem_off(kv_tab, bs_log2, 0, p0 + c, H, h, smax) * es + seg * 16;
  const int64_t d_o = kv_elem_off(kv_tab, bs_log2, b, pad[b] + c, H, h, smax) * es + seg * 16;
  st16(base + d_o, ld16<B16>(base + so));
}
}  // namespace itts

extern "C" int itts_kv_share_rows(void* kcache, void* vcache, int layers, int64_t layer_stride, int B, int H, int C, int p0,
                                  const int32_t* pad, int smax, const int32_t* kv_tab, int kv_bs, int dtype, void* stream) {
  ITTS_REQUIRE(kcache && vcache && pad && layers > 0 && B >= 1 && H > 0 && C > 0 && p0 >= 0, "itts_kv_share_rows: bad arguments");
  ITTS_REQUIRE(kv_tab != nullptr ? (kv_bs == 16 || kv_bs == 32 || kv_bs == 64) : smax >= p0 + C,
               "itts_kv_share_rows: paged cache: kv_bs must be 16, 32 or 64; contiguous cache: smax covers the block");
  ITTS_REQUIRE(layers <= 65535 && B <= 65536, "itts_kv_share_rows: too many layers / rows");
  if (B == 1) return ITTS_OK;
  const int es = dtype == ITTS_F32 ? 4 : 2;
  const int total = H * C * 2 * (64 * es / 16);
  hipLaunchKernelGGL(itts::kv_share_rows_kernel, dim3((total + 255) / 256, B - 1, layers), dim3(256), 0, (hipStream_t)stream, (char*)kcache,
                     (char*)vcache, layer_stride * es, H, C, p0, pad, smax, kv_tab, kv_bs == 64 ? 6 : kv_bs == 32 ? 5 : 4, es, total);
  return itts::check_launch("itts_kv_share_rows");
}
