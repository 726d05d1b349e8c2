// Attention kernels for the GPT-2 decoder (20 heads x 64, scale 1/8, causal, left-pad mask).
//   attn_decode : one query per (batch row, head) over the KV cache -- HBM-bound streaming of K and V.
//   attn_prefill: whole-sequence causal attention (prefill of the decode loop, teacher-forced latent pass).
// Semantics follow HF GPT2Attention as driven by indextts/gpt/model.py:169-182: scores = q.k/sqrt(64), keys visible
// iff causal and attention_mask == 1 (left padding, model.py:643-649); softmax in fp32.
#include "common.h"

namespace itts {

constexpr int HD = 64;  // head dim

// ---------------------------------------------------------------------------------------------------------------------
// Decode: workgroup = (b, h), 4 waves.  Each lane owns a 16-byte slice of a key/value row (8 bf16 / 4 fp32 dims);
// LPR = 64/E lanes cover one row, a wave-load covers RPW = 64/LPR rows (1 KiB, contiguous).  Single pass with online
// softmax: for a chunk of CH row groups the K AND V fragments are all requested before the first use (one memory round
// trip per chunk; a typical 100-300 token context is one chunk), every lane keeps a running (max, sum, o[E]) for the
// rows it saw, and the partial states are merged across row groups (shuffles) and waves (LDS) at the end.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int AD_MAXCTX = 2048;
constexpr int AD_CH = 8;

__device__ __forceinline__ void softmax_merge(float& m, float& l, float m2, float l2, float& sa, float& sb) {
  float M = fmaxf(m, m2);
  sa = (m == -INFINITY) ? 0.f : __expf(m - M);
  sb = (m2 == -INFINITY) ? 0.f : __expf(m2 - M);
  l = l * sa + l2 * sb;
  m = M;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_decode_kernel(const T* __restrict__ q, const T* __restrict__ kc,
                                                           const T* __restrict__ vc, T* __restrict__ out,
                                                           const int32_t* __restrict__ pad, const int32_t* __restrict__ pos,
                                                           int H, int smax) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  constexpr int E = EL::E;
  constexpr int LPR = HD / E;        // lanes per row: 8 (16-bit) / 16 (fp32)
  constexpr int RPW = 64 / LPR;      // rows per wave-load: 8 / 4
  __shared__ float w_m[4], w_l[4];
  __shared__ float w_o[4][HD];
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int part = lane % LPR, rg = lane / LPR;
  const int j0 = pad[b];
  const int ctx = pos[0] + 1;  // keys [j0, ctx)
  const T* kb = kc + ((int64_t)b * H + h) * smax * HD + part * E;
  const T* vb = vc + ((int64_t)b * H + h) * smax * HD + part * E;

  float qf[E];
  {
    frag qv = ld16<frag>(q + ((int64_t)b * H + h) * HD + part * E);
#pragma unroll
    for (int e = 0; e < E; ++e) qf[e] = EL::to_f(qv[e]) * 0.125f;
  }
  float m = -INFINITY, l = 0.f, o[E];
#pragma unroll
  for (int e = 0; e < E; ++e) o[e] = 0.f;

  for (int base = j0; base < ctx; base += 4 * RPW * AD_CH) {
    frag kf[AD_CH], vf[AD_CH];
#pragma unroll
    for (int i = 0; i < AD_CH; ++i) {
      int j = base + (i * 4 + wave) * RPW + rg;
      bool ok = j < ctx;
      kf[i] = ok ? ld16<frag>(kb + (int64_t)j * HD) : zero_frag<frag>();
      vf[i] = ok ? ld16<frag>(vb + (int64_t)j * HD) : zero_frag<frag>();
    }
    float sc[AD_CH];
    float cmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < AD_CH; ++i) {
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) d = fmaf(qf[e], EL::to_f(kf[i][e]), d);
#pragma unroll
      for (int off = 1; off < LPR; off <<= 1) d += __shfl_xor(d, off, 64);
      int j = base + (i * 4 + wave) * RPW + rg;
      sc[i] = (j < ctx) ? d : -INFINITY;
      cmax = fmaxf(cmax, sc[i]);
    }
    if (cmax > -INFINITY) {
      float M = fmaxf(m, cmax);
      float corr = (m == -INFINITY) ? 0.f : __expf(m - M);
      l *= corr;
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] *= corr;
#pragma unroll
      for (int i = 0; i < AD_CH; ++i) {
        float pv = __expf(sc[i] - M);  // -inf -> 0
        l += pv;
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = fmaf(pv, EL::to_f(vf[i][e]), o[e]);
      }
      m = M;
    }
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
  if (tid < HD) {
    float M = fmaxf(fmaxf(w_m[0], w_m[1]), fmaxf(w_m[2], w_m[3]));
    float L = 0.f, acc = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float sw = (w_m[w] == -INFINITY) ? 0.f : __expf(w_m[w] - M);
      L += w_l[w] * sw;
      acc += w_o[w][tid] * sw;
    }
    out[((int64_t)b * H + h) * HD + tid] = EL::from_f(L > 0.f ? acc / L : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Prefill: workgroup = one wave = 64 consecutive queries of one (b, h); each lane owns one query row (q and the output
// accumulator live in registers, fp32).  Key/value tiles of 32 rows are staged in LDS as fp32 and broadcast-read.
// Online softmax per tile.  The workgroup whose query tile covers a key tile also writes those k/v rows to the cache.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int AP_KT = 32;

template <typename T>
__global__ __launch_bounds__(64) void attn_prefill_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                           T* __restrict__ kc, T* __restrict__ vc,
                                                           const int32_t* __restrict__ pad, int S, int H, int smax) {
  typedef Elem<T> EL;
  __shared__ __attribute__((aligned(16))) float ks[AP_KT][HD];
  __shared__ __attribute__((aligned(16))) float vs[AP_KT][HD];
  const int q0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x;
  const int D = H * HD;
  const int qi = q0 + lane;
  const bool qok = qi < S;
  const int j0 = pad ? pad[b] : 0;
  const T* base = qkv + (int64_t)b * S * 3 * D;

  float qf[HD], o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    qf[d] = qok ? EL::to_f(base[(int64_t)qi * 3 * D + h * HD + d]) * 0.125f : 0.f;
    o[d] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  const int jend = min(S, q0 + 64);  // keys needed by this query tile: [0, jend)
  const int jstart = (j0 / AP_KT) * AP_KT;
  for (int jt = (kc ? 0 : jstart); jt < jend; jt += AP_KT) {
    __syncthreads();
    // stage 32 k rows and 32 v rows (64 dims each): 2048 + 2048 elements over 64 lanes
    for (int idx = lane; idx < AP_KT * HD; idx += 64) {
      int jj = idx >> 6, d = idx & 63;
      int j = jt + jj;
      float kvv = 0.f, vvv = 0.f;
      if (j < S) {
        T kraw = base[(int64_t)j * 3 * D + D + h * HD + d];
        T vraw = base[(int64_t)j * 3 * D + 2 * D + h * HD + d];
        kvv = EL::to_f(kraw);
        vvv = EL::to_f(vraw);
        if (kc && jt >= q0 && jt < q0 + 64) {
          kc[(((int64_t)b * H + h) * smax + j) * HD + d] = kraw;
          vc[(((int64_t)b * H + h) * smax + j) * HD + d] = vraw;
        }
      }
      ks[jj][d] = kvv;
      vs[jj][d] = vvv;
    }
    __syncthreads();
    if (jt + AP_KT <= j0) continue;  // tile entirely inside the left padding (only staged for the cache write)
    float s[AP_KT];
    float tmax = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < AP_KT; ++jj) {
      float d0 = 0.f;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        f32x4 kv = *reinterpret_cast<const f32x4*>(&ks[jj][d]);
        d0 = fmaf(qf[d], kv[0], d0);
        d0 = fmaf(qf[d + 1], kv[1], d0);
        d0 = fmaf(qf[d + 2], kv[2], d0);
        d0 = fmaf(qf[d + 3], kv[3], d0);
      }
      int j = jt + jj;
      bool vis = qok && (j <= qi) && (j >= j0);
      s[jj] = vis ? d0 : -INFINITY;
      tmax = fmaxf(tmax, s[jj]);
    }
    float mn = fmaxf(m, tmax);
    if (mn > -INFINITY) {               // something is visible for this lane
      float corr = __expf(m - mn);      // m = -inf -> 0
      l *= corr;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] *= corr;
#pragma unroll
      for (int jj = 0; jj < AP_KT; ++jj) {
        float pv = __expf(s[jj] - mn);  // -inf -> 0
        l += pv;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
          f32x4 vv = *reinterpret_cast<const f32x4*>(&vs[jj][d]);
          o[d] = fmaf(pv, vv[0], o[d]);
          o[d + 1] = fmaf(pv, vv[1], o[d + 1]);
          o[d + 2] = fmaf(pv, vv[2], o[d + 2]);
          o[d + 3] = fmaf(pv, vv[3], o[d + 3]);
        }
      }
      m = mn;
    }
  }
  if (qok) {
    float inv = l > 0.f ? 1.0f / l : 0.f;
    T* orow = out + ((int64_t)b * S + qi) * D + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) orow[d] = EL::from_f(o[d] * inv);
  }
}

}  // namespace itts

using namespace itts;

extern "C" int itts_attn_decode(const void* q, const void* kcache, const void* vcache, void* out, const int32_t* pad,
                                const int32_t* pos, int B, int H, int smax, int dtype, void* stream) {
  ITTS_REQUIRE(q && kcache && vcache && out && pad && pos, "itts_attn_decode: null pointer");
  ITTS_REQUIRE(B > 0 && H > 0 && smax > 0 && smax <= AD_MAXCTX, "itts_attn_decode: bad shape B=%d H=%d smax=%d (max %d)", B, H,
               smax, AD_MAXCTX);
  dim3 grid(H, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case ITTS_F32:
      hipLaunchKernelGGL(attn_decode_kernel<float>, grid, block, 0, s, (const float*)q, (const float*)kcache, (const float*)vcache, (float*)out, pad, pos, H, smax);
      break;
    case ITTS_BF16:
      hipLaunchKernelGGL(attn_decode_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)q, (const bf16_t*)kcache, (const bf16_t*)vcache, (bf16_t*)out, pad, pos, H, smax);
      break;
    case ITTS_F16:
      hipLaunchKernelGGL(attn_decode_kernel<f16_t>, grid, block, 0, s, (const f16_t*)q, (const f16_t*)kcache, (const f16_t*)vcache, (f16_t*)out, pad, pos, H, smax);
      break;
    default:
      ITTS_REQUIRE(false, "itts_attn_decode: unknown dtype %d", dtype);
  }
  return check_launch("itts_attn_decode");
}

extern "C" int itts_attn_prefill(const void* qkv, void* out, void* kcache, void* vcache, const int32_t* pad, int B, int S,
                                 int H, int smax, int dtype, void* stream) {
  ITTS_REQUIRE(qkv && out, "itts_attn_prefill: null pointer");
  ITTS_REQUIRE((kcache == nullptr) == (vcache == nullptr), "itts_attn_prefill: pass both caches or neither");
  ITTS_REQUIRE(B > 0 && S > 0 && H > 0 && (!kcache || S <= smax), "itts_attn_prefill: bad shape B=%d S=%d H=%d smax=%d", B, S, H, smax);
  ITTS_REQUIRE(B <= 65535 && H <= 65535, "itts_attn_prefill: grid too large");
  dim3 grid((S + 63) / 64, H, B), block(64);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case ITTS_F32:
      hipLaunchKernelGGL(attn_prefill_kernel<float>, grid, block, 0, s, (const float*)qkv, (float*)out, (float*)kcache, (float*)vcache, pad, S, H, smax);
      break;
    case ITTS_BF16:
      hipLaunchKernelGGL(attn_prefill_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)qkv, (bf16_t*)out, (bf16_t*)kcache, (bf16_t*)vcache, pad, S, H, smax);
      break;
    case ITTS_F16:
      hipLaunchKernelGGL(attn_prefill_kernel<f16_t>, grid, block, 0, s, (const f16_t*)qkv, (f16_t*)out, (f16_t*)kcache, (f16_t*)vcache, pad, S, H, smax);
      break;
    default:
      ITTS_REQUIRE(false, "itts_attn_prefill: unknown dtype %d", dtype);
  }
  return check_launch("itts_attn_prefill");
}
