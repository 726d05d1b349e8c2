// Bandwidth-bound kernels: anti-aliased SnakeBeta activation, LayerNorm, embedding step, tanh -> PCM16.
#include "common.h"
#include "ln_math.h"
#include "aa_tile.h"
#include <mutex>

#ifndef ITTS_AA_F16_MFMA
#define ITTS_AA_F16_MFMA 1   // build-time A/B: fp16 activation with both FIRs on the matrix cores (0: the VALU form for every type)
#endif

namespace itts {


// -------------------------------------------------------------------------------------------------------------------
// Anti-aliased SnakeBeta, channels-last [B][T][C].
// Follows alias_free_torch/act.py:10-28: UpSample1d (resample.py:10-35) -> SnakeBeta (activations.py:63-122) ->
// DownSample1d/LowPassFilter1d (resample.py:38-48, filter.py:60-95); the fused form is the one of
// anti_alias_activation_cuda.cu:44-181, re-derived in polyphase form:
//   u[2q]   = 2 * sum_{d=-3..2} x[q+d] * up[5-2d]        u[2q+1] = 2 * sum_{d=-2..3} x[q+d] * up[6-2d]   (x index clamped)
//   s[m]    = u[m] + sin^2(u[m] * e^alpha) / (e^beta + 1e-9)
//   y[t]    = sum_{j<12} down[j] * s[clamp(2t + j - 5, 0, 2T-1)]
// One workgroup = TT output rows x CS channels of one batch element: x tile (TT+10 rows) and the 2TT+10 intermediate
// rows are staged in LDS as fp32, so every input element is read from HBM once (the CUDA original re-reads a 44-element
// window per thread from global memory).
// -------------------------------------------------------------------------------------------------------------------
// Work is vectorised over 4 adjacent channels everywhere (8/16-byte LDS and global accesses) and register-blocked over
// AA_R consecutive rows, so that a thread re-uses the FIR windows it has read (the un-blocked form was LDS-bandwidth
// bound: 19 16-byte LDS reads per 4 outputs against 7 here):
//   x tile  rows  t0-6 .. t0+TT+5            (XR = TT+12, replicate-clamped at load, kept in the storage type T)
//   s tile  rows  m = 2*t0-6 .. 2*t0+2*TT+5  (SR = 2*TT+12, fp32), row pair (2i, 2i+1) from x rows i .. i+6
//   y[t0+tt] = sum_j down[j] * s_tile[2*tt + 1 + j]
// The tile computation lives in aa_tile.h (shared with the fused activation+convolution kernel).
template <typename T, int CS>
__global__ __launch_bounds__(256) void aa_snake_btc_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                            const float* __restrict__ alpha_log,
                                                            const float* __restrict__ beta_log, Fir24 f, int T_full, int C,
                                                            const int32_t* __restrict__ valid_rows) {
  constexpr int TT = AaTile<CS>::TT;
  typedef AaShape<CS, TT> SH;
  typedef T t4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) T xs[SH::XRP * CS];
  __shared__ __attribute__((aligned(16))) float ss[SH::SR * CS];
  __shared__ __attribute__((aligned(16))) float ca[CS];
  __shared__ __attribute__((aligned(16))) float cb[CS];
  const int t0 = blockIdx.x * TT;
  const int c0 = blockIdx.y * CS;
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  // ragged batches: this batch element's sequence ends at valid_rows[b] -- the replicate padding of the two filters clamps
  // there, exactly as if the element were processed alone; tiles past its end are not computed
  const int T_len = valid_rows != nullptr ? min(max(valid_rows[b], 0), T_full) : T_full;
  if (t0 >= T_len) return;
  // batch stride below uses T_full
  const T* xb = x + (int64_t)b * T_full * C + c0;
  T* yb = y + (int64_t)b * T_full * C + c0;
  if (tid < CS) {
    ca[tid] = expf(alpha_log[c0 + tid]);
    cb[tid] = 1.0f / (expf(beta_log[c0 + tid]) + 1e-9f);
  }
  aa_tile<T, CS, TT>(xb, C, T_len, t0, f, xs, ss, ca, cb, [&](int tt, int c4, const f32x4& acc) {
    const int t = t0 + tt;
    if (t < T_len) {
      t4 o = {Elem<T>::from_f(acc[0]), Elem<T>::from_f(acc[1]), Elem<T>::from_f(acc[2]), Elem<T>::from_f(acc[3])};
      *reinterpret_cast<t4*>(yb + (int64_t)t * C + c4 * 4) = o;
    }
  });
}

// -------------------------------------------------------------------------------------------------------------------
// The same activation with BOTH FIRs on the matrix cores (fp16 storage; round 4).  The VALU form above spends ~54
// instructions per output element, two thirds of them on the two 12-tap filters; as banded-matrix products
//   u[c][m]  = sum_k  x[c][k] Wup[k][m]          (16 channels x 16 u rows x 32 x rows per MFMA, 6 taps per u row)
//   y[c][t]  = sum_k  s[c][k] Wdn[k][t]          (16 channels x 16 outputs x 64 s rows: two MFMAs, 12 taps per output)
// the filters cost 2 + 4 v_mfma_f32_16x16x32_f16 per 16 x 16 block (each tap matrix is split into fp16 hi + lo parts, so the
// TAPS stay exact to ~2^-22; the products of fp16 values accumulate in fp32) and the VALU keeps only the snake term.  One
// rounding is added: s goes through LDS as fp16 (2^-11 relative) -- the storage type's own resolution.
// Both LDS images are ROW-MAJOR [time][channel] (8-byte stores on the way in), and the MFMA contraction runs over TIME: the
// signal fragments (A operand: 16 channels x 8 consecutive rows per lane group) come out of LDS through the hardware
// transpose read ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group, delivered column-major); the banded tap matrices
// are constant B fragments built once per workgroup.  The accumulator holds 4 consecutive CHANNELS of one row per lane: the
// snake output goes back to LDS, and the result to the channels-last tensor, as 8-byte vectors.
// Follows alias_free_torch/act.py:10-28 like aa_tile.h; replaces anti_alias_activation_cuda.cu:44-181 for 16-bit tensors.
// -------------------------------------------------------------------------------------------------------------------
template <int NCB>   // 16-channel blocks per workgroup (CS = 16 * NCB channels)
struct AaMfma {
  static constexpr int TT = 128;                 // output rows per tile
  static constexpr int CS = 16 * NCB;
  static constexpr int XR = TT + 32;             // X image rows: x rows t0-12 .. t0+TT+19 (8-aligned 32-row windows of the u blocks)
  static constexpr int NUB = (2 * TT + 16) / 16; // u row blocks: u rows 2*t0-8 .. 2*t0+2*TT+7
  static constexpr int SR = 2 * TT + 16 + 16;    // S image rows (+16: the last y block's 64-row window runs past the u rows; zero taps there)
  static constexpr int RS = CS + 8;              // row stride of both images (elements)
  static constexpr int NYB = TT / 16;            // y row blocks
  static constexpr size_t LDS = (size_t)(XR + SR) * RS * 2;
};

typedef short aa_v4s __attribute__((__vector_size__(4 * sizeof(short))));
// fragment of a [rows][RS] fp16 image for the MFMA A operand "16 channels x 32 rows": lane (g, r) gets rows row0 + 8g .. +7 of
// channel col0 + r.  Two transposed reads; lane 4q+p of a 16-lane group addresses row q, columns 4p .. 4p+3 of the block.
template <int RS>
__device__ __forceinline__ f16x8 aa_tr_frag(const f16_t* img, int row0, int col0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
  const f16_t* a = img + (row0 + 8 * g + q) * RS + col0 + 4 * pq;
  typedef __attribute__((address_space(3))) aa_v4s* lptr;
  const aa_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a));
  const aa_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a + 4 * RS));
  typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
  const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(f16x8, both);
}

// PAIR (C = 24 only, dense batches): the 48-channel slice holds the 24 channels of TWO batch elements side by side (b = 2 z, 2 z + 1:
// same rows, same filters) -- three full 16-channel MFMA blocks instead of 2 x (one full + one half-empty) blocks.
template <int NCB, bool PAIR = false>
__global__ __launch_bounds__(256) void aa_snake_mfma_kernel(const f16_t* __restrict__ x, f16_t* __restrict__ y,
                                                             const float* __restrict__ alpha_log,
                                                             const float* __restrict__ beta_log, Fir24 f, int T_full, int C,
                                                             const int32_t* __restrict__ valid_rows, int tiles_per_wg) {
  typedef AaMfma<NCB> G;
  constexpr int TT = G::TT, CS = G::CS, XR = G::XR, NUB = G::NUB, RS = G::RS, NYB = G::NYB;
  typedef f16_t h4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char aa_lds[];
  f16_t* Xi = reinterpret_cast<f16_t*>(aa_lds);          // [XR][RS]
  f16_t* Si = Xi + XR * RS;                              // [SR][RS]
  __shared__ float taps[24];
  const int c0 = PAIR ? 0 : blockIdx.y * CS;
  const int b = PAIR ? 2 * (int)blockIdx.z : (int)blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int T_len = (!PAIR && valid_rows != nullptr) ? min(max(valid_rows[b], 0), T_full) : T_full;
  if ((int)blockIdx.x * tiles_per_wg * TT >= T_len) return;
  const f16_t* xb = x + (int64_t)b * T_full * C;
  f16_t* yb = y + (int64_t)b * T_full * C;
  if (tid < 12) taps[tid] = 2.0f * f.up[tid];      // (the upsampler's gain of 2 rides on its taps)
  else if (tid < 24) taps[tid] = f.down[tid - 12];
  __syncthreads();

  // ---- constant B fragments (lane (g, r): column r, k = 8g + e).
  // Wup, columns = u rows m0 + r, k = x rows k0 + kk with k0 = q0 - 8, q0 = m0 / 2:
  //   u[2q] = 2 sum_{d=-3..2} x[q+d] up[5-2d],  u[2q+1] = 2 sum_{d=-2..3} x[q+d] up[6-2d]   (aa_tile.h)  ->  d = kk - 8 - (r >> 1)
  // Wdn, columns = outputs t0' + r, k = s rows ks0 + kk with ks0 = 2 t0' - 8:
  //   y[t] = sum_j down[j] s[2t + j - 5]  ->  j = kk - 2 r - 3
  f16x8 wu_hi, wu_lo, wd_hi[2], wd_lo[2];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int kk = 8 * g + e;
    const int d = kk - 8 - (r >> 1);
    const int ju = (r & 1) ? 6 - 2 * d : 5 - 2 * d;        // tap index: valid for d in [-2, 3] (odd rows) / [-3, 2] (even rows)
    const float w = (ju >= 0 && ju < 12) ? taps[min(max(ju, 0), 11)] : 0.f;
    const f16_t hi = (f16_t)w;
    wu_hi[e] = hi;
    wu_lo[e] = (f16_t)(w - (float)hi);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int j = 32 * st + kk - 2 * r - 3;
      const float wd = (j >= 0 && j < 12) ? taps[12 + min(max(j, 0), 11)] : 0.f;
      const f16_t dh = (f16_t)wd;
      wd_hi[st][e] = dh;
      wd_lo[st][e] = (f16_t)(wd - (float)dh);
    }
  }
  // snake parameters of this lane's four channels in each channel block (channels c0 + 16 cb + 4g .. +3)
  f32x4 ca[NCB], cbv[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = PAIR ? (16 * cb + 4 * g + e) % 24 : min(c0 + 16 * cb + 4 * g + e, C - 1);
      ca[cb][e] = __expf(alpha_log[ch]) * 0.15915494309189535f;    // e^alpha / (2 pi): v_sin_f32 takes revolutions
      cbv[cb][e] = __frcp_rn(__expf(beta_log[ch]) + 1e-9f);
    }
  // A workgroup walks tiles_per_wg consecutive tiles of its (batch element, channel slice): the fragments above are built once,
  // and the NEXT tile's x rows are requested as soon as the current tile's have been committed to LDS -- they arrive under the
  // current tile's MFMAs and transcendental ops (a tile's global round trip is otherwise exposed: three workgroups per CU do not
  // cover it).
  // ---- phase 1: x rows t0-12 .. t0+XR-13 -> X image.  Row indices are clamped to the sequence (the reference's replicate
  // padding of x); channels past C (a 24-channel tensor in a 32-channel slice) read as zeros.  A thread keeps its channel quad
  // and walks rows RPP apart.
  constexpr int C4 = CS / 4;
  constexpr int RPP = 256 / C4;                 // rows per pass of the workgroup (RPP * C4 threads take part)
  constexpr int NP = (XR + RPP - 1) / RPP;
  const int pr = tid / C4, pc4 = tid - pr * C4;
  const bool pact = pr < RPP && (PAIR || c0 + pc4 * 4 < C);
  // PAIR: channel quads 0..5 are element b's, 6..11 element b + 1's
  const f16_t* xc = PAIR ? xb + (int64_t)(pc4 >= 6) * T_full * C + (pc4 % 6) * 4 : xb + c0 + pc4 * 4;
  h4 xv[NP];
  auto request = [&](const int t0r) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int row = min(max(t0r - 12 + pr + q * RPP, 0), T_len - 1);
      xv[q] = pact ? *reinterpret_cast<const h4*>(xc + (int64_t)row * C) : h4{0, 0, 0, 0};
    }
  };
  // the S rows past the u rows are read by the last y block's window (against zero taps): they must hold numbers
  if (pr < 16 && pc4 < C4) *reinterpret_cast<h4*>(&Si[(2 * TT + 16 + pr) * RS + pc4 * 4]) = h4{0, 0, 0, 0};
  request((int)blockIdx.x * tiles_per_wg * TT);
  for (int it = 0; it < tiles_per_wg; ++it) {
  const int t0 = ((int)blockIdx.x * tiles_per_wg + it) * TT;
  if (t0 >= T_len) break;
  {
    f16_t* xw = &Xi[pr * RS + pc4 * 4];
#pragma unroll
    for (int q = 0; q < NP; ++q)
      if (pr < RPP && pr + q * RPP < XR) *reinterpret_cast<h4*>(xw + q * RPP * RS) = xv[q];
  }
  __syncthreads();     // X committed; every wave is also done with the previous tile's S reads
  if (it + 1 < tiles_per_wg && t0 + TT < T_len) request(t0 + TT);
  // ---- phase 2: u blocks -> snake -> S image.  Row block ub: u rows 2*t0 - 8 + 16 ub .. +15; its x window starts at image row
  // 8 ub (x row t0 - 12 + 8 ub = q0 - 8).  A wave takes every fourth row block, all channel blocks of it at once (independent
  // chains: the transposed reads, the MFMAs and the transcendental ops of different blocks overlap).
  for (int ub = wave; ub < NUB; ub += 4) {
    f16x8 xf[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) xf[cb] = aa_tr_frag<RS>(Xi, 8 * ub, 16 * cb, lane);
    f32x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[cb], wu_hi, acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[cb], wu_lo, acc[cb], 0, 0, 0);
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      h4 sv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float u = acc[cb][e];
        const float sn = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(u * ca[cb][e]));   // sin(2 pi frac(.)): the hardware's input range
        sv[e] = (f16_t)(u + cbv[cb][e] * sn * sn);
      }
      // lane (g, r): channels 16 cb + 4g .. +3 of u row r of the block
      *reinterpret_cast<h4*>(&Si[(16 * ub + r) * RS + 16 * cb + 4 * g]) = sv;
    }
  }
  __syncthreads();
  // ---- phase 2b: replicate padding of the UPSAMPLED signal at the sequence ends (tile-uniform conditions); image row of m
  // is m - (2 t0 - 8)
  const int m_base = 2 * t0 - 8;
  if (m_base < 0) {
    for (int idx = tid; idx < (-m_base) * CS; idx += 256) {
      const int mi = idx / CS, c = idx - mi * CS;
      Si[mi * RS + c] = Si[(-m_base) * RS + c];
    }
    __syncthreads();
  }
  if (m_base + 2 * TT + 16 > 2 * T_len) {
    const int last = 2 * T_len - 1 - m_base;  // image row of m = 2T-1 (>= 8: the tile starts inside the sequence)
    const int n = 2 * TT + 16 - 1 - last;
    for (int idx = tid; idx < n * CS; idx += 256) {
      const int k = idx / CS, c = idx - k * CS;
      Si[(last + 1 + k) * RS + c] = Si[last * RS + c];
    }
    __syncthreads();
  }
  // ---- phase 3: y blocks.  Row block yb: outputs t0 + 16 yb .. +15; its s window starts at image row 32 yb.
  for (int yb_ = wave; yb_ < NYB; yb_ += 4) {
    f16x8 s0[NCB], s1[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      s0[cb] = aa_tr_frag<RS>(Si, 32 * yb_, 16 * cb, lane);
      s1[cb] = aa_tr_frag<RS>(Si, 32 * yb_ + 32, 16 * cb, lane);
    }
    const int t = t0 + 16 * yb_ + r;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(s0[cb], wd_hi[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(s0[cb], wd_lo[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(s1[cb], wd_hi[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(s1[cb], wd_lo[1], acc, 0, 0, 0);
      // lane (g, r): channels 16 cb + 4g .. +3 of output row t0 + 16 yb + r
      const int ch = c0 + 16 * cb + 4 * g;
      if (PAIR ? t < T_len : (t < T_len && ch < C)) {
        h4 o = {(f16_t)acc[0], (f16_t)acc[1], (f16_t)acc[2], (f16_t)acc[3]};
        if constexpr (PAIR) *reinterpret_cast<h4*>(yb + (int64_t)(ch >= 24) * T_full * C + (int64_t)t * C + ch % 24) = o;
        else *reinterpret_cast<h4*>(yb + (int64_t)t * C + ch) = o;
      }
    }
  }
  }   // tiles of this workgroup
}

// Reference-op layout [B][C][T] (drop-in for anti_alias_activation_cuda.forward): one thread per output sample.
template <typename T>
__global__ __launch_bounds__(256) void aa_snake_bct_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                            const float* __restrict__ alpha_log,
                                                            const float* __restrict__ beta_log, Fir24 f, int T_len, int C) {
  int64_t row = blockIdx.y;  // b*C + c
  int c = (int)(row % C);
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T_len) return;
  const T* xr = x + row * T_len;
  float a = expf(alpha_log[c]);
  float ib = 1.0f / (expf(beta_log[c]) + 1e-9f);
  float xw[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) xw[i] = Elem<T>::to_f(xr[min(max(t - 5 + i, 0), T_len - 1)]);
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    int m = min(max(2 * t + j - 5, 0), 2 * T_len - 1);
    int q = m >> 1;
    float u = 0.f;
    if (m & 1) {
#pragma unroll
      for (int d = -2; d <= 3; ++d) {
        int xi = min(max(q + d, 0), T_len - 1) - (t - 5);
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 11; ++i) v = (i == xi) ? xw[i] : v;
        u = fmaf(v, f.up[6 - 2 * d], u);
      }
    } else {
#pragma unroll
      for (int d = -3; d <= 2; ++d) {
        int xi = min(max(q + d, 0), T_len - 1) - (t - 5);
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 11; ++i) v = (i == xi) ? xw[i] : v;
        u = fmaf(v, f.up[5 - 2 * d], u);
      }
    }
    u *= 2.0f;
    float sn = sinf(u * a);
    acc = fmaf(f.down[j], u + ib * sn * sn, acc);
  }
  y[row * T_len + t] = Elem<T>::from_f(acc);
}

// -------------------------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, two-pass statistics in registers (D <= 4096, D % 4 == 0), optional second LayerNorm.
// -------------------------------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 16;  // float4 per lane

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                         const float* __restrict__ b, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, void* __restrict__ yv, int y_f32,
                                                         int M, int D) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* hr = h + (int64_t)row * D;
  int nv = D / 4;  // float4 count
  f32x4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int idx = lane + i * 64;
    if (idx < nv) {
      v[i] = ld16<f32x4>(hr + idx * 4);
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    } else {
      v[i] = f32x4{0, 0, 0, 0};
    }
  }
  float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int idx = lane + i * 64;
    if (idx < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float d = v[i][e] - mean;
        q = fmaf(d, d, q);
      }
    }
  }
  float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int idx = lane + i * 64;
    if (idx < nv) {
      f32x4 ww = ld16<f32x4>(w + idx * 4), bb = ld16<f32x4>(b + idx * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * rstd * ww[e] + bb[e];
    }
  }
  if (w2 != nullptr) {
    s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
      if (lane + i * 64 < nv) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    mean = wave_sum(s) / (float)D;
    q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
      if (lane + i * 64 < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float d = v[i][e] - mean;
          q = fmaf(d, d, q);
        }
      }
    rstd = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      int idx = lane + i * 64;
      if (idx < nv) {
        f32x4 ww = ld16<f32x4>(w2 + idx * 4), bb = ld16<f32x4>(b2 + idx * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * rstd * ww[e] + bb[e];
      }
    }
  }
  if (y_f32) {
    float* yr = (float*)yv + (int64_t)row * D;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      int idx = lane + i * 64;
      if (idx < nv) st16(yr + idx * 4, v[i]);
    }
  } else {
    T* yr = (T*)yv + (int64_t)row * D;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      int idx = lane + i * 64;
      if (idx < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) yr[idx * 4 + e] = Elem<T>::from_f(v[i][e]);
      }
    }
  }
}

// Residual update (sum of split-K slabs + bias, fixed order) fused with LayerNorm (optionally two in sequence);
// one wave per row.  NV (float4 per lane) and NSLAB are compile-time so that EVERY load -- h, bias, all slabs and both
// LayerNorm parameter sets -- is issued before the first use: one memory round trip per launch.
template <typename T, int NV, int NSLAB, bool LN2>
__global__ __launch_bounds__(64) void ln_reduce_kernel(float* __restrict__ h, const float* __restrict__ slab,
                                                        const float* __restrict__ bias, const float* __restrict__ w,
                                                        const float* __restrict__ b, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, T* __restrict__ y, int M, int D,
                                                        int32_t* __restrict__ bump) {
  const int row = blockIdx.x, lane = threadIdx.x;
  if (bump != nullptr && row == 0 && lane == 0) { bump[0] += 1; bump[1] += 1; }
  float* hr = h + (int64_t)row * D;
  const int nv = D / 4;
  f32x4 v[NV], bs[NV], sl[NSLAB > 0 ? NSLAB : 1][NV], lw[NV], lb[NV], lw2[LN2 ? NV : 1], lb2[LN2 ? NV : 1];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int idx = lane + i * 64;
    bool ok = idx < nv;
    v[i] = ok ? ld16<f32x4>(hr + idx * 4) : zero;
    lw[i] = ok ? ld16<f32x4>(w + idx * 4) : zero;
    lb[i] = ok ? ld16<f32x4>(b + idx * 4) : zero;
    if constexpr (LN2) {
      lw2[i] = ok ? ld16<f32x4>(w2 + idx * 4) : zero;
      lb2[i] = ok ? ld16<f32x4>(b2 + idx * 4) : zero;
    }
    if constexpr (NSLAB > 0) {
      bs[i] = (ok && bias != nullptr) ? ld16<f32x4>(bias + idx * 4) : zero;
#pragma unroll
      for (int sidx = 0; sidx < NSLAB; ++sidx)
        sl[sidx][i] = ok ? ld16<f32x4>(slab + ((int64_t)sidx * M + row) * D + idx * 4) : zero;
    }
  }
  if constexpr (NSLAB > 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] += bs[i];
#pragma unroll
      for (int sidx = 0; sidx < NSLAB; ++sidx) v[i] += sl[sidx][i];
      int idx = lane + i * 64;
      if (idx < nv) st16(hr + idx * 4, v[i]);
    }
  }
#pragma unroll
  for (int pass = 0; pass < (LN2 ? 2 : 1); ++pass) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];  // padding lanes hold zeros
    float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + i * 64 < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float d = v[i][e] - mean;
          q = fmaf(d, d, q);
        }
      }
    float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (lane + i * 64 < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float ww = pass == 0 ? lw[i][e] : lw2[LN2 ? i : 0][e];
          float bb = pass == 0 ? lb[i][e] : lb2[LN2 ? i : 0][e];
          v[i][e] = (v[i][e] - mean) * rstd * ww + bb;
        }
      }
    }
  }
  T* yr = y + (int64_t)row * D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int idx = lane + i * 64;
    if (idx < nv) {
      if constexpr (sizeof(T) == 4) {
        st16(yr + idx * 4, v[i]);
      } else {
        typedef T t4 __attribute__((ext_vector_type(4)));
        t4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f(v[i][e]);
        *reinterpret_cast<t4*>(yr + idx * 4) = o;
      }
    }
  }
}

// One float4 per thread, D/4 threads (D/256 waves) per row: a fifth of the per-lane load queue of the one-wave-per-row
// form for D = 1280; the two row statistics cross the waves through LDS.
template <typename T, int NSLAB, bool LN2>
__global__ __launch_bounds__(1024) void ln_reduce_wide_kernel(float* __restrict__ h, const float* __restrict__ slab,
                                                              const float* __restrict__ bias, const float* __restrict__ w,
                                                              const float* __restrict__ b, const float* __restrict__ w2,
                                                              const float* __restrict__ b2, T* __restrict__ y, int M, int D,
                                                              int32_t* __restrict__ bump, int y_pa, int sstride,
                                                              const float* __restrict__ lora_b, int lora_r) {
  __shared__ float red[2][2][16];
  __shared__ float xa[64];   // runtime LoRA: (x A) of this row, summed over the split-K slabs
  const int row = blockIdx.x, tid = threadIdx.x, nw = blockDim.x >> 6;
  if (bump != nullptr && row == 0 && tid == 0) { bump[0] += 1; bump[1] += 1; }   // nobody in this launch reads them
  float* hr = h + (int64_t)row * D;
  f32x4 v = ld16<f32x4>(hr + tid * 4);
  const f32x4 lw = ld16<f32x4>(w + tid * 4), lb = ld16<f32x4>(b + tid * 4);
  f32x4 lw2 = {0.f, 0.f, 0.f, 0.f}, lb2 = lw2;
  if constexpr (LN2) {
    lw2 = ld16<f32x4>(w2 + tid * 4);
    lb2 = ld16<f32x4>(b2 + tid * 4);
  }
  if constexpr (NSLAB > 0) {
    f32x4 sl[NSLAB];
    f32x4 bs = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) bs = ld16<f32x4>(bias + tid * 4);
#pragma unroll
    for (int sidx = 0; sidx < NSLAB; ++sidx) sl[sidx] = ld16<f32x4>(slab + ((int64_t)sidx * M + row) * sstride + tid * 4);
    v += bs;
#pragma unroll
    for (int sidx = 0; sidx < NSLAB; ++sidx) v += sl[sidx];  // same association order as the one-wave form
    if (lora_r > 0) {
      // Runtime LoRA of the producing projection: its packed weight carried A (scaled) as lora_r extra output columns, so
      // columns [D, D + r) of the slabs hold the split-K partials of x A; the update is  v += (x A) B  with B fp32 [r][D].
      // Fixed orders (slabs, then j = 0..r-1): deterministic.  Equal to the merged weight W + A B within rounding.
      if (tid < lora_r) {
        float t = 0.f;
#pragma unroll
        for (int sidx = 0; sidx < NSLAB; ++sidx) t += slab[((int64_t)sidx * M + row) * sstride + D + tid];
        xa[tid] = t;
      }
      __syncthreads();
      for (int j = 0; j < lora_r; ++j) {
        const f32x4 bj = ld16<f32x4>(lora_b + (int64_t)j * D + tid * 4);
        const float t = xa[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(t, bj[e], v[e]);
      }
    }
    st16(hr + tid * 4, v);
  }
  wide_layernorm<LN2>(v, lw, lb, lw2, lb2, &red[0][0][0], tid, nw, D, true);
  store_row4<T>(y + (y_pa ? pa_off<T>(row, tid * 4, (M + 15) >> 4) : (int64_t)row * D + tid * 4), v);
}

template <typename T, int NV>
static int launch_ln_reduce(const itts_ln_reduce_args& a, hipStream_t s) {
  float* h = a.h;
  const float *slab = a.slab, *bias = a.bias, *w = a.w, *b = a.b, *w2 = a.w2, *b2 = a.b2;
  T* y = (T*)a.y;
  const int M = a.M, D = a.D, nslab = a.nslab, y_pa = a.y_packed, lora_r = a.lora_b ? a.lora_r : 0;
  const int sstride = a.slab_stride > 0 ? a.slab_stride : D;
  int32_t* bump = a.state_bump;
  const float* lora_b = a.lora_b;
  const bool wide = (D % 256 == 0) && D <= 4096;
  if ((y_pa || lora_r > 0 || sstride != D) && !wide) {
    set_error("itts_ln_reduce: packed y / slab_stride / LoRA need D %% 256 == 0 (D=%d)", D);
    return ITTS_ERR_INVALID;
  }
  dim3 grid(M), block(wide ? D / 4 : 64);
  const bool two = w2 != nullptr;
#define ITTS_LNR(NS, L2)                                                                                                   \
  do {                                                                                                                     \
    if (wide) hipLaunchKernelGGL((ln_reduce_wide_kernel<T, NS, L2>), grid, block, 0, s, h, slab, bias, w, b, w2, b2, y, M, D, bump, y_pa, sstride, lora_b, lora_r); \
    else hipLaunchKernelGGL((ln_reduce_kernel<T, NV, NS, L2>), grid, block, 0, s, h, slab, bias, w, b, w2, b2, y, M, D, bump);     \
  } while (0)
  if (nslab == 0) {
    if (two) ITTS_LNR(0, true); else ITTS_LNR(0, false);
  } else if (nslab == 6) {
    if (two) ITTS_LNR(6, true); else ITTS_LNR(6, false);
  } else if (nslab == 5) {
    if (two) ITTS_LNR(5, true); else ITTS_LNR(5, false);
  } else if (nslab == 4) {
    if (two) ITTS_LNR(4, true); else ITTS_LNR(4, false);
  } else if (nslab == 3) {
    if (two) ITTS_LNR(3, true); else ITTS_LNR(3, false);
  } else if (nslab == 2) {
    if (two) ITTS_LNR(2, true); else ITTS_LNR(2, false);
  } else if (nslab == 1) {
    if (two) ITTS_LNR(1, true); else ITTS_LNR(1, false);
  } else {
    set_error("itts_ln_reduce: nslab must be 0..6 (got %d)", nslab);
    return ITTS_ERR_INVALID;
  }
#undef ITTS_LNR
  return check_launch("itts_ln_reduce");
}

// h[b] = table[tokens[b]] + pos_table[p], p = *step - row_step0[b] + pos_add clamped to the table (a slot whose row has
// finished keeps stepping formally until the loop ends; its value is discarded, its read must stay inside the table).
// hp (optional): the same rows as T in the packed activation layout -- what the LayerNorm-folded QKV GEMM multiplies.
template <typename T>
__global__ __launch_bounds__(256) void embed_step_kernel(const int32_t* __restrict__ tokens, const float* __restrict__ table,
                                                          const float* __restrict__ pos_table,
                                                          const int32_t* __restrict__ step, int pos_add,
                                                          float* __restrict__ h, int D, int32_t* __restrict__ bump,
                                                          const int32_t* __restrict__ row_step0, int pos_rows,
                                                          T* __restrict__ hp, int mtp) {
  constexpr int E = Elem<T>::E;
  int b = blockIdx.x;
  int tok = tokens[b];
  const int32_t* s0p = row_step0 != nullptr ? row_step0 + b : step;   // a readable word either way (no branch around the load)
  const int s0 = row_step0 != nullptr ? *s0p : 0;
  int p = step[0] - s0 + pos_add;
  p = min(max(p, 0), pos_rows - 1);
  const float* e = table + (int64_t)tok * D;
  const float* pe = pos_table + (int64_t)p * D;
  for (int c = threadIdx.x; c < D / E; c += 256) {
    float v[E];
#pragma unroll
    for (int i = 0; i < E; ++i) v[i] = e[c * E + i] + pe[c * E + i];
#pragma unroll
    for (int i = 0; i < E; ++i) h[(int64_t)b * D + c * E + i] = v[i];
    if (hp != nullptr) {
      T* dst = hp + pa_off<T>(b, c * E, mtp);
#pragma unroll
      for (int i = 0; i < E; ++i) dst[i] = Elem<T>::from_f(v[i]);
    }
  }
  // the loop-state word this launch advances (the cache position: nothing here reads it)
  if (bump != nullptr && b == 0 && threadIdx.x == 0) bump[0] += 1;
}

template <typename T>
__global__ __launch_bounds__(256) void tanh_pcm_kernel(const T* __restrict__ x, float* __restrict__ wav,
                                                        int16_t* __restrict__ pcm, int64_t n, int apply_tanh) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    float v = Elem<T>::to_f(x[i]);
    if (apply_tanh) v = tanhf(v);
    if (wav) wav[i] = v;
    if (pcm) {
      float s = fminf(fmaxf(32767.0f * v, -32767.0f), 32767.0f);
      pcm[i] = (int16_t)s;  // truncation toward zero, as numpy astype(int16) (infer.py:911)
    }
  }
}


}  // namespace itts

using namespace itts;

// consecutive tiles one workgroup of the MFMA activation walks: enough workgroups to fill the chip several times over (768
// resident at 3 per CU), as few fragment set-ups as that allows
#ifndef ITTS_AA_PAIR24
#define ITTS_AA_PAIR24 1         // build-time A/B: C = 24 with two batch elements per 48-channel slice
#endif
#ifndef ITTS_AA_MFMA_MAXC
#define ITTS_AA_MFMA_MAXC 192    // build-time A/B: widest tensor the MFMA form of the activation takes
#endif
// Workgroups the MFMA form aims for: 3072 (four rounds of the 768 resident ones) at C <= 96; at C = 192 a batch element has only
// 70 tiles per 48-channel slice, and twice the tiles per workgroup (1536 workgroups) amortise the tap-fragment set-up better:
// 85 us against 94 for either form with 3072 (profiles/r04_act_variants.txt; C = 96: 150 / 163 / 179 us for 3072 / 1536 / 6144).
static int aa_tiles_per_wg(int64_t tiles, int C) {
  int64_t t = tiles / (C > 96 ? 1536 : 3072);
  return (int)(t < 1 ? 1 : t > 8 ? 8 : t);
}

extern "C" int itts_aa_snake_fwd(const void* x, void* y, const float* alpha_log, const float* beta_log,
                                 const float* up_filter12, const float* down_filter12, int B, int T, int C, int dtype,
                                 int layout, const int32_t* valid_rows, void* stream) {
  ITTS_REQUIRE(valid_rows == nullptr || layout == 0, "itts_aa_snake_fwd: valid_rows needs the channels-last layout (0)");
  ITTS_REQUIRE(x && y && alpha_log && beta_log && up_filter12 && down_filter12, "itts_aa_snake_fwd: null pointer");
  ITTS_REQUIRE(B >= 0 && T >= 0 && C > 0, "itts_aa_snake_fwd: bad shape B=%d T=%d C=%d", B, T, C);
  if (B == 0 || T == 0) return ITTS_OK;
  Fir24 f;
  for (int i = 0; i < 12; ++i) {
    f.up[i] = up_filter12[i];
    f.down[i] = down_filter12[i];
  }
  hipStream_t s = (hipStream_t)stream;
  if (layout == 0) {
    ITTS_REQUIRE(C % 4 == 0, "itts_aa_snake_fwd: C=%d must be a multiple of 4", C);
    int CS = (C % 64 == 0) ? 64 : (C % 48 == 0) ? 48 : (C % 32 == 0) ? 32 : (C % 24 == 0) ? 24 : 0;
    ITTS_REQUIRE(CS != 0, "itts_aa_snake_fwd: unsupported channel count %d (need a multiple of 24 or 32)", C);
    const int tt = CS == 64 ? AaTile<64>::TT : CS == 48 ? AaTile<48>::TT : CS == 32 ? AaTile<32>::TT : AaTile<24>::TT;
    dim3 grid((T + tt - 1) / tt, C / CS, B), block(256);
    ITTS_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "itts_aa_snake_fwd: grid too large");
#define ITTS_AA_LAUNCH(TT_, CS_) \
  hipLaunchKernelGGL((aa_snake_btc_kernel<TT_, CS_>), grid, block, 0, s, (const TT_*)x, (TT_*)y, alpha_log, beta_log, f, T, C, valid_rows)
#define ITTS_AA_BY_CS(TT_)                         \
  switch (CS) {                                    \
    case 64: ITTS_AA_LAUNCH(TT_, 64); break;       \
    case 48: ITTS_AA_LAUNCH(TT_, 48); break;       \
    case 32: ITTS_AA_LAUNCH(TT_, 32); break;       \
    default: ITTS_AA_LAUNCH(TT_, 24); break;       \
  }
    switch (dtype) {
      case ITTS_F32:
        ITTS_AA_BY_CS(float);
        break;
      case ITTS_BF16:
        ITTS_AA_BY_CS(bf16_t);
        break;
      case ITTS_F16:
        if (ITTS_AA_F16_MFMA && C <= ITTS_AA_MFMA_MAXC && (int64_t)B * T >= 32768) {
          // Both FIRs on the matrix cores: 48-channel slices where the channel count allows, one 32-channel slice for C = 24.
          // Measured (batch 32, MI355X, us per launch, MFMA form | VALU form): C = 96: 163 | 207, C = 48: 151 | 193, C = 24: 166 |
          // 191; C = 192: 97 | 96 (85 with twice the tiles per workgroup: round 4), C = 384: 56 | 46, C = 768: 37 | 27 (few rows per batch
          // element: the per-workgroup set-up of the tap fragments is not amortised) -- hence the last four stages only.
          if (C % 48 == 0) {
            static std::once_flag once3;
            std::call_once(once3, [] {
              (void)hipFuncSetAttribute((const void*)aa_snake_mfma_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AaMfma<3>::LDS);
            });
            const int nt = (T + AaMfma<3>::TT - 1) / AaMfma<3>::TT;
            const int tpw = aa_tiles_per_wg((int64_t)nt * (C / 48) * B, C);
            dim3 g3((nt + tpw - 1) / tpw, C / 48, B);
            hipLaunchKernelGGL(aa_snake_mfma_kernel<3>, g3, block, AaMfma<3>::LDS, s, (const f16_t*)x, (f16_t*)y, alpha_log, beta_log, f, T, C, valid_rows, tpw);
          } else if (C == 24 && valid_rows == nullptr && B % 2 == 0 && ITTS_AA_PAIR24) {
            // C = 24, dense batch: two batch elements share a 48-channel slice (3 full MFMA blocks instead of 2 x 1.5)
            static std::once_flag once3p;
            std::call_once(once3p, [] {
              (void)hipFuncSetAttribute((const void*)aa_snake_mfma_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AaMfma<3>::LDS);
            });
            const int nt = (T + AaMfma<3>::TT - 1) / AaMfma<3>::TT;
            const int tpw = aa_tiles_per_wg((int64_t)nt * (B / 2), C);
            dim3 g3((nt + tpw - 1) / tpw, 1, B / 2);
            hipLaunchKernelGGL((aa_snake_mfma_kernel<3, true>), g3, block, AaMfma<3>::LDS, s, (const f16_t*)x, (f16_t*)y, alpha_log, beta_log, f, T, C, valid_rows, tpw);
          } else {
            static std::once_flag once2;
            std::call_once(once2, [] {
              (void)hipFuncSetAttribute((const void*)aa_snake_mfma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AaMfma<2>::LDS);
            });
            const int nt = (T + AaMfma<2>::TT - 1) / AaMfma<2>::TT;
            const int tpw = aa_tiles_per_wg((int64_t)nt * ((C + 31) / 32) * B, C);
            dim3 g2((nt + tpw - 1) / tpw, (C + 31) / 32, B);
            hipLaunchKernelGGL(aa_snake_mfma_kernel<2>, g2, block, AaMfma<2>::LDS, s, (const f16_t*)x, (f16_t*)y, alpha_log, beta_log, f, T, C, valid_rows, tpw);
          }
          break;
        }
        ITTS_AA_BY_CS(f16_t);
        break;
      default:
        ITTS_REQUIRE(false, "itts_aa_snake_fwd: unknown dtype %d", dtype);
    }
#undef ITTS_AA_BY_CS
#undef ITTS_AA_LAUNCH
  } else {
    ITTS_REQUIRE((int64_t)B * C <= 65535, "itts_aa_snake_fwd: B*C too large for layout 1");
    dim3 grid((T + 255) / 256, B * C), block(256);
    switch (dtype) {
      case ITTS_F32:
        hipLaunchKernelGGL(aa_snake_bct_kernel<float>, grid, block, 0, s, (const float*)x, (float*)y, alpha_log, beta_log, f, T, C);
        break;
      case ITTS_BF16:
        hipLaunchKernelGGL(aa_snake_bct_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, (bf16_t*)y, alpha_log, beta_log, f, T, C);
        break;
      case ITTS_F16:
        hipLaunchKernelGGL(aa_snake_bct_kernel<f16_t>, grid, block, 0, s, (const f16_t*)x, (f16_t*)y, alpha_log, beta_log, f, T, C);
        break;
      default:
        ITTS_REQUIRE(false, "itts_aa_snake_fwd: unknown dtype %d", dtype);
    }
  }
  return check_launch("itts_aa_snake_fwd");
}

extern "C" int itts_layernorm(const float* h, const float* w, const float* b, const float* w2, const float* b2, void* y,
                              int y_f32, int M, int D, int dtype, void* stream) {
  ITTS_REQUIRE(h && w && b && y, "itts_layernorm: null pointer");
  ITTS_REQUIRE(D % 4 == 0 && D <= 4 * 64 * LN_MAXV && D > 0, "itts_layernorm: unsupported D=%d", D);
  if (M == 0) return ITTS_OK;
  dim3 grid((M + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case ITTS_F32:
      hipLaunchKernelGGL(layernorm_kernel<float>, grid, block, 0, s, h, w, b, w2, b2, y, y_f32, M, D);
      break;
    case ITTS_BF16:
      hipLaunchKernelGGL(layernorm_kernel<bf16_t>, grid, block, 0, s, h, w, b, w2, b2, y, y_f32, M, D);
      break;
    case ITTS_F16:
      hipLaunchKernelGGL(layernorm_kernel<f16_t>, grid, block, 0, s, h, w, b, w2, b2, y, y_f32, M, D);
      break;
    default:
      ITTS_REQUIRE(false, "itts_layernorm: unknown dtype %d", dtype);
  }
  return check_launch("itts_layernorm");
}

extern "C" int itts_ln_reduce(const itts_ln_reduce_args* a, void* stream) {
  ITTS_REQUIRE(a && a->h && a->w && a->b && a->y, "itts_ln_reduce: null pointer");
  const int D = a->D;
  ITTS_REQUIRE(a->nslab >= 0 && (a->nslab == 0 || a->slab != nullptr), "itts_ln_reduce: slab missing");
  ITTS_REQUIRE((a->w2 == nullptr) == (a->b2 == nullptr), "itts_ln_reduce: pass both or neither of w2/b2");
  // D % 256 == 0 (<= 4096): one float4 per thread, D/4 threads per row; any other D <= 1280: one wave per row
  ITTS_REQUIRE(D % 4 == 0 && D > 0 && ((D % 256 == 0 && D <= 4096) || D <= 4 * 64 * 5),
               "itts_ln_reduce: unsupported D=%d (a multiple of 256 up to 4096, or any multiple of 4 up to 1280)", D);
  ITTS_REQUIRE(a->slab_stride == 0 || a->slab_stride >= D, "itts_ln_reduce: slab_stride %d < D", a->slab_stride);
  if (a->lora_b != nullptr)
    ITTS_REQUIRE(a->lora_r > 0 && a->lora_r <= 64 && a->nslab > 0 && a->slab_stride >= D + a->lora_r,
                 "itts_ln_reduce: LoRA needs 1 <= r <= 64, slabs, and slab_stride >= D + r");
  if (a->M == 0) return ITTS_OK;
  hipStream_t s = (hipStream_t)stream;
  switch (a->dtype) {
    case ITTS_F32:
      return launch_ln_reduce<float, 5>(*a, s);
    case ITTS_BF16:
      return launch_ln_reduce<bf16_t, 5>(*a, s);
    case ITTS_F16:
      return launch_ln_reduce<f16_t, 5>(*a, s);
  }
  ITTS_REQUIRE(false, "itts_ln_reduce: unknown dtype %d", a->dtype);
}

extern "C" int itts_embed_step(const int32_t* tokens, const float* table, const float* pos_table, const int32_t* step,
                               int pos_add, float* h, int B, int D, int32_t* bump, const int32_t* row_step0, int pos_rows,
                               void* h_packed, int dtype, void* stream) {
  ITTS_REQUIRE(tokens && table && pos_table && step && h && B > 0 && D > 0 && pos_rows > 0, "itts_embed_step: bad arguments");
  ITTS_REQUIRE(D % 8 == 0, "itts_embed_step: D=%d must be a multiple of 8", D);
  if (h_packed != nullptr) ITTS_REQUIRE(D % (dtype == ITTS_F32 ? 16 : 32) == 0, "itts_embed_step: a packed copy needs D %% k-step == 0");
  hipStream_t s = (hipStream_t)stream;
  const int mtp = (B + 15) / 16;
  switch (dtype) {
    case ITTS_F32:
      hipLaunchKernelGGL(embed_step_kernel<float>, dim3(B), dim3(256), 0, s, tokens, table, pos_table, step, pos_add, h, D, bump,
                         row_step0, pos_rows, (float*)h_packed, mtp);
      break;
    case ITTS_BF16:
      hipLaunchKernelGGL(embed_step_kernel<bf16_t>, dim3(B), dim3(256), 0, s, tokens, table, pos_table, step, pos_add, h, D, bump,
                         row_step0, pos_rows, (bf16_t*)h_packed, mtp);
      break;
    case ITTS_F16:
      hipLaunchKernelGGL(embed_step_kernel<f16_t>, dim3(B), dim3(256), 0, s, tokens, table, pos_table, step, pos_add, h, D, bump,
                         row_step0, pos_rows, (f16_t*)h_packed, mtp);
      break;
    default:
      ITTS_REQUIRE(false, "itts_embed_step: unknown dtype %d", dtype);
  }
  return check_launch("itts_embed_step");
}

extern "C" int itts_tanh_pcm(const void* x, float* wav, int16_t* pcm, int64_t n, int dtype, int apply_tanh, void* stream) {
  ITTS_REQUIRE(x && n >= 0, "itts_tanh_pcm: bad arguments");
  if (n == 0) return ITTS_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  dim3 grid((unsigned)blocks), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case ITTS_F32:
      hipLaunchKernelGGL(tanh_pcm_kernel<float>, grid, block, 0, s, (const float*)x, wav, pcm, n, apply_tanh);
      break;
    case ITTS_BF16:
      hipLaunchKernelGGL(tanh_pcm_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, wav, pcm, n, apply_tanh);
      break;
    case ITTS_F16:
      hipLaunchKernelGGL(tanh_pcm_kernel<f16_t>, grid, block, 0, s, (const f16_t*)x, wav, pcm, n, apply_tanh);
      break;
    default:
      ITTS_REQUIRE(false, "itts_tanh_pcm: unknown dtype %d", dtype);
  }
  return check_launch("itts_tanh_pcm");
}

