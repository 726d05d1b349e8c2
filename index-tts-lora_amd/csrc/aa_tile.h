// Anti-aliased SnakeBeta on one [TT rows x CS channels] tile of a channels-last signal, staged in LDS.
// Shared by aa_snake_btc_kernel (elementwise.hip) and the fused activation + narrow-convolution kernel (gemm_conv.hip).
//
// Follows alias_free_torch/act.py:10-28: UpSample1d (resample.py:10-35) -> SnakeBeta (activations.py:63-122) ->
// DownSample1d/LowPassFilter1d (resample.py:38-48, filter.py:60-95), in polyphase form:
//   u[2q]   = 2 * sum_{d=-3..2} x[q+d] * up[5-2d]        u[2q+1] = 2 * sum_{d=-2..3} x[q+d] * up[6-2d]   (x index clamped)
//   s[m]    = u[m] + sin^2(u[m] * e^alpha) / (e^beta + 1e-9)
//   y[t]    = sum_{j<12} down[j] * s[clamp(2t + j - 5, 0, 2T-1)]
#pragma once
#include "common.h"

namespace itts {

struct Fir24 {
  float up[12];
  float down[12];
};

constexpr int AA_R = 4;     // consecutive rows per thread (register blocking along time)
// Output rows per workgroup, by channel-slice width: the largest tile whose two compute phases (ceil((TT+6)/R) pair groups
// and TT/R output groups, times CS/4 channel quads) each fit one pass of the 256 threads.
template <int CS> struct AaTile { static constexpr int TT = CS == 64 ? 56 : CS == 48 ? 76 : CS == 32 ? 120 : 160; };

// LDS geometry of a tile: x rows t0-6 .. (kept in the storage type), s rows m = 2*t0-6 .. 2*t0+2*TT+5 (fp32)
template <int CS, int TT>
struct AaShape {
  static constexpr int R = AA_R, SR = 2 * TT + 12, NQ = TT + 6, C4 = CS / 4;
  static constexpr int QG = (NQ + R - 1) / R;   // pair groups (the last one may be partial)
  static constexpr int XRP = QG * R + 6;        // x rows the pair groups may touch (>= TT + 12; the excess is never used)
};

// sin for the periodic term: exact library sinf in fp32 (parity) mode, hardware v_sin_f32 for 16-bit storage types
template <typename T>
__device__ __forceinline__ float aa_sin(float x) {
  if constexpr (sizeof(T) == 4) return sinf(x);
  else return __sinf(x);
}

// Computes the activation for rows t0 .. t0+TT-1 (any t0, also negative or past the end: the x rows are clamped to the
// sequence like the reference's replicate padding, and `store(tt, c4, value)` decides what to do with row t0+tt) of the
// CS channels starting at xb.  xs: T[XRP*CS], ss: float[SR*CS], ca/cb: float[CS] holding e^alpha and 1/(e^beta+1e-9)
// (filled by the caller BEFORE the call; a barrier inside orders them).  Ends with all threads past their stores but
// WITHOUT a trailing barrier.  Work is vectorised over 4 adjacent channels everywhere (8/16-byte LDS and global accesses)
// and register-blocked over AA_R consecutive rows, so that a thread re-uses the FIR windows it has read.
template <typename T, int CS, int TT, typename StoreFn>
__device__ __forceinline__ void aa_tile(const T* __restrict__ xb, int C, int T_len, int t0, const Fir24& f, T* xs, float* ss,
                                        const float* ca, const float* cb, StoreFn store) {
  typedef AaShape<CS, TT> SH;
  constexpr int R = SH::R, SR = SH::SR, NQ = SH::NQ, C4 = SH::C4, QG = SH::QG, XRP = SH::XRP;
  typedef T t4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x;
  // phase 1: x tile.  All of a thread's loads are issued before the first LDS write: one HBM round trip per tile.
  {
    constexpr int NLD = (XRP * C4 + 255) / 256;
    t4 v[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      int idx = min(tid + q * 256, XRP * C4 - 1);
      int i = idx / C4, c4 = idx - i * C4;
      int row = min(max(t0 - 6 + i, 0), T_len - 1);
      v[q] = *reinterpret_cast<const t4*>(xb + (int64_t)row * C + c4 * 4);
    }
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      int idx = tid + q * 256;
      int i = idx / C4, c4 = idx - i * C4;
      if (idx < XRP * C4) *reinterpret_cast<t4*>(&xs[i * CS + c4 * 4]) = v[q];
    }
  }
  __syncthreads();
  // phase 2: upsample (polyphase, gain 2) + SnakeBeta -> s tile; R pairs per thread from R+6 x rows
  for (int idx = tid; idx < QG * C4; idx += 256) {
    int qg = idx / C4, c4 = idx - qg * C4;
    const int i0 = qg * R;
    f32x4 xv[R + 6];
#pragma unroll
    for (int k = 0; k < R + 6; ++k) {
      t4 v = *reinterpret_cast<const t4*>(&xs[(i0 + k) * CS + c4 * 4]);
      xv[k] = f32x4{Elem<T>::to_f(v[0]), Elem<T>::to_f(v[1]), Elem<T>::to_f(v[2]), Elem<T>::to_f(v[3])};
    }
    const f32x4 a = *reinterpret_cast<const f32x4*>(&ca[c4 * 4]);
    const f32x4 ib = *reinterpret_cast<const f32x4*>(&cb[c4 * 4]);
#pragma unroll
    for (int q = 0; q < R; ++q) {
      if (i0 + q < NQ) {
        f32x4 ue = {0.f, 0.f, 0.f, 0.f}, uo = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          float we = f.up[11 - 2 * k], wo = f.up[10 - 2 * k];  // even: x[i-3+k]*up[11-2k]; odd: x[i-2+k]*up[10-2k]
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            ue[e] = fmaf(xv[q + k][e], we, ue[e]);
            uo[e] = fmaf(xv[q + k + 1][e], wo, uo[e]);
          }
        }
        f32x4 se, so;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float u = 2.0f * ue[e];
          float sn = aa_sin<T>(u * a[e]);
          se[e] = u + ib[e] * sn * sn;
          u = 2.0f * uo[e];
          sn = aa_sin<T>(u * a[e]);
          so[e] = u + ib[e] * sn * sn;
        }
        *reinterpret_cast<f32x4*>(&ss[(2 * (i0 + q)) * CS + c4 * 4]) = se;
        *reinterpret_cast<f32x4*>(&ss[(2 * (i0 + q) + 1) * CS + c4 * 4]) = so;
      }
    }
  }
  __syncthreads();
  // phase 2b: replicate padding of the UPSAMPLED signal at the sequence ends (tile-uniform conditions)
  const int m_base = 2 * t0 - 6;
  if (m_base < 0 && -m_base < SR) {
    for (int idx = tid; idx < (-m_base) * CS; idx += 256) {
      int mi = idx / CS, c = idx - mi * CS;
      ss[mi * CS + c] = ss[(-m_base) * CS + c];
    }
    __syncthreads();
  }
  if (m_base + SR > 2 * T_len && 2 * T_len - 1 - m_base >= 0) {
    const int last = 2 * T_len - 1 - m_base;  // tile row of m = 2T-1
    for (int idx = tid; idx < (SR - 1 - last) * CS; idx += 256) {
      int mi = last + 1 + idx / CS, c = idx % CS;
      ss[mi * CS + c] = ss[last * CS + c];
    }
    __syncthreads();
  }
  // phase 3: 12-tap low-pass, stride 2; R outputs per thread from 2R+10 s rows
  static_assert(TT % R == 0, "row groups tile the tile");
  for (int idx = tid; idx < (TT / R) * C4; idx += 256) {
    int tg = idx / C4, c4 = idx - tg * C4;
    const int tt0 = tg * R;
    f32x4 sv[2 * R + 10];
#pragma unroll
    for (int j = 0; j < 2 * R + 10; ++j) sv[j] = *reinterpret_cast<const f32x4*>(&ss[(2 * tt0 + 1 + j) * CS + c4 * 4]);
#pragma unroll
    for (int q = 0; q < R; ++q) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 12; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(f.down[j], sv[2 * q + j][e], acc[e]);
      store(tt0 + q, c4, acc);
    }
  }
}

}  // namespace itts
