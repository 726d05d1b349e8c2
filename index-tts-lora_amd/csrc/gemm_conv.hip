// Tiled MFMA GEMM / channels-last 1-D convolution (implicit GEMM over taps).
//
//   acc(b,t,n) = sum_j sum_c X[b][t + off0 + j*dil][c] * W[j][c][n]
//
// Used for: GPT prefill + teacher-forced latent pass GEMMs (taps = 1), BigVGAN conv_pre / AMP-block dilated convs /
// conv_post (taps = 3,7,11) and the transposed-conv upsamplers (rewritten as 1- or 2-tap convs).
// Workgroup = 4 waves.  The activation tile (BM rows + dilation halo, CK k-steps of channels) is staged ONCE per
// channel chunk in LDS and reused by every tap (a k-tap conv reads its input once, not k times); rows are padded to
// 144 B so the 16-row fragment reads (ds_read_b128) are bank-conflict free.  Packed weight blocks go straight from
// L2 into B-fragment registers (they are shared only by workgroups, which L2 serves), double-buffered one step ahead;
// the next chunk's activation rows are fetched into registers while the current chunk is being multiplied.
#include "common.h"

namespace itts {

constexpr int CV_MAX_HALO = 64;                 // (taps-1)*dil must not exceed this

struct ConvParams {
  int B, Tin, Tout, Cin, N;
  int taps, off0, dil;
  const void* x;
  int64_t x_bstride;
  const void* wp;
  const float* bias;
  const float* bias2;
  int act;
  void* y;
  int y_f32;
  int64_t y_bstride, y_shift, y_limit;
  const void* resid;
  int accumulate;
  float scale;
  int NT, KT;
};

// CK = k-steps of channels staged per chunk (2 for convolutions, whose taps multiply the MFMA work per chunk; 4 for
// plain GEMMs).  HALO = compile-time bound on (taps-1)*dil (0 for plain GEMMs) that sizes the staging registers.
template <typename T, int WM, int WN, int TM, int TN, int CK, int HALO>
__global__ __launch_bounds__(256) void gemm_conv_kernel(ConvParams p) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  constexpr int E = EL::E, KS = EL::KS;
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
  constexpr int SEGS = CK * 4;                 // 16-byte segments per staged row
  constexpr int ROWB = CK * 64 + 16;           // LDS bytes per staged row (payload + 16 B pad: conflict-free b128 reads)
  constexpr int MAXST = ((BM + HALO) * SEGS + 255) / 256;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;
  const int wm = wave / WN, wn = wave % WN;
  const int t0 = blockIdx.x * BM;
  const int nt0 = blockIdx.y * (BN / 16) + wn * TN;
  const int b = blockIdx.z;
  const int HR = BM + (p.taps - 1) * p.dil;   // staged rows
  const int NC = (p.KT + CK - 1) / CK;        // channel chunks
  const T* xb = (const T*)p.x + (int64_t)b * p.x_bstride;
  const char* wp = (const char*)p.wp;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  frag stg[MAXST];
  auto prefetch_a = [&](int c) {
#pragma unroll
    for (int q = 0; q < MAXST; ++q) {
      int idx = tid + q * 256;
      int i = idx / SEGS, seg = idx - i * SEGS;
      int tin = t0 + p.off0 + i;
      int col = c * (CK * KS) + seg * E;
      bool ok = (i < HR) && (tin >= 0) && (tin < p.Tin) && (col < p.Cin);
      stg[q] = ok ? ld16<frag>(xb + (int64_t)tin * p.Cin + col) : zero_frag<frag>();
    }
  };
  auto commit_a = [&]() {
#pragma unroll
    for (int q = 0; q < MAXST; ++q) {
      int idx = tid + q * 256;
      int i = idx / SEGS, seg = idx - i * SEGS;
      if (i < HR) st16(lds + i * ROWB + seg * 16, stg[q]);
    }
  };
  // Flat step space: step s = (chunk c, tap j, k-step kk within the chunk); weight block = ((j*NT + nt)*KT + c*CK + kk).
  // The last chunk may hold fewer than CK k-steps.
  auto fetch_b = [&](frag (&bf)[TN], int c, int j, int kk) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      int nt = nt0 + tn;
      bool ok = (nt < p.NT) && (c < NC);
      bf[tn] = ok ? ld16<frag>(wp + ((((int64_t)j * p.NT + nt) * p.KT + c * CK + kk) * 64 + lane) * 16) : zero_frag<frag>();
    }
  };
  auto advance = [&](int& c, int& j, int& kk) {
    int nkk = min(CK, p.KT - c * CK);
    if (++kk == nkk) {
      kk = 0;
      if (++j == p.taps) {
        j = 0;
        ++c;
      }
    }
  };

  frag b0[TN], b1[TN], b2[TN];
  int fc = 0, fj = 0, fk = 0;  // cursor of the next weight fetch
  fetch_b(b0, fc, fj, fk);
  advance(fc, fj, fk);
  fetch_b(b1, fc, fj, fk);
  advance(fc, fj, fk);
  prefetch_a(0);
  for (int c = 0; c < NC; ++c) {
    __syncthreads();  // everyone finished reading the previous chunk
    commit_a();
    __syncthreads();
    if (c + 1 < NC) prefetch_a(c + 1);
    const int nkk = min(CK, p.KT - c * CK);
    for (int j = 0; j < p.taps; ++j) {
      for (int kk = 0; kk < nkk; ++kk) {
        fetch_b(b2, fc, fj, fk);  // two steps ahead; crosses chunk boundaries, so no bubble after the barriers
        advance(fc, fj, fk);
        const unsigned char* abase = lds + (wm * TM * 16 + r + j * p.dil) * ROWB + kk * 64 + g * 16;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          frag af = ld16<frag>(abase + tm * 16 * ROWB);
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = EL::mma(af, b0[tn], acc[tm][tn]);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          b0[tn] = b1[tn];
          b1[tn] = b2[tn];
        }
      }
    }
  }

  // ---- epilogue
  char* yb = (char*)p.y;
  const char* rb = (const char*)p.resid;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int col = (nt0 + tn) * 16 + r;
    if (col >= p.N) continue;
    float bs = p.bias ? p.bias[col] : 0.f;
    if (p.bias2) bs += p.bias2[(int64_t)b * p.N + col];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        int t = t0 + wm * TM * 16 + tm * 16 + g * 4 + jj;
        if (t >= p.Tout) continue;
        int64_t flat = (int64_t)t * p.N + col + p.y_shift;
        if (flat < 0 || flat >= p.y_limit) continue;
        int64_t off = (int64_t)b * p.y_bstride + flat;
        float v = acc[tm][tn][jj] + bs;
        if (p.act == 1) v = gelu_new(v);
        if (p.y_f32) {
          if (rb) v += ((const float*)rb)[off];
          v *= p.scale;
          if (p.accumulate) v += ((float*)yb)[off];
          ((float*)yb)[off] = v;
        } else {
          if (rb) v += EL::to_f(((const T*)rb)[off]);
          v *= p.scale;
          if (p.accumulate) v += EL::to_f(((T*)yb)[off]);
          ((T*)yb)[off] = EL::from_f(v);
        }
      }
    }
  }
}

template <typename T, int WM, int WN, int TM, int TN, int CK, int HALO>
static int launch_conv(const ConvParams& p, hipStream_t s) {
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
  int HR = BM + (p.taps - 1) * p.dil;
  size_t ldsb = (size_t)HR * (CK * 64 + 16);
  dim3 grid((p.Tout + BM - 1) / BM, (p.N + BN - 1) / BN, p.B);
  if (grid.y > 65535 || grid.z > 65535) {
    set_error("itts_gemm_conv: grid too large (%u,%u,%u)", grid.x, grid.y, grid.z);
    return ITTS_ERR_INVALID;
  }
  static bool attr = false;
  if (!attr && ldsb > 64 * 1024) {
    hipFuncSetAttribute((const void*)gemm_conv_kernel<T, WM, WN, TM, TN, CK, HALO>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_conv_kernel<T, WM, WN, TM, TN, CK, HALO>), grid, dim3(256), ldsb, s, p);
  return check_launch("itts_gemm_conv");
}

template <typename T>
static int dispatch_conv(const ConvParams& p, hipStream_t s) {
  const bool plain = (p.taps == 1);  // GEMM: no halo, 4 k-steps per chunk
  if (p.N % 128 == 0) return plain ? launch_conv<T, 1, 4, 16, 2, 4, 0>(p, s) : launch_conv<T, 1, 4, 16, 2, 2, CV_MAX_HALO>(p, s);
  if (p.N % 64 == 0) return plain ? launch_conv<T, 2, 2, 8, 2, 4, 0>(p, s) : launch_conv<T, 2, 2, 8, 2, 2, CV_MAX_HALO>(p, s);
  if (p.N % 96 == 0) return launch_conv<T, 2, 2, 8, 3, 2, CV_MAX_HALO>(p, s);
  if (p.N % 48 == 0) return launch_conv<T, 4, 1, 8, 3, 2, CV_MAX_HALO>(p, s);
  return launch_conv<T, 4, 1, 8, 2, 2, CV_MAX_HALO>(p, s);
}

}  // namespace itts

using namespace itts;

extern "C" int itts_gemm_conv(const itts_conv_args* a, void* stream) {
  ITTS_REQUIRE(a && a->x && a->wp && a->y, "itts_gemm_conv: null pointer");
  ITTS_REQUIRE(a->B >= 0 && a->Tin >= 0 && a->Tout >= 0 && a->Cin > 0 && a->N > 0 && a->taps > 0 && a->dil >= 0,
               "itts_gemm_conv: bad shape");
  ITTS_REQUIRE(a->Cin % 8 == 0, "itts_gemm_conv: Cin=%d must be a multiple of 8", a->Cin);
  ITTS_REQUIRE((a->taps - 1) * a->dil <= CV_MAX_HALO, "itts_gemm_conv: (taps-1)*dil = %d exceeds %d", (a->taps - 1) * a->dil,
               CV_MAX_HALO);
  if (a->B == 0 || a->Tout == 0) return ITTS_OK;
  ConvParams p;
  p.B = a->B;
  p.Tin = a->Tin;
  p.Tout = a->Tout;
  p.Cin = a->Cin;
  p.N = a->N;
  p.taps = a->taps;
  p.off0 = a->off0;
  p.dil = a->dil;
  p.x = a->x;
  p.x_bstride = a->x_bstride;
  p.wp = a->wp;
  p.bias = a->bias;
  p.bias2 = a->bias2;
  p.act = a->act;
  p.y = a->y;
  p.y_f32 = a->y_f32;
  p.y_bstride = a->y_bstride;
  p.y_shift = a->y_shift;
  p.y_limit = a->y_limit;
  p.resid = a->resid;
  p.accumulate = a->accumulate;
  p.scale = a->scale;
  const int ks = a->dtype == ITTS_F32 ? 16 : 32;
  p.NT = (a->N + 15) / 16;
  p.KT = (a->Cin + ks - 1) / ks;
  hipStream_t s = (hipStream_t)stream;
  switch (a->dtype) {
    case ITTS_F32:
      return dispatch_conv<float>(p, s);
    case ITTS_BF16:
      return dispatch_conv<bf16_t>(p, s);
    case ITTS_F16:
      return dispatch_conv<f16_t>(p, s);
  }
  ITTS_REQUIRE(false, "itts_gemm_conv: unknown dtype %d", a->dtype);
}
