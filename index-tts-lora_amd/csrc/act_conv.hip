// Anti-aliased SnakeBeta activation FUSED into the narrow convolution that consumes it (fp16, C = 24 / 48: the last two
// upsampling stages of BigVGAN, 143 360 / 71 680 rows x 32 batch elements).
//
// An AMP block is [activation -> conv k, dil d -> activation -> conv k, dil 1 (+ residual)] x 3; un-fused that is 9 passes over
// the [B][T][C] tensor per group (2 + 2 + 2 + 3), and at C <= 48 every one of these kernels is bandwidth-shaped.  Here the
// activation's output never exists in HBM: a workgroup computes the activated rows its convolution tile needs (its BM output
// rows plus the (taps - 1) * dil halo) straight into the LDS row tile of the narrow convolution kernel, then runs that kernel's
// MFMA loop and epilogue over it: 5 passes per group.  Both halves are the existing kernels' code paths:
//   activation  = aa_snake_mfma_kernel's tile routine (elementwise.hip): both FIRs as banded-matrix MFMAs over row-major LDS
//                 images read through ds_read_b64_tr_b16, the snake term on the VALU; output rows go to LDS (rounded to fp16
//                 exactly as the stand-alone kernel stores them, rows outside the sequence as the convolution's zero padding)
//   convolution = conv_narrow_lds_kernel's loop (gemm_conv.hip): weights resident in LDS, weights as the A operand (a lane ends
//                 with 4 consecutive output channels of one row), residual / accumulate operands requested a tile ahead
// so the result is bit-identical to the two launches (tests/test_kernels_gpu.py).  The activation works in 128-row sub-tiles; a
// convolution tile of BM rows takes ceil((BM + halo) / 128) of them, the last one cut to the row blocks that are needed.
// Follows indextts/BigVGAN/models.py:65-74 (AMPBlock1.forward) and alias_free_torch/act.py:10-28.
#include "common.h"
#include "aa_tile.h"
#include <mutex>

namespace itts {

constexpr int AC_MAX_HALO = 64;     // (taps - 1) * dil must not exceed this

struct ActConvParams {
  int B, T, C;
  int taps, off0, dil;
  const f16_t* x;            // input of the activation [B][T][C]
  const void* wp;            // packed convolution weights
  const float* bias;
  f16_t* y;                  // [B][T][C]
  const f16_t* resid;
  int accumulate;
  float scale;
  const int32_t* valid_rows;
  const float* alpha_log;
  const float* beta_log;
  Fir24 f;
  int MB;                    // convolution tiles per batch element
};

template <int CIN>
struct ActConvGeo {
  static constexpr int NCB = (CIN + 15) / 16;                                // 16-channel blocks of the activation: 2 / 3
  static constexpr int DB = CIN * 2;                                         // data bytes of a convolution-tile row
  static constexpr int RSB = ((DB + 16) >> 4) & 1 ? DB + 16 : DB + 32;       // its stride: an odd number of 16-byte units
  static constexpr int KT = (CIN + 31) / 32, NT = (CIN + 15) / 16;
};

template <int CIN, int TAPS, int TM>
__global__ __launch_bounds__(256) void act_conv_kernel(ActConvParams p) {
  typedef ActConvGeo<CIN> CG;
  constexpr int NCB = CG::NCB, KT = CG::KT, NT = CG::NT, rsb = CG::RSB;
  typedef AaMfma<NCB> G;
  constexpr int TT = G::TT, CS = G::CS, XR = G::XR, NUB = G::NUB, RS = G::RS;
  constexpr int NW = 4, BM = NW * TM * 16;
  constexpr int WB = TAPS * NT * KT * 1024;
  constexpr int TROWS = BM + AC_MAX_HALO + 16;
  typedef f16_t h4 __attribute__((ext_vector_type(4)));
  typedef f16x8 frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char ac_lds[];
  unsigned char* wl = ac_lds;                                        // [TAPS][NT][KT] 1-KiB weight blocks
  unsigned char* tile = ac_lds + WB;                                 // [TROWS][rsb]: the activated rows of this convolution tile
  f16_t* Xi = reinterpret_cast<f16_t*>(tile + TROWS * rsb);          // [XR][RS]
  f16_t* Si = Xi + XR * RS;                                          // [SR][RS]
  __shared__ float taps[24];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  constexpr unsigned OOB = 0xFFFFFFFFu;

  for (int off = tid * 16; off < WB; off += 256 * 16) st16(wl + off, ld16<frag>((const unsigned char*)p.wp + off));
  for (int off = tid * 16; off < TROWS * rsb; off += 256 * 16) st16(tile + off, zero_frag<frag>());
  if (tid < 12) taps[tid] = 2.0f * p.f.up[tid];
  else if (tid < 24) taps[tid] = p.f.down[tid - 12];
  __syncthreads();

  // ---- activation: constant tap fragments and snake parameters (as aa_snake_mfma_kernel)
  f16x8 wu_hi, wu_lo, wd_hi[2], wd_lo[2];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int kk = 8 * g + e;
    const int d = kk - 8 - (r >> 1);
    const int ju = (r & 1) ? 6 - 2 * d : 5 - 2 * d;
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
  f32x4 ca[NCB], cbv[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = min(16 * cb + 4 * g + e, p.C - 1);
      ca[cb][e] = __expf(p.alpha_log[ch]) * 0.15915494309189535f;
      cbv[cb][e] = __frcp_rn(__expf(p.beta_log[ch]) + 1e-9f);
    }
  constexpr int C4 = CS / 4;
  constexpr int RPP = 256 / C4;
  constexpr int NP = (XR + RPP - 1) / RPP;
  const int pr = tid / C4, pc4 = tid - pr * C4;
  const bool pact = pr < RPP && pc4 * 4 < p.C;
  if (pr < 16 && pc4 < C4) *reinterpret_cast<h4*>(&Si[(2 * TT + 16 + pr) * RS + pc4 * 4]) = h4{0, 0, 0, 0};

  // The activated rows of a tile start at a multiple of 16 (`shift` rows in front of the first row the convolution reads): the
  // activation's 16-row blocks then coincide with the blocks of the stand-alone kernel (tiles at multiples of 128), every
  // product sits in the same k slot of the same MFMA, and the two forms round identically.
  const int halo = (TAPS - 1) * p.dil;
  const int shift = ((p.off0 % 16) + 16) % 16;
  const int nyb_tot = (BM + halo + shift + 15) >> 4;  // 16-row blocks of activated rows a convolution tile reads
  const int nsub = (nyb_tot + 7) >> 3;                // 128-row activation sub-tiles
  const int ntiles = p.MB * p.B;
  auto tile_geo = [&](int tile_id, int& b, int& t0c, int& T_len) {
    b = tile_id / p.MB;
    t0c = (tile_id - b * p.MB) * BM;
    T_len = p.valid_rows != nullptr ? min(max(p.valid_rows[b], 0), p.T) : p.T;
  };
  // x rows of one activation sub-tile, requested one sub-tile ahead (they arrive under the current one's MFMAs)
  h4 xv[NP];
  auto request = [&](int tile_id, int sub) {
    int b, t0c, T_len;
    tile_geo(tile_id, b, t0c, T_len);
    const int ta = t0c + p.off0 - shift + TT * sub;
    const f16_t* xc = p.x + (int64_t)b * p.T * p.C + pc4 * 4;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int row = min(max(ta - 12 + pr + q * RPP, 0), max(T_len - 1, 0));
      xv[q] = (pact && T_len > 0) ? *reinterpret_cast<const h4*>(xc + (int64_t)row * p.C) : h4{0, 0, 0, 0};
    }
  };
  // ---- convolution epilogue operands, requested a whole tile ahead (as conv_narrow_lds_kernel)
  const bool has_r = p.resid != nullptr, has_a = p.accumulate != 0;
  f32x4 bs[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias ? p.bias : (const float*)p.wp), 0, p.bias ? p.C * 4 : 0, 0x00020000);
    bs[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb1, (unsigned)((nt * 16 + g * 4) * 4), 0, 0));
  }
  u32x2 rres[TM][NT], racc[TM][NT];
  auto epi_request = [&](int tile_id) {
    int b, t0c, T_len;
    tile_geo(tile_id, b, t0c, T_len);
    const int row0 = t0c + wave * (TM * 16);
    const int64_t lim = (int64_t)p.T * p.C * 2;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)b * p.T * p.C, 0, (int)lim, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<f16_t*>(p.resid ? p.resid : p.y) + (int64_t)b * p.T * p.C, 0, (int)lim, 0x00020000);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int t = row0 + tm * 16 + r, col0 = nt * 16 + g * 4;
        const bool ok = t < p.T && col0 < p.C;
        const unsigned off = ok ? (unsigned)(((int64_t)t * p.C + col0) * 2) : OOB;
        rres[tm][nt] = has_r ? __builtin_amdgcn_raw_buffer_load_b64(rr, off, 0, 0) : u32x2{0u, 0u};
        racc[tm][nt] = has_a ? __builtin_amdgcn_raw_buffer_load_b64(ry, off, 0, 0) : u32x2{0u, 0u};
      }
  };

  int tile_id = blockIdx.x;
  if (tile_id < ntiles) {
    request(tile_id, 0);
    epi_request(tile_id);
  }
  for (; tile_id < ntiles; tile_id += gridDim.x) {
    int b, t0c, T_len;
    tile_geo(tile_id, b, t0c, T_len);
    const int next_tile = tile_id + (int)gridDim.x;
    const bool live = t0c + p.off0 < T_len;      // (ragged batch: a tile that only sees the padding computes and stores nothing)
    // ================= activation: the rows [t0c + off0 - shift, ... + 16 nyb_tot) into the convolution's row tile =================
    for (int sub = 0; sub < nsub; ++sub) {
      const int ta = t0c + p.off0 - shift + TT * sub;
      const int nyb = min(8, nyb_tot - 8 * sub);
      const int nub = min(NUB, 2 * nyb + 2);
      {
        f16_t* xw = &Xi[pr * RS + pc4 * 4];
#pragma unroll
        for (int q = 0; q < NP; ++q)
          if (pr < RPP && pr + q * RPP < XR) *reinterpret_cast<h4*>(xw + q * RPP * RS) = xv[q];
      }
      __syncthreads();     // X committed; every wave is done with the previous sub-tile's S reads (and the previous tile's MFMAs)
      if (sub + 1 < nsub) request(tile_id, sub + 1);
      else if (next_tile < ntiles) request(next_tile, 0);
      const bool inside = live && ta < T_len;    // a sub-tile past the sequence end is all zero padding
      if (inside) {
        for (int ub = wave; ub < nub; ub += 4) {
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
              const float sn = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(u * ca[cb][e]));
              sv[e] = (f16_t)(u + cbv[cb][e] * sn * sn);
            }
            *reinterpret_cast<h4*>(&Si[(16 * ub + r) * RS + 16 * cb + 4 * g]) = sv;
          }
        }
      }
      __syncthreads();
      if (inside) {
        // replicate padding of the UPSAMPLED signal at the sequence ends (tile-uniform conditions); image row of m is m - m_base
        const int m_base = 2 * ta - 8;
        if (m_base < 0) {
          for (int idx = tid; idx < (-m_base) * CS; idx += 256) {
            const int mi = idx / CS, c = idx - mi * CS;
            Si[mi * RS + c] = Si[(-m_base) * RS + c];
          }
          __syncthreads();
        }
        if (m_base + 2 * TT + 16 > 2 * T_len) {
          const int last = 2 * T_len - 1 - m_base;
          const int n = 2 * TT + 16 - 1 - last;
          for (int idx = tid; idx < n * CS; idx += 256) {
            const int k = idx / CS, c = idx - k * CS;
            Si[(last + 1 + k) * RS + c] = Si[last * RS + c];
          }
          __syncthreads();
        }
      }
      // y blocks -> rows 128 sub + 16 yb + r of the convolution tile (zeros outside the sequence: the convolution's padding)
      for (int yb_ = wave; yb_ < nyb; yb_ += 4) {
        const int t = ta + 16 * yb_ + r;
        const bool in_seq = inside && t >= 0 && t < T_len;
        unsigned char* trow = tile + (TT * sub + 16 * yb_ + r) * rsb;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          if (inside) {
            const f16x8 s0 = aa_tr_frag<RS>(Si, 32 * yb_, 16 * cb, lane);
            const f16x8 s1 = aa_tr_frag<RS>(Si, 32 * yb_ + 32, 16 * cb, lane);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(s0, wd_hi[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(s0, wd_lo[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(s1, wd_hi[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(s1, wd_lo[1], acc, 0, 0, 0);
          }
          const int ch = 16 * cb + 4 * g;
          if (ch < CIN) {
            const h4 o = in_seq ? h4{(f16_t)acc[0], (f16_t)acc[1], (f16_t)acc[2], (f16_t)acc[3]} : h4{0, 0, 0, 0};
            *reinterpret_cast<h4*>(trow + ch * 2) = o;
          }
        }
      }
    }
    __syncthreads();       // the row tile is complete
    // ================= convolution over the row tile (conv_narrow_lds_kernel's loop and epilogue) =================
    const int row0 = t0c + wave * (TM * 16);
    if (live && row0 + p.off0 < T_len) {
      f32x4 acc[TM][NT];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned char* ar = tile + (wave * (TM * 16) + r + shift) * rsb + g * 16;
#pragma unroll
      for (int j = 0; j < TAPS; ++j) {
        frag af[TM][KT];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int ks = 0; ks < KT; ++ks) af[tm][ks] = ld16<frag>(ar + (tm * 16 + j * p.dil) * rsb + ks * 64);
        const unsigned char* wb = wl + (size_t)j * NT * KT * 1024 + lane * 16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int ks = 0; ks < KT; ++ks) {
            const frag bf = ld16<frag>(wb + (nt * KT + ks) * 1024);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) acc[tm][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf, af[tm][ks], acc[tm][nt], 0, 0, 0);
          }
      }
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)b * p.T * p.C, 0, (int)((int64_t)p.T * p.C * 2), 0x00020000);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int t = row0 + tm * 16 + r, col0 = nt * 16 + g * 4;
          const bool ok = t < p.T && col0 < p.C;
          const h4 rv = __builtin_bit_cast(h4, rres[tm][nt]), av = __builtin_bit_cast(h4, racc[tm][nt]);
          h4 o;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const float v = acc[tm][nt][jj] + bs[nt][jj];
            o[jj] = (f16_t)fmaf(v + (has_r ? (float)rv[jj] : 0.f), p.scale, has_a ? (float)av[jj] : 0.f);   // as conv_epilogue_impl
          }
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ry, ok ? (unsigned)(((int64_t)t * p.C + col0) * 2) : OOB, 0, 0);
        }
    }
    if (next_tile < ntiles) epi_request(next_tile);
    // (no barrier here: the next tile's first two barriers stand between this tile's MFMA reads and the next writes of the row tile)
  }
}

static int ac_num_cus() {
  static int n = [] {
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipGetLastError();
    return v > 0 ? v : 256;
  }();
  return n;
}

template <int CIN, int TAPS, int TM>
static int launch_act_conv(ActConvParams& p, hipStream_t s) {
  typedef ActConvGeo<CIN> CG;
  typedef AaMfma<CG::NCB> G;
  constexpr int BM = 4 * TM * 16;
  constexpr size_t ldsb = (size_t)TAPS * CG::NT * CG::KT * 1024 + (size_t)(BM + AC_MAX_HALO + 16) * CG::RSB + G::LDS;
  static_assert(ldsb <= 160 * 1024 - 512, "act_conv: LDS budget");
  p.MB = (p.T + BM - 1) / BM;
  const int64_t tiles = (int64_t)p.MB * p.B;
  static std::once_flag attr;
  std::call_once(attr, [] {
    (void)hipFuncSetAttribute((const void*)act_conv_kernel<CIN, TAPS, TM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    (void)hipGetLastError();
  });
  static thread_local int occ_wgs = 0;
  if (occ_wgs == 0) {
    int q = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, (const void*)act_conv_kernel<CIN, TAPS, TM>, 256, ldsb) != hipSuccess || q < 1) q = 1;
    (void)hipGetLastError();
    occ_wgs = q;
  }
  int64_t grid = (int64_t)ac_num_cus() * occ_wgs;
  if (grid > tiles) grid = tiles;
  hipLaunchKernelGGL((act_conv_kernel<CIN, TAPS, TM>), dim3((unsigned)grid), dim3(256), ldsb, s, p);
  return check_launch("itts_act_conv");
}

}  // namespace itts

using namespace itts;

extern "C" int itts_act_conv_supported(int dtype, int C, int taps, int dil) {
  return dtype == ITTS_F16 && (C == 24 || C == 48) && (taps == 3 || taps == 7 || taps == 11) && dil >= 1 && (taps - 1) * dil <= AC_MAX_HALO;
}

extern "C" int itts_act_conv(const itts_conv_args* a, const float* alpha_log, const float* beta_log, const float* up_filter12,
                             const float* down_filter12, void* stream) {
  ITTS_REQUIRE(a && a->x && a->wp && a->y && alpha_log && beta_log && up_filter12 && down_filter12, "itts_act_conv: null pointer");
  ITTS_REQUIRE(itts_act_conv_supported(a->dtype, a->Cin, a->taps, a->dil) && a->N == a->Cin,
               "itts_act_conv: built for fp16, C = 24 / 48, 3 / 7 / 11 taps, (taps - 1) * dil <= %d (ask itts_act_conv_supported)", AC_MAX_HALO);
  ITTS_REQUIRE(a->Tin == a->Tout && a->B >= 0 && a->Tin >= 0 && !a->y_f32 && a->bias2 == nullptr && a->act == 0 && a->ksplit <= 1 &&
                   a->y_shift == 0 && a->y_limit == (int64_t)a->Tout * a->N && a->y_bstride == (int64_t)a->Tout * a->N &&
                   a->x_bstride == (int64_t)a->Tin * a->Cin,
               "itts_act_conv: a plain same-shape convolution (Tin == Tout, dense batches, T-typed y, no per-batch bias / activation)");
  ITTS_REQUIRE((int64_t)a->Tout * a->N * 2 < (1ll << 31), "itts_act_conv: a batch element must stay under 2 GiB");
  if (a->B == 0 || a->Tout == 0) return ITTS_OK;
  ActConvParams p;
  p.B = a->B; p.T = a->Tout; p.C = a->Cin;
  p.taps = a->taps; p.off0 = a->off0; p.dil = a->dil;
  p.x = (const f16_t*)a->x; p.wp = a->wp; p.bias = a->bias;
  p.y = (f16_t*)a->y; p.resid = (const f16_t*)a->resid; p.accumulate = a->accumulate; p.scale = a->scale;
  p.valid_rows = a->valid_rows; p.alpha_log = alpha_log; p.beta_log = beta_log;
  for (int i = 0; i < 12; ++i) {
    p.f.up[i] = up_filter12[i];
    p.f.down[i] = down_filter12[i];
  }
  p.MB = 0;
  hipStream_t s = (hipStream_t)stream;
#define ITTS_AC(C_, K_, TM_) if (a->Cin == C_ && a->taps == K_) return launch_act_conv<C_, K_, TM_>(p, s)
  ITTS_AC(24, 3, 2);
  ITTS_AC(24, 7, 2);
  ITTS_AC(24, 11, 2);
  ITTS_AC(48, 3, 2);
  ITTS_AC(48, 7, 2);
  ITTS_AC(48, 11, 2);
#undef ITTS_AC
  ITTS_REQUIRE(false, "itts_act_conv: no instantiation");
}
