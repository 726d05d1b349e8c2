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
#include <mutex>

#include "common.h"
#include <type_traits>

namespace itts {

#ifndef ITTS_NARROW_C96
#define ITTS_NARROW_C96 1   // build-time A/B: the LDS-staged narrow kernel for C = 96, 3 taps (246-266 -> 198-219 us per layer at batch 32 x 35 840 rows; one row tile per wave, 8 waves: 2 or 4 row tiles per wave spill)
#endif
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
  int ksplit;      // plain GEMM only: > 1 = the batch index of a tile is a K SLICE (slabs, see itts_conv_args.ksplit)
  int MB, NB, GM;  // m-blocks per batch element, n-blocks, m-blocks per L2 group (XCD-aware tile order)
  const int32_t* valid_rows;   // [B] or null: input rows >= valid_rows[b] read as zeros, tiles wholly beyond are skipped
  int exp;         // diagnostic build only (itts_debug_set key 5): ablation switches of the tiled kernel, 0 in the product
#if ITTS_STAMPS
  unsigned long long* stamps;   // diagnostic build: 16 x u64 per workgroup (itts_debug_stamps_conv)
#endif
};

#if ITTS_STAMPS
unsigned long long* g_stamp_buf_conv = nullptr;
// stamps live in LDS, not in registers: 16 x u64 of VGPRs pushed the big tiles into scratch in the diagnostic build
#define CSTAMP(i)                                                                                        \
  do {                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    if (p.stamps != nullptr && threadIdx.x == 0) {                                                       \
      unsigned long long t_;                                                                             \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
      st_[i] = t_;                                                                                       \
    }                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  } while (0)
#else
#define CSTAMP(i) do { } while (0)
#endif
#if ITTS_DIAG
int g_conv_exp = 0;
#define ITTS_CONV_EXP(p) ((p).exp)
#else
#define ITTS_CONV_EXP(p) 0
#endif

// Ragged batches: valid input rows of batch element b (the rest is the convolution's zero padding)
__device__ __forceinline__ int conv_valid_rows(const ConvParams& p, int b) {
  return p.valid_rows != nullptr ? min(max(p.valid_rows[b], 0), p.Tin) : p.Tin;
}

// XCD-aware tile order (speed only): workgroups are dealt round-robin to the 8 XCDs, so give each XCD a CONTIGUOUS
// run of the tile sequence, and order the sequence so that 32 consecutive tiles form a compact GM x (32/GM) patch of
// the output -- its activation rows and weight columns then stay in that XCD's 4-MiB L2 instead of being re-fetched
// from the Infinity Cache by every tile.
__device__ __forceinline__ void tile_of_workgroup(const ConvParams& p, int bid, int nwg, int& mblk, int& nblk, int& b) {
  const int xcd = bid & 7, q = bid >> 3, base = nwg >> 3, rem = nwg & 7;
  const int L = xcd * base + min(xcd, rem) + q;
  const int MBt = p.MB * p.B;
  const int per_group = p.GM * p.NB;
  const int mgi = L / per_group, r1 = L - mgi * per_group;
  const int gm_eff = min(p.GM, MBt - mgi * p.GM);
  nblk = r1 / gm_eff;
  const int m = mgi * p.GM + (r1 - nblk * gm_eff);
  b = m / p.MB;
  mblk = m - b * p.MB;
}

// Epilogue.  The MFMAs ran with the weight fragment as the A operand, so each lane holds the transposed tile:
// acc[tm][tn][jj] = out(row = tile row (lane & 15), channel = 16*n-tile + 4*(lane >> 4) + jj), i.e. FOUR CONSECUTIVE
// CHANNELS of one output row -> one 8/16-byte access per tile for residual, accumulate and store instead of four
// 2/4-byte ones.  Range-checked buffer accesses throughout (invalid elements get offset OOB: loads return 0, stores
// are dropped; a null bias is a zero-length buffer), all residual / accumulate operands of a batch of m-tiles
// requested before the first use.  VEC needs N, y_shift, y_limit to be multiples of 4 (validity is then per quad).
template <typename T, typename YT, bool VEC, int TM, int TN>
__device__ __forceinline__ void conv_epilogue_impl(const ConvParams& p, f32x4 (&acc)[TM][TN], int b, int row0, int nt0, int g, int r) {
  constexpr unsigned OOB = 0xFFFFFFFFu;
  constexpr int ES = (int)sizeof(YT);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      (YT*)p.y + (int64_t)b * p.y_bstride, 0, (int)(p.y_limit * ES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<YT*>((const YT*)(p.resid ? p.resid : p.y)) + (int64_t)b * p.y_bstride, 0, (int)(p.y_limit * ES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.bias ? p.bias : (const float*)p.wp), 0, p.bias ? p.N * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.bias2 ? p.bias2 + (int64_t)b * p.N : (const float*)p.wp), 0, p.bias2 ? p.N * 4 : 0, 0x00020000);
  auto ld1 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off) -> float {
    if constexpr (ES == 4) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
    else return Elem<YT>::to_f(__builtin_bit_cast(YT, __builtin_amdgcn_raw_buffer_load_b16(rs, off, 0, 0)));
  };
  auto st1 = [&](unsigned off, float v) {
    if constexpr (ES == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, Elem<YT>::from_f(v)), ry, off, 0, 0);
  };
  auto ld4 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off, float (&o)[4]) {
    if constexpr (ES == 4) {
      f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = v[i];
    } else {
      typedef YT yt4 __attribute__((ext_vector_type(4)));
      yt4 v = __builtin_bit_cast(yt4, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (float)v[i];
    }
  };
  auto st4 = [&](unsigned off, const float (&o)[4]) {
    if constexpr (ES == 4) {
      f32x4 v = {o[0], o[1], o[2], o[3]};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, off, 0, 0);
    } else {
      typedef YT yt4 __attribute__((ext_vector_type(4)));
      yt4 v = {(YT)o[0], (YT)o[1], (YT)o[2], (YT)o[3]};
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), ry, off, 0, 0);
    }
  };
  const bool has_r = p.resid != nullptr, has_a = p.accumulate != 0;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col0 = (nt0 + tn) * 16 + g * 4;
    const bool cok = (nt0 + tn) < p.NT;
    float bs[4];
    {
      // zero-length / short descriptors make out-of-range channels (and a null bias) read as 0
      unsigned boff = cok ? (unsigned)(col0 * 4) : OOB;
      if constexpr (VEC) {
        f32x4 v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb1, boff, 0, 0));
        f32x4 v2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb2, boff, 0, 0));
#pragma unroll
        for (int i = 0; i < 4; ++i) bs[i] = v1[i] + v2[i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned o = cok ? (unsigned)((col0 + i) * 4) : OOB;
          bs[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb1, o, 0, 0)) +
                  __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb2, o, 0, 0));
        }
      }
    }
    constexpr int TC = TM < 4 ? TM : 4;  // m-tiles per batch of in-flight epilogue loads
#pragma unroll
    for (int tm0 = 0; tm0 < TM; tm0 += TC) {
      unsigned offs[TC][VEC ? 1 : 4];
      float rv[TC][4], av[TC][4];
#pragma unroll
      for (int u = 0; u < TC; ++u) {
        const int t = row0 + (tm0 + u) * 16 + r;
        const int64_t flat = (int64_t)t * p.N + col0 + p.y_shift;
        const bool rok = cok && (t < p.Tout);
        if constexpr (VEC) {
          bool ok = rok && (col0 < p.N) && (flat >= 0) && (flat < p.y_limit);
          offs[u][0] = ok ? (unsigned)(flat * ES) : OOB;
          if (has_r) ld4(rr, offs[u][0], rv[u]);
          if (has_a) ld4(ry, offs[u][0], av[u]);
        } else {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            bool ok = rok && (col0 + jj < p.N) && (flat + jj >= 0) && (flat + jj < p.y_limit);
            offs[u][jj] = ok ? (unsigned)((flat + jj) * ES) : OOB;
            if (has_r) rv[u][jj] = ld1(rr, offs[u][jj]);
            if (has_a) av[u][jj] = ld1(ry, offs[u][jj]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < TC; ++u) {
        float o[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          float v = acc[tm0 + u][tn][jj] + bs[jj];
          if (p.act == 1) v = gelu_new(v);
          v = fmaf(v + (has_r ? rv[u][jj] : 0.f), p.scale, has_a ? av[u][jj] : 0.f);  // explicit fma: same rounding in every instantiation
          o[jj] = v;
        }
        if constexpr (VEC) st4(offs[u][0], o);
        else {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) st1(offs[u][jj], o[jj]);
        }
      }
    }
  }
}

// row0 = first output row of this wave's tile column; YT = storage type of y / resid (float or T)
template <typename T, int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x4 (&acc)[TM][TN], int b, int row0, int nt0, int g, int r) {
  const bool vec = ((p.N | p.y_shift | p.y_limit) & 3) == 0;
  if (p.y_f32) {
    if (vec) conv_epilogue_impl<T, float, true>(p, acc, b, row0, nt0, g, r);
    else conv_epilogue_impl<T, float, false>(p, acc, b, row0, nt0, g, r);
  } else {
    if (vec) conv_epilogue_impl<T, T, true>(p, acc, b, row0, nt0, g, r);
    else conv_epilogue_impl<T, T, false>(p, acc, b, row0, nt0, g, r);
  }
}

static int conv_num_cus() {
  static int n = 0;
  static std::once_flag once;
  std::call_once(once, [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
    (void)hipGetLastError();
  });
  return n;
}

// CK = k-steps of channels staged per chunk (2 for convolutions, whose taps multiply the MFMA work per chunk; 4 for
// plain GEMMs).  HALO = compile-time bound on (taps-1)*dil (0 for plain GEMMs) that sizes the staging registers.
// Waves per SIMD the register budget is sized for: 8-wave workgroups put 2 waves on each SIMD (one wave's MFMAs run under
// the other's LDS reads / address arithmetic); 4-wave workgroups with a small accumulator tile ask for 2 workgroups per CU.
template <int WM, int WN, int TM, int TN>
struct ConvOcc {
  static constexpr int NW = WM * WN;
  static constexpr int WPS = (NW == 8) ? 2 : ((TM * TN <= 16 && !(WM == 4 && WN == 1)) ? 2 : 1);   // the 4x1 stacks stage 10 row fragments per thread
};

// PERSISTENT, cross-tile pipelined: the grid is one round of resident workgroups (launch_conv asks the occupancy API) and
// a workgroup walks the tile sequence with stride gridDim.x.  The activation rows of the NEXT tile's first chunk are
// requested while the current tile's last chunk is multiplied, the weight fragments of the next TAP (CK k-steps x TN
// column blocks) are requested at the start of the current tap -- straight across chunk and tile boundaries -- and the
// epilogue's stores drain under the next tile's work.
//
// Everything that addresses weights or picks the LDS row of a tap is WAVE-UNIFORM and kept in SGPRs (the wave index goes
// through readfirstlane so the compiler can prove it): a weight load is `buffer_load  v(lane*16), s(block offset)`, the tap
// cursor is three scalar adds.  The first version of this loop carried the cursor in VGPRs (it inherited divergence from
// threadIdx.x >> 6): ~100 VALU instructions per 16 MFMAs, and in-kernel stamps showed the tap loop at 390 ns per k-step
// with the MFMAs, weight loads and LDS reads all REMOVED (profiles/r02_conv_ablation.txt) -- index arithmetic, not data
// movement, bounded the kernel.
template <typename T, int WM, int WN, int TM, int TN, int CK, int HALO>
__global__ __launch_bounds__(WM * WN * 64, (ConvOcc<WM, WN, TM, TN>::WPS)) void gemm_conv_kernel(ConvParams p) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  constexpr int E = EL::E, KS = EL::KS;
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
  constexpr int SEGS = CK * 4;                 // 16-byte segments per staged row
  constexpr int ROWB = CK * 64 + 16;           // LDS bytes per staged row (payload + 16 B pad: conflict-free b128 reads)
  constexpr int NTH = WM * WN * 64;            // threads per workgroup (4 or 8 waves)
  constexpr int MAXST = ((BM + HALO) * SEGS + NTH - 1) / NTH;
  static_assert(WM * WN >= 2 && WM * WN <= 8, "2 to 8 waves per workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if ITTS_STAMPS
  __shared__ unsigned long long st_[16];
  if (tid < 16) st_[tid] = 0;
  if (p.stamps != nullptr && tid == 0) st_[14] = __builtin_amdgcn_s_memrealtime();
#endif
  CSTAMP(0);
  const int g = lane >> 4, r = lane & 15;
  const int wm = wave / WN, wn = wave % WN;
  const int total = p.MB * p.NB * p.B;         // tiles; tile L of the XCD-aware order is handled by workgroup L % gridDim.x
  const int HR = BM + (p.taps - 1) * p.dil;   // staged rows
  const int NC = (p.KT + CK - 1) / CK;        // channel chunks
  const int nkk_last = p.KT - (NC - 1) * CK;  // k-steps of the last chunk (== CK unless KT % CK)
  // Range-checked buffer descriptors: out-of-range rows (conv zero padding, M tail) and disabled lanes read zeros with
  // no branch around the load, so the compiler keeps counted waits instead of draining the queue at every join.
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.wp), 0, (int)((int64_t)p.taps * p.NT * p.KT * 1024), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFFFu;
  const int w_tap = p.NT * p.KT * 1024;        // bytes between the weight blocks of neighbouring taps
  const int w_col = p.KT * 1024;               // ... of neighbouring 16-column blocks
  const unsigned w_lane = (unsigned)lane * 16;

  struct Tile {
    int b, t0, nt0, vrows;
  };
  auto tile_at = [&](int L) {
    int mblk, nblk, b;
    tile_of_workgroup(p, L, total, mblk, nblk, b);
    return Tile{b, mblk * BM, nblk * (BN / 16) + wn * TN, conv_valid_rows(p, b)};
  };
  // ragged batches: a tile whose every tap reads only the zero padding past its batch element's valid rows is not computed
  // (its output rows are left as they are: nobody reads them).  next_from(L) = the first tile at or after L, in this
  // workgroup's stride, that has to be computed.
  auto next_from = [&](int L) {
    if (p.valid_rows != nullptr) {
      while (L < total) {
        const Tile t = tile_at(L);
        if (t.t0 + p.off0 < t.vrows) break;
        L += gridDim.x;
      }
    }
    return L;
  };

  f32x4 acc[TM][TN];
  frag stg[MAXST];
  auto prefetch_a = [&](int c, const Tile& tl) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>((const T*)p.x + (int64_t)tl.b * p.x_bstride), 0, (int)((int64_t)tl.vrows * p.Cin * (int)sizeof(T)), 0x00020000);
#pragma unroll
    for (int q = 0; q < MAXST; ++q) {
      int idx = tid + q * NTH;
      int i = idx / SEGS, seg = idx - i * SEGS;
      int tin = tl.t0 + p.off0 + i;
      int col = c * (CK * KS) + seg * E;
      bool ok = (i < HR) && (col < p.Cin);
      unsigned off = ok ? (unsigned)((tin * p.Cin + col) * (int)sizeof(T)) : OOB;  // tin < 0 wraps to a huge offset
      stg[q] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
    }
  };
  auto commit_a = [&]() {
#pragma unroll
    for (int q = 0; q < MAXST; ++q) {
      int idx = tid + q * NTH;
      int i = idx / SEGS, seg = idx - i * SEGS;
      if (i < HR) st16(lds + i * ROWB + seg * 16, stg[q]);
    }
  };
  // Weight cursor (scalar): w_off = byte offset of block (tap fj, column block 0 of this wave's tile, k-step fc*CK).
  // A column block past NT (N tail) is clamped onto the last one: its products land in accumulators the epilogue drops.
  // k-steps past KT in the last chunk read the neighbouring block (or zeros past the end): their activations are zero.
  int w_tn[TN];
  int fc = 0, fj = 0, f_next = 0, w_off = 0;
  auto w_tile = [&](int nt0) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) w_tn[tn] = min(nt0 + tn, p.NT - 1) * w_col;
  };
  auto fetch_b = [&](frag (&bf)[CK][TN]) {
#pragma unroll
    for (int kk = 0; kk < CK; ++kk)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        bf[kk][tn] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rw, w_lane, w_off + w_tn[tn] + kk * 1024, 0));
      }
    // advance one tap; at the end of a tile move on to the column blocks of the workgroup's next tile (or stay: the
    // loads after the last tile are never used)
    w_off += w_tap;
    if (++fj == p.taps) {
      fj = 0;
      w_off += CK * 1024 - p.taps * w_tap;
      if (++fc == NC) {
        fc = 0;
        w_off = 0;
        if (f_next < total) w_tile(tile_at(f_next).nt0);
        f_next = next_from(f_next + gridDim.x);
      }
    }
  };
  const unsigned char* a_lane = lds + (wm * TM * 16 + r) * ROWB + g * 16;
  const int a_tap = p.dil * ROWB;
  auto load_a = [&](frag (&af)[TM], int j, int kk) {   // j, kk wave-uniform
    const unsigned char* ab = a_lane + j * a_tap + kk * 64;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) af[tm] = ld16<frag>(ab + tm * 16 * ROWB);
  };
  auto mma_all = [&](frag (&af)[TM], frag (&bf)[TN]) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = EL::mma(bf[tn], af[tm], acc[tm][tn]);  // weights as A: transposed tile
  };

  frag bA[CK][TN], bB[CK][TN];   // weight fragments of the current / the next tap (roles swap every tap)
  frag a0[TM], a1[TM];           // activation fragments ping-pong between k-steps
  // One tap of a FULL chunk (CK k-steps): the next tap's weights are requested first, then every k-step issues the LDS
  // reads of the following k-step before its own MFMAs.  P = parity of the tap inside the statically unrolled pair, so
  // that the a0/a1 roles are compile-time: k-step (j, kk) uses set (P*CK + kk) & 1.
  auto tap_full = [&](auto Ptag, frag (&cur)[CK][TN], frag (&nxt)[CK][TN], int j) {
    constexpr int P = decltype(Ptag)::value;
    fetch_b(nxt);
    const bool more = j + 1 < p.taps;
    // sched_barrier: the scheduler otherwise sinks the early LDS reads down to their first use to save registers, which
    // puts the LDS latency in front of every pair of MFMAs
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < CK; ++kk) {
      const bool cur0 = (((P * CK + kk) & 1) == 0);
      if (kk + 1 < CK) {
        if (cur0) load_a(a1, j, kk + 1); else load_a(a0, j, kk + 1);
      } else if (more) {
        if (cur0) load_a(a1, j + 1, 0); else load_a(a0, j + 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (cur0) mma_all(a0, cur[kk]); else mma_all(a1, cur[kk]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  int L = next_from(blockIdx.x);
  if (L >= total) return;
  Tile cur = tile_at(L);
  w_tile(cur.nt0);
  f_next = next_from(L + gridDim.x);
  fetch_b(bA);
  prefetch_a(0, cur);
  CSTAMP(1);
#if ITTS_STAMPS
  bool first_ = true;
#endif
  while (L < total) {
    const int Ln = next_from(L + gridDim.x);
    const bool has_next = Ln < total;
    Tile nxt = cur;
    if (has_next) nxt = tile_at(Ln);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < NC; ++c) {
#if ITTS_STAMPS
      const bool stc = (c == 1 && first_);   // one steady-state chunk of the first tile, decomposed
      if (stc) CSTAMP(6);
#endif
      // Raw barriers with an LDS-only wait: __syncthreads() would also drain vmcnt, i.e. wait for the weight fragments of the
      // next tap that were requested a moment ago (measured: 3 us per chunk in the two barriers).
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everyone finished reading the previous chunk
      commit_a();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#if ITTS_STAMPS
      if (c == 0 && first_) CSTAMP(2);
      if (stc) CSTAMP(8);
#endif
      if (c + 1 < NC) prefetch_a(c + 1, cur);
      else if (has_next) prefetch_a(0, nxt);   // the next tile's first chunk, under this tile's last chunk + epilogue
      const int nkk = (c + 1 < NC) ? CK : nkk_last;
      if (nkk == CK) {
        // k-step (0, 0) of a chunk uses a0 when the unrolled pair starts at an even tap: (P*CK + 0) & 1 == 0 for P = 0
        load_a(a0, 0, 0);
        int j = 0;
        for (; j + 1 < p.taps; j += 2) {
          tap_full(std::integral_constant<int, 0>{}, bA, bB, j);
          if constexpr ((CK & 1) != 0) {
            tap_full(std::integral_constant<int, 1>{}, bB, bA, j + 1);
          } else {
            tap_full(std::integral_constant<int, 0>{}, bB, bA, j + 1);
          }
#if ITTS_STAMPS
          if (stc && j == 0) CSTAMP(9);
          if (stc && j == 2) CSTAMP(10);
          if (stc && j == 4) CSTAMP(11);
#endif
        }
        if (j < p.taps) {   // odd tap count: the last tap ran out of bA and fetched the next one into bB
          tap_full(std::integral_constant<int, 0>{}, bA, bB, j);
#pragma unroll
          for (int kk = 0; kk < CK; ++kk)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bA[kk][tn] = bB[kk][tn];
        }
#if ITTS_STAMPS
        if (stc) CSTAMP(7);
#endif
      } else {
        // short last chunk (KT % CK != 0): plain order, no register ping-pong
        for (int j = 0; j < p.taps; ++j) {
          fetch_b(bB);
#pragma unroll
          for (int kk = 0; kk < CK; ++kk)
            if (kk < nkk) {
              load_a(a0, j, kk);
              mma_all(a0, bA[kk]);
            }
#pragma unroll
          for (int kk = 0; kk < CK; ++kk)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bA[kk][tn] = bB[kk][tn];
        }
      }
    }
#if ITTS_STAMPS
    if (first_) CSTAMP(3);
#endif
    conv_epilogue<T, TM, TN>(p, acc, cur.b, cur.t0 + wm * TM * 16, cur.nt0, g, r);
#if ITTS_STAMPS
    if (first_) CSTAMP(4);
    first_ = false;
#endif
    cur = nxt;
    L = Ln;
  }
#if ITTS_STAMPS
  if (p.stamps != nullptr && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long te_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te_)::"memory");
    unsigned xcc_, hwid_;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid_));
    st_[5] = te_;
    st_[12] = hwid_;
    st_[13] = xcc_ & 0xF;
    st_[15] = __builtin_amdgcn_s_memrealtime();
  }
  __syncthreads();
  if (p.stamps != nullptr && tid < 16) p.stamps[(size_t)blockIdx.x * 16 + tid] = st_[tid];
#endif
}

template <typename T, int WM, int WN, int TM, int TN, int CK, int HALO, bool PERSIST = false>
static int launch_conv(const ConvParams& p, hipStream_t s) {
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
  int HR = BM + (p.taps - 1) * p.dil;
  size_t ldsb = (size_t)HR * (CK * 64 + 16);
  ConvParams q = p;
  q.MB = (p.Tout + BM - 1) / BM;
  q.NB = (p.N + BN - 1) / BN;
  // weight bytes one n-block touches; keep the n-extent of a 32-tile patch within ~2 MiB of weights
  const int64_t wbytes = (int64_t)BN * p.KT * 64 * p.taps;  // BN/16 n-tiles x KT k-steps x 1 KiB x taps
  int gn = (int)((2 << 20) / (wbytes > 0 ? wbytes : 1));
  gn = gn < 1 ? 1 : (gn > 8 ? 8 : gn);
  int gm = 32 / gn;  // 4..32 m-blocks per patch
  gm = gm >= 32 ? 32 : (gm >= 16 ? 16 : (gm >= 8 ? 8 : 4));
  q.GM = gm;
  const int64_t total = (int64_t)q.MB * q.NB * p.B;
  if (total > 0x7fffffff) {
    set_error("itts_gemm_conv: too many tiles (%lld)", (long long)total);
    return ITTS_ERR_INVALID;
  }
  static std::once_flag attr;   // one-shot per instantiation, safe under concurrent first calls (RequestPool threads)
  std::call_once(attr, [] {
    // 160 KiB minus the static block of the diagnostic build's stamps; a refusal must not stay behind as the thread's last error
    (void)hipFuncSetAttribute((const void*)gemm_conv_kernel<T, WM, WN, TM, TN, CK, HALO>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024 - 256);
    (void)hipGetLastError();
  });
  // PERSIST: one round of resident workgroups walks the tile sequence (a multiple of 8 workgroups, so that a workgroup's
  // tiles stay on its XCD's run of the tile order); nothing waits on another workgroup, so an over-estimate only costs a
  // second round.  Otherwise one workgroup per tile, dispatched by the hardware as CUs free up.
  int64_t g = total;
  if (PERSIST || (ITTS_CONV_EXP(q) & 32)) {
    // one occupancy query per (instantiation, LDS size) and thread: a vocoder pass launches ~100 of these
    static thread_local size_t occ_lds = ~(size_t)0;
    static thread_local int occ_wgs = 1;
    if (occ_lds != ldsb) {
      int q_wgs = 1;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q_wgs, (const void*)gemm_conv_kernel<T, WM, WN, TM, TN, CK, HALO>,
                                                       WM * WN * 64, ldsb) != hipSuccess || q_wgs < 1)
        q_wgs = 1;
      (void)hipGetLastError();
      occ_lds = ldsb;
      occ_wgs = q_wgs;
    }
    const int per_cu = occ_wgs;
    g = (int64_t)conv_num_cus() * per_cu;
    if (g > total) g = total;
  }
  dim3 grid((unsigned)g);
  hipLaunchKernelGGL((gemm_conv_kernel<T, WM, WN, TM, TN, CK, HALO>), grid, dim3(WM * WN * 64), ldsb, s, q);
  return check_launch("itts_gemm_conv");
}

// -------------------------------------------------------------------------------------------------------------------
// Plain GEMM (taps == 1): software-pipelined so that no wait in the k-loop ever drains the memory queue.
//   * activations: chunk c+2 is in flight global -> registers, chunk c+1 is written to the OTHER LDS buffer while chunk c
//     is multiplied out of this one: one barrier per chunk, two chunks (~2 x 1024 MFMA cycles per wave) of latency cover;
//   * weights: ring of 4 fragment sets with STATIC indices (the chunk body is fully unrolled over its 4 k-steps), filled 3
//     steps ahead; the ring slot of step s of the next chunk is refilled right after step s of this chunk has issued.
//   vmcnt is an in-order counter, so the issue order is what makes every wait a counted one: B(c,3) -> A(c+2) ->
//   B(c+1,0..2); the waits inside chunk c are all on loads older than A(c+2).
// One 8-wave workgroup per CU (two 128-register accumulator/operand sets per SIMD would not fit twice).
// -------------------------------------------------------------------------------------------------------------------
template <typename T, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_plain_kernel(ConvParams p) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  constexpr int E = EL::E, KS = EL::KS, CK = 4;
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
  constexpr int SEGS = CK * 4, ROWB = CK * 64 + 16, NTH = WM * WN * 64;
  constexpr int MAXST = (BM * SEGS + NTH - 1) / NTH;
  constexpr int BUFB = BM * ROWB;
  static_assert(WM * WN == 8 || WM * WN == 4, "4 or 8 waves per workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;
  const int wm = wave / WN, wn = wave % WN;
  int mblk, nblk, b;
  tile_of_workgroup(p, blockIdx.x, gridDim.x, mblk, nblk, b);
  const int t0 = mblk * BM;
  const int nt0 = nblk * (BN / 16) + wn * TN;
  // split-K (p.ksplit > 1): the "batch element" b is a K slice -- k-steps [kt0, kt0 + KTs) of the SAME rows, result into slab b
  const bool split = p.ksplit > 1;
  const int kt0 = split ? (int)(((int64_t)b * p.KT) / p.ksplit) : 0;
  const int KTs = split ? (int)(((int64_t)(b + 1) * p.KT) / p.ksplit) - kt0 : p.KT;
  const int cin_s = split ? min(KTs * KS, p.Cin - kt0 * KS) : p.Cin;     // columns of x this tile multiplies
  const int NC = (KTs + CK - 1) / CK;
  const int vrows = split ? p.Tin : conv_valid_rows(p, b);
  if (t0 + p.off0 >= vrows) return;   // ragged batch: the whole tile lies in this batch element's padding
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>((const T*)p.x + (split ? (int64_t)kt0 * KS : (int64_t)b * p.x_bstride)), 0,
      (int)((int64_t)vrows * p.Cin * (int)sizeof(T) - (split ? kt0 * KS * (int)sizeof(T) : 0)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.wp), 0, (int)((int64_t)p.NT * p.KT * 1024), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFFFu;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-thread staging slots: row / segment are chunk-independent
  unsigned st_goff[MAXST];   // byte offset of (row, seg) within chunk 0, or OOB
  int st_lds[MAXST];
  int st_col[MAXST];
#pragma unroll
  for (int q = 0; q < MAXST; ++q) {
    int idx = tid + q * NTH;
    int i = idx / SEGS, seg = idx - i * SEGS;
    int tin = t0 + p.off0 + i;
    bool ok = (i < BM) && (tin >= 0) && (tin < vrows);
    st_col[q] = seg * E;
    st_goff[q] = ok ? (unsigned)((tin * p.Cin + seg * E) * (int)sizeof(T)) : OOB;
    st_lds[q] = (i < BM) ? i * ROWB + seg * 16 : -1;
  }
  frag stg[MAXST];
  auto prefetch_a = [&](int c) {
#pragma unroll
    for (int q = 0; q < MAXST; ++q) {
      bool ok = (st_goff[q] != OOB) && (c * (CK * KS) + st_col[q] < cin_s);
      unsigned off = ok ? st_goff[q] + (unsigned)(c * (CK * KS) * (int)sizeof(T)) : OOB;
      stg[q] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
    }
  };
  auto commit_a = [&](unsigned char* buf) {
#pragma unroll
    for (int q = 0; q < MAXST; ++q)
      if (st_lds[q] >= 0) st16(buf + st_lds[q], stg[q]);
  };
  // weight block of (n-tile nt, flat k-step ks) = nt*KT + ks
  unsigned wbase[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) wbase[tn] = (nt0 + tn) < p.NT ? (unsigned)((((nt0 + tn) * p.KT + kt0) * 64 + lane) * 16) : OOB;
  auto fetch_b = [&](frag (&bf)[TN], int ks) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      unsigned off = (wbase[tn] != OOB && ks < KTs) ? wbase[tn] + (unsigned)ks * 1024u : OOB;
      bf[tn] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0));
    }
  };
  const int a_off = (wm * TM * 16 + r) * ROWB + g * 16;
  auto load_a = [&](frag (&af)[TM], const unsigned char* buf, int kk) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) af[tm] = ld16<frag>(buf + a_off + tm * 16 * ROWB + kk * 64);
  };
  auto mma_all = [&](frag (&af)[TM], frag (&bf)[TN]) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = EL::mma(bf[tn], af[tm], acc[tm][tn]);  // weights as A: transposed tile
  };

  frag bq0[TN], bq1[TN], bq2[TN], bq3[TN];
  frag a0[TM], a1[TM];
  // prologue: chunk 0 into LDS buffer 0, chunk 1 in flight, weight steps 0..2 in flight
  prefetch_a(0);
  fetch_b(bq0, 0);
  fetch_b(bq1, 1);
  fetch_b(bq2, 2);
  commit_a(lds);
  prefetch_a(1);
  __syncthreads();
  for (int c = 0; c < NC; ++c) {
    unsigned char* cur = lds + (c & 1) * BUFB;
    unsigned char* nxt = lds + ((c + 1) & 1) * BUFB;
    const int ks0 = c * CK;
    fetch_b(bq3, ks0 + 3);
    load_a(a0, cur, 0);
    commit_a(nxt);              // chunk c+1 (requested one chunk ago) -> the buffer everyone left at the last barrier
    prefetch_a(c + 2);
    __builtin_amdgcn_sched_barrier(0);  // keep the long-latency requests at the top of the chunk
    load_a(a1, cur, 1);
    mma_all(a0, bq0);
    fetch_b(bq0, ks0 + 4);
    load_a(a0, cur, 2);
    mma_all(a1, bq1);
    fetch_b(bq1, ks0 + 5);
    load_a(a1, cur, 3);
    mma_all(a0, bq2);
    fetch_b(bq2, ks0 + 6);
    mma_all(a1, bq3);
    __syncthreads();            // nxt complete for everyone; everyone done reading cur
  }
  conv_epilogue<T, TM, TN>(p, acc, b, t0 + wm * TM * 16, nt0, g, r);
}

template <typename T, int WM, int WN, int TM, int TN>
static int launch_plain(const ConvParams& p, hipStream_t s) {
  constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
  size_t ldsb = (size_t)2 * BM * (4 * 64 + 16);
  ConvParams q = p;
  q.MB = (p.Tout + BM - 1) / BM;
  q.NB = (p.N + BN - 1) / BN;
  const int64_t wbytes = (int64_t)BN * p.KT * 64;
  int gn = (int)((2 << 20) / (wbytes > 0 ? wbytes : 1));
  gn = gn < 1 ? 1 : (gn > 8 ? 8 : gn);
  int gm = 32 / gn;
  gm = gm >= 32 ? 32 : (gm >= 16 ? 16 : (gm >= 8 ? 8 : 4));
  q.GM = gm;
  if (p.ksplit > 1) {            // K slices in place of batch elements; slab ks of y = the rows' partial products over slice ks
    q.B = p.ksplit;
    q.y_bstride = (int64_t)p.Tout * p.N;
    q.y_limit = (int64_t)p.Tout * p.N;
    q.y_shift = 0;
  }
  const int64_t total = (int64_t)q.MB * q.NB * q.B;
  if (total > 0x7fffffff) {
    set_error("itts_gemm_conv: too many tiles (%lld)", (long long)total);
    return ITTS_ERR_INVALID;
  }
  static std::once_flag attr;
  std::call_once(attr, [] {
    (void)hipFuncSetAttribute((const void*)gemm_plain_kernel<T, WM, WN, TM, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  hipLaunchKernelGGL((gemm_plain_kernel<T, WM, WN, TM, TN>), dim3((unsigned)total), dim3(WM * WN * 64), ldsb, s, q);
  return check_launch("itts_gemm_conv");
}

// -------------------------------------------------------------------------------------------------------------------
// Narrow convolution (Cin <= 64, N <= 64: the last two BigVGAN stages, conv_post and the last upsampler).  These layers
// are HBM-bound (a k = 7 conv over 48 channels does 84 FLOP per byte moved), so the kernel is built around the row
// stream, not the MFMA pipe:
//   * the whole packed weight tensor of the layer (taps x KT x NT KiB, <= 66 KiB) is copied to LDS once per workgroup and
//     the workgroup then walks several 64-row-per-wave tiles (persistent grid), so weights cost nothing per row;
//   * activation fragments go straight from global memory to MFMA operand registers (a 16-row x 64-byte fragment is one
//     16-byte load per lane; the taps' overlapping rows are served by L1), no activation staging and no barrier in the
//     row loop; the next tap's fragments are requested before the current tap is multiplied.
// -------------------------------------------------------------------------------------------------------------------
template <typename T, int KT, int NT>
__global__ __launch_bounds__(256, (KT * NT <= 1 ? 4 : (KT * NT <= 4 ? 3 : 2))) void conv_narrow_kernel(ConvParams p) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  constexpr int E = EL::E, KS = EL::KS, TM = 4, BM = 4 * TM * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;
  constexpr unsigned OOB = 0xFFFFFFFFu;

#if ITTS_STAMPS
  __shared__ unsigned long long st_[16];   // [0] start [1] weights staged; tiles 0 / 1 of wave 0: [2,3,4] / [6,7,8] = start, taps done, epilogue issued; [5] end; [9] tiles
  if (tid < 16) st_[tid] = 0;
  int ntile_ = 0;
#endif
  CSTAMP(0);
  // weights -> LDS (same block order as in memory: ((tap*NT + nt)*KT + ks) KiB)
  const int wbytes = p.taps * NT * KT * 1024;
  for (int off = tid * 16; off < wbytes; off += 256 * 16) st16(lds + off, ld16<frag>((const unsigned char*)p.wp + off));
  __syncthreads();
  CSTAMP(1);

  const int ntiles = p.MB * p.B;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / p.MB, mblk = tile - b * p.MB;
    const int row0 = mblk * BM + wave * (TM * 16);
    const int vrows = conv_valid_rows(p, b);
    if (row0 + p.off0 >= vrows) continue;   // ragged batch: this wave's rows lie in the batch element's padding (no barrier in the loop)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>((const T*)p.x + (int64_t)b * p.x_bstride), 0, (int)((int64_t)vrows * p.Cin * (int)sizeof(T)), 0x00020000);
    f32x4 acc[TM][NT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_a = [&](frag (&af)[TM][KT], int j) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int tin = row0 + tm * 16 + r + p.off0 + j * p.dil;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
          const int col = ks * KS + g * E;
          const bool ok = (j < p.taps) && (tin >= 0) && (tin < vrows) && (col < p.Cin);
          const unsigned off = ok ? (unsigned)((tin * p.Cin + col) * (int)sizeof(T)) : OOB;
          af[tm][ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
      }
    };
    auto mma_tap = [&](frag (&af)[TM][KT], int j) {
      const unsigned char* wb = lds + (size_t)j * NT * KT * 1024 + lane * 16;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
          const frag bf = ld16<frag>(wb + (nt * KT + ks) * 1024);
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) acc[tm][nt] = EL::mma(bf, af[tm][ks], acc[tm][nt]);  // weights as A: transposed tile
        }
    };
    frag a0[TM][KT], a1[TM][KT];
#if ITTS_STAMPS
    if (ntile_ == 0) CSTAMP(2);
    if (ntile_ == 1) CSTAMP(6);
#endif
    load_a(a0, 0);
    for (int j = 0; j < p.taps; j += 2) {
      load_a(a1, j + 1);            // past the last tap: out-of-range offsets, no memory traffic
      mma_tap(a0, j);
      load_a(a0, j + 2);
      if (j + 1 < p.taps) mma_tap(a1, j + 1);
    }
#if ITTS_STAMPS
    asm volatile("" ::"v"(acc[0][0]));
    if (ntile_ == 0) CSTAMP(3);
    if (ntile_ == 1) CSTAMP(7);
#endif
    conv_epilogue<T, TM, NT>(p, acc, b, row0, 0, g, r);
#if ITTS_STAMPS
    if (ntile_ == 0) CSTAMP(4);
    if (ntile_ == 1) CSTAMP(8);
    ++ntile_;
#endif
  }
#if ITTS_STAMPS
  if (p.stamps != nullptr && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long te_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te_)::"memory");
    st_[5] = te_;
    st_[9] = (unsigned long long)ntile_;
    st_[14] = 1;
    st_[15] = 2;
  }
  __syncthreads();
  if (p.stamps != nullptr && tid < 16) p.stamps[(size_t)blockIdx.x * 16 + tid] = st_[tid];
#endif
}

template <typename T, int KT, int NT>
static int launch_narrow(const ConvParams& p, hipStream_t s) {
  constexpr int BM = 256;
  ConvParams q = p;
  q.MB = (p.Tout + BM - 1) / BM;
  q.NB = 1;
  q.GM = 1;
  const int64_t tiles = (int64_t)q.MB * p.B;
  const size_t ldsb = (size_t)p.taps * NT * KT * 1024;
  // persistent grid: as many workgroups as stay resident (LDS- and register-bound), each walking tiles with stride grid
  static std::once_flag attr;
  std::call_once(attr, [] {
    (void)hipFuncSetAttribute((const void*)conv_narrow_kernel<T, KT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024 - 256);   // minus the diagnostic build's static stamp block
    (void)hipGetLastError();
  });
  // persistent grid = the workgroups that are actually resident (registers and LDS both limit them; the first version
  // asked for up to 6 per CU by LDS alone and ran in rounds: tools/timeline_narrow.py)
  static thread_local size_t occ_lds = ~(size_t)0;
  static thread_local int occ_wgs = 1;
  if (occ_lds != ldsb) {
    int q_wgs = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q_wgs, (const void*)conv_narrow_kernel<T, KT, NT>, 256, ldsb) != hipSuccess ||
        q_wgs < 1)
      q_wgs = 1;
    (void)hipGetLastError();
    occ_lds = ldsb;
    occ_wgs = q_wgs;
  }
  int64_t grid = (int64_t)conv_num_cus() * occ_wgs;
  if (grid > tiles) grid = tiles;
  hipLaunchKernelGGL((conv_narrow_kernel<T, KT, NT>), dim3((unsigned)grid), dim3(256), ldsb, s, q);
  return check_launch("itts_gemm_conv");
}

// Second form of the narrow kernel (round 3) for the layers that dominate the last two vocoder stages (Cin = N = 24 / 48,
// taps 3 / 7 / 11): per-wave tiles of TM x 16 rows and EVERY tap's activation fragments of a tile requested before the first
// MFMA (in one or two groups).  The first form walks the taps with one tap of look-ahead: a 64-row tile pays 4-6 dependent
// L1 / L2 round trips (tools/timeline_narrow.py: 4.1-6.3 us in the taps against 0.4 us of MFMA time).  Here a tile pays one
// or two, and the smaller register footprint lets more waves share a SIMD.  Same arithmetic in the same order (taps outer,
// column tiles, k-steps inner): bit-identical to the first form.
// taps per request group: at most 16 activation fragments (64 VGPRs) in flight per lane, so that 3-4 waves share a SIMD
// (measured at C = 48, k = 7: one group of 7 taps at 2 waves per SIMD 195 us, at 3 waves per SIMD 174 us)
template <int TM, int KT, int TAPS>
struct NarrowGroup {
  static constexpr int CAP = 28 / (TM * KT) < 1 ? 1 : 28 / (TM * KT);
  static constexpr int G = TAPS < CAP ? TAPS : CAP;
};

template <typename T, int KT, int NT, int TAPS, int TM, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 8 ? 4 : 3))
void conv_narrow_taps_kernel(ConvParams p) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  constexpr int E = EL::E, KS = EL::KS, BM = NW * TM * 16, NTH = NW * 64;
  constexpr int G = NarrowGroup<TM, KT, TAPS>::G;   // taps per request group
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the tile bookkeeping stays in SGPRs
  const int g = lane >> 4, r = lane & 15;
  constexpr unsigned OOB = 0xFFFFFFFFu;
  const int wbytes = TAPS * NT * KT * 1024;
  for (int off = tid * 16; off < wbytes; off += NTH * 16) st16(lds + off, ld16<frag>((const unsigned char*)p.wp + off));
  __syncthreads();
  const int ntiles = p.MB * p.B;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / p.MB, mblk = tile - b * p.MB;
    const int row0 = mblk * BM + wave * (TM * 16);
    const int vrows = conv_valid_rows(p, b);
    if (row0 + p.off0 >= vrows) continue;   // ragged batch: this wave's rows lie in the batch element's padding (no barrier in the loop)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>((const T*)p.x + (int64_t)b * p.x_bstride), 0, (int)((int64_t)vrows * p.Cin * (int)sizeof(T)), 0x00020000);
    f32x4 acc[TM][NT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j0 = 0; j0 < TAPS; j0 += G) {
      frag af[G][TM][KT];
#pragma unroll
      for (int jj = 0; jj < G; ++jj)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const int tin = row0 + tm * 16 + r + p.off0 + (j0 + jj) * p.dil;
#pragma unroll
          for (int ks = 0; ks < KT; ++ks) {
            const int col = ks * KS + g * E;
            const bool ok = (j0 + jj < TAPS) && (tin >= 0) && (tin < vrows) && (col < p.Cin);
            const unsigned off = ok ? (unsigned)((tin * p.Cin + col) * (int)sizeof(T)) : OOB;
            af[jj][tm][ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
          }
        }
#pragma unroll
      for (int jj = 0; jj < G; ++jj) {
        if (j0 + jj < TAPS) {
          const unsigned char* wb = lds + (size_t)(j0 + jj) * NT * KT * 1024 + lane * 16;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
              const frag bf = ld16<frag>(wb + (nt * KT + ks) * 1024);
#pragma unroll
              for (int tm = 0; tm < TM; ++tm) acc[tm][nt] = EL::mma(bf, af[jj][tm][ks], acc[tm][nt]);  // weights as A: transposed tile
            }
        }
      }
    }
    conv_epilogue<T, TM, NT>(p, acc, b, row0, 0, g, r);
  }
}

template <typename T, int KT, int NT, int TAPS, int TM, int NW>
static int launch_narrow_taps(const ConvParams& p, hipStream_t s) {
  constexpr int BM = NW * TM * 16;
  ConvParams q = p;
  q.MB = (p.Tout + BM - 1) / BM;
  q.NB = 1;
  q.GM = 1;
  const int64_t tiles = (int64_t)q.MB * p.B;
  const size_t ldsb = (size_t)TAPS * NT * KT * 1024;
  static std::once_flag attr;
  static int per_cu = 1;
  std::call_once(attr, [] {
    constexpr size_t ldsb = (size_t)TAPS * NT * KT * 1024;
    (void)hipFuncSetAttribute((const void*)conv_narrow_taps_kernel<T, KT, NT, TAPS, TM, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024 - 256);
    int q_wgs = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q_wgs, (const void*)conv_narrow_taps_kernel<T, KT, NT, TAPS, TM, NW>, NW * 64, ldsb) !=
            hipSuccess || q_wgs < 1)
      q_wgs = 1;
    (void)hipGetLastError();
    per_cu = q_wgs;
  });
  int64_t grid = (int64_t)conv_num_cus() * per_cu;
  if (grid > tiles) grid = tiles;
  hipLaunchKernelGGL((conv_narrow_taps_kernel<T, KT, NT, TAPS, TM, NW>), dim3((unsigned)grid), dim3(NW * 64), ldsb, s, q);
  return check_launch("itts_gemm_conv");
}

// Third form (round 4): the activation rows of a tile are STAGED IN LDS.  The second form requests every tap's fragments from
// global memory -- a k = 11 layer reads each input row eleven times through the CU's L1 path and a tile still pays one or two
// dependent L2 round trips, 150-320 us per launch against an HBM time of ~100 us.  Here a workgroup's BM output rows plus the
// dilation halo (BM + (taps - 1) dil rows x Cin channels) go through registers into LDS ONCE; every tap's fragments are
// ds_read_b128 row reads (row stride = an odd number of 16-byte units: conflict-free), weights in LDS as before.  The NEXT tile's
// rows are requested right after the current tile has been committed to LDS and arrive under its MFMAs (persistent grid,
// register staging, two barriers per tile).  A fragment read may run up to (KT * KS - Cin) channels past a row's data -- into
// the row's pad or the next row's first bytes -- against ZERO weights (the packed weights are zero-padded in K), so the buffer
// only has to hold finite numbers there: it is cleared once.  Same arithmetic in the same order as the other two forms.
template <int CIN, int ES>
struct NarrowLds {
  static constexpr int DB = CIN * ES;                                      // data bytes of a row
  static constexpr int RSB = ((DB + 16) >> 4) & 1 ? DB + 16 : DB + 32;     // row stride: an odd number of 16-byte units
  static constexpr int CPR = DB / 16;                                      // 16-byte chunks per row
};

template <typename T, int CIN, int TAPS, int TM, int NW, int MAXH = CV_MAX_HALO>   // MAXH: largest (taps - 1) * dil the row tile is sized for
__global__ __launch_bounds__(NW * 64, 2)
void conv_narrow_lds_kernel(ConvParams p) {
  typedef Elem<T> EL;
  typedef typename EL::frag frag;
  constexpr int KS = EL::KS, ES = (int)sizeof(T), BM = NW * TM * 16, NTH = NW * 64;
  constexpr int KT = (CIN + KS - 1) / KS, NT = (CIN + 15) / 16;
  constexpr int WB = TAPS * NT * KT * 1024;
  constexpr int MAXROWS = BM + MAXH + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  constexpr unsigned OOB = 0xFFFFFFFFu;
  constexpr int DB = NarrowLds<CIN, ES>::DB, rsb = NarrowLds<CIN, ES>::RSB, cpr = NarrowLds<CIN, ES>::CPR;
  const int halo = (TAPS - 1) * p.dil;
  const int rows_in = BM + halo;
  unsigned char* tile = lds + WB;
  for (int off = tid * 16; off < WB; off += NTH * 16) st16(lds + off, ld16<frag>((const unsigned char*)p.wp + off));
  for (int off = tid * 16; off < MAXROWS * rsb; off += NTH * 16) st16(tile + off, zero_frag<frag>());
  __syncthreads();
  constexpr int NLD = (MAXROWS * cpr + NTH - 1) / NTH;   // staging requests per thread (upper bound: the largest halo)
  const int ntiles = p.MB * p.B;
  frag st[NLD];
  auto request = [&](int tile_id) {
    const int b = tile_id / p.MB, mblk = tile_id - b * p.MB;
    const int vrows = conv_valid_rows(p, b);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>((const T*)p.x + (int64_t)b * p.x_bstride), 0, (int)((int64_t)vrows * p.Cin * ES), 0x00020000);
    const int t_in0 = mblk * BM + p.off0;
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int idx = tid + q * NTH;
      const int row = idx / cpr, ch = idx - row * cpr;
      const int tin = t_in0 + row;
      const bool ok = row < rows_in && tin >= 0 && tin < vrows;
      st[q] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)(tin * DB + ch * 16) : OOB, 0, 0));
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const int idx = tid + q * NTH;
      const int row = idx / cpr, ch = idx - row * cpr;
      if (row < rows_in) st16(tile + row * rsb + ch * 16, st[q]);
    }
  };
  // Epilogue operands (residual rows, the running sum of an accumulating layer) are requested a WHOLE TILE AHEAD, with the next
  // tile's activation rows: a lane's epilogue positions (row r of each row tile, channels 16 nt + 4g .. +3) are the same in every
  // tile, so they wait in registers and the epilogue issues no load.  v = fma(acc + bias + resid, scale, sum) as conv_epilogue
  // (y is T-typed, N / y_shift / y_limit multiples of 4, no per-batch bias, no activation: checked by the dispatcher).
  typedef T t4 __attribute__((ext_vector_type(4)));
  const bool has_r = p.resid != nullptr, has_a = p.accumulate != 0;
  f32x4 bs[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias ? p.bias : (const float*)p.wp), 0, p.bias ? p.N * 4 : 0, 0x00020000);
    bs[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb1, (unsigned)((nt * 16 + g * 4) * 4), 0, 0));
  }
  u32x2 rres[TM][NT], racc[TM][NT];
  auto epi_request = [&](int tid_) {
    const int b = tid_ / p.MB, mblk = tid_ - b * p.MB;
    const int row0 = mblk * BM + wave * (TM * 16);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((T*)p.y + (int64_t)b * p.y_bstride, 0, (int)(p.y_limit * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>((const T*)(p.resid ? p.resid : p.y)) + (int64_t)b * p.y_bstride, 0, (int)(p.y_limit * ES), 0x00020000);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int t = row0 + tm * 16 + r, col0 = nt * 16 + g * 4;
        const int64_t flat = (int64_t)t * p.N + col0 + p.y_shift;
        const bool ok = (t < p.Tout) && (col0 < p.N) && (flat >= 0) && (flat < p.y_limit);
        const unsigned off = ok ? (unsigned)(flat * ES) : OOB;
        rres[tm][nt] = has_r ? __builtin_amdgcn_raw_buffer_load_b64(rr, off, 0, 0) : u32x2{0u, 0u};
        racc[tm][nt] = has_a ? __builtin_amdgcn_raw_buffer_load_b64(ry, off, 0, 0) : u32x2{0u, 0u};
      }
  };
  int tile_id = blockIdx.x;
  if (tile_id < ntiles) {
    request(tile_id);
    epi_request(tile_id);
  }
  for (; tile_id < ntiles; tile_id += gridDim.x) {
    commit();
    __syncthreads();
    if (tile_id + (int)gridDim.x < ntiles) request(tile_id + gridDim.x);    // in flight under this tile's MFMAs
    const int b = tile_id / p.MB, mblk = tile_id - b * p.MB;
    const int row0 = mblk * BM + wave * (TM * 16);
    const int vrows = conv_valid_rows(p, b);
    if (row0 + p.off0 < vrows) {     // (ragged batch: a wave whose rows lie in the element's padding computes and stores nothing)
      f32x4 acc[TM][NT];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned char* ar = tile + (wave * (TM * 16) + r) * rsb + g * 16;
#pragma unroll
      for (int j = 0; j < TAPS; ++j) {
        frag af[TM][KT];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int ks = 0; ks < KT; ++ks) af[tm][ks] = ld16<frag>(ar + (tm * 16 + j * p.dil) * rsb + ks * (KS * ES));
        const unsigned char* wb = lds + (size_t)j * NT * KT * 1024 + lane * 16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int ks = 0; ks < KT; ++ks) {
            const frag bf = ld16<frag>(wb + (nt * KT + ks) * 1024);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) acc[tm][nt] = EL::mma(bf, af[tm][ks], acc[tm][nt]);  // weights as A: transposed tile
          }
      }
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((T*)p.y + (int64_t)b * p.y_bstride, 0, (int)(p.y_limit * ES), 0x00020000);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int t = row0 + tm * 16 + r, col0 = nt * 16 + g * 4;
          const int64_t flat = (int64_t)t * p.N + col0 + p.y_shift;
          const bool ok = (t < p.Tout) && (col0 < p.N) && (flat >= 0) && (flat < p.y_limit);
          const t4 rv = __builtin_bit_cast(t4, rres[tm][nt]), av = __builtin_bit_cast(t4, racc[tm][nt]);
          t4 o;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const float v = acc[tm][nt][jj] + bs[nt][jj];
            o[jj] = (T)fmaf(v + (has_r ? (float)rv[jj] : 0.f), p.scale, has_a ? (float)av[jj] : 0.f);   // as conv_epilogue_impl
          }
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ry, ok ? (unsigned)(flat * ES) : OOB, 0, 0);
        }
    }
    // the NEXT tile's epilogue operands: requested now (their registers are free again), consumed at the end of the next tile
    if (tile_id + (int)gridDim.x < ntiles) epi_request(tile_id + gridDim.x);
    __syncthreads();     // every wave has read its rows: the next tile may be committed
  }
}

template <typename T, int CIN, int TAPS, int TM, int NW, int MAXH = CV_MAX_HALO>
static int launch_narrow_lds(const ConvParams& p, hipStream_t s) {
  constexpr int BM = NW * TM * 16, ES = (int)sizeof(T), KS = Elem<T>::KS;
  constexpr int KT = (CIN + KS - 1) / KS, NT = (CIN + 15) / 16;
  ConvParams q = p;
  q.MB = (p.Tout + BM - 1) / BM;
  q.NB = 1;
  q.GM = 1;
  const int64_t tiles = (int64_t)q.MB * p.B;
  constexpr size_t ldsb = (size_t)TAPS * NT * KT * 1024 + (size_t)(BM + MAXH + 1) * NarrowLds<CIN, ES>::RSB;
  static_assert(ldsb <= 160 * 1024 - 256, "conv_narrow_lds: weights + row tile exceed the LDS");
  static std::once_flag attr;
  std::call_once(attr, [] {
    (void)hipFuncSetAttribute((const void*)conv_narrow_lds_kernel<T, CIN, TAPS, TM, NW, MAXH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024 - 256);
    (void)hipGetLastError();
  });
  static thread_local size_t occ_lds = ~(size_t)0;
  static thread_local int occ_wgs = 1;
  if (occ_lds != ldsb) {
    int q_wgs = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q_wgs, (const void*)conv_narrow_lds_kernel<T, CIN, TAPS, TM, NW, MAXH>, NW * 64, ldsb) !=
            hipSuccess || q_wgs < 1)
      q_wgs = 1;
    (void)hipGetLastError();
    occ_lds = ldsb;
    occ_wgs = q_wgs;
  }
  int64_t grid = (int64_t)conv_num_cus() * occ_wgs;
  if (grid > tiles) grid = tiles;
  hipLaunchKernelGGL((conv_narrow_lds_kernel<T, CIN, TAPS, TM, NW, MAXH>), dim3((unsigned)grid), dim3(NW * 64), ldsb, s, q);
  return check_launch("itts_gemm_conv");
}

#if ITTS_DIAG
int g_conv_cfg = 0;  // diagnostic build: itts_debug_set(3, id) kernel override for A/B measurements (0 = default)
#else
constexpr int g_conv_cfg = 0;  // product build: the heuristic below, no mutable state (the override branches fold away)
#endif

// (k-steps, column tiles) pairs built for the narrow kernel; everything else takes the tiled kernel
template <typename T>
static int dispatch_narrow(const ConvParams& p, hipStream_t s, bool& handled) {
  handled = true;
  const int kt = p.KT, nt = p.NT;
  if constexpr (sizeof(T) == 2) if (g_conv_cfg != 30 && g_conv_cfg != 31 && p.Cin == p.N && (p.Cin == 24 || p.Cin == 48) && !p.y_f32 &&
                                    p.bias2 == nullptr && p.act == 0 && ((p.N | p.y_shift | p.y_limit) & 3) == 0) {
    // third form (rows staged in LDS); diagnostic build: cfg 31 = the second form instead, for A/B runs
#define ITTS_NL_CASE(CIN_, NW3_, NW7_, NW11_)                                                                       \
    if (p.Cin == CIN_ && p.taps == 3) return launch_narrow_lds<T, CIN_, 3, 16 / NW3_, NW3_>(p, s);                  \
    if (p.Cin == CIN_ && p.taps == 7) return launch_narrow_lds<T, CIN_, 7, 16 / NW7_, NW7_>(p, s);                  \
    if (p.Cin == CIN_ && p.taps == 11) return launch_narrow_lds<T, CIN_, 11, 16 / NW11_, NW11_>(p, s);
    ITTS_NL_CASE(24, 4, 4, 4)       // C = 24: 256-row tiles, 4 waves x 4 row tiles; weights + rows <= 48 KB -> 3 workgroups per CU
    ITTS_NL_CASE(48, 4, 4, 8)       // C = 48: the same; 11 taps: 66 KB of weights -> one 8-wave workgroup per CU (2 row tiles per wave)
#undef ITTS_NL_CASE
  }
#if ITTS_NARROW_C96
  // C = 96, 3 taps only (54 KB of weights; 7 taps would be 126 KB): bandwidth-shaped like the C = 48 layers
  if constexpr (sizeof(T) == 2) if (g_conv_cfg != 30 && g_conv_cfg != 31 && p.Cin == 96 && p.N == 96 && p.taps == 3 && !p.y_f32 && p.bias2 == nullptr &&
                                    p.act == 0 && ((p.N | p.y_shift | p.y_limit) & 3) == 0)
    return launch_narrow_lds<T, 96, 3, 1, 8>(p, s);
  // ... and 7 taps with the row tile sized for the vocoder's dilations (halo <= 30 rows): 126 KB of weights + 159 rows = 158 KB
  if constexpr (sizeof(T) == 2) if (g_conv_cfg != 30 && g_conv_cfg != 31 && p.Cin == 96 && p.N == 96 && p.taps == 7 && 6 * p.dil <= 30 && !p.y_f32 &&
                                    p.bias2 == nullptr && p.act == 0 && ((p.N | p.y_shift | p.y_limit) & 3) == 0)
    return launch_narrow_lds<T, 96, 7, 1, 8, 30>(p, s);
#endif
  if (g_conv_cfg != 30) {   // (diagnostic build: cfg 30 = the first form everywhere, for A/B runs)
#define ITTS_NT_CASE(KT_, NT_, TM_, NW_)                                                               \
    if (kt == KT_ && nt == NT_ && p.taps == 3) return launch_narrow_taps<T, KT_, NT_, 3, TM_, NW_>(p, s);   \
    if (kt == KT_ && nt == NT_ && p.taps == 7) return launch_narrow_taps<T, KT_, NT_, 7, TM_, NW_>(p, s);   \
    if (kt == KT_ && nt == NT_ && p.taps == 11) return launch_narrow_taps<T, KT_, NT_, 11, TM_, NW_>(p, s);
    // measured on MI355X (fp16, batch 32, k = 7; first form 211 / 184 us at C = 48 / 24): C = 48: 2 row tiles per wave, 4 waves
    // 177-181 us (8 waves 197, 1 row tile x 8 waves 176 with spills); C = 24: 4 row tiles per wave 144-148 us (2 tiles 154-162)
    ITTS_NT_CASE(1, 2, 4, 4)
    ITTS_NT_CASE(2, 3, 2, 4)
#undef ITTS_NT_CASE
  }
  if (kt == 1 && nt == 1) return launch_narrow<T, 1, 1>(p, s);
  if (kt == 1 && nt == 2) return launch_narrow<T, 1, 2>(p, s);
  if (kt == 2 && nt == 1) return launch_narrow<T, 2, 1>(p, s);
  if (kt == 2 && nt == 2) return launch_narrow<T, 2, 2>(p, s);
  if (kt == 2 && nt == 3) return launch_narrow<T, 2, 3>(p, s);
  if (kt == 3 && nt == 3) return launch_narrow<T, 3, 3>(p, s);
  handled = false;
  return ITTS_OK;
}

template <typename T>
static int dispatch_conv(const ConvParams& p, hipStream_t s) {
  const bool plain = (p.taps == 1);  // GEMM: no halo, 4 k-steps per chunk
  if (((p.Cin <= 64 && p.N <= 64) || (ITTS_NARROW_C96 && p.Cin == 96 && p.N == 96 && (p.taps == 3 || p.taps == 7))) && g_conv_cfg != 2 &&
      (size_t)p.taps * p.NT * p.KT * 1024 <= 150 * 1024) {   // (C = 96, 7 taps: 126 KB)
    bool handled;
    int rc = dispatch_narrow<T>(p, s, handled);
    if (handled) return rc;
  }
  if (plain && p.N % 128 == 0) {
    // Measured on MI355X (bf16, M = 3008 / 7488, N = 1280..5120, K = 1280 / 5120): 128 x 128 pipelined 435-720 TFLOP/s,
    // 256 x 128 pipelined 300-625, the unpipelined 256 x 128 tile of the convolution kernel 260-540.
    if (g_conv_cfg == 5) return launch_plain<T, 2, 4, 8, 2>(p, s);   // 256 x 128
    // (a 256 x 128 tile with both operands through LDS and four waves of 128 x 64 was built in round 3, measured slower on every
    // shape -- 300-475 against 580-670 TFLOP/s, profiles/r03_big_gemm.txt -- and removed in round 4)
    // Tile shape by rounds of the chip (all tiles of a launch take the same time, so a launch costs its number of ROUNDS over the
    // 512 workgroup slots times the tile's work): 128 x 160 tiles (4 x 2 waves of 32 x 80, 1.25x the work) where they save a round --
    // the prefill's FC at M ~ 2 000 (640 tiles of 128 x 128 = two rounds, 512 of 128 x 160 = one), the latent pass's QKV at M ~ 4 500
    // (1 080 = three rounds against 864 = two).
    if (p.ksplit <= 1 && p.N % 160 == 0 && p.B == 1) {
      const int64_t mb = (p.Tout + 127) / 128, slots = 2 * (int64_t)conv_num_cus();
      const int64_t r128 = (mb * (p.N / 128) + slots - 1) / slots, r160 = (mb * (p.N / 160) + slots - 1) / slots;
      if (5 * r160 < 4 * r128) return launch_plain<T, 4, 2, 2, 5>(p, s);
    }
    return launch_plain<T, 2, 4, 4, 2>(p, s);                        // 128 x 128, two workgroups per CU
  }
  if (plain && p.N % 64 == 0) {
    const int64_t rows = (int64_t)p.B * ((p.Tout + 255) / 256);
    return (rows * (p.N / 64) >= 448) ? launch_conv<T, 4, 2, 4, 2, 4, 0>(p, s) : launch_conv<T, 4, 2, 2, 2, 4, 0>(p, s);
  }
#if ITTS_DIAG
  if (g_conv_cfg == 10) {   // the 8-wave tiles of the first version (256-row tiles, one workgroup per CU), for A/B runs
    if (p.N % 128 == 0) return launch_conv<T, 2, 4, 8, 2, 2, CV_MAX_HALO>(p, s);
    if (p.N % 64 == 0) return launch_conv<T, 4, 2, 4, 2, 2, CV_MAX_HALO>(p, s);
    if (p.N % 96 == 0) return launch_conv<T, 4, 2, 4, 3, 2, CV_MAX_HALO>(p, s);
  }
  if (g_conv_cfg == 15) {   // 6 waves side by side for 192 columns
    if (p.N % 192 == 0) return launch_conv<T, 1, 6, 8, 2, 2, CV_MAX_HALO, true>(p, s);
  }
#endif
  // Tile choice (MI355X, fp16, batch 32; profiles/r02_conv_tiles.txt).  Two 4-wave workgroups per CU beat one 8-wave
  // workgroup everywhere: each workgroup stalls once per channel chunk on the in-order memory queue (its weight loads sit
  // behind the next chunk's activation rows), and the other workgroup's MFMAs fill that hole.
  //   N % 128: four waves SIDE BY SIDE (128 x 32 each, 128 x 128 per workgroup): no weight fragment is loaded by two waves
  //            of a workgroup, the per-CU L2->L1 weight traffic is that of the old 256 x 128 tile.  C = 768 k = 11: 244 us
  //            (953 TFLOP/s) against 388 us before; k = 3 d = 5: 100 against 134; C = 384: 146-153 against 185.
  //   N % 64 / N % 96 (C = 192, 96: HBM-heavy, 1.1 / 2.3 FLOP per byte per tap): 2 x 2 waves of 64 x 32 / 64 x 48,
  //            persistent grid (the epilogue's stores overlap the next tile): 192 / 233 us against 268 / 273.
  if (p.N % 128 == 0) return launch_conv<T, 1, 4, 8, 2, 2, CV_MAX_HALO>(p, s);
  if (p.N % 64 == 0) return launch_conv<T, 2, 2, 4, 2, 2, CV_MAX_HALO, true>(p, s);
  if (p.N % 96 == 0) return launch_conv<T, 2, 2, 4, 3, 2, CV_MAX_HALO, true>(p, s);
  if (p.N % 48 == 0) return launch_conv<T, 4, 1, 4, 3, 2, CV_MAX_HALO>(p, s);
  return launch_conv<T, 4, 1, 4, 2, 2, CV_MAX_HALO>(p, s);
}

}  // namespace itts

using namespace itts;

static int conv_params_from_args(const itts_conv_args* a, ConvParams& p, const char* who) {
  if (!(a && a->x && a->wp && a->y)) { set_error("%s: null pointer", who); return ITTS_ERR_INVALID; }
  if (!(a->B >= 0 && a->Tin >= 0 && a->Tout >= 0 && a->Cin > 0 && a->N > 0 && a->taps > 0 && a->dil >= 0)) {
    set_error("%s: bad shape", who);
    return ITTS_ERR_INVALID;
  }
  if (a->Cin % 8 != 0) { set_error("%s: Cin=%d must be a multiple of 8", who, a->Cin); return ITTS_ERR_INVALID; }
  if ((a->taps - 1) * a->dil > CV_MAX_HALO) {
    set_error("%s: (taps-1)*dil = %d exceeds %d", who, (a->taps - 1) * a->dil, CV_MAX_HALO);
    return ITTS_ERR_INVALID;
  }
  p.B = a->B;
  p.Tin = a->Tin;
  p.valid_rows = a->valid_rows;
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
#if ITTS_DIAG
  p.exp = g_conv_exp;
#else
  p.exp = 0;
#endif
#if ITTS_STAMPS
  p.stamps = g_stamp_buf_conv;
#endif
  p.NT = (a->N + 15) / 16;
  p.KT = (a->Cin + ks - 1) / ks;
  p.MB = p.NB = p.GM = 1;
  p.ksplit = a->ksplit > 1 ? a->ksplit : 1;
  if (p.ksplit > 1 && !(a->taps == 1 && a->N % 128 == 0 && a->B == 1 && a->y_f32 && a->bias == nullptr && a->bias2 == nullptr &&
                        a->resid == nullptr && !a->accumulate && a->act == 0 && a->valid_rows == nullptr && a->off0 == 0 &&
                        a->Tin == a->Tout && a->Cin % ks == 0 && p.ksplit <= p.KT && p.ksplit <= 64)) {
    set_error("%s: ksplit > 1 is for the plain GEMM (taps 1, N %% 128 == 0, B 1, fp32 y = slabs, no bias / residual / activation)", who);
    return ITTS_ERR_INVALID;
  }
  return ITTS_OK;
}

extern "C" int itts_gemm_conv(const itts_conv_args* a, void* stream) {
  ConvParams p;
  int rc = conv_params_from_args(a, p, "itts_gemm_conv");
  if (rc != ITTS_OK) return rc;
  if (a->B == 0 || a->Tout == 0) return ITTS_OK;
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
