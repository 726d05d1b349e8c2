// L2 run-ahead of the decode step (include/indextts_hip.h, itts_prefetch): a launch of the token loop touches the packed
// weight bytes a LATER skinny-GEMM launch will stream, from the XCD whose workgroups will read them, so that the HBM fetch
// of launch k+1 overlaps launch k's dependent work instead of heading launch k+1's critical path.
//
// Geometry.  The consumer (plan_skinny) runs gx x gy workgroups; workgroup lin' = ks * gx + bx streams, for each of its ntb
// column tiles t, the contiguous bytes [(t * KT + ks * SB) KiB, + min(SB, KT - ks * SB) KiB) of the packed weight.
// Workgroups are dealt round-robin over the 8 XCDs, so consumer workgroups with lin' % 8 == r share an XCD (and its L2)
// with the workgroups lin % 8 == r of THIS launch: those touch exactly that residue's bytes, in pieces of 4 KiB (one wave
// instruction, one dword per 64 bytes), dealt evenly over their waves.  The touched values are never used: a different
// placement makes the consumer's reads L2 misses again, nothing else.
#pragma once
#include "common.h"

namespace itts {

struct PfParams {
  const char* base;   // nullptr: off
  unsigned bytes;     // size of the packed weight (buffer range check)
  int gx, nblk;       // consumer grid.x, workgroups
  int ntb, KT, SB, NT;
  int pp;             // 4-KiB pieces per (tile, K slice)
  int part, parts;
  float inv_pp, inv_per_blk;
};

constexpr int PF_SLOTS = 4;   // pieces a wave can touch (more are simply left cold)
struct PfRegs {
  uint32_t v[PF_SLOTS];
};

// wave `wave` of `nwaves` of workgroup `lin` of `nblk` (all wave-uniform) requests its pieces; returns at once
__device__ __forceinline__ void pf_issue(const PfParams& pf, unsigned lin, unsigned nblk, int wave, int nwaves, int lane, PfRegs& o) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pf.base), 0, pf.base ? (int)pf.bytes : 0, 0x00020000);
  const int r = lin & 7, rank = lin >> 3;
  const int cnt = ((int)nblk - r + 7) >> 3;          // workgroups of this launch on "my" XCD
  const int nres = (pf.nblk - r + 7) >> 3;           // consumer workgroups there
  const int per_blk = pf.ntb * pf.pp;
  const int total = nres > 0 ? nres * per_blk : 0;
  const int W = cnt * nwaves;
  const int me = __builtin_amdgcn_readfirstlane(rank * nwaves + wave);
#pragma unroll
  for (int i = 0; i < PF_SLOTS; ++i) {
    int j = me + i * W;
    if (pf.parts > 1) j = j * pf.parts + pf.part;
    const int m = (int)(((float)j + 0.5f) * pf.inv_per_blk);
    const int q = j - m * per_blk;
    const int tl = (int)(((float)q + 0.5f) * pf.inv_pp);
    const int p = q - tl * pf.pp;
    const int lb = r + 8 * m;                          // consumer workgroup
    int ks = 0;   // gy <= 8; sums of compares, no branch (a join would drain vmcnt)
#pragma unroll
    for (int i = 1; i < 8; ++i) ks += (lb >= i * pf.gx);
    const int bx = lb - ks * pf.gx;
    const int t = bx * pf.ntb + tl;
    const int kbeg = ks * pf.SB;
    const int klen = min(pf.SB, pf.KT - kbeg);
    const int within = p * 4096 + lane * 64;
    const bool ok = j < total && t < pf.NT && within < klen * 1024;
    const unsigned off = ok ? (unsigned)((t * pf.KT + kbeg) * 1024 + within) : 0xFFFFFFFFu;   // out of range: no access
    o.v[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
  }
}

// keeps the requests alive up to this point (the loads have no other consumer); put it at the very end of the kernel
__device__ __forceinline__ void pf_keep(const PfRegs& o) {
#pragma unroll
  for (int i = 0; i < PF_SLOTS; ++i) asm volatile("" ::"v"(o.v[i]));
}

}  // namespace itts

// host: consumer geometry -> kernel parameters (itts_skinny_plan is this library's own entry point)
static inline int itts_make_prefetch(const itts_prefetch& a, itts::PfParams* o) {
  o->base = nullptr;
  o->bytes = 0;
  o->gx = o->nblk = o->ntb = o->KT = o->SB = o->NT = o->pp = 1;
  o->part = 0;
  o->parts = 1;
  o->inv_pp = o->inv_per_blk = 1.f;
  if (a.wp == nullptr) return ITTS_OK;
  const int ksplit = a.ksplit > 0 ? a.ksplit : 1;
  int q[6];
  const int rc = itts_skinny_plan(a.dtype, a.M, a.N, a.K, ksplit, q);
  if (rc != ITTS_OK) return rc;
  const int kstep = a.dtype == ITTS_F32 ? 16 : 32;
  ITTS_REQUIRE(a.K % kstep == 0 && ksplit <= 8, "itts_prefetch: bad consumer shape N=%d K=%d ksplit=%d", a.N, a.K, ksplit);
  const int KT = a.K / kstep, NT = (a.N + 15) / 16;
  const int64_t bytes = (int64_t)NT * KT * 1024;
  ITTS_REQUIRE(bytes < ((int64_t)1 << 31), "itts_prefetch: packed weight too large");
  o->base = (const char*)a.wp;
  o->bytes = (unsigned)bytes;
  o->gx = q[0];
  o->nblk = q[0] * q[1];
  o->ntb = q[3];
  o->KT = KT;
  o->SB = (KT + ksplit - 1) / ksplit;
  o->NT = NT;
  o->pp = (o->SB * 1024 + 4095) / 4096;
  o->parts = a.parts > 1 ? a.parts : 1;
  o->part = a.parts > 1 ? a.part % a.parts : 0;
  o->inv_pp = 1.0f / (float)o->pp;
  o->inv_per_blk = 1.0f / (float)(o->ntb * o->pp);
  return ITTS_OK;
}
