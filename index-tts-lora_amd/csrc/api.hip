// Error plumbing + weight packing for libindextts_hip.so
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace itts {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return ITTS_ERR_LAUNCH;
  }
  return ITTS_OK;
}

// one thread per 16-byte chunk of the packed image
template <typename T>
__global__ void pack_weight_kernel(const T* __restrict__ w, T* __restrict__ out, int taps, int K, int N, int NT, int KT) {
  constexpr int E = Elem<T>::E, KS = Elem<T>::KS;
  int64_t chunk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)taps * NT * KT * 64;
  if (chunk >= total) return;
  int lane = (int)(chunk & 63);
  int64_t blk = chunk >> 6;
  int ks = (int)(blk % KT);
  int nt = (int)((blk / KT) % NT);
  int tap = (int)(blk / ((int64_t)KT * NT));
  int g = lane >> 4, c = lane & 15;
  int n = nt * 16 + c;
  T vals[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    int k = ks * KS + g * E + e;
    vals[e] = (k < K && n < N) ? w[((int64_t)tap * K + k) * N + n] : Elem<T>::from_f(0.f);
  }
  T* dst = out + chunk * E;
#pragma unroll
  for (int e = 0; e < E; ++e) dst[e] = vals[e];
}
}  // namespace itts

using namespace itts;

extern "C" int itts_abi_version(void) { return ITTS_ABI_VERSION; }
extern "C" const char* itts_last_error(void) { return g_err; }

extern "C" int64_t itts_packed_bytes(int taps, int K, int N, int dtype) {
  int ks = dtype == ITTS_F32 ? 16 : 32;
  int64_t NT = (N + 15) / 16, KT = (K + ks - 1) / ks;
  return (int64_t)taps * NT * KT * 1024;
}

extern "C" int itts_pack_weight(const void* w, void* packed, int taps, int K, int N, int dtype, void* stream) {
  ITTS_REQUIRE(w && packed && taps > 0 && K > 0 && N > 0, "itts_pack_weight: bad arguments");
  int ks = dtype == ITTS_F32 ? 16 : 32;
  int NT = (N + 15) / 16, KT = (K + ks - 1) / ks;
  int64_t total = (int64_t)taps * NT * KT * 64;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case ITTS_F32:
      hipLaunchKernelGGL(pack_weight_kernel<float>, grid, block, 0, s, (const float*)w, (float*)packed, taps, K, N, NT, KT);
      break;
    case ITTS_BF16:
      hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)w, (bf16_t*)packed, taps, K, N, NT, KT);
      break;
    case ITTS_F16:
      hipLaunchKernelGGL(pack_weight_kernel<f16_t>, grid, block, 0, s, (const f16_t*)w, (f16_t*)packed, taps, K, N, NT, KT);
      break;
    default:
      ITTS_REQUIRE(false, "itts_pack_weight: unknown dtype %d", dtype);
  }
  return check_launch("itts_pack_weight");
}
