// Probe (MI355X): (1) which XCD do the workgroups of consecutive launches land on?  (2) does a concurrent, throttled
// "weight pump" on a second stream -- one wave per CU that touches the NEXT launch's weight bytes from the XCD whose
// workgroups will read them -- shorten a chain of dependent weight-streaming launches?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/build/xcc_pump_probe tools/probes/xcc_pump_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xF;
}

// ---------------------------------------------------------------- (1) placement
__global__ void where_kernel(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

// ---------------------------------------------------------------- (2) chain + pump
constexpr int TILE_BYTES = 48 * 1024;   // per workgroup and launch: 8 waves x 6 x 1 KiB
constexpr int NTILES = 256;
constexpr int SPIN_LIMIT = 1 << 20;

struct ChainArgs {
  const char* w;          // this launch's weight set (NTILES * TILE_BYTES)
  float* out;             // [NTILES] (dependency carrier)
  const float* in;        // previous launch's out
  unsigned* progress;     // word the pump watches
  unsigned* ticket;       // [8] per-XCD tickets of THIS launch (32-byte apart?) -- 8 words, 128 B apart
  unsigned* stat;         // [0] overflow count
  unsigned step;
  int mode;               // 0: tile = blockIdx; 1: tile = (xcc, ticket)
};

__global__ __launch_bounds__(512) void chain_kernel(ChainArgs a) {
  __shared__ float red[8];
  __shared__ int s_tile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    if (blockIdx.x == 0) {
      if (a.step == 0xFFFFFFFFu) (void)__hip_atomic_fetch_add(a.progress, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(a.progress, a.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int tile = blockIdx.x;
    if (a.mode == 1) {
      const unsigned x = xcc_id() & 7;
      unsigned slot = __hip_atomic_fetch_add(a.ticket + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (slot < 32) tile = (int)(x + 8 * slot);
      else {
        atomicAdd(a.stat, 1u);
        tile = -1;
        for (unsigned d = 1; d < 8 && tile < 0; ++d) {   // another XCD's queue has room (placement was not even)
          const unsigned y = (x + d) & 7;
          slot = __hip_atomic_fetch_add(a.ticket + y * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (slot < 32) tile = (int)(y + 8 * slot);
        }
      }
    }
    s_tile = tile;
  }
  __syncthreads();
  const int tile = s_tile;
  if (tile < 0) return;
  const char* p = a.w + (size_t)tile * TILE_BYTES + (size_t)wave * 6 * 1024 + lane * 16;
  float4 v[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) v[i] = *reinterpret_cast<const float4*>(p + i * 1024);
  float dep = a.in[(tile * 7 + 3) & (NTILES - 1)];
  float s = dep;
#pragma unroll
  for (int i = 0; i < 6; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < 8; ++w) t += red[w];
    a.out[tile] = t * 1e-9f;
  }
}

struct PumpArgs {
  const char* w;          // all weight sets, consecutive
  size_t set_bytes;
  int nsets;
  int nsteps;
  int ahead;              // touch step j's bytes once progress >= j - ahead
  unsigned* progress;
  unsigned* pticket;      // [8] per-XCD rank tickets of the pump itself (128 B apart)
  unsigned* stat;         // [1] timeouts
  int mode;               // 1: my tiles = the ones the chain's workgroups on MY XCD take (xcc + 8*slot); 0: tile = blockIdx
  int first_step;
  unsigned base;          // progress value at which step 0 of this pump starts (eager pump: replay * nsteps)
  int shift;              // tile = (blockIdx + shift) % NTILES: shift 4 = deliberately the WRONG XCD (Infinity Cache only)
};

// one wave per workgroup, one workgroup per CU; touches one dword per 64 B
__global__ __launch_bounds__(64) void pump_kernel(PumpArgs a) {
  const int lane = threadIdx.x;
  int tile = (blockIdx.x + a.shift) % NTILES;
  if (a.mode == 1) {
    unsigned x = xcc_id() & 7, slot = 0;
    if (lane == 0) slot = __hip_atomic_fetch_add(a.pticket + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    slot = __shfl(slot, 0);
    tile = (int)(x + 8 * (slot & 31));
  }
  for (int j = a.first_step; j < a.nsteps; ++j) {
    const unsigned want = a.base + (j > a.ahead ? (unsigned)(j - a.ahead) : 0u);
    int spins = 0;
    while ((int)(__hip_atomic_load(a.progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > SPIN_LIMIT) { if (lane == 0) atomicAdd(a.stat + 1, 1u); return; }
    }
    const char* p = a.w + (size_t)(j % a.nsets) * a.set_bytes + (size_t)tile * TILE_BYTES + lane * 64;
    unsigned sink;
#pragma unroll
    for (int i = 0; i < TILE_BYTES / 4096; ++i)
      asm volatile("global_load_dword %0, %1, off" : "=v"(sink) : "v"(p + i * 4096) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

static double run_chain(int mode_chain, int pump, int ahead, int nsets, int nsteps, char* w, size_t set_bytes, float* bufA, float* bufB,
                        unsigned* ctl, hipStream_t sA, hipStream_t sB, int replays) {
  // ctl layout (words): [0] progress, [64..] stat(2), [128 .. 128+8*32) pump tickets, [512 + step*256 ...] chain tickets
  hipGraph_t g; hipGraphExec_t ge;
  hipEvent_t fork, join;
  CK(hipEventCreate(&fork)); CK(hipEventCreate(&join));
  CK(hipStreamBeginCapture(sA, hipStreamCaptureModeGlobal));
  CK(hipMemsetAsync(ctl, 0, (512 + (size_t)nsteps * 256) * 4, sA));
  if (pump) {
    CK(hipEventRecord(fork, sA));
    CK(hipStreamWaitEvent(sB, fork, 0));
    PumpArgs pa{w, set_bytes, nsets, nsteps, ahead, ctl, ctl + 128, ctl + 64, pump == 2 ? 1 : 0, 0, 0u, 0};
    hipLaunchKernelGGL(pump_kernel, dim3(NTILES), dim3(64), 0, sB, pa);
    CK(hipEventRecord(join, sB));
  }
  for (int k = 0; k < nsteps; ++k) {
    ChainArgs ca{w + (size_t)(k % nsets) * set_bytes, (k & 1) ? bufB : bufA, (k & 1) ? bufA : bufB, ctl, ctl + 512 + (size_t)k * 256,
                 ctl + 64, (unsigned)(k + 1), mode_chain};
    hipLaunchKernelGGL(chain_kernel, dim3(NTILES), dim3(512), 0, sA, ca);
  }
  if (pump) CK(hipStreamWaitEvent(sA, join, 0));
  CK(hipStreamEndCapture(sA, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, sA)); CK(hipStreamSynchronize(sA));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, sA));
  for (int r = 0; r < replays; ++r) CK(hipGraphLaunch(ge, sA));
  CK(hipEventRecord(e1, sA));
  CK(hipStreamSynchronize(sA));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned st[2]; CK(hipMemcpy(st, ctl + 64, 8, hipMemcpyDeviceToHost));
  if (st[0] || st[1]) printf("    [overflow tickets %u, pump timeouts %u]\n", st[0], st[1]);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 1e3 * ms / (replays * (double)nsteps);
}

// the pump as an EAGER launch on a second stream before every replay of the chain graph (graph branches did not overlap)
static double run_chain_eager_pump(int ahead, int shift, int nsets, int nsteps, char* w, size_t set_bytes, float* bufA, float* bufB,
                                   unsigned* ctl, hipStream_t sA, hipStream_t sB, int replays, bool with_pump) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipMemset(ctl, 0, (512 + (size_t)nsteps * 256) * 4));
  CK(hipStreamBeginCapture(sA, hipStreamCaptureModeGlobal));
  for (int k = 0; k < nsteps; ++k) {
    ChainArgs ca{w + (size_t)(k % nsets) * set_bytes, (k & 1) ? bufB : bufA, (k & 1) ? bufA : bufB, ctl, ctl + 512, ctl + 64, 0xFFFFFFFFu, 0};
    hipLaunchKernelGGL(chain_kernel, dim3(NTILES), dim3(512), 0, sA, ca);
  }
  CK(hipStreamEndCapture(sA, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  unsigned rep = 0;
  auto one = [&]() {
    if (with_pump) {
      PumpArgs pa{w, set_bytes, nsets, nsteps, ahead, ctl, ctl + 128, ctl + 64, 0, 0, rep * (unsigned)nsteps, shift};
      hipLaunchKernelGGL(pump_kernel, dim3(NTILES), dim3(64), 0, sB, pa);
    }
    CK(hipGraphLaunch(ge, sA));
    ++rep;
  };
  one(); CK(hipStreamSynchronize(sA)); CK(hipStreamSynchronize(sB));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, sA));
  for (int r = 0; r < replays; ++r) one();
  CK(hipEventRecord(e1, sA));
  CK(hipStreamSynchronize(sA)); CK(hipStreamSynchronize(sB));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned st[2]; CK(hipMemcpy(st, ctl + 64, 8, hipMemcpyDeviceToHost));
  if (st[0] || st[1]) printf("    [overflow tickets %u, pump timeouts %u]\n", st[0], st[1]);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 1e3 * ms / (replays * (double)nsteps);
}

int main(int argc, char** argv) {
  hipStream_t sA, sB; CK(hipStreamCreate(&sA)); CK(hipStreamCreate(&sB));
  // ---- (1) placement of consecutive launches inside a graph
  {
    const int grids[] = {240, 256, 160, 640, 80, 240, 32, 240, 97, 256, 256, 240};
    const int NL = sizeof(grids) / sizeof(int);
    unsigned* d; CK(hipMalloc(&d, NL * 1024 * 4));
    for (int threads : {512, 256, 64}) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(sA, hipStreamCaptureModeGlobal));
      for (int l = 0; l < NL; ++l) hipLaunchKernelGGL(where_kernel, dim3(grids[l]), dim3(threads), 0, sA, d + l * 1024);
      CK(hipStreamEndCapture(sA, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipGraphLaunch(ge, sA)); CK(hipStreamSynchronize(sA));
        std::vector<unsigned> h(NL * 1024); CK(hipMemcpy(h.data(), d, NL * 1024 * 4, hipMemcpyDeviceToHost));
        printf("placement threads=%d replay %d:", threads, rep);
        for (int l = 0; l < NL; ++l) {
          int off0 = (int)((h[l * 1024] + 8 - 0) % 8), bad = 0, cnt[8] = {0};
          for (int b = 0; b < grids[l]; ++b) { if ((int)((h[l * 1024 + b] + 800 - b) % 8) != off0) ++bad; cnt[h[l * 1024 + b] & 7]++; }
          printf(" g%d:off%d/bad%d", grids[l], off0, bad);
          (void)cnt;
        }
        printf("\n");
      }
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    CK(hipFree(d));
  }
  // ---- (2) chain with and without the pump
  const int nsteps = 96;
  const size_t set_bytes = (size_t)NTILES * TILE_BYTES;
  const int maxsets = 24;
  char* w; CK(hipMalloc(&w, set_bytes * maxsets)); CK(hipMemset(w, 1, set_bytes * maxsets));
  float *bA, *bB; CK(hipMalloc(&bA, NTILES * 4)); CK(hipMalloc(&bB, NTILES * 4));
  CK(hipMemset(bA, 0, NTILES * 4)); CK(hipMemset(bB, 0, NTILES * 4));
  unsigned* ctl; CK(hipMalloc(&ctl, (512 + (size_t)nsteps * 256) * 4));
  printf("chain of %d launches, %d workgroups x 512 threads, %.1f MB per launch; us per launch\n", nsteps, NTILES, set_bytes / 1e6);
  for (int nsets : {1, 2, 4, 24}) {
    for (int mode : {0, 1}) {
      double t = run_chain(mode, 0, 0, nsets, nsteps, w, set_bytes, bA, bB, ctl, sA, sB, 20);
      printf("  sets=%2d chain-map=%s no pump: %6.2f us\n", nsets, mode ? "xcc-ticket" : "blockIdx  ", t);
    }
  }
  {
    double t = run_chain_eager_pump(0, 0, 24, nsteps, w, set_bytes, bA, bB, ctl, sA, sB, 20, false);
    printf("  sets=24 eager form, no pump:              %6.2f us\n", t);
  }
  for (int shift : {0, 4})
    for (int ahead : {0, 1, 2}) {
      double t = run_chain_eager_pump(ahead, shift, 24, nsteps, w, set_bytes, bA, bB, ctl, sA, sB, 20, true);
      printf("  sets=24 eager pump %s ahead=%d: %6.2f us\n", shift ? "WRONG xcd (Infinity Cache only)" : "same xcd (L2)                  ", ahead, t);
    }
  printf("PROBE DONE\n");
  return 0;
}
