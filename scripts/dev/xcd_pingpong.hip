// Dev lab: round-trip latency of a tagged 8-byte word between two workgroups, on the same XCD and on different XCDs,
// with agent-scope relaxed atomics (what lstm_persist_kernel uses) and with plain L1-bypassing accesses.
// Workgroups are dispatched to XCDs round-robin by id (wg % 8); each participant reports the XCC_ID it really ran on.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_pingpong scripts/dev/xcd_pingpong.hip && /tmp/xcd_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

template <int MODE>
__device__ __forceinline__ unsigned long long ld(const unsigned long long* p) {
  if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long v;
  if (MODE == 1) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (MODE == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (MODE == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int MODE>
__device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) {
  if (MODE == 0) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
  if (MODE == 1) asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
  if (MODE == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
  if (MODE == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

// block `a` and block `b` ping-pong `n` times; everybody else exits.  word[0]: a -> b, word[32]: b -> a (own lines).
template <int MODE>
__global__ void pingpong(unsigned long long* word, int a, int b, int n, long long* out) {
  if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
  if (threadIdx.x != 0) return;
  const bool first = (int)blockIdx.x == a;
  out[first ? 2 : 3] = xcc_id();
  unsigned long long* mine = word + (first ? 0 : 32);
  const unsigned long long* theirs = word + (first ? 32 : 0);
  const long long t0 = wall_clock64();
  long long spins = 0;
  for (int i = 1; i <= n; ++i) {
    if (first) st<MODE>(mine, (unsigned long long)i);
    while (ld<MODE>(theirs) < (unsigned long long)i) { if (++spins > (1ll << 26)) { out[4] = -1; return; } }
    if (!first) st<MODE>(mine, (unsigned long long)i);
  }
  if (first) { out[0] = wall_clock64() - t0; out[1] = spins; }
}

template <int MODE>
static void run(const char* name, int a, int b) {
  unsigned long long* word; long long* out;
  hipMalloc(&word, 64 * 8); hipMalloc(&out, 8 * 8);
  hipMemset(word, 0, 64 * 8); hipMemset(out, 0, 64);
  const int n = 2000;
  hipLaunchKernelGGL(pingpong<MODE>, dim3(256), dim3(64), 0, 0, word, a, b, n, out);
  hipError_t e = hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
  // wall_clock64 ticks at 100 MHz
  printf("%-34s blocks %3d/%3d on XCC %lld/%lld: %.2f us per round trip (%lld spins)%s %s\n", name, a, b, h[2], h[3],
         (double)h[0] / n * 0.01, h[1], h[4] < 0 ? "  TIMED OUT" : "", e == hipSuccess ? "" : hipGetErrorString(e));
  hipFree(word); hipFree(out);
}

int main() {
  const int pairs[3][2] = {{0, 8}, {0, 1}, {0, 4}};
  for (auto& p : pairs) {
    run<0>("agent-scope relaxed atomics", p[0], p[1]);
    run<1>("plain accesses, sc1", p[0], p[1]);
    run<2>("plain accesses, sc0", p[0], p[1]);
    run<3>("plain accesses, sc0 sc1", p[0], p[1]);
  }
  return 0;
}
