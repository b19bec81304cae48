// Dev lab: the split-bf16 full-mesh row-block kernel on the bench shape (T frames, V = 6890), timing + per-wave phase stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iem_pose_amd/csrc -Iinclude scripts/dev/mesh_bf16_lab.hip -o /tmp/mesh_bf16_lab && /tmp/mesh_bf16_lab
#define EMPOSE_MESH_TRACE 1
#include "../../em_pose_amd/csrc/mesh.hip"

#include <algorithm>
#include <cstdio>
#include <vector>
using namespace empose;
#include "lab_stubs.h"
#ifdef LAB_PIPE   // scripts/dev/experiments/mesh_bf16_pipe.hip pasted into mesh.hip
#define LAB_KERNEL mesh_rows_bf16_pipe_kernel
#else
#define LAB_KERNEL mesh_rows_bf16_kernel<false>
#endif

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u ^ seed;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  p[i] = ((x & 0xffffff) / 8388608.f - 1.f) * scale;
}
static float* dev_rand(size_t n, unsigned seed, float scale) {
  float* p; (void)hipMalloc(&p, n * 4);
  fill_kernel<<<(n + 255) / 256, 256>>>(p, n, seed, scale);
  return p;
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 16384, V = 6890, NT = (V + 31) / 32;
  MeshSkinArgs a{};
  a.T = T; a.V = V; a.kb = 4;
  a.feat = dev_rand((size_t)T * 200, 1, 1.f);
  a.xf = dev_rand((size_t)T * 264, 2, 1.f);
  a.trans = dev_rand((size_t)T * 3, 3, 1.f);
  a.wc_frag = dev_rand((size_t)NT * 25 * 3 * 256, 4, 0.05f);
  a.wc_bf16 = dev_rand((size_t)NT * mb::TILE_BYTES / 4, 6, 0.05f);   // bit patterns of small floats: finite bf16 pairs
  a.skin_w4 = dev_rand((size_t)NT * 32 * 4, 5, 0.25f);
  std::vector<int> idx((size_t)NT * 32 * 4);
  for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)((i / 4 / 300 + (i & 3)) % 22);   // neighbouring vertices share bones
  int* d_idx; (void)hipMalloc(&d_idx, idx.size() * 4);
  (void)hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
  a.skin_idx4 = d_idx; a.skin_idx = d_idx; a.skin_w = a.skin_w4;
  float* out; (void)hipMalloc(&out, (size_t)T * V * 12);
  a.vertices = out;
  const int bx = (T + 63) / 64;
  auto launch = [&]() {
    hipLaunchKernelGGL(LAB_KERNEL, dim3(bx, 1), dim3(mb::NW * 64), mb::LDS_BYTES, 0, a);
  };
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(LAB_KERNEL),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)mb::LDS_BYTES);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms / 5 < best ? ms / 5 : best;
  }
  const double flops = 2.0 * T * (double)NT * 32 * 3 * 200;
  printf("T=%d: %.1f us/launch  %.1f TFLOP/s  %.2f M frames/s (%s)\n", T, best * 1e3,
         flops / best * 1e-9, T / best * 1e-3, hipGetErrorString(hipGetLastError()));
  static long long tr[8 * 32 * 4];
  (void)hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_mesh_trace), sizeof(tr));
  long long last = 0;
  for (int w = 0; w < 8; ++w) last = std::max(last, tr[(w * 32 + 26) * 4 + 2] - tr[0]);
  printf("  block (0,0): %lld cycles to the last wave's end; at %.1f us per launch that is >= %.2f GHz\n", last, best * 1e3,
         last / (best * 1e6));
  for (int w = 0; w < 8; w += (argc > 2 ? 1 : 4)) {
    printf("  wave %d: start %lld |", w, tr[(w * 32) * 4] - tr[0]);
    for (int t = 0; t < 27; t += 1)
      printf(" %lld+%lld", (tr[(w * 32 + t) * 4 + 1] - tr[(w * 32 + t) * 4]) / 100, (tr[(w * 32 + t) * 4 + 2] - tr[(w * 32 + t) * 4 + 1]) / 100);
    printf(" | end %lld\n", tr[(w * 32 + 26) * 4 + 2] - tr[0]);
  }
  return 0;
}
