// Stand-alone reproducer of the accumulator corruption seen with v_mfma_f32_32x32x16_bf16 when TWO waves of a SIMD run
// the split-bf16 mesh kernel ("One wave per SIMD, on purpose").  No Python, no library: the kernel
// source is included as it is and launched with random operands; every launch is compared bit for bit with the first.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iem_pose_amd/csrc -DMB_NW=8 scripts/dev/bf16_hazard_repro.hip -o /tmp/repro8
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iem_pose_amd/csrc -DMB_NW=4 scripts/dev/bf16_hazard_repro.hip -o /tmp/repro4
//   /tmp/repro8 [frames] [launches]      (scripts/dev/bf16_hazard_repro.sh runs both and a few variants)
//
// MB_NW = waves per workgroup: 8 = two per SIMD (the failing configuration), 4 = one per SIMD (what the tree ships).
// Variants: -DREPRO_FP32_MFMA is not available here (the fp32 kernel is another function: mesh_rows_kernel, 8 waves,
// clean); -DMB_RING=n changes the prefetch depth of the coefficient ring (did not matter).
#include "../../em_pose_amd/csrc/mesh.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace empose;
namespace empose { Options& options() { static Options o; return o; } }

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
// differences of a launch against the reference launch: count, and the first few (element index, got bits, want bits)
__global__ void compare_kernel(const unsigned* got, const unsigned* want, size_t n, unsigned long long* count, unsigned* first) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n || got[i] == want[i]) return;
  const unsigned long long k = atomicAdd(count, 1ull);
  if (k < 8) { first[3 * k] = (unsigned)i; first[3 * k + 1] = got[i]; first[3 * k + 2] = want[i]; }
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 16384, R = argc > 2 ? atoi(argv[2]) : 20, V = 6890, NT = (V + 31) / 32;
  MeshSkinArgs a{};
  a.T = T; a.V = V; a.kb = 4;
  a.feat = dev_rand((size_t)T * 200, 1, 1.f);
  a.xf = dev_rand((size_t)T * 264, 2, 1.f);
  a.trans = dev_rand((size_t)T * 3, 3, 1.f);
  a.wc_frag = dev_rand((size_t)NT * 25 * 3 * 256, 4, 0.05f);
  a.wc_bf16 = dev_rand((size_t)NT * mb::TILE_BYTES / 4, 6, 0.05f);   // bit patterns of small floats: finite bf16 pairs
  a.skin_w4 = dev_rand((size_t)NT * 32 * 4, 5, 0.25f);
  std::vector<int> idx((size_t)NT * 32 * 4);
  for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)((i / 4 / 300 + (i & 3)) % 22);
  int* d_idx; (void)hipMalloc(&d_idx, idx.size() * 4);
  (void)hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
  a.skin_idx4 = d_idx; a.skin_idx = d_idx; a.skin_w = a.skin_w4;
  const size_t n = (size_t)T * V * 3;
  float *ref, *out;
  (void)hipMalloc(&ref, n * 4); (void)hipMalloc(&out, n * 4);
  unsigned long long* d_count; unsigned* d_first;
  (void)hipMalloc(&d_count, 8); (void)hipMalloc(&d_first, 96);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mesh_rows_bf16_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)mb::LDS_BYTES);
  auto launch = [&](float* dst) {
    a.vertices = dst;
    hipLaunchKernelGGL(mesh_rows_bf16_kernel<false>, dim3((T + 63) / 64, 1), dim3(mb::NW * 64), mb::LDS_BYTES, 0, a);
  };
  launch(ref);
  unsigned long long total = 0; int bad_launches = 0;
  for (int r = 0; r < R; ++r) {
    launch(out);
    (void)hipMemset(d_count, 0, 8);
    compare_kernel<<<(n + 255) / 256, 256>>>((const unsigned*)out, (const unsigned*)ref, n, d_count, d_first);
    unsigned long long c; unsigned f[24];
    (void)hipMemcpy(&c, d_count, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(f, d_first, 96, hipMemcpyDeviceToHost);
    if (c) {
      ++bad_launches; total += c;
      printf("  launch %2d: %llu elements differ from launch 0; first:", r + 1, c);
      for (unsigned long long k = 0; k < (c < 4 ? c : 4); ++k) {
        const unsigned e = f[3 * k]; const unsigned frame = e / (V * 3), vert = (e / 3) % V, coord = e % 3;
        printf(" (frame %u [%u in its 64-block], vertex %u [%u in its 32-tile], %c: %08x vs %08x)", frame, frame % 64, vert,
               vert % 32, "xyz"[coord], f[3 * k + 1], f[3 * k + 2]);
      }
      printf("\n");
    }
  }
  printf("MB_NW=%d (%s wave%s per SIMD), T=%d, %d launches: %d launches differ from the first, %llu elements in all (%s)\n", mb::NW,
         mb::NW == 8 ? "two" : "one", mb::NW == 8 ? "s" : "", T, R, bad_launches, total, hipGetErrorString(hipGetLastError()));
  return 0;
}
