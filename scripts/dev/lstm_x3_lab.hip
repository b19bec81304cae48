// Dev lab: one wavefront step of the bench LSTM (2 layers, 296 -> 512 -> 512, B rows) on lstm_chain_x3_kernel, timed per
// launch; built with -DLX_LAB_NOLOAD / -DLX_LAB_NOPART / -DLX_LAB_NOFINISH it times the kernel without that part.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iem_pose_amd/csrc scripts/dev/lstm_x3_lab.hip -o /tmp/lstm_x3_lab
#include "../../em_pose_amd/csrc/kernels.h"
#include "../../em_pose_amd/csrc/lstm_x3.hip"
#include "lab_stubs.h"

#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
using namespace empose;

template <typename T>
static T* dev_rand(size_t n, std::mt19937& rng, float scale) {
  std::vector<T> h(n);
  std::normal_distribution<float> nd(0.f, scale);
  for (auto& v : h) {
    const float x = nd(rng);
    if (sizeof(T) == 2) { unsigned u; memcpy(&u, &x, 4); v = (T)(u >> 16); } else { memcpy(&v, &x, 4); }
  }
  T* p; (void)hipMalloc(&p, n * sizeof(T)); (void)hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice);
  return p;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 1024, H = 512, IN = 296, F = 32;
  std::mt19937 rng(3);
  const int KS_in = (IN + 15) / 16, KS_h = H / 16, RT = (B + 31) / 32;
  LstmX3Args a;
  a.n_units = 2; a.units_per_block = 2; a.seq_lengths = nullptr; a.B = B; a.F = F; a.H = H;
  float* y = dev_rand<float>((size_t)B * F * H, rng, 0.1f);
  for (int l = 0; l < 2; ++l) {
    LstmX3Unit& u = a.unit[l];
    const int ks_in = l == 0 ? KS_in : KS_h;
    u.w3_ih = dev_rand<unsigned short>((size_t)ks_in * (H / 32) * 4 * 3 * 512, rng, 0.02f);
    u.w3_hh = dev_rand<unsigned short>((size_t)KS_h * (H / 32) * 4 * 3 * 512, rng, 0.02f);
    u.bias = dev_rand<float>(4 * H, rng, 0.1f);
    u.a3_in = dev_rand<unsigned short>((size_t)RT * ks_in * 3 * 512, rng, 0.5f);
    u.ks_in = ks_in;
    u.a3_rec = dev_rand<unsigned short>((size_t)RT * KS_h * 3 * 512, rng, 0.5f);
    u.a3_out = dev_rand<unsigned short>((size_t)RT * KS_h * 3 * 512, rng, 0.5f);
    u.h_prev = dev_rand<float>((size_t)B * H, rng, 0.5f);
    u.h_next = dev_rand<float>((size_t)B * H, rng, 0.5f);
    u.c = dev_rand<float>((size_t)B * H, rng, 0.5f);
    u.y = l == 1 ? y : nullptr; u.y_ld = H; u.y_col = 0; u.t = 3 - l;
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto time_it = [&](const char* name, const LstmX3Args& x) {
    for (int i = 0; i < 5; ++i) (void)launch_lstm_chain_x3(x, 0);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      (void)hipEventRecord(e0);
      for (int i = 0; i < 33; ++i) (void)launch_lstm_chain_x3(x, 0);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms / 33 < best ? ms / 33 : best;
    }
    printf("%-44s B=%d: %.1f us/launch (%s)\n", name, B, best * 1e3, hipGetErrorString(hipGetLastError()));
  };
#ifdef LX_LAB_TIMES
  long long* times; (void)hipMalloc(&times, 512 * 16 * 8); (void)hipMemset(times, 0, 512 * 16 * 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(lx_lab_times), &times, sizeof(times));
#endif
  time_it("both units, one workgroup walks both", a);
#ifdef LX_LAB_TIMES
  {
    std::vector<long long> t(512 * 16);
    (void)hipMemcpy(t.data(), times, t.size() * 8, hipMemcpyDeviceToHost);
    const char* names[11] = {"", "u0 products", "u0 sums in LDS", "u0 barrier", "u0 cell math", "u0 stored", "u1 products",
                             "u1 sums in LDS", "u1 barrier", "u1 cell math", "u1 stored"};
    double mean[11] = {0}; int cnt = 0;
    for (int b = 0; b < 256; ++b) {
      if (!t[b * 16 + 10]) continue;
      ++cnt;
      for (int i = 1; i <= 10; ++i) mean[i] += (double)(t[b * 16 + i] - t[b * 16]);
    }
    printf("  mean over %d workgroups, shader clocks since the workgroup's start:\n   ", cnt);
    for (int i = 1; i <= 10; ++i) printf(" %s %.0f |", names[i], mean[i] / cnt);
    printf("\n");
  }
#endif
  LstmX3Args b = a; b.units_per_block = 1;
  time_it("both units, a workgroup each", b);
  LstmX3Args c = a; c.n_units = 1; c.units_per_block = 1;
  time_it("layer 0 only (K = 296 + 512)", c);
  LstmX3Args d = c; d.unit[0] = a.unit[1];
  time_it("layer 1 only (K = 512 + 512)", d);
  return 0;
}
