// Dev lab: one forward layer of both update networks at the reference's training batch (384 rows, 512 -> 512) on
// cols_kernel (train_cols.hip): time per launch back to back, and -- built with -DTC_LAB_TIMES -- where a workgroup's
// time goes (shader-clock stamps: start, loads issued, products done, after the barrier, statistics ready, exchange done, end).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTC_LAB_TIMES -Iem_pose_amd/csrc scripts/dev/train_cols_lab.hip -o /tmp/train_cols_lab
#include "../../em_pose_amd/csrc/kernels.h"
#include "../../em_pose_amd/csrc/train_cols.hip"
#include "lab_stubs.h"

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
using namespace empose;

static float* dev_rand(size_t n, std::mt19937& rng, float scale, float offset = 0.f) {
  std::vector<float> h(n);
  std::normal_distribution<float> nd(0.f, scale);
  for (auto& v : h) v = offset + nd(rng);
  float* p; (void)hipMalloc(&p, n * 4); (void)hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice);
  return p;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 384, H = 512;
  std::mt19937 rng(5);
  ColsArgs a{};
  a.n_nets = 2; a.M = M; a.eps = 1e-5f; a.momentum = 0.1f; a.tag = 1;
  (void)hipMalloc(&a.mailbox, cols_mailbox_words(H) * 8);
  (void)hipMemset(a.mailbox, 0, cols_mailbox_words(H) * 8);
  for (int i = 0; i < 2; ++i) {
    ColsNet& c = a.net[i];
    c.A = dev_rand((size_t)M * H, rng, 1.f); c.lda = H;
    c.W = dev_rand((size_t)H * H, rng, 0.05f); c.ldw = H; c.bias = dev_rand(H, rng, 0.1f);
    c.N = H; c.K = H;
    c.gamma = dev_rand(H, rng, 0.1f, 1.f); c.beta = dev_rand(H, rng, 0.1f); c.slope = dev_rand(1, rng, 0.f, 0.25f);
    c.running_mean = dev_rand(H, rng, 0.1f); c.running_var = dev_rand(H, rng, 0.f, 1.f); c.num_batches = nullptr;
    c.z = dev_rand((size_t)M * H, rng, 1.f); c.ldz = H; c.out = dev_rand((size_t)M * H, rng, 1.f); c.ld_out = H;
    c.mean = dev_rand(H, rng, 1.f); c.rstd = dev_rand(H, rng, 1.f);
  }
#ifdef TC_LAB_TIMES
  long long* times; (void)hipMalloc(&times, 256 * 8 * 8); (void)hipMemset(times, 0, 256 * 8 * 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(tc_lab_times), &times, sizeof(times));
#endif
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  unsigned tag = 1;
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < 5; ++i) { a.tag = tag++; (void)launch_cols(a, mode, 0); }
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      (void)hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) { a.tag = tag++; (void)launch_cols(a, mode, 0); }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      best = std::min(best, ms / 20);
    }
    printf("%s M=%d: %.2f us/launch back to back (%s)\n", mode == 0 ? "forward with BatchNorm" : "forward, product only", M,
           best * 1e3, hipGetErrorString(hipGetLastError()));
#ifdef TC_LAB_TIMES
    std::vector<long long> t(256 * 8);
    (void)hipMemcpy(t.data(), times, t.size() * 8, hipMemcpyDeviceToHost);
    // every XCD has its own counter base: differences inside a workgroup only; mean over the workgroups that ran
    const char* names[7] = {"start", "loads issued", "products done", "after barrier", "statistics", "exchange done", "end"};
    const int last_i = mode == 0 ? 6 : 3;
    double mean[7] = {0, 0, 0, 0, 0, 0, 0}; int cnt = 0;
    for (int b = 0; b < 256; ++b) {
      if (!t[b * 8 + last_i]) continue;
      ++cnt;
      for (int i = 1; i <= last_i; ++i) mean[i] += (double)(t[b * 8 + i] - t[b * 8]);
    }
    printf("  mean over %d workgroups, shader clocks since the workgroup's start:", cnt);
    for (int i = 1; i <= last_i; ++i) printf("  %s %.0f", names[i], mean[i] / cnt);
    printf("\n");
    (void)hipMemset(times, 0, 256 * 8 * 8);
#endif
  }
  return 0;
}
