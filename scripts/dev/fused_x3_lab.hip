// Dev lab: the fused MLP on the bench shape (two nets, 296-512x4-66/10, T rows) -- the fp32-MFMA kernel (mlp_fused.hip)
// next to the three-piece bf16 kernel (mlp_fused_x3.hip): time per launch, and the error of both against a float64
// evaluation of the same nets on the host (first / last rows of a few row blocks).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iem_pose_amd/csrc scripts/dev/fused_x3_lab.hip -o /tmp/fused_x3_lab
#include "../../em_pose_amd/csrc/mlp_fused.hip"
#include "../../em_pose_amd/csrc/mlp_fused_x3.hip"
#include "lab_stubs.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
using namespace empose;

static unsigned short bf16_rn(float x) {
  unsigned u; memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

static float* upload_f(const std::vector<float>& v) {
  float* p; (void)hipMalloc(&p, v.size() * 4); (void)hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice); return p;
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 32768, LDX = 296, Hd = 512;
  const float xs = argc > 2 ? atof(argv[2]) : 1.f;
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> hx((size_t)T * LDX);
  for (auto& v : hx) v = nd(rng) * xs;
  float* x = upload_f(hx);
  FusedMlpArgs a32, ax3;
  a32.count = ax3.count = 2; a32.M = ax3.M = T;
  struct HostLayer { std::vector<float> W, sc, sh; int K, N; float slope; int act; };
  std::vector<HostLayer> host[2];
  double flops = 0;
  for (int n = 0; n < 2; ++n) {
    const int nout = n == 0 ? 66 : 10;
    for (FusedMlpArgs* a : {&a32, &ax3}) {
      FusedNet& fn = a->net[n];
      fn.x = x; fn.ldx = LDX; fn.ld_out = nout; fn.n_layers = 6;
      (void)hipMalloc(&fn.out, (size_t)T * nout * 4);
    }
    for (int l = 0; l < 6; ++l) {
      HostLayer h;
      h.K = l == 0 ? LDX : Hd; h.N = l == 5 ? nout : Hd; h.slope = 0.25f; h.act = l < 5;
      h.W.resize((size_t)h.N * h.K); h.sc.resize(h.N); h.sh.resize(h.N);
      const float ws = 1.4f / std::sqrt((float)h.K);
      for (auto& v : h.W) v = nd(rng) * ws;
      for (auto& v : h.sc) v = 0.75f + 0.5f * (float)(rng() & 0xffff) / 65536.f;
      for (auto& v : h.sh) v = 0.1f * nd(rng);
      flops += 2.0 * T * h.N * h.K;
      // fp32 fragment order (api.hip pack_fragments_raw)
      const int KG = (h.K + 7) / 8, NT = (h.N + 31) / 32, KG4 = (KG + 3) & ~3;
      std::vector<float> p32((size_t)KG4 * NT * 256, 0.f);
      for (int kg = 0; kg < KG; ++kg) for (int nt = 0; nt < NT; ++nt) for (int lane = 0; lane < 64; ++lane) {
        const int nn = nt * 32 + (lane & 31);
        if (nn >= h.N) continue;
        for (int e = 0; e < 4; ++e) { const int k = kg * 8 + (lane >> 5) * 4 + e; if (k < h.K) p32[(((size_t)kg * NT + nt) * 64 + lane) * 4 + e] = h.W[(size_t)nn * h.K + k]; }
      }
      // three bf16 pieces in fragment order: [ks][tile][piece][lane][8]
      const int KS = (h.K + 15) / 16, KS4 = (KS + 3) & ~3;   // zero-padded to whole quads of k-steps
      std::vector<unsigned short> px((size_t)KS4 * NT * 3 * 512, 0);
      for (int ks = 0; ks < KS; ++ks) for (int nt = 0; nt < NT; ++nt) for (int lane = 0; lane < 64; ++lane) {
        const int nn = nt * 32 + (lane & 31);
        if (nn >= h.N) continue;
        for (int e = 0; e < 8; ++e) {
          const int k = ks * 16 + (lane >> 5) * 8 + e;
          if (k >= h.K) continue;
          const float w = h.W[(size_t)nn * h.K + k];
          const unsigned short p0 = bf16_rn(w); const float r = w - bf16_f(p0);
          const unsigned short p1 = bf16_rn(r); const float s = r - bf16_f(p1);
          const unsigned short p2 = bf16_rn(s);
          const size_t base = (((size_t)ks * NT + nt) * 3) * 512 + (size_t)lane * 8 + e;
          px[base] = p0; px[base + 512] = p1; px[base + 1024] = p2;
        }
      }
      void* dpx; (void)hipMalloc(&dpx, px.size() * 2); (void)hipMemcpy(dpx, px.data(), px.size() * 2, hipMemcpyHostToDevice);
      float* dsc = upload_f(h.sc); float* dsh = upload_f(h.sh); float* d32 = upload_f(p32);
      for (int which = 0; which < 2; ++which) {
        FusedLayer& L = (which ? ax3 : a32).net[n].layer[l];
        L.K = h.K; L.N = h.N; L.scale = dsc; L.shift = dsh; L.slope = h.slope; L.act = h.act;
        L.W = which ? (const float*)dpx : d32;
      }
      host[n].push_back(std::move(h));
    }
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto time_it = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) (void)launch();
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
      (void)hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) (void)launch();
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms / 10 < best ? ms / 10 : best;
    }
    printf("%-30s T=%d: %.1f us/launch  %.1f TFLOP/s fp32-equivalent  (%s)\n", name, T, best * 1e3, flops / best * 1e-9,
           hipGetErrorString(hipGetLastError()));
  };
  time_it("fp32 mfma", [&] { return launch_mlp_fused(a32, 0); });
  options().mlp_x3 = 2;
  time_it("3 x bf16 (cooperative split)", [&] { return launch_mlp_fused_x3(ax3, 0); });
  options().mlp_x3 = 1;
  time_it("3 x bf16 (every wave splits)", [&] { return launch_mlp_fused_x3(ax3, 0); });

  // accuracy on sampled rows
  std::vector<int> rows;
  for (int b : {0, 1, T / 128, T / 64 - 1}) for (int r : {0, 31, 32, 63}) rows.push_back(b * 64 + r);
  double e32 = 0, ex3 = 0, d32x3 = 0, omax = 0;
  for (int n = 0; n < 2; ++n) {
    const int nout = n == 0 ? 66 : 10;
    std::vector<float> o32((size_t)T * nout), ox3((size_t)T * nout);
    (void)hipMemcpy(o32.data(), a32.net[n].out, o32.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(ox3.data(), ax3.net[n].out, ox3.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < o32.size(); ++i) d32x3 = std::max(d32x3, (double)std::fabs(o32[i] - ox3[i]));
    for (int row : rows) {
      std::vector<double> cur(hx.begin() + (size_t)row * LDX, hx.begin() + (size_t)(row + 1) * LDX), nxt;
      for (auto& h : host[n]) {
        nxt.assign(h.N, 0.0);
        for (int j = 0; j < h.N; ++j) {
          double s = 0;
          for (int k = 0; k < h.K; ++k) s += cur[k] * (double)h.W[(size_t)j * h.K + k];
          s = s * (double)h.sc[j] + (double)h.sh[j];
          if (h.act) s = s >= 0 ? s : s * (double)h.slope;
          nxt[j] = s;
        }
        cur = nxt;
      }
      for (int j = 0; j < nout; ++j) {
        omax = std::max(omax, std::fabs(cur[j]));
        e32 = std::max(e32, std::fabs(cur[j] - (double)o32[(size_t)row * nout + j]));
        ex3 = std::max(ex3, std::fabs(cur[j] - (double)ox3[(size_t)row * nout + j]));
        if (std::isnan(ox3[(size_t)row * nout + j]) || std::isnan(o32[(size_t)row * nout + j])) ex3 = e32 = 1e30;
      }
    }
  }
  printf("max |out| %.3f;  max error vs float64 on %zu rows: fp32 mfma %.3e, 3 x bf16 %.3e;  max |fp32 - x3| over all rows %.3e\n",
         omax, rows.size(), e32, ex3, d32x3);
  return 0;
}
