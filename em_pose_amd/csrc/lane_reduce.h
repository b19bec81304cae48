// Sums over the 64 lanes of a wave of N values per lane as a reduce-scatter: every step that can halves what a lane carries
// (the lane keeps one half of its values and receives the partner's), so N sums cost about N lane exchanges instead of
// 6 N.  Each step adds a lane's value and its partner's -- the same pairs as the butterfly all-reduce (v += shfl_xor(v, off),
// off = 32 .. 1), so every sum has the same bits as that one.
#pragma once
#include <hip/hip_runtime.h>

namespace empose {

template <int N, int OFF, int CAP>
struct LaneReduceScatter {   // v[0 .. N) summed over lanes; on return the lane holds sums base .. base + count - 1 of the original N
  static __device__ __forceinline__ void run(float (&v)[CAP], int lane, int& base, int& count) {
    if constexpr (OFF == 0) {
      count = N;
    } else if constexpr (N % 2 == 0) {
      const bool up = (lane & OFF) != 0;
#pragma unroll
      for (int i = 0; i < N / 2; ++i) {
        const float send = up ? v[i] : v[i + N / 2];
        const float keep = up ? v[i + N / 2] : v[i];
        v[i] = keep + __shfl_xor(send, OFF, 64);
      }
      base += up ? N / 2 : 0;
      LaneReduceScatter<N / 2, OFF / 2, CAP>::run(v, lane, base, count);
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] += __shfl_xor(v[i], OFF, 64);
      LaneReduceScatter<N, OFF / 2, CAP>::run(v, lane, base, count);
    }
  }
};

}  // namespace empose
