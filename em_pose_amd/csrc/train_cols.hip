// One train-mode layer of the update MLPs at the reference's training batch (12 windows x 32 frames = 384 rows, reference
// scripts/train.py:125-152; layer = Linear -> BatchNorm1d(train) -> PReLU, nn/layers.py:13-77) as ONE launch, forward or
// backward, both networks of an iteration side by side.
//
// At 384 rows a layer is 0.2 GFLOP: the step was a chain of ~390 dependent launches of 3-20 us on 24-48 workgroups (a
// split-K product, then a BatchNorm kernel that walks all rows of 32 columns), each near the floor of a launch.  Here a
// layer's product and its BatchNorm meet in one kernel, on the whole chip:
//   * a workgroup owns 16 output columns x one quarter of the rows (R = 4 row parts of up to 8 tiles of 16 rows);
//     hidden width 512 and two networks: 32 x 4 x 2 = 256 workgroups, mapped so that an XCD serves one network;
//   * its eight waves split K (a contiguous range of 16-k blocks each), every wave keeps all row tiles of the part in registers
//     (v_mfma_f32_16x16x4_f32, fp32 operands straight from the row-major arrays: a lane reads 16 bytes of a row and feeds
//     one float to each of four MFMAs -- the k order inside a block is permuted the same way on both operands);
//     the partial sums meet in LDS and are added in wave order;
//   * BatchNorm needs column sums over ALL rows: the R workgroups of a column slice exchange their per-column partial
//     statistics through a mailbox of tagged 8-byte words in device memory (value | tag, relaxed agent-scope atomics: the
//     path lstm_persist_kernel uses between its workgroups; 0.5-0.6 us per hand-over, scripts/dev/xcd_pingpong.hip) and
//     every one of them combines the R parts in part order -- the same bits in all of them, no second launch;
//       forward : per part (n, mean, M2 = sum (y - mean)^2), combined pairwise (Chan et al.) -> mean, biased var -> rstd
//       backward: per part (sum dy, sum dy xhat, sum_{y <= 0} dA y), added in part order
//   * the tag is the layer's number inside the call; the caller zeroes the mailbox once per call (a memset node when the
//     step is captured as a graph).  A poll that exceeds its spin limit poisons the outputs and is counted
//     (poll_timeout_word), it does not hang.  All workgroups of a launch fit the chip at once (<= 256 at width 512).
// What it writes is what the layer-by-layer path writes (save layout 1: z | a | mean | rstd), so forward and backward
// choose their path independently.
#include "gemm_epilogue.h"
#include "kernels.h"

namespace empose {

namespace tc {
constexpr int NT = 512, NW8 = NT / 64, MAX_TILES = 8, RG = NT / 16, VT = MAX_TILES * 16 / RG;
constexpr int CH = 4;                                     // 16-k blocks of a wave per round (its whole K at width 512): loads two blocks ahead of the products
constexpr int TLD = 17, TSZ = 16 * TLD;                   // padded 16 x 16 tile of partial sums
constexpr int PART_FLOATS = NW8 * MAX_TILES * TSZ;        // [wave][tile][row][col]
}  // namespace tc

__device__ __forceinline__ unsigned long long tc_pack(float v, unsigned tag) {
  return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}

// sum over the row groups of a column (every thread of the column gets the same sum, added in group order)
__device__ __forceinline__ float tc_reduce(float v, float* red, int rg, int c) {
  red[rg * 16 + c] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < tc::RG; ++g) s += red[g * 16 + c];
  __syncthreads();
  return s;
}

// MODE 0: forward (a.plain: product + bias only, the output layer; else with BatchNorm + PReLU); 2: backward
#ifdef TC_LAB_TIMES      // (scripts/dev/train_cols_lab.hip: shader-clock stamps of thread 0 of every workgroup)
__device__ long long* tc_lab_times;
#define TC_STAMP(i) do { if (threadIdx.x == 0) tc_lab_times[blockIdx.x * 8 + (i)] = clock64(); } while (0)
#else
#define TC_STAMP(i) do { } while (0)
#endif

template <int MODE>
__global__ __launch_bounds__(tc::NT) void cols_kernel(ColsArgs a) {
  using namespace tc;
  TC_STAMP(0);
  __shared__ float part[PART_FLOATS];
  __shared__ float red[3 * RG * 16];
  __shared__ float xs[4 * 3 * 16];     // [part][word][col]: what the parts posted
  __shared__ int bad_lds;
  // Workgroup -> (network, column slice, row part), XCD-aware: consecutive workgroup ids go round the 8 XCDs, so id & 7 is
  // the XCD; one XCD takes ONE network and 1 / groups of its column slices with all their row parts -- its L2 then holds
  // that network's rows once and only its own slices' weights (1.0 MB instead of 1.8 at two networks of width 512), and
  // the parts that exchange statistics sit behind the same L2.
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int groups = 8 / a.n_nets, net_i = xcd / groups;
  const int slice = (xcd % groups) * a.slices_per_group + j % a.slices_per_group, r = j / a.slices_per_group;
  const ColsNet& n = a.net[net_i];
  const int N = n.N, K = n.K, M = a.M;
  if (slice * 16 >= N) return;
  const int T16 = (M + 15) >> 4;
  const int t0 = r * a.tiles_per_part;
  const int ntl = min(a.tiles_per_part, T16 - t0);
  if (ntl <= 0) return;
  const int row0 = t0 * 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;

  // the epilogue's element group: thread (row group rg of 32, column c) owns rows row0 + 32 t + rg.  What the reverse reads
  // besides the product -- the saved pre-BatchNorm rows and the layer's constants -- is fetched now, under the product.
  const int c = tid & 15, rg = tid >> 4;
  const int col = slice * 16 + c;
  const bool colok = col < N;
  const int cc = colok ? col : N - 1;
  bool ok[VT];
#pragma unroll
  for (int t = 0; t < VT; ++t) ok[t] = ((t * RG + rg) >> 4) < ntl && row0 + t * RG + rg < M;
  float zin[VT], e_mean = 0.f, e_rstd = 0.f, e_gm = 0.f, e_bt = 0.f, e_slope = 0.f;
  if constexpr (MODE == 2) {
    e_mean = n.mean[cc]; e_rstd = n.rstd[cc]; e_gm = n.gamma[cc]; e_bt = n.beta[cc]; e_slope = n.slope[0];
#pragma unroll
    for (int t = 0; t < VT; ++t) zin[t] = n.z_in[(size_t)min(row0 + t * RG + rg, M - 1) * n.ldz + cc];
  }

  // ---- the product: wave w's contiguous range of 16-k blocks, all row tiles of this part.  A layer at this size is bound
  // by getting the operands into the CU (L2 -> CU at ~25 bytes per clock), not by the matrix cores: the loads run two
  // blocks ahead of the products so that the memory pipeline stays full.
  f32x4 acc[MAX_TILES];
#pragma unroll
  for (int t = 0; t < MAX_TILES; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const int KB = (K + 15) >> 4;
    const int per = (KB + NW8 - 1) / NW8;
    const int kb0 = wave * per, kb1 = min(KB, kb0 + per);
    const int wcol = min(slice * 16 + r16, N - 1);
    const float* wrow = n.W + (size_t)wcol * n.ldw;
    int aoff[MAX_TILES];
#pragma unroll
    for (int t = 0; t < MAX_TILES; ++t) aoff[t] = min(row0 + t * 16 + r16, M - 1) * n.lda;
#pragma unroll 1
    for (int kb = kb0; kb < kb1; kb += CH) {
      f32x4 fa[CH][MAX_TILES], fw[CH];
      auto load = [&](int q) {
        // K % 4 == 0: a lane's four floats are inside or outside together.  Outside (the last block of a ragged K, blocks
        // past the wave's range) the lane reads the first four floats of the same rows instead -- no branch around a load,
        // nothing read past a row's K -- and its weights count as zero (the row's own values times zero: what the row's
        // sum holds anyway if they are not finite)
        const bool in = kb + q < kb1 && (kb + q) * 16 + 4 * g < K;
        const int k0 = in ? (kb + q) * 16 + 4 * g : 0;
#pragma unroll
        for (int t = 0; t < MAX_TILES; ++t)
          if (t < ntl) fa[q][t] = *reinterpret_cast<const f32x4*>(n.A + aoff[t] + k0);
        f32x4 wv;
        if (MODE == 2 && n.w_kmajor) {   // the layer's own W [k][columns] (no transposed copy): four 64-byte row pieces
#pragma unroll
          for (int e = 0; e < 4; ++e) {          // (W has Kw <= K rows: the cotangent's zero padding columns have none)
            const float wk = n.W[(size_t)min(k0 + e, n.Kw - 1) * n.ldw + wcol];
            wv[e] = k0 + e < n.Kw ? wk : 0.f;
          }
        } else {
          wv = *reinterpret_cast<const f32x4*>(wrow + k0);
        }
        fw[q] = in ? wv : f32x4{0.f, 0.f, 0.f, 0.f};
      };
      auto mma = [&](int q) {
#pragma unroll
        for (int sI = 0; sI < 4; ++sI)
#pragma unroll
          for (int t = 0; t < MAX_TILES; ++t)
            if (t < ntl) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[q][t][sI], fw[q][sI], acc[t], 0, 0, 0);
      };
      // loads two blocks ahead of the products: the memory pipeline of the CU stays full (issuing all of a wave's loads at
      // once stalls it for as long as the data takes to arrive, with the matrix cores idle: scripts/dev/train_cols_lab.hip)
      load(0); load(1);
      __builtin_amdgcn_sched_barrier(0);
      load(2); mma(0);
      __builtin_amdgcn_sched_barrier(0);
      load(3); mma(1);
      __builtin_amdgcn_sched_barrier(0);
      TC_STAMP(1);
      mma(2); mma(3);
    }
  }
  // the C/D layout of the 16 x 16 MFMA: a lane holds column r16, rows 4 g .. 4 g + 3
#pragma unroll
  for (int t = 0; t < MAX_TILES; ++t)
    if (t < ntl) {
#pragma unroll
      for (int j = 0; j < 4; ++j) part[(wave * MAX_TILES + t) * TSZ + (4 * g + j) * TLD + r16] = acc[t][j];
    }
  if (tid == 0) bad_lds = 0;
  TC_STAMP(2);
  __syncthreads();
  TC_STAMP(3);

  // ---- epilogue
  float v[VT];
#pragma unroll
  for (int t = 0; t < VT; ++t) {
    const int lr = t * RG + rg, tile = lr >> 4;       // row inside the part, its 16-row tile
    float sum = 0.f;
    if (tile < ntl) {
      const float* ps = part + tile * TSZ + (lr & 15) * TLD + c;
#pragma unroll
      for (int w = 0; w < NW8; ++w) sum += ps[w * MAX_TILES * TSZ];     // wave order
    }
    v[t] = sum;
  }
  if (MODE == 0 && a.plain) {      // the output layer: product + bias, nothing to exchange
    const float b = n.bias[cc];
#pragma unroll
    for (int t = 0; t < VT; ++t)
      if (ok[t] && colok) n.out[(size_t)(row0 + t * RG + rg) * n.ld_out + col] = v[t] + b;
    return;
  }

  // what the parts post: NW words per column; word w of part q, column c at mailbox[((q * 16 + c) * 3 + w]
  constexpr int NW = MODE == 0 ? 2 : 3;
  unsigned long long* mb = a.mailbox + ((size_t)(net_i * a.s_max + slice) * 4) * 16 * 3;
  auto exchange = [&](const float (&mine)[NW]) {
    if (rg == 0) {
#pragma unroll
      for (int w = 0; w < NW; ++w)
        __hip_atomic_store(mb + ((size_t)r * 16 + c) * 3 + w, tc_pack(mine[w], a.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (rg < a.R && min(a.tiles_per_part, T16 - rg * a.tiles_per_part) > 0) {     // wave 0: row group q polls part q
      unsigned long long wv[NW];
      const unsigned long long* src = mb + ((size_t)rg * 16 + c) * 3;
#pragma unroll
      for (int w = 0; w < NW; ++w) wv[w] = __hip_atomic_load(src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        int spins = 0;
        while ((unsigned)(wv[w] >> 32) != a.tag) {
          if (++spins > a.spin_limit) { bad_lds = 1; wv[w] = tc_pack(__builtin_nanf(""), a.tag); break; }
          __builtin_amdgcn_s_sleep(1);
          wv[w] = __hip_atomic_load(src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        xs[(rg * 3 + w) * 16 + c] = __uint_as_float((unsigned)wv[w]);
      }
    }
    __syncthreads();
    if (bad_lds && tid == 0) __hip_atomic_fetch_add(a.timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  const float inv_m = 1.f / (float)M;

  if constexpr (MODE == 0) {
    const float b = n.bias[cc];
    const int n_mine = min(ntl * 16, M - row0);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < VT; ++t) { v[t] += b; s += ok[t] ? v[t] : 0.f; }
    const float mean_l = tc_reduce(s, red, rg, c) / (float)n_mine;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < VT; ++t) { const float d = v[t] - mean_l; q += ok[t] ? d * d : 0.f; }
    const float m2_l = tc_reduce(q, red, rg, c);
    if (colok) {                          // (the pre-BatchNorm rows do not wait for the other parts)
#pragma unroll
      for (int t = 0; t < VT; ++t)
        if (ok[t]) n.z[(size_t)(row0 + t * RG + rg) * n.ldz + col] = v[t];
    }
    const float mine[2] = {mean_l, m2_l};
    TC_STAMP(4);
    exchange(mine);
    TC_STAMP(5);
    // the parts in part order (pairwise update of mean and centred sum of squares)
    float cnt = 0.f, mean = 0.f, m2 = 0.f;
    for (int p = 0; p < a.R; ++p) {
      const int np = min(a.tiles_per_part * 16, M - p * a.tiles_per_part * 16);
      if (np <= 0) break;
      const float mp = xs[(p * 3 + 0) * 16 + c], qp = xs[(p * 3 + 1) * 16 + c];
      const float tot = cnt + (float)np, d = mp - mean;
      mean += d * ((float)np / tot);
      m2 += qp + d * d * (cnt * (float)np / tot);
      cnt = tot;
    }
    const float var = m2 * inv_m;                                   // biased: what normalises the batch
    const float rstd = 1.f / sqrtf(var + a.eps);
    if (colok) {
      const float gm = n.gamma[col], bt = n.beta[col], slope = n.slope[0];
#pragma unroll
      for (int t = 0; t < VT; ++t)
        if (ok[t]) {
          const size_t row = (size_t)(row0 + t * RG + rg);
          const float y = gm * ((v[t] - mean) * rstd) + bt;
          n.out[row * n.ld_out + col] = y > 0.f ? y : slope * y;
        }
      if (r == 0 && rg == 0) {
        n.mean[col] = mean;
        n.rstd[col] = rstd;
        if (n.running_mean) {
          const float unbiased = M > 1 ? var * (float)M / (float)(M - 1) : var;
          n.running_mean[col] = (1.f - a.momentum) * n.running_mean[col] + a.momentum * mean;
          n.running_var[col] = (1.f - a.momentum) * n.running_var[col] + a.momentum * unbiased;
        }
      }
    }
    if (slice == 0 && r == 0 && tid == 0 && n.num_batches) n.num_batches[0] += 1;
    TC_STAMP(6);
    return;
  }

  if constexpr (MODE == 2) {
    const float mean = e_mean, rstd = e_rstd, gm = e_gm, bt = e_bt, slope = e_slope;
    float xh[VT], dy[VT];
    float s_b = 0.f, s_g = 0.f, s_a = 0.f;
#pragma unroll
    for (int t = 0; t < VT; ++t) {
      const float z = ok[t] ? zin[t] : mean;
      const float da = ok[t] ? v[t] : 0.f;
      xh[t] = (z - mean) * rstd;
      const float y = gm * xh[t] + bt;
      dy[t] = y > 0.f ? da : slope * da;
      s_b += dy[t];
      s_g += dy[t] * xh[t];
      s_a += y > 0.f ? 0.f : da * y;
    }
    // the three column sums over the row groups in one pass through LDS (each in group order, as tc_reduce adds them)
    float mine[3] = {0.f, 0.f, 0.f};
    red[rg * 16 + c] = s_b;
    red[RG * 16 + rg * 16 + c] = s_g;
    red[2 * RG * 16 + rg * 16 + c] = colok ? s_a : 0.f;
    __syncthreads();
#pragma unroll
    for (int gI = 0; gI < RG; ++gI) {
      mine[0] += red[gI * 16 + c];
      mine[1] += red[RG * 16 + gI * 16 + c];
      mine[2] += red[2 * RG * 16 + gI * 16 + c];
    }
    __syncthreads();
    exchange(mine);
    float dbeta = 0.f, dgamma = 0.f, dslope_c = 0.f;
    for (int p = 0; p < a.R; ++p) {
      if (M - p * a.tiles_per_part * 16 <= 0) break;
      dbeta += xs[(p * 3 + 0) * 16 + c];
      dgamma += xs[(p * 3 + 1) * 16 + c];
      dslope_c += xs[(p * 3 + 2) * 16 + c];
    }
    if (colok) {
      const float k = gm * rstd * inv_m;
#pragma unroll
      for (int t = 0; t < VT; ++t)
        if (ok[t]) n.out[(size_t)(row0 + t * RG + rg) * n.ld_out + col] = k * ((float)M * dy[t] - dbeta - xh[t] * dgamma);
      if (r == 0 && rg == 0) {
        n.dgamma[col] = dgamma + (a.accumulate ? n.dgamma[col] : 0.f);
        n.dbeta[col] = dbeta + (a.accumulate ? n.dbeta[col] : 0.f);
      }
    }
    // slope: part 0's workgroup of a slice adds its columns and posts the sum like the statistics (a tagged word per slice
    // behind the statistics' words); slice 0's workgroup waits for all of them and adds them in slice order.  (A last-arriver
    // counter needs a release fence in every workgroup -- an L2 write-back on this chip, 5 us per launch.)
    if (r == 0) {
      if (rg == 0) red[c] = dslope_c;
      __syncthreads();
      unsigned long long* sw = a.mailbox + (size_t)a.n_nets * a.s_max * 4 * 16 * 3 + (size_t)net_i * a.s_max;
      if (tid == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += red[i];
        __hip_atomic_store(sw + slice, tc_pack(t, a.tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (slice == 0) {
        const int n_slices = (N + 15) >> 4;
        for (int i = tid; i < n_slices; i += NT) {
          unsigned long long wv = __hip_atomic_load(sw + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          int spins = 0;
          while ((unsigned)(wv >> 32) != a.tag) {
            if (++spins > a.spin_limit) { bad_lds = 2; wv = tc_pack(__builtin_nanf(""), a.tag); break; }
            __builtin_amdgcn_s_sleep(1);
            wv = __hip_atomic_load(sw + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          part[i] = __uint_as_float((unsigned)wv);       // (the partial sums of the product are long consumed)
        }
        __syncthreads();
        if (tid == 0) {
          if (bad_lds == 2) __hip_atomic_fetch_add(a.timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          float total = 0.f;
          for (int i = 0; i < n_slices; ++i) total += part[i];
          n.dslope[0] = total + (a.accumulate ? n.dslope[0] : 0.f);
        }
      }
    }
  }
}

// per network and column slice: 4 parts x 16 columns x 3 statistics words, then one slope word
size_t cols_mailbox_words(int n_max) { return (size_t)2 * ((n_max + 15) / 16) * (4 * 16 * 3 + 1); }

bool cols_launchable(int n_max, int n_nets) {
  static int resident[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // per device: workgroups of the heaviest instantiation
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 8) return false;
  if (resident[dev] == 0) {
    int b0 = 0, b2 = 0;
    if (coresident_blocks(reinterpret_cast<const void*>(cols_kernel<0>), tc::NT, 0, &b0) != hipSuccess) return false;
    if (coresident_blocks(reinterpret_cast<const void*>(cols_kernel<2>), tc::NT, 0, &b2) != hipSuccess) return false;
    resident[dev] = b0 < b2 ? b0 : b2;
  }
  const int groups = 8 / n_nets, s_max = (n_max + 15) / 16;
  return (long)8 * ((s_max + groups - 1) / groups) * 4 <= resident[dev];
}

hipError_t launch_cols(ColsArgs a, int mode, hipStream_t stream) {
  if (a.n_nets < 1 || a.n_nets > 2 || a.M < 1 || a.M > COLS_MAX_ROWS) return hipErrorInvalidValue;
  const int T16 = (a.M + 15) / 16;
  a.R = T16 < 4 ? T16 : 4;
  a.tiles_per_part = (T16 + a.R - 1) / a.R;
  a.spin_limit = options().spin_limit > 0 ? options().spin_limit : 1 << 20;
  a.timeouts = poll_timeout_word();
  if (!a.timeouts) return hipErrorOutOfMemory;
  int n_max = 0;
  for (int i = 0; i < a.n_nets; ++i) n_max = a.net[i].N > n_max ? a.net[i].N : n_max;
  a.s_max = (n_max + 15) / 16;
  const int groups = 8 / a.n_nets;
  a.slices_per_group = (a.s_max + groups - 1) / groups;
  const dim3 grid(8 * a.slices_per_group * a.R);
  a.plain = mode == 1;
  // The workgroups of a launch wait for each other's exchange words, so ALL of them must be resident.  Eager launches ask
  // the runtime for exactly that guarantee (hipLaunchCooperativeKernel: the dispatch does not start until the whole grid
  // fits, whatever else this or another stream has on the device) -- round 6; before, residency was only ARGUED from the
  // occupancy query and a kernel of another stream could starve a late workgroup into a poll timeout.  A stream that is
  // being CAPTURED takes the ordinary launch (a cooperative node is not something the graph of this runtime records): the
  // replay is then safe only alone on the device, which is what the caller of a capture asserts (nn/train_engine.py
  // refuses to capture with side streams forked; option "cols_coop" = 0 forces the ordinary launch for A/B timing).
  const void* fn = mode <= 1 ? reinterpret_cast<const void*>(cols_kernel<0>) : reinterpret_cast<const void*>(cols_kernel<2>);
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &capturing) != hipSuccess) { (void)hipGetLastError(); capturing = hipStreamCaptureStatusNone; }
  if (capturing == hipStreamCaptureStatusNone && options().cols_coop) {
    void* params[] = {&a};
    if (hipLaunchCooperativeKernel(fn, grid, dim3(tc::NT), params, 0u, stream) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();      // no cooperative launch in this context: fall back to the argued residency
  }
  if (mode <= 1) hipLaunchKernelGGL(cols_kernel<0>, grid, dim3(tc::NT), 0, stream, a);
  else hipLaunchKernelGGL(cols_kernel<2>, grid, dim3(tc::NT), 0, stream, a);
  return hipGetLastError();
}

}  // namespace empose
