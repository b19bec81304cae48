// EXPERIMENT, not built into the library (round 2): the software-pipelined form of mesh_rows_bf16_kernel.
// Result (MI355X, T = 16384, scripts/dev/mesh_bf16_lab.hip -DLAB_PIPE with this text pasted into mesh.hip):
// bit-identical output, 828 us per launch against 795 us for the plain kernel (K loop, then skinning).  Per-tile stamps:
// K loop alone 8.2 k cycles (252 MFMAs x 32 = 8.1 k: ideal), skinning alone 11.7 k, interleaved 23-28 k -- a wave's bf16
// MFMAs and its vector FMAs do NOT run concurrently on this chip (nor do those of two waves of one SIMD: see the header of
// mesh_rows_bf16_kernel), so interleaving them buys nothing and costs registers.  Kept for the record; what would help
// is fewer vector instructions in the skinning (e.g. the bone blend as a second bf16 contraction).
// Needs: mb_stage() (the staging block of mesh_rows_bf16_kernel as a function), MeshSkinArgs, mb:: constants.
// ---------------------------------------------------------------------------------------------------------------
// The software-pipelined form of mesh_rows_bf16_kernel (bodies with at most four bones per vertex): the K loop of a
// wave's NEXT tile runs interleaved with the skinning of the tile it just finished.  bf16 MFMAs execute in the matrix
// pipe while the same wave's vector FMAs issue (unlike fp32 MFMAs, which occupy the vector FMA lanes), so a tile costs
// about max(8.1 k MFMA cycles, 11 k skinning cycles) instead of their sum.  One wave per SIMD with the whole register
// file: two accumulator sets (the tile being skinned, the tile being contracted) and a seven-slot coefficient ring
// that runs six k-steps ahead, straight across tile boundaries (14 k-steps = 2 x 7 slots).  No branch inside the
// pipelined loop: the vertices are stored through a buffer resource that covers exactly this block's valid frames, so
// frames past T and the padding vertices of the last tile fall outside its range and are dropped by the hardware.
// ---------------------------------------------------------------------------------------------------------------
namespace mp {
constexpr int RING = 4;
static_assert((2 * mb::KS) % RING == 0, "ring slots must line up across a pair of tiles");
}
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ int n_tiles_all(int V) { return (V + 31) / 32; }

__global__ __launch_bounds__(mb::NW * 64) void mesh_rows_bf16_pipe_kernel(MeshSkinArgs a) {
  using namespace mb;
  constexpr int RING = mp::RING;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned short* Ab = reinterpret_cast<unsigned short*>(lds);
  float* XFs = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + A_BYTES);
  float* TRs = XFs + XF_FLOATS;
  const int T = a.T, V = a.V;
  const int f0 = blockIdx.x * BM;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  mb_stage(a, Ab, XFs, TRs, f0, tid);
  __syncthreads();

  const int n_tiles = (V + 31) / 32;
  const int per_block = (n_tiles + gridDim.y - 1) / gridDim.y;
  const int first = blockIdx.y * per_block;
  const int end = min(first + per_block, n_tiles);
  int vt = first + wave;
  if (vt >= end) return;

  // the coefficient table as a raw buffer: a fragment load is buffer_load_dwordx4 with the lane offset in one VGPR, the
  // tile / k-step offset in an SGPR and the (plane, piece) offset in the instruction -- no address arithmetic in VGPRs
  const __amdgpu_buffer_rsrc_t wtab = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.wc_bf16), 0, (int)((size_t)n_tiles_all(a.V) * TILE_BYTES), 0x00020000);
  const int lane16 = lane * 16;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  auto wload = [&](int tile_off, int ks, int c, int p) {
    const int f = c * 2 + p;   // fragment of the k-step: the part below 4 KB rides in the instruction's offset field
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wtab, lane16 + (f & 3) * 1024, tile_off + ks * 6144 + (f >> 2) * 4096, 0));
  };
  const char* a_lane = reinterpret_cast<const char*>(Ab) + l31 * (LDA * 2) + lh * 16;
  const char* xfl = reinterpret_cast<const char*>(XFs) + lh * (4 * NB * 48);
  const char* trl = reinterpret_cast<const char*>(TRs) + lh * 64;
  // this block's valid frames as a raw buffer: anything at or past num_records bytes is not written
  const int nf = min(BM, T - f0);
  const __amdgpu_buffer_rsrc_t vout = __builtin_amdgcn_make_buffer_rsrc(
      a.vertices + (size_t)f0 * V * 3, 0, (int)((size_t)nf * V * 12), 0x00020000);
  const unsigned vrow_bytes = (unsigned)V * 12u;

  f32x4 ring[RING][3][2];
  f32x4 fa[2][2][2];
  f32x16 acc0[2][3], acc1[2][3];   // ping-pong: one is contracted into while the other is skinned

  struct Skin { const char* xk[4]; f32x2 wp[4]; unsigned off; };
  auto make_skin = [&](int tile, const int4& bone4, const f32x4& w4) {
    Skin k;
    k.xk[0] = xfl + bone4.x * 48; k.xk[1] = xfl + bone4.y * 48; k.xk[2] = xfl + bone4.z * 48; k.xk[3] = xfl + bone4.w * 48;
#pragma unroll
    for (int j = 0; j < 4; ++j) k.wp[j] = f32x2{w4[j], w4[j]};
    const int s = tile * 32 + l31;
    k.off = s < V ? ((unsigned)(4 * lh) * (unsigned)V + (unsigned)s) * 12u : 0x80000000u;
    return k;
  };
  // one (frame, vertex) pair of the lane: blended 3x4 transform, mat-vec, 12-byte store (as mesh_rows_kernel)
  auto skin_item = [&](const f32x16 (&vp)[2][3], const Skin& k, int i, int r) {
    const int dm = i * 32 + (r & 3) + 8 * (r >> 2);
    const float vx = vp[i][0][r], vy = vp[i][1][r], vz = vp[i][2][r];
    const f32x4 tr = *reinterpret_cast<const f32x4*>(trl + dm * 16);
    float out[3];
#pragma unroll
    for (int row = 0; row < 3; ++row) {
      f32x4 gk = *reinterpret_cast<const f32x4*>(k.xk[0] + dm * (NB * 48) + row * 16);
      f32x2 Ta = k.wp[0] * f32x2{gk[0], gk[1]}, Tb = k.wp[0] * f32x2{gk[2], gk[3]};
#pragma unroll
      for (int j = 1; j < 4; ++j) {
        gk = *reinterpret_cast<const f32x4*>(k.xk[j] + dm * (NB * 48) + row * 16);
        Ta = __builtin_elementwise_fma(k.wp[j], f32x2{gk[0], gk[1]}, Ta);
        Tb = __builtin_elementwise_fma(k.wp[j], f32x2{gk[2], gk[3]}, Tb);
      }
      out[row] = __builtin_fmaf(Ta[0], vx, __builtin_fmaf(Ta[1], vy, __builtin_fmaf(Tb[0], vz, Tb[1]))) + tr[row];
    }
    const u32x3 o = {__float_as_uint(out[0]), __float_as_uint(out[1]), __float_as_uint(out[2])};
    __builtin_amdgcn_raw_buffer_store_b96(o, vout, (int)(k.off + (unsigned)dm * vrow_bytes), 0, 0);
  };
  // The K loop of tile `bt` into acc; with SKIN the 32 pairs of the previous tile (vp) are skinned between its MFMAs.
  // PAR = parity of the tile in the wave's sequence: k-step ks of the tile is step PAR * KS + ks of the ring.
  // A group is the six MFMAs of one product (six different accumulators).  32 of the 42 groups of a tile carry one
  // pair, spread over the group's six slots as an explicit pipeline -- slots 0..2 blend transform rows 0..2 from the
  // LDS reads issued two slots earlier, slot 3 does the mat-vec and the store:
  //   MFMA | blend row 0, read row 2      MFMA | blend row 1, read row 0 of the next pair
  //   MFMA | blend row 2, read row 1 of the next pair      MFMA | mat-vec, store, read the next translation      MFMA      MFMA
  // Nothing crosses a slot boundary (sched_barrier(0)): each MFMA is followed by about its own duration of vector work.
  auto kpass = [&](auto skin_tag, auto par_tag, f32x16 (&acc)[2][3], const f32x16 (&vp)[2][3], const Skin& k,
                   int bt, int bnext) {
    constexpr bool SKIN = decltype(skin_tag)::value;
    constexpr int S0 = decltype(par_tag)::value * KS;
    // two row buffers: rows 0 / 1 / 2 of pair `it` live in buffers p / p^1 / p (p = it & 1), so a row is read two slots
    // before it is blended and three rows are never live at once
    f32x4 gk[2][4], trv;
    f32x2 Ta[3], Tb[3];
    auto dm_of = [](int it) { const int i = it >> 4, r = it & 15; return i * 32 + (r & 3) + 8 * (r >> 2); };
    auto buf_of = [](int it, int row) { return (it + row) & 1; };
    auto rd_row = [&](int it, int row) {
      f32x4 (&b)[4] = gk[buf_of(it, row)];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f32x4*>(k.xk[j] + dm_of(it) * (NB * 48) + row * 16);
    };
    auto rd_tr = [&](int it) { trv = *reinterpret_cast<const f32x4*>(trl + dm_of(it) * 16); };
    auto blend = [&](int it, int row) {
      const f32x4 (&b)[4] = gk[buf_of(it, row)];
      Ta[row] = k.wp[0] * f32x2{b[0][0], b[0][1]};
      Tb[row] = k.wp[0] * f32x2{b[0][2], b[0][3]};
#pragma unroll
      for (int j = 1; j < 4; ++j) {
        Ta[row] = __builtin_elementwise_fma(k.wp[j], f32x2{b[j][0], b[j][1]}, Ta[row]);
        Tb[row] = __builtin_elementwise_fma(k.wp[j], f32x2{b[j][2], b[j][3]}, Tb[row]);
      }
    };
    auto finish = [&](int it) {
      const int i = it >> 4, r = it & 15;
      const float vx = vp[i][0][r], vy = vp[i][1][r], vz = vp[i][2][r];
      float out[3];
#pragma unroll
      for (int row = 0; row < 3; ++row)
        out[row] = __builtin_fmaf(Ta[row][0], vx, __builtin_fmaf(Ta[row][1], vy, __builtin_fmaf(Tb[row][0], vz, Tb[row][1]))) + trv[row];
      const u32x3 o = {__float_as_uint(out[0]), __float_as_uint(out[1]), __float_as_uint(out[2])};
      __builtin_amdgcn_raw_buffer_store_b96(o, vout, (int)(k.off + (unsigned)dm_of(it) * vrow_bytes), 0, 0);
    };
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;
    if (SKIN) { rd_row(0, 0); rd_row(0, 1); rd_tr(0); }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int kn = ks + RING - 1;                    // coefficient step requested now: this tile's, or the next one's
      const int bsrc = kn < KS ? bt : bnext;
      const int ksrc = kn < KS ? kn : kn - KS;
      f32x4 (&bn)[3][2] = ring[(S0 + kn) % RING];
      f32x4 (&fn)[2][2] = fa[(ks + 1) & 1];
      const int an = ks + 1 < KS ? ks + 1 : 0;
      const f32x4 (&fc)[2][2] = fa[ks & 1];
      const f32x4 (&bc)[3][2] = ring[(S0 + ks) % RING];
#pragma unroll
      for (int prod = 0; prod < 3; ++prod) {           // lo.hi, hi.lo, hi.hi
#pragma unroll
        for (int p = 0; p < 2; ++p)
          bn[prod][p] = wload(bsrc, ksrc, prod, p);
        if (prod < 2) {
#pragma unroll
          for (int p = 0; p < 2; ++p)
            fn[prod][p] = *reinterpret_cast<const f32x4*>(a_lane + p * A_PIECE_BYTES + prod * 32 * (LDA * 2) + an * 32);
        }
        const int g = ks * 3 + prod;
        const int it = (g * 32) / 42;
        const bool has = SKIN && ((g + 1) * 32) / 42 > it;   // this group carries pair `it`
#pragma unroll
        for (int slot = 0; slot < 6; ++slot) {
          const int i = slot / 3, c = slot % 3;
          const bf16x8 av = __builtin_bit_cast(bf16x8, fc[i][prod == 0 ? 1 : 0]);
          const bf16x8 bv = __builtin_bit_cast(bf16x8, bc[c][prod == 1 ? 1 : 0]);
          acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i][c], 0, 0, 0);
          if (has) {
            // slot 0: blend row 0, read row 2 | 1: blend row 1, read row 0 of the next pair | 2: blend row 2, read
            // row 1 of the next pair | 3: finish, read the next pair's translation
            if (slot < 3) blend(it, slot);
            if (slot == 0) rd_row(it, 2);
            if (slot == 1 && it + 1 < 32) rd_row(it + 1, 0);
            if (slot == 2 && it + 1 < 32) rd_row(it + 1, 1);
            if (slot == 3) { finish(it); if (it + 1 < 32) rd_tr(it + 1); }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (!has) __builtin_amdgcn_sched_barrier(0x6);   // never two MFMAs on one accumulator back to back
      }                                                  // (see mesh_rows_bf16_kernel)
    }
  };
  auto tile_ptr = [&](int t) { return __builtin_amdgcn_readfirstlane(t) * TILE_BYTES; };   // byte offset, wave-uniform
  auto next_ptr = [&](int t) { return tile_ptr(t + NW < end ? t + NW : t); };   // past the last tile: re-read it
  auto load_meta = [&](int t, int4& bone4, f32x4& w4) {
    const int s = t * 32 + l31;
    bone4 = *reinterpret_cast<const int4*>(a.skin_idx4 + (size_t)s * 4);
    w4 = *reinterpret_cast<const f32x4*>(a.skin_w4 + (size_t)s * 4);
  };
  auto skin_all = [&](const f32x16 (&vp)[2][3], const Skin& k) {
#pragma unroll
    for (int it = 0; it < 32; ++it) skin_item(vp, k, it >> 4, it & 15);
  };

  // ---- first tile: plain K loop
  int4 bone4; f32x4 w4;
  load_meta(vt, bone4, w4);
  {
    const int b0 = tile_ptr(vt);
#pragma unroll
    for (int ks = 0; ks < RING - 1; ++ks)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int p = 0; p < 2; ++p) ring[ks][c][p] = wload(b0, ks, c, p);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        fa[0][i][p] = *reinterpret_cast<const f32x4*>(a_lane + p * A_PIECE_BYTES + i * 32 * (LDA * 2));
  }
  Skin sk = make_skin(vt, bone4, w4);
  MR_STAMP(0, 0)
  kpass(std::false_type{}, std::integral_constant<int, 0>{}, acc0, acc1, sk, tile_ptr(vt), next_ptr(vt));
  MR_STAMP(0, 1)
  MR_STAMP(0, 2)

  // ---- steady state, two tiles per trip: contract the next tile while skinning the one just contracted
  int seq = 1;
#pragma unroll 1
  for (;;) {
    if (vt + NW >= end) { skin_all(acc0, sk); return; }
    {
      MR_STAMP(seq, 0)
      const int nx = vt + NW;
      load_meta(nx, bone4, w4);
      kpass(std::true_type{}, std::integral_constant<int, 1>{}, acc1, acc0, sk, tile_ptr(nx), next_ptr(nx));
      sk = make_skin(nx, bone4, w4);
      vt = nx;
      MR_STAMP(seq, 1)
      MR_STAMP(seq, 2)
      ++seq;
    }
    if (vt + NW >= end) { skin_all(acc1, sk); return; }
    {
      MR_STAMP(seq, 0)
      const int nx = vt + NW;
      load_meta(nx, bone4, w4);
      kpass(std::true_type{}, std::integral_constant<int, 0>{}, acc0, acc1, sk, tile_ptr(nx), next_ptr(nx));
      sk = make_skin(nx, bone4, w4);
      vt = nx;
      MR_STAMP(seq, 1)
      MR_STAMP(seq, 2)
      ++seq;
    }
  }
}

