// SMPL-H on the sensor sub-mesh, one FRAME PER LANE (round 3; replaces chain_sensors_kernel for large launches).
//
// chain_sensors_kernel (smpl.hip) maps (frame, item) pairs to lanes: every lane decodes index tables, splits its flat
// index into (frame, item) and re-reads bone transforms from LDS for a third of a mat-vec -- 3290 VALU wave-instructions
// per frame for ~15 k useful multiply-adds (profiles/r02_chain_counters.txt: 14x instruction overhead, issue-bound).
// Here a wave owns 64 FRAMES (lane = frame) and walks the items itself:
//   * everything that is an index, a weight or a loop count is wave-uniform: it lives in scalar registers, comes from
//     one small table (TileTables, scalar loads) and costs no vector instruction; branches on it are scalar branches;
//   * every vector instruction is a useful multiply-add (or load / store) for 64 frames at once;
//   * per-frame data in global memory is laid out frame-minor ("tile layout": [tile of 64 frames][column][64]), so a
//     column load of a wave is one coalesced 256-byte read; the blend-shape GEMMs write / read that layout
//     (gemm_rows_t_kernel, mlp_fused.hip);
//   * a sensor's local patch (centre + ring of <= 8 vertices, <= 4 bones per vertex) is register-resident from skinning to
//     its cotangents; the tables give each sensor its own copies of its vertices (columns of the blend matrix are
//     repeated for vertices two sensors share -- the transposed GEMM adds the copies' cotangents by itself), the ring in
//     fan order (face k = (centre, ring k, ring k+1)) and its vertices' (bone, weight) lists, so no register array is
//     ever indexed by a run-time value;
//   * only what crosses sensors lives in LDS: the 22 joint transforms (written by the chain phase) and the per-bone
//     force / rest-space moment sums (added by the sensors with plain read-add-write IN SENSOR ORDER -- a turn counter
//     in LDS serialises just that short update -- so the sums are ordered and reproducible).
// Phases of a workgroup (4 waves with the reverse pass, 8 without; 64 frames): Rodrigues + chain, a wave per limb path |
// sensors, wave w takes w, w + NW, ... | reverse chain, a wave per limb, then the spine.  Maths: reference models.py:471-483, 560-579, virtual_sensors.py:16-38, utils.py:126-146,
// loss.py:23-41; reverse pass as in oracle/analytic_np.py (same formulas as chain_sensors_kernel).
#include "kernels.h"
#include "smpl_math.h"
#include "feat_rows.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace empose {

namespace tl {
constexpr int FR = TL_FR;
constexpr int GW = 15;                              // per bone in LDS: R (9) | t (3) | A^t = t - R J (3)
constexpr int LDS_G = NB * GW * FR;                 // floats
constexpr int LDS_M = NB * 12 * FR;                 // per bone: rest-space moment N (9) | force F (3); first the 22 R_j
constexpr size_t LDS_BYTES = (size_t)(LDS_G + LDS_M) * sizeof(float) + 64;   // + the sensors' turn counter
}  // namespace tl

// ---------------------------------------------------------------------------------------------------------------
// host: tables
// ---------------------------------------------------------------------------------------------------------------
bool build_tile_tables(int nv, int kb, int max_deg, int j_off, const float* wc, const int* parents, const int* skin_idx,
                       const float* skin_w, const int* s_center, const int* s_helper, const int* s_deg,
                       const int* s_faces, TileTables* out, std::vector<float>* wc2) {
  using namespace tl;
  TileTables& tb = *out;
  std::memset(&tb, 0, sizeof(tb));
  if (max_deg > TL_NR || nv <= 0) return false;
  int nloc_max = 0, nbl_max = 0;
  std::vector<std::vector<int>> local(12);   // needed-vertex ids of each sensor's local vertices: centre, ring 0..deg-1
  for (int m = 0; m < 12; ++m) {
    const int deg = s_deg[m], c = s_center[m];
    if (deg < 3 || deg > TL_NR) return false;
    // every face (a, b, cc) of the sensor, rotated so that the centre comes first: (c, p, q); (v1-v0)x(v2-v0) does not
    // change under a cyclic rotation of the triangle.  A closed, consistently oriented fan: q of one face is p of the next.
    std::vector<int> p(deg), q(deg);
    for (int k = 0; k < deg; ++k) {
      const int* f = s_faces + ((size_t)m * max_deg + k) * 3;
      int at = -1;
      for (int e = 0; e < 3; ++e) if (f[e] == c) at = at < 0 ? e : 3;
      if (at < 0 || at > 2) return false;
      p[k] = f[(at + 1) % 3]; q[k] = f[(at + 2) % 3];
      if (p[k] == c || q[k] == c || p[k] == q[k]) return false;
    }
    std::vector<int> ring;
    std::vector<char> used(deg, 0);
    int cur = 0;
    for (int step = 0; step < deg; ++step) {
      if (used[cur]) return false;
      used[cur] = 1;
      ring.push_back(p[cur]);
      int nxt = -1;
      for (int k = 0; k < deg; ++k) if (p[k] == q[cur]) nxt = nxt < 0 ? k : -2;
      if (nxt < 0) return false;       // open fan or a vertex that starts two faces
      cur = nxt;
    }
    if (cur != 0) return false;        // the cycle closes on the face it started from
    for (size_t i = 0; i < ring.size(); ++i)
      for (size_t j = i + 1; j < ring.size(); ++j) if (ring[i] == ring[j]) return false;
    TileSensor& S = tb.s[m];
    S.deg = deg;
    S.helper = -1;
    for (int k = 0; k < deg; ++k) if (ring[k] == s_helper[m]) S.helper = k;
    if (S.helper < 0) return false;    // a caller-given helper vertex outside the ring: the general kernel takes it
    local[m].push_back(c);
    for (int k = 0; k < deg; ++k) local[m].push_back(ring[k]);
    nloc_max = std::max(nloc_max, deg + 1);
    // bones of the patch and dense weights over them
    S.nb = 0;
    S.bones = 0;
    for (int i = 0; i <= deg; ++i) {
      const int s = local[m][i];
      if (s < 0 || s >= nv) return false;
      for (int k = 0; k < kb; ++k) {
        const float w = skin_w[(size_t)s * kb + k];
        if (w == 0.f) continue;
        const int b = skin_idx[(size_t)s * kb + k];
        if (b < 0 || b >= NB) return false;
        int slot = -1;
        for (int u = 0; u < S.nb; ++u) if (S.bone[u] == b) slot = u;
        if (slot < 0) {
          if (S.nb == TL_NBL) return false;
          slot = S.nb++;
          S.bone[slot] = b;
          S.bones |= 1 << b;
        }
        S.w[i][slot] += w;
      }
    }
    nbl_max = std::max(nbl_max, S.nb);
  }
  // columns: sensor m, local vertex i, coordinate c -> (m * nloc_max + i) * 3 + c; then the 22 rest joints
  const int nloc = nloc_max;
  tb.nloc = nloc;
  tb.nbl = nbl_max;
  tb.j_off2 = (12 * nloc * 3 + 3) & ~3;
  tb.ncp2 = (tb.j_off2 + NB * 3 + 31) & ~31;   // whole 32-column tiles of the matrix-core GEMMs
  wc2->assign((size_t)tb.ncp2 * 200, 0.f);
  for (int m = 0; m < 12; ++m) {
    tb.s[m].col = m * nloc * 3;
    for (size_t i = 0; i < local[m].size(); ++i)
      for (int c = 0; c < 3; ++c)
        std::memcpy(wc2->data() + ((size_t)tb.s[m].col + i * 3 + c) * 200, wc + ((size_t)local[m][i] * 3 + c) * 200,
                    200 * sizeof(float));
  }
  for (int r = 0; r < NB * 3; ++r)
    std::memcpy(wc2->data() + ((size_t)tb.j_off2 + r) * 200, wc + ((size_t)j_off + r) * 200, 200 * sizeof(float));
  tb.n_rounds = 0;   // (the kernels deal sensor m to wave m % waves in round m / waves themselves)
  // the chain phases are written out for the SMPL body tree (reference configuration.py:118); any other tree keeps the
  // general kernel
  static const int smpl_tree[NB] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19};
  for (int j = 0; j < NB; ++j) {
    if (parents[j] != smpl_tree[j]) return false;
    tb.parent[j] = parents[j];
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------------------------
#ifdef EMPOSE_CHAIN_TRACE   // dev build only (scripts/dev/chain_trace.sh): shader-clock stamps per wave of block 0
__device__ long long g_tile_trace[8][32];
#define TL_STAMP(i) if (blockIdx.x == 0 && lane == 0) g_tile_trace[wave][(i)] = clock64();
#define TL_SSTAMP(i) if (blockIdx.x == 0 && lane == 0 && rnd == 0) g_tile_trace[wave][(i)] = clock64();
#else
#define TL_STAMP(i)
#define TL_SSTAMP(i)
#endif

namespace {

template <int NLOC>
struct SensorIn {            // what a sensor reads from global memory
  float vp[NLOC][3];
  float Ro[9], to[3];
  float tp[3], tori[9], scale;     // targets (residual) ...
  float dpos[3], dori[9];          // ... or external cotangents
};

// Everything a sensor reads from global memory.  Issued one round ahead of its use (the loads of the next sensor fly
// while the current one computes: a wave has nobody to hide their latency behind).
template <bool BWD, int NLOC>
__device__ __forceinline__ void tile_sensor_load(const TileArgs& a, const TileTables& tb, int m, int lane, int tc,
                                                 const float* __restrict__ ot, SensorIn<NLOC>& in) {
  using namespace tl;
  const TileSensor& S = tb.s[m];
  const int col = S.col, nloc = tb.nloc;
  float (&vp)[NLOC][3] = in.vp;
  float (&Ro)[9] = in.Ro;
  float (&to)[3] = in.to;
  float (&tp)[3] = in.tp;
  float (&tori)[9] = in.tori;
  float (&dpos)[3] = in.dpos;
  float (&dori)[9] = in.dori;
#pragma unroll
  for (int i = 0; i < NLOC; ++i) {
#pragma unroll
    for (int c = 0; c < 3; ++c) vp[i][c] = 0.f;
    if (i < nloc) {   // (columns past a patch's last vertex hold zeros: zero rows of the blend matrix)
#pragma unroll
      for (int c = 0; c < 3; ++c) vp[i][c] = ot[(size_t)(col + i * 3 + c) * FR + lane];
    }
  }
  const int w = tc / a.F;
  {
    const float* pr = a.offset_r + ((size_t)w * 12 + m) * 9;
    const float* pt = a.offset_t + ((size_t)w * 12 + m) * 3;
#pragma unroll
    for (int k = 0; k < 9; ++k) Ro[k] = pr[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) to[k] = pt[k];
  }
  const bool cot = BWD && a.cot_pos != nullptr;
  const int slot_m = a.used_slot[m];
  float scale = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) { tp[k] = 0.f; dpos[k] = 0.f; }
#pragma unroll
  for (int k = 0; k < 9; ++k) { tori[k] = 0.f; dori[k] = 0.f; }
  if (BWD) {
    if (cot) {
      const float* cp = a.cot_pos + ((size_t)tc * 12 + m) * 3;
      const float* co = a.cot_ori + ((size_t)tc * 12 + m) * 9;
#pragma unroll
      for (int k = 0; k < 3; ++k) dpos[k] = cp[k];
#pragma unroll
      for (int k = 0; k < 9; ++k) dori[k] = co[k];
    } else if (slot_m >= 0) {
      if (a.tgt_t) {
        const float* pt = a.tgt_t + (size_t)(tc >> 6) * (12 * a.n_markers) * FR + (tc & 63);
#pragma unroll
        for (int k = 0; k < 3; ++k) tp[k] = pt[(size_t)(slot_m * 3 + k) * FR];
#pragma unroll
        for (int k = 0; k < 9; ++k) tori[k] = pt[(size_t)(a.n_markers * 3 + slot_m * 9 + k) * FR];
      } else {
        const float* p3 = a.tgt + (size_t)tc * a.ld_tgt + slot_m * 3;
        const float* p9 = a.tgt + (size_t)tc * a.ld_tgt + a.n_markers * 3 + slot_m * 9;
#pragma unroll
        for (int k = 0; k < 3; ++k) tp[k] = p3[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) tori[k] = p9[k];
      }
      scale = a.frame_scale[tc];
    }
  }
  in.scale = scale;
}

// One sensor for the wave's 64 frames.  BWD: also the cotangents (residual of the targets, or external ones).
// NLOC / NBL: local vertices (centre + ring) and bones the code is unrolled for; a patch with fewer has zero weights
// there, so nothing in the arithmetic is conditional -- the compiler is free to batch every load of the patch.
template <bool BWD, int NLOC, int NBL>
__device__ __forceinline__ void tile_sensor(const TileArgs& a, const TileTables& tb, int m, int lane, int t,
                                            int wave, int rnd, bool valid, const SensorIn<NLOC>& in,
                                            float* __restrict__ dot, const float* sG, float* sM) {
  using namespace tl;
  constexpr int NR = NLOC - 1;
  const TileSensor& S = tb.s[m];
  const int deg = S.deg, hk = S.helper, col = S.col, nb = S.nb, nloc = tb.nloc;
  const bool cot = BWD && a.cot_pos != nullptr;
  const int slot_m = a.used_slot[m];
  const float (&vp)[NLOC][3] = in.vp;
  const float (&Ro)[9] = in.Ro;
  const float (&to)[3] = in.to;
  const float (&tp)[3] = in.tp;
  const float (&tori)[9] = in.tori;
  const float scale = in.scale;
  float dpos[3], dori[9];
#pragma unroll
  for (int k = 0; k < 3; ++k) dpos[k] = in.dpos[k];
#pragma unroll
  for (int k = 0; k < 9; ++k) dori[k] = in.dori[k];
  // the patch's dense weights: scalar registers for the whole sensor
  float W[NLOC][NBL];
#pragma unroll
  for (int i = 0; i < NLOC; ++i)
#pragma unroll
    for (int q = 0; q < NBL; ++q) W[i][q] = S.w[i][q];
  TL_SSTAMP(14)
  // ---- linear blend skinning: T_i = sum_q w_iq [R | A^t]_q (blended transform, as the reference), v_i = T_i (v_p; 1).
  // The patch's bone transforms come from LDS for this block only (unused slots: bone 0, weight 0); the reverse pass
  // reads them again rather than keep 72 registers alive across the whole sensor (two waves share a SIMD's file).
  float v[NLOC][3];
  {
    float G[NBL][12];
#pragma unroll
    for (int q = 0; q < NBL; ++q) {
      const float* g = sG + (size_t)S.bone[q] * GW * FR + lane;
#pragma unroll
      for (int e = 0; e < 9; ++e) G[q][e] = g[e * FR];
#pragma unroll
      for (int e = 0; e < 3; ++e) G[q][9 + e] = g[(12 + e) * FR];
    }
#pragma unroll
    for (int i = 0; i < NLOC; ++i) {
      float T[12];
#pragma unroll
      for (int e = 0; e < 12; ++e) T[e] = W[i][0] * G[0][e];
#pragma unroll
      for (int q = 1; q < NBL; ++q)
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] += W[i][q] * G[q][e];
#pragma unroll
      for (int r = 0; r < 3; ++r)
        v[i][r] = T[r * 3 + 0] * vp[i][0] + T[r * 3 + 1] * vp[i][1] + T[r * 3 + 2] * vp[i][2] + T[9 + r];
    }
  }
  TL_SSTAMP(15)
  // ---- vertex normal of the centre: sum of the un-normalised face normals / degree
  float n[3] = {0.f, 0.f, 0.f};
  auto edges = [&](int k, float (&e1)[3], float (&e2)[3]) {   // face k = (centre, ring k, ring k + 1), k < deg (static k)
    const bool wrap = k + 1 == deg;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float nxt = (k + 2 < NLOC) ? (wrap ? v[1][c] : v[k + 2 < NLOC ? k + 2 : 1][c]) : v[1][c];
      e1[c] = v[1 + k][c] - v[0][c];
      e2[c] = nxt - v[0][c];
    }
  };
#pragma unroll
  for (int k = 0; k < NR; ++k)
    if (k < deg) {
      float e1[3], e2[3], fn[3];
      edges(k, e1, e2);
      cross3(e1, e2, fn);
      n[0] += fn[0]; n[1] += fn[1]; n[2] += fn[2];
    }
  // 1 / |x| as the hardware's reciprocal square root (1 ulp) plus one Newton step: the IEEE sqrt + divide sequences are
  // ~30 dependent instructions each, four of them in a row, on a wave that has nothing else to issue meanwhile
  auto inv_norm = [](const float* x) -> float {
    const float q = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    const float y = __builtin_amdgcn_rsqf(q);
    return y * (1.5f - 0.5f * q * y * y);
  };
  const float inv_deg = 1.f / (float)deg;
  n[0] *= inv_deg; n[1] *= inv_deg; n[2] *= inv_deg;
  const float inv_nn = inv_norm(n);
  const float nh[3] = {n[0] * inv_nn, n[1] * inv_nn, n[2] * inv_nn};
  float vh[3] = {v[1][0], v[1][1], v[1][2]};
#pragma unroll
  for (int k = 1; k < NR; ++k)
    if (hk == k) { vh[0] = v[1 + k][0]; vh[1] = v[1 + k][1]; vh[2] = v[1 + k][2]; }
  const float e[3] = {vh[0] - v[0][0], vh[1] - v[0][1], vh[2] - v[0][2]};
  const float inv_ne = inv_norm(e);
  const float sv[3] = {e[0] * inv_ne, e[1] * inv_ne, e[2] * inv_ne};
  float bb[3];
  cross3(nh, sv, bb);
  const float inv_nb = inv_norm(bb);
  const float tv[3] = {bb[0] * inv_nb, bb[1] * inv_nb, bb[2] * inv_nb};
  float aa[3];
  cross3(tv, nh, aa);
  const float inv_na = inv_norm(aa);
  const float s2[3] = {aa[0] * inv_na, aa[1] * inv_na, aa[2] * inv_na};
  const float Rm[9] = {s2[0], tv[0], nh[0], s2[1], tv[1], nh[1], s2[2], tv[2], nh[2]};   // columns (s', t, n)
  float ori[9], pos[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
      ori[r * 3 + cc] = Rm[r * 3 + 0] * Ro[cc] + Rm[r * 3 + 1] * Ro[3 + cc] + Rm[r * 3 + 2] * Ro[6 + cc];
    pos[r] = v[0][r] + (Rm[r * 3 + 0] * to[0] + Rm[r * 3 + 1] * to[1] + Rm[r * 3 + 2] * to[2]);
  }
  if (valid) {
    if (a.pos) {
      float* po = a.pos + ((size_t)t * 12 + m) * 3;
      float* oo = a.ori + ((size_t)t * 12 + m) * 9;
      po[0] = pos[0]; po[1] = pos[1]; po[2] = pos[2];
#pragma unroll
      for (int k = 0; k < 9; ++k) oo[k] = ori[k];
    }
    if (a.pos2) {
      float* po = a.pos2 + ((size_t)t * 12 + m) * 3;
      float* oo = a.ori2 + ((size_t)t * 12 + m) * 9;
      po[0] = pos[0]; po[1] = pos[1]; po[2] = pos[2];
#pragma unroll
      for (int k = 0; k < 9; ++k) oo[k] = ori[k];
    }
  }
  TL_SSTAMP(16)
  if (!BWD) return;
  // ---- cotangents of pos / ori: the residual's (reference loss.py:23-41 times the frame weight) or the caller's
  if (!cot) {
    const bool live = slot_m >= 0 && scale != 0.f;
    const float r0 = pos[0] - tp[0], r1 = pos[1] - tp[1], r2 = pos[2] - tp[2];
    const float sp = live ? scale / sqrtf(r0 * r0 + r1 * r1 + r2 * r2) : 0.f;
    dpos[0] = live ? r0 * sp : 0.f; dpos[1] = live ? r1 * sp : 0.f; dpos[2] = live ? r2 * sp : 0.f;
    float qq = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { dori[k] = ori[k] - tori[k]; qq += dori[k] * dori[k]; }
    const float so = live ? scale / sqrtf(qq) : 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) dori[k] = live ? dori[k] * so : 0.f;
  }
  // dR_m = dori Ro^T + dpos (x) to, then back through the Gram-Schmidt frame
  float dRm[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
      dRm[r * 3 + cc] = dori[r * 3 + 0] * Ro[cc * 3 + 0] + dori[r * 3 + 1] * Ro[cc * 3 + 1] +
                        dori[r * 3 + 2] * Ro[cc * 3 + 2] + dpos[r] * to[cc];
  float ds2[3] = {dRm[0], dRm[3], dRm[6]};
  float dt[3] = {dRm[1], dRm[4], dRm[7]};
  float dnh[3] = {dRm[2], dRm[5], dRm[8]};
  float da[3], tmp[3];
  unit_bwd(ds2, s2, inv_na, da);   // a = t x nh
  cross3(nh, da, tmp); dt[0] += tmp[0]; dt[1] += tmp[1]; dt[2] += tmp[2];
  cross3(da, tv, tmp); dnh[0] += tmp[0]; dnh[1] += tmp[1]; dnh[2] += tmp[2];
  float db[3];
  unit_bwd(dt, tv, inv_nb, db);    // b = nh x s
  cross3(sv, db, tmp); dnh[0] += tmp[0]; dnh[1] += tmp[1]; dnh[2] += tmp[2];
  float dsv[3];
  cross3(db, nh, dsv);
  float de[3];
  unit_bwd(dsv, sv, inv_ne, de);
  float dn[3];
  unit_bwd(dnh, nh, inv_nn, dn);
  const float dfn[3] = {dn[0] * inv_deg, dn[1] * inv_deg, dn[2] * inv_deg};   // the same for every face
  // ---- cotangents of the patch's skinned vertices
  float dv[NLOC][3];
#pragma unroll
  for (int i = 0; i < NLOC; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) dv[i][c] = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) dv[0][c] = dpos[c] - de[c];
#pragma unroll
  for (int k = 0; k < NR; ++k)
    if (hk == k) { dv[1 + k][0] += de[0]; dv[1 + k][1] += de[1]; dv[1 + k][2] += de[2]; }
#pragma unroll
  for (int k = 0; k < NR; ++k)
    if (k < deg) {
      float e1[3], e2[3], d1[3], d2[3];
      edges(k, e1, e2);
      cross3(e2, dfn, d1);         // d e1
      cross3(dfn, e1, d2);         // d e2
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dv[0][c] -= d1[c] + d2[c];
        dv[1 + k][c] += d1[c];
      }
      if (k + 1 == deg || k + 2 >= NLOC) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dv[1][c] += d2[c];
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) dv[k + 2 < NLOC ? k + 2 : 1][c] += d2[c];
      }
    }
  TL_SSTAMP(17)
  // ---- d v_posed = T^R^T dv (to the transposed GEMM)
  {
  float G[NBL][9];
#pragma unroll
  for (int q = 0; q < NBL; ++q) {
    const float* g = sG + (size_t)S.bone[q] * GW * FR + lane;
#pragma unroll
    for (int e = 0; e < 9; ++e) G[q][e] = g[e * FR];
  }
#pragma unroll
  for (int i = 0; i < NLOC; ++i) {
    float TR[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) TR[e] = W[i][0] * G[0][e];
#pragma unroll
    for (int q = 1; q < NBL; ++q)
#pragma unroll
      for (int e = 0; e < 9; ++e) TR[e] += W[i][q] * G[q][e];
    if (i < nloc) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        dot[(size_t)(col + i * 3 + c) * FR + lane] = TR[0 + c] * dv[i][0] + TR[3 + c] * dv[i][1] + TR[6 + c] * dv[i][2];
    }
  }
  }
  // ---- per bone of the patch: force sum_i w_iq dv_i and rest-space moment sum_i w_iq dv_i (x) v_p,i ...
  float acc[NBL][12];
#pragma unroll
  for (int q = 0; q < NBL; ++q) {
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[q][e] = 0.f;
#pragma unroll
    for (int i = 0; i < NLOC; ++i) {
      const float d0 = W[i][q] * dv[i][0], d1 = W[i][q] * dv[i][1], d2 = W[i][q] * dv[i][2];
      acc[q][0] += d0 * vp[i][0]; acc[q][1] += d0 * vp[i][1]; acc[q][2] += d0 * vp[i][2];
      acc[q][3] += d1 * vp[i][0]; acc[q][4] += d1 * vp[i][1]; acc[q][5] += d1 * vp[i][2];
      acc[q][6] += d2 * vp[i][0]; acc[q][7] += d2 * vp[i][1]; acc[q][8] += d2 * vp[i][2];
      acc[q][9] += d0; acc[q][10] += d1; acc[q][11] += d2;
    }
  }
  // ... added to the tile's sums in LDS with plain read-add-write, the sensors IN INDEX ORDER: a sensor waits for its
  // turn (a counter in LDS; the waves of a workgroup share a CU and take their sensors in increasing order, so the wait
  // cannot deadlock), adds its <= NBL x 12 sums and passes the turn on.  Only this short update is serial -- the sensors
  // themselves all run concurrently whatever bones they share -- and the order of the additions is fixed, so the sums
  // are reproducible (LDS float atomics would not be, and measured ~400 cycles each here).
  int* turn = reinterpret_cast<int*>(sM + LDS_M);
  while (__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != m) __builtin_amdgcn_s_sleep(2);
#pragma unroll
  for (int q = 0; q < NBL; ++q)
    if (q < nb) {   // (an unused slot names bone 0 and holds zeros)
      float* mq = sM + (size_t)S.bone[q] * 12 * FR + lane;
#pragma unroll
      for (int e = 0; e < 12; ++e) mq[e * FR] += acc[q][e];
    }
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS updates are done before the turn moves on
  if (lane == 0) __hip_atomic_store(turn, m + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace

// NW waves per tile: the forward-only launch runs 8 (two per SIMD cover each other's latencies; 176 VGPRs), the launch
// with the reverse pass 4 (its sensors need ~380 registers: with 8 waves they spill and the launch is slower).
template <bool BWD, int NLOC, int NBL, int NW>
__global__ __launch_bounds__(64 * NW) void smpl_tile_kernel(TileArgs a) {
  using namespace tl;
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sG = lds;
  float* sM = lds + LDS_G;
  const TileTables& tb = *a.tab;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x;
  const int t = tile * FR + lane;
  const bool valid = t < a.T;
  const int tc = valid ? t : a.T - 1;
  const float* __restrict__ ot = a.out_t + (size_t)tile * tb.ncp2 * FR;
  float* __restrict__ dot = BWD ? a.d_out_t + (size_t)tile * tb.ncp2 * FR : nullptr;
  const float* oj = ot + (size_t)tb.j_off2 * FR + lane;   // rest joint j, coordinate c: oj[(j * 3 + c) * FR]
  TL_STAMP(0)

  // ---- the bone sums start at zero; columns of d_out no sensor or joint writes must be finite (the transposed GEMM
  // multiplies them with zero rows)
  if (BWD) {
    for (int i = threadIdx.x; i < LDS_M; i += NT) sM[i] = 0.f;
    if (threadIdx.x == 0) *reinterpret_cast<int*>(sM + LDS_M) = 0;   // the sensors' turn counter
    for (int c = 12 * tb.nloc * 3 + wave; c < tb.j_off2; c += NW) dot[(size_t)c * FR + lane] = 0.f;
    for (int c = tb.j_off2 + NB * 3 + wave; c < tb.ncp2; c += NW) dot[(size_t)c * FR + lane] = 0.f;
  }
  // ---- Rodrigues + forward chain G_j = G_p [R_j | J_j - J_p] on the SMPL body tree (reference configuration.py:118;
  // the host checks the model's parents against it).  A wave walks whole root-to-leaf paths, parents first, with the
  // running transform in registers -- no table look-up from memory, no LDS read, no barrier inside.  The path is a
  // packed constant (5 bits per joint); the loop body exists once (the whole kernel has to stay well inside the 64 KB
  // instruction cache two CUs share: unrolled per path and joint it ran at ~10 cycles per instruction).  Spine joints
  // 0, 3, 6, 9 are computed by the three waves whose paths run through them (same inputs, same instructions, same
  // bits) and stored by the first.
  {
    //                      step:  0   1   2   3   4   5   6   7   8        (5 bits each)
    // wave 0 (left arm)    joint: 0   3   6   9  13  16  18  20           own: all
    // wave 1 (right arm)          0   3   6   9  14  17  19  21           own: from step 4
    // wave 2 (head)               0   3   6   9  12  15                   own: from step 4
    // wave 3 (left leg)           0   1   4   7  10                       own: from step 1
    // wave 4 (right leg)          0   2   5   8  11                       own: from step 1       (waves 5.. : nothing)
    auto pack = [](int j0, int j1, int j2, int j3, int j4, int j5, int j6, int j7, int j8) -> unsigned long long {
      return (unsigned long long)j0 | ((unsigned long long)j1 << 5) | ((unsigned long long)j2 << 10) |
             ((unsigned long long)j3 << 15) | ((unsigned long long)j4 << 20) | ((unsigned long long)j5 << 25) |
             ((unsigned long long)j6 << 30) | ((unsigned long long)j7 << 35) | ((unsigned long long)j8 << 40);
    };
    // (with four waves the two legs share wave 3: 0 1 4 7 10 2 5 8 11, step 5 restarting from the root)
    const unsigned long long path = wave == 0 ? pack(0, 3, 6, 9, 13, 16, 18, 20, 0)
                                  : wave == 1 ? pack(0, 3, 6, 9, 14, 17, 19, 21, 0)
                                  : wave == 2 ? pack(0, 3, 6, 9, 12, 15, 0, 0, 0)
                                  : NW == 4 ? pack(0, 1, 4, 7, 10, 2, 5, 8, 11)
                                  : wave == 3 ? pack(0, 1, 4, 7, 10, 0, 0, 0, 0) : pack(0, 2, 5, 8, 11, 0, 0, 0, 0);
    const int len = wave < 2 ? 8 : wave == 2 ? 6 : NW == 4 ? 9 : wave < 5 ? 5 : 0;
    const int own_from = wave == 0 ? 0 : wave < 3 ? 4 : 1;
    const int restart = NW == 4 ? (wave == 3 ? 5 : -1) : -1;
    auto joint_at = [&](int k) -> int { return (int)((path >> (5 * k)) & 31ull); };
    const float* tht = a.theta_t ? a.theta_t + (size_t)tile * 66 * FR + lane : nullptr;
    auto fetch = [&](int j, float (&th)[3], float (&Jr)[3]) {
      if (tht) {   // tile layout: one coalesced 256-byte read per value (a strided row read costs 64 cache-line look-ups)
#pragma unroll
        for (int c = 0; c < 3; ++c) th[c] = tht[(size_t)(j * 3 + c) * FR];
      } else {
        const float* pth = a.theta + (size_t)tc * a.ld_theta + j * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) th[c] = pth[c];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) Jr[c] = oj[(size_t)(j * 3 + c) * FR];
    };
    float th[3], Jr[3], thn[3], Jn[3];
    float G[12], Groot[12], Jp[3], Jroot[3];
    if (len > 0) fetch(0, th, Jr);
#pragma unroll 1
    for (int k = 0; k < len; ++k) {
      const int j = joint_at(k);
      if (k + 1 < len) fetch(joint_at(k + 1), thn, Jn);   // the next joint's inputs fly while this one is computed
      Rod q; float R[9];
      rodrigues_fast(th[0], th[1], th[2], a.rod_conv, q, R);
      if (k == 0) {
#pragma unroll
        for (int e = 0; e < 9; ++e) G[e] = R[e];
#pragma unroll
        for (int c = 0; c < 3; ++c) G[9 + c] = Jr[c];
#pragma unroll
        for (int e = 0; e < 12; ++e) Groot[e] = G[e];
#pragma unroll
        for (int c = 0; c < 3; ++c) Jroot[c] = Jr[c];
      } else {
        if (k == restart) {
#pragma unroll
          for (int e = 0; e < 12; ++e) G[e] = Groot[e];
#pragma unroll
          for (int c = 0; c < 3; ++c) Jp[c] = Jroot[c];
        }
        const float d0 = Jr[0] - Jp[0], d1 = Jr[1] - Jp[1], d2 = Jr[2] - Jp[2];
        float Gn[12];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          Gn[9 + r] = G[r * 3 + 0] * d0 + G[r * 3 + 1] * d1 + G[r * 3 + 2] * d2 + G[9 + r];
#pragma unroll
          for (int c = 0; c < 3; ++c)
            Gn[r * 3 + c] = G[r * 3 + 0] * R[c] + G[r * 3 + 1] * R[3 + c] + G[r * 3 + 2] * R[6 + c];
        }
#pragma unroll
        for (int e = 0; e < 12; ++e) G[e] = Gn[e];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) Jp[c] = Jr[c];
      if (k >= own_from) {
        float* g = sG + (size_t)j * GW * FR + lane;
#pragma unroll
        for (int e = 0; e < 12; ++e) g[e * FR] = G[e];
#pragma unroll
        for (int r = 0; r < 3; ++r)
          g[(12 + r) * FR] = G[9 + r] - (G[r * 3 + 0] * Jr[0] + G[r * 3 + 1] * Jr[1] + G[r * 3 + 2] * Jr[2]);
        if (valid) {
          if (a.joints) { float* o = a.joints + (size_t)t * 66 + j * 3; o[0] = G[9]; o[1] = G[10]; o[2] = G[11]; }
          if (a.joints2) { float* o = a.joints2 + (size_t)t * 66 + j * 3; o[0] = G[9]; o[1] = G[10]; o[2] = G[11]; }
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) { th[c] = thn[c]; Jr[c] = Jn[c]; }
    }
  }
  TL_STAMP(1)
  __syncthreads();   // the chain's transforms are complete
  TL_STAMP(3)
  // ---- sensors: wave w takes sensors w, w + NW, ... (no barrier in between: the bone sums are ordered by the turn
  // counter).  With one wave per SIMD the loop runs one step ahead: step r issues the global loads of sensor r + 1 and
  // then computes sensor r; with two waves per SIMD they cover each other's latencies and registers are what is short.
  if (NW <= 4) {
    SensorIn<NLOC> cur, nxt;
#pragma unroll 1
    for (int r = -1; r * NW + wave < 12; ++r) {
      const int m = r >= 0 ? r * NW + wave : -1, mn = (r + 1) * NW + wave < 12 ? (r + 1) * NW + wave : -1;
      if (r >= 0) { TL_STAMP(4 + 2 * r) }
      if (mn >= 0) tile_sensor_load<BWD, NLOC>(a, tb, mn, lane, tc, ot, nxt);
      if (m >= 0) tile_sensor<BWD, NLOC, NBL>(a, tb, m, lane, t, wave, r, valid, cur, dot, sG, sM);
      if (mn >= 0) cur = nxt;
      if (r >= 0) { TL_STAMP(5 + 2 * r) }
    }
  } else {
#pragma unroll 1
    for (int r = 0; r * NW + wave < 12; ++r) {
      const int m = r * NW + wave;
      TL_STAMP(4 + 2 * r)
      SensorIn<NLOC> in;
      tile_sensor_load<BWD, NLOC>(a, tb, m, lane, tc, ot, in);
      tile_sensor<BWD, NLOC, NBL>(a, tb, m, lane, t, wave, r, valid, in, dot, sG, sM);
      TL_STAMP(5 + 2 * r)
    }
  }
  if (BWD) __syncthreads();   // every sensor has added its sums
  TL_STAMP(28)
  if (!BWD) return;
  // ---- reverse chain on the same static tree.  Per bone b the sensors left N_b = sum w dv (x) v_p and F_b = sum w dv;
  // the world-space moment about the origin is M_b = N_b R_b^T + F_b (x) A^t_b.  With subtree sums (children first):
  //   X_j = sum_sub M_b - Fs_j (x) t_j,   dR_j = G_p^T X_j G_j,   dJ_j = (G_p - G_j)^T Fs_j
  // Each wave climbs its limbs with the running subtree sums in registers and leaves the sums of the limb roots
  // (13, 14, 12, 1, 2) in LDS; after ONE barrier wave 0 finishes the spine 9, 6, 3, 0.
  auto rev = [&](int j, int p, float (&Ms)[12]) {   // Ms in: sums over the subtrees of j's children; out: over j's subtree
    const float* g = sG + (size_t)j * GW * FR + lane;
    const float* mj = sM + (size_t)j * 12 * FR + lane;
    float Gj[15], N[12];
#pragma unroll
    for (int e = 0; e < 15; ++e) Gj[e] = g[e * FR];
#pragma unroll
    for (int e = 0; e < 12; ++e) N[e] = mj[e * FR];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        Ms[i * 3 + c] += N[i * 3 + 0] * Gj[c * 3 + 0] + N[i * 3 + 1] * Gj[c * 3 + 1] + N[i * 3 + 2] * Gj[c * 3 + 2] +
                         N[9 + i] * Gj[12 + c];
      Ms[9 + i] += N[9 + i];
    }
    float X[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c) X[i * 3 + c] = Ms[i * 3 + c] - Ms[9 + i] * Gj[9 + c];
    float XG[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c) XG[i * 3 + c] = X[i * 3 + 0] * Gj[0 + c] + X[i * 3 + 1] * Gj[3 + c] + X[i * 3 + 2] * Gj[6 + c];
    float P[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    if (p >= 0) {
      const float* gp = sG + (size_t)p * GW * FR + lane;
#pragma unroll
      for (int e = 0; e < 9; ++e) P[e] = gp[e * FR];
    }
    float* dr = a.d_rot_t + ((size_t)tile * (NB * 9) + j * 9) * FR + lane;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        dr[(r * 3 + c) * FR] = P[0 + r] * XG[0 + c] + P[3 + r] * XG[3 + c] + P[6 + r] * XG[6 + c];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      dot[(size_t)(tb.j_off2 + j * 3 + c) * FR + lane] =
          (P[0 + c] - Gj[0 + c]) * Ms[9] + (P[3 + c] - Gj[3 + c]) * Ms[10] + (P[6 + c] - Gj[6 + c]) * Ms[11];
  };
  auto put = [&](int j, const float (&Ms)[12]) {
    float* mj = sM + (size_t)j * 12 * FR + lane;
#pragma unroll
    for (int e = 0; e < 12; ++e) mj[e * FR] = Ms[e];
  };
  auto add = [&](int j, float (&Ms)[12]) {
    const float* mj = sM + (size_t)j * 12 * FR + lane;
#pragma unroll
    for (int e = 0; e < 12; ++e) Ms[e] += mj[e * FR];
  };
  {
    // limbs, bottom-up (5 bits per joint, the limb root last; 31 ends the list):
    //   wave 0: 20 18 16 13 | wave 1: 21 19 17 14 | wave 2: 15 12 | wave 3: 10 7 4 1 | wave 4: 11 8 5 2
    auto pack = [](int j0, int j1, int j2, int j3, int j4, int j5, int j6) -> unsigned long long {
      return (unsigned long long)j0 | ((unsigned long long)j1 << 5) | ((unsigned long long)j2 << 10) |
             ((unsigned long long)j3 << 15) | ((unsigned long long)j4 << 20) | ((unsigned long long)j5 << 25) |
             ((unsigned long long)j6 << 30);
    };
    // (with four waves: wave 2: 15 12, then 10 7 4 1 | wave 3: 11 8 5 2)
    const unsigned long long limb = wave == 0 ? pack(20, 18, 16, 13, 31, 31, 31) : wave == 1 ? pack(21, 19, 17, 14, 31, 31, 31)
                                  : wave == 2 ? (NW == 4 ? pack(15, 12, 10, 7, 4, 1, 31) : pack(15, 12, 31, 31, 31, 31, 31))
                                  : wave == 3 ? (NW == 4 ? pack(11, 8, 5, 2, 31, 31, 31) : pack(10, 7, 4, 1, 31, 31, 31))
                                  : wave == 4 ? pack(11, 8, 5, 2, 31, 31, 31) : pack(31, 31, 31, 31, 31, 31, 31);
    const unsigned long long par = wave == 0 ? pack(18, 16, 13, 9, 0, 0, 0) : wave == 1 ? pack(19, 17, 14, 9, 0, 0, 0)
                                 : wave == 2 ? (NW == 4 ? pack(12, 9, 7, 4, 1, 0, 0) : pack(12, 9, 0, 0, 0, 0, 0))
                                 : wave == 3 ? (NW == 4 ? pack(8, 5, 2, 0, 0, 0, 0) : pack(7, 4, 1, 0, 0, 0, 0))
                                 : pack(8, 5, 2, 0, 0, 0, 0);
    float Ms[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) Ms[e] = 0.f;
#pragma unroll 1
    for (int k = 0; k < 7; ++k) {
      const int j = (int)((limb >> (5 * k)) & 31ull), p = (int)((par >> (5 * k)) & 31ull);
      if (j == 31) break;
      rev(j, p, Ms);
      if (p == 9 || p == 0) {   // a limb root: its subtree sums go to the spine pass; a second limb starts from zero
        put(j, Ms);
#pragma unroll
        for (int e = 0; e < 12; ++e) Ms[e] = 0.f;
      }
    }
  }
  __syncthreads();
  if (wave == 0) {   // the spine: 9 (children 12, 13, 14), 6, 3, 0 (children 1, 2, 3)
    float Ms[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) Ms[e] = 0.f;
    add(12, Ms); add(13, Ms); add(14, Ms);
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      const int j = 9 - 3 * k, p = j - 3;
      if (j == 0) { add(1, Ms); add(2, Ms); }
      rev(j, p, Ms);
    }
  }
  TL_STAMP(29)
}

// ---------------------------------------------------------------------------------------------------------------
// d R -> d theta and the gradient features, one frame per lane: d_rot and d_feat come in tile layout, the 76 outputs
// of a frame leave through LDS as contiguous row pieces (the caller's rows have a stride of ~300 floats).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rodrigues_bwd_t_kernel(RodBwdTArgs a) {
  __shared__ float sg[TL_FR * ROD_BWD_LD];
  const int tile = blockIdx.x;
  const float* df_t = a.d_feat_t + (size_t)tile * a.ld_feat_t * TL_FR + (threadIdx.x & 63);
  rodrigues_bwd_tile(a, tile, sg, [&](int col) { return df_t[(size_t)col * TL_FR]; });   // feat_rows.h
}

template <bool BWD, int NLOC, int NBL>
static hipError_t launch_tile_cfg(const TileArgs& a, hipStream_t stream) {
#ifndef TL_BWD_WAVES
#define TL_BWD_WAVES 4
#endif
  constexpr int NWAVES = BWD ? TL_BWD_WAVES : 8;
  if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(smpl_tile_kernel<BWD, NLOC, NBL, NWAVES>), tl::LDS_BYTES))
    return e;
  const int tiles = (a.T + TL_FR - 1) / TL_FR;
  hipLaunchKernelGGL((smpl_tile_kernel<BWD, NLOC, NBL, NWAVES>), dim3(tiles), dim3(64 * NWAVES), tl::LDS_BYTES, stream, a);
  return hipGetLastError();
}

// `nloc`, `nbl`: the model's largest patch (TileTables::nloc / nbl, host copies).  Two unrolled sizes: the common
// degree-6 patch over at most 6 bones, and the largest the tables admit.
hipError_t launch_smpl_tile(const TileArgs& a, bool backward, int nloc, int nbl, hipStream_t stream) {
  const bool small = nloc <= 7 && nbl <= 6;
  if (backward) return small ? launch_tile_cfg<true, 7, 6>(a, stream) : launch_tile_cfg<true, TL_NLOC, TL_NBL>(a, stream);
  return small ? launch_tile_cfg<false, 7, 6>(a, stream) : launch_tile_cfg<false, TL_NLOC, TL_NBL>(a, stream);
}

// Rows [T][ld] (the first `cols` columns) -> tile layout [tiles][cols][64], through LDS so that both sides are
// coalesced (a thread-per-element copy is coalesced on one side only: 25 -> 58 us inside the pack kernel).
__global__ __launch_bounds__(256) void rows_to_tile_kernel(const float* __restrict__ src, int ld, int cols,
                                                           float* __restrict__ dst, int T) {
  extern __shared__ float sx[];   // [64][cols + 1]
  const int tile = blockIdx.x, t0 = tile * TL_FR, ldx = cols + 1;
  for (int i = threadIdx.x; i < TL_FR * cols; i += 256) {
    const int f = i / cols, c = i - f * cols;
    const int t = t0 + f < T ? t0 + f : T - 1;
    sx[f * ldx + c] = src[(size_t)t * ld + c];
  }
  __syncthreads();
  float* d = dst + (size_t)tile * cols * TL_FR;
  for (int i = threadIdx.x; i < TL_FR * cols; i += 256) {
    const int c = i >> 6, f = i & 63;
    d[i] = sx[f * ldx + c];
  }
}

hipError_t launch_rows_to_tile(const float* src, int ld, int cols, float* dst_t, int T, hipStream_t stream) {
  const int tiles = (T + TL_FR - 1) / TL_FR;
  hipLaunchKernelGGL(rows_to_tile_kernel, dim3(tiles), dim3(256), (size_t)TL_FR * (cols + 1) * sizeof(float), stream, src, ld,
                     cols, dst_t, T);
  return hipGetLastError();
}

hipError_t launch_rodrigues_bwd_t(const RodBwdTArgs& a, hipStream_t stream) {
  const int tiles = (a.T + TL_FR - 1) / TL_FR;
  hipLaunchKernelGGL(rodrigues_bwd_t_kernel, dim3(tiles), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace empose

#ifdef EMPOSE_CHAIN_TRACE
extern "C" int empose_debug_tile_trace(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(empose::g_tile_trace), sizeof(long long) * 8 * 32);
}
#endif
