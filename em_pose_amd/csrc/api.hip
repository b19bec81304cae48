// C ABI (include/empose_hip.h): model packing, workspace carving and the launch sequence of the LGD loop.
#include "../../include/empose_hip.h"
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace empose;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                            \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) return fail(EMPOSE_EHIP, "%s: %s", #expr, hipGetErrorString(e_));       \
  } while (0)

// ---- optional per-launch timing (HIP events on the launch stream), used by bench.py for the roofline numbers ---------
enum ProfTag { P_PACK = 0, P_LSTM_STEP, P_HEADS, P_UPDATE_FEAT, P_BLEND_GEMM, P_CHAIN, P_BLEND_T_GEMM,
               P_ROD_BWD, P_MLP_IN, P_MLP_HIDDEN, P_MLP_OUT, P_MLP_FUSED, P_INIT_MLP, P_COPY, P_EVENT_PAIR, P_END, P_NTAGS };
const char* const kProfNames[P_NTAGS] = {"pack_inputs", "lstm_step", "init_heads_gemm",
                                         "update_feat", "blend_gemm", "chain_sensors", "blend_T_gemm",
                                         "rodrigues_bwd", "mlp_in_gemm", "mlp_hidden_gemm", "mlp_out_gemm",
                                         "mlp_fused", "init_mlp_gemm", "copies", "event_pair", "end"};
struct Profiler {
  bool on = false;
  int only = -1;            // >= 0: bracket launches of this tag only (and one empty event pair per forward: P_EVENT_PAIR)
  std::vector<hipEvent_t> ev;
  std::vector<int> tags;
  size_t used = 0;
};
Profiler g_prof;

void prof_mark(int tag, hipStream_t stream) {
  if (!g_prof.on) return;
  // Single-kernel mode: events go around the launches of ONE tag (start, then P_END right after the launch), so the
  // rest of the step runs unperturbed; P_EVENT_PAIR / P_END pairs with nothing in between measure what the two event
  // packets themselves cost (the caller subtracts it).
  if (g_prof.only >= 0 && tag != g_prof.only && tag != P_EVENT_PAIR &&
      !(tag == P_END && !g_prof.tags.empty() && (g_prof.tags.back() == g_prof.only || g_prof.tags.back() == P_EVENT_PAIR)))
    return;
  if (g_prof.used == g_prof.ev.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    g_prof.ev.push_back(e);
  }
  (void)hipEventRecord(g_prof.ev[g_prof.used++], stream);
  g_prof.tags.push_back(tag);
}

// A packed Linear(+BN)(+PReLU): device weight and the per-column epilogue (scale, shift).
struct Dense {
  int in_dim = 0, out_dim = 0;
  float* w = nullptr;      // [out][in]
  float* wp = nullptr;     // the same weights in MFMA fragment order (mlp_fused.hip), only for MLP layers
  float* wp3 = nullptr;    // ... and as three bf16 pieces per weight in bf16-MFMA fragment order (mlp_fused_x3.hip)
  float* scale = nullptr;  // nullptr => 1
  float* shift = nullptr;  // bias (and folded BN)
  int act = 0;
  float slope = 0.f;
};

struct Mlp {
  int n_layers = 0, skip = 0;
  Dense layers[EMPOSE_MAX_DENSE];
};

struct Lstm {   // unit u = layer * dirs + direction
  int num_layers = 0, input_size = 0, H = 0, dirs = 1;
  float* w_ih[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  float* w_hh[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  float* bias[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // b_ih + b_hh
  // the same matrices as three bf16 pieces per weight in the fragment order of lstm_x3.hip (uni-directional stacks only)
  unsigned short* w3_ih[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  unsigned short* w3_hh[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // ... and in the fragment order of lstm_mid_x3.hip (8-unit blocks, the four gates of a unit in one 32-column tile)
  unsigned short* w3m_ih[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  unsigned short* w3m_hh[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // ... and in the order of lstm_mid16_x3.hip (4-unit blocks, k-steps of 32)
  unsigned short* w3q_ih[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  unsigned short* w3q_hh[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

}  // namespace

struct empose_model {
  std::vector<void*> allocs;
  SmplTables tab;
  float* wc_frag = nullptr;    // tab.wc / tab.wct in MFMA fragment order (row-block GEMM)
  float* wct_frag = nullptr;
  int n_markers = 12;
  int marker_idx[12];
  int used_slot[12];
  int N = 4;
  float step = 0.1f;
  int shape_avg = 1, use_gradient = 1, rnn_init = 1;
  int d_in = 144, d_x = 296;
  Lstm rnn;
  Dense pose_head, shape_head;
  float* heads_frag = nullptr;   // both heads stacked ([66 + 10][H]) in fragment order, and their stacked bias
  float* heads_frag3 = nullptr;  // ... as three bf16 pieces per weight (x3_rows_layer)
  float* heads_bias = nullptr;
  Mlp pose_init, shape_init, pose_iter, shape_iter;
  int hidden_max = 0;
  int any_skip = 0;
  int smpl_only = 0;
  int rod_conv = 0;            // EMPOSE_RODRIGUES_*
  // frame-per-lane path (smpl_tile.hip): tables, the blend matrix with per-patch vertex copies in fragment order
  int tile_ok = 0, ncp2 = 0, tile_nloc = 0, tile_nbl = 0;
  TileTables* tile_tab = nullptr;
  float* wc2_frag = nullptr;   // [ncp2][200]
  float* wc2t_frag = nullptr;  // [200][ncp2]
  float* wc2_frag3 = nullptr;  // the same two as three bf16 pieces per weight (x3_rows_layer)
  float* wc2t_frag3 = nullptr;
};

struct empose_rnn {
  std::vector<void*> allocs;
  Lstm rnn;
};

struct empose_mesh {
  std::vector<void*> allocs;
  int V = 0, j_off = 0, ncp = 0, kb = 0;
  int n_joints = 22;            // posed joints returned (22 body, or all 52 of SMPL-H)
  int rod_conv = 0;             // EMPOSE_RODRIGUES_*
  float* wc = nullptr;
  float* wc_frag = nullptr;     // vertex rows of wc in matrix-core fragment order, per 32-vertex tile (mesh.hip)
  int* skin_idx = nullptr;
  float* skin_w = nullptr;
  int* skin_idx4 = nullptr;     // first four bones / weights per vertex, padded to whole tiles
  float* skin_w4 = nullptr;
  int* parents = nullptr;
  unsigned short* wc_bf16 = nullptr;   // split-bf16 pieces of wc in fragment order (only when the handle asked for them)
  unsigned short* wc_x3 = nullptr;     // three bf16 pieces of wc in fragment order (mesh_x3.hip), kb <= 4
  unsigned short* skin_bf16 = nullptr; // dense skin weights per 32-vertex tile, bf16 hi + lo, B-fragment order (ditto)
};

namespace {

template <typename T>
int upload(std::vector<void*>& allocs, const T* host, size_t count, T** dev) {
  *dev = nullptr;
  if (count == 0) return EMPOSE_OK;
  if (!host) return fail(EMPOSE_EINVAL, "null host pointer in model descriptor");
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, count * sizeof(T)));
  allocs.push_back(p);
  HIP_TRY(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
  *dev = static_cast<T*>(p);
  return EMPOSE_OK;
}

#define TRY(expr)            \
  do {                       \
    int rc_ = (expr);        \
    if (rc_ != EMPOSE_OK) return rc_; \
  } while (0)

// Packs every index/weight table the chain kernel needs into one array of 32-bit words (staged into LDS per block).
int build_chain_blob(const empose_smpl_desc& s, std::vector<uint32_t>& blob, ChainTabs& off, int* n_chunks) {
  auto put_i = [&](const std::vector<int>& v) { int o = (int)blob.size(); for (int x : v) blob.push_back((uint32_t)x); return o; };
  auto put_f = [&](const float* p, size_t n) {
    int o = (int)blob.size();
    for (size_t i = 0; i < n; ++i) { uint32_t u; std::memcpy(&u, p + i, 4); blob.push_back(u); }
    return o;
  };
  std::vector<int> path_mask(22, 0), sub_mask(22, 0), parents(s.parents, s.parents + 22);
  for (int j = 0; j < 22; ++j) {
    for (int q = s.path_ptr[j]; q < s.path_ptr[j + 1]; ++q) path_mask[j] |= 1 << s.path[q];
    for (int q = s.sub_ptr[j]; q < s.sub_ptr[j + 1]; ++q) sub_mask[j] |= 1 << s.sub[q];
  }
  off.path_mask = put_i(path_mask);
  off.sub_mask = put_i(sub_mask);
  off.parents = put_i(parents);
  {
    // depth-first pre-order: every subtree is a contiguous range, so a subtree sum is a difference of prefix sums
    std::vector<int> pos(22, 0), size(22, 0), stack, order;
    stack.push_back(0);
    while (!stack.empty()) {
      const int j = stack.back();
      stack.pop_back();
      pos[j] = (int)order.size();
      order.push_back(j);
      for (int c = 21; c >= 1; --c)
        if (parents[c] == j) stack.push_back(c);
    }
    for (int j = 0; j < 22; ++j) size[j] = __builtin_popcount((unsigned)sub_mask[j]);
    off.dfs_pos = put_i(pos);
    off.sub_size = put_i(size);
  }
  off.skin_idx = put_i(std::vector<int>(s.skin_idx, s.skin_idx + (size_t)s.nv * s.kb));
  off.skin_w = put_f(s.skin_w, (size_t)s.nv * s.kb);
  // per-bone (vertex, weight) lists cut into chunks of CHAIN_CHUNK pairs, each chunk padded with (vertex 0, weight 0)
  std::vector<int> cb, cbeg, bcp(23, 0), pv;
  std::vector<float> pw;
  for (int b = 0; b < 22; ++b) {
    bcp[b] = (int)cb.size();
    for (int q = s.bone_ptr[b]; q < s.bone_ptr[b + 1]; q += CHAIN_CHUNK) {
      cb.push_back(b);
      cbeg.push_back((int)pv.size());
      for (int k = 0; k < CHAIN_CHUNK; ++k) {
        const bool in = q + k < s.bone_ptr[b + 1];
        pv.push_back(in ? s.bone_vert[q + k] : 0);
        pw.push_back(in ? s.bone_w[q + k] : 0.f);
      }
    }
  }
  bcp[22] = (int)cb.size();
  *n_chunks = (int)cb.size();
  off.chunk_bone = put_i(cb); off.chunk_beg = put_i(cbeg);
  off.bone_chunk_ptr = put_i(bcp);
  off.bone_vert = put_i(pv);
  off.bone_w = put_f(pw.data(), pw.size());
  off.s_center = put_i(std::vector<int>(s.s_center, s.s_center + 12));
  off.s_helper = put_i(std::vector<int>(s.s_helper, s.s_helper + 12));
  off.s_deg = put_i(std::vector<int>(s.s_deg, s.s_deg + 12));
  {
    std::vector<int> faces(s.s_faces, s.s_faces + (size_t)12 * s.max_deg * 3);
    for (int m = 0; m < 12; ++m)
      for (int k = s.s_deg[m]; k < s.max_deg; ++k)
        for (int c = 0; c < 3; ++c) faces[((size_t)m * s.max_deg + k) * 3 + c] = s.s_center[m];
    off.s_faces = put_i(faces);
  }
  // Incidence lists of P4d as packed words (see the kernel): offsets are relative to the frame's LDS record.
  const ChainLds lay = chain_layout(s.nv, s.ncp, s.max_deg, *n_chunks);
  if (lay.total >= (1 << 13)) return fail(EMPOSE_EINVAL, "sensor sub-mesh too large for the packed incidence words");
  if (s.max_deg > 64) return fail(EMPOSE_EINVAL, "more than 64 faces around a sensor vertex");
  auto pack = [](int a, int b, int use_b, int neg) {
    return (int)((uint32_t)a | ((uint32_t)b << 13) | ((uint32_t)use_b << 26) | ((uint32_t)neg << 27));
  };
  std::vector<int> inc_ptr(s.nv + 1, 0), inc_code;
  for (int v = 0; v < s.nv; ++v) {
    inc_ptr[v] = (int)inc_code.size();
    for (int m = 0; m < 12; ++m) {
      if (s.s_center[m] == v) { const int o = lay.scr + m * 9 + 3; inc_code.push_back(pack(o, o, 0, 0)); }
      if (s.s_helper[m] == v) { const int o = lay.scr + m * 9 + 6; inc_code.push_back(pack(o, o, 0, 0)); }
      for (int k = 0; k < s.s_deg[m]; ++k)
        for (int c = 0; c < 3; ++c)
          if (s.s_faces[((size_t)m * s.max_deg + k) * 3 + c] == v) {
            const int fg = lay.fg + (m * s.max_deg + k) * 6;
            if (c == 0) inc_code.push_back(pack(fg, fg + 3, 1, 1));       // v0: -(d e1 + d e2)
            else if (c == 1) inc_code.push_back(pack(fg, fg, 0, 0));      // v1: + d e1
            else inc_code.push_back(pack(fg + 3, fg + 3, 0, 0));          // v2: + d e2
          }
    }
    while ((inc_code.size() - (size_t)inc_ptr[v]) % 4 != 0) inc_code.push_back((int)(1u << 28));   // null
  }
  inc_ptr[s.nv] = (int)inc_code.size();
  off.inc_ptr = put_i(inc_ptr);
  off.inc_code = put_i(inc_code);
  while (blob.size() % 4 != 0) blob.push_back(0u);   // staged into LDS in 16-byte pieces
  off.total = (int)blob.size();
  return EMPOSE_OK;
}

int pack_dense(std::vector<void*>& allocs, const empose_dense_desc& d, Dense* out) {
  if (d.in_dim <= 0 || d.out_dim <= 0 || !d.weight) return fail(EMPOSE_EINVAL, "dense layer: bad dims / null weight");
  if (d.in_dim % 4 != 0) return fail(EMPOSE_EINVAL, "dense layer: in_dim %d must be a multiple of 4", d.in_dim);
  out->in_dim = d.in_dim;
  out->out_dim = d.out_dim;
  TRY(upload(allocs, d.weight, (size_t)d.in_dim * d.out_dim, &out->w));
  std::vector<float> shift(d.out_dim, 0.f), scale;
  if (d.bn_weight) {
    if (!d.bn_bias || !d.bn_mean || !d.bn_var) return fail(EMPOSE_EINVAL, "dense layer: incomplete batch norm");
    scale.resize(d.out_dim);
    for (int n = 0; n < d.out_dim; ++n) {
      const double s = (double)d.bn_weight[n] / std::sqrt((double)d.bn_var[n] + (double)d.bn_eps);
      const double b = d.bias ? (double)d.bias[n] : 0.0;
      scale[n] = (float)s;
      shift[n] = (float)((b - (double)d.bn_mean[n]) * s + (double)d.bn_bias[n]);
    }
    TRY(upload(allocs, scale.data(), scale.size(), &out->scale));
  } else if (d.bias) {
    for (int n = 0; n < d.out_dim; ++n) shift[n] = d.bias[n];
  }
  TRY(upload(allocs, shift.data(), shift.size(), &out->shift));
  out->act = d.has_prelu ? 1 : 0;
  out->slope = d.prelu;
  return EMPOSE_OK;
}

// Weights in the order the matrix cores consume them (mlp_fused.hip): for every k-group of 8 and every 32-column tile,
// lane (n = lane & 31, half = lane >> 5) owns W[tile * 32 + n][kg * 8 + half * 4 .. + 3]; columns / k past the matrix
// are zero, so a wave's fragment is one coalesced 1 KB read and ragged K needs no masking on this operand.
int pack_fragments_raw(std::vector<void*>& allocs, const float* weight, int N, int K, float** out) {
  const int KG = (K + 7) / 8, NT = (N + 31) / 32;
  const int KG4 = (KG + 3) & ~3;   // the kernels walk four k-groups per iteration
  std::vector<float> buf((size_t)KG4 * NT * 256, 0.f);
  for (int kg = 0; kg < KG; ++kg)
    for (int nt = 0; nt < NT; ++nt)
      for (int lane = 0; lane < 64; ++lane) {
        const int n = nt * 32 + (lane & 31);
        if (n >= N) continue;
        for (int e = 0; e < 4; ++e) {
          const int k = kg * 8 + (lane >> 5) * 4 + e;
          if (k < K) buf[(((size_t)kg * NT + nt) * 64 + lane) * 4 + e] = weight[(size_t)n * K + k];
        }
      }
  return upload(allocs, buf.data(), buf.size(), out);
}

// The same weights as THREE bf16 pieces each (w = h + m + l, every piece the round-to-nearest bf16 of what the previous
// ones leave: 8 + 8 + 8 mantissa bits, all of an fp32's 24) in the order v_mfma_f32_32x32x16_bf16 consumes them
// (mlp_fused_x3.hip): for every k-step of 16, every 32-column tile and every piece one 1 KB wave fragment -- lane
// (n = lane & 31, half = lane >> 5) owns piece[tile * 32 + n][ks * 16 + half * 8 .. + 7]; k-steps padded with zeros to a
// multiple of four (the kernel walks quads).  6 bytes per weight.
static unsigned short bf16_round(float x) {
  unsigned u;
  std::memcpy(&u, &x, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);   // inf / nan: as they are
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf16_value(unsigned short h) {
  const unsigned u = (unsigned)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
int pack_fragments_x3_raw(std::vector<void*>& allocs, const float* weight, int N, int K, float** out) {
  const int KS = (K + 15) / 16, NT = (N + 31) / 32;
  const int KS4 = (KS + 3) & ~3;
  std::vector<unsigned short> buf((size_t)KS4 * NT * 3 * 512, 0);
  for (int ks = 0; ks < KS; ++ks)
    for (int nt = 0; nt < NT; ++nt)
      for (int lane = 0; lane < 64; ++lane) {
        const int n = nt * 32 + (lane & 31);
        if (n >= N) continue;
        for (int e = 0; e < 8; ++e) {
          const int k = ks * 16 + (lane >> 5) * 8 + e;
          if (k >= K) continue;
          const float w = weight[(size_t)n * K + k];
          const unsigned short h = bf16_round(w);
          const float r = w - bf16_value(h);
          const unsigned short m = bf16_round(r);
          const unsigned short l = bf16_round(r - bf16_value(m));
          const size_t at = (((size_t)ks * NT + nt) * 3) * 512 + (size_t)lane * 8 + e;
          buf[at] = h; buf[at + 512] = m; buf[at + 1024] = l;
        }
      }
  static_assert(sizeof(float) == 2 * sizeof(unsigned short), "");
  std::vector<float> as_f(buf.size() / 2);
  std::memcpy(as_f.data(), buf.data(), buf.size() * 2);
  return upload(allocs, as_f.data(), as_f.size(), out);
}

int pack_fragments(std::vector<void*>& allocs, const empose_dense_desc& d, Dense* out) {
  TRY(pack_fragments_x3_raw(allocs, d.weight, d.out_dim, d.in_dim, &out->wp3));
  return pack_fragments_raw(allocs, d.weight, d.out_dim, d.in_dim, &out->wp);
}

int pack_mlp(std::vector<void*>& allocs, const empose_mlp_desc& d, Mlp* out, int* hidden_max, int* any_skip) {
  out->n_layers = d.n_layers;
  out->skip = d.skip;
  if (d.n_layers == 0) return EMPOSE_OK;
  if (d.skip) *any_skip = 1;
  if (d.n_layers < 2 || d.n_layers > EMPOSE_MAX_DENSE || (d.n_layers % 2) != 0)
    return fail(EMPOSE_EINVAL, "mlp: n_layers=%d unsupported", d.n_layers);
  for (int i = 0; i < d.n_layers; ++i) {
    TRY(pack_dense(allocs, d.layers[i], &out->layers[i]));
    TRY(pack_fragments(allocs, d.layers[i], &out->layers[i]));
    if (i > 0 && d.layers[i].in_dim != d.layers[i - 1].out_dim) return fail(EMPOSE_EINVAL, "mlp: layer dims do not chain");
    if (i + 1 < d.n_layers && d.layers[i].out_dim > *hidden_max) *hidden_max = d.layers[i].out_dim;
  }
  return EMPOSE_OK;
}

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  float* f(size_t count) {
    float* r = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += align_up(count * sizeof(float));
    return r;
  }
};

// ---- workspace layouts ------------------------------------------------------------------------------------------
struct SmplWs {
  float *rot, *feat, *out, *d_out, *d_feat, *d_rot;
  float* theta_t;   // theta in tile layout for the frame-per-lane kernel
  float* tgt_t;     // targets in tile layout (stand-alone entry points; the LGD loop has its own copy)
};
constexpr int D_FEAT_T_COLS = 224;   // the 200 feature cotangents in tile layout, whole 32-column tiles
SmplWs carve_smpl(Carver& c, const empose_model* m, int T) {
  // Either path fits: row-major [T][cols] for chain_sensors_kernel, tile layout [ceil(T / 64)][cols][64] for
  // smpl_tile_kernel (which does not use `rot`: it evaluates Rodrigues itself).
  SmplWs w;
  const size_t Tp = (size_t)(T + TL_FR - 1) / TL_FR * TL_FR;
  const size_t ncp = m->tab.ncp > m->ncp2 ? m->tab.ncp : m->ncp2;
  w.rot = c.f((size_t)T * 198);
  w.feat = c.f((size_t)T * 200);
  w.out = c.f(Tp * ncp);
  w.d_out = c.f(Tp * ncp);
  w.d_feat = c.f(Tp * D_FEAT_T_COLS);
  w.d_rot = c.f(Tp * 198);
  w.theta_t = c.f(Tp * 66);
  w.tgt_t = c.f(Tp * 144);
  return w;
}
// The frame-per-lane path pays once its 64-frame workgroups fill the 256 CUs (one per CU, 152 KB of LDS each): from
// 16384 frames on.  Measured at 8192 frames (the training step at 256 windows): 2 % slower than the general kernel.
// Option "smpl_tile": 0 never, 1 by size, 2 always (tests).
bool use_tile_path(const empose_model* m, int T, const float* cot_joints = nullptr) {
  const int opt = options().smpl_tile;
  return m->tile_ok && opt != 0 && !cot_joints && (opt == 2 || T >= 16384);
}

struct UpdWs {
  float* buf[3];  // [2 nets][T][hidden_max] each; buf[2] only when a net uses skip connections
};
UpdWs carve_upd(Carver& c, const empose_model* m, int T) {
  UpdWs w;
  w.buf[0] = c.f((size_t)2 * T * m->hidden_max);
  w.buf[1] = c.f((size_t)2 * T * m->hidden_max);
  w.buf[2] = m->any_skip ? c.f((size_t)2 * T * m->hidden_max) : nullptr;
  return w;
}

struct LstmWs {
  float* h[8][2];
  float* c[8];
  float* yb[2];   // [B][F][2H] ping-pong between the layers of a bidirectional stack
  float* xch;     // exchange words of the whole-sequence small-batch kernel (lstm_persist_kernel), or nullptr
  float* h3[8];   // third hidden-state buffer per unit + counters of the whole-sequence large-batch kernel, or nullptr
  unsigned* seq_cnt;
  // lstm_x3.hip: the stored input of every time step and the hidden states (ping-pong) as bf16 piece planes, or nullptr
  unsigned short* x3; size_t x3_t_stride;
  unsigned short* a3[8][2];
  // lstm_midseq_x3.hip: [F + 1] sets of hidden-state planes per layer and the progress counters, or nullptr
  unsigned short* xa[8];
  unsigned* midseq_flags;
};
// bf16 elements of one set of A planes: [32-row tiles][k-steps][3 pieces][512]
size_t lstm_x3_plane_elems(int B, int K) { return (size_t)((B + 31) / 32) * ((K + 15) / 16) * 3 * 512; }
bool lstm_x3_covers(const Lstm& r, int B) {
  if (options().lstm_x3 == 0 || r.dirs != 1 || r.num_layers > 4 || B < LSTM_SEQ_MIN_B) return false;
  if (r.H % 32 != 0 || r.input_size % 4 != 0) return false;
  for (int l = 0; l < r.num_layers; ++l)
    if (!r.w3_ih[l] || !r.w3_hh[l]) return false;
  return true;
}
constexpr int LSTM_MID16_MIN_B = 9;
// medium batches (the batched evaluation driver's chunks): lstm_mid_x3.hip / lstm_mid16_x3.hip
bool lstm_x3_mid_covers(const Lstm& r, int B) {
  if (options().lstm_x3 == 0 || options().lstm_mid_x3 == 0 || r.dirs != 1 || r.num_layers > 4) return false;
  // from 9 rows with the 4-unit tiles of lstm_mid16_x3.hip (7.0 us per step against 7.8 - 10.7 of lstm_fewrows_kernel at 9 - 16
  // rows), from 17 with the 8-unit tiles
  const bool tiles16 = options().lstm_mid16 != 0 && lstm_mid16_shape_ok(B, r.H) && r.w3q_ih[0] && r.w3q_hh[0];
  if (B < (tiles16 ? LSTM_MID16_MIN_B : LSTM_PERSIST_B + 1) || B >= LSTM_SEQ_MIN_B) return false;
  if (r.H % 32 != 0 || r.input_size % 4 != 0) return false;
  for (int l = 0; l < r.num_layers; ++l)
    if (!r.w3m_ih[l] || !r.w3m_hh[l]) return false;
  return true;
}
// medium batches, whole sequence in one launch: lstm_midseq_x3.hip (needs the 8-unit-block weight order too)
constexpr int LSTM_MIDSEQ_MIN_B = 4, LSTM_MIDSEQ_MAX_B = 64;   // (up to 3 rows: lstm_persist_kernel)
bool lstm_x3_midseq_covers(const Lstm& r, int B) {
  if (options().lstm_x3 == 0 || options().lstm_mid_x3 == 0 || options().lstm_midseq == 0 || r.dirs != 1) return false;
  if (B < LSTM_MIDSEQ_MIN_B || B > LSTM_MIDSEQ_MAX_B || r.num_layers > 4 || r.input_size % 4 != 0) return false;
  int ks_in[4];
  for (int l = 0; l < r.num_layers; ++l) {
    if (!r.w3m_ih[l] || !r.w3m_hh[l]) return false;
    ks_in[l] = l == 0 ? (r.input_size + 15) / 16 : r.H / 16;
  }
  return lstm_midseq_shape_ok(B, r.H, r.num_layers, ks_in);
}
LstmWs carve_lstm_of(Carver& c, const Lstm& r, int B, int F) {
  LstmWs w;
  const int H = r.H, U = r.num_layers * r.dirs;
  for (int u = 0; u < 8; ++u) {
    const bool used = u < U;
    w.h[u][0] = used ? c.f((size_t)B * H) : nullptr;
    w.h[u][1] = used ? c.f((size_t)B * H) : nullptr;
    w.c[u] = used ? c.f((size_t)B * H) : nullptr;
  }
  const bool need_y = r.dirs == 2 && r.num_layers > 1;
  w.yb[0] = need_y ? c.f((size_t)B * F * 2 * H) : nullptr;
  w.yb[1] = (need_y && r.num_layers > 2) ? c.f((size_t)B * F * 2 * H) : nullptr;
  w.xch = (r.dirs == 1 && B <= LSTM_PERSIST_B) ? c.f(lstm_persist_xch_floats(r.num_layers, B, H)) : nullptr;
  const bool seq = r.dirs == 1 && B >= LSTM_SEQ_MIN_B && r.num_layers <= 4;
  for (int u = 0; u < 8; ++u) w.h3[u] = (seq && u < U) ? c.f((size_t)B * H) : nullptr;
  w.seq_cnt = seq ? reinterpret_cast<unsigned*>(c.f(lstm_seq_counter_uints(B))) : nullptr;
  const bool midseq = lstm_x3_midseq_covers(r, B);
  const bool x3 = lstm_x3_covers(r, B) || lstm_x3_mid_covers(r, B) || midseq;
  w.x3_t_stride = lstm_x3_plane_elems(B, r.input_size);
  w.x3 = x3 ? reinterpret_cast<unsigned short*>(c.f((w.x3_t_stride * F + 1) / 2)) : nullptr;
  for (int u = 0; u < 8; ++u)
    for (int k = 0; k < 2; ++k)
      w.a3[u][k] = (x3 && u < U) ? reinterpret_cast<unsigned short*>(c.f((lstm_x3_plane_elems(B, H) + 1) / 2)) : nullptr;
  for (int u = 0; u < 8; ++u)
    w.xa[u] = (midseq && u < U) ? reinterpret_cast<unsigned short*>(c.f((lstm_x3_plane_elems(B, H) * (size_t)(F + 1) + 1) / 2))
                                : nullptr;
  w.midseq_flags = midseq ? reinterpret_cast<unsigned*>(c.f(lstm_midseq_flag_uints(U, H))) : nullptr;
  return w;
}
LstmWs carve_lstm(Carver& c, const empose_model* m, int B, int F) { return carve_lstm_of(c, m->rnn, B, F); }

GemmProb linear_prob(const float* A, int lda, const Dense& d, float* C, int ldc, int M) {
  GemmProb p;
  p.A = A; p.lda = lda; p.W = d.w; p.ldw = d.in_dim; p.C = C; p.ldc = ldc;
  p.M = M; p.N = d.out_dim; p.K = d.in_dim;
  p.scale = d.scale; p.shift = d.shift; p.resid = nullptr; p.ldr = 0; p.act = d.act; p.slope = d.slope;
  return p;
}

// Runs one or two MLPs that share the input x (the update nets / the init nets) layer by layer, both nets per launch.
// Hidden blocks are layer pairs (1,2), (3,4), ...; with skip connections the block input is added to the block
// output (reference layers.py:35-43), which needs the block input kept alive in a third buffer.
int run_mlps(const Mlp* nets[2], int n_nets, float* outs[2], const int out_ld[2], const float* x, int ldx, int T,
             const UpdWs& ws, int hidden_max, hipStream_t stream, bool init_net = false) {
  const int L = nets[0]->n_layers;
  for (int i = 1; i < n_nets; ++i)
    if (nets[i]->n_layers != L) return fail(EMPOSE_EINVAL, "paired MLPs must have the same depth");

  // Large batches: every layer of both nets in ONE launch (mlp_fused.hip); a workgroup keeps 128 rows through all the
  // layers. Needs enough row panels to fill the chip and layers no wider than the four 128-column waves.
  {
    bool ok = options().mlp_fused != 0 && L <= FUSED_MAX_LAYERS && (long)((T + 63) / 64) * n_nets >= 256;
    for (int i = 0; i < n_nets && ok; ++i) {
      if (nets[i]->skip || nets[i]->layers[0].in_dim > FUSED_MAX_WIDTH) ok = false;   // no room for a block input
      for (int l = 0; l < L; ++l) {
        const Dense& d = nets[i]->layers[l];
        if (d.out_dim > FUSED_MAX_WIDTH || d.act > 1) ok = false;
      }
    }
    if (ok) {
      // fp32 products from three bf16 pieces per operand on the bf16 matrix path (mlp_fused_x3.hip; fp32-equivalent, 2.7
      // times the fp32 instruction's rate): needs every hidden width to be whole quads of k-steps of the next layer
      bool x3 = options().mlp_x3 != 0;
      for (int i = 0; i < n_nets && x3; ++i)
        for (int l = 0; l + 1 < L; ++l)
          if (nets[i]->layers[l].out_dim % 64 != 0 || !nets[i]->layers[l].wp3) x3 = false;
      FusedMlpArgs fa;
      fa.count = n_nets; fa.M = T;
      for (int i = 0; i < n_nets; ++i) {
        FusedNet& fn = fa.net[i];
        fn.x = x; fn.ldx = ldx; fn.out = outs[i]; fn.ld_out = out_ld[i];
        fn.n_layers = L;   // the activations stay in LDS: no scratch
        for (int l = 0; l < L; ++l) {
          const Dense& d = nets[i]->layers[l];
          FusedLayer& fl = fn.layer[l];
          fl.W = x3 ? d.wp3 : d.wp; fl.K = d.in_dim; fl.N = d.out_dim; fl.scale = d.scale; fl.shift = d.shift;
          fl.slope = d.slope; fl.act = d.act;
        }
      }
      prof_mark(init_net ? P_INIT_MLP : P_MLP_FUSED, stream);
      hipError_t e = x3 ? launch_mlp_fused_x3(fa, stream) : launch_mlp_fused(fa, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "fused mlp launch: %s", hipGetErrorString(e));
      prof_mark(P_END, stream);   // close the dominant kernel's interval at its completion, not at the next launch
      return EMPOSE_OK;
    }
  }

  int cur[2] = {-1, -1}, block_in[2] = {-1, -1};
  for (int l = 0; l < L; ++l) {
    GemmBatch b;
    b.count = n_nets;
    int nxt[2] = {-1, -1};
    for (int i = 0; i < n_nets; ++i) {
      const Dense& d = nets[i]->layers[l];
      auto buf = [&](int k) { return ws.buf[k] + (size_t)i * T * hidden_max; };
      const float* in = (l == 0) ? x : buf(cur[i]);
      const int ld_in = (l == 0) ? ldx : nets[i]->layers[l - 1].out_dim;
      const bool block_first = (l >= 1) && (l % 2 == 1) && (l < L - 1);
      const bool block_last = (l >= 2) && (l % 2 == 0) && (l < L - 1);
      if (block_first) block_in[i] = cur[i];
      float* out;
      int ld_out;
      if (l == L - 1) {
        out = outs[i];
        ld_out = out_ld[i];
      } else {
        int k = 0;
        while (k == cur[i] || (nets[i]->skip && k == block_in[i])) ++k;
        if (k > 2 || !ws.buf[k]) return fail(EMPOSE_EINVAL, "internal: MLP scratch buffers exhausted");
        nxt[i] = k;
        out = buf(k);
        ld_out = d.out_dim;
      }
      b.p[i] = linear_prob(in, ld_in, d, out, ld_out, T);
      if (block_last && nets[i]->skip) {
        b.p[i].resid = buf(block_in[i]);
        b.p[i].ldr = d.out_dim;
      }
    }
    b.role = (!init_net && l > 0 && l < L - 1) ? 1 : 0;
    prof_mark(init_net ? P_INIT_MLP : (l == 0 ? P_MLP_IN : (l == L - 1 ? P_MLP_OUT : P_MLP_HIDDEN)), stream);
    hipError_t e = launch_gemm(b, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "gemm launch: %s", hipGetErrorString(e));
    for (int i = 0; i < n_nets; ++i) cur[i] = nxt[i];
  }
  return EMPOSE_OK;
}

void fill_unit(LstmUnitArgs& ua, const Lstm& r, const LstmWs& ws, int u) {
  ua.w_ih = r.w_ih[u]; ua.w_hh = r.w_hh[u]; ua.bias = r.bias[u];
  ua.h[0] = ws.h[u][0]; ua.h[1] = ws.h[u][1]; ua.c = ws.c[u];
  ua.in_seq = nullptr; ua.in_ld = 0; ua.in_from = -1; ua.t_offset = 0; ua.reverse = 0;
  ua.y = nullptr; ua.y_ld = 0; ua.y_col = 0;
}

// A poll of a cooperative kernel launched by an EARLIER call gave up (that call's outputs are NaN): reported by every entry
// point that launches or consumes such kernels -- the recurrences and, round 6, the training layers (empose_mlp_train_*,
// empose_lstm_train_*) -- without synchronising (the counter is a host-mapped word), and STICKY until
// empose_async_status() has reported and cleared it.
int earlier_poll_timeouts() {
  if (const unsigned n = poll_timeouts_peek())
    return fail(EMPOSE_ETIMEOUT, "%u poll(s) of a cooperative kernel (whole-sequence LSTM / one-launch training layer) launched by an earlier call timed out waiting for "
                "another workgroup's exchange word; that call's outputs are NaN (the state it carried too); "
                "empose_async_status() reports and clears this", n);
  return EMPOSE_OK;
}

// State layout of h0/c0/h_n/c_n: [num_layers * dirs][B][H], unit u = layer * dirs + direction (PyTorch's order).
int run_lstm(const Lstm& r, int B, int F, const float* x, int ldx, const int* seq_lengths, const float* h0,
             const float* c0, float* y, float* h_n, float* c_n, const LstmWs& ws, hipStream_t stream) {
  const int H = r.H, L = r.num_layers, D = r.dirs, U = L * D;
  const size_t bh = (size_t)B * H;
  // a poll of a cooperative kernel of an EARLIER call gave up: everything that call (and what was fed from it) produced
  // is NaN.  Reported once, here, without synchronising (the counter is a host-mapped word).
  // STICKY: the count is only looked at here; it stays set -- and every recurrence of the process keeps failing, whichever
  // model, stream or thread it belongs to -- until empose_async_status() has reported and cleared it.  (Clearing it here
  // let the one call that happened to come next swallow the report while the call that produced the NaNs returned OK.)
  TRY(earlier_poll_timeouts());
  // the wavefront kernel addresses its operands with 32-bit byte offsets from a per-segment base
  if ((size_t)B * F * (size_t)(ldx > 2 * H ? ldx : 2 * H) * sizeof(float) >= ((size_t)1 << 32))
    return fail(EMPOSE_EINVAL, "LSTM batch of %d x %d frames is too large for one call; split the batch", B, F);
  prof_mark(P_COPY, stream);
  if (!h0 && !c0) {
    // new sequences: the state buffers of all units are carved back to back (carve_lstm_of) -- one fill instead of 2 U
    const char* lo = reinterpret_cast<const char*>(ws.h[0][0]);
    const char* hi = reinterpret_cast<const char*>(ws.c[U - 1] + bh);
    HIP_TRY(hipMemsetAsync(ws.h[0][0], 0, (size_t)(hi - lo), stream));
  } else {
    for (int u = 0; u < U; ++u) {
      if (h0) HIP_TRY(hipMemcpyAsync(ws.h[u][0], h0 + u * bh, bh * sizeof(float), hipMemcpyDeviceToDevice, stream));
      else HIP_TRY(hipMemsetAsync(ws.h[u][0], 0, bh * sizeof(float), stream));
      if (c0) HIP_TRY(hipMemcpyAsync(ws.c[u], c0 + u * bh, bh * sizeof(float), hipMemcpyDeviceToDevice, stream));
      else HIP_TRY(hipMemsetAsync(ws.c[u], 0, bh * sizeof(float), stream));
    }
  }
  LstmWaveArgs a;
  a.seq_lengths = seq_lengths; a.B = B; a.F = F; a.H = H;
  bool seq_done = false;
  if (D == 1) {
    // Stacked uni-directional layers: wavefront over (layer, time), launch s advances layer l by its step s - l.
    if (L > 4) return fail(EMPOSE_EINVAL, "at most 4 stacked layers per wavefront");
    a.n_units = L;
    for (int l = 0; l < L; ++l) {
      LstmUnitArgs& ua = a.unit[l];
      fill_unit(ua, r, ws, l);
      ua.in_k = (l == 0) ? r.input_size : H;
      if (l == 0) { ua.in_seq = x; ua.in_ld = ldx; }
      else ua.in_from = l - 1;
      ua.t_offset = l;
      if (l == L - 1) { ua.y = y; ua.y_ld = H; }
    }
    // Small batches: the whole sequence in one cooperative launch (weights in registers, grid barrier per step).
    bool done = false;
    if (ws.xch && F >= 4 && options().lstm_persist != 0) {
      prof_mark(P_LSTM_STEP, stream);
      a.s = 0;
      hipError_t e = launch_lstm_persist(a, ws.xch, stream, &done);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "lstm sequence kernel: %s", hipGetErrorString(e));
    }
    // Large batches: the whole sequence in one cooperative launch too (lstm_seq_kernel; the workgroups of a row group
    // synchronise through counters); falls back to the step launches when its workgroups cannot all be resident.
    if (!done && ws.seq_cnt && F >= 4 && options().lstm_seq != 0 && !a.unit[0].sv_gates) {
      prof_mark(P_LSTM_STEP, stream);
      a.s = 0;
      hipError_t e = launch_lstm_seq(a, ws.h3, ws.seq_cnt, stream, &done);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "lstm sequence kernel (large batch): %s", hipGetErrorString(e));
      seq_done = done;
    }
    // Medium batches, inference: the whole sequence in one cooperative launch on three bf16 pieces per operand, weights in
    // registers (lstm_midseq_x3.hip); falls back to the step launches below when it cannot be launched here.
    if (!done && ws.xa[0] && ws.x3 && F >= 4 && !a.unit[0].sv_gates && lstm_x3_midseq_covers(r, B)) {
      prof_mark(P_COPY, stream);
      const int KS_in = (r.input_size + 15) / 16, KS_h = H / 16;
      hipError_t e = launch_lstm_split_rows(x, (long)F * ldx, ldx, F, B, r.input_size, KS_in, ws.x3, (long)ws.x3_t_stride, stream);
      for (int l = 0; l < L && e == hipSuccess; ++l)
        e = launch_lstm_split_rows(ws.h[l][0], H, 0, 1, B, H, KS_h, ws.xa[l], 0, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "lstm operand split: %s", hipGetErrorString(e));
      LstmMidSeqArgs qa;
      qa.n_units = L; qa.seq_lengths = seq_lengths; qa.B = B; qa.F = F; qa.H = H; qa.flags = ws.midseq_flags;
      for (int l = 0; l < 4; ++l) {
        const int ll = l < L ? l : 0;
        LstmMidSeqUnit& qu = qa.unit[l];
        qu.w3_ih = r.w3m_ih[ll]; qu.w3_hh = r.w3m_hh[ll]; qu.bias = r.bias[ll];
        qu.in3 = ws.x3; qu.in_t_stride = ws.x3_t_stride; qu.ks_in = ll == 0 ? KS_in : KS_h;
        qu.xa = ws.xa[ll]; qu.h0 = ws.h[ll][0]; qu.h_last = ws.h[ll][F & 1]; qu.c = ws.c[ll];
        qu.y = ll == L - 1 ? y : nullptr; qu.y_ld = H; qu.y_col = 0;
      }
      prof_mark(P_LSTM_STEP, stream);
      e = launch_lstm_midseq_x3(qa, stream, &done);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "lstm sequence kernel (medium batch): %s", hipGetErrorString(e));
    }
    // Large batches, inference: the steps on the bf16 matrix path with three bf16 pieces per operand (lstm_x3.hip)
    const bool mid3 = lstm_x3_mid_covers(r, B);
    // ... up to 64 rows with half the tile, on all 256 CUs (lstm_mid16_x3.hip)
    const bool mid16 = mid3 && options().lstm_mid16 != 0 && lstm_mid16_shape_ok(B, H) && r.w3q_ih[0] && r.w3q_hh[0];
    if (!done && ws.x3 && !a.unit[0].sv_gates && (lstm_x3_covers(r, B) || mid3)) {
      prof_mark(P_COPY, stream);
      const int KS_in = (r.input_size + 15) / 16, KS_h = H / 16;
      hipError_t e = launch_lstm_split_rows(x, (long)F * ldx, ldx, F, B, r.input_size, KS_in, ws.x3, (long)ws.x3_t_stride, stream);
      for (int l = 0; l < L && e == hipSuccess; ++l) {
        e = launch_lstm_split_rows(ws.h[l][0], H, 0, 1, B, H, KS_h, ws.a3[l][0], 0, stream);
        if (e == hipSuccess) e = hipMemsetAsync(ws.a3[l][1], 0, lstm_x3_plane_elems(B, H) * sizeof(unsigned short), stream);
      }
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "lstm operand split: %s", hipGetErrorString(e));
      const int tiles = (H / 32) * ((B + 63) / 64);
      for (int s = 0; s < F + L - 1; ++s) {
        LstmX3Args xa;
        xa.n_units = 0; xa.seq_lengths = seq_lengths; xa.B = B; xa.F = F; xa.H = H;
        for (int l = 0; l < L; ++l) {
          const int t = s - l;
          if (t < 0 || t >= F) continue;
          LstmX3Unit& xu = xa.unit[xa.n_units++];
          xu.w3_ih = mid16 ? r.w3q_ih[l] : mid3 ? r.w3m_ih[l] : r.w3_ih[l];
          xu.w3_hh = mid16 ? r.w3q_hh[l] : mid3 ? r.w3m_hh[l] : r.w3_hh[l]; xu.bias = r.bias[l];
          xu.a3_in = l == 0 ? ws.x3 + (size_t)t * ws.x3_t_stride : ws.a3[l - 1][(t + 1) & 1];
          xu.ks_in = l == 0 ? KS_in : KS_h;
          xu.a3_rec = ws.a3[l][t & 1]; xu.a3_out = ws.a3[l][(t + 1) & 1];
          xu.h_prev = ws.h[l][t & 1]; xu.h_next = ws.h[l][(t + 1) & 1]; xu.c = ws.c[l];
          xu.y = l == L - 1 ? y : nullptr; xu.y_ld = H; xu.y_col = 0; xu.t = t;
        }
        xa.units_per_block = tiles >= 192 ? xa.n_units : 1;
        prof_mark(P_LSTM_STEP, stream);
        e = mid16 ? launch_lstm_mid16_x3(xa, stream) : mid3 ? launch_lstm_mid_x3(xa, stream)
                 : options().lstm_x3 == 2 ? launch_lstm_rows_x3(xa, stream) : launch_lstm_chain_x3(xa, stream);
        if (e != hipSuccess) return fail(EMPOSE_EHIP, "lstm step (bf16 pieces): %s", hipGetErrorString(e));
      }
      done = true;
    }
    for (int s = 0; !done && s < F + L - 1; ++s) {
      a.s = s;
      prof_mark(P_LSTM_STEP, stream);
      hipError_t e = launch_lstm_wave(a, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "lstm step: %s", hipGetErrorString(e));
    }
  } else {
    // Bidirectional: a layer needs the whole output sequence of the layer below, so layers run one after the other;
    // the two directions of a layer share each launch.
    if (!seq_lengths) return fail(EMPOSE_EINVAL, "bidirectional LSTM needs seq_lengths");
    a.n_units = 2;
    for (int l = 0; l < L; ++l) {
      const float* in = (l == 0) ? x : ws.yb[(l - 1) & 1];
      const int in_ld = (l == 0) ? ldx : 2 * H;
      float* out = (l == L - 1) ? y : ws.yb[l & 1];
      for (int d = 0; d < 2; ++d) {
        LstmUnitArgs& ua = a.unit[d];
        fill_unit(ua, r, ws, l * 2 + d);
        ua.in_k = (l == 0) ? r.input_size : 2 * H;
        ua.in_seq = in; ua.in_ld = in_ld; ua.reverse = d;
        ua.y = out; ua.y_ld = 2 * H; ua.y_col = d * H;
      }
      for (int s = 0; s < F; ++s) {
        a.s = s;
        prof_mark(P_LSTM_STEP, stream);
        hipError_t e = launch_lstm_wave(a, stream);
        if (e != hipSuccess) return fail(EMPOSE_EHIP, "lstm step: %s", hipGetErrorString(e));
      }
    }
  }
  prof_mark(P_COPY, stream);
  for (int u = 0; u < U; ++u) {
    // the final hidden state: buffer F & 1 after step launches, F % 3 of (h[0], h[1], h3) after the large-batch sequence kernel
    const float* h_last = seq_done ? (F % 3 == 2 ? ws.h3[u] : ws.h[u][F % 3]) : ws.h[u][F & 1];
    if (h_n) HIP_TRY(hipMemcpyAsync(h_n + u * bh, h_last, bh * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (c_n) HIP_TRY(hipMemcpyAsync(c_n + u * bh, ws.c[u], bh * sizeof(float), hipMemcpyDeviceToDevice, stream));
  }
  return EMPOSE_OK;
}

// An LSTM weight matrix [4H][K] (gate-major rows) as three bf16 pieces per weight in the fragment order of lstm_x3.hip:
// [k-step of 16][32-unit block][gate][piece] -> one wave fragment of 512 bf16, lane (n = lane & 31, half = lane >> 5) owns
// W[gate * H + block * 32 + n][ks * 16 + half * 8 .. + 7]; k past K is zero.
// `mid`: the order of lstm_mid_x3.hip instead -- [k-step of 16][8-unit block][piece] -> one fragment whose column
// n = lane & 31 is gate n >> 3 of unit block * 8 + (n & 7).
int pack_lstm_x3(std::vector<void*>& allocs, const float* w, int H, int K, unsigned short** out, bool mid = false) {
  const int KS = (K + 15) / 16, JB = mid ? H / 8 : H / 32, NQ = mid ? 1 : 4;
  std::vector<unsigned short> buf((size_t)KS * JB * NQ * 3 * 512, 0);
  for (int ks = 0; ks < KS; ++ks)
    for (int jb = 0; jb < JB; ++jb)
      for (int q = 0; q < NQ; ++q)
        for (int lane = 0; lane < 64; ++lane) {
          const int n = lane & 31;
          const float* row = mid ? w + (size_t)((n >> 3) * H + jb * 8 + (n & 7)) * K
                                 : w + (size_t)(q * H + jb * 32 + n) * K;
          for (int e = 0; e < 8; ++e) {
            const int k = ks * 16 + (lane >> 5) * 8 + e;
            if (k >= K) continue;
            const unsigned short h = bf16_round(row[k]);
            const float r1 = row[k] - bf16_value(h);
            const unsigned short m = bf16_round(r1);
            const unsigned short l = bf16_round(r1 - bf16_value(m));
            const size_t at = ((((size_t)ks * JB + jb) * NQ + q) * 3) * 512 + (size_t)lane * 8 + e;
            buf[at] = h; buf[at + 512] = m; buf[at + 1024] = l;
          }
        }
  std::vector<float> as_f((buf.size() + 1) / 2);
  std::memcpy(as_f.data(), buf.data(), buf.size() * 2);
  float* dev = nullptr;
  TRY(upload(allocs, as_f.data(), as_f.size(), &dev));
  *out = reinterpret_cast<unsigned short*>(dev);
  return EMPOSE_OK;
}

// ... and in the order of lstm_mid16_x3.hip: [k-step of 32][4-unit block][piece] -> one fragment of the 16x16x32 instruction,
// lane (n = lane & 15, q = lane >> 4) owns W[gate (n >> 2) * H + block * 4 + (n & 3)][ks * 32 + q * 8 .. + 7]; k past K is zero.
int pack_lstm_x3_mid16(std::vector<void*>& allocs, const float* w, int H, int K, unsigned short** out) {
  const int K2 = (K + 31) / 32, JB = H / 4;
  std::vector<unsigned short> buf((size_t)K2 * JB * 3 * 512, 0);
  for (int ks = 0; ks < K2; ++ks)
    for (int jb = 0; jb < JB; ++jb)
      for (int lane = 0; lane < 64; ++lane) {
        const int n = lane & 15, q = lane >> 4;
        const float* row = w + (size_t)((n >> 2) * H + jb * 4 + (n & 3)) * K;
        for (int e = 0; e < 8; ++e) {
          const int k = ks * 32 + q * 8 + e;
          if (k >= K) continue;
          const unsigned short h = bf16_round(row[k]);
          const float r1 = row[k] - bf16_value(h);
          const unsigned short m = bf16_round(r1);
          const unsigned short l = bf16_round(r1 - bf16_value(m));
          const size_t at = (((size_t)ks * JB + jb) * 3) * 512 + (size_t)lane * 8 + e;
          buf[at] = h; buf[at + 512] = m; buf[at + 1024] = l;
        }
      }
  std::vector<float> as_f((buf.size() + 1) / 2);
  std::memcpy(as_f.data(), buf.data(), buf.size() * 2);
  float* dev = nullptr;
  TRY(upload(allocs, as_f.data(), as_f.size(), &dev));
  *out = reinterpret_cast<unsigned short*>(dev);
  return EMPOSE_OK;
}

int pack_lstm(std::vector<void*>& allocs, const empose_lstm_desc& r, int dirs, const float* const* w_ih,
              const float* const* w_hh, const float* const* b_ih, const float* const* b_hh, Lstm* out) {
  if (r.num_layers < 1 || r.num_layers * dirs > 8 || r.hidden_size % 4 != 0 || r.input_size % 4 != 0)
    return fail(EMPOSE_EINVAL, "unsupported LSTM configuration");
  out->num_layers = r.num_layers; out->input_size = r.input_size; out->H = r.hidden_size; out->dirs = dirs;
  for (int l = 0; l < r.num_layers; ++l)
    for (int d = 0; d < dirs; ++d) {
      const int u = l * dirs + d;
      const int k_in = l == 0 ? r.input_size : r.hidden_size * dirs;
      if (!w_ih[u] || !w_hh[u] || !b_ih[u] || !b_hh[u]) return fail(EMPOSE_EINVAL, "null LSTM parameter");
      TRY(upload(allocs, w_ih[u], (size_t)4 * r.hidden_size * k_in, &out->w_ih[u]));
      TRY(upload(allocs, w_hh[u], (size_t)4 * r.hidden_size * r.hidden_size, &out->w_hh[u]));
      std::vector<float> bias(4 * r.hidden_size);
      for (int i = 0; i < 4 * r.hidden_size; ++i) bias[i] = b_ih[u][i] + b_hh[u][i];
      TRY(upload(allocs, bias.data(), bias.size(), &out->bias[u]));
      if (dirs == 1 && r.hidden_size % 32 == 0) {
        TRY(pack_lstm_x3(allocs, w_ih[u], r.hidden_size, k_in, &out->w3_ih[u]));
        TRY(pack_lstm_x3(allocs, w_hh[u], r.hidden_size, r.hidden_size, &out->w3_hh[u]));
        TRY(pack_lstm_x3(allocs, w_ih[u], r.hidden_size, k_in, &out->w3m_ih[u], true));
        TRY(pack_lstm_x3(allocs, w_hh[u], r.hidden_size, r.hidden_size, &out->w3m_hh[u], true));
        TRY(pack_lstm_x3_mid16(allocs, w_ih[u], r.hidden_size, k_in, &out->w3q_ih[u]));
        TRY(pack_lstm_x3_mid16(allocs, w_hh[u], r.hidden_size, r.hidden_size, &out->w3q_hh[u]));
      }
    }
  return EMPOSE_OK;
}

// Where the residual gradient of one SMPL evaluation goes (null: no gradient wanted).
struct GradOut {
  float* g_theta; int ld_g; float* g_beta; int ld_gb;
  float* trace_g_theta; float* trace_g_beta;
};
// One SMPL evaluation: pose / shape update + feature row (fa: what to update and where the copies go; rot / feat /
// theta_t are filled in here), blend GEMM, chain + skinning + sensors (+ reverse), transposed GEMM, Rodrigues reverse.
// On the frame-per-lane path the first and the last step ride on the GEMMs (option "smpl_fuse", default on).
int run_smpl_eval(const empose_model* m, int T, int F, const SmplWs& ws, FeatArgs fa, const float* offset_r,
                  const float* offset_t, const float* tgt, int ld_tgt, const float* frame_scale, float* pos, float* ori,
                  float* joints, float* pos2, float* ori2, float* joints2, hipStream_t stream,
                  const float* cot_pos = nullptr, const float* cot_ori = nullptr, const float* cot_joints = nullptr,
                  const float* tgt_t = nullptr, const GradOut* go = nullptr) {
  const bool bwd = tgt || cot_pos;
  const bool tile = use_tile_path(m, T, cot_joints);
  const bool fuse = tile && options().smpl_fuse != 0;
  fa.rot = tile ? nullptr : ws.rot; fa.feat = ws.feat; fa.theta_t = tile ? ws.theta_t : nullptr;
  fa.T = T; fa.F = F; fa.rod_conv = m->rod_conv;
  if (bwd && !go) return fail(EMPOSE_EINVAL, "gradient outputs missing");
  if (!fuse) {
    prof_mark(P_UPDATE_FEAT, stream);
    hipError_t e = launch_update_feat(fa, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "update_feat kernel: %s", hipGetErrorString(e));
  }
  if (tile) {
    // frame-per-lane path: blend GEMM -> tile layout -> smpl_tile_kernel -> tile layout -> transposed GEMM
    prof_mark(P_BLEND_GEMM, stream);
    const bool rx3 = options().rows_x3 != 0 && m->wc2_frag3 && m->wc2t_frag3;
    hipError_t e = fuse ? launch_blend_feat_gemm(fa, rx3 ? m->wc2_frag3 : m->wc2_frag, ws.out, m->ncp2, m->ncp2, rx3, stream)
                        : launch_gemm_rows_t(ws.feat, 200, false, m->wc2_frag, ws.out, m->ncp2, T, m->ncp2, 200, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "blend gemm (tile): %s", hipGetErrorString(e));
    TileArgs a;
    a.tab = m->tile_tab; a.theta = fa.theta; a.ld_theta = fa.ld_theta; a.out_t = ws.out;
    a.theta_t = ws.theta_t; a.tgt_t = tgt ? tgt_t : nullptr;
    a.offset_r = offset_r; a.offset_t = offset_t; a.tgt = tgt; a.ld_tgt = ld_tgt; a.frame_scale = frame_scale;
    a.n_markers = m->n_markers;
    for (int i = 0; i < 12; ++i) a.used_slot[i] = m->used_slot[i];
    a.pos = pos; a.ori = ori; a.joints = joints; a.pos2 = pos2; a.ori2 = ori2; a.joints2 = joints2;
    a.d_out_t = ws.d_out; a.d_rot_t = ws.d_rot; a.T = T; a.F = F; a.rod_conv = m->rod_conv;
    a.cot_pos = cot_pos; a.cot_ori = cot_ori;
    prof_mark(P_CHAIN, stream);
    e = launch_smpl_tile(a, bwd, m->tile_nloc, m->tile_nbl, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "smpl tile kernel: %s", hipGetErrorString(e));
    if (bwd) {
      RodBwdTArgs ra;
      ra.theta = fa.theta; ra.ld_theta = fa.ld_theta; ra.theta_t = ws.theta_t; ra.d_rot_t = ws.d_rot;
      ra.d_feat_t = ws.d_feat; ra.ld_feat_t = D_FEAT_T_COLS;
      ra.g_theta = go->g_theta; ra.ld_g = go->ld_g; ra.g_beta = go->g_beta; ra.ld_gb = go->ld_gb;
      ra.trace_g_theta = go->trace_g_theta; ra.trace_g_beta = go->trace_g_beta;
      ra.T = T; ra.rod_conv = m->rod_conv;
      prof_mark(P_BLEND_T_GEMM, stream);
      if (fuse) {
        e = launch_blend_t_gemm_rod(ws.d_out, m->ncp2, rx3 ? m->wc2t_frag3 : m->wc2t_frag, m->ncp2, ra, rx3, stream);
        if (e != hipSuccess) return fail(EMPOSE_EHIP, "blend^T gemm + rodrigues_bwd (tile): %s", hipGetErrorString(e));
        return EMPOSE_OK;
      }
      e = launch_gemm_rows_t(ws.d_out, m->ncp2, true, m->wc2t_frag, ws.d_feat, D_FEAT_T_COLS, T, 200, m->ncp2, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "blend^T gemm (tile): %s", hipGetErrorString(e));
      prof_mark(P_ROD_BWD, stream);
      e = launch_rodrigues_bwd_t(ra, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "rodrigues_bwd (tile) kernel: %s", hipGetErrorString(e));
    }
    return EMPOSE_OK;
  }
  GemmBatch b;
  b.count = 1;
  GemmProb& p = b.p[0];
  p.A = ws.feat; p.lda = 200; p.W = m->tab.wc; p.ldw = 200; p.C = ws.out; p.ldc = m->tab.ncp;
  p.M = T; p.N = m->tab.ncp; p.K = 200;
  p.scale = nullptr; p.shift = nullptr; p.resid = nullptr; p.ldr = 0; p.act = 0; p.slope = 0.f;
  prof_mark(P_BLEND_GEMM, stream);
  hipError_t e = (m->wc_frag && gemm_rows_applicable(T, m->tab.ncp, 200))
                     ? launch_gemm_rows(ws.feat, 200, m->wc_frag, ws.out, m->tab.ncp, T, m->tab.ncp, 200, stream)
                     : launch_gemm(b, stream);
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "blend gemm: %s", hipGetErrorString(e));
  ChainArgs c;
  c.tab = m->tab;
  c.rot = ws.rot; c.out = ws.out; c.offset_r = offset_r; c.offset_t = offset_t;
  c.tgt = tgt; c.ld_tgt = ld_tgt; c.frame_scale = frame_scale;
  c.n_markers = m->n_markers;
  for (int i = 0; i < 12; ++i) c.used_slot[i] = m->used_slot[i];
  c.pos = pos; c.ori = ori; c.joints = joints; c.pos2 = pos2; c.ori2 = ori2; c.joints2 = joints2;
  c.d_out = ws.d_out; c.d_rot = ws.d_rot; c.T = T; c.F = F;
  c.cot_pos = cot_pos; c.cot_ori = cot_ori; c.cot_joints = cot_joints;
  prof_mark(P_CHAIN, stream);
  e = launch_chain_sensors(c, stream);
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "chain kernel: %s", hipGetErrorString(e));
  if (tgt || cot_pos) {
    p.A = ws.d_out; p.lda = m->tab.ncp; p.W = m->tab.wct; p.ldw = m->tab.ncp; p.C = ws.d_feat; p.ldc = 200;
    p.M = T; p.N = 200; p.K = m->tab.ncp;
    prof_mark(P_BLEND_T_GEMM, stream);
    e = (m->wct_frag && gemm_rows_applicable(T, 200, m->tab.ncp))
            ? launch_gemm_rows(ws.d_out, m->tab.ncp, m->wct_frag, ws.d_feat, 200, T, 200, m->tab.ncp, stream)
            : launch_gemm(b, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "blend^T gemm: %s", hipGetErrorString(e));
    RodBwdArgs ra;
    ra.theta = fa.theta; ra.ld_theta = fa.ld_theta; ra.d_rot = ws.d_rot; ra.d_feat = ws.d_feat;
    ra.g_theta = go->g_theta; ra.ld_g = go->ld_g; ra.g_beta = go->g_beta; ra.ld_gb = go->ld_gb;
    ra.trace_g_theta = go->trace_g_theta; ra.trace_g_beta = go->trace_g_beta; ra.T = T; ra.rod_conv = m->rod_conv;
    prof_mark(P_ROD_BWD, stream);
    e = launch_rodrigues_bwd(ra, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "rodrigues_bwd kernel: %s", hipGetErrorString(e));
  }
  return EMPOSE_OK;
}

}  // namespace

namespace empose {
Options& options() {
  static Options o;
  return o;
}

namespace {
unsigned* g_timeout_host = nullptr;   // host-mapped, device-visible (fine-grained): written by kernels, read here
unsigned* g_timeout_dev = nullptr;
}  // namespace
unsigned* poll_timeout_word() {
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    // (Portable: the word is pinned for every device of the process, not only the one that happens to be current here)
    if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return; }
    std::memset(h, 0, 64);
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return; }
    g_timeout_host = static_cast<unsigned*>(h);
    g_timeout_dev = static_cast<unsigned*>(d);
  });
  return g_timeout_dev;
}
unsigned poll_timeouts_take() {
  if (!g_timeout_host) return 0;
  return __atomic_exchange_n(g_timeout_host, 0u, __ATOMIC_RELAXED);
}
unsigned poll_timeouts_peek() {
  if (!g_timeout_host) return 0;
  return __atomic_load_n(g_timeout_host, __ATOMIC_RELAXED);
}

hipError_t coresident_blocks(const void* fn, int threads, size_t lds_bytes, int* blocks) {
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev)) return e;
  struct Entry { size_t lds = 0; int blocks = -1; };
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, Entry> cache;
  std::lock_guard<std::mutex> lock(mu);
  Entry& c = cache[{dev, fn}];
  if (c.blocks < 0 || lds_bytes > c.lds) {   // (a figure computed for more LDS is a safe one for less)
    if (hipError_t e = allow_dynamic_lds(fn, lds_bytes)) return e;
    int per_cu = 0;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, dev);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds_bytes);
    if (e != hipSuccess) return e;
    c.lds = lds_bytes;
    c.blocks = per_cu * prop.multiProcessorCount;
  }
  *blocks = c.blocks;
  return hipSuccess;
}

hipError_t allow_dynamic_lds(const void* fn, size_t bytes) {
  if (bytes > LDS_BYTES_PER_CU) return hipErrorInvalidValue;
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev)) return e;
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> allowed;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = allowed[{dev, fn}];
  if (bytes <= have) return hipSuccess;
  if (hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)) return e;
  have = bytes;
  return hipSuccess;
}
}  // namespace empose

extern "C" {

const char* empose_last_error(void) { return g_err.c_str(); }

int empose_set_option(const char* name, int value) {
  if (!name) return fail(EMPOSE_EINVAL, "null option name");
  Options& o = options();
  const struct { const char* n; int* v; } tab[] = {
      {"mlp_fused", &o.mlp_fused}, {"lstm_persist", &o.lstm_persist}, {"gemm_splitk", &o.gemm_splitk},
      {"smpl_tile", &o.smpl_tile}, {"smpl_fuse", &o.smpl_fuse}, {"heads_rows", &o.heads_rows}, {"lstm_seq", &o.lstm_seq}, {"bptt_wave", &o.bptt_wave}, {"train_fused", &o.train_fused},
      {"gemm_wide", &o.gemm_wide},
      {"atb_target", &o.atb_target},
      {"atb_chunk", &o.atb_chunk},
      {"spin_limit", &o.spin_limit},
      {"train_epi", &o.train_epi},
      {"mesh_skin_mfma", &o.mesh_skin_mfma},
      {"mlp_x3", &o.mlp_x3},
      {"lstm_x3", &o.lstm_x3},
      {"rows_x3", &o.rows_x3},
      {"train_cols", &o.train_cols},
      {"cols_coop", &o.cols_coop},
      {"mesh_x3", &o.mesh_x3},
      {"lstm_mid_x3", &o.lstm_mid_x3},
      {"lstm_midseq", &o.lstm_midseq},
      {"lstm_mid16", &o.lstm_mid16},
      {"train_x3", &o.train_x3},
      {"lstm_fewrows", &o.lstm_fewrows},
      {"atb_fast", &o.atb_fast}};
  for (const auto& e : tab)
    if (std::strcmp(name, e.n) == 0) { *e.v = value; return EMPOSE_OK; }
  return fail(EMPOSE_EINVAL, "unknown option '%s'", name);
}

size_t empose_pack_weight_x3_bytes(int N, int K) {
  if (N <= 0 || K <= 0) return 0;
  return pack_x3_elems(N, K) * sizeof(unsigned short);
}

int empose_pack_weight_x3(const float* W, int ldw, int N, int K, void* out, empose_stream_t stream_) {
  if (!W || !out) return fail(EMPOSE_EINVAL, "null argument");
  if (N <= 0 || K <= 0 || K % 4 != 0 || ldw < K) return fail(EMPOSE_EINVAL, "bad sizes");
  hipError_t e = launch_pack_x3(W, ldw, N, K, static_cast<unsigned short*>(out), static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "weight pack: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

int empose_async_status(void) {
  if (const unsigned n = poll_timeouts_take())
    return fail(EMPOSE_ETIMEOUT, "%u poll(s) of a cooperative kernel (whole-sequence LSTM / one-launch training layer) timed out waiting for another workgroup's exchange "
                "word; the outputs of that call are NaN", n);
  return EMPOSE_OK;
}

int empose_reset_options(void) {
  options() = Options{};
  return EMPOSE_OK;
}

int empose_get_option(const char* name) {
  if (!name) return -1;
  const Options& o = options();
  const struct { const char* n; int v; } tab[] = {
      {"mlp_fused", o.mlp_fused}, {"lstm_persist", o.lstm_persist}, {"gemm_splitk", o.gemm_splitk},
      {"smpl_tile", o.smpl_tile}, {"smpl_fuse", o.smpl_fuse}, {"heads_rows", o.heads_rows}, {"lstm_seq", o.lstm_seq}, {"bptt_wave", o.bptt_wave}, {"train_fused", o.train_fused},
      {"gemm_wide", o.gemm_wide},
      {"atb_target", o.atb_target},
      {"atb_chunk", o.atb_chunk},
      {"spin_limit", o.spin_limit},
      {"train_epi", o.train_epi},
      {"mesh_skin_mfma", o.mesh_skin_mfma},
      {"mlp_x3", o.mlp_x3},
      {"lstm_x3", o.lstm_x3},
      {"rows_x3", o.rows_x3},
      {"train_cols", o.train_cols},
      {"cols_coop", o.cols_coop},
      {"mesh_x3", o.mesh_x3},
      {"lstm_mid_x3", o.lstm_mid_x3},
      {"lstm_midseq", o.lstm_midseq},
      {"lstm_mid16", o.lstm_mid16},
      {"train_x3", o.train_x3},
      {"lstm_fewrows", o.lstm_fewrows},
      {"atb_fast", o.atb_fast}};
  for (const auto& e : tab)
    if (std::strcmp(name, e.n) == 0) return e.v;
  return -1;
}
int empose_version(void) { return 3; }   // 2: empose_lgd_io gained suppress_missing / mask_value; 3 (round 6): empose_mlp_params gained
                                         // weight_x3 / weight_t_x3, empose_lstm_grads gained d_h0 / d_c0 (appended fields)
const char* empose_arch(void) { return "gfx950"; }

int empose_profile_enable(int on) {
  g_prof.on = on != 0;
  g_prof.only = -1;
  g_prof.used = 0;
  g_prof.tags.clear();
  return EMPOSE_OK;
}

int empose_profile_enable_only(const char* tag_name) {
  if (!tag_name) return fail(EMPOSE_EINVAL, "null tag name");
  for (int i = 0; i < P_NTAGS; ++i)
    if (std::strcmp(tag_name, kProfNames[i]) == 0) {
      g_prof.on = true;
      g_prof.only = i;
      g_prof.used = 0;
      g_prof.tags.clear();
      return EMPOSE_OK;
    }
  return fail(EMPOSE_EINVAL, "unknown profile tag '%s'", tag_name);
}

int empose_profile_ntags(void) { return P_NTAGS; }

const char* empose_profile_tag_name(int tag) { return (tag >= 0 && tag < P_NTAGS) ? kProfNames[tag] : ""; }

const char* empose_profile_gemm_kernel_name(int M, int N, int K, int count, int role) {
  return gemm_kernel_name(M, N, K, count < 1 ? 1 : (count > 2 ? 2 : count), role);
}

int empose_profile_read(double* total_ms, long long* count) {
  if (!total_ms || !count) return fail(EMPOSE_EINVAL, "null argument");
  for (int i = 0; i < P_NTAGS; ++i) { total_ms[i] = 0.0; count[i] = 0; }
  if (g_prof.used == 0) return EMPOSE_OK;
  HIP_TRY(hipEventSynchronize(g_prof.ev[g_prof.used - 1]));
  for (size_t i = 0; i + 1 < g_prof.used; ++i) {
    const int tag = g_prof.tags[i];
    if (tag == P_END) continue;  // gap between two forwards
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]));
    total_ms[tag] += ms;
    count[tag] += 1;
  }
  g_prof.used = 0;
  g_prof.tags.clear();
  return EMPOSE_OK;
}

void empose_model_destroy(empose_model_t* model) {
  if (!model) return;
  for (void* p : model->allocs) (void)hipFree(p);
  delete model;
}

int empose_model_create(const empose_model_desc* d, empose_model_t** out) {
  if (!d || !out) return fail(EMPOSE_EINVAL, "null argument");
  *out = nullptr;
  const empose_smpl_desc& s = d->smpl;
  if (s.n_sensors != EMPOSE_N_SENSORS) return fail(EMPOSE_EINVAL, "n_sensors must be 12");
  if (s.nv <= 0 || s.ncp % 4 != 0 || s.j_off < s.nv * 3 || s.j_off + 66 > s.ncp || s.kb <= 0 || s.max_deg <= 0)
    return fail(EMPOSE_EINVAL, "inconsistent SMPL table sizes");
  if (d->n_markers != 6 && d->n_markers != 12) return fail(EMPOSE_EINVAL, "n_markers must be 6 or 12");
  if (d->n_iterations < 0) return fail(EMPOSE_EINVAL, "n_iterations < 0");
  if (s.rodrigues != EMPOSE_RODRIGUES_SMPLX && s.rodrigues != EMPOSE_RODRIGUES_SO3)
    return fail(EMPOSE_EINVAL, "unknown Rodrigues convention %d", s.rodrigues);
  empose_model* m = new empose_model();
  m->rod_conv = s.rodrigues;
  auto bail = [&](int rc) { empose_model_destroy(m); return rc; };
#define MTRY(expr) do { int rc_ = (expr); if (rc_ != EMPOSE_OK) return bail(rc_); } while (0)
  SmplTables& t = m->tab;
  t.n_sensors = s.n_sensors; t.nv = s.nv; t.j_off = s.j_off; t.ncp = s.ncp; t.kb = s.kb; t.max_deg = s.max_deg;
  float* fp; int* ip;
  MTRY(upload(m->allocs, s.wc, (size_t)s.ncp * 200, &fp)); t.wc = fp;
  MTRY(upload(m->allocs, s.wct, (size_t)s.ncp * 200, &fp)); t.wct = fp;
  MTRY(pack_fragments_raw(m->allocs, s.wc, s.ncp, 200, &m->wc_frag));     // the same two matrices in MFMA fragment order
  MTRY(pack_fragments_raw(m->allocs, s.wct, 200, s.ncp, &m->wct_frag));
  MTRY(upload(m->allocs, s.parents, 22, &ip)); t.parents = ip;
  MTRY(upload(m->allocs, s.skin_idx, (size_t)s.nv * s.kb, &ip)); t.skin_idx = ip;
  MTRY(upload(m->allocs, s.skin_w, (size_t)s.nv * s.kb, &fp)); t.skin_w = fp;
  if (!s.bone_ptr || !s.path_ptr || !s.sub_ptr) return bail(fail(EMPOSE_EINVAL, "null CSR pointer"));
  MTRY(upload(m->allocs, s.bone_ptr, 23, &ip)); t.bone_ptr = ip;
  MTRY(upload(m->allocs, s.bone_vert, (size_t)s.bone_ptr[22], &ip)); t.bone_vert = ip;
  MTRY(upload(m->allocs, s.bone_w, (size_t)s.bone_ptr[22], &fp)); t.bone_w = fp;
  MTRY(upload(m->allocs, s.s_center, 12, &ip)); t.s_center = ip;
  MTRY(upload(m->allocs, s.s_helper, 12, &ip)); t.s_helper = ip;
  MTRY(upload(m->allocs, s.s_deg, 12, &ip)); t.s_deg = ip;
  MTRY(upload(m->allocs, s.s_faces, (size_t)12 * s.max_deg * 3, &ip)); t.s_faces = ip;
  MTRY(upload(m->allocs, s.path_ptr, 23, &ip)); t.path_ptr = ip;
  MTRY(upload(m->allocs, s.path, (size_t)s.path_ptr[22], &ip)); t.path = ip;
  MTRY(upload(m->allocs, s.sub_ptr, 23, &ip)); t.sub_ptr = ip;
  MTRY(upload(m->allocs, s.sub, (size_t)s.sub_ptr[22], &ip)); t.sub = ip;
  {
    for (int i = 0; i < 12; ++i)
      if (s.s_deg[i] < 1 || s.s_deg[i] > s.max_deg) return bail(fail(EMPOSE_EINVAL, "sensor degree out of range"));
    std::vector<uint32_t> blob;
    MTRY(build_chain_blob(s, blob, t.off, &t.n_chunks));
    uint32_t* bp;
    MTRY(upload(m->allocs, blob.data(), blob.size(), &bp));
    t.blob = bp;
  }
  {
    // frame-per-lane path: only for patches that are closed fans of at most TL_NR faces over at most TL_NBL bones
    // (closed manifold meshes; anything else keeps chain_sensors_kernel)
    TileTables tt;
    std::vector<float> wc2;
    if (s.n_sensors == 12 && build_tile_tables(s.nv, s.kb, s.max_deg, s.j_off, s.wc, s.parents, s.skin_idx, s.skin_w,
                                               s.s_center, s.s_helper, s.s_deg, s.s_faces, &tt, &wc2)) {
      std::vector<float> wc2t((size_t)200 * tt.ncp2);
      for (int r = 0; r < tt.ncp2; ++r)
        for (int k = 0; k < 200; ++k) wc2t[(size_t)k * tt.ncp2 + r] = wc2[(size_t)r * 200 + k];
      MTRY(upload(m->allocs, &tt, 1, &m->tile_tab));
      MTRY(pack_fragments_raw(m->allocs, wc2.data(), tt.ncp2, 200, &m->wc2_frag));
      MTRY(pack_fragments_raw(m->allocs, wc2t.data(), 200, tt.ncp2, &m->wc2t_frag));
      MTRY(pack_fragments_x3_raw(m->allocs, wc2.data(), tt.ncp2, 200, &m->wc2_frag3));
      MTRY(pack_fragments_x3_raw(m->allocs, wc2t.data(), 200, tt.ncp2, &m->wc2t_frag3));
      m->ncp2 = tt.ncp2; m->tile_nloc = tt.nloc; m->tile_nbl = tt.nbl;
      m->tile_ok = tt.ncp2 <= 320;   // the widest tile gemm_rows_t_kernel covers
    }
  }

  m->n_markers = d->n_markers;
  for (int i = 0; i < 12; ++i) { m->marker_idx[i] = 0; m->used_slot[i] = -1; }
  for (int i = 0; i < d->n_markers; ++i) {
    const int v = d->marker_idx[i];
    if (v < 0 || v >= 12) return bail(fail(EMPOSE_EINVAL, "marker_idx out of range"));
    m->marker_idx[i] = v;
    m->used_slot[v] = i;
  }
  m->N = d->n_iterations; m->step = d->step_size; m->shape_avg = d->shape_avg; m->use_gradient = d->use_gradient;
  m->rnn_init = d->rnn_init;
  m->d_in = d->n_markers * 12;
  m->d_x = m->d_in + 76 + (d->use_gradient ? 76 : 0);

  if (d->rnn_init) {
    const empose_lstm_desc& r = d->rnn;
    if (r.num_layers > 4 || r.input_size != m->d_in) return bail(fail(EMPOSE_EINVAL, "unsupported LSTM configuration"));
    MTRY(pack_lstm(m->allocs, r, 1, r.w_ih, r.w_hh, r.b_ih, r.b_hh, &m->rnn));
    MTRY(pack_dense(m->allocs, d->pose_head, &m->pose_head));
    MTRY(pack_dense(m->allocs, d->shape_head, &m->shape_head));
    if (d->pose_head.out_dim == 66 && d->shape_head.out_dim == 10 && d->pose_head.in_dim == d->shape_head.in_dim &&
        !d->pose_head.bn_weight && !d->shape_head.bn_weight && !d->pose_head.has_prelu && !d->shape_head.has_prelu) {
      const int K = d->pose_head.in_dim;
      std::vector<float> wst((size_t)76 * K), bst(76, 0.f);
      std::memcpy(wst.data(), d->pose_head.weight, (size_t)66 * K * sizeof(float));
      std::memcpy(wst.data() + (size_t)66 * K, d->shape_head.weight, (size_t)10 * K * sizeof(float));
      for (int n = 0; n < 66; ++n) bst[n] = d->pose_head.bias ? d->pose_head.bias[n] : 0.f;
      for (int n = 0; n < 10; ++n) bst[66 + n] = d->shape_head.bias ? d->shape_head.bias[n] : 0.f;
      MTRY(pack_fragments_raw(m->allocs, wst.data(), 76, K, &m->heads_frag));
      MTRY(pack_fragments_x3_raw(m->allocs, wst.data(), 76, K, &m->heads_frag3));
      MTRY(upload(m->allocs, bst.data(), bst.size(), &m->heads_bias));
    }
    if (m->pose_head.out_dim != 66 || m->shape_head.out_dim != 10 || m->pose_head.in_dim != r.hidden_size)
      return bail(fail(EMPOSE_EINVAL, "init head dims"));
  } else if (d->pose_init.n_layers == 0 && d->n_iterations == 0) {
    // body-model-only handle: serves empose_smpl_sensors_fwd_bwd / _vjp (training path), not empose_lgd_forward
    m->smpl_only = 1;
  } else {
    MTRY(pack_mlp(m->allocs, d->pose_init, &m->pose_init, &m->hidden_max, &m->any_skip));
    MTRY(pack_mlp(m->allocs, d->shape_init, &m->shape_init, &m->hidden_max, &m->any_skip));
    if (m->pose_init.n_layers == 0 || m->pose_init.layers[0].in_dim != m->d_in)
      return bail(fail(EMPOSE_EINVAL, "init MLP dims"));
  }
  if (m->N > 0) {
    MTRY(pack_mlp(m->allocs, d->pose_iter, &m->pose_iter, &m->hidden_max, &m->any_skip));
    MTRY(pack_mlp(m->allocs, d->shape_iter, &m->shape_iter, &m->hidden_max, &m->any_skip));
    if (m->pose_iter.n_layers == 0 || m->pose_iter.layers[0].in_dim != m->d_x ||
        m->pose_iter.layers[m->pose_iter.n_layers - 1].out_dim != 66 ||
        m->shape_iter.layers[m->shape_iter.n_layers - 1].out_dim != 10)
      return bail(fail(EMPOSE_EINVAL, "update MLP dims (expected input %d)", m->d_x));
  }
  if (m->hidden_max == 0) m->hidden_max = 4;
#undef MTRY
  *out = m;
  return EMPOSE_OK;
}

int empose_smpl_tile_supported(const empose_model_t* m) { return m && m->tile_ok ? 1 : 0; }

size_t empose_smpl_workspace_bytes(const empose_model_t* m, int T) {
  Carver c(nullptr);
  carve_smpl(c, m, T);
  return c.off;
}

size_t empose_update_workspace_bytes(const empose_model_t* m, int T) {
  Carver c(nullptr);
  carve_upd(c, m, T);
  return c.off;
}

size_t empose_lstm_workspace_bytes(const empose_model_t* m, int B, int F) {
  Carver c(nullptr);
  carve_lstm(c, m, B, F);
  return c.off;
}

struct LgdWs {
  float *x, *scale, *d_pose, *d_shape, *pos, *ori, *joints;
  float* x_t;   // the sensor columns of x in tile layout (targets of the frame-per-lane kernel)
  SmplWs smpl;
  UpdWs upd;
  LstmWs lstm;
  float* y;
};
static LgdWs carve_lgd(Carver& c, const empose_model* m, int B, int F) {
  LgdWs w;
  const size_t T = (size_t)B * F;
  w.x = c.f(T * m->d_x);
  w.scale = c.f(T);
  w.d_pose = c.f(T * 66);
  w.d_shape = c.f(T * 10);
  w.pos = c.f(T * 36);
  w.ori = c.f(T * 108);
  w.joints = c.f(T * 66);
  w.x_t = c.f((T + TL_FR - 1) / TL_FR * TL_FR * m->d_in);
  w.smpl = carve_smpl(c, m, (int)T);
  w.upd = carve_upd(c, m, (int)T);
  if (m->rnn_init) {
    w.lstm = carve_lstm(c, m, B, F);
    w.y = c.f(T * m->rnn.H);
  } else {
    w.y = nullptr;
  }
  return w;
}

size_t empose_lgd_workspace_bytes(const empose_model_t* m, int B, int F) {
  if (!m || B <= 0 || F <= 0) return 0;
  Carver c(nullptr);
  carve_lgd(c, m, B, F);
  return c.off;
}

int empose_lgd_forward(const empose_model_t* m, const empose_lgd_io* io, void* workspace, size_t workspace_bytes,
                       empose_stream_t stream_) {
  return empose_lgd_forward_phase(m, io, workspace, workspace_bytes, stream_, EMPOSE_LGD_PHASE_INIT | EMPOSE_LGD_PHASE_ITER);
}

int empose_lgd_forward_phase(const empose_model_t* m, const empose_lgd_io* io, void* workspace, size_t workspace_bytes,
                             empose_stream_t stream_, int phases) {
  if (!m || !io || !workspace) return fail(EMPOSE_EINVAL, "null argument");
  if (phases < 1 || phases > 3) return fail(EMPOSE_EINVAL, "phases: EMPOSE_LGD_PHASE_INIT, _ITER or both");
  if (m->smpl_only) return fail(EMPOSE_EINVAL, "this handle holds the body model only (no networks)");
  const int B = io->B, F = io->F;
  if (B <= 0 || F <= 0) return fail(EMPOSE_EINVAL, "B and F must be positive");
  if (!io->marker_pos || !io->marker_oris || !io->offset_t || !io->offset_r || !io->pose_hat || !io->shape_hat ||
      !io->joints_hat)
    return fail(EMPOSE_EINVAL, "null input/output tensor");
  if (workspace_bytes < empose_lgd_workspace_bytes(m, B, F)) return fail(EMPOSE_ENOMEM, "workspace too small");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int T = B * F;
  Carver c(workspace);
  LgdWs w = carve_lgd(c, m, B, F);
  const int dx = m->d_x, din = m->d_in;
  float* x_theta = w.x + din;
  float* x_beta = w.x + din + 66;
  float* x_gtheta = w.x + din + 76;
  float* x_gbeta = w.x + din + 142;

  if (phases & EMPOSE_LGD_PHASE_INIT) {
  PackArgs pa;
  pa.marker_pos = io->marker_pos; pa.marker_oris = io->marker_oris; pa.marker_masks = io->marker_masks;
  pa.seq_lengths = io->seq_lengths; pa.x = w.x; pa.ldx = dx; pa.frame_scale = w.scale;
  pa.B = B; pa.F = F; pa.n_markers = m->n_markers;
  pa.rows_as_unpadded = (m->shape_avg == 2) ? 1 : 0;
  pa.suppress_missing = io->suppress_missing; pa.mask_value = io->mask_value;
  for (int i = 0; i < 12; ++i) pa.marker_idx[i] = m->marker_idx[i];
  prof_mark(P_PACK, stream);
  hipError_t e = launch_pack_inputs(pa, stream);
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "pack kernel: %s", hipGetErrorString(e));
  if (m->use_gradient && m->N > 0 && use_tile_path(m, T)) {   // the targets of the frame-per-lane kernel, once per forward
    e = launch_rows_to_tile(w.x, dx, m->d_in, w.x_t, T, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "tile transpose: %s", hipGetErrorString(e));
  }

  // ---- initial estimate (reference models.py:511-526)
  if (m->rnn_init) {
    TRY(run_lstm(m->rnn, B, F, w.x, dx, io->seq_lengths, io->h0, io->c0, w.y, io->h_n, io->c_n, w.lstm, stream));
    GemmBatch b;
    b.count = 2;
    b.p[0] = linear_prob(w.y, m->rnn.H, m->pose_head, x_theta, dx, T);
    b.p[1] = linear_prob(w.y, m->rnn.H, m->shape_head, w.d_shape, 10, T);
    prof_mark(P_HEADS, stream);
    if (m->heads_frag && options().heads_rows != 0 && heads_rows_applicable(T, m->rnn.H))
      e = launch_heads_rows(w.y, m->rnn.H, (options().rows_x3 != 0 && m->heads_frag3) ? m->heads_frag3 : m->heads_frag,
                            m->heads_bias, x_theta, dx, w.d_shape, 10, T, m->rnn.H, 66, 10,
                            options().rows_x3 != 0 && m->heads_frag3, stream);
    else
    e = launch_gemm(b, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "head gemm: %s", hipGetErrorString(e));
  } else {
    const Mlp* nets[2] = {&m->pose_init, &m->shape_init};
    float* outs[2] = {x_theta, w.d_shape};
    const int lds[2] = {dx, 10};
    TRY(run_mlps(nets, 2, outs, lds, w.x, dx, T, w.upd, m->hidden_max, stream, true));
  }
  }   // EMPOSE_LGD_PHASE_INIT
  if (!(phases & EMPOSE_LGD_PHASE_ITER)) return EMPOSE_OK;

  hipError_t e = hipSuccess;
  const int N = m->N;
  auto hist = [&](float* base, int i, size_t width) -> float* { return base ? base + (size_t)i * T * width : nullptr; };
  for (int i = 0; i <= N; ++i) {
    FeatArgs fa;
    fa.theta = x_theta; fa.ld_theta = dx; fa.beta = x_beta; fa.ld_beta = dx;
    fa.shape_avg = m->shape_avg;
    fa.seq_lengths = io->seq_lengths;
    if (i == 0) {
      fa.d_theta = nullptr; fa.theta_step = 0.f;
      fa.d_beta = w.d_shape; fa.beta_keep = 0.f; fa.beta_step = 1.f;
    } else {
      fa.d_theta = w.d_pose; fa.theta_step = m->step;
      fa.d_beta = w.d_shape; fa.beta_keep = 1.f; fa.beta_step = m->step;
    }
    const bool tile = use_tile_path(m, T);
    fa.out_theta = hist(io->hist_pose, i, 66); fa.out_beta = hist(io->hist_shape, i, 10);
    fa.out_theta2 = (i == N) ? io->pose_hat : nullptr;
    fa.out_beta2 = (i == N) ? io->shape_hat : nullptr;

    const bool need_grad = (i < N) && m->use_gradient;
    float* hm = hist(io->hist_markers, i, 36);
    float* ho = hist(io->hist_markers_ori, i, 108);
    float* hj = hist(io->hist_joints, i, 66);
    if ((hm == nullptr) != (ho == nullptr)) return fail(EMPOSE_EINVAL, "hist_markers and hist_markers_ori go together");
    GradOut go{x_gtheta, dx, x_gbeta, dx, hist(io->trace_g_pose, i, 66), hist(io->trace_g_shape, i, 10)};
    // (the frame-per-lane kernel skips outputs nobody asked for; the general kernel always writes its scratch copies)
    TRY(run_smpl_eval(m, T, F, w.smpl, fa, io->offset_r, io->offset_t, need_grad ? w.x : nullptr, dx, w.scale,
                      hm ? hm : (tile ? nullptr : w.pos), ho ? ho : (tile ? nullptr : w.ori),
                      (i == N) ? io->joints_hat : (hj ? hj : (tile ? nullptr : w.joints)),
                      nullptr, nullptr, (i == N) ? hj : nullptr, stream, nullptr, nullptr, nullptr, w.x_t,
                      need_grad ? &go : nullptr));
    if (i == N) break;
    const Mlp* nets[2] = {&m->pose_iter, &m->shape_iter};
    float* outs[2] = {w.d_pose, w.d_shape};
    const int lds[2] = {66, 10};
    TRY(run_mlps(nets, 2, outs, lds, w.x, dx, T, w.upd, m->hidden_max, stream));
  }
  prof_mark(P_END, stream);
  if (g_prof.on && g_prof.only >= 0) {   // calibration: two event packets with nothing in between
    prof_mark(P_EVENT_PAIR, stream);
    prof_mark(P_END, stream);
  }
  return EMPOSE_OK;
}

int empose_smpl_sensors_fwd_bwd(const empose_model_t* m, int T, int F, const float* theta, int ld_theta,
                                const float* beta, int ld_beta, const float* offset_r, const float* offset_t,
                                const float* tgt, int ld_tgt, const float* frame_scale, float* pos, float* ori,
                                float* joints, float* g_theta, int ld_g, float* g_beta, int ld_gb, void* workspace,
                                size_t workspace_bytes, empose_stream_t stream_) {
  if (!m || !theta || !beta || !offset_r || !offset_t || !pos || !ori || !joints || !workspace)
    return fail(EMPOSE_EINVAL, "null argument");
  if (T <= 0 || F <= 0 || T % F != 0) return fail(EMPOSE_EINVAL, "T must be a positive multiple of F");
  if (tgt && (!frame_scale || !g_theta || !g_beta)) return fail(EMPOSE_EINVAL, "gradient outputs missing");
  if (workspace_bytes < empose_smpl_workspace_bytes(m, T)) return fail(EMPOSE_ENOMEM, "workspace too small");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  Carver c(workspace);
  SmplWs ws = carve_smpl(c, m, T);
  FeatArgs fa;   // the caller's rows are read in place (no update: the kernel does not write them back)
  fa.theta = const_cast<float*>(theta); fa.ld_theta = ld_theta; fa.beta = const_cast<float*>(beta); fa.ld_beta = ld_beta;
  fa.d_theta = nullptr; fa.d_beta = nullptr; fa.theta_step = 0.f; fa.beta_keep = 1.f; fa.beta_step = 0.f;
  const bool tile = use_tile_path(m, T);
  fa.shape_avg = 0;
  fa.out_theta = fa.out_beta = fa.out_theta2 = fa.out_beta2 = nullptr;
  if (tgt && tile) {
    hipError_t e = launch_rows_to_tile(tgt, ld_tgt, 12 * m->n_markers, ws.tgt_t, T, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "tile transpose: %s", hipGetErrorString(e));
  }
  GradOut go{g_theta, ld_g, g_beta, ld_gb, nullptr, nullptr};
  TRY(run_smpl_eval(m, T, F, ws, fa, offset_r, offset_t, tgt, ld_tgt, frame_scale, pos, ori, joints, nullptr, nullptr,
                    nullptr, stream, nullptr, nullptr, nullptr, tile ? ws.tgt_t : nullptr, tgt ? &go : nullptr));
  return EMPOSE_OK;
}

int empose_smpl_sensors_vjp(const empose_model_t* m, int T, int F, const float* theta, int ld_theta, const float* beta,
                            int ld_beta, const float* offset_r, const float* offset_t, const float* d_pos,
                            const float* d_ori, const float* d_joints, float* g_theta, float* g_beta, void* workspace,
                            size_t workspace_bytes, empose_stream_t stream_) {
  if (!m || !theta || !beta || !offset_r || !offset_t || !d_pos || !d_ori || !g_theta || !g_beta || !workspace)
    return fail(EMPOSE_EINVAL, "null argument");
  if (T <= 0 || F <= 0 || T % F != 0) return fail(EMPOSE_EINVAL, "T must be a positive multiple of F");
  if (workspace_bytes < empose_smpl_workspace_bytes(m, T) + (size_t)T * (36 + 108 + 66) * sizeof(float) + 1024)
    return fail(EMPOSE_ENOMEM, "workspace too small (need empose_smpl_vjp_workspace_bytes)");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  Carver c(workspace);
  SmplWs ws = carve_smpl(c, m, T);
  float* pos = c.f((size_t)T * 36);
  float* ori = c.f((size_t)T * 108);
  float* joints = c.f((size_t)T * 66);
  FeatArgs fa;   // the caller's rows are read in place (no update: the kernel does not write them back)
  fa.theta = const_cast<float*>(theta); fa.ld_theta = ld_theta; fa.beta = const_cast<float*>(beta); fa.ld_beta = ld_beta;
  fa.d_theta = nullptr; fa.d_beta = nullptr; fa.theta_step = 0.f; fa.beta_keep = 1.f; fa.beta_step = 0.f;
  const bool tile = use_tile_path(m, T, d_joints);
  fa.shape_avg = 0;
  fa.out_theta = fa.out_beta = fa.out_theta2 = fa.out_beta2 = nullptr;
  GradOut go{g_theta, 66, g_beta, 10, nullptr, nullptr};
  TRY(run_smpl_eval(m, T, F, ws, fa, offset_r, offset_t, nullptr, 0, nullptr, tile ? nullptr : pos, tile ? nullptr : ori,
                    tile ? nullptr : joints, nullptr, nullptr, nullptr, stream, d_pos, d_ori, d_joints, nullptr, &go));
  return EMPOSE_OK;
}

size_t empose_smpl_vjp_workspace_bytes(const empose_model_t* m, int T) {
  return empose_smpl_workspace_bytes(m, T) + (size_t)T * (36 + 108 + 66) * sizeof(float) + 1024;
}

int empose_update_nets_fwd(const empose_model_t* m, int T, const float* x, int ldx, float* d_pose, float* d_shape,
                           void* workspace, size_t workspace_bytes, empose_stream_t stream_) {
  if (!m || !x || !d_pose || !d_shape || !workspace) return fail(EMPOSE_EINVAL, "null argument");
  if (m->pose_iter.n_layers == 0) return fail(EMPOSE_EINVAL, "model has no update nets");
  if (ldx < m->d_x || ldx % 4 != 0) return fail(EMPOSE_EINVAL, "ldx must be >= %d and a multiple of 4", m->d_x);
  if (workspace_bytes < empose_update_workspace_bytes(m, T)) return fail(EMPOSE_ENOMEM, "workspace too small");
  Carver c(workspace);
  UpdWs ws = carve_upd(c, m, T);
  const Mlp* nets[2] = {&m->pose_iter, &m->shape_iter};
  float* outs[2] = {d_pose, d_shape};
  const int lds[2] = {66, 10};
  return run_mlps(nets, 2, outs, lds, x, ldx, T, ws, m->hidden_max, static_cast<hipStream_t>(stream_));
}

int empose_lstm_fwd(const empose_model_t* m, int B, int F, const float* x, int ldx, const int* seq_lengths,
                    const float* h0, const float* c0, float* y, float* h_n, float* c_n, void* workspace,
                    size_t workspace_bytes, empose_stream_t stream_) {
  if (!m || !x || !y || !workspace) return fail(EMPOSE_EINVAL, "null argument");
  if (!m->rnn_init) return fail(EMPOSE_EINVAL, "model has no LSTM");
  if (ldx % 4 != 0) return fail(EMPOSE_EINVAL, "ldx must be a multiple of 4");
  if (workspace_bytes < empose_lstm_workspace_bytes(m, B, F)) return fail(EMPOSE_ENOMEM, "workspace too small");
  Carver c(workspace);
  LstmWs ws = carve_lstm(c, m, B, F);
  return run_lstm(m->rnn, B, F, x, ldx, seq_lengths, h0, c0, y, h_n, c_n, ws, static_cast<hipStream_t>(stream_));
}

void empose_rnn_destroy(empose_rnn_t* rnn) {
  if (!rnn) return;
  for (void* p : rnn->allocs) (void)hipFree(p);
  delete rnn;
}

int empose_rnn_create(const empose_rnn_desc* d, empose_rnn_t** out) {
  if (!d || !out) return fail(EMPOSE_EINVAL, "null argument");
  *out = nullptr;
  empose_rnn* r = new empose_rnn();
  empose_lstm_desc base;
  base.num_layers = d->num_layers; base.input_size = d->input_size; base.hidden_size = d->hidden_size;
  const int rc = pack_lstm(r->allocs, base, d->bidirectional ? 2 : 1, d->w_ih, d->w_hh, d->b_ih, d->b_hh, &r->rnn);
  if (rc != EMPOSE_OK) { empose_rnn_destroy(r); return rc; }
  if (!d->bidirectional && d->num_layers > 4) { empose_rnn_destroy(r); return fail(EMPOSE_EINVAL, "at most 4 stacked layers"); }
  *out = r;
  return EMPOSE_OK;
}

size_t empose_rnn_workspace_bytes(const empose_rnn_t* rnn, int B, int F) {
  if (!rnn || B <= 0 || F <= 0) return 0;
  Carver c(nullptr);
  carve_lstm_of(c, rnn->rnn, B, F);
  return c.off;
}

int empose_rnn_fwd(const empose_rnn_t* rnn, int B, int F, const float* x, int ldx, const int* seq_lengths,
                   const float* h0, const float* c0, float* y, float* h_n, float* c_n, void* workspace,
                   size_t workspace_bytes, empose_stream_t stream_) {
  if (!rnn || !x || !y || !workspace) return fail(EMPOSE_EINVAL, "null argument");
  if (B <= 0 || F <= 0) return fail(EMPOSE_EINVAL, "B and F must be positive");
  if (ldx % 4 != 0 || ldx < rnn->rnn.input_size) return fail(EMPOSE_EINVAL, "ldx must be a multiple of 4 and >= input_size");
  if (workspace_bytes < empose_rnn_workspace_bytes(rnn, B, F)) return fail(EMPOSE_ENOMEM, "workspace too small");
  Carver c(workspace);
  LstmWs ws = carve_lstm_of(c, rnn->rnn, B, F);
  return run_lstm(rnn->rnn, B, F, x, ldx, seq_lengths, h0, c0, y, h_n, c_n, ws, static_cast<hipStream_t>(stream_));
}

int empose_linear_f32_ex(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                         const float* scale, const float* shift, const float* resid, int ldr, int act, float slope,
                         empose_stream_t stream_) {
  if (!A || !W || !C) return fail(EMPOSE_EINVAL, "null argument");
  if (K % 4 != 0 || lda % 4 != 0 || ldw % 4 != 0) return fail(EMPOSE_EINVAL, "K, lda, ldw must be multiples of 4");
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return fail(EMPOSE_EINVAL, "A and W must be 16-byte aligned");
  if (act < 0 || act > 2) return fail(EMPOSE_EINVAL, "act must be 0 (none), 1 (PReLU, residual added after) or 2 (residual, then ReLU)");
  GemmBatch b;
  b.count = 1;
  GemmProb& p = b.p[0];
  p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.scale = scale; p.shift = shift; p.resid = resid; p.ldr = ldr; p.act = act; p.slope = slope;
  hipError_t e = launch_gemm(b, static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "gemm launch: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

int empose_gemm_strided_f32(int M, int N, int K, const float* A, long a_rs, long a_ks, const float* W, long w_rs,
                            long w_ks, float* C, int ldc, const float* bias, empose_stream_t stream_) {
  if (!A || !W || !C) return fail(EMPOSE_EINVAL, "null argument");
  if (M <= 0 || N <= 0 || K <= 0 || ldc < N) return fail(EMPOSE_EINVAL, "bad sizes");
  if (!strided_gemm_applicable(M, N))
    return fail(EMPOSE_EINVAL, "problem too large for the small-problem GEMM (%d x %d outputs)", M, N);
  StridedGemm p;
  p.A = A; p.a_rs = a_rs; p.a_ks = a_ks; p.W = W; p.w_rs = w_rs; p.w_ks = w_ks; p.C = C; p.ldc = ldc; p.bias = bias;
  p.M = M; p.N = N; p.K = K;
  hipError_t e = launch_strided_gemm(p, static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "strided gemm launch: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

int empose_gemm_strided_applicable(int M, int N) { return strided_gemm_applicable(M, N) ? 1 : 0; }

size_t empose_bn_prelu_workspace_bytes(int M, int C) {
  return (M > 0 && C > 0) ? bn_prelu_workspace_floats(M, C) * sizeof(float) : 0;
}

int empose_bn_prelu_train_fwd(int M, int C, const float* x, int ldx, const float* gamma, const float* beta,
                              const float* slope, float eps, float momentum, float* running_mean, float* running_var,
                              long long* num_batches_tracked, float* z, int ldz, float* save_mean, float* save_rstd,
                              void* workspace, size_t workspace_bytes, empose_stream_t stream_) {
  if (!x || !gamma || !beta || !slope || !z || !save_mean || !save_rstd) return fail(EMPOSE_EINVAL, "null argument");
  if (M <= 0 || C <= 0 || ldx < C || ldz < C) return fail(EMPOSE_EINVAL, "bad sizes");
  if (bn_prelu_workspace_floats(M, C) * sizeof(float) > (workspace ? workspace_bytes : 0))
    return fail(EMPOSE_ENOMEM, "workspace too small (empose_bn_prelu_workspace_bytes)");
  if ((running_mean == nullptr) != (running_var == nullptr)) return fail(EMPOSE_EINVAL, "running_mean and running_var go together");
  BnPreluArgs a{};
  a.M = M; a.C = C; a.x = x; a.ldx = ldx; a.gamma = gamma; a.beta = beta; a.slope = slope; a.eps = eps;
  a.momentum = momentum; a.running_mean = running_mean; a.running_var = running_var;
  a.num_batches_tracked = num_batches_tracked; a.z = z; a.ldz = ldz; a.save_mean = save_mean; a.save_rstd = save_rstd;
  a.workspace = static_cast<float*>(workspace);
  hipError_t e = launch_bn_prelu(a, false, static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "bn_prelu forward: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

int empose_bn_prelu_train_bwd(int M, int C, const float* x, int ldx, const float* dz, int lddz, const float* gamma,
                              const float* beta, const float* slope, const float* save_mean, const float* save_rstd,
                              float* dx, int lddx, float* dgamma, float* dbeta, float* dslope, float* dslope_partial,
                              int* counter, void* workspace, size_t workspace_bytes, empose_stream_t stream_) {
  if (!x || !dz || !gamma || !beta || !slope || !save_mean || !save_rstd || !dx || !dgamma || !dbeta || !dslope ||
      !dslope_partial || !counter)
    return fail(EMPOSE_EINVAL, "null argument");
  if (M <= 0 || C <= 0 || ldx < C || lddz < C || lddx < C) return fail(EMPOSE_EINVAL, "bad sizes");
  if (bn_prelu_workspace_floats(M, C) * sizeof(float) > (workspace ? workspace_bytes : 0))
    return fail(EMPOSE_ENOMEM, "workspace too small (empose_bn_prelu_workspace_bytes)");
  BnPreluArgs a{};
  a.workspace = static_cast<float*>(workspace);
  a.M = M; a.C = C; a.x = x; a.ldx = ldx; a.gamma = gamma; a.beta = beta; a.slope = slope;
  a.save_mean = const_cast<float*>(save_mean); a.save_rstd = const_cast<float*>(save_rstd);
  a.dz = dz; a.lddz = lddz; a.dx = dx; a.lddx = lddx; a.dgamma = dgamma; a.dbeta = dbeta; a.dslope_partial = dslope_partial; a.dslope = dslope; a.counter = counter;
  hipError_t e = launch_bn_prelu(a, true, static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "bn_prelu backward: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

// ---- training backward building blocks ------------------------------------------------------------------------
size_t empose_gemm_atb_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return atb_workspace_floats(M, N, K) * sizeof(float) + 256;
}

int empose_gemm_atb_f32(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                        float* bias, void* workspace, size_t workspace_bytes, empose_stream_t stream_) {
  if (!A || !B || !C) return fail(EMPOSE_EINVAL, "null argument");
  if (M <= 0 || N <= 0 || K <= 0 || lda < N || ldb < K || ldc < K) return fail(EMPOSE_EINVAL, "bad sizes");
  const size_t need = atb_workspace_floats(M, N, K) * sizeof(float);
  if (need > 0 && (!workspace || workspace_bytes < need)) return fail(EMPOSE_ENOMEM, "workspace too small");
  AtbArgs a{};
  a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.bias = bias; a.M = M; a.N = N; a.K = K;
  a.accumulate = 0;
  hipError_t e = launch_gemm_atb(a, static_cast<float*>(workspace), workspace_bytes / sizeof(float),
                                 static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "A^T B gemm: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

int empose_transpose_f32(int rows, int cols, const float* src, int ld_src, float* dst, int ld_dst,
                         empose_stream_t stream_) {
  if (!src || !dst) return fail(EMPOSE_EINVAL, "null argument");
  if (rows <= 0 || cols <= 0 || ld_src < cols || ld_dst < rows) return fail(EMPOSE_EINVAL, "bad sizes");
  hipError_t e = launch_transpose(src, ld_src, dst, ld_dst, rows, cols, static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "transpose: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

int empose_pack_inputs(int B, int F, int n_markers, const int* marker_idx, const float* marker_pos,
                       const float* marker_oris, const float* marker_masks, const int* seq_lengths, float* x, int ldx,
                       float* frame_weight, empose_stream_t stream_) {
  if (!marker_idx || !marker_pos || !marker_oris || !x) return fail(EMPOSE_EINVAL, "null argument");
  if (B <= 0 || F <= 0 || n_markers < 1 || n_markers > 12 || ldx < 12 * n_markers) return fail(EMPOSE_EINVAL, "bad sizes");
  PackArgs pa;
  pa.marker_pos = marker_pos; pa.marker_oris = marker_oris; pa.marker_masks = marker_masks; pa.seq_lengths = seq_lengths;
  pa.x = x; pa.ldx = ldx; pa.frame_scale = frame_weight; pa.B = B; pa.F = F; pa.n_markers = n_markers;
  for (int i = 0; i < 12; ++i) {
    pa.marker_idx[i] = i < n_markers ? marker_idx[i] : 0;
    if (pa.marker_idx[i] < 0 || pa.marker_idx[i] > 11) return fail(EMPOSE_EINVAL, "sensor index out of range");
  }
  hipError_t e = launch_pack_inputs(pa, static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "pack kernel: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

int empose_window_mean(int T, int F, int C, const float* in, int ld_in, float* out, int ld_out, empose_stream_t stream_) {
  if (!in || !out) return fail(EMPOSE_EINVAL, "null argument");
  if (T <= 0 || F <= 0 || C <= 0 || T % F != 0 || ld_in < C || ld_out < C) return fail(EMPOSE_EINVAL, "bad sizes");
  hipError_t e = launch_window_mean(in, ld_in, out, ld_out, T, F, C, static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "window mean: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

int empose_axpby2d(int rows, int cols, float alpha, const float* x, int ldx, float beta, const float* y, int ldy,
                   float* out, int ldo, empose_stream_t stream_) {
  if (!out) return fail(EMPOSE_EINVAL, "null argument");
  if (rows <= 0 || cols <= 0 || ldo < cols || (x && ldx < cols) || (y && ldy < cols)) return fail(EMPOSE_EINVAL, "bad sizes");
  hipError_t e = launch_axpby2d(rows, cols, alpha, x, ldx, beta, y, ldy, out, ldo, static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "axpby: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

int empose_lgd_assemble_inputs(int T, int d_in, const float* x0, int ld_x0, const float* pose, const float* shape,
                               float* X, int ldx, empose_stream_t stream_) {
  if (!x0 || !pose || !shape || !X) return fail(EMPOSE_EINVAL, "null argument");
  if (T <= 0 || d_in <= 0 || ld_x0 < d_in || ldx < d_in + 76) return fail(EMPOSE_EINVAL, "bad sizes");
  hipError_t e = launch_lgd_assemble(T, d_in, x0, ld_x0, pose, shape, X, ldx, static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "assemble: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

int empose_lgd_additive_update(int B, int F, float step, int shape_avg, const float* pose, const float* d_pose,
                               const float* shape, const float* d_shape, float* pose_next, float* shape_next,
                               empose_stream_t stream_) {
  if (!pose || !d_pose || !shape || !d_shape || !pose_next || !shape_next) return fail(EMPOSE_EINVAL, "null argument");
  if (B <= 0 || F <= 0) return fail(EMPOSE_EINVAL, "bad sizes");
  hipError_t e = launch_lgd_update(B, F, step, shape_avg, pose, d_pose, shape, d_shape, pose_next, shape_next,
                                   static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "update: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

int empose_lgd_cotangent_step(int B, int F, int first, const float* d_pose, const float* d_shape, const float* vp,
                              const float* vs, const float* g_theta, int ld_g, const float* g_beta, int ld_gb, float* Dp,
                              float* Ds, float step, int shape_avg, float* dpad, float* dspad, empose_stream_t stream_) {
  if (!d_pose || !d_shape || !vp || !vs || !Dp || !Ds) return fail(EMPOSE_EINVAL, "null argument");
  if (B <= 0 || F <= 0 || (size_t)F * 10 * sizeof(float) > 48 * 1024) return fail(EMPOSE_EINVAL, "bad sizes");
  if ((g_theta && ld_g < 66) || (g_beta && ld_gb < 10) || ((dpad == nullptr) != (dspad == nullptr)))
    return fail(EMPOSE_EINVAL, "bad arguments");
  hipError_t e = launch_lgd_cotangent(B, F, first, d_pose, d_shape, vp, vs, g_theta, ld_g, g_beta, ld_gb, Dp, Ds, step,
                                      shape_avg, dpad, dspad, static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "cotangent step: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

size_t empose_lgd_losses_workspace_bytes(int B, int F, int n_hist) {
  if (B <= 0 || F <= 0 || n_hist <= 0) return 0;
  return (size_t)4 * n_hist * B * F * sizeof(float) + 256;
}

int empose_lgd_losses(const empose_loss_io* io, void* workspace, size_t workspace_bytes, empose_stream_t stream_) {
  if (!io || !workspace) return fail(EMPOSE_EINVAL, "null argument");
  if (io->B <= 0 || io->F <= 0 || io->n_hist <= 0 || (io->n_markers != 6 && io->n_markers != 12))
    return fail(EMPOSE_EINVAL, "bad sizes");
  if (!io->pose_hist || !io->shape_hist || !io->markers_hist || !io->markers_ori_hist || !io->joints_final ||
      !io->pose_gt || !io->shape_gt || !io->inputs || !io->d_pose || !io->d_shape || !io->d_markers ||
      !io->d_markers_ori || !io->d_joints || !io->loss_vals)
    return fail(EMPOSE_EINVAL, "null tensor");
  if (workspace_bytes < empose_lgd_losses_workspace_bytes(io->B, io->F, io->n_hist)) return fail(EMPOSE_ENOMEM, "workspace too small");
  LossArgs a;
  a.B = io->B; a.F = io->F; a.N1 = io->n_hist; a.n_markers = io->n_markers;
  for (int m = 0; m < 12; ++m) a.used_slot[m] = -1;
  for (int i = 0; i < io->n_markers; ++i) {
    if (io->marker_idx[i] < 0 || io->marker_idx[i] >= 12) return fail(EMPOSE_EINVAL, "marker_idx out of range");
    a.used_slot[io->marker_idx[i]] = i;
  }
  a.pose_hist = io->pose_hist; a.shape_hist = io->shape_hist; a.pos_hist = io->markers_hist; a.ori_hist = io->markers_ori_hist;
  a.joints_final = io->joints_final; a.pose_gt = io->pose_gt; a.shape_gt = io->shape_gt; a.joints_gt = io->joints_gt;
  a.x_in = io->inputs; a.ldx = io->ld_inputs; a.seq_lengths = io->seq_lengths; a.masks = io->marker_masks;
  a.w_pose = io->w_pose; a.w_shape = io->w_shape; a.w_fk = io->w_fk; a.w_rec = io->w_rec;
  a.d_pose = io->d_pose; a.d_shape = io->d_shape; a.d_pos = io->d_markers; a.d_ori = io->d_markers_ori;
  a.d_joints = io->d_joints; a.partial = static_cast<float*>(workspace); a.loss_vals = io->loss_vals;
  hipError_t e = launch_lgd_losses(a, static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "loss kernels: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

int empose_adam_step(int n_chunks, const void* params, const void* grads, const void* exp_avg, const void* exp_avg_sq,
                     const void* sizes, const void* chunk_tensor, const void* chunk_offset, float lr, float beta1,
                     float beta2, float eps, int step, empose_stream_t stream_) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !sizes || !chunk_tensor || !chunk_offset)
    return fail(EMPOSE_EINVAL, "null argument");
  if (n_chunks <= 0 || step < 1) return fail(EMPOSE_EINVAL, "bad sizes");
  AdamArgs a;
  a.params = static_cast<void* const*>(params); a.grads = static_cast<void* const*>(grads);
  a.exp_avg = static_cast<void* const*>(exp_avg); a.exp_avg_sq = static_cast<void* const*>(exp_avg_sq);
  a.sizes = static_cast<const long long*>(sizes); a.chunk_tensor = static_cast<const int*>(chunk_tensor);
  a.chunk_offset = static_cast<const long long*>(chunk_offset);
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
  a.step_size = (float)((double)lr / bc1);
  a.inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2));
  hipError_t e = launch_adam(a, n_chunks, static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "adam: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

// ---- one MLP in training mode -------------------------------------------------------------------------------------
namespace {
int check_mlp_params(const empose_mlp_params* p) {
  if (!p) return fail(EMPOSE_EINVAL, "null argument");
  if (p->n_layers < 2 || p->n_layers > EMPOSE_MAX_DENSE || p->in_dim <= 0 || p->hidden <= 0 || p->out_dim <= 0 ||
      p->in_dim % 4 != 0 || p->hidden % 4 != 0)
    return fail(EMPOSE_EINVAL, "unsupported MLP configuration");
  for (int l = 0; l < p->n_layers; ++l) {
    if (!p->weight[l] || !p->bias[l]) return fail(EMPOSE_EINVAL, "null MLP parameter");
    if (l < p->n_layers - 1 && (!p->bn_weight[l] || !p->bn_bias[l] || !p->prelu[l]))
      return fail(EMPOSE_EINVAL, "the training MLP needs BatchNorm + PReLU on every hidden layer");
  }
  return EMPOSE_OK;
}
struct MlpTrainWs {
  float* d[2];       // [M][hidden] cotangent ping-pong
  float* wt;         // transposed weight [hidden][max(hidden, out_pad)]
  float* atb; size_t atb_floats; float* bn; float* slope_partial; int* counter;
  float* part; float* coef;   // fused path: per-row-block partial sums, BatchNorm-reverse coefficients [3][H]
  // one-launch layers (train_cols.hip): mailbox words, zeroed once per call
  unsigned long long* mbox; size_t mbox_bytes;
};
size_t cols_zero_bytes(const empose_mlp_params* p) {
  return cols_mailbox_words(p->hidden > p->out_dim ? p->hidden : p->out_dim) * sizeof(unsigned long long);
}
// every A^T B product of one MLP over M rows: (H, in_dim), (H, H), (out_dim, H)
size_t mlp_atb_floats(const empose_mlp_params* p, int M) {
  return atb_workspace_floats_max(M, {{p->hidden, p->in_dim}, {p->hidden, p->hidden}, {p->out_dim, p->hidden}});
}
MlpTrainWs carve_mlp_train(Carver& c, const empose_mlp_params* p, int M) {
  MlpTrainWs w;
  const int H = p->hidden, op = (p->out_dim + 3) & ~3;
  w.d[0] = c.f((size_t)M * H); w.d[1] = c.f((size_t)M * H);
  w.wt = c.f((size_t)H * (H > op ? H : op));
  w.atb_floats = mlp_atb_floats(p, M);
  w.atb = c.f(w.atb_floats + 64);
  w.bn = c.f(bn_prelu_workspace_floats(M, H) + 64);
  w.slope_partial = c.f((size_t)(H + 31) / 32 + 8);
  w.counter = reinterpret_cast<int*>(c.f(64));
  w.part = c.f(bn_fused_partial_floats(M, H) + 64);
  w.coef = c.f((size_t)3 * H + 64);
  w.mbox_bytes = cols_zero_bytes(p);
  w.mbox = reinterpret_cast<unsigned long long*>(c.f(w.mbox_bytes / sizeof(float)));
  return w;
}
// The BatchNorm / PReLU passes folded into the GEMMs (train_fused.hip).  Opt-in: gradient parity with the reference is
// tested, but at 256 windows the step is no faster (the operand transform and the statistics epilogue cost the GEMMs
// about what the removed passes cost: 701-711 k against 705-720 k frames/s).  Option "train_fused": 0 never (default),
// 1 from BN_SINGLE_PASS_ROWS rows on, 2 always (tests).
// (empose_mlp_params::save_layout != 0: the layout chosen when the step's forward ran wins over the options of the moment)
bool mlp_train_fused(const empose_mlp_params* p, int M) {
  if (p->save_layout) return p->save_layout == 2;
  const int opt = options().train_fused;
  return opt != 0 && (opt == 2 || M > BN_SINGLE_PASS_ROWS) && p->hidden % 4 == 0 && p->in_dim % 4 == 0;
}
// Round 4, the default above BN_SINGLE_PASS_ROWS rows: the statistics still come out of the GEMM epilogues, but the
// activations / cotangents are materialised by ONE combine-and-apply launch per layer and direction (train_fused.hip,
// bn_finish_*): GEMM + 1 launch instead of GEMM + 3, and every consumer reads a ready operand.  Option "train_epi":
// 0 never, 1 above BN_SINGLE_PASS_ROWS rows (default), 2 always (tests).  "train_fused" takes precedence when both apply.
bool mlp_train_epi(const empose_mlp_params* p, int M) {
  if (p->save_layout) return p->save_layout == 3;
  const int opt = options().train_epi;
  return !mlp_train_fused(p, M) && opt != 0 && (opt == 2 || M > BN_SINGLE_PASS_ROWS) && p->hidden % 4 == 0 &&
         p->in_dim % 4 == 0;
}
// per hidden layer: passes  z [M][H] | a [M][H] | mean [H] | rstd [H];  fused  y [M][H] | mean | rstd | s | t;
//                   epi     y [M][H] | a [M][H] | mean | rstd | s | t
size_t mlp_layer_save(const empose_mlp_params* p, int M) {
  if (mlp_train_fused(p, M)) return (size_t)M * p->hidden + 4 * (size_t)p->hidden;
  if (mlp_train_epi(p, M)) return (size_t)2 * M * p->hidden + 4 * (size_t)p->hidden;
  return (size_t)2 * M * p->hidden + 2 * (size_t)p->hidden;
}
// At the reference's training batch a layer is one launch: product, BatchNorm and PReLU of both update networks in
// train_cols.hip (option "train_cols": 0 never, 1 up to COLS_MAX_ROWS rows).  Reads and writes save layout 1 (passes).
bool mlp_train_cols(const empose_mlp_params* p, int M) {
  if (options().train_cols == 0 || M > COLS_MAX_ROWS) return false;
  if (mlp_train_fused(p, M) || mlp_train_epi(p, M)) return false;
  return cols_launchable(p->hidden > p->out_dim ? p->hidden : p->out_dim, 2);
}
bool mlp_cols_pairable(const empose_mlp_params* a, const empose_mlp_params* b, int M) {
  return a->n_layers == b->n_layers && a->hidden == b->hidden && a->bn_eps == b->bn_eps &&
         a->bn_momentum == b->bn_momentum && mlp_train_cols(a, M) && mlp_train_cols(b, M) &&
         b->out_dim <= (a->hidden > a->out_dim ? a->hidden : a->out_dim);
}

int mlp_fwd_cols(const empose_mlp_params* const* ps, int n, int M, const float* x, int ldx, float* const* outs,
                 const int* ld_outs, float* const* saves, const MlpTrainWs& w, hipStream_t stream) {
  const int L = ps[0]->n_layers;
  HIP_TRY(hipMemsetAsync(w.mbox, 0, w.mbox_bytes, stream));
  for (int l = 0; l < L; ++l) {
    const bool last = l == L - 1;
    ColsArgs a{};
    a.n_nets = n; a.M = M; a.eps = ps[0]->bn_eps; a.momentum = ps[0]->bn_momentum; a.tag = (unsigned)l + 1;
    a.mailbox = w.mbox;
    for (int i = 0; i < n; ++i) {
      const empose_mlp_params* p = ps[i];
      const int H = p->hidden;
      const size_t lsz = (size_t)2 * M * H + 2 * (size_t)H;
      float* sv = saves[i] + (size_t)l * lsz;                                   // z | a | mean | rstd
      ColsNet& c = a.net[i];
      c.A = l == 0 ? x : saves[i] + (size_t)(l - 1) * lsz + (size_t)M * H; c.lda = l == 0 ? ldx : H;
      c.W = p->weight[l]; c.ldw = l == 0 ? p->in_dim : H; c.bias = p->bias[l];
      c.N = last ? p->out_dim : H; c.K = l == 0 ? p->in_dim : H;
      if (last) { c.out = outs[i]; c.ld_out = ld_outs[i]; continue; }
      c.gamma = p->bn_weight[l]; c.beta = p->bn_bias[l]; c.slope = p->prelu[l];
      c.running_mean = p->bn_running_mean[l]; c.running_var = p->bn_running_var[l]; c.num_batches = p->bn_num_batches[l];
      c.z = sv; c.ldz = H; c.out = sv + (size_t)M * H; c.ld_out = H;
      c.mean = sv + (size_t)2 * M * H; c.rstd = c.mean + H;
    }
    hipError_t e = launch_cols(a, last ? 1 : 0, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "one-launch layer forward: %s", hipGetErrorString(e));
  }
  return EMPOSE_OK;
}
}  // namespace

int empose_mlp_train_uses_weight_t(const empose_mlp_params* p, int M) {
  if (!p || M <= 0) return fail(EMPOSE_EINVAL, "null parameters / no rows");
  return mlp_train_cols(p, M) ? 0 : 1;
}

int empose_mlp_train_save_layout(const empose_mlp_params* p, int M) {
  if (!p || M <= 0) return fail(EMPOSE_EINVAL, "null parameters / no rows");
  if (p->save_layout < 0 || p->save_layout > 3) return fail(EMPOSE_EINVAL, "save_layout must be 0 .. 3");
  return mlp_train_fused(p, M) ? 2 : (mlp_train_epi(p, M) ? 3 : 1);
}

size_t empose_mlp_train_save_floats(const empose_mlp_params* p, int M) {
  if (!p || M <= 0) return 0;
  return (size_t)(p->n_layers - 1) * mlp_layer_save(p, M);
}

size_t empose_mlp_train_workspace_bytes(const empose_mlp_params* p, int M) {
  if (!p || M <= 0) return 0;
  Carver c(nullptr);
  carve_mlp_train(c, p, M);
  return c.off;
}

int empose_mlp_train_fwd(const empose_mlp_params* p, int M, const float* x, int ldx, float* out, int ld_out,
                         float* save, void* workspace, size_t workspace_bytes, empose_stream_t stream_) {
  TRY(check_mlp_params(p));
  TRY(earlier_poll_timeouts());
  if (!x || !out || !save || !workspace) return fail(EMPOSE_EINVAL, "null argument");
  if (M <= 0 || ldx < p->in_dim || ldx % 4 != 0 || ld_out < p->out_dim) return fail(EMPOSE_EINVAL, "bad sizes");
  if (workspace_bytes < empose_mlp_train_workspace_bytes(p, M)) return fail(EMPOSE_ENOMEM, "workspace too small");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  Carver c(workspace);
  MlpTrainWs w = carve_mlp_train(c, p, M);
  const int H = p->hidden, L = p->n_layers;
  if (mlp_train_cols(p, M)) return mlp_fwd_cols(&p, 1, M, x, ldx, &out, &ld_out, &save, w, stream);
  if (mlp_train_epi(p, M)) {
    // y_l = a_{l-1} W_l^T + b_l on the materialised a_{l-1}; the epilogue leaves the column statistics of y_l per row
    // block; ONE launch turns them into (mean, rstd, s, t), updates the running statistics and writes a_l = PReLU(s y_l + t)
    const size_t lsz = mlp_layer_save(p, M);
    for (int l = 0; l < L; ++l) {
      const bool last = l == L - 1;
      float* sv = save + (size_t)l * lsz;                      // this layer's y | a | mean | rstd | s | t
      const float* pa = l > 0 ? save + (size_t)(l - 1) * lsz + (size_t)M * H : nullptr;
      TrainGemmArgs g{};
      g.A = l == 0 ? x : pa; g.lda = l == 0 ? ldx : H; g.W = p->weight[l]; g.ldw = l == 0 ? p->in_dim : H;
      g.C = last ? out : sv; g.ldc = last ? ld_out : H;
      g.M = M; g.N = last ? p->out_dim : H; g.K = l == 0 ? p->in_dim : H; g.bias = p->bias[l];
      g.part = w.part;
      const bool x3 = !last && options().train_x3 != 0 && p->weight_x3[l] && gemm_train_x3_applicable(g.M, g.N, g.K);
      hipError_t e = x3 ? launch_gemm_train_x3(g, p->weight_x3[l], 1, stream) : launch_gemm_train(g, 0, last ? 0 : 1, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "mlp forward gemm (statistics epilogue): %s", hipGetErrorString(e));
      if (last) break;
      BnFinishFwdArgs c{};
      c.M = M; c.C = H; c.part = w.part; c.gamma = p->bn_weight[l]; c.beta = p->bn_bias[l];
      c.eps = p->bn_eps; c.momentum = p->bn_momentum; c.running_mean = p->bn_running_mean[l];
      c.running_var = p->bn_running_var[l]; c.num_batches_tracked = p->bn_num_batches[l];
      c.mean = sv + (size_t)2 * M * H; c.rstd = c.mean + H; c.s = c.rstd + H; c.t = c.s + H;
      c.y = sv; c.ldy = H; c.act = sv + (size_t)M * H; c.ld_act = H; c.slope = p->prelu[l];
      e = launch_bn_finish_fwd(c, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "bn finish forward: %s", hipGetErrorString(e));
    }
    return EMPOSE_OK;
  }
  if (mlp_train_fused(p, M)) {
    // y_l = a_{l-1} W_l^T + b_l with a_{l-1} = PReLU(s y_{l-1} + t) formed while the GEMM stages its A operand; the
    // epilogue leaves the column statistics of y_l per row block, a small kernel turns them into (mean, rstd, s, t)
    const size_t lsz = mlp_layer_save(p, M);
    for (int l = 0; l < L; ++l) {
      const bool last = l == L - 1;
      float* sv = save + (size_t)l * lsz;                      // this layer's y | mean | rstd | s | t
      const float* pv = l > 0 ? save + (size_t)(l - 1) * lsz : nullptr;
      TrainGemmArgs g{};
      g.A = l == 0 ? x : pv; g.lda = l == 0 ? ldx : H; g.W = p->weight[l]; g.ldw = l == 0 ? p->in_dim : H;
      g.C = last ? out : sv; g.ldc = last ? ld_out : H;
      g.M = M; g.N = last ? p->out_dim : H; g.K = l == 0 ? p->in_dim : H; g.bias = p->bias[l];
      if (l > 0) { g.a_s = pv + (size_t)M * H + 2 * H; g.a_t = g.a_s + H; g.a_slope = p->prelu[l - 1]; }
      g.part = w.part;
      hipError_t e = launch_gemm_train(g, l > 0 ? 1 : 0, last ? 0 : 1, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "fused mlp forward gemm: %s", hipGetErrorString(e));
      if (last) break;
      BnFusedFwdArgs c{};
      c.M = M; c.C = H; c.part = w.part; c.gamma = p->bn_weight[l]; c.beta = p->bn_bias[l];
      c.eps = p->bn_eps; c.momentum = p->bn_momentum; c.running_mean = p->bn_running_mean[l];
      c.running_var = p->bn_running_var[l]; c.num_batches_tracked = p->bn_num_batches[l];
      c.mean = sv + (size_t)M * H; c.rstd = c.mean + H; c.s = c.rstd + H; c.t = c.s + H;
      e = launch_bn_fused_combine_fwd(c, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "fused bn combine: %s", hipGetErrorString(e));
    }
    return EMPOSE_OK;
  }
  const float* in = x;
  int ld_in = ldx, k_in = p->in_dim;
  for (int l = 0; l < L; ++l) {
    const bool last = l == L - 1;
    float* sv = save + (size_t)l * mlp_layer_save(p, M);
    float* z = last ? out : sv;
    GemmBatch b;
    b.count = 1;
    GemmProb& g = b.p[0];
    g.A = in; g.lda = ld_in; g.W = p->weight[l]; g.ldw = k_in; g.C = z; g.ldc = last ? ld_out : H;
    g.M = M; g.N = last ? p->out_dim : H; g.K = k_in;
    g.scale = nullptr; g.shift = p->bias[l]; g.resid = nullptr; g.ldr = 0; g.act = 0; g.slope = 0.f;
    hipError_t e = launch_gemm(b, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "mlp forward gemm: %s", hipGetErrorString(e));
    if (last) break;
    float* act = sv + (size_t)M * H;
    BnPreluArgs a{};
    a.M = M; a.C = H; a.x = z; a.ldx = H; a.gamma = p->bn_weight[l]; a.beta = p->bn_bias[l]; a.slope = p->prelu[l];
    a.eps = p->bn_eps; a.momentum = p->bn_momentum; a.running_mean = p->bn_running_mean[l];
    a.running_var = p->bn_running_var[l]; a.num_batches_tracked = p->bn_num_batches[l];
    a.z = act; a.ldz = H; a.save_mean = sv + (size_t)2 * M * H; a.save_rstd = a.save_mean + H;
    a.workspace = w.bn;
    e = launch_bn_prelu(a, false, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "bn_prelu forward: %s", hipGetErrorString(e));
    in = act; ld_in = H; k_in = H;
  }
  return EMPOSE_OK;
}

namespace {
// stash of one application: dZ of the hidden layers [M][hidden] each, then a copy of d_out [M][out_pad]
size_t mlp_stash_floats(const empose_mlp_params* p, int M) {
  return (size_t)M * ((size_t)(p->n_layers - 1) * p->hidden + ((p->out_dim + 3) & ~3));
}
// The reverse sweep of one or two MLPs on the one-launch layers: layer l's launch forms dA_{l-1} = dZ_l W_l and, in its
// epilogue, the BatchNorm / PReLU reverse of layer l - 1 (whose column sums the row parts exchange) -> dZ_{l-1}.  With
// stashes the weight gradients are deferred (empose_mlp_train_wgrad); without (one network only) they are formed here.
int mlp_bwd_cols(const empose_mlp_params* const* ps, int n, int M, const float* x, int ldx, const float* const* d_outs,
                 const int* ld_douts, const float* const* saves, const empose_mlp_grads* const* grs, int accumulate,
                 float* const* stashes, const MlpTrainWs& w, hipStream_t stream) {
  const int L = ps[0]->n_layers;
  const bool deferred = stashes && stashes[0];
  if (!deferred && n != 1) return fail(EMPOSE_EINVAL, "a pair of networks runs its reverse sweep with deferred weight gradients");
  HIP_TRY(hipMemsetAsync(w.mbox, 0, w.mbox_bytes, stream));
  auto layer_save = [&](int i, int l) { return saves[i] + (size_t)l * ((size_t)2 * M * ps[i]->hidden + 2 * (size_t)ps[i]->hidden); };
  auto dz_of = [&](int i, int l) -> float* { return deferred ? stashes[i] + (size_t)M * l * ps[i]->hidden : w.d[l & 1]; };
  auto atb = [&](int l) -> int {   // dW_l, db_l of the single network (not deferred)
    const empose_mlp_params* p = ps[0];
    const int H = p->hidden;
    const bool last = l == L - 1;
    AtbArgs ab{};
    ab.A = last ? d_outs[0] : dz_of(0, l); ab.lda = last ? ld_douts[0] : H;
    ab.B = l == 0 ? x : layer_save(0, l - 1) + (size_t)M * H; ab.ldb = l == 0 ? ldx : H;
    ab.C = grs[0]->weight[l]; ab.ldc = l == 0 ? p->in_dim : H; ab.bias = grs[0]->bias[l];
    ab.M = M; ab.N = last ? p->out_dim : H; ab.K = l == 0 ? p->in_dim : H; ab.accumulate = accumulate;
    hipError_t e = launch_gemm_atb(ab, w.atb, w.atb_floats, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "dW: %s", hipGetErrorString(e));
    return EMPOSE_OK;
  };
  for (int i = 0; i < n && deferred; ++i) {   // keep d_out for empose_mlp_train_wgrad (no copy when it was produced in its slot)
    const int op = (ps[i]->out_dim + 3) & ~3;
    float* slot = stashes[i] + (size_t)M * (L - 1) * ps[i]->hidden;
    if (d_outs[i] != slot || ld_douts[i] != op) {
      hipError_t e = launch_axpby2d(M, op, 1.f, d_outs[i], ld_douts[i], 0.f, nullptr, 0, slot, op, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "stash: %s", hipGetErrorString(e));
    }
  }
  for (int l = L - 1; l >= 1; --l) {
    const bool last = l == L - 1;
    if (!deferred) TRY(atb(l));
    ColsArgs a{};
    a.n_nets = n; a.M = M; a.eps = ps[0]->bn_eps; a.momentum = ps[0]->bn_momentum; a.accumulate = accumulate;
    a.tag = (unsigned)l; a.mailbox = w.mbox;
    for (int i = 0; i < n; ++i) {
      const empose_mlp_params* p = ps[i];
      const int H = p->hidden, op = (p->out_dim + 3) & ~3, kdim = last ? op : H;
      const float* sv = layer_save(i, l - 1);
      ColsNet& c = a.net[i];
      c.A = last ? d_outs[i] : dz_of(i, l); c.lda = last ? ld_douts[i] : H;
      c.N = H; c.K = kdim;
      if (p->weight_t[l]) { c.W = p->weight_t[l]; c.ldw = kdim; }
      else { c.W = p->weight[l]; c.ldw = H; c.w_kmajor = 1; c.Kw = last ? p->out_dim : H; }   // the layer's own W, read by rows
      c.gamma = p->bn_weight[l - 1]; c.beta = p->bn_bias[l - 1]; c.slope = p->prelu[l - 1];
      c.z_in = sv; c.ldz = H; c.mean = const_cast<float*>(sv + (size_t)2 * M * H); c.rstd = c.mean + H;
      c.out = dz_of(i, l - 1); c.ld_out = H;
      c.dgamma = grs[i]->bn_weight[l - 1]; c.dbeta = grs[i]->bn_bias[l - 1]; c.dslope = grs[i]->prelu[l - 1];
    }
    hipError_t e = launch_cols(a, 2, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "one-launch layer backward: %s", hipGetErrorString(e));
  }
  if (!deferred) TRY(atb(0));
  return EMPOSE_OK;
}

int mlp_train_bwd_impl(const empose_mlp_params* p, int M, const float* x, int ldx, const float* d_out, int ld_dout,
                       const float* save, const empose_mlp_grads* gr, int accumulate, float* stash, void* workspace,
                       size_t workspace_bytes, empose_stream_t stream_) {
  TRY(check_mlp_params(p));
  TRY(earlier_poll_timeouts());
  if (!x || !d_out || !save || !gr || !workspace) return fail(EMPOSE_EINVAL, "null argument");
  const int H = p->hidden, L = p->n_layers, op = (p->out_dim + 3) & ~3;
  if (M <= 0 || ldx < p->in_dim || ld_dout < op || ld_dout % 4 != 0) return fail(EMPOSE_EINVAL, "bad sizes");
  if (workspace_bytes < empose_mlp_train_workspace_bytes(p, M)) return fail(EMPOSE_ENOMEM, "workspace too small");
  for (int l = 0; l < L; ++l) {
    if (!gr->weight[l] || !gr->bias[l]) return fail(EMPOSE_EINVAL, "null gradient output");
    if (l < L - 1 && (!gr->bn_weight[l] || !gr->bn_bias[l] || !gr->prelu[l])) return fail(EMPOSE_EINVAL, "null gradient output");
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  Carver c(workspace);
  MlpTrainWs w = carve_mlp_train(c, p, M);
  if (mlp_train_cols(p, M))
    return mlp_bwd_cols(&p, 1, M, x, ldx, &d_out, &ld_dout, &save, &gr, accumulate, &stash, w, stream);
  // the last-arriver counter of the single-pass BatchNorm reverse kernel (it re-arms itself; the workspace may be fresh)
  const bool epi = mlp_train_epi(p, M);
  if (M <= BN_SINGLE_PASS_ROWS || epi) HIP_TRY(hipMemsetAsync(w.counter, 0, sizeof(int), stream));
  auto gemm = [&](const float* A, int lda, const float* W, int ldw, float* C, int ldc, int N, int K) -> hipError_t {
    GemmBatch b;
    b.count = 1;
    GemmProb& g = b.p[0];
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.scale = nullptr; g.shift = nullptr; g.resid = nullptr; g.ldr = 0; g.act = 0; g.slope = 0.f;
    return launch_gemm(b, stream);
  };
  auto layer_save = [&](int l) { return save + (size_t)l * mlp_layer_save(p, M); };
  if (mlp_train_fused(p, M) || epi) {
    // The dX GEMM's epilogue writes dyh_l = dA_l * PReLU'(yhat_l) and the column sums BatchNorm's reverse needs; a
    // small kernel turns the sums into dgamma / dbeta / dslope and three per-column coefficients, one pass forms
    // dY_l = c1 dyh_l + c3 y_l + c0 in place (`epi`: both in ONE launch, bn_finish_bwd).  Fused: the layer inputs a_{l-1}
    // are not stored, the A^T B product re-forms them from y_{l-1} while it stages its B operand; `epi`: they are.
    auto stats_of = [&](int l) { return layer_save(l) + (size_t)(epi ? 2 : 1) * M * H; };   // mean | rstd | s | t
    auto dz_of = [&](int l) -> float* { return stash ? stash + (size_t)M * l * H : w.d[l & 1]; };
    auto atb = [&](int l) -> int {   // dW_l, db_l (not deferred)
      const bool last = l == L - 1;
      AtbArgs ab{};
      ab.A = last ? d_out : dz_of(l); ab.lda = last ? ld_dout : H;
      ab.B = l == 0 ? x : layer_save(l - 1) + (epi ? (size_t)M * H : 0); ab.ldb = l == 0 ? ldx : H;
      ab.C = gr->weight[l]; ab.ldc = l == 0 ? p->in_dim : H; ab.bias = gr->bias[l];
      ab.M = M; ab.N = last ? p->out_dim : H; ab.K = l == 0 ? p->in_dim : H; ab.accumulate = accumulate;
      if (l > 0 && !epi) { ab.b_mode = 1; ab.Bs_seg[0] = stats_of(l - 1) + 2 * H; ab.b_slope = p->prelu[l - 1]; }
      hipError_t e = launch_gemm_atb(ab, w.atb, w.atb_floats, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "fused dW: %s", hipGetErrorString(e));
      return EMPOSE_OK;
    };
    for (int l = L - 1; l >= 0; --l) {
      const bool last = l == L - 1;
      if (last && stash) {   // keep d_out for empose_mlp_train_wgrad (no copy when the caller produced it in its slot)
        float* slot = stash + (size_t)M * (L - 1) * H;
        if (d_out != slot || ld_dout != op) {
          hipError_t e = launch_axpby2d(M, op, 1.f, d_out, ld_dout, 0.f, nullptr, 0, slot, op, stream);
          if (e != hipSuccess) return fail(EMPOSE_EHIP, "stash: %s", hipGetErrorString(e));
        }
      }
      if (!stash) TRY(atb(l));
      if (l == 0) break;
      // dA_{l-1} = dY_l W_l on the forward tile against W_l^T, its epilogue already in terms of layer l - 1
      const float* wt = p->weight_t[l];
      const int kdim = last ? op : H;
      if (!wt) {
        if (last) HIP_TRY(hipMemsetAsync(w.wt, 0, (size_t)H * op * sizeof(float), stream));
        hipError_t e = launch_transpose(p->weight[l], H, w.wt, kdim, last ? p->out_dim : H, H, stream);
        if (e != hipSuccess) return fail(EMPOSE_EHIP, "transpose: %s", hipGetErrorString(e));
        wt = w.wt;
      }
      TrainGemmArgs g{};
      g.A = last ? d_out : dz_of(l); g.lda = last ? ld_dout : H; g.W = wt; g.ldw = kdim;
      g.C = dz_of(l - 1); g.ldc = H; g.M = M; g.N = H; g.K = kdim; g.bias = nullptr;
      g.part = w.part; g.e_y = layer_save(l - 1); g.ld_ey = H;
      g.e_mean = stats_of(l - 1); g.e_rstd = g.e_mean + H; g.e_s = g.e_rstd + H; g.e_t = g.e_s + H; g.e_slope = p->prelu[l - 1];
      const bool x3 = options().train_x3 != 0 && p->weight_t[l] && p->weight_t_x3[l] && gemm_train_x3_applicable(g.M, g.N, g.K);
      hipError_t e = x3 ? launch_gemm_train_x3(g, p->weight_t_x3[l], 2, stream) : launch_gemm_train(g, 0, 2, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "fused dX gemm: %s", hipGetErrorString(e));
      if (epi) {
        BnFinishBwdArgs f{};
        f.M = M; f.C = H; f.part = w.part; f.gamma = p->bn_weight[l - 1]; f.mean = stats_of(l - 1); f.rstd = f.mean + H;
        f.dgamma = gr->bn_weight[l - 1]; f.dbeta = gr->bn_bias[l - 1]; f.dslope = gr->prelu[l - 1];
        f.dslope_partial = w.slope_partial; f.counter = w.counter; f.accumulate = accumulate;
        f.dyh = dz_of(l - 1); f.ld = H; f.y = layer_save(l - 1); f.ldy = H;
        e = launch_bn_finish_bwd(f, stream);
        if (e != hipSuccess) return fail(EMPOSE_EHIP, "bn finish backward: %s", hipGetErrorString(e));
        continue;
      }
      BnFusedBwdArgs c{};
      c.M = M; c.C = H; c.part = w.part; c.gamma = p->bn_weight[l - 1]; c.mean = stats_of(l - 1); c.rstd = stats_of(l - 1) + H;
      c.dgamma = gr->bn_weight[l - 1]; c.dbeta = gr->bn_bias[l - 1]; c.dslope = gr->prelu[l - 1];
      c.dslope_partial = w.slope_partial; c.coef = w.coef; c.accumulate = accumulate;
      e = launch_bn_fused_combine_bwd(c, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "fused bn reverse combine: %s", hipGetErrorString(e));
      e = launch_bn_fused_apply_bwd(dz_of(l - 1), layer_save(l - 1), w.coef, M, H, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "fused bn reverse apply: %s", hipGetErrorString(e));
    }
    return EMPOSE_OK;
  }
  // output layer: dW = d_out^T a_{L-2}, db, dA = d_out . W
  {
    const int l = L - 1;
    hipError_t e = hipSuccess;
    if (stash) {   // weight gradients deferred: keep d_out for empose_mlp_train_wgrad (no copy when the caller
                   // produced it in its stash slot already, include/empose_hip.h)
      float* slot = stash + (size_t)M * (L - 1) * H;
      if (d_out != slot || ld_dout != op) {
        e = launch_axpby2d(M, op, 1.f, d_out, ld_dout, 0.f, nullptr, 0, slot, op, stream);
        if (e != hipSuccess) return fail(EMPOSE_EHIP, "stash: %s", hipGetErrorString(e));
      }
    } else {
      AtbArgs ab{};
      ab.A = d_out; ab.lda = ld_dout; ab.B = layer_save(l - 1) + (size_t)M * H; ab.ldb = H; ab.C = gr->weight[l]; ab.ldc = H;
      ab.bias = gr->bias[l]; ab.M = M; ab.N = p->out_dim; ab.K = H; ab.accumulate = accumulate;
      e = launch_gemm_atb(ab, w.atb, w.atb_floats, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "dW: %s", hipGetErrorString(e));
    }
    const float* wt = p->weight_t[l];
    if (!wt) {
      HIP_TRY(hipMemsetAsync(w.wt, 0, (size_t)H * op * sizeof(float), stream));
      e = launch_transpose(p->weight[l], H, w.wt, op, p->out_dim, H, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "transpose: %s", hipGetErrorString(e));
      wt = w.wt;
    }
    e = gemm(d_out, ld_dout, wt, op, w.d[0], H, H, op);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "dX gemm: %s", hipGetErrorString(e));
  }
  const int cur = 0;   // w.d[0]: cotangent of the current layer's activation; w.d[1]: dZ when it is not stashed
  for (int l = L - 2; l >= 0; --l) {
    const float* sv = layer_save(l);
    BnPreluArgs a{};
    a.M = M; a.C = H; a.x = sv; a.ldx = H; a.gamma = p->bn_weight[l]; a.beta = p->bn_bias[l]; a.slope = p->prelu[l];
    a.save_mean = const_cast<float*>(sv + (size_t)2 * M * H); a.save_rstd = a.save_mean + H;
    float* dz = stash ? stash + (size_t)M * l * H : w.d[cur ^ 1];   // dZ_l: into the stash when the dW are deferred
    a.dz = w.d[cur]; a.lddz = H; a.dx = dz; a.lddx = H;
    a.dgamma = gr->bn_weight[l]; a.dbeta = gr->bn_bias[l]; a.dslope = gr->prelu[l];
    a.dslope_partial = w.slope_partial; a.counter = w.counter; a.workspace = w.bn; a.accumulate = accumulate;
    hipError_t e = launch_bn_prelu(a, true, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "bn_prelu backward: %s", hipGetErrorString(e));
    const float* in = l == 0 ? x : layer_save(l - 1) + (size_t)M * H;
    const int ld_in = l == 0 ? ldx : H, k_in = l == 0 ? p->in_dim : H;
    if (!stash) {
      AtbArgs ab{};
      ab.A = dz; ab.lda = H; ab.B = in; ab.ldb = ld_in; ab.C = gr->weight[l]; ab.ldc = k_in;
      ab.bias = gr->bias[l]; ab.M = M; ab.N = H; ab.K = k_in; ab.accumulate = accumulate;
      e = launch_gemm_atb(ab, w.atb, w.atb_floats, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "dW: %s", hipGetErrorString(e));
    }
    if (l == 0) break;
    const float* wt = p->weight_t[l];
    if (!wt) {
      e = launch_transpose(p->weight[l], H, w.wt, H, H, H, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "transpose: %s", hipGetErrorString(e));
      wt = w.wt;
    }
    e = gemm(dz, H, wt, H, w.d[cur], H, H, H);   // the cotangent of the layer below overwrites the consumed one
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "dX gemm: %s", hipGetErrorString(e));
  }
  return EMPOSE_OK;
}
}  // namespace

int empose_mlp_train_bwd(const empose_mlp_params* p, int M, const float* x, int ldx, const float* d_out, int ld_dout,
                         const float* save, const empose_mlp_grads* gr, int accumulate, void* workspace,
                         size_t workspace_bytes, empose_stream_t stream) {
  return mlp_train_bwd_impl(p, M, x, ldx, d_out, ld_dout, save, gr, accumulate, nullptr, workspace, workspace_bytes, stream);
}

size_t empose_mlp_train_stash_floats(const empose_mlp_params* p, int M) {
  if (!p || M <= 0 || check_mlp_params(p) != EMPOSE_OK) return 0;
  return mlp_stash_floats(p, M);
}

int empose_mlp_train_bwd_deferred(const empose_mlp_params* p, int M, const float* x, int ldx, const float* d_out,
                                  int ld_dout, const float* save, const empose_mlp_grads* gr, int accumulate,
                                  float* dz_stash, void* workspace, size_t workspace_bytes, empose_stream_t stream) {
  if (!dz_stash) return fail(EMPOSE_EINVAL, "null stash");
  return mlp_train_bwd_impl(p, M, x, ldx, d_out, ld_dout, save, gr, accumulate, dz_stash, workspace, workspace_bytes, stream);
}

// ---- both update networks of an iteration in one call: paired launches on the one-launch layers, else one after the other
size_t empose_mlp_train_pair_workspace_bytes(const empose_mlp_params* p0, const empose_mlp_params* p1, int M) {
  const size_t a = empose_mlp_train_workspace_bytes(p0, M), b = empose_mlp_train_workspace_bytes(p1, M);
  return a > b ? a : b;
}

int empose_mlp_train_fwd_pair(const empose_mlp_params* p0, const empose_mlp_params* p1, int M, const float* x, int ldx,
                              float* out0, int ld_out0, float* out1, int ld_out1, float* save0, float* save1,
                              void* workspace, size_t workspace_bytes, empose_stream_t stream_) {
  TRY(check_mlp_params(p0));
  TRY(check_mlp_params(p1));
  TRY(earlier_poll_timeouts());
  if (workspace_bytes < empose_mlp_train_pair_workspace_bytes(p0, p1, M)) return fail(EMPOSE_ENOMEM, "workspace too small");
  if (M > 0 && x && out0 && out1 && save0 && save1 && workspace && ldx % 4 == 0 && ldx >= p0->in_dim && ldx >= p1->in_dim &&
      ld_out0 >= p0->out_dim && ld_out1 >= p1->out_dim && mlp_cols_pairable(p0, p1, M)) {
    Carver c(workspace);
    MlpTrainWs w = carve_mlp_train(c, p0, M);
    const empose_mlp_params* ps[2] = {p0, p1};
    float* outs[2] = {out0, out1};
    const int lds[2] = {ld_out0, ld_out1};
    float* saves[2] = {save0, save1};
    return mlp_fwd_cols(ps, 2, M, x, ldx, outs, lds, saves, w, static_cast<hipStream_t>(stream_));
  }
  TRY(empose_mlp_train_fwd(p0, M, x, ldx, out0, ld_out0, save0, workspace, workspace_bytes, stream_));
  return empose_mlp_train_fwd(p1, M, x, ldx, out1, ld_out1, save1, workspace, workspace_bytes, stream_);
}

int empose_mlp_train_bwd_deferred_pair(const empose_mlp_params* p0, const empose_mlp_params* p1, int M, const float* x,
                                       int ldx, const float* d_out0, int ld_dout0, const float* d_out1, int ld_dout1,
                                       const float* save0, const float* save1, const empose_mlp_grads* gr0,
                                       const empose_mlp_grads* gr1, int accumulate, float* dz_stash0, float* dz_stash1,
                                       void* workspace, size_t workspace_bytes, empose_stream_t stream_) {
  TRY(check_mlp_params(p0));
  TRY(check_mlp_params(p1));
  TRY(earlier_poll_timeouts());
  if (!dz_stash0 || !dz_stash1) return fail(EMPOSE_EINVAL, "null stash");
  if (workspace_bytes < empose_mlp_train_pair_workspace_bytes(p0, p1, M)) return fail(EMPOSE_ENOMEM, "workspace too small");
  bool pair = M > 0 && x && d_out0 && d_out1 && save0 && save1 && gr0 && gr1 && workspace && mlp_cols_pairable(p0, p1, M) &&
              ld_dout0 % 4 == 0 && ld_dout1 % 4 == 0 && ld_dout0 >= ((p0->out_dim + 3) & ~3) && ld_dout1 >= ((p1->out_dim + 3) & ~3);
  for (int l = 0; l < p0->n_layers - 1 && pair; ++l)
    if (!gr0->bn_weight[l] || !gr0->bn_bias[l] || !gr0->prelu[l] || !gr1->bn_weight[l] || !gr1->bn_bias[l] || !gr1->prelu[l])
      pair = false;
  if (pair) {
    Carver c(workspace);
    MlpTrainWs w = carve_mlp_train(c, p0, M);
    const empose_mlp_params* ps[2] = {p0, p1};
    const float* d_outs[2] = {d_out0, d_out1};
    const int lds[2] = {ld_dout0, ld_dout1};
    const float* saves[2] = {save0, save1};
    const empose_mlp_grads* grs[2] = {gr0, gr1};
    float* stashes[2] = {dz_stash0, dz_stash1};
    return mlp_bwd_cols(ps, 2, M, x, ldx, d_outs, lds, saves, grs, accumulate, stashes, w, static_cast<hipStream_t>(stream_));
  }
  TRY(empose_mlp_train_bwd_deferred(p0, M, x, ldx, d_out0, ld_dout0, save0, gr0, accumulate, dz_stash0, workspace,
                                    workspace_bytes, stream_));
  return empose_mlp_train_bwd_deferred(p1, M, x, ldx, d_out1, ld_dout1, save1, gr1, accumulate, dz_stash1, workspace,
                                       workspace_bytes, stream_);
}

size_t empose_mlp_train_wgrad_workspace_bytes(const empose_mlp_params* p, int n_app, int M) {
  if (!p || M <= 0 || n_app <= 0 || check_mlp_params(p) != EMPOSE_OK) return 0;
  // batched: one product over n_app * M rows; row counts off the 32-row grid run per application (M rows each)
  const size_t a = mlp_atb_floats(p, n_app * M), b = mlp_atb_floats(p, M);
  return ((a > b ? a : b) + 64) * sizeof(float);
}

int empose_mlp_train_wgrad(const empose_mlp_params* p, int n_app, int M, const float* const* x, int ldx,
                           const float* const* save, const float* const* dz_stash, const empose_mlp_grads* gr,
                           int accumulate, void* workspace, size_t workspace_bytes, empose_stream_t stream_) {
  TRY(check_mlp_params(p));
  TRY(earlier_poll_timeouts());
  if (!x || !save || !dz_stash || !gr || !workspace) return fail(EMPOSE_EINVAL, "null argument");
  if (n_app < 1 || n_app > ATB_MAX_SEG || M <= 0 || ldx < p->in_dim) return fail(EMPOSE_EINVAL, "bad sizes");
  if (workspace_bytes < empose_mlp_train_wgrad_workspace_bytes(p, n_app, M)) return fail(EMPOSE_ENOMEM, "workspace too small");
  const int H = p->hidden, L = p->n_layers, op = (p->out_dim + 3) & ~3;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  float* ws = static_cast<float*>(workspace);
  const size_t ws_floats = workspace_bytes / sizeof(float);
  // one product over all applications when their rows can be addressed as 32-row aligned segments, else one per application
  bool batched = M % 32 == 0 && ldx % 4 == 0;
  for (int s = 0; s < n_app && batched; ++s)
    batched = x[s] && save[s] && dz_stash[s] && ((uintptr_t)x[s] & 15) == 0 && ((uintptr_t)save[s] & 15) == 0 &&
              ((uintptr_t)dz_stash[s] & 15) == 0;
  for (int s = 0; s < n_app; ++s)
    if (!x[s] || !save[s] || !dz_stash[s]) return fail(EMPOSE_EINVAL, "null argument");
  const bool fused = mlp_train_fused(p, M);
  for (int l = 0; l < L; ++l) {
    if (!gr->weight[l] || !gr->bias[l]) return fail(EMPOSE_EINVAL, "null gradient output");
    const bool last = l == L - 1;
    const int ld_a = last ? op : H, n_out = last ? p->out_dim : H;
    const int ld_b = l == 0 ? ldx : H, k_in = l == 0 ? p->in_dim : H;
    if (fused) {
      // operands as the fused sweeps left them: dY_l (stash), y_{l-1} and its (s, t) (save)
      const size_t lsz = mlp_layer_save(p, M);
      AtbArgs ab{};
      ab.lda = ld_a; ab.ldb = ld_b; ab.C = gr->weight[l]; ab.ldc = k_in; ab.bias = gr->bias[l]; ab.N = n_out; ab.K = k_in;
      ab.b_mode = l > 0 ? 1 : 0; ab.b_slope = l > 0 ? p->prelu[l - 1] : nullptr;
      auto fill = [&](int slot, int s) {
        ab.A_seg[slot] = dz_stash[s] + (size_t)M * l * H;
        ab.B_seg[slot] = l == 0 ? x[s] : save[s] + (size_t)(l - 1) * lsz;
        ab.Bs_seg[slot] = l > 0 ? save[s] + (size_t)(l - 1) * lsz + (size_t)M * H + 2 * H : nullptr;
      };
      if (batched) {
        for (int s = 0; s < n_app; ++s) fill(s, s);
        ab.A = ab.A_seg[0]; ab.B = ab.B_seg[0]; ab.M = n_app * M; ab.accumulate = accumulate;
        ab.n_seg = n_app; ab.seg_rows = M;
        hipError_t e = launch_gemm_atb(ab, ws, ws_floats, stream);
        if (e != hipSuccess) return fail(EMPOSE_EHIP, "fused dW: %s", hipGetErrorString(e));
      } else {
        for (int s = 0; s < n_app; ++s) {
          fill(0, s);
          ab.A = ab.A_seg[0]; ab.B = ab.B_seg[0]; ab.M = M; ab.accumulate = accumulate || s > 0;
          hipError_t e = launch_gemm_atb(ab, ws, ws_floats, stream);
          if (e != hipSuccess) return fail(EMPOSE_EHIP, "fused dW: %s", hipGetErrorString(e));
        }
      }
      continue;
    }
    auto a_of = [&](int s) { return dz_stash[s] + (size_t)M * l * H; };
    auto b_of = [&](int s) { return l == 0 ? x[s] : save[s] + (size_t)(l - 1) * mlp_layer_save(p, M) + (size_t)M * H; };
    AtbArgs ab{};
    ab.lda = ld_a; ab.ldb = ld_b; ab.C = gr->weight[l]; ab.ldc = k_in; ab.bias = gr->bias[l]; ab.N = n_out; ab.K = k_in;
    if (batched) {
      ab.A = a_of(0); ab.B = b_of(0); ab.M = n_app * M; ab.accumulate = accumulate;
      ab.n_seg = n_app; ab.seg_rows = M;
      for (int s = 0; s < n_app; ++s) { ab.A_seg[s] = a_of(s); ab.B_seg[s] = b_of(s); }
      hipError_t e = launch_gemm_atb(ab, ws, ws_floats, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "dW: %s", hipGetErrorString(e));
    } else {
      for (int s = 0; s < n_app; ++s) {
        ab.A = a_of(s); ab.B = b_of(s); ab.M = M; ab.accumulate = accumulate || s > 0;
        hipError_t e = launch_gemm_atb(ab, ws, ws_floats, stream);
        if (e != hipSuccess) return fail(EMPOSE_EHIP, "dW: %s", hipGetErrorString(e));
      }
    }
  }
  return EMPOSE_OK;
}

size_t empose_lstm_train_save_floats(int L, int B, int F, int H) {
  return (size_t)L * B * F * 7 * H;
}

namespace {
struct TrainLstmWs {
  LstmWs st;               // h[l][2], c[l]
  float* bias[4];          // b_ih + b_hh
  float* dgates;           // [B*F][4H]
  float* dyl;              // [B*F][H] cotangent of the layer below's output
  float* dh[2]; float* carry; float* dc;   // [B][H]
  float* wt;               // transposed weights, max(H, in) x 4H
  float* atb;              // A^T B partials
  size_t atb_floats;
  float* ksplit;           // partial tiles of the K-split recurrent product, or nullptr
  size_t ksplit_floats;
  // the reverse recurrences of two layers as a wavefront (bptt_wave): the upper layer's pre-activation gradients, cell
  // state cotangent and carry next to the lower layer's, three transposed weight matrices at once
  int wave = 0;            // 0 no, 1 matrix-vector kernel, 2 K-split tiles
  float* dgates_up; float* carry_up; float* dc_up;
  float* wt_hh_up; float* wt_ih_up;
  float* rec;              // partial tiles of both problems of a wavefront step (K-split form)
};
int check_lstm_params(const empose_lstm_params* p) {
  if (!p) return fail(EMPOSE_EINVAL, "null argument");
  if (p->num_layers < 1 || p->num_layers > 4 || p->hidden_size % 4 != 0 || p->input_size % 4 != 0 ||
      p->hidden_size <= 0 || p->input_size <= 0)
    return fail(EMPOSE_EINVAL, "unsupported LSTM configuration");
  for (int l = 0; l < p->num_layers; ++l)
    if (!p->w_ih[l] || !p->w_hh[l] || !p->b_ih[l] || !p->b_hh[l]) return fail(EMPOSE_EINVAL, "null LSTM parameter");
  return EMPOSE_OK;
}
Lstm lstm_view(const empose_lstm_params* p) {   // device pointers as they are: nothing is uploaded
  Lstm r;
  r.num_layers = p->num_layers; r.input_size = p->input_size; r.H = p->hidden_size; r.dirs = 1;
  return r;
}
TrainLstmWs carve_train_lstm(Carver& c, const empose_lstm_params* p, int B, int F) {
  TrainLstmWs w;
  const int H = p->hidden_size, L = p->num_layers;
  const int in_max = p->input_size > H ? p->input_size : H;
  Lstm r = lstm_view(p);
  w.st = carve_lstm_of(c, r, B, F);   // B > LSTM_PERSIST_B or not, the exchange buffer is unused here
  for (int l = 0; l < 4; ++l) w.bias[l] = l < L ? c.f((size_t)4 * H) : nullptr;
  w.dgates = c.f((size_t)B * F * 4 * H);
  w.dyl = c.f((size_t)B * F * H);
  w.dh[0] = c.f((size_t)B * H); w.dh[1] = c.f((size_t)B * H); w.carry = c.f((size_t)B * H); w.dc = c.f((size_t)B * H);
  w.wt = c.f((size_t)in_max * 4 * H);
  w.atb_floats = atb_workspace_floats_max(B * F, {{4 * H, p->input_size}, {4 * H, H}});   // dW_ih (layer 0 / above), dW_hh
  w.atb = c.f(w.atb_floats + 64);
  w.ksplit_floats = gemm_ksplit_applicable(B, H, 4 * H) ? gemm_ksplit_workspace_floats(B, H, 4 * H) : 0;
  w.ksplit = w.ksplit_floats ? c.f(w.ksplit_floats) : nullptr;
  w.wave = 0;
  w.dgates_up = w.carry_up = w.dc_up = w.wt_hh_up = w.wt_ih_up = w.rec = nullptr;
  if (L == 2 && options().bptt_wave != 0 && (4 * H) % 256 == 0) {
    if (gemm_fewrows_applicable(B, H, 4 * H) && 4 * H <= 2048) w.wave = 1;
    else if (w.ksplit_floats && 8 * H / 256 <= 16) w.wave = 2;   // (not the pointer: null while sizes are counted)
  }
  if (w.wave) {
    w.dgates_up = c.f((size_t)B * F * 4 * H);
    w.carry_up = c.f((size_t)B * H); w.dc_up = c.f((size_t)B * H);
    w.wt_hh_up = c.f((size_t)H * 4 * H); w.wt_ih_up = c.f((size_t)H * 4 * H);
    if (w.wave == 2) w.rec = c.f(rec_ksplit_workspace_floats(B, H, 8 * H, 2));
  }
  return w;
}
}  // namespace

size_t empose_lstm_train_workspace_bytes(const empose_lstm_params* p, int B, int F) {
  if (!p || B <= 0 || F <= 0) return 0;
  Carver c(nullptr);
  carve_train_lstm(c, p, B, F);
  return c.off;
}

int empose_lstm_train_fwd(const empose_lstm_params* p, int B, int F, const float* x, int ldx, const int* seq_lengths,
                          const float* h0, const float* c0, float* y, float* h_n, float* c_n, float* save,
                          void* workspace, size_t workspace_bytes, empose_stream_t stream_) {
  TRY(check_lstm_params(p));
  TRY(earlier_poll_timeouts());
  if (!x || !y || !save || !workspace) return fail(EMPOSE_EINVAL, "null argument");
  if (B <= 0 || F <= 0 || ldx < p->input_size || ldx % 4 != 0) return fail(EMPOSE_EINVAL, "bad sizes");
  if (workspace_bytes < empose_lstm_train_workspace_bytes(p, B, F)) return fail(EMPOSE_ENOMEM, "workspace too small");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int H = p->hidden_size, L = p->num_layers;
  const size_t bh = (size_t)B * H, bfh = (size_t)B * F * H;
  if ((size_t)B * F * (size_t)(ldx > 4 * H ? ldx : 4 * H) * sizeof(float) >= ((size_t)1 << 32))
    return fail(EMPOSE_EINVAL, "LSTM batch of %d x %d frames is too large for one call; split the batch", B, F);
  Carver c(workspace);
  TrainLstmWs w = carve_train_lstm(c, p, B, F);
  LstmWaveArgs a;
  a.seq_lengths = seq_lengths; a.B = B; a.F = F; a.H = H; a.n_units = L;
  for (int l = 0; l < L; ++l) {
    hipError_t e = launch_add2(p->b_ih[l], p->b_hh[l], w.bias[l], 4 * H, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "bias sum: %s", hipGetErrorString(e));
    float* sv = save + (size_t)l * 7 * bfh;
    LstmUnitArgs& ua = a.unit[l];
    ua.w_ih = p->w_ih[l]; ua.w_hh = p->w_hh[l]; ua.bias = w.bias[l];
    ua.h[0] = w.st.h[l][0]; ua.h[1] = w.st.h[l][1]; ua.c = w.st.c[l];
    ua.in_k = l == 0 ? p->input_size : H;
    ua.in_seq = l == 0 ? x : nullptr; ua.in_ld = l == 0 ? ldx : 0; ua.in_from = l == 0 ? -1 : l - 1;
    ua.t_offset = l; ua.reverse = 0;
    ua.y = l == L - 1 ? y : sv + 6 * bfh; ua.y_ld = H; ua.y_col = 0;
    ua.sv_gates = sv; ua.sv_c = sv + 4 * bfh; ua.sv_hprev = sv + 5 * bfh;
    if (h0) {
      HIP_TRY(hipMemcpyAsync(ua.h[0], h0 + l * bh, bh * sizeof(float), hipMemcpyDeviceToDevice, stream));
      HIP_TRY(hipMemcpy2DAsync(ua.sv_hprev, (size_t)F * H * sizeof(float), h0 + l * bh, (size_t)H * sizeof(float),
                               (size_t)H * sizeof(float), B, hipMemcpyDeviceToDevice, stream));
    } else {
      HIP_TRY(hipMemsetAsync(ua.h[0], 0, bh * sizeof(float), stream));
      HIP_TRY(hipMemset2DAsync(ua.sv_hprev, (size_t)F * H * sizeof(float), 0, (size_t)H * sizeof(float), B, stream));
    }
    if (c0) HIP_TRY(hipMemcpyAsync(ua.c, c0 + l * bh, bh * sizeof(float), hipMemcpyDeviceToDevice, stream));
    else HIP_TRY(hipMemsetAsync(ua.c, 0, bh * sizeof(float), stream));
  }
  // Small batches (the reference's training batch of 12 windows): the whole sequence in one cooperative launch with the
  // weights in registers, as in inference -- the step-by-step kernel streams 13.8 MB of weights per wavefront step.
  bool done = false;
  if (w.st.xch && F >= 4 && options().lstm_persist != 0) {
    a.s = 0;
    hipError_t e = launch_lstm_persist(a, w.st.xch, stream, &done);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "lstm sequence kernel: %s", hipGetErrorString(e));
  }
  for (int s = 0; !done && s < F + L - 1; ++s) {
    a.s = s;
    hipError_t e = launch_lstm_wave(a, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "lstm step: %s", hipGetErrorString(e));
  }
  for (int l = 0; l < L; ++l) {
    if (h_n) HIP_TRY(hipMemcpyAsync(h_n + l * bh, w.st.h[l][F & 1], bh * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (c_n) HIP_TRY(hipMemcpyAsync(c_n + l * bh, w.st.c[l], bh * sizeof(float), hipMemcpyDeviceToDevice, stream));
  }
  return EMPOSE_OK;
}

int empose_lstm_train_bwd(const empose_lstm_params* p, int B, int F, const float* x, int ldx, const int* seq_lengths,
                          const float* c0, const float* save, const float* dy, float* dx,
                          const empose_lstm_grads* grads, void* workspace, size_t workspace_bytes,
                          empose_stream_t stream_) {
  TRY(check_lstm_params(p));
  TRY(earlier_poll_timeouts());
  if (!x || !save || !dy || !grads || !workspace) return fail(EMPOSE_EINVAL, "null argument");
  if (B <= 0 || F <= 0 || ldx < p->input_size || ldx % 4 != 0) return fail(EMPOSE_EINVAL, "bad sizes");
  if (workspace_bytes < empose_lstm_train_workspace_bytes(p, B, F)) return fail(EMPOSE_ENOMEM, "workspace too small");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int H = p->hidden_size, L = p->num_layers;
  const size_t bh = (size_t)B * H, bfh = (size_t)B * F * H;
  for (int l = 0; l < L; ++l)
    if (!grads->w_ih[l] || !grads->w_hh[l] || !grads->b_ih[l] || !grads->b_hh[l])
      return fail(EMPOSE_EINVAL, "null gradient output");
  Carver c(workspace);
  TrainLstmWs w = carve_train_lstm(c, p, B, F);
  auto gemm = [&](const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                  const float* resid, int ldr) -> hipError_t {
    GemmBatch b;
    b.count = 1;
    GemmProb& g = b.p[0];
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.scale = nullptr; g.shift = nullptr; g.resid = resid; g.ldr = ldr; g.act = 0; g.slope = 0.f;
    return launch_gemm(b, stream);
  };
  if (w.wave) {
    // ---- two layers as a wavefront: stage s runs the cell of (layer 1, step s) and of (layer 0, step s + 1).  Both
    // need only what stage s + 1 left: dh1_s = dG1_{s+1} . W_hh1, and dh0_{s+1} = dG0_{s+2} . W_hh0 + dG1_{s+1} . W_ih1
    // -- the cotangent of layer 0's output, which the layer-after-layer form gets from one batched product over all
    // steps afterwards, is the second K segment of layer 0's recurrent product here.  F + 1 stages of one launch (or
    // one launch pair) instead of 2 F; the batched dX product of layer 1 disappears.
    hipError_t e = launch_transpose(p->w_hh[1], H, w.wt_hh_up, 4 * H, 4 * H, H, stream);
    if (e == hipSuccess) e = launch_transpose(p->w_ih[1], H, w.wt_ih_up, 4 * H, 4 * H, H, stream);
    if (e == hipSuccess) e = launch_transpose(p->w_hh[0], H, w.wt, 4 * H, 4 * H, H, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "transpose: %s", hipGetErrorString(e));
    HIP_TRY(hipMemsetAsync(w.dc, 0, bh * sizeof(float), stream));
    HIP_TRY(hipMemsetAsync(w.dc_up, 0, bh * sizeof(float), stream));
    auto cell_of = [&](int l, int t) {
      const float* sv = save + (size_t)l * 7 * bfh;
      LstmCellBwdArgs ca;
      ca.gates = sv; ca.c_all = sv + 4 * bfh; ca.c0 = c0 ? c0 + l * bh : nullptr;
      ca.dy = l == 1 ? dy : nullptr; ca.ld_dy = H; ca.dh_in = nullptr;
      ca.dc = l == 1 ? w.dc_up : w.dc; ca.dgates = l == 1 ? w.dgates_up : w.dgates;
      ca.dh_carry = l == 1 ? w.carry_up : w.carry;
      ca.seq_lengths = seq_lengths; ca.B = B; ca.F = F; ca.H = H; ca.t = t;
      return ca;
    };
    e = launch_lstm_cell_bwd(cell_of(1, F - 1), stream);   // stage F - 1: nothing flows into the last step of the top layer
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "lstm cell backward: %s", hipGetErrorString(e));
    for (int s = F - 2; s >= -1; --s) {
      RecBatch rb;
      LstmCellBwdArgs cells[2];
      rb.count = 0;
      if (s >= 0) {   // (layer 1, step s)
        RecProb& q = rb.p[rb.count];
        q.nseg = 1; q.seg[0] = RecSeg{w.dgates_up + (size_t)(s + 1) * 4 * H, F * 4 * H, w.wt_hh_up, 4 * H, 4 * H};
        q.M = B; q.N = H; q.resid = w.carry_up; q.ldr = H;
        cells[rb.count++] = cell_of(1, s);
      }
      {               // (layer 0, step s + 1)
        const int t0 = s + 1;
        RecProb& q = rb.p[rb.count];
        q.nseg = 0;
        if (t0 + 1 <= F - 1) q.seg[q.nseg++] = RecSeg{w.dgates + (size_t)(t0 + 1) * 4 * H, F * 4 * H, w.wt, 4 * H, 4 * H};
        q.seg[q.nseg++] = RecSeg{w.dgates_up + (size_t)t0 * 4 * H, F * 4 * H, w.wt_ih_up, 4 * H, 4 * H};
        q.M = B; q.N = H; q.resid = t0 + 1 <= F - 1 ? w.carry : nullptr; q.ldr = H;
        cells[rb.count++] = cell_of(0, t0);
      }
      e = w.wave == 2 ? launch_rec_ksplit(rb, cells, w.rec, stream) : launch_rec_fewrows(rb, cells, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "recurrent backward (wavefront): %s", hipGetErrorString(e));
    }
    // ---- the cotangents of the initial state, where asked for: dh_{-1} = dG_0 . W_hh + carry, dc_{-1} = what the cell of
    // step 0 left in the running cell cotangent
    for (int l = 0; l < 2; ++l) {
      if (grads->d_h0[l]) {
        e = gemm(l == 1 ? w.dgates_up : w.dgates, F * 4 * H, l == 1 ? w.wt_hh_up : w.wt, 4 * H, grads->d_h0[l], H, B, H, 4 * H,
                 l == 1 ? w.carry_up : w.carry, H);
        if (e != hipSuccess) return fail(EMPOSE_EHIP, "initial-state cotangent: %s", hipGetErrorString(e));
      }
      if (grads->d_c0[l])
        HIP_TRY(hipMemcpyAsync(grads->d_c0[l], l == 1 ? w.dc_up : w.dc, bh * sizeof(float), hipMemcpyDeviceToDevice, stream));
    }
    // ---- the batched products: weight gradients of both layers, the input cotangent of layer 0
    for (int l = 1; l >= 0; --l) {
      const float* sv = save + (size_t)l * 7 * bfh;
      const int in_l = l == 0 ? p->input_size : H;
      const float* x_l = l == 0 ? x : save + 6 * bfh;
      const int ldx_l = l == 0 ? ldx : H;
      float* dg = l == 1 ? w.dgates_up : w.dgates;
      AtbArgs ab{};
      ab.A = dg; ab.lda = 4 * H; ab.B = x_l; ab.ldb = ldx_l; ab.C = grads->w_ih[l]; ab.ldc = in_l;
      ab.bias = grads->b_ih[l]; ab.M = B * F; ab.N = 4 * H; ab.K = in_l;
      e = launch_gemm_atb(ab, w.atb, w.atb_floats, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "dW_ih: %s", hipGetErrorString(e));
      HIP_TRY(hipMemcpyAsync(grads->b_hh[l], grads->b_ih[l], (size_t)4 * H * sizeof(float), hipMemcpyDeviceToDevice, stream));
      ab.B = sv + 5 * bfh; ab.ldb = H; ab.C = grads->w_hh[l]; ab.ldc = H; ab.bias = nullptr; ab.K = H;
      e = launch_gemm_atb(ab, w.atb, w.atb_floats, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "dW_hh: %s", hipGetErrorString(e));
    }
    if (dx) {
      e = launch_transpose(p->w_ih[0], p->input_size, w.wt, 4 * H, 4 * H, p->input_size, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "transpose: %s", hipGetErrorString(e));
      e = gemm(w.dgates, 4 * H, w.wt, 4 * H, dx, p->input_size, B * F, p->input_size, 4 * H, nullptr, 0);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "dX gemm: %s", hipGetErrorString(e));
    }
    return EMPOSE_OK;
  }
  for (int l = L - 1; l >= 0; --l) {
    const float* sv = save + (size_t)l * 7 * bfh;
    const int in_l = l == 0 ? p->input_size : H;
    const float* x_l = l == 0 ? x : save + (size_t)(l - 1) * 7 * bfh + 6 * bfh;
    const int ldx_l = l == 0 ? ldx : H;
    const float* dy_l = l == L - 1 ? dy : w.dyl;
    // W_hh^T for the recurrent product dh_{t-1} = dG_t . W_hh on the forward GEMM kernel
    hipError_t e = launch_transpose(p->w_hh[l], H, w.wt, 4 * H, 4 * H, H, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "transpose: %s", hipGetErrorString(e));
    HIP_TRY(hipMemsetAsync(w.dc, 0, bh * sizeof(float), stream));
    const float* dh_in = nullptr;
    auto cell_args = [&](int t, const float* dh) {
      LstmCellBwdArgs ca;
      ca.gates = sv; ca.c_all = sv + 4 * bfh; ca.c0 = c0 ? c0 + l * bh : nullptr;
      ca.dy = dy_l; ca.ld_dy = H; ca.dh_in = dh; ca.dc = w.dc; ca.dgates = w.dgates; ca.dh_carry = w.carry;
      ca.seq_lengths = seq_lengths; ca.B = B; ca.F = F; ca.H = H; ca.t = t;
      return ca;
    };
    bool cell_done = false;   // the cell of step t already ran inside the previous step's reduce kernel
    for (int t = F - 1; t >= 0; --t) {
      if (!cell_done) {
        LstmCellBwdArgs ca = cell_args(t, dh_in);
        e = launch_lstm_cell_bwd(ca, stream);
        if (e != hipSuccess) return fail(EMPOSE_EHIP, "lstm cell backward: %s", hipGetErrorString(e));
      }
      cell_done = false;
      if (t == 0) break;
      float* out = w.dh[t & 1];
      if (w.ksplit) {   // a few hundred rows: K split over the workgroups (gemm_ksplit_kernel); its reduce kernel
        GemmProb g;     // feeds dh straight into the cell of step t - 1 (the carry of step t is the residual)
        g.A = w.dgates + (size_t)t * 4 * H; g.lda = F * 4 * H; g.W = w.wt; g.ldw = 4 * H; g.C = out; g.ldc = H;
        g.M = B; g.N = H; g.K = 4 * H; g.scale = nullptr; g.shift = nullptr; g.resid = w.carry; g.ldr = H; g.act = 0;
        g.slope = 0.f;
        LstmCellBwdArgs cn = cell_args(t - 1, nullptr);
        e = launch_gemm_ksplit(g, w.ksplit, stream, &cn);
        cell_done = true;
      } else if (gemm_fewrows_applicable(B, H, 4 * H)) {   // the reference's batch: matrix-vector kernel, same fusion
        GemmProb g;
        g.A = w.dgates + (size_t)t * 4 * H; g.lda = F * 4 * H; g.W = w.wt; g.ldw = 4 * H; g.C = out; g.ldc = H;
        g.M = B; g.N = H; g.K = 4 * H; g.scale = nullptr; g.shift = nullptr; g.resid = w.carry; g.ldr = H; g.act = 0;
        g.slope = 0.f;
        e = launch_gemm_fewrows_cell(g, cell_args(t - 1, nullptr), stream);
        cell_done = true;
      } else
      e = gemm(w.dgates + (size_t)t * 4 * H, F * 4 * H, w.wt, 4 * H, out, H, B, H, 4 * H, w.carry, H);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "recurrent backward gemm: %s", hipGetErrorString(e));
      dh_in = out;
    }
    // the cotangents of this layer's initial state, where asked for (w.wt still holds W_hh^T, w.carry / w.dc what the cell
    // of step 0 left)
    if (grads->d_h0[l]) {
      e = gemm(w.dgates, F * 4 * H, w.wt, 4 * H, grads->d_h0[l], H, B, H, 4 * H, w.carry, H);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "initial-state cotangent: %s", hipGetErrorString(e));
    }
    if (grads->d_c0[l])
      HIP_TRY(hipMemcpyAsync(grads->d_c0[l], w.dc, bh * sizeof(float), hipMemcpyDeviceToDevice, stream));
    AtbArgs ab{};
    ab.A = w.dgates; ab.lda = 4 * H; ab.B = x_l; ab.ldb = ldx_l; ab.C = grads->w_ih[l]; ab.ldc = in_l;
    ab.bias = grads->b_ih[l]; ab.M = B * F; ab.N = 4 * H; ab.K = in_l;
    e = launch_gemm_atb(ab, w.atb, w.atb_floats, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "dW_ih: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(grads->b_hh[l], grads->b_ih[l], (size_t)4 * H * sizeof(float), hipMemcpyDeviceToDevice, stream));
    ab.B = sv + 5 * bfh; ab.ldb = H; ab.C = grads->w_hh[l]; ab.ldc = H; ab.bias = nullptr; ab.K = H;
    e = launch_gemm_atb(ab, w.atb, w.atb_floats, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "dW_hh: %s", hipGetErrorString(e));
    float* dx_l = l > 0 ? w.dyl : dx;
    if (dx_l) {
      e = launch_transpose(p->w_ih[l], in_l, w.wt, 4 * H, 4 * H, in_l, stream);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "transpose: %s", hipGetErrorString(e));
      e = gemm(w.dgates, 4 * H, w.wt, 4 * H, dx_l, in_l, B * F, in_l, 4 * H, nullptr, 0);
      if (e != hipSuccess) return fail(EMPOSE_EHIP, "dX gemm: %s", hipGetErrorString(e));
    }
  }
  return EMPOSE_OK;
}

int empose_linear_f32(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                      const float* scale, const float* shift, int prelu, float slope, empose_stream_t stream_) {
  if (!A || !W || !C) return fail(EMPOSE_EINVAL, "null argument");
  if (K % 4 != 0 || lda % 4 != 0 || ldw % 4 != 0) return fail(EMPOSE_EINVAL, "K, lda, ldw must be multiples of 4");
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return fail(EMPOSE_EINVAL, "A and W must be 16-byte aligned");
  GemmBatch b;
  b.count = 1;
  GemmProb& p = b.p[0];
  p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.scale = scale; p.shift = shift; p.resid = nullptr; p.ldr = 0; p.act = prelu ? 1 : 0; p.slope = slope;
  hipError_t e = launch_gemm(b, static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "gemm launch: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

int empose_virtual_sensors_fwd(int T, int V, const float* vertices, int M, int max_deg, const int* center,
                               const int* helper, const int* deg, const int* faces, float* pos, float* ori,
                               float* normals, empose_stream_t stream_) {
  if (!vertices || !center || !helper || !deg || !faces || !pos || !ori) return fail(EMPOSE_EINVAL, "null argument");
  if (T <= 0 || V <= 0 || M <= 0 || max_deg <= 0) return fail(EMPOSE_EINVAL, "sizes must be positive");
  VirtualSensorArgs a;
  a.vertices = vertices; a.center = center; a.helper = helper; a.deg = deg; a.faces = faces;
  a.pos = pos; a.ori = ori; a.normals = normals; a.T = T; a.V = V; a.M = M; a.max_deg = max_deg;
  hipError_t e = launch_virtual_sensors(a, static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "virtual sensors kernel: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

int empose_metrics_rows(int T, const float* joints_gt, const float* joints_hat, const float* pose_gt,
                        const float* pose_hat, const int* parents_host, double* rows, empose_stream_t stream_) {
  if (!joints_gt || !joints_hat || !parents_host || !rows) return fail(EMPOSE_EINVAL, "null argument");
  if ((pose_gt == nullptr) != (pose_hat == nullptr)) return fail(EMPOSE_EINVAL, "pose_gt and pose_hat go together");
  if (T <= 0) return fail(EMPOSE_EINVAL, "T must be positive");
  MetricsArgs a;
  a.joints_gt = joints_gt; a.joints_hat = joints_hat; a.pose_gt = pose_gt; a.pose_hat = pose_hat; a.rows = rows; a.T = T;
  for (int j = 0; j < 22; ++j) {
    if (parents_host[j] >= j) return fail(EMPOSE_EINVAL, "parents must be topologically ordered");
    a.parents[j] = parents_host[j] < 0 ? 0 : parents_host[j];
  }
  hipError_t e = launch_metrics_rows(a, static_cast<hipStream_t>(stream_));
  if (e != hipSuccess) return fail(EMPOSE_EHIP, "metrics kernel: %s", hipGetErrorString(e));
  return EMPOSE_OK;
}

// ---- full mesh -----------------------------------------------------------------------------------------------
void empose_mesh_destroy(empose_mesh_t* mesh) {
  if (!mesh) return;
  for (void* p : mesh->allocs) (void)hipFree(p);
  delete mesh;
}

// Tables of mesh_rows_kernel: for every 32-vertex tile, k-group of 8 and coordinate c, lane (v = lane & 31,
// half = lane >> 5) owns wc[(tile * 32 + v) * 3 + c][kg * 8 + half * 4 .. + 3] -- the three coordinate planes of a tile
// are separate 32-column operands, so a lane's accumulators hold x, y and z of the same vertex.  Vertices past V are
// zero.  Skinning tables: the first four (bone, weight) pairs per vertex, zero-padded.
static int pack_mesh_tiles(empose_mesh* m, const empose_mesh_desc* d) {
  const int V = d->n_vertices, NT = (V + 31) / 32, KG = 25, K = 200;
  std::vector<float> buf((size_t)NT * KG * 3 * 256, 0.f);
  for (int t = 0; t < NT; ++t)
    for (int kg = 0; kg < KG; ++kg)
      for (int c = 0; c < 3; ++c)
        for (int lane = 0; lane < 64; ++lane) {
          const int v = t * 32 + (lane & 31);
          if (v >= V) continue;
          const float* src = d->wc + ((size_t)v * 3 + c) * K + kg * 8 + (lane >> 5) * 4;
          float* dst = &buf[((((size_t)t * KG + kg) * 3 + c) * 64 + lane) * 4];
          for (int e = 0; e < 4; ++e) dst[e] = src[e];
        }
  TRY(upload(m->allocs, buf.data(), buf.size(), &m->wc_frag));
  std::vector<int> idx4((size_t)NT * 32 * 4, 0);
  std::vector<float> w4((size_t)NT * 32 * 4, 0.f);
  for (int v = 0; v < V; ++v)
    for (int k = 0; k < 4 && k < d->kb; ++k) {
      idx4[(size_t)v * 4 + k] = d->skin_idx[(size_t)v * d->kb + k];
      w4[(size_t)v * 4 + k] = d->skin_w[(size_t)v * d->kb + k];
    }
  TRY(upload(m->allocs, idx4.data(), idx4.size(), &m->skin_idx4));
  TRY(upload(m->allocs, w4.data(), w4.size(), &m->skin_w4));
  return EMPOSE_OK;
}

// Tables of mesh_rows_bf16_kernel: per 32-vertex tile, k-step of 16, coordinate plane and (hi, lo) piece, lane
// (v = lane & 31, half = lane >> 5) owns the eight values k = kstep * 16 + half * 8 .. + 7 of row (tile * 32 + v) * 3 + c.
// Columns (mesh.hip): 0..188 pose, (hi, lo) = (w0, w1); 189..199 shape/template, (b0, b2); 200..210 the same
// coefficients again, (b1, b0); zero up to 223.
static unsigned short host_bf16_rne(float x) {
  uint32_t u;
  std::memcpy(&u, &x, 4);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float host_bf16_f32(unsigned short h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
// Tables of mesh_rows_x3_kernel (mesh_x3.hip): per 32-vertex tile, k-step of 16, coordinate plane c and piece p one
// fragment of 1 KB -- lane (v = lane & 31, half = lane >> 5) owns the eight values k = kstep * 16 + half * 8 .. + 7 of
// row (tile * 32 + v) * 3 + c, as piece p of their three-piece bf16 split (bf16x3.h); k >= 200 and vertices past V are zero.
static int pack_mesh_tiles_x3(empose_mesh* m, const empose_mesh_desc* d) {
  const int V = d->n_vertices, NT = (V + 31) / 32, K = 200, KS = 13;
  const size_t tile_shorts = MESH_X3_TILE_BYTES / 2;
  std::vector<unsigned short> buf((size_t)NT * tile_shorts, 0);
  for (int t = 0; t < NT; ++t)
    for (int lane = 0; lane < 64; ++lane) {
      const int v = t * 32 + (lane & 31), half = lane >> 5;
      if (v >= V) continue;
      for (int c = 0; c < 3; ++c) {
        const float* row = d->wc + ((size_t)v * 3 + c) * K;
        for (int ks = 0; ks < KS; ++ks)
          for (int e = 0; e < 8; ++e) {
            const int k = ks * 16 + half * 8 + e;
            if (k >= K) continue;
            const float x = row[k];
            const unsigned short p0 = host_bf16_rne(x);
            const float r1 = x - host_bf16_f32(p0);
            const unsigned short p1 = host_bf16_rne(r1);
            const unsigned short p2 = host_bf16_rne(r1 - host_bf16_f32(p1));
            unsigned short* dst = &buf[(size_t)t * tile_shorts + ((size_t)((ks * 3 + c) * 3) * 64 + lane) * 8 + e];
            dst[0] = p0; dst[64 * 8] = p1; dst[2 * 64 * 8] = p2;
          }
      }
    }
  return upload(m->allocs, buf.data(), buf.size(), &m->wc_x3);
}

static int pack_mesh_tiles_bf16(empose_mesh* m, const empose_mesh_desc* d) {
  const int V = d->n_vertices, NT = (V + 31) / 32, K = 200, KS = 14;
  const size_t tile_shorts = MESH_BF16_TILE_BYTES / 2;
  std::vector<unsigned short> buf((size_t)NT * tile_shorts, 0);
  for (int t = 0; t < NT; ++t)
    for (int lane = 0; lane < 64; ++lane) {
      const int v = t * 32 + (lane & 31), half = lane >> 5;
      if (v >= V) continue;
      for (int c = 0; c < 3; ++c) {
        const float* row = d->wc + ((size_t)v * 3 + c) * K;
        for (int ks = 0; ks < KS; ++ks)
          for (int e = 0; e < 8; ++e) {
            const int k = ks * 16 + half * 8 + e;
            if (k >= 211) continue;
            const float x = row[k < 200 ? k : k - 11];
            const unsigned short p0 = host_bf16_rne(x);
            const float r1 = x - host_bf16_f32(p0);
            const unsigned short p1 = host_bf16_rne(r1);
            const unsigned short p2 = host_bf16_rne(r1 - host_bf16_f32(p1));
            unsigned short* dst = &buf[(size_t)t * tile_shorts + ((size_t)((ks * 3 + c) * 2) * 64 + lane) * 8 + e];
            dst[0] = k < 200 ? p0 : p1;                          // hi: w0 | b0 | b1
            dst[64 * 8] = k < 189 ? p1 : (k < 200 ? p2 : p0);    // lo: w1 | b2 | b0
          }
      }
    }
  TRY(upload(m->allocs, buf.data(), buf.size(), &m->wc_bf16));
  // The skin weights as a dense [32 bone slots][32 vertices] block per tile for the bone blend on the matrix cores
  // (mesh.hip mesh_rows_bf16s_kernel): k-step ks, piece p, lane (vertex = lane & 31, half = lane >> 5) holds the eight
  // weights of bones 16 ks + 8 half + 0..7 -- hi = bf16(w), lo = bf16(w - hi); bones a vertex does not have, and the
  // slots past the 22 body bones, are zero.  Weights of the same bone listed twice add up.
  {
    const size_t tile = MESH_SKIN_BF16_TILE_BYTES / 2;
    std::vector<unsigned short> sk((size_t)NT * tile, 0);
    std::vector<float> dense(32);
    for (int t = 0; t < NT; ++t)
      for (int lane = 0; lane < 64; ++lane) {
        const int v = t * 32 + (lane & 31), half = lane >> 5;
        if (v >= V) continue;
        std::fill(dense.begin(), dense.end(), 0.f);
        for (int k = 0; k < d->kb; ++k) {
          const int b = d->skin_idx[(size_t)v * d->kb + k];
          if (b < 0 || b >= NB) return fail(EMPOSE_EINVAL, "skin index %d of vertex %d outside the %d body bones", b, v, NB);
          dense[b] += d->skin_w[(size_t)v * d->kb + k];
        }
        for (int ks = 0; ks < 2; ++ks)
          for (int e = 0; e < 8; ++e) {
            const float x = dense[ks * 16 + half * 8 + e];
            const unsigned short hi = host_bf16_rne(x);
            const unsigned short lo = host_bf16_rne(x - host_bf16_f32(hi));
            unsigned short* dst = &sk[(size_t)t * tile + ((size_t)(ks * 2) * 64 + lane) * 8 + e];
            dst[0] = hi;
            dst[64 * 8] = lo;
          }
      }
    TRY(upload(m->allocs, sk.data(), sk.size(), &m->skin_bf16));
  }
  return EMPOSE_OK;
}

int empose_mesh_create(const empose_mesh_desc* d, empose_mesh_t** out) {
  if (!d || !out) return fail(EMPOSE_EINVAL, "null argument");
  *out = nullptr;
  const int nj = d->n_joints == 0 ? 22 : d->n_joints;
  if (nj < 22 || nj > MESH_MAX_JOINTS) return fail(EMPOSE_EINVAL, "n_joints must be in [22, %d]", MESH_MAX_JOINTS);
  if (d->rodrigues != EMPOSE_RODRIGUES_SMPLX && d->rodrigues != EMPOSE_RODRIGUES_SO3)
    return fail(EMPOSE_EINVAL, "unknown Rodrigues convention %d", d->rodrigues);
  if (d->n_vertices <= 0 || d->ncp % 4 != 0 || d->j_off != d->n_vertices * 3 || d->j_off + nj * 3 > d->ncp ||
      d->ncp - d->j_off > nj * 3 + 3 || d->kb <= 0 || !d->parents)
    return fail(EMPOSE_EINVAL, "inconsistent mesh table sizes");
  for (int j = 0; j < nj; ++j)
    if (d->parents[j] >= j || (j > 0 && d->parents[j] < 0))
      return fail(EMPOSE_EINVAL, "parents must be topologically ordered with a single root");
  empose_mesh* m = new empose_mesh();
  m->V = d->n_vertices; m->j_off = d->j_off; m->ncp = d->ncp; m->kb = d->kb;
  m->n_joints = nj; m->rod_conv = d->rodrigues;
  int rc;
  if ((rc = upload(m->allocs, d->wc, (size_t)d->ncp * 200, &m->wc)) ||
      (rc = upload(m->allocs, d->skin_idx, (size_t)d->n_vertices * d->kb, &m->skin_idx)) ||
      (rc = upload(m->allocs, d->skin_w, (size_t)d->n_vertices * d->kb, &m->skin_w)) ||
      (rc = upload(m->allocs, d->parents, (size_t)nj, &m->parents)) || (rc = pack_mesh_tiles(m, d)) ||
      (d->kb <= 4 && (rc = pack_mesh_tiles_x3(m, d))) ||
      (d->with_bf16x3 && (rc = pack_mesh_tiles_bf16(m, d)))) {
    empose_mesh_destroy(m);
    return rc;
  }
  *out = m;
  return EMPOSE_OK;
}

int empose_mesh_n_joints(const empose_mesh_t* mesh) { return mesh ? mesh->n_joints : 0; }

static const int MESH_SLAB = 16384;  // frames per pass: bounds the scratch (rot, feat, rest joints, transforms)

struct MeshWs { float *rot, *feat, *jrest, *xf; };
static MeshWs carve_mesh(Carver& c, const empose_mesh* mesh, size_t S) {
  MeshWs w;
  w.rot = c.f(S * 198); w.feat = c.f(S * 200); w.jrest = c.f(S * (size_t)(mesh->ncp - mesh->j_off));
  w.xf = c.f(S * 264);
  return w;
}

size_t empose_mesh_workspace_bytes(const empose_mesh_t* mesh, int T) {
  if (!mesh || T <= 0) return 0;
  Carver c(nullptr);
  carve_mesh(c, mesh, T < MESH_SLAB ? T : MESH_SLAB);
  return c.off;
}

// Slab by slab: Rodrigues + feature row, rest joints (the joint rows of wc), kinematic chain (all n_joints posed
// joints + the 22 skinning transforms) and, when `vertices` is given, the full-mesh kernel.
static int run_mesh(const empose_mesh_t* mesh, int T, const float* poses, const float* betas, const float* trans,
                    float* vertices, float* joints, void* workspace, hipStream_t stream, bool bf16x3 = false) {
  const int S = T < MESH_SLAB ? T : MESH_SLAB;
  const int jw = mesh->ncp - mesh->j_off, nj = mesh->n_joints;
  Carver c(workspace);
  const MeshWs w = carve_mesh(c, mesh, (size_t)S);
  for (int t0 = 0; t0 < T; t0 += S) {
    const int n = (T - t0) < S ? (T - t0) : S;
    FeatArgs fa;   // plain evaluation: the caller's rows are read in place
    fa.theta = const_cast<float*>(poses) + (size_t)t0 * 66; fa.ld_theta = 66;
    fa.beta = const_cast<float*>(betas) + (size_t)t0 * 10; fa.ld_beta = 10;
    fa.d_theta = nullptr; fa.d_beta = nullptr; fa.theta_step = 0.f; fa.beta_keep = 1.f; fa.beta_step = 0.f;
    fa.shape_avg = 0; fa.rot = w.rot; fa.feat = w.feat;
    fa.out_theta = fa.out_beta = fa.out_theta2 = fa.out_beta2 = nullptr;
    fa.T = n; fa.F = 1; fa.rod_conv = mesh->rod_conv;
    hipError_t e = launch_update_feat(fa, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "update_feat kernel: %s", hipGetErrorString(e));
    GemmBatch b;
    b.count = 1;
    GemmProb& p = b.p[0];
    p.A = w.feat; p.lda = 200; p.W = mesh->wc + (size_t)mesh->j_off * 200; p.ldw = 200; p.C = w.jrest; p.ldc = jw;
    p.M = n; p.N = jw; p.K = 200;
    p.scale = nullptr; p.shift = nullptr; p.resid = nullptr; p.ldr = 0; p.act = 0; p.slope = 0.f;
    e = launch_gemm(b, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "rest-joint gemm: %s", hipGetErrorString(e));
    const float* tr = trans ? trans + (size_t)t0 * 3 : nullptr;
    MeshChainArgs ca;
    ca.rot = w.rot; ca.out = w.jrest; ca.ncp = jw; ca.j_off = 0; ca.parents = mesh->parents;
    ca.trans = tr; ca.xf = w.xf; ca.joints = joints + (size_t)t0 * nj * 3; ca.T = n; ca.n_joints = nj;
    e = launch_mesh_chain(ca, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "mesh chain: %s", hipGetErrorString(e));
    if (!vertices) continue;
    MeshSkinArgs sa;
    sa.feat = w.feat; sa.wc = mesh->wc; sa.xf = w.xf; sa.skin_idx = mesh->skin_idx; sa.skin_w = mesh->skin_w;
    sa.kb = mesh->kb; sa.trans = tr; sa.vertices = vertices + (size_t)t0 * mesh->V * 3; sa.T = n; sa.V = mesh->V;
    sa.wc_frag = mesh->wc_frag; sa.skin_idx4 = mesh->skin_idx4; sa.skin_w4 = mesh->skin_w4;
    sa.wc_bf16 = mesh->wc_bf16; sa.skin_bf16 = mesh->skin_bf16; sa.wc_x3 = mesh->wc_x3;
    // default: the three-piece bf16 contraction (fp32-equivalent); `bf16x3`: the explicitly selected two-piece variant
    if (bf16x3)
      e = options().mesh_skin_mfma && mesh->skin_bf16 ? launch_mesh_rows_bf16s(sa, stream) : launch_mesh_rows_bf16(sa, stream);
    else if (options().mesh_x3 != 0 && mesh->wc_x3 && mesh->kb <= 4) {
      sa.stagger = options().mesh_x3 == 1;
      e = launch_mesh_rows_x3(sa, options().mesh_x3 == 3, stream);
    }
    else
      e = launch_mesh_rows(sa, stream);
    if (e != hipSuccess) return fail(EMPOSE_EHIP, "fused mesh kernel: %s", hipGetErrorString(e));
  }
  return EMPOSE_OK;
}

int empose_mesh_vertices_fwd(const empose_mesh_t* mesh, int T, const float* poses, const float* betas,
                             const float* trans, float* vertices, float* joints, void* workspace,
                             size_t workspace_bytes, empose_stream_t stream_) {
  if (!mesh || !poses || !betas || !vertices || !joints || !workspace) return fail(EMPOSE_EINVAL, "null argument");
  if (T <= 0) return fail(EMPOSE_EINVAL, "T must be positive");
  if (workspace_bytes < empose_mesh_workspace_bytes(mesh, T)) return fail(EMPOSE_ENOMEM, "workspace too small");
  return run_mesh(mesh, T, poses, betas, trans, vertices, joints, workspace, static_cast<hipStream_t>(stream_));
}

int empose_mesh_vertices_fwd_bf16x3(const empose_mesh_t* mesh, int T, const float* poses, const float* betas,
                                    const float* trans, float* vertices, float* joints, void* workspace,
                                    size_t workspace_bytes, empose_stream_t stream_) {
  if (!mesh || !poses || !betas || !vertices || !joints || !workspace) return fail(EMPOSE_EINVAL, "null argument");
  if (!mesh->wc_bf16) return fail(EMPOSE_EINVAL, "the mesh handle was created without with_bf16x3");
  if (T <= 0) return fail(EMPOSE_EINVAL, "T must be positive");
  if (workspace_bytes < empose_mesh_workspace_bytes(mesh, T)) return fail(EMPOSE_ENOMEM, "workspace too small");
  return run_mesh(mesh, T, poses, betas, trans, vertices, joints, workspace, static_cast<hipStream_t>(stream_), true);
}

int empose_mesh_joints_fwd(const empose_mesh_t* mesh, int T, const float* poses, const float* betas,
                           const float* trans, float* joints, void* workspace, size_t workspace_bytes,
                           empose_stream_t stream_) {
  if (!mesh || !poses || !betas || !joints || !workspace) return fail(EMPOSE_EINVAL, "null argument");
  if (T <= 0) return fail(EMPOSE_EINVAL, "T must be positive");
  if (workspace_bytes < empose_mesh_workspace_bytes(mesh, T)) return fail(EMPOSE_ENOMEM, "workspace too small");
  return run_mesh(mesh, T, poses, betas, trans, nullptr, joints, workspace, static_cast<hipStream_t>(stream_));
}

}  // extern "C"
