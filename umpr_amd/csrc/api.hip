// extern "C" entry points of libumpr_hip.so (declared in include/umpr_hip.h) and the composite ops they launch.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <vector>

#include "../../include/umpr_hip.h"
#include "umpr_common.h"
#include "umpr_internal.h"

static thread_local char g_err[512] = "";
void umpr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- profiling registry -------------------------------------------------------------------------------------
namespace {
// UMPR_FC_SMALL=0: batch-sized-M products stay on the tiled GEMM (A/B runs)
const bool g_fc_small = umpr_env_on("UMPR_FC_SMALL");
// UMPR_MERGE_SMALL=0: linear_u / linear_i of the review merge on the fc kernels above (two launches forward, five backward)
const bool g_merge_small = umpr_env_on("UMPR_MERGE_SMALL");
struct ProfRec { hipEvent_t e0, e1; int family; double work; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<ProfRec> g_prof_pool;
double g_prof_ms[UMPR_K_COUNT] = {0}, g_prof_work[UMPR_K_COUNT] = {0};
long g_prof_n[UMPR_K_COUNT] = {0};
}  // namespace
UmprProfScope::UmprProfScope(int family, double work, hipStream_t s) : idx(-1), stream(s) {
  if (!g_prof_on) return;
  ProfRec r;
  if (!g_prof_pool.empty()) { r = g_prof_pool.back(); g_prof_pool.pop_back(); }
  else { (void)hipEventCreate(&r.e0); (void)hipEventCreate(&r.e1); }
  r.family = family; r.work = work;
  (void)hipEventRecord(r.e0, s);
  g_prof.push_back(r);
  idx = (int)g_prof.size() - 1;
}
UmprProfScope::~UmprProfScope() {
  if (idx >= 0) (void)hipEventRecord(g_prof[idx].e1, stream);
}

namespace {
inline hipStream_t S(void* s) { return static_cast<hipStream_t>(s); }
constexpr int D = 128, H = 64, G3 = 192, AT = 64;

// VGG16-D geometry (torchvision cfg "D"): 13 convs in 5 blocks, a 2x2 max pool after each block
constexpr int kConvPerBlock[5] = {2, 3 - 1, 3, 3, 3};  // 2,2,3,3,3
constexpr int kBlockCh[5] = {64, 128, 256, 512, 512};
constexpr int kFc[3][2] = {{25088, 4096}, {4096, 4096}, {4096, 1000}};

struct VggLayout {
  // feature activations in order: for each block, conv outputs (post-ReLU) then the pooled output
  size_t conv_off[13]; size_t pool_off[5];
  int conv_cin[13], conv_cout[13], conv_hw[13], conv_block[13];
  size_t fc_off[2];     // ReLU outputs of fc1, fc2 ([n][4096])
  size_t drop_off[2];   // dropout outputs
  size_t v_off[13], v_floats[13];   // transformed inputs the forward keeps for the weight gradient (umpr_wino_v_floats; 0: none)
  size_t total;         // floats
};
VggLayout vgg_layout(int n) {
  VggLayout L;
  size_t off = 0;
  int cin = 3, hw = 224, ci = 0;
  for (int b = 0; b < 5; ++b) {
    for (int j = 0; j < kConvPerBlock[b]; ++j) {
      L.conv_cin[ci] = cin; L.conv_cout[ci] = kBlockCh[b]; L.conv_hw[ci] = hw; L.conv_block[ci] = b;
      L.conv_off[ci] = off;
      off += (size_t)n * kBlockCh[b] * hw * hw;
      cin = kBlockCh[b];
      ++ci;
    }
    hw /= 2;
    L.pool_off[b] = off;
    off += (size_t)n * kBlockCh[b] * hw * hw;
  }
  for (int j = 0; j < 2; ++j) { L.fc_off[j] = off; off += (size_t)n * 4096; }
  for (int j = 0; j < 2; ++j) { L.drop_off[j] = off; off += (size_t)n * 4096; }
  for (int i = 0; i < 13; ++i) {
    // only layers whose forward takes the Winograd path (56 / 28 maps, conv2_2 at 112; the 14x14 maps' weight gradient is on
    // the 2x2 tile and umpr_wino_v_floats returns 0 for them)
    const bool wl = umpr_conv3x3_fwd_is_wino(L.conv_cin[i], L.conv_cout[i], L.conv_hw[i], L.conv_hw[i]);
    L.v_floats[i] = wl ? umpr_wino_v_floats(n, L.conv_cin[i], L.conv_cout[i], L.conv_hw[i], L.conv_hw[i]) : 0;
    L.v_off[i] = off;
    off += L.v_floats[i];
  }
  L.total = off;
  return L;
}
}  // namespace

extern "C" {

const char* umpr_version(void) { return "umpr_hip 0.1 (gfx950)"; }
const char* umpr_last_error(void) { return g_err; }

int umpr_device_info(int dev, char* name, int name_len, int* compute_units, size_t* hbm_bytes) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { umpr_set_error("device_info: no device %d", dev); return -1; }
  if (name && name_len > 0) { strncpy(name, prop.name, name_len - 1); name[name_len - 1] = 0; }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return 0;
}

int umpr_gemm_f32(const float* A, long lda, int transA, const float* B, long ldb, int transB, float* C, long ldc,
                  int M, int N, int K, const float* bias, int bias_mode, int act, int accumulate, float alpha,
                  float* ws, size_t ws_bytes, void* stream) {
  UmprGemm g;
  g.A = A; g.lda = lda; g.transA = transA != 0; g.B = B; g.ldb = ldb; g.transB = transB != 0; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.bias = bias; g.bias_mode = bias_mode; g.act = act; g.accumulate = accumulate != 0;
  g.alpha = alpha; g.split_k = ws ? 0 : 1; g.ws = ws; g.ws_bytes = ws_bytes;
  return umpr_gemm(g, S(stream));
}

// Mixed precision for the GEMM-shaped text products (GRU input projections, co-attention / S-Net / C-Net projections and
// their gradients): while set on the calling host thread, every product the library issues through its generic GEMM rounds
// its operands to bf16 on the way into LDS and runs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (what
// torch.autocast does to the same nn.GRU / nn.Linear / nn.Conv1d layers).  Hosts set it around a call and clear it after.
int umpr_set_gemm_bf16(int on) { umpr_gemm_set_b16(on != 0); return 0; }
int umpr_set_conv_inference(int on) { umpr_wino_set_inference(on != 0); return 0; }
int umpr_set_conv_pool_follows(int on) { umpr_wino_set_pool_follows(on != 0); return 0; }
long umpr_debug_wino_fix_count(void) { return umpr_wino_last_fix_count(); }

// ------------------------------------------------------------------------------------------------ GRU
// split-K slabs of the dW_ih product ([384][Ep] each, two slabs per unit): 256 units = up to 128 splits of the [384 x Ep] x N*L
// product (64 left its 3 x 64 = 192 workgroups walking 800 reduction steps each at GloVe-50d: 42 us at batch 32)
constexpr size_t kSlabs = 256;
size_t umpr_embed_gru_bidir_ws_bytes(int N, int L, int E) {
  const size_t tiles = umpr_gru_tiles(N);
  // gx / dgx [N*L*384] + dWhh slabs + bias slabs + split-K slab for dW_ih (forward: the stacked input weights) + the
  // stacked [384][Ep] weight gradient + the gathered embedding rows [N*L][Ep]; Ep = E rounded up to 4 floats (16-B rows)
  const size_t Ep = (size_t)((E + 3) & ~3);
  return ((size_t)N * L * 384 + tiles * 2 * G3 * H + tiles * 2 * 2 * G3 + (size_t)kSlabs * G3 * Ep + (size_t)2 * G3 * Ep +
          (size_t)N * L * Ep) * sizeof(float);
}
namespace {
// The token rows of the embedding table, gathered ONCE into a contiguous [N*L][E] matrix (one wave per row, whole 1200-B rows)
// instead of inside the projection GEMM, whose 128 x 128 tiles fetched 64-B pieces of random rows of the 480 MB table stage
// by stage and three times over (one per column tile): 228 us for 260 MB at E = 300.  UMPR_EMB_GATHER=0: gather in the GEMM.
const bool g_emb_gather = umpr_env_on("UMPR_EMB_GATHER");
// Rows are written with a pitch of Ep = E rounded up to 4 floats, the tail zero-filled: the projection GEMMs then stage both
// operands as float4 (GloVe-50d: 200-B rows made every load of them a scalar one - 63 us for the [25600 x 50] x [50 x 384]
// product at batch 32, against the 39 MB it writes).  Zero columns add exact zeros to the sums.
__global__ __launch_bounds__(256) void gather_rows_kernel(const int64_t* __restrict__ ids, const float* __restrict__ emb,
                                                          int E, int Ep, float* __restrict__ out, long rows) {
  const int lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
    const float* src = emb + ids[r] * (long)E;
    float* dst = out + r * (long)Ep;
    if ((E & 3) == 0) {
      for (int c = lane; c < E / 4; c += 64) reinterpret_cast<float4*>(dst)[c] = reinterpret_cast<const float4*>(src)[c];
    } else {
      for (int c = lane; c < Ep; c += 64) dst[c] = c < E ? src[c] : 0.f;
    }
  }
}
// [W_ih_f ; W_ih_r] as one [384][Ep] matrix (zero tail columns) + [b_ih_f ; b_ih_r]
__global__ __launch_bounds__(256) void stack_wih_kernel(const float* __restrict__ wf, const float* __restrict__ wr,
                                                        const float* __restrict__ bf, const float* __restrict__ br, int E, int Ep,
                                                        float* __restrict__ wst, float* __restrict__ bst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 2 * G3 * Ep) {
    const int row = i / Ep, c = i - row * Ep;
    const float* src = row < G3 ? wf + (long)row * E : wr + (long)(row - G3) * E;
    wst[i] = c < E ? src[c] : 0.f;
  }
  if (i < 2 * G3) bst[i] = i < G3 ? bf[i] : br[i - G3];
}
// rows 0..191 / 192..383 of the stacked [384][Ep] weight gradient to (or onto) the two parameters' [192][E] gradients
__global__ __launch_bounds__(256) void unstack_dwih_kernel(const float* __restrict__ stacked, int E, int Ep, float* __restrict__ df,
                                                           float* __restrict__ dr, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * G3 * E) return;
  const int row = i / E, c = i - row * E;
  const float v = stacked[(long)row * Ep + c];
  float* dst = row < G3 ? df + (long)row * E + c : dr + (long)(row - G3) * E + c;
  *dst = accumulate ? *dst + v : v;
}
inline int emb_pitch(int E) { return g_emb_gather ? (E + 3) & ~3 : E; }
float* gathered_rows(float* ws, int N, int L, int Ep) {
  return ws + (size_t)N * L * 384 + (size_t)umpr_gru_tiles(N) * (2 * G3 * H + 2 * 2 * G3) + (size_t)kSlabs * G3 * Ep + (size_t)2 * G3 * Ep;
}
int gather_rows(const int64_t* ids, const float* emb, int E, int Ep, float* out, long rows, hipStream_t s) {
  long blocks = (rows + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  gather_rows_kernel<<<(unsigned)blocks, 256, 0, s>>>(ids, emb, E, Ep, out, rows);
  UMPR_LAUNCH_CHECK("gather_rows");
  return 0;
}
}  // namespace

int umpr_embed_gru_bidir_fwd(const int64_t* ids, const float* emb, int E,
                             const float* w_ih_f, const float* w_hh_f, const float* b_ih_f, const float* b_hh_f,
                             const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                             const int32_t* lengths, const int32_t* order, const int32_t* dst_row, int N, int L,
                             float* out, float* saved, float* ws, size_t ws_bytes, void* stream) {
  UMPR_REQUIRE(N > 0 && L > 0 && E > 0, "embed_gru: bad shape N=%d L=%d E=%d", N, L, E);
  UMPR_REQUIRE(ws_bytes >= (size_t)N * L * 384 * sizeof(float), "embed_gru: workspace too small");
  float* gx = ws;
  // gx[:, 0:192 | 192:384] = emb[ids] [W_ih_f ; W_ih_r]^T + [b_ih_f ; b_ih_r]: ONE gather-GEMM for both directions (the
  // embedding rows are fetched once instead of twice, and N = 384 fills three 128-column tiles where 192 wasted half of
  // its second one).  The stacked [384][E] weight / [384] bias live behind gx in the workspace.
  if (ws_bytes >= umpr_embed_gru_bidir_ws_bytes(N, L, E)) {
    const int Ep = emb_pitch(E);
    float* wst = gx + (size_t)N * L * 384 + (size_t)umpr_gru_tiles(N) * (2 * G3 * H + 2 * 2 * G3);   // the dW_ih slab area
    float* bst = wst + (size_t)2 * G3 * Ep;
    const hipStream_t s = S(stream);
    // stack the two directions' input weights and biases: one launch
    stack_wih_kernel<<<cdiv(2 * G3 * Ep, 256), 256, 0, s>>>(w_ih_f, w_ih_r, b_ih_f, b_ih_r, E, Ep, wst, bst);
    UMPR_LAUNCH_CHECK("stack_wih");
    UmprGemm g;
    if (g_emb_gather) {
      float* xg = gathered_rows(ws, N, L, Ep);
      if (int rc = gather_rows(ids, emb, E, Ep, xg, (long)N * L, s)) return rc;
      g.A = xg; g.lda = Ep;
    } else {
      g.A = emb; g.lda = E; g.gatherA = ids;
    }
    g.B = wst; g.ldb = Ep; g.transB = true;
    g.C = gx; g.ldc = 384; g.M = N * L; g.N = 2 * G3; g.K = Ep; g.bias = bst; g.bias_mode = 1;
    if (int rc = umpr_gemm(g, s)) return rc;
  } else {
    for (int d = 0; d < 2; ++d) {  // small workspace (inference callers sized for gx only): one GEMM per direction
      UmprGemm g;
      g.A = emb; g.lda = E; g.gatherA = ids; g.B = d ? w_ih_r : w_ih_f; g.ldb = E; g.transB = true;
      g.C = gx + d * G3; g.ldc = 384; g.M = N * L; g.N = G3; g.K = E;
      g.bias = d ? b_ih_r : b_ih_f; g.bias_mode = 1;
      if (int rc = umpr_gemm(g, S(stream))) return rc;
    }
  }
  return umpr_gru_recurrent_fwd(gx, w_hh_f, b_hh_f, w_hh_r, b_hh_r, lengths, order, dst_row, out, saved, N, L, S(stream));
}

int umpr_embed_gru_bidir_bwd_acc(const int64_t* ids, const float* emb, int E, const float* w_hh_f, const float* w_hh_r,
                                 const int32_t* lengths, const int32_t* order, const int32_t* dst_row, int N, int L,
                                 const float* dout, const float* out, const float* saved,
                                 float* dw_ih_f, float* dw_hh_f, float* db_ih_f, float* db_hh_f,
                                 float* dw_ih_r, float* dw_hh_r, float* db_ih_r, float* db_hh_r,
                                 int accumulate, float* ws, size_t ws_bytes, void* stream) {
  UMPR_REQUIRE(dout != nullptr && out != nullptr && saved != nullptr, "embed_gru_bwd: dout / out / saved must not be NULL");
  UMPR_REQUIRE(ws_bytes >= umpr_embed_gru_bidir_ws_bytes(N, L, E), "embed_gru_bwd: workspace too small");
  const int tiles = umpr_gru_tiles(N);
  float* dgx = ws;
  float* wslab = dgx + (size_t)N * L * 384;
  float* bslab = wslab + (size_t)tiles * 2 * G3 * H;
  float* kslab = bslab + (size_t)tiles * 2 * 2 * G3;
  if (int rc = umpr_gru_bptt(dout, out, saved, w_hh_f, w_hh_r, lengths, order, dst_row, dgx, wslab, bslab, N, L, S(stream)))
    return rc;
  float* dwhh[2] = {dw_hh_f, dw_hh_r};
  float* dbih[2] = {db_ih_f, db_ih_r};
  float* dbhh[2] = {db_hh_f, db_hh_r};
  float* dwih[2] = {dw_ih_f, dw_ih_r};
  {  // the six per-tile slab reductions (dW_hh, db_ih, db_hh of both directions): one launch
    const float* src[6]; float* dst[6]; int rows[6]; long cols[6], rs[6]; int acc[6];
    for (int d = 0; d < 2; ++d) {
      src[3 * d] = wslab + (size_t)d * G3 * H;            dst[3 * d] = dwhh[d];     cols[3 * d] = G3 * H;     rs[3 * d] = 2 * G3 * H;
      src[3 * d + 1] = bslab + (size_t)d * 2 * G3;        dst[3 * d + 1] = dbih[d]; cols[3 * d + 1] = G3;     rs[3 * d + 1] = 4 * G3;
      src[3 * d + 2] = bslab + (size_t)d * 2 * G3 + G3;   dst[3 * d + 2] = dbhh[d]; cols[3 * d + 2] = G3;     rs[3 * d + 2] = 4 * G3;
    }
    for (int q = 0; q < 6; ++q) { rows[q] = tiles; acc[q] = accumulate; }
    if (int rc = umpr_multi_colsum_rows(src, rows, cols, rs, dst, acc, 6, S(stream))) return rc;
  }
  {  // [dW_ih_f ; dW_ih_r] [384][E] = dgx^T emb[ids]: one split-K gather-GEMM for both directions, then rows 0..191 /
     // 192..383 are copied (or added) to their parameters' gradients
    const int Ep = emb_pitch(E);
    float* stacked = kslab + (size_t)kSlabs * G3 * Ep;
    UmprGemm g;
    g.A = dgx; g.lda = 384; g.transA = true;
    if (g_emb_gather) {
      float* xg = gathered_rows(ws, N, L, Ep);
      if (int rc = gather_rows(ids, emb, E, Ep, xg, (long)N * L, S(stream))) return rc;
      g.B = xg; g.ldb = Ep;
    } else {
      g.B = emb; g.ldb = E; g.gatherB = ids;
    }
    g.C = stacked; g.ldc = Ep; g.M = 2 * G3; g.N = Ep; g.K = N * L; g.split_k = 0; g.ws = kslab;   // (tail columns: zeros)
    g.ws_bytes = (size_t)kSlabs * G3 * Ep * sizeof(float);
    if (int rc = umpr_gemm(g, S(stream))) return rc;
    unstack_dwih_kernel<<<cdiv(2 * G3 * E, 256), 256, 0, S(stream)>>>(stacked, E, Ep, dwih[0], dwih[1], accumulate);
    UMPR_LAUNCH_CHECK("unstack_dwih");
  }
  return 0;
}

int umpr_embed_gru_bidir_bwd(const int64_t* ids, const float* emb, int E, const float* w_hh_f, const float* w_hh_r,
                             const int32_t* lengths, const int32_t* order, const int32_t* dst_row, int N, int L,
                             const float* dout, const float* out, const float* saved,
                             float* dw_ih_f, float* dw_hh_f, float* db_ih_f, float* db_hh_f,
                             float* dw_ih_r, float* dw_hh_r, float* db_ih_r, float* db_hh_r,
                             float* ws, size_t ws_bytes, void* stream) {
  return umpr_embed_gru_bidir_bwd_acc(ids, emb, E, w_hh_f, w_hh_r, lengths, order, dst_row, N, L, dout, out, saved, dw_ih_f,
                                      dw_hh_f, db_ih_f, db_hh_f, dw_ih_r, dw_hh_r, db_ih_r, db_hh_r, 0, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------ co-attention
size_t umpr_coattention_fwd_ws_bytes(int B, int SL) { return umpr_coattn_fwd_ws_bytes(B, SL); }
int umpr_coattention_fwd(const float* Gu, const float* Gi, const float* M, int B, int SL, float* T, float* soft_u,
                         float* soft_i, float* atte_u, long ld_u, float* atte_i, long ld_i, float* colmax,
                         int32_t* argcol, float* rowmax, int32_t* argrow, float* ws, size_t ws_bytes, void* stream) {
  return umpr_coattn_fwd_impl(Gu, Gi, M, B, SL, T, soft_u, soft_i, atte_u, ld_u, atte_i, ld_i, colmax, argcol, rowmax,
                              argrow, ws, ws_bytes, S(stream));
}
int umpr_coattention_fwd_bf16(const float* Gu, const float* Gi, const float* M, int B, int SL, float* T, float* soft_u,
                              float* soft_i, float* atte_u, long ld_u, float* atte_i, long ld_i, float* colmax,
                              int32_t* argcol, float* rowmax, int32_t* argrow, float* ws, size_t ws_bytes, void* stream) {
  return umpr_coattn_fwd_impl(Gu, Gi, M, B, SL, T, soft_u, soft_i, atte_u, ld_u, atte_i, ld_i, colmax, argcol, rowmax,
                              argrow, ws, ws_bytes, S(stream), 1);
}
size_t umpr_coattention_bwd_ws_bytes(int B, int SL) { return umpr_coattn_bwd_ws_bytes(B, SL); }
int umpr_coattention_bwd(const float* Gu, const float* Gi, const float* M, const float* T, const float* soft_u,
                         const float* soft_i, const float* colmax, const int32_t* argcol, const float* rowmax,
                         const int32_t* argrow, const float* d_atte_u, long ld_du, const float* d_atte_i, long ld_di,
                         const float* d_soft_u, const float* d_soft_i, int B, int SL, float* dGu, float* dGi,
                         float* dM, int accumulate, float* ws, size_t ws_bytes, void* stream) {
  return umpr_coattn_bwd_impl(Gu, Gi, M, T, soft_u, soft_i, colmax, argcol, rowmax, argrow, d_atte_u, ld_du, d_atte_i,
                              ld_di, d_soft_u, d_soft_i, B, SL, dGu, dGi, dM, accumulate, ws, ws_bytes, S(stream));
}

// ------------------------------------------------------------------------------------------------ S-Net
int umpr_snet_fwd(const float* X, const float* Ms, const float* Ws, const float* word_soft, int wl, int B, int S_,
                  int L, float* U, float* P, float* wsum, float* self_atte, float* senti, long ld_senti, void* stream) {
  return umpr_snet_fwd_impl(X, Ms, Ws, word_soft, wl, B, S_, L, U, P, wsum, self_atte, senti, ld_senti, S(stream));
}
size_t umpr_snet_bwd_ws_bytes(int B, int S_, int L) { return umpr_snet_bwd_ws_bytes_impl(B, S_, L); }
int umpr_snet_bwd(const float* X, const float* Ms, const float* Ws, const float* U, const float* P, const float* wsum,
                  const float* self_atte, const float* d_senti, long ld_ds, const float* d_self_atte, int B, int S_,
                  int L, int wl, float* dX, float* dMs, float* dWs, float* d_word_soft, float* ws, size_t ws_bytes,
                  void* stream) {
  return umpr_snet_bwd_impl(X, Ms, Ws, U, P, wsum, self_atte, d_senti, ld_ds, d_self_atte, B, S_, L, wl, dX, dMs, dWs,
                            d_word_soft, ws, ws_bytes, S(stream));
}

// ------------------------------------------------------------------------------------------------ merge
int umpr_review_merge_fwd(const float* repr_u, const float* repr_i, const float* W_u, const float* W_i, int B,
                          float* out, void* stream) {
  if (g_merge_small && B <= 256) return umpr_review_merge_fwd_impl(repr_u, repr_i, W_u, W_i, B, out, nullptr, S(stream));
  if (g_fc_small && umpr_fc_small_ok(B, D, 2 * D)) {
    // batch-sized M: the register-streaming kernels (no LDS stage, no barrier per k-tile) instead of one workgroup of
    // the tiled GEMM walking 16 k-tiles one global round trip at a time (35 us -> 4 us at B = 32)
    if (int rc = umpr_fc_small_fwd(repr_u, W_u, nullptr, out, B, D, 2 * D, UMPR_ACT_NONE, nullptr, 0, S(stream))) return rc;
    return umpr_fc_small_fwd(repr_i, W_i, nullptr, out, B, D, 2 * D, UMPR_ACT_TANH, nullptr, 0, S(stream), 1);
  }
  UmprGemm g;
  g.A = repr_u; g.lda = 2 * D; g.B = W_u; g.ldb = 2 * D; g.transB = true; g.C = out; g.ldc = D; g.M = B; g.N = D; g.K = 2 * D;
  if (int rc = umpr_gemm(g, S(stream))) return rc;
  g.A = repr_i; g.B = W_i; g.accumulate = true; g.act = UMPR_ACT_TANH;
  return umpr_gemm(g, S(stream));
}
size_t umpr_review_merge_bwd_ws_bytes(int B) { return (size_t)B * D * sizeof(float); }
int umpr_review_merge_bwd(const float* repr_u, const float* repr_i, const float* W_u, const float* W_i,
                          const float* out, const float* d_out, int B, float* d_repr_u, float* d_repr_i, float* dW_u,
                          float* dW_i, float* ws, size_t ws_bytes, void* stream) {
  UMPR_REQUIRE(ws_bytes >= umpr_review_merge_bwd_ws_bytes(B), "review_merge_bwd: workspace too small");
  if (g_merge_small && B <= 256)   // tanh' folded into the two kernels
    return umpr_review_merge_bwd_impl(repr_u, repr_i, W_u, W_i, out, d_out, B, d_repr_u, d_repr_i, dW_u, dW_i, S(stream));
  float* dpre = ws;
  if (int rc = umpr_tanh_bwd(out, d_out, dpre, (long)B * D, S(stream))) return rc;
  const float* reprs[2] = {repr_u, repr_i};
  const float* Wm[2] = {W_u, W_i};
  float* drepr[2] = {d_repr_u, d_repr_i};
  float* dW[2] = {dW_u, dW_i};
  if (g_fc_small && umpr_fc_small_ok(B, D, 2 * D)) {
    for (int q = 0; q < 2; ++q) {
      if (int rc = umpr_fc_small_dx(dpre, Wm[q], drepr[q], B, D, 2 * D, nullptr, 0, S(stream))) return rc;
      if (int rc = umpr_fc_small_dw(dpre, reprs[q], dW[q], B, D, 2 * D, S(stream))) return rc;
    }
    return 0;
  }
  for (int q = 0; q < 2; ++q) {
    UmprGemm g;  // d_repr = dpre W
    g.A = dpre; g.lda = D; g.B = Wm[q]; g.ldb = 2 * D; g.C = drepr[q]; g.ldc = 2 * D; g.M = B; g.N = 2 * D; g.K = D;
    if (int rc = umpr_gemm(g, S(stream))) return rc;
    UmprGemm h;  // dW = dpre^T repr
    h.A = dpre; h.lda = D; h.transA = true; h.B = reprs[q]; h.ldb = 2 * D; h.C = dW[q]; h.ldc = 2 * D; h.M = D;
    h.N = 2 * D; h.K = B;
    if (int rc = umpr_gemm(h, S(stream))) return rc;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ C-Net head
size_t umpr_cnet_head_fwd_ws_bytes(int B, int S_, int L, int KS) { return umpr_cnet_fwd_ws_bytes(B, S_, L, KS); }
int umpr_cnet_head_fwd(const float* X, const float* Wc, const float* bc, const float* Wl, const float* bl, float thr,
                       int B, int S_, int L, int KC, int KS, int V, float* Y, float* cmax, int32_t* argl, float* sp,
                       float* view_p, float* final_, float* ws, size_t ws_bytes, void* stream) {
  return umpr_cnet_head_fwd_impl(X, Wc, bc, Wl, bl, thr, B, S_, L, KC, KS, V, Y, cmax, argl, sp, view_p, final_, ws,
                                 ws_bytes, S(stream));
}
size_t umpr_cnet_head_bwd_ws_bytes(int B, int S_, int L, int KC, int KS, int V) {
  return umpr_cnet_bwd_ws_bytes(B, S_, L, KC, KS, V);
}
int umpr_cnet_head_bwd(const float* X, const float* Wc, const float* Wl, const float* cmax, const int32_t* argl,
                       const float* sp, const float* view_p, const float* d_final, const float* d_view_p, int B,
                       int S_, int L, int KC, int KS, int V, float* dX, int accumulate_dX, int accumulate_w, float* dWc,
                       float* dbc, float* dWl, float* dbl, float* ws, size_t ws_bytes, void* stream) {
  return umpr_cnet_head_bwd_impl(X, Wc, Wl, cmax, argl, sp, view_p, d_final, d_view_p, B, S_, L, KC, KS, V, dX,
                                 accumulate_dX, accumulate_w, dWc, dbc, dWl, dbl, ws, ws_bytes, S(stream));
}

// ------------------------------------------------------------------------------------------------ gate
int umpr_control_gate_fwd(const float* self_atte, const float* w, const float* bias, const float* view_p,
                          const float* c_out, int B, int S_, int V, float* senti, float* view_score,
                          float* prefer_pos, float* prefer_neg, void* stream) {
  return umpr_gate_fwd_impl(self_atte, w, bias, view_p, c_out, B, S_, V, senti, view_score, prefer_pos, prefer_neg, S(stream));
}
size_t umpr_control_gate_bwd_ws_bytes(int B) { return umpr_gate_bwd_ws_bytes(B); }
int umpr_control_gate_bwd(const float* self_atte, const float* w, const float* view_p, const float* c_out,
                          const float* senti, const float* view_score, const float* d_prefer_pos,
                          const float* d_prefer_neg, int B, int S_, int V, float* d_self_atte, float* d_view_p,
                          float* d_c_out, float* dw, float* db, float* ws, size_t ws_bytes, void* stream) {
  return umpr_gate_bwd_impl(self_atte, w, view_p, c_out, senti, view_score, d_prefer_pos, d_prefer_neg, B, S_, V,
                            d_self_atte, d_view_p, d_c_out, dw, db, ws, ws_bytes, S(stream));
}

// ------------------------------------------------------------------------------------------------ VGG16
size_t umpr_vgg16_act_bytes(int n_img) { return vgg_layout(n_img).total * sizeof(float); }

namespace {
size_t vgg_scratch_bytes(int n) {
  // largest of: conv wgrad slabs, split-K slabs of the small-batch classifier GEMMs
  size_t slab = 0;
  const VggLayout L = vgg_layout(n);
  for (int i = 0; i < 13; ++i) {
    const size_t b = umpr_conv3x3_wgrad_ws_bytes(n, L.conv_cin[i], L.conv_cout[i], L.conv_hw[i], L.conv_hw[i]);
    if (b > slab) slab = b;
  }
  for (int i = 0; i < 13; ++i) {  // packed weights / Winograd buffers of the forward convs share this region
    const size_t b = umpr_conv3x3_pack_floats(n, L.conv_cin[i], L.conv_cout[i], L.conv_hw[i], L.conv_hw[i]) * sizeof(float);
    if (b > slab) slab = b;
  }
  size_t fcs = (size_t)16 * n * 25088;
  if ((size_t)128 * n * 4096 > fcs) fcs = (size_t)128 * n * 4096;
  fcs *= sizeof(float);
  return align_up(slab > fcs ? slab : fcs, 256);
}
// scratch of the data-gradient convs (packed weights / Winograd buffers): as large as the forward scratch
static size_t wt_floats(int n) { return vgg_scratch_bytes(n) / sizeof(float); }
}  // namespace

size_t umpr_vgg16_fwd_ws_bytes(int n_img) { return vgg_scratch_bytes(n_img); }

size_t umpr_vgg16_ws_bytes(int n_img) {
  // [d_pool5][3 rotating gradient slots] (largest activation each) [packed weights][scratch]
  const size_t big = (size_t)n_img * 64 * 224 * 224;
  return ((size_t)n_img * 25088 + 3 * big + wt_floats(n_img)) * sizeof(float) + vgg_scratch_bytes(n_img);
}

size_t umpr_conv3x3_pack_bytes(int N, int Cin, int Cout, int H_, int W) {
  return umpr_conv3x3_pack_floats(N, Cin, Cout, H_, W) * sizeof(float);
}
int umpr_conv3x3_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H_, int W,
                     int Cout, int relu, float* wpack, size_t wpack_bytes, void* stream) {
  return umpr_conv3x3_run(x, w, 0, bias, nullptr, y, N, Cin, Cout, H_, W, relu, wpack, wpack_bytes / sizeof(float),
                          S(stream));
}
int umpr_conv3x3_bwd_data(const float* dy, const float* w, const float* mask_src, float* dx, int N, int Cin, int H_,
                          int W, int Cout, float* wt, size_t wt_bytes, void* stream) {
  return umpr_conv3x3_run(dy, w, 1, nullptr, mask_src, dx, N, Cin, Cout, H_, W, 0, wt, wt_bytes / sizeof(float),
                          S(stream));
}
size_t umpr_conv3x3_bwd_weight_ws_bytes(int N, int Cin, int Cout, int H_, int W) {
  return umpr_conv3x3_wgrad_ws_bytes(N, Cin, Cout, H_, W);
}
int umpr_conv3x3_bwd_weight(const float* dy, const float* x, float* dw, float* db, int N, int Cin, int H_, int W,
                            int Cout, float* ws, size_t ws_bytes, void* stream) {
  return umpr_conv3x3_wgrad(dy, x, dw, db, N, Cin, Cout, H_, W, 0, ws, ws_bytes, S(stream));
}
int umpr_maxpool2_fwd(const float* x, float* y, long planes, int H_, int W, void* stream) {
  return umpr_maxpool2_fwd_impl(x, y, planes, H_, W, S(stream));
}
int umpr_maxpool2_bwd_relu(const float* x, const float* dy, float* dx, long planes, int H_, int W, void* stream) {
  return umpr_maxpool2_bwd_relu_impl(x, dy, dx, planes, H_, W, S(stream));
}

size_t umpr_vgg16_pool5_offset(int n_img) { return vgg_layout(n_img).pool_off[4] * sizeof(float); }

int umpr_vgg16_features_fwd(const float* images, const float* const* params, int n, float* acts, float* ws,
                            size_t ws_bytes, void* stream) {
  UMPR_REQUIRE(n > 0 && images && params && acts, "vgg16_features_fwd: bad arguments");
  UMPR_REQUIRE(ws_bytes >= umpr_vgg16_fwd_ws_bytes(n), "vgg16_features_fwd: workspace too small");
  const VggLayout L = vgg_layout(n);
  hipStream_t s = S(stream);
  const float* x = images;
  int ci = 0;
  for (int b = 0; b < 5; ++b) {
    for (int j = 0; j < kConvPerBlock[b]; ++j, ++ci) {
      float* y = acts + L.conv_off[ci];
      const int hw = L.conv_hw[ci];
      umpr_wino_set_pool_follows(j == kConvPerBlock[b] - 1);   // the block's last convolution feeds the max-pool
      if (L.v_floats[ci] && !umpr_wino_inference()) umpr_wino_set_v_slot(acts + L.v_off[ci], L.v_floats[ci]);
      const int rc = umpr_conv3x3_run(x, params[2 * ci], 0, params[2 * ci + 1], nullptr, y, n, L.conv_cin[ci],
                                      L.conv_cout[ci], hw, hw, 1, ws, ws_bytes / sizeof(float), s);
      umpr_wino_set_v_slot(nullptr, 0);
      umpr_wino_set_pool_follows(0);
      if (rc) return rc;
      x = y;
    }
    const int hw = L.conv_hw[ci - 1];
    float* y = acts + L.pool_off[b];
    if (int rc = umpr_maxpool2_fwd_impl(x, y, (long)n * kBlockCh[b], hw, hw, s)) return rc;
    x = y;
  }
  return 0;
}

namespace {
// classifier activations: pool5 input [n][25088], ReLU outputs of fc1 / fc2 and their dropout outputs ([n][4096] each)
struct ClsActs { const float* pool5; float* fc[2]; float* drop[2]; };

int classifier_fwd_impl(const float* const* params, int n, int train, int use_masks, uint64_t seed, const ClsActs& A,
                        uint8_t* masks, float* out, float* ws, size_t ws_bytes, hipStream_t s, int bf16 = 0) {
  // AdaptiveAvgPool2d(7) is the identity on the 7x7 map a 224x224 image produces
  const float* x = A.pool5;
  for (int j = 0; j < 3; ++j) {
    float* y = j < 2 ? A.fc[j] : out;
    const int act = j < 2 ? UMPR_ACT_RELU : UMPR_ACT_NONE;
    if (g_fc_small && umpr_fc_small_ok(n, kFc[j][1], kFc[j][0])) {
      // batch-sized M: register-streaming kernels (fc_small.hip) instead of the LDS-tiled GEMM
      if (int rc = umpr_fc_small_fwd(x, params[26 + 2 * j], params[27 + 2 * j], y, n, kFc[j][1], kFc[j][0], act, ws,
                                     ws_bytes, s, 0, bf16)) return rc;
    } else {
      UmprGemm g;
      g.A = x; g.lda = kFc[j][0]; g.B = params[26 + 2 * j]; g.ldb = kFc[j][0]; g.transB = true;
      g.C = y; g.ldc = kFc[j][1]; g.M = n; g.N = kFc[j][1]; g.K = kFc[j][0];
      g.bias = params[27 + 2 * j]; g.bias_mode = 1; g.act = act;
      g.split_k = 0; g.ws = ws; g.ws_bytes = ws_bytes;
      if (int rc = umpr_gemm(g, s)) return rc;
    }
    x = y;
    if (j < 2 && (train || use_masks)) {
      float* y = A.drop[j];
      if (int rc = umpr_dropout_fwd_impl(x, y, masks + (size_t)j * n * 4096, (long)n * 4096, 0.5f,
                                         seed + 0x9E3779B97F4A7C15ULL * (j + 1), use_masks ? 0 : 1, s)) return rc;
      x = y;
    }
  }
  return 0;
}
ClsActs cls_acts_full(float* acts, int n) {
  const VggLayout L = vgg_layout(n);
  return ClsActs{acts + L.pool_off[4], {acts + L.fc_off[0], acts + L.fc_off[1]}, {acts + L.drop_off[0], acts + L.drop_off[1]}};
}
ClsActs cls_acts_compact(float* arena, int n) {
  float* fc = arena + (size_t)n * 25088;
  const size_t q = (size_t)n * 4096;
  return ClsActs{arena, {fc, fc + q}, {fc + 2 * q, fc + 3 * q}};
}
}  // namespace

int umpr_vgg16_classifier_fwd(const float* const* params, int n, int train, int use_masks, uint64_t seed, float* acts,
                              uint8_t* masks, float* out, float* ws, size_t ws_bytes, void* stream) {
  UMPR_REQUIRE(n > 0 && params && acts && out, "vgg16_classifier_fwd: bad arguments");
  UMPR_REQUIRE(ws_bytes >= umpr_vgg16_fwd_ws_bytes(n), "vgg16_classifier_fwd: workspace too small");
  return classifier_fwd_impl(params, n, train, use_masks, seed, cls_acts_full(acts, n), masks, out, ws, ws_bytes, S(stream));
}

size_t umpr_vgg16_cls_arena_bytes(int n_img) { return ((size_t)n_img * 25088 + (size_t)4 * n_img * 4096) * sizeof(float); }

int umpr_vgg16_classifier_fwd_compact(const float* const* params, int n, int train, int use_masks, uint64_t seed,
                                      float* cls_arena, uint8_t* masks, float* out, float* ws, size_t ws_bytes,
                                      void* stream) {
  UMPR_REQUIRE(n > 0 && params && cls_arena && out, "vgg16_classifier_fwd_compact: bad arguments");
  UMPR_REQUIRE(ws_bytes >= umpr_vgg16_fwd_ws_bytes(n), "vgg16_classifier_fwd_compact: workspace too small");
  return classifier_fwd_impl(params, n, train, use_masks, seed, cls_acts_compact(cls_arena, n), masks, out, ws, ws_bytes,
                             S(stream));
}

int umpr_vgg16_fwd(const float* images, const float* const* params, int n, int train, int use_masks, uint64_t seed,
                   float* acts, uint8_t* masks, float* out, float* ws, size_t ws_bytes, void* stream) {
  if (int rc = umpr_vgg16_features_fwd(images, params, n, acts, ws, ws_bytes, stream)) return rc;
  return umpr_vgg16_classifier_fwd(params, n, train, use_masks, seed, acts, masks, out, ws, ws_bytes, stream);
}

size_t umpr_vgg16_classifier_bwd_ws_bytes(int n_img) {
  return (size_t)2 * n_img * 4096 * sizeof(float) + vgg_scratch_bytes(n_img);
}

// d_pool5 [n][25088] receives the gradient w.r.t. the pooled feature map; grads: the 32-pointer array (only the six
// classifier entries 26..31 are written).
namespace {
int classifier_bwd_impl(const float* const* params, int n, int train, const ClsActs& A, const uint8_t* masks,
                        const float* d_out, float* const* grads, float* d_pool5, float* ws, size_t ws_bytes, hipStream_t s,
                        int bf16 = 0) {
  float* gA = ws;
  float* gB = ws + (size_t)n * 4096;
  float* scratch = gB + (size_t)n * 4096;
  const size_t slab_bytes = vgg_scratch_bytes(n);
  const float* g = d_out;  // gradient w.r.t. the current layer's output
  float* cur = gA; float* oth = gB;
  for (int j = 2; j >= 0; --j) {
    const int fin = kFc[j][0], fout = kFc[j][1];
    const float* xin = j == 0 ? A.pool5 : (train ? A.drop[j - 1] : A.fc[j - 1]);
    if (j < 2) {
      // g is d(dropout output or relu output); fold dropout mask and ReLU into gz
      if (int rc = umpr_dropout_bwd_impl(g, train ? masks + (size_t)j * n * 4096 : nullptr, A.fc[j], cur,
                                         (long)n * 4096, 0.5f, s)) return rc;
      g = cur; float* t = cur; cur = oth; oth = t;
    }
    const bool small = g_fc_small && umpr_fc_small_ok(n, fout, fin);
    if (small) {  // dW[fout][fin] = g^T xin
      if (int rc = umpr_fc_small_dw(g, xin, grads[26 + 2 * j], n, fout, fin, s, bf16)) return rc;
    } else {
      UmprGemm w;
      w.A = g; w.lda = fout; w.transA = true; w.B = xin; w.ldb = fin; w.C = grads[26 + 2 * j]; w.ldc = fin;
      w.M = fout; w.N = fin; w.K = n;
      if (int rc = umpr_gemm(w, s)) return rc;
    }
    if (int rc = umpr_colsum_rows(g, n, fout, fout, grads[27 + 2 * j], 0, s)) return rc;
    float* dxo = j == 0 ? d_pool5 : cur;
    if (small) {  // dx[n][fin] = g W
      if (int rc = umpr_fc_small_dx(g, params[26 + 2 * j], dxo, n, fout, fin, scratch, slab_bytes, s, bf16)) return rc;
    } else {
      UmprGemm d;
      d.A = g; d.lda = fout; d.B = params[26 + 2 * j]; d.ldb = fin; d.C = dxo; d.ldc = fin; d.M = n;
      d.N = fin; d.K = fout; d.split_k = 0; d.ws = scratch; d.ws_bytes = slab_bytes;
      if (int rc = umpr_gemm(d, s)) return rc;
    }
    g = dxo; float* t = cur; cur = oth; oth = t;
  }
  return 0;
}
}  // namespace

int umpr_vgg16_classifier_bwd(const float* const* params, int n, int train, const float* acts, const uint8_t* masks,
                              const float* d_out, float* const* grads, float* d_pool5, float* ws, size_t ws_bytes,
                              void* stream) {
  UMPR_REQUIRE(ws_bytes >= umpr_vgg16_classifier_bwd_ws_bytes(n), "vgg16_classifier_bwd: workspace too small");
  return classifier_bwd_impl(params, n, train, cls_acts_full(const_cast<float*>(acts), n), masks, d_out, grads, d_pool5,
                             ws, ws_bytes, S(stream));
}

int umpr_vgg16_classifier_bwd_compact(const float* const* params, int n, int train, const float* cls_arena,
                                      const uint8_t* masks, const float* d_out, float* const* grads, float* d_pool5,
                                      float* ws, size_t ws_bytes, void* stream) {
  UMPR_REQUIRE(ws_bytes >= umpr_vgg16_classifier_bwd_ws_bytes(n), "vgg16_classifier_bwd_compact: workspace too small");
  return classifier_bwd_impl(params, n, train, cls_acts_compact(const_cast<float*>(cls_arena), n), masks, d_out, grads,
                             d_pool5, ws, ws_bytes, S(stream));
}

// Mixed precision (BASELINE.json configs[4]): the three products of each layer take bf16-rounded operands on the bf16
// matrix pipe with fp32 accumulation; weights, activations, biases, dropout and the gradients stay fp32 in memory.
int umpr_vgg16_classifier_fwd_compact_bf16(const float* const* params, int n, int train, int use_masks, uint64_t seed,
                                           float* cls_arena, uint8_t* masks, float* out, float* ws, size_t ws_bytes,
                                           void* stream) {
  UMPR_REQUIRE(n > 0 && params && cls_arena && out, "vgg16_classifier_fwd_compact_bf16: bad arguments");
  return classifier_fwd_impl(params, n, train, use_masks, seed, cls_acts_compact(cls_arena, n), masks, out, ws, ws_bytes,
                             S(stream), 1);
}
int umpr_vgg16_classifier_bwd_compact_bf16(const float* const* params, int n, int train, const float* cls_arena,
                                           const uint8_t* masks, const float* d_out, float* const* grads, float* d_pool5,
                                           float* ws, size_t ws_bytes, void* stream) {
  UMPR_REQUIRE(ws_bytes >= umpr_vgg16_classifier_bwd_ws_bytes(n), "vgg16_classifier_bwd_compact_bf16: workspace too small");
  return classifier_bwd_impl(params, n, train, cls_acts_compact(const_cast<float*>(cls_arena), n), masks, d_out, grads,
                             d_pool5, ws, ws_bytes, S(stream), 1);
}

size_t umpr_vgg16_features_bwd_ws_bytes(int n_img) {
  const size_t big = (size_t)n_img * 64 * 224 * 224;
  return (3 * big + wt_floats(n_img)) * sizeof(float) + vgg_scratch_bytes(n_img);
}

// Library-owned side stream for the weight-gradient kernels of the VGG backward (one per process: one GPU per process).
namespace {
struct WgradSide {
  hipStream_t stream = nullptr;
  hipEvent_t ready[13] = {};   // recorded on the caller's stream: gradient of layer ci is complete
  hipEvent_t done[13] = {};    // recorded on the side stream: wgrad of layer ci has finished reading it
  bool ok = false;
  bool init() {
    if (ok) return true;
    if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return false;
    for (int i = 0; i < 13; ++i) {
      if (hipEventCreateWithFlags(&ready[i], hipEventDisableTiming) != hipSuccess) return false;
      if (hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) return false;
    }
    ok = true;
    return true;
  }
};
WgradSide g_wside;
const bool g_wgrad_side = umpr_env_on("UMPR_WGRAD_STREAM");
// host callback after the kernels of one VGG block's backward have been enqueued (data parallel: the caller starts that
// block's gradient exchange while the blocks below are still being computed)
umpr_block_callback g_block_cb = nullptr;
void* g_block_user = nullptr;
}  // namespace

int umpr_vgg16_set_block_callback(umpr_block_callback cb, void* user) { g_block_cb = cb; g_block_user = user; return 0; }
void* umpr_vgg16_wgrad_stream(void) {
  return (g_wgrad_side && g_wside.init()) ? static_cast<void*>(g_wside.stream) : nullptr;
}

int umpr_vgg16_features_bwd(const float* images, const float* const* params, int n, const float* acts,
                            const float* d_pool5, float* const* grads, float* ws, size_t ws_bytes, void* stream) {
  UMPR_REQUIRE(ws_bytes >= umpr_vgg16_features_bwd_ws_bytes(n), "vgg16_features_bwd: workspace too small");
  const VggLayout L = vgg_layout(n);
  hipStream_t s = S(stream);
  const size_t big = (size_t)n * 64 * 224 * 224;
  float* buf[3] = {ws, ws + big, ws + 2 * big};
  float* wt = buf[2] + big;
  float* scratch = wt + wt_floats(n);
  const size_t slab_bytes = vgg_scratch_bytes(n);
  // The weight gradient of a layer depends only on that layer's output gradient, not on the data-gradient chain that
  // continues below it: it runs on a side stream, so that its MFMA work fills the HBM-bound phases of the chain
  // (pool backward, Winograd transforms) on the caller's stream.  The gradient buffers rotate through three slots:
  // a slot is rewritten two layers after it was produced, behind a wait for the wgrad that read it.
  const bool side = g_wgrad_side && g_wside.init();
  hipStream_t sw = side ? g_wside.stream : s;
  int slot_reader[3] = {-1, -1, -1};   // layer whose side-stream wgrad still reads the slot
  int nxt = 0;
  auto claim = [&]() -> float* {       // next slot to write on the caller's stream
    const int k = nxt; nxt = (nxt + 1) % 3;
    if (side && slot_reader[k] >= 0) { (void)hipStreamWaitEvent(s, g_wside.done[slot_reader[k]], 0); slot_reader[k] = -1; }
    return buf[k];
  };
  const float* g = d_pool5;
  int gslot = -1;                      // slot index holding g (-1: the caller's d_pool5)
  int ci = 12;
  for (int b = 4; b >= 0; --b) {
    const int hw = L.conv_hw[ci];
    // pool backward + ReLU mask of the conv output feeding it -> gradient w.r.t. the conv pre-activation
    float* cur = claim();
    if (int rc = umpr_maxpool2_bwd_relu_impl(acts + L.conv_off[ci], g, cur, (long)n * kBlockCh[b], hw, hw, s)) return rc;
    g = cur; gslot = (nxt + 2) % 3;
    for (int j = kConvPerBlock[b] - 1; j >= 0; --j, --ci) {
      const int cin = L.conv_cin[ci], cout = L.conv_cout[ci];
      const float* xin = ci == 0 ? images : (j == 0 ? acts + L.pool_off[b - 1] : acts + L.conv_off[ci - 1]);
      if (side) {
        (void)hipEventRecord(g_wside.ready[ci], s);
        (void)hipStreamWaitEvent(sw, g_wside.ready[ci], 0);
      }
      if (L.v_floats[ci]) umpr_wino_set_v_slot(const_cast<float*>(acts) + L.v_off[ci], L.v_floats[ci]);   // written by the forward
      const int rcw = umpr_conv3x3_wgrad(g, xin, grads[2 * ci], grads[2 * ci + 1], n, cin, cout, hw, hw, 0, scratch,
                                         slab_bytes, sw);
      umpr_wino_set_v_slot(nullptr, 0);
      if (rcw) return rcw;
      if (side) {
        (void)hipEventRecord(g_wside.done[ci], sw);
        slot_reader[gslot] = ci;
      }
      if (ci == 0) break;
      // input came straight from a conv+ReLU (j > 0): mask by it; from a pool (j == 0): the pool backward masks
      const float* mask = j > 0 ? xin : nullptr;
      cur = claim();
      if (int rc = umpr_conv3x3_run(g, params[2 * ci], 1, nullptr, mask, cur, n, cin, cout, hw, hw, 0, wt, wt_floats(n), s))
        return rc;
      g = cur; gslot = (nxt + 2) % 3;
    }
    if (g_block_cb) g_block_cb(b, g_block_user);   // every weight-gradient kernel of block b is enqueued (on sw)
    if (ci == 0 && b == 0) break;
  }
  if (side) {  // the caller's stream owns the results again: every weight gradient is complete behind this wait
    hipEvent_t& last = g_wside.done[0];
    (void)hipStreamWaitEvent(s, last, 0);
  }
  return 0;
}

int umpr_vgg16_bwd(const float* images, const float* const* params, int n, int train, const float* acts,
                   const uint8_t* masks, const float* d_out, float* const* grads, float* ws, size_t ws_bytes,
                   void* stream) {
  UMPR_REQUIRE(ws_bytes >= umpr_vgg16_ws_bytes(n), "vgg16_bwd: workspace too small (%zu < %zu)", ws_bytes,
               umpr_vgg16_ws_bytes(n));
  // d_pool5 lives at the front of the workspace; the stage workspaces follow it
  float* d_pool5 = ws;
  float* rest = ws + (size_t)n * 25088;
  const size_t rest_bytes = ws_bytes - (size_t)n * 25088 * sizeof(float);
  if (int rc = umpr_vgg16_classifier_bwd(params, n, train, acts, masks, d_out, grads, d_pool5, rest, rest_bytes, stream))
    return rc;
  return umpr_vgg16_features_bwd(images, params, n, acts, d_pool5, grads, rest, rest_bytes, stream);
}

// ------------------------------------------------------------------------------------------------ bf16 path
namespace {
struct VggB16Layout {
  UmprPF geo[5], pool_geo[5];          // map geometry of block b (224..14) and of its pooled output (112..7)
  size_t conv_off[13], pool_off[5];    // byte offsets of the CB8-PF tensors in the activation arena
  int conv_cin[13], conv_cout[13], conv_block[13];
  size_t total;                        // bytes
};
VggB16Layout vgg_b16_layout(int n) {
  VggB16Layout L;
  size_t off = 0;
  int cin = 3, hw = 224, ci = 0;
  for (int b = 0; b < 5; ++b) {
    L.geo[b] = umpr_pf(n, hw, hw);
    for (int j = 0; j < kConvPerBlock[b]; ++j) {
      L.conv_cin[ci] = cin; L.conv_cout[ci] = kBlockCh[b]; L.conv_block[ci] = b;
      L.conv_off[ci] = off;
      off += align_up(umpr_pf_bytes(L.geo[b], kBlockCh[b]), 1024);
      cin = kBlockCh[b];
      ++ci;
    }
    hw /= 2;
    L.pool_geo[b] = umpr_pf(n, hw, hw);
    L.pool_off[b] = off;
    off += align_up(umpr_pf_bytes(L.pool_geo[b], kBlockCh[b]), 1024);
  }
  L.total = off;
  return L;
}
// packed weights of the 12 MFMA layers (conv index 1..12), one region each
struct B16Packs { size_t off[13]; size_t total; };
B16Packs b16_packs() {
  B16Packs P;
  size_t off = 0;
  int cin = 64, ci = 1;
  P.off[0] = 0;
  for (int b = 0; b < 5; ++b)
    for (int j = (b == 0 ? 1 : 0); j < kConvPerBlock[b]; ++j, ++ci) {
      P.off[ci] = off;
      off += align_up(umpr_conv_bf16_pack_bytes(cin, kBlockCh[b]), 1024);
      cin = kBlockCh[b];
    }
  P.total = off;
  return P;
}
size_t b16_pack_bytes() { return b16_packs().total; }
int b16_pack_all(const VggB16Layout& L, const float* const* params, int transposed, void* wpack, hipStream_t s) {
  const B16Packs P = b16_packs();
  const float* w[12]; int cin[12], cout[12], width[12];
  for (int ci = 1; ci < 13; ++ci) {
    w[ci - 1] = params[2 * ci]; cin[ci - 1] = L.conv_cin[ci]; cout[ci - 1] = L.conv_cout[ci]; width[ci - 1] = L.geo[L.conv_block[ci]].W;
  }
  return umpr_conv_bf16_pack_all(w, cin, cout, width, 12, transposed, wpack, P.off + 1, s);
}
size_t b16_wgrad_ws_bytes(int n) {
  const VggB16Layout L = vgg_b16_layout(n);
  size_t m = umpr_conv1_bf16_wgrad_ws_bytes();
  for (int i = 1; i < 13; ++i) {
    const size_t b = umpr_wgrad_bf16_ws_bytes(L.geo[L.conv_block[i]], L.conv_cin[i], L.conv_cout[i]);
    if (b > m) m = b;
  }
  return align_up(m, 1024);
}
inline char* BP(void* p) { return static_cast<char*>(p); }
inline const char* CBP(const void* p) { return static_cast<const char*>(p); }
}  // namespace

size_t umpr_bf16_tensor_bytes(int N, int C, int H_, int W) { return umpr_pf_bytes(umpr_pf(N, H_, W), C); }
int umpr_bf16_from_nchw_f32(const float* x, void* y, int N, int C, int H_, int W, void* stream) {
  UMPR_REQUIRE(N > 0 && C > 0 && H_ > 0 && W > 0 && x && y, "bf16_from_nchw_f32: bad arguments");
  return umpr_nchw_to_cb8(x, y, umpr_pf(N, H_, W), C, S(stream));
}
int umpr_bf16_to_nchw_f32(const void* x, float* y, int N, int C, int H_, int W, void* stream) {
  UMPR_REQUIRE(N > 0 && C > 0 && H_ > 0 && W > 0 && x && y, "bf16_to_nchw_f32: bad arguments");
  return umpr_cb8_to_nchw(x, y, umpr_pf(N, H_, W), C, S(stream));
}
size_t umpr_conv3x3_bf16_ws_bytes(int N, int Cin, int Cout, int H_, int W) {
  const size_t a = umpr_conv_bf16_pack_bytes(Cin, Cout), b = umpr_wgrad_bf16_ws_bytes(umpr_pf(N, H_, W), Cin, Cout);
  return align_up(a > b ? a : b, 1024);
}
int umpr_conv3x3_bf16_fwd(const void* x, const float* w, const float* bias, void* y, int N, int Cin, int H_, int W,
                          int Cout, int relu, void* ws, size_t ws_bytes, void* stream) {
  return umpr_conv_bf16_run(x, w, 0, bias, nullptr, y, umpr_pf(N, H_, W), Cin, Cout, relu, ws, ws_bytes, S(stream));
}
int umpr_conv3x3_bf16_bwd_data(const void* dy, const float* w, const void* mask_src, void* dx, int N, int Cin, int H_,
                               int W, int Cout, void* ws, size_t ws_bytes, void* stream) {
  return umpr_conv_bf16_run(dy, w, 1, nullptr, mask_src, dx, umpr_pf(N, H_, W), Cin, Cout, 0, ws, ws_bytes, S(stream));
}
int umpr_conv3x3_bf16_bwd_weight(const void* dy, const void* x, float* dw, float* db, int N, int Cin, int H_, int W,
                                 int Cout, void* ws, size_t ws_bytes, void* stream) {
  return umpr_wgrad_bf16_run(dy, x, dw, db, umpr_pf(N, H_, W), Cin, Cout, 0, static_cast<float*>(ws), ws_bytes, S(stream));
}
int umpr_maxpool2_bf16_fwd(const void* x, void* y, int N, int C, int H_, int W, void* stream) {
  UMPR_REQUIRE((H_ % 2) == 0 && (W % 2) == 0 && (C % 8) == 0, "maxpool2_bf16: bad shape");
  return umpr_maxpool2_bf16_fwd_run(x, y, umpr_pf(N, H_, W), umpr_pf(N, H_ / 2, W / 2), C, S(stream));
}
int umpr_maxpool2_bf16_bwd_relu(const void* x, const void* dy, void* dx, int N, int C, int H_, int W, void* stream) {
  UMPR_REQUIRE((H_ % 2) == 0 && (W % 2) == 0 && (C % 8) == 0, "maxpool2_bf16: bad shape");
  return umpr_maxpool2_bf16_bwd_run(x, dy, dx, umpr_pf(N, H_, W), umpr_pf(N, H_ / 2, W / 2), C, S(stream));
}

size_t umpr_vgg16_bf16_act_bytes(int n_img) { return vgg_b16_layout(n_img).total; }
// forward scratch: [packed weights]
size_t umpr_vgg16_bf16_fwd_ws_bytes(int n_img) { (void)n_img; return b16_pack_bytes(); }
// backward scratch: [3 rotating gradient slots][packed weights][wgrad slabs]
size_t umpr_vgg16_bf16_bwd_ws_bytes(int n_img) {
  const size_t slot = align_up(umpr_pf_bytes(umpr_pf(n_img, 224, 224), 64), 1024);
  return 3 * slot + b16_pack_bytes() + b16_wgrad_ws_bytes(n_img);
}

// Convolutional stage in bf16 (activations kept in `acts` for the backward pass); pool5 [n][25088] fp32 out, in the
// NCHW flatten order the classifier's fc1 expects (torchvision: x.flatten(1) of [n][512][7][7]).
int umpr_vgg16_bf16_features_fwd(const float* images, const float* const* params, int n, void* acts, float* pool5,
                                 void* ws, size_t ws_bytes, void* stream) {
  UMPR_REQUIRE(n > 0 && images && params && acts && pool5, "vgg16_bf16_features_fwd: bad arguments");
  UMPR_REQUIRE(ws_bytes >= umpr_vgg16_bf16_fwd_ws_bytes(n), "vgg16_bf16_features_fwd: workspace too small");
  const VggB16Layout L = vgg_b16_layout(n);
  hipStream_t s = S(stream);
  void* wpack = ws;
  {  // the zero guards of all 18 activation tensors in one launch
    void* bases[18]; UmprPF geos[18]; int ch[18];
    for (int i = 0; i < 13; ++i) { bases[i] = BP(acts) + L.conv_off[i]; geos[i] = L.geo[L.conv_block[i]]; ch[i] = L.conv_cout[i]; }
    for (int b = 0; b < 5; ++b) { bases[13 + b] = BP(acts) + L.pool_off[b]; geos[13 + b] = L.pool_geo[b]; ch[13 + b] = kBlockCh[b]; }
    if (int rc = umpr_pf_zero_guards_multi(bases, geos, ch, 18, s)) return rc;
  }
  const B16Packs PK = b16_packs();
  if (int rc = b16_pack_all(L, params, 0, wpack, s)) return rc;
  // first layer (3 input channels, K = 27 padded to 32): its own kernel, fp32 image in, bf16 CB8-PF out
  if (int rc = umpr_conv1_bf16_fwd(images, params[0], params[1], BP(acts) + L.conv_off[0], L.geo[0], s, false)) return rc;
  const void* x = BP(acts) + L.conv_off[0];
  int ci = 1;
  for (int b = 0; b < 5; ++b) {
    for (int j = (b == 0 ? 1 : 0); j < kConvPerBlock[b]; ++j, ++ci) {
      void* y = BP(acts) + L.conv_off[ci];
      if (int rc = umpr_conv_bf16_run(x, nullptr, 0, params[2 * ci + 1], nullptr, y, L.geo[b], L.conv_cin[ci],
                                      L.conv_cout[ci], 1, BP(wpack) + PK.off[ci], PK.total - PK.off[ci], s, false)) return rc;
      x = y;
    }
    void* y = BP(acts) + L.pool_off[b];
    if (int rc = umpr_maxpool2_bf16_fwd_run(x, y, L.geo[b], L.pool_geo[b], kBlockCh[b], s, false)) return rc;
    x = y;
  }
  return umpr_cb8_to_nchw(x, pool5, L.pool_geo[4], 512, s);
}

int umpr_vgg16_bf16_features_bwd(const float* images, const float* const* params, int n, const void* acts,
                                 const float* d_pool5, float* const* grads, void* ws, size_t ws_bytes, void* stream) {
  UMPR_REQUIRE(ws_bytes >= umpr_vgg16_bf16_bwd_ws_bytes(n), "vgg16_bf16_features_bwd: workspace too small");
  const VggB16Layout L = vgg_b16_layout(n);
  hipStream_t s = S(stream);
  const size_t slot = align_up(umpr_pf_bytes(umpr_pf(n, 224, 224), 64), 1024);
  char* buf[3] = {BP(ws), BP(ws) + slot, BP(ws) + 2 * slot};
  void* wpack = BP(ws) + 3 * slot;
  float* slabs = reinterpret_cast<float*>(BP(ws) + 3 * slot + b16_pack_bytes());
  const size_t slab_bytes = b16_wgrad_ws_bytes(n);
  // weight gradients on the library's side stream, exactly as in the fp32 path (umpr_vgg16_features_bwd): `ready[ci]`
  // gates the wgrad of layer ci behind its output gradient, `done[ci]` gates the reuse of that gradient's slot
  const bool side = g_wgrad_side && g_wside.init();
  hipStream_t sw = side ? g_wside.stream : s;
  int slot_reader[3] = {-1, -1, -1};
  int nxt = 0;
  auto claim = [&]() -> int {
    const int k = nxt; nxt = (nxt + 1) % 3;
    if (side && slot_reader[k] >= 0) { (void)hipStreamWaitEvent(s, g_wside.done[slot_reader[k]], 0); slot_reader[k] = -1; }
    return k;
  };
  const B16Packs PK = b16_packs();
  if (int rc = b16_pack_all(L, params, 1, wpack, s)) return rc;
  // gradient w.r.t. the pooled 7x7 map arrives in fp32 NCHW order from the classifier
  int gs = claim();
  if (int rc = umpr_nchw_to_cb8(d_pool5, buf[gs], L.pool_geo[4], 512, s)) return rc;
  int ci = 12;
  for (int b = 4; b >= 0; --b) {
    // pool backward + ReLU mask of the conv output that fed the pool -> gradient w.r.t. that conv's pre-activation
    int cs = claim();
    if (int rc = umpr_maxpool2_bf16_bwd_run(CBP(acts) + L.conv_off[ci], buf[gs], buf[cs], L.geo[b], L.pool_geo[b],
                                            kBlockCh[b], s)) return rc;
    gs = cs;
    for (int j = kConvPerBlock[b] - 1; j >= 0; --j, --ci) {
      const int cin = L.conv_cin[ci], cout = L.conv_cout[ci];
      if (side) {
        (void)hipEventRecord(g_wside.ready[ci], s);
        (void)hipStreamWaitEvent(sw, g_wside.ready[ci], 0);
      }
      if (ci == 0) {
        if (int rc = umpr_conv1_bf16_wgrad(buf[gs], images, grads[0], grads[1], L.geo[0], 0, slabs, slab_bytes, sw)) return rc;
      } else {
        const void* xin = j == 0 ? CBP(acts) + L.pool_off[b - 1] : CBP(acts) + L.conv_off[ci - 1];
        if (int rc = umpr_wgrad_bf16_run(buf[gs], xin, grads[2 * ci], grads[2 * ci + 1], L.geo[b], cin, cout, 0, slabs,
                                         slab_bytes, sw)) return rc;
      }
      if (side) {
        (void)hipEventRecord(g_wside.done[ci], sw);
        slot_reader[gs] = ci;
      }
      if (ci == 0) break;
      const void* xin = j == 0 ? CBP(acts) + L.pool_off[b - 1] : CBP(acts) + L.conv_off[ci - 1];
      // the input came straight from a conv+ReLU (j > 0): mask by it; from a pool (j == 0): the pool backward masks
      const void* mask = j > 0 ? xin : nullptr;
      cs = claim();
      if (int rc = umpr_conv_bf16_run(buf[gs], nullptr, 1, nullptr, mask, buf[cs], L.geo[b], cin, cout, 0,
                                      BP(wpack) + PK.off[ci], PK.total - PK.off[ci], s)) return rc;
      gs = cs;
    }
    if (g_block_cb) g_block_cb(b, g_block_user);   // every weight-gradient kernel of block b is enqueued (on sw)
    if (ci == 0 && b == 0) break;
  }
  if (side) (void)hipStreamWaitEvent(s, g_wside.done[0], 0);   // every weight gradient is complete behind this wait
  return 0;
}

// ------------------------------------------------------------------------------------------------ head
int umpr_head_fwd(const float* rr, const float* c_u, const float* c_i, const float* prefer_pos,
                  const float* prefer_neg, const float* vgg, const float* pos_v_emb, const float* neg_v_emb,
                  const float* lin_w, const float* lin_b, const float* fus_w, const float* fus_b, const float* labels,
                  float loss_v_rate, int B, int V, int P, float* pred, float* loss, float* z, float* img_emb,
                  float* pos_match, float* neg_match, float* posneg_emb, void* stream) {
  UMPR_REQUIRE(B > 0 && V >= 0 && P >= 0, "head_fwd: bad shape");
  UmprHead h;
  memset(&h, 0, sizeof(h));
  h.rr = rr; h.c_u = c_u; h.c_i = c_i; h.pp = prefer_pos; h.pn = prefer_neg; h.vgg = vgg; h.pos_v = pos_v_emb;
  h.neg_v = neg_v_emb; h.lw = lin_w; h.lb = lin_b; h.fw = fus_w; h.fb = fus_b; h.labels = labels; h.rate = loss_v_rate;
  h.B = B; h.V = V; h.P = P; h.F = 1000; h.pred = pred; h.loss = loss; h.z = z; h.img_emb = img_emb;
  h.pos_match = pos_match; h.neg_match = neg_match; h.posneg_emb = posneg_emb;
  return umpr_head_launch(h, 0, S(stream));
}
int umpr_head_bwd(const float* rr, const float* c_u, const float* c_i, const float* prefer_pos,
                  const float* prefer_neg, const float* vgg, const float* pos_v_emb, const float* neg_v_emb,
                  const float* lin_w, const float* fus_w, const float* labels, float loss_v_rate, int B, int V, int P,
                  const float* pred, const float* z, const float* img_emb, const float* pos_match,
                  const float* neg_match, const float* posneg_emb, const float* d_loss, const float* d_pred,
                  float* d_rr, float* d_cu, float* d_ci, float* d_pp, float* d_pn, float* d_vgg, float* d_pos_v,
                  float* d_neg_v, float* d_lin_w, float* d_lin_b, float* d_fus_w, float* d_fus_b, void* stream) {
  UMPR_REQUIRE(((size_t)B + 3 * (size_t)B * V + 2 * V) * sizeof(float) <= 60000, "head_bwd: batch too large for one workgroup");
  UmprHead h;
  memset(&h, 0, sizeof(h));
  h.rr = rr; h.c_u = c_u; h.c_i = c_i; h.pp = prefer_pos; h.pn = prefer_neg; h.vgg = vgg; h.pos_v = pos_v_emb;
  h.neg_v = neg_v_emb; h.lw = lin_w; h.fw = fus_w; h.labels = labels; h.rate = loss_v_rate;
  h.B = B; h.V = V; h.P = P; h.F = 1000;
  h.pred = const_cast<float*>(pred); h.z = const_cast<float*>(z); h.img_emb = const_cast<float*>(img_emb);
  h.pos_match = const_cast<float*>(pos_match); h.neg_match = const_cast<float*>(neg_match);
  h.posneg_emb = const_cast<float*>(posneg_emb);
  h.d_loss = d_loss; h.d_pred = d_pred; h.d_rr = d_rr; h.d_cu = d_cu; h.d_ci = d_ci; h.d_pp = d_pp; h.d_pn = d_pn;
  h.d_vgg = d_vgg; h.d_pos_v = d_pos_v; h.d_neg_v = d_neg_v; h.d_lw = d_lin_w; h.d_lb = d_lin_b; h.d_fw = d_fus_w;
  h.d_fb = d_fus_b;
  return umpr_head_launch(h, 1, S(stream));
}

// ------------------------------------------------------------------------------------------------ debug
namespace {
__global__ void poison_lds_kernel(int* sink) {
  __shared__ float s[163840 / 4];  // all 160 KiB of the CU
  for (int e = threadIdx.x; e < 163840 / 4; e += blockDim.x) s[e] = __int_as_float(0x7fc00000);  // quiet NaN
  __syncthreads();
  if (s[(threadIdx.x * 97) % (163840 / 4)] == 0.f) sink[0] = 1;  // keep the stores alive
}
}  // namespace
int umpr_debug_poison_lds(void* sink, void* stream) {
  poison_lds_kernel<<<1024, 256, 0, S(stream)>>>(static_cast<int*>(sink));
  UMPR_LAUNCH_CHECK("poison_lds");
  return 0;
}

// ------------------------------------------------------------------------------------------------ profiling
int umpr_profile_enable(int on) { g_prof_on = on != 0; return 0; }
int umpr_profile_reset(void) {
  for (auto& r : g_prof) g_prof_pool.push_back(r);
  g_prof.clear();
  for (int i = 0; i < UMPR_K_COUNT; ++i) { g_prof_ms[i] = 0; g_prof_work[i] = 0; g_prof_n[i] = 0; }
  return 0;
}
int umpr_profile_read(int family, double* total_ms, double* total_work, long* launches) {
  UMPR_REQUIRE(family >= 0 && family < UMPR_K_COUNT, "profile_read: bad family %d", family);
  for (auto& r : g_prof) {  // fold finished records
    if (hipEventSynchronize(r.e1) != hipSuccess) { umpr_set_error("profile_read: event sync failed"); return -2; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    g_prof_ms[r.family] += ms; g_prof_work[r.family] += r.work; g_prof_n[r.family] += 1;
    g_prof_pool.push_back(r);
  }
  g_prof.clear();
  if (total_ms) *total_ms = g_prof_ms[family];
  if (total_work) *total_work = g_prof_work[family];
  if (launches) *launches = g_prof_n[family];
  return 0;
}

// ------------------------------------------------------------------------------------------------ Adam
int umpr_bce_head_fwd(const float* att, long ld, const float* w, const float* b, const float* target, int B, int K,
                      float* result, float* loss, float* ws, size_t ws_bytes, void* stream) {
  return umpr_bce_head_fwd_impl(att, ld, w, b, target, B, K, result, loss, ws, ws_bytes, S(stream));
}
int umpr_bce_head_bwd(const float* att, long ld, const float* w, const float* result, const float* target,
                      const float* d_result, const float* d_loss, int B, int K, float* d_att, long ld_d, float* dw,
                      float* db, float* ws, size_t ws_bytes, void* stream) {
  return umpr_bce_head_bwd_impl(att, ld, w, result, target, d_result, d_loss, B, K, d_att, ld_d, dw, db, ws, ws_bytes,
                                S(stream));
}

int umpr_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                   double eps, double weight_decay, long step, double grad_scale, void* stream) {
  UMPR_REQUIRE(step >= 1 && n >= 0, "adam: bad step/n");
  if (n == 0) return 0;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const double step_size = lr / bc1;
  const double inv_bc2_sqrt = 1.0 / sqrt(bc2);
  return umpr_adam_impl(p, g, m, v, n, (float)grad_scale, (float)weight_decay, (float)beta1, (float)beta2, (float)eps,
                        (float)step_size, (float)inv_bc2_sqrt, S(stream));
}

int umpr_adam_step_dev(float* p, const float* g, float* m, float* v, long n, double beta1, double beta2, double eps,
                       const float* hyper, void* stream) {
  UMPR_REQUIRE(n >= 0 && hyper != nullptr, "adam_dev: bad arguments");
  if (n == 0) return 0;
  return umpr_adam_dev_impl(p, g, m, v, n, (float)beta1, (float)beta2, (float)eps, hyper, S(stream));
}

int umpr_sq_err_accumulate(const float* pred, const float* label, long n, double* acc, void* stream) {
  UMPR_REQUIRE(n >= 0 && acc != nullptr && (n == 0 || (pred != nullptr && label != nullptr)), "sq_err_accumulate: null argument");
  if (n == 0) return 0;
  return umpr_sq_err_accumulate_impl(pred, label, n, acc, S(stream));
}

}  // extern "C"
