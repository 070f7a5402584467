// The text path of UMPR.forward as TWO calls per direction instead of ~25: the whole ReviewNet (R-Net GRU over the user+item
// review pair, co-attention, the two S-Nets, textual matching - src/model.py:157-169) and the whole ControlNet (C-Net GRU over the
// ui reviews and over the pair, three C-Net heads, control S-Net, SS-Net gate - src/model.py:179-198).  Round 3: a training step
// of UMPR-R is 0.9 ms of kernels, and a host that issues every stage through Python (one ctypes call, a handful of tensor
// allocations and an autograd node each) needs longer than that to enqueue them; here the stages are issued back to back from
// C++ into ONE caller-owned arena (everything the backward pass reads) and one scratch buffer.  The stage functions are the
// per-stage entry points of api.hip, unchanged: these wrappers only carve buffers and call them in order.
#include "umpr_common.h"
#include "umpr_internal.h"
#include "../../include/umpr_hip.h"

namespace {

constexpr int D = 128, H = 64, AT = 64;

struct Carve {   // hands out 256-byte aligned pieces of one buffer; with base == nullptr it only measures
  char* base; size_t off = 0;
  explicit Carve(void* b) : base(static_cast<char*>(b)) {}
  template <typename T> T* take(size_t n) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += (n * sizeof(T) + 255) / 256 * 256;
    return p;
  }
};

struct B16Scope {   // the text path's GEMM-shaped products on the bf16 pipe while alive (per host thread)
  int on;
  explicit B16Scope(int o) : on(o) { if (on) umpr_gemm_set_b16(1); }
  ~B16Scope() { if (on) umpr_gemm_set_b16(0); }
};

struct SnetSaved { float *U, *P, *wsum, *sa; };
SnetSaved take_snet(Carve& c, long B, long S, long L) {
  SnetSaved s;
  s.U = c.take<float>(B * S * L * AT); s.P = c.take<float>(B * S * L); s.wsum = c.take<float>(B * S); s.sa = c.take<float>(B * S * D);
  return s;
}

// ---- ReviewNet arena ----------------------------------------------------------------------------------------------
struct ReviewArena {
  float *gru_out, *gru_saved, *T, *soft_u, *soft_i, *colmax, *rowmax, *repr_u, *repr_i, *merged;
  int32_t *argcol, *argrow;
  SnetSaved su, si;
  size_t bytes;
};
ReviewArena review_arena(void* base, long B, long S, long L) {
  Carve c(base);
  const long N = B * S, SL = S * L;
  ReviewArena a;
  a.gru_out = c.take<float>(2 * N * L * D);
  a.gru_saved = c.take<float>(2 * 2 * N * L * 4 * H);
  a.T = c.take<float>(B * SL * D);
  a.soft_u = c.take<float>(B * SL); a.soft_i = c.take<float>(B * SL);
  a.colmax = c.take<float>(B * SL); a.rowmax = c.take<float>(B * SL);
  a.argcol = c.take<int32_t>(B * SL); a.argrow = c.take<int32_t>(B * SL);
  a.repr_u = c.take<float>(B * 2 * D); a.repr_i = c.take<float>(B * 2 * D);
  a.merged = c.take<float>(B * D);
  a.su = take_snet(c, B, S, L); a.si = take_snet(c, B, S, L);
  a.bytes = c.off;
  return a;
}

size_t max2(size_t a, size_t b) { return a > b ? a : b; }

// ---- ControlNet arena ---------------------------------------------------------------------------------------------
struct HeadSaved { float *cmax, *sp, *vp, *fin; int32_t* argl; };
struct ControlArena {
  float *gru_ui, *saved_ui, *gru_pair, *saved_pair, *senti, *vs;
  HeadSaved h[3];   // ui, user, item
  SnetSaved sn;
  size_t bytes;
};
ControlArena control_arena(void* base, long B, long S_ui, long L_ui, long S, long L, long KC, long V) {
  Carve c(base);
  const long Nui = B * S_ui, N = B * S;
  ControlArena a;
  a.gru_ui = c.take<float>(Nui * L_ui * D);
  a.saved_ui = c.take<float>(2 * Nui * L_ui * 4 * H);
  a.gru_pair = c.take<float>(2 * N * L * D);
  a.saved_pair = c.take<float>(2 * 2 * N * L * 4 * H);
  const long ss[3] = {S_ui, S, S};
  for (int q = 0; q < 3; ++q) {
    a.h[q].cmax = c.take<float>(B * ss[q] * KC); a.h[q].argl = c.take<int32_t>(B * ss[q] * KC);
    a.h[q].sp = c.take<float>(B * ss[q] * V); a.h[q].vp = c.take<float>(B * ss[q] * V); a.h[q].fin = c.take<float>(B * V);
  }
  a.sn = take_snet(c, B, S_ui, L_ui);
  a.senti = c.take<float>(B * S_ui); a.vs = c.take<float>(B * V);
  a.bytes = c.off;
  return a;
}

// UMPR_POISON_WS=1 (debugging aid): every entry point below first fills its workspace - and the forward ones their arena - with
// 0xFF bytes (NaN as floats), so that any read of memory the call did not write itself shows up as NaN in its results instead of
// depending on what the allocator handed out (tests/test_gpu_parity.py::test_text_path_reads_no_uninitialised_memory).
// UMPR_DEBUG_SYNC=<bit mask> (debugging aid): device-wide synchronisation at numbered points of umpr_review_net_bwd (bit 0: on
// entry, 1: after the merge, 2 / 3: after the user / item S-Net, 4: after the co-attention) - narrows a cross-stream dependency down
void debug_sync(int point) {
  static const int mask = umpr_env_int("UMPR_DEBUG_SYNC", 0);
  if (mask & (1 << point)) (void)hipDeviceSynchronize();
}
int poison(void* p, size_t bytes, void* stream) {
  static const bool on = umpr_env_int("UMPR_POISON_WS", 0) == 1;
  if (!on || !p || !bytes) return 0;
  if (hipMemsetAsync(p, 0xFF, bytes, static_cast<hipStream_t>(stream)) != hipSuccess) { umpr_set_error("poison: memset failed"); return -2; }
  return 0;
}

}  // namespace

extern "C" {

// ids of the user and of the item reviews as ONE [2N][L] tensor for the GRUs the two share (src/model.py:45-46, 183-184)
int umpr_concat_ids(const int64_t* ids_u, const int64_t* ids_i, long n_each, int64_t* dst, void* stream) {
  UMPR_REQUIRE(ids_u && ids_i && dst && n_each > 0, "concat_ids: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(dst, ids_u, (size_t)n_each * 8, hipMemcpyDeviceToDevice, s) != hipSuccess ||
      hipMemcpyAsync(dst + n_each, ids_i, (size_t)n_each * 8, hipMemcpyDeviceToDevice, s) != hipSuccess) {
    umpr_set_error("concat_ids: copy failed");
    return -2;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------------ ReviewNet
size_t umpr_review_net_arena_bytes(int B, int S, int L) { return review_arena(nullptr, B, S, L).bytes; }
size_t umpr_review_net_ws_bytes(int B, int S, int L, int E) {
  const long N = (long)B * S, SL = (long)S * L;
  Carve c(nullptr);
  c.take<float>(2 * N * L * D);   // dG (also the GRUs' dout)
  c.take<float>(B * 2 * D); c.take<float>(B * 2 * D);   // d_repr_u, d_repr_i
  c.take<float>(B * SL); c.take<float>(B * SL);         // d_soft_u, d_soft_i
  size_t stage = umpr_embed_gru_bidir_ws_bytes((int)(2 * N), L, E);
  stage = max2(stage, umpr_coattention_fwd_ws_bytes(B, (int)SL));
  stage = max2(stage, umpr_coattention_bwd_ws_bytes(B, (int)SL));
  stage = max2(stage, umpr_snet_bwd_ws_bytes(B, S, L));
  stage = max2(stage, umpr_review_merge_bwd_ws_bytes(B));
  return c.off + stage + 256;
}

// params: w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r, M, Ms_u, Ws_u, Ms_i, Ws_i, W_u, W_i
int umpr_review_net_fwd(const int64_t* ids_pair, const float* emb, int E, const float* const* P, const int32_t* lengths,
                        const int32_t* order, int B, int S, int L, int b16_gemm, int b16_scores, int need_grad, void* arena,
                        float* out, float* ws, size_t ws_bytes, void* stream) {
  UMPR_REQUIRE(ids_pair && emb && P && lengths && order && arena && out && B > 0 && S > 0 && L > 0, "review_net_fwd: bad arguments");
  UMPR_REQUIRE(ws_bytes >= umpr_review_net_ws_bytes(B, S, L, E), "review_net_fwd: workspace too small");
  if (poison(ws, ws_bytes, stream) || poison(arena, umpr_review_net_arena_bytes(B, S, L), stream)) return -2;
  const ReviewArena A = review_arena(arena, B, S, L);
  const int N = B * S, SL = S * L;
  B16Scope b16(b16_gemm);
  if (int rc = umpr_embed_gru_bidir_fwd(ids_pair, emb, E, P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], lengths, order, order,
                                        2 * N, L, A.gru_out, need_grad ? A.gru_saved : nullptr, ws, ws_bytes, stream)) return rc;
  const float* gru_u = A.gru_out;
  const float* gru_i = A.gru_out + (size_t)N * L * D;
  if (int rc = (b16_scores ? umpr_coattention_fwd_bf16 : umpr_coattention_fwd)(gru_u, gru_i, P[8], B, SL, A.T, A.soft_u, A.soft_i, A.repr_u,
                                                                         2 * D, A.repr_i, 2 * D, A.colmax, A.argcol, A.rowmax,
                                                                         A.argrow, ws, ws_bytes, stream)) return rc;
  if (int rc = umpr_snet_fwd(gru_u, P[9], P[10], A.soft_u, L, B, S, L, A.su.U, A.su.P, A.su.wsum, A.su.sa, A.repr_u + D, 2 * D, stream)) return rc;
  if (int rc = umpr_snet_fwd(gru_i, P[11], P[12], A.soft_i, L, B, S, L, A.si.U, A.si.P, A.si.wsum, A.si.sa, A.repr_i + D, 2 * D, stream)) return rc;
  static const bool merge_small = umpr_env_on("UMPR_MERGE_SMALL");
  if (B <= 256 && merge_small)   // one kernel writes the caller's tensor and the arena's copy for backward
    return umpr_review_merge_fwd_impl(A.repr_u, A.repr_i, P[13], P[14], B, out, A.merged, static_cast<hipStream_t>(stream));
  if (int rc = umpr_review_merge_fwd(A.repr_u, A.repr_i, P[13], P[14], B, A.merged, stream)) return rc;
  if (hipMemcpyAsync(out, A.merged, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)) != hipSuccess) {
    umpr_set_error("review_net_fwd: copy failed");
    return -2;
  }
  return 0;
}

// grads: 15 pointers in the order of params, each overwritten
int umpr_review_net_bwd(const int64_t* ids_pair, const float* emb, int E, const float* const* P, const int32_t* lengths,
                        const int32_t* order, int B, int S, int L, int b16_gemm, const void* arena, const float* d_out,
                        float* const* G, float* ws, size_t ws_bytes, void* stream) {
  UMPR_REQUIRE(ids_pair && emb && P && G && lengths && order && arena && d_out, "review_net_bwd: bad arguments");
  UMPR_REQUIRE(ws_bytes >= umpr_review_net_ws_bytes(B, S, L, E), "review_net_bwd: workspace too small");
  if (poison(ws, ws_bytes, stream)) return -2;
  const ReviewArena A = review_arena(const_cast<void*>(arena), B, S, L);
  const long N = (long)B * S, SL = (long)S * L;
  Carve c(ws);
  float* dG = c.take<float>(2 * N * L * D);
  float* d_repr_u = c.take<float>(B * 2 * D);
  float* d_repr_i = c.take<float>(B * 2 * D);
  float* ds_u = c.take<float>(B * SL);
  float* ds_i = c.take<float>(B * SL);
  float* sws = reinterpret_cast<float*>(static_cast<char*>(static_cast<void*>(ws)) + c.off);
  const size_t swsb = ws_bytes - c.off;
  const float* gru_u = A.gru_out;
  const float* gru_i = A.gru_out + (size_t)N * L * D;
  float* dGu = dG;
  float* dGi = dG + (size_t)N * L * D;
  B16Scope b16(b16_gemm);
  debug_sync(0);
  if (int rc = umpr_review_merge_bwd(A.repr_u, A.repr_i, P[13], P[14], A.merged, d_out, B, d_repr_u, d_repr_i, G[13], G[14], sws, swsb, stream)) return rc;
  debug_sync(1);
  if (int rc = umpr_snet_bwd(gru_u, P[9], P[10], A.su.U, A.su.P, A.su.wsum, A.su.sa, d_repr_u + D, 2 * D, nullptr, B, S, L, L, dGu,
                             G[9], G[10], ds_u, sws, swsb, stream)) return rc;
  debug_sync(2);
  if (int rc = umpr_snet_bwd(gru_i, P[11], P[12], A.si.U, A.si.P, A.si.wsum, A.si.sa, d_repr_i + D, 2 * D, nullptr, B, S, L, L, dGi,
                             G[11], G[12], ds_i, sws, swsb, stream)) return rc;
  debug_sync(3);
  if (int rc = umpr_coattention_bwd(gru_u, gru_i, P[8], A.T, A.soft_u, A.soft_i, A.colmax, A.argcol, A.rowmax, A.argrow, d_repr_u,
                                    2 * D, d_repr_i, 2 * D, ds_u, ds_i, B, (int)SL, dGu, dGi, G[8], 1, sws, swsb, stream)) return rc;
  debug_sync(4);
  return umpr_embed_gru_bidir_bwd_acc(ids_pair, emb, E, P[1], P[5], lengths, order, order, (int)(2 * N), L, dG, A.gru_out, A.gru_saved,
                                      G[0], G[1], G[2], G[3], G[4], G[5], G[6], G[7], 0, sws, swsb, stream);
}

// ------------------------------------------------------------------------------------------------------ ControlNet
size_t umpr_control_net_arena_bytes(int B, int S_ui, int L_ui, int S, int L, int KC, int V) {
  return control_arena(nullptr, B, S_ui, L_ui, S, L, KC, V).bytes;
}
size_t umpr_control_net_ws_bytes(int B, int S_ui, int L_ui, int S, int L, int E, int KC, int KS, int V) {
  const long Nui = (long)B * S_ui, N = (long)B * S;
  Carve c(nullptr);
  c.take<float>(Nui * L_ui * D);     // dX_ui (the ui GRU's dout)
  c.take<float>(2 * N * L * D);      // dX_u, dX_i (the pair GRU's dout)
  c.take<float>(B * S_ui * D);       // d_self_atte
  c.take<float>(B * S_ui * V);       // d_view_p
  c.take<float>(B * V);              // d_c_out
  c.take<float>(B * D);              // zero d_senti of the control S-Net (its sentiment output is unused, model.py:185)
  c.take<float>((long)B * (S_ui > S ? S_ui : S) * (L_ui > L ? L_ui : L) * KC);   // Y of a C-Net head (forward scratch)
  size_t stage = max2(umpr_embed_gru_bidir_ws_bytes((int)(2 * N), L, E), umpr_embed_gru_bidir_ws_bytes((int)Nui, L_ui, E));
  stage = max2(stage, umpr_cnet_head_fwd_ws_bytes(B, S_ui, L_ui, KS));
  stage = max2(stage, umpr_cnet_head_fwd_ws_bytes(B, S, L, KS));
  stage = max2(stage, umpr_cnet_head_bwd_ws_bytes(B, S_ui, L_ui, KC, KS, V));
  stage = max2(stage, umpr_cnet_head_bwd_ws_bytes(B, S, L, KC, KS, V));
  stage = max2(stage, umpr_snet_bwd_ws_bytes(B, S_ui, L_ui));
  stage = max2(stage, umpr_control_gate_bwd_ws_bytes(B));
  return c.off + stage + 256;
}

// params: the C-Net GRU's eight (as above), Wc [KC][128][KS], bc, Wl [V][KC], bl, Ms, Ws, ssW [128], ssb [1]
int umpr_control_net_fwd(const int64_t* ids_ui, const int64_t* ids_pair, const float* emb, int E, const float* const* P,
                         const int32_t* len_ui, const int32_t* ord_ui, const int32_t* len_pair, const int32_t* ord_pair, int B,
                         int S_ui, int L_ui, int S, int L, int KC, int KS, int V, float thr, int b16_gemm, int need_grad, void* arena,
                         float* c_u, float* c_i, float* prefer_pos, float* prefer_neg, float* ws, size_t ws_bytes, void* stream) {
  UMPR_REQUIRE(ids_ui && ids_pair && emb && P && arena && c_u && c_i && prefer_pos && prefer_neg, "control_net_fwd: bad arguments");
  UMPR_REQUIRE(ws_bytes >= umpr_control_net_ws_bytes(B, S_ui, L_ui, S, L, E, KC, KS, V), "control_net_fwd: workspace too small");
  if (poison(ws, ws_bytes, stream) || poison(arena, umpr_control_net_arena_bytes(B, S_ui, L_ui, S, L, KC, V), stream)) return -2;
  const ControlArena A = control_arena(arena, B, S_ui, L_ui, S, L, KC, V);
  const long Nui = (long)B * S_ui, N = (long)B * S;
  hipStream_t s = static_cast<hipStream_t>(stream);
  B16Scope b16(b16_gemm);
  if (int rc = umpr_embed_gru_bidir_fwd(ids_ui, emb, E, P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], len_ui, ord_ui, ord_ui, (int)Nui,
                                        L_ui, A.gru_ui, need_grad ? A.saved_ui : nullptr, ws, ws_bytes, stream)) return rc;
  if (int rc = umpr_embed_gru_bidir_fwd(ids_pair, emb, E, P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], len_pair, ord_pair, ord_pair,
                                        (int)(2 * N), L, A.gru_pair, need_grad ? A.saved_pair : nullptr, ws, ws_bytes, stream)) return rc;
  Carve c(ws);
  float* Y = c.take<float>((long)B * (S_ui > S ? S_ui : S) * (L_ui > L ? L_ui : L) * KC);
  float* sws = reinterpret_cast<float*>(static_cast<char*>(static_cast<void*>(ws)) + c.off);
  const size_t swsb = ws_bytes - c.off;
  const float* X[3] = {A.gru_ui, A.gru_pair, A.gru_pair + (size_t)N * L * D};
  const int ss[3] = {S_ui, S, S}, ll[3] = {L_ui, L, L};
  for (int q = 0; q < 3; ++q)
    if (int rc = umpr_cnet_head_fwd(X[q], P[8], P[9], P[10], P[11], thr, B, ss[q], ll[q], KC, KS, V, Y, A.h[q].cmax, A.h[q].argl,
                                    A.h[q].sp, A.h[q].vp, A.h[q].fin, sws, swsb, stream)) return rc;
  // control S-Net weighted by view_p (model.py:185); its sentiment vector is not used: written into scratch
  if (int rc = umpr_snet_fwd(A.gru_ui, P[12], P[13], A.h[0].vp, V, B, S_ui, L_ui, A.sn.U, A.sn.P, A.sn.wsum, A.sn.sa, Y, D, stream)) return rc;
  if (int rc = umpr_control_gate_fwd(A.sn.sa, P[14], P[15], A.h[0].vp, A.h[0].fin, B, S_ui, V, A.senti, A.vs, prefer_pos, prefer_neg, stream)) return rc;
  if (hipMemcpyAsync(c_u, A.h[1].fin, (size_t)B * V * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess ||
      hipMemcpyAsync(c_i, A.h[2].fin, (size_t)B * V * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
    umpr_set_error("control_net_fwd: copy failed");
    return -2;
  }
  return 0;
}

// d_cu, d_ci, d_pp, d_pn [B][V]; grads: 16 pointers in the order of params, each overwritten
int umpr_control_net_bwd(const int64_t* ids_ui, const int64_t* ids_pair, const float* emb, int E, const float* const* P,
                         const int32_t* len_ui, const int32_t* ord_ui, const int32_t* len_pair, const int32_t* ord_pair, int B,
                         int S_ui, int L_ui, int S, int L, int KC, int KS, int V, int b16_gemm, const void* arena, const float* d_cu,
                         const float* d_ci, const float* d_pp, const float* d_pn, float* const* G, float* ws, size_t ws_bytes,
                         void* stream) {
  UMPR_REQUIRE(ids_ui && ids_pair && emb && P && G && arena && d_cu && d_ci && d_pp && d_pn, "control_net_bwd: bad arguments");
  UMPR_REQUIRE(ws_bytes >= umpr_control_net_ws_bytes(B, S_ui, L_ui, S, L, E, KC, KS, V), "control_net_bwd: workspace too small");
  if (poison(ws, ws_bytes, stream)) return -2;
  const ControlArena A = control_arena(const_cast<void*>(arena), B, S_ui, L_ui, S, L, KC, V);
  const long Nui = (long)B * S_ui, N = (long)B * S;
  hipStream_t s = static_cast<hipStream_t>(stream);
  Carve c(ws);
  float* dX_ui = c.take<float>(Nui * L_ui * D);
  float* dX_pair = c.take<float>(2 * N * L * D);
  float* d_sa = c.take<float>(B * S_ui * D);
  float* d_vp = c.take<float>(B * S_ui * V);
  float* d_cout = c.take<float>(B * V);
  float* zero_senti = c.take<float>(B * D);
  float* sws = reinterpret_cast<float*>(static_cast<char*>(static_cast<void*>(ws)) + c.off);
  const size_t swsb = ws_bytes - c.off;
  B16Scope b16(b16_gemm);
  if (hipMemsetAsync(zero_senti, 0, (size_t)B * D * sizeof(float), s) != hipSuccess) { umpr_set_error("control_net_bwd: memset"); return -2; }
  if (int rc = umpr_control_gate_bwd(A.sn.sa, P[14], A.h[0].vp, A.h[0].fin, A.senti, A.vs, d_pp, d_pn, B, S_ui, V, d_sa, d_vp, d_cout,
                                     G[14], G[15], sws, swsb, stream)) return rc;
  if (int rc = umpr_snet_bwd(A.gru_ui, P[12], P[13], A.sn.U, A.sn.P, A.sn.wsum, A.sn.sa, zero_senti, D, d_sa, B, S_ui, L_ui, V, dX_ui,
                             G[12], G[13], nullptr, sws, swsb, stream)) return rc;
  const float* X[3] = {A.gru_ui, A.gru_pair, A.gru_pair + (size_t)N * L * D};
  float* dX[3] = {dX_ui, dX_pair, dX_pair + (size_t)N * L * D};
  const float* dfin[3] = {d_cout, d_cu, d_ci};
  const float* dvp[3] = {d_vp, nullptr, nullptr};
  const int ss[3] = {S_ui, S, S}, ll[3] = {L_ui, L, L};
  for (int q = 0; q < 3; ++q)   // the ui head adds onto the S-Net's dX and writes the weight gradients; the other two add onto those
    if (int rc = umpr_cnet_head_bwd(X[q], P[8], P[10], A.h[q].cmax, A.h[q].argl, A.h[q].sp, A.h[q].vp, dfin[q], dvp[q], B, ss[q], ll[q],
                                    KC, KS, V, dX[q], q == 0 ? 1 : 0, q == 0 ? 0 : 1, G[8], G[9], G[10], G[11], sws, swsb, stream)) return rc;
  if (int rc = umpr_embed_gru_bidir_bwd_acc(ids_ui, emb, E, P[1], P[5], len_ui, ord_ui, ord_ui, (int)Nui, L_ui, dX_ui, A.gru_ui, A.saved_ui,
                                            G[0], G[1], G[2], G[3], G[4], G[5], G[6], G[7], 0, sws, swsb, stream)) return rc;
  return umpr_embed_gru_bidir_bwd_acc(ids_pair, emb, E, P[1], P[5], len_pair, ord_pair, ord_pair, (int)(2 * N), L, dX_pair, A.gru_pair,
                                      A.saved_pair, G[0], G[1], G[2], G[3], G[4], G[5], G[6], G[7], 1, sws, swsb, stream);
}

}  // extern "C"
