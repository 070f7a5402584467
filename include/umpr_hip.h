/* libumpr_hip.so - C ABI of the MI355X-native UMPR hot path.
 *
 * The reference (iamwinter/UMPR) has no FFI layer: its hot path is the Python nn.Module API of src/model.py
 * (UMPR.__init__ model.py:233-255, UMPR.forward model.py:257-278) and all arithmetic runs inside torch ATen /
 * torchvision.  This header is the boundary a maintainer binds instead (ctypes stub in INTEGRATION.md): each
 * entry point replaces the torch ops of the cited reference lines with hand-written gfx950 kernels.
 *
 * Conventions: every pointer is DEVICE memory owned by the caller (plain pointers and sizes, no torch types);
 * fp32 unless noted; token ids are int64 (the reference's LongTensor); lengths / permutations are int32 arrays the
 * host computes (the reference keeps lengths on the host too, model.py:18); `stream` is a hipStream_t passed as
 * void*; workspaces come from the caller (size queries below) - the library never allocates or frees.
 * Return value: 0 = ok, <0 = error (message: umpr_last_error(), thread-local).  Hidden sizes are the reference's
 * configuration constants: gru_size 64 (2u = 128), self_atte_size 64 (config.py:34-35).
 */
#ifndef UMPR_HIP_H
#define UMPR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* umpr_version(void);
const char* umpr_last_error(void);
/* name of device `dev`, number of CUs, bytes of HBM; used by bench.py to print what it ran on */
int umpr_device_info(int dev, char* name, int name_len, int* compute_units, size_t* hbm_bytes);

/* ---- generic fp32 MFMA GEMM: C = act(alpha * op(A) op(B) + bias + (accumulate ? C : 0)) --------------------
 * Replaces torch.mm/addmm/Linear call sites on the path (model.py:50,75,120,154-155,167-168; VGG classifier).
 * op(A)(m,k) = transA ? A[k*lda+m] : A[m*lda+k];  op(B)(k,n) = transB ? B[n*ldb+k] : B[k*ldb+n].
 * bias_mode: 0 none, 1 bias[n], 2 bias[m].  act: 0 none, 1 relu, 2 tanh, 3 sigmoid.
 * ws/ws_bytes: optional split-K workspace (used when the grid would not fill the GPU). */
int umpr_gemm_f32(const float* A, long lda, int transA, const float* B, long ldb, int transB, float* C, long ldc,
                  int M, int N, int K, const float* bias, int bias_mode, int act, int accumulate, float alpha,
                  float* ws, size_t ws_bytes, void* stream);

/* Mixed precision for the GEMM-shaped products of the text path (BASELINE.json configs[4]; not in the reference, which
 * would get it from torch.autocast): while set on the calling host thread, every product the library issues through this
 * GEMM (GRU input projections, co-attention / S-Net / C-Net projections and their gradients) rounds its operands to bf16 on
 * the way into LDS and runs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Set it around a call, clear it after. */
int umpr_set_gemm_bf16(int on);

/* Inference hint for the fp32 convolution stack (the reference's counterpart is torch.no_grad() around evaluate.py:8-13):
 * while set on the calling host thread, forward convolutions may use the F(4x4,3x3) Winograd tile that training reserves for
 * the backward pass - its rounding (5e-6 of max|y|) is irrelevant when no gradient will be taken through the ReLU / pool
 * decisions of this forward.  Predictions move by ~3e-6.  Set it around umpr_vgg16_features_fwd / umpr_conv3x3_fwd. */
int umpr_set_conv_inference(int on);

/* A 2x2 max-pool will consume the output of the next umpr_conv3x3_fwd calls of this host thread (umpr_vgg16_features_fwd sets
 * it itself around conv3_3 / conv4_3 / conv5_3): the decision fix-up of the F(4x4,3x3) training forward (winograd.hip) then
 * covers the pool windows' argmax as well as the ReLU signs.  Callers that chain umpr_conv3x3_fwd + umpr_maxpool2_fwd
 * themselves set it likewise. */
int umpr_set_conv_pool_follows(int on);

/* Test / tooling aid: number of outputs the most recent decision fix-up pass of this host thread listed for recomputation
 * (-1: none yet).  Synchronises the device. */
long umpr_debug_wino_fix_count(void);

/* ---- K1-K3: embedding lookup + bidirectional packed GRU + the reference's double un-sort ---------------------
 * Replaces nn.Embedding (model.py:262-264) + ImprovedRnn.forward (model.py:12-21) for one review tensor.
 * ids [N*L] int64; emb [vocab][E]; GRU weights in nn.GRU layout (gate order r,z,n): w_ih [192][E], w_hh [192][64],
 * b_ih/b_hh [192], forward direction then "_reverse".  lengths [N]; order [N] = sorted_indices (descending length,
 * the tie order torch.sort produced); dst_row [N]: input row n lands in output row dst_row[n] (= sorted_indices[n]
 * for the reference's semantics; must be a permutation of 0..N-1: each output row is written - values up to its length, zeros
 * past it - by the sequence that owns it).  out [N][L][128] (zeros past each length).  saved [2][N][L][4][64] (gates for
 * backward) or NULL for inference. */
size_t umpr_embed_gru_bidir_ws_bytes(int N, int L, int E);
int umpr_embed_gru_bidir_fwd(const int64_t* ids, const float* emb, int E,
                             const float* w_ih_f, const float* w_hh_f, const float* b_ih_f, const float* b_hh_f,
                             const float* w_ih_r, const float* w_hh_r, const float* b_ih_r, const float* b_hh_r,
                             const int32_t* lengths, const int32_t* order, const int32_t* dst_row, int N, int L,
                             float* out, float* saved, float* ws, size_t ws_bytes, void* stream);
/* dout [N][L][128]; d* receive the parameter gradients (overwritten).  The embedding table is frozen
 * (nn.Embedding.from_pretrained, model.py:237) so no gradient is produced for it. */
int umpr_embed_gru_bidir_bwd(const int64_t* ids, const float* emb, int E,
                             const float* w_hh_f, const float* w_hh_r,
                             const int32_t* lengths, const int32_t* order, const int32_t* dst_row, int N, int L,
                             const float* dout, const float* out, const float* saved,
                             float* dw_ih_f, float* dw_hh_f, float* db_ih_f, float* db_hh_f,
                             float* dw_ih_r, float* dw_hh_r, float* db_ih_r, float* db_hh_r,
                             float* ws, size_t ws_bytes, void* stream);

/* The same with `accumulate` != 0 adding onto the eight gradient buffers instead of overwriting them: the reference uses
 * ONE GRU for user and item reviews (model.py:45-46) and one for the three C-Net calls (model.py:182-184); a caller that
 * owns flat gradient storage lets the second and third call accumulate in place instead of adding temporaries. */
int umpr_embed_gru_bidir_bwd_acc(const int64_t* ids, const float* emb, int E,
                                 const float* w_hh_f, const float* w_hh_r,
                                 const int32_t* lengths, const int32_t* order, const int32_t* dst_row, int N, int L,
                                 const float* dout, const float* out, const float* saved,
                                 float* dw_ih_f, float* dw_hh_f, float* db_ih_f, float* db_hh_f,
                                 float* dw_ih_r, float* dw_hh_r, float* db_ih_r, float* db_hh_r,
                                 int accumulate, float* ws, size_t ws_bytes, void* stream);

/* ---- K4-K5: R-Net co-attention (model.py:50-55) --------------------------------------------------------------
 * Gu, Gi [B][SL][128]; M [128][128].  Outputs soft_u/soft_i [B][SL], atte_u/atte_i rows of 128 written at
 * atte_x + b*ld_x (lets the caller place them inside the [B][256] concat of model.py:166-167).
 * Saved for backward: T [B][SL][128], colmax/rowmax [B][SL], argcol/argrow [B][SL] int32. */
size_t umpr_coattention_fwd_ws_bytes(int B, int SL);
int umpr_coattention_fwd(const float* Gu, const float* Gi, const float* M, int B, int SL, float* T, float* soft_u,
                         float* soft_i, float* atte_u, long ld_u, float* atte_i, long ld_i, float* colmax,
                         int32_t* argcol, float* rowmax, int32_t* argrow, float* ws, size_t ws_bytes, void* stream);
/* The same with the score contraction tanh(T G_u^T) on bf16 MFMA (operands rounded to bf16, fp32 accumulation; tanh,
 * max, argmax, softmax in fp32) - the "attention" half of BASELINE.json configs[4].  Same outputs / saved tensors, and
 * umpr_coattention_bwd consumes them unchanged. */
int umpr_coattention_fwd_bf16(const float* Gu, const float* Gi, const float* M, int B, int SL, float* T, float* soft_u,
                              float* soft_i, float* atte_u, long ld_u, float* atte_i, long ld_i, float* colmax,
                              int32_t* argcol, float* rowmax, int32_t* argrow, float* ws, size_t ws_bytes, void* stream);
size_t umpr_coattention_bwd_ws_bytes(int B, int SL);
int umpr_coattention_bwd(const float* Gu, const float* Gi, const float* M, const float* T, const float* soft_u,
                         const float* soft_i, const float* colmax, const int32_t* argcol, const float* rowmax,
                         const int32_t* argrow, const float* d_atte_u, long ld_du, const float* d_atte_i, long ld_di,
                         const float* d_soft_u, const float* d_soft_i, int B, int SL, float* dGu, float* dGi,
                         float* dM, int accumulate /* add onto dGu,dGi */, float* ws, size_t ws_bytes, void* stream);

/* ---- K6: S-Net (model.py:71-81) ------------------------------------------------------------------------------
 * X [B][S][L][128]; Ms [64][128]; Ws [64]; word_soft [B][S][wl] (wl = L for soft_u/soft_i, V for view_p).
 * Outputs: self_atte [B][S][128], senti rows at senti + b*ld_senti.  Saved: U [B][S][L][64], P [B][S][L], wsum [B][S]. */
int umpr_snet_fwd(const float* X, const float* Ms, const float* Ws, const float* word_soft, int wl, int B, int S,
                  int L, float* U, float* P, float* wsum, float* self_atte, float* senti, long ld_senti, void* stream);
size_t umpr_snet_bwd_ws_bytes(int B, int S, int L);
int umpr_snet_bwd(const float* X, const float* Ms, const float* Ws, const float* U, const float* P, const float* wsum,
                  const float* self_atte, const float* d_senti, long ld_ds, const float* d_self_atte /*or NULL*/,
                  int B, int S, int L, int wl, float* dX, float* dMs, float* dWs, float* d_word_soft /*or NULL*/,
                  float* ws, size_t ws_bytes, void* stream);

/* ---- K7: ReviewNet merge tanh(W_u [atte_u;senti_u] + W_i [atte_i;senti_i]) (model.py:166-168) ----------------
 * repr_u, repr_i [B][256]; W_u, W_i [128][256]; out [B][128]. */
int umpr_review_merge_fwd(const float* repr_u, const float* repr_i, const float* W_u, const float* W_i, int B,
                          float* out, void* stream);
size_t umpr_review_merge_bwd_ws_bytes(int B);
int umpr_review_merge_bwd(const float* repr_u, const float* repr_i, const float* W_u, const float* W_i,
                          const float* out, const float* d_out, int B, float* d_repr_u, float* d_repr_i, float* dW_u,
                          float* dW_i, float* ws, size_t ws_bytes, void* stream);

/* ---- K8: C-Net head: Conv1d(128->KC,k=KS,pad)+ReLU+max_L, Linear(KC->V)+Sigmoid, threshold, sum p^2
 * (model.py:118-125).  X [B][S][L][128]; Wc [KC][128][KS]; Wl [V][KC].  Outputs view_p [B][S][V], final [B][V].
 * Saved: Y [B][S][L][KC], cmax [B][S][KC], argl int32 [B][S][KC], sp [B][S][V]. */
size_t umpr_cnet_head_fwd_ws_bytes(int B, int S, int L, int KS);
int umpr_cnet_head_fwd(const float* X, const float* Wc, const float* bc, const float* Wl, const float* bl, float thr,
                       int B, int S, int L, int KC, int KS, int V, float* Y, float* cmax, int32_t* argl, float* sp,
                       float* view_p, float* final_, float* ws, size_t ws_bytes, void* stream);
size_t umpr_cnet_head_bwd_ws_bytes(int B, int S, int L, int KC, int KS, int V);
int umpr_cnet_head_bwd(const float* X, const float* Wc, const float* Wl, const float* cmax, const int32_t* argl,
                       const float* sp, const float* view_p, const float* d_final /*or NULL*/,
                       const float* d_view_p /*or NULL*/, int B, int S, int L, int KC, int KS, int V, float* dX,
                       int accumulate_dX, int accumulate_w /* add onto dX / the weight grads */, float* dWc,
                       float* dbc, float* dWl, float* dbl, float* ws, size_t ws_bytes, void* stream);

/* ---- K9: control gate incl. SS-Net (model.py:142-143,186-197) ------------------------------------------------ */
int umpr_control_gate_fwd(const float* self_atte, const float* w, const float* bias, const float* view_p,
                          const float* c_out, int B, int S, int V, float* senti, float* view_score,
                          float* prefer_pos, float* prefer_neg, void* stream);
size_t umpr_control_gate_bwd_ws_bytes(int B);
int umpr_control_gate_bwd(const float* self_atte, const float* w, const float* view_p, const float* c_out,
                          const float* senti, const float* view_score, const float* d_prefer_pos,
                          const float* d_prefer_neg, int B, int S, int V, float* d_self_atte, float* d_view_p,
                          float* d_c_out, float* dw, float* db, float* ws, size_t ws_bytes, void* stream);

/* ---- the text path in two calls per direction (round 3; csrc/text_path.hip) -----------------------------------------------
 * The whole ReviewNet (src/model.py:157-169: R-Net GRU over the user + item pair, co-attention, S-Net u / i, merge) and the whole
 * ControlNet (src/model.py:179-198: C-Net GRU over ui and over the pair, three C-Net heads, control S-Net, SS-Net gate), each
 * issued back to back from C++ into ONE caller-owned arena (everything the backward reads; *_arena_bytes) and one scratch buffer
 * (*_ws_bytes).  Same kernels and the same arithmetic as the per-stage entry points above, which these call in order; what goes
 * away is ~20 host round trips per direction (a UMPR-R training step is 0.9 ms of kernels).
 * ids_pair [2N][L]: user rows then item rows (umpr_concat_ids); lengths / order [2N] with the item half's order offset by N.
 * b16_gemm: GEMM-shaped products on the bf16 pipe (as umpr_set_gemm_bf16); b16_scores: the co-attention score contraction in
 * bf16 (as umpr_coattention_fwd_bf16); need_grad = 0 skips the GRU gate records (inference). */
int umpr_concat_ids(const int64_t* ids_u, const int64_t* ids_i, long n_each, int64_t* dst, void* stream);
size_t umpr_review_net_arena_bytes(int B, int S, int L);
size_t umpr_review_net_ws_bytes(int B, int S, int L, int E);
/* params (15): w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r, M, Ms_u, Ws_u, Ms_i, Ws_i, W_u, W_i; out [B][128] */
int umpr_review_net_fwd(const int64_t* ids_pair, const float* emb, int E, const float* const* params, const int32_t* lengths,
                        const int32_t* order, int B, int S, int L, int b16_gemm, int b16_scores, int need_grad, void* arena,
                        float* out, float* ws, size_t ws_bytes, void* stream);
/* grads: 15 pointers in the order of params, each overwritten */
int umpr_review_net_bwd(const int64_t* ids_pair, const float* emb, int E, const float* const* params, const int32_t* lengths,
                        const int32_t* order, int B, int S, int L, int b16_gemm, const void* arena, const float* d_out,
                        float* const* grads, float* ws, size_t ws_bytes, void* stream);
size_t umpr_control_net_arena_bytes(int B, int S_ui, int L_ui, int S, int L, int KC, int V);
size_t umpr_control_net_ws_bytes(int B, int S_ui, int L_ui, int S, int L, int E, int KC, int KS, int V);
/* params (16): the C-Net GRU's eight, Wc [KC][128][KS], bc, Wl [V][KC], bl, Ms, Ws, ssW [128], ssb [1];
 * outputs c_u, c_i, prefer_pos, prefer_neg [B][V] */
int umpr_control_net_fwd(const int64_t* ids_ui, const int64_t* ids_pair, const float* emb, int E, const float* const* params,
                         const int32_t* len_ui, const int32_t* ord_ui, const int32_t* len_pair, const int32_t* ord_pair, int B,
                         int S_ui, int L_ui, int S, int L, int KC, int KS, int V, float thr, int b16_gemm, int need_grad, void* arena,
                         float* c_u, float* c_i, float* prefer_pos, float* prefer_neg, float* ws, size_t ws_bytes, void* stream);
int umpr_control_net_bwd(const int64_t* ids_ui, const int64_t* ids_pair, const float* emb, int E, const float* const* params,
                         const int32_t* len_ui, const int32_t* ord_ui, const int32_t* len_pair, const int32_t* ord_pair, int B,
                         int S_ui, int L_ui, int S, int L, int KC, int KS, int V, int b16_gemm, const void* arena, const float* d_cu,
                         const float* d_ci, const float* d_pp, const float* d_pn, float* const* grads, float* ws, size_t ws_bytes,
                         void* stream);

/* ---- K10: VGG16-D feature extractor (torchvision.models.vgg16, call site model.py:204-207,217) ---------------
 * images [n][3][224][224]; params: 32 pointers = 13 x (conv weight [Cout][Cin][3][3], bias) then 3 x (fc weight
 * [out][in], bias) in torchvision order.  acts: activation arena (umpr_vgg16_act_bytes), kept for backward.
 * train!=0 applies Dropout(0.5) after fc1/fc2 with a counter-hash mask from `seed`; use_masks!=0 reads the caller's
 * keep-masks (uint8 [2][n][4096] in `masks`) instead (parity tests).  out [n][1000]. */
size_t umpr_vgg16_act_bytes(int n_img);
size_t umpr_vgg16_fwd_ws_bytes(int n_img);
size_t umpr_vgg16_ws_bytes(int n_img); /* backward workspace */
int umpr_vgg16_fwd(const float* images, const float* const* params, int n_img, int train, int use_masks,
                   uint64_t seed, float* acts, uint8_t* masks, float* out, float* ws, size_t ws_bytes, void* stream);
/* grads: 32 pointers matching params (overwritten).  train != 0 <=> dropout was applied in the forward pass
 * (train or use_masks). */
int umpr_vgg16_bwd(const float* images, const float* const* params, int n_img, int train, const float* acts,
                   const uint8_t* masks, const float* d_out, float* const* grads, float* ws, size_t ws_bytes,
                   void* stream);
/* The same in two stages, so that a host can start exchanging the classifier gradients (89 % of all gradient bytes:
 * fc1 alone is 411 MB) while the convolutional backward still runs.  pool5 = acts + umpr_vgg16_pool5_offset(n) bytes
 * is the [n][512][7][7] output of the feature stage; d_pool5 [n][25088]. */
size_t umpr_vgg16_pool5_offset(int n_img);
int umpr_vgg16_features_fwd(const float* images, const float* const* params, int n_img, float* acts, float* ws,
                            size_t ws_bytes, void* stream);
int umpr_vgg16_classifier_fwd(const float* const* params, int n_img, int train, int use_masks, uint64_t seed,
                              float* acts, uint8_t* masks, float* out, float* ws, size_t ws_bytes, void* stream);
size_t umpr_vgg16_classifier_bwd_ws_bytes(int n_img);
int umpr_vgg16_classifier_bwd(const float* const* params, int n_img, int train, const float* acts,
                              const uint8_t* masks, const float* d_out, float* const* grads, float* d_pool5, float* ws,
                              size_t ws_bytes, void* stream);
size_t umpr_vgg16_features_bwd_ws_bytes(int n_img);
int umpr_vgg16_features_bwd(const float* images, const float* const* params, int n_img, const float* acts,
                            const float* d_pool5, float* const* grads, float* ws, size_t ws_bytes, void* stream);
/* Data parallel: `cb(block, user)` is called on the calling thread each time umpr_vgg16_features_bwd /
 * umpr_vgg16_bf16_features_bwd has ENQUEUED all kernels of one VGG block (4, 3, 2, 1, 0), i.e. as soon as that block's
 * weight gradients are ordered on umpr_vgg16_wgrad_stream() (the library's side stream; NULL when weight gradients run on
 * the caller's stream).  A host that owns flat gradient storage starts that block's all-reduce there, ordered behind that
 * stream, while the blocks below are still being computed.  cb = NULL clears it. */
typedef void (*umpr_block_callback)(int block, void* user);
int umpr_vgg16_set_block_callback(umpr_block_callback cb, void* user);
void* umpr_vgg16_wgrad_stream(void);
/* per-layer entry points (also what the composite calls) */
/* wpack / wt: scratch of umpr_conv3x3_pack_bytes(...) - packed weights, or on the 56/28/14 maps the Winograd
 * F(2x2,3x3) buffers (with a smaller scratch those layers fall back to the direct kernel) */
size_t umpr_conv3x3_pack_bytes(int N, int Cin, int Cout, int H, int W);
int umpr_conv3x3_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W,
                     int Cout, int relu, float* wpack, size_t wpack_bytes, void* stream);
/* dx = conv_transpose(dy, w) [* (mask_src > 0)] */
int umpr_conv3x3_bwd_data(const float* dy, const float* w, const float* mask_src /*or NULL*/, float* dx, int N,
                          int Cin, int H, int W, int Cout, float* wt, size_t wt_bytes, void* stream);
size_t umpr_conv3x3_bwd_weight_ws_bytes(int N, int Cin, int Cout, int H, int W);
int umpr_conv3x3_bwd_weight(const float* dy, const float* x, float* dw, float* db, int N, int Cin, int H, int W,
                            int Cout, float* ws, size_t ws_bytes, void* stream);
int umpr_maxpool2_fwd(const float* x, float* y, long planes, int H, int W, void* stream);
int umpr_maxpool2_bwd_relu(const float* x, const float* dy, float* dx, long planes, int H, int W, void* stream);

/* ---- bf16 mixed precision (BASELINE.json configs[4]: "MFMA bf16 conv + attention; MSE within 1e-3 of fp32") ---------
 * Replaces the same torchvision VGG16 call site (model.py:204-207,217) with bf16 activations / gradients, fp32
 * accumulation and fp32 master weights: v_mfma_f32_32x32x16_bf16 implicit GEMM.  Activations live in the library's
 * CB8-PF layout (umpr_amd/csrc/bf16_conv.hip: channel blocks of 8, padded-flat pixels with zero borders); a tensor of
 * N x C x H x W takes umpr_bf16_tensor_bytes(...) bytes and is produced / read back by the two converters.  Per-layer
 * entry points (parity tests): channels must be multiples of 64 (32 for a reduction), maps 224/112/56/28/14 square.
 * ws: umpr_conv3x3_bf16_ws_bytes(...) bytes of scratch (packed bf16 weights, or split-K slabs of the weight gradient). */
size_t umpr_bf16_tensor_bytes(int N, int C, int H, int W);
int umpr_bf16_from_nchw_f32(const float* x, void* y, int N, int C, int H, int W, void* stream);
int umpr_bf16_to_nchw_f32(const void* x, float* y, int N, int C, int H, int W, void* stream);
size_t umpr_conv3x3_bf16_ws_bytes(int N, int Cin, int Cout, int H, int W);
int umpr_conv3x3_bf16_fwd(const void* x, const float* w, const float* bias, void* y, int N, int Cin, int H, int W,
                          int Cout, int relu, void* ws, size_t ws_bytes, void* stream);
/* dx = conv_transpose(dy, w) [* (mask_src > 0)], mask_src in the same layout as dx */
int umpr_conv3x3_bf16_bwd_data(const void* dy, const float* w, const void* mask_src /*or NULL*/, void* dx, int N,
                               int Cin, int H, int W, int Cout, void* ws, size_t ws_bytes, void* stream);
/* dw [Cout][Cin][3][3], db [Cout] in fp32 (overwritten) */
int umpr_conv3x3_bf16_bwd_weight(const void* dy, const void* x, float* dw, float* db, int N, int Cin, int H, int W,
                                 int Cout, void* ws, size_t ws_bytes, void* stream);
int umpr_maxpool2_bf16_fwd(const void* x, void* y, int N, int C, int H, int W, void* stream);
int umpr_maxpool2_bf16_bwd_relu(const void* x, const void* dy, void* dx, int N, int C, int H, int W, void* stream);
/* The convolutional stage of VGG16 in bf16.  images fp32 [n][3][224][224]; params / grads: the same 32-pointer fp32
 * tables as umpr_vgg16_features_*; acts: umpr_vgg16_bf16_act_bytes(n) bytes kept for backward; pool5 / d_pool5: fp32
 * [n][25088] in NCHW flatten order (what the classifier consumes / produces). */
size_t umpr_vgg16_bf16_act_bytes(int n_img);
size_t umpr_vgg16_bf16_fwd_ws_bytes(int n_img);
size_t umpr_vgg16_bf16_bwd_ws_bytes(int n_img);
int umpr_vgg16_bf16_features_fwd(const float* images, const float* const* params, int n_img, void* acts, float* pool5,
                                 void* ws, size_t ws_bytes, void* stream);
int umpr_vgg16_bf16_features_bwd(const float* images, const float* const* params, int n_img, const void* acts,
                                 const float* d_pool5, float* const* grads, void* ws, size_t ws_bytes, void* stream);
/* The classifier on a compact fp32 arena of umpr_vgg16_cls_arena_bytes(n): [pool5 n x 25088][fc1, fc2 ReLU outputs]
 * [their dropout outputs] - for callers whose feature stage does not use the fp32 activation arena (the bf16 path).
 * Workspaces as for umpr_vgg16_classifier_fwd / _bwd. */
size_t umpr_vgg16_cls_arena_bytes(int n_img);
int umpr_vgg16_classifier_fwd_compact(const float* const* params, int n_img, int train, int use_masks, uint64_t seed,
                                      float* cls_arena, uint8_t* masks, float* out, float* ws, size_t ws_bytes,
                                      void* stream);
int umpr_vgg16_classifier_bwd_compact(const float* const* params, int n_img, int train, const float* cls_arena,
                                      const uint8_t* masks, const float* d_out, float* const* grads, float* d_pool5,
                                      float* ws, size_t ws_bytes, void* stream);
/* The same under mixed precision (torch.autocast would run these nn.Linear layers in bf16 too): operands rounded to
 * bf16 in registers, v_mfma_f32_32x32x16_bf16 with fp32 accumulation; everything in memory stays fp32. */
int umpr_vgg16_classifier_fwd_compact_bf16(const float* const* params, int n_img, int train, int use_masks,
                                           uint64_t seed, float* cls_arena, uint8_t* masks, float* out, float* ws,
                                           size_t ws_bytes, void* stream);
int umpr_vgg16_classifier_bwd_compact_bf16(const float* const* params, int n_img, int train, const float* cls_arena,
                                           const uint8_t* masks, const float* d_out, float* const* grads,
                                           float* d_pool5, float* ws, size_t ws_bytes, void* stream);

/* ---- K11-K12: visual head + fusion + losses (model.py:218-228,267-277) ----------------------------------------
 * V = 0 selects the review_net_only branch (model.py:267-269).  loss [3] = (loss, loss_r, loss_v). */
int umpr_head_fwd(const float* rr, const float* c_u, const float* c_i, const float* prefer_pos,
                  const float* prefer_neg, const float* vgg, const float* pos_v_emb, const float* neg_v_emb,
                  const float* lin_w, const float* lin_b, const float* fus_w, const float* fus_b, const float* labels,
                  float loss_v_rate, int B, int V, int P, float* pred, float* loss, float* z, float* img_emb,
                  float* pos_match, float* neg_match, float* posneg_emb, void* stream);
int umpr_head_bwd(const float* rr, const float* c_u, const float* c_i, const float* prefer_pos,
                  const float* prefer_neg, const float* vgg, const float* pos_v_emb, const float* neg_v_emb,
                  const float* lin_w, const float* fus_w, const float* labels, float loss_v_rate, int B, int V, int P,
                  const float* pred, const float* z, const float* img_emb, const float* pos_match,
                  const float* neg_match, const float* posneg_emb, const float* d_loss, const float* d_pred /*or NULL*/,
                  float* d_rr, float* d_cu, float* d_ci, float* d_pp, float* d_pn, float* d_vgg, float* d_pos_v,
                  float* d_neg_v, float* d_lin_w, float* d_lin_b, float* d_fus_w, float* d_fus_b, void* stream);

/* ---- R-Net pre-training head (pretrain/pretrain_rnet.py:147-169): result = sigmoid(Linear(K -> 1)(att)),
 * loss = BCELoss(mean)(result, target) with torch's log clamp at -100.  att rows at att + b*ld (K = 256: [atte_u;atte_i]
 * as umpr_coattention_fwd leaves them).  ws: B floats.  Backward follows ATen's binary_cross_entropy_backward
 * (denominator max(p(1-p), 1e-12)); d_result may be NULL, d_loss is a device scalar. */
int umpr_bce_head_fwd(const float* att, long ld, const float* w, const float* b, const float* target, int B, int K,
                      float* result, float* loss, float* ws, size_t ws_bytes, void* stream);
int umpr_bce_head_bwd(const float* att, long ld, const float* w, const float* result, const float* target,
                      const float* d_result /*or NULL*/, const float* d_loss, int B, int K, float* d_att, long ld_d,
                      float* dw, float* db, float* ws, size_t ws_bytes, void* stream);

/* ---- K13: Adam step with coupled L2, as torch.optim.Adam drives it in main.py:22-26,37 ----------------------
 * One flat parameter segment: p,g,m,v [n]; step >= 1; grad_scale multiplies g first (1/world for data parallel). */
int umpr_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                   double eps, double weight_decay, long step, double grad_scale, void* stream);

/* The same step with its per-step scalars in DEVICE memory - hyper[4] = {grad_scale, lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t),
 * weight_decay} in the kernel's fp32 - so that a captured hipGraph of a training step can be replayed while t advances: the
 * host refreshes the four floats before each replay (umpr_amd/graphs.py). */
int umpr_adam_step_dev(float* p, const float* g, float* m, float* v, long n, double beta1, double beta2, double eps,
                       const float* hyper, void* stream);

/* ---- evaluate_mse (src/evaluate.py:12-13, `mse_loss(pred, labels, reduction='sum')` accumulated over batches) -----
 * acc[0] += sum_i (pred[i] - label[i])^2, acc[1] += n; acc = two device doubles the caller zeroed once and reads back
 * once after the last batch (the reference's `.item()` per batch is a host sync per batch). */
int umpr_sq_err_accumulate(const float* pred, const float* label, long n, double* acc, void* stream);

/* ---- test aid: fills the LDS of every CU with NaN (LDS is not cleared between kernels, so a kernel that reads LDS
 * it never wrote shows up as NaN in the parity tests instead of passing by luck).  sink: one int of device memory. */
int umpr_debug_poison_lds(void* sink, void* stream);

/* ---- kernel timing for bench.py's roofline line: while enabled, the library brackets every launch of a kernel
 * family with HIP events on the launch stream.  family: 0 conv3x3 forward (direct implicit GEMM / Winograd),
 * 1 conv3x3 wgrad, 2 generic GEMM, 3 GRU, 4 the Winograd batched GEMM alone (nested inside families 0 and 5; its
 * work is the MFMA FLOPs executed, 1/2.25 of the direct-convolution FLOPs counted for the same launch), 5 the
 * family-0 kernels run as data gradient (these overlap with wgrad on the library's side stream), 6 the Winograd
 * weight-gradient GEMM alone (nested inside family 1, executed FLOPs), 3 the recurrent GRU kernels (work = bytes),
 * 7 / 8 / 9 the bf16 convolution forward / data-gradient / weight-gradient kernels (FLOPs over the padded pixel grid).  read() synchronises the recorded events and returns totals since reset():
 * milliseconds, algorithmic FLOPs, launches. */
int umpr_profile_enable(int on);
int umpr_profile_reset(void);
int umpr_profile_read(int family, double* total_ms, double* total_work, long* launches);

#ifdef __cplusplus
}
#endif
#endif
