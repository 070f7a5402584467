// Internal (C++) interfaces between the translation units of libumpr_hip.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct UmprGemm {
  const float* A = nullptr; long lda = 0; bool transA = false;
  const float* B = nullptr; long ldb = 0; bool transB = false;
  float* C = nullptr; long ldc = 0;
  int M = 0, N = 0, K = 0;
  const int64_t* gatherA = nullptr;  // !transA: physical row of logical row m (<0: zero row)
  const int64_t* gatherB = nullptr;  // !transB: physical row of logical k   (<0: zero row)
  const float* bias = nullptr; int bias_mode = 0;  // 1: bias[n]  2: bias[m]
  int act = 0; bool accumulate = false; float alpha = 1.0f;
  int split_k = 1;                   // 0 = auto (needs ws)
  float* ws = nullptr; size_t ws_bytes = 0;
  // Sliding-window operand (1-D convolution as a GEMM without im2col): the operand is a matrix X [rows][winD] of
  // sentences of winL rows, and its logical element (row r, column j*winD + c) is X[r + j - winPad][c], zero where
  // (r % winL) + j - winPad leaves the sentence.  winA: A [M][KS*winD] (!transA);  winB: B [K][KS*winD] (!transB).
  int winA_L = 0, winA_D = 1, winA_pad = 0;
  int winB_L = 0, winB_D = 1, winB_pad = 0;
};
void umpr_gemm_set_b16(int on);
int umpr_gemm(const UmprGemm& g, hipStream_t stream);

// fc_small.hip - batch-sized fully connected layers (register-streaming fp32 MFMA, no LDS)
bool umpr_fc_small_ok(int M, int N, int K);
size_t umpr_fc_small_ws_bytes(int M, int N, int K);
int umpr_fc_small_fwd(const float* x, const float* W, const float* bias, float* out, int M, int N, int K, int act,
                      float* ws, size_t ws_bytes, hipStream_t s, int accumulate = 0, int bf16 = 0);
int umpr_fc_small_dx(const float* g, const float* W, float* dx, int M, int N, int K, float* ws, size_t ws_bytes,
                     hipStream_t s, int bf16 = 0);
int umpr_fc_small_dw(const float* g, const float* x, float* dW, int M, int N, int K, hipStream_t s, int bf16 = 0);

// small shared launch helpers (util.hip)
int umpr_fill(float* p, long n, float v, hipStream_t s);
int umpr_multi_copy(const float* const* src, float* const* dst, const long* n, const int* accumulate, int nseg, hipStream_t s);
int umpr_multi_colsum_rows(const float* const* src, const int* rows, const long* cols, const long* row_stride, float* const* dst,
                           const int* accumulate, int nseg, hipStream_t s);
int umpr_copy_or_add(const float* src, float* dst, long n, int accumulate, hipStream_t s);   // dst (+)= src

// visual head + fusion + losses (text_ops.hip)
struct UmprHead {
  const float* rr;   // [B][128] review_net representation
  const float* c_u; const float* c_i; const float* pp; const float* pn;  // [B][V] (null when review_net_only)
  const float* vgg;  // [B*V*P][1000]
  const float* pos_v; const float* neg_v;  // [V][1000]
  const float* lw; const float* lb;        // [1000], [1]
  const float* fw; const float* fb;        // [128 (+2V)], [1]
  const float* labels;                     // [B]
  float rate;
  int B, V, P, F;                          // F = 1000; V = 0 for review_net_only
  // forward outputs / saved
  float* pred; float* loss;                // [B], [3] = (loss, loss_r, loss_v)
  float* z;                                // [B] pre-ReLU
  float* img_emb; float* pos_match; float* neg_match;  // [B][V]
  float* posneg_emb;                       // [2][V]
  // backward inputs / outputs
  const float* d_loss;                     // [1]
  const float* d_pred;                     // [B] or null
  float* d_rr; float* d_cu; float* d_ci; float* d_pp; float* d_pn;
  float* d_vgg; float* d_pos_v; float* d_neg_v; float* d_lw; float* d_lb; float* d_fw; float* d_fb;
};
int umpr_head_launch(const UmprHead& p, int backward, hipStream_t s);

// conv3x3.hip
size_t umpr_conv3x3_pack_floats(int N, int Cin, int Cout, int H, int W);
int umpr_conv3x3_run(const float* x, const float* w, int transposed, const float* bias, const float* mask, float* y,
                     int N, int Cin, int Cout, int H, int W, int relu, float* wpack, size_t wpack_floats,
                     hipStream_t s);
// winograd.hip
size_t umpr_wino_ws_floats(int N, int C, int M, int H, int W, int transposed);
void umpr_wino_set_inference(int on);   // per host thread, see umpr_set_conv_inference
void umpr_wino_set_pool_follows(int on);   // per host thread, see umpr_set_conv_pool_follows
long umpr_wino_last_fix_count();            // see umpr_debug_wino_fix_count
void umpr_wino_set_v_slot(float* p, size_t floats);   // per host thread: where a training forward keeps V for the weight gradient
size_t umpr_wino_v_floats(int N, int Cin, int Cout, int H, int W);   // 0: the layer's V is not shared
bool umpr_conv3x3_fwd_is_wino(int Cin, int Cout, int H, int W);     // the training forward of this layer runs Winograd
int umpr_wino_inference();
int umpr_wino_f4_mode();   // UMPR_WINO_F4: 0 = F(2x2,3x3) only, 1 = F(4x4,3x3) in the backward pass, 2 = forward as well
size_t umpr_wino_wgrad_ws_floats(int N, int Cin, int Cout, int H, int W);
int umpr_wino_wgrad(const float* dy, const float* x, float* dw, float* db, int N, int Cin, int Cout, int H, int W,
                    int accumulate, float* ws, size_t ws_floats, hipStream_t s);
int umpr_wino_conv3x3(const float* x, const float* w, int transposed, const float* bias, const float* mask, float* y,
                      int N, int Cin, int Cout, int H, int W, int relu, float* ws, size_t ws_floats, hipStream_t s);
int umpr_conv3x3_flip_transpose(const float* w, float* wt, int Cout, int Cin, hipStream_t s);
size_t umpr_conv3x3_wgrad_ws_bytes(int N, int Cin, int Cout, int H, int W);
int umpr_conv3x3_wgrad(const float* gz, const float* x, float* dw, float* db, int N, int Cin, int Cout, int H, int W,
                       int accumulate, float* ws, size_t ws_bytes, hipStream_t s);
int umpr_maxpool2_fwd_impl(const float* x, float* y, long planes, int H, int W, hipStream_t s);
int umpr_maxpool2_bwd_relu_impl(const float* x, const float* gy, float* gx, long planes, int H, int W, hipStream_t s);

int umpr_wgrad_reduce(const float* slab, const float* bslab, int splits, int Cout, int Cin, float* dw, float* db,
                      int accumulate, hipStream_t s);

// bf16_conv.hip - bf16 mixed-precision conv stack on CB8-PF tensors (see the file header for the layout)
struct UmprPF {      // padded-flat geometry of N images of H x W
  int N, H, W, RW;   // RW = W + 1: row pitch in pixels (column 0 is the zero pad)
  long rows;         // N * (H + 1) + 1 map rows incl. the zero rows before / between / after the images
  long ptot;         // rows * RW: pixels [0, ptot) are produced by the kernels
  long lead, tail;   // zeroed guard pixels in front of pixel 0 and behind ptot
  long ps;           // plane stride in pixels = lead + ptot + tail (multiple of 64)
};
UmprPF umpr_pf(int N, int H, int W);
size_t umpr_pf_bytes(const UmprPF& g, int C);
size_t umpr_conv_bf16_pack_bytes(int Cin, int Cout);
int umpr_conv_bf16_run(const void* x, const float* w, int transposed, const float* bias, const void* mask, void* y,
                       const UmprPF& g, int Cin, int Cout, int relu, void* wpack, size_t wpack_bytes, hipStream_t s,
                       bool zero_guards = true);
int umpr_conv_bf16_pack_all(const float* const* w, const int* Cin, const int* Cout, const int* W, int n, int transposed,
                            void* wpack, const size_t* offsets, hipStream_t s);
int umpr_pf_zero_guards_multi(void* const* bases, const UmprPF* geos, const int* channels, int n, hipStream_t s);
size_t umpr_wgrad_bf16_ws_bytes(const UmprPF& g, int Cin, int Cout);
int umpr_wgrad_bf16_run(const void* dy, const void* x, float* dw, float* db, const UmprPF& g, int Cin, int Cout,
                        int accumulate, float* ws, size_t ws_bytes, hipStream_t s);
int umpr_conv1_bf16_fwd(const float* x, const float* w, const float* bias, void* y, const UmprPF& g, hipStream_t s,
                        bool zero_guards = true);
size_t umpr_conv1_bf16_wgrad_ws_bytes();
int umpr_conv1_bf16_wgrad(const void* dy, const float* x, float* dw, float* db, const UmprPF& g, int accumulate, float* ws,
                          size_t ws_bytes, hipStream_t s);
int umpr_nchw_to_cb8(const float* x, void* y, const UmprPF& g, int C, hipStream_t s);
int umpr_cb8_to_nchw(const void* x, float* y, const UmprPF& g, int C, hipStream_t s);
int umpr_maxpool2_bf16_fwd_run(const void* x, void* y, const UmprPF& gi, const UmprPF& go, int C, hipStream_t s,
                               bool zero_guards = true);
int umpr_maxpool2_bf16_bwd_run(const void* x, const void* gy, void* gx, const UmprPF& gi, const UmprPF& go, int C,
                               hipStream_t s);

// gru.hip
int umpr_colsum_rows(const float* src, int rows, long cols, long row_stride, float* dst, int accumulate, hipStream_t s);
int umpr_gru_recurrent_fwd(const float* gx, const float* whh_f, const float* bhh_f, const float* whh_r,
                           const float* bhh_r, const int* lengths, const int* order, const int* dst_row, float* out,
                           float* saved, int N, int L, hipStream_t s);
int umpr_gru_tiles(int N);
int umpr_gru_bptt(const float* dout, const float* out, const float* saved, const float* whh_f, const float* whh_r,
                  const int* lengths, const int* order, const int* dst_row, float* dgx, float* dwhh_slab,
                  float* dbias_slab, int N, int L, hipStream_t s);

// coattn.hip
size_t umpr_coattn_fwd_ws_bytes(int B, int SL);
int umpr_coattn_fwd_impl(const float* Gu, const float* Gi, const float* M, int B, int SL, float* T, float* soft_u,
                         float* soft_i, float* atte_u, long ld_u, float* atte_i, long ld_i, float* colmax, int* argcol,
                         float* rowmax, int* argrow, float* ws, size_t ws_bytes, hipStream_t s, int bf16_scores = 0);
size_t umpr_coattn_bwd_ws_bytes(int B, int SL);
int umpr_coattn_bwd_impl(const float* Gu, const float* Gi, const float* M, const float* T, const float* soft_u,
                         const float* soft_i, const float* colmax, const int* argcol, const float* rowmax,
                         const int* argrow, const float* d_atte_u, long ld_du, const float* d_atte_i, long ld_di,
                         const float* d_soft_u, const float* d_soft_i, int B, int SL, float* dGu, float* dGi, float* dM,
                         int accumulate, float* ws, size_t ws_bytes, hipStream_t s);

// text_ops.hip
int umpr_tanh_bwd(const float* y, const float* gy, float* gx, long n, hipStream_t s);
// tanh(repr_u W_u^T + repr_i W_i^T) (model.py:150-158) and its backward: repr [B][256], W [128][256], out [B][128]; out2 (optional)
// receives a second copy of out
int umpr_review_merge_fwd_impl(const float* ru, const float* ri, const float* Wu, const float* Wi, int B, float* out, float* out2,
                               hipStream_t s);
int umpr_review_merge_bwd_impl(const float* ru, const float* ri, const float* Wu, const float* Wi, const float* out,
                               const float* d_out, int B, float* dru, float* dri, float* dWu, float* dWi, hipStream_t s);
int umpr_relu_bwd(const float* y, const float* gy, float* gx, long n, hipStream_t s);
size_t umpr_colsum_ws_bytes(long rows, int cols);
int umpr_colsum(const float* src, long rows, int cols, long ld, float* dst, int accumulate, float* ws, size_t ws_bytes,
                hipStream_t s);
int umpr_dropout_fwd_impl(const float* x, float* y, uint8_t* mask, long n, float p, uint64_t seed, int gen, hipStream_t s);
int umpr_dropout_bwd_impl(const float* gy, const uint8_t* mask, const float* a, float* gx, long n, float p, hipStream_t s);
int umpr_bce_head_fwd_impl(const float* att, long ld, const float* w, const float* b, const float* target, int B, int K,
                           float* result, float* loss, float* ws, size_t ws_bytes, hipStream_t s);
int umpr_bce_head_bwd_impl(const float* att, long ld, const float* w, const float* result, const float* target,
                           const float* d_result, const float* d_loss, int B, int K, float* d_att, long ld_d, float* dw,
                           float* db, float* ws, size_t ws_bytes, hipStream_t s);
int umpr_adam_impl(float* p, const float* g, float* m, float* v, long n, float gscale, float wd, float b1, float b2,
                   float eps, float step_size, float inv_bc2_sqrt, hipStream_t s);
int umpr_sq_err_accumulate_impl(const float* pred, const float* label, long n, double* acc, hipStream_t s);
int umpr_adam_dev_impl(float* p, const float* g, float* m, float* v, long n, float b1, float b2, float eps, const float* hyper,
                       hipStream_t s);
int umpr_snet_fwd_impl(const float* X, const float* Ms, const float* Ws, const float* word_soft, int wl, int B, int S,
                       int L, float* U, float* P, float* wsum, float* self_atte, float* senti, long ld_senti,
                       hipStream_t s);
size_t umpr_snet_bwd_ws_bytes_impl(int B, int S, int L);
int umpr_snet_bwd_impl(const float* X, const float* Ms, const float* Ws, const float* U, const float* P,
                       const float* wsum, const float* self_atte, const float* d_senti, long ld_ds,
                       const float* d_self_atte, int B, int S, int L, int wl, float* dX, float* dMs, float* dWs,
                       float* d_word_soft, float* ws, size_t ws_bytes, hipStream_t s);
size_t umpr_cnet_fwd_ws_bytes(int B, int S, int L, int KS);
int umpr_cnet_head_fwd_impl(const float* X, const float* Wc, const float* bc, const float* Wl, const float* bl,
                            float thr, int B, int S, int L, int KC, int KS, int V, float* Y, float* cmax, int* argl,
                            float* sp, float* view_p, float* final_, float* ws, size_t ws_bytes, hipStream_t s);
size_t umpr_cnet_bwd_ws_bytes(int B, int S, int L, int KC, int KS, int V);
int umpr_cnet_head_bwd_impl(const float* X, const float* Wc, const float* Wl, const float* cmax, const int* argl,
                            const float* sp, const float* view_p, const float* d_final, const float* d_view_p, int B,
                            int S, int L, int KC, int KS, int V, float* dX, int accumulate_dX, int accumulate_w, float* dWc,
                            float* dbc, float* dWl, float* dbl, float* ws, size_t ws_bytes, hipStream_t s);
int umpr_gate_fwd_impl(const float* sa, const float* w, const float* bias, const float* view_p, const float* c_out,
                       int B, int S, int V, float* senti, float* vs, float* pp, float* pn, hipStream_t s);
size_t umpr_gate_bwd_ws_bytes(int B);
int umpr_gate_bwd_impl(const float* sa, const float* w, const float* view_p, const float* c_out, const float* senti,
                       const float* vs, const float* d_pp, const float* d_pn, int B, int S, int V, float* d_sa,
                       float* d_view_p, float* d_c_out, float* dw, float* db, float* ws, size_t ws_bytes, hipStream_t s);

// ---- in-library kernel timing (bench.py roofline): HIP events recorded on the launch stream around the kernels of
// one family while profiling is enabled.  Zero cost when disabled.
enum UmprKernelFamily { UMPR_K_CONV_IGEMM = 0, UMPR_K_CONV_WGRAD = 1, UMPR_K_GEMM = 2, UMPR_K_GRU = 3,
                        UMPR_K_WINO_GEMM = 4 /* nested inside CONV_IGEMM: executed MFMA FLOPs of the Winograd GEMM */,
                        UMPR_K_CONV_DGRAD = 5 /* the conv kernels of family 0 run as data gradient */,
                        UMPR_K_WINO_WGRAD_GEMM = 6 /* nested inside CONV_WGRAD: executed FLOPs of the Winograd wgrad GEMM */,
                        UMPR_K_B16_FWD = 7, UMPR_K_B16_DGRAD = 8, UMPR_K_B16_WGRAD = 9 /* bf16 conv kernels (bf16_conv.hip) */,
                        UMPR_K_COUNT = 10 };
struct UmprProfScope {
  UmprProfScope(int family, double work, hipStream_t s);
  ~UmprProfScope();
  int idx; hipStream_t stream;
};
