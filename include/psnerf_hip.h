/*
 * psnerf_hip.h -- C ABI of libpsnerf_hip.so, the MI355X (gfx950) implementation
 * of the PS-NeRF hot path.
 *
 * The reference (ywq/psnerf) has no native boundary: its hot path is PyTorch
 * eager code under nn.Module.forward().  These entry points are what a
 * ctypes/cffi binding on the reference side would call for each torch
 * expression they replace (citations are relative to /root/reference).  All
 * pointers are DEVICE pointers to contiguous fp32 (or int32 where stated)
 * buffers owned by the caller (PyTorch's allocator in our host code); the
 * library never allocates or frees device memory and keeps no global state.
 * `stream` is a hipStream_t passed as void* (0 = default stream).  Every
 * function returns 0 on success or a negative PSN_E_* code; psn_last_error()
 * returns a thread-local description of the most recent failure.  Kernels are
 * launched asynchronously on `stream`; no function synchronises the device.
 */
#ifndef PSNERF_HIP_H
#define PSNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSN_OK 0
#define PSN_E_ARG (-1)     /* bad shape / null pointer / misaligned buffer */
#define PSN_E_LAUNCH (-2)  /* hipLaunchKernel / hipGetLastError failure */
#define PSN_E_UNSUPPORTED (-3)

const char* psn_last_error(void);
int psn_version(void);

/* ------------------------------------------------------------------------
 * Alpha composite -- stage1/model/rendering.py:196-197,214-216 (and :405-406)
 *   w_i   = alpha_i * prod_{j<i} (1 - alpha_j + 1e-6)
 *   rgb   = sum_i w_i c_i (+ 1 - acc if white_bg),   acc = sum_i w_i
 * alpha [n_rays, n_samples], rgb [n_rays, n_samples, 3] (may be NULL: acc only),
 * weights [n_rays, n_samples] (saved for backward; may be NULL),
 * rgb_out [n_rays, 3] (NULL iff rgb NULL), acc_out [n_rays].
 * n_samples <= 1024.
 * ---------------------------------------------------------------------- */
int psn_composite_fwd(const float* alpha, const float* rgb, int64_t n_rays, int n_samples, int white_bg,
                      float* weights, float* rgb_out, float* acc_out, void* stream);
/* d_rgb_out [n_rays,3], d_acc_out [n_rays] (may be NULL) -> d_alpha [n_rays,n_samples], d_rgb [n_rays,n_samples,3] */
int psn_composite_bwd(const float* alpha, const float* rgb, const float* d_rgb_out, const float* d_acc_out,
                      int64_t n_rays, int n_samples, int white_bg, float* d_alpha, float* d_rgb, void* stream);

/* ------------------------------------------------------------------------
 * Positional encodings.
 *   layout 0: stage1 PositionalEncoding (stage1/model/network.py:141-150)
 *   layout 0 is also stage2 Embedder (stage2/model/embedder.py:6-54): both are
 *   [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] with 3-vectors per block.
 * x [n, 3] -> out [n, out_stride] (first 3+6*n_freqs columns written, the
 * rest zero-filled up to out_stride).  `scale` multiplies x first (1/rescale).
 * ---------------------------------------------------------------------- */
int psn_pe_encode(const float* x, int64_t n, int n_freqs, float scale, float* out, int out_stride, void* stream);
/* Input table of the stage-1 appearance network, stage1/model/network.py:128-138 (infer_app: cat[p, gamma(view), normal,
 * features]; gamma from :141-150 on the normalised view direction, rendering.py:185-190): out [n, 64] row r =
 * [p (3) | v / |v| (3) | sin, cos bands of v / |v| (6 n_freqs) | normal (3) | zeros].  The 256 features are not part of the
 * table: they are the fused chain's initial activations (psn_mlp_infer act_init). */
int psn_app_input(const float* p, const float* v, const float* normal, int64_t n, int n_freqs, float* out, void* stream);

/* forward-mode: t [n,3] tangent of x -> d(encoding) [n, out_stride] = J_PE(x) t  (the transpose of
 * psn_pe_encode_bwd; needed for the backward of the gradient sweep) */
int psn_pe_encode_jvp(const float* x, const float* t, int64_t n, int n_freqs, float scale, float* out, int out_stride,
                      void* stream);
/* d_out [n, out_stride] -> d_x [n,3] (chain rule through sin/cos; x is re-read).  d_out2 (or NULL): a second gradient of
 * the encoding, row stride out2_stride, added column by column first (both may be column ranges of wider tensors). */
int psn_pe_encode_bwd(const float* x, const float* d_out, int64_t n, int n_freqs, float scale, int out_stride,
                      const float* d_out2, int out2_stride, float* d_x, void* stream);

/* ------------------------------------------------------------------------
 * fp32-MFMA GEMM with fused epilogues: the torch.nn.Linear / addmm / mm calls
 * of stage1/model/network.py:85-106 and stage2/model/renderer.py:17-49 and
 * their autograd backward.
 *   C[M,N] = epi( opA(A)[M,K] * opB(B)[K,N] )
 * trans_a = 0: A is [M,K] row-major (lda);  1: A is stored [K,M] row-major.
 * trans_b = 0: B is [K,N] row-major (ldb);  1: B is stored [N,K] row-major
 *              (a torch Linear weight -> y = x W^T is trans_a=0, trans_b=1).
 * split_k > 1 (only epi NONE/ACCUM): K is cut into split_k slices, partial
 * tiles go to `workspace` (>= split_k*M*N floats) and are reduced
 * deterministically by a second kernel.
 * ----------------------------------------------------------------------  * colsum_a (trans_a only, or NULL): receives sum_k A[k, m] for m < M as a by-product of staging the A tiles -- with
 * A = dZ this is the bias gradient that accompanies the weight gradient dW = dZ^T X (nn.Linear backward), so no
 * separate pass over dZ is needed.  With split_k > 1 the workspace must hold split_k * M * (N + 1) floats.
 */
enum {
    PSN_EPI_NONE = 0,          /* C = acc                                            */
    PSN_EPI_BIAS = 1,          /* C = acc + bias[n]                                  */
    PSN_EPI_BIAS_RELU = 2,     /* C = relu(acc + bias[n])                            */
    PSN_EPI_BIAS_SOFTPLUS = 3, /* C = softplus_100(z), aux_out = sigmoid(100 z), z = acc+bias */
    PSN_EPI_MUL_AUX = 4,       /* C = acc * aux_in[m,n]                              */
    PSN_EPI_MUL_POS = 5,       /* C = acc * (aux_in[m,n] > 0)   (ReLU backward)      */
    PSN_EPI_BIAS_SIGMOID = 6,  /* C = sigmoid(acc + bias[n])                         */
    PSN_EPI_ACCUM = 7,         /* C += acc                                           */
    PSN_EPI_MUL2 = 8,          /* C = acc * aux_in,  aux_out = acc * aux_in2         */
    PSN_EPI_SOFTPLUS_BWD = 9,  /* C = s * (acc + 100 * aux_in2 * (1 - s)), s = aux_in:
                                  d/dz of softplus_100 plus the sigmoid'(100 z) term of the
                                  gradient-sweep branch (double backward of network.py:108-120) */
    PSN_EPI_MUL_AUX_RAW = 10   /* C = acc * aux_in,  aux_out = acc                   */
};
int psn_gemm(int trans_a, int trans_b, int64_t M, int N, int K, const float* A, int64_t lda, const float* B,
             int64_t ldb, float* C, int64_t ldc, const float* bias, int epilogue, const float* aux_in,
             int64_t ld_aux_in, const float* aux_in2, int64_t ld_aux_in2, float* aux_out, int64_t ld_aux_out,
             int split_k, float* workspace, float* colsum_a, void* stream);

/* Grouped weight gradients of one backward pass (nn.Linear backward of every layer of a network at once):
 *   C_i [M_i, N_i] (+)= A_i^T B_i (+ A2_i^T B2_i),    colsum_a_i [M_i] = column sums of A_i      for i < n_items
 * with A_i = dZ_i [K, M_i], B_i = the layer input [K, N_i] (row-major, row strides lda / ldb), all sharing K = the
 * number of rows of the pass.  One launch over all items + one reduction launch; split_k slices of K per product.
 * The optional second product covers a layer that is used twice (stage1/model/network.py:108-120: value pass and
 * gradient sweep share the weights).  n_items <= 12.  workspace: sum over items of
 * n_products * split_k * M * N + split_k * M floats (+ 8 floats of padding per item). */
typedef struct {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    const float* A2; int64_t lda2;   /* NULL: single product */
    const float* B2; int64_t ldb2;
    float* C; int64_t ldc;
    int M, N;
    int accumulate;                  /* 0: C = ..., 1: C += ... */
    float* colsum_a;                 /* NULL or [M] */
    int64_t b_div, b_mod;            /* b_div > 0: B is a TABLE and row k of the product reads its row (k / b_div) % b_mod
                                        (the virtual input rows of psn_mlp_infer: d W_in of a layer whose input block
                                        repeats per light or per point, summed over all K rows without expanding it) */
    const float* B_tab2; int64_t ldb_tab2;   /* optional second table (needs b_div > 0): the columns b_split .. N-1 of the */
    int64_t b2_div, b2_mod;                  /* virtual B come from B_tab2[(k / b2_div) % b2_mod, n - b_split]           */
    int b_split;                             /* multiple of 4; N = b_split + columns taken from B_tab2                    */
    int64_t k_rows;                          /* 0 = K; else this product sums over the first k_rows (<= K) rows only       */
} PsnGemmTnItem;
int psn_gemm_tn_grouped(int n_items, const PsnGemmTnItem* items, int64_t K, int split_k, float* workspace,
                        int64_t workspace_floats, void* stream);
/* Partial products per multiply of the psn_gemm_tn_grouped_x3 launches that follow: 6 (default; three bf16 pieces per operand,
 * fp32-class), 3 (two pieces: hi hi + hi mid + mid hi, ~16 significant bits) or 1 (plain bf16 operands, fp32 accumulation -- the
 * "bf16 MFMA path" of BASELINE configs[4] in its literal sense).  Process-wide; returns the previous value. */
int psn_gemm_tn_x3_set_products(int n_products);
/* psn_gemm_tn_grouped with the 256 x 256-tile products (128 < M, N <= 256: the hidden-layer weight gradients of
 * stage1/model/network.py:85-106 and of the stage-2 visibility net) evaluated on the bf16 matrix pipe with split operands --
 * every fp32 element as three bf16 pieces, six partial products per multiply, fp32 accumulation ("bf16x6"): fp32-class results,
 * HBM-bound instead of MFMA-bound.  EXPERIMENT for BASELINE configs[4]'s bf16 path; every other product of the group takes the
 * exact fp32 kernels of psn_gemm_tn_grouped.  Same arguments, same workspace size. */
int psn_gemm_tn_grouped_x3(int n_items, const PsnGemmTnItem* items, int64_t K, int split_k, float* workspace,
                        int64_t workspace_floats, void* stream);

/* column sums: out[j, n] (+)= sum_m w[m, j] X[m, n] for n_w <= 4 weight columns (row_weight [M, n_w], row stride ldw), or
 * plain column sums out[n] (+)= sum_m X[m, n] with row_weight = NULL, n_w = 0 -- bias gradients, and with weights the
 * n_w x N weight gradient g^T H of a layer with few outputs.  workspace >= 2048*N floats */
int psn_colsum(const float* X, const float* row_weight, int n_w, int64_t ldw, int64_t M, int N, int64_t ldx, float* out,
               int accumulate, float* workspace, void* stream);

/* ------------------------------------------------------------------------
 * Points along rays.  Replaces the depth-profile arithmetic of stage1/model/rendering.py:110-176 (interval
 * sampling around the surface, outer samples after iteration 5000, free-space sampling of rays that miss,
 * stratified jitter) and :431-436 (the ray-march sweep), bit-identically for the same u tables and noise.
 *   origin, dir [N,3]; dist [N] (surface depth of hit rays, NULL otherwise); far [N] (sphere exit depth)
 *   idx [n] int64: the rays of this group = rows of out [N, c0+c1, 3] that are written (NULL: rays 0..n-1)
 *   hit = 0: d = near (1 - u0) + far u0.   hit = 1: [dnp, dfp] = [max(dist - delta, near), min(dist + delta, far)]
 *   sampled with u0 (c1 == 0), or [near, dnp] with u0 followed by [dnp, dfp] with u1.
 *   u*, omu* = linspace(0, 1, c) and 1 - linspace; noise [n, c0+c1] in [0,1) or NULL (no jitter).
 * ---------------------------------------------------------------------- */
int psn_sample_points(const float* origin, const float* dir, const float* dist, const float* far, const int64_t* idx,
                      int64_t n, int hit, float near, float delta, const float* u0, const float* omu0, int c0,
                      const float* u1, const float* omu1, int c1, const float* noise, float* out, void* stream);

/* Both ray groups of a stage-1 step in one launch: ray r uses the hit profile (as hit = 1 above, with dist / delta /
 * u0 / u1) if flags[r] != 0 and the free-space profile d = near * omum + far * um over all c0 + c1 samples otherwise
 * (flags [n] bytes: the renderer's hit mask).  No index lists, hence no nonzero() / host synchronisation. */
int psn_sample_points_flagged(const float* origin, const float* dir, const float* dist, const float* far,
                              const unsigned char* flags, int64_t n, float near, float delta, const float* u0, const float* omu0,
                              int c0, const float* u1, const float* omu1, int c1, const float* um, const float* omum,
                              const float* noise, float* out, void* stream);

/* ------------------------------------------------------------------------
 * Fully fused MLP inference (activations never leave registers).  Replaces the
 * no-grad network evaluations: stage2 visibility_net over L*Ns rows
 * (stage2/model/renderer.py:191-200, vis.detach()), stage1 occupancy queries
 * in ray_marching / secant / light_visibility (stage1/model/rendering.py:
 * 456-462, 537-540, 394-399).
 *
 * Hidden width is 256, 128 or 64 (8, 4 or 2 tiles of 32; one width per network).  A network is described by
 * PsnMlpDesc; weights are pre-packed into MFMA fragment order by
 * psn_mlp_pack_layer (one call per layer, into one contiguous buffer).
 * ---------------------------------------------------------------------- */
#define PSN_MLP_MAX_LAYERS 12
/* Per-layer activation "programs" of the fused kernel.  z = accumulator; a1 / a2 = optional row-major [n_rows,256]
 * operands (mask_ptrs / aux2_ptrs); the new activation (and for some codes a second value) can be dumped row-major
 * (save_ptrs / save2_ptrs).  Forward, backward and gradient-sweep chains of stage1/model/network.py:85-120 and
 * stage2/model/renderer.py:17-49 are all sequences of these. */
enum {
    PSN_ACT_NONE = 0,
    PSN_ACT_RELU = 1,
    PSN_ACT_SOFTPLUS100 = 2,   /* act = softplus_100(z);           second = sigmoid(100 z)                  */
    PSN_ACT_RELU_MASK = 3,     /* act = z * (a1 > 0)               (ReLU backward with the saved activation) */
    PSN_ACT_MUL_AUX = 4,       /* act = z * a1;                    second = z                               */
    PSN_ACT_MUL2 = 5,          /* act = z * a1;                    second = z * a2                          */
    PSN_ACT_SOFTPLUS_BWD = 6,  /* act = a1 z + 100 (1 - a1) a2     (softplus double-backward combine)        */
    PSN_ACT_HEAD = 7,          /* side output: z is dumped, the activations are left untouched              */
    PSN_ACT_RELU_BITS = 8,     /* act = z * bit: RELU_MASK with the mask as sign bits -- a1 = [n_rows, 4] uint64 words written
                                  by psn_mlp_infer_bits (bit 4 mt + r of word (row, g) = feature 16 mt + 4 g + r was > 0)    */
    /* single-dump forms of the softplus chains (stage1/model/network.py:85-120): a1 = the dumped softplus OUTPUT a of the forward
       layer; s = sigmoid(100 z) = 1 - exp(-100 a) is re-formed in the kernel (a < 0, impossible for a softplus, gives s = 0) */
    PSN_ACT_MUL_AUX_A = 9,     /* act = z * s(a1);                 second = z                               */
    PSN_ACT_MUL2_A = 10,       /* act = z * s(a1);                 second = z * a2                          */
    PSN_ACT_SOFTPLUS_BWD_A = 11 /* act = s(a1) z + 100 (1 - s(a1)) a2                                        */
};
enum { PSN_OUT_NONE = 0, PSN_OUT_SIGMOID = 1, PSN_OUT_OCC = 2 /* sigmoid(-10 x), network.py:125 */ };

/* Weight-stage formats.  PSN_W_F32: fp32 fragments of v_mfma_f32_16x16x4_f32 (exact; every parity-gated path).  PSN_W_BF16X2
 * (opt-in experiment, BASELINE configs[4] "bf16 MFMA path"): the same 32 KB stage geometry [k-tile][2][16-row tile][64 lanes][16 B],
 * the two halves being the bf16 heads and the bf16 remainders (x = hi + mid + O(2^-16 x)) of the lane's eight weights of the
 * k-tile; the chain engine multiplies hi hi + hi mid + mid hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- the matrix
 * work of stage1/model/network.py:85-120 (value pass, gradient sweep and their adjoints) and :98-106 on the bf16 pipe,
 * activation programs, dumps and epilogues unchanged (fp32). */
enum { PSN_W_F32 = 0, PSN_W_BF16X2 = 1 };

typedef struct {
    int n_kt_in;   /* 32-wide K tiles taken from the input-feature registers (0..4) */
    int n_kt_act;  /* 32-wide K tiles taken from the previous layer's activations: 0, the hidden n_mt, or n_mt - 1 when the
                      last 32 activation columns are padding (7 of 8 behind a 217-output layer) */
    int n_mt;      /* 32-wide output tiles: 8, 4 or 2 (hidden: 256- / 128- / 64-wide network) or 1 (final) */
    int act;       /* PSN_ACT_* applied to this layer's output */
    int64_t w_off; /* float offset of this layer's packed weights */
    int64_t b_off; /* float offset of this layer's bias (n_mt*32 floats, zero padded) */
    int64_t init_off; /* >= 0: the accumulator additionally starts from init_a[ia, init_off + f] + init_b[ib, init_off + f]
                         (per-row precomputed partial products of this layer's input-feature block); -1: unused */
} PsnMlpLayer;

typedef struct {
    int n_layers;
    int n_out;    /* real outputs of the final layer (<= 32; <= 64 in a 256-wide chain launch: final n_mt = 2, two weight stages) */
    int out_act;  /* PSN_OUT_* */
    int in_kt_a;  /* K tiles (of 32 floats) per row of feature table A (1..4) */
    int in_kt_b;  /* K tiles per row of table B (0 = unused); in_kt_a + in_kt_b <= 4 */
    int init_stride; /* floats per row of the init tables (multiple of the hidden width), 0 if no layer uses init_off */
    int w_format; /* PSN_W_F32, or PSN_W_BF16X2 (experiment; 256-wide networks through psn_mlp_infer* / psn_mlp_infer_pe* /
                     psn_march_sweep, not psn_root_find): every block with n_mt >= 2 is packed as two bf16 planes and multiplied as
                     three bf16 partial products, see PsnPackItem.format */
    PsnMlpLayer layers[PSN_MLP_MAX_LAYERS];
} PsnMlpDesc;

/* Pack W[rows, cols] (row-major, ldw floats per row; with transpose != 0 the memory holds W^T, i.e. element (r, c) sits
 * at W[c * ldw + r]), zero-extended to [n_mt*32, k_tiles*32], into `dst` (n_mt*32 * k_tiles*32 floats) in stage order.
 * A layer whose K dimension consists of several blocks (activations | input features) is packed block by block into
 * consecutive k-tile ranges of its slot. */
int psn_mlp_pack_layer(const float* W, int64_t ldw, int rows, int cols, int transpose, int n_mt, int k_tiles, float* dst,
                       void* stream);

/* The same for up to PSN_PACK_MAX_ITEMS blocks in one launch (all layers of a network after an optimiser step).
 * HOST array of items holding device pointers. */
#define PSN_PACK_MAX_ITEMS 24
typedef struct {
    const float* W;
    float* dst;
    int64_t ldw;
    int rows, cols, transpose, n_mt, k_tiles;
    int format; /* PSN_W_F32 or PSN_W_BF16X2 (same size and geometry, see above) */
} PsnPackItem;
int psn_mlp_pack_layers(int n_items, const PsnPackItem* items, void* stream);

/* Row q of the virtual input matrix is [ A[(q / a_div) % a_mod, :] | B[(q / b_div) % b_mod, :] ];
 * tables are row-major with strides in_kt_a*32 / in_kt_b*32 floats, 16-byte aligned.
 * Because a layer that reads the input block is linear in it, W_in [A_row | B_row] = W_a A_row + W_b B_row
 * can be precomputed once per table row instead of once per (A,B) pair: init_a / init_b
 * ([rows, init_stride], indexed like tab_a / tab_b; init_b may be NULL) hold those partial products and the
 * layers with init_off >= 0 start their accumulators from them (tab_a may then be NULL if no layer has n_kt_in > 0).
 * save_ptrs (HOST array of n_layers-1 device pointers, entries or the array itself may be NULL): for rows >=
 * save_row0 the post-activation output of hidden layer l is also written row-major to save_ptrs[l]
 * [n_rows - save_row0, 256] -- the training rows of stage2/model/renderer.py:251-262 ride along with the
 * gradient-free rows and leave exactly what their backward pass needs.
 * mask_ptrs / aux2_ptrs (HOST arrays of n_layers device pointers or NULL): row-major [n_rows, 256] operands a1 / a2
 * of the layers' activation programs; save2_ptrs: second dumps (same indexing as save_ptrs).  With n_out == 0 there is no final layer and `out` may be NULL: together with
 * transposed weight packs, an init table holding d h of the last hidden layer, masks = the dumped activations and
 * save_ptrs = the d z outputs this runs the ReLU BACKWARD chain d h_{l-1} = W_l^T (d h_l * relu'(h_l)).
 * act_init (or NULL): row-major [act_init_rows, 256] initial activations, so that layer 0 may already read them; rows >=
 * act_init_rows (1 .. n_rows) start from zeros (a gradient that exists for a row prefix only: the colour network's d features
 * when the surface-normal points ride behind the render samples, stage1/model/rendering.py:196-212).
 * rk_coef [n_rows, rk_k], rk_basis [rk_k, init_stride] (or NULL, NULL, 0; rk_k <= 4): rank-k init -- the layers with
 * init_off >= 0 additionally start from sum_c rk_coef[row, c] * rk_basis[c, init_off + f].  This is the init table of a
 * backward chain whose network has 1..4 outputs, d h = g_out W_last (stage1/model/network.py:104-106 colour head,
 * :93-95 occupancy logit; stage2/model/renderer.py:47-49), formed in registers instead of a [n_rows, 256] tensor.
 * dump_tiles (HOST array of 2 * PSN_MLP_MAX_LAYERS words, or NULL = everything): bit mt of word l / word
 * PSN_MLP_MAX_LAYERS + l set = the 16-column tile mt of layer l's first / second dump is written; for dumps of which only
 * a column range is read afterwards (the raw sweep values that feed d logit / d pe, network.py:108-120).
 * out [n_rows, n_out]. */
int psn_mlp_infer(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* tab_a,
                  int64_t a_div, int64_t a_mod, const float* tab_b, int64_t b_div, int64_t b_mod,
                  const float* init_a, const float* init_b, float* const* save_ptrs, int64_t save_row0,
                  const float* const* mask_ptrs, const float* const* aux2_ptrs, float* const* save2_ptrs,
                  const float* act_init, int64_t act_init_rows, const float* rk_coef, const float* rk_basis, int rk_k,
                  const uint32_t* dump_tiles, int64_t n_rows, float* out, void* stream);
/* psn_mlp_infer (plain forward: no chain operands) over a PADDED row set -- the stage-2 visibility rows of
 * stage2/model/renderer.py:191-200 when the surface-pixel list has a fixed capacity (psn_surface_index: one captured HIP graph
 * serves batches with different surface counts).  The rows [0, save_row0) come in groups of live_period rows (one group per
 * shading light; live_period a multiple of 64, save_row0 a multiple of live_period), of which the first live_count[0] -- a
 * float ON THE DEVICE, psn_surface_index's count -- are real.  64-row blocks that hold padding only are not evaluated: their
 * outputs are zeros.  Every other row is evaluated exactly as by psn_mlp_infer (bit-identical), the rows from save_row0 on
 * (the supervision rows, whose dumps a backward pass reads) all of them.  The grid enumerates the evaluated blocks first, so
 * that the skipped ones do not unbalance the XCDs. */
int psn_mlp_infer_padded(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* tab_a,
                         int64_t a_div, int64_t a_mod, const float* tab_b, int64_t b_div, int64_t b_mod,
                         const float* init_a, const float* init_b, float* const* save_ptrs, int64_t save_row0,
                         int64_t n_rows, float* out, const float* live_count, int64_t live_period, void* stream);
/* Workgroup order of the psn_mlp_infer* launches whose rows are (group, point) pairs, row = group * P + point with a_div = 1,
 * b_div = a_mod = P, n_rows = P * b_mod -- the light-major row set of stage2/model/renderer.py:163,183-193.  1 (default): the
 * 64-row blocks are visited point-tile-major with the group (light) running fastest and every XCD takes a contiguous eighth of
 * that order, so that the workgroups resident at one time share a few point tiles and the per-point init table is read from
 * the fabric once instead of once per light; 0: row order (the light-major sweep).  A row's result does not depend on the
 * order (bit-identical outputs and dumps).  Process-wide; returns the previous value. */
int psn_mlp_block_order(int point_major);
/* psn_mlp_infer, plain forward with activation dumps (stage2/model/renderer.py:251-262: the supervision rows of the visibility
 * network, :127-143 / :163-189 the normal / BRDF networks), that ALSO leaves the sign bits of every dumped activation behind:
 * save_bits_ptrs[l] [n_rows - save_row0, 4] uint64 (NULL per layer: none).  The ReLU-backward chain (PSN_ACT_RELU_BITS, the words
 * passed as that layer's aux1 in mask_ptrs) reads 32 bytes per row and layer instead of the 1 KB activation row -- the same
 * d z = d h * (h > 0), bit for bit.  live_count / live_period as for psn_mlp_infer_padded (NULL, 0: every row is real). */
int psn_mlp_infer_bits(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* tab_a,
                       int64_t a_div, int64_t a_mod, const float* tab_b, int64_t b_div, int64_t b_mod,
                       const float* init_a, const float* init_b, float* const* save_ptrs,
                       unsigned long long* const* save_bits_ptrs, int64_t save_row0, int64_t n_rows, float* out,
                       const float* live_count, int64_t live_period, void* stream);

/* Dense per-pixel outputs of the stage-2 model, stage2/model/renderer.py:145-152,204-264: dense [B, N, C] = fill
 * everywhere except dense[b, idx[r], c] = rows[(b Ns + r) row_stride + c col_stride] (light-major surface rows;
 * col_stride 0 broadcasts one column, as the reference's .expand(-1, 3) does).  Up to PSN_SCATTER_MAX_ITEMS outputs in one
 * launch.  inv [N] int32: surface row of a pixel or -1.  psn_gather_rows is the adjoint: rows [B Ns, C] (contiguous,
 * written) = dense[b, idx[r], c] with idx [Ns] int64 (the `rows` pointer of an item is then the OUTPUT). */
#define PSN_SCATTER_MAX_ITEMS 16
typedef struct {
    const float* rows;
    float* dense;
    int64_t row_stride, col_stride;
    int B, C;
    float fill;
} PsnScatterItem;
int psn_scatter_rows(int n_items, const PsnScatterItem* items, const int* inv, int64_t N, int64_t Ns, void* stream);
int psn_gather_rows(int n_items, const PsnScatterItem* items, const int64_t* idx, int64_t N, int64_t Ns, void* stream);
/* psn_gather_rows for an index list that may be PADDED (psn_surface_index): rows r with inv[idx[r]] != r receive zeros. */
int psn_gather_rows_valid(int n_items, const PsnScatterItem* items, const int64_t* idx, const int* inv, int64_t N, int64_t Ns, void* stream);

/* One secant (regula falsi) iteration of the surface refinement, stage1/model/rendering.py:525-555, for n hit rays:
 * with occ [n] (occupancy at the current d_pred; NULL for the initial step) the bracket (d_low, d_high, f_low, f_high,
 * all [n], updated in place) moves -- f_mid = occ - tau replaces the end with its sign -- then
 * d_pred = -f_low (d_high - d_low) / (f_high - f_low) + d_low and, if p_mid != NULL, p_mid [n,3] = origin + d_pred dir. */
int psn_secant_step(const float* occ, float tau, float* d_pred, float* d_low, float* d_high, float* f_low, float* f_high,
                    const float* origin, const float* dir, float* p_mid, int64_t n, void* stream);

/* First free -> occupied crossing of the ray-march sweep, stage1/model/rendering.py:457-504: occ [n_rays, n_steps]
 * (occupancy of the sweep points), far [n_rays] (sphere exit depth), u / omu [n_steps] = linspace(0, 1) and 1 - it (the
 * sweep depths are near * omu + far * u).  bracket [4, n_rays] = d_low, d_high, f_low, f_high of the crossing (values are
 * occupancy - tau); flags [n_rays] int32: bit 0 = a free -> occupied crossing exists on a ray that starts in free space
 * (the reference's `mask`), bit 1 = the ray starts in free space (`mask_0_not_occupied`). */
int psn_first_crossing(const float* occ, const float* far, const float* u, const float* omu, float near, float tau,
                       int64_t n_rays, int n_steps, float* bracket, int* flags, void* stream);

/* Shadow-ray sample points that lie inside the object box, stage1/model/rendering.py:378-408: for the dense rows
 * q = (l n_surf + s) n_steps + m the point p = surf[s] + ldir[l] (lnear omu[m] + lfar u[m]) is generated and, if
 * |p|_inf <= box, appended (atomic counter, caller zeroes it) as pts[k] = p, rows[k] = q.  pts [>= in-box count, 3],
 * rows [>= in-box count] int64, counter [1] uint64.  The caller evaluates the occupancy on pts only; every other row has
 * occupancy 0 by the reference's own rule (alpha[~amask] = 0). */
int psn_shadow_points(const float* surf, const float* ldir, int64_t n_surf, int n_lights, int n_steps, float lnear, float lfar,
                      const float* u, const float* omu, float box, float* pts, int64_t* rows, unsigned long long* counter,
                      void* stream);

/* Positional encoding + network in one launch (stage1/model/network.py:141-150 feeding :85-101; the occupancy queries of
 * the march sweep, rendering.py:410-470, and of the shadow rays, :378-408): points [n_rows, 3]; the encoding
 * gamma(pe_scale * p) with pe_octaves bands is formed in the kernel prologue, in registers, with the expressions of
 * psn_pe_encode -- out equals psn_pe_encode (stride 64) followed by psn_mlp_infer bit for bit, and the [n_rows, 64] table
 * never exists.  desc / packed_w / packed_b as for psn_mlp_infer, restricted to: 256-wide hidden layers without init tables,
 * in_kt_a = 2 (one 64-column block), biases packed back to back, n_out <= 32.  out [n_rows, n_out]. */
int psn_mlp_infer_pe(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* points, int64_t n_rows,
                     int pe_octaves, float pe_scale, float* out, void* stream);
/* psn_mlp_infer_pe over a compacted point list whose length lives on the device (the counter psn_shadow_points writes): the
 * grid covers `capacity` rows, workgroups behind *n_rows_dev leave at once; with out_rows, row r's outputs go to
 * out[out_rows[r] * n_out ...] (scatter).  The shadow-ray visibility of stage1/model/rendering.py:378-408 without a host
 * synchronisation between the +-1.1 box test (:400-401) and the occupancy network. */
int psn_mlp_infer_pe_indirect(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* points,
                              int64_t capacity, const long long* n_rows_dev, const int64_t* out_rows, int pe_octaves,
                              float pe_scale, float* out, void* stream);

/* Ray-march sweep, stage1/model/rendering.py:447-462 (+ the early termination its consumer :472-504 allows): the occupancy
 * sigmoid(-10 logit) of n_steps proposal points per ray, p = origin + dir * (near (1 - u_m) + far u_m), generated and
 * encoded inside the kernel -- no [n_rays, n_steps, 3] point tensor, same bits as psn_sample_points + psn_mlp_infer_pe.
 * One workgroup = 64 consecutive steps of one ray (n_steps % 64 == 0), workgroups in block-major order.  skip: int32
 * [n_rays], ZEROED by the caller, or NULL: a block that contains a sign change of (occ - tau) between neighbouring steps
 * (or a ray whose first value is not free) records its block index in the ray's flag (INT_MAX - block, the lowest block
 * wins; 0 = none) and only blocks BEHIND the recorded one leave unevaluated, whatever order the workgroups ran in -- their
 * entries of occ stay untouched; every value up to and including the pair of the FIRST sign change is always written,
 * which is all psn_first_crossing reads.  n_blocks: NULL, or a device counter that receives the number of 64-step blocks
 * evaluated (measurement).  desc / packed_w / packed_b as for psn_root_find.  occ [n_rays, n_steps]. */
int psn_march_sweep(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* origin, const float* dir,
                    const float* far, const float* u, const float* omu, float near, int64_t n_rays, int n_steps, float tau,
                    int pe_octaves, float pe_scale, int* skip, unsigned long long* n_blocks, float* occ, void* stream);

/* Fused secant refinement, stage1/model/rendering.py:525-555: n_iter regula-falsi iterations for every ray inside ONE
 * launch -- query point origin + d_pred * dir, its positional encoding (pe_octaves bands, input scaled by pe_scale), the
 * occupancy network (desc / packed_w / packed_b as for psn_mlp_infer: 256-wide, one output, PSN_OUT_OCC, input block =
 * one 64-column encoding) and the bracket update -- instead of one encoding + network + update launch per iteration.
 * bracket [4, n_rays] as written by psn_first_crossing (not modified); d_out [n_rays] = the final d_pred. */
int psn_root_find(const PsnMlpDesc* desc, const float* packed_w, const float* packed_b, const float* origin, const float* dir,
                  const float* bracket, int64_t n_rays, float tau, int n_iter, int pe_octaves, float pe_scale, float* d_out,
                  void* stream);

/* ------------------------------------------------------------------------
 * Stage-2 losses, stage2/model/loss.py:27-58,76-92 (MainLoss) and :123-141 (NormalLoss), over the dense outputs of the
 * model with mask = mask_a & mask_b (network_object_mask & object_mask, bool [N]):
 *   term 0  rgb        sum |rgb - rgb_gt| (l2 = 0) or squared (l2 = 1) over [L, N, 3]
 *   term 1  albedo     sum |alb - alb_j| over [N, 3]              term 2  SG weights  sum |wgt - wgt_j| over [N, nb]
 *   term 3  visibility sum |vis[v, n, 0] - vis_gt[v, n]| (or squared) over [V, N]   (vis is [V, N, 3])
 *   term 4  normal     sum (nrm - normalize(nrm_gt))^2 over [N, 3]   term 5  normal smoothness  sum |nrm - nrm_j|
 * A term whose first pointer is NULL is skipped (0).  out[i] = sum_i * inv_denom[i] (i < 6), out[6] = sum_i weight[i] out[i];
 * inv_denom, weight: HOST arrays [6] (passed by value).  count_dev (optional, device [1] float): the masked-pixel count;
 * when given, inv_denom holds only the per-pixel multiplicities (1 / (L 3), 1 / 3, 1 / nb, 1 / V, 1 / 3, 1 / 3) and
 * every term (and every gradient in _bwd) is divided by count_dev[0] on the device (0 for an empty mask), so the count --
 * possibly all-reduced over ranks -- never has to reach the host.  partial: device workspace of >= 2048 * 6 floats.
 * Deterministic summation order.
 * psn_stage2_loss_bwd writes d(out[6]) / d(input) * g_total[0] for every non-NULL d_* (k_i = weight_i * inv_denom_i on the
 * host); masked-out pixels get zeros; d_vis channels 1, 2 are zero. */
int psn_stage2_loss_fwd(const float* rgb, const float* rgb_gt, int L, const float* alb, const float* alb_j, const float* wgt,
                        const float* wgt_j, int nb, const float* vis, const float* vis_gt, int V, const float* nrm,
                        const float* nrm_gt, const float* nrm_j, const unsigned char* mask_a, const unsigned char* mask_b,
                        int64_t N, int l2, const float* inv_denom, const float* weight, const float* count_dev, float* partial,
                        float* out, void* stream);
int psn_stage2_loss_bwd(const float* g_total, const float* rgb, const float* rgb_gt, int L, float k_rgb, float* d_rgb,
                        const float* alb, const float* alb_j, float k_alb, float* d_alb, float* d_alb_j, const float* wgt,
                        const float* wgt_j, int nb, float k_wgt, float* d_wgt, float* d_wgt_j, const float* vis, const float* vis_gt,
                        int V, float k_vis, float* d_vis, const float* nrm, const float* nrm_gt, const float* nrm_j, float k_nrm,
                        float k_nrmj, float* d_nrm, float* d_nrm_j, const unsigned char* mask_a, const unsigned char* mask_b,
                        int64_t N, int l2, const float* count_dev, void* stream);

/* ------------------------------------------------------------------------
 * Stage-1 losses of the sync-free training forward, stage1/model/losses.py:24-70, over per-ray tensors ([N] / [N, 3]):
 *   sums[0] = sum |rgb - rgb_gt|                               sums[4] = #hit
 *   sums[1] = sum over hit rays of diff (smoothness)           sums[5] = #norm_mask
 *   sums[2] = sum over norm_mask of |normal - normal_gt|       sums[6] = #mask_valid
 *   sums[3] = sum over mask_valid of BCE(clamp(acc, 0, 1), mask_gt) with the log terms clamped at -100
 * terms[0..3] = sums[0] / n_rays, sums[1] / max(sums[4], 1), sums[2] / max(sums[5], 1), sums[3] / max(sums[6], 1);
 * terms[4] = weights[0] terms[0] + weights[1] terms[1] (+ weights[2] terms[2]) (+ weights[3] terms[3]), a term with NULL
 * inputs (or weight 0 for the first two) being skipped.  weights: HOST array [4] (full, grad, norm, mask).  n_rays: the
 * ray count of the whole batch (all ranks).  psn_stage1_loss_fwd with terms == NULL stops after the sums, so that the caller
 * can all-reduce sums[4..6] over ranks and finish with psn_stage1_loss_terms.  partial: device workspace of
 * psn_stage1_loss_partial_floats() floats.  Deterministic summation order.
 * psn_stage1_loss_bwd: d terms[4] / d input * g_loss[0] for every non-NULL d_* (counts read from sums on the device).
 * psn_surface_normals_{fwd,bwd}: stage1/model/rendering.py:200-212 for g [2 N, 3] (gradients of the geometry field at the N
 * surface points and at their N jittered neighbours): n = g / (|g| + eps); norm_pred [N, 3] = hit ? n[:N] : 0;
 * diff [N] = |n[:N] - n[N:]|; backward: dg [2 N, 3] from d_norm_pred / d_diff (either may be NULL). */
int psn_stage1_loss_partial_floats(void);
int psn_stage1_loss_fwd(const float* rgb, const float* rgb_gt, const float* diff, const unsigned char* hit, const float* normal,
                        const float* normal_gt, const unsigned char* norm_mask, const float* acc, const float* mask_gt,
                        const unsigned char* mask_valid, int64_t N, int64_t n_rays, const float* weights, float* partial,
                        float* sums, float* terms, void* stream);
int psn_stage1_loss_terms(const float* sums, int64_t n_rays, const float* weights, int has_grad, int has_norm, int has_mask,
                          float* terms, void* stream);
int psn_stage1_loss_bwd(const float* g_loss, const float* sums, const float* rgb, const float* rgb_gt, const unsigned char* hit,
                        const float* normal, const float* normal_gt, const unsigned char* norm_mask, const float* acc,
                        const float* mask_gt, const unsigned char* mask_valid, int64_t N, int64_t n_rays, const float* weights,
                        float* d_rgb, float* d_diff, float* d_normal, float* d_acc, void* stream);
int psn_surface_normals_fwd(const float* g, const unsigned char* hit, int64_t N, float eps, float* norm_pred, float* diff,
                            void* stream);
int psn_surface_normals_bwd(const float* g, const unsigned char* hit, int64_t N, float eps, const float* d_norm_pred,
                            const float* d_diff, float* dg, void* stream);

/* Sums of x [V, Ns, C] (C <= 256, V <= PSN_PAIR_SUMS_MAX_V) over V and over Ns in one pass: sx [Ns, C] = sum_v x and
 * sl_part [*n_chunks, V, C] = per-chunk partial sums over Ns (the caller adds the chunks; at most PSN_PAIR_SUMS_MAX_CHUNKS).
 * Used for the separable weight gradient of a layer whose input block is [table(x_n) | table(l_v)]. */
#define PSN_PAIR_SUMS_MAX_V 16
#define PSN_PAIR_SUMS_MAX_CHUNKS 2048
int psn_pair_sums(const float* x, int V, int64_t Ns, int C, float* sx, float* sl_part, int* n_chunks, void* stream);
/* The separable input-block weight gradients of the visibility network's backward (stage2/model/renderer.py:191-200, 251-262:
 * input row k = v Ns + n is [table(x_n) | table(l_v)]) for up to PSN_PAIR_GROUP_MAX input layers in two launches (three when the
 * per-chunk partial sums are many: their reduction is then sliced over 32 blocks per output tile).  Per item,
 * with x = d z [V, Ns, C]:  sx [Ns, C] = sum_v x (the K = Ns operand of d W_x);  dWl [C, ld_w] (columns < n_pe written) =
 * (sum_n x[v])^T pe_l[v] summed over v, pe_l [V, ld_pe];  bias [C] = sum over v and n of x (NULL: not wanted).
 * workspace: psn_pair_sums_group_workspace(...) floats, 16-byte aligned.  Fixed summation order (deterministic). */
#define PSN_PAIR_GROUP_MAX 4
typedef struct { const float* x; float* sx; float* dWl; float* bias; } PsnPairSumsItem;
int64_t psn_pair_sums_group_workspace(int n_items, int V, int64_t Ns, int C);
int psn_pair_sums_group(int n_items, const PsnPairSumsItem* items, int V, int64_t Ns, int C, const float* pe_l, int64_t ld_pe, int n_pe,
                        int64_t ld_w, float* workspace, void* stream);

/* torch.optim.SparseAdam on the touched rows of up to PSN_ROW_ADAM_MAX tables in one launch (the per-light direction
 * [n, 3] and intensity [n, 1] embeddings, stage2/trainer.py:126-168): rows listed in idx [n_idx] int64 (duplicates
 * allowed) advance their moments and move by -step_size m / (sqrt(v) + eps), step_size = lr sqrt(1 - b2^t) / (1 - b1^t)
 * computed by the caller; all other rows and their moments stay untouched.  grad is the DENSE gradient [rows, cols]. */
#define PSN_ROW_ADAM_MAX 4
typedef struct {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    int64_t rows; int cols;
    float one_minus_beta1, one_minus_beta2, eps, step_size;  /* 1 - beta computed in double by the caller, like torch */
} PsnRowAdamItem;
int psn_row_adam(int n_items, const PsnRowAdamItem* items, const int64_t* idx, int n_idx, void* stream);
/* The same launch with the step sizes read from DEVICE memory (step_sizes_dev [n_items], item i uses element i; the host
 * values in items[] are ignored): identical from step to step, so it can be replayed from a HIP graph while the host
 * refreshes the scalars (psnerf_amd/stage2/graph.py; the lr schedule and the bias corrections of stage2/trainer.py:402-410,462-464). */
int psn_row_adam_dev(int n_items, const PsnRowAdamItem* items, const int64_t* idx, int n_idx, const float* step_sizes_dev, void* stream);

/* ------------------------------------------------------------------------
 * The launch-bound tail of a stage-2 train step as single kernels (csrc/small.hip).
 *
 * psn_normalize_rows_{fwd,bwd}: torch.nn.functional.normalize(x, p=2, dim=-1, eps) of [n, 3] rows, y = x / max(|x|, eps),
 *   and autograd's backward of it (div -> clamp_min -> norm); stage2/model/renderer.py:129,137 (the predicted normals).
 * psn_light_rows_{fwd,bwd}: the light-table lookups of a step, stage2/trainer.py:376-379: dir_out [n_idx, 3] =
 *   normalize(dir_table[idx]), int_out [n_idx] = int_table[idx] (int_table / int_out may both be NULL).  Backward writes
 *   the DENSE table gradients d_dir_table [n_rows, 3], d_int_table [n_rows] (every row: zeros where untouched, duplicate
 *   indices summed in list order -- what nn.Embedding(sparse=False) + F.normalize produce); a NULL gradient pair is skipped.
 * psn_camera_rays: stage2/utils/rend_util.py:90-147 for a 4 x 4 pose: out[i] = scale * normalize(R [(u - cx) / fx,
 *   (v - cy) / fy, 1]) for pixel idx[i] (idx NULL: pixel i) of uv [N, 2]; pose / intrinsics: 16 floats each, row-major, on
 *   the device.
 * psn_stage1_rays: the ray set-up of a stage-1 batch (batch size 1) in one launch: cam[i] = world_mat[:3, 3]
 *   (stage1/model/common.py:205-207), rays[i] = normalize(R [(px - cx) / fx, (py - cy) / fx, 1]) (common.py:210-226: BOTH
 *   axes over fx = K[0][0], the reference's quirk; rendering.py:64-65), far[i] = the sphere exit depth max(sqrt(b^2 - (|cam|^2 -
 *   radius2)) - b, 0), 0 for rays that miss the sphere (rendering.py:576-596).  camera_mat: k_ld x k_ld row-major (3 or 4),
 *   world_mat 4 x 4, both on the device.
 * psn_surface_points: rendering.py:516-522 + :84-108 -- d = flags & 1 ? d_pred : inf; d = flags & 2 ? d : 0 (d_i, optional
 *   output); obj_mask = finite(d) & d != 0; dists = obj_mask ? d : (d == 0 ? 0 : 1); points = cam + rays * dists.
 * psn_stage1_targets: the ground truth of the sampled pixels, stage1/model/common.py:172-202 (nearest-neighbour
 *   grid_sample, align_corners, at x = 2 px / w - 1, y = 2 py / h - 1) applied to the image [3, h, w], the masks [h, w] (a NULL
 *   mask / mask_valid image is all ones; outputs: mask_gt float 0/1, mask_valid_out / norm_mask_out bytes 0/1 = torch.bool
 *   storage) and the normal map [3, h, w] (training.py:176-191: norm_mask cleared where the UNROTATED n_z < cos_thresh when
 *   use_angle, then normal_gt = R diag(1, -1, -1) n with R = world_mat[:3, :3]).  Optional outputs may be NULL.
 * psn_adam_flat: torch.optim.Adam's update (torch/optim/adam.py::_multi_tensor_adam; amsgrad off, weight_decay 0;
 *   stage2/trainer.py:126-133) over ranges of one flat parameter / gradient / exp_avg / exp_avg_sq allocation:
 *   m += (1 - beta1)(g - m); v = v beta2 + (1 - beta2) g g; p += neg_step_size * m / (sqrt(v) / bias_correction2_sqrt + eps),
 *   neg_step_size = -lr / (1 - beta1^t) per range (parameters that joined the optimisation later have their own t).
 * ---------------------------------------------------------------------- */
int psn_normalize_rows_fwd(const float* x, int64_t n, float eps, float* y, void* stream);
int psn_normalize_rows_bwd(const float* x, const float* g, int64_t n, float eps, float* dx, void* stream);
int psn_light_rows_fwd(const float* dir_table, const float* int_table, const int64_t* idx, int n_idx, float eps, float* dir_out,
                       float* int_out, void* stream);
int psn_light_rows_bwd(const float* dir_table, const int64_t* idx, int n_idx, int64_t n_rows, float eps, const float* g_dir,
                       const float* g_int, float* d_dir_table, float* d_int_table, void* stream);
int psn_camera_rays(const float* uv, const float* pose, const float* intrinsics, const int64_t* idx, int64_t n, float scale,
                    float* out, void* stream);
int psn_stage1_rays(const float* pix, const float* camera_mat, int k_ld, const float* world_mat, float radius2, int64_t n,
                    float* cam, float* rays, float* far, void* stream);
int psn_surface_points(const float* d_pred, const int* flags, const float* cam, const float* rays, int64_t n, float* d_i,
                       float* dists, unsigned char* obj_mask, float* points, void* stream);
int psn_stage1_targets(const float* pix, int64_t n, int h, int w, const float* img, const float* mask, const float* mask_valid,
                       const float* normal, const float* norm_mask, const float* world_mat, int use_angle, float cos_thresh,
                       float* rgb_gt, float* mask_gt, unsigned char* mask_valid_out, float* normal_gt,
                       unsigned char* norm_mask_out, void* stream);
/* Front of a stage-2 step (stage2/trainer.py:355-392, stage2/model/renderer.py:110-152), one launch each:
 * psn_mask_count: out[0] = #{i : mask_a[i] && mask_b[i]} as a float (masks = torch.bool storage, mask_b may be NULL; n < 2^24) --
 *   the masked-pixel count that normalises the losses (stage2/model/loss.py:27-38);
 * psn_inverse_index: inv[p] = position of pixel p in the ascending surface-pixel list idx[0 .. ns), or -1 (the pixel -> row map
 *   of psn_scatter_rows). */
/* psn_copy2d_group: up to PSN_COPY2D_MAX row-major 2-D copies dst[r][c] = src[r][c] (r < rows, c < cols; row strides ld_*) in
 *   one launch -- the side tables of a weight pack (init-table column slices of the parameters, bias segments), which were
 *   a slice copy or a concatenation each.  HOST array of items holding device pointers. */
#define PSN_COPY2D_MAX 24
typedef struct {
    const float* src; int64_t ld_src;
    float* dst; int64_t ld_dst;
    int rows, cols;
} PsnCopy2dItem;
int psn_copy2d_group(int n_items, const PsnCopy2dItem* items, void* stream);
/* Up to PSN_COPY_BYTES_MAX contiguous copies of any element type in ONE launch: a training batch (stage2/trainer.py:364-392: ~17
 * tensors of floats, bools and indices) into the fixed input buffers of a captured HIP graph (one device-to-device copy each cost
 * ~5 us of dependent-launch latency).  `aligned` is filled in by the library (both pointers 16-byte aligned). */
#define PSN_COPY_BYTES_MAX 24
typedef struct { const void* src; void* dst; int64_t n_bytes; int aligned; } PsnCopyBytesItem;
int psn_copy_bytes_group(int n_items, const PsnCopyBytesItem* items, void* stream);
int psn_mask_count(const unsigned char* mask_a, const unsigned char* mask_b, int64_t n, float* out, void* stream);
/* psn_surface_index: idx[0 .. ns) = ascending positions of the set bytes of mask [n] (= surface_mask[0].nonzero(),
 *   stage2/model/renderer.py:125, without a host synchronisation), idx[ns .. cap) = idx[ns - 1]: a FIXED-size list whose
 *   dead tail repeats the last surface pixel (0 for an empty mask); count[0] (may be NULL) = ns as a float.  A list longer than
 *   cap is truncated.  Kernels that read such a list treat entries behind the first of equal neighbours as dead rows
 *   (psn_gather_rows_valid); psn_inverse_index maps a pixel to the FIRST of equal entries; count (device float [1], or NULL = ns):
 *   only the first count[0] entries of idx are real -- an EMPTY mask then gives no pixel a row. */
int psn_surface_index(const unsigned char* mask, int64_t n, int64_t cap, int64_t* idx, float* count, void* stream);
int psn_inverse_index(const int64_t* idx, int64_t ns, int64_t n_pix, int* inv, const float* count, void* stream);
/* psn_view_batch: ONE training batch of stage 2 gathered on the device from view tables that stay resident in HBM -- replaces the
 * host-side item construction of stage2/datasets/dataset.py:137-199 (light subset :149-151 -> rgb = imgs[view][lidx] * object_mask
 * :172; pixel subset :182-195 -> every per-pixel tensor indexed with sampling_idx), the un-batching of stage2/trainer.py:364-367 and
 * the vis_plus selection of stage2/trainer.py:384-392 (vis_train_gt = vis_plus_v[sidx][:, sampling_idx]) + the upload of the batch.
 * Only the drawn INDEX lists (lidx, pix, vidx: device int64) are per-step inputs; a data-parallel rank passes its slice of pix.
 *   images [n_view_lights, hw, 3] of image_type 0 = float32 | 1 = uint8 | 2 = uint16; integer images are decoded through lut
 *     ([256] / [65536] floats = value / 255 exactly as the reference's numpy division forms it, dataset.py:121);
 *   object_mask / surface_mask [hw] bytes, points / normal [hw, 3], visibility [n_view_lights, hw], vis_plus [n_rows, hw],
 *     light_direction [n_view_lights, 3] floats;
 *   pix [n] pixel indices, or NULL = the identity range pix0 .. pix0 + n (a test-split item, dataset.py:182 not taken); width = image
 *     width (uv = (pix % width, pix / width) as floats, dataset.py:138-140).
 * Outputs (any may be NULL = not produced; all contiguous): rgb [n_lights, n, 3]; object_mask_out / surface_mask_out [n] bytes;
 * uv [n, 2]; points_out / normal_out [n, 3]; visibility_out [n_lights, n]; vis_train_gt [n_vis, n]; sampling_idx_out [n] int64;
 * light_direction_out [n_lights, 3] = light_direction[lidx]. */
typedef struct {
    const void* images; int image_type; const float* lut;
    const unsigned char* object_mask; const unsigned char* surface_mask;
    const float* points; const float* normal; const float* visibility; const float* vis_plus; const float* light_direction;
    int64_t hw; int width;
    const int64_t* lidx; int n_lights;
    const int64_t* pix; int64_t pix0; int64_t n;
    const int64_t* vidx; int n_vis;
    float* rgb; unsigned char* object_mask_out; unsigned char* surface_mask_out; float* uv; float* points_out; float* normal_out;
    float* visibility_out; float* vis_train_gt; int64_t* sampling_idx_out; float* light_direction_out;
} PsnViewBatch;
int psn_view_batch(const PsnViewBatch* b, void* stream);
#define PSN_ADAM_MAX_SEGS 16
typedef struct {
    int64_t offset, grad_offset, n;    /* elements [offset, offset + n) of param / exp_avg / exp_avg_sq, [grad_offset, ..+n) of grad */
    float neg_step_size;               /* -lr / (1 - beta1^t) */
    float bias_correction2_sqrt;       /* sqrt(1 - beta2^t) */
} PsnAdamSeg;
int psn_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int n_segs, const PsnAdamSeg* segs,
                  float one_minus_beta1, float beta2, float one_minus_beta2, float eps, void* stream);
/* psn_adam_flat with (neg_step_size, bias_correction2_sqrt) of range i read from DEVICE memory, seg_scalars_dev[2 i],
 * seg_scalars_dev[2 i + 1] (the host values in segs[] are ignored): a launch that is identical from step to step (HIP-graph
 * replay; the host refreshes the scalars before each replay). */
int psn_adam_flat_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int n_segs, const PsnAdamSeg* segs,
                      float one_minus_beta1, float beta2, float one_minus_beta2, float eps, const float* seg_scalars_dev, void* stream);

/* ------------------------------------------------------------------------
 * Weight normalisation of up to PSN_WN_MAX_ITEMS layers in one launch: nn.utils.weight_norm(nn.Linear) as used by every
 * layer of stage1/model/network.py:37-66 (state_dict keys weight_g [rows,1], weight_v [rows,cols]).
 *   fwd: w = v * (g / |v|_row) [* scale]       (scale = 1/sqrt(2) folds the cat[x, pe]/sqrt(2) of network.py:90-91)
 *   bwd: given dw (gradient of the scaled w):  dv, dg
 * All matrices row-major and dense (row stride = cols).  HOST array of items holding device pointers.
 * ---------------------------------------------------------------------- */
#define PSN_WN_MAX_ITEMS 16
typedef struct {
    const float* v;   /* [rows, cols] */
    const float* g;   /* [rows] */
    float* w;         /* fwd: [rows, cols] output */
    const float* dw;  /* bwd: [rows, cols] */
    float* dv;        /* bwd: [rows, cols] output */
    float* dg;        /* bwd: [rows] output */
    int rows, cols;
    float scale;
} PsnWnItem;
int psn_weight_norm_fwd(int n_items, const PsnWnItem* items, void* stream);
int psn_weight_norm_bwd(int n_items, const PsnWnItem* items, void* stream);

/* ------------------------------------------------------------------------
 * bf16 inference engine (evaluation / relighting only; BASELINE config 5 "bf16 MFMA path ... envmap relight eval"):
 * the 256-wide ReLU networks of stage2/model/renderer.py:34-49 on v_mfma_f32_32x32x16_bf16 -- weights, input
 * features and post-ReLU activations rounded to bf16 (RNE), fp32 accumulation, fp32 output.  Replaces the no-grad
 * evaluation of visibility_net in stage2/eval.py:199-218 (via renderer.py:191-200) when the caller opts in; training
 * and every parity-gated path use psn_mlp_infer (exact fp32).
 *
 * Network: n_hidden hidden layers of width 256 with ReLU; layer 0 reads the input block only, layer l >= 1 reads the
 * previous activations and, if has_in[l], the input block again (skip connection cat[y, x]); a final layer with
 * n_out <= 32 outputs followed by out_act.  The input block of row q is
 * [ A[(q / a_div) % a_mod, 0:64] | B[(q / b_div) % b_mod, 0:64] ] with bf16 tables of 64 columns (128-byte rows).
 *
 * Weight stream (bf16, one contiguous buffer, in execution order), each piece written by psn_mlp_pack_bf16:
 *   layer 0:      input block k-steps 0..7 [W_a | W_b] natural order, 1 bias k-step            (72 KB)
 *   layer l >= 1: activation k-steps 0..7 (permuted), 1 bias k-step (72 KB); if has_in[l]: input block k-steps 0..7
 *                 (64 KB); activation k-steps 8..15 (64 KB)
 *   final:        activation k-steps 0..15 with n_ot = 1 (16 KB); its bias is added in fp32 from final_bias[32].
 *                 The kernel always requests whole 72 KB stages: the buffer must extend (any content) 56 KB past
 *                 the final block.
 * A k-step covers 16 K indices; a bias k-step is the [256, 2] matrix (bf16(b), b - bf16(b)) packed in natural order.
 * ---------------------------------------------------------------------- */
typedef struct {
    int32_t n_hidden;
    int32_t n_out;
    int32_t out_act; /* PSN_OUT_* */
    int32_t reserved;
    uint8_t has_in[PSN_MLP_MAX_LAYERS + 4];
} PsnBf16Desc;

/* k-steps [ks0, ks0 + n_ks) of W[rows, cols] (row-major fp32, ldw floats per row, zero-extended) for n_ot output tiles
 * of 32 (8 = hidden layer, 1 = final layer) -> dst[n_ks][n_ot][64 lanes][8] bf16.  permuted != 0: K follows the
 * register layout of the previous layer's activations; 0: natural order (input block, bias columns). */
int psn_mlp_pack_bf16(const float* W, int64_t ldw, int rows, int cols, int permuted, int n_ot, int ks0, int n_ks,
                      uint16_t* dst, void* stream);

/* out [n_rows, n_out] fp32.  tab_b may be NULL (input block = table A only).  All device buffers 16-byte aligned. */
int psn_mlp_infer_bf16(const PsnBf16Desc* desc, const uint16_t* packed_w, const float* final_bias, const uint16_t* tab_a,
                       int64_t a_div, int64_t a_mod, const uint16_t* tab_b, int64_t b_div, int64_t b_mod, int64_t n_rows,
                       float* out, void* stream);

/* Grouped form for the light-major rows (g, n) -> g * rows_per_group + n of stage2/model/renderer.py:163,193 (g = light,
 * n = surface point; the 512-light evaluation of stage2/eval.py:199-218): the input block of a row is table A row n
 * ([rows_per_group, 64] bf16) and the group's part of every input layer, W_b * B[g] + b, arrives as that group's bias:
 * group_bias [n_groups][n_in][8 KB] = one bias k-step per (group, input layer in network order), written by
 * psn_bf16_pack_group_bias from fp32 rows.  The light's encoding and its weight columns therefore stay in fp32 (one small
 * product per light and input layer on the host side) and cost no k-steps per row.
 * Weight stream: as above with 4 input k-steps (table A columns only) instead of 8 and WITHOUT the bias k-step of the
 * layers that read the input block:
 *   layer 0: input k-steps 0..3 (32 KB);  layer l >= 1: activation k-steps 0..7, bias k-step unless has_in[l];
 *   if has_in[l]: input k-steps 0..3 (32 KB); activation k-steps 8..15;  final: 16 KB, then >= 56 KB of padding.
 * out [n_groups * rows_per_group, n_out] fp32. */
int psn_mlp_infer_bf16_grouped(const PsnBf16Desc* desc, const uint16_t* packed_w, const float* final_bias,
                               const uint16_t* tab_a, int64_t rows_per_group, const uint16_t* group_bias, int64_t n_groups,
                               float* out, void* stream);
/* V [n, 256] fp32 -> dst [n][4096] bf16: bias k-steps (K slot 0 = bf16(v), slot 1 = bf16(v - slot 0), the rest 0). */
int psn_bf16_pack_group_bias(const float* V, int64_t n, uint16_t* dst, void* stream);

/* ------------------------------------------------------------------------
 * Split-bf16 ("bf16x6") inference engine, csrc/mlp_infer_x3.hip -- EXPERIMENT, opt-in, gradient-free rows only.  The
 * network of psn_mlp_infer_bf16_grouped with every fp32 operand carried as three bf16 planes (hi / mid / lo, exact sum)
 * and every product as the six partial products of weight >= 2^-16: fp32-class results from the bf16 matrix pipe.
 *   psn_x3_pack        W [rows, cols] fp32 -> k-steps [ks0, ks0 + n_ks) as [ks][n_ot tiles][3 planes][64 lanes][8] bf16 (K order
 *                      as psn_mlp_pack_bf16: natural for the input block, permuted for activations)
 *   psn_x3_pack_bias   V [n, 256] fp32 -> n bias k-steps [8 tiles][64 lanes][8] bf16 (K slots 0..2 = the three pieces)
 *   psn_mlp_infer_x3_grouped: rows (g, n) -> g * rows_per_group + n; packed_w = the weight stream in execution order (48 KB
 *     stages of 2 k-steps: every hidden layer >= 1 = its 16 activation k-steps; final layer = one stage [16 k-steps][3 planes],
 *     + 56 KB of padding); bias_steps [n_hidden][8 KB] (used for the layers without an input block).  The layers that read
 *     the input block start from U[n] + V[g] (both fp32, [., n_input_layers * 256]): U = W_a x_n per row, V = W_b x_g + b per
 *     group, as the init tables of psn_mlp_infer; layer 0 is that sum alone.  out [n_groups * rows_per_group, n_out] fp32.
 * ---------------------------------------------------------------------- */
int psn_x3_pack(const float* W, int64_t ldw, int rows, int cols, int permuted, int n_ot, int ks0, int n_ks, uint16_t* dst, void* stream);
int psn_x3_pack_bias(const float* V, int64_t n, uint16_t* dst, void* stream);
int psn_mlp_infer_x3_grouped(const PsnBf16Desc* desc, const uint16_t* packed_w, const uint16_t* bias_steps, const float* final_bias,
                             const float* U, int64_t rows_per_group, const float* V, int64_t n_groups, float* out, void* stream);
/* The stage-1 occupancy network (stage1/model/network.py:85-101: softplus(beta = 100) stack on the positional encoding of the
 * point, cat[h, pe] / sqrt(2) in front of the skip layer, output row 0) on the split-bf16 engine, for GRADIENT-FREE queries
 * (shadow rays rendering.py:378-408, ray march :447-462, shape_extract :297-376): out[r] = sigmoid(-10 logit(points[r])).
 * desc: n_hidden, n_out = 1, out_act; packed_w: layer 0 as 4 natural-order k-steps of its encoding columns (psn_x3_pack,
 * permuted = 0), every further hidden layer 16 permuted k-steps (the skip layer as ONE 256-input matrix, scaled by 1 / sqrt(2)),
 * the final row as one stage; bias_steps [n_hidden][4096] (psn_x3_pack_bias); final_bias [32].  The kernel forms the encoding
 * (pe_octaves, pe_scale: the expressions of psn_pe_encode) and writes it into input features pe_first .. of layer skip_layer
 * (-1: none).  n_rows_dev / out_rows: as psn_mlp_infer_pe_indirect (row count on the device, outputs scattered). */
int psn_mlp_infer_x3_occ(const PsnBf16Desc* desc, const uint16_t* packed_w, const uint16_t* bias_steps, const float* final_bias,
                         const float* points, int64_t n_rows, const long long* n_rows_dev, const int64_t* out_rows, int pe_octaves,
                         float pe_scale, int skip_layer, int pe_first, float* out, void* stream);
/* The ray-march sweep of stage1/model/rendering.py:447-462 on the split-bf16 engine (opt-in experiment; psn_march_sweep is the exact
 * form): occ [n_rays, n_steps] = sigmoid(-10 logit) at the points origin + dir (near (1 - u_m) + far u_m), formed and encoded in the
 * kernel; a workgroup = 128 consecutive steps of one ray, block-major.  skip [n_rays] int32, zeroed by the caller (or NULL = every
 * block is evaluated): the blocks behind a ray's first sign change are left out -- their entries stay unwritten, and nothing
 * reads them (psn_first_crossing stops at the first crossing).  n_steps a multiple of 128; desc / packed_w / bias_steps /
 * final_bias / skip_layer / pe_first as for psn_mlp_infer_x3_occ.  n_blocks (or NULL): counts the evaluated blocks. */
int psn_march_sweep_x3(const PsnBf16Desc* desc, const uint16_t* packed_w, const uint16_t* bias_steps, const float* final_bias,
                       const float* origin, const float* dir, const float* far, const float* u, const float* omu, float near,
                       int64_t n_rays, int n_steps, float tau, int pe_octaves, float pe_scale, int skip_layer, int pe_first,
                       int* skip, float* occ, unsigned long long* n_blocks, void* stream);


/* ------------------------------------------------------------------------
 * Spherical-Gaussian shading over the light-major rows (l, n) -> l*Ns + n:
 * stage2/model/sgbasis.py:16-32 + stage2/model/renderer.py:174-204
 *   h = normalize(l + v); D_k = exp(max(lobe_k,0) (h.n - 1)); spec_c = max(sum_k w_{c,k} D_k, 0)
 *   rgb = clamp((albedo + spec) * I_l * (l.n) * clamp(vis,0,1), 0, 1)
 * light_dir [L,3]; view/normal/albedo [Ns,3]; weights [Ns, 3*nb] (specular_rgb, channel-major as
 * sgbasis.py:27 view(-1,3,nb)) or [Ns, nb]; lobe [nb<=9]; light_int [L, int_ch] or NULL (then light_int_scalar),
 * int_ch = 1, or 3 for RGB environment-map lights (stage2/eval.py:200; forward only);
 * vis [L*Ns] or NULL.  rgb [L*Ns,3]; spec [L*Ns,3] (specular_rgb) or [L*Ns].
 * Backward: g_rgb [L*Ns,3], g_spec like spec or NULL -> d_albedo [Ns,3], d_weights like weights,
 * d_normal [Ns,3], d_vis [L*Ns] or NULL, d_light_dir [L,3], d_light_int [L] or NULL;
 * workspace >= ceil(Ns/64)*L*4 floats.
 * ---------------------------------------------------------------------- */
int psn_sg_shade_fwd(const float* light_dir, const float* view, const float* normal, const float* albedo,
                     const float* weights, const float* lobe, const float* light_int, int int_ch, float light_int_scalar,
                     const float* vis, int L, int64_t Ns, int nb, int specular_rgb, float* rgb, float* spec,
                     void* stream);
int psn_sg_shade_bwd(const float* light_dir, const float* view, const float* normal, const float* albedo,
                     const float* weights, const float* lobe, const float* light_int, float light_int_scalar,
                     const float* vis, int L, int64_t Ns, int nb, int specular_rgb, const float* g_rgb,
                     const float* g_spec, float* d_albedo, float* d_weights, float* d_normal, float* d_vis,
                     float* d_light_dir, float* d_light_int, float* workspace, void* stream);

/* ------------------------------------------------------------------------
 * GGX microfacet shading (train.render_model = microfacet): stage2/model/microfacet.py:35-114 followed by
 * stage2/model/renderer.py:187-204, same row layout as psn_sg_shade_*.  rough [Ns] (sigmoid output of rough_net),
 * f0 = brdf.fresnel_f0.  rgb [L*Ns,3].  Backward: d_albedo [Ns,3], d_rough [Ns], d_normal [Ns,3], d_vis [L*Ns] or
 * NULL, d_light_dir [L,3], d_light_int [L] or NULL; workspace >= ceil(Ns/64)*L*4 floats.
 * ---------------------------------------------------------------------- */
int psn_mf_shade_fwd(const float* light_dir, const float* view, const float* normal, const float* albedo,
                     const float* rough, const float* light_int, float light_int_scalar, float f0, const float* vis,
                     int L, int64_t Ns, float* rgb, void* stream);
int psn_mf_shade_bwd(const float* light_dir, const float* view, const float* normal, const float* albedo,
                     const float* rough, const float* light_int, float light_int_scalar, float f0, const float* vis,
                     int L, int64_t Ns, const float* g_rgb, float* d_albedo, float* d_rough, float* d_normal,
                     float* d_vis, float* d_light_dir, float* d_light_int, float* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PSNERF_HIP_H */
