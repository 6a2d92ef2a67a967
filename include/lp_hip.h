/* lp_hip.h - C ABI of liblp_hip.so: the MI355X (gfx950) kernels behind the Lightning Pose heatmap-tracker
 * training step.
 *
 * The reference (paninski-lab/lightning-pose v2.4.0) is pure Python; the functions below replace the third-party
 * native ops its hot path invokes.  Each entry point cites the reference interface
 * it stands in for (paths relative to the reference tree).  INTEGRATION.md shows the ctypes binding a maintainer
 * of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned, contiguous memory (torch tensor .data_ptr());
 *     the library never allocates, frees or retains pointers; structs are passed by host pointer and only read
 *     during the call
 *   - `stream` is the hipStream_t the kernels are enqueued on (torch.cuda.current_stream().cuda_stream);
 *     calls only enqueue - no device synchronisation, no host read-back
 *   - return value: 0 = LP_OK, < 0 = argument / shape error (nothing was launched), > 0 = hipError_t
 *   - re-entrant and thread-safe: no entry point keeps state between calls or reads the environment.  The only process-wide data is
 *     the table of A/B switches (LP_CONV_PIPE, LP_CONV_HALO, LP_CONV_RES2D, LP_INFER_PIPE, LP_GEMM_PIPE, LP_WGRAD_PIPE,
 *     LP_STEM_2D, LP_STEM_WGRAD_NB, LP_POOL_V2, LP_CONV_MAX_WGS, LP_BN_BWD_WGS_PER_CU), read from the environment ONCE when the library is loaded and immutable afterwards -
 *     except through lp_config_reload_env(), a test / A-B hook that must not run concurrently with other calls
 *   - limits: lp_bn_bwd_apply WITHOUT its terms_ws workspace covers C <= 2048 channels (the per-launch correction table then lives in
 *     LDS; LP_ERR_UNSUPPORTED beyond); with the workspace any C that is a multiple of 8
 *   - heat-maps are fp32 NCHW (B, K, h, w); keypoints are fp32 (B, K, 2) = the reference's (B, 2K) row layout;
 *     backbone activations are bf16 NHWC; weights fp32 masters + bf16 GEMM copies
 */
#ifndef LP_HIP_H
#define LP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lp_stream_t; /* hipStream_t */

/* Re-read the LP_* switches from the environment into the library's switch table (done once, implicitly, at load).  For tests and A/B
 * scripts that flip a switch inside one process; not to be called while another thread is inside an lp_* call. */
int lp_config_reload_env(void);

enum {
    LP_OK = 0,
    LP_ERR_ARGUMENT = -1,    /* null pointer / non-positive dimension  -> ValueError on the Python side */
    LP_ERR_UNSUPPORTED = -2, /* shape outside the instantiated kernels -> NotImplementedError          */
};

enum { LP_TF_NONE = 0, LP_TF_SINGLE = 1, LP_TF_PER_FRAME = 2, LP_TF_PER_VIEW = 3 };

/* ABI version of THIS header: bumped whenever the signature of an existing entry point changes (an argument inserted, a struct field added),
 * not only when symbols come or go.  lp_version() returns the value the library was built with; a caller compares the two before its first
 * call (lightning_pose_amd/_lib.py raises LpHipUnavailable on a mismatch) - a library built against an older header would otherwise take,
 * e.g., the stream argument for an inserted flag without any error.  History: 131 = round 5 (decode `prune`, bn_bwd `terms_ws`), 140 = round 6. */
#define LP_HIP_ABI_VERSION 142
int lp_version(void);
const char* lp_strerror(int code);

/* ------------------------------------------------------------------------------------------------------
 * Decode: heat-map -> sub-pixel keypoints + confidence   (models/heads/heatmap.py:103-144 run_subpixelmaxima,
 * :86-100 upsample, data/heatmaps.py:90-142 evaluate_heatmaps_at_location, data/utils.py:191-234
 * undo_affine_transform_batch, data/bboxes.py:222-288 model_to_frame_batch)
 * ------------------------------------------------------------------------------------------------------ */

/* Banded tap tables of the composite `downsample_factor` x (bicubic x2, 5x5 binomial) operator along each axis,
 * built on the host in fp64 (lightning_pose_amd.ops.decode_tables).  ty = lp_decode_window(ds, h). */
typedef struct lp_decode_tables {
    const int* row_base;     /* [h]           first heat-map row of output-row group j's window                */
    const float* row_taps;   /* [h][R][ty]    taps of output row j*R+rr relative to row_base[j], R = 2^ds        */
    const int* col_start;    /* [w*R]         first heat-map column feeding output column c                    */
    const float* col_taps;   /* [w*R][12]     zero padded                                                      */
    const int* colT_start;   /* [w]           first output column fed by heat-map column q   (backward only)   */
    const float* colT_taps;  /* [w][tc]                                                                        */
    int ty, tx, tc;
} lp_decode_tables;

/* keypoint epilogue: undo the augmentation affine, then model px -> frame px */
typedef struct lp_frame_map {
    const float* transforms; /* LP_TF_SINGLE (2,3) | LP_TF_PER_FRAME (B,2,3) | LP_TF_PER_VIEW (V,2,3) | NULL     */
    int tf_mode;
    const float* bbox;       /* (B, 4*V) rows [x, y, h, w] per view, or NULL for identity                       */
    int bbox_stride;         /* 4*V */
    int kp_per_view;         /* K / V */
    float model_h, model_w;  /* network input size */
} lp_frame_map;

int lp_decode_window(int downsample_factor, int n);

/* `prune` (lp_decode_fwd / lp_decode_bwd): which instantiation of the decode kernels runs - 1 = the exactly-pruned ones (terms below e^-50
 * of the largest are skipped: results equal to the last bit or two, 1.4x / 1.8x faster on the peaked maps of a trained head, slower on flat
 * maps), 0 = the plain ones.  An argument of the call since round 5 (no process state; the product chooses it per model from the decode's
 * own sumexp output, ops.py).  Replaces nothing in the reference (models/heads/heatmap.py:103-144 has one code path). */

/* heat (B,K,h,w) -> kp_aug (B,K,2) model px, kp_frame (B,K,2) frame px, conf (B,K), stats (B,K,4) = {max, sumexp, E[x] - x0, E[y] - y0}: opaque
 * to the caller except [1] (ops.py reads it as "how many pixels carry weight"); (x0, y0) = the up-sampled position of the tile's own maximum, which
 * lp_decode_bwd re-derives from the same tile - the two moments are accumulated about that point (round 6: fp32 resolves offsets of a few pixels
 * 100x finer than coordinates up to 384) */
int lp_decode_fwd(const float* heat, int B, int K, int h, int w, int downsample_factor, float temperature,
                  const lp_decode_tables* tables, const lp_frame_map* frame_map, float* kp_aug, float* kp_frame, float* conf,
                  float* stats, int prune, lp_stream_t stream);

/* g_heat (B,K,h,w) (+)= d loss / d heat given d loss / d kp_aug and/or d loss / d kp_frame (either may be NULL) */
int lp_decode_bwd(const float* heat, int B, int K, int h, int w, int downsample_factor, float temperature,
                  const lp_decode_tables* tables, const lp_frame_map* frame_map, const float* stats, const float* g_kp_aug,
                  const float* g_kp_frame, float* g_heat, int accumulate, int prune, lp_stream_t stream);

/* data/utils.py:191-234 undo_affine_transform_batch + data/bboxes.py:222-288 model_to_frame_batch on their own
 * (target keypoints; backward = 1 applies the transposed Jacobian to a gradient). */
int lp_frame_map_apply(const float* kp_in, int B, int K, const lp_frame_map* frame_map, int backward, float* kp_out,
                       lp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Heat-map targets and heat-map losses
 * ------------------------------------------------------------------------------------------------------ */

/* data/heatmaps.py:11-87 generate_heatmaps.  keypoints (B,K,2) image px; visibility int32 (B,K) or NULL. */
int lp_heatmap_gen(const float* keypoints, const int* visibility, int B, int K, int img_h, int img_w, int h, int w, float sigma,
                   float* out, lp_stream_t stream);
/* ... and its gradient with respect to the keypoints (keep_gradients=True, data/heatmaps.py:37-40): grad_keypoints (B,K,2); zero / uniform maps
 * (NaN, out of bounds, visibility < 2) contribute zeros.  Heat-map axes up to 512. */
int lp_heatmap_gen_bwd(const float* keypoints, const int* visibility, int B, int K, int img_h, int img_w, int h, int w, float sigma,
                       const float* grad_out, float* grad_keypoints, lp_stream_t stream);

/* losses/losses.py:706-869 TemporalHeatmapLoss ("temporal_heatmap_mse" = LP_HM_MSE, "temporal_heatmap_kl" = LP_HM_KL): distance
 * between the heat-maps of consecutive frames per keypoint, zeroed where either frame's confidence < prob_threshold, relu(. - epsilon[k]),
 * mean over all (S-1)*K entries.  pred (S,K,h,w), conf (S,K), epsilon (K) device floats; workspace persists to bwd; loss is a device
 * scalar (NaN for S = 1, the mean of an empty tensor).  bwd: gpred (S,K,h,w) = or += gout * d loss / d pred. */
size_t lp_temporal_heatmap_workspace_bytes(int S, int K);
int lp_temporal_heatmap_fwd(int kind, const float* pred, const float* conf, int S, int K, int h, int w, const float* epsilon,
                            float prob_threshold, float* loss, void* workspace, lp_stream_t stream);
int lp_temporal_heatmap_bwd(int kind, const float* pred, int S, int K, int h, int w, const void* workspace, const float* gout,
                            float* gpred, int accumulate, lp_stream_t stream);

/* data/heatmaps.py:90-142 evaluate_heatmaps_at_location: out (B,K) = sum of the (2*radius+1)^2 window of heat (B,K,h,w) around
 * int64(locs (B,K,2) = x, y), zero padded; radius = floor(sigma * num_stds) (2 for the defaults).  The training step gets the
 * same number from lp_decode_fwd's epilogue; this is the standalone form the reference exports. */
int lp_heatmap_confidence(const float* heat, const float* locs, int B, int K, int h, int w, int radius, float* out, lp_stream_t stream);

/* losses/losses.py:229-290,314-335 HeatmapMSELoss (remove_nans -> mse*h*w -> mean).  workspace persists to bwd. */
size_t lp_heatmap_mse_workspace_bytes(int B, int K);
int lp_heatmap_mse_fwd(const float* targ, const float* pred, int B, int K, int h, int w, float* loss, void* workspace,
                       lp_stream_t stream);
int lp_heatmap_mse_bwd(const float* targ, const float* pred, int B, int K, int h, int w, const void* workspace, const float* gout,
                       float* gpred, int accumulate, lp_stream_t stream);
/* The same masked mean for the three supervised heat-map losses: kind = LP_HM_MSE (above), LP_HM_KL (HeatmapKLLoss,
 * losses/losses.py:338-379) or LP_HM_JS (HeatmapJSLoss, :382-423) - kornia's kl_div_loss_2d / js_div_loss_2d of the maps
 * + 1e-10, summed over each map, averaged over the labelled maps.  Workspace as lp_heatmap_mse_workspace_bytes. */
enum { LP_HM_MSE = 0, LP_HM_KL = 1, LP_HM_JS = 2 };
int lp_heatmap_loss_fwd(int kind, const float* targ, const float* pred, int B, int K, int h, int w, float* loss, void* workspace,
                        lp_stream_t stream);
int lp_heatmap_loss_bwd(int kind, const float* targ, const float* pred, int B, int K, int h, int w, const void* workspace,
                        const float* gout, float* gpred, int accumulate, lp_stream_t stream);

/* "unimodal_mse" (not in the reference snapshot; defined by oracle/restated.py unimodal_mse_loss,
 * structurally losses/losses.py:1129-1260).  Same workspace size as lp_heatmap_mse. */
int lp_unimodal_mse_fwd(const float* kp_aug, const float* pred, const float* conf, int S, int K, int img_h, int img_w, int h, int w,
                        float sigma, float prob_threshold, float* loss, void* workspace, lp_stream_t stream);
int lp_unimodal_mse_bwd(const float* kp_aug, const float* pred, int S, int K, int img_h, int img_w, int h, int w, float sigma,
                        const void* workspace, const float* gout, float* gpred, int accumulate, lp_stream_t stream);

/* models/heads/heatmap.py:209-211 spatial_softmax2d(T=1).  Input element (b,k,i) at in[b*sb + i*si + k*sk]. */
int lp_softmax2d_fwd(const float* in, long stride_b, long stride_i, long stride_k, int B, int K, int n, float* out,
                     lp_stream_t stream);
/* gin_bf16: bf16 gradient of the logits in the same strided layout (it feeds the MFMA kernels).  Pixel-major rows (stride_k == 1,
 * stride_i % 8 == 0) are written whole: the pad channels [K, stride_i) of every pixel come out as zeros. */
int lp_softmax2d_bwd(const float* prob, const float* gprob, int B, int K, int n, void* gin_bf16, long stride_b, long stride_i,
                     long stride_k, lp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Keypoint-space losses (loss scalar + gradient for unit upstream, one launch each)
 * ------------------------------------------------------------------------------------------------------ */

/* losses/losses.py:576-703 TemporalLoss.  kp (S,K,2), conf (S,K) or NULL, eps_per_kp (K). */
int lp_temporal_fwd_bwd(const float* kp, const float* conf, int S, int K, const float* eps_per_kp, float prob_threshold,
                        float* loss, float* grad_unit, lp_stream_t stream);

/* losses/losses.py:528-573 PCALoss + utils/pca.py:97-190,266-309.  index (rows, points) int32 keypoint ids. */
int lp_pca_fwd_bwd(const float* kp, int S, int K, const int* index, int rows, int points, const float* mean,
                   const float* kept_eigenvectors, int ncomp, float epsilon, float* loss, float* grad_unit, lp_stream_t stream);

/* losses/losses.py:880-996 RegressionRMSELoss (always-on diagnostic, models/base.py:528). */
int lp_rmse_fwd(const float* kp_targ, const float* kp_pred, int n_points, float* loss, lp_stream_t stream);
/* LossFactory.__call__ (reference losses/factory.py:229-285) in one launch: weighted[i] = w[i] * x[i][0] (logged as "<stage>_<name>_loss_weighted"),
 * total = sum_i a[i] * weighted[i] in registry order (a = the anneal value for the unsupervised terms, 1 for the heat-map losses).  x: HOST array of
 * n <= 8 DEVICE scalars; w, a: host arrays.  _bwd: gx[i] = w[i] * (g_weighted[i] + a[i] * g_total[0]); either gradient may be NULL. */
int lp_loss_combine(const float* const* x, const float* w, const float* a, int n, float* weighted, float* total, lp_stream_t stream);
int lp_loss_combine_bwd(const float* w, const float* a, int n, const float* g_weighted, const float* g_total, float* gx, lp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Backbone / head contractions on the matrix cores (bf16 NHWC activations, fp32 accumulate).
 * Replace cuDNN behind `self.backbone(images)` (models/base.py:398; models/backbones/factory.py:322-325) and
 * nn.ConvTranspose2d in the head (models/heads/heatmap.py:60-69).  A ConvTranspose2d(k3,s2,p1,op1) forward is
 * lp_conv_dgrad of the mirrored convolution, its data-gradient is lp_conv_fwd, its weight-gradient lp_conv_wgrad.
 * ------------------------------------------------------------------------------------------------------ */
typedef struct lp_conv_geom {
    int B, Hi, Wi, Ci; /* input  tensor, NHWC */
    int Ho, Wo, Co;    /* output tensor, NHWC */
    int R, S, stride, pad;
} lp_conv_geom;

/* Which kernel family the most recent convolution entry point called from this thread launched: conv_igemm_kernel (register-staged,
 * 128-row tiles), conv_pipe_kernel (direct-to-LDS ring, 256-row tiles; LP_CONV_PIPE=0 disables it), or the two weight-gradient kernels.
 * Diagnostic only (bench labels, tests); replaces nothing in the reference. */
#define LP_CONV_KERNEL_IGEMM 0
#define LP_CONV_KERNEL_PIPE 1
#define LP_CONV_KERNEL_WGRAD 2
#define LP_CONV_KERNEL_WGRAD_PIPE 3
#define LP_CONV_KERNEL_PIPE_HALO 4 /* conv_pipe_kernel<..., HALO>: 3x3 / stride 1, the input neighbourhood staged once (LP_CONV_HALO=0 disables) */
/* (6, 7: round 5's producer / consumer kernel conv_spec_kernel - measured not faster, retired in round 6: profiles/retired/r05_conv_spec.h.txt) */
#define LP_CONV_KERNEL_STEM_WGRAD_NB 8 /* stem_wgrad_nb_kernel: the stem's weight gradient from a staged input neighbourhood (LP_STEM_WGRAD_NB=0 disables) */
#define LP_CONV_KERNEL_RES2D 5     /* conv_res2d_kernel: 3x3 / stride 1, 64 -> 64 channels, 16 x 16 tiles, filter resident in LDS (LP_CONV_RES2D=0 disables) */
int lp_conv_last_kernel(void);

/* w: bf16 [Co][R][S][Ci] (Ci % 64 == 0).  Output row-major [B*Ho*Wo][ldo], columns < n_store written. */
int lp_conv_fwd(const void* x, const void* w, const lp_conv_geom* geom, const float* bias, void* out_bf16, float* out_f32, int ldo,
                int n_store, lp_stream_t stream);
/* wd: bf16 [Ci][R][S][Co] (Co % 64 == 0); optional bf16 addend (same layout as dx) is summed in; optional relu_mask (the
 * bf16 activation dx is the gradient of) zeroes dx where the activation is <= 0, i.e. the ReLU backward is fused here.
 * addend may alias dx (in-place accumulation).  A stride-2 gradient runs as 4 parity-class launches; with skip_empty_classes the
 * classes no filter tap reaches (3 of 4 for a 1x1) are not touched - use it when dx already holds addend there. */
int lp_conv_dgrad(const void* dy, const void* wd, const lp_conv_geom* geom, const float* bias, const void* addend,
                  const void* relu_mask, void* dx_bf16, float* dx_f32, int ldo, int n_store, int skip_empty_classes,
                  lp_stream_t stream);
/* The forward kernel as a plain NT GEMM (the token-wise Linear layers and the attention products of the ViT backbone,
 * models/backbones/vit.py:16-49 -> transformers ViTModel):  C[z][m][n] = sum_k A[z][m][k] * B[z][n][k] (+ bias[n]).
 * A: bf16 rows of pitch lda, B: bf16 rows of pitch ldb (K % 64 == 0, pitches % 8 == 0), C: bf16 or fp32 rows of pitch ldc.
 * Columns n < n_store are written (n_store may exceed N up to the tile: the extra columns repeat row N-1 of B); with n_store a
 * multiple of 8 and bf16 output the coalesced store path is taken.  `batch` (optional): nb x nh independent products whose
 * operands start at a_b*zb + a_h*zh (elements; likewise b_*, c_*), e.g. per (image, head) slices of a fused QKV tensor. */
typedef struct lp_gemm_batch {
    int nb, nh;
    long long a_b, a_h, b_b, b_h, c_b, c_h;
} lp_gemm_batch;
int lp_gemm_nt(const void* a, int lda, const void* b, int ldb, void* c_bf16, float* c_f32, int ldc, int M, int N, int K, int n_store,
               const float* bias, const lp_gemm_batch* batch, lp_stream_t stream);
/* out[z][j][n] (bf16, pitch ldo) = sum_m x[z][m][j] * y[z][m][n]: both operands are contracted over their ROW index (the
 * weight-gradient kernel used directly; attention's dV = P^T dO and dK = dS^T Q).  batch: a_* = x, b_* = y, c_* = out strides. */
int lp_gemm_tn(const void* x, int ldx, const void* y, int ldy, void* out_bf16, int ldo, int M, int J, int N, const lp_gemm_batch* batch,
               lp_stream_t stream);
/* Fused soft-max attention forward, head dimension 64 (HF ViTSelfAttention eager attention, reference models/backbones/vit.py:38-43):
 * per (image b, head h):  P = softmax(scale * Q K^T) -> p_bf16[(b*nh + h)*T + q][ldp] (pad columns [T, ldp) zeroed; the backward pass
 * reads it),  O = P V -> out_bf16[(b*T + q)*ldo + h*64 ..].  Token rows qkv_bf16[(b*T + t)*ld_qkv + ..] hold Q at column h*64, K at
 * k_off + h*64 and V at v_off + h*64.  The scores stay on chip (two passes over the keys: running max / exp-sum, then P and O).
 * p_bf16 == NULL (inference): P is used for O and dropped - nothing of size T x T is written. */
int lp_attn_fwd(const void* qkv_bf16, int ld_qkv, int k_off, int v_off, int B, int nh, int T, float scale, void* p_bf16, int ldp,
                void* out_bf16, int ldo, lp_stream_t stream);
/* Attention backward, key / value side, in one pass over the stored probabilities (the autograd of HF ViTSelfAttention's eager
 * attention, reference models/backbones/vit.py:38-43 -> transformers ViTModel; replaces lp_attn_dscores + 2 x lp_gemm_tn):
 *   dS[z][q][k] = scale * P[z][q][k] * (sum_d dO[q][d] V[k][d] - D[q])   -> ds_bf16 (layout / pitch of P, pad columns zeroed; dQ = dS K reads it)
 *   dV[k][d] = sum_q P[q][k] dO[q][d]  -> dqkv_bf16[(b*T + k)*ld_dqkv + dv_off + h*64 + d]
 *   dK[k][d] = sum_q dS[q][k] Q[q][d]  -> dqkv_bf16[(b*T + k)*ld_dqkv + dk_off + h*64 + d]
 * Q / V come from the token rows of lp_attn_fwd, dO from d_out_bf16[(b*T + q)*ld_do + h*64 ..], D = lp_attn_rowdot's [B*T][nh]. */
int lp_attn_bwd_kv(const void* qkv_bf16, int ld_qkv, int v_off, const void* d_out_bf16, int ld_do, const void* p_bf16, int ldp, const float* d_rows,
                   int B, int nh, int T, float scale, void* ds_bf16, void* dqkv_bf16, int ld_dqkv, int dk_off, int dv_off, lp_stream_t stream);
/* Attention backward without materialising dP (replaces lp_gemm_nt + lp_softmax_rows_bwd of the composition; the reference's
 * arithmetic is HF ViTSelfAttention's eager soft-max attention, models/backbones/vit.py:38-43):
 *   lp_attn_rowdot   D[row][h] = sum_d a[row][h*64+d] * b[row][h*64+d]   (a = dO, b = O, head dimension 64)
 *   lp_attn_dscores  dS[z][m][n] = scale * P[z][m][n] * (sum_k dO[z][m][k] V[z][n][k] - D[z][m]), pad columns [N, ldc) zeroed;
 *                    P has the layout of dS; D[z][m] sits at d_rows[zb*d_b + zh*d_h + m*d_row_stride]; batch as in lp_gemm_nt. */
int lp_attn_rowdot(const void* a_bf16, const void* b_bf16, int rows, int nh, int ld, float* out, lp_stream_t stream);
int lp_attn_dscores(const void* d_out, int ld_do, const void* v, int ldv, const void* p_bf16, const float* d_rows, int d_row_stride,
                    long long d_b, long long d_h, float scale, void* ds_bf16, int ldc, int M, int N, int K, const lp_gemm_batch* batch,
                    lp_stream_t stream);
/* A BatchNorm sum in FIXED POINT (round 4): value = hi * 2^-12 + lo * 2^-60.  The reductions over rows that are spread across workgroups -
 * the sums the convolution store passes take, lp_bn_pool_bwd_reduce - split each workgroup's fp32 partial sum t into hi = rint(t 2^12),
 * lo = rint((t - hi 2^-12) 2^60) and add both with 64-bit INTEGER atomics (lp_bn_stats / lp_bn_bwd_reduce: per-workgroup fp32 rows added in
 * a fixed order, then one such addition per channel).  Integer addition commutes, so
 * the totals do not depend on the order in which workgroups arrive: a training step repeats bit for bit (the reference is deterministic on
 * a fixed seed, models/heatmap_tracker.py:69-70; rounds 2 - 3 used fp32 atomics and did not).  |sum| < 2^50; a partial of magnitude
 * >= 1e-10 is represented exactly, smaller ones to 4e-19.  Buffers are accumulated into: zero them first.  SyncBatchNorm all-reduces the
 * integers (SUM is exact).  lp_bn_finalize / lp_bn_bwd_apply / lp_bn_pool_bwd_apply read them. */
typedef struct lp_fxsum {
    long long hi, lo;
} lp_fxsum;
/* The data gradient of a Linear layer that follows a GELU, leaving as the gradient of the GELU's INPUT (a ViT block's fc2 -> fc1's
 * output; HF ViTIntermediate / ViTOutput, reference models/backbones/vit.py:29-49 through transformers' ViTLayer):
 *   c[m][n] = bf16( bf16(sum_k a[m][k] * b[n][k]) * GELU'(u[m][n]) ),   a (M, K), b (N, K), u and c (M, N), all dense bf16;
 * the inner rounding is the one lp_gemm_nt's output would have had, so the result equals lp_gemm_nt followed by lp_gelu_bwd bit for
 * bit, without the activation gradient's write and read.  colsum (optional, (2, N) lp_fxsum, zeroed by the caller): row 0 receives the
 * column sums of c - the bias gradient of the layer that produced u (lp_fxsum_accumulate turns them into fp32); row 1 is scratch.
 * LP_ERR_UNSUPPORTED unless K % 64 == 0 and N % 128 == 0 (the pipelined kernel's shapes): the caller then runs the two-pass form. */
int lp_gemm_nt_gelu_bwd(const void* a, const void* b, const void* u_bf16, void* c_bf16, int M, int N, int K, lp_fxsum* colsum,
                        lp_stream_t stream);
/* ... and the forward side of the same pair: c = bf16(a b^T + bias) as lp_gemm_nt writes it, and act = bf16(GELU(c)) beside it (what
 * lp_gelu_fwd would compute from c, bit for bit): fc1 of a ViT block leaves with its activation.  Same shape limits. */
int lp_gemm_nt_gelu_fwd(const void* a, const void* b, const float* bias, void* c_bf16, void* act_bf16, int M, int N, int K,
                        lp_stream_t stream);
/* BatchNorm reductions fused into the store pass of the convolution next to it, so the normalised tensor is not re-read
 * for them (torch.nn.BatchNorm2d training forward / backward).  Every persistent workgroup adds the
 * column sums of the tiles it walked into `sums` (fixed point, integer atomics: see lp_fxsum).
 *   lp_conv_fwd_bn / lp_stem_fwd_bn:  sums (2,Co) += [sum z, sum z^2] of the bf16 output z  (== lp_bn_stats on it);
 *                                     only sums / seg_images are read.
 *   lp_conv_dgrad_bn:                 dx is the gradient of a = relu(BN(z) [+ residual]); sums (2,Ci) += [sum dx, sum dx*xhat]
 *                                     (== lp_bn_bwd_reduce; the BatchNorm's d beta / d gamma ARE these sums: lp_bn_bwd_apply adds them
 *                                     into the parameter gradients).
 *                                     mask_from_z = 1 recomputes the ReLU mask as bf16(gamma*invstd*(z-mean)+beta) > 0 (layers
 *                                     without a residual branch; relu_mask must then be NULL), otherwise pass relu_mask or
 *                                     bn->relu_bits. */
typedef struct lp_bn_fuse {
    const void* z;        /* bf16 [rows][C] pre-normalisation tensor (dgrad only) */
    const float* mean;    /* (C,) batch mean      (dgrad only) */
    const float* invstd;  /* (C,) 1/sqrt(var+eps) (dgrad only) */
    const float* gamma;   /* (C,) weight, mask_from_z only */
    const float* beta;    /* (C,) bias,   mask_from_z only */
    int mask_from_z;
    const void* relu_bits; /* dgrad only, optional: 1-bit ReLU mask written by lp_bn_apply (then relu_mask must be NULL) */
    lp_fxsum* sums;       /* (2,C) fixed-point sums, accumulated into (zero first) */
    /* Two BatchNorm segments in ONE launch: images [0, seg_images) and [seg_images, B) keep separate batch statistics - the labeled
     * and the unlabeled frames of a semi-supervised step, which the reference normalises in two forward calls (models/base.py:682-695).
     * Then sums is (2 segments, 2, C), mean / invstd are (2, C).  seg_images times the
     * launch's rows per image must be a multiple of 128 (LP_ERR_UNSUPPORTED otherwise: run the segments as two calls).  0 = one segment. */
    int seg_images;
    /* lp_conv_dgrad_bn only (stride 1, relu_bits given): `addend` is the data gradient of the block's stride-2 projection shortcut on ITS grid,
     * [B][ceil(Hi / 2)][ceil(Wi / 2)][Ci] bf16, added at the pixels with even row and column only - the projection shortcut then needs no
     * read-modify-write pass over dx and this launch, the last writer, can take the BatchNorm sums (round 5).  0 = a dense addend. */
    int addend_half;
} lp_bn_fuse;
/* lp_conv_dgrad (bf16 result, no bias) with the ReLU mask read at 1 BIT per element - relu_bits[(row * Ci + c) / 8] bit c % 8, the bytes
 * lp_bn_apply writes beside the activation - instead of from the bf16 activation itself (round 5: the two data gradients into a layer's first
 * block's input read 1/16 of the mask bytes).  Results are identical to lp_conv_dgrad(relu_mask = that activation).
 * Reference: autograd of relu(bn3(z) + identity) feeding conv1 / downsample under models/base.py:398. */
int lp_conv_dgrad_bits(const void* dy, const void* wd, const lp_conv_geom* geom, const void* addend, const void* relu_bits, void* dx_bf16,
                       int skip_empty_classes, lp_stream_t stream);
/* Inference (predict_step, models/heatmap_tracker.py:155-191; eval-mode nn.BatchNorm2d uses its running statistics): the BatchNorm
 * after a convolution is folded into it once per set of weights - lp_bn_fold: w_bf16[co][:] = bf16(w[co][:] * a[co]),
 * bias[co] = beta[co] - running_mean[co] * a[co], a = gamma / sqrt(running_var + eps) - and the layer becomes ONE launch,
 * lp_conv_fwd_act: out = [relu](conv(x, w_bf16) + bias + residual), residual = the block's identity / shortcut (bf16, layout of out)
 * or NULL.  No BatchNorm pass, no pre-normalisation tensor, nothing kept for a backward pass.  Runs on conv_pipe_kernel<.., kEkInfer> where
 * the training forward would (LP_INFER_PIPE=0: conv_igemm_kernel<infer>); there the residual and the ReLU act on bf16(accumulator + bias). */
int lp_bn_fold(const float* w, const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
               int Co, int per_co, void* w_bf16, float* bias, lp_stream_t stream);
int lp_conv_fwd_act(const void* x, const void* w, const lp_conv_geom* geom, const float* bias, const void* residual_bf16, int relu,
                    void* out_bf16, lp_stream_t stream);
int lp_conv_fwd_bn(const void* x, const void* w, const lp_conv_geom* geom, void* out_bf16, const lp_bn_fuse* bn, lp_stream_t stream);
int lp_stem_fwd_bn(const void* x4, const void* w, const lp_conv_geom* geom, void* out_bf16, const lp_bn_fuse* bn, lp_stream_t stream);
int lp_conv_dgrad_bn(const void* dy, const void* wd, const lp_conv_geom* geom, const void* addend, const void* relu_mask,
                     void* dx_bf16, const lp_bn_fuse* bn, lp_stream_t stream);
/* dw: fp32 [Co][R][S][Ci], accumulated into (zero it first); split_hint <= 0 picks the pixel split.  The pixel slices leave
 * partial tiles in `workspace` (lp_conv_wgrad_workspace_bytes) and a second kernel sums them in a fixed order: deterministic. */
size_t lp_conv_wgrad_workspace_bytes(const lp_conv_geom* geom, int split_hint);
int lp_conv_wgrad(const void* x, const void* dy, const lp_conv_geom* geom, float* dw, int split_hint, void* workspace,
                  size_t workspace_bytes, lp_stream_t stream);
/* same, plus dbias[co] += sum_m dy[m][co] (bias gradient of a Linear layer of the ViT backbone, reference models/backbones/vit.py:16-49,
 * or of a ConvTranspose2d of the head, models/heads/heatmap.py:20-71) out of the same pass over dy: fp32
 * atomics, one per column and pixel slice, so the summation order (not the set of addends) can vary between runs */
int lp_conv_wgrad_bias(const void* x, const void* dy, const lp_conv_geom* geom, float* dw, float* dbias, int split_hint, void* workspace,
                       size_t workspace_bytes, lp_stream_t stream);
/* ResNet stem 7x7/2: x4 = NHWC4 bf16 (channel 3 zero), weights / gradients in the padded [64][8][8][4] layout. */
int lp_stem_fwd(const void* x4, const void* w, const lp_conv_geom* geom, void* out_bf16, lp_stream_t stream);
int lp_stem_wgrad(const void* x4, const void* dy, const lp_conv_geom* geom, float* dw, int split_hint, void* workspace,
                  size_t workspace_bytes, lp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * fp32 VALIDATION path (csrc/fp32.hip): the same layers in fp32 end to end on v_mfma_f32_32x32x2_f32, so a whole training step can
 * be held against the reference - which trains in fp32 only (lightning_pose/train.py:411-428 passes no precision) - at the 1e-4
 * tolerance BASELINE.json's north_star states for fp32.  Untuned (scalar operand loads, one wave per 32 x 32 tile): it exists to
 * prove the wiring, the bf16-mixed entry points above are the product and the measured path.
 * Tensors are dense NHWC fp32; weights are read straight from the flat fp32 parameter buffer in its [Co][KH][KW][CiS] storage
 * (KH, KW, CiS >= R, S, Ci: the stem is stored [64][8][8][4]); dgrad needs no transposed copy.  `addend` (may alias the output) is
 * added to the result; lp_f32_conv_wgrad accumulates into dw with fp32 atomics.
 * ------------------------------------------------------------------------------------------------------ */
int lp_f32_conv_fwd(const float* x, const float* w, const lp_conv_geom* geom, int KH, int KW, int CiS, const float* bias,
                    const float* addend, float* out, lp_stream_t stream);
int lp_f32_conv_dgrad(const float* dy, const float* w, const lp_conv_geom* geom, int KH, int KW, int CiS, const float* bias,
                      const float* addend, float* dx, lp_stream_t stream);
int lp_f32_conv_wgrad(const float* x, const float* dy, const lp_conv_geom* geom, int KH, int KW, int CiS, float* dw, lp_stream_t stream);
/* fp32 forms of lp_bn_stats / lp_bn_apply / lp_bn_bwd_reduce / lp_bn_bwd_apply (lp_bn_finalize is shared), of the 3x3/2 max-pool, of the
 * input layout conversion (NCHW -> NHWC4), of PixelShuffle(2) and of the spatial soft-max backward (fp32 gradient out) */
int lp_f32_bn_stats(const float* x, int M, int C, float* sums, lp_stream_t stream);
/* the same sums through per-stripe partials in `workspace`, added in a fixed order (no atomics): the validation executor's forward statistics,
 * so that its outputs repeat bit for bit from run to run */
size_t lp_f32_bn_stats_workspace_bytes(int M, int C);
int lp_f32_bn_stats_ordered(const float* x, int M, int C, float* sums, void* workspace, size_t workspace_bytes, lp_stream_t stream);
int lp_f32_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const float* residual,
                    int relu, int M, int C, float* y, lp_stream_t stream);
int lp_f32_bn_bwd_reduce(const float* dy, const float* y_out, const float* x, const float* mean, const float* invstd, int M, int C,
                         float* sums, float* dbeta_acc, float* dgamma_acc, lp_stream_t stream);
int lp_f32_bn_bwd_apply(const float* dy, const float* y_out, const float* x, const float* mean, const float* invstd, const float* gamma,
                        const float* sums, float count, int M, int C, float* dx, float* dres, lp_stream_t stream);
int lp_f32_maxpool_fwd(const float* x, int B, int Hi, int Wi, int C, float* y, void* argmax_u8, lp_stream_t stream);
int lp_f32_maxpool_bwd(const void* argmax_u8, const float* dy, int B, int Hi, int Wi, int C, float* dx, lp_stream_t stream);
int lp_f32_images_to_nhwc4(const float* images, int B, int H, int W, float* out, lp_stream_t stream);
int lp_f32_pixel_shuffle(const float* in, int B, int h, int w, int c_out, int ld, int inverse, float* out, lp_stream_t stream);
int lp_f32_softmax2d_bwd(const float* prob, const float* gprob, int B, int K, int n, float* gin, long stride_b, long stride_i, long stride_k,
                         lp_stream_t stream);
/* fp32 forms of the ViT-S/16 glue (csrc/vit_f32.hip) - patch rows, token assembly, LayerNorm (+ residual add in front, [CLS] rows dropped
 * with drop_T), exact GELU - and of the attention over the fused qkv tensor (head dimension 64; p [B][nh][T][T] keeps the probabilities,
 * ds_workspace the same size): config C4 (lightning_pose/models/backbones/vit.py:16-49 over HuggingFace ViTModel) at the reference's own
 * precision.  The Linear layers of that path are lp_f32_conv_fwd / _dgrad / _wgrad with a 1x1 geometry. */
int lp_f32_vit_patchify(const float* images, int B, int H, int W, int patch, float* out, lp_stream_t stream);
int lp_f32_vit_tokens_fwd(const float* patch, const float* cls, const float* pos, int B, int Np, int D, float* x, lp_stream_t stream);
int lp_f32_vit_tokens_bwd(const float* dx, int B, int Np, int D, float* dpatch, float* dpos, lp_stream_t stream);
int lp_f32_layernorm_fwd(const float* x, const float* delta, float* x_out, const float* gamma, const float* beta, float eps, int M, int D,
                         int drop_T, float* y, float* mean, float* rstd, lp_stream_t stream);
int lp_f32_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, int M, int D, int drop_T,
                         float* dx_acc, float* dgamma_acc, float* dbeta_acc, lp_stream_t stream);
int lp_f32_gelu_fwd(const float* x, size_t n, float* y, lp_stream_t stream);
int lp_f32_gelu_bwd(const float* x, const float* dy, size_t n, float* dx, lp_stream_t stream);
int lp_f32_attn_fwd(const float* qkv, int ld, int k_off, int v_off, int B, int nh, int T, float scale, float* p, float* o, int ldo,
                    lp_stream_t stream);
int lp_f32_attn_bwd(const float* qkv, int ld, int k_off, int v_off, const float* dout, int ldo, const float* p, int B, int nh, int T,
                    float scale, float* ds_workspace, float* dqkv, int ldd, lp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * HBM-bound glue of the trunk (NHWC bf16): BatchNorm2d training mode, ReLU, residual add, MaxPool2d(3,2,1),
 * PixelShuffle(2), input layout.  torchvision Bottleneck semantics; called from
 * `self.backbone(images)` (models/base.py:398) and HeatmapHead.forward (models/heads/heatmap.py:44,208).
 * ------------------------------------------------------------------------------------------------------ */
/* sums (2,C) fixed point (lp_fxsum), accumulated into (zero first): [sum x, sum x^2] over the M rows.  SyncBatchNorm = all-reduce it
 * (int64 SUM).  `workspace` (lp_bn_reduce_workspace_bytes; also for lp_bn_bwd_reduce): one fp32 row of partial sums per workgroup, added in
 * a fixed order by a second launch - up to 1024 workgroups finish together here, too many for atomics on one address each. */
size_t lp_bn_reduce_workspace_bytes(int M, int C);
int lp_bn_stats(const void* x, int M, int C, lp_fxsum* sums, void* workspace, size_t workspace_bytes, lp_stream_t stream);
/* dst[i] += value of sums[i], i < n (fixed-point totals as fp32) */
int lp_fxsum_accumulate(const lp_fxsum* sums, int n, float* dst, lp_stream_t stream);
int lp_bn_finalize(const lp_fxsum* sums, float count, int C, float eps, float momentum, float* mean, float* invstd,
                   float* running_mean, float* running_var, lp_stream_t stream);
/* the same from plain fp32 [sum, sum of squares] (the fp32 validation executor: lp_f32_bn_stats_ordered) */
int lp_bn_finalize_f32(const float* sums, float count, int C, float eps, float momentum, float* mean, float* invstd,
                       float* running_mean, float* running_var, lp_stream_t stream);
/* two segments at once: sums (2,2,C), mean / invstd (2,C); the running statistics take segment 0's update, then segment 1's - the
 * order of the reference's two forward calls (labeled, then unlabeled: models/base.py:682-695) */
int lp_bn_finalize2(const lp_fxsum* sums, float count0, float count1, int C, float eps, float momentum, float* mean, float* invstd,
                    float* running_mean, float* running_var, lp_stream_t stream);
/* y = [relu]((x - mean) * invstd * gamma + beta [+ residual]); relu_bits (optional, M*C/8 bytes): bit q of byte i = (y[8*i + q] > 0),
 * a 16x smaller ReLU mask for the backward pass */
int lp_bn_apply(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const void* residual,
                int relu, int M, int C, void* y, void* relu_bits, lp_stream_t stream);
/* both BatchNorm segments of a joint pass (lp_bn_fuse.seg_images) in ONE launch: rows [0, seg_rows) use row 0 of mean / invstd (2, C)
 * [and of sums (2, 2, C), with count0], the other rows use row 1 [count1] */
int lp_bn_apply_seg(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const void* residual,
                    int relu, int M, int C, int seg_rows, void* y, void* relu_bits, lp_stream_t stream);
/* The block-output pass of the optional "fp32 residual stream" policy (round 6; LP_RESIDUAL_FP32=1 in the engine - not the benchmarked default):
 * a block output o is kept as a bf16 pair, y = bf16(o) (what the convolutions read) and y_lo = bf16(o - y) (may be NULL: nobody adds this
 * output as an identity shortcut), and the residual added here is `residual` + `residual_lo` (the previous block's pair; residual_lo may be
 * NULL) or - `zd` given, residual / residual_lo NULL - the projection shortcut zd normalised in this pass and added unrounded.  Otherwise as
 * lp_bn_apply_seg / lp_bn_apply_seg_rbn.  (The reference keeps fp32 activations throughout: train.py:411-428.) */
int lp_bn_apply_seg_lo(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const void* residual,
                       const void* residual_lo, const void* zd, const float* mean_d, const float* invstd_d, const float* gamma_d,
                       const float* beta_d, int relu, int M, int C, int seg_rows, void* y, void* y_lo, void* relu_bits, lp_stream_t stream);
/* lp_bn_apply_seg whose residual is a PRE-normalisation tensor with its own BatchNorm (a block's projection shortcut, round 5): the shortcut is
 * normalised in the same pass - rounded to bf16 as its own lp_bn_apply would have stored it - instead of being written and read back.
 * Bit-identical to lp_bn_apply_seg(zd -> idt, no ReLU) followed by lp_bn_apply_seg(x, ..., residual = idt).
 * Reference: torchvision Bottleneck.forward (out = relu(bn3(conv3(.)) + downsample(x))) under models/base.py:398. */
int lp_bn_apply_seg_rbn(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const void* zd,
                        const float* mean_d, const float* invstd_d, const float* gamma_d, const float* beta_d, int relu, int M, int C,
                        int seg_rows, void* y, void* relu_bits, lp_stream_t stream);
/* sums (2,C) += [sum dz, sum dz*xhat], dz = dy masked by relu'(y_out) (y_out may be NULL) */
int lp_bn_bwd_reduce(const void* dy, const void* y_out, const void* x, const float* mean, const float* invstd, int M, int C,
                     lp_fxsum* sums, void* workspace, size_t workspace_bytes, lp_stream_t stream);
/* dx = gamma * invstd * (dz - sum(dz)/N - xhat * sum(dz*xhat)/N) [, dres = dz].  `sums`: the totals the two correction terms use - the
 * buffer after the SyncBatchNorm all-reduce, with count = rows x world size - or NULL: no batch-statistics terms (eval-mode BatchNorm is a
 * fixed affine map).  `sums_local` (this rank's sums, before any exchange; may equal sums) + dbeta_acc / dgamma_acc (optional): the
 * BatchNorm's parameter gradients d beta += sum dz, d gamma += sum dz*xhat over the segments, added by one thread per channel.
 * `terms_ws` (round 5): caller-owned scratch of segments x 2 x C floats, or NULL.  With it the call is two launches - a one-thread-per-value
 * conversion of the fixed-point sums to sum / count (+ the parameter gradients), then the streaming kernel reading plain floats; without it
 * one self-contained launch that converts the table into LDS per workgroup (C <= 2048 only; ~10 us slower per launch). */
int lp_bn_bwd_apply(const void* dy, const void* y_out, const void* x, const float* mean, const float* invstd, const float* gamma,
                    const lp_fxsum* sums, float count, int M, int C, void* dx, void* dres, const lp_fxsum* sums_local, float* dbeta_acc,
                    float* dgamma_acc, float* terms_ws, lp_stream_t stream);
int lp_bn_bwd_apply_seg(const void* dy, const void* y_out, const void* x, const float* mean, const float* invstd, const float* gamma,
                        const lp_fxsum* sums, float count0, float count1, int M, int C, int seg_rows, void* dx, void* dres,
                        const lp_fxsum* sums_local, float* dbeta_acc, float* dgamma_acc, float* terms_ws, lp_stream_t stream);
/* lp_bn_bwd_apply_seg for a block output whose masked gradient ALSO feeds the block's projection-shortcut BatchNorm (round 5): the same walk reads
 * that BatchNorm's pre-normalisation tensor zd and leaves its two backward reductions [sum dz, sum dz * xhat_d] per segment in sums_d
 * ([segments][2][C], added into) - what lp_bn_bwd_reduce(dres, NULL, zd, mean_d, invstd_d, ...) computes from one more pass over the gradient.
 * terms_ws is required; C / 8 must divide 256 (LP_ERR_UNSUPPORTED otherwise); workspace: lp_bn_bwd_ds_workspace_bytes(M, C).
 * Reference: autograd of bn3(z3) + downsample-bn(zd) -> relu under models/base.py:398 (torchvision Bottleneck.forward). */
size_t lp_bn_bwd_ds_workspace_bytes(int M, int C);
int lp_bn_bwd_apply_seg_ds(const void* dy, const void* y_out, const void* x, const float* mean, const float* invstd, const float* gamma,
                           const lp_fxsum* sums, float count0, float count1, int M, int C, int seg_rows, void* dx, void* dres,
                           const lp_fxsum* sums_local, float* dbeta_acc, float* dgamma_acc, float* terms_ws, const void* zd,
                           const float* mean_d, const float* invstd_d, lp_fxsum* sums_d, void* workspace, size_t workspace_bytes,
                           lp_stream_t stream);
/* 3x3 / stride 2 / pad 1; argmax_u8 (B,Ho,Wo,C) records the winning tap (first maximum, ATen tie rule) for the backward gather */
int lp_maxpool_fwd(const void* x, int B, int Hi, int Wi, int C, void* y, void* argmax_u8, lp_stream_t stream);
int lp_maxpool_bwd(const void* argmax_u8, const void* dy, int B, int Hi, int Wi, int C, void* dx, lp_stream_t stream);
/* Stem: BatchNorm + ReLU + MaxPool2d(3, 2, 1) without ever storing the activation between them (torchvision resnet50 children
 * bn1, relu, maxpool; reference models/backbones/factory.py:322-348).  Forward = lp_bn_apply(relu) -> lp_maxpool_fwd, bit for bit
 * (each tap is rounded to bf16 as lp_bn_apply would have stored it).  Backward: the activation's gradient is rebuilt on the fly from
 * the pooled gradient dy, the arg-max bytes and z (ReLU gate recomputed from z): lp_bn_pool_bwd_reduce leaves [sum g, sum g * xhat]
 * in sums[2][C] (fixed point), lp_bn_pool_bwd_apply writes d z (and adds the sums into d beta / d gamma, as lp_bn_bwd_apply does). */
int lp_bn_relu_maxpool_fwd(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta, int B, int Hi, int Wi,
                           int C, void* y, void* argmax_u8, lp_stream_t stream);
int lp_bn_pool_bwd_reduce(const void* argmax_u8, const void* dy, const void* z, const float* mean, const float* invstd, const float* gamma,
                          const float* beta, int B, int Hi, int Wi, int C, lp_fxsum* sums, lp_stream_t stream);
int lp_bn_pool_bwd_apply(const void* argmax_u8, const void* dy, const void* z, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, const lp_fxsum* sums, float count, int B, int Hi, int Wi, int C, void* dx,
                         const lp_fxsum* sums_local, float* dbeta_acc, float* dgamma_acc, lp_stream_t stream);
int lp_images_to_nhwc4(const float* images_nchw, int B, int H, int W, void* out_bf16, lp_stream_t stream);
/* (B,h,w,4*c_out) -> (B,2h,2w,c_out) stored with channel pitch ld >= c_out (pad channels untouched); inverse = 1 maps the
 * gradient (pitch ld) back */
int lp_pixel_shuffle(const void* in, int B, int h, int w, int c_out, int ld, int inverse, void* out, lp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * ViT-S/16 backbone glue (models/backbones/vit.py:16-49 -> transformers ViTModel; backbone = "vits_dino",
 * models/backbones/factory.py:188-190).  The Linear layers and attention products run on lp_gemm_nt / lp_conv_wgrad.
 * Residual stream and statistics fp32, GEMM operands bf16.
 * ------------------------------------------------------------------------------------------------------ */
/* (B,3,H,W) fp32 -> [B*(H/P)*(W/P)][3*P*P] bf16 patch rows, k = (c, ky, kx) (= Conv2d weight.flatten(1)) */
int lp_vit_patchify(const float* images_nchw, int B, int H, int W, int patch, void* out_bf16, lp_stream_t stream);
/* x[b][0] = cls + pos[0], x[b][1+p] = patch[b][p] + pos[1+p]  (pos already interpolated, (1+Np, D) fp32) */
int lp_vit_tokens_fwd(const void* patch_bf16, const float* cls, const float* pos, int B, int Np, int D, float* x, lp_stream_t stream);
/* dpatch = bf16(dx[:, 1:]);  dpos[t] = sum_b dx[b][t]  (d cls = dpos[0]) */
int lp_vit_tokens_bwd(const float* dx, int B, int Np, int D, void* dpatch_bf16, float* dpos, lp_stream_t stream);
/* y (R,D) (+)= w (R,Q) @ x (Q,D), or with transpose_w: y (Q,D) (+)= w^T @ x (R,D): bicubic position-embedding interpolation */
int lp_small_matmul(const float* w, const float* x, int R, int Q, int D, int transpose_w, int accumulate, float* y, lp_stream_t stream);
/* x_out = x (+ delta_bf16);  y = LayerNorm(x_out) in bf16;  drop_T > 0: rows with row % drop_T == 0 ([CLS]) are dropped from y and
 * the rest compacted (the (B, h, w, D) feature map).  x_out may alias x; mean / rstd (M,) are kept for the backward pass. */
int lp_layernorm_fwd(const float* x, const void* delta_bf16, float* x_out, const float* gamma, const float* beta, float eps, int M,
                     int D, int drop_T, void* y_bf16, float* mean, float* rstd, lp_stream_t stream);
/* dx_acc += LayerNorm backward of dy (bf16, same row mapping as y);  dgamma_acc / dbeta_acc accumulate too */
int lp_layernorm_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd, const float* gamma, int M, int D,
                     int drop_T, float* dx_acc, float* dgamma_acc, float* dbeta_acc, lp_stream_t stream);
/* same, and the updated dx_acc is also written rounded to bf16 (what the next Linear layer's backward reads; the LayerNorms are
 * those of transformers ViTLayer / ViTModel.layernorm behind reference models/backbones/vit.py:26-27) */
int lp_layernorm_bwd_bf16(const void* dy_bf16, const float* x, const float* mean, const float* rstd, const float* gamma, int M, int D,
                          int drop_T, float* dx_acc, void* dx_bf16, float* dgamma_acc, float* dbeta_acc, lp_stream_t stream);
int lp_gelu_fwd(const void* x_bf16, size_t n, void* y_bf16, lp_stream_t stream);                       /* exact (erf) GELU */
int lp_gelu_bwd(const void* x_bf16, const void* dy_bf16, size_t n, void* dx_bf16, lp_stream_t stream);
/* The two producers of a Linear layer's dy in the ViT backward, leaving that layer's BIAS gradient (the column sums of the bf16 tensor they
 * write, accumulated into colsum_acc) on the way: the weight gradient then needs no bias pass (lp_conv_wgrad instead of lp_conv_wgrad_bias). */
int lp_gelu_bwd_colsum(const void* x_bf16, const void* dy_bf16, int rows, int cols, void* dx_bf16, float* colsum_acc, lp_stream_t stream);
int lp_layernorm_bwd_bf16_colsum(const void* dy_bf16, const float* x, const float* mean, const float* rstd, const float* gamma, int M, int D,
                                 int drop_T, float* dx_acc, void* dx_bf16, float* dgamma_acc, float* dbeta_acc, float* colsum_acc,
                                 lp_stream_t stream);
/* in place: s[r][:n] = softmax(scale * s[r][:n]), s[r][n:ld] = 0   /   dp <- scale * p * (dp - sum(dp * p)) */
int lp_softmax_rows_fwd(void* s_bf16, int rows, int n, int ld, float scale, lp_stream_t stream);
int lp_softmax_rows_bwd(const void* p_bf16, void* dp_bf16, int rows, int n, int ld, float scale, lp_stream_t stream);
/* out[z][c][r] = in[z][r][c] (r < R, c < Cc), out[z][c][R:ldo] = 0;  z = (zb, zh) with element strides */
int lp_transpose_batched(const void* in_bf16, int R, int Cc, int ldi, long long in_b, long long in_h, void* out_bf16, int ldo,
                         long long out_b, long long out_h, int nb, int nh, lp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Batch producers: what sits directly BEFORE the step.
 *   lp_frames_resize / lp_frames_augment   the device half of data/video/dali.py:135-192 video_pipe after the decoder:
 *        fn.resize -> [fn.warp_affine(matrix, fill_value=0, inverse_map=False) -> fn.brightness_contrast -> fn.noise.shot]
 *        -> /255 -> fn.crop_mirror_normalize(mean, std, output_layout="FCHW"); the (2,3) matrix is the `transforms` entry of
 *        UnlabeledBatchDict (data/video/dali.py:267-330) that lp_decode_fwd's frame map undoes.  DALI is not vendored:
 *        its published operator definitions are restated (oracle/restated.py), parity unpinned.
 *   lp_labeled_keypoints                   data/datasets.py:262-376 (imgaug Resize keypoint projection, hflip + left/right
 *        swap), :465-472 (visibility synthesised from NaN labels), :496-508 (out-of-frame keypoints become NaN); feed the
 *        result to lp_heatmap_gen for HeatmapDataset.compute_heatmap's targets.
 * ------------------------------------------------------------------------------------------------------ */
enum { LP_BORDER_RENORM = 0, /* filter window cut at the image edge and renormalised (PIL / torch antialias)       */
       LP_BORDER_CLAMP = 1   /* edge pixels replicated under the full window (DALI resampling)                     */ };

typedef struct lp_frame_norm {
    float mean[3], std[3]; /* per channel, in [0,1] units (ImageNet statistics in the reference, data/__init__.py:46-47) */
} lp_frame_norm;

typedef struct lp_frame_augment {
    int has_matrix;
    float matrix[6];        /* row-major (2,3), maps SOURCE to DESTINATION pixel-centre coordinates (inverse_map=False)   */
    float brightness, contrast, contrast_center; /* out = brightness * (center + contrast * (in - center)); DALI: 0.5 for float input */
    float shot_factor;      /* out = Poisson(max(in,0) / factor) * factor on the [0,255] scale; 0 = off                   */
    unsigned long long seed; /* counter-based generator: the noise depends only on (seed, frame, pixel, channel)         */
} lp_frame_augment;

/* src u8 (S, Hs, Ws, 3) with byte strides -> antialiased linear resize to (H, W).  finish_norm == NULL: dst = fp32 (S,H,W,3) in
 * [0,255] (input of lp_frames_augment); else dst = fp32 (S,3,H,W) = (v/255 - mean) / std (imgaug="default": no augmentation) */
int lp_frames_resize(const void* src_u8, int S, int Hs, int Ws, long long frame_stride, int row_stride, int H, int W, int border,
                     const lp_frame_norm* finish_norm, float* dst, lp_stream_t stream);
/* Same layouts, bicubic interpolation without antialiasing (Keys kernel, A = -0.75, half-pixel centres, taps clamped): imgaug
 * iaa.Resize's default (OpenCV INTER_CUBIC), the last imgaug step of every LABELED image (data/datasets.py:137-143).  round_u8 != 0:
 * the interpolated value is rounded and saturated to [0, 255] first - imgaug returns uint8 images. */
int lp_frames_resize_cubic(const void* src_u8, int S, int Hs, int Ws, long long frame_stride, int row_stride, int H, int W, int round_u8,
                           const lp_frame_norm* finish_norm, float* dst, lp_stream_t stream);
/* src fp32 (S,H,W,3) in [0,255] -> dst fp32 (S,3,H,W); one parameter set per call = per DALI sample (a whole sequence) */
int lp_frames_augment(const float* src_hwc, int S, int H, int W, const lp_frame_augment* aug, const lp_frame_norm* norm,
                      float* dst_nchw, lp_stream_t stream);
/* kp_src (B,K,2) source px, src_hw (B,2) = source (height, width); optional affine (B,2,3) on source px, hflip (B) 0/1 with the
 * keypoint permutation swap (K), vis_in (B,K) 0/1/2 (NULL: synthesised from NaN labels, NaN -> uniform_heatmaps ? 1 : 0, else 2).
 * kp_out (B,K,2) model px with out-of-frame points set to NaN, vis_out (B,K).  kp_out must not alias kp_src. */
int lp_labeled_keypoints(const float* kp_src, const float* src_hw, const float* affine, const int* hflip, const int* swap,
                         const int* vis_in, int uniform_heatmaps, int B, int K, int H, int W, float* kp_out, int* vis_out,
                         lp_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Optimiser: torch.optim.Adam / AdamW semantics (models/base.py:458-479) over one flat fp32 range, also emitting
 * the bf16 copy the GEMMs read.  lr may be 0 (frozen backbone, callbacks.py:79-196): moments still move.
 * ------------------------------------------------------------------------------------------------------ */
int lp_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int decoupled, int step, float grad_scale, void* params_bf16, lp_stream_t stream);
int lp_cast_bf16(const float* src, size_t n, void* dst, lp_stream_t stream);
/* dst[c][b][a] = src[a][b][c] on bf16: weights [Co][R*S][Ci] -> data-gradient copy [Ci][R*S][Co] */
int lp_permute_cba(const void* src, int A, int B, int C, void* dst, lp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LP_HIP_H */
