/*
 * ssdhip.h -- C ABI of libssdhip.so, the MI355X (gfx950) implementation of the SSD300 hot path of
 * rs1004/object-detection-torch2.
 *
 * The reference has no FFI layer: its boundary is the Python surface `model.SSD` + `utils.*`
 * (SURVEY.md 8(b)).  Each entry point below replaces the stock-ATen work behind one reference
 * call site (cited per function, paths relative to the reference's src/).  The Python host layer
 * (object_detection_torch2_amd/) mirrors the reference surface and binds these with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed for the call (caller-owned, e.g. torch storage);
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous and stream-ordered;
 *   - no hidden allocation, no host sync, no global mutable state: workspaces are caller-supplied,
 *     so every call can be captured into a hipGraph;
 *   - return 0 on success, a negative SSDHIP_E* code on bad arguments / launch failure
 *     (the Python side turns it into ValueError / RuntimeError, the reference's error style);
 *   - activations are NHWC fp32 with C a multiple of 32 (the 3-channel image is padded to 4);
 *     NCHW exists only at the API edge (input image, OIHW state_dict weights).
 */
#ifndef SSDHIP_H
#define SSDHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSDHIP_OK 0
#define SSDHIP_EINVAL (-1)   /* bad shape / argument            */
#define SSDHIP_ELAUNCH (-2)  /* hipLaunch / runtime error       */
#define SSDHIP_ENOSUP (-3)   /* configuration not supported     */

/* library / device probe (no reference equivalent) */
int ssdhip_version(void);
const char* ssdhip_last_error(void);
/* compute units of the current device (256 on an MI355X in SPX mode); persistent grids and split-K targets are derived from it */
int ssdhip_cu_count(void);

/* ---- A9  SSD._get_default_bboxes (model/ssd.py:108-133) ------------------------------------------------
 * out: (P,4) fp32 rows [cx,cy,w,h]; HOST table maps[k] = {m, n, anchors}; computed in fp64, rounded once.
 * Returns P (>0) on success.                                                                              */
int ssdhip_default_boxes(float* out, const int32_t* maps_host, int n_maps, double s_min, double s_max, void* stream);

/* ---- A2  VGG16.normalize (model/vgg16.py:103-115) fused with NCHW -> NHWC4 -----------------------------
 * x: (N,3,H,W) -> y: (N,H,W,4), y[...,c] = (x-mean[c])/std[c], y[...,3] = 0.                               */
int ssdhip_normalize_nchw_to_nhwc4(const float* x, float* y, int N, int H, int W, void* stream);

/* ---- weights: OIHW state_dict tensor -> packed implicit-GEMM operand ------------------------------------
 * mode 0 (fwd):   packed[co][k], k = ((ci/32)*R*S + r*S+s)*32 + ci%32  (k = (r*S+s)*4 + ci for the 4-channel padded image);
 *                 rows co < rows_pad, zero-filled beyond Cout / Cin / taps, row length KP
 * mode 1 (dgrad): packed[ci][k], k = ((co/32)*R*S + r*S+s)*32 + co%32, rows ci < rows_pad, row length KP
 * (model/ssd.py:61,70-77 nn.Conv2d weights; repacked after each optimizer step for trainable layers)     */
int ssdhip_pack_conv_weight(const float* w_oihw, float* packed, int Cout, int Cin, int R, int S,
                            int rows_pad, int inner_pad, int KP, int mode, void* stream);
/* the same for many tensors in one launch (an optimizer step changes every trainable weight: ~26 layouts): `jobs_device` = n_jobs records in
 * DEVICE memory, each the argument list of one ssdhip_pack_conv_weight call (validated by the caller); max_elements = the largest
 * rows_pad * KP among them */
typedef struct {
    const float* w;
    float* packed;
    int32_t Cout, Cin, R, S, rows_pad, inner_pad, KP, mode;
} ssdhip_pack_job;
int ssdhip_pack_conv_weights(const ssdhip_pack_job* jobs_device, int n_jobs, int64_t max_elements, void* stream);

/* ---- A3/A7  nn.Conv2d forward as NHWC implicit GEMM on v_mfma_f32_32x32x2_f32 ---------------------------
 * (model/vgg16.py:33, model/ssd.py:61,70-77,103-104).  Also used for dgrad (mode 1) with mode-1 packed weights.
 * Optional fused input transform: a = relu(x*in_scale[c] + in_shift[c]) applied while staging (train-mode
 * BatchNorm2d + ReLU of the producer layer, model/vgg16.py:34-35), zero padding applied after it.
 * Optional epilogue: + bias[co]; per-(m-block, channel) partial sum / sum-of-squares for BatchNorm statistics. */
typedef struct {
    const float* x;         /* input activations (N,Hi,Wi,Cin)                                   */
    const float* w;         /* packed weights (see ssdhip_pack_conv_weight)                      */
    const float* bias;      /* (Cout) or NULL                                                    */
    const float* in_scale;  /* (Cin) or NULL: fused BN+ReLU of the input                         */
    const float* in_shift;  /* (Cin)                                                             */
    float* y;               /* output base; element (n,ho,wo,co) at y[n*y_batch_stride + (ho*Wo+wo)*Cout + co] */
    float* stats;           /* NULL or (ceil(M/BM), Cout, 2) partial [sum, sumsq] per m-block     */
    int64_t y_batch_stride; /* floats between images in y (Ho*Wo*Cout for dense NHWC)            */
    int32_t N, Hi, Wi, Cin; /* Cin: 4 (padded RGB image) or a multiple of 32                     */
    int32_t Ho, Wo, Cout;
    int32_t R, S, stride, pad;
    int32_t KP;             /* packed row length (multiple of 32)                                */
    int32_t mode;           /* 0 fwd gather, 1 dgrad gather (x = dy, roles of Hi/Ho swapped by caller) */
    int32_t accumulate;     /* 1: y += result (second dgrad contribution)                        */
    int32_t tile;           /* 0 auto, else BM*1000+BN (128128, 128064, 064128, 064064); nothing else */
    int32_t rows_per_img;   /* 0, or batched-GEMM use: output rows per weight plane (multiple of 128) */
    int64_t w_img_stride;   /* floats between consecutive weight planes (with rows_per_img)      */
    int32_t compute;        /* 0: exact fp32 MFMA (default, the headline path); 1: operands rounded to bf16 at LDS staging,
                               v_mfma_f32_32x32x16_bf16 with fp32 accumulate (BASELINE configs[3]); tensors stay fp32 in HBM;
                               2 (opt-in, 128 x 128 plane GEMMs only): every fp32 operand as three bf16 terms, the six products down to
                               2^-24 on the bf16 MFMA with fp32 accumulate -- error vs fp64 no larger than mode 0, 1.45x its speed */
    float* ws;              /* NULL, or split-K workspace of ws_bytes bytes (ssdhip_conv2d_ws_bytes): launches whose tile grid cannot
                               fill the chip (tiny maps) split K over blocks and sum the partial tiles in a second, fixed-order pass */
    int64_t ws_bytes;
    int32_t max_cus;        /* 0 = all; > 0: the persistent work-queue plane GEMM fills at most this many CUs (two blocks each) and leaves the
                               rest to kernels of other streams (a full grid holds every register of every CU for the whole launch) */
} ssdhip_conv_params;
int ssdhip_conv2d_nhwc(const ssdhip_conv_params* p, void* stream);
int64_t ssdhip_conv2d_ws_bytes(const ssdhip_conv_params* p);   /* split-K workspace this launch can use (0 = it does not split) */
/* For a plane GEMM (rows_per_img != 0) with tile = 128128 and K >= 256 whose tile count fits the resident block slots badly, the same query returns
 * the workspace of the persistent work-queue kernel (csrc/gemm_planes.hip: per-XCD atomic queues, last round split along K, fixed-order reduce:
 * bit-reproducible): 4 KiB of control words -- ZERO before the first launch on this workspace, left zero by every launch -- plus one 128 x 128
 * partial tile per block slot.  Launches that share a workspace must be stream-ordered.  Without it the launch runs one block per tile slot. */
/* m-block size the auto/explicit tile choice will use for `stats` (rows of the partial table = ceil(M/BM)) */
int ssdhip_conv2d_block_m(const ssdhip_conv_params* p);
/* BM*1000+BN the call will run with (profiling / bench bookkeeping) */
int ssdhip_conv2d_tile(const ssdhip_conv_params* p);

/* ---- A3 (deep 3x3 / stride 1 / pad 1 layers)  Winograd F(2x2,3x3) around the same GEMM kernel: 2.25x fewer MFMA flops.
 * T = N*ceil(H/2)*ceil(W/2) tiles, Tp = T rounded up to 128.  V and M are (16, Tp, C) fp32 planes.
 *   weight:  U (16, rows_pad, KP) = G g G^T per (co, ci), zero padded
 *   input:   V = B^T d B of every 4x4 patch, producer BN scale/shift + ReLU fused on load (in_scale may be NULL)
 *   GEMM:    ssdhip_conv2d_nhwc with N=16, Hi=Ho=1, Wi=Wo=Tp, R=S=1, w_img_stride = rows_pad*KP, rows_per_img = Tp
 *   output:  y (N,H,W,Cout) = A^T m A + bias, BN partial sums (rows, Cout, 2); returns the number of partial rows     */
int ssdhip_winograd_tiles(int N, int H, int W, int* T, int* Tp);
int ssdhip_winograd_weight(const float* w_oihw, float* U, int Cout, int Cin, int rows_pad, int KP, void* stream);
int ssdhip_winograd_input(const float* x, const float* in_scale, const float* in_shift, float* V, int N, int H, int W, int C,
                          void* stream);
int ssdhip_winograd_output(const float* M, const float* bias, float* y, float* stats, int N, int H, int W, int Cout, void* stream);
/* detector heads: M planes have Cp = Cout padded to 32 channels; y is the (N, P, 4+C) prediction tensor at the head's prior
 * offset: element (n,h,w,co) at y[n*y_batch_stride + (h*W+w)*Cout + co]  (model/ssd.py:103-104)                          */
int ssdhip_winograd_output_head(const float* M, const float* bias, float* y, int64_t y_batch_stride, int N, int H, int W, int Cout,
                                int Cp, void* stream);

/* ---- A3 (frozen 3x3 / stride 1 / pad 1 backbone layers, forward)  Winograd F(4x4,3x3), points (0, +-3/2, +-2/3, inf): 36 products per
 * 4x4 output tile = 4x fewer MFMA flops than the direct form of nn.Conv2d (model/vgg16.py:33), 1.78x fewer than F(2x2,3x3).
 * T = N*ceil(H/4)*ceil(W/4) tiles, Tp = T rounded up to row_pad (64 or 128 = the GEMM's m-tile).  V and M are (36, Tp, C) fp32 planes.
 *   weight:  U (36, rows_pad, KP) = G g G^T per (co, ci) in fp64, rounded once, zero padded
 *   input:   V = B^T d B of every 6x6 patch, producer BN scale/shift + ReLU fused on load (in_scale may be NULL), zero pad after it
 *   GEMM:    ssdhip_conv2d_nhwc with N=36, Hi=Ho=1, Wi=Wo=Tp, R=S=1, w_img_stride = rows_pad*KP, rows_per_img = Tp
 *   output:  y (N,H,W,Cout) = A^T m A + bias, BN partial sums (rows, Cout, 2); returns the number of partial rows
 *            (= ssdhip_winograd4_stat_rows, for sizing the table)
 * Per-layer error against an fp64 convolution: 1.1e-6 rms / 2.3e-6 max of the output scale (F(2x2): 4e-7 / 5e-7); csrc/winograd4.hip. */
int ssdhip_winograd4_tiles(int N, int H, int W, int row_pad, int* T, int* Tp);
int ssdhip_winograd4_weight(const float* w_oihw, float* U, int Cout, int Cin, int rows_pad, int KP, void* stream);
int ssdhip_winograd4_input(const float* x, const float* in_scale, const float* in_shift, float* V, int N, int H, int W, int C, int Tp,
                           void* stream);
int ssdhip_winograd4_stat_rows(int N, int H, int W, int Cout);
/* accumulate != 0: y += result (data gradients: two producers of one activation gradient) */
int ssdhip_winograd4_output(const float* M, const float* bias, float* y, float* stats, int accumulate, int N, int H, int W, int Cout,
                            int Tp, void* stream);
/* data gradient of the same layers (autograd backward of nn.Conv2d, train.py:121) = the ADJOINT convolution of dy through the same three
 * passes: U (36, rows_pad >= Cin, KP >= Cout) from the filter flipped with its channel roles swapped; input = dy (N,H,W,KP), output
 * dx (N,H,W,Cin) via ssdhip_winograd4_output(bias = stats = NULL) */
int ssdhip_winograd4_weight_adjoint(const float* w_oihw, float* U, int Cout, int Cin, int rows_pad, int KP, void* stream);
/* detector heads (as ssdhip_winograd_output_head): M planes have Cp = Cout padded to 32 channels; y is the (N, P, 4+C) prediction
 * tensor at the head's prior offset (model/ssd.py:103-104) */
int ssdhip_winograd4_output_head(const float* M, const float* bias, float* y, int64_t y_batch_stride, int N, int H, int W, int Cout,
                                 int Cp, int Tp, void* stream);

/* ---- A3, opt-in: the same F(4x4,3x3) convolution in TWO passes for maps too large for the three-pass form (the 150 / 75 px layers
 * conv_2_1 .. conv_3_1 of model/vgg16.py:33): the products M never reach memory.  Cin % 32 == 0, Cout % 64 == 0.
 *   pack:   U (36, rows_pad, Cin) of ssdhip_winograd4_weight (KP = Cin) -> upk (36 * Cout * Cin floats), the per-K-step LDS images
 *   input:  x (N,H,W,C) -> vb (ssdhip_wino4f_v_floats floats): V = B^T d B in K-step-blocked tiles of 32; producer BN.ReLU fused on load
 *   gemm:   vb, upk -> z (N,H,W,Cout) = A^T (sum_c V U) A + bias and BN partial sums (ssdhip_wino4f_tile_blocks rows, Cout, 2; stats may be
 *           NULL); returns the number of partial rows.  _pool: even H, W; stores per 2x2 / stride 2 pooling window the maximum and the
 *           minimum of z, (N, H/2, W/2, Cout) each, instead of z (as ssdhip_wino_fused_conv_pool; consumer: ssdhip_bn_relu_select)
 * csrc/wino4_fused.hip; same arithmetic as the three-pass form up to the summation order of the GEMM.                                   */
int ssdhip_wino4f_tile_blocks(int N, int H, int W);
int64_t ssdhip_wino4f_v_floats(int N, int H, int W, int Cin);
int ssdhip_wino4f_pack(const float* U, float* upk, int Cout, int Cin, int rows_pad, void* stream);
int ssdhip_wino4f_input(const float* x, const float* in_scale, const float* in_shift, float* vb, int N, int H, int W, int C, void* stream);
int ssdhip_wino4f_gemm(const float* vb, const float* upk, const float* bias, float* z, float* stats, int N, int H, int W, int Cin, int Cout,
                       void* stream);
int ssdhip_wino4f_gemm_pool(const float* vb, const float* upk, const float* bias, float* zmax, float* zmin, float* stats, int N, int H, int W,
                            int Cin, int Cout, void* stream);

/* ---- A4  nn.BatchNorm2d, training=True (model/vgg16.py:34, model/ssd.py:62) -----------------------------
 * partial (MB,C,2) -> scale = gamma*rsqrt(var+eps), shift = beta-mean*scale (biased var); running stats EMA with
 * momentum and unbiased var, exactly nn.BatchNorm2d; save_mean/save_invstd (C) for backward (may be NULL).
 * ws: >= 64*C*2 doubles scratch.  count = N*H*W must be > 1 (reference raises ValueError otherwise).       */
int ssdhip_bn_finalize(const float* partial, int MB, int C, int64_t count, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps,
                       float* scale, float* shift, float* save_mean, float* save_invstd, double* ws, void* stream);
/* eval-mode BatchNorm2d folded to scale/shift from running stats */
int ssdhip_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, int C, float* scale, float* shift, void* stream);

/* ---- A4-A6  BN-apply + ReLU (+ MaxPool2d(2,2,pad)) (model/vgg16.py:28-35) ------------------------------
 * y = maxpool(relu(x*scale+shift)); pad rows/cols contribute -inf like F.max_pool2d.  pool=0: no pooling.   */
int ssdhip_bn_relu_pool(const float* x, const float* scale, const float* shift, float* y,
                        int N, int H, int W, int C, int pool, int pool_pad, void* stream);

/* ---- A10-A16  SSD.loss (model/ssd.py:181-328) -----------------------------------------------------------
 * outputs (N,P,4+C), targets (N,G,4+C) zero-padded rows, default boxes (P,4).
 * terms (3,N,P): l_loc, l_conf_pos, l_conf_neg;  cnt (N) int32 scratch;  sel (N,8) fp32:
 *   [thr_pos, thr_neg, pos_k, neg_k, inv_pos, per_image_loss, 0, 0];  loss (1).
 * Matching is bit-exact w.r.t. the reference's fp32 op order (no FMA contraction, IEEE divide).
 * C = number of classes incl. void, 2 <= C <= 1024 (ssd.py:9 is parametric; VOC's C = 21 has a register-resident
 * specialisation, every other C runs the same arithmetic in the same order from memory).                     */
int ssdhip_multibox_loss_fwd(const float* outputs, const float* targets, const float* dboxes,
                             int N, int P, int G, int C, float alpha, float threshold,
                             float* terms, int32_t* cnt, float* sel, float* loss, void* stream);
/* grad (N,P,4+C) = d loss / d outputs * grad_out[0]  (autograd of ssd.py:226-227; masks are constants)    */
int ssdhip_multibox_loss_bwd(const float* outputs, const float* targets, const float* dboxes,
                             int N, int P, int G, int C, float alpha, float threshold,
                             const float* terms, const float* sel, const float* grad_out, float* grad, void* stream);
/* (N,P,G) uint8 match mask of SSD._match (model/ssd.py:231-250), for parity tests and evaluation            */
int ssdhip_match_mask(const float* targets, const float* dboxes, int N, int P, int G, int row,
                      float threshold, uint8_t* mask, void* stream);

/* ---- A19+A20  utils.calc_coordicate + utils.calc_score (utils.py:19-55), in place on (N,P,4+C) ----------- */
int ssdhip_decode_score(float* outputs, const float* dboxes, int N, int P, int C, void* stream);
/* ---- A21  utils.calc_iou (utils.py:58-77): t (N,T,rowt), s (N,S,rows) -> (N,T,S) ------------------------- */
int ssdhip_calc_iou(const float* t, const float* s, int N, int T, int S, int rowt, int rows, float* iou, void* stream);
/* ---- 8(f)-1  true-positive assignment of the evaluation loop (evaluate.py:134-151) on decoded, scored, NMS-filtered
 * outputs (N,P,4+C) (one positive score column per row, as utils.calc_score leaves them) and gts (N,G,4+C) one-hot rows:
 * cls (N,P) int32 = class of the row's positive score (column 5 + cls) or -1; tp (N,P) uint8 = 1 where the row is the
 * highest-scored claimant (IoU > iou_thresh with its best-IoU ground truth of the same class) of that ground truth;
 * score (N,P) fp32, may be NULL: that positive score (0 where cls = -1);
 * gt_count (C-1) int32, may be NULL: += number of ground-truth rows per class (evaluate.py:138-151 result_count).        */
int ssdhip_eval_match(const float* outputs, const float* gts, int N, int P, int G, int C, float iou_thresh, int32_t* cls, uint8_t* tp,
                      float* score, int32_t* gt_count, void* stream);
/* ---- A22  utils.non_maximum_suppression (utils.py:80-116): greedy, class-agnostic, strict '>' -------------
 * outputs (N,P,4+C) decoded+scored; zeroes outputs[n,p,4:] of suppressed / void rows in place.
 * keep (N,P) uint8 (optional, may be NULL).  P <= 9000 (all-LDS).                                                   */
int ssdhip_nms(float* outputs, int N, int P, int C, float iou_thresh, uint8_t* keep, void* stream);
/* same result with a caller workspace (ssdhip_nms_ws_bytes): sort per image, IoU bit-matrix over the whole chip, serial
 * scan per image on register-resident suppression bits (P <= 9984), or, for 9984 < P <= 32768 (SSD512 extension), a chunked
 * sort and a scan over global mask rows.  Preferred for batches (the LDS-only variant keeps one
 * CU per image busy).                                                                                        */
int64_t ssdhip_nms_ws_bytes(int N, int P);
int ssdhip_nms_ws(float* outputs, int N, int P, int C, float iou_thresh, uint8_t* keep, void* ws, void* stream);
/* option named by the north star ("per-class NMS"), not the reference's behaviour (utils.py:99-109 is class-agnostic): a box is
 * suppressed only by a kept box of the same arg-max class; same workspace, same kernels with one extra compare.              */
int ssdhip_nms_ws_per_class(float* outputs, int N, int P, int C, float iou_thresh, uint8_t* keep, void* ws, void* stream);

/* ---- backward of the trainable tail (autograd of model/ssd.py:49-77 layers; train.py:121) ---------------- */
/* wgrad: dW (Cout,R,S,Cin)->written in OIHW state_dict layout, db (Cout) optional.
 * x (N,Hi,Wi,Cin) with the same optional fused BN+ReLU input transform as the forward; dy (N,Ho,Wo,Cout) element
 * (n,ho,wo,co) at dy[n*dy_batch_stride + (ho*Wo+wo)*Cout + co].  ws: split-K scratch, ssdhip_conv2d_wgrad_ws_bytes. */
typedef struct {
    const float* x; const float* dy; const float* in_scale; const float* in_shift;
    float* dw_oihw; float* dbias; float* ws;
    int64_t dy_batch_stride;  /* floats between images in dy (Ho*Wo*dy_row when dense)          */
    int32_t N, Hi, Wi, Cin, Ho, Wo, Cout, R, S, stride, pad;
    int32_t dy_row;           /* floats per pixel row of dy (>= Cout, multiple of 4; channel padding) */
    int32_t dw_cin;           /* channels of dw_oihw's layout if smaller than Cin (RGB input padded to 4 in x), else 0 */
    const float* wino_v;      /* Winograd variant only: precomputed input transform V (16,Tp,Cin) of x, or NULL */
    int32_t compute;          /* ssdhip_conv2d_wgrad only: 0 = exact fp32 MFMA; 1 = operands rounded to bf16 at LDS staging, bf16 MFMA,
                                 fp32 accumulation and fp32 result (BASELINE configs[3]; opt-in, never the headline) */
} ssdhip_wgrad_params;
int64_t ssdhip_conv2d_wgrad_ws_bytes(const ssdhip_wgrad_params* p);
int ssdhip_conv2d_wgrad(const ssdhip_wgrad_params* p, void* stream);
/* same result for 3x3 / stride 1 / pad 1 layers through Winograd F(2x2,3x3): V = B^T d B (input, fused BN+ReLU), dM = A dY A^T,
 * 16 plane products dU = dM^T V on the wgrad MFMA kernel (2.25x fewer flops), dW = G^T dU G.  dy must be dense.     */
int64_t ssdhip_conv2d_wgrad_winograd_ws_bytes(const ssdhip_wgrad_params* p);
int ssdhip_conv2d_wgrad_winograd(const ssdhip_wgrad_params* p, void* stream);
/* the same through Winograd F(4x4,3x3) (36 planes of a quarter as many tiles: 1.78x fewer MFMA flops; dW = G^T dU G in fp64):
 * p->wino_v, when set, is the forward's F(4x4) input transform of the same activation with planes padded to 128 rows */
int64_t ssdhip_conv2d_wgrad_winograd4_ws_bytes(const ssdhip_wgrad_params* p);
int ssdhip_conv2d_wgrad_winograd4(const ssdhip_wgrad_params* p, void* stream);

/* BatchNorm2d(train)+ReLU backward.  z raw conv output (M,C); da grad w.r.t. relu output.
 * pass 1: partial sums of dy=da*(z*scale+shift>0) and dy*xhat -> (MB,C,2);  pass 2 (after finalize):
 * dz = scale*(dy - mean(dy) - xhat*mean(dy*xhat)), dgamma, dbeta.                                           */
int ssdhip_bn_relu_bwd_reduce(const float* z, const float* da, const float* scale, const float* shift,
                              const float* save_mean, const float* save_invstd, int64_t M, int C,
                              float* partial, int MB_rows, void* stream);
int ssdhip_bn_relu_bwd_finalize(const float* partial, int MB, int C, int64_t M, float* dgamma, float* dbeta,
                                float* coef /* (C,2): mean_dy, mean_dy_xhat */, double* ws, void* stream);
/* dense loss gradient (N,P,row) -> channel-padded NHWC head gradient (N,HW,Cpad), pad channels zeroed
 * (feeds dgrad/wgrad of one detector head; inverse of the zero-copy concat of model/ssd.py:103-104)        */
int ssdhip_scatter_head_grad(const float* grad, float* out, int N, int P, int row, int prior_offset, int HW,
                             int Cout, int Cpad, void* stream);
/* all heads in one launch: `jobs_device` = n_jobs records in DEVICE memory (one per head: the arguments of ssdhip_scatter_head_grad that
 * differ between heads, validated by the caller); max_elements = the largest N * HW * Cpad among them */
typedef struct {
    float* out;
    int32_t prior_offset, HW, Cout, Cpad;
} ssdhip_scatter_job;
int ssdhip_scatter_head_grads(const float* grad, const ssdhip_scatter_job* jobs_device, int n_jobs, int N, int P, int row, int64_t max_elements,
                              void* stream);
int ssdhip_bn_relu_bwd_apply(const float* z, const float* da, const float* scale, const float* shift,
                             const float* save_mean, const float* save_invstd, const float* coef,
                             int64_t M, int C, float* dz, void* stream);

/* ---- A17  torch.optim.Adam(lr, weight_decay) step over many tensors (train.py:97,122) -------------------
 * table (n_tensors,4) device pointers {param, grad, exp_avg, exp_avg_sq}, sizes (n_tensors) int64 element counts,
 * blockmap: (n_blocks,2) int32 {tensor, chunk} built by the host (all three arrays live in DEVICE memory).
 * L2 decay added to grad (not AdamW).                                                                        */
int ssdhip_adam_multi_tensor(const void* const* table, const int64_t* sizes, const int32_t* blockmap, int n_blocks,
                             int chunk, const float* hyper, void* stream);
/* hyper (device, 8 floats): [lr, beta1, beta2, eps, weight_decay, 1-beta1^t, sqrt(1-beta2^t), grad_scale];
 * adam_tick increments the device step counter and refreshes slots 5,6 (graph-replayable step bookkeeping).  */
int ssdhip_adam_tick(int32_t* step, float* hyper, void* stream);

/* ---- A3  conv_1_1 (model/vgg16.py:25,33: nn.Conv2d(3, 64, 3, padding=1) on the normalised image) as a dedicated HBM-bound
 * kernel: x (N,H,W,4) fp32 (RGB padded to 4), w OIHW (64,3,3,3) unpacked, z (N,H,W,64), stats NULL or
 * (ssdhip_conv_rgb3x3_stat_rows, 64, 2) per-wave partial [sum, sumsq] for train-mode BatchNorm.                          */
int ssdhip_conv_rgb3x3_stat_rows(int N, int H, int W);
int ssdhip_conv_rgb3x3(const float* x_nhwc4, const float* w_oihw, const float* bias, float* z, float* stats, int N, int H, int W, int Cout,
                       void* stream);

/* ---- A3  the Cin = 64 3x3/s1/p1 layers on big maps (conv_1_2, conv_2_1; model/vgg16.py:25,33): K = 576 is too short for the
 * staged implicit-GEMM kernel; here the A operand comes straight from global memory and the weights stay in LDS.
 * x (N,H,W,64), wimg = ssdhip_conv_c64_pack(w OIHW (Cout,64,3,3)) of ssdhip_conv_c64_weight_floats(Cout) floats, optional fused
 * BN.ReLU of the input (in_scale/in_shift, 64 each), z (N,H,W,Cout), stats NULL or (ssdhip_conv_c64_stat_rows(Cout), Cout, 2).
 * Cout = 64, 128 or 256.                                                                                                */
int64_t ssdhip_conv_c64_weight_floats(int Cout);
int ssdhip_conv_c64_pack(const float* w_oihw, float* wimg, int Cout, void* stream);
int ssdhip_conv_c64_stat_rows(int Cout);
int ssdhip_conv_c64(const float* x, const float* wimg, const float* bias, const float* in_scale, const float* in_shift, float* z,
                    float* stats, int N, int H, int W, int Cout, void* stream);

/* ---- A3 + A4 partials  the 3x3/s1/p1 layers of the backbone (model/vgg16.py:33: nn.Conv2d(cin, v, kernel_size=3, padding=1)) as ONE
 * fused Winograd F(2x2,3x3) kernel: input transform (with the producer's BN.ReLU), the 16 Winograd-domain products on the fp32
 * MFMA and the output transform + bias + BatchNorm partial sums, nothing of the Winograd domain in HBM.  Persistent grid (one
 * block per CU) that walks (tile block, channel tile) items as one software pipeline.
 * x (N,H,W,Cin) NHWC, Cin % 16 == 0, 32 <= Cin <= 512, W >= 9; upk = ssdhip_wino_fused_pack(w OIHW (Cout,Cin,3,3)), Cout % 64 == 0,
 * of ssdhip_wino_fused_weight_floats(Cout, Cin) floats; in_scale/in_shift (Cin) or both NULL; z (N,H,W,Cout);
 * stats NULL or (ssdhip_wino_fused_stat_rows(N,H,W), Cout, 2); geo = tile-block geometry of the map shape, filled once per
 * (N,H,W) by ssdhip_wino_fused_geo into ssdhip_wino_fused_geo_bytes(N,H,W) bytes (caller-owned, reusable by every layer and
 * step of that shape).  max_blocks: 0 = one block per CU (256); a smaller grid leaves whole CUs to kernels running on other
 * streams.  ssdhip_wino_fused_conv returns the number of stats rows (>= 0) or a negative code.                               */
int64_t ssdhip_wino_fused_weight_floats(int Cout, int Cin);
int ssdhip_wino_fused_pack(const float* w_oihw, float* upk, int Cout, int Cin, void* stream);
int ssdhip_wino_fused_stat_rows(int N, int H, int W);
int64_t ssdhip_wino_fused_geo_bytes(int N, int H, int W);
int ssdhip_wino_fused_geo(int N, int H, int W, void* geo, void* stream);
int ssdhip_wino_fused_conv(const float* x, const float* upk, const float* bias, const float* in_scale, const float* in_shift, float* z,
                           float* stats, const void* geo, int N, int H, int W, int Cin, int Cout, int max_blocks, void* stream);
/* the same layer when a 2x2 / stride 2 / pad 0 nn.MaxPool2d follows (model/vgg16.py:36; even H, W): instead of z (N,H,W,Cout) the
 * kernel stores, per pooling window, max z and min z -- zmax, zmin (N,H/2,W/2,Cout) -- and ssdhip_bn_relu_select turns them into the
 * pooled activation max-pool(relu(bn(z))) exactly (relu(bn(.)) is monotonic in z); BN partial sums still cover every z */
int ssdhip_wino_fused_conv_pool(const float* x, const float* upk, const float* bias, const float* in_scale, const float* in_shift,
                                float* zmax, float* zmin, float* stats, const void* geo, int N, int H, int W, int Cin, int Cout,
                                int max_blocks, void* stream);
/* the same, storing per pooling window only the extreme the pool will use: the maximum where bn_weight[co] >= 0, else the minimum (the sign
 * of a channel's BatchNorm scale gamma * invstd is gamma's, known before the batch statistics).  zsel (N, H/2, W/2, Cout); consumer:
 * ssdhip_bn_relu_pool(zsel, scale, shift, y, ..., pool = 0) -- bit-identical to conv -> ssdhip_bn_relu_pool(pool = 1), half the stores of the
 * pair form and a third less for the consumer to read */
int ssdhip_wino_fused_conv_pool_sel(const float* x, const float* upk, const float* bias, const float* in_scale, const float* in_shift,
                                    const float* bn_weight, float* zsel, float* stats, const void* geo, int N, int H, int W, int Cin, int Cout,
                                    int max_blocks, void* stream);
int ssdhip_bn_relu_select(const float* zmax, const float* zmin, const float* scale, const float* shift, float* y, int64_t pixels, int C,
                          void* stream);

/* ---- BASELINE configs[3] / [4] (bf16 forward / backward with fp32 master weights), backbone forward: the 3x3/s1/p1 layers of
 * model/vgg16.py:33 on bf16 activations kept in HBM, v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fed by LDS-DMA from an image
 * patch that serves all 9 taps.  Opt-in (SSD.set_compute_dtype('bf16')); never the fp32 headline path.
 * a = ACTIVATED input (BatchNorm + ReLU applied), bf16, NHWC with a one-pixel zero halo: (N, H+2, W+2, Cin) inside a zero-initialised
 * allocation of ssdhip_conv_bf16_act_bytes(N,H,W,Cin) bytes (< 4 GiB; the slack behind the tensor is read, never written);
 * Cin % 32 == 0, W >= 14.  wpk = ssdhip_conv_bf16_pack(w OIHW (Cout,Cin,3,3) fp32), ssdhip_conv_bf16_weight_bytes(Cout,Cin) bytes, Cout % 64 == 0.
 * z (N,H,W,Cout) = conv + bias, fp32 (z_bf16 = 0) or bf16 (1); stats NULL or (ssdhip_conv_bf16_stat_rows(N,H,W), Cout, 2) fp32
 * partial sum / sum of squares of the fp32 values.  max_blocks as in ssdhip_wino_fused_conv.  Returns the stats rows or < 0.   */
int64_t ssdhip_conv_bf16_weight_bytes(int Cout, int Cin);
int ssdhip_conv_bf16_pack(const float* w_oihw, void* wpk, int Cout, int Cin, void* stream);
int64_t ssdhip_conv_bf16_act_bytes(int N, int H, int W, int C);
int ssdhip_conv_bf16_stat_rows(int N, int H, int W);
int ssdhip_conv_bf16(const void* a, const void* wpk, const float* bias, void* z, int z_bf16, float* stats, int N, int H, int W, int Cin,
                     int Cout, int max_blocks, void* stream);

/* The producer side of ssdhip_conv_bf16's input: BatchNorm apply + ReLU (+ MaxPool2d(2,2,pool_pad) when pool != 0) of the pre-BN map
 * z (N,H,W,C) fp32 (z_bf16 = 0) or bf16 (1), written as bf16 into the INTERIOR of the haloed tensor out (N, Ho+2, Wo+2, C) (vgg16.py:28-35);
 * the halo and the slack of the allocation stay as the caller zero-initialised them.  C % 8 == 0.                               */
int ssdhip_bn_relu_to_bf16(const void* z, int z_bf16, const float* scale, const float* shift, void* out, int N, int H, int W, int C, int pool,
                           int pool_pad, void* stream);

/* ---- 8(f)-4  full fine-tuning (train.py:69-74 un-freezes nothing; extension): backward of MaxPool2d(2,2,pad) over the
 * recomputed a = relu(z*scale+shift) (vgg16.py:28-35): dpooled (N,Ho,Wo,C) -> da (N,H,W,C), gradient to the first maximum
 * of each window in scan order (ATen rule).                                                                            */
int ssdhip_bn_relu_pool_bwd(const float* z, const float* scale, const float* shift, const float* dpooled, float* da, int N, int H, int W,
                            int C, int pool_pad, void* stream);

/* ---- 8(f)-4  VGG16 classification head (model/vgg16.py:42-61, 96-101, 117-130).  The Linear layers run on ssdhip_conv2d_nhwc /
 * ssdhip_conv2d_wgrad as 1x1 convolutions; these are the element-wise pieces.
 * nhwc_flatten: (N,H,W,C) NHWC <-> the reference's x.view(N,-1) of NCHW (index c*H*W + h*W + w); inverse != 0 goes back.
 * relu_dropout: out = relu(x) * mask / (1-p) (dy == NULL) or dy * that mask (backward); mask[i] = 24-bit counter hash(key, i) >= p
 *   (synth.hash_u32); training == 0: plain ReLU.  n < 2^32.
 * log_softmax_dot: loss[0] = sign * mean_n sum_c t[n][c] * log_softmax(z[n])[c] (rowval: N floats scratch);
 *   bwd: dlogits = grad_out[0] * sign / N * (t - softmax(z) * sum_c t)  (scratch1: 1 float).  sign = +1: vgg16.py:127-129 literally. */
int ssdhip_nhwc_flatten(const float* src, float* dst, int N, int H, int W, int C, int inverse, void* stream);
int ssdhip_relu_dropout(const float* x, const float* dy, float* out, int64_t n, float p, uint32_t key, int training, void* stream);
int ssdhip_log_softmax_dot_fwd(const float* logits, const float* targets, int N, int C, float sign, float* rowval, float* loss, void* stream);
int ssdhip_log_softmax_dot_bwd(const float* logits, const float* targets, int N, int C, float sign, const float* grad_out, float* scratch1,
                               float* dlogits, void* stream);

/* ---- 8(f)-2  batch augmentation (augmentation/random.py:6-42, augmentation/to_tensor.py:8-16, applied per sample at
 * dataset.py:43-52; pipeline train.py:32-36): colour jitter ops in the sampled order, horizontal flip, uint8 -> [0,1]
 * fp32 (ToTensor) and up to three erased boxes, on the whole batch.
 * src: (N,H,W,3) uint8 as decoded + resized by the dataset; params: (N, ssdhip_augment_row_floats()) fp32 rows
 * [n_colour_ops (0|4), op order x4 (0 brightness 1 contrast 2 saturation 3 hue), factor of op 0..3, flip, n_erase (<=3),
 *  3 x (top,left,height,width)]; out: (N,3,H,W) fp32; ws: ssdhip_augment_ws_bytes(N) bytes.                              */
int ssdhip_augment_row_floats(void);
int64_t ssdhip_augment_ws_bytes(int N);
int ssdhip_augment_batch(const uint8_t* src_nhwc_u8, const float* params, int N, int H, int W, float* out_nchw, void* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSDHIP_H */
