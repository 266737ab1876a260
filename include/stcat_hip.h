/* libstcat_hip — C ABI of the MI355X-native STCAT hot path.
 *
 * The reference (jy0205/STCAT) is pure PyTorch: it has no FFI of its own, every
 * kernel it runs is reached through torch.nn modules.  Each entry point below
 * therefore replaces a torch call site of the hot path and cites it
 * (file:line relative to the reference checkout).  INTEGRATION.md shows the
 * ctypes binding and the factory seam (models/pipeline.py:6-8) a maintainer
 * plugs them into.
 *
 * Conventions
 *   - all tensors are fp32 device pointers (HIP), dense unless a leading
 *     dimension is given; activations are NHWC / token-major [rows][features];
 *     conv weights are OHWI (== a torch [O,I,H,W] tensor in channels_last).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - return 0 on success; <0 = invalid argument (see stcat_last_error());
 *     >0 = hipError_t of the failed launch.  Nothing is allocated or freed,
 *     no host synchronisation happens inside any call (graph-capture safe).
 */
#ifndef STCAT_HIP_H_
#define STCAT_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

int stcat_version(void);
const char* stcat_last_error(void);
/* arithmetic of the implicit-GEMM family (conv / Linear fwd, dgrad, wgrad); inputs and outputs stay fp32:
 *   0 = fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32 products)
 *   2 = split-bf16 x3: x = hi + lo in bf16, hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate
 *   3 = split-bf16 x6: three bf16 pieces, six cross terms (fp32-class products) */
int stcat_set_mma_mode(int mode);
int stcat_get_mma_mode(void);
/*   4 = split-bf16 x3 with the backbone's activations, gradients and weights PRE-SPLIT into bf16 planes in HBM (the
 *       stcat_pl_* entry points below); every other GEMM of the path runs as mode 2
 *   5 = three bf16 planes per backbone tensor (hi + mid + lo = the fp32 value exactly), six cross terms; others as mode 3
 *   6 = (round 4, experimental) TWO IEEE-fp16 planes per backbone tensor (22 significand bits), three cross terms on
 *       v_mfma_f32_32x32x16_f16; weight planes hold w * 2^wlog and gradient planes dy * 2^glog (fp16's range), undone in the
 *       consuming epilogues; others as mode 3 */
/* power-of-two operand scales of mode 6 (defaults 6 and 16: profiles/r04_plane_range_report.log); which: 0 weight, 1 gradient */
int stcat_set_f16_scales(int weight_log2, int grad_log2);
int stcat_get_f16_scale(int which);
/* tuning/test hook: force the implicit-GEMM block tile (128x128, 128x64, 64x64; 0,0 = heuristic) */
int stcat_debug_force_tile(int bm, int bn);
/* stream-K scheduling of the split-bf16 forward GEMM (opt-in experiment): 1 whenever legal,
 * 0 / -1 off (default) */
int stcat_debug_streamk(int mode);
/* one workgroup that spins for `microseconds` of wall clock on `stream`: the probe of stcat_amd.ops.pick_streams()
 * (HIP maps the streams of a process onto GPU_MAX_HW_QUEUES = 4 hardware queues; two streams that share a queue
 * serialise, and which ones share depends on what else — RCCL, the framework — created streams before us) */
int stcat_spin(int microseconds, void* stream);
/* a HIP stream for a background lane of the step — round 6: the NEXT clip's frozen backbone prefix (stem + max-pool +
 * layer1: no gradient, models/vision_model/backbone.py:78-85) runs under the current step's grounding section
 * (stcat_amd/backbone.py: Backbone.stage_next).  priority -1 / 0 / 1 = the device's greatest / default / least stream
 * priority; cus > 0 = a stream restricted to the first `cus` compute units (hipExtStreamCreateWithCUMask; wins over
 * priority).  The hipStream_t comes back in *out (host pointer); the caller destroys it. */
int stcat_stream_create(int priority, int cus, void** out);
int stcat_stream_destroy(void* stream);

/* ---- backbone: torchvision ResNet-101 + FrozenBatchNorm2d (models/vision_model/backbone.py:16-66,
 *      93-121; torch conv2d/max_pool2d underneath) ------------------------------------------------ */

/* scale = w*rsqrt(rv+eps), bias = b - rm*scale  (backbone.py:56-66) */
int stcat_frozen_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float* scale,
                         float* bias, int C, float eps, void* stream);
/* stem 7x7/2 pad 3 conv on the NCHW frame tensor [n,3,H,W] with OIHW weight [64,3,7,7],
 * + FrozenBN + ReLU -> NHWC [n,H/2,W/2,64]  (resnet conv1/bn1/relu) */
int stcat_stem_fwd(const float* frames, const float* w, const float* scale, const float* bias, float* y, int n,
                   int H, int W, void* stream);
/* The same stem fed by the video decoder's own output: uint8 frames [n,H,W,3] (HWC).  ToTensor + Normalize of the input
 * pipeline (datasets/vidstg.py:140, datasets/transforms.py:155-168) happen inside the patch gather:
 * x = u8 * in_scale[c] + in_shift[c] with in_scale = 1 / (255 std), in_shift = -mean / std; padding stays 0 AFTER
 * normalisation, exactly as conv2d pads the normalised tensor.  4x fewer input bytes, no normalisation pass. */
int stcat_stem_u8_fwd(const unsigned char* frames_hwc, const float* w, const float* in_scale, const float* in_shift,
                      const float* scale, const float* bias, float* y, int n, int H, int W, void* stream);
/* 3x3/2 pad 1 max-pool, NHWC (resnet maxpool) */
int stcat_maxpool3x3s2(const float* x, float* y, int n, int H, int W, int C, void* stream);
/* y = relu?(scale*conv(x,w) + bias + res), x NHWC [n,H,W,Cin], w OHWI [Cout,KH,KW,Cin]
 * (Bottleneck conv+bn(+add)+relu; any of scale/bias/res may be NULL) */
int stcat_conv_fwd(const float* x, const float* w, const float* scale, const float* bias, const float* res,
                   float* y, int n, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                   int relu, void* stream);
/* dx = conv_transpose(g, w) (+ add): g NHWC [n,OH,OW,Cout] -> dx NHWC [n,H,W,Cin]  (autograd of conv2d).
 * With mask_y (the [n,H,W,Cin] output of the conv+FrozenBN+ReLU that produced this conv's input) the epilogue
 * also applies that layer's backward: dx = (y > 0 ? dx : 0) * mask_scale[c]  (mask_scale may be NULL).
 * dx2 (optional, with dx2_scale[c]) receives dx * dx2_scale: at a bottleneck boundary dx is the identity-path
 * gradient of the block below and dx2 its conv3 upstream gradient.  wt: optional transposed weights (below). */
int stcat_conv_dgrad(const float* g, const float* w, const float* add, const float* mask_y, const float* mask_scale,
                     float* dx, float* dx2, const float* dx2_scale, const float* wt, int n, int H, int W, int Cin,
                     int Cout, int KH, int KW, int stride, int pad, void* stream);
/* wt = stcat_weight_transpose(w): OHWI [Cout][taps][Cin] -> [taps][Cin][Cout].  When given (and a split-bf16 mode is
 * active) the data gradient runs on the forward kernel's staging path; NULL keeps the generic path. */
int stcat_weight_transpose(const float* w, float* wt, int Cout, int taps, int Cin, void* stream);
/* the same transpose for many weights in ONE launch: DEVICE table of entries
 *   { const float* w; float* wt; int Cout, taps, Cin; int blk0, nbx, nby; }   (stcat_weight_transpose_entry_bytes() == 40)
 * where entry e owns grid blocks [blk0, blk0 + nbx*nby*taps), nbx = ceil(Cin/32), nby = ceil(Cout/32), blk0 ascending */
int stcat_weight_transpose_entry_bytes(void);
int stcat_weight_transpose_multi(const void* table, int n_entries, int total_blocks, void* stream);
/* dw (OHWI, caller-zeroed) += sum over pixels g (x) gathered x  (autograd of conv2d w.r.t. weight) */
int stcat_conv_wgrad(const float* g, const float* x, float* dw, int n, int H, int W, int Cin, int Cout, int KH,
                     int KW, int stride, int pad, void* stream);
/* backward of y = relu?(scale*z + bias (+res)): dz = dy*[y>0]; G = dz*scale[c] (may be NULL); dres = dz (may be NULL) */
int stcat_act_bwd(const float* dy, const float* y, const float* scale, float* G, float* dres, long n, int C,
                  int relu, void* stream);

/* ---- plane-format backbone (stcat_set_mma_mode(4)) -------------------------------------------------
 * Same call sites as the stcat_conv_* family above (torchvision resnet101 inside
 * models/vision_model/backbone.py:115-119, FrozenBatchNorm2d :56-66, autograd of conv2d), but every tensor is a
 * PAIR of bf16 planes (h, l) with x = h + l, h = bf16(x), l = bf16(x - h): NHWC [n,H,W,C] per plane, weights OHWI per
 * plane.  The split is done once by the producing kernel's epilogue, so the GEMM main loop moves operands
 * HBM -> LDS by LDS-DMA and issues nothing but fragment reads and MFMAs (stcat_amd/csrc/igemm_pl.h).
 * Products are hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (= mode 2 arithmetic). */
/* y = relu?(scale*conv(x,w) + bias + res): planes in, planes (yh, yl) and/or fp32 (yf) out; res planes optional.
 * ymask (optional, [n*OH*OW][Cout / 8] bytes): bit e of byte j of a row = (y[8 j + e] > 0) — the ReLU mask the
 * backward pass needs, at 1/32 of the bytes of re-reading y */
int stcat_pl_conv_fwd(const void* xh, const void* xl, const void* wh, const void* wl, const float* scale,
                      const float* bias, const void* rh, const void* rl, void* yh, void* yl, float* yf,
                      unsigned char* ymask, int n, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                      int relu, void* stream);
/* dx = conv_transpose(g, w) (+ add), then the fused ReLU + FrozenBN backward of the layer below (y planes,
 * mask_scale) and the optional second output dx2 = dx * dx2_scale — semantics of stcat_conv_dgrad.  The mask comes
 * either as the y planes (yh, yl) or as the bit mask the forward pass wrote (ybits, see stcat_pl_conv_fwd).  th / tl
 * are the TRANSPOSED weight planes [taps][Cin][Cout] written by stcat_weight_planes_multi. */
int stcat_pl_conv_dgrad(const void* gh, const void* gl, const void* th, const void* tl, const void* addh,
                        const void* addl, const void* yh, const void* yl, const unsigned char* ybits,
                        const float* mask_scale, void* dxh, void* dxl,
                        void* dx2h, void* dx2l, const float* dx2_scale, int n, int H, int W, int Cin, int Cout, int KH,
                        int KW, int stride, int pad, void* stream);
/* Block-boundary data gradient of a Bottleneck with a strided downsample branch (torchvision Bottleneck.forward:
 * out = conv3(..) + downsample(x); call site models/vision_model/backbone.py:115-119): dx = [ybits](g . w1^T +
 * scatter(addc)), addc [n][ceil(H/add_stride)][ceil(W/add_stride)][Cin] = the downsample conv's data gradient on its own
 * coarse grid (a plain 1x1 GEMM through stcat_pl_conv_dgrad with stride 1), placed at pixels (add_stride i, add_stride j).
 * 1x1, stride 1, bf16-plane modes; ybits / mask_scale as in stcat_pl_conv_dgrad. */
int stcat_pl_conv_dgrad_cadd(const void* gh, const void* gl, const void* th, const void* tl, const void* addch,
                             const void* addcl, int add_stride, const unsigned char* ybits, const float* mask_scale,
                             void* dxh, void* dxl, int n, int H, int W, int Cin, int Cout, void* stream);
/* Linear layers on the plane kernels (the spatial encoder layers' FFN, modal_encoder.py:239-240) with plane operands
 * (stcat_pl_split of the LayerNorm output / the previous plane result; weight planes from stcat_weight_planes_multi on W
 * viewed as [N,1,1,K]): y [M,N] = dropout_p(relu?(x w^T + bias + addf)) written as fp32 (yf) and / or as planes (yh, yl);
 * ymask (optional) = the bit mask y > 0 ([M][N/8] bytes); addf (optional) an fp32 [M,N] term (a residual; a gradient to sum).
 * The dropout decision of element (m, n) is counter drop_offset + m * N + n, as stcat_dropout draws it on [M,N]. */
int stcat_pl_linear_fwd(const void* xh, const void* xl, const void* wh, const void* wl, const float* bias, const float* addf,
                        float* yf, void* yh, void* yl, unsigned char* ymask, int M, int N, int K, int relu, float drop_p,
                        long drop_seed, long drop_offset, const long* drop_base, void* stream);
/* ... and the data gradient of linear2 with that pair's backward in its epilogue: dx [M,K] = [ybits] mask_scale[k]
 * (g [M,N] . w [N,K]) as fp32 (dxf) and / or planes (dxh, dxl); th / tl = the transposed weight planes [K][N]; ybits = the
 * forward's ymask, mask_scale = 1/(1-p). */
int stcat_pl_linear_dgrad_mask(const void* gh, const void* gl, const void* th, const void* tl, const unsigned char* ybits,
                               const float* mask_scale, float* dxf, void* dxh, void* dxl, int M, int N, int K, void* stream);
/* planes (h, l) of x + y and, when `sum` is given, the fp32 sum: q = k = src + pos (modal_encoder.py:234) entering the
 * in-projection on the plane kernels in one pass; n a multiple of 8, all pointers 16-byte aligned */
int stcat_pl_split_sum(const float* x, const float* y, float* sum, void* h, void* l, long n, void* stream);
/* out [N] (caller-zeroed) += column sums of a plane set [M][N]: the bias gradient where the upstream gradient is planes */
int stcat_pl_colsum(const void* h, const void* l, float* out, int M, int N, void* stream);
/* dw (fp32 OHWI, caller-zeroed) += row_scale[co] * sum over pixels g (x) gathered x; Cout % 128 == 0, Cin % 128 == 0.
 * row_scale (optional, [Cout]): a FrozenBN scale folded out of g — dz * scale is never materialised */
int stcat_pl_conv_wgrad(const void* gh, const void* gl, const void* xh, const void* xl, float* dw, const float* row_scale,
                        int n, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, void* stream);
/* the same with a caller-owned workspace of ws_floats fp32 values (one per stream: launches on a stream reuse it in order):
 * when (reduction slices) x Cout x KH KW Cin values fit, the slices STORE their partial tiles and a second launch sums them in
 * slice order — no atomics, dw bit-identical run to run; otherwise the atomic form */
int stcat_pl_conv_wgrad_ws(const void* gh, const void* gl, const void* xh, const void* xl, float* dw, const float* row_scale,
                           int n, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, float* ws,
                           long ws_floats, void* stream);
/* resnet maxpool 3x3/2 pad 1: fp32 NHWC in (stem output) -> planes out */
int stcat_pl_maxpool3x3s2(const float* x, void* yh, void* yl, int n, int H, int W, int C, void* stream);
/* fp32 <-> planes; n % 8 == 0 */
int stcat_pl_split(const float* x, void* h, void* l, long n, void* stream);
int stcat_pl_join(const void* h, const void* l, float* out, long n, void* stream);
/* stcat_act_bwd with plane outputs: dz = dy * [y > 0] (dy, y fp32; relu = 0: no mask) -> (rh, rl) if given;
 * g = dz * scale[c] -> (gh, gl) if given */
int stcat_pl_act_bwd(const float* dy, const float* y, const float* scale, void* gh, void* gl, void* rh, void* rl, long n,
                     int C, int relu, void* stream);
/* g = x * scale[c] on planes (upstream gradient of a downsample conv: dz * FrozenBN scale) */
int stcat_pl_scale(const void* xh, const void* xl, const float* scale, void* gh, void* gl, long n, int C, void* stream);
/* weight planes for many conv weights in ONE launch: DEVICE table of entries
 *   { const float* w; bf16* wh, *wl, *th, *tl; const float* tscale; int Cout, taps, Cin; int blk0, nbx, nby; int pad; }
 * (stcat_weight_planes_entry_bytes() == 80): w fp32 OHWI -> (wh, wl) OHWI planes and, when th != NULL, the transposed
 * planes (th, tl) [taps][Cin][Cout] of w * tscale[co] (tscale optional); entry e owns grid blocks [blk0, blk0 + nbx*nby*taps), nbx = ceil(Cin/32),
 * nby = ceil(Cout/32), blk0 ascending */
int stcat_weight_planes_entry_bytes(void);
int stcat_weight_planes_multi(const void* table, int n_entries, int total_blocks, void* stream);
/* tuning/test hook: force the plane-GEMM tile (0: 256x256, 1: 256x128, 2: 128x256, 3: 128x128, 4: 256x64; -1 = heuristic) */
int stcat_debug_force_pl_tile(int index);
/* timing experiments, bit set (results are WRONG when bits 0, 1, 4, 5 or 6 are set): 1 = weight gradient without its atomics,
 * 2 = forward / data-gradient epilogue without global memory traffic, 4 = three-plane K <= 512 layers on the two-workgroup
 * 128 x 64 tile, 8 = phase stagger of the first-round workgroups (flags >> 8 = 10 ns ticks per quarter period; results stay
 * correct), 16 = no K loop (epilogue only), 32 = no epilogue loads, 64 = no plane stores; 0 = off
 * (profiles/r04_plane_gemm_experiments.log) */
int stcat_debug_pl_flags(int flags);

/* ---- position embeddings ------------------------------------------------------------------------ */
/* PositionEmbeddingSine(128, normalize=True) (vision_model/position_encoding.py:70-94):
 * mask [n,h,w] bytes (1 = pad) -> pos [n,h*w,256]; dimt = 128 host-computed divisors */
int stcat_pos_sine_2d(const unsigned char* mask, const float* dimt, float* pos, int n, int h, int w, void* stream);
/* gen_sineembed_for_position (models/net_utils.py:29-56): anchors [M,4] -> [M,512], and its gradient */
int stcat_sine_embed_fwd(const float* anchor, const float* dimt, float* out, int M, void* stream);
int stcat_sine_embed_bwd(const float* anchor, const float* dimt, const float* dout, float* danchor, int M,
                         void* stream);

/* ---- Linear / LayerNorm / glue (torch.nn.Linear, LayerNorm call sites in modal_encoder.py:207-242,
 *      query_decoder.py:250-438, 553-660, net_utils.py:7-26, pipeline.py:41) -------------------- */
/* y[m, :N] = relu?(x[m,:K] . w[N,K]^T + bias + res[m]); output row m is written at
 * (m / c_group)*c_group_stride + (m % c_group)*ldy  (c_group <= 0: plain m*ldy).  N % 64 == 0, K % 16 == 0. */
int stcat_linear_fwd(const float* x, const float* w, const float* bias, const float* res, float* y, int M, int N,
                     int K, int ldx, int ldy, int ldr, int relu, int c_group, long c_group_stride, void* stream);
/* dx[M,K] = g[M,N] . w[N,K] (+ add[M,K]);  K % 64 == 0, N % 16 == 0; wt = optional w^T [K][N] (weight_transpose) */
int stcat_linear_dgrad(const float* g, const float* w, const float* add, const float* wt, float* dx, int M, int N,
                       int K, int ldg, int lddx, void* stream);
/* Accumulating forms for the decoders' skinny launches (M <= 128; the [T,256] query states, query_decoder.py:329-438,
 * 587-660): y += x w^T + bias (+ res) and dx += g w (+ add) onto outputs that already hold the value to add to (zeros
 * from the caller's zeroed arena).  The reduction is split over grid.z (>= 2 K-tiles per slice, atomic epilogue) with
 * no memset launch in front.  Split-bf16 modes only; other shapes are refused. */
int stcat_linear_fwd_acc(const float* x, const float* w, const float* bias, const float* res, float* y, int M, int N,
                         int K, int ldx, int ldy, int ldr, void* stream);
int stcat_linear_dgrad_acc(const float* g, const float* w, const float* add, float* dx, int M, int N, int K, int ldg,
                           int lddx, void* stream);
/* The feed-forward block's `dropout(activation(linear1(x)))` (modal_encoder.py:239-240, query_decoder.py:435-436,
 * 657-658) as ONE launch: y = dropout_p(relu?(x w^T + bias (+ res))), the dropout decision of element (m, n) drawn from
 * counter drop_offset + m * N + n of the site (what stcat_dropout draws on the dense [M,N] tensor).  Dense output
 * (ldy == N); split-bf16 modes only. */
int stcat_linear_fwd_drop(const float* x, const float* w, const float* bias, const float* res, float* y, int M, int N,
                          int K, int ldx, int ldy, int ldr, int relu, float drop_p, long drop_seed, long drop_offset,
                          const long* drop_base, void* stream);
/* ... and the backward of that pair folded into the data gradient of linear2 (modal_encoder.py:240, the autograd
 * of `linear2(dropout(relu(.)))`): dx = [mask_y > 0] * mask_gain * (g . w (+ add)), where mask_y = the forward's
 * dropout(relu(.)) output (positive <=> ReLU passed and the element was kept) and mask_gain = 1 / (1 - p). */
int stcat_linear_dgrad_mask(const float* g, const float* w, const float* add, const float* wt, const float* mask_y,
                            float mask_gain, float* dx, int M, int N, int K, int ldg, int lddx, void* stream);
/* Up to eight INDEPENDENT skinny Linear problems of one shape (M <= 128 rows: the [T,256] query states of the decoders)
 * in ONE launch — the seven input projections of a box-decoder layer's self-attention (sa_qcontent / sa_qtime / sa_qpos /
 * sa_kcontent / sa_ktime / sa_kpos / sa_v, query_decoder.py:329-338), the three in-projections of nn.MultiheadAttention
 * (:341), ca_qcontent / ca_qpos / ca_qpos_sine (:360-369) — and the matching data / weight gradients.  Accumulating forms:
 * y_j += x_j w_j^T + b_j, dx_j += g_j w_j (+ add_j), dw_j += g_j^T x_j (db_j += column sums) onto outputs that hold the
 * value to add to (zeros from the caller's arena); problems may share an output (q = Wqc tgt + Wqt time + Wqp pos).
 * Unused slots are NULL; split-bf16 modes only. */
int stcat_linear_fwd_multi(int n, const float* x0, const float* x1, const float* x2, const float* x3, const float* x4, const float* x5, const float* x6, const float* x7, const float* w0, const float* w1, const float* w2, const float* w3, const float* w4, const float* w5, const float* w6, const float* w7, const float* b0, const float* b1, const float* b2, const float* b3, const float* b4, const float* b5, const float* b6, const float* b7,
                           float* y0, float* y1, float* y2, float* y3, float* y4, float* y5, float* y6, float* y7, int M, int N, int K, void* stream);
int stcat_linear_dgrad_multi(int n, const float* g0, const float* g1, const float* g2, const float* g3, const float* g4, const float* g5, const float* g6, const float* g7, const float* w0, const float* w1, const float* w2, const float* w3, const float* w4, const float* w5, const float* w6, const float* w7, const float* a0, const float* a1, const float* a2, const float* a3, const float* a4, const float* a5, const float* a6, const float* a7,
                             float* d0, float* d1, float* d2, float* d3, float* d4, float* d5, float* d6, float* d7, int M, int N, int K, void* stream);
int stcat_linear_wgrad_multi(int n, const float* g0, const float* g1, const float* g2, const float* g3, const float* g4, const float* g5, const float* g6, const float* g7, const float* x0, const float* x1, const float* x2, const float* x3, const float* x4, const float* x5, const float* x6, const float* x7, float* dw0, float* dw1, float* dw2, float* dw3, float* dw4, float* dw5, float* dw6, float* dw7,
                             float* db0, float* db1, float* db2, float* db3, float* db4, float* db5, float* db6, float* db7, int M, int N, int K, void* stream);
/* dw[N,K] (caller-zeroed) += g[M,N]^T . x[M,K];  N % 64 == 0, K % 64 == 0.  db (may be NULL; caller-zeroed,
 * needs ldg == N) += column sums of g — the bias gradient of the same nn.Linear, summed inside the launch */
int stcat_linear_wgrad(const float* g, const float* x, float* dw, float* db, int M, int N, int K, int ldg, int ldx,
                       void* stream);
/* narrow heads, N <= 16, K % 4 == 0 */
int stcat_small_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K,
                           void* stream);
int stcat_small_linear_bwd(const float* g, const float* x, const float* w, float* dx, float* dw, float* db, int M,
                           int N, int K, void* stream);
/* out[N] (caller-zeroed) += sum_m a[m,n] * (b ? b[m,n] : 1) */
int stcat_colsum(const float* a, const float* b, float* out, int M, int N, void* stream);
/* LayerNorm over 256 features of (res + dropout_p(x)); saves mean / rstd per row.  The (drop_p, drop_seed,
 * drop_offset, drop_base) quadruple is the dropout of the residual branch (see "dropout" below; drop_p = 0: plain
 * x + res).  Backward: dz = gradient of res (and of x when drop_p = 0), dx (may be NULL when drop_p = 0) = mask * dz. */
int stcat_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                        float* mean, float* rstd, int M, int D, float eps, float drop_p, long drop_seed,
                        long drop_offset, const long* drop_base, void* stream);
int stcat_layernorm_bwd(const float* dy, const float* x, const float* res, const float* gamma, const float* mean,
                        const float* rstd, float* dz, float* dx, float* dgamma, float* dbeta, int M, int D,
                        float drop_p, long drop_seed, long drop_offset, const long* drop_base, void* stream);
/* element-wise glue; op codes STCAT_EW_* below; b indexed modulo bmod */
int stcat_ew(int op, const float* a, const float* b, const float* c, float* out, long n, long bmod, float alpha,
             float beta, void* stream);
/* row-strided two-operand form (ADD, MUL, AXPBY, COPY): out[r*ldo + c] = op(a[r*lda + c], b[r*ldb + c]); column blocks of
 * wider matrices (a decoder layer's slice of the layer-batched key/value projections, query_decoder.py:355-358; the
 * first half of the anchor sine embedding, query_decoder.py:193-200) without a gather copy.  lda / ldb == 0 broadcasts
 * one row (the [CLS] embeddings prepended to every frame's tokens, modal_encoder.py:145-151) */
int stcat_ew2d(int op, const float* a, long lda, const float* b, long ldb, float* out, long ldo, long rows, int cols,
               float alpha, float beta, void* stream);
enum {
  STCAT_EW_ADD = 0, STCAT_EW_MUL = 1, STCAT_EW_SIGMOID = 2, STCAT_EW_TANH = 3, STCAT_EW_RELU = 4,
  STCAT_EW_INVSIG = 5, STCAT_EW_SIGMOID_BWD = 6, STCAT_EW_TANH_BWD = 7, STCAT_EW_INVSIG_BWD = 8,
  STCAT_EW_ADD3 = 9, STCAT_EW_AXPBY = 10, STCAT_EW_COPY = 11
};

/* ---- dropout (train mode) ----------------------------------------------------------------------- */
/* y = res + dropout_p(x)  (res may be NULL): nn.Dropout at modal_encoder.py:237-240, query_decoder.py:344,
 * 431-436, 612, 653-658 and net_utils.py:24-25.  Decisions are a pure function of (seed, offset + element
 * index) — splitmix64 finaliser, see csrc/stcat_rng.h — so the same call on dY is the backward pass and no mask
 * is stored.  The attention entry points below take the same (drop_p, drop_seed, drop_offset) triple for the
 * dropout nn.MultiheadAttention / attention.py:381 applies to the softmax probabilities; drop_p = 0 disables it
 * (eval mode, and every parity test against the golden vectors).  base / drop_base (may be NULL) is a DEVICE
 * int64 added to the offset when the kernel starts: the host advances it once per step with a device-side add, so
 * a captured hipGraph replaying identical launch arguments still draws new masks each step. */
int stcat_dropout(const float* x, const float* res, float* y, long n, float p, long seed, long offset,
                  const long* base, void* stream);

/* ---- attention ---------------------------------------------------------------------------------- */
/* torch.nn.MultiheadAttention core after the in-projection (modal_encoder.py:236; query_decoder.py:341,
 * 604-610): per (batch, head) softmax(scale * q k^T + key_padding) v with head dim 32, S <= 256.
 * q/k/v/o are [B,S,ld*] with head h at column h*32.  pt receives the probabilities as
 * [B,H,Sp,Sp] (Sp = 32*ceil(S/32), key-major) for the backward pass / head-mean weights; pt = NULL skips the
 * 103 MB-per-layer store (inference: nothing reads it). */
int stcat_mha_self_fwd(const float* q, const float* k, const float* v, const unsigned char* kpm, float* o,
                       float* pt, int B, int H, int S, int ldq, int ldk, int ldv, int ldo, float scale,
                       float drop_p, long drop_seed, long drop_offset, const long* drop_base, void* stream);
/* out = forward output; dw (may be NULL) = gradient of the head-averaged weights [B,S,S] and then
 * corr = [B,H,S] scratch; dst = [B,H,Sp,Sp] scratch; dq/dk are [B,S,ldg], dv is [B,S,ldgv] */
int stcat_mha_self_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout,
                       const float* pt, const float* dw, float* corr, float* dst, float* dq, float* dk, float* dv,
                       int B, int H, int S, int ldq, int ldk, int ldv, int ldo, int ldg, int ldgv, float scale,
                       float drop_p, long drop_seed, long drop_offset, const long* drop_base, void* stream);
/* The recomputing pair (modal_encoder.py:236, query_decoder.py:341 — nn.MultiheadAttention's core when nobody reads the
 * head-mean weights): the forward keeps lse [B][H][Sp][2] = (row maximum, 1 / row sum), Sp = S rounded up to 32, instead of
 * the S x S probabilities; the backward rebuilds its probability tiles from q, k and lse.  S <= 256; kpm / drop_* as above. */
int stcat_mha_self_fwd_lse(const float* q, const float* k, const float* v, const unsigned char* kpm, float* o, float* lse,
                           int B, int H, int S, int ldq, int ldk, int ldv, int ldo, float scale, float drop_p,
                           long drop_seed, long drop_offset, const long* drop_base, void* stream);
int stcat_mha_self_bwd_lse(const float* q, const float* k, const float* v, const unsigned char* kpm, const float* out,
                           const float* dout, const float* lse, float* dq, float* dk, float* dv, int B, int H, int S,
                           int ldq, int ldk, int ldv, int ldo, int ldg, int ldgv, float scale, float drop_p,
                           long drop_seed, long drop_offset, const long* drop_base, void* stream);
/* The same core on the bf16 matrix pipe (fp32 accumulate) with an online softmax: any S, nothing but the row
 * log-sum-exp lse[B,H,S] kept for backward (no probability stash); no head-mean weights.  The arithmetic follows the
 * mma mode: two bf16 planes per operand and three products (modes bf16x3 / bf16x6 / bf16x3p), or — mode bf16x6p,
 * round 6 — THREE planes per operand (their sum is the fp32 value) and the six cross terms of the plane GEMMs, also
 * for the probabilities and dS that feed the second contraction.  stcat_mha_bs_bwd recomputes the probabilities from
 * q, k and lse (S <= 256): ONE launch for two planes; for three planes two launches (dQ with K / V planes in LDS, dK / dV
 * with Q / dO planes in LDS: all four as three planes would be 172 KB at S = 224).
 * Replaces nn.MultiheadAttention's core in modal_encoder.py:161-168, 180-185, 228-242 and query_decoder.py:341. */
int stcat_mha_bs_fwd(const float* q, const float* k, const float* v, const unsigned char* kpm, float* o, float* lse,
                     int B, int H, int S, int ldq, int ldk, int ldv, int ldo, float scale, float drop_p, long drop_seed,
                     long drop_offset, const long* drop_base, void* stream);
int stcat_mha_bs_bwd(const float* q, const float* k, const float* v, const unsigned char* kpm, const float* out,
                     const float* dout, const float* lse, float* dq, float* dk, float* dv, int B, int H, int S, int ldq,
                     int ldk, int ldv, int ldo, int ldg, int ldgv, float scale, float drop_p, long drop_seed,
                     long drop_offset, const long* drop_base, void* stream);
/* head-averaged weights [B,S,S] (need_weights=True; consumed at pipeline.py:84-85); in train mode these are
 * the DROPPED probabilities, as torch returns them */
int stcat_attn_weights_mean(const float* pt, float* w, int B, int H, int S, float drop_p, long drop_seed,
                            long drop_offset, const long* drop_base, void* stream);
/* time-aligned cross-attention with ONE query per frame (query_decoder.py:386-417 via
 * grounding_model/attention.py:184-393; query_decoder.py:618-639): per (frame, head)
 * softmax(scale*(q1.k1 + q2.k2)) v, each part 32 wide; q2/k2 may be NULL.  P [B,H,S] is kept for backward. */
int stcat_attn_q1_fwd(const float* q1, const float* q2, const float* k1, const float* k2, const float* v,
                      const unsigned char* kpm, float* out, float* P, int B, int H, int S, int ldq, int ldk,
                      int ldv, float scale, float drop_p, long drop_seed, long drop_offset, const long* drop_base, void* stream);
int stcat_attn_q1_bwd(const float* q1, const float* q2, const float* k1, const float* k2, const float* v,
                      const float* P, const float* dout, float* dq1, float* dq2, float* dk1, float* dk2, float* dv,
                      int B, int H, int S, int ldq, int ldk, int ldv, float scale, float drop_p, long drop_seed,
                      long drop_offset, const long* drop_base, void* stream);

/* ---- 2D temporal map head (models/map2d_head.py: Gen2DMap :9-62, TempConvInteraction :228-250) — optional op,
 *      forward only (the reference defines no loss for it) -------------------------------------------------------- */
/* adaptive pooling of x [b,T,D] to N steps: T > N: adaptive_avg_pool1d, else adaptive_max_pool1d (:52-55) */
int stcat_map2d_pool(const float* x, float* pooled, int b, int T, int N, int D, void* stream);
/* the 39 cascaded MaxPool1d layers written on sparse diagonals (:57-61) in closed form: cell c = (cell_i[c], cell_j[c])
 * of the caller-zeroed NHWC map [b,N,N,D] = max over pooled[b, cell_i .. cell_j, :] */
int stcat_map2d_cells(const float* pooled, const int* cell_i, const int* cell_j, int ncells, float* map, int b, int N, int D,
                      void* stream);
/* backward of the two launches above (train mode returns raw scores, map2d_head.py:122-124): a cell's gradient goes to
 * the first maximum of its range [i, j] (where the reference's cascade of MaxPool1d backward passes routes it), then
 * through the adaptive pooling (:48-51); dpooled [b,N,D] and dx [b,T,D] are zeroed by the caller (atomic accumulation) */
int stcat_map2d_cells_bwd(const float* pooled, const int* cell_i, const int* cell_j, int ncells, const float* dmap,
                          float* dpooled, int b, int N, int D, void* stream);
int stcat_map2d_pool_bwd(const float* x, const float* dpooled, float* dx, int b, int T, int N, int D, void* stream);
/* y[m, :] *= w[m % period]: the per-pixel mask-normalisation weight after each conv + ReLU (:247-249) */
int stcat_rowscale(float* y, const float* w, long rows, int C, int period, void* stream);

/* ---- VideoSTGLoss (models/criterion.py:11-208), all decoder layers in one launch --------------------------------
 * vec[k*nl + l] = un-weighted loss k of decoder layer l; k = 0 loss_bbox (criterion.py:38-55), 1 loss_giou (:56-66),
 * 2 loss_sted (:68-124), 3 loss_guided_attn (:126-145), 4 loss_actioness (:147-158; 0 when act == NULL).
 * boxes [nl][rows_total][4] (cx,cy,w,h); rows [nbox] = the GT-span rows (:168-171); tgt [nbox][4]; sted [nl][b][T][2];
 * dist [b][T][2] the normalised Gaussian span targets; time_mask / pos_or_pad [b][T] bytes; w [nl][b][T][T];
 * nb_neg [b]; act [nl][b][T] with act_tgt / act_w [b][T].  num_boxes: host value, or a device scalar (the
 * data-parallel mean box count, :175-178) when num_boxes_dev != NULL.  wmat [5][nl] (may be NULL): weight_dict laid
 * out like vec; total (may be NULL, caller-zeroed) += sum wmat * vec. */
int stcat_stg_loss_fwd(const float* boxes, const long* rows, const float* tgt, const float* sted, const float* dist,
                       const unsigned char* time_mask, const float* w, const unsigned char* pos_or_pad,
                       const float* nb_neg, const float* act, const float* act_tgt, const float* act_w,
                       const float* num_boxes_dev, float num_boxes, int nl, int rows_total, int nbox, int b, int T,
                       const float* wmat, float* vec, float* total, void* stream);
/* gradients of sum_k,l (gvec[k][l] + gtotal * wmat[k][l]) * vec[k][l] w.r.t. boxes / sted / w / act (same shapes; the
 * rows of d_boxes outside the GT span are written as zeros).  gvec or gtotal may be NULL. */
int stcat_stg_loss_bwd(const float* boxes, const long* rows, const float* tgt, const float* sted, const float* dist,
                       const unsigned char* time_mask, const float* w, const unsigned char* pos_or_pad,
                       const float* nb_neg, const float* act, const float* act_tgt, const float* act_w,
                       const float* num_boxes_dev, float num_boxes, int nl, int rows_total, int nbox, int b, int T,
                       const float* wmat, const float* gvec, const float* gtotal, float* d_boxes, float* d_sted,
                       float* d_w, float* d_act, void* stream);

/* ---- optimizer tail (scripts/train_net.py:134-143) ------------------------------------------------ */
/* Multi-tensor launches over a DEVICE table of entries
 *   { float* p; const float* g; float* m; float* v; float* ema (may be NULL); long n; int group; int pad; }
 * (stcat_optim_table_entry_bytes() == 56) cut into chunks: chunk c covers elements
 * [chunk_off[c], chunk_off[c] + chunk) of table[chunk_tensor[c]]; chunk % 4 == 0.
 * stcat_grad_sqnorm: *out_sq = sum over all tensors of |g|^2 (torch.nn.utils.clip_grad_norm_, first half).
 * stcat_adamw_ema_step: clip coefficient min(1, max_norm / (sqrt(*sqnorm) + 1e-6)) (max_norm <= 0: none),
 *   torch.optim.AdamW update (engine/optimizer.py:25-55; lr / wd are HOST arrays indexed by entry.group,
 *   rewritten by the schedule of engine/lr_scheduler.py:212-252 every step) and the EMA copy
 *   w_ema = w_ema * decay + (1 - decay) * w (engine/optimizer.py:5-22) in one pass.  step counts from 1.
 * stcat_grad_clip_scale: second half of a standalone torch.nn.utils.clip_grad_norm_ (scripts/train_net.py:136-137):
 *   every gradient of the table is scaled in place by max_norm / (sqrt(*sqnorm) + 1e-6) when that is < 1; the
 *   buffers are walked in flat memory order, so row-major and channels_last gradients are treated alike.
 * stcat_ema_update: the EMA alone, for state that is not a trained parameter. */
int stcat_optim_table_entry_bytes(void);
int stcat_grad_sqnorm(const void* table, const int* chunk_tensor, const long* chunk_off, int n_chunks, int chunk,
                      float* out_sq, void* stream);
int stcat_adamw_ema_step(const void* table, const int* chunk_tensor, const long* chunk_off, int n_chunks, int chunk,
                         const float* sqnorm, const float* lr, const float* wd, int n_groups, float beta1,
                         float beta2, float eps, int step, float max_norm, float ema_decay, void* stream);
int stcat_grad_clip_scale(const void* table, const int* chunk_tensor, const long* chunk_off, int n_chunks, int chunk,
                          const float* sqnorm, float max_norm, void* stream);
int stcat_ema_update(const void* table, const int* chunk_tensor, const long* chunk_off, int n_chunks, int chunk,
                     float decay, void* stream);

/* ---- live 2D temporal map (models/post_processor.py:30-53) -------------------------------------- */
/* sted [b,T,2], durations [b] (device int32) -> out [b,2] device int32 (start_idx, end_idx), T <= 1024 */
int stcat_temporal_map_argmax(const float* sted, const int* durations, int* out, int b, int T, void* stream);

/* ---- launch plans: the host side of a composite node of the path — models/vision_model/backbone.py:93-121 (the
 *      backbone forward / backward), grounding_model/modal_encoder.py:104-204 (encoder), grounding_model/
 *      query_decoder.py:150-247 and :478-550 (box / time decoder), models/pipeline.py:88-103 (heads), called in the
 *      order of models/pipeline.py:52-121 — as ONE call.  The launch sequence (every entry point above that takes a
 *      stream) is recorded once per input shape and replayed by stcat_plan_run(): one hipLaunchKernel per op instead
 *      of one Python -> FFI round trip per op.  Argument words: one 64-bit word per argument (pointers and integers
 *      as is, floats as their IEEE bits in the low half); the stream argument is filled in from `streams[slot]` at
 *      replay.  Relocations patch the pointers that lie inside the caller's per-step tensors ("externals"). -------- */
int stcat_plan_fn_index(const char* entry_point_name);                  /* -1: not a launch entry point */
int stcat_plan_fn_nargs(int fn);
void* stcat_plan_create(void);
int stcat_plan_destroy(void* plan);
/* returns the index of the call's first argument word (>= 0) or < 0 */
int stcat_plan_add_call(void* plan, int fn, const unsigned long long* words, int nargs, int stream_slot, int stream_arg);
/* everything issued later on `waiter_slot` waits for what is queued on `signal_slot` at this point of the replay */
int stcat_plan_add_wait(void* plan, int waiter_slot, int signal_slot);
/* zero `bytes` at ptr (the plan's accumulation buffers) at this point of the sequence (at_front: before every other
 * op); returns the pointer's word, the byte count is the word after it */
int stcat_plan_add_memset(void* plan, void* ptr, unsigned long long bytes, int stream_slot, int at_front);
/* overwrite one argument word (the byte count of a memset whose extent is only known when the recording ends) */
int stcat_plan_set_word(void* plan, int word, unsigned long long value);
/* the replay returns to the host here (tag identifies the host-side action), to be resumed at *next */
int stcat_plan_add_yield(void* plan, int tag);
int stcat_plan_add_reloc(void* plan, int word, int external, unsigned long long byte_offset);
int stcat_plan_size(void* plan, int* n_ops, int* n_words, int* n_relocs);
int stcat_plan_run(void* plan, const unsigned long long* externals, int n_externals, void* const* streams, int n_streams,
                   int start, int* next, int* tag);

#ifdef __cplusplus
}
#endif
#endif /* STCAT_HIP_H_ */
