/*
 * cmk.h — C ABI of libcmk_hip.so, the MI355X (gfx950) kernels under the CenterMask2 inference path.
 *
 * The reference (Zeng-Yan/centermask2) is pure Python; every arithmetic step of its path reaches native code
 * through PyTorch / detectron2 / torchvision operators.  Each entry point below replaces one such operator
 * call site (cited per function, paths relative to the reference root).  Conventions:
 *   - the caller owns every buffer, including workspaces; the library never allocates, frees or synchronises;
 *   - all pointers are device pointers; all launches go to the `stream` argument (a hipStream_t passed as void*);
 *   - activations are NHWC float32; a tensor "view" is (pointer, channel stride `cs`, channel offset `co`), so a
 *     conv can write straight into a slice of an OSA concat buffer (vovnet.py:324 never materialises);
 *   - returns 0, or a negative CMK_E* code; cmk_last_error() gives the message (thread-local); nothing throws;
 *   - no state beyond one-time, per-device kernel attributes (set on the first launch on each device, idempotent); one process
 *     per GPU each loads its own copy.  Launches go to the CURRENT device's stream the caller passes: the caller selects the
 *     device (hipSetDevice / torch.cuda.device) that owns the pointers before calling.
 */
#ifndef CMK_H
#define CMK_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CMK_OK 0
#define CMK_EINVAL (-1)   /* bad argument / unsupported shape */
#define CMK_ELAUNCH (-2)  /* HIP launch error */

int cmk_version(void);                 /* ABI version, currently 5 (2: cmk_conv_desc.w_wino6, cmk_groupnorm_affine_tiles takes record counts;
                                          3: cmk_conv_desc.pool_ws, cmk_ese_gate_pooled, cmk_pack_records; 4: cmk_conv_desc.w_split;
                                          5: cmk_conv_desc.w_splith, w_splith_scale) */
const char* cmk_arch(void);            /* "gfx950" */
const char* cmk_last_error(void);

/* ---- convolution as implicit GEMM on v_mfma_f32_32x32x2_f32 -------------------------------------------------
 * Replaces aten::conv2d (+ FrozenBN fold, bias, ReLU) at vovnet.py:205-236, fpn.py:27-35, d2 FPN lateral/output
 * convs, fcos.py:169-200, sam.py:58-83, maskiou_head.py:81-93 and the Linear layers maskiou_head.py:89-91.
 * Weights are pre-packed by cmk_pack_conv_weight_size/the host packer into [tap][Cin/16][cout_pad][16].
 * y = act( conv(x) * scale[c] + shift[c] (+ residual) ),  act = ReLU on channels [0, relu_upto).              */
typedef struct {
    const float* x; int x_cs; int x_co;         /* input view, (N,H,W,Cin) */
    const float* w;                             /* packed weights */
    const float* scale; const float* shift;     /* per output channel, length >= Cout */
    const float* res; int res_cs; int res_co;   /* optional residual view */
    int res_mode;                               /* 0 none, 1 same resolution, 2 nearest-upsample x2 of (N,Hr,Wr,*) */
    int Hr; int Wr;
    float* y; int y_cs; int y_co;               /* output view, (N,Ho,Wo,Cout) */
    int N; int H; int W; int Cin; int Cout;
    int ksize;                                  /* 1 or 3 (padding ksize/2) */
    int stride;                                 /* 1 or 2 */
    int relu_upto;                              /* ReLU applied to output channels < relu_upto */
    int in_relu;                                /* ReLU applied to the input while staging (fpn.py:34) */
    /* optional tile variant picked by the caller's autotuner (all 0 = let the library's cost model choose):
     * tune_wm in {1,2} (128 or 256 pixels per workgroup), tune_sc in {16,32} (sub-tile 2x16 or 1x32 pixels; 32 for 1x1),
     * tune_wn in 1..7 (32*tune_wn output channels per workgroup; must divide the padded Cout).  Results are bitwise
     * identical across variants (the K order per output does not depend on the tile).
     * tune_wm == 7 (with tune_wn in {1,2,4}) selects the gather form of a 3x3 conv (stride 1|2): a flattened-pixel GEMM whose K walks
     * 9 taps x Cin/16 chunks, each A row gathered per tap — for maps too small to fill the spatial tiles (the 14->7 maskiou conv
     * maskiou_head.py:84, P6/P7 fpn.py:32-35); same K order, so again bitwise identical to the tiled variants.
     * tune_wm == 8 (with tune_wn in {4,2}) selects the pointwise GEMM kernel (conv_pw.hip) for a 1x1 conv with Cout > 224, Cin % 32 == 0,
     * no fused input affine / input ReLU (the upsampled residual only with an even W; split-K with K chunks % (2*splitk) == 0): 64*tune_wn pixels x 128 output channels per workgroup, weights
     * fetched straight into registers (the packed layout is the same); again the same bits as the tiled variants.  It is also the
     * untuned default for such convs when they make at least 256 workgroups (vovnet.py:222-236 aggregation convs, the deconv).
     * tune_wm == 9 (tune_wn in {4,2}): the gather form of a 3x3 conv (stride 1|2) on that kernel — K walks 9 taps x Cin/16 chunks, rows
     * gathered per tap with bounds-checked loads; Cout in 97..128 or > 224, Cin % 32 == 0, one problem, no fused affine / upsampled
     * residual; bitwise identical to the tune_wm 7 form; the untuned default for stride-2 convs of at least 1024 workgroups
     * of 256 pixels (stem_3 vovnet.py:412). */
    int tune_wm; int tune_sc; int tune_wn;
    /* tune_wm == 5 selects the fused Winograd F(2x2,3x3) kernel (3x3 stride 1, no residual; the default for such convs when w_wino
     * is given): same fp32 arithmetic on the matrix pipe with 2.25x fewer multiplies; results differ from the direct kernel by fp32
     * rounding only.  It needs the weights pre-transformed to U = G g G^T (cmk_wino_packed_floats floats), packed
     * [Cin/16][ceil(Cout/64)][step 4][fh 2][ng 2][fl 2][piece 2][lane 64][4 floats]: frequency (row-major index of the 4x4 grid) =
     * 8*fh + 2*step + fl, output channel = ntile*64 + ng*32 + (lane & 31), input channel = chunk*16 + 8*(lane >> 5) + 4*piece + j —
     * every operand load of a wave is one contiguous KiB. */
    const float* w_wino;
    /* optional fused GroupNorm+ReLU of the PRODUCER (fcos.py:182-186): per (image, input channel) x' = relu(x*in_scale + in_shift)
     * is applied while the input tile is staged, so the normalised tensor is never written; arrays of N*Cin floats from
     * cmk_groupnorm_affine.  Supported by the direct kernels and the Winograd kernel. */
    const float* in_scale; const float* in_shift;
    /* split-K (direct kernels incl. the gather form, one problem, res_mode 0|1): splitk >= 2 workgroup rows each own 1/splitk of the
     * 16-channel K chunks (K chunks % (2*splitk) == 0) and write raw partial sums to splitk_ws (splitk * N*Ho*Wo * cmk_conv_cout_pad(Cout)
     * floats, caller-owned); a second launch sums them and applies the epilogue.  For skinny GEMMs (maskiou_fc1: 400 x 12544 x 1024,
     * maskiou_head.py:116) whose M x N tiles cannot fill 256 CUs.  0/1 = off. */
    int splitk; float* splitk_ws;
    /* optional fused GroupNorm STATISTICS of this conv's output (the GroupNorm that follows it, fcos.py:182-186): only with
     * tune_wm == 5, relu_upto == 0 and Cout/gn_groups a power of two <= 32.  The kernel writes {sum, sum of squares} per
     * (spatial tile of 8x16 outputs, row parity, group) to gn_ws as doubles: record index ((tile*2 + parity)*gn_groups + group),
     * tiles numbered image-major per problem (for _multi: problems back to back) with cmk_conv_gn_tiles(H, W) tiles per image;
     * cmk_groupnorm_affine_tiles turns them into the per-(image, channel) scale/shift.  NULL = off.
     * With tune_wm == 6 the records are per (spatial tile of 12x40 outputs, wave 0..3, group): index ((tile*4 + wave)*gn_groups + group);
     * cmk_conv_gn_records(H, W, tune_wm) gives the records per image of either form. */
    double* gn_ws; int gn_groups;
    /* tune_wm == 6 selects the fused Winograd F(4x4,3x3) kernel (conv_wino6.hip; 3x3 stride 1, no residual, Cin % 8 == 0; tune_wn 1 = 12x40-pixel
     * tiles of one image, tune_wn 2 = two whole maps of at most 16 rows x 14 columns per workgroup, for the 14x14 RoI features): 36 multiplies
     * per 4x4 outputs, 1.78x fewer than F(2x2,3x3); fp32 throughout, error ~1.6x the 2x2 form's (tools/wino_numerics.py).  Needs
     * U = G g G^T (6x6 per filter, points 0, +-1, +-2, inf), cmk_wino6_packed_floats floats packed
     * [Cin/8][ceil(Cout/32)][wave 4][slot 9][lane 64][4 floats]: slot k < 6 is frequency (row = wave, column = k), slot k >= 6 is
     * (row = 4 + wave/2, column = 3*(wave%2) + k - 6); output channel = tile*32 + (lane & 31), input channel = chunk*8 + 4*(lane >> 5) + j.
     * tune_sc == 64 with tune_wm == 6 selects the "shared V" form of the same arithmetic (conv_wino6s.hip; tune_sc 16 or 0 = conv_wino6.hip): one
     * 8-wave workgroup per CU computes 64 output channels of a spatial tile from ONE frequency image of the input kept in LDS, so the halo is
     * fetched and transformed once per 64 channels instead of once per 32; same packed weights, same K order: bit-identical results.  It is
     * the faster form for launches of about one round of workgroups (the 50x80 maps of stage 4, vovnet.py:90-98); the start-up tuner decides. */
    const float* w_wino6;
    /* optional fused average-pool partial sums of the (scaled, shifted, ReLU'd) OUTPUT, for the eSE gate of the OSA aggregation conv
     * (vovnet.py:255-256 avg_pool over the conv the block just produced): only the pointwise GEMM kernel produces them — ask
     * cmk_conv_pool_rows(d) first; it returns R > 0 (rows of flattened pixels per record) when this descriptor, as tuned, runs on that
     * kernel and H*W >= R, else 0.  pool_ws then receives 2 * ceil(N*H*W / R) records of Cout floats: record 2g = the sum over the rows
     * of block g (pixels gR .. gR+R-1) that belong to the image of the block's first pixel, record 2g+1 = the sum over its rows that
     * belong to the next image (0 if none); every record is written.  cmk_ese_gate_pooled turns them into the gate.  NULL = off. */
    float* pool_ws;
    /* OPT-IN, tune_wm == 10 (tune_sc 32, tune_wn 4): a plain 1x1 conv (Cin % 16 == 0, Cout > 224 or 97..128, no residual / input affine / split-K;
     * pool_ws allowed) as the same GEMM with every fp32 product rebuilt from bf16 pieces on v_mfma_f32_32x32x16_bf16: x = hi + mid + lo with
     * hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (round to nearest even; exact to 2^-24), the six products of weight >= 2^-16
     * accumulated in fp32.  The result carries the error of an fp32 accumulation (measured: that of a sequential fp32 fma chain), NOT the bits
     * of the fp32-MFMA kernels.  The activations are fp32 and are split inside the kernel; the weights are split by the caller:
     * w_split = cmk_split_packed_halves(Cout, Cin) 16-bit values, [Cin/16][cout_pad/32][piece hi|mid|lo][lane 64][8], lane = 32*hh + li holding
     * input channels 16*chunk + 8*hh + 0..7 of output channel 32*tile + li (cout_pad = Cout rounded up to 128, zero filled).
     * No caller of this repository selects it by default (ops.ALLOW_SPLIT_BF16); NULL = not available. */
    const void* w_split;
    /* (tune_wm 11 also takes tune_sc 21: the same geometries with one cout tile per wave — workgroups of 64 / 128 couts: less cout padding for
     * layers of 160 / 192 couts, more and smaller workgroups for the small maps.)
     * OPT-IN, tune_wm 12 (tune_sc 32, tune_wn 4): the pointwise GEMM / gather form of tune_wm 10 on TWO fp16 pieces per operand, three products
     * (half the MFMAs of the bf16 form, the same fp32-class error); takes pool_ws and res_mode 2 like tune_wm 10; packing w_splith (1 tap for a
     * 1x1 conv, 9 tap-major for a 3x3 conv) and w_splith_scale as described next.
     * OPT-IN, tune_wm 11 (conv_sp3.hip; tune_sc = 2 pieces, tune_wn = tile geometry 0..3): a 3x3 stride-1 conv as a DIRECT implicit GEMM on
     * v_mfma_f32_32x32x16_f16 with every fp32 operand split into TWO fp16 pieces (22 bits of significand: h = fp16(x), m = fp16(x - h), the
     * residual is exact) and the products m*h, h*m, h*h accumulated in fp32.  The representation error is below an fp32 GEMM's own accumulation
     * error, so the result carries the error of an fp32 accumulation — NOT the bits of the fp32-MFMA kernels.  Activations are split inside the
     * kernel (scaled by 2^-4, residual by 2^11: finite up to |x| = 1e6, 22 bits down to 2^-21); the caller packs the weights:
     * w' = w * S_w, S_w the power of two with max |w'| in [2^14, 2^15); w_splith = taps x cmk_splith_packed_halves(Cout, Cin) fp16 values,
     * [tap][Cin/16][cout_pad/32][piece h|m][lane 64][8] (lane = 32*hh + li: input channels 16*chunk + 8*hh + 0..7 of output channel 32*tile + li;
     * cout_pad = Cout rounded up to 128, zero filled); w_splith_scale = 1 / S_w.  Takes in_scale/in_shift, gn_ws (cmk_conv_gn_records(H, W,
     * 110 + geometry)), and in cmk_conv2d_nhwc_multi up to 10 problems that may differ in their weights.  No residual, split-K or pooled sums.
     * No caller of this repository selects it by default (ops.ALLOW_SPLIT_F16); NULL = not available. */
    const void* w_splith;
    float w_splith_scale;
} cmk_conv_desc;
int cmk_conv2d_nhwc(const cmk_conv_desc* d, void* stream);
int cmk_conv_pool_rows(const cmk_conv_desc* d);
/* Same conv applied to up to 5 inputs of different H x W in ONE launch (the FCOS towers/predictors share their weights
 * across the FPN levels, fcos.py:227-238).  All descriptors must share w, Cin, Cout, ksize, stride, views and flags and
 * carry no residual; x, y, N, H, W, scale, shift, in_scale/in_shift may differ.
 * With tune_wm == 6, tune_wn == 1 (the F(4x4) map kernels, which take the packed weights per problem) the problems may also differ in their
 * weights (w, w_wino6) and there may be up to 10 of them: conv k of the FCOS head's cls tower and of its bbox tower — same level shapes,
 * different weights, fcos.py:227-231 — run as one launch of 2 x 5 problems.  Fused GroupNorm records (gn_ws) are numbered over the spatial
 * tiles of all problems in order, so the records of problems 5..9 follow those of problems 0..4. */
int cmk_conv2d_nhwc_multi(const cmk_conv_desc* descs, int n, void* stream);
/* number of floats of the packed layout for (Cout, Cin, k): taps * ceil(Cin/16) * cout_pad * 16 */
int64_t cmk_conv_packed_floats(int Cout, int Cin, int ksize);
int cmk_conv_cout_pad(int Cout);
int64_t cmk_wino_packed_floats(int Cout, int Cin);
int64_t cmk_wino6_packed_floats(int Cout, int Cin);
int64_t cmk_split_packed_halves(int Cout, int Cin);          /* 16-bit elements of cmk_conv_desc.w_split */
int64_t cmk_splith_packed_halves(int Cout, int Cin);         /* 16-bit elements PER TAP of cmk_conv_desc.w_splith */
/* spatial tiles per image of the fused-statistics conv (8 x 16 outputs each) */
int cmk_conv_gn_tiles(int H, int W);
/* {sum, sumsq} records per image written through cmk_conv_desc.gn_ws by the Winograd kernel tune_wm (5 or 6) on an H x W map */
int cmk_conv_gn_records(int H, int W, int tune_wm);

/* ---- depth-wise 3x3, pad 1, stride 1|2, no bias / norm / activation (vovnet.py:110-119 'dw_conv3x3', the dw half of the
 * depth-wise VoVNet bodies V-19-slim-dw-eSE / V-19-dw-eSE vovnet.py:30-48).  x, y are NHWC channel-slice views
 * (pixel stride *_cs, offset *_co floats); w is tap-major [9][C]. ------------------------------------------------ */
int cmk_dwconv3x3_nhwc(const float* x, int x_cs, int x_co, const float* w, float* y, int y_cs, int y_co,
                       int N, int H, int W, int C, int stride, void* stream);

/* ---- stem_1: 3x3 stride-2 conv on the NCHW 3-channel image (vovnet.py:409), BN-folded, ReLU, NHWC out -------- */
int cmk_stem_conv_nchw3(const float* x, const float* w /* [27][Cout] */, const float* scale, const float* shift,
                        float* y, int N, int H, int W, int Cout, void* stream);

/* ---- max pool k3 s2 ceil_mode, no padding (vovnet.py:349-350); k2 s2 (maskiou_head.py:93,108) ----------------- */
/* gate: optional (N*C) non-negative channel gate applied after the max (the eSE scale of the producer block, folded in). */
int cmk_maxpool3x3s2_ceil_nhwc(const float* x, int x_cs, int x_co, float* y, int y_cs, int y_co,
                               int N, int H, int W, int C, const float* gate, void* stream);

/* MaxPool2d(kernel 1, stride 2): every second pixel — d2's LastLevelMaxPool, the top block build_vovnet_fpn_backbone (vovnet.py:504-524)
 * hands to the FPN; Ho = (H-1)/2 + 1, Wo = (W-1)/2 + 1. */
int cmk_maxpool1x1s2_nhwc(const float* x, int x_cs, int x_co, float* y, int y_cs, int y_co, int N, int H, int W, int C, void* stream);

/* ---- eSE (vovnet.py:247-260): gate = relu6(W * mean_HW(x) + b + 3) / 6 ; y = x * gate (+ identity) ------------
 * ws: N * ese_chunks * C floats of workspace for the two-stage mean.                                        */
int cmk_ese_gate(const float* x, int x_cs, int x_co, const float* fc_w /* [C][C] row = out */, const float* fc_b,
                 float* gate /* N*C */, float* ws, int ws_chunks, int N, int HW, int C, void* stream);
int cmk_ese_scale(const float* x, int x_cs, int x_co, const float* gate, const float* identity, int id_cs, int id_co,
                  float* y, int y_cs, int y_co, int N, int HW, int C, void* stream);
/* the gate from the partial sums the producing conv left in pool_ws (cmk_conv_desc.pool_ws; rows = cmk_conv_pool_rows of that conv):
 * no pass over the map.  Fixed summation order. */
int cmk_ese_gate_pooled(const float* pool_ws, int rows, const float* fc_w, const float* fc_b, float* gate, int N, int HW, int C, void* stream);

/* ---- GroupNorm(32) + ReLU in place (fcos.py:182-186) ---------------------------------------------------------
 * ws: N * groups * gn_chunks * 2 doubles.                                                                    */
int cmk_groupnorm_relu_nhwc(float* x, const float* gamma, const float* beta, double* ws, int ws_chunks,
                            int N, int HW, int C, int groups, float eps, void* stream);

/* The same without the ReLU (d2 get_norm("GN") behind a conv with no activation: MODEL.FPN.NORM "GN", vovnet.py:550) */
int cmk_groupnorm_nhwc(float* x, const float* gamma, const float* beta, double* ws, int ws_chunks,
                       int N, int HW, int C, int groups, float eps, void* stream);
/* y (N,H,W,C dense) += nearest-neighbour 2x upsampling of coarse (N,Hc,Wc,C dense): d2 FPN's top-down sum when a norm sits between the
 * lateral conv and the sum (otherwise the sum rides in the lateral conv's epilogue, cmk_conv_desc.res_mode 2) */
int cmk_upsample2x_add_nhwc(float* y, const float* coarse, int N, int H, int W, int Hc, int Wc, int C, void* stream);

/* Statistics only: out_scale/out_shift (N*C each) such that GroupNorm(x)[n,:,c] = x*out_scale[n,c] + out_shift[n,c];
 * the consumer conv applies them (cmk_conv_desc.in_scale/in_shift).  ws as above. */
int cmk_groupnorm_affine(const float* x, const float* gamma, const float* beta, double* ws, int ws_chunks,
                         int N, int HW, int C, int groups, float eps, float* out_scale, float* out_shift, void* stream);

/* The same for up to 5 tensors (the FPN levels of one tower conv) in two launches; xs/out_scale/out_shift/HWs are HOST arrays
 * of nlev entries; ws: nlev * N * groups * ws_chunks * 2 doubles. */
int cmk_groupnorm_affine_multi(const float* const* xs, const int* HWs, int nlev, const float* gamma, const float* beta, double* ws,
                               int ws_chunks, int N, int C, int groups, float eps, float* const* out_scale, float* const* out_shift,
                               void* stream);
/* Second half of the fused form: the statistics were written by the conv itself (cmk_conv_desc.gn_ws, all levels of one
 * cmk_conv2d_nhwc[_multi] call, N images each); this turns them into the same per-(image, channel) scale/shift. */
/* recs[l] = cmk_conv_gn_records(Hs[l], Ws[l], tune_wm of the producing conv): records per image of level l */
int cmk_groupnorm_affine_tiles(const double* ws, const int* Hs, const int* Ws, const int* recs, int nlev, const float* gamma, const float* beta,
                               int N, int C, int groups, float eps, float* const* out_scale, float* const* out_shift, void* stream);

/* ---- FCOS candidate selection + box decode (fcos_outputs.py:396-466) ------------------------------------------ */
typedef struct {
    const float* logits;   /* (N, HW, C) */
    const float* regctr;   /* (N, HW, 5): relu(scale*bbox_pred) (4) , ctrness logit (1) */
    int H; int W; int stride;
} cmk_fcos_level;
/* Per image i the candidates of all levels are appended in level order, location-major / class-minor
 * (the order of torch.nonzero, fcos_outputs.py:429 and Instances.cat :391-392).
 * cand_* have capacity `cap` rows per image; counts[i] receives the TRUE number: counts[i] > cap means rows beyond cap were dropped —
 * the caller compares (it already reads the counts) and re-runs with a larger capacity; the reference is unbounded (:444-449).
 * thresh_with_ctr: MODEL.FCOS.THRESH_WITH_CTR (:410-420): the candidate test is sigmoid(cls)*sigmoid(ctr) > thr instead of sigmoid(cls) > thr. */
int cmk_fcos_select(const cmk_fcos_level* levels, int num_levels, int N, int C, float pre_nms_thresh, int thresh_with_ctr,
                    float* cand_box /* N*cap*4 */, float* cand_score, int32_t* cand_cls, float* cand_loc /* N*cap*2 */,
                    int32_t* counts /* N */, int32_t* block_counts /* workspace */, int64_t block_counts_len,
                    int cap, void* stream);
int64_t cmk_fcos_select_ws_len(const cmk_fcos_level* levels, int num_levels, int N, int C);

/* ---- batched NMS + top-k (layers/ml_nms.py:93 -> d2 batched_nms; fcos_outputs.py:472-482) --------------------
 * Stable descending sort by score (ties by candidate index), torchvision's coordinate trick
 * (boxes + cls * (max_coord + 1)) for < 40000 candidates, per-class suppression otherwise, greedy IoU > thr,
 * the first `topk` (1..1024) survivors are written.  sort_ws: 4 * N * cap uint32.                                       */
int cmk_nms_topk(const float* cand_box, const float* cand_score, const int32_t* cand_cls, const float* cand_loc,
                 const int32_t* counts, int N, int cap, float iou_thr, int topk,
                 float* out_box /* N*topk*4 */, float* out_score, int64_t* out_cls, float* out_loc, int32_t* out_idx,
                 int32_t* out_count /* N */, uint32_t* sort_ws, void* stream);

/* ---- multi-level ROIAlignV2 with CenterMask's ratio level assignment (pooler.py:80-118,290-366) ---------------- */
int cmk_roi_align_ratio(const float* const* feats /* host array of device pointers */, const int* feat_h, const int* feat_w,
                        const float* scales, int num_levels, int min_level, int C,
                        const float* boxes /* N*topk*4 */, const int32_t* counts, const float* img_area /* N */,
                        int N, int topk, int out_size, int sampling_ratio,
                        float* y, int y_cs /* (N*topk, out, out, y_cs) */, int32_t* out_level, void* stream);

/* The general pooler (pooler.py:192-288): aligned = 1 ROIAlignV2 / 0 ROIAlign v1 (torchvision roi_align aligned flag, :243-255);
 * assign_by_area = 0 the "ratio" rule above / 1 FPN Eqn.(1): floor(canonical_level + log2(sqrt(area) / canonical_box_size + eps))
 * clamped to the levels (pooler.py:121-152; img_area may then be NULL). */
int cmk_roi_align_pool(const float* const* feats, const int* feat_h, const int* feat_w, const float* scales, int num_levels, int min_level,
                       int C, const float* boxes, const int32_t* counts, const float* img_area, int N, int topk, int out_size,
                       int sampling_ratio, int aligned, int assign_by_area, float canonical_box_size, int canonical_level,
                       float* y, int y_cs, int32_t* out_level, void* stream);

/* ---- SAG-Mask spatial attention (sam.py:23-28) in place ------------------------------------------------------- */
int cmk_spatial_attention(float* x /* (R,S,S,C) */, const float* w /* [2][3][3] */, const int32_t* counts, int topk,
                          int R, int S, int C, void* stream);

/* ---- mask predictor for the predicted class only + sigmoid (sam.py:83,97; mask_head.py:202-208) ---------------
 * deconv_out: (R, S, S, 4, C) = relu(deconv) laid out (dh,dw)-major; masks: (R, 2S, 2S).                     */
int cmk_mask_predict(const float* deconv_out, const float* pw /* [classes][C] */, const float* pb,
                     const int64_t* cls, const int32_t* counts, int topk, int R, int S, int C,
                     float* masks, float* mask_logits_opt, void* stream);

/* ---- maxpool 2x2 of the masks into channel `co` of the MaskIoU input (maskiou_head.py:108-112) ---------------- */
int cmk_mask_pool_concat(const float* masks /* (R,2S,2S) */, float* y, int y_cs, int y_co, int R, int S, void* stream);

/* ---- mask_scores = scores * iou[cls] (maskiou_head.py:50-60) -------------------------------------------------- */
int cmk_mask_iou_score(const float* iou /* (R, classes_stride) */, int iou_cs, const float* scores, const int64_t* cls,
                       float* mask_scores, int R, void* stream);

/* ---- input side: (x - mean) / std of one CHW image (uint8 if src_is_u8 else float32) into its zero-padded slot
 * (3,H,W) of the batched NCHW tensor (deploy_utils.py:76-98; d2 preprocess_image + ImageList.from_tensors).  mean3 / std3
 * are HOST arrays of 3 floats. ------------------------------------------------------------------------------------- */
int cmk_preprocess_chw(const void* src, int src_is_u8, float* dst, int h, int w, int H, int W, const float* mean3,
                       const float* std3, void* stream);

/* ---- output side: paste (R,S,S) soft masks into (R,H,W) uint8 bitmasks at `threshold` (deploy_utils.py:151-156 ->
 * d2 ROIMasks.to_bitmasks: bilinear grid_sample, align_corners=False, zero padding); boxes (R,4) are the rescaled and
 * clipped output boxes. ------------------------------------------------------------------------------------------------ */
int cmk_paste_masks(const float* masks, const float* boxes, int R, int S, int H, int W, float threshold, uint8_t* out,
                    void* stream);

/* ---- multi-GPU result exchange (the reference's analogue: comm.gather in evaluation/coco_evaluation.py:155-156) ----------
 * One fixed-stride float record per image, written straight into the all-gather send buffer:
 * [box 4K | score K | mask_score K | loc 2K | class K (as float) | mask K*hw*hw | count], K*(9 + hw*hw) + 1 floats. */
int cmk_pack_records(const float* box, const float* score, const float* mask_scores, const float* loc, const int64_t* cls,
                     const float* masks, const int32_t* counts, int N, int K, int mask_hw, float* rec /* N * width */, void* stream);

#ifdef __cplusplus
}
#endif
#endif
