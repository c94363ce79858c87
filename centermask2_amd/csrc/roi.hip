// CenterROIHeads gather/indexing kernels: multi-level ROIAlignV2 with the CenterMask "ratio" level rule, SAG-Mask
// spatial attention, class-selected mask predictor + sigmoid, 2x2 mask pooling into the MaskIoU input, score calibration.
// ROIs live in a padded [image][topk] layout; slot s of image n is valid iff s < counts[n] (device memory), so the
// launch geometry is static.  Features are NHWC: one wave covers the 256 channels of a sample point with float4 lanes,
// i.e. every bilinear corner is one coalesced 1 KiB read.
//
// Reference call sites: pooler.py:70-118,155-189,290-366 + detectron2 ROIAlign -> torchvision roi_align (source absent;
// restated in oracle/oracle_ops.c); sam.py:12-28,92-97; mask_head.py:174-216; maskiou_head.py:50-60,107-112.
#include "cmk_common.hpp"

namespace cmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MAXL = 4;

struct RoiLevels {
    const float* feat[MAXL];
    int H[MAXL], W[MAXL];
    float scale[MAXL];
    int num_levels, min_level;
    int assign_area;          // 0: CenterMask "ratio" rule (pooler.py:80-118); 1: FPN Eqn.(1) by box area (pooler.py:121-152)
    float canonical_size;     //    canonical_box_size, canonical_level of the area rule
    int canonical_level;
    int aligned;              // 1: ROIAlignV2 (pixel-centre shift of 0.5); 0: ROIAlign v1 (no shift, RoI at least 1x1), pooler.py:243-255
};

// grid = (ceil(out*out / 4), R); block = 4 waves, one output bin each; lanes = channel quads (C == 256 -> 64 lanes).
__global__ __launch_bounds__(256) void roi_align_kernel(const RoiLevels L, int C, const float* __restrict__ boxes,
                                                       const int32_t* __restrict__ counts, const float* __restrict__ img_area, int topk,
                                                       int out_size, int sampling_ratio, float* __restrict__ y, int y_cs,
                                                       int32_t* __restrict__ out_level, int num_rois) {
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs; all bin groups of one RoI go to the same XCD so the
    // feature rows it samples are fetched into ONE L2 (the adaptive sampling grid re-reads neighbouring pixels many times)
    const int bpr = (out_size * out_size + 3) >> 2;          // workgroups (4 bins each) per RoI
    const int xq = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int r = (xq / bpr) * 8 + xcd;
    if (r >= num_rois) return;
    const int n = r / topk, s = r - n * topk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bin = (xq % bpr) * 4 + wave;
    if (bin >= out_size * out_size) return;
    const int ph = bin / out_size, pw = bin - ph * out_size;
    float* yo = y + ((long)r * out_size * out_size + bin) * y_cs;
    const bool valid = s < counts[n];
    const int C4 = C >> 2;
    if (!valid) {
        for (int c4 = lane; c4 < C4; c4 += 64) *reinterpret_cast<f32x4*>(yo + c4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (bin == 0 && lane == 0) out_level[r] = -1;
        return;
    }
    const float* b = boxes + (long)r * 4;
    const float bx0 = b[0], by0 = b[1], bx1 = b[2], by1 = b[3];
    // pooler.py:80-118: ceil(max_level - log2(img_area / box_area + eps)), clamped; eps (2.2e-16) is added in fp32
    const float box_area = (bx1 - bx0) * (by1 - by0);
    const int max_level = L.min_level + L.num_levels - 1;
    float lvf = L.assign_area ? floorf((float)L.canonical_level + log2f(sqrtf(box_area) / L.canonical_size + 2.220446049250313e-16f))
                              : ceilf((float)max_level - log2f(img_area[n] / box_area + 2.220446049250313e-16f));
    lvf = fminf(fmaxf(lvf, (float)L.min_level), (float)max_level);   // NaN (0/0 areas) propagates like torch.clamp; see host note
    int lv = (int)lvf - L.min_level;
    if (!(lv >= 0 && lv < L.num_levels)) lv = 0;
    if (bin == 0 && lane == 0) out_level[r] = lv;

    const float* feat = L.feat[lv] + (long)n * L.H[lv] * L.W[lv] * C;
    const int height = L.H[lv], width = L.W[lv];
    const float sc = L.scale[lv];
    // torchvision roi_align: aligned shifts by half a pixel; the unaligned (v1) form forces the RoI to be at least 1x1
    const float off = L.aligned ? 0.5f : 0.0f;
    const float roi_start_w = bx0 * sc - off, roi_start_h = by0 * sc - off;
    const float roi_end_w = bx1 * sc - off, roi_end_h = by1 * sc - off;
    float roi_width = roi_end_w - roi_start_w, roi_height = roi_end_h - roi_start_h;
    if (!L.aligned) { roi_width = fmaxf(roi_width, 1.0f); roi_height = fmaxf(roi_height, 1.0f); }
    const float bin_size_h = roi_height / (float)out_size, bin_size_w = roi_width / (float)out_size;
    const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)out_size);
    const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)out_size);
    const float count = (float)max(grid_h * grid_w, 1);

    for (int c4 = lane; c4 < C4; c4 += 64) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int iy = 0; iy < grid_h; ++iy) {
            const float yy = roi_start_h + (float)ph * bin_size_h + ((float)iy + 0.5f) * bin_size_h / (float)grid_h;
            for (int ix = 0; ix < grid_w; ++ix) {
                const float xx = roi_start_w + (float)pw * bin_size_w + ((float)ix + 0.5f) * bin_size_w / (float)grid_w;
                float xq = xx, yq = yy;
                if (yq < -1.0f || yq > (float)height || xq < -1.0f || xq > (float)width) continue;
                if (yq <= 0.f) yq = 0.f;
                if (xq <= 0.f) xq = 0.f;
                int y_low = (int)yq, x_low = (int)xq, y_high, x_high;
                if (y_low >= height - 1) { y_high = y_low = height - 1; yq = (float)y_low; } else y_high = y_low + 1;
                if (x_low >= width - 1) { x_high = x_low = width - 1; xq = (float)x_low; } else x_high = x_low + 1;
                const float ly = yq - (float)y_low, lx = xq - (float)x_low, hy = 1.0f - ly, hx = 1.0f - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(feat + ((long)y_low * width + x_low) * C + c4 * 4);
                const f32x4 v2 = *reinterpret_cast<const f32x4*>(feat + ((long)y_low * width + x_high) * C + c4 * 4);
                const f32x4 v3 = *reinterpret_cast<const f32x4*>(feat + ((long)y_high * width + x_low) * C + c4 * 4);
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(feat + ((long)y_high * width + x_high) * C + c4 * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += w1 * v1[j] + w2 * v2[j] + w3 * v3[j] + w4 * v4[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] /= count;
        *reinterpret_cast<f32x4*>(yo + c4 * 4) = acc;
    }
}

// SAG-Mask spatial attention (sam.py:23-28), in place.  grid = R, block = 256; x: (R,S,S,C) dense.
__global__ __launch_bounds__(256) void spatial_attention_kernel(float* __restrict__ x, const float* __restrict__ w,
                                                               const int32_t* __restrict__ counts, int topk, int S, int C) {
    extern __shared__ float sm[];      // avg[S*S], max[S*S], att[S*S]
    const int r = blockIdx.x;
    const int n = r / topk, s = r - n * topk;
    if (s >= counts[n]) return;
    const int P = S * S, C4 = C >> 2;
    float* savg = sm;
    float* smax = sm + P;
    float* satt = sm + 2 * P;
    float* xr = x + (long)r * P * C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int p = wave; p < P; p += 4) {
        float sum = 0.f, mx = -INFINITY;
        for (int c4 = lane; c4 < C4; c4 += 64) {
            f32x4 v = *reinterpret_cast<const f32x4*>(xr + (long)p * C + c4 * 4);
            sum += (v.x + v.y) + (v.z + v.w);
            mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
        sum = wave_sum(sum);
        mx = wave_max(mx);
        if (lane == 0) { savg[p] = sum / (float)C; smax[p] = mx; }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += 256) {
        const int h = p / S, ww = p - h * S;
        float a = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                int ih = h + kh - 1, iw = ww + kw - 1;
                if (ih >= 0 && ih < S && iw >= 0 && iw < S) {
                    a += savg[ih * S + iw] * w[kh * 3 + kw];          // input channel 0 = mean (sam.py:24-26)
                    a += smax[ih * S + iw] * w[9 + kh * 3 + kw];      // input channel 1 = max
                }
            }
        satt[p] = 1.0f / (1.0f + expf(-a));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P * C4; i += 256) {
        const int p = i / C4;
        f32x4 v = *reinterpret_cast<f32x4*>(xr + (long)i * 4);
        const float g = satt[p];
        v.x *= g; v.y *= g; v.z *= g; v.w *= g;
        *reinterpret_cast<f32x4*>(xr + (long)i * 4) = v;
    }
}

// Predictor 1x1 for the predicted class only + sigmoid.  dec: (R,S,S,4,C) = relu(deconv), (dh,dw)-major.
// grid = (ceil(4*S*S/4), R); one wave per output pixel.
__global__ __launch_bounds__(256) void mask_predict_kernel(const float* __restrict__ dec, const float* __restrict__ pw,
                                                          const float* __restrict__ pb, const int64_t* __restrict__ cls,
                                                          const int32_t* __restrict__ counts, int topk, int S, int C,
                                                          float* __restrict__ masks, float* __restrict__ logits_opt) {
    const int r = blockIdx.y;
    const int n = r / topk, s = r - n * topk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;       // index over (h, w, dh, dw)
    const int P4 = S * S * 4;
    if (q >= P4) return;
    const int sub = q & 3, hw = q >> 2;
    const int h = hw / S, w = hw - h * S;
    const int oh = 2 * h + (sub >> 1), ow = 2 * w + (sub & 1);
    const long o = ((long)r * 2 * S + oh) * 2 * S + ow;
    if (s >= counts[n]) {
        if (lane == 0) { masks[o] = 0.f; if (logits_opt) logits_opt[o] = 0.f; }
        return;
    }
    const int k = (int)cls[r];
    const float* d = dec + ((long)r * P4 + q) * C;
    const float* wv = pw + (long)k * C;
    float acc = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        f32x4 a = *reinterpret_cast<const f32x4*>(d + c);
        f32x4 b = *reinterpret_cast<const f32x4*>(wv + c);
        acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        float lg = acc + pb[k];
        if (logits_opt) logits_opt[o] = lg;
        masks[o] = 1.0f / (1.0f + expf(-lg));
    }
}

// maxpool 2x2 of the (R,2S,2S) masks -> channel y_co of the (R,S,S,y_cs) MaskIoU input; the 15 pad channels after it
// are zeroed so that zero-padded weights never meet uninitialised memory.
__global__ __launch_bounds__(256) void mask_pool_kernel(const float* __restrict__ masks, float* __restrict__ y, int y_cs, int y_co, int R,
                                                       int S, int pad) {
    long total = (long)R * S * S;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int w = (int)(i % S), h = (int)((i / S) % S);
        long r = i / ((long)S * S);
        const float* m = masks + (r * 2 * S + 2 * h) * 2 * S + 2 * w;
        float v = fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2 * S], m[2 * S + 1]));
        float* o = y + i * y_cs + y_co;
        o[0] = v;
        for (int j = 1; j < pad; ++j) o[j] = 0.f;
    }
}

__global__ void mask_iou_score_kernel(const float* __restrict__ iou, int iou_cs, const float* __restrict__ scores,
                                      const int64_t* __restrict__ cls, float* __restrict__ out, int R) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) out[r] = scores[r] * iou[(long)r * iou_cs + cls[r]];
}

}  // namespace cmk

using namespace cmk;

extern "C" int cmk_roi_align_pool(const float* const* feats, const int* feat_h, const int* feat_w, const float* scales, int num_levels,
                                  int min_level, int C, const float* boxes, const int32_t* counts, const float* img_area, int N,
                                  int topk, int out_size, int sampling_ratio, int aligned, int assign_by_area, float canonical_box_size,
                                  int canonical_level, float* y, int y_cs, int32_t* out_level, void* stream) {
    if (!feats || !feat_h || !feat_w || !scales || !boxes || !counts || (!img_area && !assign_by_area) || !y || !out_level)
        return fail(CMK_EINVAL, "roi_align: null pointer%s", "");
    if (assign_by_area && !(canonical_box_size > 0.f)) return fail(CMK_EINVAL, "roi_align: canonical_box_size must be positive%s", "");
    if (num_levels < 1 || num_levels > MAXL || (C & 3) || y_cs < C || (y_cs & 3) || N < 1 || topk < 1 || out_size < 1)
        return fail(CMK_EINVAL, "roi_align: bad shape%s", "");
    RoiLevels L;
    L.num_levels = num_levels;
    L.min_level = min_level;
    L.assign_area = assign_by_area ? 1 : 0;
    L.canonical_size = canonical_box_size;
    L.canonical_level = canonical_level;
    L.aligned = aligned ? 1 : 0;
    for (int l = 0; l < MAXL; ++l) {
        bool ok = l < num_levels;
        if (ok && !feats[l]) return fail(CMK_EINVAL, "roi_align: null level%s", "");
        L.feat[l] = ok ? feats[l] : nullptr;
        L.H[l] = ok ? feat_h[l] : 1;
        L.W[l] = ok ? feat_w[l] : 1;
        L.scale[l] = ok ? scales[l] : 1.f;
    }
    int R = N * topk;
    hipLaunchKernelGGL(roi_align_kernel, dim3(cdiv(out_size * out_size, 4) * (((R + 7) / 8) * 8)), dim3(256), 0, (hipStream_t)stream, L, C, boxes,
                       counts, img_area, topk, out_size, sampling_ratio, y, y_cs, out_level, R);
    return check_launch("roi_align");
}

extern "C" int cmk_roi_align_ratio(const float* const* feats, const int* feat_h, const int* feat_w, const float* scales, int num_levels,
                                   int min_level, int C, const float* boxes, const int32_t* counts, const float* img_area, int N,
                                   int topk, int out_size, int sampling_ratio, float* y, int y_cs, int32_t* out_level, void* stream) {
    return cmk_roi_align_pool(feats, feat_h, feat_w, scales, num_levels, min_level, C, boxes, counts, img_area, N, topk, out_size,
                              sampling_ratio, 1, 0, 224.f, 4, y, y_cs, out_level, stream);
}

extern "C" int cmk_spatial_attention(float* x, const float* w, const int32_t* counts, int topk, int R, int S, int C, void* stream) {
    if (!x || !w || !counts) return fail(CMK_EINVAL, "spatial_attention: null pointer%s", "");
    if ((C & 3) || S < 1 || R < 1 || topk < 1 || R % topk) return fail(CMK_EINVAL, "spatial_attention: bad shape%s", "");
    hipLaunchKernelGGL(spatial_attention_kernel, dim3(R), dim3(256), (size_t)3 * S * S * sizeof(float), (hipStream_t)stream, x, w, counts,
                       topk, S, C);
    return check_launch("spatial_attention");
}

extern "C" int cmk_mask_predict(const float* deconv_out, const float* pw, const float* pb, const int64_t* cls, const int32_t* counts,
                                int topk, int R, int S, int C, float* masks, float* mask_logits_opt, void* stream) {
    if (!deconv_out || !pw || !pb || !cls || !counts || !masks) return fail(CMK_EINVAL, "mask_predict: null pointer%s", "");
    if ((C & 3) || S < 1 || R < 1 || topk < 1 || R % topk) return fail(CMK_EINVAL, "mask_predict: bad shape%s", "");
    hipLaunchKernelGGL(mask_predict_kernel, dim3(S * S, R), dim3(256), 0, (hipStream_t)stream, deconv_out, pw, pb, cls, counts, topk, S, C,
                       masks, mask_logits_opt);
    return check_launch("mask_predict");
}

extern "C" int cmk_mask_pool_concat(const float* masks, float* y, int y_cs, int y_co, int R, int S, void* stream) {
    if (!masks || !y) return fail(CMK_EINVAL, "mask_pool: null pointer%s", "");
    if (R < 1 || S < 1 || y_co >= y_cs) return fail(CMK_EINVAL, "mask_pool: bad shape%s", "");
    long total = (long)R * S * S;
    int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(mask_pool_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, masks, y, y_cs, y_co, R, S, y_cs - y_co);
    return check_launch("mask_pool");
}

extern "C" int cmk_mask_iou_score(const float* iou, int iou_cs, const float* scores, const int64_t* cls, float* mask_scores, int R,
                                  void* stream) {
    if (!iou || !scores || !cls || !mask_scores) return fail(CMK_EINVAL, "mask_iou_score: null pointer%s", "");
    hipLaunchKernelGGL(mask_iou_score_kernel, dim3(cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, iou, iou_cs, scores, cls, mask_scores, R);
    return check_launch("mask_iou_score");
}
