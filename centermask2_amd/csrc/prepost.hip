// The steps directly before and after the model:
//   preprocess  — (x - mean) / std of a CHW image (uint8 or float32) written into its zero-padded slot of the batched
//                 NCHW float tensor (deploy_utils.py:76-98; detectron2 preprocess_image + ImageList.from_tensors).
//   paste_masks — 28x28 soft masks -> full-image bitmasks at a threshold (deploy_utils.py:151-156 -> detectron2
//                 ROIMasks.to_bitmasks -> paste_masks_in_image: bilinear grid_sample, align_corners=False, zero padding).
// Both are HBM-bound streaming kernels.
#include "cmk_common.hpp"

namespace cmk {

template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(const T* __restrict__ src, float* __restrict__ dst, int C, int h, int w, int H, int W,
                                                        float m0, float m1, float m2, float s0, float s1, float s2) {
    long total = (long)C * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int x = (int)(i % W);
        int y = (int)((i / W) % H);
        int c = (int)(i / ((long)W * H));
        float v = 0.f;                                  // zero padding on the right/bottom (applied after normalisation)
        if (y < h && x < w) {
            float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
            v = ((float)src[((long)c * h + y) * w + x] - mean) / sd;
        }
        dst[i] = v;
    }
}

// grid = (ceil(W/256), H, R)
__global__ __launch_bounds__(256) void paste_masks_kernel(const float* __restrict__ masks, const float* __restrict__ boxes, int S, int H, int W,
                                                         float thr, uint8_t* __restrict__ out) {
    const int r = blockIdx.z, y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const float* b = boxes + (long)r * 4;
    const float x0 = b[0], y0 = b[1], x1 = b[2], y1 = b[3];
    // detectron2 _do_paste_mask: normalised coords of the pixel centre, then grid_sample's un-normalisation
    const float gx = ((float)x + 0.5f - x0) / (x1 - x0) * 2.0f - 1.0f;
    const float gy = ((float)y + 0.5f - y0) / (y1 - y0) * 2.0f - 1.0f;
    const float ix = ((gx + 1.0f) * (float)S - 1.0f) / 2.0f;
    const float iy = ((gy + 1.0f) * (float)S - 1.0f) / 2.0f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int xw = (int)fx, yn = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    const float* m = masks + (long)r * S * S;
    auto at = [&](int yy, int xx) -> float { return (yy >= 0 && yy < S && xx >= 0 && xx < S) ? m[yy * S + xx] : 0.f; };
    // aten grid_sampler bilinear: nw*(1-tx)(1-ty) + ne*tx(1-ty) + sw*(1-tx)ty + se*tx*ty
    float v = 0.f;
    if (ix > -1.0f && ix < (float)S && iy > -1.0f && iy < (float)S)
        v = at(yn, xw) * ((1.f - tx) * (1.f - ty)) + at(yn, xw + 1) * (tx * (1.f - ty)) + at(yn + 1, xw) * ((1.f - tx) * ty) +
            at(yn + 1, xw + 1) * (tx * ty);
    out[((long)r * H + y) * W + x] = (v >= thr) ? 1 : 0;
}

// Fixed-stride per-image result record for the multi-GPU all-gather: one pass over the padded result buffers,
// straight into the send buffer:  [box 4K | score K | mask_score K | loc 2K | cls K (as float) | mask K*S*S | count].  grid = (blocks, N)
__global__ __launch_bounds__(256) void pack_records_kernel(const float* __restrict__ box, const float* __restrict__ score, const float* __restrict__ mscore,
                                                          const float* __restrict__ loc, const int64_t* __restrict__ cls, const float* __restrict__ masks,
                                                          const int32_t* __restrict__ counts, int K, int SS, float* __restrict__ rec, int width) {
    const int n = blockIdx.y;
    float* r = rec + (long)n * width;
    const int o_score = 4 * K, o_ms = 5 * K, o_loc = 6 * K, o_cls = 8 * K, o_mask = 9 * K, o_cnt = 9 * K + K * SS;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < width; i += gridDim.x * 256) {
        float v;
        if (i < o_score) v = box[(long)n * 4 * K + i];
        else if (i < o_ms) v = score[(long)n * K + (i - o_score)];
        else if (i < o_loc) v = mscore[(long)n * K + (i - o_ms)];
        else if (i < o_cls) v = loc[(long)n * 2 * K + (i - o_loc)];
        else if (i < o_mask) v = (float)cls[(long)n * K + (i - o_cls)];
        else if (i < o_cnt) v = masks[(long)n * K * SS + (i - o_mask)];
        else v = (float)counts[n];
        r[i] = v;
    }
}

}  // namespace cmk

using namespace cmk;

extern "C" int cmk_pack_records(const float* box, const float* score, const float* mask_scores, const float* loc, const int64_t* cls,
                                const float* masks, const int32_t* counts, int N, int K, int mask_hw, float* rec, void* stream) {
    if (!box || !score || !mask_scores || !loc || !cls || !masks || !counts || !rec) return fail(CMK_EINVAL, "pack_records: null pointer%s", "");
    if (N < 1 || K < 1 || mask_hw < 1) return fail(CMK_EINVAL, "pack_records: bad shape%s", "");
    const int width = K * (9 + mask_hw * mask_hw) + 1;
    hipLaunchKernelGGL(pack_records_kernel, dim3(cdiv(width, 256 * 4), N), dim3(256), 0, (hipStream_t)stream, box, score, mask_scores, loc, cls, masks,
                       counts, K, mask_hw * mask_hw, rec, width);
    return check_launch("pack_records");
}

extern "C" int cmk_preprocess_chw(const void* src, int src_is_u8, float* dst, int h, int w, int H, int W, const float* mean3,
                                  const float* std3, void* stream) {
    if (!src || !dst || !mean3 || !std3) return fail(CMK_EINVAL, "preprocess: null pointer%s", "");
    if (h < 1 || w < 1 || H < h || W < w) return fail(CMK_EINVAL, "preprocess: padded size smaller than the image%s", "");
    long total = 3L * H * W;
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (src_is_u8)
        hipLaunchKernelGGL(preprocess_kernel<uint8_t>, dim3(grid), dim3(256), 0, st, (const uint8_t*)src, dst, 3, h, w, H, W, mean3[0], mean3[1],
                           mean3[2], std3[0], std3[1], std3[2]);
    else
        hipLaunchKernelGGL(preprocess_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)src, dst, 3, h, w, H, W, mean3[0], mean3[1],
                           mean3[2], std3[0], std3[1], std3[2]);
    return check_launch("preprocess");
}

extern "C" int cmk_paste_masks(const float* masks, const float* boxes, int R, int S, int H, int W, float threshold, uint8_t* out,
                               void* stream) {
    if (R == 0) return CMK_OK;
    if (!masks || !boxes || !out) return fail(CMK_EINVAL, "paste_masks: null pointer%s", "");
    if (R < 0 || S < 1 || H < 1 || W < 1 || H > 65535 || R > 65535) return fail(CMK_EINVAL, "paste_masks: bad shape%s", "");
    hipLaunchKernelGGL(paste_masks_kernel, dim3(cdiv(W, 256), H, R), dim3(256), 0, (hipStream_t)stream, masks, boxes, S, H, W, threshold, out);
    return check_launch("paste_masks");
}
