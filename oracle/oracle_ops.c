/*
 * ORACLE — test infrastructure only.  Not part of the product path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Plain-C restatement of the two third-party CPU kernels the reference reaches through
 * detectron2 (source absent from /root/reference, README.md:203 pins torchvision==0.9.0):
 *   - torchvision::nms        (called via d2 batched_nms at layers/ml_nms.py:93)
 *   - torchvision::roi_align  (called via d2 ROIAlign at modeling/centermask/pooler.py:249-255,361-364)
 * Restated from the published algorithm of torchvision 0.9 (ops/cpu/nms_kernel.cpp,
 * ops/cpu/roi_align_kernel.cpp); no reference test pins them => "parity unpinned" for these two.
 * All arithmetic is single precision, in the same operation order as the published kernels.
 * Built with -ffp-contract=off so that no fused multiply-add changes a rounding.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Greedy NMS over boxes already offset by the caller (coordinate trick of batched_nms).
 * order: indices sorted by descending score (caller supplies a stable order).
 * keep_out: kept indices in that order; returns their number. suppress when IoU > thr. */
int64_t oracle_nms(const float *boxes, const int64_t *order, int64_t n, float thr, int64_t *keep_out)
{
    uint8_t *suppressed = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    float *areas = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    int64_t num_to_keep = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float *b = boxes + 4 * i;
        areas[i] = (b[2] - b[0]) * (b[3] - b[1]);
    }
    for (int64_t _i = 0; _i < n; ++_i) {
        int64_t i = order[_i];
        if (suppressed[i]) continue;
        keep_out[num_to_keep++] = i;
        float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        float iarea = areas[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            int64_t j = order[_j];
            if (suppressed[j]) continue;
            float xx1 = fmaxf(ix1, boxes[4 * j]);
            float yy1 = fmaxf(iy1, boxes[4 * j + 1]);
            float xx2 = fminf(ix2, boxes[4 * j + 2]);
            float yy2 = fminf(iy2, boxes[4 * j + 3]);
            float w = fmaxf(0.0f, xx2 - xx1);
            float h = fmaxf(0.0f, yy2 - yy1);
            float inter = w * h;
            float ovr = inter / (iarea + areas[j] - inter);
            if (ovr > thr) suppressed[j] = 1;
        }
    }
    free(suppressed);
    free(areas);
    return num_to_keep;
}

/* ROIAlign forward, NCHW float32 input, rois (M,5) = [batch_idx, x0, y0, x1, y1].
 * aligned != 0 => half-pixel offset and no min-size clamp (ROIAlignV2).
 * sampling_ratio <= 0 => adaptive grid ceil(roi_size / pooled_size). */
void oracle_roi_align(const float *input, int64_t channels, int64_t height, int64_t width,
                      const float *rois, int64_t num_rois, float spatial_scale,
                      int64_t pooled_h, int64_t pooled_w, int64_t sampling_ratio, int aligned,
                      float *output)
{
    for (int64_t n = 0; n < num_rois; ++n) {
        const float *roi = rois + 5 * n;
        int64_t batch = (int64_t)roi[0];
        float offset = aligned ? 0.5f : 0.0f;
        float roi_start_w = roi[1] * spatial_scale - offset;
        float roi_start_h = roi[2] * spatial_scale - offset;
        float roi_end_w = roi[3] * spatial_scale - offset;
        float roi_end_h = roi[4] * spatial_scale - offset;
        float roi_width = roi_end_w - roi_start_w;
        float roi_height = roi_end_h - roi_start_h;
        if (!aligned) {
            roi_width = fmaxf(roi_width, 1.0f);
            roi_height = fmaxf(roi_height, 1.0f);
        }
        float bin_size_h = roi_height / (float)pooled_h;
        float bin_size_w = roi_width / (float)pooled_w;
        int64_t grid_h = sampling_ratio > 0 ? sampling_ratio : (int64_t)ceilf(roi_height / (float)pooled_h);
        int64_t grid_w = sampling_ratio > 0 ? sampling_ratio : (int64_t)ceilf(roi_width / (float)pooled_w);
        float count = (float)(grid_h * grid_w > 1 ? grid_h * grid_w : 1);
        for (int64_t c = 0; c < channels; ++c) {
            const float *data = input + (batch * channels + c) * height * width;
            for (int64_t ph = 0; ph < pooled_h; ++ph) {
                for (int64_t pw = 0; pw < pooled_w; ++pw) {
                    float val = 0.0f;
                    for (int64_t iy = 0; iy < grid_h; ++iy) {
                        float yy = roi_start_h + (float)ph * bin_size_h +
                                   ((float)iy + 0.5f) * bin_size_h / (float)grid_h;
                        for (int64_t ix = 0; ix < grid_w; ++ix) {
                            float xx = roi_start_w + (float)pw * bin_size_w +
                                       ((float)ix + 0.5f) * bin_size_w / (float)grid_w;
                            float x = xx, y = yy;
                            if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) continue;
                            if (y <= 0) y = 0;
                            if (x <= 0) x = 0;
                            int64_t y_low = (int64_t)y, x_low = (int64_t)x, y_high, x_high;
                            if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; }
                            else y_high = y_low + 1;
                            if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; }
                            else x_high = x_low + 1;
                            float ly = y - (float)y_low, lx = x - (float)x_low;
                            float hy = 1.0f - ly, hx = 1.0f - lx;
                            float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                            val += w1 * data[y_low * width + x_low] + w2 * data[y_low * width + x_high] +
                                   w3 * data[y_high * width + x_low] + w4 * data[y_high * width + x_high];
                        }
                    }
                    val /= count;
                    output[((n * channels + c) * pooled_h + ph) * pooled_w + pw] = val;
                }
            }
        }
    }
}
