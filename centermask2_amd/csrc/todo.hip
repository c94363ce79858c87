// Bring-up placeholders: entry points declared in cmk.h whose kernels are not written yet fail loudly.
#include "cmk_common.hpp"
using namespace cmk;
#define NOTIMPL(name) return fail(CMK_EINVAL, "%s: not implemented yet", name)
extern "C" int cmk_fcos_select(const cmk_fcos_level*, int, int, int, float, float*, float*, int32_t*, float*, int32_t*, int32_t*, int64_t, int, void*) { NOTIMPL("cmk_fcos_select"); }
extern "C" int64_t cmk_fcos_select_ws_len(const cmk_fcos_level*, int, int, int) { return -1; }
extern "C" int cmk_nms_topk(const float*, const float*, const int32_t*, const float*, const int32_t*, int, int, float, int, float*, float*, int64_t*, float*, int32_t*, int32_t*, uint32_t*, void*) { NOTIMPL("cmk_nms_topk"); }
extern "C" int cmk_roi_align_ratio(const float* const*, const int*, const int*, const float*, int, int, int, const float*, const int32_t*, const float*, int, int, int, int, float*, int, int32_t*, void*) { NOTIMPL("cmk_roi_align_ratio"); }
extern "C" int cmk_spatial_attention(float*, const float*, const int32_t*, int, int, int, int, void*) { NOTIMPL("cmk_spatial_attention"); }
extern "C" int cmk_mask_predict(const float*, const float*, const float*, const int64_t*, const int32_t*, int, int, int, int, float*, float*, void*) { NOTIMPL("cmk_mask_predict"); }
extern "C" int cmk_mask_pool_concat(const float*, float*, int, int, int, int, void*) { NOTIMPL("cmk_mask_pool_concat"); }
extern "C" int cmk_mask_iou_score(const float*, int, const float*, const int64_t*, float*, int, void*) { NOTIMPL("cmk_mask_iou_score"); }
