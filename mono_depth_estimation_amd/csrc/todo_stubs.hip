// Entry points declared in include/mde_hip.h whose kernels are not written yet.
// Each returns MDE_ENOTSUP loudly; this file shrinks to nothing as kernels land.
#include "mde_common.h"
#define STUB(name, ...) extern "C" int name(__VA_ARGS__) { mde_set_error(#name ": not implemented in this build"); return MDE_ENOTSUP; }
STUB(mde_stem_conv_fwd, const float*, const float*, void*, int, int, int, void*)
STUB(mde_stem_conv_wgrad, const float*, const void*, float*, int, int, int, void*)
STUB(mde_head_conv_fwd, const void*, const float*, float*, int, int, int, int, int, void*)
STUB(mde_head_conv_bwd, const void*, const float*, const float*, void*, float*, int, int, int, int, int, void*)
STUB(mde_bn_stats, const void*, int64_t, int, int, float*, void*)
STUB(mde_bn_finalize, const float*, int, int64_t, int, const float*, const float*, float*, float*, float, float, float*, float*, float*, float*, void*)
STUB(mde_bn_eval_scale_shift, const float*, const float*, const float*, const float*, float, int, float*, float*, void*)
STUB(mde_bn_apply, const void*, int, const float*, const float*, const void*, int, const float*, const float*, void*, int, int64_t, int, int, void*)
STUB(mde_bn_bwd_reduce, const void*, int, const void*, int, const void*, int, const float*, const float*, int64_t, int, int, float*, void*)
STUB(mde_bn_bwd_apply, const void*, int, const void*, int, const void*, int, const float*, const float*, const float*, const float*, int, int64_t, int, int, float*, float*, void*, int, int, void*, int, void*)
STUB(mde_maxpool_fwd, const void*, void*, uint8_t*, int, int, int, int, void*)
STUB(mde_maxpool_bwd, const void*, const uint8_t*, void*, int, int, int, int, void*)
STUB(mde_upsample_sigmoid_fwd, const float*, float*, int, int, int, int, int, int, void*)
STUB(mde_upsample_sigmoid_bwd, const float*, const float*, float*, int, int, int, int, int, int, void*)
STUB(mde_silog_fwd, const float*, const float*, int64_t, float, void*, float*, void*)
STUB(mde_silog_bwd, const float*, const float*, int64_t, float, const void*, const float*, float*, void*)
STUB(mde_depth_metrics, const float*, const float*, int64_t, void*, float*, void*)
STUB(mde_adam_step, float*, const float*, float*, float*, void*, int64_t, float, float, float, float, float, float, int, void*)
STUB(mde_cast_bf16, const float*, void*, int64_t, void*)
STUB(mde_pack_wt, const float*, void*, int, int, int, void*)
STUB(mde_nchw_to_nhwc_bf16, const float*, void*, int, int, int, int, void*)
STUB(mde_nhwc_bf16_to_nchw, const void*, float*, int, int, int, int, void*)
extern "C" int mde_bn_stats_blocks(int64_t, int) { return MDE_ENOTSUP; }
extern "C" size_t mde_silog_ws_bytes(void) { return 0; }
extern "C" size_t mde_metrics_ws_bytes(void) { return 0; }
