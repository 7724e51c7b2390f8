// Fine-level local expectation for PATS on gfx950.
//
// Replaces ThirdLayer.Compute_result (models/third_layer.py:184-217) and the match-label rule
// (models/third_layer.py:161-170): six gathers + two einsums over [P,16,25] index tensors in the
// reference; here one wave owns one 65x65 problem and walks its 16 centre rows (the 4x4 centre of
// the 8x8 source window).  Per row: argmax over the 64 real targets (first index on ties), a 5x5
// neighbourhood on the zero-padded (pad 2) 8x8 target map with weights sqrt(P + 1e-7) / scale
// (scale padded with 1e-2), expectation -> sub-pixel target; label from the argmax over all 65
// columns hitting the dustbin (outdoor) or a fixed cell pattern (indoor).
// kornia.create_meshgrid (un-vendored, kornia==0.5.5) is only an index grid here: (x, y) integer
// coordinates - restated inline.
#include "common.hpp"
#include "third_device.hpp"

namespace pats {

__global__ void __launch_bounds__(256)
compute_result_kernel(const float* __restrict__ scores, int input_is_log, int64_t P,
                      const float* __restrict__ scale_x, const float* __restrict__ scale_y,
                      const int64_t* __restrict__ p_s, const int64_t* __restrict__ p_t, int outdoor,
                      ComputeResultOut o) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * 4 + wave;
    if (p >= P) return;
    compute_result_problem(scores + p * (int64_t)65 * 65, input_is_log, p, scale_x + p * 64,
                           scale_y + p * 64, (float)p_s[p * 2], (float)p_s[p * 2 + 1],
                           (float)p_t[p * 2], (float)p_t[p * 2 + 1], outdoor, o, lane);
}

// whole_loss = where(wl >= 1e-2, wl, 0) / (count + 10) / 10        (:215)
__global__ void __launch_bounds__(256)
whole_loss_finish_kernel(float* __restrict__ wl, int64_t n, const int* __restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float denom = (float)(*count) + 10.0f;
    const float x = wl[i];
    wl[i] = (x >= 1e-2f ? x : 0.0f) / denom / 10.0f;
}

}  // namespace pats

using namespace pats;

static int compute_result_impl(const float* scores, int input_is_log, int64_t P, const float* scale_x, const float* scale_y,
                               const int64_t* p_s, const int64_t* p_t, int outdoor, float* mkpts0_f, float* mkpts1_f,
                               float* whole_loss, float* label, uint8_t* if_matching1, int* count, hipStream_t st) {
    if (whole_loss && fill_bytes(count, 0, sizeof(int), st)) return PATS_ERR_LAUNCH;
    hipLaunchKernelGGL(compute_result_kernel, dim3((unsigned)ceil_div(P, 4)), dim3(256), 0, st, scores,
                       input_is_log, P, scale_x, scale_y, p_s, p_t, outdoor,
                       ComputeResultOut{mkpts0_f, mkpts1_f, whole_loss, label, if_matching1, whole_loss ? count : nullptr});
    int rc = check_launch("compute_result_kernel");
    if (whole_loss && rc == PATS_OK) {
        hipLaunchKernelGGL(whole_loss_finish_kernel, dim3((unsigned)ceil_div(P * 16, 256)), dim3(256),
                           0, st, whole_loss, P * 16, count);
        rc = check_launch("whole_loss_finish_kernel");
    }
    return rc;
}

// whole_loss needs one cross-problem count (:215).  This entry takes it from the caller (4 bytes of device workspace):
// no allocation, no synchronisation, safe inside a HIP graph capture.
extern "C" int pats_compute_result_ws_f32(const float* scores, int input_is_log, int64_t P,
                                          const float* scale_x, const float* scale_y,
                                          const int64_t* p_s, const int64_t* p_t, int outdoor,
                                          float* mkpts0_f, float* mkpts1_f, float* whole_loss,
                                          float* label, uint8_t* if_matching1, void* workspace, size_t workspace_bytes,
                                          pats_stream_t stream) {
    PATS_REQUIRE(P >= 0, "compute_result: bad shape");
    if (P == 0) return PATS_OK;
    PATS_REQUIRE(scores && scale_x && scale_y && p_s && p_t && mkpts0_f && mkpts1_f && label &&
                     if_matching1, "compute_result: null pointer");
    PATS_REQUIRE(!whole_loss || (workspace && workspace_bytes >= sizeof(int)), "compute_result: whole_loss needs 4 bytes of workspace");
    return compute_result_impl(scores, input_is_log, P, scale_x, scale_y, p_s, p_t, outdoor, mkpts0_f, mkpts1_f, whole_loss, label,
                               if_matching1, static_cast<int*>(workspace), as_stream(stream));
}

// The round-1 signature, kept for callers built against it: without a workspace argument the count is a 4-byte
// stream-ordered allocation owned by the call (hipMallocAsync / hipFreeAsync on `stream`; not capturable).
extern "C" int pats_compute_result_f32(const float* scores, int input_is_log, int64_t P,
                                       const float* scale_x, const float* scale_y,
                                       const int64_t* p_s, const int64_t* p_t, int outdoor,
                                       float* mkpts0_f, float* mkpts1_f, float* whole_loss,
                                       float* label, uint8_t* if_matching1, pats_stream_t stream) {
    PATS_REQUIRE(P >= 0, "compute_result: bad shape");
    if (P == 0) return PATS_OK;
    PATS_REQUIRE(scores && scale_x && scale_y && p_s && p_t && mkpts0_f && mkpts1_f && label &&
                     if_matching1, "compute_result: null pointer");
    hipStream_t st = as_stream(stream);
    int* count = nullptr;
    if (whole_loss && hipMallocAsync((void**)&count, sizeof(int), st) != hipSuccess) {
        set_error("compute_result: hipMallocAsync failed");
        return PATS_ERR_LAUNCH;
    }
    const int rc = compute_result_impl(scores, input_is_log, P, scale_x, scale_y, p_s, p_t, outdoor, mkpts0_f, mkpts1_f, whole_loss,
                                       label, if_matching1, count, st);
    if (count) (void)hipFreeAsync(count, st);
    return rc;
}
