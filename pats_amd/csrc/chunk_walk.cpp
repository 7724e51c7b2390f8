// PATS.forward's chunk loop (models/pats.py:33-78) walked chunk by chunk: the C calls of one chunk that sit BETWEEN the network
// callbacks, issued as two entry points instead of seven.  Nothing here launches a kernel of its own - each entry point runs the
// library's existing entry points in the reference's order on one stream - what it saves is host time: walked from Python, a chunk is
// seven ctypes calls with their argument checks and ~25 output allocations, ~25 us each, and a pair of eight chunks was host-bound at
// ~2.5 ms with the GPU idle half of the time.
//   fine tail   second_layer.py:100-122 + pats.py:37-39,53-58: cost build + log_optimal_transport2 (+ ln k) + column flags, the
//               8-step area expansion on the 12 x 12 grid, the chunk's merge through the row table, the third level's inputs
//   third tail  third_layer.py:153-170 + pats.py:59-78: cost + OT + Compute_result + label over the chunk's capacity with the count on
//               the device, the scatter onto the 48 x 48 sub-cell grid, get_result with the chunk's mask as level 0
// The scratch of the sub-calls is ONE caller-provided block (they run in stream order: each one's scratch is dead when the next starts).
#include "common.hpp"

#include <algorithm>

using namespace pats;

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t pats_chunk_fine_tail_workspace_bytes(int64_t B, int64_t pairs, int H, int W) {
    if (B <= 0 || pairs <= 0) return 0;
    return al256(std::max({pats_cost_ot_workspace_bytes(B, 264, 145, 145, 2), pats_merge_batch_workspace_bytes(pairs, H, W),
                           pats_compact_workspace_bytes(B * 144)}));
}

extern "C" int pats_chunk_fine_tail_f32(const float* d0, const float* d1, int64_t B, const float* one, const float* ns, int iters, float bias_k,
                                        const float* scale_x, const float* scale_y, int merge_new, int Cmax, int c, int64_t pairs, int H,
                                        int W, int64_t row_origin, const int64_t* chunk_base, const int32_t* row_cell,
                                        const int32_t* row_slot, const uint8_t* row_forced, double* scores_back, int first_chunk,
                                        float* Z, uint8_t* col_nomatch, float* trust, float* core, float* points, float* x_scale,
                                        float* y_scale, int64_t* bound, uint8_t* row_nomatch, uint8_t* merged, float* mkpts0,
                                        float* mkpts1, int64_t* b_ids, int64_t* P_dev, void* wait_before_merge, void* record_after_merge,
                                        void* workspace, size_t workspace_bytes, pats_stream_t stream) {
    PATS_REQUIRE(B > 0 && pairs > 0, "chunk_fine_tail: bad shape");
    PATS_REQUIRE(workspace && workspace_bytes >= pats_chunk_fine_tail_workspace_bytes(B, pairs, H, W), "chunk_fine_tail: workspace too small");
    int rc;
    // second_layer.py:100-112: scores = log_optimal_transport2(0.1 d0^T d1 / sqrt(264), 1, scale_x * scale_y, iters), + ln k on the dustbins
    if ((rc = pats_cost_ot_flags_f32(d0, d1, B, 264, 145, 145, 2, one, ns, iters, bias_k, Z, col_nomatch, workspace, workspace_bytes, stream)))
        return rc;
    // second_layer.py:240-259: est_position on the log plan (exp fused into the load), 8 steps, lower bound 1e-3; row flags on the way
    if ((rc = pats_iterative_expand_f32(Z, 1, B, 145, 145, scale_x, scale_y, 12, 12, 12, 1e-3f, 8, trust, core, points, x_scale, y_scale,
                                        bound, row_nomatch, stream)))
        return rc;
    // second_layer.py:119-122 + pats.py:37-39: the chunk's merge on its own rows of the table.  The merges of a pair are ordered
    // (scores_back, pats.py:37); when consecutive chunks run on different streams the caller hands the previous chunk's event to wait
    // for HERE - not in front of the cost build and the solve, which need nothing of the previous chunk - and one to record behind it
    if (wait_before_merge && hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(wait_before_merge), 0) != hipSuccess)
        return check_launch("chunk_fine_tail: hipStreamWaitEvent");
    if ((rc = pats_merge_patches_chunks(merge_new, Cmax, c, c + 1, pairs, H, W, row_origin, B, chunk_base, row_cell, row_slot, row_forced, trust,
                                        row_nomatch, scores_back, first_chunk, merged, workspace, workspace_bytes, stream)))
        return rc;
    if (record_after_merge && hipEventRecord(reinterpret_cast<hipEvent_t>(record_after_merge), as_stream(stream)) != hipSuccess)
        return check_launch("chunk_fine_tail: hipEventRecord");
    // pats.py:53-58 over the capacity 144 B, the count stays on the device
    return pats_third_inputs_f32(merged, points, B, mkpts0, mkpts1, b_ids, B * 144, P_dev, workspace, workspace_bytes, stream);
}

extern "C" size_t pats_chunk_third_tail_workspace_bytes(int64_t B, int h, int w) {
    if (B <= 0) return 0;
    return al256(std::max(pats_compact_workspace_bytes(B * 144), pats_get_result_workspace_bytes((int64_t)h * w, B, 2304)));
}

extern "C" int pats_chunk_third_tail_f32(const float* feat0, const float* feat1, int64_t P_cap, const int64_t* P_dev, const float* scale,
                                         const int64_t* p_s, const int64_t* p_t, int iters, int outdoor, const uint8_t* merged,
                                         const float* points2, int64_t B, const uint8_t* chunk_mask, int h, int w, const float* pts_new,
                                         const float* scales, const uint8_t* ones, float* mkpts0_f, float* mkpts1_f, float* label,
                                         uint8_t* if_matching1, uint8_t* if_nomatching16, float* pts16, float* matches_l, float* matches_r,
                                         int32_t* match_row, int64_t* M_dev, void* workspace, size_t workspace_bytes, pats_stream_t stream) {
    PATS_REQUIRE(B > 0 && P_cap == B * 144 && h > 0 && w > 0, "chunk_third_tail: bad shape (P_cap must be 144 B)");
    PATS_REQUIRE(ones, "chunk_third_tail: null `ones` (B bytes of 1: the left_choice flags of pats.py:73)");
    PATS_REQUIRE(workspace && workspace_bytes >= pats_chunk_third_tail_workspace_bytes(B, h, w), "chunk_third_tail: workspace too small");
    int rc;
    // third_layer.py:153-170 for the chunk's surviving cells
    if ((rc = pats_third_level_counted_f32(feat0, feat1, P_cap, P_dev, 128, scale, nullptr, nullptr, p_s, p_t, iters, outdoor, mkpts0_f, mkpts1_f,
                                           label, if_matching1, stream)))
        return rc;
    // pats.py:59-67
    if ((rc = pats_refine_scatter_f32(merged, points2, mkpts1_f, label, 2, B, P_cap, if_nomatching16, pts16, workspace, workspace_bytes, stream)))
        return rc;
    // pats.py:68-78: get_result with this chunk's mask as the level-0 flags
    const int ps0[3] = {32, h, w}, ps1[3] = {2, 48, 48};
    return pats_get_result_chunks_f32(1, 1, chunk_mask, if_nomatching16, B, pts_new, pts16, scales, ps0, ps1, ones, ones, matches_l, matches_r,
                                      match_row, B * 2304, M_dev, workspace, workspace_bytes, stream);
}
