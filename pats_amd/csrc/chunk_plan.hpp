// split_patches, utils/utils.py:152-181, as one host/device function: the chunk plan of ONE image pair from the
// cumulative match count of its coarse grid (shared by pats_split_patches, pats_split_patches_device and the
// throughput-mode row table of batch.hip).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace pats {

__host__ __device__ inline int split_patches_plan(const int32_t* sc, int height, int width, int max_once_used,
                                                        int64_t* second, int64_t* third) {
    const int64_t L = (int64_t)height * width;
    auto at = [&](int64_t i) -> int64_t { return sc[((i % L) + L) % L]; };
    int cycle = 0, last_second = 0, last_third = 0;
    for (int i = 0; i < height; ++i) {
        const int64_t num = at((int64_t)(i + 1) * width - 1);
        if (num > (int64_t)max_once_used * (cycle + 1)) {
            const int64_t origin = last_second == 0 ? 0 : at((int64_t)last_second * width - 1);
            second[2 * cycle] = origin;
            second[2 * cycle + 1] = num;
            third[2 * cycle] = at((int64_t)last_third * width) - origin;
            third[2 * cycle + 1] = num - at((int64_t)i * width - 1);
            ++cycle;
            last_second = i;
            last_third = i + 1;
        }
    }
    const int64_t origin = last_second == 0 ? 0 : at((int64_t)last_second * width - 1);
    second[2 * cycle] = origin;
    second[2 * cycle + 1] = L;
    const int64_t end_num = (last_third == height) ? origin : at((int64_t)last_third * width);
    third[2 * cycle] = end_num - origin;
    third[2 * cycle + 1] = 0;
    return cycle + 1;
}

}  // namespace pats
