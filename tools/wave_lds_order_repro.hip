// Stand-alone reproducer for hazard (c) of the round-4 verdict (csrc/sinkhorn_blk2w.hip, wave_lds_sync()): inside ONE wave a lane reads
// from LDS a value that ANOTHER lane of the same wave stored a few instructions earlier, with no barrier in between (the hardware
// executes a wave's LDS operations in order, so none is needed for the hardware).  To the compiler the load and the store are
// unrelated - per thread their addresses provably differ - and it may hoist the load above the store: the reader then sees the
// previous sweep's value.  In sinkhorn_blk145w2_kernel that made seven lanes in eight use a stale row scaling.
//
// The kernel below is the pattern reduced to its skeleton, in the two shapes the library has:
//   A  "one lane in eight publishes": lanes with (lane & 7) == 0 store a per-sweep value to a[lane >> 3]; every lane then reads
//      a[lane >> 3] (its group's slot) - the two-wave Sinkhorn kernel's ninth row scaling;
//   B  "neighbour exchange": every lane stores to a[lane] and reads a[lane ^ 1].
// A per-sweep recurrence makes a stale read change the final value.  SYNC = 1 puts wave_lds_sync() (wavefront-scope release +
// acquire fences: an ordering, no instruction) between store and load, SYNC = 0 leaves them to the compiler.
// Output: one JSON line with the number of wrong lanes per variant.  Whether SYNC = 0 fails is a property of the COMPILER (it is
// deterministic per build): the test (tests/test_hazard_repro_gpu.py) requires SYNC = 1 to be right and records SYNC = 0.
// build: hipcc --offload-arch=gfx950 -O3 tools/wave_lds_order_repro.hip -o /tmp/wave_lds_order_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int SHAPE, int SYNC>
__global__ void __launch_bounds__(64) order_kernel(float* __restrict__ out, const float* __restrict__ in, int sweeps) {
    __shared__ float a[64];
    __shared__ float k[64 * 24];                          // per-lane "matrix row": keeps independent work between store and load
    const int lane = threadIdx.x;
    for (int j = 0; j < 24; ++j) k[lane * 24 + j] = in[(lane * 24 + j) % 1536];
    a[lane] = 0.f;
    __syncthreads();
    float v = 1.0f + 0.001f * lane;
    for (int s = 0; s < sweeps; ++s) {
        // independent work the scheduler would like to overlap with an LDS round trip (the sweep's dot product)
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 24; ++j) dot = fmaf(k[lane * 24 + j], v, dot);
        const float pub = 1.0f / (1.0f + 0.5f * fabsf(dot));            // what this lane publishes this sweep
        float got;
        if (SHAPE == 0) {
            if ((lane & 7) == 0) a[lane >> 3] = pub;
            if (SYNC) wave_lds_sync();
            got = a[lane >> 3];
        } else {
            a[lane] = pub;
            if (SYNC) wave_lds_sync();
            got = a[lane ^ 1];
        }
        v = 0.5f * v + got;                                             // a stale `got` changes every later sweep
        if (SYNC) wave_lds_sync();                                      // (the next sweep's store must not overtake this read either)
    }
    out[blockIdx.x * 64 + lane] = v;
}

// the same recurrence on the host, in the same fp32 operations
static void reference(int shape, const std::vector<float>& in, int sweeps, std::vector<float>& ref) {
    std::vector<float> v(64), pub(64), a(64, 0.f);
    for (int l = 0; l < 64; ++l) v[l] = 1.0f + 0.001f * l;
    for (int s = 0; s < sweeps; ++s) {
        for (int l = 0; l < 64; ++l) {
            float dot = 0.f;
            for (int j = 0; j < 24; ++j) dot = fmaf(in[(l * 24 + j) % 1536], v[l], dot);
            pub[l] = 1.0f / (1.0f + 0.5f * fabsf(dot));
        }
        if (shape == 0) { for (int l = 0; l < 64; l += 8) a[l >> 3] = pub[l]; }
        else { for (int l = 0; l < 64; ++l) a[l] = pub[l]; }
        for (int l = 0; l < 64; ++l) v[l] = 0.5f * v[l] + (shape == 0 ? a[l >> 3] : a[l ^ 1]);
    }
    ref = v;
}

template <int SHAPE, int SYNC>
static int run(const float* din, float* dout, const std::vector<float>& hin, int sweeps) {
    std::vector<float> ref, got(64 * 8);
    reference(SHAPE, hin, sweeps, ref);
    hipLaunchKernelGGL((order_kernel<SHAPE, SYNC>), dim3(8), dim3(64), 0, 0, dout, din, sweeps);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (size_t i = 0; i < got.size(); ++i) {
        const float r = ref[i % 64];
        bad += !(fabsf(got[i] - r) <= 1e-5f * fabsf(r));              // (1 / x on the device is v_rcp + refinement: allow rounding, not staleness)
    }
    return bad;
}

int main() {
    std::vector<float> hin(1536);
    for (int i = 0; i < 1536; ++i) hin[i] = 0.01f * (float)((i * 37) % 101 - 50);
    float *din, *dout;
    hipMalloc((void**)&din, 1536 * 4); hipMalloc((void**)&dout, 64 * 8 * 4);
    hipMemcpy(din, hin.data(), 1536 * 4, hipMemcpyHostToDevice);
    const int sweeps = 40;
    printf("{\"sweeps\": %d, \"lanes_checked\": 512, \"wrong_lanes\": {\"publish_one_in_eight\": {\"no_sync\": %d, \"wave_lds_sync\": %d}, "
           "\"neighbour_exchange\": {\"no_sync\": %d, \"wave_lds_sync\": %d}}}\n", sweeps,
           run<0, 0>(din, dout, hin, sweeps), run<0, 1>(din, dout, hin, sweeps), run<1, 0>(din, dout, hin, sweeps), run<1, 1>(din, dout, hin, sweeps));
    return 0;
}
