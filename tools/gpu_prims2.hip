// Device self-test of pats_amd/csrc/lane_reduce.hpp (diagnostic binary, not part of the product)
#include "../pats_amd/csrc/lane_reduce.hpp"
#include <vector>
#include <cmath>
namespace pats { void set_error(const char*, ...) {} int check_launch(const char*) { return 0; } int sinkhorn_mode() { return 0; } }
using namespace pats;
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    float p[8];
    for (int t = 0; t < 8; ++t) p[t] = (float)((lane * 7 + t * 13) % 23) + 0.01f * t;
    out[lane] = reduce8_consecutive(p, OpSum(), lane);
    out[64 + lane] = reduce8_strided(p, OpSum(), lane);
    out[128 + lane] = reduce8_consecutive(p, OpMax(), lane);
    out[192 + lane] = reduce8_strided(p, OpMax(), lane);
    out[256 + lane] = wave_sum_uniform(p[3]);
    out[320 + lane] = div_invariant(p[5] * 37.f - 400.f, sqrtf(128.f), 1.0f / sqrtf(128.f)) - (p[5] * 37.f - 400.f) / sqrtf(128.f);
}
int main() {
    float* d; (void)hipMalloc(&d, 384 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(384); (void)hipMemcpy(h.data(), d, 1536, hipMemcpyDeviceToHost);
    auto P = [](int lane, int t) { return (float)((lane * 7 + t * 13) % 23) + 0.01f * t; };
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int I = lane >> 3, J = lane & 7;
        float sc = 0, ss = 0, mc = -1e9, ms = -1e9;
        for (int k2 = 0; k2 < 8; ++k2) {
            sc += P(8 * I + k2, J); mc = fmaxf(mc, P(8 * I + k2, J));     // over lanes sharing I, value index J
            ss += P(8 * k2 + J, I); ms = fmaxf(ms, P(8 * k2 + J, I));     // over lanes sharing J, value index I
        }
        if (fabs(h[lane] - sc) > 1e-3) { if (bad++ < 4) printf("consec sum lane %d: %f want %f\n", lane, h[lane], sc); }
        if (fabs(h[64 + lane] - ss) > 1e-3) { if (bad++ < 8) printf("strided sum lane %d: %f want %f\n", lane, h[64 + lane], ss); }
        if (h[128 + lane] != mc) { if (bad++ < 12) printf("consec max lane %d: %f want %f\n", lane, h[128 + lane], mc); }
        if (h[192 + lane] != ms) { if (bad++ < 16) printf("strided max lane %d: %f want %f\n", lane, h[192 + lane], ms); }
    }
    float tot = 0; for (int lane = 0; lane < 64; ++lane) tot += P(lane, 3);
    for (int lane = 0; lane < 64; ++lane) {
        if (fabs(h[256 + lane] - tot) > 1e-2) { if (bad++ < 20) printf("wave_sum_uniform lane %d: %f want %f\n", lane, h[256 + lane], tot); }
        if (h[320 + lane] != 0.f) { if (bad++ < 24) printf("div_invariant lane %d: diff %g\n", lane, h[320 + lane]); }
    }
    printf("lane_reduce %s (%d bad)\n", bad ? "FAIL" : "ok", bad);
    return bad != 0;
}
