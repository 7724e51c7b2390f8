// Device self-test of the wave-level primitives in pats_amd/csrc/common.hpp (diagnostic binary,
// not part of the product):  hipcc --offload-arch=gfx950 tools/gpu_prims.hip -o /tmp/gpu_prims
#include "../pats_amd/csrc/common.hpp"
#include <vector>
#include <cmath>
namespace pats { void set_error(const char*, ...) {} int check_launch(const char*) { return 0; } }
using namespace pats;

__global__ void k(float* out) {
    const int lane = threadIdx.x;
    float v = (float)(lane * 3 % 17) + 0.25f * lane;
    out[0 * 64 + lane] = wave_sum(v);
    out[1 * 64 + lane] = wave_max(v);
    out[2 * 64 + lane] = dpp_f<DPP_QUAD_XOR1>((float)lane);
    out[3 * 64 + lane] = dpp_f<DPP_QUAD_XOR2>((float)lane);
    out[4 * 64 + lane] = dpp_f<DPP_ROW_HALF_MIRROR>((float)lane);
    out[5 * 64 + lane] = dpp_f<DPP_ROW_MIRROR>((float)lane);
    {
        unsigned x = lane;
        auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        out[6 * 64 + lane] = (float)r[0];
        out[7 * 64 + lane] = (float)r[1];
        auto q = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        out[8 * 64 + lane] = (float)q[0];
        out[9 * 64 + lane] = (float)q[1];
    }
    out[10 * 64 + lane] = fast_exp2(-0.5f * lane);
    out[11 * 64 + lane] = fast_log2(1.0f + lane);
    float bv = (float)((lane * 7) % 13); int bi = lane;
    wave_argmax(bv, bi);
    out[12 * 64 + lane] = (float)bi;
}

int main() {
    float* d; hipMalloc(&d, 13 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(13 * 64);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    double s = 0, mx = -1e9; for (int l = 0; l < 64; ++l) { double v = (l * 3 % 17) + 0.25 * l; s += v; if (v > mx) mx = v; }
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        if (fabs(h[l] - s) > 1e-3) { bad++; if (bad < 5) printf("wave_sum lane %d = %f want %f\n", l, h[l], s); }
        if (h[64 + l] != (float)mx) { bad++; if (bad < 5) printf("wave_max lane %d = %f want %f\n", l, h[64 + l], mx); }
    }
    const char* names[] = {"xor1", "xor2", "half_mirror", "mirror", "p16[0]", "p16[1]", "p32[0]", "p32[1]"};
    for (int r = 0; r < 8; ++r) { printf("%-12s:", names[r]); for (int l = 0; l < 64; ++l) printf(" %d", (int)h[(2 + r) * 64 + l]); printf("\n"); }
    printf("exp2(-0.5*l): %g %g %g  log2(1+l): %g %g %g\n", h[640], h[641], h[644], h[704], h[705], h[707]);
    printf("argmax lane0 -> %d (want first index of max 12: lane %d)\n", (int)h[768], 11); if ((int)h[768] != 11) bad++;
    printf("prims %s (%d bad)\n", bad ? "FAIL" : "ok", bad);
    return bad != 0;
}
