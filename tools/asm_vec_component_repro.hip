// hipcc (ROCm 7.2, gfx950) miscompiles component reads of a float4 defined by inline assembly: `g.y == want && g.w == want` below becomes ONE
// compare of g.x (hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only tools/asm_vec_component_repro.hip: v_cmp_eq_u32 vcc, s8, v0).
// Found in csrc/sinkhorn_stream.hip (round 5): the resident kernel polls its granules with 8-byte atomics instead.
#include <hip/hip_runtime.h>
typedef float f4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4s ld4(const float* p) { f4s v; asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void wt(f4s& a) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a) :: "memory"); }
__global__ void k(const float* p, unsigned want, float* out, int* okout) {
    f4s g0 = ld4(p + threadIdx.x * 4);
    wt(g0);
    const unsigned t0 = __builtin_bit_cast(unsigned, g0.y), t1 = __builtin_bit_cast(unsigned, g0.w);
    out[threadIdx.x * 2] = g0.x; out[threadIdx.x * 2 + 1] = g0.z;
    okout[threadIdx.x] = (t0 == want && t1 == want) ? 1 : 0;
}
