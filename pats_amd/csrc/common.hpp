// Shared host/device helpers for libpats_amd.so (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/pats_amd.h"

namespace pats {

constexpr int WAVE = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float ZERO_F = 1e-14f;  // the reference's `zero` (utils/utils.py:1201)

// ---- environment switches -----------------------------------------------------------------------------------------------------
// The PRODUCTION library reads exactly the switches INTEGRATION.md lists (section "Environment switches": each selects a TESTED
// alternative) - through env_switch().  Everything else - A/B partners of superseded kernel generations, timelines, ablations,
// occupancy pads - goes through diag_env(), which is a constant nullptr outside -DPATS_DIAG builds (libpats_amd_diag*.so:
// `python -m pats_amd.build --diag` compiles every file with it).  tests/test_host_abi.py holds the shipped library's strings to
// that table.
inline const char* env_switch(const char* name) { return getenv(name); }
#ifdef PATS_DIAG
inline const char* diag_env(const char* name) { return getenv(name); }
#else
inline const char* diag_env(const char*) { return nullptr; }
#endif
bool cost_f32_only();      // PATS_COST_F32 (host.cpp): every contraction on the fp32 MFMA (the in-kernel fallback of the fp16 split)

// ---- error plumbing ----------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);
int sinkhorn_mode();   // PATS_SINKHORN_* (host.cpp)
// device-resident count of problems whose linear-domain solve left the guard and was redone in the
// log domain (host.cpp; one counter per device, allocated on first use; null if that failed)
unsigned long long* fallback_counter();
// the GNN layers' overflow protocol (host.cpp): deferred mode = no gated redo chain, one sticky device flag the caller reads
bool gnn_redo_deferred();
int* gnn_overflow_flag();

#define PATS_REQUIRE(cond, ...)               \
    do {                                      \
        if (!(cond)) {                        \
            pats::set_error(__VA_ARGS__);     \
            return PATS_ERR_INVALID;          \
        }                                     \
    } while (0)

// ---- wave-level reductions: 4 DPP steps inside each 16-lane row, then two half-swaps -------
// Every lane ends up with the full 64-lane result (all-reduce).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}

constexpr int DPP_QUAD_XOR1 = 0xB1;     // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;     // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

struct OpSum {
    __device__ __forceinline__ float operator()(float a, float b) const { return a + b; }
};
struct OpMax {
    __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); }
};

// ---- exchanges between the halves of a wave / between neighbouring 16-lane rows --------------------------------------
// lane_swap32(a, b): lanes 32..63 of a <-> lanes 0..31 of b; lane_swap16(a, b): the odd 16-lane rows of a <-> the even rows
// of b, in place (v_permlane32_swap_b32 / v_permlane16_swap_b32).  The builtin returns both registers; hipcc (ROCm 7.2)
// mis-folds the pair when it can see through it - op(r[0], r[1]) came out as op(r[0], r[0]) (a wave sum returned 4 x the row
// total on hardware) - so the second operand and both results pass through empty asm to stay opaque.
// -DPATS_NO_PERMLANE_SWAP: the same exchanges on the LDS crossbar (ds_bpermute_b32 / ds_swizzle_b32; every lane hands its
// partner the one value the partner wants) - the A/B partner used while hunting the round-3 barrier bug (profiles/r03_determinism.md).
#ifndef PATS_NO_PERMLANE_SWAP
__device__ __forceinline__ void lane_swap32(unsigned& a, unsigned& b) {
    asm volatile("" : "+v"(b));
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
    asm volatile("" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lane_swap16(unsigned& a, unsigned& b) {
    asm volatile("" : "+v"(b));
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
    asm volatile("" : "+v"(a), "+v"(b));
}
#else
__device__ __forceinline__ void lane_swap32(unsigned& a, unsigned& b) {
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const bool lo = lane < 32u;
    const unsigned got = (unsigned)__builtin_amdgcn_ds_bpermute((int)((lane ^ 32u) << 2), (int)(lo ? b : a));
    a = lo ? a : got;
    b = lo ? got : b;
}
__device__ __forceinline__ void lane_swap16(unsigned& a, unsigned& b) {
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const bool even = (lane & 16u) == 0u;
    const unsigned got = (unsigned)__builtin_amdgcn_ds_swizzle((int)(even ? b : a), (16 << 10) | 0x1f);     // lane ^ 16
    a = even ? a : got;
    b = even ? got : b;
}
#endif

template <class Op>
__device__ __forceinline__ float wave_allreduce(float v, Op op) {
    v = op(v, dpp_f<DPP_QUAD_XOR1>(v));
    v = op(v, dpp_f<DPP_QUAD_XOR2>(v));
    v = op(v, dpp_f<DPP_ROW_HALF_MIRROR>(v));
    v = op(v, dpp_f<DPP_ROW_MIRROR>(v));
    // rows of 16 now hold their totals; exchange rows 0<->1, 2<->3 then halves 0<->1
    {
        unsigned x = __builtin_bit_cast(unsigned, v), y = x;
        lane_swap16(x, y);
        v = op(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y));
    }
    {
        unsigned x = __builtin_bit_cast(unsigned, v), y = x;
        lane_swap32(x, y);
        v = op(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y));
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return wave_allreduce(v, OpSum()); }

// 64-lane sum delivered as a wave-uniform value (SGPR): four in-row DPP steps, then the GFX9 row
// broadcasts (row 0->1 and 2->3, then lane 31 -> rows 2,3) leave the total in row 3 and one
// v_readlane hands it to every lane.  Seven issue slots against ten for the all-reduce above
// (whose two lane swaps each need an opaque register copy).  Masked-off rows add the `old` 0.
constexpr int DPP_ROW_BCAST15 = 0x142;
constexpr int DPP_ROW_BCAST31 = 0x143;
__device__ __forceinline__ float wave_sum_uniform(float v) {
    v += dpp_f<DPP_QUAD_XOR1>(v);
    v += dpp_f<DPP_QUAD_XOR2>(v);
    v += dpp_f<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f<DPP_ROW_MIRROR>(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), DPP_ROW_BCAST15, 0xa, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), DPP_ROW_BCAST31, 0xc, 0xf, false));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// x / d for a loop-invariant divisor, rd = 1.0f / d (IEEE): one residual correction of x * rd.
// Agrees with the IEEE quotient the reference computes (`/ D**.5`) on every one of 3.6e7 sampled
// operands per divisor used here, in 3 issue slots instead of the ~13 of a full fp32 division.
__device__ __forceinline__ float div_invariant(float x, float d, float rd) {
    const float q = x * rd;
    return fmaf(fmaf(-q, d, x), rd, q);
}
__device__ __forceinline__ float wave_max(float v) { return wave_allreduce(v, OpMax()); }

// argmax all-reduce with first-index tie-break (ATen CPU semantics relied on at
// first_layer.py:162, utils.py:1182,1232, third_layer.py:188)
__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        float ov = __shfl_xor(v, off);
        int oi = __shfl_xor(i, off);
        argmax_combine(v, i, ov, oi);
    }
}

// raw transcendental units (v_exp_f32 = 2^x, v_log_f32 = log2 x): no denormal range fix-up code.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// ---- workgroup barriers ------------------------------------------------------------------------------------------------
// wg_barrier(): THE barrier of this library - every kernel uses it where CUDA-style code would write __syncthreads().
// __syncthreads() is `fence release (workgroup); s_barrier; fence acquire (workgroup)`, and hipcc (ROCm 7.2, gfx950) lowers
// the release WITHOUT `s_waitcnt lgkmcnt(0)` wherever it reasons that the LDS operations of one CU execute in one total
// order - e.g. at the top of a Sinkhorn sweep loop whose latch ends in this wave's ds_write of the scaling vector.  On MI355X
// a wave does pass such a barrier with its ds_write still queued, and the waves the barrier releases read the previous
// sweep's values: 1-9 of 8 192 fine-level problems per launch came back with perturbed duals (profiles/r03_determinism.md).
// Round 3 patched the wait into the compiler's assembly; since round 4 it is in the SOURCE: the explicit
// `s_waitcnt lgkmcnt(0)` below (vmcnt 63 / expcnt 7 = "do not wait": software-pipelined kernels keep global loads in flight
// across barriers on purpose) sits between the release fence and the s_barrier, so plain `hipcc -c` builds every file
// correctly.  tools/check_code_objects.py still checks the shipped code objects (tests/test_host_abi.py runs it).
constexpr int WAITCNT_LGKM0 = 0xC07F;      // gfx9 s_waitcnt encoding: vmcnt[3:0|15:14] = 63, expcnt[6:4] = 7, lgkmcnt[11:8] = 0
__device__ __forceinline__ void wg_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(WAITCNT_LGKM0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// wg_barrier_global(): the same for data that crosses the waves of a workgroup through GLOBAL memory (written before the
// barrier, read by another wave behind it).  hipcc's workgroup-scope release also omits the vmcnt wait on gfx950 outside
// threadgroup-split mode (same in-order reasoning); after what that reasoning was worth for LDS it is explicit here.
// s_waitcnt 0 = vmcnt(0) expcnt(0) lgkmcnt(0).
__device__ __forceinline__ void wg_barrier_global() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Barriers of a wave that does nothing but LDS DMA (global_load_lds) for the others - csrc/gnn_fine.hip's attention kernel gives the
// K and V staging a wave each, so that their vmcnt queues hold nothing else and the computing waves' hold no DMA.  No fences: a
// workgroup-scope release with DMA in flight makes hipcc wait vmcnt(0) (the DMA's LDS writes are what it would publish), which is
// exactly what wg_dma_arrive() must not do (the fill issued before it is awaited one phase later); wg_dma_landed() waits for
// everything the wave has issued, then joins - the waves it releases read the staged bytes behind their own wg_barrier().
__device__ __forceinline__ void wg_dma_arrive() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(WAITCNT_LGKM0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wg_dma_landed() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// wave_lds_sync(): between a wave's LDS writes and the reads of them by OTHER lanes of the SAME wave where no wg_barrier() stands in
// between.  The hardware runs a wave's LDS operations in order, but to the compiler a lane's load does not depend on another lane's
// store - in sinkhorn_blk145w2_kernel it hoisted such a read above the store (round 4: seven lanes in eight multiplied with the
// previous sweep's value).  Wavefront-scope release / acquire fences order the two; no instruction is emitted.
// tools/wave_lds_order_repro.hip is the pattern on its own; tools/lds_handover_audit.py lists the candidate sites of a tree.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup-wide OR of a predicate (what __syncthreads_or does, on wg_barrier()).  Called uniformly by all threads.
__device__ __forceinline__ bool wg_barrier_or(bool pred) {
    __shared__ int wg_or_slot;
    if (threadIdx.x == 0) wg_or_slot = 0;
    wg_barrier();
    if (__builtin_amdgcn_ballot_w64(pred) != 0ull && (threadIdx.x & 63) == 0) wg_or_slot = 1;     // same value from every writer
    wg_barrier();
    const bool any = wg_or_slot != 0;
    wg_barrier();                                       // the slot is free for the next call
    return any;
}

inline hipStream_t as_stream(pats_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// memset as a kernel of this library (host.cpp): the throughput path's steps then consist of pats:: kernels only, and a
// fill is stream-ordered like every other launch (hipMemsetAsync turns into a runtime-internal fill kernel)
int fill_bytes(void* p, int value, size_t n, hipStream_t st);

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace pats
