// Patch-subdivision gather for PATS on gfx950: crop bounds, left fixed-grid crops, and the
// batched crop + bilinear resize that replaces the reference's only native component.
//
//   bounds      Compute_imgs                      utils/utils.py:1350-1382
//   left crops  origin_extract on the padded left utils/utils.py:1300-1318, caller :1383-1384
//   resize      tensor_resize / resize            setup/library.cpp:47-66 (binding :92-93)
//
// The reference's resize is a serial C++ loop with five `.item()` device->host syncs per crop
// around narrow + upsample_bilinear2d + index_put_; here all K crops are one launch that reads
// the bounds from device memory.  Output pixels map to consecutive lanes (coalesced 384-byte row
// stores); source taps of one output row fall in at most two source rows, served by L1/L2.
#include "common.hpp"

#include <cstdlib>

#ifndef PATS_CROPS_NT_DEFAULT
#define PATS_CROPS_NT_DEFAULT 1
#endif

namespace pats {

// ---- bounds + ordered compaction of the matched patches --------------------------------------
__device__ __forceinline__ void
imgs_bounds_block(const float* __restrict__ x_scale, const float* __restrict__ y_scale,
                  const float* __restrict__ average_point, const uint8_t* __restrict__ ifn, int Np,
                  int height, int width, int img, int64_t* __restrict__ bound5,
                  int64_t* __restrict__ K_out, float* __restrict__ xsn, float* __restrict__ ysn,
                  float* __restrict__ avn) {
    __shared__ int wave_tot[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float ps = 32.0f, margin = 128.0f;
    const float board1 = (float)(32 * height - 1), board3 = (float)(32 * width);   // :1351
    if (tid == 0) base_s = 0;
    wg_barrier();
    for (int k0 = 0; k0 < Np; k0 += 1024) {
        const int k = k0 + tid;
        int flag = 0;
        long long l0 = 0, l1 = 0, l2 = 0, l3 = 0;
        if (k < Np) {
            const float ay = average_point[2 * k], ax = average_point[2 * k + 1];
            float b0 = (ay - y_scale[k] * 3.0f / 2.0f) * ps + margin;    // :1360-1363
            float b1 = (ay + y_scale[k] * 3.0f / 2.0f) * ps + margin;
            float b2 = (ax - x_scale[k] * 3.0f / 2.0f) * ps + margin;
            float b3 = (ax + x_scale[k] * 3.0f / 2.0f) * ps + margin;
            b0 = b0 >= 0 ? b0 : 0.0f;                                     // :1364
            b1 = b1 >= 0 ? b1 : 0.0f;
            b2 = b2 >= 0 ? b2 : 0.0f;
            b3 = b3 >= 0 ? b3 : 0.0f;
            b1 = (b1 < (float)(32 * height + 256)) ? b1 : board1;          // :1365
            b3 = (b3 < (float)(32 * width + 256)) ? b3 : board3;           // :1366
            xsn[2 * k] = (b1 - b0 + 1.0f) / 96.0f;                         // :1367
            xsn[2 * k + 1] = 1.0f;                                         // :1378-1381
            ysn[2 * k] = (b3 - b2 + 1.0f) / 96.0f;                         // :1368
            ysn[2 * k + 1] = 1.0f;
            l0 = (long long)b0; l1 = (long long)b1; l2 = (long long)b2; l3 = (long long)b3;  // :1369
            avn[2 * k + 1] = (float)(l1 + l0) / 2.0f - 128.0f + 0.5f;      // :1371
            avn[2 * k + 0] = (float)(l2 + l3) / 2.0f - 128.0f + 0.5f;      // :1372
            flag = ifn[k] ? 0 : 1;
        }
        // ordered compaction: ballot within the wave, scan of wave totals across the block
        const unsigned long long mask = __ballot(flag);
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(mask);
        wg_barrier();
        int wbase = base_s;
        for (int q = 0; q < wave; ++q) wbase += wave_tot[q];
        if (flag) {
            const int64_t o = (int64_t)(wbase + before) * 5;
            bound5[o + 0] = l0; bound5[o + 1] = l1; bound5[o + 2] = l2; bound5[o + 3] = l3;
            bound5[o + 4] = (int64_t)img * 10000 + k;                      // :1374-1377
        }
        wg_barrier();
        if (tid == 0) {
            int tot = 0;
            for (int q = 0; q < 16; ++q) tot += wave_tot[q];
            base_s += tot;
        }
        wg_barrier();
    }
    if (tid == 0 && K_out) *K_out = base_s;
}

__global__ void __launch_bounds__(1024)
imgs_bounds_kernel(const float* __restrict__ x_scale, const float* __restrict__ y_scale,
                   const float* __restrict__ average_point, const uint8_t* __restrict__ ifn, int Np,
                   int height, int width, int img, int64_t* __restrict__ bound5,
                   int64_t* __restrict__ K_out, float* __restrict__ xsn, float* __restrict__ ysn,
                   float* __restrict__ avn) {
    imgs_bounds_block(x_scale, y_scale, average_point, ifn, Np, height, width, img, bound5, K_out, xsn, ysn, avn);
}

// A batch of images in one launch, no host-side counts: block i first sums the match flags of the
// images before it (its row offset into the compacted bound table; flags are one byte per patch, so
// even hundreds of images cost a few hundred KB of L2 reads), then compacts its own patches.
// K_img[i] = matches of image i; K_total (written by the last block) = rows of bound5 that are valid.
__global__ void __launch_bounds__(1024)
imgs_bounds_batch_kernel(const float* __restrict__ x_scale, const float* __restrict__ y_scale,
                         const float* __restrict__ average_point, const uint8_t* __restrict__ ifn, int Np,
                         int height, int width, int64_t* __restrict__ bound5, int64_t* __restrict__ K_img,
                         int64_t* __restrict__ K_total, float* __restrict__ xsn, float* __restrict__ ysn,
                         float* __restrict__ avn) {
    __shared__ int part[16];
    __shared__ int64_t off_s;
    const int img = blockIdx.x, tid = threadIdx.x;
    int cnt = 0;
    for (int64_t k = tid; k < (int64_t)img * Np; k += 1024) cnt += ifn[k] ? 0 : 1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) cnt += __shfl_xor(cnt, o);
    if ((tid & 63) == 0) part[tid >> 6] = cnt;
    wg_barrier();
    if (tid == 0) {
        int64_t o = 0;
        for (int q = 0; q < 16; ++q) o += part[q];
        off_s = o;
    }
    wg_barrier();
    const int64_t off = off_s;
    wg_barrier();
    const int64_t i = img;
    imgs_bounds_block(x_scale + i * Np, y_scale + i * Np, average_point + i * Np * 2, ifn + i * Np, Np, height, width,
                      img, bound5 + off * 5, K_img + i, xsn + i * Np * 2, ysn + i * Np * 2, avn + i * Np * 2);
    if (tid == 0 && img == (int)gridDim.x - 1 && K_total) *K_total = off + K_img[i];
}

// ---- left crops: 96x96 windows on the fixed grid of the 32-px zero-padded left image ---------
template <bool NT>
__global__ void __launch_bounds__(288)
left_crops_kernel(const float* __restrict__ left_all, int n_img, int H, int W, const int64_t* __restrict__ bound5,
                  int width, float* __restrict__ out, const int64_t* __restrict__ K_dev) {
    const int64_t k = blockIdx.x;
    if (K_dev && k >= *K_dev) return;          // launched over the capacity: rows past the device-side count
    const int64_t seq = bound5[k * 5 + 4];                 // img * 10000 + patch, utils.py:1374-1377
    const int patch = (int)(seq % 10000);
    const int64_t img = min(max(seq / 10000, (int64_t)0), (int64_t)n_img - 1);
    const float* left = left_all + img * (int64_t)H * W * 3;
    const int r = patch / width, c = patch - r * width;
    // a row of the window is 288 contiguous floats of the source (or zeros): 72 lanes x 16 bytes, four rows per pass of
    // the 288 threads; the source offset is only 4-byte aligned in general (W * 3 floats per image row)
    const int t = threadIdx.x, j = t % 72, sub = t / 72;
    typedef float f4a __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f4s __attribute__((ext_vector_type(4)));
    float* o = out + k * (96 * 96 * 3);
    const int ix0 = c * 32 - 32;                // first source pixel of the row
    for (int y = blockIdx.y * 8 + sub; y < blockIdx.y * 8 + 8; y += 4) {
        const int iy = r * 32 + y - 32;
        f4s v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H) {
            const float* row = left + (int64_t)iy * W * 3;
            const int e0 = ix0 * 3 + 4 * j;                            // element offset inside the source row
            if (e0 >= 0 && e0 + 3 < W * 3) {
                const f4a u = *reinterpret_cast<const f4a*>(row + e0);
                v = f4s{u.x, u.y, u.z, u.w};
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = e0 + q;
                    if (e >= 0 && e < W * 3) v[q] = row[e];
                }
            }
        }
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4s*>(o + y * 288 + 4 * j));
        else *reinterpret_cast<f4s*>(o + y * 288 + 4 * j) = v;
    }
}

// ---- tensor_resize ---------------------------------------------------------------------------
struct CropGeom {
    long long y0, x0, ih, iw, img;
    bool ok;
};
__device__ __forceinline__ CropGeom crop_geom(const int64_t* __restrict__ bound, int64_t i, int n_img,
                                              int Hp, int Wp) {
    CropGeom g;
    g.y0 = bound[i * 5];
    const long long y1 = bound[i * 5 + 1];
    g.x0 = bound[i * 5 + 2];
    const long long x1 = bound[i * 5 + 3];
    g.img = bound[i * 5 + 4] / 10000;            // library.cpp:55-56
    g.ih = y1 - g.y0;                            // narrow(1, y0, y1 - y0)        library.cpp:57-58
    g.iw = x1 - g.x0 + 1;                        // narrow(2, x0, x1 - x0 + 1)    library.cpp:58-59
    g.ok = g.ih > 0 && g.iw > 0 && g.y0 >= 0 && g.x0 >= 0 && g.y0 + g.ih <= Hp &&
           g.x0 + g.iw <= Wp && g.img >= 0 && g.img < n_img;
    return g;
}

// ATen upsample_bilinear2d(align_corners=true): scale = (in-1)/(out-1); src = scale * dst
struct Tap {
    int i1, ip;
    float l0, l1;
};
__device__ __forceinline__ Tap make_tap(float scale, int dst, long long in_size) {
    const float f = scale * (float)dst;
    Tap t;
    t.i1 = (int)f;
    t.ip = (t.i1 < in_size - 1) ? 1 : 0;
    t.l1 = f - (float)t.i1;
    t.l0 = 1.0f - t.l1;
    return t;
}

// CHW source (the padded tensor of utils.py:1352), CHW output [K,C,96,96]
__global__ void __launch_bounds__(256)
resize_chw_kernel(const float* __restrict__ input, int n_img, int C, int Hp, int Wp,
                  const int64_t* __restrict__ bound, float* __restrict__ out,
                  int32_t* __restrict__ status) {
    const int64_t i = blockIdx.x;
    const int c = blockIdx.y, slab = blockIdx.z;
    const CropGeom g = crop_geom(bound, i, n_img, Hp, Wp);
    float* o = out + ((i * C + c) * 96) * 96;
    if (!g.ok) {
        if (status && threadIdx.x == 0) atomicOr(status, 1);
        for (int idx = threadIdx.x; idx < 24 * 96; idx += 256) o[slab * 24 * 96 + idx] = 0.f;
        return;
    }
    const float sh = (float)(g.ih - 1) / 95.0f, sw = (float)(g.iw - 1) / 95.0f;
    const float* src = input + (((int64_t)g.img * C + c) * Hp + g.y0) * Wp + g.x0;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int idx = threadIdx.x + 256 * q;
        const int oy = slab * 24 + idx / 96, ox = idx % 96;
        const Tap ty = make_tap(sh, oy, g.ih), tx = make_tap(sw, ox, g.iw);
        const float* p = src + (int64_t)ty.i1 * Wp + tx.i1;
        const float p00 = p[0], p01 = p[tx.ip], p10 = p[(int64_t)ty.ip * Wp],
                    p11 = p[(int64_t)ty.ip * Wp + tx.ip];
        o[oy * 96 + ox] = ty.l0 * (tx.l0 * p00 + tx.l1 * p01) + ty.l1 * (tx.l0 * p10 + tx.l1 * p11);
    }
}

// HWC unpadded source [n_img,H,W,3] with a virtual zero margin, HWC output [K,96,96,3]:
// fuses F.pad (utils.py:1352), the NCHW permute and the caller's permute(0,2,3,1) (utils.py:1385)
template <bool NT>
__global__ void __launch_bounds__(256)
resize_hwc_kernel(const float* __restrict__ right, int n_img, int H, int W, int margin,
                  const int64_t* __restrict__ bound, float* __restrict__ out,
                  int32_t* __restrict__ status, const int64_t* __restrict__ K_dev) {
    const int64_t i = blockIdx.x;
    if (K_dev && i >= *K_dev) return;
    const int slab = blockIdx.y;
    const int Hp = H + 2 * margin, Wp = W + 2 * margin;
    const CropGeom g = crop_geom(bound, i, n_img, Hp, Wp);
    float* o = out + i * (96 * 96 * 3);
    if (!g.ok) {
        if (status && threadIdx.x == 0) atomicOr(status, 1);
        for (int idx = threadIdx.x; idx < 24 * 288; idx += 256) o[slab * 24 * 288 + idx] = 0.f;
        return;
    }
    const float sh = (float)(g.ih - 1) / 95.0f, sw = (float)(g.iw - 1) / 95.0f;
    const float* img = right + (int64_t)g.img * H * W * 3;
    // one output PIXEL per thread and pass (taps, bounds tests and addresses once for the three channels; the three
    // floats of a tap are 12 contiguous bytes); coordinates fit 32 bits once the crop geometry has been validated
    const int y0 = (int)g.y0 - margin, x0 = (int)g.x0 - margin, ih = (int)g.ih, iw = (int)g.iw;
    struct Px { float r, g, b; };
    auto at = [&](int y, int x) {
        Px v{0.f, 0.f, 0.f};
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const float* q = img + ((int64_t)y * W + x) * 3;
            v.r = q[0]; v.g = q[1]; v.b = q[2];
        }
        return v;
    };
    for (int px = threadIdx.x; px < 24 * 96; px += 256) {
        const int oy = slab * 24 + px / 96, ox = px % 96;
        const Tap ty = make_tap(sh, oy, ih), tx = make_tap(sw, ox, iw);
        const int y = y0 + ty.i1, x = x0 + tx.i1;
        const Px p00 = at(y, x), p01 = at(y, x + tx.ip), p10 = at(y + ty.ip, x), p11 = at(y + ty.ip, x + tx.ip);
        typedef float f3a __attribute__((ext_vector_type(3), aligned(4)));
        const f3a d = {ty.l0 * (tx.l0 * p00.r + tx.l1 * p01.r) + ty.l1 * (tx.l0 * p10.r + tx.l1 * p11.r),
                       ty.l0 * (tx.l0 * p00.g + tx.l1 * p01.g) + ty.l1 * (tx.l0 * p10.g + tx.l1 * p11.g),
                       ty.l0 * (tx.l0 * p00.b + tx.l1 * p01.b) + ty.l1 * (tx.l0 * p10.b + tx.l1 * p11.b)};
        f3a* dp = reinterpret_cast<f3a*>(o + oy * 288 + 3 * ox);       // one 12-byte store: a wave's 64 pixels are 768 contiguous bytes
        if (NT) __builtin_nontemporal_store(d, dp);
        else *dp = d;
    }
}

// the crops are written once and read by another kernel much later (2.3 GB per side and step): non-temporal stores of whole
// lines (see gather.hip).  PATS_CROPS_NT = 0 / 1, read once per process.
static bool crops_nt() {
    static const bool nt = [] { const char* e = env_switch("PATS_CROPS_NT"); return e ? atoi(e) != 0 : PATS_CROPS_NT_DEFAULT != 0; }();
    return nt;
}

}  // namespace pats

using namespace pats;

extern "C" int pats_compute_imgs_bounds_f32(const float* x_scale, const float* y_scale,
                                            const float* average_point, const uint8_t* if_nomatching,
                                            int Np, int height, int width, int img, int64_t* bound5,
                                            int64_t* K_out, float* x_scale_new, float* y_scale_new,
                                            float* average_new, pats_stream_t stream) {
    PATS_REQUIRE(Np > 0 && height > 0 && width > 0, "compute_imgs_bounds: bad shape");
    PATS_REQUIRE(x_scale && y_scale && average_point && if_nomatching && bound5 && K_out &&
                     x_scale_new && y_scale_new && average_new, "compute_imgs_bounds: null pointer");
    hipLaunchKernelGGL(imgs_bounds_kernel, dim3(1), dim3(1024), 0, as_stream(stream), x_scale,
                       y_scale, average_point, if_nomatching, Np, height, width, img, bound5, K_out,
                       x_scale_new, y_scale_new, average_new);
    return check_launch("imgs_bounds_kernel");
}

extern "C" int pats_left_crops_f32(const float* left, int n_img, int H, int W, const int64_t* bound5, int64_t K,
                                   int height, int width, float* out, pats_stream_t stream) {
    PATS_REQUIRE(K >= 0 && n_img > 0 && H > 0 && W > 0 && height > 0 && width > 0, "left_crops: bad shape");
    if (K == 0) return PATS_OK;
    PATS_REQUIRE(left && bound5 && out, "left_crops: null pointer");
    if (crops_nt()) hipLaunchKernelGGL(left_crops_kernel<true>, dim3((unsigned)K, 12), dim3(288), 0, as_stream(stream),
                                       left, n_img, H, W, bound5, width, out, (const int64_t*)nullptr);
    else hipLaunchKernelGGL(left_crops_kernel<false>, dim3((unsigned)K, 12), dim3(288), 0, as_stream(stream),
                            left, n_img, H, W, bound5, width, out, (const int64_t*)nullptr);
    return check_launch("left_crops_kernel");
}

extern "C" int pats_compute_imgs_bounds_batch_f32(const float* x_scale, const float* y_scale,
                                                  const float* average_point, const uint8_t* if_nomatching,
                                                  int n_img, int Np, int height, int width, int64_t* bound5,
                                                  int64_t* K_img, int64_t* K_total, float* x_scale_new,
                                                  float* y_scale_new, float* average_new, pats_stream_t stream) {
    PATS_REQUIRE(n_img > 0 && n_img <= 10000 && Np > 0 && height > 0 && width > 0, "compute_imgs_bounds_batch: bad shape");
    PATS_REQUIRE(x_scale && y_scale && average_point && if_nomatching && bound5 && K_img && x_scale_new &&
                     y_scale_new && average_new, "compute_imgs_bounds_batch: null pointer");
    hipLaunchKernelGGL(imgs_bounds_batch_kernel, dim3((unsigned)n_img), dim3(1024), 0, as_stream(stream), x_scale,
                       y_scale, average_point, if_nomatching, Np, height, width, bound5, K_img, K_total,
                       x_scale_new, y_scale_new, average_new);
    return check_launch("imgs_bounds_batch_kernel");
}

extern "C" int pats_left_crops_counted_f32(const float* left, int n_img, int H, int W, const int64_t* bound5,
                                           int64_t K_cap, const int64_t* K_dev, int height, int width, float* out,
                                           pats_stream_t stream) {
    PATS_REQUIRE(K_cap >= 0 && n_img > 0 && H > 0 && W > 0 && height > 0 && width > 0, "left_crops_counted: bad shape");
    if (K_cap == 0) return PATS_OK;
    PATS_REQUIRE(left && bound5 && out && K_dev, "left_crops_counted: null pointer");
    if (crops_nt()) hipLaunchKernelGGL(left_crops_kernel<true>, dim3((unsigned)K_cap, 12), dim3(288), 0, as_stream(stream),
                                       left, n_img, H, W, bound5, width, out, K_dev);
    else hipLaunchKernelGGL(left_crops_kernel<false>, dim3((unsigned)K_cap, 12), dim3(288), 0, as_stream(stream),
                            left, n_img, H, W, bound5, width, out, K_dev);
    return check_launch("left_crops_kernel");
}

extern "C" int pats_tensor_resize_f32(const float* input, int n_img, int C, int Hp, int Wp,
                                      const int64_t* bound, int64_t K, float* out, int32_t* status,
                                      pats_stream_t stream) {
    PATS_REQUIRE(K >= 0 && n_img > 0 && C > 0 && Hp > 0 && Wp > 0, "tensor_resize: bad shape");
    if (K == 0) return PATS_OK;     // empty [0,C,96,96] result, like the reference when nothing matches
    PATS_REQUIRE(input && bound && out, "tensor_resize: null pointer");
    PATS_REQUIRE(C <= 65535, "tensor_resize: too many channels");
    hipLaunchKernelGGL(resize_chw_kernel, dim3((unsigned)K, (unsigned)C, 4), dim3(256), 0,
                       as_stream(stream), input, n_img, C, Hp, Wp, bound, out, status);
    return check_launch("resize_chw_kernel");
}

extern "C" int pats_tensor_resize_hwc_f32(const float* right, int n_img, int H, int W, int margin,
                                          const int64_t* bound, int64_t K, float* out,
                                          int32_t* status, pats_stream_t stream) {
    PATS_REQUIRE(K >= 0 && n_img > 0 && H > 0 && W > 0 && margin >= 0, "tensor_resize_hwc: bad shape");
    if (K == 0) return PATS_OK;
    PATS_REQUIRE(right && bound && out, "tensor_resize_hwc: null pointer");
    if (crops_nt()) hipLaunchKernelGGL(resize_hwc_kernel<true>, dim3((unsigned)K, 4), dim3(256), 0, as_stream(stream), right,
                                       n_img, H, W, margin, bound, out, status, (const int64_t*)nullptr);
    else hipLaunchKernelGGL(resize_hwc_kernel<false>, dim3((unsigned)K, 4), dim3(256), 0, as_stream(stream), right,
                            n_img, H, W, margin, bound, out, status, (const int64_t*)nullptr);
    return check_launch("resize_hwc_kernel");
}

extern "C" int pats_tensor_resize_hwc_counted_f32(const float* right, int n_img, int H, int W, int margin,
                                                  const int64_t* bound, int64_t K_cap, const int64_t* K_dev,
                                                  float* out, int32_t* status, pats_stream_t stream) {
    PATS_REQUIRE(K_cap >= 0 && n_img > 0 && H > 0 && W > 0 && margin >= 0, "tensor_resize_hwc_counted: bad shape");
    if (K_cap == 0) return PATS_OK;
    PATS_REQUIRE(right && bound && out && K_dev, "tensor_resize_hwc_counted: null pointer");
    if (crops_nt()) hipLaunchKernelGGL(resize_hwc_kernel<true>, dim3((unsigned)K_cap, 4), dim3(256), 0, as_stream(stream), right,
                                       n_img, H, W, margin, bound, out, status, K_dev);
    else hipLaunchKernelGGL(resize_hwc_kernel<false>, dim3((unsigned)K_cap, 4), dim3(256), 0, as_stream(stream), right,
                            n_img, H, W, margin, bound, out, status, K_dev);
    return check_launch("resize_hwc_kernel");
}
