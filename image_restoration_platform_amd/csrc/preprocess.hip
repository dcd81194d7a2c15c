// Preprocess step in front of the hot path: the pixel work of
// server-node/src/middleware/imagePreprocess.js:24-91 -- EXIF auto-orient (:43) and fit-inside-2048 Lanczos-3 (:46-55).
// The JPEG q85 4:4:4 encode (:57-64) stays with the host codec (sharp in the Node deployment, Pillow behind FastAPI).
//
// Arithmetic: the published 8-bit resampler of Pillow (two passes, horizontal then vertical; taps normalised and
// rounded to 22-bit integers; one rounding to u8 per pass) -- integer multiply-adds, so the GPU result is bit-exact
// against oracle/preprocess.py, which tests/test_preprocess.py pins bit-exactly against Pillow itself.  libvips' own
// reducer (the reference's dependency) is not in this image: parity with it is unpinned.
//
// Both kernels are HBM-bound byte work (read each stored pixel once, write the intermediate once, read it once, write
// the result): a thread produces one output pixel (3 bytes) from <= ksize taps; the orientation is folded into the
// horizontal pass's reads, so there is no separate transpose pass.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <vector>

#include "engine.hpp"

namespace ire {

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

double lanczos3(double x) {
    if (x >= -3.0 && x < 3.0) {
        if (x == 0.0) return 1.0;
        const double px = x * M_PI;
        const double a = std::sin(px) / px;
        const double py = px / 3.0;
        return a * (std::sin(py) / py);
    }
    return 0.0;
}

// bounds[out][2] = (first, count), taps[out][ksize]
int make_taps(int in_size, int out_size, std::vector<int32_t>& bounds, std::vector<int32_t>& taps) {
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 3.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    taps.assign((size_t)out_size * ksize, 0);
    const double ss = 1.0 / filterscale;
    std::vector<double> w(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) { w[x] = lanczos3((x + xmin - center + 0.5) * ss); ww += w[x]; }
        for (int x = 0; x < xmax; ++x) {
            const double k = ww != 0.0 ? w[x] / ww : w[x];
            taps[(size_t)xx * ksize + x] = k < 0 ? (int32_t)(k * (1 << kPrecisionBits) - 0.5) : (int32_t)(k * (1 << kPrecisionBits) + 0.5);
        }
        bounds[(size_t)xx * 2] = xmin;
        bounds[(size_t)xx * 2 + 1] = xmax;
    }
    return ksize;
}

__device__ __forceinline__ int clip8(int v) { v >>= kPrecisionBits; return v < 0 ? 0 : (v > 255 ? 255 : v); }

// upright (y, x) -> stored pixel index, EXIF orientation 1..8; sh, sw = stored height, width
__device__ __forceinline__ size_t stored_index(int orientation, int y, int x, int sh, int sw) {
    int sy, sx;
    switch (orientation) {
        case 2: sy = y; sx = sw - 1 - x; break;
        case 3: sy = sh - 1 - y; sx = sw - 1 - x; break;
        case 4: sy = sh - 1 - y; sx = x; break;
        case 5: sy = x; sx = y; break;
        case 6: sy = sh - 1 - x; sx = y; break;
        case 7: sy = sh - 1 - x; sx = sw - 1 - y; break;
        case 8: sy = x; sx = sw - 1 - y; break;
        default: sy = y; sx = x; break;
    }
    return (size_t)sy * sw + sx;
}

// horizontal pass over the UPRIGHT image (uh x uw) read through the orientation map; out: uh x ow
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, int sh, int sw, int orientation, int uh,
                                                         int ow, int ksize, const int32_t* __restrict__ bounds,
                                                         const int32_t* __restrict__ taps, uint8_t* __restrict__ dst) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (xx >= ow || y >= uh) return;
    const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int32_t* k = taps + (size_t)xx * ksize;
    int r = 1 << (kPrecisionBits - 1), g = r, b = r;
    for (int t = 0; t < n; ++t) {
        const uint8_t* p = src + stored_index(orientation, y, x0 + t, sh, sw) * 3;
        const int kv = k[t];
        r += p[0] * kv; g += p[1] * kv; b += p[2] * kv;
    }
    uint8_t* o = dst + ((size_t)y * ow + xx) * 3;
    o[0] = (uint8_t)clip8(r); o[1] = (uint8_t)clip8(g); o[2] = (uint8_t)clip8(b);
}

// vertical pass: in uh x ow -> out oh x ow; threads run along x (coalesced rows)
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* __restrict__ src, int ow, int oh, int ksize,
                                                         const int32_t* __restrict__ bounds, const int32_t* __restrict__ taps,
                                                         uint8_t* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, yy = blockIdx.y;      // i = byte column in [0, ow*3)
    if (i >= ow * 3 || yy >= oh) return;
    const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int32_t* k = taps + (size_t)yy * ksize;
    int acc = 1 << (kPrecisionBits - 1);
    for (int t = 0; t < n; ++t) acc += src[(size_t)(y0 + t) * ow * 3 + i] * k[t];
    dst[(size_t)yy * ow * 3 + i] = (uint8_t)clip8(acc);
}

// orientation only (no size change): one byte-triplet copy per upright pixel
__global__ __launch_bounds__(256) void orient_kernel(const uint8_t* __restrict__ src, int sh, int sw, int orientation, int uh, int uw,
                                                     uint8_t* __restrict__ dst) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= uw || y >= uh) return;
    const uint8_t* p = src + stored_index(orientation, y, x, sh, sw) * 3;
    uint8_t* o = dst + ((size_t)y * uw + x) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}

long js_round(double x) { return (long)std::floor(x + 0.5); }

}  // namespace

// imagePreprocess.js:12-22,46-55 -- the box comes from the STORED dimensions (sharp's metadata() ignores EXIF), the upright
// image is then fitted inside it without enlargement.
void preprocess_plan(int width, int height, int orientation, int max_dim, int* out_w, int* out_h, int* resized) {
    if (width <= 0 || height <= 0 || orientation < 1 || orientation > 8 || max_dim < 1)
        fail(IRE_ERR_INVALID_INPUT, "invalid arguments to preprocess: width, height > 0, orientation in 1..8, max_dim >= 1");
    const bool swap = orientation >= 5;
    const int uw = swap ? height : width, uh = swap ? width : height;
    if (width <= max_dim && height <= max_dim) { *out_w = uw; *out_h = uh; if (resized) *resized = 0; return; }
    const double scale = (double)max_dim / (width > height ? width : height);
    const double bw = (double)js_round(width * scale), bh = (double)js_round(height * scale);
    double s2 = bw / uw;
    if (bh / uh < s2) s2 = bh / uh;
    if (s2 > 1.0) s2 = 1.0;
    long ow = js_round(uw * s2), oh = js_round(uh * s2);
    *out_w = ow < 1 ? 1 : (int)ow;
    *out_h = oh < 1 ? 1 : (int)oh;
    if (resized) *resized = 1;
}

void Engine::preprocess_device(const uint8_t* d_rgb, int h, int w, int orientation, int max_dim, uint8_t* d_out, int out_h, int out_w,
                               hipStream_t s) {
    int pw = 0, ph = 0, resized = 0;
    preprocess_plan(w, h, orientation, max_dim, &pw, &ph, &resized);
    if (pw != out_w || ph != out_h) fail(IRE_ERR_INVALID_INPUT, "invalid output size for preprocess: use ire_preprocess_plan");
    if (!d_rgb || !d_out) fail(IRE_ERR_INVALID_INPUT, "invalid arguments to preprocess: null buffer");
    const bool swap = orientation >= 5;
    const int uw = swap ? h : w, uh = swap ? w : h;
    if (!resized) {
        if (orientation == 1) { IRE_HIP(hipMemcpyAsync(d_out, d_rgb, (size_t)h * w * 3, hipMemcpyDeviceToDevice, s)); return; }
        hipLaunchKernelGGL(orient_kernel, dim3((uw + 255) / 256, uh), dim3(256), 0, s, d_rgb, h, w, orientation, uh, uw, d_out);
        IRE_HIP(hipGetLastError());
        return;
    }
    // taps are rebuilt per call (two small tables, <= a few hundred KiB): the step runs once per upload
    std::vector<int32_t> bh_, th_, bv_, tv_;
    const int kh = make_taps(uw, out_w, bh_, th_);
    const int kv = make_taps(uh, out_h, bv_, tv_);
    const size_t need_tab = (bh_.size() + th_.size() + bv_.size() + tv_.size()) * sizeof(int32_t);
    const size_t need_mid = (size_t)uh * out_w * 3;
    if (need_tab > pp_tab_cap_) { if (d_pp_tab_) IRE_HIP(hipFree(d_pp_tab_)); d_pp_tab_ = (int32_t*)dalloc(need_tab); pp_tab_cap_ = need_tab; }
    if (need_mid > pp_mid_cap_) { if (d_pp_mid_) IRE_HIP(hipFree(d_pp_mid_)); d_pp_mid_ = (uint8_t*)dalloc(need_mid); pp_mid_cap_ = need_mid; }
    int32_t* d_bh = d_pp_tab_;
    int32_t* d_th = d_bh + bh_.size();
    int32_t* d_bv = d_th + th_.size();
    int32_t* d_tv = d_bv + bv_.size();
    // the tables live in pageable host vectors: copy synchronously with respect to the host (hipMemcpyAsync from
    // pageable memory returns after staging), ordered on `s` before the kernels
    IRE_HIP(hipMemcpyAsync(d_bh, bh_.data(), bh_.size() * 4, hipMemcpyHostToDevice, s));
    IRE_HIP(hipMemcpyAsync(d_th, th_.data(), th_.size() * 4, hipMemcpyHostToDevice, s));
    IRE_HIP(hipMemcpyAsync(d_bv, bv_.data(), bv_.size() * 4, hipMemcpyHostToDevice, s));
    IRE_HIP(hipMemcpyAsync(d_tv, tv_.data(), tv_.size() * 4, hipMemcpyHostToDevice, s));
    IRE_HIP(hipStreamSynchronize(s));      // the vectors die at return
    hipLaunchKernelGGL(resample_h_kernel, dim3((out_w + 255) / 256, uh), dim3(256), 0, s, d_rgb, h, w, orientation, uh, out_w, kh, d_bh,
                       d_th, d_pp_mid_);
    hipLaunchKernelGGL(resample_v_kernel, dim3((out_w * 3 + 255) / 256, out_h), dim3(256), 0, s, d_pp_mid_, out_w, out_h, kv, d_bv, d_tv,
                       d_out);
    IRE_HIP(hipGetLastError());
}

void Engine::preprocess_host(const uint8_t* rgb, int h, int w, int orientation, int max_dim, uint8_t* out, int out_h, int out_w) {
    int pw = 0, ph = 0;
    preprocess_plan(w, h, orientation, max_dim, &pw, &ph, nullptr);
    if (pw != out_w || ph != out_h) fail(IRE_ERR_INVALID_INPUT, "invalid output size for preprocess: use ire_preprocess_plan");
    if (!rgb || !out) fail(IRE_ERR_INVALID_INPUT, "invalid arguments to preprocess: null buffer");
    const size_t in_bytes = (size_t)h * w * 3, out_bytes = (size_t)out_h * out_w * 3;
    if (in_bytes > pp_in_cap_) { if (d_pp_in_) IRE_HIP(hipFree(d_pp_in_)); d_pp_in_ = (uint8_t*)dalloc(in_bytes); pp_in_cap_ = in_bytes; }
    if (out_bytes > pp_out_cap_) { if (d_pp_out_) IRE_HIP(hipFree(d_pp_out_)); d_pp_out_ = (uint8_t*)dalloc(out_bytes); pp_out_cap_ = out_bytes; }
    hipStream_t s = main_stream_;
    IRE_HIP(hipMemcpyAsync(d_pp_in_, rgb, in_bytes, hipMemcpyHostToDevice, s));
    preprocess_device(d_pp_in_, h, w, orientation, max_dim, d_pp_out_, out_h, out_w, s);
    IRE_HIP(hipMemcpyAsync(out, d_pp_out_, out_bytes, hipMemcpyDeviceToHost, s));
    IRE_HIP(hipStreamSynchronize(s));
}

}  // namespace ire
