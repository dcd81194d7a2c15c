// fusion.hip -- Fusion-v0: align <= 3 views of one scene to view 0 and blend them (gfx950).
//
// BUILD-DEFINED: the reference has no fusion code -- its docs hand <= 3
// images to the provider in one restoreImage call (geminiClient.js:32,49;
// image-restoration-platform.md:787-857) -- so this file defines the pixel-space step that stands
// behind that seam.  Everything is integer arithmetic => bit-exact against oracle/fusion.py.
//
//   luma     L = (77 R + 150 G + 29 B + 128) >> 8 ;  quarter-res Q = (sum of 4x4 L + 8) >> 4
//   coarse   for each view v >= 1: SAD of Q0 vs Qv shifted by (dy,dx) in [-4,4]^2 over the interior
//            (4-pixel margin); winner = min (SAD, |dy|+|dx|, dy, dx)            -> +-16 px
//   fine     +-3 px around 4*coarse on full-res L, every 2nd pixel, 20-pixel margin; same tie-break
//   blend    per channel: x_v = view_v[y+dy_v][x+dx_v] (replicate-clamped), m = median (k=3) or
//            rounded mean (k=2), w_v = WLUT[|x_v - m|]; out = (sum w_v x_v + (sum w)/2) / sum w
//            WLUT[d] = max(1, round(1024 exp(-d^2 / (2 sigma^2)))), sigma = 4 + 40 * noise score
// Sign convention: the aligned sample of view v for reference pixel (y,x) is view_v[y+dy_v][x+dx_v].
// HBM-bound: k*3*H*W bytes in, 3*H*W out (+ the u8 luma planes: k*H*W written once, read <= twice).
#include "fusion.hpp"

#include <cmath>
#include <vector>

#include "engine.hpp"

namespace ire {

namespace {

constexpr int CR = 4;     // coarse search radius (quarter-res pixels)
constexpr int FR = 3;     // fine search radius (full-res pixels)
constexpr int FM = 20;    // fine margin
constexpr int NC = (2 * CR + 1) * (2 * CR + 1);  // 81
constexpr int NF = (2 * FR + 1) * (2 * FR + 1);  // 49

__global__ void fusion_luma_kernel(const uint8_t* __restrict__ rgb, int k, int H, int W, uint8_t* __restrict__ L,
                                   uint8_t* __restrict__ Q) {
    const int Hq = H >> 2, Wq = W >> 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k * Hq * Wq) return;
    const int v = i / (Hq * Wq), rem = i - v * Hq * Wq;
    const int yq = rem / Wq, xq = rem - yq * Wq;
    unsigned sum = 0;
    for (int dy = 0; dy < 4; ++dy)
        for (int dx = 0; dx < 4; ++dx) {
            const size_t p = ((size_t)v * H + (yq * 4 + dy)) * W + xq * 4 + dx;
            const uint8_t* px = rgb + p * 3;
            const unsigned l = (77u * px[0] + 150u * px[1] + 29u * px[2] + 128u) >> 8;
            L[p] = (uint8_t)l;
            sum += l;
        }
    Q[i] = (uint8_t)((sum + 8u) >> 4);
}

// SAD of the reference plane against view v's plane at every candidate shift.
// MODE 0: coarse on Q (all interior pixels, +-CR).  MODE 1: fine on L (every 2nd pixel, +-FR around 4*coarse).
template <int MODE>
__global__ __launch_bounds__(256) void fusion_sad_kernel(const uint8_t* __restrict__ P, int k, int PH, int PW,
                                                         const int* __restrict__ coarse, unsigned* __restrict__ sad) {
    constexpr int R = MODE == 0 ? CR : FR, N = MODE == 0 ? NC : NF, M = MODE == 0 ? CR : FM, STEP = MODE == 0 ? 1 : 2;
    __shared__ unsigned s_sad[N];
    const int v = blockIdx.y + 1;
    for (int i = threadIdx.x; i < N; i += 256) s_sad[i] = 0;
    __syncthreads();
    const int nx = (PW - 2 * M + STEP - 1) / STEP, ny = (PH - 2 * M + STEP - 1) / STEP;
    const int by = MODE == 0 ? 0 : 4 * coarse[v * 2], bx = MODE == 0 ? 0 : 4 * coarse[v * 2 + 1];
    const uint8_t* P0 = P;
    const uint8_t* Pv = P + (size_t)v * PH * PW;
    // per-thread partial SAD of every candidate shift in registers (N = 81 / 49), accumulated over this thread's pixels;
    // lanes are then summed by DPP row rotations + permlane swaps and each wave adds N values to LDS once.  Integer sums:
    // any order gives the same result (bit-exact against oracle/fusion.py).
    unsigned acc[N];
#pragma unroll
    for (int c = 0; c < N; ++c) acc[c] = 0u;
    for (int base = blockIdx.x * 256; base < nx * ny; base += gridDim.x * 256) {   // uniform trip count per block
        const int i = base + threadIdx.x;
        const bool valid = i < nx * ny;
        const int ii = valid ? i : 0;
        const int y = M + (ii / nx) * STEP, x = M + (ii % nx) * STEP;
        const int a = P0[(size_t)y * PW + x];
#pragma unroll
        for (int dy = -R; dy <= R; ++dy) {
            const int yy = min(max(y + by + dy, 0), PH - 1);
            const uint8_t* row = Pv + (size_t)yy * PW;
#pragma unroll
            for (int dx = -R; dx <= R; ++dx) {
                const int xx = min(max(x + bx + dx, 0), PW - 1);
                const int d = a - (int)row[xx];
                acc[(dy + R) * (2 * R + 1) + dx + R] += valid ? (unsigned)(d < 0 ? -d : d) : 0u;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < N; ++c) {
        unsigned val = acc[c];
        val += (unsigned)__builtin_amdgcn_update_dpp(0, (int)val, 0x121, 0xf, 0xf, false);   // row_ror:1
        val += (unsigned)__builtin_amdgcn_update_dpp(0, (int)val, 0x122, 0xf, 0xf, false);   // row_ror:2
        val += (unsigned)__builtin_amdgcn_update_dpp(0, (int)val, 0x124, 0xf, 0xf, false);   // row_ror:4
        val += (unsigned)__builtin_amdgcn_update_dpp(0, (int)val, 0x128, 0xf, 0xf, false);   // row_ror:8 -> row sums
        acc[c] = val;
    }
    if ((threadIdx.x & 15) == 0) {            // one lane per 16-lane row
#pragma unroll
        for (int c = 0; c < N; ++c) atomicAdd(&s_sad[c], acc[c]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 256)
        if (s_sad[i]) atomicAdd(&sad[(v - 1) * N + i], s_sad[i]);
}

template <int MODE>
__global__ void fusion_pick_kernel(const unsigned* __restrict__ sad, int k, int* __restrict__ coarse,
                                   int* __restrict__ shifts) {
    constexpr int R = MODE == 0 ? CR : FR, N = MODE == 0 ? NC : NF;
    const int v = threadIdx.x + 1;
    if (threadIdx.x == 0) {
        if (MODE == 0) { coarse[0] = 0; coarse[1] = 0; }
        else { shifts[0] = 0; shifts[1] = 0; }
    }
    if (v >= k) return;
    unsigned best = 0xffffffffu;
    int bdy = 0, bdx = 0, bman = 1 << 30;
    for (int dy = -R; dy <= R; ++dy)
        for (int dx = -R; dx <= R; ++dx) {
            const unsigned s = sad[(v - 1) * N + (dy + R) * (2 * R + 1) + dx + R];
            const int man = (dy < 0 ? -dy : dy) + (dx < 0 ? -dx : dx);
            // lexicographic (SAD, |dy|+|dx|, dy, dx); the scan order already yields ascending (dy, dx)
            if (s < best || (s == best && man < bman)) { best = s; bdy = dy; bdx = dx; bman = man; }
        }
    if (MODE == 0) { coarse[v * 2] = bdy; coarse[v * 2 + 1] = bdx; }
    else { shifts[v * 2] = 4 * coarse[v * 2] + bdy; shifts[v * 2 + 1] = 4 * coarse[v * 2 + 1] + bdx; }
}

__global__ __launch_bounds__(256) void fusion_blend_kernel(const uint8_t* __restrict__ rgb, int k, int H, int W,
                                                           const int* __restrict__ shifts,
                                                           const unsigned* __restrict__ wlut, uint8_t* __restrict__ out) {
    __shared__ unsigned s_w[256];
    s_w[threadIdx.x] = wlut[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    const uint8_t* p[3];
    for (int v = 0; v < k; ++v) {
        const int yy = min(max(y + shifts[v * 2], 0), H - 1), xx = min(max(x + shifts[v * 2 + 1], 0), W - 1);
        p[v] = rgb + (((size_t)v * H + yy) * W + xx) * 3;
    }
    for (int c = 0; c < 3; ++c) {
        int a = p[0][c], b = p[1][c], m;
        unsigned num, den;
        if (k == 2) {
            m = (a + b + 1) >> 1;
            const unsigned wa = s_w[abs(a - m)], wb = s_w[abs(b - m)];
            num = wa * a + wb * b; den = wa + wb;
        } else {
            const int d = p[2][c];
            m = max(min(a, b), min(max(a, b), d));
            const unsigned wa = s_w[abs(a - m)], wb = s_w[abs(b - m)], wd = s_w[abs(d - m)];
            num = wa * a + wb * b + wd * d; den = wa + wb + wd;
        }
        out[(size_t)i * 3 + c] = (uint8_t)((num + (den >> 1)) / den);
    }
}

void make_wlut(double noise, unsigned* lut) {
    if (!(noise >= 0.0)) noise = 0.0;
    if (noise > 1.0) noise = 1.0;
    const double sigma = 4.0 + 40.0 * noise;
    for (int d = 0; d < 256; ++d) {
        const double w = std::floor(1024.0 * std::exp(-(double)(d * d) / (2.0 * sigma * sigma)) + 0.5);
        lut[d] = w < 1.0 ? 1u : (unsigned)w;
    }
}

void check_fuse_args(int k, int h, int w) {
    if (k < 2 || k > 3) fail(IRE_ERR_INVALID_INPUT, "invalid view count for fusion: expected 2..3");
    if (h < 64 || w < 64 || h % 8 || w % 8 || h > 8192 || w > 8192)
        fail(IRE_ERR_INVALID_INPUT, "invalid image size for fusion: height and width must be multiples of 8, >= 64");
}

}  // namespace

void Engine::fuse_launch(const uint8_t* d_views, int k, int h, int w, const unsigned* host_wlut, uint8_t* d_out,
                         int32_t* d_shifts, hipStream_t s) {
    const size_t px = (size_t)h * w, qpx = (size_t)(h / 4) * (w / 4);
    if (fuse_cap_px_ < px) {
        IRE_HIP(hipDeviceSynchronize());
        for (void* p : {(void*)d_fL_, (void*)d_fQ_, (void*)d_fsad_, (void*)d_fmisc_}) if (p) (void)hipFree(p);
        d_fL_ = (uint8_t*)dalloc(3 * px);
        d_fQ_ = (uint8_t*)dalloc(3 * qpx);
        d_fsad_ = (unsigned*)dalloc(sizeof(unsigned) * 2 * (NC + NF));
        d_fmisc_ = (int*)dalloc(sizeof(int) * (6 + 6 + 256));   // coarse[3][2], shifts[3][2], wlut[256]
        fuse_cap_px_ = px;
    }
    int* d_coarse = d_fmisc_;
    int* d_sh = d_fmisc_ + 6;
    unsigned* d_wlut = reinterpret_cast<unsigned*>(d_fmisc_ + 12);
    prof_begin(FAM_FUSION, s, 0, (double)(k + 1) * px * 3);
    IRE_HIP(hipMemcpyAsync(d_wlut, host_wlut, 256 * sizeof(unsigned), hipMemcpyHostToDevice, s));
    IRE_HIP(hipMemsetAsync(d_fsad_, 0, sizeof(unsigned) * 2 * (NC + NF), s));
    const int nq = k * (int)qpx;
    hipLaunchKernelGGL(fusion_luma_kernel, dim3(ceil_div(nq, 256)), dim3(256), 0, s, d_views, k, h, w, d_fL_, d_fQ_);
    hipLaunchKernelGGL(fusion_sad_kernel<0>, dim3(128, k - 1), dim3(256), 0, s, d_fQ_, k, h / 4, w / 4, d_coarse, d_fsad_);
    hipLaunchKernelGGL(fusion_pick_kernel<0>, dim3(1), dim3(64), 0, s, d_fsad_, k, d_coarse, d_sh);
    hipLaunchKernelGGL(fusion_sad_kernel<1>, dim3(512, k - 1), dim3(256), 0, s, d_fL_, k, h, w, d_coarse, d_fsad_ + 2 * NC);
    hipLaunchKernelGGL(fusion_pick_kernel<1>, dim3(1), dim3(64), 0, s, d_fsad_ + 2 * NC, k, d_coarse, d_sh);
    hipLaunchKernelGGL(fusion_blend_kernel, dim3(ceil_div((int)px, 256)), dim3(256), 0, s, d_views, k, h, w, d_sh, d_wlut, d_out);
    IRE_HIP(hipGetLastError());
    if (d_shifts) IRE_HIP(hipMemcpyAsync(d_shifts, d_sh, sizeof(int) * 2 * k, hipMemcpyDeviceToDevice, s));
    prof_end(s);
}

double Engine::noise_of_view0(const uint8_t* d_views, int h, int w, hipStream_t s) {
    ensure_io(1, 1, 1);
    classifier_launch(tables_, d_views, 1, h, w, nullptr, d_sums_, d_scores_, d_label_, d_cond_, s);
    double sc[7];
    IRE_HIP(hipMemcpyAsync(sc, d_scores_, sizeof(sc), hipMemcpyDeviceToHost, s));
    IRE_HIP(hipStreamSynchronize(s));   // the blend LUT is built on the host (exact double exp)
    return sc[IRE_SCORE_NOISE];
}

void fuse_device(Engine& E, const uint8_t* d_rgb_views, int k, int h, int w, double noise_score, uint8_t* d_out_rgb,
                 int32_t* d_shifts, hipStream_t stream) {
    check_fuse_args(k, h, w);
    if (!d_rgb_views || !d_out_rgb) fail(IRE_ERR_INVALID_INPUT, "invalid input: null image pointer");
    if (noise_score < 0) noise_score = E.noise_of_view0(d_rgb_views, h, w, stream);
    unsigned lut[256];
    make_wlut(noise_score, lut);
    E.fuse_launch(d_rgb_views, k, h, w, lut, d_out_rgb, d_shifts, stream);
}

void fuse_host(Engine& E, const uint8_t* rgb_views, int k, int h, int w, double noise_score, uint8_t* out_rgb,
               int32_t* shifts_out, ire_timings* t) {
    check_fuse_args(k, h, w);
    if (!rgb_views || !out_rgb) fail(IRE_ERR_INVALID_INPUT, "invalid input: null pointer");
    E.fuse_host_impl(rgb_views, k, h, w, noise_score, out_rgb, shifts_out, t);
}

void Engine::fuse_host_impl(const uint8_t* rgb_views, int k, int h, int w, double noise_score, uint8_t* out_rgb,
                            int32_t* shifts_out, ire_timings* t) {
    const size_t px = (size_t)h * w;
    ensure_io(3, h, w);   // views in d_in_, result in d_out_
    hipStream_t s = main_stream_;
    IRE_HIP(hipEventRecord(ev_[0], s));
    IRE_HIP(hipMemcpyAsync(d_in_, rgb_views, (size_t)k * px * 3, hipMemcpyHostToDevice, s));
    IRE_HIP(hipEventRecord(ev_[1], s));
    if (noise_score < 0) noise_score = noise_of_view0(d_in_, h, w, s);
    IRE_HIP(hipEventRecord(ev_[2], s));
    unsigned lut[256];
    make_wlut(noise_score, lut);
    fuse_launch(d_in_, k, h, w, lut, d_out_, nullptr, s);
    IRE_HIP(hipEventRecord(ev_[3], s));
    IRE_HIP(hipMemcpyAsync(out_rgb, d_out_, px * 3, hipMemcpyDeviceToHost, s));
    if (shifts_out) IRE_HIP(hipMemcpyAsync(shifts_out, d_fmisc_ + 6, sizeof(int) * 2 * k, hipMemcpyDeviceToHost, s));
    IRE_HIP(hipStreamSynchronize(s));
    if (t) {
        float a = 0, b = 0, c = 0;
        IRE_HIP(hipEventElapsedTime(&a, ev_[1], ev_[2]));
        IRE_HIP(hipEventElapsedTime(&b, ev_[2], ev_[3]));
        IRE_HIP(hipEventElapsedTime(&c, ev_[0], ev_[3]));
        t->classify_ms = a; t->restore_ms = b; t->total_ms = c;
    }
}

}  // namespace ire
