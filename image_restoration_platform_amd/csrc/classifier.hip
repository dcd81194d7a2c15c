// classifier.hip -- fused degradation-classifier scan for gfx950 (MI355X).
//
// Replaces the seven sharp/libvips pipelines + JS reductions behind
// ClassifierService.analyze (server-node/src/services/classifier.js:40-337) with ONE pass
// over the decoded RGB bytes:
//   grey (libvips B_W, integer LUT form)                      classifier.js:108,136,200
//   3x3 Laplacian-8 / high-pass-9 / Laplacian-4, u8 clip       classifier.js:109-113,137-141,201-205
//   separable sigma=1 integer Gaussian on RGB                  classifier.js:297
//   per-channel sum / sum^2 (stats())                          classifier.js:52
//   stride-4 scratch probes                                    classifier.js:316-333
// All accumulators are exact integers (u64 atomics => order independent => bit-exact
// scores vs the CPU oracle).  Roofline: HBM, 3*H*W algorithmic bytes per image.
//
// Data layout: rgb is [N][H][W][3] u8, tightly packed.  One workgroup walks a strided set
// of 16x64 output tiles of ONE image (blockIdx.y); each tile is staged with a 1-pixel
// replicate-clamped halo into LDS as four u8 planes (R,G,B,grey); the horizontal blur
// pass goes through LDS as well (libvips rounds to u8 between the two passes).
#include "classifier.hpp"

#include "classifier_finalize.hpp"

namespace ire {

namespace {

constexpr int CT_H = 16, CT_W = 64;             // output tile
constexpr int CH_H = CT_H + 2, CH_W = CT_W + 2; // halo tile
constexpr int CPITCH = 68;                      // u8 plane row pitch (bytes)
constexpr int NBUCKETS = 5001;

struct ClsLds {
    unsigned int wR[256], wG[256], wB[256];
    unsigned int thr[260];
    unsigned char inv[5008];
    unsigned char pl[4][CH_H][CPITCH];  // R, G, B, grey
    unsigned char hb[3][CH_H][CT_W];    // horizontally blurred R,G,B for rows -1..CT_H
};

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned int lo = (unsigned int)v, hi = (unsigned int)(v >> 32);
        unsigned int olo = __shfl_xor(lo, off, 64), ohi = __shfl_xor(hi, off, 64);
        v += ((unsigned long long)ohi << 32) | olo;
    }
    return v;
}

__global__ __launch_bounds__(256) void classifier_scan_kernel(const uint8_t* __restrict__ rgb, int H, int W,
                                                              int tiles_x, int tiles_y,
                                                              const unsigned int* __restrict__ g_lin16,
                                                              const unsigned int* __restrict__ g_thr,
                                                              const unsigned char* __restrict__ g_inv,
                                                              unsigned long long* __restrict__ sums) {
    __shared__ ClsLds L;
    __shared__ unsigned long long red[4][CLS_NSUMS];
    const int tid = threadIdx.x;
    const int img = blockIdx.y;
    const uint8_t* base = rgb + (size_t)img * H * W * 3;

    // tables -> LDS once per workgroup (pre-multiplied by the luminance weights x10000)
    {
        unsigned int l = g_lin16[tid];
        L.wR[tid] = 2126u * l;
        L.wG[tid] = 7152u * l;
        L.wB[tid] = 722u * l;
        L.thr[tid] = g_thr[tid];
        if (tid == 0) L.thr[256] = g_thr[256];
        for (int i = tid; i < NBUCKETS; i += 256) L.inv[i] = g_inv[i];
    }
    __syncthreads();

    unsigned long long acc[CLS_NSUMS];
#pragma unroll
    for (int i = 0; i < CLS_NSUMS; ++i) acc[i] = 0;

    const int ntiles = tiles_x * tiles_y;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int y0 = ty * CT_H, x0 = tx * CT_W;

        // phase A: halo tile (replicate-clamped) -> R,G,B,grey planes
        for (int i = tid; i < CH_H * CH_W; i += 256) {
            int py = i / CH_W, px = i - py * CH_W;
            int gy = min(max(y0 + py - 1, 0), H - 1);
            int gx = min(max(x0 + px - 1, 0), W - 1);
            const uint8_t* p = base + ((size_t)gy * W + gx) * 3;
            unsigned int r = p[0], g = p[1], b = p[2];
            unsigned int y = L.wR[r] + L.wG[g] + L.wB[b];
            unsigned int gv = L.inv[y >> 17];
            gv += (y >= L.thr[gv + 1]) ? 1u : 0u;
            L.pl[0][py][px] = (unsigned char)r;
            L.pl[1][py][px] = (unsigned char)g;
            L.pl[2][py][px] = (unsigned char)b;
            L.pl[3][py][px] = (unsigned char)gv;
        }
        __syncthreads();

        // phase B: horizontal {12,20,12}/44 pass, rounded to u8 (libvips integer convsep)
        for (int i = tid; i < 3 * CH_H * CT_W; i += 256) {
            int c = i / (CH_H * CT_W);
            int rem = i - c * (CH_H * CT_W);
            int py = rem / CT_W, x = rem - py * CT_W;
            unsigned int v = 12u * L.pl[c][py][x] + 20u * L.pl[c][py][x + 1] + 12u * L.pl[c][py][x + 2];
            L.hb[c][py][x] = (unsigned char)((v + 22u) / 44u);
        }
        __syncthreads();

        // phase C: per-pixel accumulation, thread -> column x, rows (tid>>6) + 4k
        {
            const int x = tid & 63;
            const int gx = x0 + x;
            unsigned int s_c[3] = {0, 0, 0}, q_c[3] = {0, 0, 0};
            unsigned int s_b = 0, q_b = 0, s8 = 0, q8 = 0, s9 = 0, q9 = 0;
            if (gx < W) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int y = (tid >> 6) + 4 * k;
                    if (y0 + y < H) {
                        const int hy = y + 1, hx = x + 1;
                        int c = L.pl[3][hy][hx];
                        int sum9 = L.pl[3][hy - 1][hx - 1] + L.pl[3][hy - 1][hx] + L.pl[3][hy - 1][hx + 1] +
                                   L.pl[3][hy][hx - 1] + c + L.pl[3][hy][hx + 1] +
                                   L.pl[3][hy + 1][hx - 1] + L.pl[3][hy + 1][hx] + L.pl[3][hy + 1][hx + 1];
                        unsigned int e8 = (unsigned int)min(max(9 * c - sum9, 0), 255);
                        unsigned int e9 = (unsigned int)min(max(10 * c - sum9, 0), 255);
                        s8 += e8; q8 += e8 * e8;
                        s9 += e9; q9 += e9 * e9;
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            unsigned int v = L.pl[ch][hy][hx];
                            s_c[ch] += v; q_c[ch] += v * v;
                            unsigned int bv = 12u * L.hb[ch][hy - 1][x] + 20u * L.hb[ch][hy][x] + 12u * L.hb[ch][hy + 1][x];
                            bv = (bv + 22u) / 44u;
                            s_b += bv; q_b += bv * bv;
                        }
                    }
                }
            }
            acc[0] += s_c[0]; acc[1] += s_c[1]; acc[2] += s_c[2];
            acc[3] += q_c[0]; acc[4] += q_c[1]; acc[5] += q_c[2];
            acc[6] += s_b; acc[7] += q_b;
            acc[8] += s8; acc[9] += q8;
            acc[10] += s9; acc[11] += q9;
        }
        // scratch probes: pixels on the stride-4 grid (tile origin is a multiple of 4)
        if (tid < 64) {
            const int y = (tid >> 4) * 4, x = (tid & 15) * 4;
            const int gy = y0 + y, gx = x0 + x;
            if (gy < H && gx < W) {
                auto e4 = [&](int yy, int xx) -> int {
                    const int hy = yy + 1, hx = xx + 1;
                    int v = 4 * L.pl[3][hy][hx] - L.pl[3][hy - 1][hx] - L.pl[3][hy + 1][hx] -
                            L.pl[3][hy][hx - 1] - L.pl[3][hy][hx + 1];
                    return min(max(v, 0), 255);
                };
                if (e4(y, x) > 200) {
                    if (gx + 1 < W) acc[12] += e4(y, x + 1) > 200 ? 1 : 0;
                    if (gy + 1 < H) acc[13] += e4(y + 1, x) > 200 ? 1 : 0;
                }
            }
        }
        __syncthreads();  // planes are rewritten by the next tile
    }

    // workgroup reduction -> 14 u64 atomics per workgroup
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int i = 0; i < CLS_NSUMS; ++i) {
        unsigned long long v = wave_sum_u64(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (tid < CLS_NSUMS) {
        unsigned long long v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        if (v) atomicAdd(&sums[(size_t)img * CLS_NSUMS + tid], v);
    }
}

__global__ void classifier_finalize_kernel(const unsigned long long* __restrict__ sums,
                                           const uint8_t* __restrict__ is_jpeg, int n, int H, int W,
                                           double* __restrict__ scores, int32_t* __restrict__ label,
                                           float* __restrict__ cond) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t S[CLS_NSUMS];
    for (int k = 0; k < CLS_NSUMS; ++k) S[k] = sums[(size_t)i * CLS_NSUMS + k];
    double sc[7];
    int32_t lb;
    cls_finalize_one(S, (uint64_t)H * (uint64_t)W, is_jpeg ? is_jpeg[i] : 1, sc, &lb);
    for (int k = 0; k < 7; ++k) {
        if (scores) scores[(size_t)i * 7 + k] = sc[k];
        if (cond) cond[(size_t)i * 8 + k] = (float)sc[k];
    }
    if (cond) cond[(size_t)i * 8 + 7] = 0.f;
    if (label) label[i] = lb;
}

__global__ void scores_to_cond_kernel(const double* __restrict__ scores, int n, float* __restrict__ cond) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 8) return;
    int img = i >> 3, k = i & 7;
    cond[i] = k < 7 ? (float)scores[(size_t)img * 7 + k] : 0.f;
}

}  // namespace

void classifier_launch(const ClassifierTables& tb, const uint8_t* d_rgb, int n, int h, int w,
                       const uint8_t* d_is_jpeg, unsigned long long* d_sums, double* d_scores,
                       int32_t* d_label, float* d_cond, hipStream_t stream) {
    IRE_HIP(hipMemsetAsync(d_sums, 0, sizeof(unsigned long long) * CLS_NSUMS * n, stream));
    const int tiles_x = ceil_div(w, CT_W), tiles_y = ceil_div(h, CT_H);
    const int ntiles = tiles_x * tiles_y;
    // ~4 workgroups per CU chip-wide; every workgroup amortises its 9 KB table load over its tiles
    int per_img = std::max(1, std::min(ntiles, 1024 / std::max(1, n)));
    dim3 grid(per_img, n);
    hipLaunchKernelGGL(classifier_scan_kernel, grid, dim3(256), 0, stream, d_rgb, h, w, tiles_x, tiles_y,
                       tb.lin16, tb.thr, tb.inv, d_sums);
    IRE_HIP(hipGetLastError());
    hipLaunchKernelGGL(classifier_finalize_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, stream, d_sums, d_is_jpeg,
                       n, h, w, d_scores, d_label, d_cond);
    IRE_HIP(hipGetLastError());
}

void scores_to_cond_launch(const double* d_scores, int n, float* d_cond, hipStream_t stream) {
    hipLaunchKernelGGL(scores_to_cond_kernel, dim3(ceil_div(n * 8, 64)), dim3(64), 0, stream, d_scores, n, d_cond);
    IRE_HIP(hipGetLastError());
}

}  // namespace ire
