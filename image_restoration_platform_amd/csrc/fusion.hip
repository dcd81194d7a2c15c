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

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "engine.hpp"

namespace ire {

namespace {

constexpr int CR = 4;     // coarse search radius (quarter-res pixels)
constexpr int FR = 3;     // fine search radius (full-res pixels)
constexpr int FM = 20;    // fine margin
constexpr int NC = (2 * CR + 1) * (2 * CR + 1);  // 81
constexpr int NF = (2 * FR + 1) * (2 * FR + 1);  // 49

// `qp` = pitch of the quarter-res planes in bytes (W/4 rounded up to a multiple of 4: the SAD kernel stages dwords).
__global__ void fusion_luma_kernel(const uint8_t* __restrict__ rgb, int k, int H, int W, int qp, uint8_t* __restrict__ L,
                                   uint8_t* __restrict__ Q) {
    const int Hq = H >> 2, Wq = W >> 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k * Hq * Wq) return;
    const size_t set = blockIdx.y;                       // view set of a batched call: its own views, luma planes, quarter planes
    rgb += set * (size_t)k * H * W * 3; L += set * (size_t)3 * H * W; Q += set * (size_t)3 * Hq * qp;
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
    Q[((size_t)v * Hq + yq) * qp + xq] = (uint8_t)((sum + 8u) >> 4);
}

// SAD of the reference plane against view v's plane at every candidate shift, and -- in the launch's last workgroup --
// the winning shift of every view.
// MODE 0: coarse on Q (all interior pixels, +-CR).  MODE 1: fine on L (every 2nd pixel, +-FR around 4*coarse).
//
// A workgroup walks tiles of SR sampled rows x TW reference pixels.  Per tile it stages the reference rows and the view's
// rows (+-R, displaced by the coarse shift) in LDS as dwords -- both planes have a pitch that is a multiple of 4 and every
// tile origin is too (margins 4 / 20, coarse shifts x 4), so the staging is dword loads -- and a thread then owns ONE
// dword of reference pixels at a time: per candidate row it reads three view dwords and forms the nine (seven) shifted
// windows with v_alignbyte, v_sad_u8 accumulating into that candidate's register (the fine search masks bytes 1 and 3:
// every 2nd pixel).  The margins keep every window inside the plane: no clamping in the arithmetic (staging loads clamp
// their ADDRESS at the plane's right / bottom edge; those bytes are masked).  Integer sums: any order gives the same result
// (bit-exact against oracle/fusion.py).
//
// Workgroup sums go to the workgroup's own row of `part`; the last workgroup of the launch (ticket) adds the rows and picks
// min (SAD, |dy|+|dx|, dy, dx) per view: no atomics on the 81 / 49 totals, no memset, no second launch.  (Same device-scope
// store / wait / ticket sequence as classifier.hip.)
template <int MODE>
struct SadCfg {
    static constexpr int R = MODE == 0 ? CR : FR, N = MODE == 0 ? NC : NF, M = MODE == 0 ? CR : FM, STEP = MODE == 0 ? 1 : 2;
    static constexpr int TW = MODE == 0 ? 128 : 256;        // tile width in reference pixels
    static constexpr int SR = MODE == 0 ? 8 : 16;           // sampled reference rows per tile
    static constexpr int TWD = TW / 4;                      // reference dwords per row
    static constexpr int VR = (SR - 1) * STEP + 1 + 2 * R;  // view rows staged
    static constexpr int VWD = TWD + 2;                     // view dwords per row (window offsets 0..8 from dword j: j .. j+2)
    static constexpr int ITEMS = SR * TWD / 256;
    static_assert(SR * TWD % 256 == 0 && M % 4 == 0, "tile shape");
};

constexpr int FUSE_SAD_GRID = 512;   // workgroups per view at most (= rows of `part` per view)

template <int MODE>
__global__ __launch_bounds__(256) void fusion_sad_kernel(const uint8_t* __restrict__ P, int k, int PH, int PW, int pitch,
                                                         int* __restrict__ coarse, int* __restrict__ shifts,
                                                         int* __restrict__ shifts_out, unsigned* __restrict__ part,
                                                         unsigned* __restrict__ ticket) {
    using C = SadCfg<MODE>;
    constexpr int R = C::R, N = C::N, M = C::M, STEP = C::STEP, D = 2 * R + 1;
    __shared__ unsigned s_ref[C::SR][C::TWD];
    __shared__ unsigned s_view[C::VR][C::VWD];
    __shared__ unsigned s_sad[2][N];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const int v = blockIdx.y + 1;
    {   // view set blockIdx.z of a batched call: its own planes, shifts, partial rows and ticket
        const size_t set = blockIdx.z;
        P += set * (size_t)3 * PH * pitch;
        coarse += set * 6; shifts += set * 6;
        if (shifts_out) shifts_out += set * 2 * k;
        part += set * (size_t)2 * FUSE_SAD_GRID * NC;
        ticket += set;
    }
    for (int i = tid; i < 2 * N; i += 256) (&s_sad[0][0])[i] = 0;
    const int by = MODE == 0 ? 0 : 4 * coarse[v * 2], bx = MODE == 0 ? 0 : 4 * coarse[v * 2 + 1];
    const uint8_t* P0 = P;
    const uint8_t* Pv = P + (size_t)v * PH * pitch;
    const int pd = pitch >> 2;
    const int tiles_x = (PW - 2 * M + C::TW - 1) / C::TW;
    const int tiles_y = ((PH - 2 * M + STEP - 1) / STEP + C::SR - 1) / C::SR;
    const int wave = tid >> 6, lane = tid & 63;
    for (int tile = blockIdx.x; tile < tiles_x * tiles_y; tile += gridDim.x) {   // uniform per workgroup
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int gy0 = M + ty * C::SR * STEP, gx0 = M + tx * C::TW;
        __syncthreads();   // the previous tile's readers are done (and s_sad is zero on the first pass)
#pragma unroll
        for (int it = 0; it < C::ITEMS; ++it) {
            const int i = it * 256 + tid, r = i / C::TWD, j = i - r * C::TWD;
            const int yy = min(gy0 + r * STEP, PH - 1), xd = min((gx0 >> 2) + j, pd - 1);
            s_ref[r][j] = reinterpret_cast<const unsigned*>(P0 + (size_t)yy * pitch)[xd];
        }
        for (int i = tid; i < C::VR * C::VWD; i += 256) {
            const int r = i / C::VWD, j = i - r * C::VWD;
            const int yy = min(gy0 + by - R + r, PH - 1), xd = min(((gx0 + bx - 4) >> 2) + j, pd - 1);
            s_view[r][j] = reinterpret_cast<const unsigned*>(Pv + (size_t)yy * pitch)[xd];
        }
        __syncthreads();
        // The four waves split the candidate ROWS; a wave sweeps the tile once per row with its D accumulators.  (All N
        // accumulators per thread in one unrolled pass is 4400 instructions executed once, one wave per SIMD: 35 us per
        // launch against 12-18 us for this form, profiles/r02_experiments.md.)
#pragma unroll 1
        for (int dy = -R + wave; dy <= R; dy += 4) {
            unsigned acc[D];
#pragma unroll
            for (int c = 0; c < D; ++c) acc[c] = 0u;
#pragma unroll 2
            for (int i = lane; i < C::SR * C::TWD; i += 64) {
                const int r = i / C::TWD, j = i - r * C::TWD;
                const int y = gy0 + r * STEP, x = gx0 + 4 * j;
                unsigned mask = 0;
                if (y < PH - M) {
#pragma unroll
                    for (int b = 0; b < 4; b += STEP)
                        if (x + b < PW - M) mask |= 0xffu << (8 * b);
                }
                const unsigned a = s_ref[r][j] & mask;
                const unsigned* row = &s_view[r * STEP + dy + R][j];
                const unsigned w0 = row[0], w1 = row[1], w2 = row[2];
#pragma unroll
                for (int dx = -R; dx <= R; ++dx) {
                    const int o = dx + 4;   // byte offset of the window from dword j
                    const unsigned win = o < 4 ? __builtin_amdgcn_alignbyte(w1, w0, o & 3)
                                       : o < 8 ? __builtin_amdgcn_alignbyte(w2, w1, o & 3) : w2;
                    acc[dx + R] = __builtin_amdgcn_sad_u8(a, win & mask, acc[dx + R]);
                }
            }
#pragma unroll
            for (int c = 0; c < D; ++c) {
                unsigned val = acc[c];
                val += (unsigned)__builtin_amdgcn_update_dpp(0, (int)val, 0x121, 0xf, 0xf, false);   // row_ror:1
                val += (unsigned)__builtin_amdgcn_update_dpp(0, (int)val, 0x122, 0xf, 0xf, false);   // row_ror:2
                val += (unsigned)__builtin_amdgcn_update_dpp(0, (int)val, 0x124, 0xf, 0xf, false);   // row_ror:4
                val += (unsigned)__builtin_amdgcn_update_dpp(0, (int)val, 0x128, 0xf, 0xf, false);   // row_ror:8 -> row sums
                if ((lane & 15) == 0) atomicAdd(&s_sad[0][(dy + R) * D + c], val);   // one lane per 16-lane row
            }
        }
    }
    __syncthreads();
    const int G = gridDim.x;
    for (int i = tid; i < N; i += 256)
        __hip_atomic_store(&part[((size_t)(v - 1) * G + blockIdx.x) * N + i], s_sad[0][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
        s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(G * gridDim.y - 1);
    __syncthreads();
    if (!s_last) return;
    for (int i = tid; i < 2 * N; i += 256) (&s_sad[0][0])[i] = 0;
    __syncthreads();
    const int total = (k - 1) * G * N;
    for (int base = 0; base < total; base += 256 * 8) {    // eight loads in flight per thread (one round trip each otherwise);
        unsigned t[8];                                     // LDS atomics on N addresses per view
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 256 + tid;
            t[u] = i < total ? __hip_atomic_load(&part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 256 + tid;
            if (i < total) atomicAdd(&s_sad[i / (G * N)][i % N], t[u]);
        }
    }
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-cleaning
        if (MODE == 0) { coarse[0] = 0; coarse[1] = 0; }
        else {
            shifts[0] = 0; shifts[1] = 0;
            if (shifts_out) { shifts_out[0] = 0; shifts_out[1] = 0; }
        }
    }
    if (wave < k - 1) {   // one wave per view: min over (SAD, |dy|+|dx|, dy, dx) as one 64-bit key (candidate index ascends with (dy, dx))
        const int vv = wave + 1;
        unsigned long long best = ~0ull;
        for (int c = lane; c < N; c += 64) {
            const int dy = c / D - R, dx = c % D - R;
            const unsigned man = (unsigned)((dy < 0 ? -dy : dy) + (dx < 0 ? -dx : dx));
            const unsigned long long key = ((unsigned long long)s_sad[vv - 1][c] << 16) | (man << 8) | (unsigned)c;
            best = key < best ? key : best;
        }
        for (int off = 32; off; off >>= 1) {
            const unsigned long long o = __shfl_xor(best, off);
            best = o < best ? o : best;
        }
        if (lane == 0) {
            const int c = (int)(best & 0xffu), bdy = c / D - R, bdx = c % D - R;
            if (MODE == 0) { coarse[vv * 2] = bdy; coarse[vv * 2 + 1] = bdx; }
            else {
                const int sy = 4 * coarse[vv * 2] + bdy, sx = 4 * coarse[vv * 2 + 1] + bdx;
                shifts[vv * 2] = sy; shifts[vv * 2 + 1] = sx;
                if (shifts_out) { shifts_out[vv * 2] = sy; shifts_out[vv * 2 + 1] = sx; }
            }
        }
    }
}

struct FuseWlut { unsigned w[256]; };   // the blend weights ride in the kernel arguments: no host-to-device copy

__global__ __launch_bounds__(256) void fusion_blend_kernel(const uint8_t* __restrict__ rgb, int k, int H, int W,
                                                           const int* __restrict__ shifts,
                                                           const FuseWlut wlut, const unsigned* __restrict__ wluts, uint8_t* __restrict__ out) {
    __shared__ unsigned s_w[256];
    const size_t set = blockIdx.y;                       // view set of a batched call; its blend table comes from `wluts` (one call: kernel arguments)
    s_w[threadIdx.x] = wluts ? wluts[set * 256 + threadIdx.x] : wlut.w[threadIdx.x];
    rgb += set * (size_t)k * H * W * 3; out += set * (size_t)H * W * 3; shifts += set * 6;
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

void Engine::fuse_launch(const uint8_t* d_views, int nsets, int k, int h, int w, const unsigned* host_wluts, uint8_t* d_out,
                         int32_t* d_shifts, hipStream_t s) {
    const size_t px = (size_t)h * w;
    const int hq = h / 4, wq = w / 4, qp = (wq + 3) & ~3;
    if (fuse_cap_px_ < px || fuse_cap_sets_ < nsets) {
        IRE_HIP(hipDeviceSynchronize());
        for (void* p : {(void*)d_fL_, (void*)d_fQ_, (void*)d_fsad_, (void*)d_fmisc_, (void*)d_fwlut_}) if (p) (void)hipFree(p);
        const size_t cpx = std::max(px, fuse_cap_px_);
        const int cs = std::max(nsets, fuse_cap_sets_);
        // per view set: 3 luma planes, 3 quarter-res planes (pitch: a multiple of 4),
        // per-workgroup SAD rows (coarse and fine take turns), coarse[3][2] + shifts[3][2], a ticket, a blend table
        d_fL_ = (uint8_t*)dalloc((size_t)cs * 3 * cpx);
        d_fQ_ = (uint8_t*)dalloc((size_t)cs * 3 * (cpx / 16 + cpx / 64 + 16) + 64);      // (w >= 64: the pitch adds at most 3 to a row of >= 16)
        d_fsad_ = (unsigned*)dalloc(sizeof(unsigned) * (size_t)cs * 2 * FUSE_SAD_GRID * NC);
        d_fmisc_ = (int*)dalloc(sizeof(int) * (size_t)cs * 16);
        d_fwlut_ = (unsigned*)dalloc(sizeof(unsigned) * (size_t)cs * 256);
        IRE_HIP(hipMemsetAsync(d_fmisc_, 0, sizeof(int) * (size_t)cs * 16, s));   // the tickets start at zero; the kernels reset them
        fuse_cap_px_ = cpx; fuse_cap_sets_ = cs;
    }
    // d_fmisc_: [sets][6] coarse | [sets][6] shifts | [sets] tickets
    int* d_coarse = d_fmisc_;
    int* d_sh = d_fmisc_ + 6 * fuse_cap_sets_;
    unsigned* d_ticket = reinterpret_cast<unsigned*>(d_fmisc_ + 12 * fuse_cap_sets_);
    FuseWlut lut;
    std::memcpy(lut.w, host_wluts, sizeof(lut.w));
    prof_begin(FAM_FUSION, s, 0, (double)nsets * (k + 1) * px * 3);
    const unsigned* d_wluts = nullptr;
    if (nsets > 1) {      // one table per set (its own noise score); a single call carries its table in the kernel arguments
        IRE_HIP(hipMemcpyAsync(d_fwlut_, host_wluts, sizeof(unsigned) * 256 * nsets, hipMemcpyHostToDevice, s));   // pageable source: staged before the call returns
        d_wluts = d_fwlut_;
    }
    const int nq = k * hq * wq;
    hipLaunchKernelGGL(fusion_luma_kernel, dim3(ceil_div(nq, 256), nsets), dim3(256), 0, s, d_views, k, h, w, qp, d_fL_, d_fQ_);
    auto tiles = [](int ph, int pw, int m, int step, int tw, int sr) {
        return ceil_div(pw - 2 * m, tw) * ceil_div(ceil_div(ph - 2 * m, step), sr);
    };
    const int g0 = std::min(FUSE_SAD_GRID, tiles(hq, wq, CR, 1, SadCfg<0>::TW, SadCfg<0>::SR));
    const int g1 = std::min(FUSE_SAD_GRID, tiles(h, w, FM, 2, SadCfg<1>::TW, SadCfg<1>::SR));
    hipLaunchKernelGGL(fusion_sad_kernel<0>, dim3(g0, k - 1, nsets), dim3(256), 0, s, d_fQ_, k, hq, wq, qp, d_coarse, d_sh, (int*)nullptr,
                       d_fsad_, d_ticket);
    hipLaunchKernelGGL(fusion_sad_kernel<1>, dim3(g1, k - 1, nsets), dim3(256), 0, s, d_fL_, k, h, w, w, d_coarse, d_sh, (int*)d_shifts,
                       d_fsad_, d_ticket);
    hipLaunchKernelGGL(fusion_blend_kernel, dim3(ceil_div((int)px, 256), nsets), dim3(256), 0, s, d_views, k, h, w, d_sh, lut, d_wluts, d_out);
    IRE_HIP(hipGetLastError());
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
    fuse_batch_device(E, d_rgb_views, 1, k, h, w, &noise_score, d_out_rgb, d_shifts, stream);
}

// `nsets` independent view sets in one call: d_rgb_views [nsets][k][h][w][3], noise_scores [nsets] on the HOST (< 0: classify
// view 0 of that set), d_out_rgb [nsets][h][w][3], d_shifts [nsets][k][2] or null.  Every kernel of the chain takes the set as
// a grid dimension, so a batch costs the chain's four dependent launches once.
void fuse_batch_device(Engine& E, const uint8_t* d_rgb_views, int nsets, int k, int h, int w, const double* noise_scores,
                       uint8_t* d_out_rgb, int32_t* d_shifts, hipStream_t stream) {
    check_fuse_args(k, h, w);
    if (nsets < 1 || nsets > kFuseMaxSets) fail(IRE_ERR_INVALID_INPUT, "invalid batch size for fusion: expected 1..16 view sets");
    if (!d_rgb_views || !d_out_rgb || !noise_scores) fail(IRE_ERR_INVALID_INPUT, "invalid input: null pointer");
    std::vector<unsigned> luts((size_t)nsets * 256);
    for (int i = 0; i < nsets; ++i) {
        double ns = noise_scores[i];
        if (ns < 0) ns = E.noise_of_view0(d_rgb_views + (size_t)i * k * h * w * 3, h, w, stream);
        make_wlut(ns, luts.data() + (size_t)i * 256);
    }
    E.fuse_launch(d_rgb_views, nsets, k, h, w, luts.data(), d_out_rgb, d_shifts, stream);
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
    fuse_launch(d_in_, 1, k, h, w, lut, d_out_, nullptr, s);
    IRE_HIP(hipEventRecord(ev_[3], s));
    IRE_HIP(hipMemcpyAsync(out_rgb, d_out_, px * 3, hipMemcpyDeviceToHost, s));
    if (shifts_out) IRE_HIP(hipMemcpyAsync(shifts_out, d_fmisc_ + 6 * fuse_cap_sets_, sizeof(int) * 2 * k, hipMemcpyDeviceToHost, s));
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
