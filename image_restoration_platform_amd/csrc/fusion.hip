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
    if ((reinterpret_cast<uintptr_t>(rgb) & 3u) == 0) {
        // word-wide: a row of the block is 12 bytes = three aligned dwords (W % 8 == 0); a pixel's luma is ONE v_dot4_u32_u8 on
        // the window that starts at its R byte (weights 77, 150, 29, 0; + 128), its value byte 1 of the result (< 2^16)
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            const size_t p = ((size_t)v * H + (yq * 4 + dy)) * W + xq * 4;
            const unsigned* src = reinterpret_cast<const unsigned*>(rgb + p * 3);
            const unsigned d0 = src[0], d1 = src[1], d2 = src[2];
            constexpr unsigned LW = 0x001d964du;
            const unsigned l0 = __builtin_amdgcn_udot4(d0, LW, 128u, false);
            const unsigned l1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), LW, 128u, false);
            const unsigned l2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), LW, 128u, false);
            const unsigned l3 = __builtin_amdgcn_udot4(d2 >> 8, LW, 128u, false);
            const unsigned lw = __builtin_amdgcn_perm(l1, l0, 0x0c0c0501u) | __builtin_amdgcn_perm(l3, l2, 0x05010c0cu);   // bytes 1 of l0..l3
            *reinterpret_cast<unsigned*>(L + p) = lw;
            sum = __builtin_amdgcn_sad_u8(lw, 0u, sum);
        }
    } else {
        for (int dy = 0; dy < 4; ++dy)
            for (int dx = 0; dx < 4; ++dx) {
                const size_t p = ((size_t)v * H + (yq * 4 + dy)) * W + xq * 4 + dx;
                const uint8_t* px = rgb + p * 3;
                const unsigned l = (77u * px[0] + 150u * px[1] + 29u * px[2] + 128u) >> 8;
                L[p] = (uint8_t)l;
                sum += l;
            }
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

#ifndef FUSE_FL
#define FUSE_FL 16      // loads in flight per thread in the last workgroup's row sum (build-time A/B)
#endif
constexpr int FUSE_SAD_GRID = 512;   // workgroups per view at most (= rows of `part` per view)

template <int MODE>
__global__ __launch_bounds__(256) void fusion_sad_kernel(const uint8_t* __restrict__ P, int k, int PH, int PW, int pitch,
                                                         int* __restrict__ coarse, int* __restrict__ shifts,
                                                         int* __restrict__ shifts_out, unsigned* __restrict__ part,
                                                         unsigned* __restrict__ ticket) {
    using C = SadCfg<MODE>;
    constexpr int R = C::R, N = C::N, M = C::M, STEP = C::STEP, D = 2 * R + 1;
    // MODE 1 keeps only the SAMPLED pixels (every 2nd one): a reference dword = four samples = eight pixels, and the view's
    // rows as two planes -- its even and its odd pixels, interleaved per dword pair -- because candidate dx reads the plane of
    // dx's parity at a byte offset of (dx + 4) >> 1: one v_alignbyte and one v_sad_u8 per FOUR samples and candidate, no masks
    // (the image's interior is a whole number of such dwords: (PW - 2 M) % 8 == 0).
    constexpr int RWD = MODE == 0 ? C::TWD : C::TWD / 2;          // reference dwords per tile row
    constexpr int VPR = C::VWD / 2;                                // MODE 1: dword pairs per view row
    __shared__ __attribute__((aligned(16))) unsigned s_ref[C::SR][RWD];
    __shared__ __attribute__((aligned(16))) unsigned s_view[C::VR][C::VWD];                     // MODE 1: [row][pair][even, odd]
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
    // A tile's global loads are all issued together (one round trip), and the NEXT tile's before this tile's arithmetic.
    constexpr int NRL = MODE == 0 ? C::ITEMS : C::SR * RWD / 256;
    constexpr int NVL = MODE == 0 ? (C::VR * C::VWD + 255) / 256 : (C::VR * VPR + 255) / 256;
    constexpr int LW = MODE == 0 ? 1 : 2;                          // dwords per staged item
    unsigned tr[NRL][LW], tv[NVL][LW];
    auto load_tile = [&](int tile) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int gy0 = M + ty * C::SR * STEP, gx0 = M + tx * C::TW;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int it = 0; it < NRL; ++it) {
                const int i = it * 256 + tid, r = i / C::TWD, j = i - r * C::TWD;
                const int yy = min(gy0 + r * STEP, PH - 1), xd = min((gx0 >> 2) + j, pd - 1);
                tr[it][0] = reinterpret_cast<const unsigned*>(P0 + (size_t)yy * pitch)[xd];
            }
#pragma unroll
            for (int it = 0; it < NVL; ++it) {
                const int i = min(it * 256 + tid, C::VR * C::VWD - 1), r = i / C::VWD, j = i - r * C::VWD;
                const int yy = min(gy0 + by - R + r, PH - 1), xd = min(((gx0 + bx - 4) >> 2) + j, pd - 1);
                tv[it][0] = reinterpret_cast<const unsigned*>(Pv + (size_t)yy * pitch)[xd];
            }
        } else {
#pragma unroll
            for (int it = 0; it < NRL; ++it) {
                const int i = it * 256 + tid, r = i / RWD, c = i - r * RWD;
                const unsigned* src = reinterpret_cast<const unsigned*>(P0 + (size_t)min(gy0 + r * STEP, PH - 1) * pitch);
                tr[it][0] = src[min((gx0 >> 2) + 2 * c, pd - 1)]; tr[it][LW - 1] = src[min((gx0 >> 2) + 2 * c + 1, pd - 1)];
            }
#pragma unroll
            for (int it = 0; it < NVL; ++it) {
                const int i = min(it * 256 + tid, C::VR * VPR - 1), r = i / VPR, c = i - r * VPR;
                const unsigned* src = reinterpret_cast<const unsigned*>(Pv + (size_t)min(gy0 + by - R + r, PH - 1) * pitch);
                const int x = ((gx0 + bx - 4) >> 2) + 2 * c;
                tv[it][0] = src[min(x, pd - 1)]; tv[it][LW - 1] = src[min(x + 1, pd - 1)];
            }
        }
    };
    auto write_tile = [&]() {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int it = 0; it < NRL; ++it) (&s_ref[0][0])[it * 256 + tid] = tr[it][0];
#pragma unroll
            for (int it = 0; it < NVL; ++it)
                if (it * 256 + tid < C::VR * C::VWD) (&s_view[0][0])[it * 256 + tid] = tv[it][0];
        } else {
#pragma unroll
            for (int it = 0; it < NRL; ++it)
                (&s_ref[0][0])[it * 256 + tid] = __builtin_amdgcn_perm(tr[it][LW - 1], tr[it][0], 0x06040200u);      // pixels 0, 2, 4, 6 of the eight
#pragma unroll
            for (int it = 0; it < NVL; ++it)
                if (it * 256 + tid < C::VR * VPR)
                    reinterpret_cast<uint2*>(&s_view[0][0])[it * 256 + tid] =
                        make_uint2(__builtin_amdgcn_perm(tv[it][LW - 1], tv[it][0], 0x06040200u),
                                   __builtin_amdgcn_perm(tv[it][LW - 1], tv[it][0], 0x07050301u));
        }
    };
    const int ntiles = tiles_x * tiles_y;
    if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {   // uniform per workgroup
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int gy0 = M + ty * C::SR * STEP, gx0 = M + tx * C::TW;
        __syncthreads();   // the previous tile's readers are done (and s_sad is zero on the first pass)
        write_tile();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
        __syncthreads();
        // The four waves split the candidate ROWS; a wave sweeps the tile once per row with its D accumulators, in a LOOP.
        // Straight-line code is what these short kernels cannot afford -- every instruction is fetched cold, once: all N
        // accumulators per thread with the items unrolled (4400 instructions) took 35 us per launch (r02), all N accumulators
        // kept across the tiles with only the candidates unrolled (~2000 instructions with the final reduction) 26.5 / 22.8 us
        // (coarse / fine, one set) against 12.4 / 16.4 us for this form (profiles/r03_experiments.md).
#pragma unroll 1
        for (int dy = -R + wave; dy <= R; dy += 4) {
            unsigned acc[D];
#pragma unroll
            for (int c = 0; c < D; ++c) acc[c] = 0u;
            if constexpr (MODE == 0) {
#pragma unroll 2
                for (int i = lane; i < C::SR * C::TWD; i += 64) {
                    const int r = i / C::TWD, j = i - r * C::TWD;
                    const int y = gy0 + r * STEP, x = gx0 + 4 * j;
                    if (y >= PH - M || x >= PW - M) continue;
                    unsigned mask = 0xffffffffu;
                    if (x + 3 >= PW - M) {                   // the interior's last dword may be partial
                        mask = 0;
#pragma unroll
                        for (int b = 0; b < 4; ++b)
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
            } else {
                const int nr = min(C::SR, (PH - M - gy0 + 1) >> 1), nc = min(RWD, (PW - M - gx0) >> 3);
#pragma unroll 2
                for (int i = lane; i < C::SR * RWD; i += 64) {
                    const int r = i / RWD, c = i - r * RWD;
                    if (r >= nr || c >= nc) continue;
                    const unsigned a = s_ref[r][c];
                    const uint2 p0 = *reinterpret_cast<const uint2*>(&s_view[r * STEP + dy + R][2 * c]);
                    const uint2 p1 = *reinterpret_cast<const uint2*>(&s_view[r * STEP + dy + R][2 * c + 2]);
#pragma unroll
                    for (int dx = -R; dx <= R; ++dx) {
                        const int o = (dx + 4) >> 1;         // sample t of the dword is byte o + t of the plane of dx's parity
                        const unsigned lo = (dx & 1) ? p0.y : p0.x, hi = (dx & 1) ? p1.y : p1.x;
                        const unsigned win = o == 0 ? lo : __builtin_amdgcn_alignbyte(hi, lo, o);
                        acc[dx + R] = __builtin_amdgcn_sad_u8(a, win, acc[dx + R]);
                    }
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
    // FL loads in flight per thread (one round trip each otherwise); the view / candidate of element i = (vv G + g) N + c are
    // stepped, not divided out (i advances by 256: c by 256 % N): this code runs once, cold -- every instruction counts.
    constexpr int FL = FUSE_FL;
    const int GN = G * N;
    int ci = tid % N, i0 = tid;
    for (int base = 0; base < total; base += 256 * FL) {
        unsigned t[FL];
#pragma unroll
        for (int u = 0; u < FL; ++u) {
            const int i = base + u * 256 + tid;
            t[u] = i < total ? __hip_atomic_load(&part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        }
#pragma unroll
        for (int u = 0; u < FL; ++u) {
            if (i0 < total) atomicAdd(&s_sad[i0 >= GN ? 1 : 0][ci], t[u]);     // LDS atomics on N addresses per view
            i0 += 256;
            ci += 256 % N;
            ci -= ci >= N ? N : 0;
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

struct FuseWlut { unsigned w[256]; };
__device__ __forceinline__ unsigned cls_pack_bytes(unsigned a, unsigned b, unsigned c, unsigned d) { return a | (b << 8) | (c << 16) | (d << 24); }   // the blend weights ride in the kernel arguments: no host-to-device copy

// Rounded weighted mean of one byte position of the k views (the blend line of the header), word-wide caller below.
// The division: n = num + (den >> 1) < 2^20 and den <= 3 * 1024, so the true quotient's fraction is a multiple of 1 / den and
// (n + 0.5) / den sits at least 0.5 / 3072 = 1.6e-4 away from every integer, while v_rcp_f32 (1 ulp) and one rounded multiply
// are off by at most 255.5 * 1.8e-7 = 4.6e-5: the truncated float product IS floor(n / den), no fix-up (tests/test_host_logic.py
// walks every den with the reciprocal pushed 2 ulp either way).
template <int K>
__device__ __forceinline__ unsigned fuse_byte(const unsigned* __restrict__ s_w, unsigned w0, unsigned a, unsigned b, unsigned d) {
    unsigned n, den;                       // n = weighted sum + den / 2, built as one multiply-add chain on top of den >> 1
    if constexpr (K == 2) {
        const unsigned m = (a + b + 1u) >> 1;
        const unsigned wa = s_w[__builtin_amdgcn_sad_u8(a, m, 0u)], wb = s_w[__builtin_amdgcn_sad_u8(b, m, 0u)];
        den = wa + wb;
        n = __umul24(wb, b) + (__umul24(wa, a) + (den >> 1));
    } else {
        // the weighted sum does not care which view a value came from: sorted, the median's own weight is WLUT[0] (read once per
        // thread), so two table reads per byte, and the two distances are plain differences
        const unsigned lo = min(min(a, b), d), hi = max(max(a, b), d), m = max(min(a, b), min(max(a, b), d));      // v_min3 / v_max3 / v_med3
        const unsigned wl = s_w[m - lo], wh = s_w[hi - m];
        den = wl + wh + w0;
        n = __umul24(w0, m) + (__umul24(wh, hi) + (__umul24(wl, lo) + (den >> 1)));
    }
    const float nf = (float)n + 0.5f;
    return (unsigned)(nf * __builtin_amdgcn_rcpf((float)den));
}

// A thread blends FOUR pixels = 12 bytes = three dwords of the output row (W % 8 == 0: a row is a whole number of groups and
// its byte pitch a multiple of 4).  A view's 12 source bytes start at ((v H + y + dy) W + x + dx) * 3 -- any alignment, but the
// SAME one for every group of a view (W * 3 % 4 == 0), so they are four aligned dwords and three v_alignbyte by a per-view
// uniform amount.  The blend itself does not care which channel a byte is: twelve independent bytes.  Groups that touch the
// left / right edge under their shift (replicate clamp per PIXEL) gather their bytes one by one.
template <int K>
__global__ __launch_bounds__(256) void fusion_blend_kernel(const uint8_t* __restrict__ rgb, int H, int W,
                                                           const int* __restrict__ shifts,
                                                           const FuseWlut wlut, const unsigned* __restrict__ wluts, uint8_t* __restrict__ out) {
    __shared__ unsigned s_w[256];
    const size_t set = blockIdx.y;                       // view set of a batched call; its blend table comes from `wluts` (one call: kernel arguments)
    s_w[threadIdx.x] = wluts ? wluts[set * 256 + threadIdx.x] : wlut.w[threadIdx.x];
    // the caller's buffer [buf_lo, buf_hi): the word-wide path reads whole aligned dwords AROUND a group's 12 bytes, which for a device
    // pointer that is not 4-byte aligned would reach up to 3 bytes before the first view's first group or past the last view's last one
    const uint8_t* const buf_lo = rgb;
    const uint8_t* const buf_hi = rgb + (size_t)gridDim.y * K * H * W * 3;
    rgb += set * (size_t)K * H * W * 3; out += set * (size_t)H * W * 3; shifts += set * 6;
    __syncthreads();
    const int gpr = W >> 2;
    const int gi = blockIdx.x * 256 + threadIdx.x;
    if (gi >= H * gpr) return;
    const int y = gi / gpr, x0 = (gi - y * gpr) * 4;
    unsigned e[3][3];
#pragma unroll
    for (int v = 0; v < K; ++v) {
        const int sy = shifts[v * 2], sx = shifts[v * 2 + 1];
        const int yy = min(max(y + sy, 0), H - 1);
        const uint8_t* row = rgb + ((size_t)v * H + yy) * W * 3;
        const uint8_t* p = row + (x0 + sx) * 3;
        const unsigned rem = (unsigned)(reinterpret_cast<uintptr_t>(p) & 3u);
        if (x0 + sx >= 0 && x0 + sx + 3 <= W - 1 && p - rem >= buf_lo && p - rem + (rem ? 16 : 12) <= buf_hi) {
            const unsigned* q = reinterpret_cast<const unsigned*>(p - rem);
            const unsigned d0 = q[0], d1 = q[1], d2 = q[2], d3 = rem ? q[3] : 0u;     // rem == 0: the 12 bytes end with d2
            e[v][0] = __builtin_amdgcn_alignbyte(d1, d0, rem);
            e[v][1] = __builtin_amdgcn_alignbyte(d2, d1, rem);
            e[v][2] = __builtin_amdgcn_alignbyte(d3, d2, rem);
        } else {
            unsigned char t[12];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint8_t* p = row + min(max(x0 + i + sx, 0), W - 1) * 3;
                t[3 * i] = p[0]; t[3 * i + 1] = p[1]; t[3 * i + 2] = p[2];
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) e[v][j] = cls_pack_bytes(t[4 * j], t[4 * j + 1], t[4 * j + 2], t[4 * j + 3]);
        }
    }
    unsigned o[3];
    const unsigned w0 = s_w[0];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        unsigned r = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned a = (e[0][j] >> (8 * t)) & 0xffu, b = (e[1][j] >> (8 * t)) & 0xffu;
            const unsigned d = K == 3 ? (e[2][j] >> (8 * t)) & 0xffu : 0u;
            r |= fuse_byte<K>(s_w, w0, a, b, d) << (8 * t);
        }
        o[j] = r;
    }
    uint8_t* po = out + ((size_t)y * W + x0) * 3;
    if ((reinterpret_cast<uintptr_t>(po) & 3u) == 0) {
        unsigned* pw = reinterpret_cast<unsigned*>(po);
        pw[0] = o[0]; pw[1] = o[1]; pw[2] = o[2];
    } else {
#pragma unroll
        for (int j = 0; j < 12; ++j) po[j] = (uint8_t)(o[j >> 2] >> (8 * (j & 3)));
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
        // (copied on a stream of their own, beside the luma and SAD kernels, with the blend waiting on an event: measured, the chain is
        //  8 us LONGER -- the cross-stream wait costs more than the ~4 us copy in front: profiles/r03_experiments.md)
        IRE_HIP(hipMemcpyAsync(d_fwlut_, host_wluts, sizeof(unsigned) * 256 * nsets, hipMemcpyHostToDevice, s));   // pageable source: staged before the call returns
        d_wluts = d_fwlut_;
    }
    const int nq = k * hq * wq;
    hipLaunchKernelGGL(fusion_luma_kernel, dim3(ceil_div(nq, 256), nsets), dim3(256), 0, s, d_views, k, h, w, qp, d_fL_, d_fQ_);
    auto tiles = [](int ph, int pw, int m, int step, int tw, int sr) {
        return ceil_div(pw - 2 * m, tw) * ceil_div(ceil_div(ph - 2 * m, step), sr);
    };
    // Workgroups per view and set: a lone set spreads its tiles over the chip (one tile per workgroup at 1024^2); a batch keeps
    // ~2 workgroups per CU in total and walks several tiles each (the next tile's loads fly under this tile's arithmetic), which
    // also shortens the last workgroup's row sum.  Same box, 8 sets x 3 views at 1024^2, coarse / fine: 62 / 124 workgroups per
    // view 22.3 / 36.0 us, 32 / 32: 17.2 / 22.9 us, 16 / 16: 19.9 / 31.7 us (tools/r03_fus2.sh).
    static const int gcap_env = getenv("IRE_FUSE_GCAP") ? atoi(getenv("IRE_FUSE_GCAP")) : 0;
    const int gcap = gcap_env > 0 ? gcap_env : std::max(32, 512 / ((k - 1) * nsets));
    const int g0 = std::min(std::min(FUSE_SAD_GRID, gcap), tiles(hq, wq, CR, 1, SadCfg<0>::TW, SadCfg<0>::SR));
    const int g1 = std::min(std::min(FUSE_SAD_GRID, gcap), tiles(h, w, FM, 2, SadCfg<1>::TW, SadCfg<1>::SR));
    hipLaunchKernelGGL(fusion_sad_kernel<0>, dim3(g0, k - 1, nsets), dim3(256), 0, s, d_fQ_, k, hq, wq, qp, d_coarse, d_sh, (int*)nullptr,
                       d_fsad_, d_ticket);
    hipLaunchKernelGGL(fusion_sad_kernel<1>, dim3(g1, k - 1, nsets), dim3(256), 0, s, d_fL_, k, h, w, w, d_coarse, d_sh, (int*)d_shifts,
                       d_fsad_, d_ticket);
    if (k == 3) hipLaunchKernelGGL(fusion_blend_kernel<3>, dim3(ceil_div((int)(px / 4), 256), nsets), dim3(256), 0, s, d_views, h, w, d_sh, lut, d_wluts, d_out);
    else hipLaunchKernelGGL(fusion_blend_kernel<2>, dim3(ceil_div((int)(px / 4), 256), nsets), dim3(256), 0, s, d_views, h, w, d_sh, lut, d_wluts, d_out);
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
