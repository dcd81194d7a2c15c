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
// of 16x256 output tiles of ONE image (blockIdx.y); each tile is staged with a 1-pixel
// replicate-clamped halo into LDS as four planes (R,G,B,grey) of PACKED u8 words, four pixels
// per word; the horizontal blur pass goes through LDS as well (libvips rounds to u8 between
// the two passes).  The image's last workgroup computes the scores (classifier_finalize.hpp).
#include "classifier.hpp"

#include "classifier_finalize.hpp"

namespace ire {

namespace {

constexpr int CT_H = 16, CT_W = 256;            // output tile: 16 rows x 64 four-pixel groups
constexpr int CG = CT_W / 4;                    // groups per tile row
constexpr int PL_ROWS = CT_H + 2;               // halo rows -1 .. 16
constexpr int PL_WORDS = CG + 2;                // plane row = pixels x0-4 .. x0+259 as packed u8 words: pixel x0+dx is byte 4+dx
constexpr int NBUCKETS = 5001;
#ifndef CLS_ABL
#define CLS_ABL 0     // timing ablations (results wrong by design): 1 no grey tables, 2 no phase B, 4 no phase C, 8 no atomics / finalize
#endif
constexpr unsigned DIV44_M = 11916u;            // floor(n / 44) == (n * 11916) >> 19 for every n < 32768 (11916 * 44 = 2^19 + 16)

struct ClsLds {
    unsigned int wR[256], wG[256], wB[256];
    unsigned int thr[260];
    unsigned char inv[5008];
    unsigned int pl[4][PL_ROWS][PL_WORDS];      // R, G, B, grey: four pixels per word
    unsigned int hb[3][PL_ROWS][CG];            // horizontally blurred R, G, B (u8-rounded, libvips integer convsep) for rows -1 .. 16
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

__device__ __forceinline__ unsigned cls_byte(unsigned w, int j) { return (w >> (8 * j)) & 0xffu; }
__device__ __forceinline__ unsigned cls_pack4(unsigned a, unsigned b, unsigned c, unsigned d) { return a | (b << 8) | (c << 16) | (d << 24); }
// sum and sum of squares of the four bytes of a word: one v_sad_u8 and one v_dot4_u32_u8
__device__ __forceinline__ void cls_acc4(unsigned w, unsigned& s, unsigned& q) {
    s = __builtin_amdgcn_sad_u8(w, 0u, s);
    q = __builtin_amdgcn_udot4(w, w, q, false);
}
// {12, 20, 12} / 44 with libvips' integer rounding: (12 a + 20 b + 12 c + 22) / 44
__device__ __forceinline__ unsigned cls_blur3(unsigned a, unsigned b, unsigned c) {
    return ((12u * (a + c) + 20u * b + 22u) * DIV44_M) >> 19;
}
// The same blur with the three taps in bytes 0..2 of one word: ONE v_dot4_u32_u8 gives the numerator (byte 3 has weight 0),
// one 24-bit multiply puts floor(n / 44) into the product's TOP BYTE: n * 381301 < 2^32 and floor(n / 44) == (n * 381301) >> 24
// for every n <= 12 * 510 + 20 * 255 + 22 = 11242 (381301 * 44 = 2^24 + 28; tests/test_host_logic.py checks every n), so four
// quotients pack with two byte permutes and an OR -- no shifts, no per-byte extracts.
constexpr unsigned BLUR_W = 0x000c140cu;         // weights 12, 20, 12, 0
constexpr unsigned DIV44_TOP = 381301u;
__device__ __forceinline__ unsigned cls_blur_top(unsigned window) {
    return __umul24(__builtin_amdgcn_udot4(window, BLUR_W, 22u, false), DIV44_TOP);
}
__device__ __forceinline__ unsigned cls_pack_top(unsigned p0, unsigned p1, unsigned p2, unsigned p3) {     // bytes 3 of p0..p3
    return __builtin_amdgcn_perm(p1, p0, 0x0c0c0703u) | __builtin_amdgcn_perm(p3, p2, 0x07030c0cu);
}
// windows (x-1, x, x+1, .) of the four pixels of w1; w0 / w2 are the words to its left / right
__device__ __forceinline__ void cls_windows(unsigned w0, unsigned w1, unsigned w2, unsigned (&win)[4]) {
    win[0] = __builtin_amdgcn_alignbyte(w1, w0, 3);
    win[1] = w1;
    win[2] = __builtin_amdgcn_alignbyte(w2, w1, 1);
    win[3] = __builtin_amdgcn_alignbyte(w2, w1, 2);
}

// One workgroup walks a strided set of 16 x 256 tiles of ONE image (blockIdx.y).  Everything between HBM and the 12 + 2
// integer accumulators is word-wide: a thread owns a four-pixel group (three dwords of RGB bytes in, four packed u8 words
// out: R, G, B, grey), the planes live in LDS as packed words, neighbours come from the adjacent words by byte alignment,
// and every per-pixel quantity is packed back to bytes so that its sum and its sum of squares are one v_sad_u8 and one
// v_dot4_u32_u8 per four pixels (a byte mask drops the pixels outside a ragged image).  The LAST workgroup of an image
// (ticket counter) finalizes its seven scores: no second launch.
__global__ __launch_bounds__(256) void classifier_scan_kernel(const uint8_t* __restrict__ rgb, int H, int W,
                                                              int tiles_x, int tiles_y,
                                                              const unsigned int* __restrict__ g_lin16,
                                                              const unsigned int* __restrict__ g_thr,
                                                              const unsigned char* __restrict__ g_inv,
                                                              unsigned long long* __restrict__ sums,
                                                              unsigned long long* __restrict__ tickets,
                                                              unsigned long long* __restrict__ parts,
                                                              const uint8_t* __restrict__ is_jpeg,
                                                              double* __restrict__ scores, int32_t* __restrict__ label,
                                                              float* __restrict__ cond,
                                                              const float* __restrict__ film_w, const float* __restrict__ film_b, int film_n, float* __restrict__ film) {
    __shared__ ClsLds L;
    __shared__ unsigned long long red[4][CLS_NSUMS];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const int img = blockIdx.y;
    const uint8_t* base = rgb + (size_t)img * H * W * 3;


    unsigned long long acc[CLS_NSUMS];
#pragma unroll
    for (int i = 0; i < CLS_NSUMS; ++i) acc[i] = 0;

    auto grey_of = [&](unsigned r, unsigned g, unsigned b) -> unsigned {
        if constexpr (CLS_ABL & 1) return g;
        const unsigned y = L.wR[r] + L.wG[g] + L.wB[b];
        unsigned gv = L.inv[y >> 17];
        gv += (y >= L.thr[gv + 1]) ? 1u : 0u;
        return gv;
    };

    const int g = tid & (CG - 1), rb = tid >> 6;          // phases B, C: four-pixel group g, rows 4 rb .. 4 rb + 3
    const int ntiles = tiles_x * tiles_y;
    // a tile's input: PL_ROWS x PL_WORDS groups of four pixels = 12 bytes = three (unaligned) dwords each, A_ITERS per thread.
    // Groups that straddle the image's left / right edge (replicate padding) are gathered bytewise into the same form.
    constexpr int A_ITERS = (PL_ROWS * PL_WORDS + 255) / 256;
    unsigned pre[A_ITERS][3];
    // a thread's groups sit at the same (row, word) of every tile: decoded once (the division by the plane width and the clamps
    // of a from-scratch decode were ~40 vector instructions per group and tile: 200 of a tile's ~1 800)
    int a_py[A_ITERS], a_xo[A_ITERS];
#pragma unroll
    for (int k = 0; k < A_ITERS; ++k) {
        const int i = min(tid + k * 256, PL_ROWS * PL_WORDS - 1);
        a_py[k] = i / PL_WORDS - 1;                              // tile row -1 .. 16
        a_xo[k] = 4 * (i - (i / PL_WORDS) * PL_WORDS - 1);       // first pixel of the group relative to the tile: -4 .. 256
    }
    auto load_tile = [&](int tile) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int y0 = ty * CT_H, x0 = tx * CT_W;
#pragma unroll
        for (int k = 0; k < A_ITERS; ++k) {
            const int gy = min(max(y0 + a_py[k], 0), H - 1);
            const int xs = x0 + a_xo[k];
            if (xs >= 0 && xs + 3 < W) {
                const uint8_t* p = base + (unsigned)((gy * W + xs) * 3);          // (an image is < 2^31 bytes: check_shape caps H, W at 8192)
                __builtin_memcpy(&pre[k][0], p, 4); __builtin_memcpy(&pre[k][1], p + 4, 4); __builtin_memcpy(&pre[k][2], p + 8, 4);
            } else {
                unsigned char t[12];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gx = min(max(xs + j, 0), W - 1);
                    const uint8_t* p = base + (unsigned)((gy * W + gx) * 3);
                    t[3 * j] = p[0]; t[3 * j + 1] = p[1]; t[3 * j + 2] = p[2];
                }
                pre[k][0] = cls_pack4(t[0], t[1], t[2], t[3]); pre[k][1] = cls_pack4(t[4], t[5], t[6], t[7]); pre[k][2] = cls_pack4(t[8], t[9], t[10], t[11]);
            }
        }
    };
    if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
    // (the first tile's pixels are requested BEFORE the tables: the two round trips overlap)
    // tables -> LDS once per workgroup (pre-multiplied by the luminance weights x10000)
    {
        unsigned int l = g_lin16[tid];
        L.wR[tid] = 2126u * l;
        L.wG[tid] = 7152u * l;
        L.wB[tid] = 722u * l;
        L.thr[tid] = g_thr[tid];
        if (tid == 0) L.thr[256] = g_thr[256];
        // 5008 bytes as 313 16-byte chunks (hipMalloc'd: aligned; the table is padded to 5008 on the host)
        const uint4* gi = reinterpret_cast<const uint4*>(g_inv);
        uint4* li = reinterpret_cast<uint4*>(L.inv);
        for (int i = tid; i < 5008 / 16; i += 256) li[i] = gi[i];
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int y0 = ty * CT_H, x0 = tx * CT_W;

        // phase A: halo tile (replicate-clamped) -> R, G, B, grey planes from the 12-byte groups prefetched into `pre` (issued a
        // whole tile ahead: during the previous tile's phases B and C)
#pragma unroll
        for (int k = 0; k < A_ITERS; ++k) {
            const int i = tid + k * 256;
            if (i < PL_ROWS * PL_WORDS) {
                const int py = i / PL_WORDS, wq = i - py * PL_WORDS;
                const unsigned d0 = pre[k][0], d1 = pre[k][1], d2 = pre[k][2];          // r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
                // the three planes by byte permutes (two per plane) instead of twelve extracts and nine packs
                const unsigned rw = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x0c060300u), 0x05020100u);
                const unsigned gw = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x0c070401u), 0x06020100u);
                const unsigned bw = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x0c0c0502u), 0x07040100u);
                unsigned yv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) yv[j] = grey_of(cls_byte(rw, j), cls_byte(gw, j), cls_byte(bw, j));
                L.pl[0][py][wq] = rw;
                L.pl[1][py][wq] = gw;
                L.pl[2][py][wq] = bw;
                L.pl[3][py][wq] = cls_pack4(yv[0], yv[1], yv[2], yv[3]);
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);      // in flight across phases B and C

        // phase B: horizontal {12,20,12}/44 pass, rounded to u8; rows 4 rb .. 4 rb + 3, and plane rows 16 + rb for rb < 2
        if constexpr (!(CLS_ABL & 2))
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int py = k < 4 ? rb * 4 + k : CT_H + rb;
            if (k == 4 && rb >= 2) break;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                unsigned win[4];
                cls_windows(L.pl[c][py][g], L.pl[c][py][g + 1], L.pl[c][py][g + 2], win);
                L.hb[c][py][g] = cls_pack_top(cls_blur_top(win[0]), cls_blur_top(win[1]), cls_blur_top(win[2]), cls_blur_top(win[3]));
            }
        }
        __syncthreads();

        // phase C: this thread's 4 x 4 pixels
        if constexpr (!(CLS_ABL & 4)) {
            const int gx = x0 + 4 * g;
            unsigned xm = 0;                                   // 0xff per pixel of the group that is inside the image
#pragma unroll
            for (int j = 0; j < 4; ++j) xm |= (gx + j < W) ? (0xffu << (8 * j)) : 0u;
            unsigned s_c[3] = {0, 0, 0}, q_c[3] = {0, 0, 0};
            unsigned s_b = 0, q_b = 0, s8 = 0, q8 = 0, s9 = 0, q9 = 0;
            // sliding window over plane rows: h3[row][j] = grey(x-1) + grey(x) + grey(x+1) of that row
            unsigned h3[3][4], cw[3];
            auto row_sums = [&](int py, unsigned (&h)[4], unsigned& cword) {
                const unsigned w1 = L.pl[3][py][g + 1];
                unsigned win[4];
                cls_windows(L.pl[3][py][g], w1, L.pl[3][py][g + 2], win);
#pragma unroll
                for (int j = 0; j < 4; ++j) h[j] = __builtin_amdgcn_udot4(win[j], 0x00010101u, 0u, false);     // grey(x-1) + grey(x) + grey(x+1)
                cword = w1;
            };
            row_sums(rb * 4, h3[0], cw[0]);
            row_sums(rb * 4 + 1, h3[1], cw[1]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int y = rb * 4 + k, hy = y + 1;          // hy: plane row of the pixel row
                row_sums(hy + 1, h3[(k + 2) % 3], cw[(k + 2) % 3]);
                const unsigned m = (y0 + y < H) ? xm : 0u;
                unsigned e8[4], e9[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = (int)cls_byte(cw[(k + 1) % 3], j);
                    const int sum9 = (int)(h3[0][j] + h3[1][j] + h3[2][j]);
                    e8[j] = (unsigned)min(max(9 * c - sum9, 0), 255);
                    e9[j] = (unsigned)min(max(10 * c - sum9, 0), 255);
                }
                cls_acc4(cls_pack4(e8[0], e8[1], e8[2], e8[3]) & m, s8, q8);
                cls_acc4(cls_pack4(e9[0], e9[1], e9[2], e9[3]) & m, s9, q9);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    cls_acc4(L.pl[ch][hy][g + 1] & m, s_c[ch], q_c[ch]);
                    // vertical pass: the three rows' bytes of a pixel gathered into one window word by two byte permutes
                    const unsigned ua = L.hb[ch][hy - 1][g], ub = L.hb[ch][hy][g], uc = L.hb[ch][hy + 1][g];
                    const unsigned bw = cls_pack_top(cls_blur_top(__builtin_amdgcn_perm(uc, __builtin_amdgcn_perm(ub, ua, 0x0c0c0400u), 0x0c040100u)),
                                                     cls_blur_top(__builtin_amdgcn_perm(uc, __builtin_amdgcn_perm(ub, ua, 0x0c0c0501u), 0x0c050100u)),
                                                     cls_blur_top(__builtin_amdgcn_perm(uc, __builtin_amdgcn_perm(ub, ua, 0x0c0c0602u), 0x0c060100u)),
                                                     cls_blur_top(__builtin_amdgcn_perm(uc, __builtin_amdgcn_perm(ub, ua, 0x0c0c0703u), 0x0c070100u)));
                    cls_acc4(bw & m, s_b, q_b);
                }
            }
            acc[0] += s_c[0]; acc[1] += s_c[1]; acc[2] += s_c[2];
            acc[3] += q_c[0]; acc[4] += q_c[1]; acc[5] += q_c[2];
            acc[6] += s_b; acc[7] += q_b;
            acc[8] += s8; acc[9] += q8;
            acc[10] += s9; acc[11] += q9;
        }
        // scratch probes: the pixels on the stride-4 grid (tile origin is a multiple of 4): one per thread
        {
            const unsigned char* gp = reinterpret_cast<const unsigned char*>(&L.pl[3][0][0]);
            const int y = rb * 4, x = g * 4;
            const int gy = y0 + y, gx = x0 + x;
            if (gy < H && gx < W) {
                auto px = [&](int yy, int xx) -> int { return gp[(yy + 1) * (PL_WORDS * 4) + 4 + xx]; };
                auto e4 = [&](int yy, int xx) -> int {
                    const int v = 4 * px(yy, xx) - px(yy - 1, xx) - px(yy + 1, xx) - px(yy, xx - 1) - px(yy, xx + 1);
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

    if constexpr (CLS_ABL & 8) { if (acc[0] + acc[5] + acc[9] + acc[13] == 0x123456789ull) sums[0] = 1; return; }
    // workgroup reduction -> 14 u64 atomics per workgroup
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int i = 0; i < CLS_NSUMS; ++i) {
        unsigned long long v = wave_sum_u64(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    // Workgroup partials go to this workgroup's own row of `parts` with plain stores; the image's LAST workgroup (one ticket
    // atomic per workgroup) adds the rows -- exact integers: any order gives the same sums -- and computes the scores.  (14
    // u64 atomics per workgroup on ONE 128-byte line per image serialise at the memory-side atomic unit: 64 workgroups x 14 =
    // 896 per line took 31 of the kernel's 76 us.)
    // (Device-scope atomic stores / loads, relaxed: they go to the device's coherence point without the L2 write-back and
    // invalidate a __threadfence() costs every workgroup -- 64 workgroups per XCD queue up on those: +35 us.  The ticket is
    // ordered after the row by waiting for the stores' acknowledgement.)
    if (tid < CLS_NSUMS)
        __hip_atomic_store(&parts[((size_t)img * gridDim.x + blockIdx.x) * CLS_NSUMS + tid], red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(&tickets[img], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)(gridDim.x - 1);
    __syncthreads();
    if (!s_last) return;
    {   // 16 row lanes x 14 columns: every thread's loads are independent (one round trip, not one per row)
        __shared__ unsigned long long fin[CLS_NSUMS][16];
        const int k = tid >> 4, rl = tid & 15;
        if (k < CLS_NSUMS) {
            // a launch has at most CLS_MAX_WG = 768 workgroups per image: <= 48 rows per lane, requested eight at a time (one
            // round trip per eight rows instead of one per row: the accumulate made the loads of the rolled loop serial)
            unsigned long long v = 0;
            for (unsigned b0 = rl; b0 < gridDim.x; b0 += 128) {
                unsigned long long t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned b = b0 + 16 * u;
                    t[u] = b < gridDim.x ? __hip_atomic_load(&parts[((size_t)img * gridDim.x + b) * CLS_NSUMS + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) v += t[u];
            }
            fin[k][rl] = v;
        }
        __syncthreads();
        if (tid < CLS_NSUMS) {
            unsigned long long v = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) v += fin[tid][i];
            red[0][tid] = v;
            sums[(size_t)img * CLS_NSUMS + tid] = v;
        }
    }
    __syncthreads();
    __shared__ float s_cond[8];
    // The scores (classifier_finalize.hpp::cls_finalize_one, the same IEEE steps expression by expression) spread over the last
    // workgroup's four waves: one thread walking all of it is ~7 double square roots and ~20 double divisions in one dependent chain
    // (9.5 of the scan's 46 us at batch 8: the launch's tail).  Lanes of one wave that run DIFFERENT formulas serialise, so the
    // independent pieces go to different waves, and pieces with the same formula to different lanes of one wave:
    //   step 1: wave 0 lanes 0..2 mean / sample stdev of R, G, B; wave 1 lanes 0..3 the four exact population variances
    //   step 2: wave 0 noise, wave 1 fade, wave 2 colorShift, wave 3 blur / lowLight / compression / scratch
    //   step 3: thread 0 the label (first-max argmax), the ticket's reset
    {
#pragma clang fp contract(off)
        __shared__ double f_mean[3], f_sd[3], f_var[4], f_sc[8];
        const int wave_f = tid >> 6, lane_f = tid & 63;
        const uint64_t px = (uint64_t)H * (uint64_t)W;
        const int jpeg = is_jpeg ? is_jpeg[img] : 1;
        if (wave_f == 0 && lane_f < 3) {
            const double vals = (double)px;
            const double ds = (double)red[0][lane_f], ds2 = (double)red[0][3 + lane_f];
            f_mean[lane_f] = ds / vals;
            f_sd[lane_f] = sqrt(fabs(ds2 - (ds * ds / vals)) / (vals - 1.0));
        } else if (wave_f == 1 && lane_f < 4) {
            // var_e8, var_e9, var_rgb, var_blur
            const uint64_t n = lane_f < 2 ? px : 3 * px;
            const uint64_t sa = lane_f == 0 ? red[0][8] : lane_f == 1 ? red[0][10] : lane_f == 2 ? red[0][0] + red[0][1] + red[0][2] : red[0][6];
            const uint64_t sb = lane_f == 0 ? red[0][9] : lane_f == 1 ? red[0][11] : lane_f == 2 ? red[0][3] + red[0][4] + red[0][5] : red[0][7];
            f_var[lane_f] = popvar_exact(n, sa, sb);
        }
        __syncthreads();
        if (lane_f == 0) {
            if (wave_f == 0) {
                f_sc[1] = js_min(sqrt(f_var[1]) / 50.0, 1.0);
            } else if (wave_f == 1) {
                const double sd0 = f_sd[0], sd1 = f_sd[1], sd2 = f_sd[2];
                const double sat = sqrt((sd0 * sd0 + sd1 * sd1) + sd2 * sd2) / 255.0;
                const double colorfulness = js_min(sat, 1.0);
                const double avg_sd = (((0.0 + sd0) + sd1) + sd2) / 3.0;
                const double contrast = js_min(avg_sd / 64.0, 1.0);
                f_sc[5] = js_min((1.0 - colorfulness) * 0.6 + (1.0 - contrast) * 0.4, 1.0);
            } else if (wave_f == 2) {
                const double m0 = f_mean[0], m1 = f_mean[1], m2 = f_mean[2];
                const double avg = ((m0 + m1) + m2) / 3.0;
                const double dr = avg > 0 ? fabs(m0 - avg) / avg : 0.0;
                const double dg = avg > 0 ? fabs(m1 - avg) / avg : 0.0;
                const double db = avg > 0 ? fabs(m2 - avg) / avg : 0.0;
                f_sc[6] = js_min(js_max(js_max(dr, dg), db) * 2.0, 1.0);
            } else {
                const double nv = js_min(f_var[0] / 1000.0, 1.0);
                f_sc[0] = js_max(0.0, 1.0 - nv);
                const double mb = (((0.0 + f_mean[0]) + f_mean[1]) + f_mean[2]) / 3.0;
                const double nb = mb / 255.0;
                f_sc[2] = (nb < 0.3) ? js_min((0.3 - nb) * 2.0, 1.0) : 0.0;
                if (!jpeg) {
                    f_sc[3] = 0.0;
                } else {
                    const double delta = js_max(0.0, f_var[2] - f_var[3]);
                    f_sc[3] = js_min(js_min(delta / 500.0, 1.0), 1.0);
                }
                const double total = (double)(red[0][12] + red[0][13]);
                f_sc[4] = js_min(js_min(total / 1000.0, 1.0), 1.0);
            }
        }
        __syncthreads();
        if (tid < 7) {
            const double v = f_sc[tid];
            if (scores) scores[(size_t)img * 7 + tid] = v;
            if (cond) cond[(size_t)img * 8 + tid] = (float)v;
            s_cond[tid] = (float)v;
        }
        if (tid == 0) {
            __hip_atomic_store(&tickets[img], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-cleaning: the next launch starts from zero without a memset
            if (cond) cond[(size_t)img * 8 + 7] = 0.f;
            int best = 0;                    // first-max argmax in key order
            for (int i = 1; i < 7; ++i)
                if (f_sc[i] > f_sc[best]) best = i;
            if (label) label[img] = best;
        }
    }
    if (film == nullptr) return;
    // the restoration's FiLM vector of this image (Linear(7 -> film_n)(scores): gn.hip::film_kernel's arithmetic, term by term) from the
    // same workgroup: a restore call that classifies inside needs no film launch and no kernel boundary behind the scan
    __syncthreads();
    for (int o = tid; o < film_n; o += 256) {
        float acc = film_b[o];
#pragma unroll
        for (int k = 0; k < 7; ++k) acc = __builtin_fmaf(film_w[o * 7 + k], s_cond[k], acc);
        film[(size_t)img * film_n + o] = acc;
    }
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
                       int32_t* d_label, float* d_cond, hipStream_t stream, const float* d_film_w, const float* d_film_b, int film_n, float* d_film) {
    // d_sums (engine.cpp::ensure_io): [cap][14] sums | [cap] tickets (zero at allocation, reset by the kernel) | [cap][CLS_MAX_WG][14]
    // workgroup partials -- no memset, no second launch
    const int tiles_x = ceil_div(w, CT_W), tiles_y = ceil_div(h, CT_H);
    const int ntiles = tiles_x * tiles_y;
    // three workgroups per CU chip-wide, an equal number of tiles each where the counts allow; every workgroup amortises its
    // 9 KB table load over its tiles
    int per_img = std::max(1, std::min(ntiles, CLS_MAX_WG / std::max(1, n)));
    per_img = ceil_div(ntiles, ceil_div(ntiles, per_img));
    dim3 grid(per_img, n);
    hipLaunchKernelGGL(classifier_scan_kernel, grid, dim3(256), 0, stream, d_rgb, h, w, tiles_x, tiles_y,
                       tb.lin16, tb.thr, tb.inv, d_sums, cls_tickets(d_sums), cls_parts(d_sums), d_is_jpeg, d_scores, d_label, d_cond,
                       d_film_w, d_film_b, film_n, d_film);
    IRE_HIP(hipGetLastError());
}

void scores_to_cond_launch(const double* d_scores, int n, float* d_cond, hipStream_t stream) {
    hipLaunchKernelGGL(scores_to_cond_kernel, dim3(ceil_div(n * 8, 64)), dim3(64), 0, stream, d_scores, n, d_cond);
    IRE_HIP(hipGetLastError());
}

}  // namespace ire
