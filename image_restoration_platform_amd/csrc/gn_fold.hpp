// gn_fold.hpp -- GroupNorm finalize folded into the prologue of the convolution that consumes the coefficients.
//
// gn.hip's gn_finalize_kernel is 64 tiny workgroups per GroupNorm: 33 launches per network step, each ~5 us of kernel plus a
// ~5.7 us dependent-dispatch gap on either side (profiles/r02: 0.35 ms of a 9.7 ms step).  Here every persistent workgroup
// of the consumer reduces the tile partials of the image(s) ITS items read -- 8 groups x gn_parts (sum, sumsq) pairs, in
// double, in a fixed order -- and writes y = x * A + B coefficients to the same [image][channel] array the staging code
// fetches from.  All workgroups that touch an image compute bit-identical values from the same partials (same code, same
// order), so the redundant stores are benign and the result does not depend on which workgroup ran when: the run stays
// deterministic, batch-independent and strip-independent (cfg 4 finalizes over the complete gathered array as before).
// Cost: gn_parts x 64 B of L2 reads per workgroup and image (128 KB at 1024^2 level 0) ~ 2-3 us, against ~10.5 us saved.
#pragma once
#include "conv_mfma.hpp"

namespace ire {

// smem: >= nthr * 16 + 64 bytes of LDS not otherwise in use yet; ends with a barrier, the coefficient stores retired.
// nthr_used: the threads that take part (a power of two, a multiple of 64, <= blockDim.x; 0 = all of them): the rest only join the barriers.
// lds_ab (optional): the caller's LDS coefficient table [img - img_lo][C] float2 -- filled here as well, so that a kernel that keeps
// its coefficients in LDS (conv_pc, conv_pk) does not read back from global memory what this workgroup has just computed.
//
// Reduction order (round 4): a thread = (tile lane tl = tid >> 3, group g = tid & 7) adds its tiles in eight independent chains, the
// eight tile lanes of a WAVE are added by three cross-lane steps (lanes 8, 16, 32 apart: no LDS, no barrier), the waves' sums meet in
// LDS and one thread per group adds them in wave order.  Two barriers per image where the LDS tree took eight.  Every workgroup of
// every kernel runs this same function with the same thread count, so all of them still derive bit-identical coefficients.
__device__ __forceinline__ double gnf_xlane_add(double v, int ctrl_kind) {
    // ctrl_kind 0: lane ^ 8 (DPP row_ror:8 within a row of 16), 1: lanes 16 apart (permlane16 swap), 2: lanes 32 apart (permlane32 swap)
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32), olo, ohi;
    if (ctrl_kind == 0) {
        olo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x128, 0xf, 0xf, false);
        ohi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0x128, 0xf, 0xf, false);
    } else if (ctrl_kind == 1) {
        unsigned x = lo, y = lo, x2 = hi, y2 = hi;
        asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
        asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x2), "+v"(y2));
        // after the swap x holds (own rows 0, 2 | partner rows ...): x and y are the two operands of the pairwise add, as in *_swap16_add
        const double a = __builtin_bit_cast(double, ((unsigned long long)x2 << 32) | x), c = __builtin_bit_cast(double, ((unsigned long long)y2 << 32) | y);
        return a + c;
    } else {
        unsigned x = lo, y = lo, x2 = hi, y2 = hi;
        asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
        asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x2), "+v"(y2));
        const double a = __builtin_bit_cast(double, ((unsigned long long)x2 << 32) | x), c = __builtin_bit_cast(double, ((unsigned long long)y2 << 32) | y);
        return a + c;
    }
    return v + __builtin_bit_cast(double, ((unsigned long long)ohi << 32) | olo);
}

__device__ __forceinline__ void gn_fold(const ConvArgs& a, unsigned char* smem, int img_lo, int img_hi, int nthr_used = 0, float2* lds_ab = nullptr) {
#if defined(IRE_FOLD_ABL) && IRE_FOLD_ABL == 1      // timing ablation (results wrong by design): the consumers read stale coefficients
    __syncthreads();
    return;
#endif
    const int tid = threadIdx.x, nthr = nthr_used ? nthr_used : (int)blockDim.x;
    const bool act = tid < nthr;
    double* red = reinterpret_cast<double*>(smem);                       // [waves][8 groups][2]
    float* mr = reinterpret_cast<float*>(smem + (size_t)nthr * 16);      // [8][2] mean, rstd
    const int g = tid & 7, tl = tid >> 3, ntl = nthr >> 3;               // thread = (tile lane, group): a tile's 8 groups are 64 contiguous bytes
    const int nwaves = nthr >> 6;
    const int C = a.cin0, G = C >> 3;
    // this thread's channel of the FIRST image (C <= nthr everywhere: one channel per thread): gamma, beta and the FiLM pair are requested
    // before the partials, not behind the reduction (a second, serial round trip per fold: ~1 us of its ~3)
    const bool pre = act && tid < C && C <= nthr;
    float p_gamma = 0.f, p_beta = 0.f, p_sc = 0.f, p_sh = 0.f;
    if (pre) {
        p_gamma = a.gn_gamma[tid]; p_beta = a.gn_beta[tid];
        if (a.gn_film) {
            p_sc = a.gn_film[(size_t)img_lo * a.gn_film_stride + a.gn_film_off + tid];
            p_sh = a.gn_film[(size_t)img_lo * a.gn_film_stride + a.gn_film_off + C + tid];
        }
    }
    for (int img = img_lo; img <= img_hi; ++img) {
        const float2* st = reinterpret_cast<const float2*>(a.gn_stats) + (size_t)img * a.gn_parts * 8 + g;
        // eight independent chains per thread: the L2 round trips of a pass overlap (a level-0 image at 1024^2 is 4 passes)
        double sv[8], qv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { sv[k] = 0.0; qv[k] = 0.0; }
        int t = act ? tl : a.gn_parts;
        for (; t + 7 * ntl < a.gn_parts; t += 8 * ntl) {
            float2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = st[(size_t)(t + k * ntl) * 8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { sv[k] += (double)v[k].x; qv[k] += (double)v[k].y; }
        }
        for (; t < a.gn_parts; t += ntl) { const float2 v = st[(size_t)t * 8]; sv[0] += (double)v.x; qv[0] += (double)v.y; }
        double s1 = ((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7]));
        double q1 = ((qv[0] + qv[1]) + (qv[2] + qv[3])) + ((qv[4] + qv[5]) + (qv[6] + qv[7]));
        // the eight tile lanes of this wave that share group g sit 8 lanes apart
#pragma unroll
        for (int step = 0; step < 3; ++step) { s1 = gnf_xlane_add(s1, step); q1 = gnf_xlane_add(q1, step); }
        if (act && (tid & 63) < 8) { red[((tid >> 6) * 8 + g) * 2] = s1; red[((tid >> 6) * 8 + g) * 2 + 1] = q1; }
        __syncthreads();
        if (tid < 8) {
            double ss = 0.0, qq = 0.0;
            for (int w = 0; w < nwaves; ++w) { ss += red[(w * 8 + tid) * 2]; qq += red[(w * 8 + tid) * 2 + 1]; }
            const double cnt = (double)a.gn_hw * (double)G;
            const double mean = ss / cnt;
            double var = qq / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            mr[tid * 2] = (float)mean;
            mr[tid * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
        }
        __syncthreads();
        for (int c = act ? tid : C; c < C; c += nthr) {
            const int gg = c / G;
            const bool use_pre = pre && img == img_lo;                     // (then c == tid)
            const float gam = use_pre ? p_gamma : a.gn_gamma[c], bet = use_pre ? p_beta : a.gn_beta[c];
            const float rg = mr[gg * 2 + 1] * gam;
            float sc = 0.f, sh = 0.f;
            if (a.gn_film) {
                sc = use_pre ? p_sc : a.gn_film[(size_t)img * a.gn_film_stride + a.gn_film_off + c];
                sh = use_pre ? p_sh : a.gn_film[(size_t)img * a.gn_film_stride + a.gn_film_off + C + c];
            }
            float2 o;
            o.x = rg * (1.f + sc);
            o.y = (bet - mr[gg * 2] * rg) * (1.f + sc) + sh;
            a.ab_w[(size_t)img * C + c] = o;
            if (lds_ab) lds_ab[(size_t)(img - img_lo) * C + c] = o;
        }
        __syncthreads();
    }
    // the coefficient stores have reached L2 before any wave of this workgroup fetches them (kernels that read them back from
    // global memory; with lds_ab the table is complete behind the loop's last barrier and nothing here needs to wait)
    if (!lds_ab) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
}

}  // namespace ire
