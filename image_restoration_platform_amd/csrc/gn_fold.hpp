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
// nthr_used: the threads that take part (a power of two <= blockDim.x; 0 = all of them): the rest only join the barriers.
__device__ __forceinline__ void gn_fold(const ConvArgs& a, unsigned char* smem, int img_lo, int img_hi, int nthr_used = 0) {
#if defined(IRE_FOLD_ABL) && IRE_FOLD_ABL == 1      // timing ablation (results wrong by design): the consumers read stale coefficients
    __syncthreads();
    return;
#endif
    const int tid = threadIdx.x, nthr = nthr_used ? nthr_used : (int)blockDim.x;
    const bool act = tid < nthr;
    double* red = reinterpret_cast<double*>(smem);                       // [nthr][2]
    float* mr = reinterpret_cast<float*>(smem + (size_t)nthr * 16);      // [8][2] mean, rstd
    const int g = tid & 7, tl = tid >> 3, ntl = nthr >> 3;               // thread = (tile lane, group): a tile's 8 groups are 64 contiguous bytes
    const int C = a.cin0, G = C >> 3;
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
        if (act) {
            red[tid * 2] = ((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7]));
            red[tid * 2 + 1] = ((qv[0] + qv[1]) + (qv[2] + qv[3])) + ((qv[4] + qv[5]) + (qv[6] + qv[7]));
        }
        __syncthreads();
        for (int off = ntl >> 1; off >= 1; off >>= 1) {
            if (act && tl < off) { red[tid * 2] += red[(tid + off * 8) * 2]; red[tid * 2 + 1] += red[(tid + off * 8) * 2 + 1]; }
            __syncthreads();
        }
        if (tid < 8) {
            const double cnt = (double)a.gn_hw * (double)G;
            const double mean = red[tid * 2] / cnt;
            double var = red[tid * 2 + 1] / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            mr[tid * 2] = (float)mean;
            mr[tid * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
        }
        __syncthreads();
        for (int c = act ? tid : C; c < C; c += nthr) {
            const int gg = c / G;
            const float rg = mr[gg * 2 + 1] * a.gn_gamma[c];
            float sc = 0.f, sh = 0.f;
            if (a.gn_film) {
                sc = a.gn_film[(size_t)img * a.gn_film_stride + a.gn_film_off + c];
                sh = a.gn_film[(size_t)img * a.gn_film_stride + a.gn_film_off + C + c];
            }
            float2 o;
            o.x = rg * (1.f + sc);
            o.y = (a.gn_beta[c] - mr[gg * 2] * rg) * (1.f + sc) + sh;
            a.ab_w[(size_t)img * C + c] = o;
        }
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the coefficient stores have reached L2 before any wave of this workgroup fetches them
    __syncthreads();
}

}  // namespace ire
