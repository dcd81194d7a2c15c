// classifier_finalize.hpp -- scores from the integer accumulators of the classifier scan.
//
// Mirrors the score formulas of server-node/src/services/classifier.js:
//   blur :118-122, noise :145-146, lowLight :159-167, compression :180-186 + :299-303,
//   scratch :335-336, fade :223-228 + :272-286, colorShift :245-253,
// with the per-channel mean / sample-stdev that sharp's stats() returns (libvips:
// mean = S/N, stdev = sqrt(|S2 - S*S/N| / (N-1))) and the
// population variance of classifier.js:262-266 evaluated exactly from integer sums.
// __host__ __device__ so the engine can run it on either side with identical IEEE steps
// (compiled with FP contraction off: no FMA fusing).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace ire {

enum { CLS_NSUMS = 14 };
// sums[0..2]  sum of R,G,B            sums[3..5]  sum of squares of R,G,B
// sums[6,7]   sum / sum^2 of the sigma=1 blurred RGB bytes (all 3 channels jointly)
// sums[8,9]   sum / sum^2 of clip_u8(Laplacian-8(grey))
// sums[10,11] sum / sum^2 of clip_u8(high-pass-9(grey))
// sums[12,13] scratch vertical / horizontal pair counts

__host__ __device__ inline uint64_t umulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

__host__ __device__ inline double js_min(double a, double b) {
    if (a != a || b != b) return NAN;
    return a < b ? a : b;
}
__host__ __device__ inline double js_max(double a, double b) {
    if (a != a || b != b) return NAN;
    return a > b ? a : b;
}

#pragma clang fp contract(off)
__host__ __device__ inline double popvar_exact(uint64_t n, uint64_t s, uint64_t s2) {
    // (n*s2 - s*s) as a 128-bit integer, then (double)hi * 2^64 + (double)lo, then / n^2
    uint64_t a_lo = n * s2, a_hi = umulhi64(n, s2);
    uint64_t b_lo = s * s, b_hi = umulhi64(s, s);
    uint64_t lo = a_lo - b_lo;
    uint64_t hi = a_hi - b_hi - (a_lo < b_lo ? 1u : 0u);
    double d = (double)hi * 18446744073709551616.0 + (double)lo;
    double dn = (double)n;
    return d / (dn * dn);
}

__host__ __device__ inline void cls_finalize_one(const uint64_t* S, uint64_t px, int is_jpeg,
                                                 double* scores, int32_t* label) {
#pragma clang fp contract(off)
    double mean[3], sd[3];
    double vals = (double)px;
    for (int c = 0; c < 3; ++c) {
        double ds = (double)S[c], ds2 = (double)S[3 + c];
        mean[c] = ds / vals;
        sd[c] = sqrt(fabs(ds2 - (ds * ds / vals)) / (vals - 1.0));
    }
    double var_e8 = popvar_exact(px, S[8], S[9]);
    double var_e9 = popvar_exact(px, S[10], S[11]);
    double nv = js_min(var_e8 / 1000.0, 1.0);
    scores[0] = js_max(0.0, 1.0 - nv);
    scores[1] = js_min(sqrt(var_e9) / 50.0, 1.0);
    double mb = (((0.0 + mean[0]) + mean[1]) + mean[2]) / 3.0;
    double nb = mb / 255.0;
    scores[2] = (nb < 0.3) ? js_min((0.3 - nb) * 2.0, 1.0) : 0.0;
    if (!is_jpeg) {
        scores[3] = 0.0;
    } else {
        double var_rgb = popvar_exact(3 * px, S[0] + S[1] + S[2], S[3] + S[4] + S[5]);
        double var_blur = popvar_exact(3 * px, S[6], S[7]);
        double delta = js_max(0.0, var_rgb - var_blur);
        scores[3] = js_min(js_min(delta / 500.0, 1.0), 1.0);
    }
    double total = (double)(S[12] + S[13]);
    scores[4] = js_min(js_min(total / 1000.0, 1.0), 1.0);
    double sat = sqrt((sd[0] * sd[0] + sd[1] * sd[1]) + sd[2] * sd[2]) / 255.0;
    double colorfulness = js_min(sat, 1.0);
    double avg_sd = (((0.0 + sd[0]) + sd[1]) + sd[2]) / 3.0;
    double contrast = js_min(avg_sd / 64.0, 1.0);
    scores[5] = js_min((1.0 - colorfulness) * 0.6 + (1.0 - contrast) * 0.4, 1.0);
    double avg = ((mean[0] + mean[1]) + mean[2]) / 3.0;
    double dr = avg > 0 ? fabs(mean[0] - avg) / avg : 0.0;
    double dg = avg > 0 ? fabs(mean[1] - avg) / avg : 0.0;
    double db = avg > 0 ? fabs(mean[2] - avg) / avg : 0.0;
    scores[6] = js_min(js_max(js_max(dr, dg), db) * 2.0, 1.0);
    int best = 0;  // first-max argmax in key order
    for (int i = 1; i < 7; ++i)
        if (scores[i] > scores[best]) best = i;
    *label = best;
}

}  // namespace ire
