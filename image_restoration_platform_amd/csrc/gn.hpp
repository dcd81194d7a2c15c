// gn.hpp -- GroupNorm finalize / FiLM launch interface (gn.hip).
#pragma once
#include "common.hpp"

namespace ire {

// stats: [nimg][ntiles][8][2] partial (sum, sumsq); hw = pixels per image of the normalised tensor.
// film (may be null): [nimg][film_stride], scale at film_off..+C, shift at film_off+C..+2C.
void gn_finalize_launch(const float* d_stats, int nimg, int ntiles, int C, int hw, const float* d_gamma,
                        const float* d_beta, const float* d_film, int film_stride, int film_off,
                        float2* d_ab, hipStream_t stream);

// film[nimg][nout] = W[nout][7] * cond[nimg][8 (7 used)] + b
void film_launch(const float* d_cond, int nimg, const float* d_w, const float* d_b, int nout, float* d_film,
                 hipStream_t stream);

}  // namespace ire
