// gn.hip -- GroupNorm statistics finalize + FiLM conditioning for RestoreNet-v0 (gfx950).
//
// GroupNorm's global (per image, per group) statistics cannot be fused into the consuming
// convolution, so they are produced in two stages (SURVEY.md section 7 "hard parts"):
//   1. every producing conv writes per-workgroup partial (sum, sumsq) per group (conv_mfma.hip);
//   2. gn_finalize_kernel reduces the partials in a fixed order (deterministic, in double) and
//      folds mean/rstd, the layer's gamma/beta and the image's FiLM (scale, shift) into ONE
//      per-(image, channel) pair:   y = x * A + B,
//         A = rstd*gamma*(1+s),  B = (beta - mean*rstd*gamma)*(1+s) + t
//      which the consuming conv applies while staging its input tile, followed by SiLU.
// FiLM: the 7 classifier scores condition the restoration (the reference conditions the
// provider call on the classification: restorator.js:57-94): film = Wf * scores + bf, sliced per level.
#include "gn.hpp"
#include "gn_fold.hpp"

#include <algorithm>

namespace ire {

namespace {

// The standalone finalize (row strips: one finalize over the gathered partials array; the A/B fallback kernels): ONE workgroup of
// 512 threads per image running gn_fold -- the very function, with the very thread count, that the consuming convolutions run in
// their prologues (gn_fold.hpp).  Same partials, same double-precision addition tree, same float arithmetic afterwards: the
// coefficients are bit-identical to the folded finalize BY CONSTRUCTION, which is what makes a tiled run (this kernel) equal to
// the untiled run (the folded prologue) bit for bit.
__global__ __launch_bounds__(512) void gn_finalize_kernel(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[512 * 16 + 64];
    gn_fold(a, smem, blockIdx.x, blockIdx.x, 512);
}

__global__ void film_kernel(const float* __restrict__ cond, const float* __restrict__ w,
                            const float* __restrict__ b, int nout, float* __restrict__ film) {
    const int img = blockIdx.y;
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nout) return;
    float acc = b[o];
#pragma unroll
    for (int k = 0; k < 7; ++k) acc = __builtin_fmaf(w[o * 7 + k], cond[img * 8 + k], acc);
    film[(size_t)img * nout + o] = acc;
}

}  // namespace

void gn_finalize_launch(const float* d_stats, int nimg, int ntiles, int C, int hw, const float* d_gamma,
                        const float* d_beta, const float* d_film, int film_stride, int film_off,
                        float2* d_ab, hipStream_t stream) {
    ConvArgs a{};
    a.cin0 = C; a.gn_stats = d_stats; a.gn_parts = ntiles; a.gn_hw = hw; a.gn_gamma = d_gamma; a.gn_beta = d_beta;
    a.gn_film = d_film; a.gn_film_stride = film_stride; a.gn_film_off = film_off; a.ab_w = d_ab;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(nimg), dim3(512), 0, stream, a);
    IRE_HIP(hipGetLastError());
}

void film_launch(const float* d_cond, int nimg, const float* d_w, const float* d_b, int nout, float* d_film,
                 hipStream_t stream) {
    hipLaunchKernelGGL(film_kernel, dim3(ceil_div(nout, 128), nimg), dim3(128), 0, stream, d_cond, d_w, d_b,
                       nout, d_film);
    IRE_HIP(hipGetLastError());
}

}  // namespace ire
