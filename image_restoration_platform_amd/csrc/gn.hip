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

namespace ire {

namespace {

__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ stats, int ntiles, int C,
                                                          int hw, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ film, int film_stride,
                                                          int film_off, float2* __restrict__ ab) {
    __shared__ double red[32][8][2];
    __shared__ float s_mean[8], s_rstd[8];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int g = tid & 7, part = tid >> 3;  // 32 parts x 8 groups
    double s = 0.0, q = 0.0;
    const float* st = stats + (size_t)img * ntiles * 16;
    for (int t = part; t < ntiles; t += 32) {
        s += (double)st[(size_t)t * 16 + g * 2];
        q += (double)st[(size_t)t * 16 + g * 2 + 1];
    }
    red[part][g][0] = s;
    red[part][g][1] = q;
    __syncthreads();
    if (tid < 8) {
        double ss = 0.0, qq = 0.0;
        for (int p = 0; p < 32; ++p) { ss += red[p][tid][0]; qq += red[p][tid][1]; }
        const double cnt = (double)hw * (double)(C / 8);
        const double mean = ss / cnt;
        double var = qq / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[tid] = (float)mean;
        s_rstd[tid] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
    const int G = C / 8;
    for (int c = tid; c < C; c += 256) {
        const int gi = c / G;
        const float rg = s_rstd[gi] * gamma[c];
        float sc = 0.f, sh = 0.f;
        if (film) {
            sc = film[(size_t)img * film_stride + film_off + c];
            sh = film[(size_t)img * film_stride + film_off + C + c];
        }
        float2 o;
        o.x = rg * (1.f + sc);
        o.y = (beta[c] - s_mean[gi] * rg) * (1.f + sc) + sh;
        ab[(size_t)img * C + c] = o;
    }
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
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(nimg), dim3(256), 0, stream, d_stats, ntiles, C, hw, d_gamma,
                       d_beta, d_film, film_stride, film_off, d_ab);
    IRE_HIP(hipGetLastError());
}

void film_launch(const float* d_cond, int nimg, const float* d_w, const float* d_b, int nout, float* d_film,
                 hipStream_t stream) {
    hipLaunchKernelGGL(film_kernel, dim3(ceil_div(nout, 128), nimg), dim3(128), 0, stream, d_cond, d_w, d_b,
                       nout, d_film);
    IRE_HIP(hipGetLastError());
}

}  // namespace ire
