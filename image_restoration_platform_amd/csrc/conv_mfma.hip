// conv_mfma.hip -- NHWC bf16 direct (im2col-free) convolution on MFMA for gfx950.
//
// This is the restoration step that stands behind GeminiClient.restoreImage
// (server-node/src/clients/geminiClient.js:32-97): every conv of RestoreNet-v0
// (DESIGN.md "RestoreNet-v0") runs through this one kernel template.
//
// Mapping (implicit GEMM, no im2col buffer):
//   M = output pixels   : a workgroup owns a TH x 32 pixel tile; wave w owns rows w*MT..w*MT+MT-1,
//                         one v_mfma_f32_32x32x16_bf16 M-tile = 32 consecutive pixels of one row
//   N = output channels : NT per workgroup (NT/32 N-tiles per wave), grid covers COUT/NT blocks
//   K = taps x Cin      : K-chunks of KC8*8 input channels; inside a chunk K enumerates
//                         kk = tap*KC8 + c8 (8 channels each); one MFMA k-step = kk {2s, 2s+1}
//                         (lane half h = lane>>5 supplies kk = 2s+h for both A and B)
// LDS (one array): input halo tile [IH*IW pixels][KC8 x 16 B] XOR-swizzled so that the 16-lane
// groups of ds_read_b128 hit 16 distinct 16-B slots; weight slab [kk][NT][8] (16 B per lane,
// conflict-free); after the K loop the same memory is reused as the output tile for a
// transposed, full-line (16 B per lane) store.
// Fusions: GroupNorm+FiLM+SiLU applied as x*A[c]+B[c] -> SiLU while staging the input
// (coefficients from gn_finalize), nearest x2 upsample and stride 2 folded into the staging
// address math, concat folded into the K loop (two sources), bias + residual add + GroupNorm
// partial statistics + (head) "input + residual -> clamp -> u8" in the epilogue.
// Roofline: MFMA (2*9*Cin*Cout flop per output pixel) for C >= 128, HBM for C = 32/64
// ((Cin+Cout)*2 B per pixel; DESIGN.md "kernels").
#include "conv_mfma.hpp"

namespace ire {

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int TW = 32;
constexpr int NTHREADS = 256;

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    f32x2_t f = {a, b};
    bf16x2_t v = __builtin_convertvector(f, bf16x2_t);
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ float bf16_round(float x) {  // value after a bf16 round trip
    return bf16_lo(pack_bf16x2(x, 0.f));
}

__device__ __forceinline__ float silu_f(float y) {
    // y * sigmoid(y); exp2/rcp hardware approximations (rel. err ~1e-6, far below bf16)
    float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * y);
    return y * __builtin_amdgcn_rcpf(1.0f + e);
}

template <int KC8>
__device__ __forceinline__ int swz(int p) {
    // 16-B-chunk XOR so that 16 consecutive pixels with a fixed chunk cover 16 distinct LDS slots
    if constexpr (KC8 == 4) return (p >> 2) & 3;
    else return 0;
}

template <int KC8, int TAPS, int NT, int TH, int STRIDE, bool UPS, int PRO, bool RESID, bool STATS, bool HEAD>
struct ConvCfg {
    static constexpr int MT = TH / 4;
    static constexpr int NTL = NT / 32;
    static constexpr int PAD = (TAPS == 9) ? 1 : 0;
    static constexpr int KSZ = (TAPS == 9) ? 3 : 1;
    static constexpr int IH = (TH - 1) * STRIDE + KSZ;
    static constexpr int IW = (TW - 1) * STRIDE + KSZ;
    static constexpr int NKK = TAPS * KC8;
    static constexpr int NSTEPS = (NKK + 1) / 2;
    static constexpr int IN_CHUNKS = IH * IW * KC8;       // 16-B chunks
    static constexpr int W_CHUNKS = NSTEPS * 2 * NT;      // 16-B chunks
    static constexpr int IN_BYTES = IN_CHUNKS * 16;
    static constexpr int W_BYTES = W_CHUNKS * 16;
    static constexpr int OUT_BYTES = HEAD ? TH * TW * 16 : TH * TW * NT * 2;
    static constexpr int RED_BYTES = 4 * (NT / 8) * 4 * 4;  // [wave][cc][4] floats
    static constexpr int MAIN_BYTES = (IN_BYTES + W_BYTES) > OUT_BYTES ? (IN_BYTES + W_BYTES) : OUT_BYTES;
    static constexpr int LDS_BYTES = MAIN_BYTES + RED_BYTES;
    static constexpr int IN_ITERS = (IN_CHUNKS + NTHREADS - 1) / NTHREADS;
    static constexpr int W_ITERS = (W_CHUNKS + NTHREADS - 1) / NTHREADS;
    static constexpr int OUT_CHUNKS = TH * TW * NT / 8;
    static constexpr int OUT_ITERS = OUT_CHUNKS / NTHREADS;
    static_assert(TH % 4 == 0, "TH must be a multiple of 4 (4 waves)");
    static_assert(NT % 32 == 0 && NT <= 128, "NT");
    static_assert(KC8 == 1 || KC8 == 4, "KC8");
    static_assert(HEAD || OUT_CHUNKS % NTHREADS == 0, "output chunks per thread");
};

template <int KC8, int TAPS, int NT, int TH, int STRIDE, bool UPS, int PRO, bool RESID, bool STATS, bool HEAD>
__global__ __launch_bounds__(NTHREADS) void conv_mfma_kernel(ConvArgs a) {
    using C = ConvCfg<KC8, TAPS, NT, TH, STRIDE, UPS, PRO, RESID, STATS, HEAD>;
    constexpr int MT = C::MT, NTL = C::NTL, IW = C::IW, IH = C::IH, NSTEPS = C::NSTEPS;
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];
    uint4* lds_in = reinterpret_cast<uint4*>(smem);
    uint4* lds_w = reinterpret_cast<uint4*>(smem + C::IN_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // ---- block -> (image, tile, n-block), XCD-aware: blocks b and b+8 share an XCD (and its L2),
    // so give each XCD a contiguous run of logical ids; the NB n-blocks of a tile are adjacent.
    const int total = gridDim.x;
    int logical;
    {
        const int L = blockIdx.x, xcd = L & 7, q = total >> 3, rr = total & 7;
        logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (L >> 3);
    }
    const int nb = logical % a.nblocks;
    const int t = logical / a.nblocks;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int img = t / tiles_per_img;
    const int trem = t - img * tiles_per_img;
    const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int WV = UPS ? a.Win * 2 : a.Win;  // virtual input width (rows: a.iy_lo / a.iy_span)

    f32x16_t acc[MT][NTL];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][j][i] = 0.f;

    const uint4* wbase = reinterpret_cast<const uint4*>(a.w) + (size_t)nb * a.nkc * C::W_CHUNKS;

    for (int kc = 0; kc < a.nkc; ++kc) {
        // ---------------- stage the weight slab (already in LDS order in global memory) ----------
        {
            const uint4* wsrc = wbase + (size_t)kc * C::W_CHUNKS;
#pragma unroll
            for (int it = 0; it < C::W_ITERS; ++it) {
                int i = tid + it * NTHREADS;
                if (i < C::W_CHUNKS) lds_w[i] = wsrc[i];
            }
        }
        // ---------------- stage the input halo tile, prologue fused -------------------------------
        {
            const bool second = (kc >= a.kc_split);
            const int csrc = second ? a.cin1 : a.cin0;                 // channels per pixel of the source
            const int cbase = (second ? (kc - a.kc_split) : kc) * KC8 * 8;
            const unsigned short* src = reinterpret_cast<const unsigned short*>(second ? a.in1 : a.in0);
            const int c8_fixed = tid % KC8;  // NTHREADS % KC8 == 0 => constant per thread
            float cA[8], cB[8];
            if constexpr (PRO == PRO_GN) {
                const float2* ab = a.ab + (size_t)img * a.cin0 + cbase + c8_fixed * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) { float2 v = ab[e]; cA[e] = v.x; cB[e] = v.y; }
            }
            uint4 v[C::IN_ITERS];
            bool ok[C::IN_ITERS];
#pragma unroll
            for (int it = 0; it < C::IN_ITERS; ++it) {
                const int idx = tid + it * NTHREADS;
                const int p = idx / KC8;
                const int py = p / IW, px = p - py * IW;
                const int iy = oy0 * STRIDE + py - C::PAD, ix = ox0 * STRIDE + px - C::PAD;
                ok[it] = (idx < C::IN_CHUNKS) && (unsigned)(iy - a.iy_lo) < (unsigned)a.iy_span && ix >= 0 && ix < WV;
                v[it] = make_uint4(0, 0, 0, 0);
                if (ok[it]) {
                    const int sy = UPS ? (iy >> 1) : iy, sx = UPS ? (ix >> 1) : ix;
                    const size_t pix = ((size_t)img * a.in_rows + sy + a.in_row_off) * a.Win + sx;
                    if constexpr (PRO == PRO_U8) {
                        const unsigned char* pb = reinterpret_cast<const unsigned char*>(a.in0) + pix * 3;
                        // u8 -> bf16 is exact (integers <= 255); channels 3..7 are zero padding
                        v[it].x = pack_bf16x2((float)pb[0], (float)pb[1]);
                        v[it].y = pack_bf16x2((float)pb[2], 0.f);
                    } else {
                        v[it] = *reinterpret_cast<const uint4*>(src + pix * csrc + cbase + c8_fixed * 8);
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < C::IN_ITERS; ++it) {
                const int idx = tid + it * NTHREADS;
                if (idx < C::IN_CHUNKS) {
                    uint4 o = v[it];
                    if constexpr (PRO == PRO_GN) {
                        if (ok[it]) {  // zero padding applies AFTER the activation
                            unsigned wds[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                float y0 = __builtin_fmaf(bf16_lo(wds[d]), cA[2 * d], cB[2 * d]);
                                float y1 = __builtin_fmaf(bf16_hi(wds[d]), cA[2 * d + 1], cB[2 * d + 1]);
                                wds[d] = pack_bf16x2(silu_f(y0), silu_f(y1));
                            }
                            o = make_uint4(wds[0], wds[1], wds[2], wds[3]);
                        }
                    }
                    const int p = idx / KC8;
                    lds_in[p * KC8 + (c8_fixed ^ swz<KC8>(p))] = o;
                }
            }
        }
        __syncthreads();

        // ---------------- MFMA over this K-chunk --------------------------------------------------
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s) {
            int tap, c8;
            if constexpr (KC8 == 4) { tap = s >> 1; c8 = ((s & 1) << 1) + h; }
            else { tap = 2 * s + h; tap = tap < TAPS ? tap : TAPS - 1; c8 = 0; }
            const int ky = (TAPS == 9) ? tap / 3 : 0;
            const int kx = (TAPS == 9) ? tap - ky * 3 : 0;
            bf16x8_t bfrag[NTL], afrag[MT];
#pragma unroll
            for (int j = 0; j < NTL; ++j)
                bfrag[j] = __builtin_bit_cast(bf16x8_t, lds_w[(2 * s + h) * NT + j * 32 + r]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = wave * MT + m;
                const int p = (row * STRIDE + ky) * IW + r * STRIDE + kx;
                afrag[m] = __builtin_bit_cast(bf16x8_t, lds_in[p * KC8 + (c8 ^ swz<KC8>(p))]);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[m], bfrag[j], acc[m][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---------------- epilogue: bias, transpose through LDS, residual, stats, store ---------------
    const int cout0 = nb * NT;
    if constexpr (HEAD) {
        float* lds_o = reinterpret_cast<float*>(smem);  // [TH*TW][4] floats
        if (r < 3) {
            const float bias = a.bias[r];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int x = (i & 3) + 8 * (i >> 2) + 4 * h;
                    lds_o[((wave * MT + m) * TW + x) * 4 + r] = acc[m][0][i] + bias;
                }
        }
        __syncthreads();
        for (int pix = tid; pix < TH * TW; pix += NTHREADS) {
            const int oy = oy0 + (pix >> 5), ox = ox0 + (pix & 31);
            if (oy < a.Hout && ox < a.Wout) {
                const size_t g = (((size_t)img * a.Hout + oy) * a.Wout + ox) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float vv = (float)a.u8_in[g + c] + lds_o[pix * 4 + c];
                    vv = fminf(fmaxf(vv, 0.f), 255.f);
                    a.u8_out[g + c] = (unsigned char)(int)floorf(vv + 0.5f);
                }
            }
        }
    } else {
        unsigned short* lds_o = reinterpret_cast<unsigned short*>(smem);  // [TH*TW][NT] bf16
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const float bias = a.bias[cout0 + j * 32 + r];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int x = (i & 3) + 8 * (i >> 2) + 4 * h;
                    const unsigned u = pack_bf16x2(acc[m][j][i] + bias, 0.f);
                    lds_o[((wave * MT + m) * TW + x) * NT + j * 32 + r] = (unsigned short)(u & 0xffffu);
                }
        }
        __syncthreads();
        constexpr int NCC = NT / 8;  // 16-B chunks per pixel in this n-block
        const int cc = tid % NCC;    // constant per thread (NTHREADS % NCC == 0)
        float sA = 0.f, qA = 0.f, sB = 0.f, qB = 0.f;
        const uint4* lds_o4 = reinterpret_cast<const uint4*>(smem);
#pragma unroll
        for (int it = 0; it < C::OUT_ITERS; ++it) {
            const int idx = tid + it * NTHREADS;
            const int pix = idx / NCC;
            const int oy = oy0 + (pix >> 5), ox = ox0 + (pix & 31);
            if (oy < a.Hout && ox < a.Wout) {
                uint4 o = lds_o4[idx];
                const size_t g = (((size_t)img * a.Hout + oy) * a.Wout + ox) * a.cout + cout0 + cc * 8;
                unsigned wds[4] = {o.x, o.y, o.z, o.w};
                if constexpr (RESID) {
                    const uint4 rv = *reinterpret_cast<const uint4*>(a.resid + g);
                    const unsigned rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        wds[d] = pack_bf16x2(bf16_lo(wds[d]) + bf16_lo(rw[d]), bf16_hi(wds[d]) + bf16_hi(rw[d]));
                }
                if constexpr (STATS) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const float f0 = bf16_lo(wds[d]), f1 = bf16_hi(wds[d]);
                        if (d < 2) { sA += f0 + f1; qA += f0 * f0 + f1 * f1; }
                        else { sB += f0 + f1; qB += f0 * f0 + f1 * f1; }
                    }
                }
                *reinterpret_cast<uint4*>(a.out + g) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
            }
        }
        if constexpr (STATS) {
            // lanes with equal cc are NCC apart: butterfly over offsets NCC..32 (fixed order => deterministic)
#pragma unroll
            for (int off = NCC; off < 64; off <<= 1) {
                sA += __shfl_xor(sA, off, 64); qA += __shfl_xor(qA, off, 64);
                sB += __shfl_xor(sB, off, 64); qB += __shfl_xor(qB, off, 64);
            }
            float* red = reinterpret_cast<float*>(smem + C::MAIN_BYTES);  // [4 waves][NCC][4]
            if (lane < NCC) {
                float* d = red + (wave * NCC + lane) * 4;
                d[0] = sA; d[1] = qA; d[2] = sB; d[3] = qB;
            }
            __syncthreads();
            const int G = a.group_size;            // channels per GroupNorm group of the OUTPUT tensor
            const int ngl = NT / G;                // groups covered by this n-block
            if (tid < ngl) {
                float s = 0.f, q = 0.f;
                if (G == 4) {                      // chunk = two groups: (cc, half)
                    const int c = tid >> 1, half = tid & 1;
                    for (int w = 0; w < 4; ++w) {
                        s += red[(w * NCC + c) * 4 + 2 * half];
                        q += red[(w * NCC + c) * 4 + 2 * half + 1];
                    }
                } else {                           // group = G/8 whole chunks
                    const int cpg = G >> 3;
                    for (int w = 0; w < 4; ++w)
                        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
                            s += red[(w * NCC + c) * 4 + 0] + red[(w * NCC + c) * 4 + 2];
                            q += red[(w * NCC + c) * 4 + 1] + red[(w * NCC + c) * 4 + 3];
                        }
                }
                const int gg = cout0 / G + tid;
                float* st = a.stats + (((size_t)img * tiles_per_img + trem) * 8 + gg) * 2;
                st[0] = s; st[1] = q;
            }
        }
    }
}

template <int KC8, int TAPS, int NT, int TH, int STRIDE, bool UPS, int PRO, bool RESID, bool STATS, bool HEAD>
void launch_one(const ConvArgs& a, hipStream_t stream) {
    const int total = a.tiles_x * a.tiles_y * a.nimg * a.nblocks;
    hipLaunchKernelGGL((conv_mfma_kernel<KC8, TAPS, NT, TH, STRIDE, UPS, PRO, RESID, STATS, HEAD>), dim3(total),
                       dim3(NTHREADS), 0, stream, a);
    IRE_HIP(hipGetLastError());
}

}  // namespace

int conv_tile_h(ConvKind kind) { return kind == CONV_DOWN ? 4 : 8; }
int conv_nt(ConvKind kind, int cout) {
    if (kind == CONV_STEM || kind == CONV_HEAD) return 32;
    return cout >= 64 ? 64 : 32;
}
int conv_nsteps(ConvKind kind) {
    switch (kind) {
        case CONV_STEM: return 5;   // 9 taps x 1 chunk, padded to 10 kk
        case CONV_FUSE: return 2;   // 1 tap x 4 chunks
        default: return 18;         // 9 taps x 4 chunks
    }
}

void conv_launch(ConvKind kind, const ConvArgs& a, hipStream_t stream) {
    const int nt = conv_nt(kind, a.cout);
    switch (kind) {
        case CONV_STEM:
            launch_one<1, 9, 32, 8, 1, false, PRO_U8, false, true, false>(a, stream); break;
        case CONV_RB1:
            if (nt == 32) launch_one<4, 9, 32, 8, 1, false, PRO_GN, false, true, false>(a, stream);
            else launch_one<4, 9, 64, 8, 1, false, PRO_GN, false, true, false>(a, stream);
            break;
        case CONV_RB2:
            if (nt == 32) launch_one<4, 9, 32, 8, 1, false, PRO_GN, true, true, false>(a, stream);
            else launch_one<4, 9, 64, 8, 1, false, PRO_GN, true, true, false>(a, stream);
            break;
        case CONV_DOWN:
            launch_one<4, 9, 64, 4, 2, false, PRO_NONE, false, true, false>(a, stream); break;
        case CONV_UP:
            if (nt == 32) launch_one<4, 9, 32, 8, 1, true, PRO_NONE, false, false, false>(a, stream);
            else launch_one<4, 9, 64, 8, 1, true, PRO_NONE, false, false, false>(a, stream);
            break;
        case CONV_FUSE:
            if (nt == 32) launch_one<4, 1, 32, 8, 1, false, PRO_NONE, false, true, false>(a, stream);
            else launch_one<4, 1, 64, 8, 1, false, PRO_NONE, false, true, false>(a, stream);
            break;
        case CONV_HEAD:
            launch_one<4, 9, 32, 8, 1, false, PRO_GN, false, false, true>(a, stream); break;
    }
}

}  // namespace ire
