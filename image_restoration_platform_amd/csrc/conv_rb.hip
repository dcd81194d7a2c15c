// conv_rb.hip -- persistent, software-pipelined 3x3 C->C convolution for the ResBlock convs
// (32 of RestoreNet-v0's 43 convolutions, 79 % of its FLOPs) on gfx950.
//
// Same math, LDS images and fusions as conv_mfma.hip (GroupNorm+FiLM+SiLU prologue, bias,
// residual, GroupNorm partial statistics); what changes is the schedule, designed around one
// 512-thread workgroup per CU that never leaves it:
//   * work item = (16x32 pixel tile, 64- or 32-channel n-block); items are dealt so that the
//     32 workgroups of one XCD walk a contiguous run of tiles (halo rows and, for C >= 128, the
//     n-blocks of a tile are shared through that XCD's L2);
//   * an item is nkc "stages" (32 input channels each).  Stage s runs its 18 MFMA k-steps from
//     LDS buffer s&1 while (a) the global loads of stage s+2 are in flight into registers,
//     (b) the registers of stage s+1 (loaded one stage ago) are normalised/activated and
//     written to the other LDS buffer, interleaved between the MFMA groups so the VALU work of
//     one wave hides under the matrix work of its SIMD partner, (c) the weight slab of stage
//     s+1 is staged.  One barrier per stage.
//   * C = 32: the whole 18 KB weight set stays resident in LDS for the life of the workgroup.
// Roofline: MFMA for C >= 128 (2*9*C*C flop per pixel), HBM for C = 32/64 (4C..6C B per pixel).
#include "conv_mfma.hpp"

namespace ire {

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int RB_THREADS = 512;
constexpr int RB_TH = 16, RB_TW = 32;
constexpr int RB_IH = RB_TH + 2, RB_IW = RB_TW + 2;
constexpr int RB_IN_CHUNKS = RB_IH * RB_IW * 4;                                // 2448 x 16 B
constexpr int RB_IN_BYTES = RB_IN_CHUNKS * 16;                                  // 39168
constexpr int RB_IN_ITERS = (RB_IN_CHUNKS + RB_THREADS - 1) / RB_THREADS;       // 5
constexpr int RB_NSTEPS = 18;

__device__ __forceinline__ unsigned rb_pack(float a, float b) {
    f32x2_t f = {a, b};
    bf16x2_t v = __builtin_convertvector(f, bf16x2_t);
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float rb_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float rb_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ float rb_silu(float y) {
    float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * y);
    return y * __builtin_amdgcn_rcpf(1.0f + e);
}

struct RbItem {
    int img, ty, tx, nb, tile;  // tile = ty*tiles_x + tx
};

struct RbRegs {          // one prefetched input stage
    uint4 v[RB_IN_ITERS];
    unsigned ok;         // bit it: chunk it is inside the image
};

template <int NT, bool RESID, bool WRES>
struct RbCfg {
    static constexpr int NTL = NT / 32;
    static constexpr int W_CHUNKS = RB_NSTEPS * 2 * NT;   // 16-B chunks per (n-block, k-chunk) slab
    static constexpr int W_BYTES = W_CHUNKS * 16;
    static constexpr int W_ITERS = (W_CHUNKS + RB_THREADS - 1) / RB_THREADS;
    // streaming: [in0 | w0][in1 | w1]; resident weights: [in0][in1][w]
    static constexpr int BUF_STRIDE = WRES ? RB_IN_BYTES : RB_IN_BYTES + W_BYTES;
    static constexpr int W_OFF0 = WRES ? 2 * RB_IN_BYTES : RB_IN_BYTES;
    static constexpr int MAIN_BYTES = WRES ? 2 * RB_IN_BYTES + W_BYTES : 2 * (RB_IN_BYTES + W_BYTES);
    static constexpr int RED_BYTES = 8 * (NT / 8) * 4 * 4;
    static constexpr int LDS_BYTES = MAIN_BYTES + RED_BYTES;
    static constexpr int OUT_ITERS = RB_TH * RB_TW * (NT / 8) / RB_THREADS;    // NT/8
    static_assert(RB_TH * RB_TW * NT * 2 <= BUF_STRIDE, "output tile must fit in one stage buffer");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int NT, bool RESID, bool WRES>
__global__ __launch_bounds__(RB_THREADS) void conv_rb_kernel(ConvArgs a) {
    using C = RbCfg<NT, RESID, WRES>;
    constexpr int NTL = C::NTL;
    constexpr int NCC = NT / 8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int c8_fixed = tid & 3;  // this thread always stages the same 8-channel slice of a pixel

    // ---- persistent work assignment (XCD-aware: blocks b and b+8 share an XCD / L2) ----------------
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int items = tiles_per_img * a.nimg * a.nblocks;
    const int G = gridDim.x;
    const int X = G < 8 ? G : 8;
    const int xcd = blockIdx.x % X, jx = blockIdx.x / X;
    const int nwx = (G - xcd + X - 1) / X;                       // workgroups in this XCD group
    const int lo = (int)((long long)items * xcd / X), hi = (int)((long long)items * (xcd + 1) / X);
    const int my_items = (lo + jx < hi) ? (hi - lo - jx + nwx - 1) / nwx : 0;
    const int nkc = a.nkc;
    const int S = my_items * nkc;                                // stages this workgroup runs
    if (S == 0) return;

    auto item_of = [&](int k) -> RbItem {                        // k-th item of this workgroup
        const int L = lo + jx + k * nwx;
        RbItem it;
        it.nb = L % a.nblocks;
        const int t = L / a.nblocks;
        it.img = t / tiles_per_img;
        it.tile = t - it.img * tiles_per_img;
        it.ty = it.tile / a.tiles_x;
        it.tx = it.tile - it.ty * a.tiles_x;
        return it;
    };

    const unsigned short* src = reinterpret_cast<const unsigned short*>(a.in0);
    const int Cin = a.cin0;

    // ---- per-lane LDS offsets of the 18 (m-tile, tap) A fragments: they never change ---------------
    // p = (row+ky)*IW + r + kx ; 16-B chunk index = p*4 + (c8 ^ ((p>>2)&3)) with c8 = 2*(s&1) + h
    int a_off[2][9];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int p = (wave * 2 + m + ky) * RB_IW + r + kx;
            a_off[m][tap] = (p * 4 + (h ^ ((p >> 2) & 3))) * 16;   // (s&1) toggles bit 5 (chunk bit 1)
        }
    const int b_off = (h * NT + r) * 16;                           // + (2s*NT + j*32)*16

    // ---- helpers -------------------------------------------------------------------------------------
    auto load_stage = [&](int s, RbRegs& R) {
        const RbItem it = item_of(s / nkc);
        const int kc = s - (s / nkc) * nkc;
        const int oy0 = it.ty * RB_TH, ox0 = it.tx * RB_TW;
        const unsigned short* base = src + (size_t)it.img * a.Hin * a.Win * Cin + kc * 32 + c8_fixed * 8;
        R.ok = 0;
        int t2 = tid;
        asm volatile("" : "+v"(t2));   // recompute the chunk index math per stage instead of keeping it live
#pragma unroll
        for (int i = 0; i < RB_IN_ITERS; ++i) {
            const int idx = t2 + i * RB_THREADS;
            const int p = idx >> 2;
            const int py = p / RB_IW, px = p - py * RB_IW;
            const int iy = oy0 + py - 1, ix = ox0 + px - 1;
            const bool ok = (idx < RB_IN_CHUNKS) && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
            const int cy = min(max(iy, 0), a.Hin - 1), cx = min(max(ix, 0), a.Win - 1);
            R.v[i] = *reinterpret_cast<const uint4*>(base + ((size_t)cy * a.Win + cx) * Cin);  // always in bounds
            R.ok |= ok ? (1u << i) : 0u;
        }
    };
    auto load_coeffs = [&](int s, float (&cA)[8], float (&cB)[8]) {
        const RbItem it = item_of(s / nkc);
        const int kc = s - (s / nkc) * nkc;
        const float4* ab = reinterpret_cast<const float4*>(a.ab + (size_t)it.img * Cin + kc * 32 + c8_fixed * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float4 v = ab[e];
            cA[2 * e] = v.x; cB[2 * e] = v.y; cA[2 * e + 1] = v.z; cB[2 * e + 1] = v.w;
        }
    };
    auto store_chunk = [&](int i, const RbRegs& R, const float (&cA)[8], const float (&cB)[8], uint4* lds_in) {
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int idx = t2 + i * RB_THREADS;
        if (idx < RB_IN_CHUNKS) {
            uint4 o = make_uint4(0, 0, 0, 0);          // zero padding applies AFTER the activation
            if (R.ok & (1u << i)) {
                unsigned w[4] = {R.v[i].x, R.v[i].y, R.v[i].z, R.v[i].w};
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float y0 = __builtin_fmaf(rb_lo(w[d]), cA[2 * d], cB[2 * d]);
                    const float y1 = __builtin_fmaf(rb_hi(w[d]), cA[2 * d + 1], cB[2 * d + 1]);
                    w[d] = rb_pack(rb_silu(y0), rb_silu(y1));
                }
                o = make_uint4(w[0], w[1], w[2], w[3]);
            }
            const int p = idx >> 2;
            lds_in[p * 4 + (c8_fixed ^ ((p >> 2) & 3))] = o;
        }
    };
    auto wslab = [&](int s) -> const uint4* {
        const RbItem it = item_of(s / nkc);
        const int kc = s - (s / nkc) * nkc;
        return reinterpret_cast<const uint4*>(a.w) + ((size_t)it.nb * nkc + kc) * C::W_CHUNKS;
    };

    f32x16_t acc[2][NTL];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][j][i] = 0.f;
    };
    zero_acc();

    // pending GroupNorm partials of the previous item (finished after the stage barrier)
    int st_img = -1, st_tile = 0, st_nb = 0;
    float* red = reinterpret_cast<float*>(smem + C::MAIN_BYTES);   // [8 waves][NCC][4]
    auto flush_stats = [&]() {
        if (st_img < 0) return;
        const int Gs = a.group_size, ngl = NT / Gs;
        if (tid < ngl) {
            float s = 0.f, q = 0.f;
            if (Gs == 4) {
                const int c = tid >> 1, half = tid & 1;
                for (int w = 0; w < 8; ++w) { s += red[(w * NCC + c) * 4 + 2 * half]; q += red[(w * NCC + c) * 4 + 2 * half + 1]; }
            } else {
                const int cpg = Gs >> 3;
                for (int w = 0; w < 8; ++w)
                    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
                        s += red[(w * NCC + c) * 4 + 0] + red[(w * NCC + c) * 4 + 2];
                        q += red[(w * NCC + c) * 4 + 1] + red[(w * NCC + c) * 4 + 3];
                    }
            }
            const int gg = (st_nb * NT) / Gs + tid;
            float* st = a.stats + (((size_t)st_img * tiles_per_img + st_tile) * 8 + gg) * 2;
            st[0] = s; st[1] = q;
        }
        st_img = -1;
    };

    // ---- one pipeline stage (PAR = s & 1 selects buffers and register sets statically) ---------------
    RbRegs R0, R1;
    auto stage = [&](int s, auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        uint4* in_cur = reinterpret_cast<uint4*>(smem + PAR * C::BUF_STRIDE);
        uint4* in_nxt = reinterpret_cast<uint4*>(smem + (PAR ^ 1) * C::BUF_STRIDE);
        const unsigned char* w_cur = smem + C::W_OFF0 + (WRES ? 0 : PAR * C::BUF_STRIDE);
        unsigned char* w_nxt = smem + C::W_OFF0 + (WRES ? 0 : (PAR ^ 1) * C::BUF_STRIDE);
        RbRegs& Rn = PAR ? R0 : R1;   // holds stage s+1 (loaded during stage s-1)
        RbRegs& Rf = PAR ? R1 : R0;   // free: receives stage s+2
        const bool have1 = (s + 1 < S), have2 = (s + 2 < S);

        // (1) coefficients of stage s+1, (2) its weight slab, (3) input prefetch of stage s+2 -- in this
        // order so waiting for (1)/(2) never waits for (3) (VMEM returns in issue order)
        float cA[8], cB[8];
        if (have1) load_coeffs(s + 1, cA, cB);
        u32x4_t wreg[C::W_ITERS];
        if constexpr (!WRES) {   // unconditional (clamped) so the array stays in registers
            const u32x4_t* ws = reinterpret_cast<const u32x4_t*>(wslab(have1 ? s + 1 : s));
#pragma unroll
            for (int i = 0; i < C::W_ITERS; ++i) {
                const int idx = tid + i * RB_THREADS;
                wreg[i] = ws[idx < C::W_CHUNKS ? idx : 0];
            }
        }
        if (have2) load_stage(s + 2, Rf);

        // (4) 18 MFMA k-steps from the current buffer, the s+1 transform interleaved between groups
        const int wsel = WRES ? ((s - (s / nkc) * nkc) * C::W_BYTES) : 0;   // resident: slab of this kc
        const unsigned char* wb = w_cur + wsel + b_off;
        const unsigned char* ib = reinterpret_cast<const unsigned char*>(in_cur);
#pragma unroll
        for (int st = 0; st < RB_NSTEPS; ++st) {
            const int tap = st >> 1;
            bf16x8_t bfrag[NTL], afrag[2];
#pragma unroll
            for (int j = 0; j < NTL; ++j)
                bfrag[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wb + (2 * st * NT + j * 32) * 16));
#pragma unroll
            for (int m = 0; m < 2; ++m)
                afrag[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ib + (a_off[m][tap] ^ ((st & 1) << 5))));
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[m], bfrag[j], acc[m][j], 0, 0, 0);
            if (have1 && (st % 3) == 2 && (st / 3) < RB_IN_ITERS) store_chunk(st / 3, Rn, cA, cB, in_nxt);
        }
        // (5) weight slab of stage s+1 -> other buffer
        if constexpr (!WRES) {
#pragma unroll
            for (int i = 0; i < C::W_ITERS; ++i) {
                const int idx = tid + i * RB_THREADS;
                if (idx < C::W_CHUNKS) reinterpret_cast<u32x4_t*>(w_nxt)[idx] = wreg[i];
            }
        }
        // (6) last k-chunk of the item: epilogue through the current buffer
        const int kc = s - (s / nkc) * nkc;
        if (kc == nkc - 1) {
            const RbItem it = item_of(s / nkc);
            const int oy0 = it.ty * RB_TH, ox0 = it.tx * RB_TW;
            const int cout0 = it.nb * NT;
            __syncthreads();                                   // every wave is done reading buf[cur]
            flush_stats();                                     // (red[] of the previous item is complete)
            unsigned short* lds_o = reinterpret_cast<unsigned short*>(smem + PAR * C::BUF_STRIDE);  // [512 px][NT]
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const float bias = a.bias[cout0 + j * 32 + r];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int x = (i & 3) + 8 * (i >> 2) + 4 * h;
                        lds_o[((wave * 2 + m) * RB_TW + x) * NT + j * 32 + r] =
                            (unsigned short)(rb_pack(acc[m][j][i] + bias, 0.f) & 0xffffu);
                    }
            }
            zero_acc();
            __syncthreads();
            const int cc = tid % NCC;
            float sA = 0.f, qA = 0.f, sB = 0.f, qB = 0.f;
            const uint4* lds_o4 = reinterpret_cast<const uint4*>(lds_o);
#pragma unroll
            for (int k = 0; k < C::OUT_ITERS; ++k) {
                const int idx = tid + k * RB_THREADS;
                const int pix = idx / NCC;
                const int oy = oy0 + (pix >> 5), ox = ox0 + (pix & 31);
                if (oy < a.Hout && ox < a.Wout) {
                    const uint4 o = lds_o4[idx];
                    const size_t g = (((size_t)it.img * a.Hout + oy) * a.Wout + ox) * a.cout + cout0 + cc * 8;
                    unsigned w[4] = {o.x, o.y, o.z, o.w};
                    if constexpr (RESID) {
                        const uint4 rv = *reinterpret_cast<const uint4*>(a.resid + g);
                        const unsigned rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                        for (int d = 0; d < 4; ++d)
                            w[d] = rb_pack(rb_lo(w[d]) + rb_lo(rw[d]), rb_hi(w[d]) + rb_hi(rw[d]));
                    }
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const float f0 = rb_lo(w[d]), f1 = rb_hi(w[d]);
                        if (d < 2) { sA += f0 + f1; qA += f0 * f0 + f1 * f1; }
                        else { sB += f0 + f1; qB += f0 * f0 + f1 * f1; }
                    }
                    *reinterpret_cast<uint4*>(a.out + g) = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
#pragma unroll
            for (int off = NCC; off < 64; off <<= 1) {
                sA += __shfl_xor(sA, off, 64); qA += __shfl_xor(qA, off, 64);
                sB += __shfl_xor(sB, off, 64); qB += __shfl_xor(qB, off, 64);
            }
            if (lane < NCC) {
                float* d = red + (wave * NCC + lane) * 4;
                d[0] = sA; d[1] = qA; d[2] = sB; d[3] = qB;
            }
            st_img = it.img; st_tile = it.tile; st_nb = it.nb;
        }
        // (7) stage barrier: buf[nxt] complete, buf[cur] (and red[]) free
        __syncthreads();
    };

    // ---- prologue: stage 0 into buffer 0, stage 1 into registers --------------------------------------
    {
        float cA[8], cB[8];
        load_coeffs(0, cA, cB);
        load_stage(0, R0);
        if (WRES) {   // whole weight set of this n-block stays in LDS (nblocks == 1)
            const uint4* ws = reinterpret_cast<const uint4*>(a.w);
            uint4* wd = reinterpret_cast<uint4*>(smem + C::W_OFF0);
            for (int i = tid; i < C::W_CHUNKS * nkc; i += RB_THREADS) wd[i] = ws[i];
        } else {
            const uint4* ws = wslab(0);
            uint4* wd = reinterpret_cast<uint4*>(smem + C::W_OFF0);
            for (int i = tid; i < C::W_CHUNKS; i += RB_THREADS) wd[i] = ws[i];
        }
        uint4* in0 = reinterpret_cast<uint4*>(smem);
#pragma unroll
        for (int i = 0; i < RB_IN_ITERS; ++i) store_chunk(i, R0, cA, cB, in0);
        if (S > 1) load_stage(1, R1);
    }
    __syncthreads();

    for (int s = 0; s < S; s += 2) {
        stage(s, std::integral_constant<int, 0>{});
        if (s + 1 < S) stage(s + 1, std::integral_constant<int, 1>{});
    }
    flush_stats();
}

template <int NT, bool RESID, bool WRES>
void launch_rb(const ConvArgs& a, hipStream_t stream) {
    const int items = a.tiles_x * a.tiles_y * a.nimg * a.nblocks;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = items < cus ? items : cus;
    hipLaunchKernelGGL((conv_rb_kernel<NT, RESID, WRES>), dim3(grid), dim3(RB_THREADS), 0, stream, a);
    IRE_HIP(hipGetLastError());
}

}  // namespace

void conv_rb_launch(bool resid, const ConvArgs& a, hipStream_t stream) {
    if (a.cout == 32) {
        if (resid) launch_rb<32, true, true>(a, stream); else launch_rb<32, false, true>(a, stream);
    } else {
        if (resid) launch_rb<64, true, false>(a, stream); else launch_rb<64, false, false>(a, stream);
    }
}

}  // namespace ire
