// conv_w4.hip -- MFMA-bound variant of the persistent pipelined 3x3 convolution for the C >= 128 levels:
// ONE wave per SIMD (256-thread workgroup per CU, up to 512 registers per wave), wave tile = 4 pixel rows x 128
// output channels.
//
// Why (measured on conv_rb.hip, profiles/r01_ablation.md): with two same-program waves per SIMD the MFMA pipe
// was ~77 % busy inside the k-loop (the waves contend; static priority only swaps which one waits), every MFMA
// needed one ds_read_b128, and a 64-channel n-block re-read (and re-staged) the input tile 2-4 times.  Here a
// k-step is 8 fragment reads for 16 MFMAs (0.5 reads per MFMA), the n-block is 128 channels (input staged once
// for C = 128, twice for C = 256), and the single wave's non-MFMA instructions issue in the 24 free cycles after
// each 8-cycle MFMA issue slot.
//
// Stage = 16 input channels (pixel = 2 x 16-B chunks in the LDS tile): 9 k-steps (one per tap) x 16 MFMAs = 144
// MFMAs per wave = 4608 pipe cycles.  LDS: 2 x (19.6 KB input tile + 36.9 KB weight slab [tap][c8][128][8]).
// Pipeline, VMEM ordering rules, LDS-DMA weights, stage queue: as conv_rb.hip.  The input must already be
// activated (gn_apply_silu): plain copy while staging.  Epilogue: the 128-channel tile leaves in 4 passes of 32
// channels through the current stage buffer (bias, residual, GroupNorm partials, full-line stores).
#include "conv_mfma.hpp"

#include <cstdlib>

namespace ire {

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int W4_THREADS = 256;
constexpr int W4_TH = 16, W4_TW = 32, W4_IH = 18, W4_IW = 34;
constexpr int W4_IN_CHUNKS = W4_IH * W4_IW * 2;                              // 1224 x 16 B (16 channels per pixel)
constexpr int W4_IN_BYTES = (W4_IN_CHUNKS + 8) * 16;                          // + dummy slot (chunk slots past the tile)
constexpr int W4_IN_ITERS = (W4_IN_CHUNKS + W4_THREADS - 1) / W4_THREADS;     // 5
constexpr int W4_NSTEPS = 9;
constexpr int W4_MT = 4;

__device__ __forceinline__ unsigned w4_pack(float a, float b) {
    f32x2_t f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ float w4_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float w4_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

__device__ __forceinline__ void w4_glds16(const void* gsrc, unsigned lds_dst_uniform) {   // see conv_rb.hip
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
template <int N> __device__ __forceinline__ float w4_ror_add(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float w4_swap16_add(float v) {
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float w4_swap32_add(float v) {
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
// sum over lanes with equal (lane % 4): the drain handles 32-channel passes => 4 chunks per pixel
__device__ __forceinline__ float w4_group_sum(float v) {
    v = w4_ror_add<4>(v);
    v = w4_ror_add<8>(v);
    v = w4_swap16_add(v);
    return w4_swap32_add(v);
}

struct W4Item { int img, ty, tx, nb, tile; };
struct W4Regs { uint4 v[W4_IN_ITERS]; unsigned ok; };

template <int NT>
struct W4Cfg {
    static constexpr int NTL = NT / 32;
    static constexpr int W_CHUNKS = W4_NSTEPS * 2 * NT;        // [tap][c8][NT] x 16 B
    static constexpr int W_BYTES = W_CHUNKS * 16;
    static constexpr int W_ITERS = (W_CHUNKS + W4_THREADS - 1) / W4_THREADS;
    static constexpr int BUF_STRIDE = W4_IN_BYTES + W_BYTES;   // [in | w]
    static constexpr int MAIN_BYTES = 2 * BUF_STRIDE;
    static constexpr int RED_BYTES = 4 * 4 * 4 * 4;            // [4 waves][4 cc][4] floats (one 32-channel pass)
    static constexpr int BIAS_BYTES = 256 * 4;
    static constexpr int LDS_BYTES = MAIN_BYTES + RED_BYTES + BIAS_BYTES;
    static constexpr int PASS_CHUNKS = W4_TH * W4_TW * 4;      // one 32-channel pass: 512 px x 4 chunks
    static constexpr int OUT_ITERS = PASS_CHUNKS / W4_THREADS; // 8
    static_assert(W4_TH * W4_TW * 32 * 2 <= BUF_STRIDE, "a 32-channel pass must fit in one stage buffer");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int NT, bool RESID, bool UPS, int DBG = 0>
__global__ __launch_bounds__(W4_THREADS) void conv_w4_kernel(ConvArgs a) {
    using C = W4Cfg<NT>;
    constexpr int NTL = C::NTL;
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int c8_fixed = tid & 1;   // 2 chunks per pixel per stage

    // ---- persistent work assignment (as conv_rb.hip) ------------------------------------------------------
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int items = tiles_per_img * a.nimg * a.nblocks;
    const int G = gridDim.x;
    const int X = G < 8 ? G : 8;
    const int xcd = blockIdx.x % X, jx = blockIdx.x / X;
    const int nwx = (G - xcd + X - 1) / X;
    const int lo = (int)((long long)items * xcd / X), hi = (int)((long long)items * (xcd + 1) / X);
    const int my_items = (lo + jx < hi) ? (hi - lo - jx + nwx - 1) / nwx : 0;
    const int nkc = a.nkc;                         // 16-channel stages per item
    const int S = my_items * nkc;
    if (S == 0) return;

    struct StageInfo { W4Item it; int kc; };
    auto decode = [&](int s) -> StageInfo {
        const int k = s / nkc;
        const int L = lo + jx + k * nwx;
        StageInfo si;
        si.it.nb = L % a.nblocks;
        const int t = L / a.nblocks;
        si.it.img = t / tiles_per_img;
        si.it.tile = t - si.it.img * tiles_per_img;
        si.it.ty = si.it.tile / a.tiles_x;
        si.it.tx = si.it.tile - si.it.ty * a.tiles_x;
        si.kc = s - k * nkc;
        return si;
    };
    StageInfo sq0 = decode(0), sq1 = decode(min(1, S - 1)), sq2 = decode(min(2, S - 1));

    const int Cin = a.cin0;
    const int cin_shift = 31 - __builtin_clz(Cin);

    // ---- input tile in LDS: two planes (k-halves c8 = 0/1) of 18x34 pixels x 16 B.  Lane (r, h) reads pixel
    // p = (4*wave + m + ky)*34 + r + kx of plane h: ONE address register + immediates, and the 16 lanes a ds_read_b128
    // services together ({0-3,12-15,20-27}...) hit 16 distinct 16-B slots.  PLANE/4 mod 32 == 16, so the 8-lane groups
    // of the ds_write_b128 (4 pixels x 2 planes) are conflict-free too.
    constexpr int PLANE = (W4_IN_CHUNKS / 2) * 16;
    static_assert((PLANE / 4) % 32 == 16, "plane offset must be half a bank row");
    const int a_base = h * PLANE + (wave * W4_MT * W4_IW + r) * 16;
    const int b_off = (h * NT + r) * 16;

    auto load_stage = [&](const StageInfo& si, W4Regs& R) {
        const W4Item& it = si.it;
        const int oy1 = it.ty * W4_TH - 1, ox1 = it.tx * W4_TW - 1;
        const char* base = reinterpret_cast<const char*>(a.in0) + (size_t)it.img * a.Hin * a.Win * Cin * 2 + si.kc * 32;
        R.ok = 0;
        int t2 = tid;
        asm volatile("" : "+v"(t2));
#pragma unroll
        for (int i = 0; i < W4_IN_ITERS; ++i) {
            const int p = (t2 + i * W4_THREADS) >> 1;
            const int py = p / W4_IW, px = p - py * W4_IW;
            const int iy = oy1 + py, ix = ox1 + px;
            const int HV = UPS ? a.Hout : a.Hin, WV = UPS ? a.Wout : a.Win;
            const int cy = min(max(iy, 0), HV - 1), cx = min(max(ix, 0), WV - 1);
            const bool ok = iy == cy && ix == cx;
            const int sy = UPS ? (cy >> 1) : cy, sx = UPS ? (cx >> 1) : cx;
            const unsigned off = ((unsigned)(sy * a.Win + sx) << (cin_shift + 1)) + (unsigned)(c8_fixed * 16);
            if constexpr (!(DBG & 8)) R.v[i] = *reinterpret_cast<const uint4*>(base + off);
            R.ok |= ok ? (1u << i) : 0u;
        }
    };
    auto store_chunk = [&](int i, const W4Regs& R, uint4* lds_in) {   // plain copy; zero padding outside the image
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int idx = t2 + i * W4_THREADS;
        const bool ok = (R.ok >> i) & 1u;
        uint4 o;
        o.x = ok ? R.v[i].x : 0u; o.y = ok ? R.v[i].y : 0u; o.z = ok ? R.v[i].z : 0u; o.w = ok ? R.v[i].w : 0u;
        const int slot = c8_fixed * (W4_IN_CHUNKS / 2) + (idx >> 1);
        lds_in[idx < W4_IN_CHUNKS ? slot : W4_IN_CHUNKS] = o;
    };
    auto wslab = [&](const StageInfo& si) -> const unsigned char* {
        return reinterpret_cast<const unsigned char*>(a.w) + ((size_t)si.it.nb * nkc + si.kc) * C::W_BYTES;
    };

    f32x16_t acc[W4_MT][NTL];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < W4_MT; ++m)
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][j][i] = 0.f;
    };

    float* red = reinterpret_cast<float*>(smem + C::MAIN_BYTES);                       // [4 waves][4 cc][4]
    const float* bias_lds = reinterpret_cast<const float*>(smem + C::MAIN_BYTES + C::RED_BYTES);

    int s = 0;
    auto stamp = [&](int k) {
#ifdef IRE_W4_STAMPS
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (a.stamps && lane == 0 && (wave == 0 || wave == 3) && blockIdx.x < 8 && s < 64)
            a.stamps[(((size_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 64 + s) * 10 + k] = t;
#else
        (void)k;
#endif
    };
    W4Regs R0, R1;
    auto compute = [&](auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        unsigned char* buf_cur = smem + PAR * C::BUF_STRIDE;
        unsigned char* buf_nxt = smem + (PAR ^ 1) * C::BUF_STRIDE;
        uint4* in_nxt = reinterpret_cast<uint4*>(buf_nxt);
        W4Regs& Rn = PAR ? R0 : R1;   // stage s+1 data (loaded one stage ago)
        W4Regs& Rf = PAR ? R1 : R0;   // receives stage s+2
        stamp(0);
#pragma unroll
        for (int i = 0; i < W4_IN_ITERS; ++i) asm volatile("" : "+v"(Rn.v[i].x), "+v"(Rn.v[i].y), "+v"(Rn.v[i].z), "+v"(Rn.v[i].w));
        load_stage(sq2, Rf);

        const unsigned char* wb = buf_cur + W4_IN_BYTES + b_off;
        const unsigned char* ib = buf_cur;
        bf16x8_t bfr[2][NTL], afr[2][W4_MT];
        auto read_frags = [&](int st, bf16x8_t (&bf)[NTL], bf16x8_t (&af)[W4_MT]) {
            const int ky = st / 3, kx = st - ky * 3;
#pragma unroll
            for (int j = 0; j < NTL; ++j)
                bf[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wb + (2 * st * NT + j * 32) * 16));
#pragma unroll
            for (int m = 0; m < W4_MT; ++m)
                af[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ib + a_base + ((m + ky) * W4_IW + kx) * 16));
        };
        read_frags(0, bfr[0], afr[0]);
#pragma unroll
        for (int st = 0; st < W4_NSTEPS; ++st) {
            if (st + 1 < W4_NSTEPS && !(DBG & 2)) read_frags(st + 1, bfr[(st + 1) & 1], afr[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < W4_MT; ++m)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    if constexpr (!(DBG & 1)) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[(DBG & 2) ? 0 : (st & 1)][j], afr[(DBG & 2) ? 0 : (st & 1)][m], acc[m][j], 0, 0, 0);  // D[cout][pixel]
            if (st < W4_IN_ITERS && !(DBG & 8)) store_chunk(st, Rn, in_nxt);        // stage s+1 input -> other buffer
            if (st == 0 && !(DBG & 32)) {                                           // weight slab of stage s+1 by LDS-DMA
                const unsigned char* ws = wslab(sq1);
                const int wave_u = __builtin_amdgcn_readfirstlane(wave);
                const unsigned w_nxt_lds = smem_lds + (unsigned)(buf_nxt - smem) + W4_IN_BYTES;
#pragma unroll
                for (int i = 0; i < C::W_ITERS; ++i) {
                    static_assert(C::W_CHUNKS % W4_THREADS == 0, "branch-free DMA issue: a CFG merge here makes hipcc drain vmcnt");
                    const int cbase = i * W4_THREADS + wave_u * 64;
                    w4_glds16(ws + (size_t)(cbase + lane) * 16, w_nxt_lds + cbase * 16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // retire the s+2 prefetch and the weight DMA before the epilogue / barrier (in-order VMEM queue: see conv_rb.hip)
#pragma unroll
        for (int i = 0; i < W4_IN_ITERS; ++i) asm volatile("" : "+v"(Rf.v[i].x), "+v"(Rf.v[i].y), "+v"(Rf.v[i].z), "+v"(Rf.v[i].w));
        stamp(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(2);
    };
    // ---- epilogue: NTL passes of 32 channels through buf[cur]; only READS the accumulators (the item loop below
    // zeroes them unconditionally: with all 256 AGPRs holding acc, a conditional redefinition is a 256-register PHI
    // that the allocator can only resolve by spilling accumulators to scratch).
    auto epilogue = [&](auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        unsigned char* buf_cur = smem + PAR * C::BUF_STRIDE;
        {
            const W4Item it = sq0.it;
            const int oy0 = it.ty * W4_TH, ox0 = it.tx * W4_TW, cout0 = it.nb * NT;
            char* obase = reinterpret_cast<char*>(a.out) + (size_t)it.img * a.Hout * a.Wout * a.cout * 2;
            const char* rbase = reinterpret_cast<const char*>(a.resid) + (size_t)it.img * a.Hout * a.Wout * a.cout * 2;
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                // register-only instructions (v_accvgpr_read of the NEXT passes) may legally move across __syncthreads():
                // pin every pass, or 256 accumulator copies are live at once and the allocator spills the hot loop's state
                __builtin_amdgcn_sched_barrier(0);
                int te = tid;
                asm volatile("" : "+v"(te));
                const int cc = te & 3;
                uint4 rv[C::OUT_ITERS];
                if constexpr (RESID) {
#pragma unroll
                    for (int k = 0; k < C::OUT_ITERS; ++k) {
                        const int pix = (te + k * W4_THREADS) >> 2;
                        const int oy = min(oy0 + (pix >> 5), a.Hout - 1), ox = min(ox0 + (pix & 31), a.Wout - 1);
                        const unsigned off = ((unsigned)((oy * a.Wout + ox) * a.cout + cout0 + j * 32) << 1) + (unsigned)(cc * 16);
                        rv[k] = *reinterpret_cast<const uint4*>(rbase + off);
                    }
                }
                __syncthreads();          // previous pass drained / every wave done reading buf[cur] fragments
                // accumulator i of lane (r = pixel column, h) is cout j*32 + 8*(i>>2) + 4h + (i&3)
                int r_e = r, h_e = h;
                asm volatile("" : "+v"(r_e), "+v"(h_e));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_sched_barrier(0);      // at most 16 accumulator copies live at a time
                    const float4 bv = *reinterpret_cast<const float4*>(bias_lds + cout0 + j * 32 + 8 * q + 4 * h_e);
#pragma unroll
                    for (int m = 0; m < W4_MT; ++m) {
                        const int pix = (wave * W4_MT + m) * W4_TW + r_e;
                        uint2 v;
                        v.x = w4_pack(acc[m][j][4 * q + 0] + bv.x, acc[m][j][4 * q + 1] + bv.y);
                        v.y = w4_pack(acc[m][j][4 * q + 2] + bv.z, acc[m][j][4 * q + 3] + bv.w);
                        *reinterpret_cast<uint2*>(buf_cur + (pix * 4 + (q ^ (pix & 3))) * 16 + h_e * 8) = v;
                    }
                }
                __syncthreads();
                float sA = 0.f, qA = 0.f, sB = 0.f, qB = 0.f;
#pragma unroll
                for (int k = 0; k < C::OUT_ITERS; ++k) {
                    const int pix = (te + k * W4_THREADS) >> 2;
                    const int oy = oy0 + (pix >> 5), ox = ox0 + (pix & 31);
                    const bool inb = oy < a.Hout && ox < a.Wout;
                    const uint4 o = reinterpret_cast<const uint4*>(buf_cur)[pix * 4 + (cc ^ (pix & 3))];
                    const unsigned off = ((unsigned)((oy * a.Wout + ox) * a.cout + cout0 + j * 32) << 1) + (unsigned)(cc * 16);
                    unsigned w[4] = {o.x, o.y, o.z, o.w};
                    if constexpr (RESID) {
                        const unsigned rw[4] = {rv[k].x, rv[k].y, rv[k].z, rv[k].w};
#pragma unroll
                        for (int d = 0; d < 4; ++d)
                            w[d] = w4_pack(w4_lo(w[d]) + w4_lo(rw[d]), w4_hi(w[d]) + w4_hi(rw[d]));
                    }
                    {
                        const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
                        float ts0 = 0.f, tq0 = 0.f, ts1 = 0.f, tq1 = 0.f;
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const bf16x2_t wv = __builtin_bit_cast(bf16x2_t, w[d]);
                            if (d < 2) { ts0 = __builtin_amdgcn_fdot2_f32_bf16(wv, ones, ts0, false); tq0 = __builtin_amdgcn_fdot2_f32_bf16(wv, wv, tq0, false); }
                            else { ts1 = __builtin_amdgcn_fdot2_f32_bf16(wv, ones, ts1, false); tq1 = __builtin_amdgcn_fdot2_f32_bf16(wv, wv, tq1, false); }
                        }
                        sA += inb ? ts0 : 0.f; qA += inb ? tq0 : 0.f; sB += inb ? ts1 : 0.f; qB += inb ? tq1 : 0.f;
                    }
                    if (inb) *reinterpret_cast<uint4*>(obase + off) = make_uint4(w[0], w[1], w[2], w[3]);
                }
                if (a.stats) {
                    // GroupNorm partials of this 32-channel pass: group size G >= 16 here (C >= 128) => the pass covers
                    // 32/G groups, each = 2 or 4 whole 16-B chunks; reduce per chunk, combine on a few lanes.
                    sA = w4_group_sum(sA + sB); qA = w4_group_sum(qA + qB);     // per (wave, cc): sum over its pixels
                    if (lane < 4) { red[(wave * 4 + lane) * 4 + 0] = sA; red[(wave * 4 + lane) * 4 + 1] = qA; }
                    __syncthreads();
                    const int G = a.group_size, cpg = G >> 3, ngl = 32 / G;       // chunks per group (2 or 4), groups in the pass
                    if (tid < ngl) {
                        float s = 0.f, q = 0.f;
#pragma unroll
                        for (int w = 0; w < 4; ++w)
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (k < cpg) { s += red[(w * 4 + tid * cpg + k) * 4 + 0]; q += red[(w * 4 + tid * cpg + k) * 4 + 1]; }
                        const int gg = (cout0 + j * 32) / G + tid;
                        float* st = a.stats + (((size_t)it.img * tiles_per_img + it.tile) * 8 + gg) * 2;
                        st[0] = s; st[1] = q;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto finish = [&](int s) {
        stamp(4);
        __syncthreads();          // stage barrier: buf[nxt] complete, buf[cur] free
        stamp(5);
        sq0 = sq1; sq1 = sq2; sq2 = decode(min(s + 3, S - 1));
    };

    // ---- prologue ---------------------------------------------------------------------------------------------
    {
        float* bl = reinterpret_cast<float*>(smem + C::MAIN_BYTES + C::RED_BYTES);
        if (tid < a.cout && tid < 256) bl[tid] = a.bias[tid];
        load_stage(sq0, R0);
        const uint4* ws = reinterpret_cast<const uint4*>(wslab(sq0));
        uint4* wd = reinterpret_cast<uint4*>(smem + W4_IN_BYTES);
        for (int i = tid; i < C::W_CHUNKS; i += W4_THREADS) wd[i] = ws[i];
        uint4* in0 = reinterpret_cast<uint4*>(smem);
#pragma unroll
        for (int i = 0; i < W4_IN_ITERS; ++i) store_chunk(i, R0, in0);
        load_stage(sq1, R1);
    }
    __syncthreads();
    // nkc is even (Cin/16 with Cin >= 128), so an item starts on an even stage and ends on an odd one
    for (int k = 0; k < my_items; ++k) {
        zero_acc();
        // steady-state stage pairs, then the item's last pair peeled with the epilogue: a conditional epilogue inside the
        // loop makes the allocator split live ranges of in-flight prefetch registers mid-stage (vmcnt(0) + v_mov)
        for (int kc = 0; kc + 2 < nkc; kc += 2) {
            compute(std::integral_constant<int, 0>{});
            finish(s); ++s;
            compute(std::integral_constant<int, 1>{});
            finish(s); ++s;
        }
        compute(std::integral_constant<int, 0>{});
        finish(s); ++s;
        compute(std::integral_constant<int, 1>{});
        stamp(3);
        epilogue(std::integral_constant<int, 1>{});
        finish(s); ++s;
    }
}

template <int NT, bool RESID, bool UPS, int DBG = 0>
void launch_w4(const ConvArgs& a, hipStream_t stream) {
    const int items = a.tiles_x * a.tiles_y * a.nimg * a.nblocks;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = items < cus ? items : cus;
    hipLaunchKernelGGL((conv_w4_kernel<NT, RESID, UPS, DBG>), dim3(grid), dim3(W4_THREADS), 0, stream, a);
    IRE_HIP(hipGetLastError());
}

}  // namespace

// a.nkc = Cin/16 stages, a.nblocks = cout/128, a.w = slabs [nblock][kc16][tap][c8][128][8], tiles of 16x32.
void conv_w4_launch(bool resid, const ConvArgs& a, hipStream_t stream) {
#ifdef IRE_W4_STAMPS
    static const int dbg = std::getenv("IRE_W4_DBG") ? std::atoi(std::getenv("IRE_W4_DBG")) : 0;
    switch (dbg) {
        case 1: return launch_w4<128, false, false, 1>(a, stream);
        case 2: return launch_w4<128, false, false, 2>(a, stream);
        case 8: return launch_w4<128, false, false, 8>(a, stream);
        case 32: return launch_w4<128, false, false, 32>(a, stream);
        case 40: return launch_w4<128, false, false, 40>(a, stream);
        case 42: return launch_w4<128, false, false, 42>(a, stream);
        default: break;
    }
#endif
    if (resid) launch_w4<128, true, false>(a, stream); else launch_w4<128, false, false>(a, stream);
}

}  // namespace ire
