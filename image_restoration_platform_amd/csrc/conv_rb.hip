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
#include "persist.hpp"
#include "gn_fold.hpp"

#include <cstdlib>

namespace ire {

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int RB_TH = kRbTileH, RB_TW = 32;          // 16 (8 waves x 2 rows); IRE_RB_TH=8: 4 waves x 2 rows
constexpr int RB_WAVES = RB_TH / 2;
constexpr int RB_THREADS = RB_WAVES * 64;
constexpr int RB_IH = RB_TH + 2, RB_IW = RB_TW + 2;
constexpr int RB_IN_CHUNKS = RB_IH * RB_IW * 4;                                // 2448 x 16 B
constexpr int RB_IN_ITERS = (RB_IN_CHUNKS + RB_THREADS - 1) / RB_THREADS;       // 5
constexpr int RB_IN_BYTES = RB_IN_ITERS * RB_THREADS * 16;                     // 40960: padded so all 5x512 chunk slots exist
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

// LDS-DMA, 16 B per lane: LDS destination = wave-uniform byte address (through M0) + lane*16; the source is
// per lane.  Not visible to the compiler's s_waitcnt bookkeeping: the caller waits (vmcnt) and barriers.
__device__ __forceinline__ void rb_glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}

// Sum over the lanes that share (lane % NCC), NCC in {4, 8}: row rotations (DPP) inside the 16-lane rows, then
// the two permlane swaps across rows / wave halves.  Every lane ends with the total; fixed order => deterministic.
// The swaps are inline asm on purpose: with the builtins hipcc (ROCm 7.2) folds the two results into one register
// when both inputs are the same value (verified in the ISA); the two v_nop are the VALU-write -> permlane wait states.
template <int N> __device__ __forceinline__ float rb_ror_add(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float rb_swap16_add(float v) {
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float rb_swap32_add(float v) {
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
template <int NCC> __device__ __forceinline__ float rb_group_sum(float v) {
    if constexpr (NCC == 4) v = rb_ror_add<4>(v);
    v = rb_ror_add<8>(v);
    v = rb_swap16_add(v);
    return rb_swap32_add(v);
}

using RbItem = PersistItem;   // img, ty, tx, nb, tile = ty*tiles_x + tx

struct RbRegs {          // one prefetched input stage
    uint4 v[RB_IN_ITERS];
    unsigned ok;         // bit it: chunk it is inside the image
};

template <int NT, bool RESID, bool WRES, bool FUSED_ACT>
struct RbCfg {
    static constexpr int NTL = NT / 32;
    static constexpr int W_CHUNKS = RB_NSTEPS * 2 * NT;   // 16-B chunks per (n-block, k-chunk) slab
    static constexpr int W_BYTES = W_CHUNKS * 16;
    static constexpr int W_ITERS = (W_CHUNKS + RB_THREADS - 1) / RB_THREADS;
    // streaming: [in0 | w0][in1 | w1]; resident weights: [in0][in1][w]
    static constexpr int BUF_STRIDE = WRES ? RB_IN_BYTES : RB_IN_BYTES + W_BYTES;
    static constexpr int W_OFF0 = WRES ? 2 * RB_IN_BYTES : RB_IN_BYTES;
    static constexpr int MAIN_BYTES = WRES ? 2 * RB_IN_BYTES + W_BYTES : 2 * (RB_IN_BYTES + W_BYTES);
    static constexpr int RED_HALF = RB_WAVES * (NT / 8) * 4 * 4;     // [waves][NT/8 chunks][sA,qA,sB,qB]
    static constexpr int RED_BYTES = 2 * RED_HALF;             // two copies, alternating per item (DIRECT: no barrier before the write)
    static constexpr int COEF_BYTES = 2 * 256;              // two stages x 32 channels x (A,B) floats
    static constexpr int BIAS_BYTES = 256 * 4;              // the layer's whole bias vector (cout <= 256)
    static constexpr int LDS_BYTES = MAIN_BYTES + RED_BYTES + COEF_BYTES + BIAS_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// DBG (timing ablations only, results are wrong): 1 = no MFMA, 2 = no fragment reads + no MFMA,
// 4 = no epilogue global traffic, 8 = no input global loads.
// HEAD (NT = 32, weights resident, fused GroupNorm+SiLU): RestoreNet's last layer, 32 -> 3 channels (slab padded to 32 rows):
// the epilogue is out = clamp(round(input + y)) as u8 RGB -- couts 0..2 are accumulators 0..2 of the lanes with h = 0 -- no
// bf16 tensor, no statistics.  Same pipeline as every other C = 32 convolution instead of the single-buffered v1 kernel.
template <int NT, bool RESID, bool WRES, bool FUSED_ACT, int DBG = 0, bool UPS = false, bool HEAD = false>
__global__ __launch_bounds__(RB_THREADS) void conv_rb_kernel(ConvArgs a) {
    static_assert(!HEAD || (NT == 32 && WRES && FUSED_ACT && !RESID && !UPS), "HEAD: the C = 32 resident-weight fused variant");
    using C = RbCfg<NT, RESID, WRES, FUSED_ACT>;
    constexpr int NTL = C::NTL;
    constexpr int NCC = NT / 8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];

    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;   // LDS byte address
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int c8_fixed = tid & 3;  // this thread always stages the same 8-channel slice of a pixel
    // diagnostic stamps (DBG & 16): lane 0 of waves 0 and 4 of the first 8 workgroups, 6 stamps x 64 stages
    auto stamp = [&](int s, int k) {
        if constexpr (DBG & 16) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (a.stamps && lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 8 && s < 64)
                a.stamps[(((size_t)blockIdx.x * 2 + (wave >> 2)) * 64 + s) * 10 + k] = t;
        }
    };

    // ---- persistent work assignment (persist.hpp: XCD-aware, advanced by additions instead of per-stage divisions) ----
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    PersistCursor cursor(a.tiles_x, a.tiles_y, a.nimg, a.nblocks, a.nkc);
    const int nkc = a.nkc;
    const int S = cursor.S;                                      // stages this workgroup runs
    if (S == 0) return;
    if (a.gn_stats) gn_fold(a, smem, cursor.first_img, cursor.last_img);     // GroupNorm finalize of the input tensor, folded in (gn_fold.hpp)
    // sq0 = stage s (MFMA + epilogue), sq1 = stage s+1 (weights), sq2 = stage s+2 (prefetch, coefficients)
    using StageInfo = PersistStage;
    StageInfo sq0 = cursor.cur, sq1 = cursor.next(), sq2 = cursor.next();

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
    const int cin_shift = 31 - __builtin_clz(Cin);                 // Cin is a power of two
    auto load_stage = [&](const StageInfo& si, RbRegs& R) {
        const RbItem& it = si.it;
        const int kc = si.kc;
        const int oy1 = it.ty * RB_TH - 1, ox1 = it.tx * RB_TW - 1;
        // wave-uniform image base (SGPR pair) + 32-bit per-lane byte offset
        const char* base = reinterpret_cast<const char*>(src) + (size_t)it.img * a.in_rows * a.Win * Cin * 2 + kc * 64;
        // range-checked buffer loads: a halo pixel outside the image gets an offset past num_records and reads as zero --
        // no coordinate clamping (4 min/max per chunk); the `ok` bit still gates the value AFTER the activation
        const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, a.in_rows * a.Win * Cin * 2 - kc * 64, 0x00020000);
        R.ok = 0;
        int t2 = tid;
        asm volatile("" : "+v"(t2));   // recompute the halo coordinates per stage: cheaper than 5 live registers
#pragma unroll
        for (int i = 0; i < RB_IN_ITERS; ++i) {
            const int p = (t2 + i * RB_THREADS) >> 2;
            const int py = p / RB_IW, px = p - py * RB_IW;
            const int iy = oy1 + py, ix = ox1 + px;
            // UPS: the conv runs on the nearest-x2 upsampled grid (extent = output extent); source pixel = coord >> 1
            const int WV = UPS ? a.Wout : a.Win;
            const bool ok = (unsigned)(iy - a.iy_lo) < (unsigned)a.iy_span && (unsigned)ix < (unsigned)WV;   // slots past the tile (py >= 18) land in the LDS padding
            const int sy = (UPS ? (iy >> 1) : iy) + a.in_row_off, sx = UPS ? (ix >> 1) : ix;
            const unsigned off = ok ? ((unsigned)(sy * a.Win + sx) << (cin_shift + 1)) + (unsigned)(c8_fixed * 16) : 0xffffffffu;
            if constexpr (DBG & 8) R.v[i] = make_uint4(off, 0x3f803f80u, i, 0x3f803f80u);
            else {
                const u32x4_t lv = __builtin_amdgcn_raw_buffer_load_b128(irsrc, off, 0, 0);
                R.v[i] = make_uint4(lv.x, lv.y, lv.z, lv.w);
            }
            R.ok |= ok ? (1u << i) : 0u;
        }
    };
    // GroupNorm+FiLM coefficients travel global -> register (16 lanes, issued with the input prefetch so the
    // same forced wait covers them) -> LDS slot -> every thread's 16 floats.  Keeping them off the VMEM queue
    // at the stage top matters: vmcnt completes in order, so a late coefficient load would drag the previous
    // item's output stores into the wait.
    float* coef_lds = reinterpret_cast<float*>(smem + C::MAIN_BYTES + C::RED_BYTES);   // [2][32 ch][2]
    auto fetch_coeffs = [&](const StageInfo& si) -> float4 {     // lanes 0..15: 16 B each of the stage's 256 B
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (FUSED_ACT) {
            const RbItem& it = si.it;
            const int kc = si.kc;
            const float4* ab = reinterpret_cast<const float4*>(a.ab + (size_t)it.img * Cin + kc * 32);
            v = ab[tid & 15];
        }
        return v;
    };
    // LDS slot = A[32 channels] then B[32 channels] (gn_finalize stores (A, B) pairs): a thread's 8 A's and 8 B's then arrive as
    // two register quads each, so the (A_2d, A_2d+1) / (B_2d, B_2d+1) operands of the packed fma are adjacent registers as loaded
    // (from (A, B, A, B) quads hipcc needed three v_mov per word to pair them up: 60 per stage)
    auto put_coeffs = [&](int slot, const float4& v) {      // lane l < 16 holds (A, B) of channels 2l, 2l+1
        if constexpr (FUSED_ACT) {
            if (tid < 16) {
                reinterpret_cast<float2*>(coef_lds + slot * 64)[tid] = make_float2(v.x, v.z);
                reinterpret_cast<float2*>(coef_lds + slot * 64 + 32)[tid] = make_float2(v.y, v.w);
            }
        }
    };
    auto load_coeffs = [&](int slot, float (&cA)[8], float (&cB)[8]) {
        if constexpr (!FUSED_ACT) return;
        const float4* pa = reinterpret_cast<const float4*>(coef_lds + slot * 64 + c8_fixed * 8);
        const float4* pb = reinterpret_cast<const float4*>(coef_lds + slot * 64 + 32 + c8_fixed * 8);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float4 va = pa[e], vb = pb[e];
            cA[4 * e] = va.x; cA[4 * e + 1] = va.y; cA[4 * e + 2] = va.z; cA[4 * e + 3] = va.w;
            cB[4 * e] = vb.x; cB[4 * e + 1] = vb.y; cB[4 * e + 2] = vb.z; cB[4 * e + 3] = vb.w;
        }
    };
    // The s+1 transform is cut into 20 word-sized pieces (5 chunks x 4 words of 2 channels) so that one
    // piece can follow every MFMA k-step: straight-line code, interleaved by source order.
    auto word_of = [&](const uint4& v, int d) -> unsigned { return d == 0 ? v.x : d == 1 ? v.y : d == 2 ? v.z : v.w; };
    auto transform_word = [&](unsigned w, int d, const float (&cA)[8], const float (&cB)[8]) -> unsigned {
        if constexpr (FUSED_ACT) {
            // two channels at a time in packed f32 (v_pk_fma/mul/add_f32): y = x*A + B ; silu(y) = y / (1 + 2^(-y log2 e))
            const f32x2_t x = {rb_lo(w), rb_hi(w)};
            const f32x2_t A = {cA[2 * d], cA[2 * d + 1]}, B = {cB[2 * d], cB[2 * d + 1]};
            const f32x2_t y = __builtin_elementwise_fma(x, A, B);
            const f32x2_t t = y * (-1.4426950408889634f);
            f32x2_t e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
            e = e + 1.0f;
            const f32x2_t rinv = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
            // (an inline-asm v_pk_mul_f32 here reads the v_rcp_f32 results before they are architecturally visible: hipcc pads
            // transcendental -> VALU dependencies for its own instructions only)
            const f32x2_t sv = y * rinv;
            return rb_pack(sv.x, sv.y);
        } else {
            return w;                                  // the input carries no GroupNorm (the `up` / `down` inputs of this family)
        }
    };
    auto store_words = [&](int i, const RbRegs& R, const unsigned (&tw)[4], uint4* lds_in) {
        int t2 = tid;
        asm volatile("" : "+v"(t2));   // recompute the LDS address per use instead of keeping 5 of them live
        const int idx = t2 + i * RB_THREADS;
        const bool ok = (R.ok >> i) & 1u;              // zero padding applies AFTER the activation
        uint4 o;
        o.x = ok ? tw[0] : 0u; o.y = ok ? tw[1] : 0u; o.z = ok ? tw[2] : 0u; o.w = ok ? tw[3] : 0u;
        const int p = idx >> 2;
        lds_in[p * 4 + (c8_fixed ^ ((p >> 2) & 3))] = o;
    };
    auto store_chunk = [&](int i, const RbRegs& R, const float (&cA)[8], const float (&cB)[8], uint4* lds_in) {
        unsigned tw[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) tw[d] = transform_word(word_of(R.v[i], d), d, cA, cB);
        store_words(i, R, tw, lds_in);
    };
    auto wslab = [&](const StageInfo& si) -> const uint4* {
        const RbItem& it = si.it;
        const int kc = si.kc;
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
    int st_img = -1, st_tile = 0, st_nb = 0, st_par = 0, red_par = 0;
    float* red_base = reinterpret_cast<float*>(smem + C::MAIN_BYTES);   // 2 x [8 waves][NCC][4]
    float* red = red_base;
    auto flush_stats = [&]() {
        if (st_img < 0) return;
        if (UPS || HEAD || a.stats == nullptr) { st_img = -1; return; }
        const float* red = red_base + st_par * (C::RED_HALF / 4);
        if (__builtin_amdgcn_readfirstlane(wave) != 0) { st_img = -1; return; }   // only wave 0 has work: spare the other waves the scalar ladder
        const int Gs = a.group_size, ngl = NT / Gs;
        if (tid < ngl) {
            float s = 0.f, q = 0.f;
            const int cpg = Gs >> 3;                       // 16-B chunks per group (0 when Gs == 4)
#pragma unroll
            for (int w = 0; w < RB_WAVES; ++w) {
                if (NT == 32) {                            // cout = 32 => groups of 4: a chunk holds two groups (cc, half)
                    const float* d = red + (w * NCC + (tid >> 1)) * 4 + 2 * (tid & 1);
                    s += d[0]; q += d[1];
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < cpg) {
                            const float* d = red + (w * NCC + tid * cpg + k) * 4;
                            s += d[0] + d[2]; q += d[1] + d[3];
                        }
                }
            }
            const int gg = (st_nb * NT) / Gs + tid;
            float* st = a.stats + (((size_t)st_img * tiles_per_img + st_tile) * 8 + gg) * 2;
            st[0] = s; st[1] = q;
        }
        st_img = -1;
    };

    // ---- epilogue building blocks ------------------------------------------------------------------------
    // ---- epilogue, straight from the accumulators -----------------------------------------------------------------
    // MFMA row rho of lane (r = pixel column, h) is accumulator i with rho = 8*(i>>2) + 4h + (i&3).  The weight slab is stored
    // with its rows permuted (bits 2 and 3 of rho swapped, engine.cpp::make_conv), which makes accumulator i the cout
    // j*32 + 16*(i>>3) + 8h + (i&7): for 16-cout group g = (j, p) a lane owns ONE whole 16-B chunk cc = j*4 + 2p + h of its
    // pixel in accumulators 8p..8p+7 -- packed and stored with no lane exchange (the natural row order needed one
    // v_permlane32_swap + two wait states per register).  GroupNorm partials are reduced over the 32 lanes of a half
    // (same chunk) and land in red[wave][cc].
    constexpr int NG = NTL * 2;
#ifndef IRE_RB_NGP64
#define IRE_RB_NGP64 4
#endif
    constexpr int NGP = (NT == 64 && FUSED_ACT) ? IRE_RB_NGP64 : NG;      // residual groups requested a stage ahead
    uint4 erv[NG][2];
    unsigned eoffs[2];
    bool einb[2];
    // always_inline: a lambda hipcc declines to inline keeps every by-reference capture (acc, the kernel arguments) in scratch
    auto epi_prefetch = [&](const RbItem& it) __attribute__((always_inline)) {
        int r_e = r, h_e = h, w_e = wave;
        asm volatile("" : "+v"(r_e), "+v"(h_e), "+v"(w_e));
        const int oyb = it.ty * RB_TH + w_e * 2, ox = it.tx * RB_TW + r_e;
        const bool colok = ox < a.Wout;
        const int oxc = min(ox, a.Wout - 1);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int oy = oyb + m;
            einb[m] = colok && oy < a.Hout;
            eoffs[m] = ((unsigned)((min(oy, a.Hout - 1) * a.Wout + oxc) * a.cout + it.nb * NT) << 1) + (unsigned)(h_e * 16);
        }
        if constexpr (RESID && !(DBG & 4)) {
            const char* rbase = reinterpret_cast<const char*>(a.resid) + (size_t)it.img * a.Hout * a.Wout * a.cout * 2;
#pragma unroll
            for (int g = 0; g < NGP; ++g)
#pragma unroll
                for (int m = 0; m < 2; ++m) erv[g][m] = *reinterpret_cast<const uint4*>(rbase + eoffs[m] + (unsigned)(g * 32));
        }
    };
    // the groups not requested a stage ahead (register budget of the fused NT = 64 variant): requested when the epilogue starts
    auto epi_fetch_rest = [&](const RbItem& it) __attribute__((always_inline)) {
        if constexpr (RESID && !(DBG & 4) && NGP < NG) {
            const char* rbase = reinterpret_cast<const char*>(a.resid) + (size_t)it.img * a.Hout * a.Wout * a.cout * 2;
#pragma unroll
            for (int g = NGP; g < NG; ++g)
#pragma unroll
                for (int m = 0; m < 2; ++m) erv[g][m] = *reinterpret_cast<const uint4*>(rbase + eoffs[m] + (unsigned)(g * 32));
        }
    };
    auto direct_epilogue = [&](const RbItem& it) __attribute__((always_inline)) {
        int h_e = h;
        asm volatile("" : "+v"(h_e));
        if constexpr (HEAD) {
            int r_e = r, w_e = wave;
            asm volatile("" : "+v"(r_e), "+v"(w_e));
            const int ox = it.tx * RB_TW + r_e;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int oy = it.ty * RB_TH + w_e * 2 + m;
                if (h_e == 0 && oy < a.Hout && ox < a.Wout) {
                    const size_t gpx = (((size_t)it.img * a.Hout + oy) * a.Wout + ox) * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float vv = (float)a.u8_in[gpx + c] + acc[m][0][c];          // the bias is already in the accumulator
                        vv = fminf(fmaxf(vv, 0.f), 255.f);
                        a.u8_out[gpx + c] = (unsigned char)(int)floorf(vv + 0.5f);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) asm volatile("" : "=v"(acc[m][0]));
            return;
        }
        const int cout0 = it.nb * NT;
        epi_fetch_rest(it);
        char* obase = reinterpret_cast<char*>(a.out) + (size_t)it.img * a.Hout * a.Wout * a.cout * 2;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, a.Hout * a.Wout * a.cout * 2, 0x00020000);
        (void)obase;
        const float* bias_lds = reinterpret_cast<const float*>(smem + C::MAIN_BYTES + C::RED_BYTES + C::COEF_BYTES);
        const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
        float* redw = red_base + red_par * (C::RED_HALF / 4);
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int g = j * 2 + pp;
            // permuted slab rows (engine.cpp::make_conv): accumulators 8pp .. 8pp+7 of lane-half h are the 8 CONTIGUOUS couts
            // j*32 + 16pp + 8h + (0..7) = 16-B chunk cc = j*4 + 2pp + h of the pixel: pack and store, no lane exchange
            float sA = 0.f, qA = 0.f, sB = 0.f, qB = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f32x16_t& c = acc[m][j];
                unsigned x0, x1, y0, y1;
                // the bias is already in the accumulators (they start at it)
                x0 = rb_pack(c[8 * pp + 0], c[8 * pp + 1]); x1 = rb_pack(c[8 * pp + 2], c[8 * pp + 3]);
                y0 = rb_pack(c[8 * pp + 4], c[8 * pp + 5]); y1 = rb_pack(c[8 * pp + 6], c[8 * pp + 7]);
                unsigned w[4] = {x0, x1, y0, y1};
                if constexpr (RESID && !(DBG & 4)) {
                    const uint4 rr = erv[g][m];
                    const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                    for (int d = 0; d < 4; ++d) w[d] = rb_pack(rb_lo(w[d]) + rb_lo(rw[d]), rb_hi(w[d]) + rb_hi(rw[d]));
                }
                float ts0 = 0.f, tq0 = 0.f, ts1 = 0.f, tq1 = 0.f;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const bf16x2_t wv = __builtin_bit_cast(bf16x2_t, w[d]);
                    if (d < 2) { ts0 = __builtin_amdgcn_fdot2_f32_bf16(wv, ones, ts0, false); tq0 = __builtin_amdgcn_fdot2_f32_bf16(wv, wv, tq0, false); }
                    else { ts1 = __builtin_amdgcn_fdot2_f32_bf16(wv, ones, ts1, false); tq1 = __builtin_amdgcn_fdot2_f32_bf16(wv, wv, tq1, false); }
                }
                sA += einb[m] ? ts0 : 0.f; qA += einb[m] ? tq0 : 0.f; sB += einb[m] ? ts1 : 0.f; qB += einb[m] ? tq1 : 0.f;
                if constexpr (DBG & 4) { if (einb[m] && w[0] == 0x12345678u) obase[eoffs[m]] = 1; }
                else {
                    // range-checked buffer store: lanes outside the image get an offset past num_records and the hardware
                    // drops them -- no exec-mask branch per store
                    const u32x4_t wv4 = {w[0], w[1], w[2], w[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(wv4, orsrc, einb[m] ? eoffs[m] + (unsigned)(g * 32) : 0xffffffffu, 0, 0);
                }
            }
            if (!UPS) {        // ResBlock convs always feed a GroupNorm (a.stats != nullptr), `up` convs never do: no run-time branch
                                      // sum over the 32 lanes of each half (= one chunk each), not across halves
                // the four chains advance one step at a time, side by side: each DPP then reads a value written three
                // instructions earlier and needs no s_nop padding (a chain reduced on its own gets one per step)
                if constexpr (NT >= 64) {      // cout >= 64 => groups of >= 8 channels: a chunk never splits into two groups
                    sA += sB; qA += qB; sB = 0.f; qB = 0.f;
                    sA = rb_ror_add<1>(sA); qA = rb_ror_add<1>(qA);
                    sA = rb_ror_add<2>(sA); qA = rb_ror_add<2>(qA);
                    sA = rb_ror_add<4>(sA); qA = rb_ror_add<4>(qA);
                    sA = rb_ror_add<8>(sA); qA = rb_ror_add<8>(qA);
                    sA = rb_swap16_add(sA); qA = rb_swap16_add(qA);
                } else {
                    sA = rb_ror_add<1>(sA); qA = rb_ror_add<1>(qA); sB = rb_ror_add<1>(sB); qB = rb_ror_add<1>(qB);
                    sA = rb_ror_add<2>(sA); qA = rb_ror_add<2>(qA); sB = rb_ror_add<2>(sB); qB = rb_ror_add<2>(qB);
                    sA = rb_ror_add<4>(sA); qA = rb_ror_add<4>(qA); sB = rb_ror_add<4>(sB); qB = rb_ror_add<4>(qB);
                    sA = rb_ror_add<8>(sA); qA = rb_ror_add<8>(qA); sB = rb_ror_add<8>(sB); qB = rb_ror_add<8>(qB);
                    sA = rb_swap16_add(sA); qA = rb_swap16_add(qA); sB = rb_swap16_add(sB); qB = rb_swap16_add(qB);
                }
                if ((lane & 31) == 0) {
                    float* d = redw + (wave * NCC + j * 4 + 2 * pp + h_e) * 4;
                    d[0] = sA; d[1] = qA; d[2] = sB; d[3] = qB;
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < NTL; ++j) asm volatile("" : "=v"(acc[m][j]));     // dead until the next item
        st_img = it.img; st_tile = it.tile; st_nb = it.nb; st_par = red_par; red_par ^= 1;
    };
    // WRES (C = 32, one n-tile, nkc == 1): every item's accumulators START at the bias -- 16 registers for the whole
    // kernel instead of 32 v_add_f32 per item in the epilogue.  Accumulator i of lane (r, h) is cout 16*(i>>3) + 8h + (i&7).
    f32x16_t bias_acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

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
        // no branches in the stage body: the last stages redo harmless work on clamped stage indices

        // (1) coefficients of stage s+1 first, then (3) the input prefetch of stage s+2, so that waiting
        // for (1) never waits for (3) (VMEM returns in issue order)
        stamp(s, 0);
        // Rn was retired at the end of the previous stage (or of the prologue); say so on every path into the
        // stage, so no wait on it is placed after the weight DMA below (where it would be a vmcnt(0)).
#pragma unroll
        for (int i = 0; i < RB_IN_ITERS; ++i) asm volatile("" : "+v"(Rn.v[i].x), "+v"(Rn.v[i].y), "+v"(Rn.v[i].z), "+v"(Rn.v[i].w));
        float cA[8], cB[8];
        load_coeffs(s & 1, cA, cB);                 // coefficients of stage s+1's data (slot written last stage)
        float4 cnext = fetch_coeffs(sq2);           // for the data of stage s+2, transformed during stage s+1
        load_stage(sq2, Rf);
        if constexpr (HEAD) {}                                          // the head's epilogue reads 3 input bytes per pixel itself
        else if constexpr (WRES) epi_prefetch(sq0.it);                  // nkc == 1: every stage ends an item
        else if constexpr (PAR == 1) { if (sq0.kc == nkc - 1) epi_prefetch(sq0.it); }   // nkc is even: items end on odd stages

        if constexpr (!WRES) {           // WRES (nkc == 1): step 0 accumulates onto the bias registers
            if (sq0.kc == 0) {           // new item (not in the epilogue: 64 dead registers there)
                {
                    // accumulators start at the bias (permuted rows: accumulator i of lane-half h is cout j*32 + 16(i>>3) + 8h + (i&7)):
                    // 4*NTL LDS reads per item replace 32*NTL epilogue adds; the moves take the place of the zeroing
                    const float* bl = reinterpret_cast<const float*>(smem + C::MAIN_BYTES + C::RED_BYTES + C::COEF_BYTES) + sq0.it.nb * NT + 8 * h;
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 bv = *reinterpret_cast<const float4*>(bl + j * 32 + 16 * (q >> 1) + 4 * (q & 1));
#pragma unroll
                            for (int m = 0; m < 2; ++m) { acc[m][j][4 * q + 0] = bv.x; acc[m][j][4 * q + 1] = bv.y; acc[m][j][4 * q + 2] = bv.z; acc[m][j][4 * q + 3] = bv.w; }
                        }
                }
            }
        }
        // (4) 18 MFMA k-steps from the current buffer, the s+1 transform interleaved between groups
        const int wsel = WRES ? (sq0.kc * C::W_BYTES) : 0;   // resident: slab of this kc
        const unsigned char* wb = w_cur + wsel + b_off;
        const unsigned char* ib = reinterpret_cast<const unsigned char*>(in_cur);
        unsigned tw[4];
        // Fragment reads are software-pipelined by hand: step st+1's four ds_read_b128 are issued before step
        // st's four MFMAs, then one piece of the s+1 transform; a sched_barrier per step keeps that order (and the
        // register budget: without it the scheduler hoists far ahead and spills).
        bf16x8_t bfr[2][NTL], afr[2][2];
        auto read_frags = [&](int st, bf16x8_t (&bf)[NTL], bf16x8_t (&af)[2]) {
            if constexpr (!(DBG & 2)) {
                const int tap = st >> 1;
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    bf[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wb + (2 * st * NT + j * 32) * 16));
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    af[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ib + (a_off[m][tap] ^ ((st & 1) << 5))));
            }
        };
        read_frags(0, bfr[0], afr[0]);
#pragma unroll
        for (int st = 0; st < RB_NSTEPS; ++st) {
            if (st + 1 < RB_NSTEPS) read_frags(st + 1, bfr[(st + 1) & 1], afr[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);   // keep the reads AHEAD of this step's MFMAs (hipcc sinks them otherwise)
            if constexpr (!(DBG & 3)) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
                    {
                        f32x16_t cin = acc[m][j];
                        if constexpr (WRES) { if (st == 0) cin = bias_acc; }
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[st & 1][j], afr[st & 1][m], cin, 0, 0, 0);  // D[cout][pixel]
                    }
            } else if constexpr (!(DBG & 2)) {
#pragma unroll
                for (int j = 0; j < NTL; ++j) asm volatile("" :: "v"(bfr[st & 1][j]));   // keep the reads alive
#pragma unroll
                for (int m = 0; m < 2; ++m) asm volatile("" :: "v"(afr[st & 1][m]));
            }
            {   // piece st of the s+1 transform (chunk st>>2, word st&3); two pieces per even k-step (interleaved dependency chains) measured
                // no better (profiles/r02_experiments.md)
                tw[st & 3] = transform_word(word_of(Rn.v[st >> 2], st & 3), st & 3, cA, cB);
                if ((st & 3) == 3) store_words(st >> 2, Rn, tw, in_nxt);
            }
            if constexpr (!WRES) {
                // (2) weight slab of stage s+1 by LDS-DMA (global_load_lds_dwordx4: 16 B per lane, no registers), a
                // pure copy into the OTHER buffer, 17 k-steps ahead of the end-of-stage wait + barrier that publish
                // it.  Inline asm on purpose: with the builtin hipcc orders every later ds_write after the DMA with
                // a vmcnt(0) (same LDS array => may alias), which would also drain the s+2 input prefetch.
                if (st == 0) {
                    const unsigned char* ws = reinterpret_cast<const unsigned char*>(wslab(sq1));
                    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
                    const unsigned w_nxt_lds = smem_lds + (unsigned)(w_nxt - smem);
#pragma unroll
                    for (int i = 0; i < C::W_ITERS; ++i) {
                        const int cbase = i * RB_THREADS + wave_u * 64;          // first chunk of this wave-instruction
                        if (cbase < C::W_CHUNKS)                                 // wave-uniform
                            rb_glds16(ws + (size_t)(cbase + lane) * 16, w_nxt_lds + cbase * 16);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(s, 1);
#pragma unroll
        for (int pc = RB_NSTEPS; pc < RB_IN_ITERS * 4; ++pc) {         // pieces 18, 19 (and 20..23 with 6 chunks per thread)
            tw[pc & 3] = transform_word(word_of(Rn.v[pc >> 2], pc & 3), pc & 3, cA, cB);
            if ((pc & 3) == 3) store_words(pc >> 2, Rn, tw, in_nxt);
        }
        // The s+2 prefetch has had a whole stage to land: retire it NOW (it is the oldest thing in this wave's
        // in-order VMEM queue), so that no later wait -- in particular the next stage's first use of these
        // registers, which on the epilogue path comes after 8 output stores -- has to drain anything newer.
#pragma unroll
        for (int i = 0; i < RB_IN_ITERS; ++i) asm volatile("" : "+v"(Rf.v[i].x), "+v"(Rf.v[i].y), "+v"(Rf.v[i].z), "+v"(Rf.v[i].w));
        asm volatile("" : "+v"(cnext.x), "+v"(cnext.y), "+v"(cnext.z), "+v"(cnext.w));
        put_coeffs((s + 1) & 1, cnext);
        // (5) the DMA'd weight slab must have landed before the stage barrier publishes it
        if constexpr (!WRES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(s, 2);
        // (6) last k-chunk of the item: epilogue
        const int kc = sq0.kc;
        if constexpr (WRES) direct_epilogue(sq0.it);
        else if constexpr (PAR == 1) { if (kc == nkc - 1) direct_epilogue(sq0.it); }
        stamp(s, 4);
        // (7) stage barrier: buf[nxt] complete, buf[cur] (and red[]) free
        __syncthreads();
        stamp(s, 5);
        flush_stats();                                          // red[st_par] is complete; its next writer is two items away
        sq0 = sq1; sq1 = sq2; sq2 = cursor.next();
    };

    // ---- prologue: stage 0 into buffer 0, stage 1 into registers --------------------------------------
    {
        float* bias_lds = reinterpret_cast<float*>(smem + C::MAIN_BYTES + C::RED_BYTES + C::COEF_BYTES);
        if (tid < a.cout) bias_lds[tid] = a.bias[tid];         // off the VMEM queue for good

        put_coeffs(1, fetch_coeffs(sq0));                       // slot 1 plays "stage -1": transform of stage 0's data
        load_stage(sq0, R0);
        if (WRES) {   // whole weight set of this n-block stays in LDS (nblocks == 1)
            const uint4* ws = reinterpret_cast<const uint4*>(a.w);
            uint4* wd = reinterpret_cast<uint4*>(smem + C::W_OFF0);
            for (int i = tid; i < C::W_CHUNKS * nkc; i += RB_THREADS) wd[i] = ws[i];
        } else {
            const uint4* ws = wslab(sq0);
            uint4* wd = reinterpret_cast<uint4*>(smem + C::W_OFF0);
            for (int i = tid; i < C::W_CHUNKS; i += RB_THREADS) wd[i] = ws[i];
        }
        __syncthreads();
        float cA[8], cB[8];
        load_coeffs(1, cA, cB);
        uint4* in0 = reinterpret_cast<uint4*>(smem);
#pragma unroll
        for (int i = 0; i < RB_IN_ITERS; ++i) store_chunk(i, R0, cA, cB, in0);
        put_coeffs(0, fetch_coeffs(sq1));           // slot 0: read at the top of stage 0 for stage 1's data
        load_stage(sq1, R1);
    }
    __syncthreads();
    if constexpr (WRES) {
        const float* bl = reinterpret_cast<const float*>(smem + C::MAIN_BYTES + C::RED_BYTES + C::COEF_BYTES);
#pragma unroll
        for (int i = 0; i < 16; ++i) bias_acc[i] = bl[16 * (i >> 3) + 8 * h + (i & 7)];     // permuted slab rows
    }

    // Waves 4-7 are the later-dispatched partners on each SIMD and lose issue arbitration to waves 0-3 on every
    // stage (measured with s_memtime: their MFMA loop took 6600 vs 4600 cycles, the older half then idles at the
    // barrier).  One static priority raise, no per-stage flips (cdna guide T5, static form); IRE_RB_PRIO=0 disables.
    if (a.prio_young && __builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_setprio(1);
    for (int s = 0; s < S; s += 2) {
        stage(s, std::integral_constant<int, 0>{});
        if (s + 1 < S) stage(s + 1, std::integral_constant<int, 1>{});
    }
    flush_stats();
}

template <int NT, bool RESID, bool WRES, bool FUSED_ACT, int DBG = 0, bool UPS = false, bool HEAD = false>
void launch_rb(const ConvArgs& a, hipStream_t stream) {
    const int items = a.tiles_x * a.tiles_y * a.nimg * a.nblocks;
    const int cus = persistent_grid_cus();
    // one workgroup per CU; with 8-row tiles (256 threads) two independent workgroups per CU where their LDS fits
    const int per_cu = (RB_TH == 8 && RbCfg<NT, RESID, WRES, FUSED_ACT>::LDS_BYTES <= 80 * 1024) ? 2 : 1;
    const int grid = items < cus * per_cu ? items : cus * per_cu;
    hipLaunchKernelGGL((conv_rb_kernel<NT, RESID, WRES, FUSED_ACT, DBG, UPS, HEAD>), dim3(grid), dim3(RB_THREADS), 0, stream, a);
    IRE_HIP(hipGetLastError());
}

template <int DBG>
void rb_dispatch(bool resid, bool fused_act, const ConvArgs& a, hipStream_t stream) {
    if (a.cout == 32) {
        if (resid) launch_rb<32, true, true, true, DBG>(a, stream); else launch_rb<32, false, true, true, DBG>(a, stream);
    } else if (fused_act) {
        if (resid) launch_rb<64, true, false, true, DBG>(a, stream); else launch_rb<64, false, false, true, DBG>(a, stream);
    } else {
        if (resid) launch_rb<64, true, false, false, DBG>(a, stream); else launch_rb<64, false, false, false, DBG>(a, stream);
    }
}

}  // namespace

void conv_up_launch(const ConvArgs& a, hipStream_t stream) {
    // nearest x2 -> conv3x3 (2C -> C): weights streamed (nkc >= 2), no prologue, no residual, no statistics
    if (a.cout == 32) launch_rb<32, false, false, false, 0, true>(a, stream);
    else launch_rb<64, false, false, false, 0, true>(a, stream);
}

void conv_head_launch(const ConvArgs& a, hipStream_t stream) {
    if (a.cout != 32 || a.nkc != 1 || a.nblocks != 1 || !a.ab || !a.u8_in || !a.u8_out) fail(IRE_ERR_INTERNAL, "internal: conv_head_launch arguments");
    launch_rb<32, false, true, true, 0, false, true>(a, stream);
}

void conv_rb_launch(bool resid, bool fused_act, const ConvArgs& a, hipStream_t stream) {
    if (a.stats == nullptr) fail(IRE_ERR_INTERNAL, "internal: conv_rb_launch needs a GroupNorm partials buffer (ResBlock convs always feed a GroupNorm)");
#ifdef IRE_RB_ABLATE
    static const int dbg = std::getenv("IRE_RB_DEBUG") ? std::atoi(std::getenv("IRE_RB_DEBUG")) : 0;
    switch (dbg) {
        case 1: return rb_dispatch<1>(resid, fused_act, a, stream);
        case 2: return rb_dispatch<2>(resid, fused_act, a, stream);
        case 4: return rb_dispatch<4>(resid, fused_act, a, stream);
        case 8: return rb_dispatch<8>(resid, fused_act, a, stream);
        case 6: return rb_dispatch<6>(resid, fused_act, a, stream);
        case 14: return rb_dispatch<14>(resid, fused_act, a, stream);
        case 16: return rb_dispatch<16>(resid, fused_act, a, stream);
        default: break;
    }
#endif
    rb_dispatch<0>(resid, fused_act, a, stream);
}

}  // namespace ire
