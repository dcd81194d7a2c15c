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
// Pipeline, VMEM ordering rules, LDS-DMA weights, stage queue: as conv_rb.hip.  The engine launches the
// fused-activation 8-wave form (GroupNorm+FiLM+SiLU while staging); the plain-copy staging of a pre-activated input and the
// 4-wave form are template branches no launch instantiates since round 4.  Epilogue: the 128-channel tile leaves in 4 passes of 32
// channels through the current stage buffer (bias, residual, GroupNorm partials, full-line stores).
#include "conv_mfma.hpp"
#include "persist.hpp"
#include "gn_fold.hpp"

#include <cstdlib>
#include <type_traits>

#ifndef W4_TEPI
#define W4_TEPI 1      // 0: the direct epilogue of round 2 (build-time A/B: IRE_W4_TEPI=0)
#endif

namespace ire {

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int W4_TH = 16, W4_TW = 32, W4_IH = 18, W4_IW = 34;
constexpr int W4_IN_CHUNKS = W4_IH * W4_IW * 2;                              // 1224 x 16 B (16 channels per pixel)
constexpr int W4_IN_BYTES = (W4_IN_CHUNKS + 8) * 16;                          // + dummy slot (chunk slots past the tile)
constexpr int W4_NSTEPS = 9;

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

using W4Item = PersistItem;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int N> struct W4Regs { u32x4_t v[N]; unsigned ok; };   // native 128-bit tuples: one register quad per load

// WAVES = 4: one wave per SIMD, wave tile 4 rows x NT couts (all 256 AGPRs are accumulators at NT = 128)
// WAVES = 8: two waves per SIMD, wave tile 2 rows x NT couts (128 accumulators): a partner wave's MFMAs cover this wave's
//            VMEM / LDS-DMA / epilogue issue slots, which a single wave per SIMD can only serialise
// FP8 (IRE_PRECISION_FP8, cfg 4): both MFMA operands are OCP e4m3 (v_mfma_f32_32x32x16_fp8_fp8): an 8-channel operand chunk
// is 8 bytes instead of 16 in the LDS tile and in the weight slab (ds_read_b64 fragments, half the LDS bytes per MFMA).
// Activations are quantised while staging -- after GroupNorm+FiLM+SiLU, scaled by kFp8ActScale so that small values stay
// normal numbers, clamped to the e4m3 range -- weights offline with one scale per output channel (engine.cpp::make_conv);
// the epilogue multiplies each accumulator by its channel's (weight scale / kFp8ActScale).  HBM tensors stay bf16.
constexpr float kFp8ActScale = 16.0f;
template <int NT, int WAVES, bool FP8 = false>
struct W4Cfg {
    static constexpr int THREADS = WAVES * 64;
    static constexpr int MT = W4_TH / WAVES;
    static constexpr int IN_ITERS = (W4_IN_CHUNKS + THREADS - 1) / THREADS;
    static constexpr int NTL = NT / 32;
    static constexpr int EB = FP8 ? 8 : 16;                     // bytes of one 8-channel operand chunk in LDS
    // one k-half plane of the input tile: 612 pixels x EB, sized so that (PLANE / 4) % 32 == 16 (conflict-free tile writes)
    static constexpr int PLANE = FP8 ? 4928 : (W4_IN_CHUNKS / 2) * 16;
    static constexpr int IN_BYTES = FP8 ? 2 * 4928 + 64 : W4_IN_BYTES;      // + dummy slot (chunk slots past the tile)
    static constexpr int W_BYTES = W4_NSTEPS * 2 * NT * EB;    // slab [tap][c8][NT] x EB
    static constexpr int W_CHUNKS = W_BYTES / 16;              // 16-B pieces of a slab (LDS-DMA granule)
    static constexpr int W_ITERS = (W_CHUNKS + THREADS - 1) / THREADS;
    static constexpr int W_BASE = 2 * IN_BYTES;                // in[2] | w[3]
    static constexpr int MAIN_BYTES = W_BASE + 3 * W_BYTES;
    static constexpr int RED_BYTES = 2 * WAVES * 16 * 2 * 4;   // [item parity][waves][16 chunks of 8 couts][sum, sumsq]  (old epilogue: [waves][8 slots of 16 couts][2])
    static constexpr int PATCH_BYTES = 16 * NT * 2;            // line-coalesced epilogue: a wave's transpose patch (16 pixels x NT couts) inside the stage's dead weight slab
    static_assert(FP8 || WAVES * PATCH_BYTES <= W4_NSTEPS * 2 * NT * EB, "the patches live in one weight slab");
    static constexpr int BIAS_BYTES = 256 * 4;
    static constexpr int COEF_BYTES = WAVES * 128;             // FUSED: per wave, 16 channels x (A, B) of the stage being staged
    static constexpr int SCALE_BYTES = FP8 ? 256 * 4 : 0;      // FP8: per-cout output scale
    static constexpr int LDS_BYTES = MAIN_BYTES + RED_BYTES + BIAS_BYTES + COEF_BYTES + SCALE_BYTES;
    static_assert((PLANE / 4) % 32 == 16, "plane offset must be half a bank row");
    static_assert(W_CHUNKS % 64 == 0, "slab = whole wave-instructions of LDS-DMA");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int NT, int WAVES, bool RESID, bool UPS, int DBG = 0, bool FUSED = false, bool FP8 = false>
__global__ __launch_bounds__(WAVES * 64) void conv_w4_kernel(ConvArgs a) {
    static_assert(!FP8 || (FUSED && WAVES == 8), "fp8 operands: fused-activation 8-wave form only");
    using C = W4Cfg<NT, WAVES, FP8>;
    constexpr int EB = C::EB;
    using Regs = W4Regs<C::IN_ITERS>;
    constexpr int NTL = C::NTL;
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int c8_fixed = tid & 1;   // 2 chunks per pixel per stage

    // ---- persistent work assignment (persist.hpp) ---------------------------------------------------------------
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    PersistCursor cursor(a.tiles_x, a.tiles_y, a.nimg, a.nblocks, a.nkc);
    const int my_items = cursor.my_items;
    const int nkc = a.nkc;                         // 16-channel stages per item
    const int S = cursor.S;
    if (S == 0) return;
#ifdef IRE_W4_TL
    if (a.stamps && tid == 0) { a.stamps[(size_t)blockIdx.x * 32] = __builtin_amdgcn_s_memrealtime(); a.stamps[(size_t)blockIdx.x * 32 + 1] = __builtin_amdgcn_s_memtime(); }
#endif
    using StageInfo = PersistStage;
    StageInfo sq0 = cursor.cur, sq1 = cursor.next(), sq2 = cursor.next();

    const int Cin = a.cin0;
    const int cin_shift = 31 - __builtin_clz(Cin);

    // ---- input tile in LDS: two planes (k-halves c8 = 0/1) of 18x34 pixels x 16 B.  Lane (r, h) reads pixel
    // p = (4*wave + m + ky)*34 + r + kx of plane h: ONE address register + immediates, and the 16 lanes a ds_read_b128
    // services together ({0-3,12-15,20-27}...) hit 16 distinct 16-B slots.  PLANE/4 mod 32 == 16, so the 8-lane groups
    // of the ds_write_b128 (4 pixels x 2 planes) are conflict-free too.
    constexpr int PLANE = C::PLANE;
    const int a_base = h * PLANE + (wave * C::MT * W4_IW + r) * EB;
    const int b_off = (h * NT + r) * EB;

    // one 16-B chunk per thread and call: chunk idx = tid + i*256 of the (18 x 34 px) x 2 halves tile of stage si
    // A thread's chunks sit at the same tile position in every stage of an item: their byte offsets within the image (and
    // whether they are inside it) are computed once per ITEM (item_offsets, when the prefetch cursor sq2 enters a new item:
    // ~25 VALU instructions per chunk) and a stage's request is base(item, kc) + offset.
    unsigned coff[C::IN_ITERS], cok = 0;
    auto item_offsets = [&](const W4Item& it) {
        const int oy1 = it.ty * W4_TH - 1, ox1 = it.tx * W4_TW - 1;
        cok = 0;
#pragma unroll
        for (int i = 0; i < C::IN_ITERS; ++i) {
            int t2 = tid;
            asm volatile("" : "+v"(t2));
            const int p = (t2 + i * C::THREADS) >> 1;
            const int py = p / W4_IW, px = p - py * W4_IW;
            const int iy = oy1 + py, ix = ox1 + px;
            const int WV = UPS ? a.Wout : a.Win;
            const int cy = min(max(iy, a.iy_lo), a.iy_lo + a.iy_span - 1), cx = min(max(ix, 0), WV - 1);
            const bool ok = iy == cy && ix == cx;
            const int sy = (UPS ? (cy >> 1) : cy) + a.in_row_off, sx = UPS ? (cx >> 1) : cx;
            coff[i] = ((unsigned)(sy * a.Win + sx) << (cin_shift + 1)) + (unsigned)(c8_fixed * 16);
            cok |= ok ? (1u << i) : 0u;
        }
    };
    auto load_chunk = [&](const StageInfo& si, int i, Regs& R) {       // si's item = the item item_offsets last saw
        const char* base = reinterpret_cast<const char*>(a.in0) + (size_t)si.it.img * a.in_rows * a.Win * Cin * 2 + si.kc * 32;
        R.v[i] = *reinterpret_cast<const u32x4_t*>(base + coff[i]);
        R.ok = (R.ok & ~(1u << i)) | (cok & (1u << i));
    };
    auto load_stage = [&](const StageInfo& si, Regs& R) {
        R.ok = 0;
#pragma unroll
        for (int i = 0; i < C::IN_ITERS; ++i) load_chunk(si, i, R);
    };
    // FUSED: y = silu(x*A + B) with (A, B) = GroupNorm+FiLM coefficients of the channel (gn_finalize), applied while staging, two
    // channels at a time in packed f32 (as conv_rb.hip::transform_word); zero padding applies AFTER the activation.
    // The coefficients of a stage's 16 channels travel global -> 8 lanes' float4 (fetched with the stage's input, a stage
    // ahead) -> this wave's own 128-B LDS slot -> every lane's 8 (A, B) pairs: wave-local ordering, no barrier.
    float* coef_lds = reinterpret_cast<float*>(smem + C::MAIN_BYTES + C::RED_BYTES + C::BIAS_BYTES) + wave * 32;
    auto fetch_coeffs = [&](const StageInfo& si) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (FUSED) v = reinterpret_cast<const float4*>(a.ab + (size_t)si.it.img * Cin + si.kc * 16)[lane & 7];
        return v;
    };
    float cA[8], cB[8];
    auto stage_coeffs = [&](const float4& v) {          // publish to the wave's slot, read back this lane's half (c8)
        if constexpr (FUSED) {
            // published as (A, A', B, B') per channel pair: a lane's float4 read is then the two register pairs the packed
            // FMA takes (as (A, B, A', B') every use cost four moves to pair them up)
            if (lane < 8) reinterpret_cast<float4*>(coef_lds)[lane] = make_float4(v.x, v.z, v.y, v.w);
            const float4* ab = reinterpret_cast<const float4*>(coef_lds + c8_fixed * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float4 t = ab[e]; cA[2 * e] = t.x; cA[2 * e + 1] = t.y; cB[2 * e] = t.z; cB[2 * e + 1] = t.w; }
            if constexpr (FP8) {       // y' = kFp8ActScale * y: silu comes out pre-scaled for the e4m3 conversion at no per-element cost
#pragma unroll
                for (int e = 0; e < 8; ++e) { cA[e] *= kFp8ActScale; cB[e] *= kFp8ActScale; }
            }
        }
    };
    // (plain f32 instructions, one channel each, and the file is built with -fno-slp-vectorize: beside MFMAs a packed f32
    // instruction costs several times two plain ones -- MI355X_MICROARCH.md per-instruction constants; profiles/r02_experiments.md)
    auto transform_pair = [&](unsigned w, int d) -> f32x2_t {        // (kFp8ActScale x) silu(x*A + B) of the word's two channels
        constexpr float K = FP8 ? -1.4426950408889634f / kFp8ActScale : -1.4426950408889634f;
        const float y0 = __builtin_fmaf(w4_lo(w), cA[2 * d], cB[2 * d]), y1 = __builtin_fmaf(w4_hi(w), cA[2 * d + 1], cB[2 * d + 1]);
        const float e0 = __builtin_amdgcn_exp2f(y0 * K) + 1.0f, e1 = __builtin_amdgcn_exp2f(y1 * K) + 1.0f;
        const f32x2_t sv = {y0 * __builtin_amdgcn_rcpf(e0), y1 * __builtin_amdgcn_rcpf(e1)};
        return sv;
    };
    auto transform_word = [&](unsigned w, int d) -> unsigned {
        const f32x2_t sv = transform_pair(w, d);
        return w4_pack(sv.x, sv.y);
    };
    auto store_chunk = [&](int i, const Regs& R, uint4* lds_in) {   // copy (or activate); zero padding outside the image
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int idx = t2 + i * C::THREADS;
        const bool ok = (R.ok >> i) & 1u;
        const u32x4_t zero = {0u, 0u, 0u, 0u};
        u32x4_t v = R.v[i];
        if constexpr (FP8) {
            // 8 channels -> 8 e4m3 bytes (v_cvt_pk_fp8_f32: round to nearest even); clamped to the format's largest finite value first
            const unsigned wds[4] = {v.x, v.y, v.z, v.w};
            int q[2] = {0, 0};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const f32x2_t sv = transform_pair(wds[d], d);
                const float f0 = __builtin_fminf(sv.x, 448.0f), f1 = __builtin_fminf(sv.y, 448.0f);
                q[d >> 1] = (d & 1) ? __builtin_amdgcn_cvt_pk_fp8_f32(f0, f1, q[d >> 1], true) : __builtin_amdgcn_cvt_pk_fp8_f32(f0, f1, q[d >> 1], false);
            }
            uint2 o8;
            o8.x = ok ? (unsigned)q[0] : 0u; o8.y = ok ? (unsigned)q[1] : 0u;
            unsigned char* lb = reinterpret_cast<unsigned char*>(lds_in);
            *reinterpret_cast<uint2*>(idx < W4_IN_CHUNKS ? lb + c8_fixed * PLANE + (idx >> 1) * 8 : lb + 2 * PLANE) = o8;
            return;
        }
        if constexpr (FUSED) { v.x = transform_word(v.x, 0); v.y = transform_word(v.y, 1); v.z = transform_word(v.z, 2); v.w = transform_word(v.w, 3); }
        const u32x4_t o = ok ? v : zero;
        const int slot = c8_fixed * (W4_IN_CHUNKS / 2) + (idx >> 1);
        reinterpret_cast<u32x4_t*>(lds_in)[idx < W4_IN_CHUNKS ? slot : W4_IN_CHUNKS] = o;
    };
    auto wslab = [&](const StageInfo& si) -> const unsigned char* {
        return reinterpret_cast<const unsigned char*>(a.w) + ((size_t)si.it.nb * nkc + si.kc) * C::W_BYTES;
    };

    f32x16_t acc[C::MT][NTL];
    // accumulators start at the bias (permuted slab rows: accumulator i of lane-half h is cout nb*NT + j*32 + 16(i>>3) + 8h + (i&7)):
    // 4*NTL LDS reads per item instead of 16*NTL*MT adds in the epilogue; the moves take the place of the zeroing
    constexpr bool BIAS_INIT = WAVES == 8;     // (the 512-register form keeps zeroing: a non-constant fill of 256 AGPRs spills)
    auto zero_acc = [&](int nb) {
        if constexpr (!BIAS_INIT) {
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[m][j][i] = 0.f;
            return;
        }
        const float* bl = reinterpret_cast<const float*>(smem + C::MAIN_BYTES + C::RED_BYTES) + nb * NT + 8 * h;
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *reinterpret_cast<const float4*>(bl + j * 32 + 16 * (q >> 1) + 4 * (q & 1));
#pragma unroll
                for (int m = 0; m < C::MT; ++m) { acc[m][j][4 * q + 0] = bv.x; acc[m][j][4 * q + 1] = bv.y; acc[m][j][4 * q + 2] = bv.z; acc[m][j][4 * q + 3] = bv.w; }
            }
    };

    float* red = reinterpret_cast<float*>(smem + C::MAIN_BYTES);                       // [4 waves][4 cc][4]
    const float* bias_lds = reinterpret_cast<const float*>(smem + C::MAIN_BYTES + C::RED_BYTES);

    int s = 0;
    auto stamp = [&](int k) {
#ifdef IRE_W4_TICKS
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (a.stamps && lane == 0 && (wave == 0 || wave == WAVES - 1) && blockIdx.x < 8 && s < 64)
            a.stamps[(((size_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 64 + s) * 10 + k] = t;
#else
        (void)k;
#endif
    };
    // workgroup TIMELINE (diagnostic build -DIRE_W4_TL, tools/r04_tl.sh): wave 0 of every workgroup stamps the constant 100 MHz
    // counter (comparable across CUs and XCDs) and its own shader clock at kernel entry (0), after the folded GroupNorm finalize (1),
    // after the prologue (2: the first MFMA follows), at the end of each item (3 + k) and at exit (12); slot 13 = (XCC id, items)
    auto tl = [&](int slot) {
#ifdef IRE_W4_TL
        if (a.stamps && tid == 0) {
            a.stamps[(size_t)blockIdx.x * 32 + slot * 2] = __builtin_amdgcn_s_memrealtime();
            a.stamps[(size_t)blockIdx.x * 32 + slot * 2 + 1] = __builtin_amdgcn_s_memtime();
        }
#else
        (void)slot;
#endif
    };
    // residual prefetch state lives across the item's last stage: the first RD-1 groups are requested BEFORE that stage's
    // MFMAs (k-steps 0..3 carry no other VMEM), so the epilogue finds them landed
    constexpr int RD = FUSED ? 2 : 4;                // residual groups in flight (the fused variant has 16 registers of coefficients live)
    uint4 rv[RD][C::MT];
    unsigned offs[C::MT];
    bool inb[C::MT];
    int cout0_e = 0;
    // ---- line-coalesced epilogue (TEPI): the accumulator layout (lane = pixel, 8 couts per register group) makes every global
    // instruction of the direct epilogue touch 32 pixels x 32 B -- 32 different lines, ~3 cycles each through the L1 (stamps,
    // profiles/r03_experiments.md: the RB2 epilogue was 23 800 ticks per item, 3.2 stages' worth, RB1's 11 000).  Here a wave
    // transposes 16 pixels x 128 couts at a time through a 4-KB patch of the stage's dead weight slab (XOR-swizzled: conflict-free
    // both ways) and reads / writes global memory in whole 256-B pixel runs: 1 KB contiguous per instruction.
    constexpr bool TEPI = W4_TEPI && WAVES == 8 && !FP8 && !(DBG & 64);        // (the same-rate fp8 fallback has half-size slabs: direct epilogue)
    unsigned toffs[C::MT];        // byte offset of (this lane's read-back pixel at q = 0, k = 0; its chunk c = lane & 15) in the output image
    bool trow[C::MT];
    int tcol0 = 0;
    constexpr int NCHK = NT / 8;                  // 16-B chunks of a pixel's NT-cout run: 16 (NT = 128) or 8 (NT = 64)
    constexpr int NRD = NCHK / 4;                 // read-backs (1 KB each) per 16-pixel pass: 4 or 2
    constexpr int PPR = 64 / NCHK;                // pixels per read-back: 4 or 8
    uint4 rvt[3][NRD];            // residual rows of pass (m, q) in slot pass % 3, read-back layout: NRD loads of 1 KB each
    auto load_resid_pass = [&](int pass, uint4 (&dst)[NRD]) {
        if constexpr (RESID) {
            char* rbase = const_cast<char*>(reinterpret_cast<const char*>(a.resid)) + (size_t)sq0.it.img * a.Hout * a.Wout * a.cout * 2;
            const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(rbase, 0, a.Hout * a.Wout * a.cout * 2, 0x00020000);
            const int m = pass >> 1, q = pass & 1;
            const unsigned cstep = (unsigned)a.cout * 2u * PPR;    // bytes per read-back's PPR pixels
#pragma unroll
            for (int k = 0; k < NRD; ++k) {
                const bool ok = trow[m] && tcol0 + 16 * q + PPR * k < a.Wout;
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ok ? toffs[m] + (unsigned)(NRD * q + k) * cstep : 0xffffffffu, 0, 0);
                dst[k] = make_uint4(v.x, v.y, v.z, v.w);
            }
        }
    };
    auto load_resid = [&](int g, uint4 (&dst)[C::MT]) {
        if constexpr (RESID) {
            const char* rbase = reinterpret_cast<const char*>(a.resid) + (size_t)sq0.it.img * a.Hout * a.Wout * a.cout * 2;
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
                dst[m] = *reinterpret_cast<const uint4*>(rbase + offs[m] + (unsigned)((g >> 1) * 64 + (g & 1) * 32));
        }
    };
    auto epi_prefetch = [&]() {
        if constexpr (TEPI) {
            const W4Item it = sq0.it;
            int l_e = lane, w_e = wave;
            asm volatile("" : "+v"(l_e), "+v"(w_e));
            cout0_e = it.nb * NT;
            const int oyb = it.ty * W4_TH + w_e * C::MT;
            tcol0 = it.tx * W4_TW + l_e / NCHK;
#pragma unroll
            for (int m = 0; m < C::MT; ++m) {
                const int oy = oyb + m;
                trow[m] = oy < a.Hout;
                toffs[m] = ((unsigned)((min(oy, a.Hout - 1) * a.Wout + tcol0) * a.cout + cout0_e + 8 * (l_e % NCHK)) << 1);
            }
            load_resid_pass(0, rvt[0]);       // before the stage's MFMAs (k-steps 0..3 carry no other VMEM): landed when the epilogue starts
            return;
        }
        const W4Item it = sq0.it;
        int r_e = r, h_e = h, w_e = wave;
        asm volatile("" : "+v"(r_e), "+v"(h_e), "+v"(w_e));
        cout0_e = it.nb * NT;
        const int oyb = it.ty * W4_TH + w_e * C::MT, ox = it.tx * W4_TW + r_e;
        const bool colok = ox < a.Wout;
        const int oxc = min(ox, a.Wout - 1);
#pragma unroll
        for (int m = 0; m < C::MT; ++m) {
            const int oy = oyb + m;
            inb[m] = colok && oy < a.Hout;
            offs[m] = ((unsigned)((min(oy, a.Hout - 1) * a.Wout + oxc) * a.cout + cout0_e) << 1) + (unsigned)(h_e * 16);
        }
#pragma unroll
        for (int g = 0; g + 1 < RD; ++g) load_resid(g, rv[g]);
    };
    // ---- one stage = 16 input channels: 9 k-steps (taps) of 16 MFMAs.  Pipeline (s = this stage):
    //   * weights: three LDS slabs; the slab of stage s+2 is fetched by LDS-DMA during stage s (k-steps 4, 5)
    //   * input:   two LDS tiles; one register set R holds stage s+1 (loaded during stage s-1); during k-steps 4..8 chunk i
    //              (k-step 4 + i) goes R -> LDS tile (s+1)&1 and R.v[i] is reloaded with stage s+2
    //   * k-steps 0..3 issue no VMEM at all; the single wait (vmcnt(0) at k-step 4) therefore only sees operations issued
    //     at least four k-steps earlier, and nothing is waited for at the stage end but the barrier.
    Regs R;
    float4 cnext = make_float4(0.f, 0.f, 0.f, 0.f);   // FUSED: coefficients of the stage whose input R holds
    int widx = 0;                                   // weight slab of the current stage (s % 3)
    int par = 0;                                    // input tile of the current stage (s & 1)
    // ONE stage body in the loop (tile parity and slab index are run-time offsets): unrolling stage pairs made the
    // allocator park the reloaded R.v[i] in different registers per copy and "fix" that with vmcnt(0) + v_mov after each load
    auto compute = [&](auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;   // the item's last stage: residual prefetch in flight
        const unsigned char* ib = smem + par * C::IN_BYTES;
        uint4* in_nxt = reinterpret_cast<uint4*>(smem + (par ^ 1) * C::IN_BYTES);
        const unsigned char* wb = smem + C::W_BASE + widx * C::W_BYTES + b_off;
        const int w2 = widx == 0 ? 2 : widx - 1;    // (s + 2) % 3
        const unsigned w_dst_lds = smem_lds + C::W_BASE + w2 * C::W_BYTES;
        stamp(0);
        using frag_t = typename std::conditional<FP8, long, bf16x8_t>::type;     // 8 k-values per lane: 8 e4m3 bytes or 8 bf16
        frag_t bfr[2][NTL], afr[2][C::MT];
        auto read_frags = [&](int st, frag_t (&bf)[NTL], frag_t (&af)[C::MT]) {
            const int ky = st / 3, kx = st - ky * 3;
            if constexpr (FP8) {
#pragma unroll
                for (int j = 0; j < NTL; ++j) bf[j] = *reinterpret_cast<const long*>(wb + (2 * st * NT + j * 32) * 8);
#pragma unroll
                for (int m = 0; m < C::MT; ++m) af[m] = *reinterpret_cast<const long*>(ib + a_base + ((m + ky) * W4_IW + kx) * 8);
            } else {
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    bf[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wb + (2 * st * NT + j * 32) * 16));
#pragma unroll
                for (int m = 0; m < C::MT; ++m)
                    af[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ib + a_base + ((m + ky) * W4_IW + kx) * 16));
            }
        };
        read_frags(0, bfr[0], afr[0]);
#pragma unroll
        for (int st = 0; st < W4_NSTEPS; ++st) {
            // progress feedback: a wave early in its stage outranks one that is further along.  (The SIMD's arbitration otherwise
            // favours the older of its two waves: it runs ahead, idles ~a quarter of every stage at the barrier, and the younger
            // one then runs without a partner to fill its stalls -- tools/s2_stamps.sh.  A static priority only swaps the roles.)
            // (boundaries 0 / 4 / 7 -- the MFMA-only k-steps, the first three staging k-steps, the last two -- measured against 0 / 3 / 6:
            //  RB1 / RB2 168.2 / 191.1 -> 165.0 / 187.2 us; 0 / 5 / 7 and 0 / 4 / 8 the same as 0 / 4 / 7)
            if (WAVES == 8 && st == 0) __builtin_amdgcn_s_setprio(3);
            if (WAVES == 8 && st == 4) __builtin_amdgcn_s_setprio(2);
            if (WAVES == 8 && st == 7) __builtin_amdgcn_s_setprio(1);
            if (st + 1 < W4_NSTEPS && !(DBG & 2)) read_frags(st + 1, bfr[(st + 1) & 1], afr[(st + 1) & 1]);
            if (st == 4) {
                // everything issued during the previous stage (R loads, slab s+1, epilogue stores) has had >= 4 k-steps
                stamp(6);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                stamp(7);
#pragma unroll
                for (int i = 0; i < C::IN_ITERS; ++i) asm volatile("" : "+v"(R.v[i]));
                if constexpr (FUSED) {            // coefficients of the data now in R (fetched with it, one stage ago)
                    asm volatile("" : "+v"(cnext.x), "+v"(cnext.y), "+v"(cnext.z), "+v"(cnext.w));
                    stage_coeffs(cnext);
                }
                if constexpr (LAST && RESID && TEPI) {
#pragma unroll
                    for (int k = 0; k < NRD; ++k) asm volatile("" : "+v"(rvt[0][k].x), "+v"(rvt[0][k].y), "+v"(rvt[0][k].z), "+v"(rvt[0][k].w));
                } else if constexpr (LAST && RESID) {      // landed as well: tell the compiler, or it re-waits with its own (short) count
#pragma unroll
                    for (int g = 0; g + 1 < RD; ++g)
#pragma unroll
                        for (int m = 0; m < C::MT; ++m) asm volatile("" : "+v"(rv[g][m].x), "+v"(rv[g][m].y), "+v"(rv[g][m].z), "+v"(rv[g][m].w));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    if constexpr (FP8) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(bfr[st & 1][j], afr[st & 1][m], acc[m][j], 0, 0, 0);
                    else if constexpr (!(DBG & 1)) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[(DBG & 2) ? 0 : (st & 1)][j], afr[(DBG & 2) ? 0 : (st & 1)][m], acc[m][j], 0, 0, 0);  // D[cout][pixel]
            if (st >= 4) {
                const int i = st - 4;
                if (!(DBG & 8) && i < C::IN_ITERS) {
                    // (the last chunk index only exists for the first threads: waves all of whose lanes are past the tile skip it)
                    const bool live = (i + 1) * C::THREADS <= W4_IN_CHUNKS || __builtin_amdgcn_readfirstlane(wave) * 64 + i * C::THREADS < W4_IN_CHUNKS;
                    if (live) {
                        store_chunk(i, R, in_nxt);                       // stage s+1 input -> other tile
                        load_chunk(sq2, i, R);                          // stage s+2 input -> R.v[i]
                    }
                }
                if (FUSED && i == 0) cnext = fetch_coeffs(sq2);   // after the last use of the old coefficients (stage_coeffs at k-step 4)
                // slab s+2: all DMA issues in k-steps 4 and 5, so that the stage's LAST VMEM operation is the third input chunk at
                // k-step 6 -- seven k-steps of flight before the next stage's vmcnt(0); spread over k-steps 4..8 the last issue had
                // four MFMA-only k-steps (~1 us) and the wait stalled: RB1 / RB2 182 / 200 -> 176.5 / 196 us
                if constexpr (!(DBG & 32)) {
                    const unsigned char* ws = wslab(sq2);
                    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
                    // branch-free issue (a CFG merge here makes hipcc drain vmcnt): waves past the slab's end re-fetch a
                    // chunk group another wave also fetches (same bytes, same destination)
                    static_assert(C::W_CHUNKS >= C::THREADS, "slab >= one workgroup-wide DMA issue");
#pragma unroll
                    for (int d = 5 * i; i < 2 && d < 5 * i + 5 && d < C::W_ITERS; ++d) {
                        int cbase = d * C::THREADS + wave_u * 64;
                        if ((d + 1) * C::THREADS > C::W_CHUNKS) cbase = cbase >= C::W_CHUNKS ? cbase - C::W_CHUNKS : cbase;   // wraps to a 64-aligned group
                        w4_glds16(ws + (size_t)(cbase + lane) * 16, w_dst_lds + cbase * 16);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(1);
        stamp(2);
    };
    // ---- epilogue, straight from the accumulators (no LDS transpose, no workgroup barrier but the one for the GroupNorm
    // partials).  The slab rows are permuted (bits 2 and 3 of the MFMA row swapped, engine.cpp::make_conv), so accumulator i
    // of lane (r = pixel column, h) is cout j*32 + 16*(i>>3) + 8h + (i&7): 8 contiguous couts per half tile = one 16-B
    // store per lane, a pixel's two lanes 32 contiguous bytes, no lane exchange.
    // It only READS acc: the item loop zeroes the accumulators unconditionally (a conditional redefinition of all 256
    // AGPRs is a PHI the allocator can only resolve through scratch).
    auto epilogue = [&]() {
        const W4Item it = sq0.it;
        int h_e = h;
        asm volatile("" : "+v"(h_e));
        const int cout0 = cout0_e;
        char* obase = reinterpret_cast<char*>(a.out) + (size_t)it.img * a.Hout * a.Wout * a.cout * 2;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, a.Hout * a.Wout * a.cout * 2, 0x00020000);
        const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
        float ssum[NTL][2], qsum[NTL][2];
#pragma unroll
        for (int g = 0; g < NTL * 2; ++g) {            // g = j*2 + p: chunks 2p, 2p+1 of the 32 couts j
            const int j = g >> 1, pp = g & 1;
            __builtin_amdgcn_sched_barrier(0);          // keep only one group's accumulator copies live
            if (g + RD - 1 < NTL * 2) load_resid(g + RD - 1, rv[(g + RD - 1) % RD]);
            // permuted slab rows (engine.cpp::make_conv): accumulators 8pp .. 8pp+7 are the 8 contiguous couts j*32 + 16pp + 8h + (0..7)
            float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;     // BIAS_INIT: the accumulators started at the bias
            if constexpr (!BIAS_INIT) {
                b0 = *reinterpret_cast<const float4*>(bias_lds + cout0 + j * 32 + 16 * pp + 8 * h_e);
                b1 = *reinterpret_cast<const float4*>(bias_lds + cout0 + j * 32 + 16 * pp + 8 * h_e + 4);
            }
            float ts = 0.f, tq = 0.f;
            float4 s0 = make_float4(1.f, 1.f, 1.f, 1.f), s1 = s0;
            if constexpr (FP8) {        // per-cout dequantisation: weight scale / activation scale (the accumulators started at bias / scale)
                const float* sl = reinterpret_cast<const float*>(smem + C::MAIN_BYTES + C::RED_BYTES + C::BIAS_BYTES + C::COEF_BYTES);
                s0 = *reinterpret_cast<const float4*>(sl + cout0 + j * 32 + 16 * pp + 8 * h_e);
                s1 = *reinterpret_cast<const float4*>(sl + cout0 + j * 32 + 16 * pp + 8 * h_e + 4);
            }
#pragma unroll
            for (int m = 0; m < C::MT; ++m) {
                const f32x16_t& c = acc[m][j];
                unsigned x0, x1, y0, y1;
                if constexpr (FP8) {
                    x0 = w4_pack(c[8 * pp + 0] * s0.x, c[8 * pp + 1] * s0.y); x1 = w4_pack(c[8 * pp + 2] * s0.z, c[8 * pp + 3] * s0.w);
                    y0 = w4_pack(c[8 * pp + 4] * s1.x, c[8 * pp + 5] * s1.y); y1 = w4_pack(c[8 * pp + 6] * s1.z, c[8 * pp + 7] * s1.w);
                } else if constexpr (BIAS_INIT) {
                    x0 = w4_pack(c[8 * pp + 0], c[8 * pp + 1]); x1 = w4_pack(c[8 * pp + 2], c[8 * pp + 3]);
                    y0 = w4_pack(c[8 * pp + 4], c[8 * pp + 5]); y1 = w4_pack(c[8 * pp + 6], c[8 * pp + 7]);
                } else {
                    x0 = w4_pack(c[8 * pp + 0] + b0.x, c[8 * pp + 1] + b0.y); x1 = w4_pack(c[8 * pp + 2] + b0.z, c[8 * pp + 3] + b0.w);
                    y0 = w4_pack(c[8 * pp + 4] + b1.x, c[8 * pp + 5] + b1.y); y1 = w4_pack(c[8 * pp + 6] + b1.z, c[8 * pp + 7] + b1.w);
                }
                unsigned w[4] = {x0, x1, y0, y1};
                if constexpr (RESID) {
                    const uint4 rr = rv[g % RD][m];
                    const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                    for (int d = 0; d < 4; ++d) w[d] = w4_pack(w4_lo(w[d]) + w4_lo(rw[d]), w4_hi(w[d]) + w4_hi(rw[d]));
                }
                float s1 = 0.f, q1 = 0.f;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const bf16x2_t wv = __builtin_bit_cast(bf16x2_t, w[d]);
                    s1 = __builtin_amdgcn_fdot2_f32_bf16(wv, ones, s1, false);
                    q1 = __builtin_amdgcn_fdot2_f32_bf16(wv, wv, q1, false);
                }
                ts += inb[m] ? s1 : 0.f; tq += inb[m] ? q1 : 0.f;
                const bool st_ok = inb[m] && (!(DBG & 4) || w[0] == 0x12345678u);
                if constexpr (WAVES == 8) {
                    // range-checked buffer store: out-of-image lanes get an offset past num_records and are dropped by the hardware
                    const u32x4_t wv4 = {w[0], w[1], w[2], w[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(wv4, orsrc, st_ok ? offs[m] + (unsigned)(j * 64 + pp * 32) : 0xffffffffu, 0, IRE_ST_PART);
                } else {          // 512-register form: no room for the resource descriptor's live range
                    if (st_ok) *reinterpret_cast<uint4*>(obase + offs[m] + (unsigned)(j * 64 + pp * 32)) = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
            ssum[j][pp] = ts; qsum[j][pp] = tq;
        }
        if (a.stats) {
            // GroupNorm partials: group size G = 16 or 32 here (C >= 128), slot (j, p) = 16 couts => 1 or 2 slots per group
            // all 2*NG chains advance one step at a time, side by side: every DPP reads a value written >= 3 instructions
            // earlier (no s_nop padding); then one 8-B LDS write per group
            if constexpr (NTL == 4) {
                // 16 values (8 slots x (sum, squares)) over the 64 lanes, TRANSPOSING for the first two steps (conv_pc.hip): a lane keeps
                // half of its values and hands the other half to its partner (lane ^ 1: the kind; lane ^ 2: slots 0..3 / 4..7), so
                // 16 -> 8 -> 4 values per lane; those take the plain steps over lane bits 2..5.  28 cross-lane instructions
                // instead of 96.  Lane l ends with kind b0 of slots 4 b1 .. 4 b1 + 3.
                const bool b0 = lane & 1, b1 = lane & 2;
                auto xch = [&](float keep, float give, auto ctrl_tag) __attribute__((always_inline)) -> float {
                    const int gg = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), decltype(ctrl_tag)::value, 0xf, 0xf, false);
                    return keep + __builtin_bit_cast(float, gg);
                };
                float u[8], t4[4];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const float sv = ssum[g >> 1][g & 1], qv = qsum[g >> 1][g & 1];
                    u[g] = xch(b0 ? qv : sv, b0 ? sv : qv, std::integral_constant<int, 0xb1>{});          // quad_perm [1,0,3,2]: lane ^ 1
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) t4[k] = xch(b1 ? u[4 + k] : u[k], b1 ? u[k] : u[4 + k], std::integral_constant<int, 0x4e>{});   // quad_perm [2,3,0,1]: lane ^ 2
#pragma unroll
                for (int k = 0; k < 4; ++k) t4[k] = w4_ror_add<4>(t4[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) t4[k] = w4_ror_add<8>(t4[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) t4[k] = w4_swap16_add(t4[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) t4[k] = w4_swap32_add(t4[k]);
                if ((lane & 60) == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) red[(wave * 8 + (b1 ? 4 : 0) + k) * 2 + (b0 ? 1 : 0)] = t4[k];
                }
            } else {
                float rv16[NTL * 4];
    #pragma unroll
                for (int g = 0; g < NTL * 2; ++g) { rv16[2 * g] = ssum[g >> 1][g & 1]; rv16[2 * g + 1] = qsum[g >> 1][g & 1]; }
    #pragma unroll
                for (int i = 0; i < NTL * 4; ++i) rv16[i] = w4_ror_add<1>(rv16[i]);
    #pragma unroll
                for (int i = 0; i < NTL * 4; ++i) rv16[i] = w4_ror_add<2>(rv16[i]);
    #pragma unroll
                for (int i = 0; i < NTL * 4; ++i) rv16[i] = w4_ror_add<4>(rv16[i]);
    #pragma unroll
                for (int i = 0; i < NTL * 4; ++i) rv16[i] = w4_ror_add<8>(rv16[i]);
    #pragma unroll
                for (int i = 0; i < NTL * 4; ++i) rv16[i] = w4_swap16_add(rv16[i]);
    #pragma unroll
                for (int i = 0; i < NTL * 4; ++i) rv16[i] = w4_swap32_add(rv16[i]);
                if (lane == 0) {
    #pragma unroll
                    for (int g = 0; g < NTL * 2; ++g) *reinterpret_cast<float2*>(red + (wave * 8 + g) * 2) = make_float2(rv16[2 * g], rv16[2 * g + 1]);
                }
            }
            __syncthreads();
            const int G = a.group_size, spg = G >> 4, ngl = NT / G;     // slots per group (1 or 2), groups in the item
            if (tid < ngl) {
                float sv = 0.f, qv = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w)
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        if (k < spg) { sv += red[(w * 8 + tid * spg + k) * 2 + 0]; qv += red[(w * 8 + tid * spg + k) * 2 + 1]; }
                const int gg = cout0 / G + tid;
                float* st = a.stats + (((size_t)it.img * tiles_per_img + it.tile) * 8 + gg) * 2;
                st[0] = sv; st[1] = qv;
            }
        }
    };
    // ---- TEPI: line-coalesced epilogue through wave-private LDS patches (see the note at `TEPI`) ---------------------------------
    int st_img = -1, st_tile = 0, st_cout0 = 0, st_par = 0, red_par = 0;
    auto flush_stats = [&]() {          // GroupNorm partials of the item that finished before the last stage barrier: 8 waves x 16 chunk slots -> groups
        if (st_img < 0) return;
        const int G = a.group_size, cpg = G >> 3, ngl = NT / G;      // chunks of 8 couts per group (2 or 4), groups in the item
        if (a.stats && tid < ngl) {
            const float* rd = red + st_par * (WAVES * 32);
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < cpg) { sv += rd[(w * NCHK + tid * cpg + k) * 2 + 0]; qv += rd[(w * NCHK + tid * cpg + k) * 2 + 1]; }
            float* st = a.stats + (((size_t)st_img * tiles_per_img + st_tile) * 8 + st_cout0 / G + tid) * 2;
            st[0] = sv; st[1] = qv;
        }
        st_img = -1;
    };
    auto epilogue_t = [&]() {
        const W4Item it = sq0.it;
        int h_e = h, l_e = lane;
        asm volatile("" : "+v"(h_e), "+v"(l_e));
        const int cout0 = cout0_e;
        // every wave is past its last fragment read of the stage: the slab becomes eight wave-private patches
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stamp(8);
        if constexpr (RESID) { load_resid_pass(1, rvt[1]); load_resid_pass(2, rvt[2]); }      // into the registers the fragments just freed; pass p + 3 follows pass p
        unsigned char* patch = smem + C::W_BASE + widx * C::W_BYTES + wave * C::PATCH_BYTES;
        char* obase = reinterpret_cast<char*>(a.out) + (size_t)it.img * a.Hout * a.Wout * a.cout * 2;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, a.Hout * a.Wout * a.cout * 2, 0x00020000);
        const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
        const int r16 = l_e & 15, qh = (l_e >> 4) & 1;                    // writer: pixel r = 16 qh + r16 of the row, half h
        const int pq = l_e / NCHK, cc_r = l_e % NCHK;                     // reader: pixel PPR k + pq of the half-row, chunk cc_r (8 couts)
        constexpr int PITCH = NT * 2;                                     // bytes of a pixel's run in the patch
        const unsigned cstep = (unsigned)a.cout * 2u * PPR;
        float ssum = 0.f, qsum = 0.f;
#pragma unroll
        for (int m = 0; m < C::MT; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                __builtin_amdgcn_sched_barrier(0);
                if (qh == q) {
#pragma unroll
                    for (int g = 0; g < NTL * 2; ++g) {
                        const int j = g >> 1, pp = g & 1;
                        const f32x16_t& c = acc[m][j];
                        u32x4_t wv;
                        if constexpr (FP8) {
                            const float* sl = reinterpret_cast<const float*>(smem + C::MAIN_BYTES + C::RED_BYTES + C::BIAS_BYTES + C::COEF_BYTES);
                            const float4 s0 = *reinterpret_cast<const float4*>(sl + cout0 + j * 32 + 16 * pp + 8 * h_e);
                            const float4 s1 = *reinterpret_cast<const float4*>(sl + cout0 + j * 32 + 16 * pp + 8 * h_e + 4);
                            wv = u32x4_t{w4_pack(c[8 * pp + 0] * s0.x, c[8 * pp + 1] * s0.y), w4_pack(c[8 * pp + 2] * s0.z, c[8 * pp + 3] * s0.w),
                                         w4_pack(c[8 * pp + 4] * s1.x, c[8 * pp + 5] * s1.y), w4_pack(c[8 * pp + 6] * s1.z, c[8 * pp + 7] * s1.w)};
                        } else {
                            wv = u32x4_t{w4_pack(c[8 * pp + 0], c[8 * pp + 1]), w4_pack(c[8 * pp + 2], c[8 * pp + 3]),
                                         w4_pack(c[8 * pp + 4], c[8 * pp + 5]), w4_pack(c[8 * pp + 6], c[8 * pp + 7])};
                        }
                        const int cc = 2 * g + h_e;                       // chunk of the pixel's 128-cout run: couts 8 cc .. 8 cc + 7
                        *reinterpret_cast<u32x4_t*>(patch + r16 * PITCH + ((cc ^ (r16 & (NCHK - 1))) << 4)) = wv;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < NRD; ++k) {
                    const int p = PPR * k + pq;
                    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(patch + p * PITCH + ((cc_r ^ (p & (NCHK - 1))) << 4));
                    unsigned w[4] = {v.x, v.y, v.z, v.w};
                    if constexpr (RESID) {
                        const uint4 rr = rvt[(2 * m + q) % 3][k];
                        const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                        for (int d = 0; d < 4; ++d) w[d] = w4_pack(w4_lo(w[d]) + w4_lo(rw[d]), w4_hi(w[d]) + w4_hi(rw[d]));
                    }
                    float s1 = 0.f, q1 = 0.f;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const bf16x2_t bv = __builtin_bit_cast(bf16x2_t, w[d]);
                        s1 = __builtin_amdgcn_fdot2_f32_bf16(bv, ones, s1, false);
                        q1 = __builtin_amdgcn_fdot2_f32_bf16(bv, bv, q1, false);
                    }
                    const bool ok = trow[m] && tcol0 + 16 * q + PPR * k < a.Wout && (!(DBG & 4) || w[0] == 0x12345678u);
                    ssum += ok ? s1 : 0.f; qsum += ok ? q1 : 0.f;
                    const u32x4_t o4 = {w[0], w[1], w[2], w[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(o4, orsrc, ok ? toffs[m] + (unsigned)(NRD * q + k) * cstep : 0xffffffffu, 0, IRE_ST_LINE);
                }
                if constexpr (RESID) { if (2 * m + q + 3 < 2 * C::MT) load_resid_pass(2 * m + q + 3, rvt[(2 * m + q) % 3]); }
            }
        stamp(9);
        // this lane's chunk cc_r over its read-backs; the other lanes with the same chunk sit NCHK apart
        if constexpr (NCHK == 8) { ssum = w4_ror_add<8>(ssum); qsum = w4_ror_add<8>(qsum); }
        ssum = w4_swap16_add(ssum); qsum = w4_swap16_add(qsum);
        ssum = w4_swap32_add(ssum); qsum = w4_swap32_add(qsum);
        if (l_e < NCHK) *reinterpret_cast<float2*>(red + red_par * (WAVES * 32) + (wave * NCHK + cc_r) * 2) = make_float2(ssum, qsum);
        st_img = it.img; st_tile = it.tile; st_cout0 = cout0; st_par = red_par; red_par ^= 1;
    };
    auto finish = [&](int s) {
        stamp(4);
        __syncthreads();          // stage barrier: buf[nxt] complete, buf[cur] free
        stamp(5);
        sq0 = sq1; sq1 = sq2; sq2 = cursor.next();
        if (sq2.kc == 0) item_offsets(sq2.it);       // (past the queue's end the cursor stays on the last stage: kc != 0)
        widx = widx == 2 ? 0 : widx + 1;
        par ^= 1;
    };

    // ---- prologue: the first stage's raw rows (registers) and the first two weight slabs (LDS-DMA) are requested BEFORE the folded
    // GroupNorm finalize -- neither needs its result, and its scratch (the first 8 KB of input tile 0) is not where they land -- so their
    // fetch rides under its reduction (round 4 timelines: finalize 3.2 us + 7 us of dependent prologue loads per launch, serialised);
    // behind it: coefficients, stage 0 -> tile 0, R <- stage 1
    {
        float* bl = reinterpret_cast<float*>(smem + C::MAIN_BYTES + C::RED_BYTES);
        if (tid < a.cout && tid < 256) bl[tid] = a.bias[tid];
        if constexpr (FP8) {
            float* sl = reinterpret_cast<float*>(smem + C::MAIN_BYTES + C::RED_BYTES + C::BIAS_BYTES + C::COEF_BYTES);
            if (tid < a.cout && tid < 256) sl[tid] = a.oscale[tid];
        }
        item_offsets(sq0.it);        // stages 0, 1, 2 belong to one item (nkc >= 8)
        load_stage(sq0, R);
        {
            const unsigned char* ws0 = wslab(sq0);
            const unsigned char* ws1 = wslab(sq1);
            const int wave_u = __builtin_amdgcn_readfirstlane(wave);
            for (int p = wave_u; p < C::W_CHUNKS / 64; p += WAVES) {
                w4_glds16(ws0 + ((size_t)p * 64 + lane) * 16, smem_lds + C::W_BASE + p * 1024);
                w4_glds16(ws1 + ((size_t)p * 64 + lane) * 16, smem_lds + C::W_BASE + C::W_BYTES + p * 1024);
            }
        }
        if (a.gn_stats) gn_fold(a, smem, cursor.first_img, cursor.last_img);     // GroupNorm finalize of the input tensor, folded in (gn_fold.hpp)
#ifdef IRE_W4_TL
        if (a.stamps && tid == 0) {
            a.stamps[(size_t)blockIdx.x * 32 + 2] = __builtin_amdgcn_s_memrealtime(); a.stamps[(size_t)blockIdx.x * 32 + 3] = __builtin_amdgcn_s_memtime();
            a.stamps[(size_t)blockIdx.x * 32 + 26] = __builtin_amdgcn_s_getreg((31 << 11) | 20); a.stamps[(size_t)blockIdx.x * 32 + 27] = (unsigned long long)my_items;
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // R and this wave's slab pieces have landed
#pragma unroll
        for (int i = 0; i < C::IN_ITERS; ++i) asm volatile("" : "+v"(R.v[i]));
        if constexpr (FUSED) stage_coeffs(fetch_coeffs(sq0));
        uint4* in0 = reinterpret_cast<uint4*>(smem);
#pragma unroll
        for (int i = 0; i < C::IN_ITERS; ++i) store_chunk(i, R, in0);
        load_stage(sq1, R);
        if constexpr (FUSED) cnext = fetch_coeffs(sq1);
    }
    __syncthreads();
    tl(2);
    // nkc is even (Cin/16 with Cin >= 128), so an item starts on an even stage and ends on an odd one
    for (int k = 0; k < my_items; ++k) {
        zero_acc(sq0.it.nb);
        // steady-state stages, then the item's last stage peeled with the epilogue: a conditional epilogue inside the
        // loop makes the allocator split live ranges of in-flight prefetch registers mid-stage (vmcnt(0) + v_mov)
        for (int kc = 0; kc + 1 < nkc; ++kc) {
            compute(std::false_type{});
            finish(s); ++s;
        }
        epi_prefetch();
        compute(std::true_type{});
        stamp(3);
        if constexpr (TEPI) epilogue_t(); else epilogue();
        finish(s); ++s;
        if constexpr (TEPI) flush_stats();      // after the stage barrier: every wave's chunk sums are in LDS (parity red_par ^ 1)
        tl(3 + (k < 8 ? k : 8));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may be in flight when the workgroup's LDS is released
    tl(12);
}

template <int NT, int WAVES, bool RESID, bool UPS, int DBG = 0, bool FUSED = false, bool FP8 = false>
void launch_w4(const ConvArgs& a, hipStream_t stream) {
    const int items = a.tiles_x * a.tiles_y * a.nimg * a.nblocks;
    const int cus = persistent_grid_cus();
    const int grid = items < cus ? items : cus;
    hipLaunchKernelGGL((conv_w4_kernel<NT, WAVES, RESID, UPS, DBG, FUSED, FP8>), dim3(grid), dim3(WAVES * 64), 0, stream, a);
    IRE_HIP(hipGetLastError());
}

}  // namespace

// a.nkc = Cin/16 stages, a.nblocks = cout/128, a.w = slabs [nblock][kc16][tap][c8][128][8], tiles of 16x32.
void conv_w4_launch(bool resid, const ConvArgs& a, hipStream_t stream) {
    // the engine launches the 8-wave form with the activation fused into the staging (the 4-wave form and the pre-activated input
    // it needed lost every A/B from round 1 to round 3; their launches were removed in round 4)
    if (a.ab == nullptr) fail(IRE_ERR_INTERNAL, "internal: conv_w4 needs the GroupNorm coefficients of its input");
    if (a.fp8) {
        if (a.oscale == nullptr) fail(IRE_ERR_INTERNAL, "internal: fp8 conv_w4 needs the per-channel scales");
        if (resid) launch_w4<128, 8, true, false, 0, true, true>(a, stream); else launch_w4<128, 8, false, false, 0, true, true>(a, stream);
        return;
    }
    if (a.w4_nt == 64) { if (resid) launch_w4<64, 8, true, false, 0, true>(a, stream); else launch_w4<64, 8, false, false, 0, true>(a, stream); }
    else { if (resid) launch_w4<128, 8, true, false, 0, true>(a, stream); else launch_w4<128, 8, false, false, 0, true>(a, stream); }
}

}  // namespace ire
