// conv_pc.hip -- the C = 32 and C = 64 ResBlock convolutions of RestoreNet-v0 and the 32 -> 3 head (17 launches, half of a
// step) as a PRODUCER / CONSUMER workgroup: 12 waves per CU = three per SIMD, gfx950.
//
// Why: conv_rb.hip's C = 32 stage is instruction-issue bound with two in-order waves per SIMD (profiles/r02: the VALU pipe is
// 62 % busy, yet the 2 x 4 700 issue cycles of the two waves add up to the measured 9 450 cycles per item: LDS, scalar and
// matrix issue never overlap another wave's VALU).  The work of an item is two different programs -- GroupNorm+FiLM+SiLU of
// the 18 x 34 x 32 input tile (transcendental-heavy VALU, global loads, LDS stores) and the 36 MFMAs + epilogue of a wave's
// 2 x 32 output pixels (LDS reads, matrix issue, stores) -- so here they run as two roles on every SIMD:
//   waves 0..7  consumers: fragments from LDS tile t & 1, 18 k-steps of 2 MFMAs, epilogue straight from the accumulators
//                (residual, bf16 stores, GroupNorm partials -- conv_rb.hip's, unchanged);
//   waves 8..11 producers: tile t + 1: raw bf16 rows (prefetched a whole item ahead into registers) -> y = silu(x A + B) in
//                packed f32 -> bf16 -> LDS tile (t + 1) & 1, then the loads of tile t + 2.
// One workgroup barrier per 32-channel stage (C = 32: one stage per item; C = 64: two, 64 couts per item = two n-tiles per
// wave).  Every wave gets 168 registers (three per SIMD); neither role needs more.  Weights (18 / 72 KB) stay in LDS for the
// whole kernel.  Tiles, accumulation order, epilogue arithmetic and the partials layout are those of
// conv_rb.hip's C = 32 variants: same results class (bf16 roundings identical, GroupNorm partials summed in the same order),
// row strips included.  Roofline: HBM for the residual variant (1.5 GB per launch), VALU issue otherwise.
#include "conv_mfma.hpp"
#include "gn_fold.hpp"
#include "persist.hpp"

#include <type_traits>

namespace ire {

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

#ifndef C3_RES_PRE
#define C3_RES_PRE -1     // residual groups requested before the MFMAs: -1 = per width (C = 32: 2, C = 64: 1); >= 0 forces one value (build-time A/B)
#endif
#ifndef C3_SCALAR
#define C3_SCALAR 1
#endif
#ifndef C3_PRIO
#define C3_PRIO 1     // measured: priority 1 for the producer waves +0.5 % (911 -> 916 img/s); 3 the same
#endif
#ifndef C3_TEPI
#define C3_TEPI 1    // C = 32 line-coalesced epilogue through a wave-private LDS transpose patch (0: the direct epilogue; build-time A/B).  With plain
                     // (write-back) stores it measured NEUTRAL (RB2 314.3 / 318.8 vs 314.8 / 315.3 us): the addresser's busy cycles fall, the kernel's time
                     // does not.  With its whole-line stores written THROUGH (IRE_ST_LINE) it pays on the residual kernel: same box, three times
                     // each, RB2 314.0 / 317.6 / 317.3 -> 302.3 / 302.0 / 302.5 us, RB1 unchanged, step 1014.8 / 1018.2 / 1016.8 -> 1023.9 / 1022.3 / 1022.0 img/s
#endif
#ifndef IRE_LD_ONCE32
#define IRE_LD_ONCE32 2      // the C = 32 line-coalesced residual loads are non-temporal (build-time A/B: RB2 304.5 / 303.4 -> 299.0 / 301.7 us, step +0.3 %)
#endif
#ifndef C3_TEPI64
#define C3_TEPI64 1  // C = 64: line-coalesced epilogue through the item's released input tile (0: the direct epilogue; build-time A/B)
#endif
#ifndef C3_PROD8
#define C3_PROD8 0   // 1: C = 32 ResBlock convs with EIGHT producer waves (16 waves per CU = four per SIMD, 128 registers each) instead of four
                     // (build-time A/B).  Stamps: at C = 32 the four producers need 6 300-8 700 ticks for an item's tile while the consumers'
                     // k-loop takes 2 300 and their epilogue ~3 000: the item waits for its producers.  Twice the producer waves: RB1 257.1 / 255.6
                     // vs 260.5 / 259.8 us, RB2 313.9 / 317.2 vs 319.9 / 315.0 (same box): nothing -- the SIMD's vector issue is what the producers'
                     // transform and the epilogues fill (85 % busy: profiles/r02_experiments.md), not a wave's own speed.  Off.
#endif
#ifndef C3_INTERIOR
#define C3_INTERIOR 0   // producers: tiles that touch no image border skip the zero-padding masks (two copies of the stage under a wave-uniform branch;
                        // as a select hipcc if-converts it and the count goes up): 4 of ~70 vector instructions per chunk
#endif
#ifndef C3_ABL
#define C3_ABL 0     // timing ablations (results wrong by design): 1 no transform, 2 no epilogue, 4 no MFMA loop, 8 no input loads, 16 no output stores, 32 no statistics, 64 no residual loads
#endif
constexpr int C3_CONS = 512, C3_PROD = 256, C3_THREADS = C3_CONS + C3_PROD;
// producer threads of an instantiation: 512 for the C = 32 ResBlock convs (C3_PROD8), 256 otherwise
constexpr int pc_prod(int C, bool head) { return (C3_PROD8 && C == 32 && !head) ? 512 : 256; }
constexpr int C3_TH = 16, C3_TW = 32, C3_IH = C3_TH + 2, C3_IW = C3_TW + 2;
constexpr int C3_IN_CHUNKS = C3_IH * C3_IW * 4;                                 // 2448 x 16 B (32 channels per pixel and stage)
constexpr int C3_P_ITERS = (C3_IN_CHUNKS + C3_PROD - 1) / C3_PROD;               // 10 chunks per producer thread and stage
constexpr int C3_IN_BYTES = C3_P_ITERS * C3_PROD * 16;                           // 40960: every chunk slot exists
template <int C> struct PcCfg {
    static constexpr int NT = C < 64 ? 32 : 64;                                 // couts per item
    static constexpr int NBLK = C / NT;                                         // n-blocks (items per tile)
    static constexpr int NKC = C / 32;                                          // 32-channel stages per item
    static constexpr int NTL = NT / 32;                                         // 32-cout n-tiles per wave
    static constexpr int W_STAGE = 36 * NT * 16;                                // [kk = tap*4 + c8][NT rows] x 16 B per stage
    static constexpr int W_STAGE_CHUNKS = 36 * NT;
    static constexpr int W_LDS = NKC * W_STAGE;                                 // the layer's weights stay in LDS for the whole kernel
    static constexpr int W_OFF = 2 * C3_IN_BYTES;
    static constexpr int BIAS_OFF = W_OFF + W_LDS;
    static constexpr int NCC = NT / 8;                                          // 16-B chunks of an item's couts per pixel
    static constexpr int RED_OFF = BIAS_OFF + C * 4;                            // 2 x [8 waves][NCC chunks][sA, qA, sB, qB]
    static constexpr int RED_HALF = 8 * NCC * 4;                                // floats
    static constexpr int COEF_IMGS = C == 32 ? 64 : 8;                          // images whose (A, B) the LDS table holds
    static constexpr int COEF_OFF = RED_OFF + 2 * RED_HALF * 4;
    static constexpr int HEAD_OFF = COEF_OFF + COEF_IMGS * C * 2 * 4;           // head: [8 waves][in | out][2 rows][96 B]
    static constexpr int PATCH_OFF = HEAD_OFF + (C == 32 ? 8 * 2 * 2 * 96 : 0);   // C = 32: [8 waves][32 pixels x 64 B] transpose patches of the line-coalesced epilogue
    static constexpr int CNT_OFF = PATCH_OFF + (C == 32 ? 8 * 2048 : 0);          // C = 64: the consumers' rendezvous counter (16 B)
    static constexpr int LDS = CNT_OFF + (C == 64 ? 16 : 0);
    static_assert(LDS <= 160 * 1024, "LDS");
};

__device__ __forceinline__ unsigned c3_pack(float a, float b) {
    f32x2_t f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ float c3_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float c3_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
template <int N> __device__ __forceinline__ float c3_ror_add(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float c3_swap16_add(float v) {       // see conv_rb.hip::rb_swap16_add
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}

// The per-stage barrier, spelled out: only LDS traffic has to be ordered (the tile a producer just wrote, the partials a consumer
// just wrote); global loads and stores stay in flight across it.  (hipcc's __syncthreads() is the same two instructions on
// gfx950 -- its workgroup-scope fence waits for lgkmcnt only -- measured equal; the asm form states the requirement.)
__device__ __forceinline__ void c3_glds16(const void* gsrc, unsigned lds_dst_uniform) {   // LDS-DMA, 1 KB per wave-instruction (conv_rb.hip)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ void c3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int C, bool RESID, bool HEAD>
__global__ __launch_bounds__(C3_CONS + pc_prod(C, HEAD)) void conv_pc_kernel(ConvArgs a) {
    using K = PcCfg<C>;
    constexpr int PROD = pc_prod(C, HEAD), THREADS = C3_CONS + PROD;
    constexpr int P_ITERS = (C3_IN_CHUNKS + PROD - 1) / PROD;        // chunks per producer thread and stage: 10 or 5 (P_ITERS * PROD * 16 == C3_IN_BYTES either way)
    static_assert(P_ITERS * PROD * 16 == C3_IN_BYTES, "tile slots");
    constexpr int NKC = K::NKC, NTL = K::NTL, NCC = K::NCC, NT = K::NT;
    static_assert(!HEAD || C == 32, "the head is a C = 32 layer");
    __shared__ __attribute__((aligned(16))) unsigned char smem[K::LDS];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // diagnostic build (IRE_RB_ABLATE=2, IRE_RB_STAMPS=<C>[r]): s_memtime stamps of consumer wave 0 and producer wave 8, per item
    int stamp_item = 0;
    auto stamp = [&](int k) {
#ifdef IRE_PC_TICKS
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (a.stamps && lane == 0 && (wave == 0 || wave == 8) && blockIdx.x < 8 && stamp_item < 64)
            a.stamps[(((size_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 64 + stamp_item) * 10 + k] = t;
#else
        (void)k;
#endif
    };

    // workgroup TIMELINE (diagnostic build -DIRE_W4_TL, tools/r04_tl.sh; layout as conv_w4.hip): entry (0), fold done (1), prologue done (2), exit (12)
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
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    PersistCursor cursor(a.tiles_x, a.tiles_y, a.nimg, K::NBLK, NKC);   // item = (tile, 64-cout block); a cursor step is one 32-channel stage
    const int n_items = cursor.my_items;
    if (n_items == 0) return;
    const int n_stages = n_items * NKC;
    tl(0);
    {   // the weights (all of them, or the first slab of the streamed form) leave for LDS by DMA BEFORE the folded GroupNorm finalize: their
        // fetch rides under its reduction (the finalize's scratch is the first 8 KB of the tile area, the weights sit behind both tiles)
        const unsigned char* ws = reinterpret_cast<const unsigned char*>(a.w);
        const unsigned wl = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + K::W_OFF;
        constexpr int PIECES = NKC * K::W_STAGE_CHUNKS / 64;                 // 1-KB pieces: 18 (C = 32), 72 (C = 64)
        static_assert((NKC * K::W_STAGE_CHUNKS) % 64 == 0, "whole wave-instructions");
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        for (int p = wv; p < PIECES; p += THREADS / 64) c3_glds16(ws + ((size_t)p * 64 + lane) * 16, wl + p * 1024);
    }
    // the folded GroupNorm finalize writes this workgroup's coefficient table straight into LDS (and to global memory for nobody's benefit here)
    if (a.gn_stats) gn_fold(a, smem, cursor.first_img, cursor.last_img, 512, reinterpret_cast<float2*>(smem + K::COEF_OFF));
    tl(1);

    {   // weights and bias stay in LDS for the whole kernel
        // resident weights: all of them; streamed: the slab of this workgroup's first stage (slot 0), the producers do the rest
        if (tid < C) reinterpret_cast<float*>(smem + K::BIAS_OFF)[tid] = a.bias[tid];
        if constexpr (C == 64) { if (tid == 0) *reinterpret_cast<unsigned*>(smem + K::CNT_OFF) = 0u; }
        // the GroupNorm+FiLM coefficients of the images this workgroup's items belong to (gn_fold just wrote them, or
        // gn_finalize_kernel did): the producers read them from LDS, so that their only VMEM traffic is the input stream
        const int nim = cursor.last_img - cursor.first_img + 1;                // <= COEF_IMGS (conv_pc_launch checks nimg)
        const float2* ab = a.ab + (size_t)cursor.first_img * C;
        float2* cd = reinterpret_cast<float2*>(smem + K::COEF_OFF);
        if (!a.gn_stats) for (int i = tid; i < nim * C; i += THREADS) cd[i] = ab[i];      // (gn_finalize_kernel ran: row strips, IRE_GN_FOLD=0)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's weight pieces are in LDS
    __syncthreads();
    tl(2);

    if (__builtin_amdgcn_readfirstlane(wave) >= 8) {
        // =============================== producers: waves 8..11 ===============================================================
#if C3_PRIO
        // (C = 32 ResBlock convs: the producers at the consumers' priority is faster -- RB1 278 -> 267 us; the head and C = 64 want them above)
        if (C3_PRIO != 1 || C != 32 || HEAD) asm volatile("s_setprio %0" :: "n"(C3_PRIO));            // the producers are the youngest waves of their SIMD: without this they issue in the consumers' leftover slots
#endif
        const int tp = tid - C3_CONS;
        const int c8 = tp & 3;                                   // this thread always stages the same 8-channel slice of a pixel
        // chunk i of this thread: halo-tile pixel p = (tp + PROD i) >> 2 -- constant for the whole kernel
        int rel[P_ITERS], lds_off[P_ITERS];
        unsigned pyx[P_ITERS];
#pragma unroll
        for (int i = 0; i < P_ITERS; ++i) {
            const int q = tp + i * PROD;
            const int p = q >> 2;
            const int py = p / C3_IW, px = p - py * C3_IW;       // py == 18: a slot past the tile (LDS padding), never valid
            pyx[i] = ((unsigned)py << 8) | (unsigned)px;
            rel[i] = (py * a.Win + px) * (2 * C) + c8 * 16;
            lds_off[i] = (p * 4 + (c8 ^ ((p >> 2) & 3))) * 16;
        }
        uint4 R[P_ITERS];
        unsigned okbits = 0;           // bit i: the chunk in R[i] lies inside the image (travels with the data; one register, not one mask per chunk)
        float cA[8], cB[8];
        // raw rows of stage `ls` (item, 32-channel slice), requested chunk by chunk (load_chunk), and its coefficients (from LDS)
        PersistStage ls = cursor.cur;
        auto load_chunk = [&](int i) __attribute__((always_inline)) {
            const PersistItem& lit = ls.it;
            const int oy1 = lit.ty * C3_TH - 1, ox1 = lit.tx * C3_TW - 1;
            const char* base = reinterpret_cast<const char*>(a.in0) + (size_t)lit.img * a.in_rows * a.Win * (2 * C) + ls.kc * 64;
            const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, a.in_rows * a.Win * (2 * C) - ls.kc * 64, 0x00020000);
            const int base_off = ((oy1 + a.in_row_off) * a.Win + ox1) * (2 * C);
            // tile-local bounds (wave-uniform): halo row py is readable iff y_lo <= py < y_hi, column px iff x_lo <= px < x_hi
            const int y_lo = max(a.iy_lo - oy1, 0), y_hi = min(a.iy_lo + a.iy_span - oy1, C3_IH);
            const int x_lo = max(-ox1, 0), x_hi = min(a.Win - ox1, C3_IW);
            const int py = (int)(pyx[i] >> 8), px = (int)(pyx[i] & 0xffu);
            const bool ok = (unsigned)(py - y_lo) < (unsigned)(y_hi - y_lo) && (unsigned)(px - x_lo) < (unsigned)(x_hi - x_lo);
            const unsigned off = ok ? (unsigned)(base_off + rel[i]) : 0xffffffffu;       // out of range: reads as zero
            if constexpr (C3_ABL & 8) { R[i] = make_uint4(off, 0x3f803f80u, i, 0x3f803f80u); }
            else {
                const u32x4_t lv = __builtin_amdgcn_raw_buffer_load_b128(irsrc, off, 0, IRE_LD_IN);
                R[i] = make_uint4(lv.x, lv.y, lv.z, lv.w);
            }
            okbits = (okbits & ~(1u << i)) | (ok ? (1u << i) : 0u);
        };
        auto load_coeffs = [&](const PersistStage& st) __attribute__((always_inline)) {      // (A, B) of this thread's 8 channels
            const float4* ab = reinterpret_cast<const float4*>(smem + K::COEF_OFF) + ((st.it.img - cursor.first_img) * C + st.kc * 32 + c8 * 8) / 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float4 v = ab[e]; cA[2 * e] = v.x; cB[2 * e] = v.y; cA[2 * e + 1] = v.z; cB[2 * e + 1] = v.w; }
        };
        // Two chunks (16 channel values in 8 packed pairs) move through the transform STAGE BY STAGE (16 independent v_exp, then
        // 16 independent v_rcp: a single in-order wave hides a transcendental's latency only behind its own independent work);
        // as soon as a pair has been transformed the same registers are reloaded with the NEXT stage's pair, so every load has
        // most of a stage's time to land and the first pair of the next transform is the oldest request.  Straight-line code on
        // purpose: with the reloads under a branch hipcc loses the order of the pending loads at the join and waits for all.
        auto transform_stage_v = [&](unsigned char* tile, auto interior_tag) __attribute__((always_inline)) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
            auto pair = [&](auto pp_tag) __attribute__((always_inline)) {
                constexpr int pp_ = decltype(pp_tag)::value, i = 2 * pp_;
                constexpr int NCH = i + 1 < P_ITERS ? 2 : 1;              // chunks in this group (the last group of an odd P_ITERS is a single chunk)
                constexpr int NW = 4 * NCH, NV = 8 * NCH;
                constexpr int i1 = NCH == 2 ? i + 1 : i;
                const unsigned wds[8] = {R[i].x, R[i].y, R[i].z, R[i].w, R[i1].x, R[i1].y, R[i1].z, R[i1].w};
                unsigned o[8];
                if constexpr (C3_ABL & 1) {
#pragma unroll
                    for (int w = 0; w < NW; ++w) o[w] = wds[w];
                } else {
#if C3_SCALAR
                    // plain f32 instructions, one channel each (build flag -fno-slp-vectorize keeps them so): beside the consumers'
                    // MFMAs a packed f32 instruction costs several times two plain ones (MI355X_MICROARCH.md, per-instruction
                    // constants; measured here: profiles/r02_experiments.md)
                    float y[16], e[16];
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        const int d = w & 3;
                        y[2 * w] = __builtin_fmaf(c3_lo(wds[w]), cA[2 * d], cB[2 * d]);
                        y[2 * w + 1] = __builtin_fmaf(c3_hi(wds[w]), cA[2 * d + 1], cB[2 * d + 1]);
                    }
#pragma unroll
                    for (int k = 0; k < NV; ++k) e[k] = y[k] * (-1.4426950408889634f);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < NV; ++k) e[k] = __builtin_amdgcn_exp2f(e[k]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < NV; ++k) e[k] = e[k] + 1.0f;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < NV; ++k) e[k] = __builtin_amdgcn_rcpf(e[k]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int w = 0; w < NW; ++w) o[w] = c3_pack(y[2 * w] * e[2 * w], y[2 * w + 1] * e[2 * w + 1]);
#else
                    static_assert(NCH == 2, "the packed-f32 form handles whole pairs");
                    f32x2_t y[8], e[8];
#pragma unroll
                    for (int w = 0; w < 8; ++w) {
                        // two channels at a time in packed f32: y = x A + B; silu(y) = y / (1 + 2^(-y log2 e))
                        const int d = w & 3;
                        const f32x2_t x = {c3_lo(wds[w]), c3_hi(wds[w])};
                        const f32x2_t A = {cA[2 * d], cA[2 * d + 1]}, B = {cB[2 * d], cB[2 * d + 1]};
                        y[w] = __builtin_elementwise_fma(x, A, B);
                        e[w] = y[w] * (-1.4426950408889634f);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int w = 0; w < 8; ++w) e[w] = f32x2_t{__builtin_amdgcn_exp2f(e[w].x), __builtin_amdgcn_exp2f(e[w].y)};
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int w = 0; w < 8; ++w) e[w] = e[w] + 1.0f;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int w = 0; w < 8; ++w) e[w] = f32x2_t{__builtin_amdgcn_rcpf(e[w].x), __builtin_amdgcn_rcpf(e[w].y)};
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int w = 0; w < 8; ++w) { const f32x2_t sv = y[w] * e[w]; o[w] = c3_pack(sv.x, sv.y); }
#endif
                }
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    uint4 ov = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
                    if constexpr (!INTERIOR) {
                        const unsigned m = (okbits >> (i + k)) & 1u ? 0xffffffffu : 0u;   // zero padding applies AFTER the activation
                        ov = make_uint4(o[4 * k] & m, o[4 * k + 1] & m, o[4 * k + 2] & m, o[4 * k + 3] & m);
                    }
                    *reinterpret_cast<uint4*>(tile + lds_off[i + k]) = ov;
                }
                load_chunk(i);
                if constexpr (NCH == 2) load_chunk(i + 1);
                __builtin_amdgcn_sched_barrier(0);
            };
            pair(std::integral_constant<int, 0>{}); pair(std::integral_constant<int, 1>{}); pair(std::integral_constant<int, 2>{});
            if constexpr (P_ITERS > 6) { pair(std::integral_constant<int, 3>{}); pair(std::integral_constant<int, 4>{}); }
        };
        // chunk slots past the tile (halo-tile row 18: only the last chunk of the upper threads) are never read: they do not count against "interior"
        unsigned past_last = 0u;
        {
            const int q = tp + (P_ITERS - 1) * PROD;
            past_last = (q >> 2) >= C3_IH * C3_IW ? (1u << (P_ITERS - 1)) : 0u;
        }
        auto transform_stage = [&](unsigned char* tile) __attribute__((always_inline)) {
            if constexpr (C3_INTERIOR) {
                if (__builtin_amdgcn_ballot_w64((okbits | past_last) != (1u << P_ITERS) - 1u) == 0) { transform_stage_v(tile, std::true_type{}); return; }
            }
            transform_stage_v(tile, std::false_type{});
        };
        // stage 0 -> R; then every transform reloads R with the stage after (past the last stage the cursor stays on it: a
        // redundant reload of rows that are never used).  `ps` = the stage whose rows are in R = the stage being produced.
#pragma unroll
        for (int i = 0; i < P_ITERS; ++i) load_chunk(i);
        load_coeffs(ls);
        ls = cursor.next();
        transform_stage(smem);
        c3_barrier();                                           // tile (and slab) of stage 0 are staged
        for (int t = 0; t < n_stages; ++t) {
            // tile / slab of stage t + 1 (for t + 1 == n_stages: the last stage again, into the slots nobody reads any more)
            stamp_item = t / NKC;
            stamp(3 * (t % NKC) + 0);
            load_coeffs(ls);
            ls = cursor.next();
            transform_stage(smem + ((t + 1) & 1) * C3_IN_BYTES);   // the consumers finished reading this slot before the last barrier
            stamp(3 * (t % NKC) + 1);
            c3_barrier();
            stamp(3 * (t % NKC) + 2);
        }
        return;
    }

    // =================================== consumers: waves 0..7 ======================================================================
    // per-lane LDS offsets of the 18 (row m, tap) pixel fragments: p = (2 wave + m + ky) IW + r + kx, 16-B chunk index
    // p*4 + (c8 ^ ((p >> 2) & 3)), c8 = 2 (k-step & 1) + h  (the k-step parity toggles bit 5)
    int a_off[2][9];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int p = (wave * 2 + m + ky) * C3_IW + r + kx;
            a_off[m][tap] = (p * 4 + (h ^ ((p >> 2) & 3))) * 16;
        }
    const unsigned char* wb = smem + K::W_OFF + (h * NT + r) * 16;        // + slab offset + ((tap*4 + 2 cp) * NT + j*32) * 16
    const float* bias_lds = reinterpret_cast<const float*>(smem + K::BIAS_OFF);
    f32x16_t bias_acc;                    // C = 32: the C operand of every item's first MFMAs (the accumulators start at the bias, no moves)
    if constexpr (C == 32) {
#pragma unroll
        for (int i = 0; i < 16; ++i) bias_acc[i] = bias_lds[16 * (i >> 3) + 8 * h + (i & 7)];     // permuted slab rows
    }
    float* red_base = reinterpret_cast<float*>(smem + K::RED_OFF);
    int st_img = -1, st_tile = 0, st_nb = 0, st_par = 0, red_par = 0;
    auto flush_stats = [&]() {              // partials of the item finished before the last barrier: its NT / (C / 8) groups
        if (HEAD || st_img < 0) return;
        constexpr int G = C / 8, NGL = NT / G, CPG = G / 8;       // couts per group, groups per item, 16-B chunks per group (C >= 64)
        if (tid < NGL) {
            const float* red = red_base + st_par * K::RED_HALF;
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                if constexpr (C == 32) {           // groups of 4: a 16-B chunk holds two groups (sA, qA | sB, qB)
                    const float* d = red + (w * NCC + (tid >> 1)) * 4 + 2 * (tid & 1);
                    s += d[0]; q += d[1];
                } else {                           // groups of 8 / 16 / 32: CPG whole chunks each
#pragma unroll
                    for (int k = 0; k < CPG; ++k) {
                        const float* d = red + (w * NCC + tid * CPG + k) * 4;
                        s += d[0] + d[2]; q += d[1] + d[3];
                    }
                }
            }
            float* st = a.stats + (((size_t)st_img * tiles_per_img + st_tile) * 8 + (st_nb * NT) / G + tid) * 2;
            st[0] = s; st[1] = q;
        }
        st_img = -1;
    };

    f32x16_t acc[2][NTL];
    uint4 erv[2 * NTL][2];
    // residual groups requested before the MFMAs, the rest after the k-loop.  Same box, C = 64 with the line-coalesced epilogue: 0 / 1 / 2 / 3 / 4
    // = 218.4 / 216.9 / 223.5 / 294.8 / 299.6 us (from 3 on the consumer spills into its k-loop); C = 32: 326.8 / 329.4 / 320.3
    constexpr int RES_WANT = C3_RES_PRE >= 0 ? C3_RES_PRE : (C == 64 ? 1 : 2);
    constexpr int RES_PRE = RES_WANT < 2 * NTL ? RES_WANT : 2 * NTL;
    unsigned eoffs[2];
    bool einb[2];
    // C = 32 line-coalesced epilogue (TEPI, off by default -- see C3_TEPI): the accumulator layout (lane = pixel) makes every lane
    // of a quad address a different pixel, and the texture addresser coalesces only within a quad: a 16-B-per-lane store of 32
    // pixels x 32 B costs ~66 of its cycles, a residual load of that shape ~115, against 16 for quad-contiguous addresses (PMC: TA
    // busy 72 % of the C = 32 residual kernel, profiles/r03_experiments.md).  With TEPI a wave transposes a pixel row through its
    // own 2-KB LDS patch (XOR-swizzled: conflict-free both ways) so that four consecutive lanes hold one pixel's 64 B and every
    // global instruction of the epilogue is 1 KB contiguous.  Correct (same tests), and neutral in time: the addresser was busy,
    // not limiting.
    constexpr bool TEPI = C3_TEPI && C == 32 && !HEAD;
    // C = 64 (TEPI64): the same transposition, with the patches in the input tile the item's LAST stage has just finished with (the
    // kernel has no other 32 KB of LDS).  That tile belongs to the consumers until the stage barrier -- the producers are writing the
    // other one -- but every consumer wave must be past its last fragment read first: the consumers meet on an LDS counter (no
    // workgroup barrier: the producers must not wait here).  Why it pays at C = 64 and not at C = 32: here the item's stores sit on
    // the critical path -- stamps (profiles/r03_experiments.md): the eight consumer waves issue their 64 partial-line stores
    // together, ~66 addresser cycles each = 4 200 of the epilogue's 5 000 ticks, and the producers wait 4 000 ticks at the barrier.
    constexpr bool TEPI64 = C3_TEPI64 && C == 64 && !HEAD;
    unsigned toffs[2];            // byte offset of (read-back pixel lane >> 2 of row m, chunk lane & 3) in the output image
    bool trow[2];
    int tcol = 0;
    unsigned hin[2] = {0u, 0u};           // head: this lane's dword of the two image rows
    size_t hoff[2] = {0, 0};
    bool hok[2] = {false, false};
    PersistStage cs = cursor.cur;
    int stage_no = 0;
    c3_barrier();                                               // tile of stage 0 is staged
    for (int t = 0; t < n_items; ++t) {
        const PersistItem it = cs.it;
        stamp_item = t;
        stamp(0);
        flush_stats();
        {   // output / residual offsets of this lane's two pixels; the residual rows are requested before the MFMAs
            const int oyb = it.ty * C3_TH + wave * 2, ox = it.tx * C3_TW + r;
            const bool colok = ox < a.Wout;
            const int oxc = min(ox, a.Wout - 1);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int oy = oyb + m;
                einb[m] = colok && oy < a.Hout;
                eoffs[m] = ((unsigned)((min(oy, a.Hout - 1) * a.Wout + oxc) * C + it.nb * NT) << 1) + (unsigned)(h * 16);
            }
            if constexpr (TEPI || TEPI64) {
                constexpr int LPP = C / 8;                  // lanes per pixel in the read-back layout: 4 (C = 32) / 8 (C = 64)
                tcol = it.tx * C3_TW + lane / LPP;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int oy = oyb + m;
                    trow[m] = oy < a.Hout;
                    toffs[m] = ((unsigned)((min(oy, a.Hout - 1) * a.Wout + tcol) * C + 8 * (lane % LPP)) << 1);
                }
            }
            if constexpr (RESID && TEPI64 && !(C3_ABL & 64)) {
                // the residual rows in the read-back layout (1 KB = 8 pixels per request): RES_PRE of the 4 quarter-rows per row before the MFMAs
                char* rbase = const_cast<char*>(reinterpret_cast<const char*>(a.resid)) + (size_t)it.img * a.Hout * a.Wout * (2 * C);
                const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(rbase, 0, a.Hout * a.Wout * (2 * C), 0x00020000);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int k = 0; k < RES_PRE; ++k) {
                        const bool ok = trow[m] && tcol + 8 * k < a.Wout;
                        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ok ? toffs[m] + (unsigned)(8 * k * 2 * C) : 0xffffffffu, 0, IRE_LD_ONCE32);
                        erv[k][m] = make_uint4(v.x, v.y, v.z, v.w);
                    }
            } else
            if constexpr (RESID && TEPI && !(C3_ABL & 64)) {
                // the residual rows in the read-back layout: two 1-KB requests per row, before the MFMAs
                char* rbase = const_cast<char*>(reinterpret_cast<const char*>(a.resid)) + (size_t)it.img * a.Hout * a.Wout * (2 * C);
                const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(rbase, 0, a.Hout * a.Wout * (2 * C), 0x00020000);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const bool ok = trow[m] && tcol + 16 * k < a.Wout;
                        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ok ? toffs[m] + (unsigned)(16 * k * 2 * C) : 0xffffffffu, 0, IRE_LD_ONCE);
                        erv[k][m] = make_uint4(v.x, v.y, v.z, v.w);
                    }
            } else if constexpr (RESID && (C3_ABL & 64)) {
#pragma unroll
                for (int g = 0; g < 2 * NTL; ++g)
#pragma unroll
                    for (int m = 0; m < 2; ++m) erv[g][m] = make_uint4(eoffs[m], g, 0x3f803f80u, m);
            } else if constexpr (RESID) {
                const char* rbase = reinterpret_cast<const char*>(a.resid) + (size_t)it.img * a.Hout * a.Wout * (2 * C);
                // row by row, a pixel's 16-B pieces back to back: the requests for the two halves of a 64-B sector (and the
                // sectors of a line) leave the CU together
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int g = 0; g < RES_PRE; ++g) erv[g][m] = *reinterpret_cast<const uint4*>(rbase + eoffs[m] + (unsigned)(g * 32));
            }
            if constexpr (HEAD) {
                // the wave's two rows of the ORIGINAL image, 32 pixels = 96 contiguous bytes each: 24 dwords per row, requested
                // before the MFMAs (a row starts dword-aligned: W is a multiple of 8 and the tile column of 32)
                const int vbytes = 3 * min(C3_TW, a.Wout - it.tx * C3_TW);          // a multiple of 24
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int oy = oyb + m;
                    hok[m] = oy < a.Hout && 4 * lane < vbytes;
                    hoff[m] = (((size_t)it.img * a.Hout + min(oy, a.Hout - 1)) * a.Wout + it.tx * C3_TW) * 3 + 4 * (size_t)min(lane, 23);
                    hin[m] = hok[m] ? *reinterpret_cast<const unsigned*>(a.u8_in + hoff[m]) : 0u;
                }
            }
        }
        if constexpr (C != 32) {          // the accumulators start at the bias: straight from the LDS table (permuted slab rows)
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias_lds + it.nb * NT + j * 32 + 16 * (q >> 1) + 8 * h + 4 * (q & 1));
#pragma unroll
                    for (int m = 0; m < 2; ++m) { acc[m][j][4 * q + 0] = bv.x; acc[m][j][4 * q + 1] = bv.y; acc[m][j][4 * q + 2] = bv.z; acc[m][j][4 * q + 3] = bv.w; }
                }
        }
        // NKC stages of 18 k-steps (tap, channel half): two pixel fragments + NTL weight fragments, read one k-step ahead
#pragma unroll NKC <= 2 ? NKC : 1
        for (int kc = 0; kc < NKC; ++kc) {
          // PAR: the stage's LDS tile as a compile-time constant where that pays (the head: two copies of the k-loop under a
          // wave-uniform branch on the item's parity, 219 -> 205 us; C = 64: the unrolled stage index) -- the tile base then rides in
          // the ds_read offset field instead of one v_add per fragment address (24 per item).  Run-time (-1) for the C = 32
          // ResBlock convs (two copies measured 3 us slower there) and the streamed forms
          auto stage_body = [&](auto par_tag) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_tag)::value;
            const int par = PAR >= 0 ? PAR : (stage_no & 1);
            const unsigned char* ib = smem + par * C3_IN_BYTES;
            const unsigned char* wk = wb + kc * K::W_STAGE;
            bf16x8_t af[2][2], bf[2][NTL];
            auto read_k = [&](int g, bf16x8_t (&pa)[2], bf16x8_t (&pw)[NTL]) __attribute__((always_inline)) {
                const int tap = g >> 1, cp = g & 1;
#pragma unroll
                for (int m = 0; m < 2; ++m) pa[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ib + (a_off[m][tap] ^ (cp << 5))));
#pragma unroll
                for (int j = 0; j < NTL; ++j) pw[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wk + ((tap * 4 + 2 * cp) * NT + j * 32) * 16));
            };
            read_k(0, af[0], bf[0]);
            if constexpr (!(C3_ABL & 4))
#pragma unroll
            for (int g = 0; g < 18; ++g) {
                if (g + 1 < 18) read_k(g + 1, af[(g + 1) & 1], bf[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int j = 0; j < NTL; ++j) {
                        if constexpr (C == 32) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[g & 1][j], af[g & 1][m], (kc == 0 && g == 0) ? bias_acc : acc[m][j], 0, 0, 0);
                        else acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[g & 1][j], af[g & 1][m], acc[m][j], 0, 0, 0);   // D[cout][pixel]
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
          };
            if constexpr (NKC == 1 && HEAD) { if (stage_no & 1) stage_body(std::integral_constant<int, 1>{}); else stage_body(std::integral_constant<int, 0>{}); }
            else if constexpr (NKC == 2) { if (kc & 1) stage_body(std::integral_constant<int, 1>{}); else stage_body(std::integral_constant<int, 0>{}); }   // (an item starts on an even stage)
            else stage_body(std::integral_constant<int, -1>{});
            ++stage_no;
            stamp(kc == 0 ? 1 : 3);
            if (kc + 1 < NKC) { cs = cursor.next(); c3_barrier(); stamp(2); }      // the item's next stage: its tile is staged, this one is free
        }
        if constexpr (RESID && TEPI64 && RES_PRE < 2 * NTL) {
            char* rbase = const_cast<char*>(reinterpret_cast<const char*>(a.resid)) + (size_t)it.img * a.Hout * a.Wout * (2 * C);
            const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(rbase, 0, a.Hout * a.Wout * (2 * C), 0x00020000);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int k = RES_PRE; k < 2 * NTL; ++k) {
                    const bool ok = trow[m] && tcol + 8 * k < a.Wout;
                    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ok ? toffs[m] + (unsigned)(8 * k * 2 * C) : 0xffffffffu, 0, 0);
                    erv[k][m] = make_uint4(v.x, v.y, v.z, v.w);
                }
        } else if constexpr (RESID && RES_PRE < 2 * NTL) {
            // the rest of the residual rows: requested once the fragment registers are free (all of them before the MFMAs did not
            // fit 168 registers: 18 were spilled, and a scratch reload in the k-loop waits, in order, for the residual rows first);
            // the epilogue reaches them ~200 instructions later
            const char* rbase = reinterpret_cast<const char*>(a.resid) + (size_t)it.img * a.Hout * a.Wout * (2 * C);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = RES_PRE; g < 2 * NTL; ++g) erv[g][m] = *reinterpret_cast<const uint4*>(rbase + eoffs[m] + (unsigned)(g * 32));
        }
        // ---- epilogue (conv_rb.hip's): accumulator i of n-tile j of lane (r, h) is cout 32 j + 16 (i >> 3) + 8 h + (i & 7) -----------
        if constexpr (HEAD) {
            // out = clamp(round(in + y)) for the 3 channels: the image bytes travel as dwords (one load and one store instruction
            // per row instead of three byte-wide ones each) and are re-sliced per pixel through a 192-byte wave-private LDS patch
            unsigned char* hp = smem + K::HEAD_OFF + wave * (2 * 2 * 96);
#pragma unroll
            for (int m = 0; m < 2; ++m)
                if (lane < 24) *reinterpret_cast<unsigned*>(hp + m * 96 + 4 * lane) = hin[m];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if (h == 0) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float vv = (float)hp[m * 96 + 3 * r + c] + acc[m][0][c];          // the bias is already in the accumulator
                        vv = fminf(fmaxf(vv, 0.f), 255.f);
                        hp[192 + m * 96 + 3 * r + c] = (unsigned char)(int)floorf(vv + 0.5f);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
                if (hok[m]) *reinterpret_cast<unsigned*>(a.u8_out + hoff[m]) = *reinterpret_cast<const unsigned*>(hp + 192 + m * 96 + 4 * min(lane, 23));
        } else if constexpr (C3_ABL & 2) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < NTL; ++j) asm volatile("" :: "v"(acc[m][j]));       // every accumulator stays live: no MFMA may be optimised away
        } else if constexpr (TEPI64) {
            char* obase = reinterpret_cast<char*>(a.out) + (size_t)it.img * a.Hout * a.Wout * (2 * C);
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, a.Hout * a.Wout * (2 * C), 0x00020000);
            const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
            float* redw = red_base + red_par * K::RED_HALF;
            {   // consumers' rendezvous: the item's last tile (an item ends on an odd stage: tile 1) is free once all 8 waves are here
                volatile unsigned* cnt = reinterpret_cast<volatile unsigned*>(smem + K::CNT_OFF);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(const_cast<unsigned*>(cnt), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned target = 8u * (unsigned)(t + 1);
                while ((unsigned)__builtin_amdgcn_readfirstlane((int)*cnt) < target) __builtin_amdgcn_s_sleep(1);
            }
            unsigned char* patch = smem + C3_IN_BYTES + wave * 4096;       // [32 pixels][128 B], chunk index XOR (pixel & 7)
            const int wsw = r & 7;                                         // writer: pixel r, chunks 2 g + h
            const int pq = lane >> 3, cq = lane & 7;                       // reader: pixel pq + 8 k, chunk cq = couts 8 cq .. 8 cq + 7 = GroupNorm group cq
            float sA = 0.f, qA = 0.f;
            // Order of the passes: quarter-rows k = 0, 1 of BOTH rows first, then k = 2, 3 -- the residual quarter-rows k >= RES_PRE were
            // requested only after the k-loop (registers), so they get the first half's arithmetic and stores to land.  A row goes
            // through the patch once per half (4 more LDS stores per row, against an exposed memory latency per item).
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int g = 0; g < 2 * NTL; ++g) {
                    const f32x16_t& c = acc[m][g >> 1];
                    const int pp = g & 1;
                    const u32x4_t wv = {c3_pack(c[8 * pp + 0], c[8 * pp + 1]), c3_pack(c[8 * pp + 2], c[8 * pp + 3]),
                                        c3_pack(c[8 * pp + 4], c[8 * pp + 5]), c3_pack(c[8 * pp + 6], c[8 * pp + 7])};
                    *reinterpret_cast<u32x4_t*>(patch + r * 128 + (((2 * g + h) ^ wsw) << 4)) = wv;
                }
#pragma unroll
                for (int k = 2 * half; k < 2 * half + 2; ++k) {
                    const int p = pq + 8 * k;
                    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(patch + p * 128 + ((cq ^ (p & 7)) << 4));
                    unsigned w[4] = {v.x, v.y, v.z, v.w};
                    if constexpr (RESID) {
                        const uint4 rr = erv[k][m];
                        const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                        for (int d = 0; d < 4; ++d) w[d] = c3_pack(c3_lo(w[d]) + c3_lo(rw[d]), c3_hi(w[d]) + c3_hi(rw[d]));
                    }
                    const bool ok = trow[m] && tcol + 8 * k < a.Wout;
                    const float mf = ok ? 1.f : 0.f;
                    float ts0 = 0.f, tq0 = 0.f;
                    if constexpr (!(C3_ABL & 32))
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const bf16x2_t bv = __builtin_bit_cast(bf16x2_t, w[d]);
                        ts0 = __builtin_amdgcn_fdot2_f32_bf16(bv, ones, ts0, false); tq0 = __builtin_amdgcn_fdot2_f32_bf16(bv, bv, tq0, false);
                    }
                    sA = __builtin_fmaf(ts0, mf, sA); qA = __builtin_fmaf(tq0, mf, qA);
                    const u32x4_t o4 = {w[0], w[1], w[2], w[3]};
                    if constexpr (C3_ABL & 16) asm volatile("" :: "v"(o4));
                    else __builtin_amdgcn_raw_buffer_store_b128(o4, orsrc, ok ? toffs[m] + (unsigned)(8 * k * 2 * C) : 0xffffffffu, 0, IRE_ST_LINE);
                }
            }
            if constexpr (!(C3_ABL & 32)) {
                // the eight lanes that share a chunk sit 8 apart: one rotation within the row of 16, then the rows
                float t2[2] = {sA, qA};
#pragma unroll
                for (int k = 0; k < 2; ++k) t2[k] = c3_ror_add<8>(t2[k]);
#pragma unroll
                for (int k = 0; k < 2; ++k) t2[k] = c3_swap16_add(t2[k]);
#pragma unroll
                for (int k = 0; k < 2; ++k) { float x = t2[k], y = t2[k]; asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y)); t2[k] = x + y; }
                if (lane < 8) *reinterpret_cast<float4*>(redw + (wave * NCC + lane) * 4) = make_float4(t2[0], t2[1], 0.f, 0.f);
            }
            st_img = it.img; st_tile = it.tile; st_nb = it.nb; st_par = red_par; red_par ^= 1;
        } else if constexpr (TEPI) {
            char* obase = reinterpret_cast<char*>(a.out) + (size_t)it.img * a.Hout * a.Wout * (2 * C);
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, a.Hout * a.Wout * (2 * C), 0x00020000);
            const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
            float* redw = red_base + red_par * K::RED_HALF;
            unsigned char* patch = smem + K::PATCH_OFF + wave * 2048;
            const int wsw = (r >> 1) & 3;                                  // writer: pixel r, chunks 2 pp + h
            const int pq = lane >> 2, cq = lane & 3;                       // reader: pixel pq + 16 k, chunk cq (couts 8 cq .. 8 cq + 7)
            float sA = 0.f, qA = 0.f, sB = 0.f, qB = 0.f;                  // groups 2 cq (couts 0..3 of the chunk) and 2 cq + 1
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f32x16_t& c = acc[m][0];
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const u32x4_t wv = {c3_pack(c[8 * pp + 0], c[8 * pp + 1]), c3_pack(c[8 * pp + 2], c[8 * pp + 3]),
                                        c3_pack(c[8 * pp + 4], c[8 * pp + 5]), c3_pack(c[8 * pp + 6], c[8 * pp + 7])};
                    *reinterpret_cast<u32x4_t*>(patch + r * 64 + (((2 * pp + h) ^ wsw) << 4)) = wv;
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int p = pq + 16 * k;
                    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(patch + p * 64 + ((cq ^ ((p >> 1) & 3)) << 4));
                    unsigned w[4] = {v.x, v.y, v.z, v.w};
                    if constexpr (RESID) {
                        const uint4 rr = erv[k][m];
                        const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                        for (int d = 0; d < 4; ++d) w[d] = c3_pack(c3_lo(w[d]) + c3_lo(rw[d]), c3_hi(w[d]) + c3_hi(rw[d]));
                    }
                    const bool ok = trow[m] && tcol + 16 * k < a.Wout;
                    const float mf = ok ? 1.f : 0.f;
                    float ts0 = 0.f, tq0 = 0.f, ts1 = 0.f, tq1 = 0.f;
                    if constexpr (!(C3_ABL & 32))
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const bf16x2_t bv = __builtin_bit_cast(bf16x2_t, w[d]);
                        if (d < 2) { ts0 = __builtin_amdgcn_fdot2_f32_bf16(bv, ones, ts0, false); tq0 = __builtin_amdgcn_fdot2_f32_bf16(bv, bv, tq0, false); }
                        else { ts1 = __builtin_amdgcn_fdot2_f32_bf16(bv, ones, ts1, false); tq1 = __builtin_amdgcn_fdot2_f32_bf16(bv, bv, tq1, false); }
                    }
                    sA = __builtin_fmaf(ts0, mf, sA); qA = __builtin_fmaf(tq0, mf, qA);
                    sB = __builtin_fmaf(ts1, mf, sB); qB = __builtin_fmaf(tq1, mf, qB);
                    const u32x4_t o4 = {w[0], w[1], w[2], w[3]};
                    if constexpr (C3_ABL & 16) asm volatile("" :: "v"(o4));
                    else __builtin_amdgcn_raw_buffer_store_b128(o4, orsrc, ok ? toffs[m] + (unsigned)(16 * k * 2 * C) : 0xffffffffu, 0, IRE_ST_LINE);
                }
            }
            if constexpr (!(C3_ABL & 32)) {
                // the sixteen lanes that share a chunk sit 4 apart: two rotations within the row of 16, then the rows
                float t4[4] = {sA, qA, sB, qB};
#pragma unroll
                for (int k = 0; k < 4; ++k) t4[k] = c3_ror_add<4>(t4[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) t4[k] = c3_ror_add<8>(t4[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) t4[k] = c3_swap16_add(t4[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) { float x = t4[k], y = t4[k]; asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y)); t4[k] = x + y; }
                if (lane < 4) *reinterpret_cast<float4*>(redw + (wave * NCC + lane) * 4) = make_float4(t4[0], t4[1], t4[2], t4[3]);
            }
            st_img = it.img; st_tile = it.tile; st_nb = it.nb; st_par = red_par; red_par ^= 1;
        } else {
            char* obase = reinterpret_cast<char*>(a.out) + (size_t)it.img * a.Hout * a.Wout * (2 * C);
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, a.Hout * a.Wout * (2 * C), 0x00020000);
            const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
            float* redw = red_base + red_par * K::RED_HALF;
            const float mf[2] = {einb[0] ? 1.f : 0.f, einb[1] ? 1.f : 0.f};      // a pixel outside the image counts for nothing
            // 16-cout group g = (j, pp): this lane's 8 couts 32 j + 16 pp + 8 h .. + 7 = 16-B chunk cc = 4 j + 2 pp + h of its pixels
            // C = 32 (GroupNorm groups of 4): vs/vq[2 g + e] = (sum, squares) of the chunk's couts 4 e .. 4 e + 3 -> 8 values
            // C = 64 (groups of 8):           vs/vq[g] = of the whole chunk                                    -> 8 values
            float vs[4], vq[4];
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int g = 2 * j + pp;
                float sA = 0.f, qA = 0.f, sB = 0.f, qB = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const f32x16_t& c = acc[m][j];
                    unsigned w[4] = {c3_pack(c[8 * pp + 0], c[8 * pp + 1]), c3_pack(c[8 * pp + 2], c[8 * pp + 3]),
                                     c3_pack(c[8 * pp + 4], c[8 * pp + 5]), c3_pack(c[8 * pp + 6], c[8 * pp + 7])};
                    if constexpr (RESID) {
                        const uint4 rr = erv[g][m];
                        const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                        for (int d = 0; d < 4; ++d) w[d] = c3_pack(c3_lo(w[d]) + c3_lo(rw[d]), c3_hi(w[d]) + c3_hi(rw[d]));
                    }
                    float ts0 = 0.f, tq0 = 0.f, ts1 = 0.f, tq1 = 0.f;
                    if constexpr (!(C3_ABL & 32))
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const bf16x2_t wv = __builtin_bit_cast(bf16x2_t, w[d]);
                        if (d < 2 || C != 32) { ts0 = __builtin_amdgcn_fdot2_f32_bf16(wv, ones, ts0, false); tq0 = __builtin_amdgcn_fdot2_f32_bf16(wv, wv, tq0, false); }
                        else { ts1 = __builtin_amdgcn_fdot2_f32_bf16(wv, ones, ts1, false); tq1 = __builtin_amdgcn_fdot2_f32_bf16(wv, wv, tq1, false); }
                    }
                    sA = __builtin_fmaf(ts0, mf[m], sA); qA = __builtin_fmaf(tq0, mf[m], qA);
                    if constexpr (C == 32) { sB = __builtin_fmaf(ts1, mf[m], sB); qB = __builtin_fmaf(tq1, mf[m], qB); }
                    const u32x4_t wv4 = {w[0], w[1], w[2], w[3]};
                    if constexpr (C3_ABL & 16) asm volatile("" :: "v"(wv4));
                    else __builtin_amdgcn_raw_buffer_store_b128(wv4, orsrc, einb[m] ? eoffs[m] + (unsigned)(g * 32) : 0xffffffffu, 0, IRE_ST_PART);
                }
                if constexpr (C == 32) { vs[2 * pp] = sA; vq[2 * pp] = qA; vs[2 * pp + 1] = sB; vq[2 * pp + 1] = qB; }
                else { vs[g] = sA; vq[g] = qA; }
            }
            // Sum of each of the 8 values over the 32 lanes of a half, TRANSPOSING for the first two steps: a lane keeps half of
            // its values and hands the other half to its partner (lane ^ 1, then lane ^ 2), so 8 -> 4 -> 2 values per lane; those two
            // take the plain steps over lane bits 2, 3 and 4.  28 instead of 56 cross-lane instructions; fixed order.  Lane l of a
            // half ends with (kind = b0: sum / sum of squares) of the values 2 b1 and 2 b1 + 1.
            if constexpr (!(C3_ABL & 32)) {
                const bool b0 = lane & 1, b1 = lane & 2;
                auto xch = [&](float keep, float give, auto ctrl_tag) __attribute__((always_inline)) -> float {
                    const int gg = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), decltype(ctrl_tag)::value, 0xf, 0xf, false);
                    return keep + __builtin_bit_cast(float, gg);
                };
                float u[4], t2[2];
#pragma unroll
                for (int k = 0; k < 4; ++k) u[k] = xch(b0 ? vq[k] : vs[k], b0 ? vs[k] : vq[k], std::integral_constant<int, 0xb1>{});     // quad_perm [1,0,3,2]: lane ^ 1
#pragma unroll
                for (int k = 0; k < 2; ++k) t2[k] = xch(b1 ? u[2 + k] : u[k], b1 ? u[k] : u[2 + k], std::integral_constant<int, 0x4e>{});   // quad_perm [2,3,0,1]: lane ^ 2
                t2[0] = c3_ror_add<4>(t2[0]); t2[1] = c3_ror_add<4>(t2[1]);
                t2[0] = c3_ror_add<8>(t2[0]); t2[1] = c3_ror_add<8>(t2[1]);
                t2[0] = c3_swap16_add(t2[0]); t2[1] = c3_swap16_add(t2[1]);
                if ((lane & 28) == 0) {
                    if constexpr (C == 32) {      // values 2 b1 + e = (chunk pp = b1, half-chunk e): red[wave][cc = 2 pp + h][2 e + kind]
                        float* d = redw + (wave * NCC + (b1 ? 2 : 0) + h) * 4 + (b0 ? 1 : 0);
                        d[0] = t2[0]; d[2] = t2[1];
                    } else {                      // values g = 2 b1 + e = (j = b1, pp = e): red[wave][cc = 4 j + 2 pp + h][kind]
                        float* d = redw + (wave * NCC + (b1 ? 4 : 0) + h) * 4 + (b0 ? 1 : 0);
                        d[0] = t2[0]; d[8] = t2[1];
                        d[2] = 0.f; d[10] = 0.f;  // (the second pair of a chunk's slot is unused at C = 64: flush adds it)
                    }
                }
            }
            st_img = it.img; st_tile = it.tile; st_nb = it.nb; st_par = red_par; red_par ^= 1;
        }
        stamp(NKC == 1 ? 3 : 4);
        cs = cursor.next();
        c3_barrier();
        stamp(5);
        tl(3 + (t < 8 ? t : 8));
    }
    flush_stats();
    tl(12);
}

}  // namespace

// C = 32 / 64 ResBlock convs with the activation applied while staging (a.ab required), and the head: a.w = permuted-row slabs
// [k-chunk][kk = tap*4 + c8][C rows][8] (engine.cpp::make_conv d_wp), 16x32 tiles, a.stats = partials [img][tile][8][2] (not for the
// head).  The producers' coefficient table holds 64 / 8 images (C = 32 / 64): conv_pc_fits() tells the engine whether every
// workgroup of a launch stays within it.  (The C >= 128 form of this kernel -- 64-cout items, weight slabs streamed by the producers,
// IRE_PC=7 -- lost to conv_w4 in round 2 and was superseded by conv_pk.hip in round 4: removed.)
bool conv_pc_fits(int C, int tiles_per_img, int nimg) {
    if (C != 32 && C != 64) return false;
    const int imgs = C == 32 ? PcCfg<32>::COEF_IMGS : PcCfg<64>::COEF_IMGS;
    if (nimg <= imgs) return true;
    // the workgroups of XCD group x walk items [items x / X, items (x + 1) / X) (persist.hpp): images spanned by a range
    const int nblk = 1;
    const long long ipi = (long long)tiles_per_img * nblk, items = ipi * nimg;
    const int cus = persistent_grid_cus();
    const long long G = items < cus ? items : cus, X = G < 8 ? G : 8;
    for (long long x = 0; x < X; ++x) {
        const long long lo = items * x / X, hi = items * (x + 1) / X;
        if (hi > lo && (hi - 1) / ipi - lo / ipi + 1 > imgs) return false;
    }
    return true;
}

void conv_pc_launch(bool resid, bool head, const ConvArgs& a, hipStream_t stream) {
    const int C = a.cout;
    if ((C != 32 && C != 64) || a.cin0 != C || a.nkc != C / 32 || a.nblocks != 1 || !a.ab)
        fail(IRE_ERR_INTERNAL, "internal: conv_pc arguments");
    if (head ? (C != 32 || !a.u8_in || !a.u8_out) : !a.stats) fail(IRE_ERR_INTERNAL, "internal: conv_pc arguments");
    if (!conv_pc_fits(C, a.tiles_x * a.tiles_y, a.nimg)) fail(IRE_ERR_INTERNAL, "internal: conv_pc batch");
    const int items = a.tiles_x * a.tiles_y * a.nimg * a.nblocks;
    const int cus = persistent_grid_cus();
    const int grid = items < cus ? items : cus;
#define PC_GO(CC, RS, HD) hipLaunchKernelGGL((conv_pc_kernel<CC, RS, HD>), dim3(grid), dim3(C3_CONS + pc_prod(CC, HD)), 0, stream, a)
    if (head) PC_GO(32, false, true);
    else if (C == 32) { if (resid) PC_GO(32, true, false); else PC_GO(32, false, false); }
    else { if (resid) PC_GO(64, true, false); else PC_GO(64, false, false); }
#undef PC_GO
    IRE_HIP(hipGetLastError());
}

}  // namespace ire
