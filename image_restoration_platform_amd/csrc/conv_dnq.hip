// conv_dnq.hip -- the encoder's stride-2 convolutions with cout >= 128 (`down1`: 64 -> 128, `down2`: 128 -> 256) as 128-cout
// items staged entirely by LDS-DMA, gfx950.  Round 4; conv_down.hip keeps `down0` (cout = 64) and stays the A/B fallback (IRE_DNQ=0).
//
// conv_down.hip's cut (profiles/r03_experiments.md): items of 64 couts stage a tile's phase images once per 64-cout block (2x / 4x
// at cout = 128 / 256), through registers (five 16-B loads + five LDS stores per thread and stage), with 8 - 32 MFMAs per wave between
// barriers; 156 / 132 us per launch against ~80 / ~40 us of HBM time and ~40 us of matrix-pipe time.  The input of a `down` carries
// no GroupNorm, so -- as in conv_upq.hip -- nothing has to pass through registers on its way in:
//   * item = (16 x 32 OUTPUT tile, 128 couts); wave w = output rows 2w, 2w + 1 x 128 couts = 128 accumulators;
//   * the stride-2 3 x 3 convolution is a unit-stride convolution over the four pixel phases P_ab[Y][X] = in[2Y + a][2X + b]
//     (conv_down.hip's header): a stage = (32 input channels, one phase): its 17 x 33 window of the phase image (36 KB) and the
//     phase's 1 / 2 / 2 / 4 taps of weights (8 / 16 / 16 / 32 KB) land by LDS-DMA in the buffer pair the previous stage does not
//     use, issued by the eight computing waves BETWEEN the MFMA pairs of the running stage; zero padding = lanes outside the image
//     fetch from a page of zeros;
//   * 16 / 32 / 32 / 64 MFMAs per wave and stage on 2 pixel + 4 weight fragments per (tap, k-step): 0.75 LDS reads per MFMA;
//   * epilogue: conv_upq.hip's (8-pixel passes through a 2-KB patch, the pieces of the stage after next in front of the stores,
//     counted wait), plain output tile; GroupNorm partials as conv_down's: one (sum, sumsq) per group per tile.
// Weights: a.w = [n-block of 128][kc32][the 9 taps in phase order][c8][128 rows, permuted like conv_w4's][8] bf16
// (engine.cpp::make_conv d_wdq).  Roofline: input staging (every output pixel reads four input pixels), then MFMA.
#include "conv_mfma.hpp"
#include "persist.hpp"

#include <type_traits>

namespace ire {

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

#ifndef DQ_ST
#define DQ_ST IRE_ST_LINE   // cache policy of the output stores (conv_mfma.hpp): build-time A/B
#endif
#ifndef DQ_ABL
#define DQ_ABL 0      // timing ablations (results wrong by design): 2 no epilogue, 4 no MFMA loop, 8 epilogue without its stores, 16 without its statistics, 32 without the pre-issued DMA
#endif

constexpr int DQ_THREADS = 512;
constexpr int DQ_TH = 16, DQ_TW = 32, DQ_IH = 17, DQ_IW = 33, DQ_NT = 128, DQ_NTL = 4;
constexpr int DQ_TILE_PIECES = 36;                              // 17 x 33 pixels x 64 B = 35 904 B as 1-KB DMA pieces
constexpr int DQ_TILE_BYTES = DQ_TILE_PIECES * 1024;
constexpr int DQ_SLAB_BYTES = 4 * 4 * DQ_NT * 16;               // the largest phase: [4 taps][c8][128][8 bf16] = 32 768
constexpr int DQ_TAP_BYTES = 4 * DQ_NT * 16;                    // one tap of a slab: 8 192
constexpr int DQ_GROUP_BYTES = 9 * DQ_TAP_BYTES;                // the nine taps of an (n-block, k-chunk)
constexpr int DQ_W_BASE = 2 * DQ_TILE_BYTES;                    // LDS: tile[2] | slab[2] | red | bias | patches
constexpr int DQ_RED_BASE = DQ_W_BASE + 2 * DQ_SLAB_BYTES;
constexpr int DQ_RED_BYTES = 2 * 8 * 16 * 2 * 4;                // [item parity][8 waves][16 chunks of 8 couts][sum, sumsq]
constexpr int DQ_BIAS_BASE = DQ_RED_BASE + DQ_RED_BYTES;
constexpr int DQ_PATCH_BASE = DQ_BIAS_BASE + 256 * 4;
constexpr int DQ_PATCH_BYTES = 8 * DQ_NT * 2;                   // 2 048: 8 pixels x 128 couts
constexpr int DQ_LDS = DQ_PATCH_BASE + 8 * DQ_PATCH_BYTES;
static_assert(DQ_LDS <= 160 * 1024, "LDS");

__host__ __device__ constexpr int dq_ntaps(int ph) { return ((ph >> 1) ? 2 : 1) * ((ph & 1) ? 2 : 1); }
__host__ __device__ constexpr int dq_tap_base(int ph) { return ph == 0 ? 0 : ph == 1 ? 1 : ph == 2 ? 3 : 5; }     // first tap of the phase in the group's 9
// tap t of phase (a, b) reads window pixel (row + oy, column + ox): oy = a ? t / ntx : 1, ox = b ? t % ntx : 1 (offsets -1 / 0 from the output pixel)
__host__ __device__ constexpr int dq_tap_win(int ph, int t) {
    const int a = ph >> 1, b = ph & 1, ntx = b ? 2 : 1;
    const int oy = a ? t / ntx : 1, ox = b ? t % ntx : 1;
    return oy * 2 + ox;
}

__device__ __forceinline__ unsigned dq_pack(float a, float b) {
    f32x2_t f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ void dq_glds16(const void* gsrc, unsigned lds_dst_uniform) {   // LDS-DMA, 1 KB per wave-instruction (conv_rb.hip::rb_glds16)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ float dq_swap16_add(float v) {
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float dq_swap32_add(float v) {
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
// conv_upq.hip::uq_stage_barrier
template <int KEEP> __device__ __forceinline__ void dq_stage_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(KEEP) : "memory");
}

template <int COUT>
__global__ __launch_bounds__(DQ_THREADS) void conv_dnq_kernel(ConvArgs a) {
    constexpr int NT = DQ_NT, NTL = DQ_NTL, C = COUT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[DQ_LDS];
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(wave);

    const int NKC = a.nkc;                                   // 32-channel chunks of Cin; a stage = (chunk, phase)
    const int Cin = a.cin0;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    PersistCursor cursor(a.tiles_x, a.tiles_y, a.nimg, C / NT, 4 * NKC);          // stage index within an item = phase * NKC + kc (PHASE-major: below)
    const int n_items = cursor.my_items, S = cursor.S;
    if (S == 0) return;

    // ---- DMA addressing (conv_upq.hip).  Wave w stages tile pieces 4w .. 4w + 3 and, w < 4, piece 32 + w; slot s = piece * 64 + lane =
    // window pixel p = s >> 2 (17 x 33, pitch 33), chunk c8 = (s & 3) ^ ((p >> 2) & 3).  Per item: the byte offset of the slot's pixel in
    // phase (0, 0) -- full-res (2 (Y0 - 1 + py), 2 (X0 - 1 + px)) -- and, per phase, whether its pixel lies inside the image (phase (a, b)
    // sits a rows down, b pixels right: a wave-uniform displacement).
    unsigned slotc[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        int l2 = lane;
        asm volatile("" : "+v"(l2));
        const int piece = d < 4 ? 4 * wv + d : 32 + (wv & 3);
        const int s = piece * 64 + l2, p = s >> 2, c8 = (s & 3) ^ ((p >> 2) & 3);
        const int py = p / DQ_IW, px = p - py * DQ_IW;
        slotc[d] = (unsigned)py | ((unsigned)px << 8) | ((unsigned)c8 << 16);
    }
    const char* const zeros = reinterpret_cast<const char*>(a.zeros);
    struct Plan { const void* src[10]; unsigned dst[10]; int n; };
    unsigned moff[5], mok = 0;                               // mok bit ph * 5 + d
    auto item_offsets = [&](const PersistItem& it) __attribute__((always_inline)) {
        const int y0 = 2 * (it.ty * DQ_TH - 1), x0 = 2 * (it.tx * DQ_TW - 1);
        mok = 0;
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            const int py = slotc[d] & 0xff, px = (slotc[d] >> 8) & 0xff, c8 = slotc[d] >> 16;
            const int iy = y0 + 2 * py, ix = x0 + 2 * px;
            moff[d] = (unsigned)((iy + a.in_row_off) * a.Win + ix) * (unsigned)(2 * Cin) + (unsigned)(c8 * 16);      // (may wrap below zero: the phase displacement brings it back)
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const bool ok = py < DQ_IH && (unsigned)(iy + (ph >> 1) - a.iy_lo) < (unsigned)a.iy_span && (unsigned)(ix + (ph & 1)) < (unsigned)a.Win;
                mok |= ok ? (1u << (ph * 5 + d)) : 0u;
            }
        }
    };
    // the pieces of stage `st` (phase PH) for buffer pair b, in issue order: tile pieces 4w .. 4w + 3, the wave's slab pieces (taps of the
    // phase: 1 / 2 / 2 / 4 x 8 KB = 8 .. 32 pieces over the eight waves), tile piece 32 + w (waves 0..3)
    auto plan_stage = [&](auto ph_tag, const PersistStage& st, int b, Plan& P) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value;              // == st.kc / NKC
        constexpr int NSL = dq_ntaps(PH);                        // slab pieces per wave: the phase's taps x 8 KB over eight waves
        const unsigned tdst = smem_lds + b * DQ_TILE_BYTES, sdst = smem_lds + DQ_W_BASE + b * DQ_SLAB_BYTES;
        const int kc = st.kc - PH * NKC;
        const char* base = reinterpret_cast<const char*>(a.in0) + (size_t)st.it.img * a.in_rows * a.Win * (2 * Cin) + kc * 64;
        const unsigned delta = (unsigned)((PH >> 1) * a.Win + (PH & 1)) * (unsigned)(2 * Cin);
        const unsigned okp = mok >> (PH * 5);
#pragma unroll
        for (int d = 0; d < 4; ++d) { P.src[d] = ((okp >> d) & 1u) ? base + (unsigned)(moff[d] + delta) : zeros; P.dst[d] = tdst + (4 * wv + d) * 1024; }
        const unsigned char* ws = reinterpret_cast<const unsigned char*>(a.w) + ((size_t)st.it.nb * NKC + kc) * DQ_GROUP_BYTES + dq_tap_base(PH) * DQ_TAP_BYTES;
#pragma unroll
        for (int d = 0; d < NSL; ++d) { const int piece = wv + 8 * d; P.src[4 + d] = ws + (size_t)(piece * 64 + lane) * 16; P.dst[4 + d] = sdst + piece * 1024; }
#pragma unroll
        for (int d = 4 + NSL; d < 10; ++d) { P.src[d] = ((okp >> 4) & 1u) ? base + (unsigned)(moff[4] + delta) : zeros; P.dst[d] = tdst + (32 + (wv & 3)) * 1024; }
        P.n = 4 + NSL + (wv < 4 ? 1 : 0);
    };

    // ---- fragment addressing.  Pixel fragment (row m, window offset (oy, ox), k): p = (2w + m + oy) 33 + r + ox, byte
    // (p * 4 + ((2k + h) ^ ((p >> 2) & 3))) * 16 (+ the tile base: a constant per stage, the phases of a chunk alternate the pairs);
    // weight fragment (tap t of the phase, k, rows 32 j + r): ((t * 4 + 2k + h) * 128 + 32 j + r) * 16.
    int a_off[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int p = (wave * 2 + m + (o >> 1)) * DQ_IW + r + (o & 1);
            a_off[m][o] = (p * 4 + (h ^ ((p >> 2) & 3))) * 16;
        }
    const int b_off = (h * NT + r) * 16;
    const float* bias_lds = reinterpret_cast<const float*>(smem + DQ_BIAS_BASE);
    float* red = reinterpret_cast<float*>(smem + DQ_RED_BASE);

    int st_img = -1, st_tile = 0, st_cout0 = 0, st_par = 0, red_par = 0;
    auto flush_stats = [&]() {          // GroupNorm partials of the item whose epilogue ended before the last barrier (conv_pk.hip)
        if (st_img < 0) return;
        constexpr int G = C / 8, CPG = G >> 3, NGL = NT / G;      // couts per group, chunks of 8 couts per group (2 or 4), groups in the item
        if (a.stats && tid < NGL) {
            const float* rd = red + st_par * (8 * 32);
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w)
#pragma unroll
                for (int k = 0; k < CPG; ++k) { sv += rd[(w * 16 + tid * CPG + k) * 2 + 0]; qv += rd[(w * 16 + tid * CPG + k) * 2 + 1]; }
            float* st = a.stats + (((size_t)st_img * tiles_per_img + st_tile) * 8 + st_cout0 / G + tid) * 2;
            st[0] = sv; st[1] = qv;
        }
        st_img = -1;
    };

    f32x16_t acc[2][NTL];
    PersistStage cs = cursor.cur;
    PersistStage cn = cursor.next();
    {   // prologue: stage 0 into pair 0
        Plan P;
        item_offsets(cs.it);
        plan_stage(std::integral_constant<int, 0>{}, cs, 0, P);
#pragma unroll
        for (int i = 0; i < 10; ++i) if (i < P.n) dq_glds16(P.src[i], P.dst[i]);
    }
    if (tid < C) reinterpret_cast<float*>(smem + DQ_BIAS_BASE)[tid] = a.bias[tid];
    dq_stage_barrier<0>();
    int stage_no = 0;
    bool pre_issued = false;              // the next stage's pieces are already on their way (issued by the previous item's epilogue)

    for (int t = 0; t < n_items; ++t) {
        const PersistItem it = cs.it;
        const int cout0 = it.nb * NT;
        {   // accumulators start at the bias (permuted slab rows: accumulator i of lane-half h is cout 32 j + 16 (i >> 3) + 8 h + (i & 7))
            const float* bl = bias_lds + cout0 + 8 * h;
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bv = *reinterpret_cast<const float4*>(bl + j * 32 + 16 * (q >> 1) + 4 * (q & 1));
#pragma unroll
                    for (int m = 0; m < 2; ++m) { acc[m][j][4 * q + 0] = bv.x; acc[m][j][4 * q + 1] = bv.y; acc[m][j][4 * q + 2] = bv.z; acc[m][j][4 * q + 3] = bv.w; }
                }
        }
        // one stage: phase PH of a chunk on buffer pair PH & 1 (an item's stages start on pair 0: four per chunk) while the next stage's
        // pieces land in the other pair; NS = 2 * taps k-steps of 4 weight fragments x 2 pixel rows, fragments rotating as in conv_pk.hip
        // Stage order is PHASE-major (all chunks of phase 0, then of phase 1, ..): a phase's 32-channel chunks read neighbouring 64-byte
        // pieces of the same 128-byte lines, and back to back the second piece is an L2 hit -- chunk-major (the phases of a chunk, then the
        // next chunk, 8 us later) fetched every line of `down1`'s input from HBM twice (2.1 x its tensor; DMA-only time 102 us).
        auto stage = [&](auto ph_tag, auto buf_tag, bool last_chunk) __attribute__((always_inline)) {
            constexpr int PH = decltype(ph_tag)::value, BUF = decltype(buf_tag)::value, NTAPS = dq_ntaps(PH);
            Plan P;
            P.n = 0;
            if (stage_no + 1 < S && !pre_issued) {
                if (cn.kc == 0) item_offsets(cn.it);
                if (last_chunk) plan_stage(std::integral_constant<int, (PH + 1) & 3>{}, cn, BUF ^ 1, P);       // the next phase's (or the next item's) first chunk
                else plan_stage(std::integral_constant<int, PH>{}, cn, BUF ^ 1, P);
            }
            const unsigned char* ib = smem + BUF * DQ_TILE_BYTES;
            const unsigned char* wb = smem + DQ_W_BASE + BUF * DQ_SLAB_BYTES + b_off;
            if constexpr (!(DQ_ABL & 4)) {
                constexpr int NS = 2 * NTAPS, NG = NS * NTL;
                auto rd_b = [&](int st, int j) __attribute__((always_inline)) -> bf16x8_t {
                    return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wb + (((st >> 1) * 4 + 2 * (st & 1)) * NT + j * 32) * 16));
                };
                auto rd_a = [&](int st, int m) __attribute__((always_inline)) -> bf16x8_t {
                    return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ib + (a_off[m][dq_tap_win(PH, st >> 1)] ^ ((st & 1) << 5))));
                };
                bf16x8_t bq[2], aq[2][2];
                bq[0] = rd_b(0, 0);
                aq[0][0] = rd_a(0, 0); aq[0][1] = rd_a(0, 1);
#pragma unroll
                for (int st = 0; st < NS; ++st)
#pragma unroll
                    for (int j = 0; j < NTL; ++j) {
                        const int g = st * NTL + j;
                        // the next stage's DMA pieces, spread over the k-loop: piece i rides in front of MFMA pair (i NG) / 10
#pragma unroll
                        for (int i = 0; i < 10; ++i)
                            if ((i * NG) / 10 == g && i < P.n) dq_glds16(P.src[i], P.dst[i]);
                        if (g + 1 < NG) bq[(g + 1) & 1] = rd_b((g + 1) / NTL, (g + 1) % NTL);
                        if (st + 1 < NS && j == 1) aq[(st + 1) & 1][0] = rd_a(st + 1, 0);
                        if (st + 1 < NS && j == 2) aq[(st + 1) & 1][1] = rd_a(st + 1, 1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[g & 1], aq[st & 1][m], acc[m][j], 0, 0, 0);   // D[cout][pixel]
                        __builtin_amdgcn_sched_barrier(0);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 10; ++i) if (i < P.n) dq_glds16(P.src[i], P.dst[i]);
            }
            ++stage_no;
            cs = cn; cn = cursor.next();
            // the next stage is staged; nobody reads this stage's pair any more
            if (pre_issued) dq_stage_barrier<16>(); else dq_stage_barrier<0>();
            pre_issued = false;
            flush_stats();
        };
        // NKC is even (Cin = 64 / 128): a phase's chunks alternate the pairs starting on pair 0
        auto phase = [&](auto ph_tag) __attribute__((always_inline)) {
#pragma unroll 1
            for (int kc = 0; kc < NKC; kc += 2) {
                stage(ph_tag, std::integral_constant<int, 0>{}, false);
                stage(ph_tag, std::integral_constant<int, 1>{}, kc + 2 >= NKC);
            }
        };
        phase(std::integral_constant<int, 0>{});
        phase(std::integral_constant<int, 1>{});
        phase(std::integral_constant<int, 2>{});
        phase(std::integral_constant<int, 3>{});
        // ---- epilogue (conv_upq.hip): the pieces of the stage after next (the next item's stage 1: pair 1) first, then 8 passes of 8
        // pixels x 128 couts through the wave's patch, 16 stores of whole 256-B pixel runs
        if (stage_no + 1 < S && !(DQ_ABL & 32)) {
            Plan P;
            plan_stage(std::integral_constant<int, 0>{}, cn, 1, P);       // the next item's stage 1 = phase 0, chunk 1
#pragma unroll
            for (int i = 0; i < 10; ++i) if (i < P.n) dq_glds16(P.src[i], P.dst[i]);
            pre_issued = true;
        }
        if constexpr (DQ_ABL & 2) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < NTL; ++j) asm volatile("" :: "v"(acc[m][j]));
            if (pre_issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            unsigned char* patch = smem + DQ_PATCH_BASE + wave * DQ_PATCH_BYTES;
            int l_e = lane, w_e = wave;
            asm volatile("" : "+v"(l_e), "+v"(w_e));
            const int oyb = it.ty * DQ_TH + w_e * 2;
            const int tcol0 = it.tx * DQ_TW + (l_e >> 4);
            unsigned toffs[2];
            bool trow[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int oy = oyb + m;
                trow[m] = oy < a.Hout;
                toffs[m] = ((unsigned)((min(oy, a.Hout - 1) * a.Wout + tcol0) * C + cout0 + 8 * (l_e & 15)) << 1);
            }
            char* obase = reinterpret_cast<char*>(a.out) + (size_t)it.img * a.Hout * a.Wout * (2 * C);
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, a.Hout * a.Wout * (2 * C), 0x00020000);
            const unsigned cstep = (unsigned)(2 * C) * 4u;                    // bytes per read-back's 4 pixels
            const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
            const int r8 = l_e & 7, eg = (l_e >> 3) & 3;                      // writer: pixel r = 8 eg + r8 of the row, half h
            const int h_e = l_e >> 5;
            const int pq = l_e >> 4, cc_r = l_e & 15;                         // reader: pixel 4 k + pq of the 8-pixel group, chunk cc_r (8 couts)
            constexpr int PITCH = NT * 2;
            u32x4_t pkd[2][NTL * 2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < NTL * 2; ++g) {
                    const f32x16_t& c = acc[m][g >> 1];
                    const int pp = g & 1;
                    pkd[m][g] = u32x4_t{dq_pack(c[8 * pp + 0], c[8 * pp + 1]), dq_pack(c[8 * pp + 2], c[8 * pp + 3]),
                                        dq_pack(c[8 * pp + 4], c[8 * pp + 5]), dq_pack(c[8 * pp + 6], c[8 * pp + 7])};
                }
            __builtin_amdgcn_sched_barrier(0);
            float ssum = 0.f, qsum = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (eg == e) {
#pragma unroll
                        for (int g = 0; g < NTL * 2; ++g) {
                            const int cc = 2 * g + h_e;                       // chunk of the pixel's 128-cout run: couts 8 cc .. 8 cc + 7
                            *reinterpret_cast<u32x4_t*>(patch + r8 * PITCH + ((cc ^ r8) << 4)) = pkd[m][g];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int p = 4 * k + pq;
                        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(patch + p * PITCH + ((cc_r ^ p) << 4));
                        const unsigned w[4] = {v.x, v.y, v.z, v.w};
                        float s1 = 0.f, q1 = 0.f;
                        if constexpr (!(DQ_ABL & 16))
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const bf16x2_t bv = __builtin_bit_cast(bf16x2_t, w[d]);
                            s1 = __builtin_amdgcn_fdot2_f32_bf16(bv, ones, s1, false);
                            q1 = __builtin_amdgcn_fdot2_f32_bf16(bv, bv, q1, false);
                        }
                        const bool ok = trow[m] && tcol0 + 8 * e + 4 * k < a.Wout;
                        ssum += ok ? s1 : 0.f; qsum += ok ? q1 : 0.f;
                        if constexpr (DQ_ABL & 8) asm volatile("" :: "v"(v));
                        else __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, ok ? toffs[m] + (unsigned)(2 * e + k) * cstep : 0xffffffffu, 0, DQ_ST);
                    }
                }
            // this lane's chunk cc_r over its read-backs; the other lanes with the same chunk sit 16 apart
            ssum = dq_swap16_add(ssum); qsum = dq_swap16_add(qsum);
            ssum = dq_swap32_add(ssum); qsum = dq_swap32_add(qsum);
            if (l_e < 16) *reinterpret_cast<float2*>(red + red_par * (8 * 32) + (wave * 16 + cc_r) * 2) = make_float2(ssum, qsum);
            st_img = it.img; st_tile = it.tile; st_cout0 = cout0; st_par = red_par; red_par ^= 1;
            if constexpr (DQ_ABL & 8) { if (pre_issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }      // (no stores behind the pieces: the counted wait would let them pass)
        }
    }
    __syncthreads();                     // the last item's chunk sums
    flush_stats();
}

}  // namespace

// a.in0 = full-res input [img][in_rows][Win][Cin] (a.Hin, a.Win full-res), a.out [img][Hout][Wout][cout] at half resolution;
// a.tiles_x / tiles_y = 16 x 32 OUTPUT tiles, a.nkc = Cin / 32, a.nblocks = cout / 128, a.w = d_wdq, a.zeros; a.stats = partials [img][tile][8][2].
void conv_dnq_launch(const ConvArgs& a, hipStream_t stream) {
    if ((a.cout != 128 && a.cout != 256) || a.cin0 % 64 || a.nkc != a.cin0 / 32 || a.nblocks != a.cout / DQ_NT || !a.w || !a.zeros)
        fail(IRE_ERR_INTERNAL, "internal: conv_dnq arguments");
    const int items = a.tiles_x * a.tiles_y * a.nimg * a.nblocks;
    const int cus = persistent_grid_cus();
    const int grid = items < cus ? items : cus;
    if (a.cout == 128) hipLaunchKernelGGL(conv_dnq_kernel<128>, dim3(grid), dim3(DQ_THREADS), 0, stream, a);
    else hipLaunchKernelGGL(conv_dnq_kernel<256>, dim3(grid), dim3(DQ_THREADS), 0, stream, a);
    IRE_HIP(hipGetLastError());
}

}  // namespace ire
