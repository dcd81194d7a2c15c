// conv_upq.hip -- the decoder's level-2 `up` + `fuse` (256 -> 128 channels) as PARITY-MAJOR items with 128 couts, gfx950.  Round 4.
//
// conv_up.hip runs this layer as items of (16 x 32 low-res tile, 32 couts, all four output parities): with 128 couts a tile's
// halo tile is staged FOUR times (once per 32-cout block), every item reads all 128 channels of its 2 048 skip pixels (512 KB,
// re-read by each of the four blocks) and the k-loop needs 68 LDS fragment reads per 64 MFMAs (four parities x their own weights:
// a weight fragment serves two MFMAs).  Measured 222 us per launch against 79 us of matrix-pipe time: LDS-bound main loop, and an
// epilogue (skip term + stores) as long as the main loop (profiles/r02_experiments.md ablations, r03 counters).
//
// Here an item is (16 x 32 LOW-res tile, ONE output parity (pa, pb), all 128 couts) -> 16 x 32 output pixels (2Y + pa, 2X + pb):
//   * a parity of the sub-pixel form is a 2 x 2 convolution over low-res rows Y - 1 + pa .. Y + pa, columns X - 1 + pb .. X + pb with
//     its own pre-summed weights (conv_up.hip's header): wave w = low-res rows 2w, 2w + 1 x 128 couts = 128 accumulators, per
//     32-channel stage 4 taps x 2 k-steps x (2 pixel + 4 weight fragments) = 48 reads per 64 MFMAs;
//   * the 1 x 1 skip term (K = 128 skip channels = eight 16-channel k-steps) rides on the eight main stages, one k-step each: a
//     lane's two skip pixel fragments (B layout: 16 B per lane) and the step's four weight fragments (A layout: 1 KB contiguous per
//     load) come straight from global memory into registers at the top of a stage and feed 8 MFMAs behind its k-loop -- each skip
//     pixel is read by exactly one item (128 KB per item).  (First cut: four more pipeline STAGES with the skip pixels by LDS-DMA like
//     an input tile: 31 us per launch for 11 % of the MFMAs -- each such stage is a bare DMA round trip; this form: 25, ~10 us saved);
//   * nothing is activated on the way in (the `up` input carries no GroupNorm), so ALL tile / slab staging is LDS-DMA, issued by the eight
//     computing waves one stage ahead into the other buffer pair (tile 36 KB + slab 32 KB, two pairs): no staging registers, no
//     vector instructions besides the addresses; zero padding = lanes outside the image fetch from a page of zeros;
//   * epilogue: conv_pk.hip's line-coalesced form (a wave transposes 8 pixels x 128 couts at a time through a 2-KB patch of its own and
//     stores whole 256-B pixel runs); the DMA pieces of the stage after next are issued IN FRONT of the stores and the next barrier
//     waits for "all but the 16 youngest" operations: the 128-KB-per-CU store burst drains behind the next stage's MFMAs.
// One barrier per stage (8 per item).  GroupNorm partials: one (sum, sumsq) per group per ITEM, i.e. FOUR partial rows per low-res
// tile (index tile * 4 + parity): the engine tells the finalize so (exec_conv: stat_parts).
// Weights: a.w = [parity][kc32][tap4][c8][128 rows, permuted like conv_w4's][8] bf16 (engine.cpp::make_up_fused d_wuq),
// a.w1 = skip weights [ks16][h][128 rows][8] (d_wsq; as bytes the same as [ks32][c8][128][8]), a.bias = composed bias.  Roofline: MFMA (executed 2*4*Cin*C + 2*C*C flop per pixel).
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

#ifndef UQ_ST
#define UQ_ST IRE_ST_LINE   // cache policy of the output stores (conv_mfma.hpp): build-time A/B
#endif
#ifndef UQ_DMA_SPAN
#define UQ_DMA_SPAN 32   // the next stage's DMA pieces are issued within the first UQ_DMA_SPAN / 32 of the k-loop's MFMA pairs
#endif
#ifndef UQ_ABL
#define UQ_ABL 0      // timing ablations (results wrong by design): 1 no skip term, 2 no epilogue, 4 no MFMA loop, 8 no tile DMA, 16 no slab DMA, 64 the k-loop twice
#endif

constexpr int UQ_THREADS = 512;
constexpr int UQ_TH = 16, UQ_TW = 32, UQ_IH = 17, UQ_IW = 33, UQ_NT = 128, UQ_NTL = 4;
constexpr int UQ_TILE_PIECES = 36;                              // 17 x 33 pixels x 64 B = 35 904 B as 1-KB DMA pieces (the last 960 B: dummy slots)
constexpr int UQ_TILE_BYTES = UQ_TILE_PIECES * 1024;
constexpr int UQ_SLAB_BYTES = 4 * 4 * UQ_NT * 16;               // main stage: [tap4][c8][128][8 bf16] = 32 768
constexpr int UQ_W_BASE = 2 * UQ_TILE_BYTES;                    // LDS: tile[2] | slab[2] | red | bias
constexpr int UQ_RED_BASE = UQ_W_BASE + 2 * UQ_SLAB_BYTES;
constexpr int UQ_RED_BYTES = 2 * 8 * 16 * 2 * 4;                // [item parity][8 waves][16 chunks of 8 couts][sum, sumsq]
constexpr int UQ_BIAS_BASE = UQ_RED_BASE + UQ_RED_BYTES;
constexpr int UQ_PATCH_BASE = UQ_BIAS_BASE + UQ_NT * 4;
constexpr int UQ_PATCH_BYTES = 8 * UQ_NT * 2;                   // 2 048: a wave's transpose patch, 8 pixels x 128 couts (LDS of its own: see the epilogue)
constexpr int UQ_LDS = UQ_PATCH_BASE + 8 * UQ_PATCH_BYTES;
static_assert(UQ_LDS <= 160 * 1024, "LDS");

__device__ __forceinline__ unsigned uq_pack(float a, float b) {
    f32x2_t f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ void uq_glds16(const void* gsrc, unsigned lds_dst_uniform) {   // LDS-DMA, 1 KB per wave-instruction (conv_rb.hip::rb_glds16)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ float uq_swap16_add(float v) {
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float uq_swap32_add(float v) {
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
// the stage barrier: this wave's DMA pieces of the next stage have landed, its LDS reads are done; behind the barrier everybody's are.
// KEEP = 16: the stage that follows an epilogue -- its successor's pieces were issued IN FRONT of the epilogue's 16 output stores
// (vector-memory operations retire in issue order: "all but the 16 youngest" = exactly those pieces), so the stores drain in the
// background instead of in front of a barrier (all 256 workgroups store 128 KB at the same moment: 30 us per launch when waited for).
template <int KEEP> __device__ __forceinline__ void uq_stage_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(KEEP) : "memory");
}

__global__ __launch_bounds__(UQ_THREADS) void conv_upq_kernel(ConvArgs a) {
    constexpr int NT = UQ_NT, NTL = UQ_NTL, C = UQ_NT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[UQ_LDS];
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(wave);

    const int NKC = a.nkc;                                   // 32-channel stages of the sub-pixel convolution (Cin / 32)
    const int NST = NKC;                                     // (the skip term's k-steps ride at the end of the main stages: below)
    const int Cin = a.cin0;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    PersistCursor cursor(a.tiles_x, a.tiles_y, a.nimg, 4, NST);          // it.nb = output parity pa * 2 + pb
    const int n_items = cursor.my_items, S = cursor.S;
    if (S == 0) return;

    // ---- DMA addressing.  Wave w stages tile pieces 4w .. 4w + 3 and, w < 4, piece 32 + w.  Chunk slot s = piece * 64 + lane holds
    // pixel p = s >> 2 of the 17 x 33 window (pitch 33), 8-channel chunk c8 = (s & 3) ^ ((p >> 2) & 3) (conv_up.hip's swizzle: conflict-free
    // fragment reads).
    // A lane's slots sit at the same window position for the whole kernel: (py, px, c8) of its five pieces, packed once.
    unsigned slotc[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        int l2 = lane;
        asm volatile("" : "+v"(l2));
        const int piece = d < 4 ? 4 * wv + d : 32 + (wv & 3);
        const int s = piece * 64 + l2, p = s >> 2, c8 = (s & 3) ^ ((p >> 2) & 3);
        const int py = p / UQ_IW, px = p - py * UQ_IW;
        slotc[d] = (unsigned)py | ((unsigned)px << 8) | ((unsigned)c8 << 16);
    }
    const char* const zeros = reinterpret_cast<const char*>(a.zeros);
    // The pieces of stage `st` for buffer pair b (its readers are behind the last barrier): sources and LDS destinations only -- the
    // caller issues them (in the prologue at once; in a stage BETWEEN the MFMA pairs of the k-loop: a DMA instruction holds the wave's
    // issue for 100+ cycles, and nine of them in front of the k-loop left the matrix pipe idle for a third of the stage).
    // Order: tile pieces 4w .. 4w + 3, the wave's four slab pieces, tile piece 32 + w (waves 0..3).
    struct Plan { const void* src[10]; unsigned dst[10]; int n; };
    // a lane's byte offsets within the image, once per ITEM (main: low-res input pixel, skip: the parity's output pixel, both at
    // channel 8 c8 of stage 0); ~0u = outside the image / a dummy slot: that lane fetches from the page of zeros.  A stage's plan
    // is then base(image, stage) + offset: ~30 vector instructions in front of the k-loop instead of ~120.
    unsigned moff[5];
    unsigned skoff[2];                                       // the lane's skip pixels (rows m = 0, 1) as B fragments: byte offset in the skip image at channel 8 h, ~0u outside
    auto item_offsets = [&](const PersistItem& it) __attribute__((always_inline)) {
        const int pa = it.nb >> 1, pb = it.nb & 1;
        const int Y0 = it.ty * UQ_TH, X0 = it.tx * UQ_TW;
        const int ybase = Y0 - 1 + pa - a.iy_lo, xbase = X0 - 1 + pb;
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            const int py = slotc[d] & 0xff, px = (slotc[d] >> 8) & 0xff, c8 = slotc[d] >> 16;
            const int ry = ybase + py, ix = xbase + px;               // ry: row relative to the first readable one
            const bool okm = py < UQ_IH && (unsigned)ry < (unsigned)a.iy_span && (unsigned)ix < (unsigned)a.Win;
            moff[d] = okm ? (unsigned)((ry + a.iy_lo + a.in_row_off) * a.Win + ix) * (unsigned)(2 * Cin) + (unsigned)(c8 * 16) : ~0u;
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int ly = Y0 + 2 * wave + m, lx = X0 + r;
            skoff[m] = (ly < a.Hin && lx < a.Win) ? (unsigned)((2 * ly + pa) * a.Wout + 2 * lx + pb) * (unsigned)(2 * C) + (unsigned)(h * 16) : ~0u;
        }
    };
    auto plan_stage = [&](const PersistStage& st, int b, Plan& P) __attribute__((always_inline)) {
        const unsigned tdst = smem_lds + b * UQ_TILE_BYTES, sdst = smem_lds + UQ_W_BASE + b * UQ_SLAB_BYTES;
        const char* base = reinterpret_cast<const char*>(a.in0) + (size_t)st.it.img * a.in_rows * a.Win * (2 * Cin) + st.kc * 64;
        const char* tsrc[5];
#pragma unroll
        for (int d = 0; d < 5; ++d) tsrc[d] = moff[d] != ~0u ? base + moff[d] : zeros;
#pragma unroll
        for (int d = 0; d < 4; ++d) { P.src[d] = tsrc[d]; P.dst[d] = tdst + (4 * wv + d) * 1024; }
        const unsigned char* ws = reinterpret_cast<const unsigned char*>(a.w) + ((size_t)st.it.nb * NKC + st.kc) * UQ_SLAB_BYTES;
#pragma unroll
        for (int d = 0; d < 4; ++d) { const int piece = wv + 8 * d; P.src[4 + d] = ws + (size_t)(piece * 64 + lane) * 16; P.dst[4 + d] = sdst + piece * 1024; }
        P.src[8] = tsrc[4]; P.dst[8] = tdst + (32 + (wv & 3)) * 1024;
        P.src[9] = tsrc[4]; P.dst[9] = P.dst[8];
        P.n = wv < 4 ? 9 : 8;
    };

    // ---- fragment addressing.  Pixel fragment (row m, tap (dy, dx), k-step k): window pixel p = (2w + m + dy) 33 + r + dx, chunk
    // c8 = 2k + h: byte (p * 4 + (c8 ^ ((p >> 2) & 3))) * 16 -- k toggles bit 5.  Weight fragment (tap, k, rows 32 j + r): ((tap*4 + 2k + h) 128 + 32 j + r) 16.
    // The eight pixel-fragment addresses carry the tile base of the buffer pair in use (they step by +- one tile per stage).
    int a_off[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            const int p = (wave * 2 + m + (tap >> 1)) * UQ_IW + r + (tap & 1);
            a_off[m][tap] = (p * 4 + (h ^ ((p >> 2) & 3))) * 16;
        }
    const int b_off = (h * NT + r) * 16;
    const float* bias_lds = reinterpret_cast<const float*>(smem + UQ_BIAS_BASE);
    float* red = reinterpret_cast<float*>(smem + UQ_RED_BASE);

    int st_img = -1, st_slot = 0, st_par = 0, red_par = 0;
    auto flush_stats = [&]() {          // GroupNorm partials of the item whose epilogue ended before the last barrier: 8 waves x 16 chunk slots -> 8 groups
        if (st_img < 0) return;
        constexpr int G = C / 8, CPG = G >> 3, NGL = NT / G;
        if (a.stats && tid < NGL) {
            const float* rd = red + st_par * (8 * 32);
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w)
#pragma unroll
                for (int k = 0; k < CPG; ++k) { sv += rd[(w * 16 + tid * CPG + k) * 2 + 0]; qv += rd[(w * 16 + tid * CPG + k) * 2 + 1]; }
            float* st = a.stats + (((size_t)st_img * tiles_per_img * 4 + st_slot) * 8 + tid) * 2;
            st[0] = sv; st[1] = qv;
        }
        st_img = -1;
    };

    f32x16_t acc[2][NTL];
    PersistStage cs = cursor.cur;
    PersistStage cn = cursor.next();
    // ---- prologue: stage 0 into pair 0
    {
        Plan P;
        item_offsets(cs.it);
        plan_stage(cs, 0, P);
#pragma unroll
        for (int i = 0; i < 10; ++i) if (i < P.n) uq_glds16(P.src[i], P.dst[i]);
    }
    if (tid < C) reinterpret_cast<float*>(smem + UQ_BIAS_BASE)[tid] = a.bias[tid];
    uq_stage_barrier<0>();
    int buf = 0, stage_no = 0;
    bool pre_issued = false;              // the next stage's pieces are already on their way (issued by the previous item's epilogue)

    for (int t = 0; t < n_items; ++t) {
        const PersistItem it = cs.it;
        {   // accumulators start at the composed bias (permuted slab rows: accumulator i of lane-half h is cout 32 j + 16 (i >> 3) + 8 h + (i & 7))
            const float* bl = bias_lds + 8 * h;
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bv = *reinterpret_cast<const float4*>(bl + j * 32 + 16 * (q >> 1) + 4 * (q & 1));
#pragma unroll
                    for (int m = 0; m < 2; ++m) { acc[m][j][4 * q + 0] = bv.x; acc[m][j][4 * q + 1] = bv.y; acc[m][j][4 * q + 2] = bv.z; acc[m][j][4 * q + 3] = bv.w; }
                }
        }
        // one pipeline stage: NTAPS taps of 2 k-steps on the pair in use while the next stage's tile and slab land in the other
        // (ONE k-loop per loop body: two variants under a branch make the accumulators PHIs that hipcc copies and spills)
        auto stage = [&](auto ntaps_tag) __attribute__((always_inline)) {
            // the 1 x 1 skip term, one 16-channel k-step per main stage (Cin / 32 == C / 16 stages): this lane's two skip pixel fragments
            // (B layout: 16 B of pixel (2 (Y0 + 2w + m) + pa, 2 (X0 + r) + pb), channels 16 ks + 8 h) and the four weight fragments (A layout:
            // rows 32 j + r of [ks][h][128][8]: 1 KB contiguous per load) come STRAIGHT FROM GLOBAL MEMORY into registers at the top of the
            // stage and are used behind its k-loop -- the separate skip stages (a DMA round trip for 16 MFMAs each) are gone
            u32x4_t skp[2], skw[NTL];
            if constexpr (!(UQ_ABL & 1)) {
                const int ks = cs.kc;
                char* sbase = const_cast<char*>(reinterpret_cast<const char*>(a.in1)) + (size_t)cs.it.img * a.Hout * a.Wout * (2 * C) + ks * 32;
                const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(sbase, 0, a.Hout * a.Wout * (2 * C) - ks * 32, 0x00020000);
#pragma unroll
                for (int m = 0; m < 2; ++m) skp[m] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, skoff[m], 0, 0);      // ~0u: out of range reads as zero
                const char* wsk = reinterpret_cast<const char*>(a.w1) + ((size_t)(ks * 2 + h) * NT + r) * 16;
#pragma unroll
                for (int j = 0; j < NTL; ++j) skw[j] = *reinterpret_cast<const u32x4_t*>(wsk + j * 512);
            }
            Plan P;
            P.n = 0;
            if (stage_no + 1 < S && !pre_issued) {
                if (cn.kc == 0) item_offsets(cn.it);
                plan_stage(cn, buf ^ 1, P);
            }
            const unsigned char* ib = smem;
            const unsigned char* wb = smem + UQ_W_BASE + buf * UQ_SLAB_BYTES + b_off;
            // NS k-steps (tap, k) x 4 weight fragments x 2 pixel rows; fragments rotate through two weight registers and two pairs of pixel
            // registers, every ds_read_b128 issued one MFMA pair ahead (conv_pk.hip's k-loop)
            if constexpr (!(UQ_ABL & 4)) {
                constexpr int NS = 2 * decltype(ntaps_tag)::value, NG = NS * NTL;
                auto rd_b = [&](int st, int j) __attribute__((always_inline)) -> bf16x8_t {
                    return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wb + (((st >> 1) * 4 + 2 * (st & 1)) * NT + j * 32) * 16));
                };
                auto rd_a = [&](int st, int m) __attribute__((always_inline)) -> bf16x8_t {
                    return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ib + (a_off[m][st >> 1] ^ ((st & 1) << 5))));
                };
#pragma unroll 1
                for (int rep = 0; rep < ((UQ_ABL & 64) ? 2 : 1); ++rep) {       // (ablation 64: the k-loop twice per stage -- what does a k-loop cost by itself?)
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
                            if ((i * (NG * UQ_DMA_SPAN / 32)) / 10 == g && i < P.n && !(((UQ_ABL & 8) && (i < 4 || i >= P.n - (wv < 4 ? 1 : 0))) || ((UQ_ABL & 16) && i >= 4 && i < P.n - (wv < 4 ? 1 : 0)))) uq_glds16(P.src[i], P.dst[i]);
                        if (g + 1 < NG) bq[(g + 1) & 1] = rd_b((g + 1) / NTL, (g + 1) % NTL);
                        if (st + 1 < NS && j == 1) aq[(st + 1) & 1][0] = rd_a(st + 1, 0);
                        if (st + 1 < NS && j == 2) aq[(st + 1) & 1][1] = rd_a(st + 1, 1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[g & 1], aq[st & 1][m], acc[m][j], 0, 0, 0);   // D[cout][pixel]
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if constexpr (UQ_ABL & 4) {
#pragma unroll
                for (int i = 0; i < 10; ++i) if (i < P.n) uq_glds16(P.src[i], P.dst[i]);
            }
            if constexpr (!(UQ_ABL & 1) && !(UQ_ABL & 4)) {
#pragma unroll
                for (int j = 0; j < NTL; ++j)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, skw[j]), __builtin_bit_cast(bf16x8_t, skp[m]), acc[m][j], 0, 0, 0);
            }
            {   // the fragment addresses move to the other tile
                const int step = buf ? -UQ_TILE_BYTES : UQ_TILE_BYTES;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int tap = 0; tap < 4; ++tap) a_off[m][tap] += step;
            }
            buf ^= 1;
            ++stage_no;
            cs = cn; cn = cursor.next();
            // the next stage is staged; nobody reads this stage's pair any more
            if (pre_issued) uq_stage_barrier<16>(); else uq_stage_barrier<0>();
            pre_issued = false;
            flush_stats();
        };
#pragma unroll 1
        for (int kc = 0; kc < NKC; ++kc) stage(std::integral_constant<int, 4>{});
        // ---- epilogue: conv_pk.hip's line-coalesced form in 8-pixel passes through a 2-KB patch of the wave's own, so that both buffers of
        // the other pair are free NOW: the pieces of the stage after next (the next item's stage 1) are issued first, the stores behind them.
        if (stage_no + 1 < S) {
            Plan P;
            plan_stage(cn, buf ^ 1, P);
#pragma unroll
            for (int i = 0; i < 10; ++i) if (i < P.n) uq_glds16(P.src[i], P.dst[i]);
            pre_issued = true;
        }
        if constexpr (UQ_ABL & 2) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < NTL; ++j) asm volatile("" :: "v"(acc[m][j]));
            if (pre_issued) {       // keep the counted wait honest: 16 dummy-free -- no stores were issued, so wait for everything
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            unsigned char* patch = smem + UQ_PATCH_BASE + wave * UQ_PATCH_BYTES;
            const int pa = it.nb >> 1, pb = it.nb & 1;
            int l_e = lane, w_e = wave;
            asm volatile("" : "+v"(l_e), "+v"(w_e));
            const int lyb = it.ty * UQ_TH + w_e * 2;
            const int lcol0 = it.tx * UQ_TW + (l_e >> 4);
            unsigned toffs[2];
            bool trow[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int ly = lyb + m;
                trow[m] = ly < a.Hin;
                toffs[m] = ((unsigned)(((2 * min(ly, a.Hin - 1) + pa) * a.Wout + 2 * lcol0 + pb) * C + 8 * (l_e & 15)) << 1);
            }
            char* obase = reinterpret_cast<char*>(a.out) + (size_t)it.img * a.Hout * a.Wout * (2 * C);
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, a.Hout * a.Wout * (2 * C), 0x00020000);
            const unsigned cstep = (unsigned)(2 * C) * 8u;                    // bytes per read-back's 4 low-res pixels = 8 output pixels
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
                    pkd[m][g] = u32x4_t{uq_pack(c[8 * pp + 0], c[8 * pp + 1]), uq_pack(c[8 * pp + 2], c[8 * pp + 3]),
                                        uq_pack(c[8 * pp + 4], c[8 * pp + 5]), uq_pack(c[8 * pp + 6], c[8 * pp + 7])};
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
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const bf16x2_t bv = __builtin_bit_cast(bf16x2_t, w[d]);
                            s1 = __builtin_amdgcn_fdot2_f32_bf16(bv, ones, s1, false);
                            q1 = __builtin_amdgcn_fdot2_f32_bf16(bv, bv, q1, false);
                        }
                        const bool ok = trow[m] && lcol0 + 8 * e + 4 * k < a.Win;
                        ssum += ok ? s1 : 0.f; qsum += ok ? q1 : 0.f;
                        __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, ok ? toffs[m] + (unsigned)(2 * e + k) * cstep : 0xffffffffu, 0, UQ_ST);
                    }
                }
            // this lane's chunk cc_r over its read-backs; the other lanes with the same chunk sit 16 apart
            ssum = uq_swap16_add(ssum); qsum = uq_swap16_add(qsum);
            ssum = uq_swap32_add(ssum); qsum = uq_swap32_add(qsum);
            if (l_e < 16) *reinterpret_cast<float2*>(red + red_par * (8 * 32) + (wave * 16 + cc_r) * 2) = make_float2(ssum, qsum);
            st_img = it.img; st_slot = it.tile * 4 + it.nb; st_par = red_par; red_par ^= 1;
        }
    }
    __syncthreads();                     // the last item's chunk sums
    flush_stats();
}


}  // namespace

// a.in0 = low-res input [img][in_rows][Win][Cin], a.in1 = skip [img][Hout][Wout][128] at its first real row, a.out likewise;
// a.tiles_x / tiles_y = 16 x 32 LOW-res tiles, a.nkc = Cin / 32, a.nblocks = 4 (parities); a.stats = partials [img][tile * 4 + parity][8][2].
void conv_upq_launch(const ConvArgs& a, hipStream_t stream) {
    if (a.cout != UQ_NT || a.cin1 != UQ_NT || a.cin0 != 2 * UQ_NT || a.nkc != a.cin0 / 32 || a.nblocks != 4 || !a.w || !a.w1 || !a.in1 || !a.zeros ||
        a.Hout != 2 * a.Hin || a.Wout != 2 * a.Win)
        fail(IRE_ERR_INTERNAL, "internal: conv_upq arguments");
    const int items = a.tiles_x * a.tiles_y * a.nimg * 4;
    const int cus = persistent_grid_cus();
    const int grid = items < cus ? items : cus;
    hipLaunchKernelGGL(conv_upq_kernel, dim3(grid), dim3(UQ_THREADS), 0, stream, a);
    IRE_HIP(hipGetLastError());
}

}  // namespace ire
