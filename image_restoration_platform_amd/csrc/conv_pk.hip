// conv_pk.hip -- the C >= 128 ResBlock convolutions (16 launches, a third of a step) as a PRODUCER / CONSUMER workgroup with
// 128-cout items: 12 waves per CU = three per SIMD, gfx950.  Round 4; conv_w4.hip stays as the A/B fallback (IRE_PK=0), for
// pre-activated inputs, for fp8 and for launches whose workgroups would span more images than the coefficient table holds.
//
// Why (profiles/r04_experiments.md, workgroup timelines + profiles/r03_experiments.md stamps): conv_w4's 16-channel stage takes
// ~6 400 ticks against 4 608 of matrix-pipe time -- its eight waves are the SAME program, each computing AND staging, and what a
// stage loses it loses in the five k-steps that carry the GroupNorm+FiLM+SiLU transform, the LDS tile writes, the weight-slab DMA
// and the prefetch loads between the MFMAs of two in-order waves per SIMD.  conv_pc.hip showed the cure at C <= 64 (two roles on
// every SIMD), and its C >= 128 trial (IRE_PC=7) showed what not to do: 64-cout items activate every input element once per 64
// couts, so the four producer waves (5 900 ticks per 32-channel tile) bound the item.  Here an item is 16 x 32 pixels x 128 couts:
//   waves 0..7  consumers: wave tile 2 rows x 128 couts (128 accumulators), a stage = 16 input channels = 9 k-steps (taps) of 8
//                MFMAs on 2 pixel + 4 weight fragments read from LDS -- NO vector-memory instruction, no transform, no LDS store
//                in the k-loop; conv_w4's line-coalesced epilogue through wave-private LDS patches that now have LDS of their own
//                (no rendezvous before the epilogue);
//   waves 8..11 producers: tile s + 1 while the consumers run stage s: raw bf16 chunks (registers, requested a stage ahead) ->
//                y = silu(x A + B) in plain f32, coefficients from an LDS table -> bf16 -> LDS tile (s + 1) & 1; the weight slab
//                of stage s + 1 by LDS-DMA into slab slot (s + 1) & 1 (TWO slots: a dedicated loader can afford to wait for its
//                DMA inside the stage, which conv_w4's computing waves could not -- that is where the 37 KB for the patches come from).
// One workgroup barrier per stage.  An input element is activated once per 128 couts (half of the producers' work per MFMA of
// conv_pc's C >= 128 trial): ~3 000 producer ticks per 4 608-tick stage.  168 registers per wave (three per SIMD): 128 accumulators
// + single-buffered fragments; the two consumer waves of a SIMD cover each other's fragment reads.
// Same tiles, weight slabs (engine.cpp::make_conv d_w4), accumulation order, epilogue arithmetic and partials layout as conv_w4's
// 8-wave fused form: bit-identical results (tests/test_restore_gpu.py holds the two schedules to equal bytes), row strips included.
// Roofline: MFMA.  `2*9*C*C` flop per output pixel, `4C` B (+ `2C` residual).
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

#ifndef PK_PRIO
#define PK_PRIO 1    // s_setprio of the producer waves (they are the youngest waves of their SIMD: at 0 they issue in the consumers' leftover slots)
#endif
#ifndef PK_DMA_SPLIT
#define PK_DMA_SPLIT 1   // 1: three slab DMA pieces in front of the first transform group, six behind it; 0: all nine at the top of the stage
#endif
#ifndef PK_BDEPTH
#define PK_BDEPTH 2  // weight-fragment registers of the consumers' rotation: 2 = the next fragment is read one MFMA pair ahead, 3 = two pairs ahead
#endif
#ifndef PK_CPRIO
#define PK_CPRIO 0   // 1: consumers raise their priority early in a stage (k-steps 0..3: 3, 4..6: 2, 7..8: 1 -- conv_w4's progress feedback: the wave that is behind outranks its partner)
#endif
#ifndef PK_DBUF
#define PK_DBUF 0    // 1: two producer register sets, a stage's rows requested two stages ahead; 0: one set, one stage ahead (vmcnt(0) at the top of a stage)
#endif
#ifndef PK_INTERIOR
#define PK_INTERIOR 1   // tiles that touch no image border skip the zero-padding masks (a wave-uniform branch per stage)
#endif
#ifndef PK_CDMA
#define PK_CDMA 0    // 1: the CONSUMER waves issue the slab DMA (4-5 one-KB pieces each, at the top of their stage, where they would otherwise wait
                     // for the producers); 0: the producers do (9 pieces each: ~900 of their ~5 000 ticks per stage)
#endif
#ifndef PK_TOUCH
#define PK_TOUCH 0   // residual variant: the producers pull the item's residual rows into L2 three stages before the epilogue
#endif
#ifndef PK_RING
#define PK_RING 2    // residual variant without PK_RDMA: passes of residual rows in flight in the epilogue
#endif
#ifndef PK_RDMA
#define PK_RDMA 1    // residual variant: the FIRST pass's residual rows (16 pixels x 256 B per wave) travel by LDS-DMA into the wave's own -- still
                     // idle -- transpose patch at the top of the item's last stage: no register waits for them through the k-loop, and the
                     // epilogue starts with its first residual rows in LDS; passes 1..3 then follow through two register sets, each
                     // requested a whole pass ahead (pass 1 at the epilogue's start, pass 2 behind pass 0, pass 3 behind pass 1)
#endif
#ifndef PK_ABL
#define PK_ABL 0     // timing ablations (results wrong by design): 1 no transform, 2 no epilogue, 4 no MFMA loop, 8 no slab DMA, 16 epilogue without its global stores
#endif

constexpr int PK_CONS = 512, PK_PROD = 256, PK_THREADS = PK_CONS + PK_PROD;
constexpr int PK_TH = 16, PK_TW = 32, PK_IH = 18, PK_IW = 34, PK_NT = 128, PK_NTL = 4;
constexpr int PK_IN_CHUNKS = PK_IH * PK_IW * 2;                     // 1224 x 16 B: 16 channels per pixel and stage, two k-half planes
constexpr int PK_P_ITERS = (PK_IN_CHUNKS + PK_PROD - 1) / PK_PROD;  // 5 chunks per producer thread and stage
constexpr int PK_PLANE = (PK_IN_CHUNKS / 2) * 16;                   // 9792: (PLANE / 4) % 32 == 16 -> conflict-free tile writes (conv_w4.hip)
constexpr int PK_IN_BYTES = (PK_IN_CHUNKS + 8) * 16;                // + dummy slot (chunk slots past the tile)
constexpr int PK_W_BYTES = 9 * 2 * PK_NT * 16;                      // slab [tap][c8][128][8 bf16] = 36 864
constexpr int PK_W_CHUNKS = PK_W_BYTES / 16;                        // 2304 = 9 DMA issues of 256 threads
constexpr int PK_W_BASE = 2 * PK_IN_BYTES;                          // in[2] | w[2] | patches[8] | red | bias | coef
constexpr int PK_PATCH_BASE = PK_W_BASE + 2 * PK_W_BYTES;
constexpr int PK_PATCH_BYTES = 16 * PK_NT * 2;                      // a consumer wave's transpose patch: 16 pixels x 128 couts
constexpr int PK_RED_BASE = PK_PATCH_BASE + 8 * PK_PATCH_BYTES;
constexpr int PK_RED_BYTES = 2 * 8 * 16 * 2 * 4;                    // [item parity][8 waves][16 chunks of 8 couts][sum, sumsq]
constexpr int PK_BIAS_BASE = PK_RED_BASE + PK_RED_BYTES;
constexpr int PK_COEF_BASE = PK_BIAS_BASE + 256 * 4;
constexpr int PK_IMGS = 4;                                          // images whose (A, B) the coefficient table holds
template <int C> struct PkCfg {
    static constexpr int SINK = PK_COEF_BASE + PK_IMGS * C * 8;      // 256 B nobody reads: where the residual prefetch lands
    static constexpr int LDS = SINK + 256;
    static_assert(LDS <= 160 * 1024, "LDS");
};
static_assert((PK_PLANE / 4) % 32 == 16 && PK_W_CHUNKS == 9 * PK_PROD, "layout");

__device__ __forceinline__ unsigned pk_pack(float a, float b) {
    f32x2_t f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ float pk_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float pk_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ void pk_glds16(const void* gsrc, unsigned lds_dst_uniform) {   // LDS-DMA, 1 KB per wave-instruction (conv_rb.hip)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
// one dword per lane into a 256-B LDS sink: a load whose only purpose is to pull its 128-B line into the XCD's L2 (no register waits for it)
__device__ __forceinline__ void pk_touch(const void* gsrc, unsigned lds_sink_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_sink_uniform) : "memory");
}
template <int N> __device__ __forceinline__ float pk_ror_add(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float pk_swap16_add(float v) {
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float pk_swap32_add(float v) {
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
// only LDS traffic is ordered by the stage barrier; global loads, stores and LDS-DMA stay in flight across it
__device__ __forceinline__ void pk_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int C, bool RESID>
__global__ __launch_bounds__(PK_THREADS) void conv_pk_kernel(ConvArgs a) {
    using K = PkCfg<C>;
    constexpr int NKC = C / 16, NBLK = C / PK_NT, NT = PK_NT, NTL = PK_NTL;
    __shared__ __attribute__((aligned(16))) unsigned char smem[K::LDS];
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const int tiles_per_img = a.tiles_x * a.tiles_y;
    PersistCursor cursor(a.tiles_x, a.tiles_y, a.nimg, NBLK, NKC);
    const int n_items = cursor.my_items, S = cursor.S;
    if (S == 0) return;
    const bool producer = __builtin_amdgcn_readfirstlane(wave) >= 8;
    // diagnostic build (-DPK_TICKS, tools/r04_pkstamps.sh): s_memtime stamps of consumer wave 0 (role 0) and producer wave 8 (role 1) of
    // the first 8 workgroups, stages 0..31: a.stamps[((wg * 2 + role) * 32 + stage) * 4 + k]
    auto stamp = [&](int role, int stage, int k) {
#ifdef PK_TICKS
        if (a.stamps && lane == 0 && blockIdx.x < 8 && stage < 32) a.stamps[(((size_t)blockIdx.x * 2 + role) * 32 + stage) * 4 + k] = __builtin_amdgcn_s_memtime();
#else
        (void)role; (void)stage; (void)k;
#endif
    };
    // workgroup TIMELINE (diagnostic build -DIRE_W4_TL; layout as conv_w4.hip): entry (0), fold done (1), first tile staged (2), item k done (3 + k), exit (12)
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
    // ... and, for workgroup 0, every wave's arrival at / release from the barrier that ends stages 0..15: a.stamps[4096 + ((stage * 12 + wave) * 2 + k)]
    auto wstamp = [&](int stage, int k) {
#ifdef PK_TICKS
        if (a.stamps && lane == 0 && blockIdx.x == 0 && stage < 16) a.stamps[4096 + (stage * 12 + wave) * 2 + k] = __builtin_amdgcn_s_memtime();
#else
        (void)stage; (void)k;
#endif
    };
    const int tp = tid - PK_CONS;                          // producer thread 0..255
    const int c8_fixed = tp & 1;                           // this thread always stages the same 8-channel half of a pixel

    // ---- producers' input addressing: a thread's chunk i (idx = tp + 256 i: pixel idx >> 1 of the 18 x 34 halo tile, half c8_fixed)
    // sits at the same tile position in every stage of an item: offsets once per item, a stage's request is base(item, kc) + offset
    unsigned coff[PK_P_ITERS], cok = 0;
    auto item_offsets = [&](const PersistItem& it) {
        const int oy1 = it.ty * PK_TH - 1, ox1 = it.tx * PK_TW - 1;
        cok = 0;
#pragma unroll
        for (int i = 0; i < PK_P_ITERS; ++i) {
            int t2 = tp;
            asm volatile("" : "+v"(t2));
            const int p = (t2 + i * PK_PROD) >> 1;
            const int py = p / PK_IW, px = p - py * PK_IW;
            const int iy = oy1 + py, ix = ox1 + px;
            const int cy = min(max(iy, a.iy_lo), a.iy_lo + a.iy_span - 1), cx = min(max(ix, 0), a.Win - 1);
            const bool ok = iy == cy && ix == cx && py < PK_IH;
            coff[i] = ((unsigned)((cy + a.in_row_off) * a.Win + cx) * (unsigned)(2 * C)) + (unsigned)(c8_fixed * 16);
            cok |= ok ? (1u << i) : 0u;
        }
    };
    // TWO register sets: stage s waits in set s & 1, requested TWO stages before it is transformed -- the transform never waits for
    // its own stage's loads (one set, requested a stage ahead: ~340 ticks of vmcnt(0) at the top of every producer stage)
    u32x4_t RA[PK_P_ITERS], RB[PK_P_ITERS];
    unsigned rokA = 0, rokB = 0;                            // bit i: the chunk in R[i] lies inside the image (travels with the data)
    auto load_chunk = [&](u32x4_t (&R)[PK_P_ITERS], const PersistStage& st, int i) __attribute__((always_inline)) {
        const char* base = reinterpret_cast<const char*>(a.in0) + (size_t)st.it.img * a.in_rows * a.Win * (2 * C) + st.kc * 32;
        R[i] = *reinterpret_cast<const u32x4_t*>(base + coff[i]);
    };
    auto wslab = [&](const PersistStage& st) -> const unsigned char* {
        return reinterpret_cast<const unsigned char*>(a.w) + ((size_t)st.it.nb * NKC + st.kc) * PK_W_BYTES;
    };
    auto dma_slab = [&](const PersistStage& st, int slot, int d0 = 0, int d1 = 9) __attribute__((always_inline)) {     // pieces d0 .. d1 - 1 of the 9 x 1 KB per producer wave
        if constexpr (PK_ABL & 8) return;
        const unsigned char* ws = wslab(st);
        const int wave_p = __builtin_amdgcn_readfirstlane(wave) - 8;
        const unsigned dst = smem_lds + PK_W_BASE + slot * PK_W_BYTES;
#pragma unroll
        for (int d = d0; d < d1; ++d) {
            const int cbase = d * PK_PROD + wave_p * 64;
            pk_glds16(ws + (size_t)(cbase + lane) * 16, dst + cbase * 16);
        }
    };
    PersistStage q1 = cursor.cur, q2 = q1, q3 = q1;         // producers: the next stages to transform (q1) .. to request (q3)

    // ---- before the folded GroupNorm finalize: everything that does not need its result is already on its way -- the first two
    // stages' raw rows (registers) and the first weight slab (LDS-DMA into slot 0; gn_fold's scratch is the first 8 KB of the tile area)
    if (producer) {
        item_offsets(q1.it);
#pragma unroll
        for (int i = 0; i < PK_P_ITERS; ++i) load_chunk(RA, q1, i);
        rokA = cok;
        dma_slab(q1, 0);
        q2 = cursor.next();
        if (PK_DBUF) {
            if (q2.kc == 0) item_offsets(q2.it);
#pragma unroll
            for (int i = 0; i < PK_P_ITERS; ++i) load_chunk(RB, q2, i);
            rokB = cok;
        }
        q3 = cursor.next();
    }
    tl(0);
    if (a.gn_stats) gn_fold(a, smem, cursor.first_img, cursor.last_img, 512, reinterpret_cast<float2*>(smem + PK_COEF_BASE));     // the coefficient table, straight into LDS
    tl(1);
    {
        if (tid < C) reinterpret_cast<float*>(smem + PK_BIAS_BASE)[tid] = a.bias[tid];
        // the GroupNorm+FiLM coefficients of the images this workgroup's items belong to (gn_fold just wrote them, or
        // gn_finalize_kernel did): the producers read them from LDS -- their only vector-memory traffic is the input stream and the slabs
        const int nim = cursor.last_img - cursor.first_img + 1;                // <= PK_IMGS (conv_pk_fits)
        const float2* ab = a.ab + (size_t)cursor.first_img * C;
        float2* cd = reinterpret_cast<float2*>(smem + PK_COEF_BASE);
        if (!a.gn_stats) for (int i = tid; i < nim * C; i += PK_THREADS) cd[i] = ab[i];      // (gn_finalize_kernel ran: row strips, IRE_GN_FOLD=0)
    }
    __syncthreads();

    if (producer) {
        // =============================== producers: waves 8..11 ===============================================================
        if (PK_PRIO) asm volatile("s_setprio %0" :: "n"(PK_PRIO));
        unsigned pastbits = 0;                                   // bit i: this thread's chunk i is a slot past the tile (idx >= 1224): its value is never read
#pragma unroll
        for (int i = 0; i < PK_P_ITERS; ++i) pastbits |= (tp + i * PK_PROD >= PK_IN_CHUNKS) ? (1u << i) : 0u;
        float cA[8], cB[8];
        auto load_coeffs = [&](const PersistStage& st) __attribute__((always_inline)) {      // (A, B) of this thread's 8 channels
            const float4* ab = reinterpret_cast<const float4*>(smem + PK_COEF_BASE) + ((st.it.img - cursor.first_img) * C + st.kc * 16 + c8_fixed * 8) / 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float4 v = ab[e]; cA[2 * e] = v.x; cB[2 * e] = v.y; cA[2 * e + 1] = v.z; cB[2 * e + 1] = v.w; }
        };
        // NCH chunks (8 NCH channel values) move through the transform STAGE BY STAGE (independent v_exp, then independent v_rcp: a
        // single in-order wave hides a transcendental's latency only behind its own independent work: conv_pc.hip); plain f32
        // instructions (-fno-slp-vectorize: beside MFMAs a packed f32 instruction costs several plain ones)
        auto transform_group = [&](u32x4_t (&R)[PK_P_ITERS], auto i_tag, auto n_tag, auto interior_tag, unsigned char* tile, const PersistStage& nxt, unsigned okbits) __attribute__((always_inline)) {
            constexpr bool interior = decltype(interior_tag)::value;
            constexpr int i0 = decltype(i_tag)::value, NCH = decltype(n_tag)::value, NW = 4 * NCH, NV = 8 * NCH;
            unsigned wds[NW], o[NW];
#pragma unroll
            for (int k = 0; k < NCH; ++k) { wds[4 * k] = R[i0 + k].x; wds[4 * k + 1] = R[i0 + k].y; wds[4 * k + 2] = R[i0 + k].z; wds[4 * k + 3] = R[i0 + k].w; }
            if constexpr (PK_ABL & 1) {
#pragma unroll
                for (int w = 0; w < NW; ++w) o[w] = wds[w];
            } else {
                float y[NV], e[NV];
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const int d = w & 3;
                    y[2 * w] = __builtin_fmaf(pk_lo(wds[w]), cA[2 * d], cB[2 * d]);
                    y[2 * w + 1] = __builtin_fmaf(pk_hi(wds[w]), cA[2 * d + 1], cB[2 * d + 1]);
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
                for (int w = 0; w < NW; ++w) o[w] = pk_pack(y[2 * w] * e[2 * w], y[2 * w + 1] * e[2 * w + 1]);
            }
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                int t2 = tp;
                asm volatile("" : "+v"(t2));
                const int idx = t2 + (i0 + k) * PK_PROD;
                u32x4_t ov = {o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]};
                if constexpr (!interior) {                                                // (tiles that touch no image border skip the masks: two copies of the stage under a wave-uniform branch)
                    const unsigned m = (okbits >> (i0 + k)) & 1u ? 0xffffffffu : 0u;      // zero padding applies AFTER the activation
                    ov = u32x4_t{o[4 * k] & m, o[4 * k + 1] & m, o[4 * k + 2] & m, o[4 * k + 3] & m};
                }
                const int slot = idx < PK_IN_CHUNKS ? c8_fixed * (PK_IN_CHUNKS / 2) + (idx >> 1) : PK_IN_CHUNKS;
                reinterpret_cast<u32x4_t*>(tile)[slot] = ov;
                load_chunk(R, nxt, i0 + k);                                              // the same registers take the chunk of the stage after next
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // one producer stage: register set R (stage `ps`, landed) -> tile; R <- stage `nxt` (two stages on)
        // slab_slot >= 0 (PK_CDMA = 0): the DMA of ps's weight slab rides along (PK_DMA_SPLIT: three pieces in front, six behind the first group)
        auto produce = [&](u32x4_t (&R)[PK_P_ITERS], unsigned& rok, const PersistStage& ps, unsigned char* tile, const PersistStage& nxt, int slab_slot) __attribute__((always_inline)) {
            // this set's loads are older than the 5 requests of the stage in between: everything but the 5 youngest operations has landed
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PK_DBUF ? PK_P_ITERS : 0) : "memory");
#pragma unroll
            for (int i = 0; i < PK_P_ITERS; ++i) asm volatile("" : "+v"(R[i]));
            if (!PK_CDMA && slab_slot >= 0) dma_slab(ps, slab_slot, 0, PK_DMA_SPLIT ? 3 : 9);
            load_coeffs(ps);
            const unsigned okbits = rok;
            if (nxt.kc == 0) item_offsets(nxt.it);       // (past the queue's end the cursor stays on the last stage: kc != 0)
            // every real chunk of every lane of this wave inside the image: one scalar test per stage (the slots past the tile are never read)
            const bool interior = PK_INTERIOR && __builtin_amdgcn_ballot_w64((okbits | pastbits) != 0x1fu) == 0;
            auto groups = [&](auto interior_tag) __attribute__((always_inline)) {
                transform_group(R, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, interior_tag, tile, nxt, okbits);
                if (!PK_CDMA && PK_DMA_SPLIT && slab_slot >= 0) dma_slab(ps, slab_slot, 3, 9);
                transform_group(R, std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, interior_tag, tile, nxt, okbits);
                transform_group(R, std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{}, interior_tag, tile, nxt, okbits);
            };
            if (interior) groups(std::true_type{}); else groups(std::false_type{});
            rok = cok;
        };
        // stage 0 (set A) -> tile 0 (its slab is in slot 0 already), set A <- stage 2
        PersistStage cc = q1;                                       // the stage the consumers are in
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the slab of stage 0 among them)
        produce(RA, rokA, q1, smem, PK_DBUF ? q3 : q2, -1);
        q1 = q2; q2 = q3; q3 = cursor.next();
        pk_barrier();                                              // tile 0 and slab 0 are staged
        // the consumers run stage t; this is stage t + 1 (for t + 1 == S: the last stage again, into slots nobody reads any more): set
        // (t + 1) & 1 -> tile (t + 1) & 1, the set then takes stage t + 3.  S is even (NKC is): the loop runs stage pairs.
        auto body = [&](int t, u32x4_t (&R)[PK_P_ITERS], unsigned& rok, int slot) __attribute__((always_inline)) {
            if (wave == 8) stamp(1, t, 0);
            if constexpr (RESID) {
                // the residual rows of the item the consumers are in, touched (one dword per 128-B line, 1 024 lines) three stages before
                // its epilogue (PK_TOUCH; off: measured slower -- 32 CUs x 128 KB is the XCD's whole L2)
                if (PK_TOUCH && cc.kc == NKC - 3) {
                    const char* rbase = reinterpret_cast<const char*>(a.resid) + (size_t)cc.it.img * a.Hout * a.Wout * (2 * C) + (size_t)cc.it.nb * NT * 2;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = tp + j * PK_PROD;                      // line k: tile row k >> 6, 128-B segment k & 63 of the row's 32 x 256 B
                        const int oy = min(cc.it.ty * PK_TH + (k >> 6), a.Hout - 1), ox = min(cc.it.tx * PK_TW + ((k & 63) >> 1), a.Wout - 1);
                        pk_touch(rbase + ((size_t)oy * a.Wout + ox) * (2 * C) + (k & 1) * 128, smem_lds + K::SINK);
                    }
                }
            }
            if (wave == 8) stamp(1, t, 1);
            // (PK_CDMA = 0: slot (t + 1) & 1 held slab t - 1: free since the last barrier)
            produce(R, rok, q1, smem + slot * PK_IN_BYTES, PK_DBUF ? q3 : q2, slot);
            cc = q1;
            q1 = q2; q2 = q3; q3 = cursor.next();
            if (wave == 8) stamp(1, t, 2);
            // PK_CDMA = 0: the slab must have landed before the barrier; the requests that follow its last DMA piece may stay in flight
            if (!PK_CDMA) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PK_DMA_SPLIT ? 3 : PK_P_ITERS) : "memory");
            if (wave == 8) stamp(1, t, 3);
            wstamp(t, 0);
            pk_barrier();
            wstamp(t, 1);
        };
        for (int t = 0; t < S; t += 2) {
            if (PK_DBUF) body(t, RB, rokB, 1); else body(t, RA, rokA, 1);
            body(t + 1, RA, rokA, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may be in flight when the workgroup's LDS is released
        return;
    }

    // =================================== consumers: waves 0..7 ======================================================================
    // Lane (r, h) reads pixel p = (2 wave + m + ky) 34 + r + kx of plane h: ONE address register + immediates; weight fragment
    // (tap, half h, rows 32 j + r): (2 tap NT + h NT + 32 j + r) x 16 B
    const int a_base = h * PK_PLANE + (wave * 2 * PK_IW + r) * 16;
    const int b_off = (h * NT + r) * 16;
    const float* bias_lds = reinterpret_cast<const float*>(smem + PK_BIAS_BASE);
    float* red = reinterpret_cast<float*>(smem + PK_RED_BASE);
    unsigned char* patch = smem + PK_PATCH_BASE + wave * PK_PATCH_BYTES;

    int st_img = -1, st_tile = 0, st_cout0 = 0, st_par = 0, red_par = 0;
    auto flush_stats = [&]() {          // GroupNorm partials of the item that finished before the last stage barrier: 8 waves x 16 chunk slots -> groups
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
    PersistStage cn = cursor.next();                               // the stage after `cs` (clamped at the end of the queue)
    int par = 0;
    int stage_no = 0;
    // (timeline slot 2 is stamped behind the first stage barrier below)
    // PK_CDMA: the slab of stage s + 1 is requested by the consumers at the top of stage s into slot (s + 1) & 1 (free since the barrier they
    // just passed): 36 one-KB pieces, wave w takes pieces w, w + 8, w + 16, w + 24 (and w + 32 for w < 4).  LDS-DMA needs no destination
    // registers; the pieces land during the k-loop and the wait behind it is free.
    auto consumer_dma = [&](const PersistStage& st, int slot) __attribute__((always_inline)) {
        if constexpr (!PK_CDMA || (PK_ABL & 8)) return;
        const unsigned char* ws = wslab(st);
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const unsigned dst = smem_lds + PK_W_BASE + slot * PK_W_BYTES;
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            const int piece = wv + 8 * d;
            if (piece < PK_W_CHUNKS / 64) pk_glds16(ws + (size_t)(piece * 64 + lane) * 16, dst + piece * 1024);
        }
    };
    pk_barrier();                                                  // tile 0 and slab 0 are staged
    tl(2);
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
#pragma unroll 1
        for (int kc = 0; kc < NKC; ++kc) {
            const unsigned char* ib = smem + par * PK_IN_BYTES + a_base;
            const unsigned char* wb = smem + PK_W_BASE + par * PK_W_BYTES + b_off;
            if (wave == 0) stamp(0, stage_no, 0);
            if (stage_no + 1 < S) consumer_dma(cn, par ^ 1);
            if constexpr (RESID && PK_RDMA) {
                if (kc == NKC - 1) {
                    int l_e = lane, w_e = wave;
                    asm volatile("" : "+v"(l_e), "+v"(w_e));
                    const int oy = min(it.ty * PK_TH + w_e * 2, a.Hout - 1);                       // pass 0 = the wave's first row, its first 16 pixels
                    const char* rbase = reinterpret_cast<const char*>(a.resid) + (size_t)it.img * a.Hout * a.Wout * (2 * C) + (size_t)(cout0 + 8 * (l_e & 15)) * 2;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {                                                    // read-back k: pixels 4 k + (lane >> 4), this lane's 8-cout chunk
                        const int ox = min(it.tx * PK_TW + 4 * k + (l_e >> 4), a.Wout - 1);         // (a column past the image: any valid address, the value is never used)
                        pk_glds16(rbase + ((size_t)oy * a.Wout + ox) * (2 * C), smem_lds + PK_PATCH_BASE + __builtin_amdgcn_readfirstlane(wave) * PK_PATCH_BYTES + k * 1024);
                    }
                }
            }
            if constexpr (!(PK_ABL & 4)) {
                // 9 k-steps (taps) x 4 weight fragments x 2 pixel rows.  Fragments rotate through TWO weight registers and two PAIRS of
                // pixel registers (24 of the 40 registers the accumulators leave): the read of the next weight fragment is issued in
                // front of the two MFMAs of the current one, the next k-step's pixel fragments during this k-step's second and third
                // pair -- every ds_read_b128 has at least one MFMA pair (64 matrix-pipe cycles, plus whatever the SIMD's other consumer
                // wave issues in between) to land.  The sched_barriers pin that order (left alone, hipcc reads each weight fragment
                // right before its MFMAs and waits lgkmcnt(0): ~100 exposed cycles per pair).
                auto rd_b = [&](int st, int j) __attribute__((always_inline)) -> bf16x8_t {
                    return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wb + (2 * st * NT + j * 32) * 16));
                };
                auto rd_a = [&](int st, int m) __attribute__((always_inline)) -> bf16x8_t {
                    const int ky = st / 3, kx = st - ky * 3;
                    return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ib + ((m + ky) * PK_IW + kx) * 16));
                };
                constexpr int BD = PK_BDEPTH;                      // weight-fragment registers in rotation
                constexpr int NG = 9 * NTL;                        // weight fragments (= MFMA pairs) of a stage
                auto rd_bg = [&](int g) __attribute__((always_inline)) -> bf16x8_t { return rd_b(g / NTL, g % NTL); };
                bf16x8_t bq[BD], aq[2][2];
#pragma unroll
                for (int d = 0; d + 1 < BD; ++d) bq[d] = rd_bg(d);
                aq[0][0] = rd_a(0, 0); aq[0][1] = rd_a(0, 1);
#pragma unroll
                for (int st = 0; st < 9; ++st)
#pragma unroll
                    for (int j = 0; j < NTL; ++j) {
                        const int g = st * NTL + j;
                        if (PK_CPRIO && j == 0 && st == 0) __builtin_amdgcn_s_setprio(3);
                        if (PK_CPRIO && j == 0 && st == 4) __builtin_amdgcn_s_setprio(2);
                        if (PK_CPRIO && j == 0 && st == 7) __builtin_amdgcn_s_setprio(1);
                        if (g + BD - 1 < NG) bq[(g + BD - 1) % BD] = rd_bg(g + BD - 1);
                        if (st + 1 < 9 && j == 1) aq[(st + 1) & 1][0] = rd_a(st + 1, 0);
                        if (st + 1 < 9 && j == 2) aq[(st + 1) & 1][1] = rd_a(st + 1, 1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[g % BD], aq[st & 1][m], acc[m][j], 0, 0, 0);   // D[cout][pixel]
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
            par ^= 1;
            if (PK_CDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's slab pieces (issued a whole k-loop ago) are in LDS
            if (wave == 0) stamp(0, stage_no, 1);
            ++stage_no;
            if (kc + 1 < NKC) { cs = cn; cn = cursor.next(); wstamp(stage_no - 1, 0); pk_barrier(); wstamp(stage_no - 1, 1); }     // the item's next stage: its tile and slab are staged, these are free
        }
        // ---- epilogue: conv_w4's line-coalesced form -- a wave transposes 16 pixels x 128 couts at a time through its own 4-KB patch
        // (XOR-swizzled: conflict-free both ways) and reads / writes global memory in whole 256-B pixel runs, 1 KB per instruction
        if constexpr (PK_ABL & 2) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < NTL; ++j) asm volatile("" :: "v"(acc[m][j]));
        } else {
            int l_e = lane, w_e = wave;
            asm volatile("" : "+v"(l_e), "+v"(w_e));
            const int oyb = it.ty * PK_TH + w_e * 2;
            const int tcol0 = it.tx * PK_TW + (l_e >> 4);
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
            uint4 rvt[PK_RDMA ? 2 : PK_RING][4];                                            // residual rows of pass (m, q) in slot pass % PK_RING
            auto load_resid_pass = [&](int pass, uint4 (&dst)[4]) __attribute__((always_inline)) {
                if constexpr (RESID) {
                    char* rbase = const_cast<char*>(reinterpret_cast<const char*>(a.resid)) + (size_t)it.img * a.Hout * a.Wout * (2 * C);
                    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(rbase, 0, a.Hout * a.Wout * (2 * C), 0x00020000);
                    const int m = pass >> 1, q = pass & 1;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool ok = trow[m] && tcol0 + 16 * q + 4 * k < a.Wout;
                        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ok ? toffs[m] + (unsigned)(4 * q + k) * cstep : 0xffffffffu, 0, 0);
                        dst[k] = make_uint4(v.x, v.y, v.z, v.w);
                    }
                }
            };
            uint4 r0[4];                                                      // PK_RDMA: pass 0's residual rows, from the patch
            if constexpr (RESID && PK_RDMA) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the DMA of the last stage's top: landed a k-loop ago
#pragma unroll
                for (int k = 0; k < 4; ++k) r0[k] = *reinterpret_cast<const uint4*>(patch + k * 1024 + l_e * 16);
                load_resid_pass(1, rvt[1]);
            } else if constexpr (RESID) {
#pragma unroll
                for (int p0 = 0; p0 < PK_RING; ++p0) load_resid_pass(p0, rvt[p0]);
            }
            const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
            const int r16 = l_e & 15, qh = (l_e >> 4) & 1;                    // writer: pixel r = 16 qh + r16 of the row, half h
            const int h_e = l_e >> 5;
            const int pq = l_e >> 4, cc_r = l_e & 15;                         // reader: pixel 4 k + pq of the half-row, chunk cc_r (8 couts)
            constexpr int PITCH = NT * 2;
            // all 128 accumulators to packed bf16 FIRST (the same 64 conversions the passes would do one by one): 64 registers come
            // free at once -- the residual rows of the next passes wait in them instead of in scratch
            u32x4_t pkd[2][NTL * 2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < NTL * 2; ++g) {
                    const f32x16_t& c = acc[m][g >> 1];
                    const int pp = g & 1;
                    pkd[m][g] = u32x4_t{pk_pack(c[8 * pp + 0], c[8 * pp + 1]), pk_pack(c[8 * pp + 2], c[8 * pp + 3]),
                                        pk_pack(c[8 * pp + 4], c[8 * pp + 5]), pk_pack(c[8 * pp + 6], c[8 * pp + 7])};
                }
            __builtin_amdgcn_sched_barrier(0);
            float ssum = 0.f, qsum = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (qh == q) {
#pragma unroll
                        for (int g = 0; g < NTL * 2; ++g) {
                            const int cc = 2 * g + h_e;                       // chunk of the pixel's 128-cout run: couts 8 cc .. 8 cc + 7
                            *reinterpret_cast<u32x4_t*>(patch + r16 * PITCH + ((cc ^ r16) << 4)) = pkd[m][g];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int p = 4 * k + pq;
                        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(patch + p * PITCH + ((cc_r ^ p) << 4));
                        unsigned w[4] = {v.x, v.y, v.z, v.w};
                        if constexpr (RESID) {
                            const uint4 rr = PK_RDMA ? ((2 * m + q) == 0 ? r0[k] : rvt[(2 * m + q) & 1][k]) : rvt[(2 * m + q) % PK_RING][k];
                            const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                            for (int d = 0; d < 4; ++d) w[d] = pk_pack(pk_lo(w[d]) + pk_lo(rw[d]), pk_hi(w[d]) + pk_hi(rw[d]));
                        }
                        float s1 = 0.f, q1 = 0.f;
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const bf16x2_t bv = __builtin_bit_cast(bf16x2_t, w[d]);
                            s1 = __builtin_amdgcn_fdot2_f32_bf16(bv, ones, s1, false);
                            q1 = __builtin_amdgcn_fdot2_f32_bf16(bv, bv, q1, false);
                        }
                        const bool ok = trow[m] && tcol0 + 16 * q + 4 * k < a.Wout;
                        ssum += ok ? s1 : 0.f; qsum += ok ? q1 : 0.f;
                        const u32x4_t o4 = {w[0], w[1], w[2], w[3]};
                        if constexpr (PK_ABL & 16) asm volatile("" :: "v"(o4));
                        else __builtin_amdgcn_raw_buffer_store_b128(o4, orsrc, ok ? toffs[m] + (unsigned)(4 * q + k) * cstep : 0xffffffffu, 0, IRE_ST_LINE);
                    }
                    if constexpr (RESID && PK_RDMA) { if (2 * m + q + 2 < 4) load_resid_pass(2 * m + q + 2, rvt[(2 * m + q) & 1]); }      // pass p + 2 into the set pass p (or nobody) used
                    else if constexpr (RESID) { if (2 * m + q + PK_RING < 4) load_resid_pass(2 * m + q + PK_RING, rvt[(2 * m + q) % PK_RING]); }
                }
            // this lane's chunk cc_r over its read-backs; the other lanes with the same chunk sit 16 apart
            ssum = pk_swap16_add(ssum); qsum = pk_swap16_add(qsum);
            ssum = pk_swap32_add(ssum); qsum = pk_swap32_add(qsum);
            if (l_e < 16) *reinterpret_cast<float2*>(red + red_par * (8 * 32) + (wave * 16 + cc_r) * 2) = make_float2(ssum, qsum);
            st_img = it.img; st_tile = it.tile; st_cout0 = cout0; st_par = red_par; red_par ^= 1;
        }
        if (wave == 0) stamp(0, stage_no - 1, 2);          // epilogue done
        cs = cn; cn = cursor.next();
        wstamp(stage_no - 1, 0);
        pk_barrier();                          // stage barrier: the next item's first tile and slab are staged; every wave's chunk sums are in LDS
        wstamp(stage_no - 1, 1);
        flush_stats();
        tl(3 + (t < 8 ? t : 8));
    }
    tl(12);
}

}  // namespace

// Every workgroup's items must stay within PK_IMGS images (the producers' coefficient table): the workgroups of XCD group x walk
// items [items x / X, items (x + 1) / X) (persist.hpp).
bool conv_pk_fits(int C, int tiles_per_img, int nimg) {
    if (C != 128 && C != 256) return false;
    if (nimg <= PK_IMGS) return true;
    const long long ipi = (long long)tiles_per_img * (C / PK_NT), items = ipi * nimg;
    const int cus = persistent_grid_cus();
    const long long G = items < cus ? items : cus, X = G < 8 ? G : 8;
    for (long long x = 0; x < X; ++x) {
        const long long lo = items * x / X, hi = items * (x + 1) / X;
        if (hi > lo && (hi - 1) / ipi - lo / ipi + 1 > PK_IMGS) return false;
    }
    return true;
}

// a.w = conv_w4's slabs [n-block of 128 couts][kc16][tap][c8][128][8] (engine.cpp::make_conv d_w4), a.nkc = C / 16, a.nblocks = C / 128,
// a.ab required (fused activation), 16 x 32 tiles, a.stats = partials [img][tile][8][2].
void conv_pk_launch(bool resid, const ConvArgs& a, hipStream_t stream) {
    const int C = a.cout;
    if ((C != 128 && C != 256) || a.cin0 != C || a.nkc != C / 16 || a.nblocks != C / PK_NT || !a.ab)
        fail(IRE_ERR_INTERNAL, "internal: conv_pk arguments");
    if (!conv_pk_fits(C, a.tiles_x * a.tiles_y, a.nimg)) fail(IRE_ERR_INTERNAL, "internal: conv_pk batch");
    const int items = a.tiles_x * a.tiles_y * a.nimg * a.nblocks;
    const int cus = persistent_grid_cus();
    const int grid = items < cus ? items : cus;
#define PK_GO(CC, RS) hipLaunchKernelGGL((conv_pk_kernel<CC, RS>), dim3(grid), dim3(PK_THREADS), 0, stream, a)
    if (C == 128) { if (resid) PK_GO(128, true); else PK_GO(128, false); }
    else { if (resid) PK_GO(256, true); else PK_GO(256, false); }
#undef PK_GO
    IRE_HIP(hipGetLastError());
}

}  // namespace ire
