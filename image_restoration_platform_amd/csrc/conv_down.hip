// conv_down.hip -- the encoder's stride-2 3x3 convolutions (RestoreNet-v0 `down` layers, C -> 2C) on the persistent pipelined
// schedule, gfx950.
//
// A stride-2 convolution read straight from an NHWC tile makes lane r fetch pixel 2r + kx: a 2-pixel lane stride that the
// LDS banks serve two-way (13.9 % conflict cycles in the v1 kernel).  Here the input is taken apart by PIXEL PHASE instead:
// with in[2Y + a][2X + b] =: P_ab[Y][X] the layer is a unit-stride convolution over the four half-resolution phase images,
//     out[Y][X] = sum over phases (a, b), offsets dy in D(a), dx in D(b) of  W[ky(a,dy)][kx(b,dx)] . P_ab[Y + dy][X + dx],
//     D(0) = {0} (ky = 1),   D(1) = {-1, 0} (ky = 0, 2)
// i.e. phase (0,0) carries 1 tap, (0,1) and (1,0) two, (1,1) four: 9 taps, no wasted K.  The phase images are never
// materialised: staging simply reads row 2(Y0-1+py)+a, column 2(X0-1+px)+b of the NHWC tensor into a 17x33 halo tile, and
// every fragment read is conv_rb.hip's conflict-free unit-stride pattern.
//
// Schedule: conv_rb.hip's (512-thread workgroup per CU, two LDS input tiles, input register-prefetched two stages ahead, weights
// by LDS-DMA two stages ahead into a ring of three slabs, one barrier per stage, counted waits: see `stage`).  Item = (16x32 OUTPUT tile, 64-cout block); a stage = 32 input channels of ONE phase
// (4 * Cin/32 stages per item, 2 k-steps per tap).  Epilogue as the ResBlock convs': bias-initialised accumulators, bf16
// stores straight from the accumulators (permuted slab rows), GroupNorm partial statistics per tile.
// Roofline: input staging (the layer reads the full-resolution tensor: 4 staged pixels per output pixel), then MFMA.
#include "conv_mfma.hpp"
#include "persist.hpp"

namespace ire {

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

#ifndef DN_LD_NT
#define DN_LD_NT 0     // 1: the input rows non-temporal (build-time A/B)
#endif
constexpr int DN_THREADS = 512;
constexpr int DN_TH = 16, DN_TW = 32;                  // output tile
constexpr int DN_IH = DN_TH + 1, DN_IW = DN_TW + 1;    // phase-image halo tile: offsets -1 .. 0
constexpr int DN_IN_CHUNKS = DN_IH * DN_IW * 4;         // 2244 x 16 B
constexpr int DN_IN_ITERS = (DN_IN_CHUNKS + DN_THREADS - 1) / DN_THREADS;    // 5
constexpr int DN_IN_BYTES = DN_IN_ITERS * DN_THREADS * 16;                   // 40960
constexpr int DN_NT = 64, DN_NTL = 2;
constexpr int DN_W_BYTES_MAX = 4 * 4 * DN_NT * 16;     // phase (1,1): 4 taps x 4 c8 x 64 rows x 16 B = 16 KB
constexpr int DN_W_BASE = 2 * DN_IN_BYTES;             // LDS: in[2] | w[3] | partial statistics | bias
constexpr int DN_MAIN = DN_W_BASE + 3 * DN_W_BYTES_MAX;
constexpr int DN_RED_HALF = 8 * (DN_NT / 8) * 4 * 4;
constexpr int DN_LDS = DN_MAIN + 2 * DN_RED_HALF + 256 * 4;
static_assert(DN_LDS <= 160 * 1024, "LDS");

__device__ __forceinline__ unsigned dn_pack(float a, float b) {
    f32x2_t f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ void dn_glds16(const void* gsrc, unsigned lds_dst_uniform) {   // see conv_rb.hip::rb_glds16
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
template <int N> __device__ __forceinline__ float dn_ror_add(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float dn_swap16_add(float v) {
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}

// taps of phase p = a*2 + b, in slab order: tap index t = (dy index)*ntx + (dx index); offsets dy, dx in {-1, 0}
__host__ __device__ constexpr int dn_ntaps(int p) { return ((p >> 1) ? 2 : 1) * ((p & 1) ? 2 : 1); }
__host__ __device__ constexpr int dn_slab_off(int p) { return p == 0 ? 0 : p == 1 ? 4096 : p == 2 ? 12288 : 20480; }   // bytes within a (nb, kc) group
constexpr int DN_GROUP_BYTES = 9 * 4 * DN_NT * 16;      // 36 KB per (n-block, k-chunk): 9 taps

struct DnRegs { u32x4_t v[DN_IN_ITERS]; };
__host__ __device__ constexpr int dn_dma_iters(int p) { return (dn_ntaps(p) * 4 * DN_NT + DN_THREADS - 1) / DN_THREADS; }   // LDS-DMA issues per wave for a phase's slab: 1, 1, 1, 2
template <int N> __device__ __forceinline__ void dn_wait_vm(bool counted) {
    // counted: at most N newer VMEM operations outstanding => everything issued before them is done (VMEM completes in order)
    if (counted) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(DN_THREADS) void conv_down_kernel(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[DN_LDS];
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int c8_fixed = tid & 3;

    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int nkc = a.nkc;                                // 32-channel chunks of Cin
    PersistCursor cursor(a.tiles_x, a.tiles_y, a.nimg, a.nblocks, 4 * nkc);      // stage index within an item = kc*4 + phase
    const int my_items = cursor.my_items;
    const int S = cursor.S;
    if (S == 0) return;
    using StageInfo = PersistStage;
    StageInfo sq0 = cursor.cur, sq1 = cursor.next(), sq2 = cursor.next();

    const int Cin = a.cin0;
    const int cin_shift = 31 - __builtin_clz(Cin);

    // per-lane LDS offsets of the pixel fragments of output rows 2*wave + m at tile offsets (dy, dx) in {-1, 0}^2:
    // p = (2*wave + m + 1 + dy)*IW + r + 1 + dx ; chunk index p*4 + (c8 ^ ((p>>2)&3)), c8 = 2*cp + h
    int a_off[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int dy = (o >> 1) - 1, dx = (o & 1) - 1;
            const int p = (wave * 2 + m + 1 + dy) * DN_IW + r + 1 + dx;
            a_off[m][o] = (p * 4 + (h ^ ((p >> 2) & 3))) * 16;
        }
    const int b_off = (h * DN_NT + r) * 16;

    // input of one stage: the (a, b) phase image of a 32-channel chunk, halo tile rows/cols -1 .. TH-1 / TW-1.
    // A thread's chunks sit at the same tile positions in every stage of an item: their byte offsets (phase (0, 0)) and, per
    // phase, whether they lie inside the image are computed once per ITEM (item_offsets, when the prefetch cursor sq2 enters a
    // new item); a stage's request is then offset + the phase's scalar displacement, or out of range (reads as zero: the padding).
    unsigned coff[DN_IN_ITERS], cokb = 0;         // cokb bit ph * DN_IN_ITERS + i
    auto item_offsets = [&](const PersistItem& it) {
        const int y0 = 2 * (it.ty * DN_TH - 1), x0 = 2 * (it.tx * DN_TW - 1);               // full-res coordinates of tile pixel (0, 0), phase (0, 0)
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        cokb = 0;
#pragma unroll
        for (int i = 0; i < DN_IN_ITERS; ++i) {
            const int p = (t2 + i * DN_THREADS) >> 2;
            const int py = p / DN_IW, px = p - py * DN_IW;
            const int iy = y0 + 2 * py, ix = x0 + 2 * px;
            coff[i] = ((unsigned)((iy + a.in_row_off) * a.Win + ix) << (cin_shift + 1)) + (unsigned)(c8_fixed * 16);
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const bool ok = (unsigned)(iy + (ph >> 1) - a.iy_lo) < (unsigned)a.iy_span && (unsigned)(ix + (ph & 1)) < (unsigned)a.Win;   // slots past the tile: harmless extra rows of the padded buffer
                cokb |= ok ? (1u << (ph * DN_IN_ITERS + i)) : 0u;
            }
        }
    };
    auto load_stage = [&](const StageInfo& si, DnRegs& R) {          // si's item = the item item_offsets last saw
        const PersistItem& it = si.it;
        const int kc = si.kc >> 2, ph = si.kc & 3;
        const char* base = reinterpret_cast<const char*>(a.in0) + (size_t)it.img * a.in_rows * a.Win * Cin * 2 + kc * 64;
        // the requests are inline asm (hipcc then neither tracks nor waits for them: every wait in the stage loop is a counted
        // s_waitcnt placed by hand, see `stage`); the resource descriptor by hand for the same reason
        const unsigned long long ba = (unsigned long long)base;
        u32x4_t irsrc = {(unsigned)ba, (unsigned)(ba >> 32) & 0xffffu, (unsigned)(a.in_rows * a.Win * Cin * 2 - kc * 64), 0x00020000u};
        irsrc.x = __builtin_amdgcn_readfirstlane(irsrc.x); irsrc.y = __builtin_amdgcn_readfirstlane(irsrc.y); irsrc.z = __builtin_amdgcn_readfirstlane(irsrc.z);
        const unsigned delta = (unsigned)((ph >> 1) * a.Win + (ph & 1)) << (cin_shift + 1);        // phase (a, b): a rows down, b pixels right
        const unsigned okp = cokb >> (ph * DN_IN_ITERS);
#pragma unroll
        for (int i = 0; i < DN_IN_ITERS; ++i) {
            const unsigned off = ((okp >> i) & 1u) ? coff[i] + delta : 0xffffffffu;
#if DN_LD_NT
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen nt" : "+v"(R.v[i]) : "v"(off), "s"(irsrc) : "memory");
#else
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(R.v[i]) : "v"(off), "s"(irsrc) : "memory");      // out of range reads as zero: the padding
#endif
        }
    };
    auto store_chunk = [&](int i, const DnRegs& R, uint4* lds_in) {
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int idx = t2 + i * DN_THREADS;
        const int p = idx >> 2;
        reinterpret_cast<u32x4_t*>(lds_in)[p * 4 + (c8_fixed ^ ((p >> 2) & 3))] = R.v[i];
    };
    auto wslab = [&](const StageInfo& si) -> const unsigned char* {
        const int ph = si.kc & 3;
        const int off = ph == 0 ? 0 : ph == 1 ? 4096 : ph == 2 ? 12288 : 20480;
        return reinterpret_cast<const unsigned char*>(a.w) + ((size_t)si.it.nb * nkc + (si.kc >> 2)) * DN_GROUP_BYTES + off;
    };

    f32x16_t acc[2][DN_NTL];
    float* red_base = reinterpret_cast<float*>(smem + DN_MAIN);       // 2 x [8 waves][8 chunks][4]
    const float* bias_lds = reinterpret_cast<const float*>(smem + DN_MAIN + 2 * DN_RED_HALF);
    int st_img = -1, st_tile = 0, st_nb = 0, st_par = 0, red_par = 0;
    auto flush_stats = [&]() {            // partials of the previous item: complete after the stage barrier
        if (st_img < 0) return;
        const float* red = red_base + st_par * (DN_RED_HALF / 4);
        if (__builtin_amdgcn_readfirstlane(wave) != 0) { st_img = -1; return; }
        const int Gs = a.group_size, ngl = DN_NT / Gs, cpg = Gs >> 3;   // couts per GroupNorm group (>= 8 here), groups in this n-block, chunks per group
        if (tid < ngl) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < cpg) {
                        const float* d = red + (w * 8 + tid * cpg + k) * 4;
                        s += d[0] + d[2]; q += d[1] + d[3];
                    }
            const int gg = (st_nb * DN_NT) / Gs + tid;
            float* st = a.stats + (((size_t)st_img * tiles_per_img + st_tile) * 8 + gg) * 2;
            st[0] = s; st[1] = q;
        }
        st_img = -1;
    };
    auto init_acc = [&](int nb) {
        const float* bl = bias_lds + nb * DN_NT + 8 * h;
#pragma unroll
        for (int j = 0; j < DN_NTL; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *reinterpret_cast<const float4*>(bl + j * 32 + 16 * (q >> 1) + 4 * (q & 1));
#pragma unroll
                for (int m = 0; m < 2; ++m) { acc[m][j][4 * q + 0] = bv.x; acc[m][j][4 * q + 1] = bv.y; acc[m][j][4 * q + 2] = bv.z; acc[m][j][4 * q + 3] = bv.w; }
            }
    };
    // accumulator i of lane (r, h), n-tile j is cout nb*64 + j*32 + 16*(i>>3) + 8h + (i&7) (permuted slab rows)
    auto epilogue = [&](const PersistItem& it) __attribute__((always_inline)) {
        int r_e = r, h_e = h, w_e = wave;
        asm volatile("" : "+v"(r_e), "+v"(h_e), "+v"(w_e));
        char* obase = reinterpret_cast<char*>(a.out) + (size_t)it.img * a.Hout * a.Wout * a.cout * 2;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, a.Hout * a.Wout * a.cout * 2, 0x00020000);
        const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
        float* redw = red_base + red_par * (DN_RED_HALF / 4);
        const int ox = it.tx * DN_TW + r_e;
        bool inb[2];
        unsigned offs[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int oy = it.ty * DN_TH + w_e * 2 + m;
            inb[m] = oy < a.Hout && ox < a.Wout;
            offs[m] = ((unsigned)((oy * a.Wout + ox) * a.cout + it.nb * DN_NT) << 1) + (unsigned)(h_e * 16);
        }
#pragma unroll
        for (int j = 0; j < DN_NTL; ++j)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                float sA = 0.f, qA = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const f32x16_t& c = acc[m][j];
                    const unsigned w[4] = {dn_pack(c[8 * pp + 0], c[8 * pp + 1]), dn_pack(c[8 * pp + 2], c[8 * pp + 3]),
                                           dn_pack(c[8 * pp + 4], c[8 * pp + 5]), dn_pack(c[8 * pp + 6], c[8 * pp + 7])};
                    float ts = 0.f, tq = 0.f;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const bf16x2_t wv = __builtin_bit_cast(bf16x2_t, w[d]);
                        ts = __builtin_amdgcn_fdot2_f32_bf16(wv, ones, ts, false);
                        tq = __builtin_amdgcn_fdot2_f32_bf16(wv, wv, tq, false);
                    }
                    sA += inb[m] ? ts : 0.f; qA += inb[m] ? tq : 0.f;
                    const u32x4_t wv4 = {w[0], w[1], w[2], w[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(wv4, orsrc, inb[m] ? offs[m] + (unsigned)(j * 64 + pp * 32) : 0xffffffffu, 0, IRE_ST_PART);
                }
                // sum over the 32 lanes of each half (one 16-B chunk each): cout >= 64 => a chunk never splits into two groups
                sA = dn_ror_add<1>(sA); qA = dn_ror_add<1>(qA);
                sA = dn_ror_add<2>(sA); qA = dn_ror_add<2>(qA);
                sA = dn_ror_add<4>(sA); qA = dn_ror_add<4>(qA);
                sA = dn_ror_add<8>(sA); qA = dn_ror_add<8>(qA);
                sA = dn_swap16_add(sA); qA = dn_swap16_add(qA);
                if ((lane & 31) == 0) {
                    float* d = redw + (wave * 8 + j * 4 + 2 * pp + h_e) * 4;
                    d[0] = sA; d[1] = qA; d[2] = 0.f; d[3] = 0.f;
                }
            }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < DN_NTL; ++j) asm volatile("" : "=v"(acc[m][j]));
        st_img = it.img; st_tile = it.tile; st_nb = it.nb; st_par = red_par; red_par ^= 1;
    };

    DnRegs R0, R1;
    int widx = 0;               // weight slab of the current stage (stage index % 3)
    bool warm = false;          // false in the first stage after the prologue: its predecessors are the prologue's requests
    // ---- one stage: PHASE = a*2 + b of the stage's input (= stage index & 3; the input tile is PHASE & 1) ----------------
    // Pipeline (s = this stage).  A stage is short -- 8 to 32 MFMAs per wave, ~0.3-1 us -- so everything it consumes is
    // requested TWO stages ahead: the input rows of s+2 into the free register set, the weight slab of s+2 by LDS-DMA into
    // the third slab.  (With the slab requested one stage ahead and a vmcnt(0) per stage the kernel ran at one memory
    // latency per stage: 157 / 131 us at levels 1 / 2 for 41 us of MFMA.)  All requests are inline asm, so every wait is a
    // counted one placed here; VMEM order per stage and wave: 5 input requests, then D(phase + 2) slab pieces.
    auto stage = [&](auto phase_tag, auto last_tag) {
        constexpr int PHASE = decltype(phase_tag)::value, PAR = PHASE & 1;
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr int PA = PHASE >> 1, PB = PHASE & 1, NTY = PA ? 2 : 1, NTX = PB ? 2 : 1, NTAPS = NTY * NTX;
        constexpr int D1 = dn_dma_iters((PHASE + 1) & 3), D2 = dn_dma_iters((PHASE + 2) & 3);   // slab pieces requested in stage s-1, s
        const unsigned char* ib = smem + PAR * DN_IN_BYTES;
        uint4* in_nxt = reinterpret_cast<uint4*>(smem + (PAR ^ 1) * DN_IN_BYTES);
        const unsigned char* wb = smem + DN_W_BASE + widx * DN_W_BYTES_MAX + b_off;
        const int w2 = widx == 0 ? 2 : widx - 1;         // (s + 2) % 3
        DnRegs& Rn = PAR ? R0 : R1;   // holds stage s+1 (requested at the start of stage s-1)
        DnRegs& Rf = PAR ? R1 : R0;   // free: receives stage s+2
        load_stage(sq2, Rf);
        {   // weight slab of stage s+2 (phase (PHASE+2)&3: 4, 8, 8 or 16 KB) by LDS-DMA; branch-free: a wave past the slab's end
            // re-fetches a 64-piece group another wave also fetches (same bytes, same destination)
            constexpr int NPH = (PHASE + 2) & 3;
            constexpr int pieces = dn_ntaps(NPH) * 4 * DN_NT;           // 16-B pieces: 256, 512, 512, 1024
            const unsigned char* ws = wslab(sq2);
            const int wave_u = __builtin_amdgcn_readfirstlane(wave);
            const unsigned w_dst_lds = smem_lds + (unsigned)(DN_W_BASE + w2 * DN_W_BYTES_MAX);
#pragma unroll
            for (int i = 0; i < D2; ++i) {
                int cbase = i * DN_THREADS + wave_u * 64;
                if ((i + 1) * DN_THREADS > pieces) cbase = cbase % pieces;
                dn_glds16(ws + (size_t)(cbase + lane) * 16, w_dst_lds + cbase * 16);
            }
        }
        // the rows of stage s+1: requested at the start of stage s-1; since then: that stage's slab pieces, this stage's requests
        dn_wait_vm<DN_IN_ITERS + D1 + D2>(warm);
#pragma unroll
        for (int i = 0; i < DN_IN_ITERS; ++i) asm volatile("" : "+v"(Rn.v[i]));
        // NTAPS taps x 2 channel pairs; tap t = ty*NTX + tx reads tile offset (dy, dx): dy = PA ? ty - 1 : 0, dx = PB ? tx - 1 : 0
        constexpr int NG = NTAPS * 2;
        bf16x8_t afr[2][2], bfr[2][DN_NTL];
        auto read_group = [&](int g, bf16x8_t (&af)[2], bf16x8_t (&bf)[DN_NTL]) __attribute__((always_inline)) {
            const int t = g >> 1, cp = g & 1;
            const int ty = t / NTX, tx = t - ty * NTX;
            const int o = ((PA ? ty : 1) << 1) | (PB ? tx : 1);        // index into a_off: (dy + 1)*2 + (dx + 1)
#pragma unroll
            for (int m = 0; m < 2; ++m) af[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ib + (a_off[m][o] ^ (cp << 5))));
#pragma unroll
            for (int j = 0; j < DN_NTL; ++j) bf[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wb + ((t * 4 + 2 * cp) * DN_NT + j * 32) * 16));
        };
        read_group(0, afr[0], bfr[0]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) read_group(g + 1, afr[(g + 1) & 1], bfr[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < DN_NTL; ++j) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[g & 1][j], afr[g & 1][m], acc[m][j], 0, 0, 0);   // D[cout][pixel]
            // stage s+1's input: registers -> the other LDS tile, spread over the groups (all 5 chunks by the last group)
            constexpr int per = (DN_IN_ITERS + NG - 1) / NG;
#pragma unroll
            for (int i = g * per; i < (g + 1) * per && i < DN_IN_ITERS; ++i) store_chunk(i, Rn, in_nxt);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the slab of stage s+1 (requested during stage s-1) is in LDS before the barrier; newer: this stage's requests
        dn_wait_vm<DN_IN_ITERS + D2>(warm);
        warm = true;
        if constexpr (LAST) epilogue(sq0.it);
        __syncthreads();
        flush_stats();
        sq0 = sq1; sq1 = sq2; sq2 = cursor.next();
        if (sq2.kc == 0) item_offsets(sq2.it);       // (past the queue's end the cursor stays on the last stage: kc != 0)
        widx = widx == 2 ? 0 : widx + 1;
    };

    // ---- prologue: stage 0 (phase 0) -> LDS buffer 0, stage 1 -> registers --------------------------------------------------
    {
        float* bl = reinterpret_cast<float*>(smem + DN_MAIN + 2 * DN_RED_HALF);
        if (tid < a.cout && tid < 256) bl[tid] = a.bias[tid];
        item_offsets(sq0.it);        // stages 0, 1, 2 belong to one item (an item has 4 * nkc >= 4 stages)
        load_stage(sq0, R0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < DN_IN_ITERS; ++i) asm volatile("" : "+v"(R0.v[i]));
        const uint4* ws0 = reinterpret_cast<const uint4*>(wslab(sq0));
        const uint4* ws1 = reinterpret_cast<const uint4*>(wslab(sq1));
        uint4* wd = reinterpret_cast<uint4*>(smem + DN_W_BASE);
        for (int i = tid; i < dn_ntaps(0) * 4 * DN_NT; i += DN_THREADS) wd[i] = ws0[i];                            // slab of stage 0 -> slot 0
        for (int i = tid; i < dn_ntaps(1) * 4 * DN_NT; i += DN_THREADS) wd[DN_W_BYTES_MAX / 16 + i] = ws1[i];      // stage 1 -> slot 1
        uint4* in0 = reinterpret_cast<uint4*>(smem);
#pragma unroll
        for (int i = 0; i < DN_IN_ITERS; ++i) store_chunk(i, R0, in0);
        load_stage(sq1, R1);
    }
    __syncthreads();
    for (int k = 0; k < my_items; ++k) {
        init_acc(sq0.it.nb);
        for (int kc = 0; kc + 1 < nkc; ++kc) {
            stage(std::integral_constant<int, 0>{}, std::false_type{}); stage(std::integral_constant<int, 1>{}, std::false_type{});
            stage(std::integral_constant<int, 2>{}, std::false_type{}); stage(std::integral_constant<int, 3>{}, std::false_type{});
        }
        stage(std::integral_constant<int, 0>{}, std::false_type{}); stage(std::integral_constant<int, 1>{}, std::false_type{});
        stage(std::integral_constant<int, 2>{}, std::false_type{}); stage(std::integral_constant<int, 3>{}, std::true_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may be in flight when the workgroup's LDS is released
    __syncthreads();
    flush_stats();
}

}  // namespace

// a.in0 = [nimg][Hin (+halo)][Win][Cin] bf16, a.out = [nimg][Hin/2][Win/2][cout]; a.nkc = Cin/32, a.nblocks = cout/64, tiles of
// 16x32 OUTPUT pixels; a.w = phase slabs [nblock][kc][phase: 1+2+2+4 taps][tap*4 + c8][64 permuted rows][8]; a.stats required
void conv_down_launch(const ConvArgs& a, hipStream_t stream) {
    if (a.cout % 64 || a.cout > 256 || a.nkc < 1 || a.stats == nullptr) fail(IRE_ERR_INTERNAL, "internal: conv_down shape");
    const int items = a.tiles_x * a.tiles_y * a.nimg * a.nblocks;
    const int cus = persistent_grid_cus();
    const int grid = items < cus ? items : cus;
    hipLaunchKernelGGL(conv_down_kernel, dim3(grid), dim3(DN_THREADS), 0, stream, a);
    IRE_HIP(hipGetLastError());
}

}  // namespace ire
