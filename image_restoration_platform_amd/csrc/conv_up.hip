// conv_up.hip -- the decoder's "nearest x2 -> 3x3 conv" (RestoreNet-v0 `up` layers, 15 % of the network's FLOPs) as a
// SUB-PIXEL convolution on the low-resolution grid, gfx950.
//
// A 3x3 convolution of a nearest-x2 upsampled tensor reads, for the output pixel (2Y + a, 2X + b), only a 2x2 window of
// low-resolution pixels: rows Y-1+a .. Y+a, columns X-1+b .. X+b.  The taps that fall on the same low-resolution pixel can be
// summed ahead of time:   a = 0: row Y-1 <- ky 0,  row Y <- ky 1 + ky 2;     a = 1: row Y <- ky 0 + ky 1,  row Y+1 <- ky 2
// (same for the columns), so each of the four output parities (a, b) is a 2x2 convolution with its own pre-summed weights
// (engine.cpp::make_conv builds them; zero padding at the image border carries over exactly because a summed pair never
// straddles the border).  K per output drops from 9*Cin to 4*Cin: 2.25x fewer MFMAs for the same result, the upsampled
// tensor never exists, and one staged low-resolution tile serves 4x as many output pixels.
//
// Schedule: the persistent, software-pipelined structure of conv_rb.hip (one 512-thread workgroup per CU, 32-channel K
// stages double-buffered in LDS, input register-prefetched two stages ahead, weights by LDS-DMA, one barrier per stage).
// Work item = (16x32 LOW-resolution tile, 32-cout block) -> a 32x64 block of output pixels, all four parities: wave w owns
// low-res rows 2w, 2w+1 and keeps 4 parities x 2 rows x 32 couts = 128 accumulators.  Per stage a tap's two pixel fragments
// are read once and feed every parity whose window contains the tap: 36 + 32 fragment reads for 64 MFMAs.
// Roofline: MFMA (2*9*Cin*Cout algorithmic flop per OUTPUT pixel; executed: 2*4*Cin*Cout).
//
// FUSED form (NKS > 0): RestoreNet-v0 follows every `up` with the 1x1 `fuse` over concat(up, skip) and nothing non-linear
// sits between them, so   fuse(concat(up(x), skip)) = (Wf_up . Wup) * x_up  +  Wf_skip . skip  +  (Wf_up . b_up + b_f):
// engine.cpp::make_up_fused composes the two weight tensors once per load (fp64, then the sub-pixel pre-sums, then bf16) and
// this kernel adds the skip term in its epilogue: the accumulator tile of a (parity, row) is D[cout][pixel], the skip pixels
// come straight from HBM in the MFMA B layout (lane = pixel, 8 channels = 16 B) and the 32 x C skip weights of the item's
// cout block wait in LDS (LDS-DMA one stage ahead).  The `up` tensor is never written or read (2 x 2 B x C per output pixel
// less traffic, one launch less per level), `fuse`'s 2*C*C flop per pixel over the up half are gone, and the epilogue also
// writes the GroupNorm partials `fuse` used to write: one (sum, sumsq) per group per ITEM (32 x 64 output pixels).
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

#ifndef IRE_UP_TEPI
#define IRE_UP_TEPI 1     // fused form: line-coalesced epilogue (skip rows and output rows through a wave-private LDS patch); 0: the direct epilogue (build-time A/B)
#endif
#ifndef IRE_UP_ABL
#define IRE_UP_ABL 0      // timing ablations of the fused epilogue (results wrong by design): 1 no skip term, 2 no statistics, 4 no stores, 8 L2 touches in the last stage
#endif
constexpr int UP_THREADS = 512;
constexpr int UP_TH = 16, UP_TW = 32;                 // low-resolution tile
constexpr int UP_IH = UP_TH + 2, UP_IW = UP_TW + 2;
constexpr int UP_IN_CHUNKS = UP_IH * UP_IW * 4;        // 2448 x 16 B (32 channels per pixel)
constexpr int UP_IN_ITERS = (UP_IN_CHUNKS + UP_THREADS - 1) / UP_THREADS;   // 5
constexpr int UP_IN_BYTES = UP_IN_ITERS * UP_THREADS * 16;                  // 40960: all 5 x 512 chunk slots exist
constexpr int UP_NT = 32;                             // couts per item
constexpr int UP_W_CHUNKS = 4 * 16 * UP_NT;           // [parity][kk = tap4*4 + c8][32 rows] x 16 B = 32 KB per stage
constexpr int UP_W_BYTES = UP_W_CHUNKS * 16;
constexpr int UP_W_ITERS = UP_W_CHUNKS / UP_THREADS;   // 4
constexpr int UP_BUF = UP_IN_BYTES + UP_W_BYTES;
constexpr int UP_BIAS_OFF = 2 * UP_BUF;                // 256 floats
constexpr int UP_SKW_OFF = UP_BIAS_OFF + 256 * 4;      // fused form: skip weights of the item's cout block [ks][h][32 rows][8] bf16, <= 8 KB
constexpr int UP_RED_OFF = UP_SKW_OFF + 8192;          // fused form: [8 waves][8 cout quads][2] floats
constexpr int UP_LDS = UP_RED_OFF + 8 * 8 * 2 * 4;
static_assert(UP_LDS <= 160 * 1024, "LDS");
constexpr int UP_PATCH = UP_BUF / 8;                   // line-coalesced epilogue: 9 KB per wave of the buffer the item's last stage has finished with

__device__ __forceinline__ unsigned up_pack(float a, float b) {
    f32x2_t f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
// LDS-DMA (see conv_rb.hip::rb_glds16): invisible to the compiler's s_waitcnt bookkeeping, the caller waits and barriers
__device__ __forceinline__ void up_glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}

template <int N> __device__ __forceinline__ float up_ror_add(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float up_swap16_add(float v) {
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}

template <int I, int N, class F> __device__ __forceinline__ void up_static_for(F& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); up_static_for<I + 1, N>(f); }
}

using UpItem = PersistItem;
struct UpRegs { uint4 v[UP_IN_ITERS]; };

// NKS = 0: the plain `up` convolution.  NKS = C/16 in {2, 4, 8}: up + fuse composed, the skip term (K = C = 16 NKS) in the epilogue.
template <int NKS>
__global__ __launch_bounds__(UP_THREADS) void conv_up_kernel(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[UP_LDS];
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int c8_fixed = tid & 3;

    // ---- persistent work assignment (persist.hpp) ---------------------------------------------------------------
    PersistCursor cursor(a.tiles_x, a.tiles_y, a.nimg, a.nblocks, a.nkc);
    const int my_items = cursor.my_items;
    const int nkc = a.nkc;                               // 32-channel stages per item (>= 2, even)
    const int S = cursor.S;
    if (S == 0) return;
    using StageInfo = PersistStage;
    StageInfo sq0 = cursor.cur, sq1 = cursor.next(), sq2 = cursor.next();

    const int Cin = a.cin0;
    const int cin_shift = 31 - __builtin_clz(Cin);

    // per-lane LDS offsets of the 18 (row m, tap) pixel fragments: p = (2*wave + m + ky)*IW + r + kx,
    // 16-B chunk index p*4 + (c8 ^ ((p>>2)&3)), c8 = 2*(k-step & 1) + h  (the k-step parity toggles bit 5)
    int a_off[2][9];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int p = (wave * 2 + m + ky) * UP_IW + r + kx;
            a_off[m][tap] = (p * 4 + (h ^ ((p >> 2) & 3))) * 16;
        }
    const int b_off = (h * UP_NT + r) * 16;               // + ((parity*16 + tap4*4 + 2*c8pair) * 32) * 16

    auto load_chunk = [&](const StageInfo& si, int i, UpRegs& R) {
        const UpItem& it = si.it;
        const int oy1 = it.ty * UP_TH - 1, ox1 = it.tx * UP_TW - 1;
        const char* base = reinterpret_cast<const char*>(a.in0) + (size_t)it.img * a.in_rows * a.Win * Cin * 2 + si.kc * 64;
        const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, a.in_rows * a.Win * Cin * 2 - si.kc * 64, 0x00020000);
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int p = (t2 + i * UP_THREADS) >> 2;
        const int py = p / UP_IW, px = p - py * UP_IW;
        const int iy = oy1 + py, ix = ox1 + px;
        const bool ok = (unsigned)(iy - a.iy_lo) < (unsigned)a.iy_span && (unsigned)ix < (unsigned)a.Win;
        const unsigned off = ok ? ((unsigned)((iy + a.in_row_off) * a.Win + ix) << (cin_shift + 1)) + (unsigned)(c8_fixed * 16) : 0xffffffffu;
        const u32x4_t lv = __builtin_amdgcn_raw_buffer_load_b128(irsrc, off, 0, 0);      // out of range reads as zero: the padding
        R.v[i] = make_uint4(lv.x, lv.y, lv.z, lv.w);
    };
    auto load_stage = [&](const StageInfo& si, UpRegs& R) {
#pragma unroll
        for (int i = 0; i < UP_IN_ITERS; ++i) load_chunk(si, i, R);
    };
    auto store_chunk = [&](int i, const UpRegs& R, uint4* lds_in) {
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int idx = t2 + i * UP_THREADS;
        const int p = idx >> 2;
        lds_in[p * 4 + (c8_fixed ^ ((p >> 2) & 3))] = R.v[i];       // a pixel outside the image was an out-of-range load: already zero
    };
    auto wslab = [&](const StageInfo& si) -> const unsigned char* {
        return reinterpret_cast<const unsigned char*>(a.w) + ((size_t)si.it.nb * nkc + si.kc) * UP_W_BYTES;
    };

    f32x16_t acc[4][2];
    const float* bias_lds = reinterpret_cast<const float*>(smem + UP_BIAS_OFF);
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    UpRegs R;          // one register set: holds stage s+1's input; chunk i goes to LDS and is reloaded with stage s+2 in place

    // ---- epilogue: accumulator i of lane (r, h) of parity (pa, pb), row m is output pixel (2*(ty*16 + 2*wave + m) + pa,
    // 2*(tx*32 + r) + pb), cout nb*32 + 16*(i>>3) + 8h + (i&7) (permuted slab rows): two 16-B stores per (parity, m) ------------
    auto epilogue = [&](const UpItem& it) __attribute__((always_inline)) {
        int r_e = r, h_e = h, w_e = wave;
        asm volatile("" : "+v"(r_e), "+v"(h_e), "+v"(w_e));
        char* obase = reinterpret_cast<char*>(a.out) + (size_t)it.img * a.Hout * a.Wout * a.cout * 2;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, a.Hout * a.Wout * a.cout * 2, 0x00020000);
        const int ly = it.ty * UP_TH + w_e * 2, lx = it.tx * UP_TW + r_e;       // low-res coordinates of row m = 0
        if constexpr (NKS > 0) {
            // ---- fused form: skip term by MFMA, bf16 stores, GroupNorm partials ------------------------------------------------
            constexpr int NKG = NKS / 2;              // units (two k-steps = 32 skip channels) per (parity, row)
            constexpr int U = 8 * NKG;                // unit u = (par * 2 + m) * NKG + kg
            // units in flight (2 x 16 B per lane each): as many as fit WITHOUT a spill -- a scratch reload in this epilogue queues
            // behind the skip loads in flight (VMEM completes in order) and drains the ring.  Same-box A/B (IRE_UP_D = 4 / 3 / 2 for
            // all): NKS 2: 337 / 362 / 342 us, NKS 4 (2 spills at 4): 259 / 253 / 270, NKS 8 (13 spills at 4, 5 at 3): 236 / 233 / 226
#ifdef IRE_UP_D
            constexpr int D = IRE_UP_D;
#else
            constexpr int D = NKS <= 2 ? 4 : NKS == 4 ? 3 : 2;
#endif
            const int C = a.cout;
            const char* sbase = reinterpret_cast<const char*>(a.in1) + (size_t)it.img * a.Hout * a.Wout * C * 2;
            // loads past the image's last byte return zero; a pixel past the right edge reads a neighbour's bytes: either way the
            // MFMA column (= that pixel) is never stored nor counted
            const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sbase), 0, a.Hout * a.Wout * C * 2, 0x00020000);
            const unsigned rowB = (unsigned)(a.Wout * C * 2), pxB = (unsigned)(C * 2);
            const unsigned s_off = (unsigned)((2 * ly) * a.Wout + 2 * lx) * pxB + (unsigned)(h_e * 16);
            const bool inb[2] = {ly < a.Hin && lx < a.Win, ly + 1 < a.Hin && lx < a.Win};
            const unsigned char* skw = smem + UP_SKW_OFF + (h_e * 32 + r_e) * 16;      // + ks * 1024
            const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
            u32x4_t ring[D][2];
            float ssum[4] = {0.f, 0.f, 0.f, 0.f}, qsum[4] = {0.f, 0.f, 0.f, 0.f};   // per cout quad of this lane-half: i = 4k .. 4k + 3
            auto issue = [&](auto u_tag) __attribute__((always_inline)) {
                constexpr int u = decltype(u_tag)::value;
                constexpr int pm = u / NKG, kg = u % NKG, par = pm >> 1, m = pm & 1, pa = par >> 1, pb = par & 1;
                const unsigned off = s_off + (unsigned)(2 * m + pa) * rowB + (unsigned)pb * pxB + (unsigned)(kg * 64);
                ring[u % D][0] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, off, 0, 0);
                ring[u % D][1] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, off + 32u, 0, 0);
            };
            auto consume = [&](auto u_tag) __attribute__((always_inline)) {
                constexpr int u = decltype(u_tag)::value;
                constexpr int pm = u / NKG, kg = u % NKG, par = pm >> 1, m = pm & 1;
                const bf16x8_t w0 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(skw + (2 * kg) * 1024));
                const bf16x8_t w1 = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(skw + (2 * kg + 1) * 1024));
                acc[par][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, __builtin_bit_cast(bf16x8_t, ring[u % D][0]), acc[par][m], 0, 0, 0);
                acc[par][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, __builtin_bit_cast(bf16x8_t, ring[u % D][1]), acc[par][m], 0, 0, 0);
            };
            auto finish = [&](auto par_tag) __attribute__((always_inline)) {
                constexpr int par = decltype(par_tag)::value;
                constexpr int pa = par >> 1, pb = par & 1;
                const int ox = 2 * lx + pb;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int oy = 2 * (ly + m) + pa;
                    const unsigned off = ((unsigned)((oy * a.Wout + ox) * a.cout + it.nb * UP_NT) << 1) + (unsigned)(h_e * 16);
                    const f32x16_t& c = acc[par][m];
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        unsigned w[4] = {up_pack(c[8 * pp + 0], c[8 * pp + 1]), up_pack(c[8 * pp + 2], c[8 * pp + 3]),
                                         up_pack(c[8 * pp + 4], c[8 * pp + 5]), up_pack(c[8 * pp + 6], c[8 * pp + 7])};
                        const u32x4_t wv4 = {w[0], w[1], w[2], w[3]};
                        if constexpr (!(IRE_UP_ABL & 4)) __builtin_amdgcn_raw_buffer_store_b128(wv4, orsrc, inb[m] ? off + (unsigned)(pp * 32) : 0xffffffffu, 0, IRE_ST_PART);
                        else asm volatile("" :: "v"(wv4));
                        if constexpr (!(IRE_UP_ABL & 2))
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const bf16x2_t wv = __builtin_bit_cast(bf16x2_t, inb[m] ? w[d] : 0u);     // the statistics are those of the STORED values
                            ssum[2 * pp + (d >> 1)] = __builtin_amdgcn_fdot2_f32_bf16(wv, ones, ssum[2 * pp + (d >> 1)], false);
                            qsum[2 * pp + (d >> 1)] = __builtin_amdgcn_fdot2_f32_bf16(wv, wv, qsum[2 * pp + (d >> 1)], false);
                        }
                    }
                }
            };
            auto unit = [&](auto u_tag) __attribute__((always_inline)) {
                constexpr int u = decltype(u_tag)::value;
                if constexpr (!(IRE_UP_ABL & 1)) {
                consume(u_tag);
                if constexpr (u + D < U) issue(std::integral_constant<int, u + D>{});
                }
                if constexpr ((u + 1) % (2 * NKG) == 0) finish(std::integral_constant<int, u / (2 * NKG)>{});
            };
            if constexpr (!(IRE_UP_ABL & 1)) up_static_for<0, D>(issue);
            up_static_for<0, U>(unit);
            if constexpr (!(IRE_UP_ABL & 2)) {
            // GroupNorm partials of the item: quad k of lane-half h is couts 16 (k >> 1) + 8 h + 4 (k & 1) .. + 3 of the block
            float rv[8] = {ssum[0], qsum[0], ssum[1], qsum[1], ssum[2], qsum[2], ssum[3], qsum[3]};
#pragma unroll
            for (int i = 0; i < 8; ++i) rv[i] = up_ror_add<1>(rv[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) rv[i] = up_ror_add<2>(rv[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) rv[i] = up_ror_add<4>(rv[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) rv[i] = up_ror_add<8>(rv[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) rv[i] = up_swap16_add(rv[i]);
            float* red = reinterpret_cast<float*>(smem + UP_RED_OFF);
            if (r_e == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<float2*>(red + (w_e * 8 + (k >> 1) * 4 + 2 * h_e + (k & 1)) * 2) = make_float2(rv[2 * k], rv[2 * k + 1]);
            }
            __syncthreads();
            const int G = a.group_size, qpg = G >> 2, ngl = UP_NT / G;       // quads per group (1, 2 or 4), groups in the item's 32 couts
            if (tid < ngl) {
                float sv = 0.f, qv = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < qpg) { sv += red[(w * 8 + tid * qpg + k) * 2 + 0]; qv += red[(w * 8 + tid * qpg + k) * 2 + 1]; }
                float* st = a.stats + (((size_t)it.img * tiles_per_img + it.tile) * 8 + (it.nb * UP_NT) / G + tid) * 2;
                st[0] = sv; st[1] = qv;
            }
            }
        } else {
#pragma unroll
        for (int par = 0; par < 4; ++par) {
            const int pa = par >> 1, pb = par & 1;
            const int ox = 2 * lx + pb;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int oy = 2 * (ly + m) + pa;
                const bool inb = oy < a.Hout && ox < a.Wout;
                const unsigned off = ((unsigned)((oy * a.Wout + ox) * a.cout + it.nb * UP_NT) << 1) + (unsigned)(h_e * 16);
                const f32x16_t& c = acc[par][m];
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const u32x4_t wv4 = {up_pack(c[8 * pp + 0], c[8 * pp + 1]), up_pack(c[8 * pp + 2], c[8 * pp + 3]),
                                         up_pack(c[8 * pp + 4], c[8 * pp + 5]), up_pack(c[8 * pp + 6], c[8 * pp + 7])};
                    __builtin_amdgcn_raw_buffer_store_b128(wv4, orsrc, inb ? off + (unsigned)(pp * 32) : 0xffffffffu, 0, IRE_ST_PART);
                }
            }
        }
        }
#pragma unroll
        for (int par = 0; par < 4; ++par)
#pragma unroll
            for (int m = 0; m < 2; ++m) asm volatile("" : "=v"(acc[par][m]));     // dead until the next item
    };

    // ---- fused form, line-coalesced epilogue (IRE_UP_TEPI).  The direct epilogue below fetches the skip pixels in the MFMA B layout
    // and stores from the accumulator layout: lane = pixel, so the four lanes of a quad address four different pixels and the
    // texture addresser -- which coalesces within a quad -- spends ~100 cycles on such a load and ~66 on such a store instead of
    // ~16-20 (profiles/r03_experiments.md): 21 000 / 34 000 / 60 000 addresser cycles per item at levels 0 / 1 / 2, half of the
    // kernel, all in a phase where nothing else runs.  Here a wave works through its four output rows (m, pa); per row
    //   * the skip row segment (64 pixels x 32 channels per piece) is loaded COALESCED (a quad = 64 contiguous bytes), written to
    //     the wave's LDS patch (chunk index XOR-swizzled by the pixel pair) and read back as B fragments (2-way bank conflicts at
    //     worst: a parity's pixels sit 2 apart) for the skip MFMAs of both column parities; the next piece's loads are in flight meanwhile;
    //   * the two finished accumulator tiles (pb = 0, 1) go through the same patch the other way: written in the accumulator layout,
    //     read back four lanes per pixel, GroupNorm partials per 8-cout chunk, stored 64 contiguous bytes per quad.
    // The patch is the wave's ninth of the LDS buffer the item's last stage has just finished with, so the stage barrier comes BEFORE
    // this epilogue; the barrier that publishes the partials closes it (the same two barriers per item as before).
    // the first skip piece of an item is requested BEFORE the stage barrier that precedes the epilogue (the barrier wait and the
    // patch set-up then cover its latency: otherwise every item pays one exposed round trip)
    u32x4_t skpre[4];
    auto epi_prefetch = [&](const UpItem& it) __attribute__((always_inline)) {
        constexpr int C = 16 * (NKS > 0 ? NKS : 2);
        int l_e = lane, w_e = wave;
        asm volatile("" : "+v"(l_e), "+v"(w_e));
        const char* sbase = reinterpret_cast<const char*>(a.in1) + (size_t)it.img * a.Hout * a.Wout * C * 2;
        const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sbase), 0, a.Hout * a.Wout * C * 2, 0x00020000);
        const int ly = it.ty * UP_TH + w_e * 2, ox0 = 2 * it.tx * UP_TW;
        const int lc = l_e & 3, lp0 = l_e >> 2;
        const bool rowok = ly < a.Hin;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = lp0 + i * 16;
            const bool ok = rowok && ox0 + p < a.Wout;
            skpre[i] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, ok ? (unsigned)((((2 * ly) * a.Wout + ox0 + p) * C + lc * 8) * 2) : 0xffffffffu, 0, C == 32 ? IRE_LD_ONCE : 0);
        }
    };
    auto epilogue_t = [&](const UpItem& it) __attribute__((always_inline)) {
        constexpr int C = 16 * (NKS > 0 ? NKS : 2);      // skip channels = cout (the plain-up instantiation NKS = 0 never calls this)
        constexpr int PCH = 32;                          // skip channels per piece: a quad of lanes = one pixel's 64 contiguous bytes (64-channel pieces: 16-21 spills)
        constexpr int NPIECE = C / PCH;                  // 1, 2, 4 (levels 0, 1, 2)
        constexpr int CPP = PCH / 8;                     // 16-B chunks per pixel in a piece: 4 or 8 = wave-loads per piece
        constexpr int KSP = PCH / 16;                    // skip k-steps per piece: 2 or 4
        constexpr int PSTEP = 64 / CPP;                  // pixels per wave-load: 16 or 8
        int l_e = lane, w_e = wave;
        asm volatile("" : "+v"(l_e), "+v"(w_e));
        const int r_e = l_e & 31, h_e = l_e >> 5;
        unsigned char* patch = smem + UP_BUF + w_e * UP_PATCH;          // an item ends on an odd stage: buffer 1
        char* obase = reinterpret_cast<char*>(a.out) + (size_t)it.img * a.Hout * a.Wout * C * 2;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, a.Hout * a.Wout * C * 2, 0x00020000);
        const char* sbase = reinterpret_cast<const char*>(a.in1) + (size_t)it.img * a.Hout * a.Wout * C * 2;
        const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sbase), 0, a.Hout * a.Wout * C * 2, 0x00020000);
        const int ly = it.ty * UP_TH + w_e * 2, ox0 = 2 * it.tx * UP_TW;
        const unsigned char* skw = smem + UP_SKW_OFF + (h_e * 32 + r_e) * 16;      // + ks * 1024
        const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
        // loader: chunk q = lane + 64 i of the piece = (pixel q / CPP, chunk q % CPP); 64 % CPP == 0, so the chunk is the lane's own
        const int lc = l_e % CPP, lp0 = l_e / CPP;
        const unsigned ld_lds = (unsigned)(lp0 * (CPP * 16) + ((lc ^ ((lp0 >> 1) & (CPP - 1))) << 4));      // + i * PSTEP * CPP * 16 (PSTEP is even: the swizzle term is i-invariant)
        // store read-back: q = lane + 64 k = (pixel (lane >> 2) + 16 k, chunk lane & 3)
        const int sp0 = l_e >> 2, scq = l_e & 3;
        float ssum[2] = {0.f, 0.f}, qsum[2] = {0.f, 0.f};       // the two cout quads of this lane's read-back chunk: couts 8 scq .. +3, +4 .. +7
        u32x4_t sk[CPP];
        auto issue_piece = [&](int rr, int pc) __attribute__((always_inline)) {
            const int m = rr >> 1, pa = rr & 1;
            const int oy = 2 * (ly + m) + pa;
            const bool rowok = ly + m < a.Hin;
#pragma unroll
            for (int i = 0; i < CPP; ++i) {
                const int p = lp0 + i * PSTEP;
                const bool ok = rowok && ox0 + p < a.Wout;
                sk[i] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, ok ? (unsigned)(((oy * a.Wout + ox0 + p) * C + pc * PCH + lc * 8) * 2) : 0xffffffffu, 0, C == 32 ? IRE_LD_ONCE : 0);
            }
        };
        if constexpr (!(IRE_UP_ABL & 1)) {
#pragma unroll
            for (int i = 0; i < CPP; ++i) sk[i] = skpre[i];        // piece (row 0, channels 0..31): requested by epi_prefetch before the stage barrier
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int m = rr >> 1, pa = rr & 1;
            if constexpr (!(IRE_UP_ABL & 1))
#pragma unroll
            for (int pc = 0; pc < NPIECE; ++pc) {
#pragma unroll
                for (int i = 0; i < CPP; ++i) *reinterpret_cast<u32x4_t*>(patch + ld_lds + i * (PSTEP * CPP * 16)) = sk[i];
                if (rr * NPIECE + pc + 1 < 4 * NPIECE) issue_piece((rr * NPIECE + pc + 1) / NPIECE, (rr * NPIECE + pc + 1) % NPIECE);     // in flight across this piece's MFMAs (and the row's stores)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int k = 0; k < KSP; ++k) {
                        const bf16x8_t wk = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(skw + (pc * KSP + k) * 1024));
                        const bf16x8_t fr = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(patch + (2 * r_e + pb) * (CPP * 16) + (((2 * k + h_e) ^ (r_e & (CPP - 1))) << 4)));
                        acc[pa * 2 + pb][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk, fr, acc[pa * 2 + pb][m], 0, 0, 0);
                    }
            }
            // ---- the row's two accumulator tiles -> patch (accumulator layout) -> read back four lanes per pixel -> statistics, stores
            const int oy = 2 * (ly + m) + pa;
            const bool rowok = ly + m < a.Hin;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const f32x16_t& c = acc[pa * 2 + pb][m];
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const u32x4_t wv = {up_pack(c[8 * pp + 0], c[8 * pp + 1]), up_pack(c[8 * pp + 2], c[8 * pp + 3]),
                                        up_pack(c[8 * pp + 4], c[8 * pp + 5]), up_pack(c[8 * pp + 6], c[8 * pp + 7])};
                    *reinterpret_cast<u32x4_t*>(patch + (2 * r_e + pb) * 64 + (((2 * pp + h_e) ^ (r_e & 3)) << 4)) = wv;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = sp0 + 16 * k;
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(patch + p * 64 + ((scq ^ ((p >> 1) & 3)) << 4));
                const bool ok = rowok && ox0 + p < a.Wout;
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
                if constexpr (!(IRE_UP_ABL & 2))
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const bf16x2_t bv = __builtin_bit_cast(bf16x2_t, ok ? w[d] : 0u);      // the statistics are those of the STORED values
                    ssum[d >> 1] = __builtin_amdgcn_fdot2_f32_bf16(bv, ones, ssum[d >> 1], false);
                    qsum[d >> 1] = __builtin_amdgcn_fdot2_f32_bf16(bv, bv, qsum[d >> 1], false);
                }
                if constexpr (!(IRE_UP_ABL & 4)) __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, ok ? (unsigned)(((oy * a.Wout + ox0 + p) * C + it.nb * UP_NT + scq * 8) * 2) : 0xffffffffu, 0, (C == 32 ? IRE_ST_LINE : IRE_ST_PART));
                else asm volatile("" :: "v"(v));
            }
        }
        if constexpr (!(IRE_UP_ABL & 2)) {
            // the 16 lanes that share a read-back chunk sit 4 apart: two rotations within the row of 16, then the rows and the halves
            float rv[4] = {ssum[0], qsum[0], ssum[1], qsum[1]};
#pragma unroll
            for (int i = 0; i < 4; ++i) rv[i] = up_ror_add<4>(rv[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) rv[i] = up_ror_add<8>(rv[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) rv[i] = up_swap16_add(rv[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) { float x = rv[i], y = rv[i]; asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y)); rv[i] = x + y; }
            float* red = reinterpret_cast<float*>(smem + UP_RED_OFF);
            if (l_e < 4) {          // chunk scq = lane: cout quads 2 scq, 2 scq + 1
                *reinterpret_cast<float2*>(red + (w_e * 8 + 2 * l_e) * 2) = make_float2(rv[0], rv[1]);
                *reinterpret_cast<float2*>(red + (w_e * 8 + 2 * l_e + 1) * 2) = make_float2(rv[2], rv[3]);
            }
            __syncthreads();
            const int G = a.group_size, qpg = G >> 2, ngl = UP_NT / G;       // quads per group (1, 2 or 4), groups in the item's 32 couts
            if (tid < ngl) {
                float sv = 0.f, qv = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < qpg) { sv += red[(w * 8 + tid * qpg + k) * 2 + 0]; qv += red[(w * 8 + tid * qpg + k) * 2 + 1]; }
                float* st = a.stats + (((size_t)it.img * tiles_per_img + it.tile) * 8 + (it.nb * UP_NT) / G + tid) * 2;
                st[0] = sv; st[1] = qv;
            }
        } else __syncthreads();
#pragma unroll
        for (int par = 0; par < 4; ++par)
#pragma unroll
            for (int m = 0; m < 2; ++m) asm volatile("" : "=v"(acc[par][m]));     // dead until the next item
    };

    // ---- one pipeline stage ------------------------------------------------------------------------------------------
    auto init_acc = [&](int nb) {            // new item: the accumulators start at the bias (permuted rows: cout 16*(i>>3) + 8h + (i&7))
        const float* bl = bias_lds + nb * UP_NT + 8 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bv = *reinterpret_cast<const float4*>(bl + 16 * (q >> 1) + 4 * (q & 1));
#pragma unroll
            for (int par = 0; par < 4; ++par)
#pragma unroll
                for (int m = 0; m < 2; ++m) { acc[par][m][4 * q + 0] = bv.x; acc[par][m][4 * q + 1] = bv.y; acc[par][m][4 * q + 2] = bv.z; acc[par][m][4 * q + 3] = bv.w; }
        }
    };
    auto stage = [&](int s, auto par_tag, auto last_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr int LASTV = decltype(last_tag)::value;      // 1: the item's last stage (epilogue before the barrier); 2: the one before it
        constexpr bool LAST = LASTV == 1, PRELAST = LASTV == 2;
        const unsigned char* ib = smem + PAR * UP_BUF;
        uint4* in_nxt = reinterpret_cast<uint4*>(smem + (PAR ^ 1) * UP_BUF);
        const unsigned char* wb = smem + PAR * UP_BUF + UP_IN_BYTES + b_off;
        unsigned char* w_nxt = smem + (PAR ^ 1) * UP_BUF + UP_IN_BYTES;
        // R (stage s+1's input) was retired by the vmcnt(0) that ended the previous stage: say so, or hipcc re-waits with a short count
#pragma unroll
        for (int i = 0; i < UP_IN_ITERS; ++i) asm volatile("" : "+v"(R.v[i].x), "+v"(R.v[i].y), "+v"(R.v[i].z), "+v"(R.v[i].w));
        // 18 (tap, channel-pair) groups: the two pixel fragments of a group feed every parity whose 2x2 window holds the tap.
        // Fragment reads run one group ahead of the MFMAs (two register sets), as in conv_rb.hip.
        bf16x8_t afr[2][2], bfr[2][4];
        auto read_group = [&](auto g_tag, bf16x8_t (&af)[2], bf16x8_t (&bf)[4]) __attribute__((always_inline)) {
            constexpr int g = decltype(g_tag)::value;
            constexpr int tap = g >> 1, cp = g & 1, ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int m = 0; m < 2; ++m)
                af[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ib + (a_off[m][tap] ^ (cp << 5))));
            int n = 0;
#pragma unroll
            for (int pa = 0; pa < 2; ++pa)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    const int dy = ky - pa, dx = kx - pb;
                    if (dy < 0 || dy > 1 || dx < 0 || dx > 1) continue;
                    const int par = pa * 2 + pb, tap4 = dy * 2 + dx;
                    bf[n++] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wb + ((par * 16 + tap4 * 4 + 2 * cp) * UP_NT) * 16));
                }
        };
        auto mfma_group = [&](auto g_tag, const bf16x8_t (&af)[2], const bf16x8_t (&bf)[4]) __attribute__((always_inline)) {
            constexpr int g = decltype(g_tag)::value;
            constexpr int tap = g >> 1, ky = tap / 3, kx = tap - ky * 3;
            int n = 0;
#pragma unroll
            for (int pa = 0; pa < 2; ++pa)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    const int dy = ky - pa, dx = kx - pb;
                    if (dy < 0 || dy > 1 || dx < 0 || dx > 1) continue;
                    const int par = pa * 2 + pb;
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[par][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[n], af[m], acc[par][m], 0, 0, 0);   // D[cout][pixel]
                    ++n;
                }
        };
        // fused form, levels whose skip tensor comes from HBM (C <= 64): the item's skip pixels are touched one dword per 32 B
        // during the last stage, so that the epilogue's 16-B fragment loads find them in L2 (the epilogue is a pure
        // load -> MFMA -> store phase: what it waits for is exactly these loads)
        constexpr bool PREFETCH = NKS > 0 && NKS <= 4 && LAST && (IRE_UP_ABL & 8);   // measured: 325 -> 376 us at level 0 (the touches' 64 sectors per instruction cost more than they hide): off
        unsigned pf_off = 0, pf_lim = 0, pf_dummy = 0;
        const char* pf_base = nullptr;
        if constexpr (PREFETCH) {
            const int C = a.cout;
            pf_base = reinterpret_cast<const char*>(a.in1) + (size_t)sq0.it.img * a.Hout * a.Wout * C * 2;
            pf_off = (unsigned)((2 * (sq0.it.ty * UP_TH + wave * 2)) * a.Wout + 2 * (sq0.it.tx * UP_TW + r)) * (unsigned)(C * 2) + (unsigned)(h * 32);
            pf_lim = (unsigned)(a.Hout * a.Wout * C * 2 - 4);
        }
        auto group = [&](auto g_tag) __attribute__((always_inline)) {
            constexpr int g = decltype(g_tag)::value;
            if constexpr (PREFETCH && g < 4 * NKS) {
                constexpr int NKG = NKS / 2, pm = g / NKG, kg = g % NKG, par = pm >> 1, m = pm & 1, pa = par >> 1, pb = par & 1;
                unsigned off = pf_off + (unsigned)(2 * m + pa) * (unsigned)(a.Wout * a.cout * 2) + (unsigned)(pb * a.cout * 2) + (unsigned)(kg * 64);
                off = off < pf_lim ? off : pf_lim;
                asm volatile("global_load_dword %0, %1, %2" : "+v"(pf_dummy) : "v"(off), "s"(pf_base) : "memory");
            }
            if constexpr (g + 1 < 18) read_group(std::integral_constant<int, g + 1>{}, afr[(g + 1) & 1], bfr[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);       // keep the reads AHEAD of this group's MFMAs
            mfma_group(g_tag, afr[g & 1], bfr[g & 1]);
            // stage s+1's input: registers -> the other LDS tile, and the same registers reloaded with stage s+2, in groups 0..4:
            // the reloads then have 13 groups of MFMAs (~3000 cycles) before the stage's one vmcnt(0)
            if constexpr (g < UP_IN_ITERS) { store_chunk(g, R, in_nxt); load_chunk(sq2, g, R); }
            // weight slab of stage s+1 by LDS-DMA into the other buffer, early in the stage
            if constexpr (g == 0) {
                const unsigned char* ws = wslab(sq1);
                const int wave_u = __builtin_amdgcn_readfirstlane(wave);
                const unsigned w_nxt_lds = smem_lds + (unsigned)(w_nxt - smem);
#pragma unroll
                for (int i = 0; i < UP_W_ITERS; ++i) {
                    const int cbase = i * UP_THREADS + wave_u * 64;
                    up_glds16(ws + (size_t)(cbase + lane) * 16, w_nxt_lds + cbase * 16);
                }
            }
            // fused form: the skip weights of this item's cout block (NKS KB) for the epilogue of the NEXT stage, one wave
            // instruction per KB; published by this stage's barrier, and the previous item's epilogue ended before the last one
            if constexpr (NKS > 0 && PRELAST && g == 1) {
                const int wave_u = __builtin_amdgcn_readfirstlane(wave);
                if (wave_u < NKS)
                    up_glds16(reinterpret_cast<const unsigned char*>(a.w1) + ((size_t)sq0.it.nb * NKS + wave_u) * 1024 + lane * 16,
                              smem_lds + UP_SKW_OFF + wave_u * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        read_group(std::integral_constant<int, 0>{}, afr[0], bfr[0]);
        group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{}); group(std::integral_constant<int, 2>{});
        group(std::integral_constant<int, 3>{}); group(std::integral_constant<int, 4>{}); group(std::integral_constant<int, 5>{});
        group(std::integral_constant<int, 6>{}); group(std::integral_constant<int, 7>{}); group(std::integral_constant<int, 8>{});
        group(std::integral_constant<int, 9>{}); group(std::integral_constant<int, 10>{}); group(std::integral_constant<int, 11>{});
        group(std::integral_constant<int, 12>{}); group(std::integral_constant<int, 13>{}); group(std::integral_constant<int, 14>{});
        group(std::integral_constant<int, 15>{}); group(std::integral_constant<int, 16>{}); group(std::integral_constant<int, 17>{});
        // the stage's one wait: the s+2 reloads and the DMA'd slab of s+1 (and the previous item's output stores)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (PREFETCH) asm volatile("" :: "v"(pf_dummy));      // the touches' destination register stayed reserved until here
        if constexpr (LAST && NKS > 0 && IRE_UP_TEPI) {
            if constexpr (!(IRE_UP_ABL & 1)) epi_prefetch(sq0.it);
            __syncthreads();              // the stage barrier first: the finished buffer becomes the waves' patches
            epilogue_t(sq0.it);           // ends with the barrier that publishes the partials: nobody restages that buffer before it
        } else {
            if constexpr (LAST) epilogue(sq0.it);
            __syncthreads();
        }
        sq0 = sq1; sq1 = sq2; sq2 = cursor.next();
    };

    // ---- prologue: stage 0 -> LDS buffer 0, stage 1 -> registers -----------------------------------------------------------
    {
        float* bl = reinterpret_cast<float*>(smem + UP_BIAS_OFF);
        if (tid < a.cout && tid < 256) bl[tid] = a.bias[tid];
        load_stage(sq0, R);
        const uint4* ws = reinterpret_cast<const uint4*>(wslab(sq0));
        uint4* wd = reinterpret_cast<uint4*>(smem + UP_IN_BYTES);
        for (int i = tid; i < UP_W_CHUNKS; i += UP_THREADS) wd[i] = ws[i];
        uint4* in0 = reinterpret_cast<uint4*>(smem);
#pragma unroll
        for (int i = 0; i < UP_IN_ITERS; ++i) store_chunk(i, R, in0);
        load_stage(sq1, R);
    }
    __syncthreads();
    // nkc is even: an item starts on an even stage and ends on an odd one; its last stage pair is peeled with the epilogue
    // (a conditional epilogue inside the loop splits live ranges of in-flight prefetch registers: conv_w4.hip)
    int s = 0;
    for (int k = 0; k < my_items; ++k) {
        init_acc(sq0.it.nb);
        for (int kc = 0; kc + 2 < nkc; kc += 2) {
            stage(s, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); ++s;
            stage(s, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); ++s;
        }
        stage(s, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}); ++s;
        stage(s, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}); ++s;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may be in flight when the workgroup's LDS is released
}

}  // namespace

// a.in0 = low-res tensor [nimg][Hin (+halo)][Win][Cin], a.out = [nimg][2*Hin][2*Win][cout]; a.nkc = Cin/32 (even), a.nblocks =
// cout/32, a.tiles_x/y = low-res tiles of 16x32; a.w = sub-pixel slabs [nblock][kc][parity][kk][32 permuted rows][8].
// Fused with the 1x1 `fuse` (a.in1 != null): a.w / a.bias = the composed weights, a.in1 = skip tensor [nimg][2*Hin][2*Win][cout]
// (first real row), a.w1 = skip weights [nblock][ks = cout/16][h][32 permuted rows][8], a.stats = partials per (image, low-res tile).
void conv_up_subpixel_launch(const ConvArgs& a, hipStream_t stream) {
    if (a.nkc < 2 || (a.nkc & 1) || a.cout > 256) fail(IRE_ERR_INTERNAL, "internal: conv_up_subpixel shape");
    if (a.in1 && (!a.w1 || !a.stats || (a.cout != 32 && a.cout != 64 && a.cout != 128))) fail(IRE_ERR_INTERNAL, "internal: fused conv_up arguments");
    const int items = a.tiles_x * a.tiles_y * a.nimg * a.nblocks;
    const int cus = persistent_grid_cus();
    const int grid = items < cus ? items : cus;
    if (!a.in1) hipLaunchKernelGGL(conv_up_kernel<0>, dim3(grid), dim3(UP_THREADS), 0, stream, a);
    else if (a.cout == 32) hipLaunchKernelGGL(conv_up_kernel<2>, dim3(grid), dim3(UP_THREADS), 0, stream, a);
    else if (a.cout == 64) hipLaunchKernelGGL(conv_up_kernel<4>, dim3(grid), dim3(UP_THREADS), 0, stream, a);
    else hipLaunchKernelGGL(conv_up_kernel<8>, dim3(grid), dim3(UP_THREADS), 0, stream, a);
    IRE_HIP(hipGetLastError());
}

}  // namespace ire
