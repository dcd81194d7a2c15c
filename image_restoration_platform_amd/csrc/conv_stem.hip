// conv_stem.hip -- RestoreNet-v0's first layer (u8 RGB -> 32 channels, 3x3) as a kernel of its own, gfx950.
//
// K = 3 x 3 x 3 = 27: one 16x32-pixel tile of u8 input is 1.8 KB and the whole layer is three MFMAs per 32 pixels, so the
// kernel is its output stream (64 B per pixel: 537 MB per 8 x 1024^2) plus a small epilogue.  The v1 template (conv_mfma.hip)
// padded the 3 channels to 8 and ran the generic 9-tap schedule: 177 us; this kernel 150 us (181 before the next tile's bytes were
// requested ahead).  An ideal fill of the output's size takes 84 us.
//
//   * a workgroup (512 threads) stages a tile's 18 x 34 x 3 bytes as bf16 (u8 -> bf16 is exact) in LDS, zero outside the image;
//     the bytes of the NEXT tile are requested before this tile's arithmetic (four per thread, in registers)
//   * the LDS tile holds a pixel as FOUR bf16 (R, G, B, 0): 8 bytes.  K is laid out as k = 16 ky + 4 kx + c with kx in 0..3 and
//     c in 0..3 (kx = 3 and c = 3 carry zero weights): k-step ky of `v_mfma_f32_32x32x16_bf16` is one tile row, and the B
//     fragment of lane (pixel r, half h) -- k = 16 ky + 8 h .. + 8 = pixels r + 2h, r + 2h + 1 -- is 16 CONTIGUOUS bytes: one
//     `ds_read2_b64` per fragment, conflict-free.  (Round 2 packed K to 27 = two k-steps and gathered each fragment with eight
//     16-bit reads from per-lane offsets: 40 % of its LDS cycles were bank conflicts, 62 vector instructions per MFMA.)
//   * a wave owns two pixel rows: per row three MFMAs, couts on rows (permuted like every other slab: a lane owns 8 contiguous couts)
//   * epilogue as conv_pc's: accumulators start at the bias, bf16 stores straight from the accumulators, GroupNorm partial
//     statistics (8 groups of 4 channels) of the stored values per tile
#include "conv_mfma.hpp"

#include <type_traits>

namespace ire {

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int ST_THREADS = 512, ST_TH = 16, ST_TW = 32, ST_IH = ST_TH + 2, ST_IW = ST_TW + 2;
constexpr int ST_RB = ST_IW * 3;            // bytes of a tile row in the image: 102
constexpr int ST_PW = ST_IW + 2;            // pixels per LDS row: the fragment of pixel r = 31, h = 1 reaches pixel 34 (zero weight) -- kept finite (zero)
constexpr int ST_ROW = ST_PW * 4;           // LDS row pitch in bf16 elements (4 per pixel)
constexpr int ST_ITERS = (ST_IH * ST_RB + ST_THREADS - 1) / ST_THREADS;     // 4 bytes per thread and tile

__device__ __forceinline__ unsigned st_pack(float a, float b) {
    f32x2_t f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
template <int N> __device__ __forceinline__ float st_ror_add(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float st_swap16_add(float v) {
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}

// two workgroups per CU (119 registers): they overlap one another's staging, arithmetic and stores; three would need <= 80
// registers (36 spilled: 289 us against 150)
__global__ __launch_bounds__(ST_THREADS) void conv_stem_kernel(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[2][ST_IH * ST_ROW];
    __shared__ float red[2][8][8][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;

    // A fragments (weights): [ky][h][32 permuted rows][8] bf16, engine.cpp::make_conv (d_wstem)
    const u32x4_t* wf = reinterpret_cast<const u32x4_t*>(a.w);
    const bf16x8_t w0 = __builtin_bit_cast(bf16x8_t, wf[(0 * 2 + h) * 32 + r]);
    const bf16x8_t w1 = __builtin_bit_cast(bf16x8_t, wf[(1 * 2 + h) * 32 + r]);
    const bf16x8_t w2 = __builtin_bit_cast(bf16x8_t, wf[(2 * 2 + h) * 32 + r]);
    // accumulator i of lane-half h is cout 16 (i >> 3) + 8 h + (i & 7) (permuted slab rows)
    f32x16_t bias_acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) bias_acc[i] = a.bias[16 * (i >> 3) + 8 * h + (i & 7)];
    // this lane's B fragments: 8 elements from pixel r + 2h of tile row 2 wave + m + ky
    const int foff = (2 * wave * ST_PW + r + 2 * h) * 4;      // element index for m = 0, ky = 0
    // the fourth element of every pixel and the two pad pixels of every row are never staged: zero them once (a zero weight does
    // not neutralise a NaN bit pattern left in LDS)
    for (int i = tid; i < 2 * ST_IH * ST_ROW / 2; i += ST_THREADS) reinterpret_cast<unsigned*>(&tile[0][0])[i] = 0u;
    __syncthreads();
    // this thread's bytes of a tile: element i = tid + 512 it -> (row py, byte b of the row): the same for every tile
    int spy[ST_ITERS], sb[ST_ITERS], spx[ST_ITERS], srel[ST_ITERS];
#pragma unroll
    for (int it = 0; it < ST_ITERS; ++it) {
        const int i = tid + it * ST_THREADS;
        spy[it] = i / ST_RB; sb[it] = i - spy[it] * ST_RB;
        spx[it] = sb[it] / 3;
        srel[it] = (spy[it] * a.Win + spx[it]) * 3 + (sb[it] - spx[it] * 3);        // byte offset from the tile's first halo pixel
    }

    const int tiles_per_img = a.tiles_x * a.tiles_y, total = tiles_per_img * a.nimg;
    const unsigned char* in = reinterpret_cast<const unsigned char*>(a.in0);
    const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
    unsigned pv[ST_ITERS];
    auto request = [&](int item) {                  // the tile's bytes -> pv (zero outside the image)
        const int img = item / tiles_per_img, t = item - img * tiles_per_img;
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        const int iy0 = ty * ST_TH - 1, ix0 = tx * ST_TW - 1;
        const long long base = (((long long)img * a.in_rows + iy0 + a.in_row_off) * a.Win + ix0) * 3;      // (may point before the buffer: only used where ok)
#pragma unroll
        for (int it = 0; it < ST_ITERS; ++it) {
            const bool ok = spy[it] < ST_IH && (unsigned)(iy0 + spy[it] - a.iy_lo) < (unsigned)a.iy_span && (unsigned)(ix0 + spx[it]) < (unsigned)a.Win;
            const unsigned v = in[ok ? base + srel[it] : 0];
            pv[it] = ok ? v : 0u;
        }
    };
    int par = 0;
    if ((int)blockIdx.x < total) request(blockIdx.x);
    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        const int img = item / tiles_per_img, t = item - img * tiles_per_img;
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        unsigned short* tl = tile[par];
#pragma unroll
        for (int it = 0; it < ST_ITERS; ++it)
            if (spy[it] < ST_IH) tl[spy[it] * ST_ROW + spx[it] * 4 + (sb[it] - spx[it] * 3)] = (unsigned short)(__builtin_bit_cast(unsigned, (float)pv[it]) >> 16);      // u8 -> bf16: exact
        if (item + (int)gridDim.x < total) request(item + gridDim.x);      // in flight across this tile's arithmetic and stores
        __syncthreads();                               // tile[par] is staged (its previous readers passed the barrier of the tile before last)
        // ---- per pixel row: two MFMAs, then stores + GroupNorm partials (groups of 4 channels: a lane's 8 contiguous couts are two groups) ----
        const int ox = tx * ST_TW + r;
        char* obase = reinterpret_cast<char*>(a.out) + (size_t)img * a.Hout * a.Wout * 64;
        float gs[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, gq[2][2] = {{0.f, 0.f}, {0.f, 0.f}};      // [pp][half of the 8 couts]
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            u32x4_t f[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const uint2* fp = reinterpret_cast<const uint2*>(tl + foff + (m + ky) * ST_ROW);      // 8-byte aligned, 16 contiguous bytes
                const uint2 lo = fp[0], hi = fp[1];
                f[ky] = u32x4_t{lo.x, lo.y, hi.x, hi.y};
            }
            f32x16_t c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, __builtin_bit_cast(bf16x8_t, f[0]), bias_acc, 0, 0, 0);     // D[cout][pixel]
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, __builtin_bit_cast(bf16x8_t, f[1]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, __builtin_bit_cast(bf16x8_t, f[2]), c, 0, 0, 0);
            const int oy = ty * ST_TH + 2 * wave + m;
            const bool inb = ox < a.Wout && oy < a.Hout;
            const float mf = inb ? 1.f : 0.f;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const unsigned w[4] = {st_pack(c[8 * pp + 0], c[8 * pp + 1]), st_pack(c[8 * pp + 2], c[8 * pp + 3]),
                                       st_pack(c[8 * pp + 4], c[8 * pp + 5]), st_pack(c[8 * pp + 6], c[8 * pp + 7])};
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const bf16x2_t wv = __builtin_bit_cast(bf16x2_t, w[d]);
                    const float s1 = __builtin_amdgcn_fdot2_f32_bf16(wv, ones, 0.f, false), q1 = __builtin_amdgcn_fdot2_f32_bf16(wv, wv, 0.f, false);
                    gs[pp][d >> 1] = __builtin_fmaf(s1, mf, gs[pp][d >> 1]);
                    gq[pp][d >> 1] = __builtin_fmaf(q1, mf, gq[pp][d >> 1]);
                }
                if (inb) {
                    const u32x4_t wv4 = {w[0], w[1], w[2], w[3]};
                    *reinterpret_cast<u32x4_t*>(obase + ((size_t)oy * a.Wout + ox) * 64 + (16 * pp + 8 * h) * 2) = wv4;
                }
            }
            __builtin_amdgcn_sched_barrier(0);         // one row's accumulators live at a time
        }
        if (a.stats) {
            // Sum of the 8 values (4 groups x (sum, squares)) over the 32 lanes of a half, TRANSPOSING for the first two steps as
            // conv_pc.hip does: a lane keeps half of its values and hands the other half to its partner (lane ^ 1, then lane ^ 2), so
            // 8 -> 4 -> 2 values per lane; those take the plain steps over lane bits 2, 3 and 4.  Value index v = 2 pp + hf: group
            // 4 pp + 2 h + hf.  Lane l of a half ends with (kind = b0) of the values 2 b1 and 2 b1 + 1, i.e. pp = b1, hf = 0 / 1.
            const bool b0 = lane & 1, b1 = lane & 2;
            auto xch = [&](float keep, float give, auto ctrl_tag) __attribute__((always_inline)) -> float {
                const int gg = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), decltype(ctrl_tag)::value, 0xf, 0xf, false);
                return keep + __builtin_bit_cast(float, gg);
            };
            const float vs[4] = {gs[0][0], gs[0][1], gs[1][0], gs[1][1]}, vq[4] = {gq[0][0], gq[0][1], gq[1][0], gq[1][1]};
            float u[4], t2[2];
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = xch(b0 ? vq[k] : vs[k], b0 ? vs[k] : vq[k], std::integral_constant<int, 0xb1>{});     // quad_perm [1,0,3,2]: lane ^ 1
#pragma unroll
            for (int k = 0; k < 2; ++k) t2[k] = xch(b1 ? u[2 + k] : u[k], b1 ? u[k] : u[2 + k], std::integral_constant<int, 0x4e>{});   // quad_perm [2,3,0,1]: lane ^ 2
            t2[0] = st_ror_add<4>(t2[0]); t2[1] = st_ror_add<4>(t2[1]);
            t2[0] = st_ror_add<8>(t2[0]); t2[1] = st_ror_add<8>(t2[1]);
            t2[0] = st_swap16_add(t2[0]); t2[1] = st_swap16_add(t2[1]);
            if ((lane & 28) == 0) {
                const int g0 = 4 * (b1 ? 1 : 0) + 2 * h;      // hf = 0; hf = 1 is the next group
                red[par][wave][g0][b0 ? 1 : 0] = t2[0];
                red[par][wave][g0 + 1][b0 ? 1 : 0] = t2[1];
            }
            __syncthreads();                           // (also: every wave is done with tile[par] long before it is staged again, two tiles on)
            if (tid < 8) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) { s += red[par][w][tid][0]; q += red[par][w][tid][1]; }
                float* st = a.stats + (((size_t)img * tiles_per_img + t) * 8 + tid) * 2;
                st[0] = s; st[1] = q;
            }
        }
        par ^= 1;
    }
}

}  // namespace

// a.in0 = u8 [nimg][in_rows][Win][3] (rows iy_lo .. iy_lo + iy_span readable, the rest zero), a.out = bf16 [nimg][Hout][Wout][32],
// a.w = A fragments [3][2][32][8] bf16 (engine.cpp::make_conv), a.bias[32], a.stats partials [img][tile][8][2]; 16x32 tiles
void conv_stem_launch(const ConvArgs& a, hipStream_t stream) {
    if (a.cout != 32 || a.Hout != a.Hin || a.Wout != a.Win || !a.stats) fail(IRE_ERR_INTERNAL, "internal: conv_stem shape");
    const int items = a.tiles_x * a.tiles_y * a.nimg;
    const int cus = persistent_grid_cus();
    const int grid = items < 2 * cus ? items : 2 * cus;
    hipLaunchKernelGGL(conv_stem_kernel, dim3(grid), dim3(ST_THREADS), 0, stream, a);
    IRE_HIP(hipGetLastError());
}

}  // namespace ire
