// conv_f8.hip -- IRE_PRECISION_FP8 (BASELINE.json cfg 4 "fp8 conv MFMA"): the C >= 128 ResBlock convolutions on the
// block-scaled fp8 matrix instruction of gfx950, v_mfma_scale_f32_32x32x64_f8f6f4 -- K = 64 per instruction at twice the cycles
// of the bf16 32x32x16 form, i.e. 2x the bf16 MFMA rate -- with OCP e4m3 operands and unit block scales (E8M0 127); the real
// scales are one per output channel (weights, applied in the epilogue) and one constant (activations, x16).
//
// Operand map (tools/microbench/mfma_scale_probe.hip, run on an MI355X): lane l holds 32 bytes; byte j of lane-half h of A
// pairs with byte j of half h of B (identity k map), rows / columns on l & 31, the standard 32x32 C/D map.  So ANY assignment
// of (half, byte) to (tap, channel) works if both operands use it: here half h <-> tap 2i + h of k-step i, byte <-> one of
// the stage's 32 input channels.  9 taps = 5 k-steps; the missing tenth tap reads a block of zero weights.
//
// Schedule: conv_up.hip / conv_down.hip's (512-thread workgroup per CU, two LDS buffers, one prefetch register set stored to
// LDS and reloaded in place, weights by LDS-DMA, one barrier per stage), tile = 16x32 pixels x 128 couts, wave = 2 rows x 128
// couts (128 accumulators), stage = 32 input channels = 40 MFMAs of 64 cycles per wave (bf16: 144 of 32 for the same work).
// GroupNorm+FiLM+SiLU and the e4m3 conversion happen while staging (activations stay bf16 in HBM); epilogue as conv_w4.hip's
// (accumulators start at bias / scale; dequantise, residual, bf16 stores, GroupNorm partials).  LDS tiles are two 16-channel
// planes per pixel / per cout row, so every fragment is two conflict-free ds_read_b128.
#include "conv_mfma.hpp"
#include "persist.hpp"
#include "gn_fold.hpp"

namespace ire {

namespace {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x8_t __attribute__((ext_vector_type(8)));

constexpr float kF8ActScale = 16.0f;
constexpr int F8_THREADS = 512;
constexpr int F8_TH = 16, F8_TW = 32, F8_IH = 18, F8_IW = 34;
constexpr int F8_PX = F8_IH * F8_IW;                    // 612 halo-tile pixels
constexpr int F8_IN_CHUNKS = F8_PX * 4;                  // 8-channel chunks of a 32-channel stage: 2448
constexpr int F8_IN_ITERS = (F8_IN_CHUNKS + F8_THREADS - 1) / F8_THREADS;     // 5
constexpr int F8_PLANE = F8_PX * 16;                     // one 16-channel plane: 9792 B ((9792/4) % 32 == 16: conflict-free tile writes)
constexpr int F8_IN_BYTES = 2 * F8_PLANE + 64;           // + dummy slot for the chunk slots past the tile
constexpr int F8_NT = 128, F8_NTL = 4;
constexpr int F8_W_BYTES = 9 * 2 * F8_NT * 16;           // slab [tap][half][128 rows][16 B] = 36 864 B
constexpr int F8_W_CHUNKS = F8_W_BYTES / 16;             // 2304
constexpr int F8_W_ITERS = (F8_W_CHUNKS + F8_THREADS - 1) / F8_THREADS;       // 5 (the last one wraps)
constexpr int F8_BUF = F8_IN_BYTES + F8_W_BYTES;
constexpr int F8_ZERO_OFF = 2 * F8_BUF;                  // 4 KB of zero weights: the tenth tap
constexpr int F8_RED_OFF = F8_ZERO_OFF + 4096;           // [8 waves][8 slots][2] floats
constexpr int F8_BIAS_OFF = F8_RED_OFF + 8 * 8 * 2 * 4;
constexpr int F8_SCALE_OFF = F8_BIAS_OFF + 1024;
constexpr int F8_COEF_OFF = F8_SCALE_OFF + 1024;         // [2 slots][A 32 | B 32] floats
constexpr int F8_LDS = F8_COEF_OFF + 2 * 256;
static_assert((F8_PLANE / 4) % 32 == 16, "plane offset must be half a bank row");
static_assert(F8_LDS <= 160 * 1024, "LDS");

__device__ __forceinline__ unsigned f8_pack(float a, float b) {
    f32x2_t f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ float f8_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float f8_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ void f8_glds16(const void* gsrc, unsigned lds_dst_uniform) {   // see conv_rb.hip::rb_glds16
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
template <int N> __device__ __forceinline__ float f8_ror_add(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float f8_swap16_add(float v) {
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float f8_swap32_add(float v) {
    float x = v, y = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}

struct F8Regs { uint4 v[F8_IN_ITERS]; unsigned ok; };

template <bool RESID>
__global__ __launch_bounds__(F8_THREADS) void conv_f8_kernel(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[F8_LDS];
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int c8_fixed = tid & 3;                         // this thread always stages the same 8-channel slice of a pixel

    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int nkc = a.nkc;                                // 32-channel stages per item (4 or 8: even)
    PersistCursor cursor(a.tiles_x, a.tiles_y, a.nimg, a.nblocks, nkc);
    const int my_items = cursor.my_items;
    const int S = cursor.S;
    if (S == 0) return;
    if (a.gn_stats) gn_fold(a, smem, cursor.first_img, cursor.last_img);     // GroupNorm finalize of the input tensor, folded in (gn_fold.hpp)
    using StageInfo = PersistStage;
    StageInfo sq0 = cursor.cur, sq1 = cursor.next(), sq2 = cursor.next();

    const int Cin = a.cin0;
    const int cin_shift = 31 - __builtin_clz(Cin);

    // per-lane LDS offsets: pixel fragment of row m at the tap this lane-half serves in k-step i (tap 2i + h; the tenth tap does
    // not exist: its weights are zero, any pixel will do), weight fragment of that tap (or the zero block)
    int p_off[2][5], w_off[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int tap = min(2 * i + h, 8);
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int m = 0; m < 2; ++m) p_off[m][i] = ((wave * 2 + m + ky) * F8_IW + r + kx) * 16;
        w_off[i] = (2 * i + h > 8) ? F8_ZERO_OFF + r * 16 : F8_IN_BYTES + (2 * i + h) * 2 * F8_NT * 16 + r * 16;   // + buffer base (not for the zero block)
    }

    auto load_chunk = [&](const StageInfo& si, int i, F8Regs& R) {
        const PersistItem& it = si.it;
        const int oy1 = it.ty * F8_TH - 1, ox1 = it.tx * F8_TW - 1;
        const char* base = reinterpret_cast<const char*>(a.in0) + (size_t)it.img * a.in_rows * a.Win * Cin * 2 + si.kc * 64;
        const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, a.in_rows * a.Win * Cin * 2 - si.kc * 64, 0x00020000);
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int p = (t2 + i * F8_THREADS) >> 2;
        const int py = p / F8_IW, px = p - py * F8_IW;
        const int iy = oy1 + py, ix = ox1 + px;
        const bool ok = (unsigned)(iy - a.iy_lo) < (unsigned)a.iy_span && (unsigned)ix < (unsigned)a.Win;
        const unsigned off = ok ? ((unsigned)((iy + a.in_row_off) * a.Win + ix) << (cin_shift + 1)) + (unsigned)(c8_fixed * 16) : 0xffffffffu;
        const u32x4_t lv = __builtin_amdgcn_raw_buffer_load_b128(irsrc, off, 0, 0);
        R.v[i] = make_uint4(lv.x, lv.y, lv.z, lv.w);
        R.ok = (R.ok & ~(1u << i)) | (ok ? (1u << i) : 0u);           // zero padding applies AFTER the activation
    };
    auto load_stage = [&](const StageInfo& si, F8Regs& R) {
        R.ok = 0;
#pragma unroll
        for (int i = 0; i < F8_IN_ITERS; ++i) load_chunk(si, i, R);
    };
    // GroupNorm+FiLM coefficients of a stage's 32 channels: global -> 16 lanes' float4 -> LDS slot (A[32] | B[32]) -> every thread's
    // 8 + 8 floats, pre-multiplied by the activation scale so silu comes out ready for the e4m3 conversion
    float* coef_lds = reinterpret_cast<float*>(smem + F8_COEF_OFF);
    auto fetch_coeffs = [&](const StageInfo& si) -> float4 {
        return reinterpret_cast<const float4*>(a.ab + (size_t)si.it.img * Cin + si.kc * 32)[tid & 15];
    };
    auto put_coeffs = [&](int slot, const float4& v) {
        if (tid < 16) {
            reinterpret_cast<float2*>(coef_lds + slot * 64)[tid] = make_float2(v.x, v.z);
            reinterpret_cast<float2*>(coef_lds + slot * 64 + 32)[tid] = make_float2(v.y, v.w);
        }
    };
    float cA[8], cB[8];
    auto load_coeffs = [&](int slot) {
        const float4* pa = reinterpret_cast<const float4*>(coef_lds + slot * 64 + c8_fixed * 8);
        const float4* pb = reinterpret_cast<const float4*>(coef_lds + slot * 64 + 32 + c8_fixed * 8);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float4 va = pa[e], vb = pb[e];
            cA[4 * e] = va.x * kF8ActScale; cA[4 * e + 1] = va.y * kF8ActScale; cA[4 * e + 2] = va.z * kF8ActScale; cA[4 * e + 3] = va.w * kF8ActScale;
            cB[4 * e] = vb.x * kF8ActScale; cB[4 * e + 1] = vb.y * kF8ActScale; cB[4 * e + 2] = vb.z * kF8ActScale; cB[4 * e + 3] = vb.w * kF8ActScale;
        }
    };
    // 8 channels: y' = 16 (x A + B); 16 silu(y) = y' / (1 + 2^(-y' log2e / 16)); clamp at 448; e4m3 (round to nearest even)
    auto store_chunk = [&](int i, const F8Regs& R, unsigned char* lds_in) {
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int idx = t2 + i * F8_THREADS;
        const bool ok = (R.ok >> i) & 1u;
        const unsigned wds[4] = {R.v[i].x, R.v[i].y, R.v[i].z, R.v[i].w};
        int q[2] = {0, 0};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            // plain f32 instructions (built with -fno-slp-vectorize): packed f32 is slower beside MFMAs, see conv_w4.hip
            constexpr float K = -1.4426950408889634f / kF8ActScale;
            const float y0 = __builtin_fmaf(f8_lo(wds[d]), cA[2 * d], cB[2 * d]), y1 = __builtin_fmaf(f8_hi(wds[d]), cA[2 * d + 1], cB[2 * d + 1]);
            const float e0 = __builtin_amdgcn_exp2f(y0 * K) + 1.0f, e1 = __builtin_amdgcn_exp2f(y1 * K) + 1.0f;
            const f32x2_t sv = {y0 * __builtin_amdgcn_rcpf(e0), y1 * __builtin_amdgcn_rcpf(e1)};
            const float f0 = __builtin_fminf(sv.x, 448.0f), f1 = __builtin_fminf(sv.y, 448.0f);
            q[d >> 1] = (d & 1) ? __builtin_amdgcn_cvt_pk_fp8_f32(f0, f1, q[d >> 1], true) : __builtin_amdgcn_cvt_pk_fp8_f32(f0, f1, q[d >> 1], false);
        }
        uint2 o8;
        o8.x = ok ? (unsigned)q[0] : 0u; o8.y = ok ? (unsigned)q[1] : 0u;
        // pixel p, channels 8*c8 .. 8*c8+7: plane c8 >> 1, bytes (c8 & 1) * 8 of the pixel's 16
        unsigned char* dst = idx < F8_IN_CHUNKS ? lds_in + (c8_fixed >> 1) * F8_PLANE + (idx >> 2) * 16 + (c8_fixed & 1) * 8 : lds_in + 2 * F8_PLANE;
        *reinterpret_cast<uint2*>(dst) = o8;
    };
    auto wslab = [&](const StageInfo& si) -> const unsigned char* {
        return reinterpret_cast<const unsigned char*>(a.w) + ((size_t)si.it.nb * nkc + si.kc) * F8_W_BYTES;
    };

    f32x16_t acc[2][F8_NTL];
    float* red = reinterpret_cast<float*>(smem + F8_RED_OFF);
    const float* bias_lds = reinterpret_cast<const float*>(smem + F8_BIAS_OFF);
    const float* scale_lds = reinterpret_cast<const float*>(smem + F8_SCALE_OFF);
    auto init_acc = [&](int nb) {          // accumulators start at bias / oscale (a.bias holds that quotient): accumulator i of lane-half h is cout nb*128 + j*32 + 16(i>>3) + 8h + (i&7)
        const float* bl = bias_lds + nb * F8_NT + 8 * h;
#pragma unroll
        for (int j = 0; j < F8_NTL; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *reinterpret_cast<const float4*>(bl + j * 32 + 16 * (q >> 1) + 4 * (q & 1));
#pragma unroll
                for (int m = 0; m < 2; ++m) { acc[m][j][4 * q + 0] = bv.x; acc[m][j][4 * q + 1] = bv.y; acc[m][j][4 * q + 2] = bv.z; acc[m][j][4 * q + 3] = bv.w; }
            }
    };

    // ---- epilogue (conv_w4.hip's): dequantise, residual, bf16 stores straight from the accumulators, GroupNorm partials ------------
    auto epilogue = [&](const PersistItem& it) __attribute__((always_inline)) {
        int r_e = r, h_e = h, w_e = wave;
        asm volatile("" : "+v"(r_e), "+v"(h_e), "+v"(w_e));
        const int cout0 = it.nb * F8_NT;
        char* obase = reinterpret_cast<char*>(a.out) + (size_t)it.img * a.Hout * a.Wout * a.cout * 2;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, a.Hout * a.Wout * a.cout * 2, 0x00020000);
        const char* rbase = RESID ? reinterpret_cast<const char*>(a.resid) + (size_t)it.img * a.Hout * a.Wout * a.cout * 2 : nullptr;
        const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
        const int ox = it.tx * F8_TW + r_e;
        bool inb[2];
        unsigned offs[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int oy = it.ty * F8_TH + w_e * 2 + m;
            inb[m] = oy < a.Hout && ox < a.Wout;
            offs[m] = ((unsigned)((min(oy, a.Hout - 1) * a.Wout + min(ox, a.Wout - 1)) * a.cout + cout0) << 1) + (unsigned)(h_e * 16);
        }
        float ssum[F8_NTL][2], qsum[F8_NTL][2];
        uint4 rv[2][2];
        auto load_resid = [&](int g, uint4 (&dst)[2]) {
            if constexpr (RESID) {
#pragma unroll
                for (int m = 0; m < 2; ++m) dst[m] = *reinterpret_cast<const uint4*>(rbase + offs[m] + (unsigned)((g >> 1) * 64 + (g & 1) * 32));
            }
        };
        load_resid(0, rv[0]);
#pragma unroll
        for (int g = 0; g < F8_NTL * 2; ++g) {
            const int j = g >> 1, pp = g & 1;
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < F8_NTL * 2) load_resid(g + 1, rv[(g + 1) & 1]);
            const float4 s0 = *reinterpret_cast<const float4*>(scale_lds + cout0 + j * 32 + 16 * pp + 8 * h_e);
            const float4 s1 = *reinterpret_cast<const float4*>(scale_lds + cout0 + j * 32 + 16 * pp + 8 * h_e + 4);
            float ts = 0.f, tq = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f32x16_t& c = acc[m][j];
                unsigned w[4] = {f8_pack(c[8 * pp + 0] * s0.x, c[8 * pp + 1] * s0.y), f8_pack(c[8 * pp + 2] * s0.z, c[8 * pp + 3] * s0.w),
                                 f8_pack(c[8 * pp + 4] * s1.x, c[8 * pp + 5] * s1.y), f8_pack(c[8 * pp + 6] * s1.z, c[8 * pp + 7] * s1.w)};
                if constexpr (RESID) {
                    const uint4 rr = rv[g & 1][m];
                    const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                    for (int d = 0; d < 4; ++d) w[d] = f8_pack(f8_lo(w[d]) + f8_lo(rw[d]), f8_hi(w[d]) + f8_hi(rw[d]));
                }
                float s1v = 0.f, q1v = 0.f;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const bf16x2_t wv = __builtin_bit_cast(bf16x2_t, w[d]);
                    s1v = __builtin_amdgcn_fdot2_f32_bf16(wv, ones, s1v, false);
                    q1v = __builtin_amdgcn_fdot2_f32_bf16(wv, wv, q1v, false);
                }
                ts += inb[m] ? s1v : 0.f; tq += inb[m] ? q1v : 0.f;
                const u32x4_t wv4 = {w[0], w[1], w[2], w[3]};
                __builtin_amdgcn_raw_buffer_store_b128(wv4, orsrc, inb[m] ? offs[m] + (unsigned)(j * 64 + pp * 32) : 0xffffffffu, 0, 0);
            }
            ssum[j][pp] = ts; qsum[j][pp] = tq;
        }
        // GroupNorm partials: group size 16 or 32 here (C >= 128), slot (j, pp) = 16 couts
        float rv16[F8_NTL * 4];
#pragma unroll
        for (int g = 0; g < F8_NTL * 2; ++g) { rv16[2 * g] = ssum[g >> 1][g & 1]; rv16[2 * g + 1] = qsum[g >> 1][g & 1]; }
#pragma unroll
        for (int i = 0; i < F8_NTL * 4; ++i) rv16[i] = f8_ror_add<1>(rv16[i]);
#pragma unroll
        for (int i = 0; i < F8_NTL * 4; ++i) rv16[i] = f8_ror_add<2>(rv16[i]);
#pragma unroll
        for (int i = 0; i < F8_NTL * 4; ++i) rv16[i] = f8_ror_add<4>(rv16[i]);
#pragma unroll
        for (int i = 0; i < F8_NTL * 4; ++i) rv16[i] = f8_ror_add<8>(rv16[i]);
#pragma unroll
        for (int i = 0; i < F8_NTL * 4; ++i) rv16[i] = f8_swap16_add(rv16[i]);
#pragma unroll
        for (int i = 0; i < F8_NTL * 4; ++i) rv16[i] = f8_swap32_add(rv16[i]);
        if (lane == 0) {
#pragma unroll
            for (int g = 0; g < F8_NTL * 2; ++g) *reinterpret_cast<float2*>(red + (wave * 8 + g) * 2) = make_float2(rv16[2 * g], rv16[2 * g + 1]);
        }
        __syncthreads();
        const int G = a.group_size, spg = G >> 4, ngl = F8_NT / G;     // slots per group (1 or 2), groups in the item
        if (tid < ngl) {
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w)
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (k < spg) { sv += red[(w * 8 + tid * spg + k) * 2 + 0]; qv += red[(w * 8 + tid * spg + k) * 2 + 1]; }
            const int gg = cout0 / G + tid;
            float* st = a.stats + (((size_t)it.img * tiles_per_img + it.tile) * 8 + gg) * 2;
            st[0] = sv; st[1] = qv;
        }
    };

    F8Regs R;            // stage s+1's input; chunk i goes to LDS (activated, e4m3) and is reloaded with stage s+2 in place
    float4 cnext = make_float4(0.f, 0.f, 0.f, 0.f);
    auto stage = [&](auto par_tag, auto last_tag) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;
        const unsigned char* ib = smem + PAR * F8_BUF;
        unsigned char* in_nxt = smem + (PAR ^ 1) * F8_BUF;
        const unsigned char* wb = smem + PAR * F8_BUF;          // w_off already carries F8_IN_BYTES (or points at the zero block)
        unsigned char* w_nxt = smem + (PAR ^ 1) * F8_BUF + F8_IN_BYTES;
        // R was retired by the vmcnt(0) that ended the previous stage; its coefficients were published to slot PAR before that
        // stage's barrier
#pragma unroll
        for (int i = 0; i < F8_IN_ITERS; ++i) asm volatile("" : "+v"(R.v[i].x), "+v"(R.v[i].y), "+v"(R.v[i].z), "+v"(R.v[i].w));
        {   // weight slab of stage s+1 by LDS-DMA into the other buffer (the wrap re-fetches a group another wave also fetches)
            const unsigned char* ws = wslab(sq1);
            const int wave_u = __builtin_amdgcn_readfirstlane(wave);
            const unsigned w_nxt_lds = smem_lds + (unsigned)(w_nxt - smem);
#pragma unroll
            for (int i = 0; i < F8_W_ITERS; ++i) {
                int cbase = i * F8_THREADS + wave_u * 64;
                if ((i + 1) * F8_THREADS > F8_W_CHUNKS) cbase = cbase >= F8_W_CHUNKS ? cbase - F8_W_CHUNKS : cbase;
                f8_glds16(ws + (size_t)(cbase + lane) * 16, w_nxt_lds + cbase * 16);
            }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            // k-step i: taps 2i (lane-half 0) and 2i + 1 (half 1), 32 channels each: two 16-channel planes per operand.  The weight
            // fragments come in two rounds of two n-tiles and the transform runs in its own scheduling region, so that at most
            // 128 (accumulators) + 16 + 16 (fragments) registers are live beside the prefetch set: 256 is all a wave has here
            asm volatile("" ::: "memory");     // no fragment load may be hoisted above the previous k-step (they would all be spilled)
            i32x8_t pf[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const uint4 q0 = *reinterpret_cast<const uint4*>(ib + p_off[m][i]);
                const uint4 q1 = *reinterpret_cast<const uint4*>(ib + F8_PLANE + p_off[m][i]);
                pf[m] = i32x8_t{(int)q0.x, (int)q0.y, (int)q0.z, (int)q0.w, (int)q1.x, (int)q1.y, (int)q1.z, (int)q1.w};
            }
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                asm volatile("" ::: "memory");
                i32x8_t wf[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = jh * 2 + jj;
                    // half 1 of the last k-step reads the zero block, which has no buffer base
                    const unsigned char* wp = (i == 4 && h) ? smem + w_off[i] + j * 512 : wb + w_off[i] + j * 512;
                    const uint4 q0 = *reinterpret_cast<const uint4*>(wp);
                    const uint4 q1 = *reinterpret_cast<const uint4*>(wp + F8_NT * 16);
                    wf[jj] = i32x8_t{(int)q0.x, (int)q0.y, (int)q0.z, (int)q0.w, (int)q1.x, (int)q1.y, (int)q1.z, (int)q1.w};
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        acc[m][jh * 2 + jj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[jj], pf[m], acc[m][jh * 2 + jj], 0, 0, 0, 127, 0, 127);   // D[cout][pixel], e4m3 x e4m3, unit block scales
                // the results are not read before the epilogue: without this the MFMAs are sunk below the staging branches to the
                // end of the stage and every fragment is spilled to scratch on the way
                asm volatile("" : "+v"(acc[0][jh * 2]), "+v"(acc[0][jh * 2 + 1]), "+v"(acc[1][jh * 2]), "+v"(acc[1][jh * 2 + 1]));
                __builtin_amdgcn_sched_barrier(0);
            }
            // stage s+1's input: activate + quantise chunk i into the other tile, reload the registers with stage s+2
            load_coeffs(PAR);
            store_chunk(i, R, in_nxt);
            load_chunk(sq2, i, R);
            if (i == 4) cnext = fetch_coeffs(sq2);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the reloads, the coefficients, the DMA'd slab (and earlier output stores)
        asm volatile("" : "+v"(cnext.x), "+v"(cnext.y), "+v"(cnext.z), "+v"(cnext.w));
        put_coeffs(PAR ^ 1, cnext);        // for the next stage (which transforms the data just reloaded into R); published by the barrier below
        if constexpr (LAST) epilogue(sq0.it);
        __syncthreads();
        sq0 = sq1; sq1 = sq2; sq2 = cursor.next();
    };

    // ---- prologue --------------------------------------------------------------------------------------------------------
    {
        float* bl = reinterpret_cast<float*>(smem + F8_BIAS_OFF);
        float* sl = reinterpret_cast<float*>(smem + F8_SCALE_OFF);
        if (tid < a.cout && tid < 256) { bl[tid] = a.bias[tid]; sl[tid] = a.oscale[tid]; }
        for (int i = tid; i < 4096 / 16; i += F8_THREADS) reinterpret_cast<uint4*>(smem + F8_ZERO_OFF)[i] = make_uint4(0, 0, 0, 0);
        load_stage(sq0, R);
        put_coeffs(1, fetch_coeffs(sq0));
        const uint4* ws = reinterpret_cast<const uint4*>(wslab(sq0));
        uint4* wd = reinterpret_cast<uint4*>(smem + F8_IN_BYTES);
        for (int i = tid; i < F8_W_CHUNKS; i += F8_THREADS) wd[i] = ws[i];
        __syncthreads();
        load_coeffs(1);
#pragma unroll
        for (int i = 0; i < F8_IN_ITERS; ++i) store_chunk(i, R, smem);
        load_stage(sq1, R);
        put_coeffs(0, fetch_coeffs(sq1));      // stage 0 transforms stage 1's data with slot 0
    }
    __syncthreads();
    for (int k = 0; k < my_items; ++k) {
        init_acc(sq0.it.nb);
        for (int kc = 0; kc + 2 < nkc; kc += 2) {
            stage(std::integral_constant<int, 0>{}, std::false_type{});
            stage(std::integral_constant<int, 1>{}, std::false_type{});
        }
        stage(std::integral_constant<int, 0>{}, std::false_type{});
        stage(std::integral_constant<int, 1>{}, std::true_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may be in flight when the workgroup's LDS is released
}

}  // namespace

// a.nkc = Cin/32, a.nblocks = cout/128, tiles of 16x32; a.w = e4m3 slabs [nblock][kc32][tap][half][128 permuted rows][16 ch],
// a.bias = bias / oscale, a.oscale = weight scale of the channel / 16, a.ab = GroupNorm+FiLM coefficients (required)
void conv_f8_launch(bool resid, const ConvArgs& a, hipStream_t stream) {
    if (a.cout % 128 || a.cout > 256 || a.nkc < 2 || (a.nkc & 1) || !a.ab || !a.oscale || !a.stats) fail(IRE_ERR_INTERNAL, "internal: conv_f8 arguments");
    const int items = a.tiles_x * a.tiles_y * a.nimg * a.nblocks;
    const int cus = persistent_grid_cus();
    const int grid = items < cus ? items : cus;
    if (resid) hipLaunchKernelGGL(conv_f8_kernel<true>, dim3(grid), dim3(F8_THREADS), 0, stream, a);
    else hipLaunchKernelGGL(conv_f8_kernel<false>, dim3(grid), dim3(F8_THREADS), 0, stream, a);
    IRE_HIP(hipGetLastError());
}

}  // namespace ire
