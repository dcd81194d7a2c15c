// conv_mfma.hpp -- launch interface of the MFMA convolution template (conv_mfma.hip).
#pragma once
#include "common.hpp"

// Cache policy of the activation stores (raw_buffer_store aux: 1 = sc0, 2 = nt, 16 = sc1).  IRE_ST_LINE: the line-coalesced
// epilogues (a wave-instruction = whole 128-B lines) write THROUGH (sc1): the XCD's L2 keeps no dirty copy, so the kernel
// boundary has nothing to write back and the L2 stays with the input halos and weights -- same box, plain -> sc1: conv_pc<64>
// RB2 211.8 / 216.8 -> 196.8 / 196.3 us, RB1 172 -> 164 / 165.5, conv_w4 168.2 / 155.0 -> 165.6 / 152.0.  IRE_ST_PART: epilogues
// whose store instruction covers PARTS of lines (32 B per lane from the accumulator layout, 64 B of a 128-B pixel row) stay
// write-back -- written through, every part is its own memory write: conv_pc<32> 311 -> 449 us, conv_down 157 -> 201
// (profiles/r03_experiments.md).  Build-time A/B: IRE_ST_LINE=..., IRE_ST_PART=...
#ifndef IRE_ST_LINE
#define IRE_ST_LINE 16
#endif
// IRE_LD_ONCE: loads of bytes a kernel reads exactly once AS WHOLE LINES (conv_pc<64>'s residual rows, conv_up's level-0 skip rows)
// are non-temporal (nt): same box, conv_up<2> 299.4 / 296.1 -> 286.4 / 289.4 us, conv_pc<64> RB2 195.4 / 195.9 -> 191.9 / 191.9.
// Not where a line is read in two halves at different times (conv_up<4>, <8>: 64-B pieces of 128-B rows: 246 -> 267 us, 219 -> 234),
// and no effect on conv_w4's and conv_pc<32>'s residual loads.
#ifndef IRE_LD_ONCE
#define IRE_LD_ONCE 2
#endif
#ifndef IRE_LD_IN
#define IRE_LD_IN 0        // cache policy of conv_pc's input tile loads (build-time A/B)
#endif
#ifndef IRE_ST_PART
#define IRE_ST_PART 0
#endif

#include <cstdlib>

namespace ire {

// Workgroups a persistent convolution kernel launches at most: the device's CU count, or IRE_GRID_CUS when set (an A/B switch: with two
// lanes -- two HIP streams restoring half a batch each -- grids of HALF the CUs let the two lanes' kernels co-reside, so that one lane's
// dispatch gap, folded finalize, prologue and tail run beside the other lane's main loop instead of beside nothing).  Speed only: the
// results do not depend on which workgroup ran which tile.
inline int persistent_grid_cus() {
    static const int forced = [] { const char* v = std::getenv("IRE_GRID_CUS"); return v ? std::atoi(v) : 0; }();
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return forced > 0 && forced < cus ? forced : cus;
}

enum { PRO_NONE = 0, PRO_GN = 1, PRO_U8 = 2 };

// The seven ways RestoreNet-v0 uses a convolution (DESIGN.md "RestoreNet-v0").
enum ConvKind {
    CONV_STEM,  // u8 RGB (padded to 8 ch) -> 32, 3x3, GroupNorm stats out
    CONV_RB1,   // C->C 3x3, GN+FiLM+SiLU prologue, stats out
    CONV_RB2,   // C->C 3x3, GN+FiLM+SiLU prologue, + residual, stats out
    CONV_DOWN,  // C->2C 3x3 stride 2, stats out
    CONV_UP,    // nearest x2 then 2C->C 3x3
    CONV_FUSE,  // concat(up, skip) 2C->C 1x1, stats out
    CONV_HEAD   // GN+SiLU prologue, 32->3 3x3, out = clamp(round(input + y)) u8
};

struct ConvArgs {
    const void* in0;             // bf16 NHWC activations (CONV_STEM: u8 RGB)
    const void* in1;             // second source of the concat (CONV_FUSE), else null
    int cin0, cin1;              // channels per pixel of in0 / in1
    int kc_split;                // number of 32-channel K-chunks read from in0 (rest from in1)
    int nkc;                     // total K-chunks
    const unsigned short* w;     // weights pre-arranged [nblock][kchunk][kk][NT][8] bf16
    const unsigned short* w1;    // conv_up.hip fused with `fuse`: skip weights [nblock32][ks][h][32][8] bf16, else null
    const float* bias;           // [cout]
    const float2* ab;            // [nimg][cin0] GroupNorm+FiLM coefficients (PRO_GN)
    // GroupNorm finalize folded into this kernel's prologue (gn_fold.hpp): when gn_stats != null every workgroup first reduces
    // the partials of the image(s) it reads and writes `ab_w` (== ab); null: `ab` was filled by gn_finalize_kernel
    const float* gn_stats;       // [nimg][gn_parts][8][2] partials of the INPUT tensor
    int gn_parts, gn_hw;         // partials per image; pixels per image of the input tensor
    const float* gn_gamma;       // [cin0]
    const float* gn_beta;
    const float* gn_film;        // [nimg][gn_film_stride] FiLM (scale | shift) rows, or null
    int gn_film_stride, gn_film_off;
    float2* ab_w;
    const unsigned short* resid; // [nimg][Hout][Wout][cout] bf16 (CONV_RB2)
    unsigned short* out;         // [nimg][Hout][Wout][cout] bf16
    const unsigned char* u8_in;  // CONV_HEAD: original image
    unsigned char* u8_out;       // CONV_HEAD: restored image
    float* stats;                // [nimg][tiles][8][2] partial (sum, sumsq) per GroupNorm group
    int Hin, Win, Hout, Wout, cout;
    int tiles_x, tiles_y, nimg, nblocks;
    int group_size;              // cout / 8
    int prio_young;              // conv_rb: raise the issue priority of waves 4-7
    // Row-strip execution (cfg 4, engine.cpp "strips"): the input tensors of a strip carry `in_row_off` halo rows above local
    // row 0 (and as many below); a (virtual) input row iy of the strip is readable iff iy_lo <= iy < iy_lo + iy_span -- the
    // halo row of a neighbouring strip is data, a row outside the image is zero padding.  Whole images: in_rows = Hin,
    // in_row_off = 0, iy_lo = 0, iy_span = virtual input height.
    int in_rows;                 // allocated rows per image of in0 / in1
    int in_row_off;              // buffer row of local row 0
    int iy_lo, iy_span;
    int fp8;                     // conv_w4: OCP e4m3 operands (a.w = fp8 slabs, a.bias = bias / oscale)
    const float* oscale;         // fp8: [cout] accumulator -> output scale (weight scale of the channel / activation scale)
    int w4_nt;                   // conv_w4 cout block: 128 (default, 0) or 64 (launches whose 128-cout items would leave CUs idle; a.w = the 64-cout slabs)
    const void* zeros;           // conv_upq.hip: >= 16 bytes of zeros (what a DMA lane outside the image fetches)
    unsigned long long* stamps;  // diagnostic builds only (IRE_RB_ABLATE, DBG bit 16): s_memtime stamps, else null
};

int conv_tile_h(ConvKind kind);       // output rows per workgroup tile (columns: 32)
int conv_nt(ConvKind kind, int cout); // output channels per workgroup
int conv_nsteps(ConvKind kind);       // MFMA k-steps per K-chunk (weight slab = nsteps*2*NT*16 B)
void conv_launch(ConvKind kind, const ConvArgs& a, hipStream_t stream);

// Persistent software-pipelined variant for CONV_RB1 / CONV_RB2 (conv_rb.hip): 16x32 tiles
// (a.tiles_y must be ceil(Hout/16)), same ConvArgs; the weight slab rows are in permuted cout order (row n of a 32-row tile =
// cout n with bits 2 and 3 swapped: engine.cpp::make_conv), so the epilogue stores straight from the accumulators.
constexpr int kRbTileH = 16;
// fused_act: apply y = silu(x*A+B) while staging (a.ab); otherwise the input is already activated.
void conv_rb_launch(bool resid, bool fused_act, const ConvArgs& a, hipStream_t stream);
// One-wave-per-SIMD variant for C >= 128 ResBlock convs on a pre-activated input (conv_w4.hip):
// a.nkc = Cin/16, a.nblocks = cout/128, a.w = slabs [nblock][kc16][tap*2 + c8][128][8], 16x32 tiles.
void conv_w4_launch(bool resid, const ConvArgs& a, hipStream_t stream);
// IRE_PRECISION_FP8: C >= 128 ResBlock convs on the block-scaled fp8 MFMA, K = 64 per instruction (conv_f8.hip).
// a.nkc = Cin/32, a.nblocks = cout/128, a.w = e4m3 slabs [nblock][kc32][tap][half][128][16], a.bias = bias / oscale, a.oscale.
void conv_f8_launch(bool resid, const ConvArgs& a, hipStream_t stream);
// CONV_DOWN on the pipelined schedule (conv_down.hip): the stride-2 conv as a unit-stride conv over the four pixel phases.
// a.Hin/Win = full-res source, a.Hout/Wout = half; a.nkc = Cin/32, a.nblocks = cout/64, tiles of 16x32 OUTPUT pixels.
void conv_stem_launch(const ConvArgs& a, hipStream_t stream);            // conv_stem.hip: u8 RGB -> 32 channels
void conv_down_launch(const ConvArgs& a, hipStream_t stream);
// CONV_HEAD through the pipelined kernel (conv_rb.hip, HEAD variant): a.w = permuted-row slab (32 rows, 3 used), 16x32 tiles.
void conv_head_launch(const ConvArgs& a, hipStream_t stream);
// C = 32 .. 256 ResBlock convs (fused activation) and the head as a producer / consumer workgroup, three waves per SIMD
// (conv_pc.hip): a.w = permuted-row slabs [n-block of min(C, 64) couts][k-chunk][kk][rows][8] (d_wp), a.nkc = C/32, a.nblocks =
// max(1, C/64), 16x32 tiles, partials layout of conv_rb_launch.  conv_pc_fits: every workgroup's images fit its coefficient table.
void conv_pc_launch(bool resid, bool head, const ConvArgs& a, hipStream_t stream);
bool conv_pc_fits(int C, int tiles_per_img, int nimg);
// C = 128 / 256 ResBlock convs (fused activation) as a producer / consumer workgroup with 128-cout items (conv_pk.hip): a.w =
// conv_w4's slabs (d_w4), a.nkc = C / 16, a.nblocks = C / 128, 16x32 tiles; results bit-identical to conv_w4's 8-wave fused form.
// conv_pk_fits: every workgroup's images fit its coefficient table (else conv_w4 takes the launch).
void conv_pk_launch(bool resid, const ConvArgs& a, hipStream_t stream);
bool conv_pk_fits(int C, int tiles_per_img, int nimg);
// CONV_UP as a sub-pixel convolution on the low-resolution grid (conv_up.hip): 4 output parities x 2x2 pre-summed taps.
// a.Hin/Win = low-res source, a.Hout/Wout = 2x; a.nkc = Cin/32 (even), a.nblocks = cout/32, tiles of 16x32 LOW-res pixels.
void conv_up_subpixel_launch(const ConvArgs& a, hipStream_t stream);
// The fused `up` + `fuse` with cout = 128 as parity-major items (conv_upq.hip): a.w = d_wuq, a.w1 = d_wsq, a.nkc = Cin / 32, a.nblocks = 4
// (parities), 16 x 32 LOW-res tiles, a.zeros, partials [img][tile * 4 + parity][8][2] (four rows per low-res tile).
void conv_upq_launch(const ConvArgs& a, hipStream_t stream);
// CONV_DOWN with cout = 128 / 256 as all-DMA 128-cout items (conv_dnq.hip): a.w = d_wdq, a.nkc = Cin / 32, a.nblocks = cout / 128, a.zeros;
// tiles, partials layout and strip arguments as conv_down_launch.
void conv_dnq_launch(const ConvArgs& a, hipStream_t stream);
// CONV_UP through the same pipelined kernel (Hin/Win = low-res source, Hout/Wout = 2x; a.stats = nullptr).
void conv_up_launch(const ConvArgs& a, hipStream_t stream);

}  // namespace ire
