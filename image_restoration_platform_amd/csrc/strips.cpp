// strips.cpp -- cfg 4 of BASELINE.json: one large image restored as ROW STRIPS (the reference's size
// cap is 2048 px: server-node/src/middleware/imagePreprocess.js:4).
//
// A strip is what one GPU of the node owns in the 8-GPU configuration; on one GPU the strips are "virtual ranks" that run
// the same code with in-device copies where the ranks would use xGMI.  The layer schedule is the engine's op program
// (engine.cpp::build_program), executed op by op:
//   * every activation tensor of a strip carries ONE halo row above and below; after an op whose output feeds a 3x3
//     convolution the strip's first / last real row is copied into the neighbour's lower / upper halo row (per level: the
//     receptive field of the whole network is far too large for one overlap);
//   * GroupNorm statistics are global: every producing conv writes its per-tile partials at the tile's GLOBAL index into one
//     partials array (ranks: each fills its slice, then an all-gather), and the finalize runs on the complete array.
// Because tiles, per-tile fp32 partial sums and the double-precision finalize order are exactly those of the untiled run,
// the tiled result is bit-identical to the untiled one (tests/test_tiled_gpu.py).
#include "strips.hpp"

#include <algorithm>
#include <cstring>

#include "gn.hpp"

namespace ire {

namespace {
const int kW[4] = {32, 64, 128, 256};
const int kFilmDimS = 960;
}  // namespace

size_t StripSession::stats_floats(int H, int W) { return (size_t)ceil_div(H, 4) * ceil_div(W, 32) * 16; }

StripSession::StripSession(Engine& eng, int H, int W, int nstrips_total, int first_strip, int nlocal, float* d_stats_external)
    : E(eng), H_(H), W_(W), total_(nstrips_total), first_(first_strip), nlocal_(nlocal) {
    if (!E.net_.loaded) fail(IRE_ERR_UNAVAILABLE, "service unavailable: RestoreNet weights are not loaded");
    if (E.rb_tile_h_ != kRbTileH)
        fail(IRE_ERR_INVALID_INPUT, "invalid configuration for strip mode: the A/B schedule switches must be at their defaults");
    if (H <= 0 || W <= 0 || H > 8192 || W > 8192 || W % 8) fail(IRE_ERR_INVALID_INPUT, "invalid image size for tiled restore");
    if (nstrips_total < 1 || nstrips_total > 64 || H % nstrips_total) fail(IRE_ERR_INVALID_INPUT, "invalid strip count: it must divide the image height");
    hr_ = H / nstrips_total;
    // a strip must start on a tile boundary of every producing kernel at every level (16 rows at 1/8 scale => 128 rows)
    if (nstrips_total > 1 && hr_ % 128) fail(IRE_ERR_INVALID_INPUT, "invalid strip height: rows per strip must be a multiple of 128");
    if (nstrips_total == 1 && (H % 8 || H < 16)) fail(IRE_ERR_INVALID_INPUT, "invalid image size for restore: height and width must be multiples of 8, >= 16");
    if (first_strip < 0 || nlocal < 1 || first_strip + nlocal > nstrips_total) fail(IRE_ERR_INVALID_INPUT, "invalid strip range");
    IRE_HIP(hipSetDevice(E.device_));
    auto alloc = [&](size_t bytes) { void* p = nullptr; IRE_HIP(hipMalloc(&p, bytes ? bytes : 16)); allocs_.push_back(p); return p; };
    try {
        strips_.resize(nlocal);
        for (int s = 0; s < nlocal; ++s) {
            Geo& g = strips_[s];
            const int gs = first_strip + s;
            g.nimg = 1; g.h = hr_; g.w = W; g.halo = 1; g.H = H; g.y0 = gs * hr_;
            g.has_up = gs > 0; g.has_down = gs + 1 < nstrips_total;
            for (int l = 0; l < 4; ++l) {
                const size_t bytes = (size_t)((hr_ >> l) + 2) * (W >> l) * kW[l] * 2;
                for (int b = 0; b < 5; ++b) {
                    if (b == 4 && l == 3) continue;
                    g.buf[l][b] = (unsigned short*)alloc(bytes);
                }
            }
            uint8_t* img = (uint8_t*)alloc((size_t)(hr_ + 2) * W * 3);
            g.img_in = img;
            g.img_out = (uint8_t*)alloc((size_t)hr_ * W * 3);
        }
        if (d_stats_external) stats_ = d_stats_external;
        else stats_ = (float*)alloc(stats_floats(H, W) * 4);
        ab_ = (float2*)alloc(256 * sizeof(float2));
        d_cond_ = (float*)alloc(8 * 4);
        d_film_ = (float*)alloc(kFilmDimS * 4);
        d_scores_ = (double*)alloc(7 * 8);
    } catch (...) {
        for (void* p : allocs_) (void)hipFree(p);
        allocs_.clear();
        throw;
    }
}

StripSession::~StripSession() {
    (void)hipSetDevice(E.device_);
    (void)hipDeviceSynchronize();
    for (void* p : allocs_) (void)hipFree(p);
}

int StripSession::num_ops() const { return (int)E.program_.size(); }

// rows: (nlocal*hr + 2) x W x 3, i.e. the local rows with one row above and one below (the rows outside the image are never
// read); scores: 7 doubles on the device (the rank that received the job classified the whole image)
void StripSession::set_input(const uint8_t* d_rows_with_halo, const double* d_scores, hipStream_t s) {
    if (!d_rows_with_halo || !d_scores) fail(IRE_ERR_INVALID_INPUT, "invalid input: null pointer");
    const size_t row = (size_t)W_ * 3;
    for (int i = 0; i < nlocal_; ++i)
        IRE_HIP(hipMemcpyAsync(const_cast<uint8_t*>(strips_[i].img_in), d_rows_with_halo + (size_t)i * hr_ * row, (size_t)(hr_ + 2) * row,
                               hipMemcpyDeviceToDevice, s));
    IRE_HIP(hipMemcpyAsync(d_scores_, d_scores, 7 * 8, hipMemcpyDeviceToDevice, s));
    scores_to_cond_launch(d_scores_, 1, d_cond_, s);
    film_launch(d_cond_, 1, E.net_.d_film_w, E.net_.d_film_b, kFilmDimS, d_film_, s);
    run_ = Run{};
    run_.stats = stats_; run_.ab = ab_; run_.film = d_film_;
}

size_t StripSession::halo_row_bytes(int k) const {
    const Op& op = E.program_[k];
    if (op.kind != Op::CONV || !op.halo_out || total_ == 1) return 0;
    return (size_t)(W_ >> op.lout) * kW[op.lout] * 2;
}

void StripSession::run_op(int k, hipStream_t s, ire_strip_xchg* info) {
    if (k < 0 || k >= num_ops()) fail(IRE_ERR_INVALID_INPUT, "invalid op index");
    const Op& op = E.program_[k];
    run_.stream = s;
    if (info) std::memset(info, 0, sizeof(*info));
    if (op.kind == Op::GN) { E.exec_op(run_, op, strips_[0]); return; }       // one finalize over the complete partials array
    for (int i = 0; i < nlocal_; ++i) E.exec_op(run_, op, strips_[i]);
    const size_t rb = halo_row_bytes(k);
    if (rb) {
        // neighbours inside this session: the copy a rank pair would do over xGMI
        const int l = op.lout, rows = hr_ >> l;
        for (int i = 0; i + 1 < nlocal_; ++i) {
            char* up = (char*)strips_[i].buf[l][op.out & 7];         // strip i  : rows 0 (halo) 1..rows (real) rows+1 (halo)
            char* dn = (char*)strips_[i + 1].buf[l][op.out & 7];
            IRE_HIP(hipMemcpyAsync(dn, up + (size_t)rows * rb, rb, hipMemcpyDeviceToDevice, s));                 // last real row of i -> upper halo of i+1
            IRE_HIP(hipMemcpyAsync(up + (size_t)(rows + 1) * rb, dn + rb, rb, hipMemcpyDeviceToDevice, s));      // first real row of i+1 -> lower halo of i
        }
        if (info) {
            info->halo_bytes = (int)rb;
            info->has_up = strips_.front().has_up ? 1 : 0;
            info->has_down = strips_.back().has_down ? 1 : 0;
        }
    }
    if (op.stats_out && info) {
        // the slice of the partials array this session's strips just wrote (whole tiles rows: contiguous)
        // the partials the op wrote: run_.stat_parts tiles per image (exec_conv), an equal slice of them per strip
        const size_t per_strip = (size_t)run_.stat_parts / total_ * 16 * 4;
        info->stats_offset_bytes = (long long)(per_strip * first_);
        info->stats_local_bytes = (long long)(per_strip * nlocal_);
        info->stats_total_bytes = (long long)(per_strip * total_);
    }
}

void StripSession::pack_halo(int k, uint8_t* d_send_up, uint8_t* d_send_down, hipStream_t s) {
    const size_t rb = halo_row_bytes(k);
    if (!rb) return;
    const Op& op = E.program_[k];
    const int l = op.lout, rows = hr_ >> l;
    if (d_send_up && strips_.front().has_up)
        IRE_HIP(hipMemcpyAsync(d_send_up, (char*)strips_.front().buf[l][op.out & 7] + rb, rb, hipMemcpyDeviceToDevice, s));
    if (d_send_down && strips_.back().has_down)
        IRE_HIP(hipMemcpyAsync(d_send_down, (char*)strips_.back().buf[l][op.out & 7] + (size_t)rows * rb, rb, hipMemcpyDeviceToDevice, s));
}

void StripSession::unpack_halo(int k, const uint8_t* d_recv_up, const uint8_t* d_recv_down, hipStream_t s) {
    const size_t rb = halo_row_bytes(k);
    if (!rb) return;
    const Op& op = E.program_[k];
    const int l = op.lout, rows = hr_ >> l;
    if (d_recv_up && strips_.front().has_up)
        IRE_HIP(hipMemcpyAsync((char*)strips_.front().buf[l][op.out & 7], d_recv_up, rb, hipMemcpyDeviceToDevice, s));
    if (d_recv_down && strips_.back().has_down)
        IRE_HIP(hipMemcpyAsync((char*)strips_.back().buf[l][op.out & 7] + (size_t)(rows + 1) * rb, d_recv_down, rb, hipMemcpyDeviceToDevice, s));
}

void StripSession::get_output(uint8_t* d_out_rows, hipStream_t s) {
    const size_t bytes = (size_t)hr_ * W_ * 3;
    for (int i = 0; i < nlocal_; ++i)
        IRE_HIP(hipMemcpyAsync(d_out_rows + (size_t)i * bytes, strips_[i].img_out, bytes, hipMemcpyDeviceToDevice, s));
}

// all strips on this GPU: the whole image in, the whole image out
void StripSession::run_all(const uint8_t* d_rgb, const double* d_scores, uint8_t* d_out, hipStream_t s) {
    if (first_ != 0 || nlocal_ != total_) fail(IRE_ERR_INVALID_INPUT, "invalid session: run_all needs every strip local");
    const size_t row = (size_t)W_ * 3;
    // strip i needs image rows [i*hr - 1, (i+1)*hr + 1): copy what exists (the rest is never read: iy_lo / iy_span)
    for (int i = 0; i < nlocal_; ++i) {
        const int r0 = std::max(0, i * hr_ - 1), r1 = std::min(H_, (i + 1) * hr_ + 1);
        uint8_t* dst = const_cast<uint8_t*>(strips_[i].img_in) + (size_t)(r0 - (i * hr_ - 1)) * row;
        IRE_HIP(hipMemcpyAsync(dst, d_rgb + (size_t)r0 * row, (size_t)(r1 - r0) * row, hipMemcpyDeviceToDevice, s));
    }
    IRE_HIP(hipMemcpyAsync(d_scores_, d_scores, 7 * 8, hipMemcpyDeviceToDevice, s));
    scores_to_cond_launch(d_scores_, 1, d_cond_, s);
    film_launch(d_cond_, 1, E.net_.d_film_w, E.net_.d_film_b, kFilmDimS, d_film_, s);
    run_ = Run{};
    run_.stats = stats_; run_.ab = ab_; run_.film = d_film_;
    for (int k = 0; k < num_ops(); ++k) run_op(k, s, nullptr);
    get_output(d_out, s);
}

}  // namespace ire
