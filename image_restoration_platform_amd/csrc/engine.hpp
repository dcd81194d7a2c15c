// engine.hpp -- the per-GPU engine object behind the C ABI (include/ire.h).
//
// One engine = one process = one MI355X.  It owns: the classifier tables, the RestoreNet-v0
// weights re-laid-out for conv_mfma.hip, per-"lane" activation workspaces (a lane = one HIP
// stream restoring a contiguous slice of the batch, so several images are in flight and the
// per-image working set of the full-resolution levels stays inside the 256 MiB Infinity Cache),
// staging buffers for the host-pointer entry points, and an event-based per-kernel profiler.
#pragma once
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "classifier.hpp"
#include "common.hpp"
#include "conv_mfma.hpp"

namespace ire {

enum Family { FAM_CLASSIFIER = 0, FAM_CONV3 = 1, FAM_CONV1 = 2, FAM_STEM = 3, FAM_HEAD = 4, FAM_GN = 5,
              FAM_FUSION = 6, FAM_COUNT = 7 };

struct ConvW {
    ConvKind kind = CONV_RB1;
    int cin = 0, cout = 0;       // logical channel counts (FLOP accounting)
    int cin0 = 0, cin1 = 0;      // channels per pixel of the two sources
    int nt = 0, nblocks = 0, nkc = 0, kc_split = 0;
    unsigned short* d_w = nullptr;
    unsigned short* d_wp = nullptr;   // slabs with permuted cout rows for conv_rb.hip's direct epilogue
    unsigned short* d_w4 = nullptr;  // second arrangement for conv_w4.hip (C >= 128 ResBlock convs): 16-channel stages, 128-cout blocks
    unsigned short* d_w4h = nullptr; // the same in 64-cout blocks: launches whose 128-cout items number fewer than the CUs (small batches, 512^2 at level 3)
    unsigned short* d_wstem = nullptr;  // CONV_STEM as MFMA A fragments (conv_stem.hip): [ky 3][h 2][32 permuted rows][8], k = 16 ky + 4 kx + c (kx = 3, c = 3: zero)
    unsigned short* d_wd = nullptr;  // CONV_DOWN by pixel phase (conv_down.hip): [nblock64][kc32][phase: 1+2+2+4 taps][tap*4 + c8][64][8]
    unsigned short* d_wu = nullptr;  // CONV_UP as a sub-pixel conv (conv_up.hip): [nblock32][kc32][parity][kk][32][8], taps pre-summed per parity
    // CONV_UP composed with the level's 1x1 `fuse` (engine.cpp::make_up_fused; conv_up.hip fused form)
    unsigned short* d_wuf = nullptr; // sub-pixel slabs of (Wf_up . Wup), layout of d_wu
    unsigned short* d_wdq = nullptr; // conv_dnq.hip (stride-2 convs with cout % 128 == 0): d_wd's taps as 128-cout slabs
    unsigned short* d_wuq = nullptr; // conv_upq.hip (cout = 128): the same composed weights as [parity][kc32][tap4][c8][128 permuted rows][8]
    unsigned short* d_wsq = nullptr; // conv_upq.hip: skip half of the fuse weights as [ks32][c8][128 permuted rows][8]
    unsigned short* d_wsk = nullptr; // skip half of the fuse weights as MFMA A fragments: [nblock32][ks = C/16][h][32 permuted rows][8]
    float* d_bias_uf = nullptr;      // Wf_up . b_up + b_f
    unsigned char* d_w8x = nullptr;  // IRE_PRECISION_FP8, K = 64 form (conv_f8.hip): [nblock128][kc32][tap][half][128][16] e4m3
    unsigned char* d_w8 = nullptr;   // IRE_PRECISION_FP8: the conv_w4 slabs as OCP e4m3, one scale per output channel
    float* d_oscale = nullptr;       // [cout] weight scale / activation scale (accumulator -> output)
    float* d_bias8 = nullptr;        // [cout] bias / oscale (the accumulators start at it)
    float* d_bias = nullptr;
};
struct GNW {
    int C = 0, level = 0;
    float* d_gamma = nullptr;
    float* d_beta = nullptr;
};
struct RBW {
    GNW gn1, gn2;
    ConvW conv1, conv2;
};
struct Net {
    bool loaded = false;
    ConvW stem, head;
    GNW head_gn;
    RBW enc[4][2], mid[2], dec[3][2];
    ConvW down[3], up[3], fuse[3];
    float* d_film_w = nullptr;
    float* d_film_b = nullptr;
    std::vector<void*> allocs;
};

struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    unsigned short* act[4][4] = {};
    unsigned short* skip[3] = {};
    float* stats = nullptr;
    float* stats2 = nullptr;          // second partials array: a conv that finalizes its input's GroupNorm itself (gn_fold.hpp) reads one and writes the other
    float2* ab = nullptr;
};

// ---- the layer schedule as a program (engine.cpp::build_program): one list of ops, executed op by op either over a whole
// batch (restore_device) or over the row strips of one image with a halo exchange after every op whose output feeds a
// 3x3 convolution and a gather of the GroupNorm partials before every OP_GN (cfg 4: engine.cpp "strips") ----
constexpr int BUF_NONE = -1;
inline int buf_id(int level, int slot) { return level * 8 + slot; }   // slot 0..3 = act[level][slot], 4 = skip[level]
struct Op {
    enum Kind { GN, CONV } kind = CONV;
    const GNW* gn = nullptr;          // GN: finalize the partials of the tensor produced last -> (A, B) per (image, channel)
    const ConvW* cw = nullptr;        // CONV
    int in0 = BUF_NONE, in1 = BUF_NONE, resid = BUF_NONE, out = BUF_NONE;
    int lin = 0, lout = 0;            // levels of in0 / out
    bool use_ab = false;              // CONV: GroupNorm+FiLM+SiLU applied while staging in0
    bool halo_out = false;            // the output is read by a 3x3 convolution: strips exchange its boundary rows
    bool stats_out = false;           // the conv writes GroupNorm partials of its output
    std::string name;                 // debug-capture name ("" = none)
};
// state shared by everything one network run touches on one stream (a lane, or all strips of a tiled image)
struct Run {
    hipStream_t stream = nullptr;
    float* stats = nullptr;           // [image][tile][8 groups][2] partials of the tensor produced last
    float2* ab = nullptr;             // [image][C] coefficients of the GroupNorm finalized last
    const float* film = nullptr;
    int stat_parts = 0;               // partials per image the last stats-producing conv wrote
    float* stats_alt = nullptr;       // where the NEXT stats-producing conv writes (ping-pong with `stats`); null: in place (strips)
    const GNW* gn_pending = nullptr;  // a GroupNorm whose finalize was deferred to its consumer (exec_op GN -> exec_conv)
    const float* gn_stats = nullptr;  // its partials and their count per image
    int gn_parts = 0;
};
// where one executor instance's rows live: a whole batch (halo = 0) or one row strip of one image (halo = 1)
struct Geo {
    int nimg = 1, h = 0, w = 0;       // rows / columns of THIS piece at level 0
    int halo = 0;                     // halo rows above and below in every activation buffer and in img_in
    bool has_up = false, has_down = false;
    int y0 = 0, H = 0;                // first global row of the piece and global image height (level 0)
    unsigned short* buf[4][6] = {};   // buffer starts (halo row included)
    const uint8_t* img_in = nullptr;  // u8 image rows of the piece, halo rows included
    uint8_t* img_out = nullptr;       // u8 output rows of the piece (no halo)
};

struct ProfRec {
    int fam;
    bool own_e0 = true;   // false: e0 is the previous record's e1 (back-to-back profiled launches share the event)
    hipEvent_t e0, e1;
    double flops, bytes;
    double flops_exec = 0;   // flops the kernel really issues (the sub-pixel `up` form runs 4 of 9 taps, its composed skip term C of 2C channels)
    int group = -1;          // index into Engine::prof_groups_ (layer group "L2.rb1", "up0", ...) or -1
};
struct ProfGroup {           // per layer group: what bench.py's roofline.per_level reports
    std::string key, kernel;
    int level = 0, cin = 0, cout = 0;
    double ms = 0, flops = 0, flops_exec = 0, bytes = 0;
    int64_t n = 0;
};

class Engine {
    friend class StripSession;
public:
    explicit Engine(const ire_config& cfg);
    ~Engine();

    void load_weights(const void* blob, size_t bytes);
    void load_weights_file(const char* path);

    // host-pointer paths (synchronous)
    void classify_host(const uint8_t* rgb, int n, int h, int w, int row_stride, const uint8_t* is_jpeg,
                       double* scores, int32_t* label);
    void restore_host(const uint8_t* rgb, int n, int h, int w, const double* scores, const uint8_t* is_jpeg,
                      uint8_t* out, ire_timings* t);
    // device-pointer paths (asynchronous on stream)
    void classify_device(const uint8_t* d_rgb, int n, int h, int w, const uint8_t* d_is_jpeg, double* d_scores,
                         int32_t* d_label, hipStream_t stream);
    void restore_device(const uint8_t* d_rgb, int n, int h, int w, const double* d_scores,
                        const uint8_t* d_is_jpeg, uint8_t* d_out, hipStream_t stream);

    // fusion (fusion.hip)
    void fuse_launch(const uint8_t* d_views, int nsets, int k, int h, int w, const unsigned* host_wluts, uint8_t* d_out,
                     int32_t* d_shifts, hipStream_t s);
    double noise_of_view0(const uint8_t* d_views, int h, int w, hipStream_t s);
    void fuse_host_impl(const uint8_t* rgb_views, int k, int h, int w, double noise_score, uint8_t* out_rgb,
                        int32_t* shifts_out, ire_timings* t);

    // preprocess step in front of the path (preprocess.hip): EXIF orientation + fit-inside Lanczos-3
    void preprocess_device(const uint8_t* d_rgb, int h, int w, int orientation, int max_dim, uint8_t* d_out, int out_h, int out_w,
                           hipStream_t s);
    void preprocess_host(const uint8_t* rgb, int h, int w, int orientation, int max_dim, uint8_t* out, int out_h, int out_w);
    // encode.hip: n images [h][w][3] -> n x png_base64_chars(h, w) characters, `stride` bytes apart (device buffers, asynchronous on s)
    void encode_png_base64_device(const uint8_t* d_rgb, int n, int h, int w, uint8_t* d_chars, size_t stride, hipStream_t s);
    void encode_png_base64_host(const uint8_t* rgb, int n, int h, int w, uint8_t* chars, size_t stride);
    uint32_t flags() const { return flags_; }

    void debug_sums(int n, uint64_t* out);
    void debug_capture(bool on) { capture_ = on; captured_.clear(); }
    bool debug_activation(const std::string& name, float* out, size_t* count);

    void profile_enable(int mode);   // low byte: 0 off, 1 every kernel family, 2 the 3x3 conv family only; bits 8..: N = time every N-th network pass (mode 2)
    void profile_reset();
    void profile_query(int fam, double* ms, int64_t* launches, double* flops, double* bytes);
    std::string profile_report();    // JSON array, one object per layer group (ire_profile_report)

    int max_batch() const { return max_batch_; }
    // Cross-stream serialisation of the shared GPU scratch (activation workspaces, d_scores_, d_film_, fusion and
    // preprocess scratch): every ABI call brackets its enqueue with enter/leave under the engine mutex.  enter makes the
    // caller's stream wait for the completion event of the previous call (a no-op on the same stream), leave records
    // the new one -- so two threads on two streams interleave whole calls on the GPU, never kernels of two calls.
    void enter(hipStream_t s);
    void leave(hipStream_t s);
    // device bytes one more image of this shape costs (activation workspace + staging), and how many fit right now
    size_t bytes_per_image(int h, int w) const;
    int capacity_for(int h, int w) const;
    // cfg 4: the whole image on this GPU as nstrips "virtual ranks" (strips.cpp); d_scores null => classify inside
    void restore_tiled_device(const uint8_t* d_rgb, int h, int w, int nstrips, const double* d_scores, const uint8_t* d_is_jpeg,
                              uint8_t* d_out, hipStream_t stream);
    void get_stats(ire_engine_stats* out);                            // counters + the images/sec gauge (queue_depth is the ABI layer's)
    // batcher form of restore_device: rows of host_scores (pinned, n*7) flagged in has_scores are used as given, the rest are
    // classified inside (one scan over the batch, skipped when every job brought its scores)
    void restore_device_mixed(const uint8_t* d_rgb, int n, int h, int w, const double* host_scores, const uint8_t* has_scores,
                              const uint8_t* d_is_jpeg, uint8_t* d_out, hipStream_t stream);
    hipStream_t main_stream() const { return main_stream_; }          // the stream of the host entry points and of the batcher's compute
    const double* scores_device() const { return d_scores_; }       // [last n][7], valid after a classify on main_stream()
    std::mutex& mutex() { return mu_; }

private:
    void check_shape(int n, int h, int w, bool for_restore) const;
    void ensure_io(int n, int h, int w);
    void ensure_workspace(int n, int h, int w);
    void free_workspace();
    void build_program();
    void run_network(Lane& L, int nimg, int h, int w, const uint8_t* d_in, uint8_t* d_out, const float* d_film);
    void exec_op(Run& R, const Op& op, const Geo& g);
    void flush_gn(Run& R, const Geo& g);       // launch the deferred finalize as its own kernel (consumers without the folded prologue)
    void exec_conv(Run& R, const Op& op, const Geo& g);
    static Geo geo_of_lane(const Lane& L, int nimg, int h, int w, const uint8_t* d_in, uint8_t* d_out);
    void prof_begin(int fam, hipStream_t s, double flops, double bytes);
    void prof_tag(const std::string& key, const char* kernel, int level, int cin, int cout, double flops_exec);   // after prof_begin: the open record's layer group
    void prof_end(hipStream_t s);
    void capture(const char* name, const unsigned short* d, size_t count, hipStream_t s);
    ConvW make_conv(ConvKind kind, const std::string& wname, const std::string& bname, int cin, int cout);
    void make_up_fused(ConvW& up, const std::string& level);
    GNW make_gn(const std::string& prefix, int C, int level);
    RBW make_rb(const std::string& prefix, int C, int level);
    void* dalloc(size_t bytes);

    int device_ = 0;
    int max_batch_ = 8;
    int num_lanes_ = 1;
    uint32_t flags_ = 0;
    int precision_ = IRE_PRECISION_BF16;
    int w4_split_ = 1;            // IRE_W4_SPLIT=0: never use the 64-cout items
    int use_dnq_ = 1;             // the stride-2 convs with cout >= 128 as all-DMA 128-cout items on conv_dnq.hip (IRE_DNQ=0: conv_down.hip)
    int use_upq_ = 1;             // the level-2 `up` + `fuse` (cout = 128) as parity-major 128-cout items on conv_upq.hip (IRE_UPQ=0: conv_up.hip)
    int use_pk_ = 1;              // C >= 128 ResBlock convs (128-cout items, fused activation) on conv_pk.hip's producer / consumer workgroups: 1 = the convs without a residual, 2 = all (IRE_PK=0: conv_w4.hip)
    int use_w4_ = 1;              // C >= 128 ResBlock convs on conv_w4.hip (IRE_W4=0: conv_rb.hip)
    int fp8_mx_ = 1;              // fp8: the block-scaled K = 64 MFMA (conv_f8.hip); IRE_FP8_MX=0: the same-rate 32x32x16 fp8 form in conv_w4.hip
    int down_rb_ = 1;             // stride-2 `down` convs on conv_down.hip's pipelined phase kernel (IRE_DOWN_RB=0: the v1 kernel)
    int head_rb_ = 1;             // the 32 -> 3 head conv on conv_rb.hip's pipelined kernel (IRE_HEAD_RB=0: the v1 kernel)
    int pc_split_ = 3;            // producer / consumer workgroups (conv_pc.hip): bit 0 = C = 32 ResBlock convs + head, bit 1 = C = 64; IRE_PC=0: conv_rb.hip
    int gn_fold_ = 1;             // GroupNorm finalize inside the consuming conv's prologue (gn_fold.hpp); IRE_GN_FOLD=0: 33 gn_finalize launches per step
    int stem_rb_ = 1;             // the stem on its own kernel (conv_stem.hip); IRE_STEM_RB=0: the v1 template
    int up_fuse_ = 1;             // `up` + 1x1 `fuse` as ONE composed convolution with the skip term in conv_up.hip's epilogue (IRE_UP_FUSE=0: two kernels)
    int up_subpixel_ = 1;         // `up` convs as sub-pixel convolutions on the low-res grid (IRE_UP_SUBPIX=0: nearest x2 + 3x3 on conv_rb.hip)
    int up_rb_min_c_ = 32;        // `up` convs with cout >= this run on conv_rb.hip (IRE_UP_RB_MINC), the rest on the v1 kernel
    int prio_young_ = 0;          // static s_setprio for waves 4-7 of conv_rb (A/B'd: it only swaps which half waits)
    int rb_tile_h_ = kRbTileH;  // 16: persistent pipelined conv_rb.hip; 8: conv_mfma.hip (IRE_CONV_V1=1)
    std::mutex mu_;
    hipStream_t main_stream_ = nullptr;
    hipEvent_t ev_[4] = {};
    hipEvent_t fork_ev_ = nullptr;
    hipEvent_t busy_ev_ = nullptr;   // completion of the last enqueued call (enter/leave)
    bool busy_recorded_ = false;
    int64_t batches_run_ = 0, images_restored_ = 0;
    int last_batch_ = 0;
    std::deque<std::pair<double, int>> recent_;      // (seconds since init, images) of the last 10 s of restore calls
    double t0_ = 0.0;

    // classifier
    ClassifierTables tables_{};
    std::vector<void*> table_allocs_;

    // io / classifier buffers (sized by ensure_io)
    size_t io_cap_imgs_ = 0, io_cap_px_ = 0;
    uint8_t* d_in_ = nullptr;
    uint8_t* d_out_ = nullptr;
    uint8_t* d_jpeg_ = nullptr;
    unsigned long long* d_sums_ = nullptr;
    double* d_scores_ = nullptr;
    int32_t* d_label_ = nullptr;
    float* d_cond_ = nullptr;
    float* d_film_ = nullptr;
    int last_n_ = 0;
    // fusion scratch
    size_t fuse_cap_px_ = 0;
    int fuse_cap_sets_ = 0;
    unsigned* d_fwlut_ = nullptr;
    uint8_t* d_fL_ = nullptr;
    uint8_t* d_fQ_ = nullptr;
    unsigned* d_fsad_ = nullptr;
    int* d_fmisc_ = nullptr;
    void* d_zero_ = nullptr;           // 256 bytes of zeros (conv_upq.hip's zero padding source)
    // preprocess scratch (tap tables, intermediate of the horizontal pass, host-path staging)
    size_t pp_tab_cap_ = 0, pp_mid_cap_ = 0, pp_in_cap_ = 0, pp_out_cap_ = 0;
    int32_t* d_pp_tab_ = nullptr;
    uint8_t* d_pp_mid_ = nullptr;
    uint8_t* d_pp_in_ = nullptr;
    uint8_t* d_pp_out_ = nullptr;
    uint8_t* d_enc_scratch_ = nullptr;    // encode.hip: per image: the PNG file + checksum state
    uint8_t* d_enc_io_ = nullptr;         // host entry: pixels in | characters out
    size_t enc_scratch_cap_ = 0, enc_io_cap_ = 0;

    // network
    Net net_;
    std::vector<Op> program_;
    class StripSession* tiled_ = nullptr;      // cached session of restore_tiled_device (one shape at a time)
    int tiled_h_ = 0, tiled_w_ = 0, tiled_n_ = 0;       // the layer schedule (build_program; rebuilt by load_weights)
    std::map<std::string, std::pair<std::vector<int>, std::vector<float>>> host_w_;
    std::vector<Lane> lanes_;
    int ws_imgs_per_lane_ = 0, ws_imgs_cap_ = 0, ws_h_ = 0, ws_w_ = 0;
    size_t ws_bytes_ = 0;
    std::vector<void*> ws_allocs_;

    // debug / profile
    // diagnostic stamps (IRE_RB_STAMPS, ablation builds)
    unsigned long long* stamps_dev_ = nullptr;
    int stamps_cout_ = 0;
    bool stamps_resid_ = false, stamps_taken_ = false;
    std::string stamps_tl_;
    bool capture_ = false;
    std::map<std::string, std::vector<float>> captured_;
    int prof_on_ = 0;
    int prof_every_ = 1, prof_pass_ = 0;   // mode 2: time every prof_every_-th pass of the network
    bool prof_skip_ = false;
    bool prof_open_ = false;
    bool prof_chain_ = false;          // the stream's last operation is prof_.back()'s end event (prof_chain_stream_): the next record starts there
    hipStream_t prof_chain_stream_ = nullptr;
    bool prof_chainable_ = false;      // true while run_network walks the op list (the only place records may share events)
    std::vector<ProfRec> prof_;
    std::vector<hipEvent_t> ev_pool_;
    double prof_ms_[FAM_COUNT] = {}, prof_flops_[FAM_COUNT] = {}, prof_bytes_[FAM_COUNT] = {};
    int64_t prof_n_[FAM_COUNT] = {};
    double prof_flops_exec_[FAM_COUNT] = {};
    std::vector<ProfGroup> prof_groups_;
    void prof_collect();
};

void preprocess_plan(int width, int height, int orientation, int max_dim, int* out_w, int* out_h, int* resized);

}  // namespace ire
