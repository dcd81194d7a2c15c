// strips.hpp -- row-strip ("tiled") restoration sessions: cfg 4 of BASELINE.json (strips.cpp).
#pragma once
#include <vector>

#include "engine.hpp"

namespace ire {

class StripSession {
public:
    // strips [first_strip, first_strip + nlocal) of nstrips_total equal row strips of an H x W image live in this session (all
    // of them on one GPU = "virtual ranks"; one per rank in the 8-GPU layout).  d_stats_external: the GroupNorm-partials
    // array to use (stats_floats(H, W) floats of device memory the caller can all-gather in place), or null = own.
    StripSession(Engine& eng, int H, int W, int nstrips_total, int first_strip, int nlocal, float* d_stats_external);
    ~StripSession();
    static size_t stats_floats(int H, int W);
    int num_ops() const;
    int rows_per_strip() const { return hr_; }
    void set_input(const uint8_t* d_rows_with_halo, const double* d_scores, hipStream_t s);
    void run_op(int k, hipStream_t s, ire_strip_xchg* info);
    void pack_halo(int k, uint8_t* d_send_up, uint8_t* d_send_down, hipStream_t s);
    void unpack_halo(int k, const uint8_t* d_recv_up, const uint8_t* d_recv_down, hipStream_t s);
    void get_output(uint8_t* d_out_rows, hipStream_t s);
    void run_all(const uint8_t* d_rgb, const double* d_scores, uint8_t* d_out, hipStream_t s);
    Engine& engine() { return E; }

private:
    size_t halo_row_bytes(int k) const;
    Engine& E;
    int H_, W_, total_, first_, nlocal_, hr_ = 0;
    std::vector<Geo> strips_;
    std::vector<void*> allocs_;
    float* stats_ = nullptr;
    float2* ab_ = nullptr;
    float* d_cond_ = nullptr;
    float* d_film_ = nullptr;
    double* d_scores_ = nullptr;
    Run run_;
};

}  // namespace ire
