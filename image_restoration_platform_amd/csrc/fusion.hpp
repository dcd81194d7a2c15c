// fusion.hpp -- <=3-view alignment + blend (Fusion-v0, build-defined; fusion.hip).
#pragma once
#include "common.hpp"

namespace ire {
class Engine;
void fuse_host(Engine& E, const uint8_t* rgb_views, int k, int h, int w, double noise_score, uint8_t* out_rgb,
               int32_t* shifts_out, ire_timings* t);
void fuse_device(Engine& E, const uint8_t* d_rgb_views, int k, int h, int w, double noise_score, uint8_t* d_out_rgb,
                 int32_t* d_shifts, hipStream_t stream);
constexpr int kFuseMaxSets = 16;
void fuse_batch_device(Engine& E, const uint8_t* d_rgb_views, int nsets, int k, int h, int w, const double* noise_scores,
                       uint8_t* d_out_rgb, int32_t* d_shifts, hipStream_t stream);
}  // namespace ire
