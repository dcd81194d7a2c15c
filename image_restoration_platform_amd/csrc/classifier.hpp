// classifier.hpp -- launch interface of the fused classifier scan (classifier.hip).
#pragma once
#include <algorithm>

#include "common.hpp"

namespace ire {

struct ClassifierTables {       // device copies of csrc/grey_tables.inc
    const unsigned int* lin16;  // [256]
    const unsigned int* thr;    // [257]
    const unsigned char* inv;   // [IRE_GREY_NBUCKETS]
};

// The accumulator block of the scan (one allocation, engine.cpp::ensure_io): for a capacity of `cap` images
//   [cap][14] u64 sums | [CLS_TICKET_CAP] u64 workgroup tickets (must be zero at allocation; the kernel resets them) | [CLS_MAX_WG][14] u64
//   workgroup partials (a launch uses at most CLS_MAX_WG workgroups in all).
constexpr int CLS_MAX_WG = 768;              // three workgroups per CU (42 KB of LDS, 142 VGPRs each)
constexpr int CLS_TICKET_CAP = 64;              // >= max_batch of any engine (ire_config: 1..64)
inline size_t cls_sums_bytes() { return ((size_t)CLS_TICKET_CAP * 14 + CLS_TICKET_CAP + (size_t)CLS_MAX_WG * 14) * 8; }
inline unsigned long long* cls_tickets(unsigned long long* sums) { return sums + (size_t)CLS_TICKET_CAP * 14; }
inline unsigned long long* cls_parts(unsigned long long* sums) { return sums + (size_t)CLS_TICKET_CAP * 14 + CLS_TICKET_CAP; }

// Asynchronous on `stream`: ONE launch -- scan, then the image's last workgroup reduces the partials and finalizes.
// d_scores [n][7] double, d_label [n] int32, d_cond [n][8] float (any may be null).  d_film != null: the image's last workgroup also
// writes its FiLM vector d_film[n][film_n] = Linear(7 -> film_n)(scores) (gn.hip::film_kernel's arithmetic).
void classifier_launch(const ClassifierTables& tb, const uint8_t* d_rgb, int n, int h, int w,
                       const uint8_t* d_is_jpeg, unsigned long long* d_sums, double* d_scores,
                       int32_t* d_label, float* d_cond, hipStream_t stream,
                       const float* d_film_w = nullptr, const float* d_film_b = nullptr, int film_n = 0, float* d_film = nullptr);

// Caller-supplied scores (ire_restore with scores != NULL) -> float conditioning vector.
void scores_to_cond_launch(const double* d_scores, int n, float* d_cond, hipStream_t stream);

}  // namespace ire
