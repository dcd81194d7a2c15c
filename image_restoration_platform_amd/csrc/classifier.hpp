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

// Asynchronous on `stream`: zero the accumulators, scan, finalize.
// d_scores [n][7] double, d_label [n] int32, d_cond [n][8] float (any may be null).
void classifier_launch(const ClassifierTables& tb, const uint8_t* d_rgb, int n, int h, int w,
                       const uint8_t* d_is_jpeg, unsigned long long* d_sums, double* d_scores,
                       int32_t* d_label, float* d_cond, hipStream_t stream);

// Caller-supplied scores (ire_restore with scores != NULL) -> float conditioning vector.
void scores_to_cond_launch(const double* d_scores, int n, float* d_cond, hipStream_t stream);

}  // namespace ire
