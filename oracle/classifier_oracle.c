/*
 * oracle/classifier_oracle.c -- TEST INFRASTRUCTURE ONLY (not product code).
 *
 * Plain-C CPU restatement of the reference's 7-score degradation classifier
 * (server-node/src/services/classifier.js) including the libvips/sharp pixel
 * semantics the JS delegates to.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file.
 *
 * PINNING STATUS: the in-tree JS arithmetic (reductions, thresholds, score
 * formulas) is followed line by line and is pinned by the analytic known-answer
 * vectors derived from the reference's own fixtures
 * (server-node/tests/utils/imageFixtures.js:5-45, tests/golden/classifier_kat.json).
 * The sharp@0.33.5 / libvips 8.15 pixel kernels (grey conversion, convolve
 * border/cast rules, gaussblur mask, stats stdev) are a third-party dependency
 * that is absent from /root/reference (package-lock.json:5190-5228); their
 * published behaviour is restated below and each assumption is tagged [A<n>]
 * after SURVEY.md Appendix A so a later session can flip it.  No real sharp
 * output exists offline => for those items "parity unpinned" beyond the
 * reference's inequality tests (tests/classifierService.test.js:19-57).
 *
 * Two evaluations are provided:
 *   ire_oracle_classify()         integer-sum form (exact Sum, Sum^2 in u64,
 *                                 variance from the exact 128-bit N*S2-S^2);
 *                                 the HIP path must match this BIT-EXACTLY.
 *   ire_oracle_classify_twopass() literal JS order: materialise the u8 edge
 *                                 buffers and run _calculateVariance's two
 *                                 sequential double passes (classifier.js:262-266).
 *                                 Differs from the integer form by O(1e-12) rel.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IRE_ORACLE_OK 0
#define IRE_ORACLE_INVALID 1

typedef struct {
    uint64_t sum_c[3];    /* per-channel sum of u8            (stats(), classifier.js:52)   */
    uint64_t sumsq_c[3];  /* per-channel sum of squares                                      */
    uint64_t sum_blur;    /* sum over all 3*px bytes of gaussblur(1) output (classifier.js:297) */
    uint64_t sumsq_blur;
    uint64_t sum_e8;      /* clip_u8(conv3(grey, Laplacian-8))  (classifier.js:107-115)     */
    uint64_t sumsq_e8;
    uint64_t sum_e9;      /* clip_u8(conv3(grey, high-pass 9))  (classifier.js:135-143)     */
    uint64_t sumsq_e9;
    uint64_t scratch_v;   /* _detectLinearFeatures verticalCount   (classifier.js:326)      */
    uint64_t scratch_h;   /* _detectLinearFeatures horizontalCount (classifier.js:329)      */
} ire_oracle_sums;

/* ------------------------------------------------------------------------- */
/* [A3] sharp .grayscale() = libvips colourspace(B_W): sRGB -> linear light ->
 * Y = 0.2126 R + 0.7152 G + 0.0722 B -> sRGB transfer curve -> u8.
 * Restated in exact integer arithmetic so CPU and GPU agree bit for bit:
 *   lin16[v] = round(65536 * srgb_to_linear(v/255))
 *   Y        = 2126*lin16[r] + 7152*lin16[g] + 722*lin16[b]   (0 .. 655,360,000)
 *   grey     = max{ v : thr[v] <= Y },  thr[v] = ceil(655360000 * srgb_to_linear((v-0.5)/255))
 * i.e. round-to-nearest in the gamma domain.  R=G=B=v gives grey=v (Appendix A.3). */
static double srgb_to_linear(double f) {
    if (f <= 0.04045) return f / 12.92;
    return pow((f + 0.055) / 1.055, 2.4);
}

void ire_oracle_grey_tables(uint32_t lin16[256], uint32_t thr[256]) {
    for (int v = 0; v < 256; ++v) {
        lin16[v] = (uint32_t)floor(65536.0 * srgb_to_linear((double)v / 255.0) + 0.5);
        thr[v] = (v == 0) ? 0u
                          : (uint32_t)ceil(655360000.0 * srgb_to_linear(((double)v - 0.5) / 255.0));
    }
}

static uint32_t g_lin16[256], g_thr[256];
static int g_tables_ready = 0;
static void ensure_tables(void) {
    if (!g_tables_ready) {
        ire_oracle_grey_tables(g_lin16, g_thr);
        g_tables_ready = 1;
    }
}

static uint8_t grey_of(uint8_t r, uint8_t g, uint8_t b) {
    uint32_t y = 2126u * g_lin16[r] + 7152u * g_lin16[g] + 722u * g_lin16[b];
    int lo = 0, hi = 255; /* largest v with thr[v] <= y */
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (g_thr[mid] <= y) lo = mid; else hi = mid - 1;
    }
    return (uint8_t)lo;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* JS Math.min / Math.max propagate NaN (ECMA-262 21.3.2.24/25). */
static double js_min(double a, double b) { if (isnan(a) || isnan(b)) return NAN; return a < b ? a : b; }
static double js_max(double a, double b) { if (isnan(a) || isnan(b)) return NAN; return a > b ? a : b; }

/* exact population variance from integer sums: (N*S2 - S^2) / N^2.
 * The 128-bit numerator is converted as (double)hi * 2^64 + (double)lo so the
 * GPU (which has no 128-bit int->fp conversion) can do the identical steps. */
static double popvar_from_sums(uint64_t n, uint64_t s, uint64_t s2) {
    unsigned __int128 num = (unsigned __int128)n * s2 - (unsigned __int128)s * s;
    uint64_t hi = (uint64_t)(num >> 64), lo = (uint64_t)num;
    double d = (double)hi * 18446744073709551616.0 + (double)lo;
    double dn = (double)n;
    return d / (dn * dn);
}

/* [A8] libvips stats: mean = S/N ; stdev = sqrt(fabs(S2 - S*S/N) / (N-1)) in doubles. */
static void vips_mean_stdev(uint64_t n, uint64_t s, uint64_t s2, double* mean, double* sd) {
    double vals = (double)n, ds = (double)s, ds2 = (double)s2;
    *mean = ds / vals;
    *sd = sqrt(fabs(ds2 - (ds * ds / vals)) / (vals - 1.0));
}

/* classifier.js:156-258 -- the three stats-only analysers + score assembly.   */
static void scores_from(const ire_oracle_sums* S, uint64_t px, int is_jpeg,
                        double var_e8, double var_e9, double var_rgb, double var_blur,
                        double scores[7], int32_t* label) {
    double mean[3], sd[3];
    for (int c = 0; c < 3; ++c) vips_mean_stdev(px, S->sum_c[c], S->sumsq_c[c], &mean[c], &sd[c]);

    /* blur: classifier.js:118-122 */
    double nv = js_min(var_e8 / 1000.0, 1.0);
    scores[0] = js_max(0.0, 1.0 - nv);
    /* noise: classifier.js:145-146 */
    scores[1] = js_min(sqrt(var_e9) / 50.0, 1.0);
    /* lowLight: classifier.js:159-167 (reduce starts at 0) */
    double mb = (((0.0 + mean[0]) + mean[1]) + mean[2]) / 3.0;
    double nb = mb / 255.0;
    scores[2] = (nb < 0.3) ? js_min((0.3 - nb) * 2.0, 1.0) : 0.0;
    /* compression: classifier.js:180-186, 299-303 */
    if (!is_jpeg) scores[3] = 0.0;
    else {
        double delta = js_max(0.0, var_rgb - var_blur);
        scores[3] = js_min(js_min(delta / 500.0, 1.0), 1.0);
    }
    /* scratch: classifier.js:335-336, 210 */
    double total = (double)(S->scratch_v + S->scratch_h);
    scores[4] = js_min(js_min(total / 1000.0, 1.0), 1.0);
    /* fade: classifier.js:223-228, 272-286 */
    double sat = sqrt((sd[0] * sd[0] + sd[1] * sd[1]) + sd[2] * sd[2]) / 255.0;
    double colorfulness = js_min(sat, 1.0);
    double avg_sd = (((0.0 + sd[0]) + sd[1]) + sd[2]) / 3.0;
    double contrast = js_min(avg_sd / 64.0, 1.0);
    scores[5] = js_min((1.0 - colorfulness) * 0.6 + (1.0 - contrast) * 0.4, 1.0);
    /* colorShift: classifier.js:245-253 */
    double avg = ((mean[0] + mean[1]) + mean[2]) / 3.0;
    double dr = avg > 0 ? fabs(mean[0] - avg) / avg : 0.0;
    double dg = avg > 0 ? fabs(mean[1] - avg) / avg : 0.0;
    double db = avg > 0 ? fabs(mean[2] - avg) / avg : 0.0;
    scores[6] = js_min(js_max(js_max(dr, dg), db) * 2.0, 1.0);

    /* label = first-max argmax in key order (SURVEY.md 8a "Argmax label"). */
    int best = 0;
    for (int i = 1; i < 7; ++i) if (scores[i] > scores[best]) best = i;
    if (label) *label = best;
}

/* Materialise the per-pixel buffers exactly as the sharp pipelines would. */
typedef struct { uint8_t *grey, *e8, *e9, *e4, *blur; } planes_t;

static void free_planes(planes_t* P) { free(P->grey); free(P->e8); free(P->e9); free(P->e4); free(P->blur); }

static int build_planes(const uint8_t* rgb, int h, int w, int row_stride, planes_t* P) {
    size_t px = (size_t)h * w;
    memset(P, 0, sizeof(*P));
    P->grey = (uint8_t*)malloc(px); P->e8 = (uint8_t*)malloc(px); P->e9 = (uint8_t*)malloc(px);
    P->e4 = (uint8_t*)malloc(px); P->blur = (uint8_t*)malloc(px * 3);
    uint8_t* tmp = (uint8_t*)malloc(px * 3);
    if (!P->grey || !P->e8 || !P->e9 || !P->e4 || !P->blur || !tmp) { free(tmp); free_planes(P); return -1; }

    for (int y = 0; y < h; ++y) {
        const uint8_t* row = rgb + (size_t)y * row_stride;
        for (int x = 0; x < w; ++x)
            P->grey[(size_t)y * w + x] = grey_of(row[3 * x], row[3 * x + 1], row[3 * x + 2]);
    }
    /* [A4][A5][A6] greyscale BEFORE convolve; scale = max(kernel sum,1) = 1; borders are
     * edge-replicated (VIPS_EXTEND_COPY); float result saturates to u8 on the raw() cast. */
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            int s9 = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx)
                    s9 += P->grey[(size_t)clampi(y + dy, 0, h - 1) * w + clampi(x + dx, 0, w - 1)];
            int c = P->grey[(size_t)y * w + x];
            int n = P->grey[(size_t)clampi(y - 1, 0, h - 1) * w + x];
            int s = P->grey[(size_t)clampi(y + 1, 0, h - 1) * w + x];
            int e = P->grey[(size_t)y * w + clampi(x + 1, 0, w - 1)];
            int wv = P->grey[(size_t)y * w + clampi(x - 1, 0, w - 1)];
            size_t i = (size_t)y * w + x;
            P->e8[i] = (uint8_t)clampi(9 * c - s9, 0, 255);            /* [-1..8..-1]  */
            P->e9[i] = (uint8_t)clampi(10 * c - s9, 0, 255);           /* [-1..9..-1]  */
            P->e4[i] = (uint8_t)clampi(4 * c - n - s - e - wv, 0, 255); /* [0,-1,0,-1,4,-1,0,-1,0] */
        }
    }
    /* [A7 revised] .blur(1) = vips gaussblur(sigma=1, min_ampl=0.2, precision=integer):
     * gaussmat stops at the first x with exp(-x^2/2) < 0.2 (x=2) => width 2*(2-1)+1 = 3,
     * integer mask round(20*{e^-0.5, 1, e^-0.5}) = {12, 20, 12}, scale 44; separable
     * (horizontal then vertical), each pass an integer convolution rounded
     * (sum + 22) / 44 back to u8, borders edge-replicated, per channel. */
    for (int y = 0; y < h; ++y) {
        const uint8_t* row = rgb + (size_t)y * row_stride;
        for (int x = 0; x < w; ++x) {
            int xl = clampi(x - 1, 0, w - 1), xr = clampi(x + 1, 0, w - 1);
            for (int c = 0; c < 3; ++c) {
                int v = 12 * row[3 * xl + c] + 20 * row[3 * x + c] + 12 * row[3 * xr + c];
                tmp[((size_t)y * w + x) * 3 + c] = (uint8_t)((v + 22) / 44);
            }
        }
    }
    for (int y = 0; y < h; ++y) {
        int yu = clampi(y - 1, 0, h - 1), yd = clampi(y + 1, 0, h - 1);
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) {
                int v = 12 * tmp[((size_t)yu * w + x) * 3 + c] + 20 * tmp[((size_t)y * w + x) * 3 + c] +
                        12 * tmp[((size_t)yd * w + x) * 3 + c];
                P->blur[((size_t)y * w + x) * 3 + c] = (uint8_t)((v + 22) / 44);
            }
    }
    free(tmp);
    return 0;
}

static void accumulate(const uint8_t* rgb, int h, int w, int row_stride, const planes_t* P, ire_oracle_sums* S) {
    memset(S, 0, sizeof(*S));
    for (int y = 0; y < h; ++y) {
        const uint8_t* row = rgb + (size_t)y * row_stride;
        for (int x = 0; x < w; ++x) {
            size_t i = (size_t)y * w + x;
            for (int c = 0; c < 3; ++c) {
                uint64_t v = row[3 * x + c], b = P->blur[i * 3 + c];
                S->sum_c[c] += v; S->sumsq_c[c] += v * v;
                S->sum_blur += b; S->sumsq_blur += b * b;
            }
            uint64_t a = P->e8[i], d = P->e9[i];
            S->sum_e8 += a; S->sumsq_e8 += a * a;
            S->sum_e9 += d; S->sumsq_e9 += d * d;
        }
    }
    /* classifier.js:316-333 stride-4 probes on the Laplacian-4 buffer */
    for (int y = 0; y < h; y += 4)
        for (int x = 0; x < w; x += 4) {
            size_t idx = (size_t)y * w + x;
            if (P->e4[idx] > 200) {
                if (x + 1 < w) S->scratch_v += P->e4[idx + 1] > 200 ? 1 : 0;
                if (y + 1 < h) S->scratch_h += P->e4[idx + w] > 200 ? 1 : 0;
            }
        }
}

/* classifier.js:262-266, literally: two sequential reduce passes in doubles. */
static double js_calculate_variance(const uint8_t* buf, size_t len) {
    double sum = 0.0;
    for (size_t i = 0; i < len; ++i) sum = sum + (double)buf[i];
    double mean = sum / (double)len;
    double acc = 0.0;
    for (size_t i = 0; i < len; ++i) { double d = (double)buf[i] - mean; acc = acc + d * d; }
    return acc / (double)len;
}

static int check_args(const uint8_t* rgb, int h, int w, int row_stride) {
    if (!rgb || h <= 0 || w <= 0 || row_stride < 3 * w) return IRE_ORACLE_INVALID;
    return IRE_ORACLE_OK;
}

int ire_oracle_classify(const uint8_t* rgb, int h, int w, int row_stride, int is_jpeg,
                        double scores[7], int32_t* label, ire_oracle_sums* sums_out) {
    if (check_args(rgb, h, w, row_stride)) return IRE_ORACLE_INVALID;
    ensure_tables();
    planes_t P;
    if (build_planes(rgb, h, w, row_stride, &P)) return IRE_ORACLE_INVALID;
    ire_oracle_sums S;
    accumulate(rgb, h, w, row_stride, &P, &S);
    free_planes(&P);
    uint64_t px = (uint64_t)h * w;
    double var_e8 = popvar_from_sums(px, S.sum_e8, S.sumsq_e8);
    double var_e9 = popvar_from_sums(px, S.sum_e9, S.sumsq_e9);
    double var_rgb = popvar_from_sums(3 * px, S.sum_c[0] + S.sum_c[1] + S.sum_c[2],
                                      S.sumsq_c[0] + S.sumsq_c[1] + S.sumsq_c[2]);
    double var_blur = popvar_from_sums(3 * px, S.sum_blur, S.sumsq_blur);
    scores_from(&S, px, is_jpeg, var_e8, var_e9, var_rgb, var_blur, scores, label);
    if (sums_out) *sums_out = S;
    return IRE_ORACLE_OK;
}

int ire_oracle_classify_twopass(const uint8_t* rgb, int h, int w, int row_stride, int is_jpeg,
                                double scores[7], int32_t* label) {
    if (check_args(rgb, h, w, row_stride)) return IRE_ORACLE_INVALID;
    ensure_tables();
    planes_t P;
    if (build_planes(rgb, h, w, row_stride, &P)) return IRE_ORACLE_INVALID;
    ire_oracle_sums S;
    accumulate(rgb, h, w, row_stride, &P, &S);
    size_t px = (size_t)h * w;
    double var_e8 = js_calculate_variance(P.e8, px);
    double var_e9 = js_calculate_variance(P.e9, px);
    double var_rgb = 0.0, var_blur = 0.0;
    if (is_jpeg) {
        /* raw().toBuffer() is tightly packed (Appendix A.1): repack if the caller passed a stride */
        uint8_t* packed = (uint8_t*)malloc(px * 3);
        if (!packed) { free_planes(&P); return IRE_ORACLE_INVALID; }
        for (int y = 0; y < h; ++y) memcpy(packed + (size_t)y * w * 3, rgb + (size_t)y * row_stride, (size_t)w * 3);
        var_rgb = js_calculate_variance(packed, px * 3);
        var_blur = js_calculate_variance(P.blur, px * 3);
        free(packed);
    }
    free_planes(&P);
    scores_from(&S, (uint64_t)px, is_jpeg, var_e8, var_e9, var_rgb, var_blur, scores, label);
    return IRE_ORACLE_OK;
}

/* Expose the intermediate planes for kernel debugging in tests (grey, e8, e9, e4: px bytes; blur: 3*px). */
int ire_oracle_planes(const uint8_t* rgb, int h, int w, int row_stride,
                      uint8_t* grey, uint8_t* e8, uint8_t* e9, uint8_t* e4, uint8_t* blur) {
    if (check_args(rgb, h, w, row_stride)) return IRE_ORACLE_INVALID;
    ensure_tables();
    planes_t P;
    if (build_planes(rgb, h, w, row_stride, &P)) return IRE_ORACLE_INVALID;
    size_t px = (size_t)h * w;
    if (grey) memcpy(grey, P.grey, px);
    if (e8) memcpy(e8, P.e8, px);
    if (e9) memcpy(e9, P.e9, px);
    if (e4) memcpy(e4, P.e4, px);
    if (blur) memcpy(blur, P.blur, px * 3);
    free_planes(&P);
    return IRE_ORACLE_OK;
}
