/*
 * ire.h -- C ABI of libire.so, the MI355X-native image-restoration engine.
 *
 * This is the drop-in boundary for the reference's queue-worker hot path
 * (degradation classification -> restoration -> optional <=3-image fusion).
 * The reference has no FFI of its own; its seams are two
 * duck-typed JS objects, and every entry point below names the reference
 * interface it stands behind:
 *
 *   ire_classify*  <- ClassifierService.analyze(Buffer) -> {7 scores}
 *                     server-node/src/services/classifier.js:40-99
 *                     (+ the ranking derived in promptEnhancer.js:121-136)
 *   ire_restore*   <- GeminiClient.restoreImage({prompt, images:[buf], userContext})
 *                     server-node/src/clients/geminiClient.js:32-97, called from
 *                     server-node/src/services/restorator.js:88-94
 *   ire_fuse*      <- the same restoreImage seam with images.length in 2..3
 *                     (geminiClient.js:32,49; docs only: image-restoration-platform.md:787-857)
 *   ire_submit/ire_poll <- restoreBatch's in-flight promises
 *                     server-node/src/services/restorator.js:181-236 (p-limit 3)
 *
 * Conventions: plain pointers and sizes only; the caller owns every buffer; the
 * engine never keeps an input pointer past return (past ire_poll completion for
 * ire_submit); one engine may be used from several threads; no exceptions cross
 * the ABI.  Every call returns an ire_status; ire_last_error() gives the
 * thread-local message.  Messages of IRE_ERR_INVALID_INPUT / _TIMEOUT /
 * _UNAVAILABLE contain "invalid" / "timeout" / "service unavailable", so the
 * reference's RestoratorService._classifyError (restorator.js:241-265) maps
 * them the way it maps provider errors; IRE_ERR_INTERNAL messages start with
 * "internal:" and fall through to UNKNOWN_ERROR there, which is how the
 * reference treats a provider error it does not recognise (restorator.js:262-264).  Images are decoded 8-bit sRGB, NHWC interleaved RGB
 * (what sharp(buf).raw() yields: SURVEY.md Appendix A.1); decode/encode lives
 * in the host adapters.  There is NO CPU fallback: without a gfx950 device
 * ire_init fails with IRE_ERR_UNAVAILABLE.
 */
#ifndef IRE_H
#define IRE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRE_ABI_VERSION 3   /* 3: ire_job_release, ire_affinity_plan, ire_engine_affinity, ire_profile_report, ire_encode_png_base64*, IRE_FLAG_RESULT_PNG_BASE64; ire_profile_enable mode bits 8..; ire_config.flags checked */

typedef enum ire_status {
    IRE_OK = 0,
    IRE_ERR_INVALID_INPUT = 1, /* message contains "invalid"             -> INVALID_INPUT       */
    IRE_ERR_TIMEOUT = 2,       /* message contains "timeout"             -> TIMEOUT             */
    IRE_ERR_UNAVAILABLE = 3,   /* message contains "service unavailable" -> SERVICE_UNAVAILABLE */
    IRE_ERR_INTERNAL = 4
} ire_status;

/* BF16: bf16 storage and MFMA operands, fp32 accumulate.  FP8 (cfg 4 of BASELINE.json): the C >= 128 ResBlock convolutions
 * take OCP e4m3 MFMA operands -- weights quantised per output channel at load, activations while staging -- everything
 * else as BF16; stated tolerance vs the fp32 oracle: max |d| <= 6/255, PSNR >= 38 dB. */
enum { IRE_PRECISION_BF16 = 0, IRE_PRECISION_FP8 = 1 };

/* Score order = the object-literal order of classifier.js:62-70. */
enum { IRE_SCORE_BLUR = 0, IRE_SCORE_NOISE, IRE_SCORE_LOWLIGHT, IRE_SCORE_COMPRESSION,
       IRE_SCORE_SCRATCH, IRE_SCORE_FADE, IRE_SCORE_COLORSHIFT, IRE_NUM_SCORES };

typedef struct ire_engine ire_engine;

typedef struct ire_config {
    uint32_t struct_size;     /* = sizeof(ire_config); lets the struct grow              */
    int32_t device_index;     /* HIP device ordinal (one engine = one GPU = one process) */
    int32_t precision;        /* IRE_PRECISION_BF16 | IRE_PRECISION_FP8                  */
    int32_t max_batch;        /* images per call, 1..64 (default 8 when 0)               */
    int32_t num_streams;      /* images restored concurrently on separate HIP streams
                                 (default 0 = engine's choice)                           */
    const char* weights_path; /* RestoreNet-v0 weight file (DESIGN.md "weight file");
                                 NULL: classify/fuse only until ire_load_weights         */
    uint32_t flags;           /* IRE_FLAG_* bits; unknown bits are rejected               */
} ire_config;
/* The batcher (ire_submit / ire_poll) delivers every result as the base64 TEXT of a PNG file of the restored image, encoded
 * on the device (ire_encode_png_base64_device below), instead of raw pixels: ire_poll's out_rgb then receives
 * ire_png_base64_bytes(h, w) ASCII characters -- the string restorator.js:108 puts on the wire, with no codec or base64 work
 * left for the host. */
#define IRE_FLAG_RESULT_PNG_BASE64 1u



typedef struct ire_timings { /* GPU time measured with HIP events on the engine's stream; mirrors
                                timings.classify_ms / restore_ms of restorator.js:48-98         */
    double classify_ms;
    double restore_ms;
    double total_ms;
} ire_timings;

/* ---- lifecycle ------------------------------------------------------------------------ */
int ire_abi_version(void);
int ire_init(const ire_config* cfg, ire_engine** out);
void ire_shutdown(ire_engine* e);
/* Thread-local message of the last failing call on this thread ("" if none). */
const char* ire_last_error(void);
/* Load RestoreNet-v0 weights from memory (same bytes as the weight file). */
int ire_load_weights(ire_engine* e, const void* blob, size_t bytes);
/* Images of this shape one restore call can take right now: min(max_batch, what the free HBM (hipMemGetInfo) holds
 * at the engine's real per-image footprint: activation workspace + staging); 0 if the shape is unsupported or not
 * even one image fits (a restore call would then fail with IRE_ERR_UNAVAILABLE "out of device memory"). */
int ire_max_batch_for(ire_engine* e, int h, int w);

/* ---- host-buffer entry points (synchronous; copy in, run, copy out) -------------------- */
/* analyze(): n images of h x w, rows row_stride bytes apart (>= 3*w), images n apart by
 * h*row_stride.  is_jpeg[i] != 0 <=> metadata.format === 'jpeg' (classifier.js:180).
 * scores_out: n*7 doubles; label_out: n int32 = first-max argmax in key order (may be NULL). */
int ire_classify(ire_engine* e, const uint8_t* rgb, int n, int h, int w, int row_stride,
                 const uint8_t* is_jpeg, double* scores_out, int32_t* label_out);
/* restoreImage() with one image per job: out_rgb is n*h*w*3 (tightly packed).  scores==NULL
 * => classify inside (is_jpeg then required); else n*7 doubles used as conditioning.
 * h and w must be multiples of 8 (three stride-2 levels), >= 16. */
int ire_restore(ire_engine* e, const uint8_t* rgb, int n, int h, int w, const double* scores,
                const uint8_t* is_jpeg, uint8_t* out_rgb, ire_timings* t);
/* restoreImage() with k = 2..3 views of one scene: align to view 0 and blend.  out: h*w*3.
 * noise_score: the classifier's noise score of view 0 (sets the blend sigma); <0 => classify inside. */
int ire_fuse(ire_engine* e, const uint8_t* rgb_views, int k, int h, int w, double noise_score,
             uint8_t* out_rgb, int32_t* shifts_out /* k*2 (dy,dx), may be NULL */, ire_timings* t);

/* ---- device-buffer entry points (asynchronous on `stream`, a hipStream_t or NULL) ------ */
/* Same semantics; every pointer is device memory of e's GPU, tightly packed. The FastAPI /
 * PyTorch-ROCm host and bench.py use these with tensor.data_ptr() and the torch stream. */
int ire_classify_device(ire_engine* e, const uint8_t* d_rgb, int n, int h, int w,
                        const uint8_t* d_is_jpeg, double* d_scores, int32_t* d_label, void* stream);
int ire_restore_device(ire_engine* e, const uint8_t* d_rgb, int n, int h, int w,
                       const double* d_scores /* NULL => classify inside */, const uint8_t* d_is_jpeg,
                       uint8_t* d_out_rgb, void* stream);
int ire_fuse_device(ire_engine* e, const uint8_t* d_rgb_views, int k, int h, int w, double noise_score,
                    uint8_t* d_out_rgb, int32_t* d_shifts, void* stream);
/* Several restoreImage() calls with 2..3 images each (geminiClient.js:32,49), coalesced like the batcher coalesces single-image
 * jobs: nsets (1..16) independent view sets of one shape in one call -- d_rgb_views [nsets][k][h][w][3], d_out_rgb
 * [nsets][h][w][3], d_shifts [nsets][k][2] (may be NULL), noise_scores [nsets] doubles in HOST memory (each < 0 => classify
 * view 0 of that set inside).  Results are those of nsets ire_fuse_device calls; the kernel chain runs once. */
int ire_fuse_batch_device(ire_engine* e, const uint8_t* d_rgb_views, int nsets, int k, int h, int w,
                          const double* noise_scores, uint8_t* d_out_rgb, int32_t* d_shifts, void* stream);

/* ---- preprocess step in front of the path (server-node/src/middleware/imagePreprocess.js:24-91) ---- */
/* Pixel part of preprocessImage(): EXIF auto-orient (:43) and fit-inside-max_dim Lanczos-3 resize (:46-55); the JPEG
 * q85 4:4:4 encode (:57-64) stays with the host codec.  width/height are the STORED dimensions (what sharp's
 * metadata() reports, :40,:46), orientation the EXIF tag 1..8, max_dim 2048 in the reference (:4).
 * ire_preprocess_plan is pure host arithmetic (no GPU): it returns the upright, fitted size. */
int ire_preprocess_plan(int width, int height, int orientation, int max_dim, int* out_w, int* out_h, int* resized);
/* rgb: h*w*3 stored pixels; out: out_h*out_w*3 as planned.  Integer two-pass resampler, bit-exact with oracle/preprocess.py. */
int ire_preprocess(ire_engine* e, const uint8_t* rgb, int h, int w, int orientation, int max_dim,
                   uint8_t* out_rgb, int out_h, int out_w);
int ire_preprocess_device(ire_engine* e, const uint8_t* d_rgb, int h, int w, int orientation, int max_dim,
                          uint8_t* d_out_rgb, int out_h, int out_w, void* stream);

/* ---- result side of the seam: restored pixels -> base64 text of a PNG file, on the device (restorator.js:108: `restoredImage` is
 * a base64 string of an ENCODED image; geminiClient.js:75-88) ----
 * The file: signature, IHDR (8-bit RGB), ONE IDAT chunk = a zlib stream of STORED deflate blocks (scanline filter 0) with its
 * Adler-32, its CRC-32, IEND -- every PNG decoder reads it; 0.2 % larger than the pixels.  The text: RFC 4648 base64, '=' padded,
 * no terminator.  Bit-exact against zlib / base64 / a PNG decoder (oracle/encode.py).  w a multiple of 8; n <= max_batch.
 * ire_png_base64_bytes: characters per image (0: unsupported size).  Images / texts are h*w*3 / stride_bytes apart. */
size_t ire_png_base64_bytes(int h, int w);
int ire_encode_png_base64_device(ire_engine* e, const uint8_t* d_rgb, int n, int h, int w, uint8_t* d_chars, size_t stride_bytes,
                                 void* stream);
int ire_encode_png_base64(ire_engine* e, const uint8_t* rgb, int n, int h, int w, uint8_t* chars, size_t stride_bytes);

/* ---- async batcher (restoreBatch's in-flight promises) ---------------------------------- */
typedef struct ire_job ire_job;
/* Queue one h x w image for restoration; jobs of equal shape are coalesced into batches of up
 * to max_batch.  The input is copied before return.  scores: the 7 doubles a previous ire_classify
 * of this image returned (the job is then not classified again), or NULL => classify inside. */
int ire_submit(ire_engine* e, const uint8_t* rgb, int h, int w, int is_jpeg, const double* scores, ire_job** job_out);
/* Wait up to timeout_ms (<0: forever) for the job; on IRE_OK out_rgb (h*w*3 pixel bytes, or with IRE_FLAG_RESULT_PNG_BASE64
 * ire_png_base64_bytes(h, w) characters), scores_out (7, may
 * be NULL) and t (may be NULL) are filled and the job is released; any other status but
 * IRE_ERR_TIMEOUT releases it too.  IRE_ERR_TIMEOUT leaves the job pending and the handle valid:
 * poll again, or give the job up with ire_job_release.  One thread at a time per handle. */
int ire_poll(ire_engine* e, ire_job* job, int timeout_ms, uint8_t* out_rgb, double* scores_out,
             ire_timings* t);
/* Give up a job without fetching its result -- the caller's promise was rejected on a timeout and
 * the retry policy (server-node/src/utils/retry.js:12-47, 3 attempts) will submit the image anew:
 * frees the handle and whatever only it kept alive (its place among a finished batch's unread
 * results, so the staging slot can cycle; its queued pixels if it never reached a batch).  A job
 * already gathered into a batch is still computed with it.  e == NULL after ire_shutdown: frees
 * the handle only.  job == NULL: no-op. */
int ire_job_release(ire_engine* e, ire_job* job);

/* ---- cfg 4: one large image restored as row strips (imagePreprocess.js:4 caps uploads at 2048 px) ----
 * The image is cut into nstrips equal row strips; every layer runs strip by strip, the boundary rows of every tensor a 3x3
 * convolution reads are exchanged between neighbouring strips after the layer that produced it (per-level halo exchange),
 * and the GroupNorm partial statistics of all strips are combined before each GroupNorm.  The result is bit-identical to
 * ire_restore_device on the whole image.  H must be a multiple of nstrips, rows per strip a multiple of 128, W of 8. */
/* All strips on this GPU ("virtual ranks": the exchange steps are in-device copies).  d_scores NULL => classify inside. */
int ire_restore_tiled_device(ire_engine* e, const uint8_t* d_rgb, int h, int w, int nstrips, const double* d_scores,
                             const uint8_t* d_is_jpeg, uint8_t* d_out_rgb, void* stream);
/* One rank of the multi-GPU layout: a session owns strips [first_strip, first_strip + nlocal) and runs the layer program one
 * op at a time; between ops the HOST moves halo rows and statistics between ranks (sharding.py: send/recv + all_gather over
 * RCCL).  d_stats_all: device buffer of ire_strips_stats_bytes(h, w) bytes owned by the caller (a torch tensor): the global
 * array of GroupNorm partials, of which this session fills its slice. */
typedef struct ire_strips ire_strips;
typedef struct ire_strip_xchg {   /* what op k left to exchange */
    int32_t halo_bytes;           /* bytes of one boundary row of the op's output (0: nothing to exchange) */
    int32_t has_up, has_down;     /* this session has a remote neighbour above / below */
    int64_t stats_offset_bytes;   /* slice of d_stats_all this session just wrote ... */
    int64_t stats_local_bytes;    /* ... its size (0: the op wrote no statistics) ... */
    int64_t stats_total_bytes;    /* ... and the size of the complete array for this op (equal slices per strip) */
} ire_strip_xchg;
size_t ire_strips_stats_bytes(int h, int w);
int ire_strips_open(ire_engine* e, int h, int w, int nstrips_total, int first_strip, int nlocal, void* d_stats_all, ire_strips** out);
void ire_strips_close(ire_strips* s);
int ire_strips_num_ops(ire_strips* s);
/* d_rows_with_halo: (nlocal*rows_per_strip + 2) x w x 3: the session's rows with one row above and below (rows outside the
 * image are never read); d_scores: the 7 classifier scores of the WHOLE image (the rank that took the job classified it). */
int ire_strips_set_input(ire_strips* s, const uint8_t* d_rows_with_halo, const double* d_scores, void* stream);
int ire_strips_run_op(ire_strips* s, int k, void* stream, ire_strip_xchg* info);
/* copy the first / last real row of op k's output into d_send_up / d_send_down (halo_bytes each; NULL or no neighbour: skipped) */
int ire_strips_pack_halo(ire_strips* s, int k, uint8_t* d_send_up, uint8_t* d_send_down, void* stream);
/* copy the neighbours' rows into the halo rows of op k's output */
int ire_strips_unpack_halo(ire_strips* s, int k, const uint8_t* d_recv_up, const uint8_t* d_recv_down, void* stream);
/* d_out_rows: nlocal*rows_per_strip x w x 3 */
int ire_strips_get_output(ire_strips* s, uint8_t* d_out_rows, void* stream);

/* ---- host feeding at 8 ranks per host (restorator.js:198-211: one independent job per image) ----
 * The engine's service threads (batch launcher, completer) run on the NUMA node of their GPU, on the share of that node's
 * CPUs that falls to this GPU among the node's GPUs (csrc/affinity.hpp); IRE_CPU_AFFINITY=off | <cpulist> overrides.
 * ire_affinity_plan is pure host arithmetic over sysfs (no GPU): sysfs_root NULL = "/sys"; pci_bdf as hipDeviceGetPCIBusId
 * prints it ("0000:c1:00.0"); cpulist_out "" when the node is unknown.  ire_engine_affinity: what engine e applied. */
int ire_affinity_plan(const char* sysfs_root, const char* pci_bdf, char* cpulist_out, size_t cap, int32_t* numa_node_out,
                      int32_t* slot_out, int32_t* nslots_out);
int ire_engine_affinity(ire_engine* e, char* cpulist_out, size_t cap, int32_t* numa_node_out);

/* ---- service gauges (getHealthStatus + /health/ready dependency entry: restorator.js:289-314, healthRouter.js:80-117) ---- */
typedef struct ire_engine_stats {
    uint32_t struct_size;   /* = sizeof(ire_engine_stats) */
    int32_t queue_depth;    /* jobs waiting in the batcher */
    int64_t batches;        /* engine batches (restore calls) since ire_init */
    int64_t images;         /* images restored since ire_init */
    int32_t last_batch;     /* images in the most recent batch: > 1 shows in-flight jobs being coalesced */
    int32_t max_batch;
    double images_per_sec;  /* gauge: images restored over the last 10 s window / its span (0 when idle) */
} ire_engine_stats;
int ire_get_stats(ire_engine* e, ire_engine_stats* out);

/* ---- measurement / diagnostics (used by bench.py and tests; not part of the job path) --- */
/* Raw integer accumulators of the classifier scan for n images (14 u64 each, order documented in
 * csrc/classifier_finalize.hpp) of the last classify / restore call, copied to host. */
int ire_debug_classifier_sums(ire_engine* e, int n, uint64_t* sums_out);
/* Turn per-layer capture on/off (slow: synchronises after every layer; tests only). */
int ire_debug_capture(ire_engine* e, int on);
/* Copy one named intermediate activation of the last ire_restore* call to host as float32 NHWC.
 * Returns the element count through *count (call with out==NULL to query). */
int ire_debug_activation(ire_engine* e, const char* name, float* out, size_t* count);
/* Accumulated HIP-event time (ms) and launch count per kernel family since the last reset:
 * family in {"classifier","conv3x3","conv1x1","stem","head","gn_finalize","fusion","all"}. */
int ire_profile_enable(ire_engine* e, int mode /* low byte: 0 off, 1 all families, 2 conv3x3 only (least overhead);
                                                   bits 8..: N > 1 = in mode 2 time only every N-th pass of the network */);
int ire_profile_query(ire_engine* e, const char* family, double* ms_out, int64_t* launches_out,
                      double* flops_out, double* bytes_out);
int ire_profile_reset(ire_engine* e);
/* Per-layer-group breakdown of the profiled convolution launches since the last reset, as a JSON array of
 * {"group" ("L0.rb1" .. "L3.rb2", "down0".., "up0".., "stem", "head"), "kernel", "level", "cin", "cout", "launches", "ms",
 *  "flops" (algorithmic), "flops_executed" (what the kernel issues: the sub-pixel `up` form runs 4 of 9 taps), "bytes"
 *  (algorithmic HBM bytes)} written to buf (cap bytes, NUL-terminated).  *needed_out = bytes required; call with
 * buf == NULL to size the buffer.  bench.py's roofline.per_level is computed from this. */
int ire_profile_report(ire_engine* e, char* buf, size_t cap, size_t* needed_out);

#ifdef __cplusplus
}
#endif
#endif /* IRE_H */
