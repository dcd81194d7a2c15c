// engine.cpp -- engine object: weights, workspaces, layer schedule, lanes, profiler.
// See engine.hpp for the design; the layer schedule follows DESIGN.md "RestoreNet-v0"
// (SURVEY.md Appendix C), which stands behind GeminiClient.restoreImage
// (server-node/src/clients/geminiClient.js:32-97).
#include "engine.hpp"
#include "encode.hpp"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "gn.hpp"
#include "strips.hpp"
#include "grey_tables.inc"

namespace ire {

namespace {

const int kWidths[4] = {32, 64, 128, 256};
const int kFilmOff[4] = {0, 64, 192, 448};
const int kFilmDim = 960;

inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline unsigned short f32_to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;  // weights are finite: no NaN handling needed
    return (unsigned short)u;
}

// fp32 -> OCP e4m3fn (1-4-3, bias 7, no infinities, max 448), round to nearest even, saturating.  Host-side, for weights.
inline unsigned char f32_to_e4m3(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    const unsigned char sign = (u >> 31) ? 0x80 : 0;
    float a = std::fabs(f);
    if (!(a == a)) return sign | 0x7f;
    if (a >= 448.0f) return sign | 0x7e;
    if (a < 0.0009765625f) return sign;                         // < 2^-10 = half the smallest subnormal (2^-9): rounds to zero
    int e;
    std::frexp(a, &e);                                          // a = m * 2^e, m in [0.5, 1)
    int E = e - 1;                                              // a = 1.xxx * 2^E
    if (E < -6) E = -6;                                         // subnormal range: fixed exponent, step 2^-9
    const float step = std::ldexp(1.0f, E - 3);
    float q = std::nearbyint(a / step);                         // default rounding mode: to nearest even
    if (E == -6 && q < 8.0f) return sign | (unsigned char)q;    // subnormal: mantissa only
    if (q >= 16.0f) { q = 8.0f; E += 1; }
    if (E > 8) return sign | 0x7e;
    return sign | (unsigned char)(((E + 7) << 3) | ((int)q - 8));
}

}  // namespace

void* Engine::dalloc(size_t bytes) {
    void* p = nullptr;
    IRE_HIP(hipMalloc(&p, bytes ? bytes : 16));
    return p;
}

Engine::Engine(const ire_config& cfg) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        fail(IRE_ERR_UNAVAILABLE, "service unavailable: no HIP device visible (the engine has no CPU fallback)");
    device_ = cfg.device_index;
    if (device_ < 0 || device_ >= ndev) fail(IRE_ERR_INVALID_INPUT, "invalid device_index");
    IRE_HIP(hipSetDevice(device_));
    hipDeviceProp_t prop;
    IRE_HIP(hipGetDeviceProperties(&prop, device_));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        fail(IRE_ERR_UNAVAILABLE, std::string("service unavailable: device is ") + prop.gcnArchName +
                                      ", this engine is built for gfx950 (MI355X) only");
    if (cfg.precision != IRE_PRECISION_BF16 && cfg.precision != IRE_PRECISION_FP8) fail(IRE_ERR_INVALID_INPUT, "invalid precision");
    precision_ = cfg.precision;
    max_batch_ = cfg.max_batch > 0 ? cfg.max_batch : 8;
    if (max_batch_ > 64) fail(IRE_ERR_INVALID_INPUT, "invalid max_batch (1..64)");
    num_lanes_ = cfg.num_streams > 0 ? cfg.num_streams : 1;
    if (num_lanes_ > 16) num_lanes_ = 16;
    if (cfg.flags & ~(uint32_t)IRE_FLAG_RESULT_PNG_BASE64) fail(IRE_ERR_INVALID_INPUT, "invalid ire_config.flags (unknown bits set)");
    flags_ = cfg.flags;
    if (const char* v = std::getenv("IRE_CONV_V1")) rb_tile_h_ = (v[0] == '1') ? 8 : kRbTileH;
    if (const char* v = std::getenv("IRE_RB_PRIO")) prio_young_ = std::atoi(v);
    if (const char* v = std::getenv("IRE_W4")) use_w4_ = std::atoi(v);
    if (const char* v = std::getenv("IRE_W4_SPLIT")) w4_split_ = std::atoi(v);
    if (const char* v = std::getenv("IRE_UP_RB_MINC")) up_rb_min_c_ = std::atoi(v);
    if (const char* v = std::getenv("IRE_UP_SUBPIX")) up_subpixel_ = std::atoi(v);
    if (const char* v = std::getenv("IRE_UP_FUSE")) up_fuse_ = std::atoi(v);
    if (const char* v = std::getenv("IRE_GN_FOLD")) gn_fold_ = std::atoi(v);
    if (const char* v = std::getenv("IRE_PC")) pc_split_ = std::atoi(v);
    if (const char* v = std::getenv("IRE_PK")) use_pk_ = std::atoi(v);
    if (const char* v = std::getenv("IRE_UPQ")) use_upq_ = std::atoi(v);
    if (const char* v = std::getenv("IRE_DNQ")) use_dnq_ = std::atoi(v);
    if (const char* v = std::getenv("IRE_HEAD_RB")) head_rb_ = std::atoi(v);
    if (const char* v = std::getenv("IRE_DOWN_RB")) down_rb_ = std::atoi(v);
    if (const char* v = std::getenv("IRE_STEM_RB")) stem_rb_ = std::atoi(v);
    if (const char* v = std::getenv("IRE_FP8_MX")) fp8_mx_ = std::atoi(v);
    if (const char* v = std::getenv("IRE_RB_STAMPS")) {   // diagnostic: "<cout>[r]" = stamp the first such ResBlock conv
        stamps_cout_ = std::atoi(v);
        stamps_resid_ = std::strchr(v, 'r') != nullptr;
    }
    if (const char* v = std::getenv("IRE_W4_TL")) stamps_tl_ = v;     // diagnostic (-DIRE_W4_TL builds): CSV path of the workgroup timeline of the LAST stamped launch

    IRE_HIP(hipStreamCreateWithFlags(&main_stream_, hipStreamNonBlocking));
    for (auto& ev : ev_) IRE_HIP(hipEventCreate(&ev));
    IRE_HIP(hipEventCreateWithFlags(&fork_ev_, hipEventDisableTiming));
    IRE_HIP(hipEventCreateWithFlags(&busy_ev_, hipEventDisableTiming));
    lanes_.resize(num_lanes_);
    for (auto& L : lanes_) {
        IRE_HIP(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        IRE_HIP(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
    }

    // classifier tables
    unsigned int* d_lin = (unsigned int*)dalloc(sizeof(kLin16));
    unsigned int* d_thr = (unsigned int*)dalloc(sizeof(kGreyThr));
    unsigned char* d_inv = (unsigned char*)dalloc(5008);          // 5001 buckets, padded to whole 16-byte chunks (classifier.hip loads it as uint4)
    IRE_HIP(hipMemset(d_inv, 0, 5008));
    table_allocs_ = {d_lin, d_thr, d_inv};
    IRE_HIP(hipMemcpy(d_lin, kLin16, sizeof(kLin16), hipMemcpyHostToDevice));
    IRE_HIP(hipMemcpy(d_thr, kGreyThr, sizeof(kGreyThr), hipMemcpyHostToDevice));
    IRE_HIP(hipMemcpy(d_inv, kGreyInv, sizeof(kGreyInv), hipMemcpyHostToDevice));
    tables_ = ClassifierTables{d_lin, d_thr, d_inv};
    if (stamps_cout_) {
        stamps_dev_ = (unsigned long long*)dalloc(8 * 2 * 64 * 10 * 8);
        IRE_HIP(hipMemset(stamps_dev_, 0, 8 * 2 * 64 * 10 * 8));
    }

    if (cfg.weights_path && cfg.weights_path[0]) load_weights_file(cfg.weights_path);
}

Engine::~Engine() {
    (void)hipSetDevice(device_);
    (void)hipDeviceSynchronize();
    if (stamps_dev_ && std::getenv("IRE_STAMPS_RAW")) {     // diagnostic builds with their own stamp layout (conv_pk.hip PK_TICKS): the whole buffer, one value per line
        std::vector<unsigned long long> h(8 * 2 * 64 * 10);
        (void)hipMemcpy(h.data(), stamps_dev_, h.size() * 8, hipMemcpyDeviceToHost);
        if (FILE* f = std::fopen(std::getenv("IRE_STAMPS_RAW"), "w")) {
            for (size_t i = 0; i < h.size(); ++i) std::fprintf(f, "%llu\n", h[i]);
            std::fclose(f);
        }
        (void)hipFree(stamps_dev_);
        stamps_dev_ = nullptr;
    }
    if (stamps_dev_ && !stamps_tl_.empty()) {       // workgroup timeline (conv_w4.hip `tl`): raw stamps, one row per workgroup and slot
        std::vector<unsigned long long> h(8 * 2 * 64 * 10);
        (void)hipMemcpy(h.data(), stamps_dev_, h.size() * 8, hipMemcpyDeviceToHost);
        if (FILE* f = std::fopen(stamps_tl_.c_str(), "w")) {
            std::fprintf(f, "wg,slot,realtime_10ns,memtime\n");
            for (int wg = 0; wg < 256; ++wg)
                for (int sl = 0; sl < 16; ++sl)
                    if (h[(size_t)wg * 32 + sl * 2] || h[(size_t)wg * 32 + sl * 2 + 1]) std::fprintf(f, "%d,%d,%llu,%llu\n", wg, sl, h[(size_t)wg * 32 + sl * 2], h[(size_t)wg * 32 + sl * 2 + 1]);
            std::fclose(f);
        }
        (void)hipFree(stamps_dev_);
        stamps_dev_ = nullptr;
    }
    if (stamps_dev_) {   // diagnostic dump: per-stage phase durations (s_memtime ticks = shader clocks / 100 MHz ref? printed raw)
        std::vector<unsigned long long> h(8 * 2 * 64 * 10);
        (void)hipMemcpy(h.data(), stamps_dev_, h.size() * 8, hipMemcpyDeviceToHost);
        const char* names[5] = {"top->mfma_loop_end", "->vmcnt0", "->epi_barrier1", "->epilogue_end", "->stage_barrier"};
        for (int wg = 0; wg < 2; ++wg)
            for (int wv = 0; wv < 2; ++wv) {
                std::fprintf(stderr, "[stamps] wg %d wave %d (ticks per segment; -1 = not taken)\n", wg, wv * 4);
                for (int s = 0; s < 40; ++s) {
                    const unsigned long long* t = &h[(((size_t)wg * 2 + wv) * 64 + s) * 10];
                    if (!t[0]) continue;
                    long long d01 = (long long)(t[1] - t[0]), d12 = (long long)(t[2] - t[1]);
                    long long d23 = t[3] ? (long long)(t[3] - t[2]) : -1, d34 = t[3] ? (long long)(t[4] - t[3]) : (long long)(t[4] - t[2]);
                    long long d45 = (long long)(t[5] - t[4]);
                    if (t[6]) std::fprintf(stderr, "      k-steps 0..3=%lld  vmcnt wait=%lld  k-steps 4..8=%lld\n",
                                           (long long)(t[6] - t[0]), (long long)(t[7] - t[6]), (long long)(t[1] - t[7]));
                    std::fprintf(stderr, "      raw deltas t1-t0..t5-t4: %lld %lld %lld %lld %lld\n", (long long)(t[1] - t[0]), (long long)(t[2] - t[1]),
                                 (long long)(t[3] - t[2]), (long long)(t[4] - t[3]), (long long)(t[5] - t[4]));
                    if (t[3] && t[8]) std::fprintf(stderr, "      epilogue: barrier A=%lld  passes=%lld  reduce+red=%lld\n",
                                                   (long long)(t[8] - t[3]), (long long)(t[9] - t[8]), (long long)(t[4] - t[9]));
                    std::fprintf(stderr, "  s%02d %s=%lld %s=%lld %s=%lld %s=%lld %s=%lld | stage=%lld\n", s, names[0], d01, names[1], d12,
                                 names[2], d23, names[3], d34, names[4], d45, (long long)(t[5] - t[0]));
                }
            }
        (void)hipFree(stamps_dev_);
    }
    free_workspace();
    delete tiled_;
    for (void* p : net_.allocs) (void)hipFree(p);
    for (void* p : table_allocs_) (void)hipFree(p);
    for (void* p : {(void*)d_in_, (void*)d_out_, (void*)d_jpeg_, (void*)d_sums_, (void*)d_scores_, (void*)d_label_,
                    (void*)d_cond_, (void*)d_film_, d_zero_, (void*)d_fL_, (void*)d_fQ_, (void*)d_fsad_, (void*)d_fmisc_, (void*)d_fwlut_, (void*)d_pp_tab_, (void*)d_pp_mid_, (void*)d_pp_in_, (void*)d_pp_out_, (void*)d_enc_scratch_, (void*)d_enc_io_})
        if (p) (void)hipFree(p);
    for (auto& L : lanes_) {
        if (L.stream) (void)hipStreamDestroy(L.stream);
        if (L.done) (void)hipEventDestroy(L.done);
    }
    for (auto& ev : ev_) if (ev) (void)hipEventDestroy(ev);
    if (fork_ev_) (void)hipEventDestroy(fork_ev_);
    if (busy_ev_) (void)hipEventDestroy(busy_ev_);
    for (auto& r : prof_) { if (r.own_e0) (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto ev : ev_pool_) (void)hipEventDestroy(ev);
    if (main_stream_) (void)hipStreamDestroy(main_stream_);
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
void Engine::load_weights_file(const char* path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) fail(IRE_ERR_INVALID_INPUT, std::string("invalid weights_path: cannot open ") + path);
    std::streamsize sz = f.tellg();
    f.seekg(0);
    std::vector<char> blob((size_t)sz);
    if (!f.read(blob.data(), sz)) fail(IRE_ERR_INVALID_INPUT, std::string("invalid weights_path: short read ") + path);
    load_weights(blob.data(), blob.size());
}

ConvW Engine::make_conv(ConvKind kind, const std::string& wname, const std::string& bname, int cin, int cout) {
    auto wi = host_w_.find(wname), bi = host_w_.find(bname);
    if (wi == host_w_.end() || bi == host_w_.end()) fail(IRE_ERR_INVALID_INPUT, "invalid weight file: missing " + wname);
    const auto& dims = wi->second.first;
    const int taps = (kind == CONV_FUSE) ? 1 : 9, ks = (kind == CONV_FUSE) ? 1 : 3;
    if (dims.size() != 4 || dims[0] != cout || dims[1] != cin || dims[2] != ks || dims[3] != ks ||
        (int)bi->second.second.size() != cout)
        fail(IRE_ERR_INVALID_INPUT, "invalid weight file: shape of " + wname);
    const float* W = wi->second.second.data();
    ConvW c;
    c.kind = kind; c.cin = cin; c.cout = cout;
    const int kc8 = (kind == CONV_STEM) ? 1 : 4;
    const int cin_pad = (kind == CONV_STEM) ? 8 : cin;
    const int cout_pad = (kind == CONV_HEAD) ? 32 : cout;
    c.nt = conv_nt(kind, cout_pad);
    if (cin_pad % (kc8 * 8) || cout_pad % c.nt) fail(IRE_ERR_INTERNAL, "internal: conv channel counts");
    c.nkc = cin_pad / (kc8 * 8);
    c.nblocks = cout_pad / c.nt;
    if (kind == CONV_FUSE) { c.cin0 = cin / 2; c.cin1 = cin / 2; c.kc_split = c.nkc / 2; }
    else { c.cin0 = cin_pad; c.cin1 = 0; c.kc_split = c.nkc; }
    const int nsteps = conv_nsteps(kind), nkk = nsteps * 2;
    // slab layout [nblock][kchunk][kk = tap*kc8 + c8][n][e]; element = W[cout][cin = kc*kc8*8 + c8*8 + e][tap]
    std::vector<unsigned short> arr((size_t)c.nblocks * c.nkc * nkk * c.nt * 8, 0);
    for (int nb = 0; nb < c.nblocks; ++nb)
        for (int kc = 0; kc < c.nkc; ++kc)
            for (int kk = 0; kk < taps * kc8; ++kk) {
                const int tap = kk / kc8, c8 = kk % kc8;
                for (int n = 0; n < c.nt; ++n) {
                    const int co = nb * c.nt + n;
                    if (co >= cout) continue;
                    for (int e = 0; e < 8; ++e) {
                        const int ci = kc * kc8 * 8 + c8 * 8 + e;
                        if (ci >= cin) continue;
                        const float v = W[((size_t)co * cin + ci) * taps + tap];
                        arr[((((size_t)nb * c.nkc + kc) * nkk + kk) * c.nt + n) * 8 + e] = f32_to_bf16(v);
                    }
                }
            }
    c.d_w = (unsigned short*)dalloc(arr.size() * 2);
    net_.allocs.push_back(c.d_w);
    IRE_HIP(hipMemcpy(c.d_w, arr.data(), arr.size() * 2, hipMemcpyHostToDevice));
    // conv_rb.hip / conv_w4.hip (direct epilogue): slab row n of a 32-row MFMA tile carries cout perm(n) = n with bits 2 and 3
    // swapped, so that the 16 accumulators of a lane-half are two runs of 8 CONTIGUOUS couts (one 16-B store each, no
    // v_permlane32_swap pairing).  The v1 kernel keeps the natural order (c.d_w).
    auto perm = [](int n) { return (n & ~12) | ((n & 4) << 1) | ((n & 8) >> 1); };
    if (kind == CONV_RB1 || kind == CONV_RB2 || kind == CONV_UP || kind == CONV_HEAD) {
        std::vector<unsigned short> arrp(arr.size(), 0);
        const size_t rows = arr.size() / ((size_t)c.nt * 8);
        for (size_t rr = 0; rr < rows; ++rr)
            for (int n = 0; n < c.nt; ++n)
                std::memcpy(&arrp[(rr * c.nt + n) * 8], &arr[(rr * c.nt + perm(n)) * 8], 16);
        c.d_wp = (unsigned short*)dalloc(arrp.size() * 2);
        net_.allocs.push_back(c.d_wp);
        IRE_HIP(hipMemcpy(c.d_wp, arrp.data(), arrp.size() * 2, hipMemcpyHostToDevice));
    }
    if (kind == CONV_STEM && cin == 3 && cout == 32) {
        // conv_stem.hip: k-step ky is one tile row; lane (row rho, half h) holds W[perm(rho)] at k = 16 ky + 8 h + e, i.e. kx = 2 h + (e >> 2),
        // c = e & 3 -- zero for the pad positions kx = 3 and c = 3 (the LDS tile holds a pixel as four bf16: R, G, B, 0)
        std::vector<unsigned short> arrs(3 * 2 * 32 * 8, 0);
        for (int ky = 0; ky < 3; ++ky)
            for (int hh = 0; hh < 2; ++hh)
                for (int rho = 0; rho < 32; ++rho)
                    for (int e = 0; e < 8; ++e) {
                        const int kx = 2 * hh + (e >> 2), ch = e & 3;
                        if (kx >= 3 || ch >= 3) continue;
                        arrs[(((size_t)ky * 2 + hh) * 32 + rho) * 8 + e] = f32_to_bf16(W[((size_t)perm(rho) * cin + ch) * 9 + ky * 3 + kx]);
                    }
        c.d_wstem = (unsigned short*)dalloc(arrs.size() * 2);
        net_.allocs.push_back(c.d_wstem);
        IRE_HIP(hipMemcpy(c.d_wstem, arrs.data(), arrs.size() * 2, hipMemcpyHostToDevice));
    }
    if (kind == CONV_DOWN && cin % 32 == 0 && cout % 64 == 0) {
        // conv_down.hip: the stride-2 conv as a unit-stride conv over the four pixel phases P_ab[Y][X] = in[2Y+a][2X+b]:
        // phase (a, b) carries the taps ky in (a ? {0, 2} : {1}) x kx in (b ? {0, 2} : {1}), in that order
        const int nbd = cout / 64, nkd = cin / 32;
        std::vector<unsigned short> arrd((size_t)nbd * nkd * 9 * 4 * 64 * 8, 0);
        size_t pos = 0;
        for (int nb = 0; nb < nbd; ++nb)
            for (int kc = 0; kc < nkd; ++kc)
                for (int ph = 0; ph < 4; ++ph) {
                    const int pa = ph >> 1, pb = ph & 1, nty = pa ? 2 : 1, ntx = pb ? 2 : 1;
                    for (int t = 0; t < nty * ntx; ++t) {
                        const int ty = t / ntx, tx = t % ntx;
                        const int ky = pa ? (ty ? 2 : 0) : 1, kx = pb ? (tx ? 2 : 0) : 1;
                        for (int c8 = 0; c8 < 4; ++c8)
                            for (int n = 0; n < 64; ++n)
                                for (int e = 0; e < 8; ++e) {
                                    const int co = nb * 64 + (n & 32) + perm(n & 31), ci = kc * 32 + c8 * 8 + e;
                                    arrd[pos++] = f32_to_bf16(W[((size_t)co * cin + ci) * 9 + ky * 3 + kx]);
                                }
                    }
                }
        c.d_wd = (unsigned short*)dalloc(arrd.size() * 2);
        net_.allocs.push_back(c.d_wd);
        IRE_HIP(hipMemcpy(c.d_wd, arrd.data(), arrd.size() * 2, hipMemcpyHostToDevice));
        if (cout % 128 == 0) {
            // conv_dnq.hip: the same taps in the same phase order as 128-cout slabs: [n-block of 128][kc32][9 taps][c8][128 permuted rows][8]
            std::vector<unsigned short> arrq((size_t)(cout / 128) * nkd * 9 * 4 * 128 * 8, 0);
            size_t qpos = 0;
            for (int nb = 0; nb < cout / 128; ++nb)
                for (int kc = 0; kc < nkd; ++kc)
                    for (int ph = 0; ph < 4; ++ph) {
                        const int pa = ph >> 1, pb = ph & 1, nty = pa ? 2 : 1, ntx = pb ? 2 : 1;
                        for (int t = 0; t < nty * ntx; ++t) {
                            const int ty = t / ntx, tx = t % ntx;
                            const int ky = pa ? (ty ? 2 : 0) : 1, kx = pb ? (tx ? 2 : 0) : 1;
                            for (int c8 = 0; c8 < 4; ++c8)
                                for (int n = 0; n < 128; ++n)
                                    for (int e = 0; e < 8; ++e) {
                                        const int co = nb * 128 + perm(n), ci = kc * 32 + c8 * 8 + e;
                                        arrq[qpos++] = f32_to_bf16(W[((size_t)co * cin + ci) * 9 + ky * 3 + kx]);
                                    }
                        }
                    }
            if (!d_zero_) { d_zero_ = dalloc(256); IRE_HIP(hipMemset(d_zero_, 0, 256)); }
            c.d_wdq = (unsigned short*)dalloc(arrq.size() * 2);
            net_.allocs.push_back(c.d_wdq);
            IRE_HIP(hipMemcpy(c.d_wdq, arrq.data(), arrq.size() * 2, hipMemcpyHostToDevice));
        }
    }
    if (kind == CONV_UP && cin % 64 == 0 && cout % 32 == 0) {
        // conv_up.hip: nearest x2 -> 3x3 == four 2x2 convolutions on the low-res grid, one per output parity (pa, pb); the taps that
        // land on the same low-res pixel are summed here (fp32), then rounded to bf16:
        //   pa = 0: window row 0 <- ky 0, row 1 <- ky 1 + ky 2;   pa = 1: row 0 <- ky 0 + ky 1, row 1 <- ky 2   (columns alike)
        const int nbu = cout / 32, nku = cin / 32;
        std::vector<unsigned short> arru((size_t)nbu * nku * 4 * 16 * 32 * 8, 0);
        auto lo_of = [](int par, int d) { return par == 0 ? (d == 0 ? 0 : 1) : (d == 0 ? 0 : 2); };
        auto hi_of = [](int par, int d) { return par == 0 ? (d == 0 ? 0 : 2) : (d == 0 ? 1 : 2); };
        for (int nb = 0; nb < nbu; ++nb)
            for (int kc = 0; kc < nku; ++kc)
                for (int par = 0; par < 4; ++par)
                    for (int kk = 0; kk < 16; ++kk) {
                        const int pa = par >> 1, pb = par & 1, tap4 = kk >> 2, c8 = kk & 3, dy = tap4 >> 1, dx = tap4 & 1;
                        for (int n = 0; n < 32; ++n)
                            for (int e = 0; e < 8; ++e) {
                                const int co = nb * 32 + perm(n), ci = kc * 32 + c8 * 8 + e;
                                float sum = 0.f;
                                for (int ky = lo_of(pa, dy); ky <= hi_of(pa, dy); ++ky)
                                    for (int kx = lo_of(pb, dx); kx <= hi_of(pb, dx); ++kx) sum += W[((size_t)co * cin + ci) * 9 + ky * 3 + kx];
                                arru[(((((size_t)nb * nku + kc) * 4 + par) * 16 + kk) * 32 + n) * 8 + e] = f32_to_bf16(sum);
                            }
                    }
        c.d_wu = (unsigned short*)dalloc(arru.size() * 2);
        net_.allocs.push_back(c.d_wu);
        IRE_HIP(hipMemcpy(c.d_wu, arru.data(), arru.size() * 2, hipMemcpyHostToDevice));
    }
    if ((kind == CONV_RB1 || kind == CONV_RB2) && cout >= 128 && cin % 16 == 0 && cout % 128 == 0) {
        // conv_w4.hip slabs: [nblock (128 couts)][kc16][kk = tap*2 + c8][128][8]
        const int nb4 = cout / 128, nk4 = cin / 16;
        std::vector<unsigned short> arr4((size_t)nb4 * nk4 * 18 * 128 * 8, 0);
        for (int nb = 0; nb < nb4; ++nb)
            for (int kc = 0; kc < nk4; ++kc)
                for (int kk = 0; kk < 18; ++kk) {
                    const int tap = kk >> 1, c8 = kk & 1;
                    for (int n = 0; n < 128; ++n)
                        for (int e = 0; e < 8; ++e) {
                            const int co = nb * 128 + perm(n), ci = kc * 16 + c8 * 8 + e;       // permuted rows, as c.d_wp
                            arr4[((((size_t)nb * nk4 + kc) * 18 + kk) * 128 + n) * 8 + e] = f32_to_bf16(W[((size_t)co * cin + ci) * 9 + tap]);
                        }
                }
        c.d_w4 = (unsigned short*)dalloc(arr4.size() * 2);
        net_.allocs.push_back(c.d_w4);
        IRE_HIP(hipMemcpy(c.d_w4, arr4.data(), arr4.size() * 2, hipMemcpyHostToDevice));
        {   // the same slabs in 64-cout blocks: [nblock (64 couts)][kc16][kk][64][8]
            std::vector<unsigned short> arrh(arr4.size(), 0);
            for (int nb = 0; nb < cout / 64; ++nb)
                for (int kc = 0; kc < nk4; ++kc)
                    for (int kk = 0; kk < 18; ++kk) {
                        const int tap = kk >> 1, c8 = kk & 1;
                        for (int n = 0; n < 64; ++n)
                            for (int e = 0; e < 8; ++e) {
                                const int co = nb * 64 + perm(n), ci = kc * 16 + c8 * 8 + e;
                                arrh[((((size_t)nb * nk4 + kc) * 18 + kk) * 64 + n) * 8 + e] = f32_to_bf16(W[((size_t)co * cin + ci) * 9 + tap]);
                            }
                    }
            c.d_w4h = (unsigned short*)dalloc(arrh.size() * 2);
            net_.allocs.push_back(c.d_w4h);
            IRE_HIP(hipMemcpy(c.d_w4h, arrh.data(), arrh.size() * 2, hipMemcpyHostToDevice));
        }
        if (precision_ == IRE_PRECISION_FP8) {
            // the same slabs as OCP e4m3 with one scale per OUTPUT channel: w_q = e4m3(w / s_w[co]), s_w[co] = max|w[co]| / 448
            // (the whole e4m3 range per channel); activations are scaled by kActScale = 16 while staging (conv_w4.hip), so the
            // kernel's accumulator times oscale = s_w / 16 is the conv output and its accumulators start at bias / oscale
            const float kActScale = 16.0f;
            std::vector<float> sw(cout), osc(cout), b8(cout);
            for (int co = 0; co < cout; ++co) {
                float m = 0.f;
                for (size_t k = 0; k < (size_t)cin * 9; ++k) m = std::max(m, std::fabs(W[(size_t)co * cin * 9 + k]));
                sw[co] = m > 0.f ? m / 448.0f : 1.0f;
                osc[co] = sw[co] / kActScale;
                b8[co] = bi->second.second[co] / osc[co];
            }
            std::vector<unsigned char> arr8((size_t)nb4 * nk4 * 18 * 128 * 8, 0);
            for (int nb = 0; nb < nb4; ++nb)
                for (int kc = 0; kc < nk4; ++kc)
                    for (int kk = 0; kk < 18; ++kk) {
                        const int tap = kk >> 1, c8 = kk & 1;
                        for (int n = 0; n < 128; ++n)
                            for (int e = 0; e < 8; ++e) {
                                const int co = nb * 128 + perm(n), ci = kc * 16 + c8 * 8 + e;
                                arr8[((((size_t)nb * nk4 + kc) * 18 + kk) * 128 + n) * 8 + e] = f32_to_e4m3(W[((size_t)co * cin + ci) * 9 + tap] / sw[co]);
                            }
                    }
            // the K = 64 form (conv_f8.hip): 32-channel stages, [tap][16-channel half][128 rows][16 bytes]
            if (cin % 32 == 0) {
                const int nk8 = cin / 32;
                std::vector<unsigned char> arrx((size_t)nb4 * nk8 * 9 * 2 * 128 * 16, 0);
                for (int nb = 0; nb < nb4; ++nb)
                    for (int kc = 0; kc < nk8; ++kc)
                        for (int tap = 0; tap < 9; ++tap)
                            for (int hf = 0; hf < 2; ++hf)
                                for (int n = 0; n < 128; ++n)
                                    for (int e = 0; e < 16; ++e) {
                                        const int co = nb * 128 + perm(n), ci = kc * 32 + hf * 16 + e;
                                        arrx[(((((size_t)nb * nk8 + kc) * 9 + tap) * 2 + hf) * 128 + n) * 16 + e] = f32_to_e4m3(W[((size_t)co * cin + ci) * 9 + tap] / sw[co]);
                                    }
                c.d_w8x = (unsigned char*)dalloc(arrx.size());
                net_.allocs.push_back(c.d_w8x);
                IRE_HIP(hipMemcpy(c.d_w8x, arrx.data(), arrx.size(), hipMemcpyHostToDevice));
            }
            c.d_w8 = (unsigned char*)dalloc(arr8.size());
            c.d_oscale = (float*)dalloc(cout * 4);
            c.d_bias8 = (float*)dalloc(cout * 4);
            net_.allocs.push_back(c.d_w8); net_.allocs.push_back(c.d_oscale); net_.allocs.push_back(c.d_bias8);
            IRE_HIP(hipMemcpy(c.d_w8, arr8.data(), arr8.size(), hipMemcpyHostToDevice));
            IRE_HIP(hipMemcpy(c.d_oscale, osc.data(), cout * 4, hipMemcpyHostToDevice));
            IRE_HIP(hipMemcpy(c.d_bias8, b8.data(), cout * 4, hipMemcpyHostToDevice));
        }
    }
    std::vector<float> bias(cout_pad, 0.f);
    std::memcpy(bias.data(), bi->second.second.data(), sizeof(float) * cout);
    c.d_bias = (float*)dalloc(bias.size() * 4);
    net_.allocs.push_back(c.d_bias);
    IRE_HIP(hipMemcpy(c.d_bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    return c;
}

// `up` (nearest x2 -> 3x3, 2C -> C) followed by `fuse` (1x1 over concat(up, skip), 2C -> C) with nothing non-linear between
// them is ONE convolution plus a 1x1 over the skip tensor:
//   fuse(concat(up(x), skip)) = (Wf_up . Wup) * x_up  +  Wf_skip . skip  +  (Wf_up . b_up + b_f),   Wf = [Wf_up | Wf_skip].
// The composition is done once here in double, then the sub-pixel pre-sums of conv_up.hip, then ONE rounding to bf16.
void Engine::make_up_fused(ConvW& up, const std::string& sl) {
    const int C = up.cout, cin = up.cin;
    if (up.kind != CONV_UP || up.d_wu == nullptr || (C != 32 && C != 64 && C != 128)) return;
    auto wu = host_w_.find("up" + sl + ".w"), bu = host_w_.find("up" + sl + ".b");
    auto wf = host_w_.find("fuse" + sl + ".w"), bf = host_w_.find("fuse" + sl + ".b");
    if (wu == host_w_.end() || bu == host_w_.end() || wf == host_w_.end() || bf == host_w_.end()) return;
    const float* Wu = wu->second.second.data();        // [C][cin][3][3]
    const float* Wf = wf->second.second.data();        // [C][2C]
    if ((int)wf->second.second.size() != C * 2 * C) return;
    std::vector<double> Wc((size_t)C * cin * 9, 0.0);
    for (int co = 0; co < C; ++co)
        for (int m = 0; m < C; ++m) {
            const double f = Wf[(size_t)co * 2 * C + m];
            const float* src = Wu + (size_t)m * cin * 9;
            double* dst = Wc.data() + (size_t)co * cin * 9;
            for (int k = 0; k < cin * 9; ++k) dst[k] += f * (double)src[k];
        }
    std::vector<float> bc(C);
    for (int co = 0; co < C; ++co) {
        double b = bf->second.second[co];
        for (int m = 0; m < C; ++m) b += (double)Wf[(size_t)co * 2 * C + m] * (double)bu->second.second[m];
        bc[co] = (float)b;
    }
    auto perm = [](int n) { return (n & ~12) | ((n & 4) << 1) | ((n & 8) >> 1); };
    const int nbu = C / 32, nku = cin / 32;
    std::vector<unsigned short> arru((size_t)nbu * nku * 4 * 16 * 32 * 8, 0);
    auto lo_of = [](int par, int d) { return par == 0 ? (d == 0 ? 0 : 1) : (d == 0 ? 0 : 2); };
    auto hi_of = [](int par, int d) { return par == 0 ? (d == 0 ? 0 : 2) : (d == 0 ? 1 : 2); };
    for (int nb = 0; nb < nbu; ++nb)
        for (int kc = 0; kc < nku; ++kc)
            for (int par = 0; par < 4; ++par)
                for (int kk = 0; kk < 16; ++kk) {
                    const int pa = par >> 1, pb = par & 1, tap4 = kk >> 2, c8 = kk & 3, dy = tap4 >> 1, dx = tap4 & 1;
                    for (int n = 0; n < 32; ++n)
                        for (int e = 0; e < 8; ++e) {
                            const int co = nb * 32 + perm(n), ci = kc * 32 + c8 * 8 + e;
                            double sum = 0.0;
                            for (int ky = lo_of(pa, dy); ky <= hi_of(pa, dy); ++ky)
                                for (int kx = lo_of(pb, dx); kx <= hi_of(pb, dx); ++kx) sum += Wc[((size_t)co * cin + ci) * 9 + ky * 3 + kx];
                            arru[(((((size_t)nb * nku + kc) * 4 + par) * 16 + kk) * 32 + n) * 8 + e] = f32_to_bf16((float)sum);
                        }
                }
    const int nks = C / 16;
    std::vector<unsigned short> arrs((size_t)nbu * nks * 2 * 32 * 8, 0);
    for (int nb = 0; nb < nbu; ++nb)
        for (int ks = 0; ks < nks; ++ks)
            for (int hh = 0; hh < 2; ++hh)
                for (int n = 0; n < 32; ++n)
                    for (int e = 0; e < 8; ++e)
                        arrs[((((size_t)nb * nks + ks) * 2 + hh) * 32 + n) * 8 + e] =
                            f32_to_bf16(Wf[(size_t)(nb * 32 + perm(n)) * 2 * C + C + ks * 16 + hh * 8 + e]);
    if (C == 128) {
        // conv_upq.hip: the same composed, pre-summed weights (the same single rounding) as 128-cout slabs per output parity, and the
        // skip half as four 32-channel stages
        std::vector<unsigned short> arrq((size_t)4 * nku * 16 * 128 * 8, 0);
        for (int par = 0; par < 4; ++par)
            for (int kc = 0; kc < nku; ++kc)
                for (int kk = 0; kk < 16; ++kk) {
                    const int pa = par >> 1, pb = par & 1, tap4 = kk >> 2, c8 = kk & 3, dy = tap4 >> 1, dx = tap4 & 1;
                    for (int n = 0; n < 128; ++n)
                        for (int e = 0; e < 8; ++e) {
                            const int co = perm(n), ci = kc * 32 + c8 * 8 + e;
                            double sum = 0.0;
                            for (int ky = lo_of(pa, dy); ky <= hi_of(pa, dy); ++ky)
                                for (int kx = lo_of(pb, dx); kx <= hi_of(pb, dx); ++kx) sum += Wc[((size_t)co * cin + ci) * 9 + ky * 3 + kx];
                            arrq[((((size_t)par * nku + kc) * 16 + kk) * 128 + n) * 8 + e] = f32_to_bf16((float)sum);
                        }
                }
        std::vector<unsigned short> arrsq((size_t)(C / 32) * 4 * 128 * 8, 0);
        for (int ks = 0; ks < C / 32; ++ks)
            for (int c8 = 0; c8 < 4; ++c8)
                for (int n = 0; n < 128; ++n)
                    for (int e = 0; e < 8; ++e)
                        arrsq[(((size_t)ks * 4 + c8) * 128 + n) * 8 + e] = f32_to_bf16(Wf[(size_t)perm(n) * 2 * C + C + ks * 32 + c8 * 8 + e]);
        if (!d_zero_) { d_zero_ = dalloc(256); IRE_HIP(hipMemset(d_zero_, 0, 256)); }
        up.d_wuq = (unsigned short*)dalloc(arrq.size() * 2);
        up.d_wsq = (unsigned short*)dalloc(arrsq.size() * 2);
        net_.allocs.push_back(up.d_wuq); net_.allocs.push_back(up.d_wsq);
        IRE_HIP(hipMemcpy(up.d_wuq, arrq.data(), arrq.size() * 2, hipMemcpyHostToDevice));
        IRE_HIP(hipMemcpy(up.d_wsq, arrsq.data(), arrsq.size() * 2, hipMemcpyHostToDevice));
    }
    up.d_wuf = (unsigned short*)dalloc(arru.size() * 2);
    up.d_wsk = (unsigned short*)dalloc(arrs.size() * 2);
    up.d_bias_uf = (float*)dalloc(C * 4);
    net_.allocs.push_back(up.d_wuf); net_.allocs.push_back(up.d_wsk); net_.allocs.push_back(up.d_bias_uf);
    IRE_HIP(hipMemcpy(up.d_wuf, arru.data(), arru.size() * 2, hipMemcpyHostToDevice));
    IRE_HIP(hipMemcpy(up.d_wsk, arrs.data(), arrs.size() * 2, hipMemcpyHostToDevice));
    IRE_HIP(hipMemcpy(up.d_bias_uf, bc.data(), C * 4, hipMemcpyHostToDevice));
}

GNW Engine::make_gn(const std::string& prefix, int C, int level) {
    GNW g;
    g.C = C; g.level = level;
    for (int k = 0; k < 2; ++k) {
        const std::string nm = prefix + (k == 0 ? ".g" : ".b");
        auto it = host_w_.find(nm);
        if (it == host_w_.end() || (int)it->second.second.size() != C)
            fail(IRE_ERR_INVALID_INPUT, "invalid weight file: missing " + nm);
        float* d = (float*)dalloc(sizeof(float) * C);
        net_.allocs.push_back(d);
        IRE_HIP(hipMemcpy(d, it->second.second.data(), sizeof(float) * C, hipMemcpyHostToDevice));
        (k == 0 ? g.d_gamma : g.d_beta) = d;
    }
    return g;
}

RBW Engine::make_rb(const std::string& p, int C, int level) {
    RBW r;
    r.gn1 = make_gn(p + ".gn1", C, level);
    r.conv1 = make_conv(CONV_RB1, p + ".conv1.w", p + ".conv1.b", C, C);
    r.gn2 = make_gn(p + ".gn2", C, level);
    r.conv2 = make_conv(CONV_RB2, p + ".conv2.w", p + ".conv2.b", C, C);
    return r;
}

void Engine::load_weights(const void* blob, size_t bytes) {
    const unsigned char* p = (const unsigned char*)blob;
    auto need = [&](size_t off, size_t n) {
        if (off + n > bytes) fail(IRE_ERR_INVALID_INPUT, "invalid weight file: truncated");
    };
    need(0, 12);
    if (std::memcmp(p, "IREW", 4) != 0) fail(IRE_ERR_INVALID_INPUT, "invalid weight file: bad magic");
    uint32_t ver, nt;
    std::memcpy(&ver, p + 4, 4);
    std::memcpy(&nt, p + 8, 4);
    if (ver != 1 || nt > 4096) fail(IRE_ERR_INVALID_INPUT, "invalid weight file: version");
    size_t off = 12;
    host_w_.clear();
    for (uint32_t i = 0; i < nt; ++i) {
        uint32_t ln, nd;
        need(off, 4); std::memcpy(&ln, p + off, 4); off += 4;
        if (ln > 256) fail(IRE_ERR_INVALID_INPUT, "invalid weight file: name");
        need(off, ln);
        std::string name((const char*)p + off, ln); off += ln;
        while (!name.empty() && name.back() == '\0') name.pop_back();
        need(off, 4); std::memcpy(&nd, p + off, 4); off += 4;
        if (nd > 4) fail(IRE_ERR_INVALID_INPUT, "invalid weight file: ndim");
        std::vector<int> dims(nd);
        size_t cnt = 1;
        for (uint32_t d = 0; d < nd; ++d) {
            uint32_t v; need(off, 4); std::memcpy(&v, p + off, 4); off += 4;
            dims[d] = (int)v; cnt *= v;
        }
        need(off, cnt * 4);
        std::vector<float> data(cnt);
        std::memcpy(data.data(), p + off, cnt * 4); off += cnt * 4;
        host_w_[name] = {dims, std::move(data)};
    }
    IRE_HIP(hipSetDevice(device_));
    IRE_HIP(hipDeviceSynchronize());
    delete tiled_; tiled_ = nullptr;
    for (void* q : net_.allocs) (void)hipFree(q);
    net_ = Net{};
    net_.stem = make_conv(CONV_STEM, "stem.w", "stem.b", 3, 32);
    for (int l = 0; l < 4; ++l) {
        for (int i = 0; i < 2; ++i)
            net_.enc[l][i] = make_rb("enc" + std::to_string(l) + ".rb" + std::to_string(i), kWidths[l], l);
        if (l < 3) net_.down[l] = make_conv(CONV_DOWN, "down" + std::to_string(l) + ".w", "down" + std::to_string(l) + ".b",
                                           kWidths[l], kWidths[l + 1]);
    }
    for (int i = 0; i < 2; ++i) net_.mid[i] = make_rb("mid.rb" + std::to_string(i), 256, 3);
    for (int l = 2; l >= 0; --l) {
        const std::string s = std::to_string(l);
        net_.up[l] = make_conv(CONV_UP, "up" + s + ".w", "up" + s + ".b", kWidths[l + 1], kWidths[l]);
        net_.fuse[l] = make_conv(CONV_FUSE, "fuse" + s + ".w", "fuse" + s + ".b", 2 * kWidths[l], kWidths[l]);
        make_up_fused(net_.up[l], s);
        for (int i = 0; i < 2; ++i) net_.dec[l][i] = make_rb("dec" + s + ".rb" + std::to_string(i), kWidths[l], l);
    }
    net_.head_gn = make_gn("head.gn", 32, 0);
    net_.head = make_conv(CONV_HEAD, "head.w", "head.b", 32, 3);
    {
        auto fw = host_w_.find("film.w"), fb = host_w_.find("film.b");
        if (fw == host_w_.end() || fb == host_w_.end() || (int)fw->second.second.size() != kFilmDim * 7 ||
            (int)fb->second.second.size() != kFilmDim)
            fail(IRE_ERR_INVALID_INPUT, "invalid weight file: film");
        net_.d_film_w = (float*)dalloc(sizeof(float) * kFilmDim * 7);
        net_.d_film_b = (float*)dalloc(sizeof(float) * kFilmDim);
        net_.allocs.push_back(net_.d_film_w);
        net_.allocs.push_back(net_.d_film_b);
        IRE_HIP(hipMemcpy(net_.d_film_w, fw->second.second.data(), sizeof(float) * kFilmDim * 7, hipMemcpyHostToDevice));
        IRE_HIP(hipMemcpy(net_.d_film_b, fb->second.second.data(), sizeof(float) * kFilmDim, hipMemcpyHostToDevice));
    }
    host_w_.clear();
    net_.loaded = true;
    build_program();
}

// ------------------------------------------------------------------------------------------------
// buffers
// ------------------------------------------------------------------------------------------------
void Engine::check_shape(int n, int h, int w, bool for_restore) const {
    if (n <= 0 || n > max_batch_) fail(IRE_ERR_INVALID_INPUT, "invalid batch size n (1..max_batch)");
    if (h <= 0 || w <= 0 || h > 8192 || w > 8192) fail(IRE_ERR_INVALID_INPUT, "invalid image size");
    if (for_restore && (h % 8 || w % 8 || h < 16 || w < 16))
        fail(IRE_ERR_INVALID_INPUT, "invalid image size for restore: height and width must be multiples of 8, >= 16");
}

void Engine::ensure_io(int n, int h, int w) {
    IRE_HIP(hipSetDevice(device_));
    const size_t px = (size_t)h * w;
    if ((size_t)n <= io_cap_imgs_ && px <= io_cap_px_) return;
    IRE_HIP(hipDeviceSynchronize());
    // small shapes: room for a whole batch at once (no regrowth per n); large ones: what was asked for
    const size_t want = px * 3 * (size_t)max_batch_ <= ((size_t)256 << 20) ? (size_t)max_batch_ : (size_t)n;
    const size_t imgs = std::max<size_t>(std::max<size_t>(io_cap_imgs_, want), (size_t)n);
    const size_t cap_px = std::max(io_cap_px_, px);
    for (void* p : {(void*)d_in_, (void*)d_out_, (void*)d_jpeg_, (void*)d_sums_, (void*)d_scores_, (void*)d_label_,
                    (void*)d_cond_, (void*)d_film_})
        if (p) (void)hipFree(p);
    d_in_ = (uint8_t*)dalloc(imgs * cap_px * 3);
    d_out_ = (uint8_t*)dalloc(imgs * cap_px * 3);
    d_jpeg_ = (uint8_t*)dalloc(imgs);
    d_sums_ = (unsigned long long*)dalloc(cls_sums_bytes());   // sums | tickets | workgroup partials (classifier.hpp)
    IRE_HIP(hipMemset(d_sums_, 0, cls_sums_bytes()));
    d_scores_ = (double*)dalloc(imgs * 7 * 8);
    d_label_ = (int32_t*)dalloc(imgs * 4);
    d_cond_ = (float*)dalloc(imgs * 8 * 4);
    d_film_ = (float*)dalloc(imgs * kFilmDim * 4);
    io_cap_imgs_ = imgs;
    io_cap_px_ = cap_px;
}

void Engine::enter(hipStream_t s) {
    IRE_HIP(hipSetDevice(device_));
    if (busy_recorded_) IRE_HIP(hipStreamWaitEvent(s, busy_ev_, 0));
}
void Engine::leave(hipStream_t s) {
    if (hipEventRecord(busy_ev_, s) == hipSuccess) busy_recorded_ = true;
}

void Engine::encode_png_base64_device(const uint8_t* d_rgb, int n, int h, int w, uint8_t* d_chars, size_t stride, hipStream_t s) {
    if (!d_rgb || !d_chars || n < 1 || n > max_batch_) fail(IRE_ERR_INVALID_INPUT, "invalid arguments to the PNG encoder (1..max_batch images)");
    if (h < 1 || w < 8 || w % 8 || h > 16384 || w > 16384) fail(IRE_ERR_INVALID_INPUT, "invalid image size for the PNG encoder: width must be a multiple of 8");
    if (stride < png_base64_chars(h, w)) fail(IRE_ERR_INVALID_INPUT, "invalid stride for the PNG encoder: smaller than ire_png_base64_bytes(h, w)");
    const size_t per = (png_scratch_bytes(h, w) + 255) / 256 * 256, need = per * (size_t)n;
    if (need > enc_scratch_cap_) {
        IRE_HIP(hipDeviceSynchronize());
        if (d_enc_scratch_) IRE_HIP(hipFree(d_enc_scratch_));
        d_enc_scratch_ = nullptr; enc_scratch_cap_ = 0;
        d_enc_scratch_ = (uint8_t*)dalloc(need);
        enc_scratch_cap_ = need;
    }
    // (the tickets must start at zero; the kernels leave them at zero: one memset per geometry change would do, one per call is 3 us)
    IRE_HIP(hipMemsetAsync(d_enc_scratch_, 0, need, s));
    for (int i = 0; i < n; ++i)
        encode_png_base64_launch(d_rgb + (size_t)i * h * w * 3, h, w, d_enc_scratch_ + per * i, d_chars + stride * i, s);
}

void Engine::encode_png_base64_host(const uint8_t* rgb, int n, int h, int w, uint8_t* chars, size_t stride) {
    if (!rgb || !chars) fail(IRE_ERR_INVALID_INPUT, "invalid arguments to the PNG encoder: null buffer");
    if (n < 1 || n > max_batch_ || h < 1 || w < 8 || w % 8 || h > 16384 || w > 16384) fail(IRE_ERR_INVALID_INPUT, "invalid arguments to the PNG encoder (1..max_batch images, width a multiple of 8)");
    const size_t ib = (size_t)h * w * 3, cb = png_base64_chars(h, w), cpad = (cb + 255) / 256 * 256;
    const size_t need = (ib + cpad) * (size_t)n;
    if (need > enc_io_cap_) {
        IRE_HIP(hipDeviceSynchronize());
        if (d_enc_io_) IRE_HIP(hipFree(d_enc_io_));
        d_enc_io_ = nullptr; enc_io_cap_ = 0;
        d_enc_io_ = (uint8_t*)dalloc(need);
        enc_io_cap_ = need;
    }
    hipStream_t s = main_stream_;
    IRE_HIP(hipMemcpyAsync(d_enc_io_, rgb, ib * n, hipMemcpyHostToDevice, s));
    uint8_t* d_txt = d_enc_io_ + ib * n;
    encode_png_base64_device(d_enc_io_, n, h, w, d_txt, cpad, s);
    for (int i = 0; i < n; ++i) IRE_HIP(hipMemcpyAsync(chars + stride * i, d_txt + cpad * i, cb, hipMemcpyDeviceToHost, s));
    IRE_HIP(hipStreamSynchronize(s));
}

void Engine::free_workspace() {
    for (void* p : ws_allocs_) (void)hipFree(p);
    ws_allocs_.clear();
    for (auto& L : lanes_) {
        std::memset(L.act, 0, sizeof(L.act));
        std::memset(L.skip, 0, sizeof(L.skip));
        L.stats = L.stats2 = nullptr;
        L.ab = nullptr;
    }
    ws_imgs_per_lane_ = ws_imgs_cap_ = ws_h_ = ws_w_ = 0;
    ws_bytes_ = 0;
}

// activation workspace (engine.cpp::ensure_workspace) + the staging of the host entry points (ensure_io) of ONE image
size_t Engine::bytes_per_image(int h, int w) const {
    size_t b = 0;
    for (int l = 0; l < 4; ++l) {
        const size_t t = (size_t)(h >> l) * (w >> l) * kWidths[l] * 2;
        b += t * (4 + (l < 3 ? 1 : 0));
    }
    b += 2 * (size_t)ceil_div(h, 4) * ceil_div(w, 32) * 16 * 4 + 256 * sizeof(float2);
    b += (size_t)h * w * 3 * 2;      // d_in_ / d_out_
    return b;
}

int Engine::capacity_for(int h, int w) const {
    if (h <= 0 || w <= 0 || h > 8192 || w > 8192 || h % 8 || w % 8 || h < 16 || w < 16) return 0;
    size_t free_b = 0, total_b = 0;
    if (hipSetDevice(device_) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    // what this engine already holds for activations would be released on a change of shape
    const double avail = 0.92 * ((double)free_b + (double)ws_bytes_);
    const double n = avail / (double)bytes_per_image(h, w);
    return n >= (double)max_batch_ ? max_batch_ : (int)n;
}

void Engine::ensure_workspace(int n, int h, int w) {
    // sized by the batch actually asked for (grow-only per shape), not by max_batch: one 8192^2 image must not need 64 slots
    if (h == ws_h_ && w == ws_w_ && n <= ws_imgs_cap_) return;
    const int cap = (h == ws_h_ && w == ws_w_) ? std::max(n, ws_imgs_cap_) : n;
    const int per = ceil_div(cap, num_lanes_);
    IRE_HIP(hipDeviceSynchronize());
    free_workspace();
    auto alloc = [&](size_t bytes) { void* p = dalloc(bytes); ws_allocs_.push_back(p); ws_bytes_ += bytes; return p; };
    try {
        for (auto& L : lanes_) {
            for (int l = 0; l < 4; ++l) {
                const size_t bytes = (size_t)per * (h >> l) * (w >> l) * kWidths[l] * 2;
                for (int b = 0; b < 4; ++b) L.act[l][b] = (unsigned short*)alloc(bytes);
                if (l < 3) L.skip[l] = (unsigned short*)alloc(bytes);
            }
            const size_t tiles0 = (size_t)ceil_div(h, 4) * ceil_div(w, 32);
            L.stats = (float*)alloc((size_t)per * tiles0 * 16 * 4);
            L.stats2 = (float*)alloc((size_t)per * tiles0 * 16 * 4);
            L.ab = (float2*)alloc((size_t)per * 256 * sizeof(float2));
        }
    } catch (...) {
        free_workspace();     // a partial allocation must not pin HBM: the caller gets "service unavailable" and may retry smaller
        throw;
    }
    ws_imgs_per_lane_ = per; ws_imgs_cap_ = per * num_lanes_; ws_h_ = h; ws_w_ = w;
}

// ------------------------------------------------------------------------------------------------
// profiler (HIP events on the stream the kernel is launched on)
// ------------------------------------------------------------------------------------------------
void Engine::prof_begin(int fam, hipStream_t s, double flops, double bytes) {
    prof_open_ = prof_on_ == 1 || (prof_on_ == 2 && fam == FAM_CONV3 && !prof_skip_);   // mode 2: dominant family only (fewer events), every prof_every_-th network pass
    if (!prof_open_) { prof_chain_ = false; return; }
    ProfRec r;
    r.fam = fam; r.flops = flops; r.bytes = bytes; r.flops_exec = flops;
    auto get = [&]() {
        hipEvent_t ev;
        if (!ev_pool_.empty()) { ev = ev_pool_.back(); ev_pool_.pop_back(); }
        else IRE_HIP(hipEventCreate(&ev));
        return ev;
    };
    // An event record is a packet of its own on the stream (~4 us between two kernels).  When the previous profiled launch's
    // end event is the stream's last operation, this launch starts its interval there: one event per boundary instead of two.
    if (prof_chainable_ && prof_chain_ && prof_chain_stream_ == s && !prof_.empty()) { r.e0 = prof_.back().e1; r.own_e0 = false; }
    else { r.e0 = get(); IRE_HIP(hipEventRecord(r.e0, s)); }
    r.e1 = get();
    prof_.push_back(r);
}
void Engine::prof_tag(const std::string& key, const char* kernel, int level, int cin, int cout, double flops_exec) {
    if (!prof_open_) return;
    int gi = -1;
    for (size_t i = 0; i < prof_groups_.size(); ++i) if (prof_groups_[i].key == key) { gi = (int)i; break; }
    if (gi < 0) { ProfGroup g; g.key = key; g.kernel = kernel; g.level = level; g.cin = cin; g.cout = cout; prof_groups_.push_back(g); gi = (int)prof_groups_.size() - 1; }
    prof_.back().group = gi; prof_.back().flops_exec = flops_exec;
}
void Engine::prof_end(hipStream_t s) {
    if (!prof_open_) return;
    IRE_HIP(hipEventRecord(prof_.back().e1, s));
    prof_chain_ = true; prof_chain_stream_ = s;
}
void Engine::prof_collect() {
    prof_chain_ = false;
    if (prof_.empty()) return;
    IRE_HIP(hipDeviceSynchronize());
    for (auto& r : prof_) {
        float ms = 0.f;
        IRE_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        prof_ms_[r.fam] += ms; prof_flops_[r.fam] += r.flops; prof_bytes_[r.fam] += r.bytes; prof_n_[r.fam] += 1;
        prof_flops_exec_[r.fam] += r.flops_exec;
        if (r.group >= 0) { ProfGroup& g = prof_groups_[r.group]; g.ms += ms; g.flops += r.flops; g.flops_exec += r.flops_exec; g.bytes += r.bytes; g.n += 1; }
        if (r.own_e0) ev_pool_.push_back(r.e0);
        ev_pool_.push_back(r.e1);
    }
    prof_.clear();
}
// mode: low byte 0 off / 1 every family / 2 the 3x3 conv family only; bits 8.. = N: in mode 2 time only every N-th pass of the network
// (0, 1: every pass).  An event record is a packet of its own between two kernels: 39 per step cost 1.35 % of a 1024^2 bs 8 step.
void Engine::profile_enable(int mode) {
    prof_collect();
    prof_on_ = mode & 0xff;
    prof_every_ = std::max(1, mode >> 8);
    prof_pass_ = 0; prof_skip_ = false;
}
void Engine::profile_reset() {
    prof_collect();
    for (int i = 0; i < FAM_COUNT; ++i) { prof_ms_[i] = prof_flops_[i] = prof_bytes_[i] = prof_flops_exec_[i] = 0; prof_n_[i] = 0; }
    prof_groups_.clear();
}
std::string Engine::profile_report() {
    prof_collect();
    std::string out = "[";
    char buf[512];
    for (size_t i = 0; i < prof_groups_.size(); ++i) {
        const ProfGroup& g = prof_groups_[i];
        std::snprintf(buf, sizeof buf, "%s{\"group\": \"%s\", \"kernel\": \"%s\", \"level\": %d, \"cin\": %d, \"cout\": %d, \"launches\": %lld, "
                      "\"ms\": %.6f, \"flops\": %.6e, \"flops_executed\": %.6e, \"bytes\": %.6e}",
                      i ? ", " : "", g.key.c_str(), g.kernel.c_str(), g.level, g.cin, g.cout, (long long)g.n, g.ms, g.flops, g.flops_exec, g.bytes);
        out += buf;
    }
    return out + "]";
}
void Engine::profile_query(int fam, double* ms, int64_t* launches, double* flops, double* bytes) {
    prof_collect();
    double m = 0, f = 0, b = 0; int64_t n = 0;
    for (int i = 0; i < FAM_COUNT; ++i)
        if (fam < 0 || fam == i) { m += prof_ms_[i]; f += prof_flops_[i]; b += prof_bytes_[i]; n += prof_n_[i]; }
    if (ms) *ms = m;
    if (launches) *launches = n;
    if (flops) *flops = f;
    if (bytes) *bytes = b;
}

// ------------------------------------------------------------------------------------------------
// debug capture
// ------------------------------------------------------------------------------------------------
void Engine::capture(const char* name, const unsigned short* d, size_t count, hipStream_t s) {
    if (!capture_ || !name) return;
    IRE_HIP(hipStreamSynchronize(s));
    std::vector<unsigned short> hbuf(count);
    IRE_HIP(hipMemcpy(hbuf.data(), d, count * 2, hipMemcpyDeviceToHost));
    std::vector<float> f(count);
    for (size_t i = 0; i < count; ++i) {
        uint32_t u = (uint32_t)hbuf[i] << 16;
        std::memcpy(&f[i], &u, 4);
    }
    captured_[name] = std::move(f);
}
bool Engine::debug_activation(const std::string& name, float* out, size_t* count) {
    auto it = captured_.find(name);
    if (it == captured_.end()) return false;
    if (count) *count = it->second.size();
    if (out) std::memcpy(out, it->second.data(), it->second.size() * sizeof(float));
    return true;
}
void Engine::debug_sums(int n, uint64_t* out) {
    if (n <= 0 || (size_t)n > io_cap_imgs_ || !d_sums_) fail(IRE_ERR_INVALID_INPUT, "invalid n for debug sums");
    IRE_HIP(hipDeviceSynchronize());
    IRE_HIP(hipMemcpy(out, d_sums_, sizeof(uint64_t) * 14 * n, hipMemcpyDeviceToHost));
}

// ------------------------------------------------------------------------------------------------
// RestoreNet-v0 schedule: a program of ops (built once per weight load), executed op by op
// ------------------------------------------------------------------------------------------------
void Engine::build_program() {
    program_.clear();
    const bool v2 = rb_tile_h_ == kRbTileH;
    auto gn = [&](const GNW& g) { Op o; o.kind = Op::GN; o.gn = &g; program_.push_back(o); };
    auto conv = [&](const ConvW& cw, int in0, int in1, int resid, int out, int lin, int lout, bool use_ab, const std::string& name) {
        Op o; o.kind = Op::CONV; o.cw = &cw; o.in0 = in0; o.in1 = in1; o.resid = resid; o.out = out; o.lin = lin; o.lout = lout;
        o.use_ab = use_ab; o.name = name;
        const bool upf = cw.kind == CONV_UP && in1 != BUF_NONE;           // `up` composed with `fuse`: its output is what fuse's was
        o.halo_out = (cw.kind != CONV_UP || upf) && cw.kind != CONV_HEAD;   // a plain `up` feeds the 1x1 fuse only, the head writes pixels
        o.stats_out = (cw.kind != CONV_UP || upf) && cw.kind != CONV_HEAD;
        program_.push_back(o);
    };
    // ResBlock: out = x + conv2(silu(gn2(conv1(silu(gn1(x)))))); the partials of x were written by x's producer
    auto resblock = [&](const RBW& rb, int x, int tmp, int out, int l, const std::string& name) {
        // GroupNorm+FiLM+SiLU is applied by the consuming conv while it stages its input (a separate activation pass lost every
        // A/B from round 1 to round 3 and was removed in round 4)
        gn(rb.gn1);
        conv(rb.conv1, x, BUF_NONE, BUF_NONE, tmp, l, l, true, name + ".h");
        gn(rb.gn2);
        conv(rb.conv2, tmp, BUF_NONE, x, out, l, l, true, name);
    };
    conv(net_.stem, BUF_NONE, BUF_NONE, BUF_NONE, buf_id(0, 0), 0, 0, false, "stem");
    int x = buf_id(0, 0);
    for (int l = 0; l < 4; ++l) {
        const int rb1_out = (l < 3) ? buf_id(l, 4) : buf_id(l, 3);
        resblock(net_.enc[l][0], x, buf_id(l, 1), buf_id(l, 2), l, "enc" + std::to_string(l) + ".rb0");
        resblock(net_.enc[l][1], buf_id(l, 2), buf_id(l, 1), rb1_out, l, "enc" + std::to_string(l) + ".rb1");
        if (l < 3) {
            conv(net_.down[l], rb1_out, BUF_NONE, BUF_NONE, buf_id(l + 1, 0), l, l + 1, false, "down" + std::to_string(l));
            x = buf_id(l + 1, 0);
        }
    }
    resblock(net_.mid[0], buf_id(3, 3), buf_id(3, 1), buf_id(3, 0), 3, "mid.rb0");
    resblock(net_.mid[1], buf_id(3, 0), buf_id(3, 1), buf_id(3, 2), 3, "mid.rb1");
    int deep = buf_id(3, 2);
    for (int l = 2; l >= 0; --l) {
        const std::string sl = std::to_string(l);
        const bool upf = v2 && up_fuse_ && up_subpixel_ && net_.up[l].d_wuf != nullptr && net_.up[l].cout >= up_rb_min_c_;
        if (upf) {
            conv(net_.up[l], deep, buf_id(l, 4), BUF_NONE, buf_id(l, 2), l + 1, l, false, "fuse" + sl);     // one kernel, the `up` tensor never exists
        } else {
            conv(net_.up[l], deep, BUF_NONE, BUF_NONE, buf_id(l, 0), l + 1, l, false, "up" + sl);
            conv(net_.fuse[l], buf_id(l, 0), buf_id(l, 4), BUF_NONE, buf_id(l, 2), l, l, false, "fuse" + sl);
        }
        resblock(net_.dec[l][0], buf_id(l, 2), buf_id(l, 1), buf_id(l, 3), l, "dec" + sl + ".rb0");
        resblock(net_.dec[l][1], buf_id(l, 3), buf_id(l, 1), buf_id(l, 0), l, "dec" + sl + ".rb1");
        deep = buf_id(l, 0);
    }
    gn(net_.head_gn);
    conv(net_.head, deep, BUF_NONE, BUF_NONE, BUF_NONE, 0, 0, true, "");
}

Geo Engine::geo_of_lane(const Lane& L, int nimg, int h, int w, const uint8_t* d_in, uint8_t* d_out) {
    Geo g;
    g.nimg = nimg; g.h = h; g.w = w; g.H = h;
    for (int l = 0; l < 4; ++l) {
        for (int b = 0; b < 4; ++b) g.buf[l][b] = L.act[l][b];
        g.buf[l][4] = l < 3 ? L.skip[l] : nullptr;
        g.buf[l][5] = nullptr;
    }
    g.img_in = d_in; g.img_out = d_out;
    return g;
}

void Engine::exec_conv(Run& R, const Op& op, const Geo& g) {
    const ConvW& cw = *op.cw;
    const int Hin = g.h >> op.lin, Win = g.w >> op.lin, Hout = g.h >> op.lout, Wout = g.w >> op.lout;
    // inputs are addressed from the buffer start (halo row included: in_row_off), outputs / residual from the first real row
    auto in_ptr = [&](int id) -> const unsigned short* { return id == BUF_NONE ? nullptr : g.buf[id >> 3][id & 7]; };
    auto out_ptr = [&](int id) -> unsigned short* {
        if (id == BUF_NONE) return nullptr;
        const int l = id >> 3;
        return g.buf[l][id & 7] + (size_t)g.halo * (g.w >> l) * kWidths[l];
    };
    ConvArgs a{};
    a.in0 = (cw.kind == CONV_STEM) ? (const void*)g.img_in : (const void*)in_ptr(op.in0);
    a.in1 = in_ptr(op.in1);
    a.cin0 = cw.cin0; a.cin1 = cw.cin1; a.kc_split = cw.kc_split; a.nkc = cw.nkc;
    a.w = cw.d_w; a.bias = cw.d_bias; a.ab = op.use_ab ? R.ab : nullptr; a.resid = out_ptr(op.resid); a.out = out_ptr(op.out);
    a.u8_in = cw.kind == CONV_HEAD ? g.img_in + (size_t)g.halo * g.w * 3 : nullptr;
    a.u8_out = cw.kind == CONV_HEAD ? g.img_out : nullptr;
    a.Hin = Hin; a.Win = Win; a.Hout = Hout; a.Wout = Wout;
    a.in_rows = Hin + 2 * g.halo; a.in_row_off = g.halo;
    {
        const int HV = cw.kind == CONV_UP ? 2 * Hin : Hin;      // virtual input rows (nearest x2 folded into the staging)
        a.iy_lo = (g.halo && g.has_up) ? -1 : 0;
        a.iy_span = HV + ((g.halo && g.has_down) ? 1 : 0) - a.iy_lo;
    }
    a.cout = (cw.kind == CONV_HEAD) ? 32 : cw.cout;
    a.tiles_x = ceil_div(Wout, 32);
    const bool rb = (cw.kind == CONV_RB1 || cw.kind == CONV_RB2);
    const bool up_rb = (cw.kind == CONV_UP) && rb_tile_h_ == kRbTileH && cw.cout >= up_rb_min_c_;
    const bool up_sub = up_rb && up_subpixel_ && cw.d_wu != nullptr;       // sub-pixel form: tiles and halo rows on the LOW-res grid
    const bool up_fused = up_sub && op.in1 != BUF_NONE;                     // composed with the 1x1 `fuse` (build_program)
    // cout = 128: parity-major items with all 128 couts (conv_upq.hip); four partial rows per low-res tile (fp8 engines too: their `up` and
    // `down` convs stay bf16).  The cout = 64 level stays on conv_up.hip: its row-parity form of this kernel measured 278 us against 241
    // (profiles/r04_experiments.md).  A function of the layer only: batch / strip invariance holds.
    const bool up_q = up_fused && use_upq_ && cw.d_wuq != nullptr && cw.cout == 128 && cw.cin % 32 == 0;
    const int parts_mul = up_q ? 4 : 1;     // partial rows per low-res tile: one per item (parity)
    const bool head_rb = cw.kind == CONV_HEAD && rb_tile_h_ == kRbTileH && head_rb_ && cw.d_wp != nullptr;    // the head on the pipelined kernel
    const bool down_rb = cw.kind == CONV_DOWN && rb_tile_h_ == kRbTileH && down_rb_ && cw.d_wd != nullptr;    // stride-2 convs by pixel phase
    const bool stem_rb = cw.kind == CONV_STEM && rb_tile_h_ == kRbTileH && stem_rb_ && cw.d_wstem != nullptr && op.stats_out;  // the stem on its own kernel
    const int th = (rb || up_rb || head_rb || down_rb || stem_rb) ? rb_tile_h_ : conv_tile_h(cw.kind);
    a.tiles_y = ceil_div(Hout, th);
    if (up_sub) {
        a.tiles_x = ceil_div(Win, 32); a.tiles_y = ceil_div(Hin, 16);
        a.iy_lo = (g.halo && g.has_up) ? -1 : 0;
        a.iy_span = Hin + ((g.halo && g.has_down) ? 1 : 0) - a.iy_lo;
    }
    a.stats = nullptr;
    if (op.stats_out) {
        // partials are indexed by the GLOBAL tile: a strip writes its tiles at its offset (strip starts are multiples of the
        // tile height at every level: checked by the strip planner), so the finalize sees exactly the whole-image layout
        // (the fused `up` writes one partial per LOW-res tile: its items are 32 x 64 output pixels)
        const int sl = up_fused ? op.lin : op.lout;
        const int ty0 = (g.y0 >> sl) / th;
        float* dst = R.stats_alt ? R.stats_alt : R.stats;         // ping-pong: this conv may still be reading R.stats in its folded finalize
        a.stats = dst + (size_t)ty0 * a.tiles_x * 16 * parts_mul;
    }
    a.nimg = g.nimg; a.nblocks = cw.nblocks;
    a.group_size = std::max(1, a.cout / 8);
    a.stamps = nullptr;
    a.prio_young = prio_young_;
    if (stamps_dev_ && rb && cw.cout == stamps_cout_ && (cw.kind == CONV_RB2) == stamps_resid_ && (!stamps_taken_ || !stamps_tl_.empty() || std::getenv("IRE_STAMPS_RAW"))) {
        a.stamps = stamps_dev_;
        stamps_taken_ = true;
    }
    const int taps = (cw.kind == CONV_FUSE) ? 1 : 9;
    const double px = (double)g.nimg * Hout * Wout;
    double flops = 2.0 * taps * cw.cin * cw.cout * px;
    const double in_px = (double)g.nimg * Hin * Win;
    double bytes = in_px * cw.cin * (cw.kind == CONV_STEM ? 1 : 2) + px * cw.cout * (cw.kind == CONV_HEAD ? 1 : 2);
    if (up_fused) { flops += 2.0 * 2 * cw.cout * cw.cout * px; bytes += px * cw.cout * 2; }   // the algorithmic work of `fuse` rides along: 1x1 over 2C channels, the skip tensor read
    if (cw.kind == CONV_RB2) bytes += px * cw.cout * 2;
    if (cw.kind == CONV_HEAD) bytes += px * 3;
    int fam = FAM_CONV3;
    if (cw.kind == CONV_FUSE) fam = FAM_CONV1;
    else if (cw.kind == CONV_STEM) fam = FAM_STEM;
    else if (cw.kind == CONV_HEAD) fam = FAM_HEAD;
    // conv_w4: the C >= 128 ResBlock convs, activation fused into its staging
    const bool w4 = rb && rb_tile_h_ == kRbTileH && use_w4_ && cw.d_w4 != nullptr && a.ab != nullptr && cw.cout >= 128;
    if (R.gn_pending) {
        // the deferred GroupNorm finalize of this conv's input: inside the kernel's prologue where it has one (gn_fold.hpp)
        const bool folds = a.ab != nullptr && (w4 || head_rb || (rb && rb_tile_h_ == kRbTileH));
        if (folds) {
            const GNW& gn = *R.gn_pending;
            a.gn_stats = R.gn_stats; a.gn_parts = R.gn_parts; a.gn_hw = (g.H >> gn.level) * (g.w >> gn.level);
            a.gn_gamma = gn.d_gamma; a.gn_beta = gn.d_beta; a.gn_film = R.film; a.gn_film_stride = kFilmDim; a.gn_film_off = kFilmOff[gn.level];
            a.ab_w = R.ab;
            R.gn_pending = nullptr;
        } else flush_gn(R, g);
    }
    prof_begin(fam, R.stream, flops, bytes);
    const char* kname = "conv_mfma";       // which kernel takes this launch (the branches below)
    if (w4 && fp8_mx_ && cw.d_w8x != nullptr && a.ab != nullptr) {   // IRE_PRECISION_FP8: the 2x-rate block-scaled fp8 MFMA
        a.fp8 = 1; a.w = reinterpret_cast<const unsigned short*>(cw.d_w8x); a.bias = cw.d_bias8; a.oscale = cw.d_oscale;
        a.nkc = cw.cin / 32; a.nblocks = cw.cout / 128;
        conv_f8_launch(cw.kind == CONV_RB2, a, R.stream); kname = "conv_f8";
    } else if (w4) {
        a.w = cw.d_w4; a.nkc = cw.cin / 16; a.nblocks = cw.cout / 128;
        {   // 64-cout items where 128-cout ones would leave CUs idle (512^2 at level 3): twice the items, each half the MFMAs.  The choice
            // is a function of the IMAGE's shape at this level only -- not of the batch size, not of the strip -- because the two forms add
            // the GroupNorm partials of a tile in different fp32 orders: a result must not depend on the batch around it or on the strip
            // decomposition (tests: batch invariance, tiled == untiled).  Rule: a batch of 8 such images would not fill the CUs.
            const int tiles_img = ceil_div(g.H >> op.lout, kRbTileH) * ceil_div(g.w >> op.lout, 32);
            if (w4_split_ && cw.d_w4h && a.ab != nullptr && cw.d_w8 == nullptr && tiles_img * a.nblocks * 8 < 256) {
                a.w = cw.d_w4h; a.nblocks = cw.cout / 64; a.w4_nt = 64;
            }
        }
        if (cw.d_w8 != nullptr && a.ab != nullptr) {      // IRE_PRECISION_FP8: e4m3 operands for the C >= 128 ResBlock convs
            a.fp8 = 1; a.w = reinterpret_cast<const unsigned short*>(cw.d_w8); a.bias = cw.d_bias8; a.oscale = cw.d_oscale;
        }
        // the producer / consumer form (conv_pk.hip) takes the 128-cout bf16 launches with a fused activation whose workgroups stay
        // within its coefficient table; same slabs, bit-identical results
        // (use_pk_ 1: the convs without a residual; 2: all of them -- the residual variant's epilogue fetches its 128 KB of residual rows per item
        //  in one burst with too few registers to wait in: 208 vs conv_w4's 181 us at C = 128, profiles/r04_experiments.md)
        if (use_pk_ && (use_pk_ >= 2 || cw.kind != CONV_RB2) && !a.fp8 && a.ab != nullptr && a.w4_nt != 64 && cw.cin == cw.cout &&
            conv_pk_fits(cw.cout, a.tiles_x * a.tiles_y, g.nimg)) {
            conv_pk_launch(cw.kind == CONV_RB2, a, R.stream); kname = "conv_pk";
        } else {
            conv_w4_launch(cw.kind == CONV_RB2, a, R.stream); kname = "conv_w4";
        }
    } else if (head_rb) { a.w = cw.d_wp; if (pc_split_ & 1) { conv_pc_launch(false, true, a, R.stream); kname = "conv_pc"; } else { conv_head_launch(a, R.stream); kname = "conv_rb"; } }
    else if (down_rb && use_dnq_ && cw.d_wdq != nullptr) {
        a.w = cw.d_wdq; a.nkc = cw.cin / 32; a.nblocks = cw.cout / 128; a.zeros = d_zero_;
        conv_dnq_launch(a, R.stream); kname = "conv_dnq";
    }
    else if (down_rb) { a.w = cw.d_wd; a.nkc = cw.cin / 32; a.nblocks = cw.cout / 64; conv_down_launch(a, R.stream); kname = "conv_down"; }
    else if (stem_rb) { a.w = cw.d_wstem; conv_stem_launch(a, R.stream); kname = "conv_stem"; }
    else if (up_q) {
        a.w = cw.d_wuq; a.w1 = cw.d_wsq; a.bias = cw.d_bias_uf; a.nkc = cw.cin / 32; a.nblocks = parts_mul;
        a.in1 = in_ptr(op.in1) + (size_t)g.halo * Wout * cw.cout;
        a.cin1 = cw.cout; a.zeros = d_zero_;
        conv_upq_launch(a, R.stream); kname = "conv_upq";
    }
    else if (up_fused) {
        a.w = cw.d_wuf; a.w1 = cw.d_wsk; a.bias = cw.d_bias_uf; a.nkc = cw.cin / 32; a.nblocks = cw.cout / 32;
        a.in1 = in_ptr(op.in1) + (size_t)g.halo * Wout * cw.cout;      // the skip tensor is read at output pixels only: first real row
        a.cin1 = cw.cout;
        conv_up_subpixel_launch(a, R.stream); kname = "conv_up";
    }
    else if (up_sub) { a.in1 = nullptr; a.w = cw.d_wu; a.nkc = cw.cin / 32; a.nblocks = cw.cout / 32; conv_up_subpixel_launch(a, R.stream); kname = "conv_up"; }
    else if (up_rb) { if (cw.d_wp) a.w = cw.d_wp; conv_up_launch(a, R.stream); kname = "conv_rb"; }
    else if (rb && rb_tile_h_ == kRbTileH) {
        if (cw.d_wp) a.w = cw.d_wp;
        const bool pc = a.ab != nullptr && cw.d_wp && cw.cin == cw.cout &&
                        ((cw.cout == 32 && (pc_split_ & 1)) || (cw.cout == 64 && (pc_split_ & 2))) && conv_pc_fits(cw.cout, a.tiles_x * a.tiles_y, g.nimg);
        if (pc) { conv_pc_launch(cw.kind == CONV_RB2, false, a, R.stream); kname = "conv_pc"; }
        else { conv_rb_launch(cw.kind == CONV_RB2, /*fused_act=*/a.ab != nullptr, a, R.stream); kname = "conv_rb"; }
    }
    else conv_launch(cw.kind, a, R.stream);
    {
        // layer group of this launch (bench.py roofline.per_level) and the flops the kernel really issues: the sub-pixel `up`
        // form runs 4 of the 9 taps, its composed `fuse` only the skip half of the 1x1 (the up half is folded into the weights)
        double fexec = flops;
        if (up_fused) fexec = 2.0 * 4 * cw.cin * cw.cout * px + 2.0 * cw.cout * cw.cout * px;
        else if (up_sub) fexec = 2.0 * 4 * cw.cin * cw.cout * px;
        std::string key;
        switch (cw.kind) {
            case CONV_RB1: key = "L" + std::to_string(op.lout) + ".rb1"; break;
            case CONV_RB2: key = "L" + std::to_string(op.lout) + ".rb2"; break;
            case CONV_DOWN: key = "down" + std::to_string(op.lin); break;
            case CONV_UP: key = "up" + std::to_string(op.lout); break;
            case CONV_FUSE: key = "fuse" + std::to_string(op.lout); break;
            case CONV_STEM: key = "stem"; break;
            case CONV_HEAD: key = "head"; break;
            default: key = "conv"; break;
        }
        prof_tag(key, kname, op.lout, cw.cin, cw.cout, fexec);
    }
    prof_end(R.stream);
    if (op.stats_out) {
        const int sl = up_fused ? op.lin : op.lout;
        R.stat_parts = a.tiles_x * ceil_div(g.H >> sl, th) * parts_mul;
        if (R.stats_alt) std::swap(R.stats, R.stats_alt);          // R.stats = the partials produced last
    }
    if (capture_ && !op.name.empty() && a.out && g.halo == 0) capture(op.name.c_str(), a.out, (size_t)g.nimg * Hout * Wout * cw.cout, R.stream);
}

void Engine::flush_gn(Run& R, const Geo& g) {
    const GNW& gn = *R.gn_pending;
    prof_begin(FAM_GN, R.stream, 0, 0);
    gn_finalize_launch(R.gn_stats, g.nimg, R.gn_parts, gn.C, (g.H >> gn.level) * (g.w >> gn.level), gn.d_gamma, gn.d_beta, R.film, kFilmDim,
                       kFilmOff[gn.level], R.ab, R.stream);
    prof_end(R.stream);
    R.gn_pending = nullptr;
}

void Engine::exec_op(Run& R, const Op& op, const Geo& g) {
    switch (op.kind) {
        case Op::GN: {
            // the partials of the tensor produced last; finalized by the consumer itself (exec_conv) unless that is switched off
            // or this is a row strip (one finalize over the gathered array serves all strips)
            R.gn_pending = op.gn; R.gn_stats = R.stats; R.gn_parts = R.stat_parts;
            if (!gn_fold_ || g.halo) flush_gn(R, g);
            break;
        }
        case Op::CONV: exec_conv(R, op, g); break;
    }
}

void Engine::run_network(Lane& L, int nimg, int h, int w, const uint8_t* d_in, uint8_t* d_out, const float* d_film) {
    Run R;
    R.stream = L.stream; R.stats = L.stats; R.stats_alt = L.stats2; R.ab = L.ab; R.film = d_film;
    const Geo g = geo_of_lane(L, nimg, h, w, d_in, d_out);
    // inside one pass of the op list a profiled launch's end event is the next one's start (prof_begin): nothing but the
    // engine's own wrapped launches goes onto the stream here.  Everywhere else (copies, host syncs between calls) records
    // keep their own start event.
    prof_chain_ = false;
    prof_chainable_ = capture_ == false;
    prof_skip_ = prof_on_ == 2 && prof_every_ > 1 && (prof_pass_ % prof_every_) != 0;
    ++prof_pass_;
    for (const Op& op : program_) exec_op(R, op, g);
    prof_chainable_ = false;
    prof_chain_ = false;
    prof_skip_ = false;
}

// ------------------------------------------------------------------------------------------------
// entry points
// ------------------------------------------------------------------------------------------------
void Engine::classify_device(const uint8_t* d_rgb, int n, int h, int w, const uint8_t* d_is_jpeg, double* d_scores,
                             int32_t* d_label, hipStream_t stream) {
    check_shape(n, h, w, false);
    if (!d_rgb) fail(IRE_ERR_INVALID_INPUT, "invalid input: null image pointer");
    ensure_io(n, 1, 1);
    Lane tmp; tmp.stream = stream;
    prof_begin(FAM_CLASSIFIER, stream, 0, (double)n * h * w * 3);
    classifier_launch(tables_, d_rgb, n, h, w, d_is_jpeg, d_sums_, d_scores ? d_scores : d_scores_, d_label ? d_label : d_label_,
                      d_cond_, stream);
    prof_end(stream);
    last_n_ = n;
}

void Engine::restore_device(const uint8_t* d_rgb, int n, int h, int w, const double* d_scores, const uint8_t* d_is_jpeg,
                            uint8_t* d_out, hipStream_t stream) {
    check_shape(n, h, w, true);
    if (!net_.loaded) fail(IRE_ERR_UNAVAILABLE, "service unavailable: RestoreNet weights are not loaded");
    if (!d_rgb || !d_out) fail(IRE_ERR_INVALID_INPUT, "invalid input: null image pointer");
    ensure_io(n, 1, 1);
    ensure_workspace(n, h, w);
    if (d_scores) {
        scores_to_cond_launch(d_scores, n, d_cond_, stream);
    } else {
        // classified inside: the scan's last workgroup per image writes the FiLM vector too (no film launch, no boundary behind the scan)
        prof_begin(FAM_CLASSIFIER, stream, 0, (double)n * h * w * 3);
        classifier_launch(tables_, d_rgb, n, h, w, d_is_jpeg, d_sums_, d_scores_, d_label_, d_cond_, stream, net_.d_film_w, net_.d_film_b, kFilmDim, d_film_);
        prof_end(stream);
    }
    if (d_scores) {
        prof_begin(FAM_GN, stream, 0, 0);
        film_launch(d_cond_, n, net_.d_film_w, net_.d_film_b, kFilmDim, d_film_, stream);
        prof_end(stream);
    }
    last_n_ = n;
    batches_run_ += 1; images_restored_ += n; last_batch_ = n;
    {
        const double t = now_s();
        recent_.emplace_back(t, n);
        while (!recent_.empty() && recent_.front().first < t - 10.0) recent_.pop_front();
    }

    const int lanes_used = std::min(num_lanes_, n);
    const size_t img_bytes = (size_t)h * w * 3;
    if (lanes_used == 1) {
        Lane L = lanes_[0];
        L.stream = stream;  // run inline on the caller's stream
        run_network(L, n, h, w, d_rgb, d_out, d_film_);
        return;
    }
    const int per = ceil_div(n, lanes_used);
    IRE_HIP(hipEventRecord(fork_ev_, stream));
    for (int i = 0; i < lanes_used; ++i) {
        const int i0 = i * per, cnt = std::min(per, n - i0);
        if (cnt <= 0) break;
        Lane& L = lanes_[i];
        IRE_HIP(hipStreamWaitEvent(L.stream, fork_ev_, 0));
        run_network(L, cnt, h, w, d_rgb + (size_t)i0 * img_bytes, d_out + (size_t)i0 * img_bytes,
                    d_film_ + (size_t)i0 * kFilmDim);
        IRE_HIP(hipEventRecord(L.done, L.stream));
        IRE_HIP(hipStreamWaitEvent(stream, L.done, 0));
    }
}

void Engine::restore_tiled_device(const uint8_t* d_rgb, int h, int w, int nstrips, const double* d_scores, const uint8_t* d_is_jpeg,
                                  uint8_t* d_out, hipStream_t stream) {
    if (!d_rgb || !d_out) fail(IRE_ERR_INVALID_INPUT, "invalid input: null image pointer");
    if (!tiled_ || tiled_h_ != h || tiled_w_ != w || tiled_n_ != nstrips) {
        IRE_HIP(hipDeviceSynchronize());
        delete tiled_; tiled_ = nullptr;
        tiled_ = new StripSession(*this, h, w, nstrips, 0, nstrips, nullptr);
        tiled_h_ = h; tiled_w_ = w; tiled_n_ = nstrips;
    }
    ensure_io(1, 1, 1);
    if (!d_scores) {
        prof_begin(FAM_CLASSIFIER, stream, 0, (double)h * w * 3);
        classifier_launch(tables_, d_rgb, 1, h, w, d_is_jpeg, d_sums_, d_scores_, d_label_, d_cond_, stream);
        prof_end(stream);
        d_scores = d_scores_;
    }
    batches_run_ += 1; images_restored_ += 1; last_batch_ = 1;
    {
        const double t = now_s();          // tiled jobs count in the images/sec gauge like whole-batch calls
        recent_.emplace_back(t, 1);
        while (!recent_.empty() && recent_.front().first < t - 10.0) recent_.pop_front();
    }
    tiled_->run_all(d_rgb, d_scores, d_out, stream);
}

void Engine::restore_device_mixed(const uint8_t* d_rgb, int n, int h, int w, const double* host_scores, const uint8_t* has_scores,
                                  const uint8_t* d_is_jpeg, uint8_t* d_out, hipStream_t stream) {
    check_shape(n, h, w, true);
    ensure_io(n, 1, 1);
    bool any_missing = false, any_given = false;
    for (int i = 0; i < n; ++i) { if (has_scores && has_scores[i]) any_given = true; else any_missing = true; }
    if (!any_given) { restore_device(d_rgb, n, h, w, nullptr, d_is_jpeg, d_out, stream); return; }
    if (any_missing) classify_device(d_rgb, n, h, w, d_is_jpeg, d_scores_, d_label_, stream);
    for (int i = 0; i < n; ++i)
        if (has_scores[i]) IRE_HIP(hipMemcpyAsync(d_scores_ + 7 * i, host_scores + 7 * i, sizeof(double) * 7, hipMemcpyHostToDevice, stream));
    restore_device(d_rgb, n, h, w, d_scores_, d_is_jpeg, d_out, stream);
}

void Engine::get_stats(ire_engine_stats* out) {
    out->batches = batches_run_;
    out->images = images_restored_;
    out->last_batch = last_batch_;
    out->max_batch = max_batch_;
    const double t = now_s();
    while (!recent_.empty() && recent_.front().first < t - 10.0) recent_.pop_front();
    double imgs = 0;
    for (auto& r : recent_) imgs += r.second;
    // span from the first call of the window to now; one lone call reports over >= 1 s so the gauge decays instead of spiking
    const double span = recent_.empty() ? 1.0 : std::max(1.0, t - recent_.front().first);
    out->images_per_sec = recent_.empty() ? 0.0 : imgs / span;
}

void Engine::classify_host(const uint8_t* rgb, int n, int h, int w, int row_stride, const uint8_t* is_jpeg, double* scores,
                           int32_t* label) {
    check_shape(n, h, w, false);
    if (!rgb || !scores) fail(IRE_ERR_INVALID_INPUT, "invalid input: null pointer");
    if (row_stride < 3 * w) fail(IRE_ERR_INVALID_INPUT, "invalid row_stride (< 3*w)");
    ensure_io(n, h, w);
    hipStream_t s = main_stream_;
    IRE_HIP(hipMemcpy2DAsync(d_in_, (size_t)3 * w, rgb, (size_t)row_stride, (size_t)3 * w, (size_t)n * h, hipMemcpyHostToDevice, s));
    std::vector<uint8_t> jp(n, 1);
    if (is_jpeg) std::memcpy(jp.data(), is_jpeg, n);
    IRE_HIP(hipMemcpyAsync(d_jpeg_, jp.data(), n, hipMemcpyHostToDevice, s));
    classify_device(d_in_, n, h, w, d_jpeg_, d_scores_, d_label_, s);
    IRE_HIP(hipMemcpyAsync(scores, d_scores_, sizeof(double) * 7 * n, hipMemcpyDeviceToHost, s));
    if (label) IRE_HIP(hipMemcpyAsync(label, d_label_, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
    IRE_HIP(hipStreamSynchronize(s));
}

void Engine::restore_host(const uint8_t* rgb, int n, int h, int w, const double* scores, const uint8_t* is_jpeg,
                          uint8_t* out, ire_timings* t) {
    check_shape(n, h, w, true);
    if (!rgb || !out) fail(IRE_ERR_INVALID_INPUT, "invalid input: null pointer");
    ensure_io(n, h, w);
    hipStream_t s = main_stream_;
    const size_t bytes = (size_t)n * h * w * 3;
    IRE_HIP(hipEventRecord(ev_[0], s));
    IRE_HIP(hipMemcpyAsync(d_in_, rgb, bytes, hipMemcpyHostToDevice, s));
    std::vector<uint8_t> jp(n, 1);
    if (is_jpeg) std::memcpy(jp.data(), is_jpeg, n);
    IRE_HIP(hipMemcpyAsync(d_jpeg_, jp.data(), n, hipMemcpyHostToDevice, s));
    const double* d_sc = nullptr;
    IRE_HIP(hipEventRecord(ev_[1], s));
    if (scores) {
        IRE_HIP(hipMemcpyAsync(d_scores_, scores, sizeof(double) * 7 * n, hipMemcpyHostToDevice, s));
        d_sc = d_scores_;
    } else {
        // classify as its own step so classify_ms / restore_ms mirror restorator.js:59-95
        prof_begin(FAM_CLASSIFIER, s, 0, (double)n * h * w * 3);
        classifier_launch(tables_, d_in_, n, h, w, d_jpeg_, d_sums_, d_scores_, d_label_, d_cond_, s);
        prof_end(s);
        d_sc = d_scores_;
    }
    IRE_HIP(hipEventRecord(ev_[2], s));
    restore_device(d_in_, n, h, w, d_sc, d_jpeg_, d_out_, s);
    IRE_HIP(hipEventRecord(ev_[3], s));
    IRE_HIP(hipMemcpyAsync(out, d_out_, bytes, hipMemcpyDeviceToHost, s));
    IRE_HIP(hipStreamSynchronize(s));
    if (t) {
        float a = 0, b = 0, c = 0;
        IRE_HIP(hipEventElapsedTime(&a, ev_[1], ev_[2]));
        IRE_HIP(hipEventElapsedTime(&b, ev_[2], ev_[3]));
        IRE_HIP(hipEventElapsedTime(&c, ev_[0], ev_[3]));
        t->classify_ms = a; t->restore_ms = b; t->total_ms = c;
    }
}

}  // namespace ire
