// encode.hip -- the result side of the restoreImage seam on the device: restored RGB pixels -> the base64 text of a PNG file.
//
// The reference's result contract is a base64 STRING of an ENCODED image (server-node/src/services/restorator.js:108:
// `restoredImage: result.base64Image`; geminiClient.js:75-88 hands the provider's base64 through).  Round 3 measured the whole Node
// seam JS-thread-bound on exactly that: V8's base64 of a raw 3-MB image is 1.03 ms per job on the one JS thread, and with a real codec
// (PIL's PNG of a 1024^2 photo: ~0.2 s of zlib per image and core; JPEG q85: ~15 ms) the codec, not the engine, sets the rate of a
// deployed worker (tools/codec_seam_rate.py).  This file removes the result side's host work altogether: three
// small kernels behind the restore write, per image, a complete PNG file -- signature, IHDR, ONE IDAT chunk holding a zlib stream of
// STORED deflate blocks (filter type 0 on every scanline), IEND -- with its Adler-32 and CRC-32 computed on the device, and then its
// base64 text.  Every PNG decoder reads it (stored blocks are plain deflate); it is 0.2 % larger than the pixels where a compressed
// PNG of a photograph is 20-30 % smaller -- the trade is host CPU seconds for 1 MB more per result on the wire.  The host receives
// ASCII it can hand to the client as it is (Node: buf.latin1Slice(); Python: bytes.decode('ascii')).
//
// Integer / byte work, bit-exact by construction against zlib.adler32, zlib.crc32, base64.b64encode and PIL's decoder
// (oracle/encode.py; tests/test_encode_gpu.py).  HBM-bound by bytes: ~3 (pixels) + 3 + 3 (file written, read) + 4 (text) = 13 B per
// pixel-byte triple... ~14 MB per 1024^2 image, a few microseconds; in practice three dependent launches.
#include "encode.hpp"

#include <cstring>

namespace ire {

namespace {

constexpr unsigned kCrcPoly = 0xedb88320u;      // reflected CRC-32 (zlib)
constexpr int kStored = 65535;                  // bytes of one stored deflate block
constexpr int kSlice = 256;                     // bytes of the IDAT chunk a thread runs its CRC over
constexpr int kCrcWG = 256;                     // threads (slices) per workgroup: 64 KB of the chunk

// ---- CRC-32 arithmetic in GF(2)[x] / p(x), reflected representation (bit 31 = x^0), as zlib's crc32.c does it -----------------------
__host__ __device__ inline unsigned gf_mul(unsigned a, unsigned b) {        // a(x) * b(x) mod p(x)
    unsigned m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}
inline unsigned gf_x_pow_8n(size_t nbytes) {     // x^(8 nbytes) mod p: what feeding nbytes zero bytes does to a CRC register
    unsigned sq = 1u << 30, p = 1u << 31;       // x^1, x^0
    size_t n = nbytes * 8;
    while (n) { if (n & 1) p = gf_mul(sq, p); sq = gf_mul(sq, sq); n >>= 1; }
    return p;
}
unsigned host_crc32(const unsigned char* d, size_t n) {
    unsigned c = 0xffffffffu;
    for (size_t i = 0; i < n; ++i) { c ^= d[i]; for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1; }
    return c ^ 0xffffffffu;
}

struct PngGeom {
    int h, w;
    unsigned row;             // bytes of a filtered scanline: 1 + 3 w
    unsigned long long raw;   // bytes of the raw (filtered) stream: h * row
    unsigned nblk;            // stored blocks
    unsigned long long zlen;  // zlib stream: 2 + 5 nblk + raw + 4
    unsigned long long file;  // 8 + 25 + (12 + zlen) + 12
    unsigned long long idat;  // file offset of the IDAT chunk's type field ("IDAT": the CRC starts here) = 8 + 25 + 4
    unsigned long long crc_len;   // bytes the IDAT CRC covers: 4 + zlen
};
PngGeom geom_of(int h, int w) {
    PngGeom g;
    g.h = h; g.w = w;
    g.row = 1u + 3u * (unsigned)w;
    g.raw = (unsigned long long)h * g.row;
    g.nblk = (unsigned)((g.raw + kStored - 1) / kStored);
    g.zlen = 2 + 5ull * g.nblk + g.raw + 4;
    g.file = 8 + 25 + 12 + g.zlen + 12;
    g.idat = 8 + 25 + 4;
    g.crc_len = 4 + g.zlen;
    return g;
}

struct PngHead { unsigned char b[41]; };       // signature + IHDR chunk (CRC included) + the IDAT chunk's length and type
__constant__ unsigned char kIend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xae, 0x42, 0x60, 0x82};

// K1: every byte of the file except the two checksums.  A thread writes one dword (4 file bytes; the file buffer is padded to a
// multiple of 4).  File regions: [0, 41) head | zlib header 78 01 | nblk x (5-byte stored-block header + <= 65535 raw bytes) |
// Adler-32 (K2) | IDAT CRC (K2) | IEND chunk.
__global__ __launch_bounds__(256) void png_frame_kernel(const unsigned char* __restrict__ rgb, PngGeom g, PngHead head, unsigned char* __restrict__ file) {
    const unsigned long long o0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (o0 >= g.file) return;
    const unsigned long long z0 = g.idat + 4;                 // first byte of the zlib stream
    const unsigned long long d0 = z0 + 2;                     // first stored block
    const unsigned long long dend = d0 + 5ull * g.nblk + g.raw;
    unsigned out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long o = o0 + k;
        unsigned v = 0;
        if (o < 41) v = head.b[o];
        else if (o < d0) v = o == z0 ? 0x78u : 0x01u;
        else if (o < dend) {
            const unsigned rel = (unsigned)(o - d0);                  // (the largest image's file fits 32 bits: 16384 x (1 + 3 x 16384) = 0.8 G)
            const unsigned blk = rel / (kStored + 5), r = rel - blk * (kStored + 5);
            if (r < 5) {
                const unsigned long long left = g.raw - (unsigned long long)blk * kStored;
                const unsigned len = left < (unsigned)kStored ? (unsigned)left : (unsigned)kStored;
                v = r == 0 ? (blk + 1 == g.nblk ? 1u : 0u) : r == 1 ? (len & 0xffu) : r == 2 ? (len >> 8) : r == 3 ? ((~len) & 0xffu) : (((~len) >> 8) & 0xffu);
            } else {
                const unsigned i = blk * kStored + (r - 5);          // raw stream index
                const unsigned y = i / g.row, c = i - y * g.row;
                v = c == 0 ? 0u : rgb[(size_t)y * (g.row - 1) + (c - 1)];                   // filter type 0 | a pixel byte
            }
        } else if (o < dend + 8) v = 0;                           // Adler-32, CRC-32: written by png_sums_kernel
        else if (o < g.file) v = kIend[o - (dend + 8)];
        out |= v << (8 * k);
    }
    *reinterpret_cast<unsigned*>(file + o0) = out;
}

// K2a: Adler-32 of the raw stream from the PIXELS (the filter bytes are zeros: they only shift positions) as two exact integer sums --
// S = sum d_i, T = sum pos_i d_i with pos_i the byte's index in the raw stream -- accumulated per workgroup and added with two 64-bit
// atomics (integers: any order gives the same result).  adler = ((N + N S - T) mod 65521) << 16 | (1 + S) mod 65521.
__global__ __launch_bounds__(256) void png_adler_kernel(const unsigned char* __restrict__ rgb, PngGeom g, unsigned long long* __restrict__ acc) {
    __shared__ unsigned long long s_s[256], s_t[256];
    const unsigned long long npix_bytes = (unsigned long long)g.h * (g.row - 1);
    unsigned long long S = 0, T = 0;
    for (unsigned long long i4 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 4; i4 < npix_bytes; i4 += (unsigned long long)gridDim.x * 1024) {
        const unsigned wv = *reinterpret_cast<const unsigned*>(rgb + i4);        // (3 w is a multiple of 4: w % 8 == 0 -- a dword never straddles two rows)
        const unsigned long long y = (unsigned)i4 / (g.row - 1);                 // (the pixel bytes of the largest image fit 32 bits)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned d = (wv >> (8 * k)) & 0xffu;
            S += d;
            T += (i4 + k + y + 1) * d;                                           // raw index = i + (rows before and including this one's filter byte)
        }
    }
    s_s[threadIdx.x] = S; s_t[threadIdx.x] = T;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) { s_s[threadIdx.x] += s_s[threadIdx.x + off]; s_t[threadIdx.x] += s_t[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicAdd(&acc[0], s_s[0]); atomicAdd(&acc[1], s_t[0]); }
}

// K2b: the Adler-32 into the file, then the CRC-32 of the IDAT chunk (type + zlib stream, Adler included) in two levels: a thread
// runs the byte-wise CRC over its 256-byte slice from register 0 (the chunk's first slice from 0xffffffff); a workgroup combines its
// 256 slices by a tree (level l: reg(A || B) = reg(A) x^(8 |B|) + reg(B), |B| = 256 * 2^l bytes: operators from the host); the per-
// workgroup registers go to `part`, and the LAST workgroup (ticket) folds them left to right with the 64-KB operator and the tail's own.
struct CrcOps { unsigned lvl[8]; unsigned wg; unsigned last_wg; };     // x^(8 * 256 * 2^l), x^(8 * 65536), x^(8 * bytes of the last workgroup's share)
__global__ __launch_bounds__(kCrcWG) void png_crc_kernel(unsigned char* __restrict__ file, PngGeom g, CrcOps ops, const unsigned long long* __restrict__ acc,
                                                          unsigned* __restrict__ part, unsigned* __restrict__ ticket) {
    __shared__ unsigned s_tab[256];
    __shared__ unsigned s_reg[kCrcWG];
    __shared__ unsigned s_last;
    {   // byte table of the reflected polynomial
        unsigned c = threadIdx.x;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
        s_tab[threadIdx.x] = c;
    }
    const unsigned long long adler_off = g.idat + 4 + g.zlen - 4;
    // every workgroup derives the same Adler-32 (two loads); the one whose slices hold it patches the bytes in before reading them
    const unsigned long long S = acc[0], T = acc[1];
    const unsigned A = (unsigned)((1 + S) % 65521ull);
    const unsigned B = (unsigned)((g.raw % 65521ull + (g.raw % 65521ull) * (S % 65521ull) + 65521ull - T % 65521ull) % 65521ull);
    const unsigned adler = (B << 16) | A;
    __syncthreads();
    const unsigned long long lo = (unsigned long long)blockIdx.x * (kCrcWG * kSlice) + (unsigned long long)threadIdx.x * kSlice;     // offset within the CRC'd range
    unsigned reg = (blockIdx.x == 0 && threadIdx.x == 0) ? 0xffffffffu : 0u;
    if (lo < g.crc_len) {
        const unsigned long long hi = lo + kSlice < g.crc_len ? lo + kSlice : g.crc_len;
        for (unsigned long long o = lo; o < hi; ++o) {
            const unsigned long long fo = g.idat + o;
            unsigned v = file[fo];
            if (fo >= adler_off && fo < adler_off + 4) { v = (adler >> (8 * (3 - (fo - adler_off)))) & 0xffu; file[fo] = (unsigned char)v; }     // big-endian
            reg = s_tab[(reg ^ v) & 0xffu] ^ (reg >> 8);
        }
    }
    // a slice shorter than 256 bytes (the chunk's tail) must still count as |B| = 256 in the tree: pad it with zero bytes on the RIGHT?
    // No -- zeros on the right change the register.  Instead the tree runs over FULL slices only and the host chose the operators so
    // that the tail is handled exactly: every slice but the chunk's last is full, and a partial / empty slice to the right of it is
    // combined with its TRUE length (0 .. 255), i.e. operator x^(8 len) computed here (rare: one thread per launch).
    s_reg[threadIdx.x] = reg;
    __syncthreads();
    // true byte count of the slices [t, t + span) of this workgroup
    auto bytes_of = [&](unsigned t, unsigned span) -> unsigned long long {
        const unsigned long long a = (unsigned long long)blockIdx.x * (kCrcWG * kSlice) + (unsigned long long)t * kSlice;
        const unsigned long long b = a + (unsigned long long)span * kSlice;
        const unsigned long long aa = a < g.crc_len ? a : g.crc_len, bb = b < g.crc_len ? b : g.crc_len;
        return bb - aa;
    };
    for (int l = 0; l < 8; ++l) {
        const unsigned span = 1u << l;
        if ((threadIdx.x & (2 * span - 1)) == 0) {
            const unsigned long long nb = bytes_of(threadIdx.x + span, span);           // bytes of the right half
            unsigned op = ops.lvl[l];
            if (nb != (unsigned long long)span * kSlice) {                               // the chunk's tail: its own operator
                unsigned sq = 1u << 30, p = 1u << 31;
                unsigned long long n = nb * 8;
                while (n) { if (n & 1) p = gf_mul(sq, p); sq = gf_mul(sq, sq); n >>= 1; }
                op = p;
            }
            s_reg[threadIdx.x] = gf_mul(op, s_reg[threadIdx.x]) ^ s_reg[threadIdx.x + span];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&part[blockIdx.x], s_reg[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        unsigned r = 0;
        for (unsigned k = 0; k < gridDim.x; ++k) {
            const unsigned pk = __hip_atomic_load(&part[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r = k == 0 ? pk : gf_mul(k + 1 == gridDim.x ? ops.last_wg : ops.wg, r) ^ pk;
        }
        const unsigned crc = r ^ 0xffffffffu;
        const unsigned long long co = g.idat + g.crc_len;
        file[co] = (unsigned char)(crc >> 24); file[co + 1] = (unsigned char)(crc >> 16); file[co + 2] = (unsigned char)(crc >> 8); file[co + 3] = (unsigned char)crc;
        *ticket = 0;                                             // the next call finds the ticket at zero: no memset per call
    }
}

// K3: base64 (RFC 4648, '=' padded): a thread turns 12 file bytes (three dwords) into 16 characters (four dwords).
__device__ __forceinline__ unsigned b64_char(unsigned v) {     // 0..63 -> 'A'..'Z' 'a'..'z' '0'..'9' '+' '/'
    return v < 26 ? v + 65 : v < 52 ? v + 71 : v < 62 ? v - 4 : v == 62 ? 43 : 47;
}
__global__ __launch_bounds__(256) void base64_kernel(const unsigned char* __restrict__ in, unsigned long long n, unsigned char* __restrict__ out) {
    const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long i0 = t * 12;
    if (i0 >= n) return;
    const unsigned* src = reinterpret_cast<const unsigned*>(in + i0);          // the file buffer is padded to a multiple of 12 readable bytes
    const unsigned w[3] = {src[0], src[1], src[2]};
    unsigned o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                             // group q: input bytes 3 q .. 3 q + 2
        unsigned b[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int j = 3 * q + k; b[k] = (w[j >> 2] >> (8 * (j & 3))) & 0xffu; }
        const unsigned long long at = i0 + 3 * q;
        const unsigned have = at >= n ? 0u : (n - at >= 3 ? 3u : (unsigned)(n - at));
        if (have < 3) b[2] = 0;
        if (have < 2) b[1] = 0;
        const unsigned v = (b[0] << 16) | (b[1] << 8) | b[2];
        const unsigned c0 = b64_char(v >> 18), c1 = b64_char((v >> 12) & 63u), c2 = have >= 2 ? b64_char((v >> 6) & 63u) : 61u, c3 = have >= 3 ? b64_char(v & 63u) : 61u;
        o[q] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    }
    const unsigned long long groups = (n + 2) / 3, g0 = t * 4;
    unsigned* dst = reinterpret_cast<unsigned*>(out + g0 * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) if (g0 + q < groups) dst[q] = o[q];
}

}  // namespace

size_t png_file_bytes(int h, int w) { return (size_t)geom_of(h, w).file; }
size_t png_base64_chars(int h, int w) { return (png_file_bytes(h, w) + 2) / 3 * 4; }
size_t png_scratch_bytes(int h, int w) {       // file (padded so that the 12-byte groups of the last threads stay inside) | 2 x u64 sums | ticket | per-workgroup CRC registers
    const PngGeom g = geom_of(h, w);
    const size_t file_pad = ((size_t)g.file + 11) / 12 * 12 + 16;
    const size_t nwg = ((size_t)g.crc_len + kCrcWG * kSlice - 1) / (kCrcWG * kSlice);
    return (file_pad + 15) / 16 * 16 + 32 + 16 + nwg * 4 + 16;
}

// d_rgb [h][w][3] -> d_chars (png_base64_chars bytes of ASCII); d_scratch: png_scratch_bytes, zero at first use (the kernels reset what
// they count in).  Three dependent launches on `s`.
void encode_png_base64_launch(const unsigned char* d_rgb, int h, int w, unsigned char* d_scratch, unsigned char* d_chars, hipStream_t s) {
    if (h <= 0 || w <= 0 || w % 8 || h > 16384 || w > 16384) fail(IRE_ERR_INVALID_INPUT, "invalid image size for the PNG encoder: width must be a multiple of 8");
    const PngGeom g = geom_of(h, w);
    const size_t file_pad = (((size_t)g.file + 11) / 12 * 12 + 16 + 15) / 16 * 16;
    unsigned char* file = d_scratch;
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(d_scratch + file_pad);
    unsigned* ticket = reinterpret_cast<unsigned*>(d_scratch + file_pad + 32);
    unsigned* part = reinterpret_cast<unsigned*>(d_scratch + file_pad + 48);
    PngHead head;
    {
        static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
        std::memcpy(head.b, sig, 8);
        unsigned char* c = head.b + 8;
        c[0] = 0; c[1] = 0; c[2] = 0; c[3] = 13; c[4] = 'I'; c[5] = 'H'; c[6] = 'D'; c[7] = 'R';
        c[8] = (unsigned char)(w >> 24); c[9] = (unsigned char)(w >> 16); c[10] = (unsigned char)(w >> 8); c[11] = (unsigned char)w;
        c[12] = (unsigned char)(h >> 24); c[13] = (unsigned char)(h >> 16); c[14] = (unsigned char)(h >> 8); c[15] = (unsigned char)h;
        c[16] = 8; c[17] = 2; c[18] = 0; c[19] = 0; c[20] = 0;
        const unsigned crc = host_crc32(c + 4, 17);
        c[21] = (unsigned char)(crc >> 24); c[22] = (unsigned char)(crc >> 16); c[23] = (unsigned char)(crc >> 8); c[24] = (unsigned char)crc;
        unsigned char* d = head.b + 33;
        d[0] = (unsigned char)(g.zlen >> 24); d[1] = (unsigned char)(g.zlen >> 16); d[2] = (unsigned char)(g.zlen >> 8); d[3] = (unsigned char)g.zlen;
        d[4] = 'I'; d[5] = 'D'; d[6] = 'A'; d[7] = 'T';
    }
    IRE_HIP(hipMemsetAsync(acc, 0, 16, s));
    const unsigned nthr = (unsigned)((g.file + 3) / 4);
    hipLaunchKernelGGL(png_frame_kernel, dim3((nthr + 255) / 256), dim3(256), 0, s, d_rgb, g, head, file);
    const unsigned long long npb = (unsigned long long)h * w * 3;
    unsigned agrid = (unsigned)((npb / 4 + 1023) / 1024);
    if (agrid > 1024) agrid = 1024;
    if (agrid < 1) agrid = 1;
    hipLaunchKernelGGL(png_adler_kernel, dim3(agrid), dim3(256), 0, s, d_rgb, g, acc);
    CrcOps ops;
    for (int l = 0; l < 8; ++l) ops.lvl[l] = gf_x_pow_8n((size_t)kSlice << l);
    const size_t wg_bytes = (size_t)kCrcWG * kSlice;
    const unsigned nwg = (unsigned)((g.crc_len + wg_bytes - 1) / wg_bytes);
    ops.wg = gf_x_pow_8n(wg_bytes);
    ops.last_wg = gf_x_pow_8n((size_t)(g.crc_len - (unsigned long long)(nwg - 1) * wg_bytes));
    hipLaunchKernelGGL(png_crc_kernel, dim3(nwg), dim3(kCrcWG), 0, s, file, g, ops, acc, part, ticket);
    const unsigned long long groups12 = (g.file + 11) / 12;
    hipLaunchKernelGGL(base64_kernel, dim3((unsigned)((groups12 + 255) / 256)), dim3(256), 0, s, file, g.file, d_chars);
    IRE_HIP(hipGetLastError());
}

}  // namespace ire
