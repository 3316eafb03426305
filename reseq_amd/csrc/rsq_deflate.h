// rsq_deflate.h -- gzip of FASTQ text that lies in device memory (SURVEY.md section 8(f) item 4 "gz FASTQ output"; the reference writes .gz through SeqAn's
// compressed stream from Simulator::Flush, reseq/Simulator.cpp:150-182, output names main.cpp:404,412).
//
// The text is cut into pieces of kPiece bytes; every piece becomes one gzip member (RFC 1952) of its own, framed like a BGZF block (an extra field "BC" with the
// member's size, input at most 65280 bytes), so that the members can be made by independent workgroups, concatenate in any grouping (ranks, batches) and are read
// by zlib's gzread, gzip -d, SeqAn and by the block-wise readers of bgzip / htslib alike.  Inside a member ONE deflate block (RFC 1951) with a dynamic Huffman
// code that is NOT the piece's own: the code is built on the host once per call from the symbol counts of a sample of the call's pieces (FASTQ text is stationary:
// the same ids, four bases, forty qualities everywhere) -- so the device never builds a code, and a piece's encoding needs nothing of another piece.
//
// A piece on the device, 256 threads (k_gzip_pieces), three workgroups per CU (51 KB of LDS each):
//   rounds of 8 KB; the round's text -- with 272 bytes beyond it and what came before -- stands in a 16 KB ring in LDS (global loads of 16 bytes, once); per round
//   A  every position finds its candidate -- the nearest earlier position with the same four bytes, through a hash table of positions in LDS, filled 256 positions
//      at a time (read all, barrier, atomicMax all: deterministic), plus the position one byte back (runs) -- and probes it for eight bytes, without a loop: most
//      positions lie inside a match that B passes over, so nothing more is measured here; a match is kept in 16 bits (length, distance up to 2048);
//   B  thread t owns the 32-byte segment t of the round: its 32 matches and 32 bytes go into registers (six 16-byte LDS reads) and an unrolled walk takes them
//      greedily (longest match at the current position, as zlib's level 1 does; a match that reaches the probe's end is measured to its end now; a match may not
//      leave the segment: the walks are independent) -- once to count the bits, then, after a scan of the counts over the workgroup, to OR the codes LSB-first
//      into the round's bit buffer in LDS (ds_or: neighbours share a word where their bits meet); the buffer's complete words go to the member in HBM in coalesced
//      stores, its last, incomplete word to the front of the next round's buffer;
//   the end-of-block code, the CRC-32 of the text (slices per thread, four bytes per step, folded with x^(8 len) mod P like zlib's crc32_combine) and the trailer.
// A piece whose bits come to no less than the piece itself (text that looks nothing like the sample, or a few bytes behind the block header), or a round of which
// needs more than 8 bits per byte, is stored instead (BTYPE 00) by a second kernel: text + 31 bytes bound every member, and the slot holds that.
// Measured on the simulator's own FASTQ text (P0, 742 MB): 86 GB/s by kernel time, 2.78 x smaller (zlib level 1: 3.19 x, level 6: 3.71 x) -- profiles/r05_*.
//
// The per-thread functions are host/device code: tests/hostemu runs the same walk on the CPU against zlib's inflate.
#pragma once
#include <stdint.h>
#include <string.h>

#include "rsq_types.h"

#ifndef RSQ_LDS
#if defined(__HIP_DEVICE_COMPILE__)
#define RSQ_LDS __attribute__((address_space(3)))
#else
#define RSQ_LDS
#endif
#endif

namespace rsq {
namespace gz {

constexpr uint32_t kPiece = 65280;                 // bytes of text per member: BGZF's limit for a block's input
constexpr uint32_t kThreads = 256, kSeg = 32, kRound = kThreads * kSeg;
constexpr uint32_t kHashBits = 11, kMinMatch = 3, kMaxMatch = 257;
constexpr uint32_t kRing = 16384, kAhead = 272;      // the device keeps the last kRing bytes of the piece in LDS: a round, kAhead bytes beyond it (the longest match + a word), and the history
constexpr uint32_t kMaxDist = 2048;               // how far back a match may reach: a match is kept as (length - 2) << 11 | (distance - 1) in 16 bits, and the ring holds far more
constexpr uint32_t kOutWords = 2048;              // words of the round's bit buffer in LDS: 8 bits per byte of the round -- a round that needs more is not worth coding
static_assert(kSeg == 32 && kMaxDist <= kRing - kRound - kAhead - 16u, "the packed match and the ring");
constexpr uint32_t kHeaderBytes = 18, kTrailerBytes = 8;
constexpr uint32_t kSlot = 65536 + 64;             // bytes of a member's slot: a stored piece needs kPiece + 5 + header + trailer
constexpr uint32_t kSlotPad = 2;                   // the member begins here in its slot: its deflate data, 18 bytes on, then lies on a 4-byte boundary (atomicOr on words)
constexpr uint32_t kSlotWords = (kSlot - kSlotPad - kHeaderBytes - kTrailerBytes) / 4u;      // words of deflate data a slot has room for
constexpr uint32_t kLitLen = 288, kDist = 32;      // code tables (286 and 30 symbols used)
constexpr uint32_t kHeaderWords = 128;             // room for the block header's bits (at most 3 + 14 + 19 * 3 + 318 * 14 bits)

// the Huffman codes of a call and the bits of the block header that announces them
struct Codes {
    uint32_t litlen[kLitLen];          // (code, bit-reversed for LSB-first output) << 4 | length (1..15)
    uint32_t dist[kDist];
    uint32_t header_bits;
    uint32_t header[kHeaderWords];     // BFINAL = 1, BTYPE = 10, HLIT, HDIST, HCLEN, the code length code, the code lengths -- LSB first
};

RSQ_HD uint32_t hash4(uint32_t v) { return (v * 2654435761u) >> (32u - kHashBits); }
RSQ_HD uint32_t load4(const uint8_t *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const uint32_t __attribute__((aligned(1))) *>(p);
#else
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
#endif
}
RSQ_HD uint32_t low_zero_bytes(uint32_t x) {           // number of low bytes of x that are zero (x != 0)
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_ctz(x) >> 3;
#else
    return (uint32_t)__builtin_ctz(x) >> 3;
#endif
}
// Where the walk reads the piece: plain memory (the host), or the ring of its last kRing bytes in LDS (the device; byte p at p mod kRing, the ring's first four
// bytes repeated behind its end so that a word may begin on its last three bytes)
struct PlainText {
    const uint8_t *t;
    uint32_t n;                          // nothing at or behind n is read (a word that reaches over the end is filled with zeros)
    RSQ_HD uint32_t byte(uint32_t p) const { return t[p]; }
    RSQ_HD uint32_t word(uint32_t p) const {
        if (p + 4u <= n) return load4(t + p);
        uint32_t v = 0;
        for (uint32_t i = 0; p + i < n; ++i) v |= (uint32_t)t[p + i] << (8u * i);
        return v;
    }
};
struct RingText {
    const RSQ_LDS uint8_t *ring;
    RSQ_HD uint32_t byte(uint32_t p) const { return ring[p & (kRing - 1u)]; }
    RSQ_HD uint32_t word(uint32_t p) const {
#if defined(__HIP_DEVICE_COMPILE__)
        // two aligned words and a byte shift: an unaligned word would be read byte by byte
        const RSQ_LDS uint32_t *w = reinterpret_cast<const RSQ_LDS uint32_t *>(ring) + ((p & (kRing - 1u)) >> 2);
        return __builtin_amdgcn_alignbyte(w[1], w[0], p & 3u);
#else
        uint32_t v;
        memcpy(&v, ring + (p & (kRing - 1u)), 4);
        return v;
#endif
    }
};
// what phase A leaves per position of a round: a match's length (0: none) and distance
struct Found {
    uint32_t len, dist;
};
// Phase A looks eight bytes far (seven for a run): enough to tell a match from none and to compare two candidates, and without a loop -- most positions lie INSIDE
// a match that the walk of phase B passes over, so what is measured here beyond that would be thrown away.  A match that reaches this far is measured to its end by
// the walk when it takes it (extend_match).
constexpr uint32_t kProbe = 7;
// A match is never used beyond the end of the 32-byte segment its position lies in (phase B cuts it there): `limit` stops there as well.
// v = text.word(p); cand_plus1: the hash table's entry for these four bytes as it stood before this group of positions was entered (position + 1, 0 = none)
template <class Text>
RSQ_HD Found find_match(const Text &text, uint32_t n, uint32_t round_lo, uint32_t p, uint32_t v, uint32_t cand_plus1) {
    Found f{0u, 0u};
    if (p + kMinMatch > n) return f;
    const uint32_t segment_end = round_lo + ((p - round_lo) / kSeg + 1u) * kSeg;
    uint32_t limit = n - p < kMaxMatch ? n - p : kMaxMatch;
    if (segment_end - p < limit) limit = segment_end - p;
    if (limit < kMinMatch) return f;
    if (cand_plus1 && limit >= 4u) {                                 // the table's candidate: all four hashed bytes, or nothing (a collision)
        const uint32_t q = cand_plus1 - 1u;
        if (p - q <= kMaxDist && text.word(q) == v) {
            const uint32_t x = text.word(p + 4u) ^ text.word(q + 4u), len = 4u + (x ? low_zero_bytes(x) : 4u);
            f = Found{len < limit ? len : limit, p - q};
        }
    }
    if (p && f.len < limit && ((text.word(p - 1u) ^ v) & 0xFFFFFFu) == 0u) {      // a run: the position one byte back (the table knows nothing nearer than a group)
        const uint32_t x = text.word(p + 3u) ^ text.word(p + 2u);
        uint32_t len = 3u + (x ? low_zero_bytes(x) : 4u);
        if (len > limit) len = limit;
        if (len > f.len) f = Found{len, 1u};
    }
    return f;
}
// the match of `len` bytes (so far) at p, `dist` back, to its end: at most `most` bytes in all
template <class Text>
RSQ_HD uint32_t extend_match(const Text &text, uint32_t p, uint32_t dist, uint32_t len, uint32_t most) {
    while (len < most) {
        uint32_t x = text.word(p + len) ^ text.word(p + len - dist);
        if (most - len < 4u) x &= (1u << (8u * (most - len))) - 1u;   // only the bytes that count
        if (x) return len + low_zero_bytes(x);
        len += 4u;
    }
    return most;
}

// RFC 1951 3.2.5: symbol, number of extra bits and their value for a length (3..257) and a distance (1..32768), by arithmetic instead of tables
struct Sym {
    uint32_t code, extra_bits, extra;
};
RSQ_HD uint32_t floor_log2(uint32_t x) { return 31u - (uint32_t)__builtin_clz(x); }
RSQ_HD Sym length_symbol(uint32_t len) {
    const uint32_t l = len - 3u;
    if (l < 8u) return Sym{257u + l, 0u, 0u};
    const uint32_t k = floor_log2(l);
    return Sym{257u + 4u * (k - 1u) + ((l >> (k - 2u)) & 3u), k - 2u, l & ((1u << (k - 2u)) - 1u)};
}
RSQ_HD Sym distance_symbol(uint32_t dist) {
    const uint32_t d = dist - 1u;
    if (d < 4u) return Sym{d, 0u, 0u};
    const uint32_t k = floor_log2(d);
    return Sym{2u * k + ((d >> (k - 1u)) & 1u), k - 1u, d & ((1u << (k - 1u)) - 1u)};
}

// Phase B: the greedy walk over one 32-byte segment, from registers: `found` = the segment's 32 packed matches (0 = none, else (length - 2) << 11 | (distance - 1)),
// `text` = its 32 bytes, n = how many of them exist (the piece's last segment).  A match that would leave the segment is cut (a cut below three bytes becomes
// literals).  The loop is unrolled: every position reads its match and byte out of a register by constant shifts; what a position does depends on `skip`, the bytes
// a match before it still covers.
struct Segment {
    uint32_t found[kSeg / 2u], text[kSeg / 4u];
};
template <class Sink, class Extend>      // Extend: (position in the segment, distance, length so far, most) -> the match's full length (extend_match on the segment's text)
RSQ_HD void walk_segment(const Segment &g, uint32_t n, Sink &sink, const Extend &extend) {
    uint32_t skip = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t i = 0; i < kSeg; ++i) {
        const uint32_t e = (g.found[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu, c = (g.text[i >> 2] >> ((i & 3u) * 8u)) & 0xFFu;
        uint32_t len = e >> 11;
        len = len ? len + 2u : 0u;
        if (i + len > n) len = n > i ? n - i : 0u;
        const bool here = skip == 0u && i < n, is_match = len >= kMinMatch;
        if (here) {
            if (is_match) {
                if (len >= kProbe && i + len < n) len = extend(i, (e & 2047u) + 1u, len, n - i);      // phase A looked no further
                sink.match(len, (e & 2047u) + 1u);
            } else sink.literal(c);
        }
        skip = here ? (is_match ? len - 1u : 0u) : (skip ? skip - 1u : 0u);
    }
}
template <class Tab>
struct CountSink {                       // bits of a segment under a code
    Tab litlen, dist;
    uint32_t bits = 0;
    RSQ_HD void literal(uint32_t b) { bits += litlen[b] & 15u; }
    RSQ_HD void match(uint32_t len, uint32_t d) {
        const Sym l = length_symbol(len), s = distance_symbol(d);
        bits += (litlen[l.code] & 15u) + l.extra_bits + (dist[s.code] & 15u) + s.extra_bits;
    }
};
template <class Add>
struct HistogramSink {                   // symbol counts of a segment (the sample the call's code is built from)
    Add add;                             // add(symbol index: 0..287 literal / length, 288.. distance)
    RSQ_HD void literal(uint32_t b) { add(b); }
    RSQ_HD void match(uint32_t len, uint32_t d) {
        add(length_symbol(len).code);
        add(kLitLen + distance_symbol(d).code);
    }
};
// Bits into the round's buffer, LSB first, at bit `at` of it: every push ORs its bits into the one or two words they fall into (`Or`: or_word(index, value) --
// an LDS atomic on the device: neighbouring threads share a word where their bits meet).  The buffer is zero where nothing was pushed.
template <class Tab, class Or>
struct BitSink {
    Tab litlen, dist;
    Or out;
    uint32_t at = 0;
    RSQ_HD void push(uint32_t value, uint32_t bits) {
        const uint64_t x = (uint64_t)value << (at & 31u);
        out.or_word(at >> 5, (uint32_t)x);
        if ((uint32_t)(x >> 32)) out.or_word((at >> 5) + 1u, (uint32_t)(x >> 32));
        at += bits;
    }
    RSQ_HD void code(uint32_t entry) { push(entry >> 4, entry & 15u); }
    RSQ_HD void literal(uint32_t b) { code(litlen[b]); }
    RSQ_HD void match(uint32_t len, uint32_t d) {
        const Sym l = length_symbol(len), s = distance_symbol(d);
        const uint32_t a = litlen[l.code], b = dist[s.code];
        push((a >> 4) | (l.extra << (a & 15u)), (a & 15u) + l.extra_bits);              // at most 15 + 3 bits (lengths up to 32)
        push((b >> 4) | (s.extra << (b & 15u)), (b & 15u) + s.extra_bits);              // at most 15 + 10 bits (distances up to 2048)
    }
};

// CRC-32 (RFC 1952 8.): bytewise with a table; x^(8 n) mod P and the product mod P as zlib's crc32_combine forms them (reflected, P = 0xedb88320)
constexpr uint32_t kCrcPoly = 0xedb88320u;
RSQ_HD uint32_t crc_table_entry(uint32_t i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = c & 1u ? kCrcPoly ^ (c >> 1) : c >> 1;
    return c;
}
template <class Table>
RSQ_HD uint32_t crc32_bytes(const Table &table, const uint8_t *p, uint32_t n) {
    uint32_t c = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return ~c;
}
RSQ_HD uint32_t multmodp(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1u)) == 0) break;
        }
        m >>= 1;
        b = b & 1u ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}
RSQ_HD uint32_t x_to_8n_modp(uint32_t n_bytes) {               // x^(8 n) mod P: square-and-multiply over the bits of 8 n (x itself is 1 << 30 in the reflected form)
    uint32_t result = 1u << 31, square = 1u << 30;            // 1 and x
    for (uint64_t e = (uint64_t)n_bytes * 8u; e; e >>= 1) {
        if (e & 1u) result = multmodp(square, result);
        square = multmodp(square, square);
    }
    return result;
}
// crc(A || B) from crc(A), crc(B) and x^(8 |B|) mod P
RSQ_HD uint32_t crc_combine(uint32_t crc_a, uint32_t crc_b, uint32_t x_len_b) { return multmodp(x_len_b, crc_a) ^ crc_b; }

// the 18 bytes in front of a member's deflate data: gzip magic, deflate, FEXTRA, no time, unknown OS, XLEN 6, subfield 'B' 'C' length 2, member size - 1
RSQ_HD void member_header(uint8_t *h, uint32_t member_bytes) {
    const uint8_t fixed[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    for (int i = 0; i < 16; ++i) h[i] = fixed[i];
    h[16] = (uint8_t)((member_bytes - 1u) & 0xFFu);
    h[17] = (uint8_t)((member_bytes - 1u) >> 8);
}


#if RSQ_DEVICE_BUILD && !defined(RSQ_SPEC)
// ------------------------------------------------------------------------------------------------------------------------- device
struct LdsOr {
    RSQ_LDS uint32_t *words;
    __device__ void or_word(uint32_t w, uint32_t v) { atomicOr(words + w, v); }
};
// the CRC-32 of a piece by the workgroup: a slice per thread (the first takes the remainder), folded pairwise; every thread returns it.  `table`: 4 x 256 words --
// four bytes per step (slicing by four), the slice read 16 bytes per load where it lies on a 16-byte boundary of memory
__device__ inline uint32_t crc32_slice(const RSQ_LDS uint32_t *table, const uint8_t *p, uint32_t n) {
    uint32_t c = 0xFFFFFFFFu, i = 0;
    for (; i < n && ((uintptr_t)(p + i) & 15u); ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    auto four = [&](uint32_t w) {
        c ^= w;
        c = table[768u + (c & 0xFFu)] ^ table[512u + ((c >> 8) & 0xFFu)] ^ table[256u + ((c >> 16) & 0xFFu)] ^ table[c >> 24];
    };
    for (; i + 16u <= n; i += 16u) {
        const uint4 v = *reinterpret_cast<const uint4 *>(p + i);
        four(v.x);
        four(v.y);
        four(v.z);
        four(v.w);
    }
    for (; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return ~c;
}
__device__ inline uint32_t piece_crc(const uint8_t *t, uint32_t len, RSQ_LDS uint32_t *table, RSQ_LDS uint32_t *part) {
    const uint32_t tid = threadIdx.x, L = len / kThreads, first = len - (kThreads - 1u) * L;
    for (uint32_t i = tid; i < 256u; i += kThreads) table[i] = crc_table_entry(i);
    __syncthreads();
    for (uint32_t k = 1; k < 4u; ++k) {                               // table k: one more zero byte behind the byte
        for (uint32_t i = tid; i < 256u; i += kThreads) table[256u * k + i] = table[table[256u * (k - 1u) + i] & 0xFFu] ^ (table[256u * (k - 1u) + i] >> 8);
        __syncthreads();
    }
    part[tid] = tid == 0 ? crc32_slice(table, t, first) : crc32_slice(table, t + first + (tid - 1u) * L, L);
    uint32_t x = x_to_8n_modp(L);
    for (uint32_t width = 1; width < kThreads; width *= 2) {
        __syncthreads();
        if (tid % (2u * width) == 0) part[tid] = crc_combine(part[tid], part[tid + width], x);
        x = multmodp(x, x);
    }
    __syncthreads();
    return part[0];
}
__device__ inline void member_frame(uint8_t *slot, uint32_t data_bytes, uint32_t crc, uint32_t len) {      // header and trailer around data_bytes of deflate data (one thread)
    member_header(slot + kSlotPad, kHeaderBytes + data_bytes + kTrailerBytes);
    uint8_t *tail = slot + kSlotPad + kHeaderBytes + data_bytes;
    for (int i = 0; i < 4; ++i) {
        tail[i] = (uint8_t)(crc >> (8 * i));
        tail[4 + i] = (uint8_t)(len >> (8 * i));
    }
}

// SAMPLE: the symbol counts of every piece_step-th piece into hist (nothing written); else piece blockIdx.x into its slot, sizes[piece] = the member's bytes (0: its
// bits did not fit -- k_gzip_stored takes it)
// bytes [lo, hi) of the piece into the ring (16 bytes per load and store where memory and ring allow: lo and the piece's address are multiples of 16 except at a
// piece's end); the ring's first bytes again behind its end
__device__ inline void ring_load(RSQ_LDS uint8_t *ring, const uint8_t *t, uint32_t lo, uint32_t hi) {
    const uint32_t tid = threadIdx.x;
    if ((((uintptr_t)t | lo) & 15u) == 0) {
        const uint32_t whole = (hi - lo) / 16u;
        for (uint32_t i = tid; i < whole; i += kThreads) *reinterpret_cast<RSQ_LDS uint4 *>(ring + ((lo + 16u * i) & (kRing - 1u))) = *reinterpret_cast<const uint4 *>(t + lo + 16u * i);
        for (uint32_t p = lo + 16u * whole + tid; p < hi; p += kThreads) ring[p & (kRing - 1u)] = t[p];
    } else
        for (uint32_t p = lo + tid; p < hi; p += kThreads) ring[p & (kRing - 1u)] = t[p];
}
// the thread's segment of the round out of LDS into registers: 32 packed matches, 32 bytes of text (both on 16-byte boundaries)
__device__ inline Segment load_segment(const RSQ_LDS uint16_t *found, const RSQ_LDS uint8_t *ring, uint32_t round_lo, uint32_t lo) {
    Segment g;
    const RSQ_LDS uint4 *f = reinterpret_cast<const RSQ_LDS uint4 *>(found + (lo - round_lo));
    const RSQ_LDS uint4 *t = reinterpret_cast<const RSQ_LDS uint4 *>(ring + (lo & (kRing - 1u)));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint4 v = f[k];
        g.found[4 * k] = v.x, g.found[4 * k + 1] = v.y, g.found[4 * k + 2] = v.z, g.found[4 * k + 3] = v.w;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint4 v = t[k];
        g.text[4 * k] = v.x, g.text[4 * k + 1] = v.y, g.text[4 * k + 2] = v.z, g.text[4 * k + 3] = v.w;
    }
    return g;
}
// The bits of `total` more bits stand in the round's buffer from bit `frac` (< 32) on: its complete words go to the member's data (coalesced), the buffer is zeroed
// and the last, incomplete word moves to its front.  Returns the number of words written.  All threads; barriers inside.
__device__ inline uint32_t flush_round(RSQ_LDS uint32_t *out, uint32_t *words, uint32_t word_base, uint32_t frac, uint32_t total, bool all) {
    const uint32_t tid = threadIdx.x, have = frac + total, complete = all ? (have + 31u) >> 5 : have >> 5;
    __syncthreads();                                                  // every thread's bits are in the buffer
    for (uint32_t w = tid; w < complete; w += kThreads)
        if (word_base + w < kSlotWords) words[word_base + w] = out[w];
    const uint32_t carry = out[complete];
    __syncthreads();
    for (uint32_t w = tid; w <= complete + 1u && w < kOutWords + 2u; w += kThreads) out[w] = 0u;
    __syncthreads();
    if (tid == 0 && !all) out[0] = carry;
    return complete;
}
template <bool SAMPLE>
__global__ void __launch_bounds__(kThreads) k_gzip_pieces(const uint8_t *text, uint64_t n, uint32_t piece_step, const Codes *codes, uint8_t *slots, uint32_t *sizes, uint32_t *hist) {
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[kRing + 16u];
    __shared__ __attribute__((aligned(16))) uint16_t s_found[kRound];
    __shared__ uint32_t head[1u << kHashBits];
    __shared__ uint32_t s_out[SAMPLE ? 1u : kOutWords + 2u], s_litlen[SAMPLE ? 1u : kLitLen], s_dist[SAMPLE ? 1u : kDist], s_hist[SAMPLE ? kLitLen + kDist : 1u];
    __shared__ uint32_t s_part[kThreads], s_wave[kThreads / 64u];
    const uint32_t tid = threadIdx.x;
    const uint64_t piece = (uint64_t)blockIdx.x * piece_step;
    const uint8_t *t = text + piece * kPiece;
    const uint32_t len = (uint32_t)(n - piece * kPiece < kPiece ? n - piece * kPiece : kPiece);
    uint8_t *slot = SAMPLE ? nullptr : slots + piece * kSlot;
    uint32_t *words = SAMPLE ? nullptr : reinterpret_cast<uint32_t *>(slot + kSlotPad + kHeaderBytes);
    RSQ_LDS uint8_t *ring = (RSQ_LDS uint8_t *)s_ring;
    RSQ_LDS uint16_t *found = (RSQ_LDS uint16_t *)s_found;
    RSQ_LDS uint32_t *out = (RSQ_LDS uint32_t *)s_out;
    const RSQ_LDS uint32_t *litlen = (const RSQ_LDS uint32_t *)s_litlen, *dist = (const RSQ_LDS uint32_t *)s_dist;
    const RingText rt{ring};
    for (uint32_t i = tid; i < (1u << kHashBits); i += kThreads) head[i] = 0u;
    uint32_t word_base = 0, frac = 0;                                 // words of deflate data written, bits waiting in front of the buffer: the same in every thread
    bool overflow = false;                                            // a round needed more than its buffer: the piece is stored (the same in every thread)
    if (SAMPLE) {
        for (uint32_t i = tid; i < kLitLen + kDist; i += kThreads) s_hist[i] = 0u;
    } else {
        for (uint32_t i = tid; i < kLitLen; i += kThreads) s_litlen[i] = codes->litlen[i];
        if (tid < kDist) s_dist[tid] = codes->dist[tid];
        for (uint32_t w = tid; w < kOutWords + 2u; w += kThreads) out[w] = w * 32u < codes->header_bits ? codes->header[w] : 0u;      // the block header: the first bits of the data
        const uint32_t header_bits = codes->header_bits;
        word_base += flush_round(out, words, word_base, 0u, header_bits, false);
        frac = header_bits & 31u;
    }
    for (uint32_t round_lo = 0; round_lo < len; round_lo += kRound) {
        const uint32_t round_hi = round_lo + kRound < len ? round_lo + kRound : len;
        // the ring: this round and kAhead bytes behind it are new (the first round brings its own bytes as well), everything older stays
        {
            const uint32_t from = round_lo ? round_lo + kAhead : 0u, to = round_lo + kRound + kAhead < len ? round_lo + kRound + kAhead : len;
            __syncthreads();
            if (from < to) ring_load(ring, t, from, to);
            __syncthreads();
            if (tid < 16u) ring[kRing + tid] = ring[tid];
        }
        __syncthreads();
        for (uint32_t group = round_lo; group < round_hi; group += kThreads) {
            const uint32_t p = group + tid;
            const bool hashed = p + 4u <= len;
            const uint32_t v = rt.word(p), h = hashed ? hash4(v) : 0u, cand = hashed ? head[h] : 0u;      // (bytes behind the piece's end may be anything: they are never counted)
            __syncthreads();
            if (hashed) atomicMax(&head[h], p + 1u);
            if (p < round_lo + kRound) {
                const Found f = p < round_hi ? find_match(rt, len, round_lo, p, v, cand) : Found{0u, 0u};
                found[p - round_lo] = (uint16_t)(f.len ? ((f.len - 2u) << 11) | (f.dist - 1u) : 0u);
            }
            __syncthreads();
        }
        const uint32_t lo = round_lo + tid * kSeg, n_seg = lo >= round_hi ? 0u : (round_hi - lo < kSeg ? round_hi - lo : kSeg);
        const Segment g = load_segment(found, ring, round_lo, lo);
        const auto extend = [&](uint32_t i, uint32_t d, uint32_t so_far, uint32_t most) { return extend_match(rt, lo + i, d, so_far, most); };
        if (SAMPLE) {
            auto add = [&](uint32_t sym) { atomicAdd(&s_hist[sym], 1u); };
            HistogramSink<decltype(add)> sink{add};
            walk_segment(g, n_seg, sink, extend);
        } else {
            CountSink<const RSQ_LDS uint32_t *> count{litlen, dist};
            walk_segment(g, n_seg, count, extend);
            // exclusive scan of the segments' bits over the workgroup: within the wave by shuffles, the waves' totals through LDS
            uint32_t incl = count.bits;
            for (uint32_t d = 1; d < 64u; d *= 2) {
                const uint32_t other = (uint32_t)__shfl_up((int)incl, (int)d, 64);
                if ((tid & 63u) >= d) incl += other;
            }
            if ((tid & 63u) == 63u) s_wave[tid >> 6] = incl;
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (uint32_t w = 0; w < kThreads / 64u; ++w) {
                if (w < (tid >> 6)) before += s_wave[w];
                total += s_wave[w];
            }
            if (frac + total > kOutWords * 32u) overflow = true;
            if (!overflow) {
                BitSink<const RSQ_LDS uint32_t *, LdsOr> sink{litlen, dist, LdsOr{out}, frac + before + incl - count.bits};
                walk_segment(g, n_seg, sink, extend);
                word_base += flush_round(out, words, word_base, frac, total, false);
                frac = (frac + total) & 31u;
            }
        }
    }
    if (SAMPLE) {
        if (tid == 0) atomicAdd(&s_hist[256], 1u);
        __syncthreads();
        for (uint32_t i = tid; i < kLitLen + kDist; i += kThreads)
            if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
        return;
    }
    const uint32_t eob_bits = s_litlen[256] & 15u;
    if (tid == 0 && !overflow) {
        BitSink<const RSQ_LDS uint32_t *, LdsOr> sink{litlen, dist, LdsOr{out}, frac};
        sink.code(s_litlen[256]);
    }
    const uint64_t bits = (uint64_t)word_base * 32u + frac + eob_bits;
    if (!overflow) flush_round(out, words, word_base, frac, eob_bits, true);
    __syncthreads();
    // the ring is done with: its memory holds the CRC's tables
    const uint32_t crc = piece_crc(t, len, reinterpret_cast<RSQ_LDS uint32_t *>(ring), (RSQ_LDS uint32_t *)s_part);
    if (tid == 0) {
        const uint32_t data_bytes = (uint32_t)((bits + 7u) / 8u);
        if (overflow || data_bytes > 5u + len) sizes[piece] = 0u;      // no smaller than stored (a piece the code does not suit, or a few bytes behind a header of forty): stored
        else {
            member_frame(slot, data_bytes, crc, len);
            sizes[piece] = kHeaderBytes + data_bytes + kTrailerBytes;
        }
    }
}
// the pieces k_gzip_pieces gave up on (sizes 0), stored (BTYPE 00)
__global__ void __launch_bounds__(kThreads) k_gzip_stored(const uint8_t *text, uint64_t n, uint8_t *slots, uint32_t *sizes) {
    __shared__ uint32_t s_table[1024], s_part[kThreads];
    const uint64_t piece = blockIdx.x;
    if (sizes[piece]) return;
    const uint8_t *t = text + piece * kPiece;
    const uint32_t len = (uint32_t)(n - piece * kPiece < kPiece ? n - piece * kPiece : kPiece), tid = threadIdx.x;
    uint8_t *slot = slots + piece * kSlot, *d = slot + kSlotPad + kHeaderBytes;
    for (uint32_t i = tid; i < len; i += kThreads) d[5u + i] = t[i];
    const uint32_t crc = piece_crc(t, len, (RSQ_LDS uint32_t *)s_table, (RSQ_LDS uint32_t *)s_part);
    if (tid == 0) {
        d[0] = 1, d[1] = (uint8_t)len, d[2] = (uint8_t)(len >> 8), d[3] = (uint8_t)~len, d[4] = (uint8_t)(~len >> 8);
        member_frame(slot, 5u + len, crc, len);
        sizes[piece] = kHeaderBytes + 5u + len + kTrailerBytes;
    }
}
// the members out of their slots, one behind the other: member i to out + at[i]
__global__ void __launch_bounds__(256) k_gzip_compact(const uint8_t *slots, const uint32_t *sizes, const uint64_t *at, uint8_t *out) {
    const uint64_t piece = blockIdx.x;
    const uint8_t *src = slots + piece * kSlot + kSlotPad;
    uint8_t *dst = out + at[piece];
    const uint32_t size = sizes[piece];
    // the destination's first bytes up to a 4-byte boundary one by one, then words (the source words are read unaligned), then the tail
    const uint32_t lead = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u) < size ? (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u) : size;
    if (threadIdx.x < lead) dst[threadIdx.x] = src[threadIdx.x];
    const uint32_t n_words = (size - lead) / 4u;
    for (uint32_t w = threadIdx.x; w < n_words; w += 256u) reinterpret_cast<uint32_t *>(dst + lead)[w] = load4(src + lead + 4u * w);
    const uint32_t done = lead + 4u * n_words;
    if (threadIdx.x < size - done) dst[done + threadIdx.x] = src[done + threadIdx.x];
}
#endif

#if !defined(__HIPCC_RTC__)
}  // namespace gz
}  // namespace rsq
#include <algorithm>
#include <stdexcept>
#include <vector>
namespace rsq {
namespace gz {
// ---------------------------------------------------------------------------------------------------------------- host: the call's code
// Code lengths of at most `limit` bits for the counts (every symbol with a count gets a code): Huffman's algorithm on the sorted counts, lengths above the limit
// folded down and the Kraft sum repaired from the longest codes, the lengths then handed out by count.
inline std::vector<uint8_t> code_lengths(const std::vector<uint64_t> &count, uint32_t limit) {
    const size_t n = count.size();
    std::vector<uint32_t> order;
    for (size_t i = 0; i < n; ++i)
        if (count[i]) order.push_back((uint32_t)i);
    std::vector<uint8_t> length(n, 0);
    if (order.empty()) return length;
    if (order.size() == 1) {
        length[order[0]] = 1;
        return length;
    }
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return count[a] != count[b] ? count[a] < count[b] : a < b; });
    // two-queue Huffman: leaves in ascending order, inner nodes are born in ascending order
    const size_t m = order.size();
    std::vector<uint64_t> weight(2 * m - 1);
    std::vector<uint32_t> parent(2 * m - 1, 0);
    for (size_t i = 0; i < m; ++i) weight[i] = count[order[i]];
    size_t leaf = 0, inner = m, born = m;
    auto take = [&]() {
        if (leaf < m && (inner >= born || weight[leaf] <= weight[inner])) return leaf++;
        return inner++;
    };
    while (born < 2 * m - 1) {
        const size_t a = take(), b = take();
        weight[born] = weight[a] + weight[b];
        parent[a] = parent[b] = (uint32_t)born;
        ++born;
    }
    std::vector<uint32_t> depth(2 * m - 1, 0);
    for (size_t i = 2 * m - 2; i-- > 0;) depth[i] = depth[parent[i]] + 1u;
    std::vector<uint32_t> per_length(std::max<uint32_t>(limit, 64) + 1, 0);
    for (size_t i = 0; i < m; ++i) ++per_length[std::min<uint32_t>(depth[i], limit)];
    // Kraft: sum of 2^(limit - length) must be 2^limit
    uint64_t total = 0;
    for (uint32_t l = 1; l <= limit; ++l) total += (uint64_t)per_length[l] << (limit - l);
    while (total > ((uint64_t)1 << limit)) {
        --per_length[limit];
        for (uint32_t l = limit - 1; l >= 1; --l)
            if (per_length[l]) {
                --per_length[l];
                per_length[l + 1] += 2;
                break;
            }
        --total;
    }
    // the most frequent symbols take the shortest lengths
    size_t at = m;
    for (uint32_t l = 1; l <= limit; ++l)
        for (uint32_t k = 0; k < per_length[l]; ++k) length[order[--at]] = (uint8_t)l;
    return length;
}
// canonical codes (RFC 1951 3.2.2), bit-reversed, packed with their lengths
inline void canonical_codes(const std::vector<uint8_t> &length, uint32_t *entry) {
    uint32_t per_length[16] = {0}, next[16] = {0};
    for (uint8_t l : length) ++per_length[l];
    per_length[0] = 0;
    uint32_t code = 0;
    for (uint32_t l = 1; l < 16; ++l) {
        code = (code + per_length[l - 1]) << 1;
        next[l] = code;
    }
    for (size_t i = 0; i < length.size(); ++i) {
        const uint32_t l = length[i];
        uint32_t rev = 0;
        if (l) {
            const uint32_t c = next[l]++;
            for (uint32_t b = 0; b < l; ++b) rev |= ((c >> b) & 1u) << (l - 1u - b);
        }
        entry[i] = (rev << 4) | l;
    }
}
struct BitString {
    std::vector<uint32_t> words;
    uint32_t bits = 0;
    void push(uint32_t value, uint32_t n) {
        for (uint32_t i = 0; i < n; ++i, ++bits) {
            if (bits / 32 >= words.size()) words.push_back(0);
            words[bits / 32] |= ((value >> i) & 1u) << (bits % 32);
        }
    }
};
// The code of a call from the symbol counts of its sample: [0, 288) literals / lengths, [288, 320) distances.  Every symbol a piece may need gets a code (a count
// of at least one), the sample only decides which are short.
inline Codes build_codes(const uint32_t *sample) {
    std::vector<uint64_t> ll(286), dd(30);
    for (size_t i = 0; i < 286; ++i) ll[i] = (uint64_t)sample[i] * 16u + 1u;           // the sample counts outweigh the one that is there for the code's sake
    for (size_t i = 0; i < 30; ++i) dd[i] = (uint64_t)sample[kLitLen + i] * 16u + 1u;
    const std::vector<uint8_t> ll_len = code_lengths(ll, 15), dd_len = code_lengths(dd, 15);
    Codes c;
    memset(&c, 0, sizeof c);
    {
        std::vector<uint8_t> padded(ll_len);
        padded.resize(kLitLen, 0);
        canonical_codes(padded, c.litlen);
        padded.assign(dd_len.begin(), dd_len.end());
        padded.resize(kDist, 0);
        canonical_codes(padded, c.dist);
    }
    // the lengths of both codes as one sequence, run-length coded with the symbols 16 (repeat the last 3-6 times), 17 (3-10 zeros), 18 (11-138 zeros)
    std::vector<uint8_t> all(ll_len);
    all.insert(all.end(), dd_len.begin(), dd_len.end());
    struct Item {
        uint8_t symbol, extra_bits, extra;
    };
    std::vector<Item> items;
    for (size_t i = 0; i < all.size();) {
        size_t run = 1;
        while (i + run < all.size() && all[i + run] == all[i]) ++run;
        if (all[i] == 0 && run >= 3) {
            const size_t take = std::min<size_t>(run, 138);
            items.push_back(take >= 11 ? Item{18, 7, (uint8_t)(take - 11)} : Item{17, 3, (uint8_t)(take - 3)});
            i += take;
        } else if (all[i] != 0 && run >= 4) {
            items.push_back(Item{all[i], 0, 0});
            const size_t take = std::min<size_t>(run - 1, 6);
            items.push_back(Item{16, 2, (uint8_t)(take - 3)});
            i += 1 + take;
        } else {
            items.push_back(Item{all[i], 0, 0});
            ++i;
        }
    }
    std::vector<uint64_t> cl_count(19, 0);
    for (const Item &it : items) ++cl_count[it.symbol];
    const std::vector<uint8_t> cl_len = code_lengths(cl_count, 7);
    uint32_t cl_code[19];
    canonical_codes(cl_len, cl_code);
    static const uint8_t kOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint32_t hclen = 19;
    while (hclen > 4 && cl_len[kOrder[hclen - 1]] == 0) --hclen;
    BitString b;
    b.push(1, 1);                        // BFINAL: the member's only block
    b.push(2, 2);                        // BTYPE 10: dynamic Huffman codes
    b.push(286 - 257, 5);
    b.push(30 - 1, 5);
    b.push(hclen - 4, 4);
    for (uint32_t i = 0; i < hclen; ++i) b.push(cl_len[kOrder[i]], 3);
    for (const Item &it : items) {
        b.push(cl_code[it.symbol] >> 4, cl_code[it.symbol] & 15u);
        if (it.extra_bits) b.push(it.extra, it.extra_bits);
    }
    if (b.words.size() > kHeaderWords) throw std::runtime_error("internal: the deflate block header does not fit its array");
    c.header_bits = b.bits;
    std::copy(b.words.begin(), b.words.end(), c.header);
    return c;
}

// ---------------------------------------------------------------------------------------- host: a piece, thread by thread (tests/hostemu)
// The device's walk with the workgroup's threads taken one after the other: `hist` != nullptr counts the piece's symbols (the sample), else the member is written
// to out (kSlot bytes, zeroed here); returns the member's size, 0 where the device gives the piece up (a round's bits beyond its buffer, or no smaller than stored).
inline uint32_t piece_on_the_host(const uint8_t *text, uint32_t n, const Codes *codes, uint8_t *out, uint32_t *hist) {
    std::vector<uint32_t> head((size_t)1 << kHashBits, 0);
    std::vector<uint16_t> found(kRound);
    std::vector<uint32_t> data((size_t)kSlotWords + kOutWords + 4, 0);      // the deflate data, all of it in one buffer of bits
    struct Or {
        uint32_t *words;
        void or_word(uint32_t w, uint32_t v) { words[w] |= v; }
    };
    uint64_t bit = 0;
    bool overflow = false;
    if (!hist) {
        for (uint32_t w = 0; w * 32u < codes->header_bits; ++w) data[w] = codes->header[w];
        bit = codes->header_bits;
    }
    const PlainText pt{text, n};
    for (uint32_t round_lo = 0; round_lo < n; round_lo += kRound) {
        const uint32_t round_hi = std::min(n, round_lo + kRound);
        std::fill(found.begin(), found.end(), 0);
        for (uint32_t group = round_lo; group < round_hi; group += kThreads) {              // phase A, kThreads positions at a time
            uint32_t cand[kThreads];
            for (uint32_t t = 0; t < kThreads; ++t) {
                const uint32_t p = group + t;
                cand[t] = p + 4u <= n ? head[hash4(load4(text + p))] : 0u;
            }
            for (uint32_t t = 0; t < kThreads; ++t) {
                const uint32_t p = group + t;
                if (p + 4u <= n) {
                    uint32_t &h = head[hash4(load4(text + p))];
                    h = std::max(h, p + 1u);
                }
            }
            for (uint32_t t = 0; t < kThreads; ++t) {
                const uint32_t p = group + t;
                if (p >= round_hi) break;
                const Found f = find_match(pt, n, round_lo, p, pt.word(p), cand[t]);
                found[p - round_lo] = (uint16_t)(f.len ? ((f.len - 2u) << 11) | (f.dist - 1u) : 0u);
            }
        }
        uint32_t round_bits = 0;
        std::vector<Segment> segs(kThreads);
        std::vector<uint32_t> seg_n(kThreads, 0), seg_bits(kThreads, 0);
        for (uint32_t t = 0; t < kThreads; ++t) {                                            // phase B: the segments into "registers", the counts
            const uint32_t lo = round_lo + t * kSeg;
            seg_n[t] = lo >= round_hi ? 0u : std::min(kSeg, round_hi - lo);
            Segment &g = segs[t];
            memset(&g, 0, sizeof g);
            const auto extend = [&](uint32_t i, uint32_t d, uint32_t so_far, uint32_t most) { return extend_match(pt, lo + i, d, so_far, most); };
            for (uint32_t i = 0; i < kSeg; ++i) {
                g.found[i >> 1] |= (uint32_t)found[lo - round_lo + i] << ((i & 1u) * 16u);
                if (lo + i < n) g.text[i >> 2] |= (uint32_t)text[lo + i] << ((i & 3u) * 8u);
            }
            if (hist) {
                auto add = [hist](uint32_t s) { ++hist[s]; };
                HistogramSink<decltype(add)> sink{add};
                walk_segment(g, seg_n[t], sink, extend);
            } else {
                CountSink<const uint32_t *> count{codes->litlen, codes->dist};
                walk_segment(g, seg_n[t], count, extend);
                seg_bits[t] = count.bits;
                round_bits += count.bits;
            }
        }
        if (hist) continue;
        if ((bit & 31u) + round_bits > kOutWords * 32u) overflow = true;                     // the device's buffer for a round's bits
        if (overflow) continue;
        for (uint32_t t = 0; t < kThreads; ++t) {
            BitSink<const uint32_t *, Or> sink{codes->litlen, codes->dist, Or{data.data() + (bit >> 5)}, (uint32_t)(bit & 31u)};
            const uint32_t lo = round_lo + t * kSeg;
            const auto extend = [&](uint32_t i, uint32_t d, uint32_t so_far, uint32_t most) { return extend_match(pt, lo + i, d, so_far, most); };
            walk_segment(segs[t], seg_n[t], sink, extend);
            bit += seg_bits[t];
        }
    }
    if (hist) {
        ++hist[256];
        return 0;
    }
    if (!overflow) {
        BitSink<const uint32_t *, Or> sink{codes->litlen, codes->dist, Or{data.data() + (bit >> 5)}, (uint32_t)(bit & 31u)};
        sink.code(codes->litlen[256]);
    }
    bit += codes->litlen[256] & 15u;
    const uint32_t data_bytes = (uint32_t)((bit + 7u) / 8u), member = kHeaderBytes + data_bytes + kTrailerBytes;
    if (overflow || data_bytes > 5u + n) return 0;                   // as the device decides: no smaller than stored
    memset(out, 0, kSlot);
    memcpy(out + kSlotPad + kHeaderBytes, data.data(), data_bytes);
    uint32_t table[256];
    for (uint32_t i = 0; i < 256; ++i) table[i] = crc_table_entry(i);
    // the CRC as the device folds it: a slice per thread (the first takes the remainder), pairs folded level by level with x^(8 L 2^k)
    const uint32_t L = n / kThreads, first = n - (kThreads - 1u) * L;
    uint32_t part[kThreads];
    for (uint32_t t = 0; t < kThreads; ++t) part[t] = t == 0 ? crc32_bytes(table, text, first) : crc32_bytes(table, text + first + (t - 1u) * L, L);
    uint32_t x = x_to_8n_modp(L);
    for (uint32_t width = 1; width < kThreads; width *= 2) {
        for (uint32_t t = 0; t < kThreads; t += 2 * width) part[t] = crc_combine(part[t], part[t + width], x);
        x = multmodp(x, x);
    }
    member_header(out + kSlotPad, member);
    uint8_t *tail = out + kSlotPad + kHeaderBytes + data_bytes;
    for (int i = 0; i < 4; ++i) {
        tail[i] = (uint8_t)(part[0] >> (8 * i));
        tail[4 + i] = (uint8_t)(n >> (8 * i));
    }
    return member;
}
// a piece without compression: one stored block (BTYPE 00), what k_gzip_stored writes for a piece whose code does not suit it
inline uint32_t stored_piece_on_the_host(const uint8_t *text, uint32_t n, uint8_t *out) {
    const uint32_t member = kHeaderBytes + 5u + n + kTrailerBytes;
    memset(out, 0, kSlot);
    member_header(out + kSlotPad, member);
    uint8_t *d = out + kSlotPad + kHeaderBytes;
    d[0] = 1;                            // BFINAL, BTYPE 00
    d[1] = (uint8_t)n, d[2] = (uint8_t)(n >> 8), d[3] = (uint8_t)~n, d[4] = (uint8_t)(~n >> 8);
    memcpy(d + 5, text, n);
    uint32_t table[256];
    for (uint32_t i = 0; i < 256; ++i) table[i] = crc_table_entry(i);
    const uint32_t crc = crc32_bytes(table, text, n);
    for (int i = 0; i < 4; ++i) {
        d[5 + n + i] = (uint8_t)(crc >> (8 * i));
        d[9 + n + i] = (uint8_t)(n >> (8 * i));
    }
    return member;
}
// which pieces of a call are the sample: at most 64, spread evenly
inline uint32_t sample_stride(uint64_t n_pieces) { return (uint32_t)std::max<uint64_t>(1, (n_pieces + 63) / 64); }
// the whole call on the host (tests/hostemu): text -> members, appended to out
inline void gzip_on_the_host(const uint8_t *text, uint64_t n, std::vector<uint8_t> &out) {
    const uint64_t n_pieces = (n + kPiece - 1) / kPiece;
    std::vector<uint32_t> hist(kLitLen + kDist, 0);
    const uint32_t stride = sample_stride(n_pieces);
    for (uint64_t i = 0; i < n_pieces; i += stride) piece_on_the_host(text + i * kPiece, (uint32_t)std::min<uint64_t>(kPiece, n - i * kPiece), nullptr, nullptr, hist.data());
    const Codes codes = build_codes(hist.data());
    std::vector<uint8_t> slot(kSlot);
    for (uint64_t i = 0; i < n_pieces; ++i) {
        const uint32_t len = (uint32_t)std::min<uint64_t>(kPiece, n - i * kPiece);
        uint32_t member = piece_on_the_host(text + i * kPiece, len, &codes, slot.data(), nullptr);
        if (!member) member = stored_piece_on_the_host(text + i * kPiece, len, slot.data());
        out.insert(out.end(), slot.begin() + kSlotPad, slot.begin() + kSlotPad + member);
    }
}
#endif

}  // namespace gz
}  // namespace rsq
