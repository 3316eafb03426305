// rsq_deflate.h -- gzip of FASTQ text that lies in device memory (SURVEY.md section 8(f) item 4 "gz FASTQ output"; the reference writes .gz through SeqAn's
// compressed stream from Simulator::Flush, reseq/Simulator.cpp:150-182, output names main.cpp:404,412).
//
// The text is cut into pieces of kPiece bytes; every piece becomes one gzip member (RFC 1952) of its own, framed like a BGZF block (an extra field "BC" with the
// member's size, input at most 65280 bytes), so that the members can be made by independent workgroups, concatenate in any grouping (ranks, batches) and are read
// by zlib's gzread, gzip -d, SeqAn and by the block-wise readers of bgzip / htslib alike.  Inside a member ONE deflate block (RFC 1951) with a dynamic Huffman
// code that is NOT the piece's own: the code is built on the host once per call from the symbol counts of a sample of the call's pieces (FASTQ text is stationary:
// the same ids, four bases, forty qualities everywhere) -- so the device never builds a code, and a piece's encoding needs nothing of another piece.
//
// A piece on the device, 256 threads (k_gzip_pieces), three or four workgroups per CU (40 KB of LDS each):
//   rounds of 8 KB; the round's text -- with 272 bytes beyond it and what came before -- stands in a 16 KB ring in LDS (global loads of 16 bytes, once); per round
//   L  the LINES: FASTQ text is searched for matches only where matches are -- the lines that begin with '@' and the lines behind them (ids and bases); quality lines
//      are coded as literals and runs (see "FASTQ text by its lines" below).  Which line a position lies in comes out of one scan over the workgroup;
//   A1 every searched position enters a hash table of positions in LDS, keyed by SIX bytes, 256 positions at a time (read all, barrier, atomicMax all: deterministic);
//      what the table held before is the position's candidate;
//   A2 every fourth searched position is PROBED, a thread per probe (all lanes at work): the candidate verified and measured to its end (the end of the 32-byte
//      segment at most: segments are coded independently), and so a run of the byte before it; 16 bits per probe.  The three positions behind a probed one inherit
//      what is left of its match;
//   B  thread t owns the 32-byte segment t of the round: its bytes and its eight probes' matches in registers, an unrolled walk takes them greedily (as zlib's level
//      1 does) without touching the text again -- once to count the bits, then, after a scan of the counts over the workgroup, to OR the codes LSB-first into the
//      round's bit buffer in LDS (ds_or: neighbours share a word where their bits meet); the buffer's complete words go to the member in HBM in coalesced stores, its
//      last, incomplete word to the front of the next round's buffer;
//   the end-of-block code, the CRC-32 of the text (slices per thread, four bytes per step, folded with x^(8 len) mod P like zlib's crc32_combine) and the trailer.
// A piece whose bits come to no less than the piece itself (text that looks nothing like the sample, or a few bytes behind the block header), or a round of which
// needs more than 8 bits per byte, is stored instead (BTYPE 00) by a second kernel: text + 31 bytes bound every member, and the slot holds that.
// Measured on the simulator's own FASTQ text (P0, 804 MB): 182 GB/s by kernel time, 2.78 x smaller (zlib level 1: 3.18 x, level 6: 3.71 x) -- profiles/r06_*.
// Round 5's kernel searched and probed every position of every line with a four-byte key and measured a match in the walk, twice: 90 GB/s, 2.75 x.  What made the
// difference (tools/micro/gzip_bench.hip, clocks per phase): the walks no longer read LDS text in divergent loops, the found array is an eighth (four workgroups
// per CU), x^(8 len) is a constant instead of 44 multiplications per thread, a fifth of the probes; the six-byte key found the better candidates.
// Text with a handful of quality values and short reads (the test profile TINY) repeats in short stretches everywhere, which fall between the probes of that rule (1.29 x
// zlib level 1).  The call's sample is therefore walked a second way -- the DENSE route, template parameter STEP = kDenseStep: every position of every line searched
// and probed, a four-byte key, which is round 5's search in this round's kernel -- and gz::dense_pays prices both samples with their own codes: the dense route is taken
// where it makes the sample more than 5 % smaller (TINY: 1.07 x zlib level 1 at 94 GB/s; P0's text stays with the lines: forced dense it gives round 5's bytes exactly,
// 1.156 x, at 93 GB/s).  zlib on host threads (option host_gzip, `reseq --hostGzip`) remains for whoever wants the smallest file.
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
constexpr uint32_t kMinRun = 4;                    // shortest run taken as a match one byte back in a line that is not searched ("FASTQ text by its lines" below)
constexpr uint32_t kProbeStep = 4;                 // every so many positions of a searched line are probed ...
constexpr uint32_t kDenseStep = 1;                 // ... or, for text with little to tell its symbols apart (dense_pays), every position of every line, with a four-byte key
static_assert(kSeg % (2u * kProbeStep) == 0, "a segment's entries of the found array fill whole words");
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
    uint32_t dense;                    // the code was built for (and the pieces are to be walked by) the dense route: kDenseStep
};

RSQ_HD uint32_t hash4(uint32_t v) { return (v * 2654435761u) >> (32u - kHashBits); }
// The table is entered with SIX bytes (v = the four a match is verified on first, more = the four behind them): among four letters any four bytes recur within the
// window by chance, and the chance candidate -- the latest -- hides the read that truly overlaps; six bytes recur by chance once in 4096 positions.  (Measured on P0's
// text: 2.74 x smaller with four bytes, 2.91 x with six, 2.89 x with eight; the shortest match found this way is six bytes, which is about where a match begins to pay.)
RSQ_HD uint32_t hash6(uint32_t v, uint32_t more) { return hash4(v ^ ((more & 0xFFFFu) * 0x9E3779B1u)); }
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
template <uint32_t STEP>
struct SegmentT {
    uint32_t found[kSeg / STEP / 2u], text[kSeg / 4u];      // the probed positions' entries (found_at), the bytes
    uint32_t kind;                       // bit i: position i lies in a line that is searched for matches (line_kinds); else only runs of its own bytes count
    uint32_t eq;                         // bit i (i >= 1): byte i equals byte i - 1 of the segment
};
// ---- FASTQ text by its lines.  A record is an id line, 150 bases, "+", 150 qualities.  Matches worth their price exist between the id lines and between the base
// lines of neighbouring records (the simulator writes its pairs in the order of their start positions: neighbours overlap); a quality line repeats nothing but, now
// and then, a byte.  So the lines that begin with '@' and the lines behind them are SEARCHED (hashed, probed: phase A), every other line is coded as literals and as
// runs of one byte (a match one byte back, found in the segment's own registers).  The rule needs no knowledge of records: a quality line that happens to begin with
// '@' is searched too, and so is the "+" behind it -- the choice only ever costs size, never correctness.  Of the searched positions every fourth is PROBED (probe); the three behind it inherit its match, a byte shorter each -- the bytes of a match that has been
// compared are equal wherever one starts among them.  What is lost is a match that begins on a position that is not probed and whose predecessor found nothing: it
// begins up to three literals later.
RSQ_HD uint32_t zero_bytes_of(uint32_t x) {        // 4 bits: byte k of x is zero
    const uint32_t t = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);      // 0x80 in every zero byte
    return ((t >> 7) & 1u) | ((t >> 14) & 2u) | ((t >> 21) & 4u) | ((t >> 28) & 8u);
}
RSQ_HD uint32_t bytes_equal_to(const uint32_t *text, uint32_t c) {                   // bit i: byte i of the segment's text is c
    uint32_t m = 0;
    for (uint32_t k = 0; k < kSeg / 4u; ++k) m |= zero_bytes_of(text[k] ^ (c * 0x01010101u)) << (4u * k);
    return m;
}
RSQ_HD uint32_t bytes_equal_to_previous(const uint32_t *text) {                      // bit i: byte i equals byte i - 1 (bit 0: never)
    uint32_t m = 0;
    for (uint32_t k = 0; k < kSeg / 4u; ++k) {
        const uint32_t before = k ? text[k - 1u] >> 24 : (~text[0]) & 0xFFu;
        m |= zero_bytes_of(text[k] ^ ((text[k] << 8) | before)) << (4u * k);
    }
    return m;
}
RSQ_HD uint32_t count_trailing_zeros(uint32_t x) { return (uint32_t)__builtin_ctz(x); }
// The state of the line a position lies in: bit 0 = the line is searched, bit 1 = it began with '@'.  A line that begins with byte c behind a line in state s is in
// state next_line_state(s, c == '@'): searched iff it or the line before it began with '@'.  kLineStateAtStart: the (partial) line a piece begins in, unknown, is searched
// and so is the line behind it.
constexpr uint32_t kLineStateAtStart = 3u;
RSQ_HD uint32_t next_line_state(uint32_t s, bool at) { return (at ? 3u : 0u) | (s >> 1); }
// `starts`: bit i = a line begins at position i of the segment (the byte before it is a newline); `at`: bytes_equal_to '@'.  The positions in searched lines, given the
// state the segment is entered in:
RSQ_HD uint32_t line_kinds(uint32_t state, uint32_t starts, uint32_t at) {
    uint32_t kind = state & 1u ? 0xFFFFFFFFu : 0u;
    for (; starts; starts &= starts - 1u) {
        const uint32_t i = count_trailing_zeros(starts), upper = 0xFFFFFFFFu << i;
        state = next_line_state(state, (at >> i) & 1u);
        kind = state & 1u ? kind | upper : kind & ~upper;
    }
    return kind;
}
// ... and what the segment does to the state, as a table of four 2-bit entries (entry s at bits 2s): the segments' tables compose (line_compose), so the state in
// front of every segment of a round comes out of one scan over the workgroup
constexpr uint32_t kLineIdentity = 0xE4u;
RSQ_HD uint32_t line_transfer(uint32_t starts, uint32_t at) {
    uint32_t table = 0;
    for (uint32_t s = 0; s < 4u; ++s) {
        uint32_t state = s;
        for (uint32_t m = starts; m; m &= m - 1u) state = next_line_state(state, (at >> count_trailing_zeros(m)) & 1u);
        table |= state << (2u * s);
    }
    return table;
}
RSQ_HD uint32_t line_compose(uint32_t first, uint32_t then) {
    uint32_t table = 0;
    for (uint32_t s = 0; s < 4u; ++s) table |= ((then >> (2u * ((first >> (2u * s)) & 3u))) & 3u) << (2u * s);
    return table;
}
RSQ_HD uint32_t line_apply(uint32_t table, uint32_t state) { return (table >> (2u * state)) & 3u; }
// A probed position's entry of the found array, 16 bits: 0 = no match, else bits 0-10 distance - 1, bits 11-15 length - 2 (the match measured to its end: at most the
// rest of the segment, 32 bytes).  The kProbeStep - 1 positions behind it have no entry: they inherit (found_at).
RSQ_HD uint32_t pack_found(uint32_t len, uint32_t dist) { return len >= kMinMatch ? ((len - 2u) << 11) | (dist - 1u) : 0u; }
// the probe of position p (a multiple of kProbeStep in its segment); cand_dist: how far back the hash table's candidate for p's bytes lies (0: none).  The candidate
// is verified on its first four bytes and measured to its end; so is a run (the byte before p three times more).  A match ends with the segment of p.
template <class Text>
RSQ_HD uint32_t probe(const Text &text, uint32_t n, uint32_t round_lo, uint32_t p, uint32_t cand_dist) {
    if (p + kMinMatch > n) return 0u;
    const uint32_t segment_end = round_lo + ((p - round_lo) / kSeg + 1u) * kSeg;
    const uint32_t limit = n - p < segment_end - p ? n - p : segment_end - p;
    if (limit < kMinMatch) return 0u;
    const uint32_t v = text.word(p);
    uint32_t len = 0, dist = 0;
    if (cand_dist && limit >= 4u && text.word(p - cand_dist) == v) {
        len = extend_match(text, p, cand_dist, 4u, limit);
        dist = cand_dist;
    }
    if (p && len < limit && ((text.word(p - 1u) ^ v) & 0xFFFFFFu) == 0u) {
        const uint32_t run = extend_match(text, p, 1u, 3u, limit);
        if (run > len) len = run, dist = 1u;
    }
    return pack_found(len, dist);
}
// what position i of a segment finds: its own entry if it is probed, else what is left of the entry of the probed position before it.  found: the segment's
// kSeg / kProbeStep entries, two per word.  (length, distance); length 0 = nothing
template <uint32_t STEP>
RSQ_HD void found_at(const uint32_t *found, uint32_t i, uint32_t &len, uint32_t &dist) {
    const uint32_t j = i / STEP, k = i % STEP, e = (found[j >> 1] >> ((j & 1u) * 16u)) & 0xFFFFu, parent = e ? (e >> 11) + 2u : 0u;
    len = parent >= kMinMatch + k ? parent - k : 0u;
    dist = (e & 2047u) + 1u;
}
template <uint32_t STEP, class Sink>
RSQ_HD void walk_segment(const SegmentT<STEP> &g, uint32_t n, Sink &sink) {
    uint32_t skip = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t i = 0; i < kSeg; ++i) {
        const uint32_t c = (g.text[i >> 2] >> ((i & 3u) * 8u)) & 0xFFu;
        const bool searched = (g.kind >> i) & 1u;
        const uint32_t run = i ? count_trailing_zeros(~(g.eq >> i)) : 0u;             // bytes from i on that equal byte i - 1
        uint32_t len, dist;
        found_at<STEP>(g.found, i, len, dist);
        if (!searched) len = run >= kMinRun ? run : 0u, dist = 1u;
        if (i + len > n) len = n > i ? n - i : 0u;
        const bool here = skip == 0u && i < n, is_match = len >= kMinMatch;
        if (here) {
            if (is_match) sink.match(len, dist);
            else sink.literal(c);
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
constexpr uint32_t multmodp_c(uint32_t a, uint32_t b) {           // multmodp for constant expressions
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
constexpr uint32_t x_to_8n_modp_c(uint32_t n_bytes) {
    uint32_t result = 1u << 31, square = 1u << 30;
    for (uint64_t e = (uint64_t)n_bytes * 8u; e; e >>= 1) {
        if (e & 1u) result = multmodp_c(square, result);
        square = multmodp_c(square, square);
    }
    return result;
}
constexpr uint32_t kFullSlice = kPiece / kThreads;              // bytes of text per thread of a whole piece (the CRC's slices)
constexpr uint32_t kXFullSlice = x_to_8n_modp_c(kFullSlice);    // x^(8 kFullSlice) mod P: every thread of every whole piece needed it, 44 multiplications mod P each
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
    uint32_t x = L == kFullSlice ? kXFullSlice : x_to_8n_modp(L);
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
// RSQ_GZ_TRACE (tools/micro/gzip_bench.hip): thread 0 of every workgroup adds the shader clocks it spent in each phase to hist[phase] (the sample kernel's argument, unused
// by the other)
#if defined(RSQ_GZ_TRACE)
#define RSQ_GZ_MARK(phase)                                              \
    do {                                                                \
        const unsigned long long now_ = clock64();                      \
        if (!SAMPLE && tid == 0) trace_[phase] += now_ - mark_;         \
        mark_ = now_;                                                   \
    } while (0)
#else
#define RSQ_GZ_MARK(phase)
#endif
template <bool SAMPLE, uint32_t STEP = kProbeStep>     // STEP = kProbeStep: FASTQ by its lines; kDenseStep: every position of every line searched and probed, a four-byte key
__global__ void __launch_bounds__(kThreads) k_gzip_pieces(const uint8_t *text, uint64_t n, uint32_t piece_step, const Codes *codes, uint8_t *slots, uint32_t *sizes, uint32_t *hist) {
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[kRing + 16u];
    __shared__ __attribute__((aligned(16))) uint16_t s_found[kRound / STEP];      // an entry per probed position
    __shared__ uint32_t head[1u << kHashBits];
    __shared__ uint32_t s_out[SAMPLE ? 1u : kOutWords + 2u], s_litlen[SAMPLE ? 1u : kLitLen], s_dist[SAMPLE ? 1u : kDist], s_hist[SAMPLE ? kLitLen + kDist : 1u];
    __shared__ uint32_t s_part[kThreads], s_wave[kThreads / 64u], s_kind[kThreads];
    __shared__ uint16_t s_list[SAMPLE && STEP > 1u ? kRound / STEP : 1u];      // the positions to probe: in the sample kernel an array of its own, else inside the round's bit buffer (dense: every position, no list)
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
    static_assert(STEP == 1u || kRound / STEP <= 2u * kOutWords, "the probe list fits the round's bit buffer behind its first word");
    RSQ_LDS uint16_t *list = SAMPLE ? (RSQ_LDS uint16_t *)s_list : reinterpret_cast<RSQ_LDS uint16_t *>(out + 1);
    uint32_t line_state = kLineStateAtStart, n_probes = 0;                 // the same in every thread
#if defined(RSQ_GZ_TRACE)
    unsigned long long trace_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mark_ = clock64();
#endif
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
        RSQ_GZ_MARK(0);
        // ---- the lines of the round: which positions of the thread's segment are searched (the state in front of it: a scan of the segments' tables over the workgroup),
        // the bytes that repeat their predecessor, and the list of the positions to probe
        const uint32_t lo = round_lo + tid * kSeg, n_seg = lo >= round_hi ? 0u : (round_hi - lo < kSeg ? round_hi - lo : kSeg);
        SegmentT<STEP> g;
        {
            const RSQ_LDS uint4 *t16 = reinterpret_cast<const RSQ_LDS uint4 *>(ring + (lo & (kRing - 1u)));
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint4 v = t16[k];
                g.text[4 * k] = v.x, g.text[4 * k + 1] = v.y, g.text[4 * k + 2] = v.z, g.text[4 * k + 3] = v.w;
            }
        }
        if constexpr (STEP == 1u) {                                          // dense: every position is searched and probed
            g.kind = n_seg >= kSeg ? 0xFFFFFFFFu : (1u << n_seg) - 1u;
            g.eq = 0u;
            n_probes = round_hi - round_lo;
        } else {
            const uint32_t valid = n_seg >= kSeg ? 0xFFFFFFFFu : (1u << n_seg) - 1u;
            const uint32_t starts = ((bytes_equal_to(g.text, '\n') << 1) | (lo && n_seg && rt.byte(lo - 1u) == '\n' ? 1u : 0u)) & valid, at = bytes_equal_to(g.text, '@');
            uint32_t incl = line_transfer(starts, at);                     // of the segments up to and including this one, within the wave
            for (uint32_t d = 1; d < 64u; d *= 2) {
                const uint32_t before = (uint32_t)__shfl_up((int)incl, (int)d, 64);
                if ((tid & 63u) >= d) incl = line_compose(before, incl);
            }
            uint32_t excl = (uint32_t)__shfl_up((int)incl, 1, 64);
            if ((tid & 63u) == 0u) excl = kLineIdentity;
            if ((tid & 63u) == 63u) s_wave[tid >> 6] = incl;
            __syncthreads();
            uint32_t state = line_state;                                   // in front of this thread's wave, then of its segment
            for (uint32_t w = 0; w < (tid >> 6); ++w) state = line_apply(s_wave[w], state);
            uint32_t after = line_state;
            for (uint32_t w = 0; w < kThreads / 64u; ++w) after = line_apply(s_wave[w], after);
            line_state = after;                                            // in front of the next round: the same in every thread
            g.kind = line_kinds(line_apply(excl, state), starts, at) & valid;
            g.eq = bytes_equal_to_previous(g.text);
            s_kind[tid] = g.kind;
            const uint32_t mine = g.kind & 0x11111111u;                     // the probed positions of the segment: every kProbeStep-th
            uint32_t upto = (uint32_t)__builtin_popcount(mine);
            const uint32_t own = upto;
            for (uint32_t d = 1; d < 64u; d *= 2) {
                const uint32_t other = (uint32_t)__shfl_up((int)upto, (int)d, 64);
                if ((tid & 63u) >= d) upto += other;
            }
            __syncthreads();                                               // s_wave has been read
            if ((tid & 63u) == 63u) s_wave[tid >> 6] = upto;
            __syncthreads();
            uint32_t at_list = upto - own;
            n_probes = 0;
            for (uint32_t w = 0; w < kThreads / 64u; ++w) {
                if (w < (tid >> 6)) at_list += s_wave[w];
                n_probes += s_wave[w];
            }
            for (uint32_t m = mine; m; m &= m - 1u) list[at_list++] = (uint16_t)(tid * kSeg + count_trailing_zeros(m));
        }
        __syncthreads();
        RSQ_GZ_MARK(1);
        // ---- phase A1: the searched positions enter the hash table, kThreads positions at a time; what the table held for a position's six bytes before its group
        // entered is its candidate, kept as a distance (for the probed positions: the others inherit)
        for (uint32_t group = round_lo; group < round_hi; group += kThreads) {
            const uint32_t p = group + tid, rel = p - round_lo;
            const bool hashed = p + 4u <= len && (STEP == 1u || ((s_kind[rel / kSeg] >> (rel % kSeg)) & 1u));      // (bytes behind the piece's end may be anything: they are never counted)
            const uint32_t h = hashed ? (STEP == 1u ? hash4(rt.word(p)) : hash6(rt.word(p), rt.word(p + 4u))) : 0u, cand = hashed ? head[h] : 0u;
            __syncthreads();
            if (hashed) atomicMax(&head[h], p + 1u);
            if (p % STEP == 0u && rel < kRound) found[rel / STEP] = (uint16_t)(cand && p + 1u - cand <= kMaxDist ? p + 1u - cand : 0u);
            __syncthreads();
        }
        RSQ_GZ_MARK(2);
        // ---- phase A2: the probes, a thread each (all lanes at work), and what the positions behind them inherit
        for (uint32_t k = tid; k < n_probes; k += kThreads) {
            const uint32_t rel = STEP == 1u ? k : list[k];
            found[rel / STEP] = (uint16_t)probe(rt, len, round_lo, round_lo + rel, found[rel / STEP]);
        }
        __syncthreads();
        if (!SAMPLE && STEP > 1u)                                          // the list stood in the round's bit buffer (behind its first word): zero again
            for (uint32_t w = 1u + tid; w < 2u + n_probes / 2u; w += kThreads) out[w] = 0u;
        {
            const RSQ_LDS uint4 *f16 = reinterpret_cast<const RSQ_LDS uint4 *>(found + (lo - round_lo) / STEP);
#pragma unroll
            for (uint32_t k = 0; k < kSeg / STEP / 8u; ++k) {
                const uint4 v = f16[k];
                g.found[4 * k] = v.x, g.found[4 * k + 1] = v.y, g.found[4 * k + 2] = v.z, g.found[4 * k + 3] = v.w;
            }
        }
        __syncthreads();                                                   // (the buffer is zero before anybody's bits go in)
        RSQ_GZ_MARK(3);
        if (SAMPLE) {
            auto add = [&](uint32_t sym) { atomicAdd(&s_hist[sym], 1u); };
            HistogramSink<decltype(add)> sink{add};
            walk_segment(g, n_seg, sink);
        } else {
            CountSink<const RSQ_LDS uint32_t *> count{litlen, dist};
            walk_segment(g, n_seg, count);
            // exclusive scan of the segments' bits over the workgroup: within the wave by shuffles, the waves' totals through LDS
            uint32_t incl = count.bits;
            for (uint32_t d = 1; d < 64u; d *= 2) {
                const uint32_t other = (uint32_t)__shfl_up((int)incl, (int)d, 64);
                if ((tid & 63u) >= d) incl += other;
            }
            if ((tid & 63u) == 63u) s_wave[tid >> 6] = incl;
            __syncthreads();
            RSQ_GZ_MARK(4);
            uint32_t before = 0, total = 0;
            for (uint32_t w = 0; w < kThreads / 64u; ++w) {
                if (w < (tid >> 6)) before += s_wave[w];
                total += s_wave[w];
            }
            if (frac + total > kOutWords * 32u) overflow = true;
            if (!overflow) {
                BitSink<const RSQ_LDS uint32_t *, LdsOr> sink{litlen, dist, LdsOr{out}, frac + before + incl - count.bits};
                walk_segment(g, n_seg, sink);
                RSQ_GZ_MARK(5);
                word_base += flush_round(out, words, word_base, frac, total, false);
                frac = (frac + total) & 31u;
                RSQ_GZ_MARK(6);
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
    RSQ_GZ_MARK(7);
#if defined(RSQ_GZ_TRACE)
    if (!SAMPLE && tid == 0 && hist)
        for (int i = 0; i < 8; ++i) atomicAdd(reinterpret_cast<unsigned long long *>(hist) + i, trace_[i]);
#endif
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
#include <cmath>
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
template <uint32_t STEP>
inline uint32_t piece_on_the_host_t(const uint8_t *text, uint32_t n, const Codes *codes, uint8_t *out, uint32_t *hist) {
    std::vector<uint32_t> head((size_t)1 << kHashBits, 0);
    std::vector<uint16_t> found(kRound / STEP);
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
    uint32_t line_state = kLineStateAtStart;
    for (uint32_t round_lo = 0; round_lo < n; round_lo += kRound) {
        const uint32_t round_hi = std::min(n, round_lo + kRound);
        std::fill(found.begin(), found.end(), 0);
        // the lines of the round: per segment which of its positions are searched, and the bytes that repeat the byte before them
        std::vector<uint32_t> kinds(kThreads, 0), eqs(kThreads, 0);
        for (uint32_t t = 0; t < kThreads; ++t) {
            const uint32_t lo = round_lo + t * kSeg;
            uint32_t words[kSeg / 4u] = {0};
            for (uint32_t i = 0; i < kSeg && lo + i < n; ++i) words[i >> 2] |= (uint32_t)text[lo + i] << ((i & 3u) * 8u);
            const uint32_t valid = lo >= round_hi ? 0u : (round_hi - lo >= kSeg ? 0xFFFFFFFFu : (1u << (round_hi - lo)) - 1u);
            const uint32_t starts = ((bytes_equal_to(words, '\n') << 1) | (lo && lo < round_hi && text[lo - 1u] == '\n' ? 1u : 0u)) & valid, at = bytes_equal_to(words, '@');
            kinds[t] = (STEP == 1u ? 0xFFFFFFFFu : line_kinds(line_state, starts, at)) & valid;
            eqs[t] = bytes_equal_to_previous(words);
            line_state = line_apply(line_transfer(starts, at), line_state);
        }
        // phase A1: the searched positions enter the hash table kThreads positions at a time; what the table held for a position's four bytes before its group entered
        // is its candidate, kept as a distance
        for (uint32_t group = round_lo; group < round_hi; group += kThreads) {
            uint32_t cand[kThreads];
            auto searched = [&](uint32_t p) { return p < round_hi && ((kinds[(p - round_lo) / kSeg] >> ((p - round_lo) % kSeg)) & 1u) && p + 4u <= n; };
            auto hash_at = [&](uint32_t p) { return STEP == 1u ? hash4(load4(text + p)) : hash6(load4(text + p), pt.word(p + 4u)); };
            for (uint32_t t = 0; t < kThreads; ++t) cand[t] = searched(group + t) ? head[hash_at(group + t)] : 0u;
            for (uint32_t t = 0; t < kThreads; ++t)
                if (searched(group + t)) {
                    uint32_t &h = head[hash_at(group + t)];
                    h = std::max(h, group + t + 1u);
                }
            for (uint32_t t = 0; t < kThreads; ++t) {
                const uint32_t p = group + t;
                if (p < round_hi && p % STEP == 0u) found[(p - round_lo) / STEP] = (uint16_t)(cand[t] && p + 1u - cand[t] <= kMaxDist ? p + 1u - cand[t] : 0u);
            }
        }
        // phase A2: every kProbeStep-th searched position is probed, the positions behind it inherit
        for (uint32_t p = round_lo; p < round_hi; p += STEP) {
            if (!((kinds[(p - round_lo) / kSeg] >> ((p - round_lo) % kSeg)) & 1u)) continue;
            found[(p - round_lo) / STEP] = (uint16_t)probe(pt, n, round_lo, p, found[(p - round_lo) / STEP]);
        }
        uint32_t round_bits = 0;
        std::vector<SegmentT<STEP>> segs(kThreads);
        std::vector<uint32_t> seg_n(kThreads, 0), seg_bits(kThreads, 0);
        for (uint32_t t = 0; t < kThreads; ++t) {                                            // phase B: the segments into "registers", the counts
            const uint32_t lo = round_lo + t * kSeg;
            seg_n[t] = lo >= round_hi ? 0u : std::min(kSeg, round_hi - lo);
            SegmentT<STEP> &g = segs[t];
            memset(&g, 0, sizeof g);
            g.kind = kinds[t];
            g.eq = eqs[t];
            for (uint32_t i = 0; i < kSeg; ++i)
                if (lo + i < n) g.text[i >> 2] |= (uint32_t)text[lo + i] << ((i & 3u) * 8u);
            for (uint32_t j = 0; j < kSeg / STEP; ++j) g.found[j >> 1] |= (uint32_t)found[(lo - round_lo) / STEP + j] << ((j & 1u) * 16u);
            if (hist) {
                auto add = [hist](uint32_t s) { ++hist[s]; };
                HistogramSink<decltype(add)> sink{add};
                walk_segment(g, seg_n[t], sink);
            } else {
                CountSink<const uint32_t *> count{codes->litlen, codes->dist};
                walk_segment(g, seg_n[t], count);
                seg_bits[t] = count.bits;
                round_bits += count.bits;
            }
        }
        if (hist) continue;
        if ((bit & 31u) + round_bits > kOutWords * 32u) overflow = true;                     // the device's buffer for a round's bits
        if (overflow) continue;
        for (uint32_t t = 0; t < kThreads; ++t) {
            BitSink<const uint32_t *, Or> sink{codes->litlen, codes->dist, Or{data.data() + (bit >> 5)}, (uint32_t)(bit & 31u)};
            walk_segment(segs[t], seg_n[t], sink);
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
inline uint32_t piece_on_the_host(const uint8_t *text, uint32_t n, const Codes *codes, uint8_t *out, uint32_t *hist, bool dense = false) {
    return dense ? piece_on_the_host_t<kDenseStep>(text, n, codes, out, hist) : piece_on_the_host_t<kProbeStep>(text, n, codes, out, hist);
}
// which pieces of a call are the sample: at most 64, spread evenly
inline uint32_t sample_stride(uint64_t n_pieces) { return (uint32_t)std::max<uint64_t>(1, (n_pieces + 63) / 64); }
// Which route a call takes: the sample is walked both ways and each way's symbol counts are priced with the code built from them (code lengths plus the extra bits of
// lengths and distances, RFC 1951 3.2.5); the dense route -- every position of every line searched and probed with a four-byte key, half the speed -- is taken where
// it makes the sample smaller by more than kDenseMustSave.  That is text whose repeats are short and everywhere (a handful of quality values, reads of thirty bases:
// the test profile TINY, 17 % smaller dense); on P0's text the rule for FASTQ lines gives the smaller sample outright (its six-byte key finds the overlapping reads).
constexpr double kDenseMustSave = 0.05;
inline uint64_t sample_bits(const uint32_t *sample, const Codes &codes) {
    uint64_t bits = 0;
    for (uint32_t c = 0; c < 286u; ++c) {
        const uint32_t extra = c < 265u || c == 285u ? 0u : (c - 261u) / 4u;
        bits += (uint64_t)sample[c] * ((codes.litlen[c] & 15u) + extra);
    }
    for (uint32_t d = 0; d < 30u; ++d) bits += (uint64_t)sample[kLitLen + d] * ((codes.dist[d] & 15u) + (d < 4u ? 0u : d / 2u - 1u));
    return bits;
}
inline bool dense_pays(const uint32_t *sample_lines, const uint32_t *sample_dense) {
    const uint64_t lines = sample_bits(sample_lines, build_codes(sample_lines)), dense = sample_bits(sample_dense, build_codes(sample_dense));
    return (double)dense < (1.0 - kDenseMustSave) * (double)lines;
}
// the whole call on the host (tests/hostemu): text -> members, appended to out
inline void gzip_on_the_host(const uint8_t *text, uint64_t n, std::vector<uint8_t> &out, int force_route = -1 /* 0 / 1: FASTQ lines / dense, whatever the sample says */) {
    const uint64_t n_pieces = (n + kPiece - 1) / kPiece;
    const uint32_t stride = sample_stride(n_pieces);
    auto sample = [&](bool dense) {
        std::vector<uint32_t> hist(kLitLen + kDist, 0);
        for (uint64_t i = 0; i < n_pieces; i += stride) piece_on_the_host(text + i * kPiece, (uint32_t)std::min<uint64_t>(kPiece, n - i * kPiece), nullptr, nullptr, hist.data(), dense);
        return hist;
    };
    std::vector<uint32_t> hist = sample(force_route == 1);
    bool dense = force_route == 1;
    if (force_route < 0) {
        const std::vector<uint32_t> other = sample(true);
        dense = dense_pays(hist.data(), other.data());
        if (dense) hist = other;
    }
    Codes codes = build_codes(hist.data());
    codes.dense = dense ? 1u : 0u;
    std::vector<uint8_t> slot(kSlot);
    for (uint64_t i = 0; i < n_pieces; ++i) {
        const uint32_t len = (uint32_t)std::min<uint64_t>(kPiece, n - i * kPiece);
        uint32_t member = piece_on_the_host(text + i * kPiece, len, &codes, slot.data(), nullptr, dense);
        if (!member) member = stored_piece_on_the_host(text + i * kPiece, len, slot.data());
        out.insert(out.end(), slot.begin() + kSlotPad, slot.begin() + kSlotPad + member);
    }
}
#endif

}  // namespace gz
}  // namespace rsq
