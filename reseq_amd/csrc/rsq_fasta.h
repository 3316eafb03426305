// rsq_fasta.h -- seqToIllumina's input parsed on the device (Simulator::ApplyErrorsAndQualityToFastaInput, reseq/Simulator.cpp:2403-2512).
//
// The caller uploads FASTA text as it stands in the file.  A record is ">{id} {1|2};{fragment length};{dominant errors};{error rates}", a line end, and the
// template bases on one line or wrapped over several; it reaches from a '>' at the start of a line to the next one.
//   k_fasta_count / k_fasta_starts   a wave per 4 KB of text finds the record starts (16 bytes per lane, word tests; a scan over the tiles' counts in between
//                                    keeps the input order)
//   k_fasta_records                  a lane per record: the reference's checks of the header in the reference's order (Simulator.cpp:2423-2485), template
//                                    bases -> codes 0..3, dominant errors -> codes 0..4, error rates -> percent (:2439-2442), packed into a half-word per base
//                                    at the record's own offset of an array as long as the text -- what k_fill_records reads (PackedRecordSrc) needs no
//                                    scan and no second copy, a record of any length fits, and the id stays where it is in the text (the formatter reads it there).
//                                    A workgroup's 256 records are one stretch of the text: it is staged into LDS with whole-line loads and the lanes read
//                                    their records there (a lane walking its record in HBM touches a cache line of its own with every load: measured 3.0 ms
//                                    per million records against 1.4 with the stretch in LDS, 0.6 since the bytes are converted eight at a time); a stretch
//                                    that does not fit is read where it is.
// A malformed record sets the smallest index of a bad record; the host fetches that record's text and words the reference's message (record_message).
// parse_record is plain code of one record: the host emulation of the tests runs it as it is.
#pragma once
#include "rsq_types.h"
#include "rsq_core.h"

namespace rsq {
namespace fasta {

enum RecordError : uint32_t {
    kRecordOk = 0,
    kTooShort,              // Simulator.cpp:2424-2427
    kErrorSeparators,       // :2431-2434
    kNoId,                  // :2447-2450
    kSegment,               // :2454-2463
    kSegmentSeparator,      // :2465-2468
    kFragmentLength,        // :2470-2477
    kContainsN              // the reference's tables have no row for N: the CLI of this repository has refused such records since round 1
};
struct RecordFields {
    uint32_t len, id_len, frag_len, seg;
};

constexpr uint64_t kOnes = 0x0101010101010101ull, kHighs = 0x8080808080808080ull;
// bytes [off, off + 8) of a record of `size` bytes, zeros beyond it
// (P: pointer to the text's bytes -- const uint8_t * or, on the device, the same in LDS)
template <class P>
struct WordOf;
template <>
struct WordOf<const uint8_t *> {
    typedef const uint64_t __attribute__((aligned(1))) *type;
};
#if defined(__HIP_DEVICE_COMPILE__)
template <>
struct WordOf<const RSQ_LDS uint8_t *> {
    typedef const RSQ_LDS uint64_t __attribute__((aligned(1))) *type;
};
#endif
template <class P>
RSQ_HD uint64_t word_at(P rec, uint64_t off, uint64_t size) {
    uint64_t w = 0;
    if (off + 8u <= size) {
#if defined(__HIP_DEVICE_COMPILE__)
        w = *reinterpret_cast<typename WordOf<P>::type>(rec + off);          // unaligned 8-byte loads are what the hardware does (amdhsa)
#else
        for (uint32_t j = 0; j < 8u; ++j) w |= (uint64_t)rec[off + j] << (8u * j);
#endif
    } else
        for (uint32_t j = 0; off + j < size; ++j) w |= (uint64_t)rec[off + j] << (8u * j);
    return w;
}
RSQ_HD void store_word(uint8_t *dst, uint64_t w) {
#if defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<uint64_t __attribute__((aligned(1))) *>(dst) = w;
#else
    for (uint32_t j = 0; j < 8u; ++j) dst[j] = (uint8_t)(w >> (8u * j));
#endif
}
// the first position >= from of byte c in the record, or size
template <class P>
RSQ_HD uint64_t find_byte(P rec, uint64_t from, uint64_t size, uint8_t c) {
    for (uint64_t off = from; off < size; off += 8u) {
        const uint64_t w = word_at(rec, off, size) ^ (kOnes * c);        // a zero byte where the text has c (bytes past the end are c itself: not zero for c != 0)
        const uint64_t t = (w - kOnes) & ~w & kHighs;                       // exact in its lowest set bit
        if (t) {
            const uint64_t at = off + ((uint64_t)__builtin_ctzll(t) >> 3);
            return at < size ? at : size;
        }
    }
    return size;
}
RSQ_HD uint32_t base_code(uint32_t c) {          // A C G T in either case -> 0..3, everything else 4
    const uint32_t u = c & 0xDFu;
    return u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : 4u;
}
RSQ_HD uint32_t rate_percent(uint32_t c) {       // Simulator.cpp:2439-2442: percents above 86 are stored halved
    uint32_t v = (c - 33u) & 0xFFu;
    if (v > 86u) v += v - 86u;
    return v & 0xFFu;
}
// Eight bytes at a time (a lane takes 450 bytes of a record apart: byte by byte that was most of the kernel's instructions).
constexpr uint64_t kLow7 = 0x7F7F7F7F7F7F7F7Full;
RSQ_HD uint64_t zero_bytes(uint64_t v) { return ~(((v & kLow7) + kLow7) | v | kLow7); }      // 0x80 in exactly the bytes of v that are zero
RSQ_HD uint64_t sub_bytes(uint64_t a, uint64_t b) { return ((a | kHighs) - (b & kLow7)) ^ ((a ^ ~b) & kHighs); }      // a - b in every byte, modulo 256
RSQ_HD uint64_t base_codes(uint64_t w) {         // base_code of every byte
    const uint64_t x = w & (kOnes * 0xDFu);
    const uint64_t letter = zero_bytes(x ^ (kOnes * 'A')) | zero_bytes(x ^ (kOnes * 'C')) | zero_bytes(x ^ (kOnes * 'G')) | zero_bytes(x ^ (kOnes * 'T'));
    const uint64_t code = ((x >> 1) ^ (x >> 2)) & (kOnes * 3u);      // bits 1-3 of 'A' 'C' 'G' 'T': 000 001 011 010 -> 0 1 2 3
    const uint64_t other = ~letter & kHighs;
    return (code & ~((other >> 7) * 3u)) | (other >> 5);
}
RSQ_HD uint64_t rate_percents(uint64_t w) {      // rate_percent of every byte
    const uint64_t v = sub_bytes(w, kOnes * 33u), k = kOnes * 87u, d = sub_bytes(v, k);
    const uint64_t below = ((~v & k) | (~(v ^ k) & d)) & kHighs;      // the borrow of v - 87 per byte: v < 87
    const uint64_t twice = sub_bytes((v << 1) & (kOnes * 0xFEu), kOnes * 86u);
    const uint64_t above = ((below ^ kHighs) >> 7) * 0xFFu;           // 0xFF in the bytes with v > 86
    return (v & ~above) | (twice & above);
}

// record_start: a '>' that begins a line (or the text)
RSQ_HD bool record_start(const uint8_t *text, uint64_t p) { return text[p] == '>' && (p == 0 || text[p - 1] == '\n'); }

// Where a record's codes go.  ByteArrays: three arrays of bytes (what rsq_sim_error_model takes from its callers, and what the tests read).  Packed: a half-word
// per base -- base code in bits 0-1, dominant error in bits 2-4, error percent in bits 8-15 -- so that the read kernel's lane gets eight bases of all three
// with ONE 16-byte load instead of three 8-byte ones (PackedRecordSrc, rsq_kernels.h: a lane's record is 450 bytes of its own, every load touches 64 lines).
struct ByteArrays {
    uint8_t *seqs, *dom, *rate;
    RSQ_HD void eight(uint64_t k, uint64_t s, uint64_t d, uint64_t r) const {
        store_word(seqs + k, s);
        store_word(dom + k, d);
        store_word(rate + k, r);
    }
    RSQ_HD void one(uint64_t k, uint32_t s, uint32_t d, uint32_t r) const {
        seqs[k] = (uint8_t)s;
        dom[k] = (uint8_t)d;
        rate[k] = (uint8_t)r;
    }
};
RSQ_HD uint64_t spread_bytes(uint32_t x) {       // bytes 0..3 of x into the low bytes of four half-words
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    return (v | (v << 8)) & 0x00FF00FF00FF00FFull;
}
struct Packed {
    uint16_t *codes;
    RSQ_HD void eight(uint64_t k, uint64_t s, uint64_t d, uint64_t r) const {
        const uint64_t low = s | (d << 2);
        store_word(reinterpret_cast<uint8_t *>(codes + k), spread_bytes((uint32_t)low) | (spread_bytes((uint32_t)r) << 8));
        store_word(reinterpret_cast<uint8_t *>(codes + k + 4u), spread_bytes((uint32_t)(low >> 32)) | (spread_bytes((uint32_t)(r >> 32)) << 8));
    }
    RSQ_HD void one(uint64_t k, uint32_t s, uint32_t d, uint32_t r) const { codes[k] = (uint16_t)(s | (d << 2) | (r << 8)); }
};

// is byte j of the word (at offset off of the record) the end of a line: '\n', or '\r' in front of one or of the record's end
template <class P>
RSQ_HD bool line_end_byte(P rec, uint64_t size, uint64_t off, uint64_t w, uint32_t j) {
    const uint32_t c = (uint32_t)(w >> (8u * j)) & 0xFFu;
    if (c == '\n') return true;
    if (c != '\r') return false;
    const uint32_t next = off + j + 1u >= size ? (uint32_t)'\n' : j < 7u ? (uint32_t)(w >> (8u * (j + 1u))) & 0xFFu : (uint32_t)rec[off + 8u];
    return next == '\n';
}

// One record rec[0, size): rec[0] is its '>'.  The codes of base k go to out (k < len; a record of n bytes has fewer than n / 3 bases).
// The checks follow the reference's order; the bases are counted first because every check needs their number, and converted last, together with the systematic
// errors the header holds for them.
template <class P, class Out>
RSQ_HD RecordError parse_record(P rec, uint64_t size, const Out &out, RecordFields &f) {
    const uint64_t line_end = find_byte(rec, 1, size, '\n');
    uint64_t header_len = line_end - 1u;
    if (header_len && rec[line_end - 1u] == '\r') --header_len;
    // the template: every line behind the header, line ends dropped
    uint64_t L = 0;
    for (uint64_t off = line_end + 1u; off < size; off += 8u) {
        const uint64_t w = word_at(rec, off, size);
        if (off + 8u <= size && !(zero_bytes(w ^ (kOnes * '\n')) | zero_bytes(w ^ (kOnes * '\r')))) L += 8u;
        else
            for (uint32_t j = 0; j < 8u && off + j < size; ++j) L += line_end_byte(rec, size, off, w, j) ? 0u : 1u;
    }
    if (header_len <= 2u * L + 2u) return kTooShort;
    const P h = rec + 1;
    uint64_t end = header_len - 2u * L - 3u;
    if (h[end + 1u] != ';' || h[end + 2u + L] != ';') return kErrorSeparators;
    const uint64_t dom_at = end + 2u, rate_at = header_len - L;
    while (end && h[end] != ' ') --end;
    if (!end) return kNoId;
    if (h[end + 1u] == '1') f.seg = 0;
    else if (h[end + 1u] == '2') f.seg = 1;
    else return kSegment;
    if (h[end + 2u] != ';') return kSegmentSeparator;
    const uint64_t fl_at = end + 3u, fl_end = header_len - 2u * L - 2u;
    if (fl_end <= fl_at) return kFragmentLength;
    uint32_t v = 0;                                   // (the reference's stoi ends the run for a number beyond 32 bits; here it wraps as in this repository's CLI since round 1)
    for (uint64_t k = fl_at; k < fl_end; ++k) {
        const uint32_t d = (uint32_t)h[k] - (uint32_t)'0';
        if (d > 9u) return kFragmentLength;
        v = v * 10u + d;
    }
    // (the fields are set HERE, not behind the conversion below: with the assignments at the function's end, hipcc 7.2 / gfx950 stored id_len = 0 in the
    // instantiation that reads the record from HBM -- its code zeroes the register behind the loop and never sets it on the path of a good record; the
    // instantiation on LDS was right.  tests: test_error_model_templates_beyond_the_staging)
    f.len = (uint32_t)L;
    f.id_len = (uint32_t)end;
    f.frag_len = v;
    // bases, dominant errors and rates: eight at a time where eight bases stand in one word of the text, one by one around the line ends
    const uint64_t hs = header_len;
    uint64_t k = 0, any = 0;
    for (uint64_t off = line_end + 1u; off < size; off += 8u) {
        const uint64_t w = word_at(rec, off, size);
        if (off + 8u <= size && !(zero_bytes(w ^ (kOnes * '\n')) | zero_bytes(w ^ (kOnes * '\r')))) {
            const uint64_t codes = base_codes(w);
            any |= codes;
            out.eight(k, codes, base_codes(word_at(h, dom_at + k, hs)), rate_percents(word_at(h, rate_at + k, hs)));
            k += 8u;
            continue;
        }
        for (uint32_t j = 0; j < 8u && off + j < size; ++j) {
            if (line_end_byte(rec, size, off, w, j)) continue;
            const uint32_t code = base_code((uint32_t)(w >> (8u * j)) & 0xFFu);
            any |= code;
            out.one(k, code, base_code(h[dom_at + k]), rate_percent(h[rate_at + k]));
            ++k;
        }
    }
    return any & (kOnes * 4u) ? kContainsN : kRecordOk;
}

#if RSQ_DEVICE_BUILD && !defined(RSQ_SPEC)
constexpr uint32_t kTileBytes = 4096, kStartsBlock = 256;        // a wave per tile, four tiles per workgroup
// record starts in tile `tile`: f(position, rank within the tile) for each, returns their number.  A lane takes 16 bytes of a step's 1024 as two words: the '>'
// bytes whose byte in front is a line end (zero_bytes of the words; the byte in front of the lane's first one is the lane before's last), counted with popcounts
// and ranked by a prefix sum over the lanes.  [A byte per lane and a ballot per 64 bytes read the text at 0.4 TB/s.]
template <class F>
__device__ uint32_t tile_starts(const uint8_t *text, uint64_t text_len, uint64_t tile, F &&f) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t count = 0;
    for (uint32_t step = 0; step < kTileBytes / 1024u; ++step) {
        const uint64_t p = tile * kTileBytes + step * 1024u + lane * 16u;
        uint64_t m0 = 0, m1 = 0;
        if (p < text_len) {
            const uint64_t left = text_len - p;
            const uint64_t w0 = word_at(text + p, 0, left), w1 = left > 8u ? word_at(text + p, 8, left) : 0u;      // (bytes behind the text's end are zeros: no '>')
            const uint64_t front = p == 0 || text[p - 1] == '\n' ? 0x80u : 0u;                     // a line's (or the text's) first byte
            const uint64_t nl0 = zero_bytes(w0 ^ (kOnes * '\n')), nl1 = zero_bytes(w1 ^ (kOnes * '\n'));
            m0 = zero_bytes(w0 ^ (kOnes * '>')) & ((nl0 << 8) | front);
            m1 = zero_bytes(w1 ^ (kOnes * '>')) & ((nl1 << 8) | (nl0 >> 56));
        }
        const uint32_t mine = (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
        uint32_t upto = mine;                                                                     // inclusive prefix sum over the lanes
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint32_t below = (uint32_t)__shfl_up((int)upto, (int)d, 64);
            if (lane >= d) upto += below;
        }
        uint32_t rank = count + upto - mine;
        for (uint64_t m = m0; m; m &= m - 1u) f(p + ((uint64_t)__builtin_ctzll(m) >> 3), rank++);
        for (uint64_t m = m1; m; m &= m - 1u) f(p + 8u + ((uint64_t)__builtin_ctzll(m) >> 3), rank++);
        count += (uint32_t)__shfl((int)upto, 63, 64);
    }
    return count;
}
__global__ void __launch_bounds__(kStartsBlock) k_fasta_count(const uint8_t *text, uint64_t text_len, uint32_t n_tiles, uint32_t *counts) {
    const uint64_t tile = (uint64_t)blockIdx.x * (kStartsBlock / 64u) + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const uint32_t n = tile_starts(text, text_len, tile, [](uint64_t, uint32_t) {});
    if ((threadIdx.x & 63u) == 0) counts[tile] = n;
}
// at[0 .. n): the starts in input order; at[n] = text_len
__global__ void __launch_bounds__(kStartsBlock) k_fasta_starts(const uint8_t *text, uint64_t text_len, uint32_t n_tiles, const uint64_t *first_of_tile, uint32_t *at) {
    const uint64_t tile = (uint64_t)blockIdx.x * (kStartsBlock / 64u) + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const uint64_t first = first_of_tile[tile];
    tile_starts(text, text_len, tile, [&](uint64_t p, uint32_t rank) { at[first + rank] = (uint32_t)p; });
    if (tile == 0 && (threadIdx.x & 63u) == 0) at[first_of_tile[n_tiles]] = (uint32_t)text_len;
}
// summary[0] = the longest template, [1] = the first malformed record (0xFFFFFFFF: none), [2] = text other than line ends in front of the first record
struct Records {
    const uint32_t *at;
    uint32_t *len, *id_len, *frag_len;
    uint8_t *seg;
};
// text[0, lead_end) lies in front of the first record: anything but line ends there is sequence data without a header
__global__ void __launch_bounds__(256) k_fasta_lead(const uint8_t *text, uint32_t lead_end, uint32_t *summary) {
    bool other = false;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < lead_end; p += (uint64_t)gridDim.x * blockDim.x) other = other || (text[p] != '\n' && text[p] != '\r');
    if (other) summary[2] = 1u;
}
// A workgroup takes kRecordsBlock consecutive records.  Their text is one stretch [at[first], at[last]): up to kStageBytes of it go to LDS in 16-byte loads
// that follow each other through the lanes (from the 16-byte boundary below the stretch: device allocations begin and end on one, so the loads stay inside).
constexpr uint32_t kRecordsBlock = 256, kStageBytes = 144u << 10;
template <class P>
__device__ uint32_t record_of_lane(P rec, uint64_t size, uint32_t i, uint32_t at, const Records &r, uint16_t *codes, uint32_t *summary) {
    RecordFields f{0, 0, 0, 0};
    const RecordError e = parse_record(rec, size, Packed{codes + at}, f);
    r.len[i] = f.len;
    r.id_len[i] = f.id_len;
    r.frag_len[i] = f.frag_len;
    r.seg[i] = (uint8_t)f.seg;
    if (e != kRecordOk) atomicMin(&summary[1], i);
    return e == kRecordOk ? f.len : 0u;
}
__global__ void __launch_bounds__(kRecordsBlock) k_fasta_records(const uint8_t *text, uint32_t n, Records r, uint16_t *codes, uint32_t *summary, uint32_t stage_limit) {
    extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
    const uint32_t first = blockIdx.x * kRecordsBlock, last = min(first + kRecordsBlock, n), i = first + threadIdx.x;
    const uint32_t begin = r.at[first], end = r.at[last];
    const uint32_t skew = (uint32_t)(reinterpret_cast<uintptr_t>(text + begin) & 15u), span = skew + (end - begin);
    const bool staged = span <= stage_limit;                          // kStageBytes; 0 (option fasta_no_stage): every workgroup reads its records where they lie in HBM
    if (staged) {
        const uint4 *src = reinterpret_cast<const uint4 *>(text + begin - skew);
        uint4 *dst = reinterpret_cast<uint4 *>(stage);
        const uint32_t words = (span + 15u) >> 4;
#pragma unroll 8
        for (uint32_t w = threadIdx.x; w < words; w += kRecordsBlock) dst[w] = src[w];
        __syncthreads();
    }
    uint32_t longest = 0;
    if (i < n) {
        const uint32_t at = r.at[i];
        const uint64_t size = (uint64_t)r.at[i + 1] - at;
        longest = staged ? record_of_lane((const RSQ_LDS uint8_t *)stage + skew + (at - begin), size, i, at, r, codes, summary)
                         : record_of_lane(text + at, size, i, at, r, codes, summary);
    }
    for (uint32_t d = 32; d; d >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, (int)d, 64));      // one atomic per wave, not per record
    if ((threadIdx.x & 63u) == 0 && longest) atomicMax(&summary[0], longest);
}
#endif

}  // namespace fasta
}  // namespace rsq
