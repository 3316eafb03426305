// rsq_kernels.h -- HIP kernels of the simulation path (gfx950).  One lane per unit of work throughout:
//   k_sys_chain     one lane per chunk of a systematic-error chain           (Simulator.h:337-382, a13)
//   k_sum_bias      one lane per run of fragment start positions              (Reference.cpp:622-659, a14)
//   k_sieve         one wave per reference start position, lanes over lengths (Simulator.cpp:2249-2357, a6)
//   k_fill_reads    one lane per read                                         (Simulator.cpp:454-594, a2/a3)
//   k_format        one lane per FASTQ record                                 (Simulator.cpp:596-632, a5)
// All arithmetic lives in rsq_core.h; this file only maps work to lanes and moves bytes.
#pragma once
#include "rsq_core.h"

namespace rsq {

// ------------------------------------------------------------------------------------ systematic errors
struct Chain {
    uint32_t kind;         // 0 reference forward, 1 reference reverse complement, 2 adapter
    uint32_t id;           // sequence id or adapter id
    uint32_t seg;          // adapter: template segment
    uint32_t len;
    uint32_t c1, c2;       // Philox counter words identifying the chain
    uint32_t first_chunk;
    uint32_t initial_dom;  // DominantBase::dom_base_ left behind by the previous chain (Clear() keeps it)
    uint16_t *out;
};

struct ChainAcc {
    const uint64_t *words;
    uint32_t kind, len;
    uint64_t word_off;
    const uint8_t *codes;
    RSQ_HD uint32_t operator()(uint32_t pos) const {
        if (kind == 0) return ref_base(words, word_off, pos);
        if (kind == 1) return 3u - ref_base(words, word_off, len - 1u - pos);
        return codes[pos];
    }
};

template <class Acc>
RSQ_HD uint32_t find_dominant(const Acc &acc, const uint32_t (&cnt)[4], uint32_t cur_pos) {     // utilities.hpp:238-262 (N-free sequence)
    uint32_t mx = cnt[0];
    for (int i = 1; i < 4; ++i) mx = cnt[i] > mx ? cnt[i] : mx;
    uint32_t pos = cur_pos;
    uint32_t b;
    do { b = acc(--pos); } while (cnt[b] != mx);
    return b;
}

// CoverageStats.cpp:379-396
RSQ_HD void update_distances(uint32_t reset_distance, uint32_t &dist, uint32_t &start_rate, uint32_t error_rate) {
    if (dist) {
        if (start_rate < error_rate) {
            dist = 0;
            start_rate = error_rate;
        } else if (++dist >= reset_distance) {
            dist = 0;
            start_rate = 0;
        }
    } else if (error_rate) {
        dist = 1;
        start_rate = error_rate;
    }
}

// Positions [lo,hi) of one chain.  Everything except (dist,start_rate) is a pure function of the sequence and is
// rebuilt at `lo`, so a chunk can start anywhere given the incoming (dist,start_rate).
template <class Acc>
RSQ_HD void sys_chain_chunk(const DevSim &S, const Acc &acc, uint32_t c1, uint32_t c2, uint32_t lo, uint32_t hi, uint32_t initial_dom, uint32_t &dist,
                            uint32_t &start_rate, uint16_t *out) {
    uint32_t cnt[4] = {0, 0, 0, 0};
    for (uint32_t p = lo > 5 ? lo - 5 : 0; p < lo; ++p) ++cnt[acc(p)];
    uint32_t last_base = lo ? acc(lo - 1) : 4u;
    uint32_t dom = lo ? find_dominant(acc, cnt, lo) : initial_dom;
    const uint32_t range = S.sys_gc_range;
    uint32_t gc_bases = lo < range ? lo : range, gc = 0;
    for (uint32_t p = lo - gc_bases; p < lo; ++p) gc += is_gc(acc(p));
    for (uint32_t pos = lo; pos < hi; ++pos) {
        const uint32_t b = acc(pos);
        const Words w = philox(S.seed, pos, c1, c2, kDomSysErr << 28);
        const uint32_t idx[3] = {transform_distance(dist), safe_percent_u16(gc, gc_bases), start_rate};
        double ps;
        uint32_t dom_error = draw<3>(S.dom_error[(b * 5u + last_base) * 5u + dom], S.pool, S.par0, idx, u32_to_unit(w.w0), ps);
        if (0.0 == ps) dom_error = 4;
        uint32_t rate = draw<3>(S.error_rate[b * 5u + dom_error], S.pool, S.par0, idx, u32_to_unit(w.w1), ps);
        if (0.0 == ps) rate = 0;
        out[pos] = (uint16_t)(dom_error | (rate << 8));
        last_base = b;
        ++cnt[b];
        if (pos >= 5) --cnt[acc(pos - 5)];
        dom = find_dominant(acc, cnt, pos + 1);
        update_distances(S.reset_distance, dist, start_rate, rate);
        if (is_gc(b)) ++gc;                                         // Simulator.h:354-366 UpdateGC
        if (gc_bases < range) ++gc_bases;
        else if (is_gc(acc(pos - gc_bases))) --gc;
    }
}

struct BiasParam {
    uint32_t seq, len;
    double general_bias;       // ref_seq_bias * insert_lengths_bias[len]
};
constexpr uint32_t kBiasRun = 32;          // start positions per lane
constexpr uint32_t kBiasBlock = 256;

// Reference::Bias of the fragment [start, start+len) (Reference.cpp:634-637,650-653 inside SumBias)
RSQ_HD double site_bias(const DevSim &S, uint64_t word_off, uint32_t L, uint32_t start, uint32_t len, uint32_t gc_count, double general_bias) {
    uint32_t ss[3], se[3];
    surrounding_forward(S.ref_words, word_off, L, start, ss);
    surrounding_reverse(S.ref_words, word_off, L, start + len - 1, se);
    return general_bias * S.gc_bias[percent_u32(gc_count, len)] * surrounding_bias(S.sur_bias, ss) * surrounding_bias(S.sur_bias, se);
}

#if defined(__HIPCC__)

// Speculative chunking: pass 0 runs every chunk from (dist,start_rate) = (0,0); later passes re-run exactly the
// chunks whose true incoming state (the outgoing state of their left neighbour) differs from the one they used.
// The fixed point is the sequential chain, bit for bit, for any seed.
__global__ void __launch_bounds__(64) k_sys_chain(DevSim S, const Chain *chains, const uint32_t *chunk_chain, uint32_t n_chunks, uint32_t chunk_len, uint32_t *used_state,
                            const uint32_t *out_prev, uint32_t *out_new, uint32_t *changed, int pass) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const Chain ch = chains[chunk_chain[c]];
    const uint32_t local = c - ch.first_chunk;
    uint32_t want = 0;
    if (pass > 0) {
        if (local == 0) {
            out_new[c] = out_prev[c];
            return;
        }
        want = out_prev[c - 1];
        if (want == used_state[c]) {
            out_new[c] = out_prev[c];
            return;
        }
        *changed = 1;
    }
    used_state[c] = want;
    ChainAcc acc{S.ref_words, ch.kind, ch.len, ch.kind < 2 ? S.seq_word_off[ch.id] : 0, ch.kind == 2 ? S.adapters[ch.seg].seqs + S.adapters[ch.seg].seq_ptr[ch.id] : nullptr};
    uint32_t dist = want & 0xFFFFFFu, start_rate = want >> 24;
    const uint32_t lo = local * chunk_len, hi = lo + chunk_len < ch.len ? lo + chunk_len : ch.len;
    sys_chain_chunk(S, acc, ch.c1, ch.c2, lo, hi, ch.initial_dom, dist, start_rate, ch.out);
    out_new[c] = dist | (start_rate << 24);
}

// ------------------------------------------------------------------------------------ bias normalisation

__global__ void __launch_bounds__(256) k_sum_bias(DevSim S, const BiasParam *params, double *partial_sum, double *partial_max) {
    __shared__ double s_sum[kBiasBlock];
    __shared__ double s_max[kBiasBlock];
    const BiasParam p = params[blockIdx.y];
    const uint32_t L = S.seq_len[p.seq];
    const uint64_t wo = S.seq_word_off[p.seq];
    const uint32_t n_starts = L - p.len + 1;                       // start positions 0 .. L-len (Reference.cpp:645)
    const uint32_t first = (blockIdx.x * kBiasBlock + threadIdx.x) * kBiasRun;
    double sum = 0.0, mx = 0.0;
    if (first < n_starts) {
        const uint32_t last = first + kBiasRun < n_starts ? first + kBiasRun : n_starts;
        uint32_t gc = ref_gc_count(S.ref_words, wo, first, first + p.len);
        for (uint32_t start = first; start < last; ++start) {
            const double bias = site_bias(S, wo, L, start, p.len, gc, p.general_bias);
            sum += bias;
            mx = bias > mx ? bias : mx;
            if (start + 1 < last) gc = gc + is_gc(ref_base(S.ref_words, wo, start + p.len)) - is_gc(ref_base(S.ref_words, wo, start));
        }
    }
    s_sum[threadIdx.x] = sum;
    s_max[threadIdx.x] = mx;
    __syncthreads();
    for (uint32_t s = kBiasBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + s];
            s_max[threadIdx.x] = s_max[threadIdx.x + s] > s_max[threadIdx.x] ? s_max[threadIdx.x + s] : s_max[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial_sum[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s_sum[0];
        partial_max[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s_max[0];
    }
}

#endif  // __HIPCC__

// --------------------------------------------------------------------------------------------- the sieve
// One cell = one (start, fragment length) pair of SimulateFromGivenBlock's double loop (Simulator.cpp:2290-2350), one
// allele, no variants.  Returns the number of pairs; cnt[j] / strand_of[j] describe the chosen strands in draw order.
struct SieveSite {
    uint32_t seq, start, L;
    uint64_t word_off;
    const double *thr;                 // thresholds of the sequence's coverage group: [insert_to][2]
    uint32_t sur_start[3];
    bool have_start;
};

RSQ_HD uint32_t sieve_cell(const DevSim &S, SieveSite &site, uint32_t len, uint32_t (&cnt)[2], uint32_t (&strand_of)[2]) {
    cnt[0] = cnt[1] = 0;
    strand_of[0] = strand_of[1] = 0;
    const Words w = philox(S.seed, site.start, site.seq, len, kDomSieve << 28);
    const double probability_chosen = u53_to_unit(w.w0, w.w1);
    const double thr0 = site.thr[2u * len], thr1 = site.thr[2u * len + 1u];
    if (!(probability_chosen >= thr1)) return 0;                                    // Simulator.h:418-420
    const uint32_t non_zero_strands = binomial(2u, 1 - thr0, probability_chosen);   // Simulator.cpp:2307
    const uint32_t end = site.start + len;
    if (!non_zero_strands || !(end < site.L)) return 0;                             // :2308,:2318
    uint32_t n_chosen;
    if (non_zero_strands <= 1u) {                                                   // :1387-1391 DrawNAlleles(1) -> SelectAllele
        strand_of[0] = (uint32_t)(u32_to_unit(w.w2) * 2.0) & 1u;
        n_chosen = 1;
    } else {                                                                        // :1392-1396 complement of the empty draw
        strand_of[0] = 0;
        strand_of[1] = 1;
        n_chosen = 2;
    }
    if (!site.have_start) {
        surrounding_forward(S.ref_words, site.word_off, site.L, site.start, site.sur_start);
        site.have_start = true;
    }
    uint32_t sur_end[3];
    surrounding_reverse(S.ref_words, site.word_off, site.L, end - 1u, sur_end);     // :1820-1832
    const uint32_t gc = percent_u32(ref_gc_count(S.ref_words, site.word_off, site.start, end), len);   // :1858-1873
    const Words w2 = philox(S.seed, site.start, site.seq, len, (kDomSieve << 28) | 1u);
    uint32_t n_here = 0;
    for (uint32_t j = 0; j < n_chosen; ++j) {
        const double u = j ? u53_to_unit(w2.w2, w2.w3) : u53_to_unit(w2.w0, w2.w1);
        const double adjusted_random = thr0 + u * (1 - thr0);                       // :2322
        cnt[j] = fragment_counts(S, site.seq, len, gc, site.sur_start, sur_end, adjusted_random);
        n_here += cnt[j];
    }
    return n_here;
}

RSQ_HD Fragment make_fragment(const SieveSite &site, uint32_t len, uint32_t dup, uint32_t strand, uint32_t block_id, uint32_t number) {
    Fragment f;
    f.seq = site.seq;
    f.start = site.start;
    f.len = len;
    f.dup = (uint16_t)dup;
    f.strand = (uint8_t)strand;
    f.pad = 0;
    f.block = block_id;
    f.number = number;
    return f;
}

#if defined(__HIPCC__)
// One wave per start position; lane l tests fragment lengths insert_from + l, + 64, ...  A cell that passes the
// zero threshold (rare) is finished by its own lane.  COUNT pass: counts[slot] = pairs starting at this position.
// EMIT pass: the same walk again, writing Fragment records at offsets[slot] in (length, chosen strand order,
// duplicate) order -- the order of the reference's loops.
template <bool EMIT>
__global__ void __launch_bounds__(256) k_sieve(DevSim S, uint32_t block_lo, uint32_t n_slots, uint32_t *counts, const uint64_t *offsets, Fragment *frags) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (slot >= n_slots) return;
    const uint32_t block_id = block_lo + slot / kBlockSize;
    SieveSite site;
    site.seq = S.block_seq[block_id];
    site.L = S.seq_len[site.seq];
    site.start = (block_id - S.first_block[site.seq]) * kBlockSize + slot % kBlockSize;
    if (site.start >= site.L) {
        if (!EMIT && lane == 0) counts[slot] = 0;
        return;
    }
    site.word_off = S.seq_word_off[site.seq];
    site.thr = S.thresholds + (size_t)S.coverage_group[site.seq] * S.insert_to * 2u;
    site.have_start = false;
    uint32_t total = 0;                                            // COUNT: pairs of this lane; EMIT: pairs of all earlier iterations
    uint64_t out_base = 0;
    uint32_t number_base = 0;
    if (EMIT) {
        out_base = offsets[slot];
        number_base = (uint32_t)(out_base - offsets[slot - slot % kBlockSize]);
    }
    for (uint32_t len0 = S.insert_from; len0 < S.insert_to; len0 += 64u) {
        const uint32_t len = len0 + lane;
        uint32_t n_here = 0, cnt[2] = {0, 0}, strand_of[2] = {0, 0};
        if (len < S.insert_to) n_here = sieve_cell(S, site, len, cnt, strand_of);
        if (!EMIT) {
            total += n_here;
        } else {
            uint32_t incl = n_here;                                 // inclusive prefix over the wave: lanes hold ascending lengths
            for (uint32_t d = 1; d < 64u; d <<= 1) {
                const uint32_t v = __shfl_up(incl, d, 64);
                if (lane >= d) incl += v;
            }
            const uint32_t wave_total = __shfl(incl, 63, 64);
            uint32_t k = total + incl - n_here;
            for (uint32_t j = 0; j < 2u; ++j)
                for (uint32_t dup = 0; dup < cnt[j]; ++dup, ++k) frags[out_base + k] = make_fragment(site, len, dup, strand_of[j], block_id, number_base + k + 1u);
            total += wave_total;
        }
    }
    if (!EMIT) {
        for (uint32_t d = 32; d > 0; d >>= 1) total += __shfl_down(total, d, 64);
        if (lane == 0) counts[slot] = total;
    }
}

// ------------------------------------------------------------------------------------------------ scans
// exclusive prefix sum of uint32 counts into uint64 offsets (n+1 entries: offsets[n] = total); three launches.
constexpr uint32_t kScanBlock = 256;
constexpr uint32_t kScanPer = 8;              // elements per thread
constexpr uint32_t kScanTile = kScanBlock * kScanPer;

__global__ void k_scan_tile_sums(const uint32_t *in, uint64_t n, uint64_t *tile_sums) {
    __shared__ uint64_t s[kScanBlock];
    uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanPer, acc = 0;
    for (uint32_t i = 0; i < kScanPer; ++i)
        if (base + i < n) acc += in[base + i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = kScanBlock / 2; d > 0; d >>= 1) {
        if (threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = s[0];
}
__global__ void k_scan_tiles(uint64_t *tile_sums, uint32_t n_tiles, uint64_t *total) {      // one block, serial over tiles (few thousand)
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint64_t acc = 0;
        for (uint32_t i = 0; i < n_tiles; ++i) {
            uint64_t v = tile_sums[i];
            tile_sums[i] = acc;
            acc += v;
        }
        *total = acc;
    }
}
__global__ void k_scan_apply(const uint32_t *in, uint64_t n, const uint64_t *tile_sums, const uint64_t *total, uint64_t *out) {
    __shared__ uint64_t s[kScanBlock];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanPer;
    uint32_t v[kScanPer];
    uint64_t acc = 0;
    for (uint32_t i = 0; i < kScanPer; ++i) {
        v[i] = base + i < n ? in[base + i] : 0u;
        acc += v[i];
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = 1; d < kScanBlock; d <<= 1) {                 // Hillis-Steele inclusive scan of the per-thread sums
        uint64_t t = threadIdx.x >= d ? s[threadIdx.x - d] : 0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    uint64_t run = tile_sums[blockIdx.x] + s[threadIdx.x] - acc;
    for (uint32_t i = 0; i < kScanPer; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;
}

#endif  // __HIPCC__

// ------------------------------------------------------------------------------------------------- reads
struct ReadOut {                        // destination of one lane's read
    uint8_t *seq, *qual;
    uint32_t *ops;
    uint32_t cur_word, cur_index;
    RSQ_HD void put(uint32_t pos, uint32_t base, uint32_t qual_char) {
        seq[pos] = (uint8_t)base;
        qual[pos] = (uint8_t)qual_char;
    }
    RSQ_HD void op(uint32_t it, uint32_t code) {                    // 2 bits per iteration, 16 per word
        const uint32_t wi = it >> 4;
        if (wi != cur_index) {
            ops[cur_index] = cur_word;
            cur_word = 0;
            cur_index = wi;
        }
        cur_word |= code << ((it & 15u) * 2u);
    }
    RSQ_HD void finish() { ops[cur_index] = cur_word; }
};

struct RawLayout {                      // per-read arrays of the read kernel, read index = segment * n_pairs + pair
    uint8_t *seq, *qual;
    uint32_t *ops;
    ReadMeta *meta;
    uint32_t read_stride;               // bytes per read in seq / qual
    uint32_t ops_stride;                // words per read in ops
};

struct FragmentSrc {                    // template of one mate cut from the 2-bit reference (Reference.cpp:483-496)
    const uint64_t *words;
    uint64_t word_off;
    uint32_t first;                     // forward: start position; reverse: end position
    uint32_t len;
    bool reverse;
    const uint16_t *sys_;               // systematic errors at the first template base
    RSQ_HD uint32_t org_len() const { return len; }
    RSQ_HD uint32_t base(uint32_t k) const {
        return reverse ? 3u - ref_base(words, word_off, first - 1u - k) : ref_base(words, word_off, first + k);
    }
    RSQ_HD uint32_t sys(uint32_t k) const { return sys_[k]; }
};

struct EmptySrc {                       // adapter-only pair: org_seq_ = "" (Simulator.cpp:2369-2371)
    RSQ_HD uint32_t org_len() const { return 0; }
    RSQ_HD uint32_t base(uint32_t) const { return 0; }
    RSQ_HD uint32_t sys(uint32_t) const { return 0; }
};

RSQ_HD uint32_t draw_tile(const DevSim &S, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3base) {      // Simulator.h:176-181
    if (S.n_tiles > 1) return discrete_draw(S.tile_cp, S.n_tiles, u32_to_unit(philox(S.seed, c0, c1, c2, c3base).w0));
    return 0;
}

// CreateReads for one mate of a fragment (Simulator.cpp:634-721, GetOrgSeq :1916-1922)
RSQ_HD void fill_fragment_read(const DevSim &S, const Fragment &f, uint32_t seg, ReadOut &out, ReadMeta &meta) {
    const uint32_t c2 = f.len | ((uint32_t)f.dup << 16);
    const uint32_t tile = draw_tile(S, f.start, f.seq, c2, pair_c3(kDomPair, f.strand, 2));
    const Stream st{S.seed, f.start, f.seq, c2, pair_c3(kDomPair, f.strand, seg)};
    const uint32_t L = S.seq_len[f.seq], end = f.start + f.len;
    const uint32_t want = S.read_lengths[seg].to + S.max_len_deletion;           // Simulator.cpp:1918-1921
    FragmentSrc src;
    src.words = S.ref_words;
    src.word_off = S.seq_word_off[f.seq];
    src.len = f.len < want ? f.len : want;
    src.reverse = seg != f.strand;                                              // block.at(strand) = start_block (:680-684)
    src.first = src.reverse ? end : f.start;
    src.sys_ = src.reverse ? S.sys_rev + S.seq_base_off[f.seq] + (L - end) : S.sys_fwd + S.seq_base_off[f.seq] + f.start;
    fill_read(S, st, seg, tile, f.len, src, out, meta);
}
// one mate of adapter-only pair i (Simulator.cpp:2359-2382)
RSQ_HD void fill_adapter_only_read(const DevSim &S, uint64_t i, uint32_t seg, ReadOut &out, ReadMeta &meta) {
    const uint32_t tile = draw_tile(S, (uint32_t)i, 0xFFFFFFFFu, (uint32_t)(i >> 32), pair_c3(kDomPair, 0, 2));
    const Stream st{S.seed, (uint32_t)i, 0xFFFFFFFFu, (uint32_t)(i >> 32), pair_c3(kDomPair, 0, seg)};
    fill_read(S, st, seg, tile, 0u, EmptySrc{}, out, meta);
}

// seqToIllumina records (Simulator.cpp:2403-2512): templates and systematic errors come from arrays
struct RecordSrc {
    const uint8_t *seq;
    const uint8_t *dom, *rate;
    uint32_t len;
    RSQ_HD uint32_t org_len() const { return len; }
    RSQ_HD uint32_t base(uint32_t k) const { return seq[k]; }
    RSQ_HD uint32_t sys(uint32_t k) const { return (uint32_t)dom[k] | ((uint32_t)rate[k] << 8); }
};
RSQ_HD void fill_record_read(const DevSim &S, uint64_t idx, uint32_t seg, uint32_t frag_len, const RecordSrc &src, ReadOut &out, ReadMeta &meta) {
    const uint32_t tile = draw_tile(S, (uint32_t)idx, (uint32_t)(idx >> 32), 0u, pair_c3(kDomErrModel, 0, 2));
    const Stream st{S.seed, (uint32_t)idx, (uint32_t)(idx >> 32), 0u, pair_c3(kDomErrModel, 0, seg)};
    fill_read(S, st, seg, tile, frag_len, src, out, meta);
}

#if defined(__HIPCC__)
// grid.y = template segment
__global__ void __launch_bounds__(64) k_fill_reads(DevSim S, const Fragment *frags, uint64_t n_pairs, uint64_t adapter_only_first, RawLayout raw) {
    const uint64_t pair = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= n_pairs) return;
    const uint32_t seg = blockIdx.y;
    const uint64_t r = (uint64_t)seg * n_pairs + pair;
    ReadOut out{raw.seq + r * raw.read_stride, raw.qual + r * raw.read_stride, raw.ops + r * raw.ops_stride, 0u, 0u};
    ReadMeta meta;
    if (frags) fill_fragment_read(S, frags[pair], seg, out, meta);
    else fill_adapter_only_read(S, adapter_only_first + pair, seg, out, meta);
    out.finish();
    raw.meta[r] = meta;
}

__global__ void __launch_bounds__(64) k_error_model(DevSim S, uint64_t first_index, uint64_t n, uint32_t read_len, const uint8_t *seqs, const uint8_t *segs,
                                                   const uint32_t *frag_len, const uint8_t *dom, const uint8_t *rate, RawLayout raw) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    RecordSrc src{seqs + i * read_len, dom + i * read_len, rate + i * read_len, read_len};
    ReadOut out{raw.seq + i * raw.read_stride, raw.qual + i * raw.read_stride, raw.ops + i * raw.ops_stride, 0u, 0u};
    ReadMeta meta;
    fill_record_read(S, first_index + i, segs[i], frag_len[i], src, out, meta);
    out.finish();
    raw.meta[i] = meta;
}

#endif  // __HIPCC__

// ---------------------------------------------------------------------------------------------- FASTQ text
// Replays the CIGAR bookkeeping of FillReadPart over the stored 2-bit ops (see fill_read_part in rsq_core.h).
template <class Sink>
RSQ_HD void cigar_replay(const uint32_t *ops, const ReadMeta &m, Sink &sink) {
    uint32_t it = 0;
    for (int part = 0; part < 2; ++part) {
        const char base = part ? 'S' : 'M';
        const uint32_t n = part ? m.n_iter_s : m.n_iter_m;
        char element = base;
        uint32_t length = 0;
        for (uint32_t i = 0; i < n; ++i, ++it) {
            const uint32_t code = (ops[it >> 4] >> ((it & 15u) * 2u)) & 3u;
            const char want = code == 0 ? base : (code == 1 ? 'D' : 'I');
            if (want == element) ++length;
            else {
                sink.element(element, length);
                element = want;
                length = 1;
            }
        }
        if (length) sink.element(element, length);
    }
    if (m.hard_clip) sink.element('H', m.hard_clip);
}

struct TextSink {                        // writes when p != nullptr, always counts
    char *p;
    uint32_t n;
    RSQ_HD void ch(char c) {
        if (p) p[n] = c;
        ++n;
    }
    RSQ_HD void str(const char *s, uint32_t len) {
        for (uint32_t i = 0; i < len; ++i) ch(s[i]);
    }
    RSQ_HD void num(uint64_t v) {
        char tmp[20];
        int k = 0;
        do {
            tmp[k++] = (char)('0' + v % 10);
            v /= 10;
        } while (v);
        while (k) ch(tmp[--k]);
    }
    RSQ_HD void element(char op, uint32_t count) {
        num(count);
        ch(op);
    }
};

struct NameTable {                       // first parts of the reference ids + the record base identifier
    const char *names;
    const uint32_t *name_ptr;
    char base_identifier[64];
    uint32_t base_len;
};

// One FASTQ record "@id\nSEQ\n+\nQUAL\n" with the id of Simulator.cpp:596-632; dst == nullptr only measures.
RSQ_HD uint32_t format_record(const DevSim &S, const NameTable &names, const Fragment *f, uint64_t adapter_only_number, const ReadMeta &m, const uint8_t *seq,
                              const uint8_t *qual, const uint32_t *ops, char *dst) {
    TextSink t{dst, 0};
    t.ch('@');
    t.str(names.base_identifier, names.base_len);
    if (f) {
        const uint32_t end = f->start + f->len;
        t.num(f->block);
        t.ch('_');
        t.num(f->number);
        t.ch(':');
        t.num(f->strand ? end : f->start + 1u);
        t.ch(':');
        t.str(names.names + names.name_ptr[f->seq], names.name_ptr[f->seq + 1] - names.name_ptr[f->seq]);
        t.ch(':');
        t.num(f->strand ? f->start + 1u : end);
    } else {
        t.ch('0');
        t.ch('_');
        t.num(adapter_only_number);
        t.str(":0:Adapter:0", 12);
    }
    t.ch(':');
    t.num(S.tiles[m.tile_id]);
    t.str(":1337:1337 ", 11);
    cigar_replay(ops, m, t);
    t.str(" E", 2);
    t.num(m.num_errors);
    t.ch('\n');
    const char *kBases = "ACGTN";
    for (uint32_t i = 0; i < m.read_len; ++i) t.ch(kBases[seq[i]]);
    t.str("\n+\n", 3);
    for (uint32_t i = 0; i < m.read_len; ++i) t.ch((char)qual[i]);
    t.ch('\n');
    return t.n;
}

#if defined(__HIPCC__)
// sizes pass (dst == nullptr) and write pass; grid.y = template segment, which is also the output file
__global__ void __launch_bounds__(64) k_format(DevSim S, NameTable names, const Fragment *frags, uint64_t n_pairs, uint64_t adapter_only_first, RawLayout raw, uint32_t *sizes,
                         const uint64_t *offsets0, const uint64_t *offsets1, char *dst0, char *dst1) {
    const uint64_t pair = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= n_pairs) return;
    const uint32_t seg = blockIdx.y;
    const uint64_t r = (uint64_t)seg * n_pairs + pair;
    const ReadMeta m = raw.meta[r];
    Fragment f;
    if (frags) f = frags[pair];
    char *dst = nullptr;
    if (!sizes) dst = (seg ? dst1 + offsets1[pair] : dst0 + offsets0[pair]);
    const uint32_t n = format_record(S, names, frags ? &f : nullptr, adapter_only_first + pair + 1u, m, raw.seq + r * raw.read_stride, raw.qual + r * raw.read_stride,
                                     raw.ops + r * raw.ops_stride, dst);
    if (sizes) sizes[r] = n;
}
#endif

}  // namespace rsq
