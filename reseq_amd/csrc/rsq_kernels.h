// rsq_kernels.h -- HIP kernels of the simulation path (gfx950).  One lane per unit of work throughout:
//   k_sys_chain     one lane per chunk of a systematic-error chain           (Simulator.h:337-382, a13)
//   k_sum_bias      one lane per run of fragment start positions              (Reference.cpp:622-659, a14)
//   k_sieve         one wave per reference start position, lanes over lengths (Simulator.cpp:2249-2357, a6)
//   k_fill_reads    one lane per read                                         (Simulator.cpp:454-594, a2/a3)
//   k_format        one lane per FASTQ record                                 (Simulator.cpp:596-632, a5)
// All arithmetic lives in rsq_core.h; this file only maps work to lanes and moves bytes.
#pragma once
#include "rsq_core.h"
#include "rsq_variants.h"

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
    // a rank of a sharded job runs only the chunks its reads can touch: chunks [chunk_lo, chunk_lo + its number of chunks) of the chain,
    // entered with in_state (dist | start_rate << 24), the outgoing state of chunk chunk_lo - 1 on the neighbouring rank (0 at a chain's start)
    uint32_t chunk_lo = 0, in_state = 0;
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

// One of the chain's two draws, screened (rsq_core.h draw_screened: single precision with a proof that the column is the double-precision
// one); undecided draws and tables outside the screen's preconditions are repeated in double precision.  Q quads per row of the table's
// float copy.  Returns the outcome value; `none`: what an all-zero row gives (prob_sum == 0 in the reference's recipe).
template <int Q>
RSQ_HD uint32_t chain_draw(const DevSim &S, const DevTable &t, const uint32_t (&idx)[3], uint32_t word, uint32_t none) {
    if (S.chain_quads && t.k && t.f32_ok) {
        const float *g = S.pool32 + t.off32;
        const uint32_t slot = 4u * (uint32_t)Q;
        const GlobalRow32 m0{g + clamp_row(t, 0, idx[0]) * slot}, m1{g + (t.rows[0] + clamp_row(t, 1, idx[1])) * slot},
            m2{g + (t.rows[0] + t.rows[1] + clamp_row(t, 2, idx[2])) * slot};
        uint32_t col = 0;
        if (draw_screened<Q>(word, col, m0, m1, m2)) return S.par0[t.par0_off + col];
    }
    double ps;
    const uint32_t value = draw<3>(t, S.pool, S.par0, idx, u32_to_unit(word), ps);
    return 0.0 == ps ? none : value;
}
RSQ_HD uint32_t chain_draw_rate_rows(const DevSim &S, const DevTable &t, const uint32_t (&idx)[3], uint32_t word) {
    switch (S.chain_quads) {
        case 8: return chain_draw<8>(S, t, idx, word, 0u);
        case 16: return chain_draw<16>(S, t, idx, word, 0u);
        default: return chain_draw<26>(S, t, idx, word, 0u);           // also 0: chain_draw goes straight to double precision
    }
}
// the word alone says "rate 0" for the lane's rows of margins 0 and 2 (rsq_pack.h): no row is read
RSQ_HD bool chain_rate_is_zero(const DevSim &S, const DevTable &t, const uint32_t (&idx)[3], uint32_t word) {
    if (!t.sure_range) return false;
    const uint32_t range = S.chain_sure[t.sure_range + clamp_row(t, 0, idx[0]) * t.rows[2] + clamp_row(t, 2, idx[2])], lo16 = range & 0xFFFFu;
    return (word >> 16) - lo16 < (range >> 16) - lo16;
}
RSQ_HD uint32_t chain_draw_rate(const DevSim &S, const DevTable &t, const uint32_t (&idx)[3], uint32_t word) {
    if (chain_rate_is_zero(S, t, idx, word)) return 0u;
    return chain_draw_rate_rows(S, t, idx, word);
}

// Positions [lo,hi) of one chain.  Everything except (dist,start_rate) is a pure function of the sequence and is
// rebuilt at `lo`, so a chunk can start anywhere given the incoming (dist,start_rate).  keep_from > lo: the positions in front of keep_from are a run-up
// (nothing is written for them) and *kept_state receives the state in front of keep_from.
template <class Acc>
RSQ_HD void sys_chain_chunk(const DevSim &S, const Acc &acc, uint32_t c1, uint32_t c2, uint32_t lo, uint32_t hi, uint32_t initial_dom, uint32_t &dist,
                            uint32_t &start_rate, uint16_t *out, uint32_t keep_from = 0, uint32_t *kept_state = nullptr) {
    uint32_t cnt[4] = {0, 0, 0, 0};
    for (uint32_t p = lo > 5 ? lo - 5 : 0; p < lo; ++p) ++cnt[acc(p)];
    uint32_t last_base = lo ? acc(lo - 1) : 4u;
    uint32_t dom = lo ? find_dominant(acc, cnt, lo) : initial_dom;
    const uint32_t range = S.sys_gc_range;
    uint32_t gc_bases = lo < range ? lo : range, gc = 0;
    for (uint32_t p = lo - gc_bases; p < lo; ++p) gc += is_gc(acc(p));
    for (uint32_t pos = lo; pos < hi; ++pos) {
        if (pos == keep_from && kept_state) *kept_state = dist | (start_rate << 24);
        const uint32_t b = acc(pos);
        const Words w = philox(S.seed, pos, c1, c2, kDomSysErr << 28);
        const uint32_t idx[3] = {transform_distance(dist), safe_percent_u16(gc, gc_bases), start_rate};
        const uint32_t dom_error = chain_draw<(int)kQuadsSmall>(S, S.dom_error[(b * 5u + last_base) * 5u + dom], idx, w.w0, 4u);
        const uint32_t rate = chain_draw_rate(S, S.error_rate[b * 5u + dom_error], idx, w.w1);
        if (pos >= keep_from) out[pos] = (uint16_t)(dom_error | (rate << 8));
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

#if RSQ_DEVICE_BUILD
// The same positions for the 64 chunks of a wave, with the expensive part of a position -- an error-rate draw that has to read its rows, about one position in
// thirty -- done for several lanes at once: a lane whose draw the random word does not decide waits (its chunk is its own: nothing orders the lanes of a wave)
// until kChainBatch lanes wait or no lane can go on, and the rows are read and multiplied by a wave most of whose lanes take part instead of two of them.
// Position by position a lane does what sys_chain_chunk does; only when it does it differs.
constexpr uint32_t kChainBatch = 16;        // human-sized chains: 8 -> 0.159 s, 16 -> 0.150, 32 -> 0.178, 48 -> 0.20 at a quarter of the size (0.185 one lane at a time)
template <class Acc>
__device__ void sys_chain_chunk_batched(const DevSim &S, const Acc &acc, uint32_t c1, uint32_t c2, uint32_t lo, uint32_t hi, uint32_t initial_dom, uint32_t &dist,
                                        uint32_t &start_rate, uint16_t *out, uint32_t keep_from, uint32_t *kept_state) {
    uint32_t cnt[4] = {0, 0, 0, 0};
    for (uint32_t p = lo > 5 ? lo - 5 : 0; p < lo; ++p) ++cnt[acc(p)];
    uint32_t last_base = lo ? acc(lo - 1) : 4u;
    uint32_t dom = lo ? find_dominant(acc, cnt, lo) : initial_dom;
    const uint32_t range = S.sys_gc_range;
    uint32_t gc_bases = lo < range ? lo : range, gc = 0;
    for (uint32_t p = lo - gc_bases; p < lo; ++p) gc += is_gc(acc(p));
    uint32_t pos = lo, b = 0, dom_error = 0, word1 = 0;
    bool waiting = false;
    for (;;) {
        const bool left = pos < hi;
        if (!__any(left)) break;
        uint32_t rate = 0;
        bool have_rate = false;
        if (left && !waiting) {
            if (pos == keep_from && kept_state) *kept_state = dist | (start_rate << 24);
            b = acc(pos);
            const Words w = philox(S.seed, pos, c1, c2, kDomSysErr << 28);
            const uint32_t idx[3] = {transform_distance(dist), safe_percent_u16(gc, gc_bases), start_rate};
            dom_error = chain_draw<(int)kQuadsSmall>(S, S.dom_error[(b * 5u + last_base) * 5u + dom], idx, w.w0, 4u);
            word1 = w.w1;
            have_rate = chain_rate_is_zero(S, S.error_rate[b * 5u + dom_error], idx, word1);
            waiting = !have_rate;
        }
        const uint32_t n_wait = (uint32_t)__popcll(__ballot(waiting)), n_go = (uint32_t)__popcll(__ballot(left && !waiting));
        if (n_wait >= kChainBatch || (n_wait && !n_go)) {
            if (waiting) {
                const uint32_t idx[3] = {transform_distance(dist), safe_percent_u16(gc, gc_bases), start_rate};      // the lane's state has not moved while it waited
                rate = chain_draw_rate_rows(S, S.error_rate[b * 5u + dom_error], idx, word1);
                have_rate = true;
                waiting = false;
            }
        }
        if (have_rate) {
            if (pos >= keep_from) out[pos] = (uint16_t)(dom_error | (rate << 8));
            last_base = b;
            ++cnt[b];
            if (pos >= 5) --cnt[acc(pos - 5)];
            dom = find_dominant(acc, cnt, pos + 1);
            update_distances(S.reset_distance, dist, start_rate, rate);
            if (is_gc(b)) ++gc;                                     // Simulator.h:354-366 UpdateGC
            if (gc_bases < range) ++gc_bases;
            else if (is_gc(acc(pos - gc_bases))) --gc;
            ++pos;
        }
    }
}
#endif

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

#if RSQ_DEVICE_BUILD && !defined(RSQ_SPEC)      // the library's kernels: not part of a read kernel compiled for one profile (rsq_spec.h)

// Speculative chunking: pass 0 runs every chunk from a guess of its incoming (dist,start_rate); later passes re-run exactly the
// chunks whose true incoming state (the outgoing state of their left neighbour) differs from the one they used.
// The fixed point is the sequential chain, bit for bit, for any seed and any guess.
// The guess: the chain run from (0,0) over the `warmup` positions in front of the chunk.  Two runs of the chain over the same positions draw from the same
// random numbers whatever their states, and meet for good as soon as both have left their error regions (measured on a human-sized reference with (0,0) as
// the guess at the chunk's own first position: 97 % of the chunks were run a second time, 12 % a third time after 256 more positions, 1.3 % a fourth: the states
// of two runs meet within about 120 positions).  Long chunks with a short run-up keep the second pass small: see chain_chunk_len.
//   k_sys_chain_select (passes > 0): one lane per chunk compares; chunks to run again are appended to `list` (their order does not
//       matter: chunks of one pass are independent), the others keep their outgoing state.  A wave of the run kernel then holds 64 chunks
//       that all have work, whatever share of the chunks changed.
//   k_sys_chain: one lane per listed chunk (pass 0: every chunk, list == nullptr).
RSQ_HD uint32_t chain_incoming(const Chain &ch, uint32_t c, const uint32_t *out_prev, int pass) {
    const uint32_t local = c - ch.first_chunk;
    if (local == 0) return ch.in_state;
    return pass > 0 ? out_prev[c - 1] : 0u;                          // pass 0: the guess (0, 0)
}
__global__ void __launch_bounds__(256) k_sys_chain_select(const Chain *chains, const uint32_t *chunk_chain, uint32_t n_chunks, const uint32_t *used_state, const uint32_t *out_prev,
                                                         uint32_t *out_new, uint32_t *list, uint32_t *n_listed, int pass) {
    // places in the list are reserved once per workgroup (ranks in LDS): one global atomic per wave queues at one L2 channel (see k_sieve_finish)
    __shared__ uint32_t s_n, s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    bool again = false;
    if (c < n_chunks) {
        again = chain_incoming(chains[chunk_chain[c]], c, out_prev, pass) != used_state[c];
        if (!again) out_new[c] = out_prev[c];
    }
    const uint32_t rank = again ? atomicAdd(&s_n, 1u) : 0u;
    __syncthreads();
    if (threadIdx.x == 0 && s_n) s_base = atomicAdd(n_listed, s_n);
    __syncthreads();
    if (again) list[s_base + rank] = c;
}
__global__ void __launch_bounds__(64) k_sys_chain(DevSim S, const Chain *chains, const uint32_t *chunk_chain, const uint32_t *list, uint32_t n_run, uint32_t chunk_len,
                                                 uint32_t warmup, uint32_t *used_state, const uint32_t *out_prev, uint32_t *out_new, int pass) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_run) return;
    const uint32_t c = list ? list[i] : i;
    const Chain ch = chains[chunk_chain[c]];
    const uint32_t local = c - ch.first_chunk;
    const uint32_t want = chain_incoming(ch, c, out_prev, pass);
    ChainAcc acc{S.ref_words, ch.kind, ch.len, ch.kind < 2 ? S.seq_word_off[ch.id] : 0, ch.kind == 2 ? S.adapters[ch.seg].seqs + S.adapters[ch.seg].seq_ptr[ch.id] : nullptr};
    uint32_t dist = want & 0xFFFFFFu, start_rate = want >> 24;
    const uint32_t lo = (ch.chunk_lo + local) * chunk_len, hi = lo + chunk_len < ch.len ? lo + chunk_len : ch.len;
    const uint32_t from = pass == 0 && local ? lo - (warmup < lo ? warmup : lo) : lo;      // pass 0: the guess is the end of a run-up from (0,0)
    uint32_t used = want;
    sys_chain_chunk_batched(S, acc, ch.c1, ch.c2, from, hi, ch.initial_dom, dist, start_rate, ch.out, lo, &used);
    used_state[c] = used;
    out_new[c] = dist | (start_rate << 24);
}

// -V: the chain state in front of every variant's position, per strand (blockIdx.y): the entering state of the position's chunk at the fixed point (used_state),
// folded over the chunk's track up to the position (at most chunk_len - 1 steps).  The host pass over the variants' own bases (variant_sys_errors_strand) needs
// nothing else of the tracks, which therefore stay on the device (12 GB for a human-sized reference).  span: per (sequence, strand) the chain and how many of its
// chunks were run (a rank of a sharded job runs a part); variants outside get state 0, which nobody reads.
struct ChainSpan {
    int32_t chain;
    uint32_t chunks;
};
__global__ void __launch_bounds__(256) k_variant_chain_states(DevSim S, const Chain *chains, const ChainSpan *span, const uint32_t *used_state, uint32_t chunk_len,
                                                             uint32_t n_variants, uint32_t *states /* [2][n_variants] */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, strand = blockIdx.y;
    if (i >= n_variants) return;
    uint32_t seq = 0;                                               // the last sequence whose variants begin at or before i
    for (uint32_t lo = 0, hi = S.n_seqs; lo < hi;) {
        const uint32_t mid = (lo + hi) >> 1;
        if (S.var_ptr[mid] <= i) seq = mid, lo = mid + 1u;
        else hi = mid;
    }
    const ChainSpan sp = span[seq * 2u + strand];
    uint32_t state = 0;
    if (sp.chain >= 0) {
        const Chain ch = chains[sp.chain];
        const uint32_t L = S.seq_len[seq], pos = strand ? L - 1u - S.variants[i].pos : S.variants[i].pos, chunk = pos / chunk_len;
        if (chunk >= ch.chunk_lo && chunk - ch.chunk_lo < sp.chunks) {
            state = used_state[ch.first_chunk + (chunk - ch.chunk_lo)];
            uint32_t dist = state & 0xFFFFFFu, start_rate = state >> 24;
            for (uint32_t p = chunk * chunk_len; p < pos; ++p) update_distances(S.reset_distance, dist, start_rate, (uint32_t)ch.out[p] >> 8);
            state = dist | (start_rate << 24);
        }
    }
    states[(size_t)strand * n_variants + i] = state;
}

// ------------------------------------------------------------------------------------ bias normalisation

// The surrounding factors of Reference::Bias depend on one position each (the start, or the end, of the fragment) and are shared by
// every sampled fragment length: computed once per position (3 table lookups in the 24 MB sur_bias table and an exp each), they turn
// k_sum_bias from a random-access kernel into a streaming one.  Same function, same values, same product order.
// [w_lo, w_hi): the part of the concatenated sequences that is needed (a sharded job computes its share); the tracks begin at w_lo
__global__ void __launch_bounds__(256) k_surrounding_bias_tracks(DevSim S, double *start_bias, double *end_bias, uint64_t w_lo, uint64_t w_hi) {
    const uint64_t at = w_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (at >= w_hi) return;
    uint32_t seq = 0;                                               // the last sequence that begins at or before `at` (empty sequences share their begin)
    for (uint32_t lo = 0, hi = S.n_seqs; lo < hi;) {
        const uint32_t mid = (lo + hi) >> 1;
        if (S.seq_base_off[mid] <= at) seq = mid, lo = mid + 1u;
        else hi = mid;
    }
    const uint32_t L = S.seq_len[seq], pos = (uint32_t)(at - S.seq_base_off[seq]);
    if (pos >= L) return;
    const uint64_t wo = S.seq_word_off[seq];
    uint32_t sur[3];
    surrounding_forward(S.ref_words, wo, L, pos, sur);
    start_bias[at - w_lo] = surrounding_bias(S.sur_bias, sur);
    surrounding_reverse(S.ref_words, wo, L, pos, sur);
    end_bias[at - w_lo] = surrounding_bias(S.sur_bias, sur);
}

// One workgroup = one chunk of kBiasBlock * kBiasRun start positions of one (sequence, sampled length).  A chunk belongs to the share
// [g_lo, g_hi) of the concatenated sequences its first start position lies in; the other chunks' partial results stay zero (the
// ranks of a sharded job add their arrays up: every entry is non-zero on one rank, so the sum is exact whatever the order).
__global__ void __launch_bounds__(256) k_sum_bias(DevSim S, const BiasParam *params, const uint32_t *chunk_param, const uint32_t *chunk_ptr, const double *start_bias,
                                                 const double *end_bias, uint64_t track_base, double *partial_sum, double *partial_max, uint64_t g_lo, uint64_t g_hi) {
    __shared__ double s_sum[kBiasBlock];
    __shared__ double s_max[kBiasBlock];
    const uint32_t param = chunk_param[blockIdx.x], chunk = blockIdx.x - chunk_ptr[param];
    const BiasParam p = params[param];
    const uint32_t L = S.seq_len[p.seq];
    const uint64_t wo = S.seq_word_off[p.seq];
    const uint32_t n_starts = L - p.len + 1;                       // start positions 0 .. L-len (Reference.cpp:645)
    const uint64_t chunk_at = S.seq_base_off[p.seq] + (uint64_t)chunk * kBiasBlock * kBiasRun;
    if (chunk_at < g_lo || chunk_at >= g_hi) return;
    const uint64_t bo = S.seq_base_off[p.seq] - track_base;        // the tracks begin at track_base of the concatenated sequences
    // lane t takes the chunk's start positions t, t + kBiasBlock, ...: neighbouring lanes read neighbouring track entries (coalesced), the
    // G/C count of a fragment comes from the prefix sums.  A lane adds its kBiasRun terms in this order, the tree below adds the lanes.
    const uint32_t chunk_first = chunk * kBiasBlock * kBiasRun;
    double sum = 0.0, mx = 0.0;
    for (uint32_t j = 0; j < kBiasRun; ++j) {
        const uint32_t start = chunk_first + j * kBiasBlock + threadIdx.x;
        if (start >= n_starts) break;
        const uint32_t gc = ref_gc_count_prefix(S.ref_words, S.gc_prefix, wo, start, start + p.len);
        const double bias = start_bias ? p.general_bias * S.gc_bias[percent_u32(gc, p.len)] * start_bias[bo + start] * end_bias[bo + start + p.len - 1u]
                                       : site_bias(S, wo, L, start, p.len, gc, p.general_bias);
        sum += bias;
        mx = bias > mx ? bias : mx;
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
        partial_sum[blockIdx.x] = s_sum[0];
        partial_max[blockIdx.x] = s_max[0];
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
    uint32_t group;                    // the coverage group
    uint32_t sur_start[3];
    bool have_start;
    uint32_t sub;                      // variants of any kind: pass at this start position and what the pass starts from
    VarStart st;
};

RSQ_HD uint32_t site_c1(const SieveSite &site) { return site.seq | (site.sub << 22); }

// Which cells of a start position pass the zero threshold (Simulator.cpp:2304-2306, Simulator.h:415-420).  The reference draws
// probability_chosen ~ U[0,1) for every (start, fragment length) and goes on iff it is >= thr1[length]: the cells are independent, one
// passes with probability 1 - thr1, and given that it passes probability_chosen ~ U[thr1, 1).  The same process drawn directly
// (hard part 3): with q[len] = product of thr1 over the lengths up to len (DevSim::gap_q), the first passing length
// behind cur-1 is the first one with q[len] <= u * q[cur-1] for one uniform u, found by bisection, and its probability_chosen is
// thr1 + v * (1 - thr1) for a second one -- 1 + passes Philox blocks per start position instead of one per four cells.
// Draw k of a start position: block (start, c1, k, 1<<28), u = u53(w0, w1), v = u53(w2, w3).  on_pass(length, probability_chosen).
template <class F>
RSQ_HD uint32_t sieve_gaps(const DevSim &S, const SieveSite &site, F &&on_pass) {
    const double *q = S.gap_q + (size_t)site.group * S.insert_to;
    const uint32_t *seg_end = S.gap_seg_end + (size_t)site.group * S.insert_to;
    const uint32_t c1 = site_c1(site);
    uint32_t cur = S.insert_from, k = 0, n = 0;
    while (cur < S.insert_to) {
        const Words w = philox(S.seed, site.start, c1, k++, kDomSieve << 28);
        const uint32_t e = seg_end[cur];
        const double base = (cur == S.insert_from || seg_end[cur - 1u] == cur) ? 1.0 : q[cur - 1u];      // a segment starts from 1
        const double target = u53_to_unit(w.w0, w.w1) * base;
        uint32_t lo = cur, hi = e;                                  // q does not increase inside a segment
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (q[mid] <= target) hi = mid;
            else lo = mid + 1u;
        }
        if (lo < e) {
            const double thr1 = site.thr[2u * lo + 1u];
            on_pass(lo, thr1 + u53_to_unit(w.w2, w.w3) * (1 - thr1));
            ++n;
            cur = lo + 1u;
        } else cur = e;                                             // nothing passes in this segment: the next one gets a fresh draw
    }
    return n;
}

RSQ_HD uint32_t sieve_cell(const DevSim &S, SieveSite &site, uint32_t len, double probability_chosen, uint32_t (&cnt)[2], uint32_t (&strand_of)[2]) {
    cnt[0] = cnt[1] = 0;
    strand_of[0] = strand_of[1] = 0;
    const double thr0 = site.thr[2u * len], thr1 = site.thr[2u * len + 1u];
    if (!(probability_chosen >= thr1)) return 0;                                    // Simulator.h:418-420
    const uint32_t non_zero_strands = binomial(2u, 1 - thr0, probability_chosen);   // Simulator.cpp:2307
    const uint32_t end = site.start + len;
    if (!non_zero_strands || !(end < site.L)) return 0;                             // :2308,:2318
    const Words w2 = philox(S.seed, site.start, site.seq, len, (kDomSieve << 28) | 1u);
    uint32_t n_chosen;
    if (non_zero_strands <= 1u) {                                                   // :1387-1391 DrawNAlleles(1) -> SelectAllele
        strand_of[0] = (uint32_t)(u32_to_unit(w2.w2) * 2.0) & 1u;
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
    const uint32_t gc = percent_u32(ref_gc_count_prefix(S.ref_words, S.gc_prefix, site.word_off, site.start, end), len);   // :1858-1873
    uint32_t n_here = 0;
    for (uint32_t j = 0; j < n_chosen; ++j) {
        const double u = j ? u53_to_unit(w2.w2, w2.w3) : u53_to_unit(w2.w0, w2.w1);
        const double adjusted_random = thr0 + u * (1 - thr0);                       // :2322
        cnt[j] = fragment_counts(S, site.seq, len, gc, site.sur_start, sur_end, adjusted_random);
        n_here += cnt[j];
    }
    return n_here;
}

// ---- the cell with variants (substitutions only): every allele is a copy of the packed reference with its substitutions applied, so
// the per-allele GC modification and the surrounding edits of Simulator.cpp:1404-1896 are plain reads of that copy -- what the
// reference's own test demands of them (SimulatorTest.cpp:116-195 compares with the sequence that has the variants applied).
constexpr uint32_t kMaxDevAlleles = 128;           // Reference::Variant::kMaxAlleles: 2 * alleles (allele, strand) slots per cell, in scratch memory
RSQ_HD const uint64_t *hap_words(const DevSim &S, uint32_t allele) { return S.hap_stride ? S.ref_words + (1u + allele) * S.hap_stride : S.ref_words; }
RSQ_HD const uint32_t *hap_gc_prefix(const DevSim &S, uint32_t allele) { return S.hap_stride ? S.gc_prefix + (1u + allele) * S.hap_stride : S.gc_prefix; }
template <uint32_t CAP>                            // CAP alleles at most: small sets keep the cell in registers (k_sieve_finish<VM, 8>)
struct VarCellT {
    uint32_t n;                                    // chosen (allele, strand) slots with pairs, in draw order
    uint16_t cnt[2 * CAP];
    uint8_t id[2 * CAP];                           // allele * 2 + strand
};
using VarCell = VarCellT<kMaxDevAlleles>;
// SelectAllele (Simulator.cpp:1341-1361); reverse_selection as a bit mask over the 2 * alleles slots
template <uint32_t CAP>
struct SlotMaskT {
    static constexpr uint32_t kWords = (2 * CAP + 31) / 32;
    uint32_t w[kWords];
    RSQ_HD void set_first(uint32_t n) {
        for (uint32_t i = 0; i < kWords; ++i) w[i] = n >= 32u * (i + 1u) ? 0xFFFFFFFFu : (n > 32u * i ? (1u << (n - 32u * i)) - 1u : 0u);
    }
    RSQ_HD bool test(uint32_t id) const { return (w[id >> 5] >> (id & 31u)) & 1u; }
    RSQ_HD void clear(uint32_t id) { w[id >> 5] &= ~(1u << (id & 31u)); }
};
template <class Mask>
RSQ_HD void select_allele(uint8_t *chosen, uint32_t &n_chosen, Mask &selectable, uint32_t possible_strands, double random_value) {
    uint32_t chosen_id = (uint32_t)(uint16_t)(random_value * (possible_strands - n_chosen));
    uint32_t replacement_correction = 0;
    for (uint32_t i = 0; i < n_chosen; ++i)
        if (chosen[i] <= chosen_id) ++replacement_correction;
    while (replacement_correction)
        if (selectable.test(++chosen_id)) --replacement_correction;
    chosen[n_chosen++] = (uint8_t)chosen_id;
    selectable.clear(chosen_id);
}
RSQ_HD uint32_t word_of(const Words &w, uint32_t k) { return k == 0u ? w.w0 : (k == 1u ? w.w1 : (k == 2u ? w.w2 : w.w3)); }
// Streams (DESIGN.md "Random streams", rows "with variants"): SelectAllele's j-th value = word j&3 of block (start, seq, length,
// 1<<28 | 2 + (j>>2)); the count uniform of the j-th chosen slot = u53 of words 2(j&1), 2(j&1)+1 of block (.., 1<<28 | 128 + (j>>1)).
template <uint32_t CAP>
RSQ_HD uint32_t sieve_cell_var(const DevSim &S, const SieveSite &site, uint32_t len, double probability_chosen, VarCellT<CAP> &cell) {
    cell.n = 0;
    const double thr0 = site.thr[2u * len], thr1 = site.thr[2u * len + 1u];
    if (!(probability_chosen >= thr1)) return 0;                                    // Simulator.h:418-420
    const uint32_t possible_strands = 2u * S.num_alleles;                           // no deletions: every allele is possible (:1330-1340)
    const uint32_t non_zero_strands = binomial(possible_strands, 1 - thr0, probability_chosen);
    const uint32_t end = site.start + len;                                          // end_pos_shift_ is 0 without insertions and deletions
    if (!non_zero_strands || !(end < site.L)) return 0;
    uint8_t chosen[2 * CAP];
    uint32_t n_chosen = 0, n_draws = 0;
    SlotMaskT<CAP> selectable;
    selectable.set_first(possible_strands);
    const bool direct = non_zero_strands <= possible_strands / 2u;                  // ChooseAlleles :1387-1397
    const uint32_t to_draw = direct ? non_zero_strands : possible_strands - non_zero_strands;
    Words ws{0, 0, 0, 0};
    while (n_chosen < to_draw) {
        if (0u == (n_draws & 3u)) ws = philox(S.seed, site.start, site.seq, len, (kDomSieve << 28) | (2u + (n_draws >> 2)));
        select_allele(chosen, n_chosen, selectable, possible_strands, u32_to_unit(word_of(ws, n_draws & 3u)));
        ++n_draws;
    }
    if (!direct) {                                                                  // ReverseSelection :1373-1385
        n_chosen = 0;
        for (uint32_t id = 0; id < possible_strands; ++id)
            if (selectable.test(id)) chosen[n_chosen++] = (uint8_t)id;
    }
    uint32_t n_here = 0;
    Words wc{0, 0, 0, 0};
#pragma unroll 1                                                                   // unrolled over the 16 slots the loop body's gathers and the count draw took 361 vector registers
    for (uint32_t j = 0; j < n_chosen; ++j) {
        const uint32_t allele = chosen[j] >> 1;
        const uint64_t *words = hap_words(S, allele);
        uint32_t sur_start[3], sur_end[3];
        surrounding_forward(words, site.word_off, site.L, site.start, sur_start, S.ref_words);       // bias_mod.surrounding_start_.at(allele)
        surrounding_reverse(words, site.word_off, site.L, end - 1u, sur_end, S.ref_words);           // bias_mod.surrounding_end_.at(allele)
        const uint32_t gc = percent_u32(ref_gc_count_prefix(words, hap_gc_prefix(S, allele), site.word_off, site.start, end), len);   // GetGCPercent with gc_mod_
        if (0u == (j & 1u)) wc = philox(S.seed, site.start, site.seq, len, (kDomSieve << 28) | (128u + (j >> 1)));
        const double u = (j & 1u) ? u53_to_unit(wc.w2, wc.w3) : u53_to_unit(wc.w0, wc.w1);
        const double adjusted_random = thr0 + u * (1 - thr0);                       // :2322
        const uint32_t c = fragment_counts(S, site.seq, len, gc, sur_start, sur_end, adjusted_random);
        if (c) {
            cell.id[cell.n] = chosen[j];
            cell.cnt[cell.n] = (uint16_t)c;
            ++cell.n;
            n_here += c;
        }
    }
    return n_here;
}

// the cell with variants of any kind: possible alleles, ChooseAlleles, and per chosen (allele, strand) the allele's own stretch (rsq_variants.h)
template <uint32_t CAP>
RSQ_HD uint32_t sieve_cell_general(const DevSim &S, const SieveSite &site, uint32_t len, double probability_chosen, VarCellT<CAP> &cell) {
    cell.n = 0;
    const double thr0 = site.thr[2u * len], thr1 = site.thr[2u * len + 1u];
    if (!(probability_chosen >= thr1)) return 0;
    const VarView r = var_view(S, site.seq);
    uint8_t possible[CAP];
    uint32_t n_possible = 0;
    for (uint32_t allele = 0; allele < S.num_alleles; ++allele)                     // GetPossibleAlleles :1330-1340
        if (allele_starts_here(r, site.st, allele, site.start)) possible[n_possible++] = (uint8_t)allele;
    const uint32_t possible_strands = 2u * n_possible;
    const uint32_t non_zero_strands = binomial(possible_strands, 1 - thr0, probability_chosen);
    if (!non_zero_strands) return 0;
    const uint32_t c1 = site_c1(site);
    uint8_t chosen[2 * CAP];
    uint32_t n_chosen = 0, n_draws = 0;
    SlotMaskT<CAP> selectable;
    selectable.set_first(possible_strands);
    const bool direct = non_zero_strands <= possible_strands / 2u;
    const uint32_t to_draw = direct ? non_zero_strands : possible_strands - non_zero_strands;
    Words ws{0, 0, 0, 0};
    while (n_chosen < to_draw) {
        if (0u == (n_draws & 3u)) ws = philox(S.seed, site.start, c1, len, (kDomSieve << 28) | (2u + (n_draws >> 2)));
        select_allele(chosen, n_chosen, selectable, possible_strands, u32_to_unit(word_of(ws, n_draws & 3u)));
        ++n_draws;
    }
    if (!direct) {
        n_chosen = 0;
        for (uint32_t id = 0; id < possible_strands; ++id)
            if (selectable.test(id)) chosen[n_chosen++] = (uint8_t)id;
    }
    uint32_t n_here = 0;
    Words wc{0, 0, 0, 0};
    for (uint32_t j = 0; j < n_chosen; ++j) {
        const uint32_t allele = possible[chosen[j] >> 1], strand = chosen[j] & 1u;
        if (0u == (j & 1u)) wc = philox(S.seed, site.start, c1, len, (kDomSieve << 28) | (128u + (j >> 1)));
        const AlleleView a = allele_view(S, site.seq, allele);
        const AlleleCell ac = allele_cell(a, site.st, site.start, len);
        if (!ac.inside) continue;                                                   // :2318
        uint32_t sur_start[3], sur_end[3];
        allele_surrounding_forward(a, ac.hs, sur_start, ac.first_entries);          // bias_mod.surrounding_start_.at(allele)
        allele_surrounding_reverse(a, ac.he - 1, sur_end, ac.last_entries);         // bias_mod.surrounding_end_.at(allele)
        const double u = (j & 1u) ? u53_to_unit(wc.w2, wc.w3) : u53_to_unit(wc.w0, wc.w1);
        const double adjusted_random = thr0 + u * (1 - thr0);
        const uint32_t c = fragment_counts(S, site.seq, len, ac.gc_percent, sur_start, sur_end, adjusted_random);
        if (c) {
            cell.id[cell.n] = (uint8_t)(allele * 2u + strand);
            cell.cnt[cell.n] = (uint16_t)c;
            ++cell.n;
            n_here += c;
        }
    }
    return n_here;
}

// a cell that passed the zero threshold: (start position slot, fragment length) and its probability_chosen, in loop order
struct SieveCand {
    uint32_t slot, len;
    double probability_chosen;
};
// a cell with fragments, recorded by the sieve pass and expanded into Fragment records after the scan (with variants: one record per
// two chosen (allele, strand) slots of the cell)
struct SieveHit {
    uint32_t slot;         // start position slot of the batch
    uint32_t cand;         // the cell: index in the batch's candidate list
    uint32_t intra;        // pairs of the same cell that come before this record's
    uint16_t len, cnt0, cnt1;
    uint8_t strand0, strand1, allele0, allele1;
};

RSQ_HD Fragment make_fragment(const SieveSite &site, uint32_t len, uint32_t dup, uint32_t strand, uint32_t block_id, uint32_t number, uint32_t allele = 0) {
    Fragment f;
    f.seq = site.seq;
    f.start = site.start;
    f.len = len;
    f.dup = (uint16_t)dup;
    f.strand = (uint8_t)strand;
    f.allele = (uint8_t)allele;
    f.block = block_id;
    f.number = number;
    return f;
}

RSQ_HD void init_site(const DevSim &S, uint32_t block_id, uint32_t offset_in_block, SieveSite &site) {
    site.seq = S.block_seq[block_id];
    site.L = S.seq_len[site.seq];
    site.start = (block_id - S.first_block[site.seq]) * kBlockSize + offset_in_block;
    site.word_off = S.seq_word_off[site.seq];
    site.group = S.coverage_group[site.seq];
    site.thr = S.thresholds + (size_t)site.group * S.insert_to * 2u;
    site.have_start = false;
    site.sub = 0;
    site.st = VarStart{0, 0};
}

// Slots of a batch.  VM 0 / 1 (no variants / substitutions): slot = block * 1000 + offset.  VM 2 (variants of any kind): a block has
// its 1000 start positions plus the extra passes inside inserted bases (DevSim::extra), merged in loop order -- the extra pass j of a
// block (0-based, extras sorted) sits at local index (pos - block start) + j + 1.  Returns the block id; first_slot_of_block = the
// batch slot of the block's first position.
struct SlotInfo {                      // VM 2: what init_site_slot finds for a slot, written once per batch by k_slot_table
    uint32_t block_id, offset_in_block, sub, first_slot;
    int32_t first_variant_id;
    uint32_t start_variant_pos;
};
template <int VM, bool NEED_START = true>                             // NEED_START false: only the position and the pass (what the cell's random stream needs)
RSQ_HD uint32_t init_site_slot(const DevSim &S, uint32_t block_lo, uint32_t block_hi, uint32_t slot, SieveSite &site, uint32_t *first_slot_of_block = nullptr,
                               const SlotInfo *table = nullptr) {
    if (VM == 2 && table) {
        const SlotInfo t = table[slot];
        init_site(S, t.block_id, t.offset_in_block, site);
        site.sub = t.sub;
        site.st = VarStart{t.first_variant_id, t.start_variant_pos};
        if (first_slot_of_block) *first_slot_of_block = t.first_slot;
        return t.block_id;
    }
    if constexpr (VM != 2) {
        const uint32_t block_id = block_lo + slot / kBlockSize;
        init_site(S, block_id, slot % kBlockSize, site);
        if (first_slot_of_block) *first_slot_of_block = slot - slot % kBlockSize;
        return block_id;
    } else {
        const uint32_t base_lo = S.block_extra_ptr[block_lo];
        uint32_t lo = block_lo, hi = block_hi;                      // the last block b with (b - block_lo) * 1000 + extras before b <= slot
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((mid - block_lo) * kBlockSize + (S.block_extra_ptr[mid] - base_lo) <= slot) lo = mid;
            else hi = mid;
        }
        const uint32_t block_id = lo, first = (block_id - block_lo) * kBlockSize + (S.block_extra_ptr[block_id] - base_lo), local = slot - first;
        if (first_slot_of_block) *first_slot_of_block = first;
        const ExtraStart *e = S.extra + S.block_extra_ptr[block_id];
        const uint32_t m = S.block_extra_ptr[block_id + 1] - S.block_extra_ptr[block_id];
        const uint32_t seq = S.block_seq[block_id], bs = (block_id - S.first_block[seq]) * kBlockSize;
        uint32_t a = 0, b = m;                                       // c = extras whose local index is below `local`
        while (a < b) {
            const uint32_t mid = (a + b) >> 1;
            if ((e[mid].pos - bs) + mid + 1u < local) a = mid + 1u;
            else b = mid;
        }
        const uint32_t c = a;
        if (c < m && (e[c].pos - bs) + c + 1u == local) {
            init_site(S, block_id, e[c].pos - bs, site);
            site.sub = e[c].sub;
            site.st = VarStart{e[c].first_variant_id, e[c].start_variant_pos};
        } else {
            init_site(S, block_id, local - c, site);
            if (NEED_START && site.start < site.L) site.st = VarStart{(int32_t)var_view(S, seq).lower_bound(site.start), 0u};      // bias_mod.first_variant_id_ at a plain position
        }
        return block_id;
    }
}

#if RSQ_DEVICE_BUILD && !defined(RSQ_SPEC)      // the library's kernels: not part of a read kernel compiled for one profile (rsq_spec.h)
// The sieve.
//   k_sieve_gaps<VM, false>: one lane per start position slot counts the cells that pass the zero threshold (sieve_gaps);
//   exclusive scan of the counts;
//   k_sieve_gaps<VM, true>: the same walk again writes the batch's candidate list, (slot, length, probability_chosen) in loop order;
//   k_sieve_finish: one lane per candidate -- full lanes -- runs the expensive part (strand / allele choice, G/C percent, surroundings,
//            negative binomial counts), records the cells with fragments in `hits` and the pairs of every candidate in pairs_of;
//   exclusive scan of pairs_of: the candidates are in the order of the reference's loops (block, start, pass, length), so the scan is
//            the position of a cell's first Fragment; k_sieve_emit writes the records (chosen strand order, duplicate).  No atomic
//            decides an order, hence deterministic read ids.
constexpr uint32_t kSieveBlock = 256;

// variants of any kind: what init_site_slot<2> finds, once per slot and batch.  One workgroup per block of 1000 start positions: the
// searches over all blocks / all variants happen once per block, the per-slot ones only over the block's own few extra starts and variants.
__global__ void __launch_bounds__(256) k_slot_table(DevSim S, uint32_t block_lo, SlotInfo *out) {
    const uint32_t block_id = block_lo + blockIdx.x, base_lo = S.block_extra_ptr[block_lo];
    const uint32_t first_slot = (block_id - block_lo) * kBlockSize + (S.block_extra_ptr[block_id] - base_lo);
    const ExtraStart *e = S.extra + S.block_extra_ptr[block_id];
    const uint32_t m = S.block_extra_ptr[block_id + 1] - S.block_extra_ptr[block_id];
    const uint32_t seq = S.block_seq[block_id], bs = (block_id - S.first_block[seq]) * kBlockSize, L = S.seq_len[seq];
    const VarView r = var_view(S, seq);
    __shared__ uint32_t s_v[2];
    if (threadIdx.x < 2) s_v[threadIdx.x] = r.lower_bound(bs + threadIdx.x * kBlockSize);
    __syncthreads();
    const uint32_t v0 = s_v[0], v1 = s_v[1];                       // the block's variants
    for (uint32_t local = threadIdx.x; local < kBlockSize + m; local += blockDim.x) {
        uint32_t a = 0, b = m;                                      // extras whose local index is below `local`
        while (a < b) {
            const uint32_t mid = (a + b) >> 1;
            if ((e[mid].pos - bs) + mid + 1u < local) a = mid + 1u;
            else b = mid;
        }
        SlotInfo t;
        t.block_id = block_id;
        t.first_slot = first_slot;
        if (a < m && (e[a].pos - bs) + a + 1u == local) {
            t.offset_in_block = e[a].pos - bs;
            t.sub = e[a].sub;
            t.first_variant_id = e[a].first_variant_id;
            t.start_variant_pos = e[a].start_variant_pos;
        } else {
            t.offset_in_block = local - a;
            t.sub = 0;
            const uint32_t pos = bs + t.offset_in_block;
            uint32_t lo = v0, hi = v1;                              // first variant at or after the position (none of it is used beyond the sequence)
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (r.v[mid].pos < pos) lo = mid + 1u;
                else hi = mid;
            }
            t.first_variant_id = pos < L ? (int32_t)lo : 0;
            t.start_variant_pos = 0;
        }
        out[first_slot + local] = t;
    }
}

template <int VM, bool FILL>
__global__ void __launch_bounds__(kSieveBlock) k_sieve_gaps(DevSim S, uint32_t block_lo, uint32_t block_hi, uint32_t n_slots, uint32_t *counts, const uint64_t *cand_off,
                                                            SieveCand *cands, uint64_t cand_cap, const SlotInfo *slots) {
    const uint32_t slot = blockIdx.x * kSieveBlock + threadIdx.x;
    if (slot >= n_slots) return;
    SieveSite site;
    init_site_slot<VM, false>(S, block_lo, block_hi, slot, site, nullptr, slots);
    if (!(site.start < site.L)) {                                   // the last block of a sequence is shorter
        if constexpr (!FILL) counts[slot] = 0;
        return;
    }
    if constexpr (FILL) {
        uint64_t at = cand_off[slot];
        sieve_gaps(S, site, [&](uint32_t len, double probability_chosen) {
            if (at < cand_cap) cands[at] = SieveCand{slot, len, probability_chosen};
            ++at;
        });
    } else counts[slot] = sieve_gaps(S, site, [](uint32_t, double) {});
}

// The cells with fragments go into the hit list.  Their order in the list is free (k_sieve_emit places fragments by the scan of pairs_of), so places are taken with an
// atomic counter -- ONE reservation per workgroup: a counter bumped once per wave (557 k times per 10 M pairs) is a queue at one L2 channel, about 10 ns per
// atomic, and was what the kernel's 6.5 ms consisted of for two rounds (VALU 13 % busy, TA 54 %: "latency-bound").  Without variants there is no list at all
// (a cell has one record at most: a word per candidate, k_sieve_emit runs over the candidates).
template <int VM, uint32_t CAP = kMaxDevAlleles>
__global__ void __launch_bounds__(kSieveBlock) k_sieve_finish(DevSim S, uint32_t block_lo, uint32_t block_hi, uint32_t n_slots, const uint64_t *cand_off, const SieveCand *cands,
                                                              uint64_t cand_cap, uint32_t *pairs_of, SieveHit *hits, uint32_t hit_cap, uint32_t *hit_count, const SlotInfo *slots,
                                                              uint32_t *cell_info) {
    __shared__ uint32_t s_records, s_base;
    if (threadIdx.x == 0) s_records = 0;
    __syncthreads();
    const uint64_t c = (uint64_t)blockIdx.x * kSieveBlock + threadIdx.x;
    const bool has_cell = c < cand_cap && c < cand_off[n_slots];
    uint32_t n_here = 0, n_records = 0;                             // pairs of the cell; its records: one per two chosen (allele, strand) slots
    SieveCand cand{};
    VarCellT<VM != 0 ? CAP : 1u> cell;                              // VM 0: the two strands' counts in cnt[0..1], their strands in id[0..1]
    cell.n = 0;
    if (has_cell) {
        cand = cands[c];
        SieveSite site;
        init_site_slot<VM>(S, block_lo, block_hi, cand.slot, site, nullptr, slots);
        if constexpr (VM == 2) n_here = sieve_cell_general(S, site, cand.len, cand.probability_chosen, cell);
        else if constexpr (VM == 1) n_here = sieve_cell_var(S, site, cand.len, cand.probability_chosen, cell);
        else {
            uint32_t cnt[2], strand_of[2];
            n_here = sieve_cell(S, site, cand.len, cand.probability_chosen, cnt, strand_of);
            cell.n = n_here ? 2u : 0u;
            for (uint32_t e = 0; e < 2u; ++e) {
                cell.cnt[e] = (uint16_t)cnt[e];
                cell.id[e] = (uint8_t)strand_of[e];                 // allele 0
            }
        }
        n_records = (cell.n + 1u) / 2u;
    }
    if constexpr (VM == 0) {
        // without variants a cell has one record at most: no list -- what k_sieve_emit needs beyond pairs_of goes into a word per candidate (coalesced), and the
        // emit kernel runs over the candidates
        if (c < cand_cap) {
            pairs_of[c] = n_here;
            cell_info[c] = (uint32_t)cell.cnt[0] | ((uint32_t)(cell.id[0] & 1u) << 16) | ((uint32_t)(cell.id[1] & 1u) << 17);
        }
        return;
    }
    const uint32_t rank = n_records ? atomicAdd(&s_records, n_records) : 0u;
    __syncthreads();
    if (threadIdx.x == 0 && s_records) s_base = atomicAdd(hit_count, s_records);
    __syncthreads();
    uint32_t intra = 0;
    for (uint32_t e = 0; e < cell.n; e += 2u) {
        const bool two = e + 1u < cell.n;
        const uint32_t at = s_base + rank + e / 2u;
        SieveHit h;
        h.slot = cand.slot;
        h.cand = (uint32_t)c;
        h.intra = intra;
        h.len = (uint16_t)cand.len;
        h.cnt0 = cell.cnt[e];
        h.cnt1 = two ? cell.cnt[e + 1u] : (uint16_t)0;
        h.strand0 = cell.id[e] & 1u;
        h.allele0 = cell.id[e] >> 1;
        h.strand1 = two ? cell.id[e + 1u] & 1u : 0;
        h.allele1 = two ? cell.id[e + 1u] >> 1 : 0;
        if (at < hit_cap) hits[at] = h;
        intra += (uint32_t)h.cnt0 + h.cnt1;
    }
    if (c < cand_cap) pairs_of[c] = n_here;
}

// one lane per recorded cell: writes its cnt0 + cnt1 Fragment records at pair_off[cell] + intra; a read's number counts the pairs of
// its block (CreateReadId, Simulator.cpp:596-632): the block's first cell is the first candidate of its first slot
// VM 0: one lane per CANDIDATE (n_hits = their number): the cell's record is put together from pairs_of, cell_info and the candidate itself
template <int VM>
__global__ void __launch_bounds__(256) k_sieve_emit(DevSim S, uint32_t block_lo, uint32_t block_hi, const SieveHit *hits, uint32_t n_hits, const uint64_t *cand_off,
                                                   const uint64_t *pair_off, Fragment *frags, FragmentVar *fvars, const SlotInfo *slots, const SieveCand *cands = nullptr,
                                                   const uint32_t *pairs_of = nullptr, const uint32_t *cell_info = nullptr) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hits) return;
    SieveHit h;
    if constexpr (VM == 0) {
        const uint32_t pairs = pairs_of[i];
        if (!pairs) return;
        const SieveCand cand = cands[i];
        const uint32_t info = cell_info[i];
        h.slot = cand.slot;
        h.cand = i;
        h.intra = 0;
        h.len = (uint16_t)cand.len;
        h.cnt0 = (uint16_t)(info & 0xFFFFu);
        h.cnt1 = (uint16_t)(pairs - (info & 0xFFFFu));
        h.strand0 = (uint8_t)((info >> 16) & 1u);
        h.strand1 = (uint8_t)((info >> 17) & 1u);
        h.allele0 = h.allele1 = 0;
    } else h = hits[i];
    SieveSite site;
    uint32_t first_slot;
    const uint32_t block_id = init_site_slot<VM>(S, block_lo, block_hi, h.slot, site, &first_slot, slots);
    const uint64_t base = pair_off[h.cand];
    const uint32_t number_base = (uint32_t)(base - pair_off[cand_off[first_slot]]);
    uint32_t k = h.intra;
    for (uint32_t e = 0; e < 2u; ++e) {
        const uint32_t cnt = e ? h.cnt1 : h.cnt0, strand = e ? h.strand1 : h.strand0, allele = e ? h.allele1 : h.allele0;
        if (!cnt) continue;
        FragmentVar fv{};
        if constexpr (VM == 2) {                                    // what SimulateFromGivenBlock hands to CreateReads (:2334-2337), derived again
            const AlleleCell ac = allele_cell(allele_view(S, site.seq, allele), site.st, site.start, h.len);
            fv.end = ac.end;
            fv.sub = site.sub;
            fv.start_var = site.st.first_variant_id;
            fv.start_var_pos = site.st.start_variant_pos;
            fv.end_var = ac.end_var.first_variant_id;
            fv.end_var_pos = ac.end_var.start_variant_pos;
        }
        for (uint32_t dup = 0; dup < cnt; ++dup, ++k) {
            frags[base + k] = make_fragment(site, h.len, dup, strand, block_id, number_base + k + 1u, allele);
            if constexpr (VM == 2) fvars[base + k] = fv;
        }
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
// one workgroup: every thread adds up a stretch of tiles, the stretches' sums are scanned in LDS, then every thread turns its stretch into
// exclusive prefixes (tens of thousands of tiles for a batch of 10 M pairs: a single serial thread took a millisecond)
constexpr uint32_t kScanTilesBlock = 1024;
// `init`: what lies in front of the whole array (nullptr: 0) -- the offsets of a sub-range continue where the sub-range in front of it ended
__global__ void __launch_bounds__(kScanTilesBlock) k_scan_tiles(uint64_t *tile_sums, uint32_t n_tiles, uint64_t *total, const uint64_t *init) {
    __shared__ uint64_t s[kScanTilesBlock];
    const uint64_t first = init ? *init : 0u;
    const uint32_t per = (n_tiles + kScanTilesBlock - 1u) / kScanTilesBlock, lo = threadIdx.x * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
    uint64_t acc = 0;
    for (uint32_t i = lo; i < hi; ++i) acc += tile_sums[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = 1; d < kScanTilesBlock; d <<= 1) {              // inclusive scan of the stretches' sums
        const uint64_t v = threadIdx.x >= d ? s[threadIdx.x - d] : 0;
        __syncthreads();
        s[threadIdx.x] += v;
        __syncthreads();
    }
    uint64_t run = first + s[threadIdx.x] - acc;                      // what lies in front of this thread's stretch
    for (uint32_t i = lo; i < hi; ++i) {
        const uint64_t v = tile_sums[i];
        tile_sums[i] = run;
        run += v;
    }
    if (threadIdx.x == kScanTilesBlock - 1u) *total = first + s[threadIdx.x];
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

// ---------------------------------------------------------------------------------------------- FASTQ text
// The read kernel's per-read word arrays are stored word-major: word w of read r lives at p[w * pitch + r], so the 64 lanes
// of a wave (64 consecutive reads at the same position) store, and later load, 256 contiguous bytes.  pitch 1 = one read alone.
struct WordColumn {
    uint32_t *p;            // word 0 of this read
    uint64_t pitch;         // reads per word row
    RSQ_HD uint32_t &at(uint32_t w) const { return p[(uint64_t)w * pitch]; }
};

// Replays the CIGAR bookkeeping of FillReadPart over the stored 2-bit ops (see fill_read_part in rsq_core.h).
template <class Sink>
RSQ_HD void cigar_replay(const WordColumn &ops, const ReadMeta &m, Sink &sink) {
    if (m.plain) {                                                  // the common read: no need to touch the ops
        if (m.n_iter_m) sink.element('M', m.n_iter_m);
        if (m.n_iter_s) sink.element('S', m.n_iter_s);
        if (m.hard_clip) sink.element('H', m.hard_clip);
        return;
    }
    uint32_t it = 0;
    for (int part = 0; part < 2; ++part) {
        const char base = part ? 'S' : 'M';
        const uint32_t n = part ? m.n_iter_s : m.n_iter_m;
        char element = base;
        uint32_t length = 0;
        for (uint32_t i = 0; i < n; ++i, ++it) {
            if (!(it & 15u) && i + 16u <= n && element == base && !ops.at(it >> 4)) {      // 16 plain iterations at once
                length += 16u;
                i += 15u;
                it += 15u;
                continue;
            }
            const uint32_t code = (ops.at(it >> 4) >> ((it & 15u) * 2u)) & 3u;
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

template <class Derived>
struct TextOps {                        // what a record is made of, on top of Derived::ch
    RSQ_HD Derived &self() { return *static_cast<Derived *>(this); }
    // four characters per push (the sinks take up to four bytes at once): the id line is mostly fixed text
    RSQ_HD void str(const char *s, uint32_t len) {
        uint32_t i = 0;
        for (; i + 4u <= len; i += 4u)
            self().bytes((uint32_t)(uint8_t)s[i] | ((uint32_t)(uint8_t)s[i + 1u] << 8) | ((uint32_t)(uint8_t)s[i + 2u] << 16) | ((uint32_t)(uint8_t)s[i + 3u] << 24), 4u);
        if (i < len) {
            uint32_t w = 0;
            for (uint32_t k = 0; i + k < len; ++k) w |= (uint32_t)(uint8_t)s[i + k] << (8u * k);
            self().bytes(w, len - i);
        }
    }
    // decimal digits without a buffer: peeled from the least significant end into a register, most significant digit lowest, then handed to the
    // sink four at a time
    RSQ_HD void num(uint32_t v) {              // 32-bit: division by 10 is a multiply and a shift
        uint64_t acc = 0;
        uint32_t n = 0;
        do {
            acc = (acc << 8) | (uint64_t)('0' + v % 10u);
            v /= 10u;
            ++n;
        } while (v && n < 8u);
        if (v) {                               // nine or ten digits: the leading ones first
            uint32_t hi = 0, nh = 0;
            do {
                hi = (hi << 8) | ('0' + v % 10u);
                v /= 10u;
                ++nh;
            } while (v);
            self().bytes(hi, nh);
        }
        self().bytes((uint32_t)acc, n < 4u ? n : 4u);
        if (n > 4u) self().bytes((uint32_t)(acc >> 32), n - 4u);
    }
    RSQ_HD void nine_digits(uint32_t v) {      // v < 10^9 with its leading zeros
        uint32_t low = 0, mid = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            low = (low << 8) | ('0' + v % 10u);
            v /= 10u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mid = (mid << 8) | ('0' + v % 10u);
            v /= 10u;
        }
        self().ch((char)('0' + v));
        self().bytes(mid, 4u);
        self().bytes(low, 4u);
    }
    RSQ_HD void num(uint64_t v) {              // beyond 32 bits (read numbers of a job of more than 4 G pairs): groups of nine digits, no buffer
        if (v <= 0xFFFFFFFFull) return num((uint32_t)v);
        const uint64_t kE9 = 1000000000ull, upper = v / kE9;
        if (upper >= kE9) {
            num((uint32_t)(upper / kE9));
            nine_digits((uint32_t)(upper % kE9));
        } else num((uint32_t)upper);
        nine_digits((uint32_t)(v % kE9));
    }
    RSQ_HD void element(char op, uint32_t count) {
        num(count);
        self().ch(op);
    }
};
template <class P>
struct TextSinkT : TextOps<TextSinkT<P>> {      // appends characters at p (no null check: LDS offset 0 is a valid destination)
    P p;
    uint32_t n;
    RSQ_HD TextSinkT(P dst, uint32_t at) : p(dst), n(at) {}
    RSQ_HD void ch(char c) {
        p[n] = c;
        ++n;
    }
    RSQ_HD void bytes(uint32_t word, uint32_t count) {        // count <= 4 characters, the first in the low byte
        for (uint32_t i = 0; i < count; ++i) ch((char)(word >> (8u * i)));
    }
};
using TextSink = TextSinkT<char *>;

// The same stream written with aligned 4-byte stores: bytes collect in a register and leave a word at a time; only the
// bytes before the first and after the last aligned word of the destination are stored singly (neighbouring records of
// other lanes share those words).
template <class P>
struct WordPtr {
    using type = uint32_t *;
};
#if defined(__HIP_DEVICE_COMPILE__)
template <>
struct WordPtr<RSQ_LDS char *> {
    using type = RSQ_LDS uint32_t *;
};
#endif
template <class P>
struct WordSinkT : TextOps<WordSinkT<P>> {
    P p;                 // next destination byte not yet stored
    uint64_t acc;        // pending bytes, first in the low byte
    uint32_t pending, lead, n;
    RSQ_HD explicit WordSinkT(P dst) : p(dst), acc(0), pending(0), lead((4u - ((uint32_t)(uintptr_t)dst & 3u)) & 3u), n(0) {}
    RSQ_HD void push(uint32_t bytes, uint32_t count) {       // count <= 4 bytes, first in the low byte, the rest zero
        acc |= (uint64_t)bytes << (8u * pending);
        pending += count;
        n += count;
        while (lead && pending) {
            *p = (char)(acc & 0xFFu);
            p += 1;
            acc >>= 8;
            --pending;
            --lead;
        }
        if (!lead && pending >= 4u) {
            *reinterpret_cast<typename WordPtr<P>::type>(p) = (uint32_t)acc;
            p += 4;
            acc >>= 32;
            pending -= 4u;
        }
    }
    RSQ_HD void ch(char c) { push((uint8_t)c, 1u); }
    RSQ_HD void bytes(uint32_t word, uint32_t count) { push(count < 4u ? word & ((1u << (8u * count)) - 1u) : word, count); }
    RSQ_HD void finish() {
        while (pending) {
            *p = (char)(acc & 0xFFu);
            p += 1;
            acc >>= 8;
            --pending;
        }
    }
};

RSQ_HD uint32_t digits_u32(uint32_t v) {
    uint32_t n = 1;
    while (v >= 10u) {
        v /= 10u;
        ++n;
    }
    return n;
}
RSQ_HD uint32_t digits_u64(uint64_t v) {
    if (v <= 0xFFFFFFFFull) return digits_u32((uint32_t)v);
    uint32_t n = 1;
    while (v >= 10) {
        v /= 10;
        ++n;
    }
    return n;
}


// One FASTQ record "@id\nSEQ\n+\nQUAL\n" with the id of Simulator.cpp:596-632: the id line ...
template <class Sink>
// (the fragment and its variant part by reference and two flags, not by pointers that may be null: a pointer chosen at run time puts the structure into scratch memory)
RSQ_HD void format_header(const DevSim &S, const NameTable &names, bool has_f, const Fragment &f, uint64_t adapter_only_number, const ReadMeta &m, const WordColumn &ops, Sink &t,
                          bool has_fv, const FragmentVar &fv) {
    t.ch('@');
    t.str(names.base_identifier, names.base_len);
    if (has_f) {
        const uint32_t end = has_fv ? fv.end : f.start + f.len;                  // end_position_forward of CreateReads
        t.num(f.block);
        t.ch('_');
        t.num(f.number);
        if (1u < S.num_alleles) {                                    // Simulator.cpp:612-614
            t.str("_allele", 7);
            t.num((uint32_t)f.allele);
        }
        t.ch(':');
        t.num(f.strand ? end : f.start + 1u);
        t.ch(':');
        t.str(names.names + names.name_ptr[f.seq], names.name_ptr[f.seq + 1] - names.name_ptr[f.seq]);
        t.ch(':');
        t.num(f.strand ? f.start + 1u : end);
    } else {
        t.ch('0');
        t.ch('_');
        t.num(adapter_only_number);
        t.str(":0:Adapter:0", 12);
    }
    t.ch(':');
    t.num((uint32_t)S.tiles[m.tile_id]);
    t.str(":1337:1337 ", 11);
    cigar_replay(ops, m, t);
    t.str(" E", 2);
    t.num((uint32_t)m.num_errors);
    t.ch('\n');
}
template <class Sink>
RSQ_HD void format_header(const DevSim &S, const NameTable &names, const Fragment *f, uint64_t adapter_only_number, const ReadMeta &m, const WordColumn &ops, Sink &t,
                          const FragmentVar *fv = nullptr) {
    format_header(S, names, f != nullptr, f ? *f : Fragment{}, adapter_only_number, m, ops, t, fv != nullptr, fv ? *fv : FragmentVar{});
}
// ... and one of its two data lines: the bases ("SEQ\n+\n", is_qual false) or the qualities ("QUAL\n").  The read kernel
// leaves both as bytes in 16-byte aligned rows; four base codes become four letters with one byte permute.
RSQ_HD uint32_t base_letters(uint32_t codes) {                       // bytes 0..3 -> "ACGT", 4 -> 'N'
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(0x4E4E4E4Eu, 0x54474341u, codes);   // selector 0-3: bytes of "ACGT", 4-7: 'N'
#else
    uint32_t out = 0;
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t b = (codes >> (8u * k)) & 0xFFu;
        out |= (uint32_t)("ACGTN"[b < 4u ? b : 4u]) << (8u * k);
    }
    return out;
#endif
}
// words [first_word, first_word + n_words) of the line, then (with_end) the line end
template <class Sink>
RSQ_HD void format_line_part(const WordColumn &row, uint32_t read_len, bool is_qual, uint32_t first_word, uint32_t n_words, bool with_end, Sink &t) {
    const uint32_t all_words = (read_len + 3u) >> 2, end_word = first_word + n_words < all_words ? first_word + n_words : all_words;
    constexpr uint32_t kAhead = 10u;                                 // loads in flight
    for (uint32_t i = first_word; i < end_word; i += kAhead) {
        uint32_t w[kAhead];
#pragma unroll
        for (uint32_t k = 0; k < kAhead; ++k) w[k] = i + k < end_word ? row.at(i + k) : 0u;
#pragma unroll
        for (uint32_t k = 0; k < kAhead; ++k) {
            const uint32_t at = 4u * (i + k);
            if (i + k >= end_word) break;
            const uint32_t left = read_len - at, text = is_qual ? w[k] : base_letters(w[k]);
            if (left >= 4u) t.push(text, 4u);
            else t.push(text & ((1u << (8u * left)) - 1u), left);
        }
    }
    if (with_end) {
        if (is_qual) t.push('\n', 1u);
        else t.push('\n' | ('+' << 8) | ('\n' << 16), 3u);
    }
}
template <class Sink>
RSQ_HD void format_line(const WordColumn &row, uint32_t read_len, bool is_qual, Sink &t) {
    format_line_part(row, read_len, is_qual, 0u, (read_len + 3u) >> 2, true, t);
}
template <class P>
RSQ_HD uint32_t format_record(const DevSim &S, const NameTable &names, bool has_f, const Fragment &f, uint64_t adapter_only_number, const ReadMeta &m, const WordColumn &seq,
                              const WordColumn &qual, const WordColumn &ops, P dst, bool has_fv, const FragmentVar &fv) {
    WordSinkT<P> t(dst);
    format_header(S, names, has_f, f, adapter_only_number, m, ops, t, has_fv, fv);
    format_line(seq, m.read_len, false, t);
    format_line(qual, m.read_len, true, t);
    t.finish();
    return t.n;
}
template <class P>
RSQ_HD uint32_t format_record(const DevSim &S, const NameTable &names, const Fragment *f, uint64_t adapter_only_number, const ReadMeta &m, const WordColumn &seq,
                              const WordColumn &qual, const WordColumn &ops, P dst, const FragmentVar *fv = nullptr) {
    return format_record(S, names, f != nullptr, f ? *f : Fragment{}, adapter_only_number, m, seq, qual, ops, dst, fv != nullptr, fv ? *fv : FragmentVar{});
}

// length of that record without producing it (the read kernel writes it next to the read)
RSQ_HD uint32_t record_size(const DevSim &S, const NameTable &names, bool has_f, const Fragment &f, uint64_t adapter_only_number, const ReadMeta &m, bool has_fv, const FragmentVar &fv) {
    uint32_t n = 1u + names.base_len;
    if (has_f) {
        const uint32_t end = has_fv ? fv.end : f.start + f.len;
        if (1u < S.num_alleles) n += 7u + digits_u64(f.allele);
        n += digits_u64(f.block) + 1u + digits_u64(f.number) + 1u + digits_u64(f.strand ? end : f.start + 1u) + 1u +
             (names.name_ptr[f.seq + 1] - names.name_ptr[f.seq]) + 1u + digits_u64(f.strand ? f.start + 1u : end);
    } else n += 2u + digits_u64(adapter_only_number) + 12u;
    n += 1u + digits_u64(S.tiles[m.tile_id]) + 11u + m.cigar_chars + 2u + digits_u64(m.num_errors) + 1u;
    return n + 2u * m.read_len + 4u;
}
RSQ_HD uint32_t record_size(const DevSim &S, const NameTable &names, const Fragment *f, uint64_t adapter_only_number, const ReadMeta &m, const FragmentVar *fv = nullptr) {
    return record_size(S, names, f != nullptr, f ? *f : Fragment{}, adapter_only_number, m, fv != nullptr, fv ? *fv : FragmentVar{});
}

// ------------------------------------------------------------------------------------------------- reads
struct ReadOut {                        // destination of one lane's read; bases and qualities leave in 4-byte stores
    WordColumn seq, qual, ops;
    uint32_t cur_word, cur_index;       // CIGAR ops: 2 bits per iteration, 16 per word
    uint32_t seq_word, qual_word, n_put;
    RSQ_HD void put(uint32_t pos, uint32_t base, uint32_t qual_char) {      // pos runs 0,1,2,... (read_pos)
        const uint32_t sh = (pos & 3u) * 8u;
        seq_word |= base << sh;
        qual_word |= qual_char << sh;
        n_put = pos + 1u;
        if ((pos & 3u) == 3u) {
            seq.at(pos >> 2) = seq_word;
            qual.at(pos >> 2) = qual_word;
            seq_word = qual_word = 0;
        }
    }
    RSQ_HD void op(uint32_t it, uint32_t code) {
        const uint32_t wi = it >> 4;
        if (wi != cur_index) {
            ops.at(cur_index) = cur_word;
            cur_word = 0;
            cur_index = wi;
        }
        cur_word |= code << ((it & 15u) * 2u);
    }
    RSQ_HD void finish() {
        ops.at(cur_index) = cur_word;
        if (n_put & 3u) {
            seq.at(n_put >> 2) = seq_word;
            qual.at(n_put >> 2) = qual_word;
        }
    }
};
RSQ_HD ReadOut make_read_out(const WordColumn &seq, const WordColumn &qual, const WordColumn &ops) { return ReadOut{seq, qual, ops, 0u, 0u, 0u, 0u, 0u}; }

struct RawLayout {                      // word-major arrays of the read kernel, read index = segment * n_pairs + pair
    uint32_t *seq, *qual;               // [read_words][pitch]: 4 bases / 4 quality characters per word
    uint32_t *ops;                      // [ops_words][pitch]: 16 two-bit CIGAR ops per word
    ReadMeta *meta;
    uint64_t pitch;                     // reads per word row (>= number of reads)
    uint64_t *templates;                // --methylation: [reads][template_words] converted templates, else nullptr
    uint32_t template_words;
    const uint32_t *order;              // seqToIllumina, after a read kernel that ran binned by tile: row r holds record order[r]; nullptr: record r
    RSQ_HD uint64_t item_of(uint64_t row) const { return order ? order[row] : row; }
    RSQ_HD WordColumn seq_of(uint64_t r) const { return WordColumn{seq + r, pitch}; }
    RSQ_HD WordColumn qual_of(uint64_t r) const { return WordColumn{qual + r, pitch}; }
    RSQ_HD WordColumn ops_of(uint64_t r) const { return WordColumn{ops + r, pitch}; }
    RSQ_HD ReadOut out_of(uint64_t r) const { return make_read_out(seq_of(r), qual_of(r), ops_of(r)); }
};

struct FragmentSrc {                    // template of one mate cut from the 2-bit reference (Reference.cpp:483-496)
    const uint64_t *words;
    uint64_t word_off;
    uint32_t first;                     // forward: start position; reverse: end position
    uint32_t len;
    bool reverse;
    const uint16_t *sys_;               // systematic errors at the first template base
    const uint64_t *converted;          // --methylation: the template after CTConversion, 2 bits per base in read orientation; else nullptr
    const uint32_t *gc_prefix;          // DevSim::gc_prefix
    // Per-lane streams: a load instruction of the wave touches 64 cache lines here, so the source holds what it last read -- the 64-bit word of the
    // template (32 bases; of the reference or of the converted template, a source reads only one of them) and a group of four systematic errors
    // (the tracks end in 8 spare entries, pack_reference).
    mutable uint32_t held_word = 0xFFFFFFFFu, held_sys = 0xFFFFFFFFu;
    mutable uint64_t word = 0, sys4 = 0;
    RSQ_HD uint64_t template_word(const uint64_t *from, uint32_t index) const {
        if (index != held_word) {
            held_word = index;
            word = from[index];
        }
        return word;
    }
    RSQ_HD uint32_t org_len() const { return len; }
    RSQ_HD uint32_t ref(uint32_t k) const {
        const uint32_t pos = reverse ? first - 1u - k : first + k, b = (uint32_t)(template_word(words + word_off, pos >> 5) >> ((pos & 31u) * 2u)) & 3u;
        return reverse ? 3u - b : b;
    }
    RSQ_HD uint32_t base(uint32_t k) const { return converted ? (uint32_t)(template_word(converted, k >> 5) >> ((k & 31u) * 2u)) & 3u : ref(k); }
    RSQ_HD uint32_t sys_base(uint32_t k) const {
        if ((k >> 2) != held_sys) {
            held_sys = k >> 2;
#if defined(__HIP_DEVICE_COMPILE__)
            sys4 = *reinterpret_cast<const uint64_t __attribute__((aligned(2))) *>(sys_ + (k & ~3u));
#else
            memcpy(&sys4, sys_ + (k & ~3u), 8);
#endif
        }
        return (uint32_t)(sys4 >> ((k & 3u) * 16u)) & 0xFFFFu;
    }
    RSQ_HD uint32_t sys_deleted(uint32_t k) const { return sys_base(k); }
    // Simulator.cpp:482-489 without a load per base: the G/C count of the template's reference range from the per-word prefix sums
    // (the complement strand has the same count), the error rates four per 8-byte load
    RSQ_HD void totals(uint32_t n, uint32_t &gc, uint32_t &rate_sum) const {
        if (!converted && !gc_prefix) return template_totals_loop(*this, n, gc, rate_sum);
        if (converted) gc += ref_gc_count(converted, 0, 0, n);                  // the converted template is packed like the reference, from base 0
        else gc += reverse ? ref_gc_count_prefix(words, gc_prefix, word_off, first - n, first) : ref_gc_count_prefix(words, gc_prefix, word_off, first, first + n);
        rate_sum += rate_total(n);
    }
    // eight entries per 16-byte load (the tracks end in 8 spare entries); the rate is an entry's high byte
    RSQ_HD uint32_t rate_total(uint32_t n) const {
        struct __attribute__((packed, aligned(2))) Eight {
            uint64_t a, b;
        };
        uint32_t sum = 0;
        for (uint32_t k = 0; k < n; k += 8u) {
            Eight e;
#if defined(__HIP_DEVICE_COMPILE__)
            e = *reinterpret_cast<const Eight *>(sys_ + k);
#else
            memcpy(&e, sys_ + k, 16);
#endif
            const uint32_t left = n - k;                               // entries of this group that count
            if (left < 8u) {
                if (left <= 4u) {
                    e.b = 0;
                    if (left < 4u) e.a &= (1ull << (16u * left)) - 1ull;
                } else e.b &= (1ull << (16u * (left - 4u))) - 1ull;
            }
            const uint64_t kHigh = 0x00FF00FF00FF00FFull, kAdd = 0x0001000100010001ull;
            sum += (uint32_t)((((e.a >> 8) & kHigh) * kAdd) >> 48) + (uint32_t)((((e.b >> 8) & kHigh) * kAdd) >> 48);
        }
        return sum;
    }
};

// ------------------------------------------------------------------------------------- bisulfite conversion (a16)
// Simulator::CTConversion (Simulator.cpp:1925-2247): a C of a template becomes a T with probability 1 - methylation where the template lies in an
// unmethylated region of the BED file; once per (start, length, strand) site and mate, so that all duplicates of a site share the converted template.
// The uniform of template position k is word k&3 of Philox block (start, sequence, length, 7<<28 | reversed<<27 | k>>2) -- a pure function of k, so
// the walk below may visit positions in any grouping.
//
// The reference walks template and reference base by base, in three overloads times two mirrored directions.  Here ONE walk serves both strands and
// both cases (with and without variants): positions are taken in the strand's own direction (MethSide: x = pos on the forward strand, -pos on the
// reverse strand, so regions and variants are met in increasing x either way), and the walk advances by EVENTS -- a region's entry and exit, the
// variant the cursor points at, the template's end -- converting whole runs of template positions at once (the C's of a run are found 32 bases per
// word).  What the reference's walk does beyond the plain geometry is kept, because it decides bytes of the output (DESIGN.md section 1 lists it):
//   * the template position is 16 bits wide (uintReadLen): a jump over more than 65535 bases wraps, and the walk goes on if the wrapped value is
//     below the template length;
//   * the reverse mate's walk begins at the fragment's end position (one past its last base), not at the last base;
//   * without variants the reverse walk never enters the sequence's first region (`while(cur_meth && ...)`); with variants it does;
//   * a deleted base inside a region takes a template position (without converting it);
//   * only the variant under the cursor is looked at: variants of other alleles at a position are passed over when the walk stands on them, but a
//     second variant at the position of one that was just used stays under the cursor and hides all later ones until the next stretch without regions;
//   * the reference reads its `deletion` flag before writing it (Simulator.cpp:2026): here it starts as false.
struct MethView {
    const uint32_t *first, *second;
    const double *rate;                 // of the allele asked for: rate[region * stride]
    uint32_t n, stride;
    RSQ_HD double rate_of(int32_t region) const { return rate[(size_t)region * stride]; }
};
// Reference::Unmethylation(seq, allele): meth_rate holds num_alleles values per region (a file with one column repeats it)
RSQ_HD MethView meth_view(const DevSim &S, uint32_t seq, uint32_t allele = 0) {
    const uint32_t off = S.meth_ptr[seq];
    return MethView{S.meth_first + off, S.meth_second + off, S.meth_rate + (size_t)off * S.num_alleles + allele, S.meth_ptr[seq + 1] - off, S.num_alleles};
}
// cur_methylation_start of SimulateFromGivenBlock (:2273,:2293-2297, CreateBlock :1214-1219): the first region that ends after pos
RSQ_HD uint32_t meth_start_index(const MethView &m, uint32_t pos) {
    uint32_t lo = 0, hi = m.n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (m.second[mid] <= pos) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
constexpr uint32_t kTemplateWordsMax = 64;      // 2048 template bases
struct MethDraws {                      // lazily evaluated Philox blocks of one template
    uint64_t seed;
    uint32_t c0, c1, c2, c3base, have;
    Words w;
    RSQ_HD double uniform(uint32_t k) {
        if (have != (k >> 2)) {
            w = philox(seed, c0, c1, c2, c3base | (k >> 2));
            have = k >> 2;
        }
        const uint32_t j = k & 3u;
        return u32_to_unit(j == 0u ? w.w0 : (j == 1u ? w.w1 : (j == 2u ? w.w2 : w.w3)));
    }
};
// template positions [t, t + n) lie in a region with conversion probability `rate`: the C's among them (code 1: low bit set, high bit clear), word by word
RSQ_HD void ct_convert_run(uint64_t *tmpl, uint32_t t, uint32_t n, double rate, MethDraws &d) {
    const uint32_t end = t + n;
    for (uint32_t w = t >> 5; (w << 5) < end; ++w) {
        const uint32_t lo = t > (w << 5) ? t - (w << 5) : 0u, hi = end - (w << 5) < 32u ? end - (w << 5) : 32u;      // bases lo .. hi-1 of the word
        uint64_t cs = tmpl[w] & ~(tmpl[w] >> 1) & 0x5555555555555555ull;
        cs &= (hi == 32u ? ~0ull : (1ull << (2u * hi)) - 1ull) & ~((1ull << (2u * lo)) - 1ull);
        while (cs) {
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t bit = (uint32_t)__ffsll((long long)cs) - 1u;
#else
            const uint32_t bit = (uint32_t)__builtin_ctzll(cs);
#endif
            cs &= cs - 1ull;
            if (d.uniform((w << 5) + (bit >> 1)) < rate) tmpl[w] |= (uint64_t)3u << bit;                             // C (1) -> T (3)
        }
    }
}
// regions and variants as the walk of one strand meets them
template <bool REV>
struct MethSide {
    const MethView &m;
    const VarView *r;                   // nullptr: no variants loaded
    uint32_t allele;
    int32_t lowest;                     // the reverse walk's last region: 1 without variants, 0 with
    RSQ_HD int64_t coord(uint32_t pos) const { return REV ? -(int64_t)pos : (int64_t)pos; }
    RSQ_HD int64_t entry(int32_t i) const { return REV ? 1 - (int64_t)m.second[i] : (int64_t)m.first[i]; }      // the first x inside the region
    RSQ_HD int64_t exit(int32_t i) const { return REV ? 1 - (int64_t)m.first[i] : (int64_t)m.second[i]; }       // the first x behind it
    RSQ_HD bool region(int32_t i) const { return REV ? i >= lowest : i < (int32_t)m.n; }
    RSQ_HD bool variant(int32_t j) const { return r && (REV ? j >= 0 : j < (int32_t)r->n); }
    RSQ_HD static int32_t next(int32_t i) { return REV ? i - 1 : i + 1; }
    RSQ_HD int64_t at(int32_t j) const { return coord(r->v[j].pos); }
    RSQ_HD uint32_t len(int32_t j) const { return r->v[j].len; }
    RSQ_HD bool mine(int32_t j) const { return r->in_allele(r->v[j], allele); }
    // the region the walk begins in or in front of: forward the first that ends behind the position, reverse the last that begins at or before it
    RSQ_HD int32_t first_region(uint32_t pos) const {
        if (!REV) return (int32_t)meth_start_index(m, pos);
        uint32_t lo = 0, hi = m.n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (m.first[mid] <= pos) lo = mid + 1;
            else hi = mid;
        }
        return (int32_t)lo - 1;
    }
};
template <bool REV>
RSQ_HD void methylation_walk(uint64_t *tmpl, uint32_t length, const MethView &m, const VarView *r, uint32_t allele, uint32_t start_pos, VarStart from, MethDraws &d) {
    const MethSide<REV> side{m, r, allele, r ? 0 : 1};
    int64_t x = side.coord(start_pos);
    uint16_t t = 0;                                                   // uintReadLen
    int32_t region = side.first_region(start_pos), var = r ? from.first_variant_id : -1;
    uint32_t left = 0;                                                // bases of the variant at x the walk has not passed yet
    if (side.variant(var) && side.at(var) == x && side.len(var) > 1u && side.mine(var)) left = side.len(var) - from.start_variant_pos;
    // the stretch without information in front of `region`: the template position moves on by what the allele holds there
    auto skip_to = [&](int32_t reg) {
        if (left) {
            t = (uint16_t)(t + left);
            left = 0;
            ++x;
            var = side.next(var);
        }
        while (side.variant(var) && side.at(var) < side.entry(reg) && t < length) {
            if (side.mine(var)) {
                t = (uint16_t)(t + (uint16_t)(side.at(var) - x) + (uint16_t)side.len(var));
                x = side.at(var) + 1;
            }
            var = side.next(var);
        }
        t = (uint16_t)(t + (uint16_t)(side.entry(reg) - x));
        x = side.entry(reg);
    };
    if (side.region(region) && side.entry(region) > x) skip_to(region);
    while (side.region(region) && t < length) {
        const double rate = m.rate_of(region);
        const int64_t out = side.exit(region);
        while (x < out && t < length) {
            if (!left) {                                              // does a variant of the allele begin here?
                while (side.variant(var) && side.at(var) == x && !side.mine(var)) var = side.next(var);
                if (side.variant(var) && side.at(var) == x) {
                    if (0u == side.len(var)) {                        // the deleted base: a template position passes unconverted
                        var = side.next(var);
                        ++x;
                        ++t;
                        continue;
                    }
                    left = side.len(var);
                }
            }
            uint32_t run;
            if (left) {                                               // the variant's bases, all at this x
                run = left < length - t ? left : length - t;
                left -= run;
                if (!left) {
                    var = side.next(var);
                    ++x;
                }
            } else {                                                  // reference bases up to the region's end or the variant under the cursor
                int64_t until = out;
                if (side.variant(var) && side.at(var) > x && side.at(var) < until) until = side.at(var);
                run = until - x < (int64_t)(length - t) ? (uint32_t)(until - x) : length - t;
                x += run;
            }
            ct_convert_run(tmpl, t, run, rate, d);
            t = (uint16_t)(t + run);
        }
        region = side.next(region);
        if (side.region(region) && side.entry(region) > x) skip_to(region);
    }
}
// CTConversion of one mate's template: `start_pos` = the fragment's start (forward mate) or END position (reverse mate), `r` = the sequence's variants or nullptr
RSQ_HD void ct_conversion(uint64_t *tmpl, uint32_t length, const MethView &m, const VarView *r, uint32_t allele, uint32_t start_pos, bool reversed, VarStart from, MethDraws &d) {
    if (reversed) methylation_walk<true>(tmpl, length, m, r, allele, start_pos, from, d);
    else methylation_walk<false>(tmpl, length, m, r, allele, start_pos, from, d);
}

struct EmptySrc {                       // adapter-only pair: org_seq_ = "" (Simulator.cpp:2369-2371)
    RSQ_HD void totals(uint32_t, uint32_t &, uint32_t &) const {}
    RSQ_HD uint32_t org_len() const { return 0; }
    RSQ_HD uint32_t base(uint32_t) const { return 0; }
    RSQ_HD uint32_t sys_base(uint32_t) const { return 0; }
    RSQ_HD uint32_t sys_deleted(uint32_t) const { return 0; }
};

RSQ_HD uint32_t draw_tile(const DevSim &S, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3base) {      // Simulator.h:176-181
    if (RSQ_SIM(S, n_tiles) > 1) return discrete_draw(S.tile_cp, RSQ_SIM(S, n_tiles), u32_to_unit(philox(S.seed, c0, c1, c2, c3base).w0));
    return 0;
}

// ------------------------------------------------------------------------------------------- LDS staging
// A lane evaluates about 250 table entries per base (quality K = 40 over four margins, base call and indel over four and three).
// Measured on gfx950 (exp/ta_bench.hip): a wave-level global_load_dwordx4 costs the CU's vector-memory path >= 23 cycles (64 lanes
// x 16 B returned at 64 B/clk) however few lanes are active, and about 2.3 cycles per distinct cache line touched; a ds_read_b128
// costs 8.  So the read kernel draws SCREENED (rsq_core.h): single-precision copies of the tables, four columns per 16-byte load,
// and the rows whose addresses scatter most across the lanes of a wave live in the workgroup's LDS image (LdsPlan, rsq_types.h):
//  * rows chosen by per-read state -- quality margins over sequence quality (0) and previous quality (1), base-call margins over
//    the quality (0) and the number of errors (2), indel margin over the indel position (0);
//  * the first rows of the error-rate margins (88 % of all positions have rate 0).  A lane whose rate is not staged reads its
//    own row from HBM (MixedRow32, rsq_core.h).
// The rows over the read position and the read's G/C percent stay in HBM (L2): the lanes of a wave share the position rows (4
// cache lines per load).  A draw the screen cannot decide is repeated in double precision from HBM (GlobalTables).
// Image layout (32-bit words), Ti = LdsPlan::img_tiles: descriptors [quality 4 Ti][base_call 20 Ti][indels 12][seq_quality Ti] (18 words
// each), the outcome values of these tables, the outcome values by column and the staged margins of the three families (FamilyGeo), error-rate rows at q3_off / b3_off.
// An image serves the reads of one template segment and of tiles first_tile .. first_tile + Ti - 1 (Ti = n_tiles: all tiles; Ti = 1: the reads
// are binned by tile and a workgroup stages the image of the bin it serves, fill_binned_loop); it is identified by the index of its first
// quality table, qbase = (segment * n_tiles + first_tile) * 4.  Descriptors in the image have par0_off relative to the image's outcome values.
RSQ_HD uint32_t lds_desc_count(uint32_t n_tiles) { return 25u * n_tiles + 12u; }
RSQ_HD uint32_t image_qbase(const DevSim &S, uint32_t seg, uint32_t first_tile) { return (seg * RSQ_SIM(S, n_tiles) + first_tile) * 4u; }
constexpr uint32_t kDescWords = sizeof(DevTable) / 4u;

#if defined(__HIP_DEVICE_COMPILE__)
#define RSQ_NOINLINE __device__ __noinline__
#else
#define RSQ_NOINLINE inline
#endif
// the host emulation counts what the screen decided (tests/hostemu): [family][0 = draws, 1 = left to double precision]
#if defined(RSQ_SCREEN_STATS) && !defined(__HIP_DEVICE_COMPILE__)
#define RSQ_SCREEN_COUNT(family, decided) (++RSQ_SCREEN_STATS[family][0], RSQ_SCREEN_STATS[family][1] += !(decided))
#else
#define RSQ_SCREEN_COUNT(family, decided) ((void)0)
#endif

// A draw the screen left open, as a call: the double-precision recipe (draw_slim) is rare and large, and inlined at every draw site it costs the read
// kernel's loop registers and a tenth of its time.  The callee reads the descriptor from the image again; the result carries prob_sum == 0 in bit 31.
template <int NM>
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __noinline__
#else
inline
#endif
    uint32_t exact_draw_call(const double *pool, const RSQ_LDS float *img, uint32_t par0_words, uint32_t desc, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3, uint32_t word) {
    const DevTable t = reinterpret_cast<const RSQ_LDS DevTable *>(img)[desc];
    uint32_t idx[NM];
    idx[0] = i0;
    idx[1] = i1;
    idx[2] = i2;
    if constexpr (NM == 4) idx[3] = i3;
    double ps;
    const uint32_t value = draw_slim<NM>(t, pool, reinterpret_cast<const RSQ_LDS uint8_t *>(img + par0_words), idx, u32_to_unit(word), ps);
    return value | (0.0 == ps ? 0x80000000u : 0u);
}

// row of margin n for value v: AdjustIndeces over the family's common range
#define RSQ_GEO_ROW(S, fam, n, v) ((uint32_t)geo_clamp((int32_t)(v) - (int32_t)RSQ_PLAN(S, fam.from[n]), (int32_t)RSQ_PLAN(S, fam.last[n])))
RSQ_HD int32_t geo_clamp(int32_t d, int32_t last) {
#if defined(__clang__)
    return __builtin_elementwise_min(__builtin_elementwise_max(d, 0), last);
#else
    return d < 0 ? 0 : (d > last ? last : d);
#endif
}

// QQ = quads per row of the quality family (LdsPlan::quads_q).  `t` is the step the wave is in: the quality rows over the read
// positions t, ... t-kRingLag are in the wave's ring (lds_ring_load / lds_ring_store).  Rows are found from (table, value) by the families' common
// geometry; a table's descriptor is read only on the double-precision route and for the rare fallbacks of FillReadPart.
template <uint32_t MASK>
struct ScreenTables {
    using Sum = uint32_t;              // 0: prob_sum is 0, else 1 (all the callers ask; a double here costs the loop moves and 64-bit compares)
    static constexpr int QQ = (int)MASK;
    const DevSim &S;
    const RSQ_LDS float *img;          // image of the workgroup
    uint32_t qbase;                    // index of the image's first quality table (image_qbase)
    const RSQ_LDS float *ring_;        // the wave's ring
    uint32_t t;
    uint32_t demand = 0xFFFFFFFFu;     // the read position whose rows the wave staged in the slot behind the ring at this step (a read that lags by more than kRingLag), or none
    RSQ_HD DevTable desc(uint32_t local) const { return reinterpret_cast<const RSQ_LDS DevTable *>(img)[local]; }
    RSQ_HD uint32_t par0_at() const { return RSQ_PLAN(S, desc_words) - RSQ_PLAN(S, par0_words); }
    RSQ_HD DevTable quality(uint32_t i) const { return desc(i - qbase); }
    RSQ_HD DevTable seq_quality(uint32_t i) const { return desc(24u * RSQ_PLAN(S, img_tiles) + 12u + i - qbase / 4u); }
    // the ring's rows of read position p; in_ring: p is one of the last steps' positions (a read lags by its deletions)
    RSQ_HD const RSQ_LDS float *ring(uint32_t p) const { return ring_ + (p % kRingSlots) * RSQ_PLAN(S, ring_stride); }
    RSQ_HD bool in_ring(uint32_t p) const { return t - p <= kRingLag; }
    // a draw the screen left open (or a table outside its preconditions): the reference's recipe in double precision
    template <int NM>
    RSQ_HD uint32_t exact(uint32_t desc, const uint32_t (&idx)[NM], uint32_t word, uint32_t &ps) const {
        const uint32_t r = exact_draw_call<NM>(S.pool, img, par0_at(), desc, idx[0], idx[1], idx[2], NM == 4 ? idx[NM - 1] : 0u, word);
        ps = (r >> 31) ^ 1u;                                  // the callers only ask whether prob_sum is 0
        return r & 0x7FFFFFFFu;
    }
    // `values`: FamilyGeo::values of the family, `column` = table of the image * slot + column
    template <int NM>
    RSQ_HD uint32_t settle(bool decided, uint32_t values, uint32_t column, uint32_t desc, const uint32_t (&idx)[NM], uint32_t u, uint32_t &ps) const {
        uint32_t value = reinterpret_cast<const RSQ_LDS uint8_t *>(img)[values + column];
        ps = 1u;
        decided = decided && !RSQ_SIM(S, force_exact);
        // a divergent branch around a call is skipped by the wave when no lane takes it (s_cbranch_execz): no ballot needed in front of it -- the ballot of a
        // predicate that is a conjunction of compares costs a select, a compare and three scalar instructions per draw
        if (!decided) value = exact<NM>(desc, idx, u, ps);
        return value;
    }

    RSQ_HD uint32_t draw_quality(uint32_t i, const uint32_t (&idx)[4], uint32_t u, uint32_t &ps) const {
        const uint32_t local = i - qbase, slot = RSQ_PLAN(S, slot_q), nr = RSQ_PLAN(S, rate_rows_q), r3 = RSQ_GEO_ROW(S, q, 3, idx[3]);
        const RSQ_LDS float *mine = img + RSQ_PLAN(S, q.lds) + local * RSQ_PLAN(S, q.lds_stride);      // margins 0 and 1 of the table
        const LdsRow32 m0{mine + RSQ_GEO_ROW(S, q, 0, idx[0]) * slot}, m1{mine + (RSQ_PLAN(S, q.before[1]) + RSQ_GEO_ROW(S, q, 1, idx[1])) * slot};
        // the rows over the read position: in the ring, or for a read that lags further in the slot behind it when the wave staged this position there
        const bool near = in_ring(idx[2]);
        const LdsRow32 m2{ring_ + (near ? idx[2] % kRingSlots : kRingSlots) * RSQ_PLAN(S, ring_stride) + local * slot};
        const LdsRow32 m3{img + RSQ_PLAN(S, q3_off) + local * RSQ_PLAN(S, q3_stride) + (r3 < nr ? r3 : 0u) * slot};
        uint32_t col = 0;
        const bool decided = draw_screened<QQ>(u, col, m0, m1, m2, m3) && (near || idx[2] == demand) && r3 < nr;      // a rate or a position whose row is not staged: double precision
        RSQ_SCREEN_COUNT(0, decided);
        return settle<4>(decided, RSQ_PLAN(S, q.values), local * slot + col, local, idx, u, ps);
    }
    RSQ_HD uint32_t draw_base_call(uint32_t i, const uint32_t (&idx)[4], uint32_t u, uint32_t &ps) const {
        const uint32_t local = i - qbase * 5u, slot = RSQ_PLAN(S, slot_b), nr = RSQ_PLAN(S, rate_rows_b), r3 = RSQ_GEO_ROW(S, b, 3, idx[3]);
        const uint32_t g = 4u * (RSQ_PLAN(S, b.off32) + i * (RSQ_PLAN(S, b.table_rows) * slot));               // the table's rows in device memory: bytes from the pool's address (below 4 GB: pack_tables)
        const LdsRow32 m0{img + RSQ_PLAN(S, b.lds) + local * RSQ_PLAN(S, b.lds_stride) + RSQ_GEO_ROW(S, b, 0, idx[0]) * slot};
        const PoolRow32 m1{S.pool32, g + 4u * (RSQ_PLAN(S, b.before[1]) + RSQ_GEO_ROW(S, b, 1, idx[1])) * slot};
        const uint32_t r2 = RSQ_GEO_ROW(S, b, 2, idx[2]);
        const bool m2_staged = RSQ_PLAN(S, b.lds2) != kNoLds, staged = r3 < nr;
        const MixedRow32 m2{LdsRow32{img + (m2_staged ? RSQ_PLAN(S, b.lds2) + local * RSQ_PLAN(S, b.lds2_stride) + r2 * slot : 0u)},
                            PoolRow32{S.pool32, g + 4u * (RSQ_PLAN(S, b.before[2]) + r2) * slot}, m2_staged};
        const MixedRow32 m3{LdsRow32{img + RSQ_PLAN(S, b3_off) + local * RSQ_PLAN(S, b3_stride) + (staged ? r3 : 0u) * slot}, PoolRow32{S.pool32, g + 4u * (RSQ_PLAN(S, b.before[3]) + r3) * slot}, staged};
        uint32_t col = 0;
        const bool decided = draw_screened<(int)kQuadsSmall>(u, col, m0, m1, m2, m3);
        RSQ_SCREEN_COUNT(1, decided);
        return settle<4>(decided, RSQ_PLAN(S, b.values), local * slot + col, 4u * RSQ_PLAN(S, img_tiles) + local, idx, u, ps);
    }
    RSQ_HD uint32_t draw_indel(uint32_t i, const uint32_t (&idx)[3], uint32_t u, uint32_t &ps) const {
        // nearly every draw: the random word alone says "no indel" (DevTable::sure_range, 0 for an empty table; margin 0 at its row 0: the index is not above the
        // margin's first); the wave skips the rows when all its lanes are that sure, and has read two words of the descriptor
        const RSQ_LDS DevTable *d = reinterpret_cast<const RSQ_LDS DevTable *>(img) + (24u * RSQ_PLAN(S, img_tiles) + i);
        const uint32_t range = d->sure_range, lo16 = range & 0xFFFFu;
        const bool sure = (u >> 16) - lo16 < (range >> 16) - lo16 && idx[0] <= d->from[0];
        RSQ_SCREEN_COUNT(3, sure);
        ps = 1u;
        if (sure) return 0;                                 // the lanes that are not sure draw among themselves; a wave without one skips the branch (s_cbranch_execz)
        const uint32_t slot = RSQ_PLAN(S, slot_i), r0 = RSQ_GEO_ROW(S, i, 0, idx[0]);
        const uint32_t g = 4u * (RSQ_PLAN(S, i.off32) + i * (RSQ_PLAN(S, i.table_rows) * slot));
        const bool m0_staged = RSQ_PLAN(S, i.lds) != kNoLds;
        const MixedRow32 m0{LdsRow32{img + (m0_staged ? RSQ_PLAN(S, i.lds) + i * RSQ_PLAN(S, i.lds_stride) + r0 * slot : 0u)}, PoolRow32{S.pool32, g + 4u * r0 * slot}, m0_staged};
        const PoolRow32 m1{S.pool32, g + 4u * (RSQ_PLAN(S, i.before[1]) + RSQ_GEO_ROW(S, i, 1, idx[1])) * slot}, m2{S.pool32, g + 4u * (RSQ_PLAN(S, i.before[2]) + RSQ_GEO_ROW(S, i, 2, idx[2])) * slot};
        uint32_t col = 0;
        const bool decided = draw_screened<(int)kQuadsSmall>(u, col, m0, m1, m2);
        RSQ_SCREEN_COUNT(2, decided);
        return settle<3>(decided, RSQ_PLAN(S, i.values), i * slot + col, 24u * RSQ_PLAN(S, img_tiles) + i, idx, u, ps);
    }
    RSQ_HD uint32_t draw_seq_quality(uint32_t i, const uint32_t (&idx)[3], uint32_t u, uint32_t &ps) const {     // once per read: double precision
        const uint32_t r = exact_draw_call<3>(S.pool, img, par0_at(), 24u * RSQ_PLAN(S, img_tiles) + 12u + i - qbase / 4u, idx[0], idx[1], idx[2], 0u, u);
        ps = (r >> 31) ^ 1u;
        return r & 0x7FFFFFFFu;
    }
};

// Builds the LDS image `qbase` (image_qbase); tid/nthreads describe the calling thread (the host emulation calls it with
// 0/1).  The caller synchronises the workgroup between the two phases and after the second.
RSQ_HD void lds_stage_descriptors(const DevSim &S, RSQ_LDS float *img, uint32_t qbase, uint32_t tid, uint32_t nthreads) {
    const uint32_t T = RSQ_PLAN(S, img_tiles);
    RSQ_LDS uint32_t *dst = reinterpret_cast<RSQ_LDS uint32_t *>(img);
    const uint32_t wq = 4u * T * kDescWords, wb = 20u * T * kDescWords, wi = 12u * kDescWords, ws = T * kDescWords;
    const uint32_t *q = reinterpret_cast<const uint32_t *>(S.quality + qbase), *b = reinterpret_cast<const uint32_t *>(S.base_call + qbase * 5u),
                   *in = reinterpret_cast<const uint32_t *>(S.indels), *sq = reinterpret_cast<const uint32_t *>(S.seq_quality + qbase / 4u);
    // outcome values: the indel tables' are the first bytes of the pool, the image's tiles' a contiguous range from its first quality table's on;
    // par0_off (word 1 of a descriptor) becomes relative to the image's copy
    const uint32_t tiles_at = S.quality[qbase].par0_off, shift = tiles_at - RSQ_PLAN(S, par0_indel_bytes);
    for (uint32_t i = tid; i < wq; i += nthreads) dst[i] = q[i] - (i % kDescWords == 1u ? shift : 0u);
    for (uint32_t i = tid; i < wb; i += nthreads) dst[wq + i] = b[i] - (i % kDescWords == 1u ? shift : 0u);
    for (uint32_t i = tid; i < wi; i += nthreads) dst[wq + wb + i] = in[i];
    for (uint32_t i = tid; i < ws; i += nthreads) dst[wq + wb + wi + i] = sq[i] - (i % kDescWords == 1u ? shift : 0u);
    const uint32_t *p0 = reinterpret_cast<const uint32_t *>(S.par0), *p1 = reinterpret_cast<const uint32_t *>(S.par0 + tiles_at);      // ranges start on words; the pool has spare bytes at its end
    const uint32_t indel_words = RSQ_PLAN(S, par0_indel_bytes) / 4u, par0_at = RSQ_PLAN(S, desc_words) - RSQ_PLAN(S, par0_words);
    for (uint32_t i = tid; i < RSQ_PLAN(S, par0_words); i += nthreads) dst[par0_at + i] = i < indel_words ? p0[i] : p1[i - indel_words];
    // the outcome values by column of the image's tables (FamilyGeo::values; whole words: slots are multiples of four columns)
    const uint32_t nq = T * RSQ_PLAN(S, slot_q), nb = 5u * T * RSQ_PLAN(S, slot_b), ni = 3u * RSQ_PLAN(S, slot_i);
    const uint32_t *vq = reinterpret_cast<const uint32_t *>(S.par0 + RSQ_PLAN(S, q.values_src)) + qbase / 4u * RSQ_PLAN(S, slot_q),
                   *vb = reinterpret_cast<const uint32_t *>(S.par0 + RSQ_PLAN(S, b.values_src)) + qbase / 4u * 5u * RSQ_PLAN(S, slot_b),
                   *vi = reinterpret_cast<const uint32_t *>(S.par0 + RSQ_PLAN(S, i.values_src));
    for (uint32_t i = tid; i < nq; i += nthreads) dst[RSQ_PLAN(S, q.values) / 4u + i] = vq[i];
    for (uint32_t i = tid; i < nb; i += nthreads) dst[RSQ_PLAN(S, b.values) / 4u + i] = vb[i];
    for (uint32_t i = tid; i < ni; i += nthreads) dst[RSQ_PLAN(S, i.values) / 4u + i] = vi[i];
}
// rows [first_row, first_row + n_rows) of `n_tables` tables of a family, from table `first` of the profile on, to [table][n_rows][slot] at dst_off: whole 16-byte groups
RSQ_HD void lds_stage_family_rows(const DevSim &S, RSQ_LDS float *img, uint32_t off32, uint32_t table_rows, uint32_t first, uint32_t n_tables, uint32_t first_row, uint32_t n_rows,
                                  uint32_t slot, uint32_t dst_off, uint32_t dst_stride, uint32_t tid, uint32_t nthreads) {
    const uint32_t per_table = n_rows * (slot / 4u);
    for (uint32_t i = tid; i < n_tables * per_table; i += nthreads) {
        const uint32_t table = i / per_table, g = i - table * per_table;
        reinterpret_cast<RSQ_LDS Quad *>(img + dst_off + table * dst_stride)[g] = reinterpret_cast<const Quad *>(S.pool32 + off32 + ((size_t)(first + table) * table_rows + first_row) * slot)[g];
    }
}
RSQ_HD void lds_stage_rows(const DevSim &S, RSQ_LDS float *img, uint32_t qbase, uint32_t tid, uint32_t nthreads) {
    const uint32_t T = RSQ_PLAN(S, img_tiles), sq = RSQ_PLAN(S, slot_q), sb = RSQ_PLAN(S, slot_b), si = RSQ_PLAN(S, slot_i);
    // quality: margins 0 and 1; base call: margin 0, margin 2; indel: margin 0; then the first rows of the two error-rate margins
    lds_stage_family_rows(S, img, RSQ_PLAN(S, q.off32), RSQ_PLAN(S, q.table_rows), qbase, 4u * T, 0u, RSQ_PLAN(S, q.lds_rows), sq, RSQ_PLAN(S, q.lds), RSQ_PLAN(S, q.lds_stride), tid, nthreads);
    lds_stage_family_rows(S, img, RSQ_PLAN(S, b.off32), RSQ_PLAN(S, b.table_rows), qbase * 5u, 20u * T, 0u, RSQ_PLAN(S, b.lds_rows), sb, RSQ_PLAN(S, b.lds), RSQ_PLAN(S, b.lds_stride), tid, nthreads);
    if (RSQ_PLAN(S, b.lds2) != kNoLds)
        lds_stage_family_rows(S, img, RSQ_PLAN(S, b.off32), RSQ_PLAN(S, b.table_rows), qbase * 5u, 20u * T, RSQ_PLAN(S, b.before[2]), RSQ_PLAN(S, b.last[2]) + 1u, sb, RSQ_PLAN(S, b.lds2),
                              RSQ_PLAN(S, b.lds2_stride), tid, nthreads);
    if (RSQ_PLAN(S, i.lds) != kNoLds)
        lds_stage_family_rows(S, img, RSQ_PLAN(S, i.off32), RSQ_PLAN(S, i.table_rows), 0u, 12u, 0u, RSQ_PLAN(S, i.lds_rows), si, RSQ_PLAN(S, i.lds), RSQ_PLAN(S, i.lds_stride), tid, nthreads);
    lds_stage_family_rows(S, img, RSQ_PLAN(S, q.off32), RSQ_PLAN(S, q.table_rows), qbase, 4u * T, RSQ_PLAN(S, q.before[3]), RSQ_PLAN(S, rate_rows_q), sq, RSQ_PLAN(S, q3_off), RSQ_PLAN(S, q3_stride), tid,
                          nthreads);
    lds_stage_family_rows(S, img, RSQ_PLAN(S, b.off32), RSQ_PLAN(S, b.table_rows), qbase * 5u, 20u * T, RSQ_PLAN(S, b.before[3]), RSQ_PLAN(S, rate_rows_b), sb, RSQ_PLAN(S, b3_off), RSQ_PLAN(S, b3_stride), tid,
                          nthreads);
}
// The ring: the quality rows (margin 2) over read position p of the segment's tables, copied by the wave itself at the beginning of
// step p into slot p % kRingSlots of its ring: one load of 16 bytes per lane instead of one per lane and quad of the row.  Item i is
// one 16-byte group of one table's row.
RSQ_HD uint32_t lds_ring_items(const DevSim &S) { return 4u * RSQ_PLAN(S, img_tiles) * RSQ_PLAN(S, quads_q); }
// What does not change from step to step is worked out once per chunk of reads (RingItem): where the table's rows over the read position begin and the
// item's place in a ring slot (first position and last row of the margin are the family's).
struct RingItem {
    uint32_t rows;                     // row 0 of margin 2, at the item's group of four columns: bytes from the pool's address
    uint32_t at;                       // floats from the slot's start
};
RSQ_HD RingItem lds_ring_item(const DevSim &S, uint32_t qbase, uint32_t item) {
    const uint32_t table = item / RSQ_PLAN(S, quads_q), c = item % RSQ_PLAN(S, quads_q), slot = RSQ_PLAN(S, slot_q);
    return RingItem{4u * (RSQ_PLAN(S, q.off32) + ((qbase + table) * RSQ_PLAN(S, q.table_rows) + RSQ_PLAN(S, q.before[2])) * slot + 4u * c), table * slot + 4u * c};
}
RSQ_HD Quad lds_ring_load(const DevSim &S, const RingItem &it, uint32_t p) { return PoolRow32{S.pool32, it.rows + 4u * RSQ_GEO_ROW(S, q, 2, p) * RSQ_PLAN(S, slot_q)}.quad(0u); }
RSQ_HD void lds_ring_store(const DevSim &S, const RingItem &it, RSQ_LDS float *ring, uint32_t p, const Quad &q) {
    *reinterpret_cast<RSQ_LDS Quad *>(ring + (p % kRingSlots) * RSQ_PLAN(S, ring_stride) + it.at) = q;
}
RSQ_HD void lds_ring_stage(const DevSim &S, const RingItem &it, RSQ_LDS float *ring, uint32_t p) { lds_ring_store(S, it, ring, p, lds_ring_load(S, it, p)); }
// the rows over position p into the slot behind the ring (ScreenTables::demand)
RSQ_HD void lds_ring_stage_demand(const DevSim &S, const RingItem &it, RSQ_LDS float *ring, uint32_t p) {
    *reinterpret_cast<RSQ_LDS Quad *>(ring + kRingSlots * RSQ_PLAN(S, ring_stride) + it.at) = lds_ring_load(S, it, p);
}
RSQ_HD void lds_ring_stage(const DevSim &S, uint32_t qbase, RSQ_LDS float *ring, uint32_t p, uint32_t item) { lds_ring_stage(S, lds_ring_item(S, qbase, item), ring, p); }
// CreateReads for one mate of a fragment (Simulator.cpp:634-721, GetOrgSeq :1916-1922)
// template and systematic errors of mate `seg` of fragment f (GetOrgSeq :1916-1922, CreateReads :680-684)
RSQ_HD FragmentSrc fragment_src(const DevSim &S, const Fragment &f, uint32_t seg, uint32_t end) {
    const uint32_t L = S.seq_len[f.seq];
    const uint32_t want = S.read_lengths[seg].to + RSQ_SIM(S, max_len_deletion);           // Simulator.cpp:1918-1921
    FragmentSrc src;
    src.words = hap_words(S, f.allele);                                        // with variants: the allele's copy (substitutions applied)
    src.word_off = S.seq_word_off[f.seq];
    src.len = f.len < want ? f.len : want;
    src.reverse = seg != f.strand;                                              // block.at(strand) = start_block
    src.first = src.reverse ? end : f.start;
    src.sys_ = src.reverse ? S.sys_rev + S.seq_base_off[f.seq] + (L - end) : S.sys_fwd + S.seq_base_off[f.seq] + f.start;
    src.converted = nullptr;
    src.gc_prefix = hap_gc_prefix(S, f.allele);
    return src;
}
RSQ_HD FragmentSrc fragment_src(const DevSim &S, const Fragment &f, uint32_t seg) { return fragment_src(S, f, seg, f.start + f.len); }

// The template of a mate with variants: FragmentSrc (on the allele's copy of the reference when all variants are substitutions,
// else with the template written beforehand by k_variant_templates), and the systematic errors through the walk of
// GetSysErrorFromBlock / IncrementBlockPos (Simulator.cpp:232-292) and of FillReadPart's deletion branch (:380-392), stated in strand
// coordinates (position on the strand the mate reads; variants in that strand's order): the reverse blocks' lists are the mirror
// image of the forward ones.  cur walks the variants of ALL alleles; a block's err_variants_ list ends where the block ends, and
// cur_var = 0 after a block change is the first variant of the new block.  As written in the reference: after a substitution is
// used cur is incremented twice (the next variant is skipped unless a block starts in between); inside an insertion the error of the
// reference position is returned, not the inserted base's; the deletion branch does not look at variants (a passed variant is
// applied late).
struct VariantSrc : FragmentSrc {
    const DevVariant *var;              // the sequence's variants in forward order
    const uint16_t *err;                // the variants' systematic errors on the strand the mate reads
    uint32_t n_var, L, allele;
    uint32_t *walk_error;               // DevSim::walk_error
    uint32_t spos0, cur0, var_pos0;     // start of the walk: strand position of the first template base, variant index, position in an insertion
    mutable uint32_t spos, cur, var_pos;
    RSQ_HD const DevVariant &var_at(uint32_t i) const { return reverse ? var[n_var - 1u - i] : var[i]; }
    RSQ_HD uint32_t var_spos(uint32_t i) const { return reverse ? L - 1u - var_at(i).pos : var_at(i).pos; }
    RSQ_HD uint32_t var_err(uint32_t i, uint32_t k) const { return err[var_at(i).off + k]; }
    RSQ_HD uint32_t lower_bound(uint32_t sp) const {
        uint32_t lo = 0, hi = n_var;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (var_spos(mid) < sp) lo = mid + 1u;
            else hi = mid;
        }
        return lo;
    }
    // lower_bound(sp) from where the walk stands: a block change asks for the first variant at or behind the new block's first position, and that is `cur` itself or
    // a neighbour (a variant skipped behind a substitution) -- one or two loads in the place of a binary search over the sequence's variants (17 dependent loads
    // for 100 000 variants, at every thousandth step of every lane)
    RSQ_HD uint32_t seek(uint32_t sp) const {
        uint32_t c = cur < n_var ? cur : n_var;
        while (c > 0u && var_spos(c - 1u) >= sp) --c;
        while (c < n_var && var_spos(c) < sp) ++c;
        return c;
    }
    // blocks are cut on the forward strand (Simulator.h:254): first strand position of the block after the one holding sp
    RSQ_HD uint32_t block_end(uint32_t sp) const { return reverse ? L - ((L - sp - 1u) / kBlockSize) * kBlockSize : (sp / kBlockSize + 1u) * kBlockSize; }
    // what the common step -- a plain reference base, no variant in reach -- needs, kept between steps: the strand position of variant `cur`
    // (none: 0xFFFFFFFF) and the end of the block holding spos; refreshed whenever cur or the block changes
    mutable uint32_t vs_cur, bend_cur;
    // ... and, for the step itself, ONE number: the strand position before which nothing of the walk can happen -- no variant of the block at or before the
    // position, not the block's last position (the step behind it changes the block), not inside an insertion, not beyond the strand.  A step in front of it
    // is the plain track's (one compare); everything else goes the long way below, which keeps its own state (cur, var_pos, the block's end) wherever the
    // compiler finds room for it -- these are read once in a few hundred steps.
    mutable uint32_t quiet_until;
    RSQ_HD void refresh() const {
        vs_cur = cur < n_var ? var_spos(cur) : 0xFFFFFFFFu;
        bend_cur = block_end(spos);
        uint32_t q = bend_cur - 1u;
        if (vs_cur < q) q = vs_cur;
        if (L < q) q = L;
        quiet_until = var_pos ? 0u : q;
    }
    RSQ_HD void rewind() const {
        spos = spos0;
        cur = cur0;
        var_pos = var_pos0;
        refresh();
    }
    RSQ_HD void increment_block_pos() const {                                   // :232-238
        if (++spos == bend_cur) cur = seek(spos);
        refresh();
    }
    // With variants a few bases apart the walk (its skipped variants, its late ones) can use up more reference positions than the
    // template has and run off the end of the strand: the reference follows a NULL next_block_ there.  Reported, not simulated.
    RSQ_HD bool off_strand() const {
        if (spos < L) return false;
        *walk_error = 1u;
        return true;
    }
    RSQ_HD uint32_t sys_base(uint32_t) const {                                  // :240-292
#if defined(RSQ_NO_QUIET)                                                        // measurements: the walk without its short cut
        if (false) {
#else
        if (__builtin_expect(spos < quiet_until, 1)) {                          // nearly every step
#endif
            const uint32_t se = FragmentSrc::sys_base(spos - spos0);
            ++spos;
            return se;
        }
        return sys_base_event();
    }
    RSQ_HD uint32_t sys_base_event() const {
        if (off_strand()) return 0;
        if (!var_pos && !(vs_cur < bend_cur && vs_cur <= spos)) {               // no variant of the block at or before this position: the block's last position
            const uint32_t se = FragmentSrc::sys_base(spos - spos0);
            if (++spos == bend_cur) {
                cur = seek(spos);
                refresh();
            }
            return se;
        }
        if (var_pos) {
            const uint32_t se = FragmentSrc::sys_base(spos - spos0);
            if (++var_pos >= var_at(cur).len) {
                var_pos = 0;
                ++cur;
                increment_block_pos();
            }
            return se;
        }
        uint32_t bend = bend_cur;
        while (cur < n_var) {
            const uint32_t vs = var_spos(cur);
            if (!(vs < bend && vs <= spos)) break;                              // cur_var < err_variants_.size() && position_ <= block_pos
            const DevVariant &v = var_at(cur);
            if ((v.allele[allele >> 6] >> (allele & 63u)) & 1u) {
                if (0u == v.len) {                                              // deletion
                    ++cur;
                    increment_block_pos();
                    if (off_strand()) return 0;
                    bend = bend_cur;
                } else {
                    const uint32_t se = var_err(cur, 0);
                    if (1u == v.len) {                                          // substitution
                        ++cur;
                        increment_block_pos();
                        ++cur;
                        refresh();
                    } else {
                        var_pos = 1;                                            // insertion
                        refresh();
                    }
                    return se;
                }
            } else ++cur;
        }
        const uint32_t se = FragmentSrc::sys_base(spos - spos0);
        increment_block_pos();
        return se;
    }
    RSQ_HD uint32_t sys_deleted(uint32_t) const {                               // :380-392
        if (off_strand()) return 0;
        const uint32_t se = FragmentSrc::sys_base(spos - spos0);
        if (var_pos && ++var_pos >= var_at(cur).len) var_pos = 0;
        if (0u == var_pos) {
            if (++spos == bend_cur) cur = seek(spos);
        }
        refresh();
        return se;
    }
    RSQ_HD void totals(uint32_t n, uint32_t &gc, uint32_t &rate_sum) const {    // :480-504: the error rates through a copy of the walk
        if (converted) gc += ref_gc_count(converted, 0, 0, n);
        else gc += reverse ? ref_gc_count_prefix(words, gc_prefix, word_off, first - n, first) : ref_gc_count_prefix(words, gc_prefix, word_off, first, first + n);
        // no variant in reach of these n steps (the walk starts at the first variant at or behind the first base, and that one lies behind the
        // last): the walk returns the strand's own errors, a block change finds the same variant again
        if (!var_pos0 && vs_cur >= spos0 + n && (0u == cur0 || var_spos(cur0 - 1u) < spos0)) {
            rate_sum += rate_total(n);
            return;
        }
        for (uint32_t k = 0; k < n; ++k) rate_sum += sys_base(k) >> 8;
        rewind();
    }
};
// fv == nullptr: substitutions only (the walk starts at the first variant at or after the first template base)
RSQ_HD VariantSrc variant_src(const DevSim &S, const Fragment &f, const FragmentVar *fv, uint32_t seg) {
    VariantSrc src;
    static_cast<FragmentSrc &>(src) = fragment_src(S, f, seg, fv ? fv->end : f.start + f.len);
    src.var = S.variants + S.var_ptr[f.seq];
    src.err = src.reverse ? S.var_err_rev : S.var_err_fwd;
    src.n_var = S.var_ptr[f.seq + 1] - S.var_ptr[f.seq];
    src.L = S.seq_len[f.seq];
    src.allele = f.allele;
    src.walk_error = S.walk_error;
    src.spos0 = src.reverse ? src.L - src.first : src.first;
    if (!fv) {
        src.cur0 = src.lower_bound(src.spos0);
        src.var_pos0 = 0;
    } else if (!src.reverse) {                                                  // CreateReads :686-688: variant.at(strand) = start variant
        src.cur0 = (uint32_t)fv->start_var;
        src.var_pos0 = fv->start_var_pos;
    } else {                                                                    // the end variant, seen from the reverse block's list
        src.cur0 = src.n_var - 1u - (uint32_t)fv->end_var;                      // end_var -1: one past the last
        src.var_pos0 = fv->end_var_pos ? src.var[fv->end_var].len - fv->end_var_pos : 0u;
    }
    src.rewind();
    return src;
}
// The converted template of mate `seg` of fragment f (CTConversion's dispatcher, Simulator.cpp:2219-2247): the forward mate is
// converted from the start position on, the reverse mate from the end position on (`reversed`).
RSQ_HD void convert_template(const DevSim &S, const Fragment &f, uint32_t seg, uint64_t *tmpl, uint32_t template_words) {
    const FragmentSrc src = fragment_src(S, f, seg);
    for (uint32_t w = 0; w < template_words; ++w) tmpl[w] = 0;
    for (uint32_t k = 0; k < src.len; ++k) tmpl[k >> 5] |= (uint64_t)src.ref(k) << ((k & 31u) * 2u);
    const MethView m = meth_view(S, f.seq);
    MethDraws d{S.seed, f.start, f.seq, f.len, (kDomMethylation << 28) | ((src.reverse ? 1u : 0u) << 27), 0xFFFFFFFFu, Words{0, 0, 0, 0}};
    ct_conversion(tmpl, src.len, m, nullptr, 0u, src.first, src.reverse, VarStart{0, 0u}, d);
}

// the template of mate `seg` with variants of any kind: the forward mate from the start variant, the reverse mate from the end variant
RSQ_HD void variant_template(const DevSim &S, const Fragment &f, const FragmentVar &fv, uint32_t seg, uint64_t *tmpl, uint32_t template_words) {
    const uint32_t want = S.read_lengths[seg].to + RSQ_SIM(S, max_len_deletion), tl = f.len < want ? f.len : want;
    const VarView r = var_view(S, f.seq);
    const bool reversed = seg != f.strand;
    const VarStart from = reversed ? VarStart{fv.end_var, fv.end_var_pos} : VarStart{fv.start_var, fv.start_var_pos};
    const uint32_t at = reversed ? fv.end : f.start;
    allele_template(allele_view(S, f.seq, f.allele), at, from, tl, reversed, tmpl, template_words);
    if (S.meth_ptr) {                                                           // CTConversion with variants (:2232-2237)
        const MethView m = meth_view(S, f.seq, f.allele);
        MethDraws d{S.seed, f.start, f.seq | (fv.sub << 22), f.len, (kDomMethylation << 28) | ((reversed ? 1u : 0u) << 27) | ((uint32_t)f.allele << 17), 0xFFFFFFFFu,
                    Words{0, 0, 0, 0}};
        ct_conversion(tmpl, tl, m, &r, f.allele, at, reversed, from, d);
    }
}

template <class Tab>
RSQ_HD void fill_fragment_read(const DevSim &S, const Tab &tab, const Fragment &f, uint32_t seg, ReadOut &out, ReadMeta &meta) {
    const uint32_t c2 = f.len | ((uint32_t)f.dup << 16);
    const uint32_t tile = draw_tile(S, f.start, f.seq, c2, pair_c3(kDomPair, f.strand, 2, f.allele));
    const Stream st{S.seed, f.start, f.seq, c2, pair_c3(kDomPair, f.strand, seg, f.allele)};
    if (S.variants_loaded) fill_read(S, tab, st, seg, tile, f.len, variant_src(S, f, nullptr, seg), out, meta);
    else fill_read(S, tab, st, seg, tile, f.len, fragment_src(S, f, seg), out, meta);
}
// one mate of adapter-only pair i (Simulator.cpp:2359-2382)
template <class Tab>
RSQ_HD void fill_adapter_only_read(const DevSim &S, const Tab &tab, uint64_t i, uint32_t seg, ReadOut &out, ReadMeta &meta) {
    const uint32_t tile = draw_tile(S, (uint32_t)i, 0xFFFFFFFFu, (uint32_t)(i >> 32), pair_c3(kDomPair, 0, 2));
    const Stream st{S.seed, (uint32_t)i, 0xFFFFFFFFu, (uint32_t)(i >> 32), pair_c3(kDomPair, 0, seg)};
    fill_read(S, tab, st, seg, tile, 0u, EmptySrc{}, out, meta);
}

// seqToIllumina records (Simulator.cpp:2403-2512): templates and systematic errors come from byte arrays, a record per lane -- 64 cache lines per
// load instruction.  The source therefore holds the 8-byte group (k >> 3) of the three arrays it last read: three loads per eight bases instead of per
// base, and the totals of FillRead's start (G/C count, rate sum) come from the same groups.  `safe`: bytes from the record's first byte to the end of
// the arrays; a group reaching beyond it is read byte by byte.
struct RecordSrc {
    const uint8_t *seq;
    const uint8_t *dom, *rate;
    uint32_t len;
    uint32_t safe;
    mutable uint32_t group;
    mutable uint64_t w_seq, w_dom, w_rate;
    RSQ_HD static uint64_t load8(const uint8_t *p, uint32_t off, uint32_t safe) {
        uint64_t w = 0;
        if (off + 8u <= safe) {
#if defined(__HIP_DEVICE_COMPILE__)
            w = *reinterpret_cast<const uint64_t __attribute__((aligned(1))) *>(p + off);      // unaligned 8-byte loads are what the hardware does (amdhsa)
#else
            memcpy(&w, p + off, 8);
#endif
        } else {
            for (uint32_t j = 0; j < 8u && off + j < safe; ++j) w |= (uint64_t)p[off + j] << (8u * j);
        }
        return w;
    }
    RSQ_HD void hold(uint32_t k) const {
        const uint32_t g = k >> 3;
        if (g == group) return;
        group = g;
        w_seq = load8(seq, g * 8u, safe);
        w_dom = load8(dom, g * 8u, safe);
        w_rate = load8(rate, g * 8u, safe);
    }
    RSQ_HD uint32_t org_len() const { return len; }
    RSQ_HD uint32_t base(uint32_t k) const {
        hold(k);
        return (uint32_t)(w_seq >> ((k & 7u) * 8u)) & 0xFFu;
    }
    RSQ_HD uint32_t sys_base(uint32_t k) const {
        hold(k);
        const uint32_t sh = (k & 7u) * 8u;
        return ((uint32_t)(w_dom >> sh) & 0xFFu) | (((uint32_t)(w_rate >> sh) & 0xFFu) << 8);
    }
    RSQ_HD uint32_t sys_deleted(uint32_t k) const { return sys_base(k); }
    // Simulator.cpp:482-489 over the groups, from the last one down (the first stays held): a base is G/C iff it is 1 or 2, the rates add up bytewise
    RSQ_HD void totals(uint32_t n, uint32_t &gc, uint32_t &rate_sum) const {
        const uint64_t kOnes = 0x0101010101010101ull, kEven = 0x00FF00FF00FF00FFull;
        for (uint32_t g = (n + 7u) >> 3; g--;) {
            hold(g * 8u);
            const uint32_t left = n - g * 8u;                                 // bases of this group that count
            const uint64_t keep = left >= 8u ? ~0ull : (1ull << (8u * left)) - 1ull;
            const uint64_t w = w_seq & keep, r = w_rate & keep;
            const uint64_t high = (w >> 2) | (w >> 3) | (w >> 4) | (w >> 5) | (w >> 6) | (w >> 7);
            const uint64_t is = (w ^ (w >> 1)) & ~high & kOnes;
            gc += (uint32_t)((is * kOnes) >> 56);
            const uint64_t pairs = (r & kEven) + ((r >> 8) & kEven);
            rate_sum += (uint32_t)((pairs * 0x0001000100010001ull) >> 48);
        }
    }
};
// record i of n: the arrays hold read_len bytes per record
RSQ_HD RecordSrc record_src(const uint8_t *seqs, const uint8_t *dom, const uint8_t *rate, uint32_t read_len, uint64_t i, uint64_t n) {
    const uint64_t rest = (n - i) * read_len;
    return RecordSrc{seqs + i * read_len, dom + i * read_len, rate + i * read_len, read_len, rest > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)rest, 0xFFFFFFFFu, 0, 0, 0};
}
// a record of `len` bytes at offset `at` of arrays of `bytes` bytes (records parsed on the device lie where their text lay, rsq_fasta.h)
RSQ_HD RecordSrc record_src_at(const uint8_t *seqs, const uint8_t *dom, const uint8_t *rate, uint32_t at, uint32_t len, uint32_t bytes) {
    return RecordSrc{seqs + at, dom + at, rate + at, len, bytes - at, 0xFFFFFFFFu, 0, 0, 0};
}
// Workgroup sizes of the read kernels: 1024 threads -- four waves per SIMD at 128 VGPRs.  The step is a chain of dependent draws that leaves VALU and LDS idle a
// fifth of the time at three waves per SIMD; a fourth wave fills more of it than the spills cost that 128 registers bring (the profile's own kernel for read pairs:
// 11 spilled VGPRs, 80 B of scratch; with variants 67 and 208 B).  Measured against 768 threads, 157-168 VGPRs and no spill (profiles/r04_zz_block_*): read pairs
// 47.2 -> 42.5 ms per 10 M pairs, seqToIllumina records 22.9 -> 22.1 ms per 8 M, configs[4] at 1/10 scale 107.5 -> 112.8 M pairs/s -- with the screen's loads
// issued a quad at a time (RSQ_SCREEN_BATCH 1; two at a time the walking kernels lose 3-7 % at 1024 threads).  RSQ_FILL_BLOCK_WALK: the kernels whose source walks
// per-lane state (variants, records), should a build want them smaller.
// The same for records parsed on the device (rsq_fasta.h Packed): a half-word per base -- base code in bits 0-1, dominant error in bits 2-4, error percent in
// bits 8-15 -- so the lane's eight bases of all three come with ONE 16-byte load (RecordSrc: three 8-byte loads, each touching a cache line of the lane's own).
struct PackedRecordSrc {
    const uint16_t *codes;
    uint32_t len;
    uint32_t safe;                   // half-words from the record's first one to the end of the array
    mutable uint32_t group;
    mutable uint64_t lo, hi;         // bases 0-3 and 4-7 of the held group
    RSQ_HD void hold(uint32_t k) const {
        const uint32_t g = k >> 3;
        if (g == group) return;
        group = g;
        const uint16_t *p = codes + 8u * g;
        if (8u * g + 8u <= safe) {
#if defined(__HIP_DEVICE_COMPILE__)
            lo = *reinterpret_cast<const uint64_t __attribute__((aligned(2))) *>(p);
            hi = *reinterpret_cast<const uint64_t __attribute__((aligned(2))) *>(p + 4);
#else
            memcpy(&lo, p, 8);
            memcpy(&hi, p + 4, 8);
#endif
        } else {
            lo = hi = 0;
            for (uint32_t j = 0; j < 8u && 8u * g + j < safe; ++j) (j < 4u ? lo : hi) |= (uint64_t)p[j] << (16u * (j & 3u));
        }
    }
    RSQ_HD uint32_t half(uint32_t k) const {
        hold(k);
        return (uint32_t)(((k & 4u) ? hi : lo) >> (16u * (k & 3u))) & 0xFFFFu;
    }
    RSQ_HD uint32_t org_len() const { return len; }
    RSQ_HD uint32_t base(uint32_t k) const { return half(k) & 3u; }
    RSQ_HD uint32_t sys_base(uint32_t k) const {
        const uint32_t h = half(k);
        return ((h >> 2) & 7u) | (h & 0xFF00u);
    }
    RSQ_HD uint32_t sys_deleted(uint32_t k) const { return sys_base(k); }
    // Simulator.cpp:482-489 over the groups, from the last one down (the first stays held): a base is G/C iff its two bits differ, the rates are the high bytes
    RSQ_HD void totals(uint32_t n, uint32_t &gc, uint32_t &rate_sum) const {
        const uint64_t kHalfOnes = 0x0001000100010001ull;
        for (uint32_t g = (n + 7u) >> 3; g--;) {
            hold(g * 8u);
            const uint32_t left = n - g * 8u;                                 // bases of this group that count
            for (uint32_t part = 0; part < 2u; ++part) {
                const uint32_t mine = left > 4u * part ? (left - 4u * part < 4u ? left - 4u * part : 4u) : 0u;
                if (!mine) continue;
                const uint64_t keep = mine >= 4u ? ~0ull : (1ull << (16u * mine)) - 1ull, w = (part ? hi : lo) & keep;
#if defined(__HIP_DEVICE_COMPILE__)
                gc += (uint32_t)__popcll((w ^ (w >> 1)) & kHalfOnes);
#else
                gc += (uint32_t)__builtin_popcountll((w ^ (w >> 1)) & kHalfOnes);
#endif
                rate_sum += (uint32_t)((((w >> 8) & 0x00FF00FF00FF00FFull) * kHalfOnes) >> 48);
            }
        }
    }
};
RSQ_HD PackedRecordSrc packed_record_src(const uint16_t *codes, uint32_t at, uint32_t len, uint32_t halfwords) {
    return PackedRecordSrc{codes + at, len, halfwords - at, 0xFFFFFFFFu, 0, 0};
}
#ifndef RSQ_FILL_BLOCK
#define RSQ_FILL_BLOCK 1024
#endif
#ifndef RSQ_FILL_BLOCK_WALK
#define RSQ_FILL_BLOCK_WALK 1024
#endif
constexpr uint32_t kFillBlock = RSQ_FILL_BLOCK, kFillBlockWalk = RSQ_FILL_BLOCK_WALK;
constexpr uint32_t kFillWavesMax = (kFillBlock > kFillBlockWalk ? kFillBlock : kFillBlockWalk) / 64u;      // the LDS image has a ring for every wave of the larger one
RSQ_HD constexpr uint32_t fill_block(bool walk) { return walk ? kFillBlockWalk : kFillBlock; }

// Reads binned by tile (LdsPlan::binned): bin = segment * n_tiles + tile.  `perm` lists the items (pairs of a batch: both segments share the list of a
// tile; seqToIllumina records: a record has one segment) bin after bin.  A workgroup joins a bin, stages its image and its waves pull the bin's chunks
// of 64 items from the bin's counter -- like the plain kernel's waves, without meeting each other -- until the bin is used up; only then does the
// workgroup synchronise, choose the bin with the most chunks left per workgroup already on it, and stage again.
struct FillBins {
    const uint32_t *perm;           // items sorted by bin
    const uint32_t *bin_first;      // [n_bins] first entry of the bin in perm
    const uint32_t *bin_count;      // [n_bins]
    const uint32_t *chunk_ptr;      // [n_bins + 1] chunks of the bins in front
    uint32_t *next_chunk;           // [n_bins] the bin's chunks handed out so far
    uint32_t *workers;              // [n_bins] workgroups on the bin
    uint32_t n_bins;
    const Fragment *frags;          // pairs: the fragments (and what the sieve found of their variants) in perm's order, so that a wave reads them in one piece and
    const FragmentVar *fvars;       // the pair index is needed only before and after a chunk's reads
};
constexpr uint32_t kSchedWords = 8;              // LDS words behind the image in which fill_binned_loop keeps the bin its workgroup is on
#ifndef RSQ_BIN_KEYS_LDS
#define RSQ_BIN_KEYS_LDS 4096
#endif
constexpr uint32_t kBinKeysLds = RSQ_BIN_KEYS_LDS;      // up to so many bin keys the counting kernels aggregate in LDS (a build with 2 runs the tile tests through the other branch)
constexpr uint32_t kBinItemsPerThread = 16, kBinBlock = 256;

// the stream of one mate of a pair (CreateReads :634-721 / SimulateAdapterOnlyPairs :2359-2382): what k_fill_reads and the tile binning agree on
struct PairStream {
    uint32_t c0, c1, c2, strand;
};
RSQ_HD PairStream pair_stream(const Fragment *f, uint32_t sub, uint64_t adapter_only_number) {
    if (f) return PairStream{f->start, f->seq | (sub << 22), f->len | ((uint32_t)f->dup << 16), f->strand};
    return PairStream{(uint32_t)adapter_only_number, 0xFFFFFFFFu, (uint32_t)(adapter_only_number >> 32), 0u};
}

#if RSQ_DEVICE_BUILD
#if !defined(RSQ_SPEC)
// bin keys of the items + their histogram.  Pairs: key = tile (TileId() once per pair, Simulator.cpp:701-704); records: key = segment * n_tiles + tile.
__device__ inline void bin_count_key(uint32_t key, bool valid, uint32_t n_keys, uint32_t *hist, uint32_t *s_hist) {
    if (n_keys <= kBinKeysLds) {
        for (uint32_t k = threadIdx.x; k < n_keys; k += blockDim.x) s_hist[k] = 0;
        __syncthreads();
        if (valid) atomicAdd(&s_hist[key], 1u);
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < n_keys; k += blockDim.x)
            if (s_hist[k]) atomicAdd(&hist[k], s_hist[k]);
    } else if (valid) atomicAdd(&hist[key], 1u);
}
__global__ void __launch_bounds__(kBinBlock) k_pair_tiles(DevSim S, const Fragment *frags, const FragmentVar *fvars, uint64_t n_pairs, uint64_t adapter_only_first, uint16_t *key_of,
                                                         uint32_t *hist) {
    __shared__ uint32_t s_hist[kBinKeysLds];
    const uint64_t pair = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = pair < n_pairs;
    uint32_t key = 0;
    if (valid) {
        Fragment f{};
        if (frags) f = frags[pair];
        const PairStream ps = pair_stream(frags ? &f : nullptr, fvars ? fvars[pair].sub : 0u, adapter_only_first + pair);
        key = draw_tile(S, ps.c0, ps.c1, ps.c2, pair_c3(kDomPair, ps.strand, 2, f.allele));
        key_of[pair] = (uint16_t)key;
    }
    bin_count_key(key, valid, S.n_tiles, hist, s_hist);
}
__global__ void __launch_bounds__(kBinBlock) k_record_tiles(DevSim S, const uint8_t *segs, uint64_t first_index, uint64_t n, uint16_t *key_of, uint32_t *hist) {
    __shared__ uint32_t s_hist[kBinKeysLds];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    uint32_t key = 0;
    if (valid) {
        const uint64_t idx = first_index + i;
        key = (segs[i] ? S.n_tiles : 0u) + draw_tile(S, (uint32_t)idx, (uint32_t)(idx >> 32), 0u, pair_c3(kDomErrModel, 0, 2));
        key_of[i] = (uint16_t)key;
    }
    bin_count_key(key, valid, 2u * S.n_tiles, hist, s_hist);
}
// one workgroup: the bins' places in perm (exclusive scan of the histogram), the scatter's cursors, the scheduler's counters.  pairs: n_keys = n_tiles, bins
// (segment, tile) of both segments share tile's entries; records: n_keys = 2 n_tiles = the bins.
__global__ void __launch_bounds__(1024) k_bins_plan(const uint32_t *hist, uint32_t n_keys, uint32_t n_bins, uint32_t *bin_first, uint32_t *bin_count, uint32_t *cursor,
                                                    uint32_t *chunk_ptr, uint32_t *next_chunk, uint32_t *workers) {
    __shared__ uint32_t s_part[1024], s_chunks[1024];
    const uint32_t t = threadIdx.x, per = (n_bins + 1023u) / 1024u, lo = t * per < n_bins ? t * per : n_bins, hi = lo + per < n_bins ? lo + per : n_bins;
    // bins lo .. hi-1 of this thread; bin b has the items of key b % n_keys (the first n_keys bins place them)
    uint32_t items = 0, chunks = 0;
    for (uint32_t b = lo; b < hi; ++b) {
        const uint32_t c = hist[b % n_keys];
        if (b < n_keys) items += c;
        chunks += (c + 63u) / 64u;
    }
    s_part[t] = items;
    s_chunks[t] = chunks;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {                       // inclusive scans over the threads
        const uint32_t a = t >= d ? s_part[t - d] : 0u, c = t >= d ? s_chunks[t - d] : 0u;
        __syncthreads();
        s_part[t] += a;
        s_chunks[t] += c;
        __syncthreads();
    }
    uint32_t at = s_part[t] - items, chunk_at = s_chunks[t] - chunks;
    for (uint32_t b = lo; b < hi; ++b) {
        const uint32_t c = hist[b % n_keys];
        if (b < n_keys) {
            cursor[b] = at;
            for (uint32_t r = b; r < n_bins; r += n_keys) bin_first[r] = at, bin_count[r] = c;
            at += c;
        }
        chunk_ptr[b] = chunk_at;
        chunk_at += (c + 63u) / 64u;
        next_chunk[b] = workers[b] = 0;
    }
    if (t == 1023u) chunk_ptr[n_bins] = s_chunks[1023];
}
// items to their bins' places: ranks inside the workgroup from LDS counters, one global reservation per workgroup and key
__global__ void __launch_bounds__(kBinBlock) k_bin_scatter(const uint16_t *key_of, uint64_t n, uint32_t n_keys, uint32_t *cursor, uint32_t *perm, const Fragment *frags,
                                                          const FragmentVar *fvars, Fragment *frags_sorted, FragmentVar *fvars_sorted) {
    __shared__ uint32_t s_count[kBinKeysLds], s_base[kBinKeysLds];
    const uint64_t first = (uint64_t)blockIdx.x * (kBinBlock * kBinItemsPerThread);
    auto place = [&](uint32_t at, uint64_t i) {
        perm[at] = (uint32_t)i;
        if (frags) frags_sorted[at] = frags[i];
        if (fvars) fvars_sorted[at] = fvars[i];
    };
    if (n_keys > kBinKeysLds) {
        for (uint32_t j = 0; j < kBinItemsPerThread; ++j) {
            const uint64_t i = first + (uint64_t)j * kBinBlock + threadIdx.x;
            if (i < n) place(atomicAdd(&cursor[key_of[i]], 1u), i);
        }
        return;
    }
    for (uint32_t k = threadIdx.x; k < n_keys; k += kBinBlock) s_count[k] = 0;
    __syncthreads();
    uint32_t rank[kBinItemsPerThread], key[kBinItemsPerThread];
#pragma unroll
    for (uint32_t j = 0; j < kBinItemsPerThread; ++j) {
        const uint64_t i = first + (uint64_t)j * kBinBlock + threadIdx.x;
        key[j] = i < n ? key_of[i] : 0xFFFFFFFFu;
        rank[j] = i < n ? atomicAdd(&s_count[key[j]], 1u) : 0u;
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_keys; k += kBinBlock)
        if (s_count[k]) s_base[k] = atomicAdd(&cursor[k], s_count[k]);
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < kBinItemsPerThread; ++j)
        if (key[j] != 0xFFFFFFFFu) place(s_base[key[j]] + rank[j], first + (uint64_t)j * kBinBlock + threadIdx.x);
}

#endif
// One lane per read, persistent waves.  A workgroup serves one LDS image at a time -- a template segment (blockIdx.x & 1) with all tiles, built once, every
// wave pulling chunks of 64 pairs from the segment's counter until the batch is exhausted (no tail); or, BINNED, the (segment, tile) of the work unit
// it took (fill_binned_loop).  All lanes of a wave walk their reads' state machines in one uniform loop.  MASK = quads per quality row of the
// screened draws (0: every table access goes to HBM in double precision).
// the LDS image `qbase` of the workgroup (all waves call it; returns after the final barrier)
template <uint32_t MASK>
__device__ RSQ_LDS float *fill_stage_image(const DevSim &S, float *lds_image, uint32_t qbase) {
    RSQ_LDS float *img = (RSQ_LDS float *)lds_image;
    if (MASK) {
        lds_stage_descriptors(S, img, qbase, threadIdx.x, blockDim.x);
        __syncthreads();
        lds_stage_rows(S, img, qbase, threadIdx.x, blockDim.x);
        __syncthreads();
    }
    return img;
}
// 64 reads of one wave through the state machine: one uniform step loop.  Screened (MASK != 0): at the beginning of a step the wave
// copies the quality rows over the step's read position into its ring.  `tile` (index among the profile's tiles) is the lane's own.
template <uint32_t MASK, class Src>
__device__ void fill_wave_reads(const DevSim &S, RSQ_LDS float *img, uint32_t qbase, uint32_t seg, bool active, const Stream &st, uint32_t tile, uint32_t fragment_length,
                                const Src &src, ReadOut &out, ReadMeta &meta) {
    ReadMachine m;
    if constexpr (MASK == 0) {
        const GlobalTables tab{S};
        if (active) {
            m.init(S, tab, st, seg, tile, fragment_length, src);
            while (m.step(S, tab, st, src, out)) {}
        }
    } else {
        const uint32_t lane = threadIdx.x & 63u, n_items = lds_ring_items(S);
        RSQ_LDS float *ring = img + RSQ_PLAN(S, ring_off) + (threadIdx.x >> 6) * kRingRows * RSQ_PLAN(S, ring_stride);
        ScreenTables<MASK> tab{S, img, qbase, ring, 0u};
        m.idle();
        if (active) m.init(S, tab, st, seg, tile, fragment_length, src);
        const RingItem mine = lds_ring_item(S, qbase, lane < n_items ? lane : 0u);      // the lane's first item (with one tile per image: its only one)
        // the lane's item of the NEXT step is loaded while this step runs (the row comes from L2: its latency would stand at the head of every step)
        Quad ahead = lane < n_items ? lds_ring_load(S, mine, 0u) : zero_quad();
        // What the wave does at the beginning of a step: its ring slot of the step (and the prefetch of the next), and the slot behind the ring for a read that has
        // lost more than kRingLag steps to deletions -- such a read no longer finds the rows over its position in the ring, and left to the double-precision call at
        // every step it would double the time of its wave's remaining steps (one read in 1600 with profile P0; the whole launch waits for such a wave when the call
        // is small).  The wave stages the rows over the first such read's position; another one at another position is rarer still.
        auto begin_step = [&](uint32_t t) {
            if (lane < n_items) {
                lds_ring_store(S, mine, ring, t, ahead);
                ahead = lds_ring_load(S, mine, t + 1u);
            }
            for (uint32_t item = lane + 64u; item < n_items; item += 64u) lds_ring_stage(S, qbase, ring, t, item);
            __builtin_amdgcn_wave_barrier();                 // the wave's LDS writes precede its reads (in order in hardware; this orders the compiler)
            tab.t = t;
            const uint64_t lagging = __builtin_amdgcn_ballot_w64(m.phase != ReadMachine::kDone && t - m.par.read_pos > kRingLag);
            tab.demand = 0xFFFFFFFFu;
            if (lagging) {
                const uint32_t p = __builtin_amdgcn_readlane(m.par.read_pos, __builtin_ctzll(lagging));
                if (lane < n_items) lds_ring_stage_demand(S, mine, ring, p);
                for (uint32_t item = lane + 64u; item < n_items; item += 64u) lds_ring_stage_demand(S, lds_ring_item(S, qbase, item), ring, p);
                __builtin_amdgcn_wave_barrier();
                tab.demand = p;
            }
        };
        uint32_t t = 0;
        // TWO loops.  While EVERY lane of the wave has a template base in front of it -- all but a chunk's last steps -- the step has no lane that sits it out, no
        // adapter and no tail: the iteration runs unmasked and compiled for the template part alone.  The second loop is the general one (lanes whose read is
        // complete or that have none return at once; phase changes; adapter and tail iterations).  One loop for both kept the read's state in two register sets with
        // moves between them around the "lane not running" join of EVERY step: 20.36 G -> 19.57 G vector instructions per launch of 10 M pairs, 216 -> 223 M pairs/s
        // (What did NOT move those copies, each measured on the device -- DESIGN_LOG.md section 11: tied asm operands on the state, both kinds of step behind a
        // wave-uniform branch inside ONE loop (11 % more instructions), the loop tested at its bottom, the phase changes behind a call.)
        for (; !RSQ_ANY(!m.in_template()); ++t) {
            begin_step(t);
            m.template iterate<true>(S, tab, st, src, out);
            __builtin_amdgcn_wave_barrier();
        }
        for (; RSQ_ANY(m.phase != ReadMachine::kDone); ++t) {      // a read is complete (or a lane has none) exactly when its machine is in kDone: a plain compare for the ballot
            begin_step(t);
            m.step(S, tab, st, src, out);                    // a lane whose read is complete (or that has none: phase kDone from the start) returns at once
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (active) {
        m.finalize(meta);
        out.finish();
    }
}

// The scheduler of the binned kernels.  The workgroup's first wave chooses the bin: the one with the most chunks left per workgroup on it (counting the
// newcomer), found by a strided scan of the bins' counters (the first choice: by the workgroup's number); the choice is left in the LDS words behind the image ([0] the bin or 0xFFFFFFFF: nothing
// left anywhere, [1] the bin whose image is staged).  chunk(img, qbase, seg, tile, place, active) runs the 64 items perm[place + lane] (place is wave-uniform).
template <uint32_t MASK, class Chunk>
__device__ void fill_binned_loop(const DevSim &S, float *lds_image, const FillBins &bins, Chunk &&chunk) {
    RSQ_LDS float *img = (RSQ_LDS float *)lds_image;
    RSQ_LDS uint32_t *sched = reinterpret_cast<RSQ_LDS uint32_t *>(img + (MASK ? RSQ_PLAN(S, total_words) : 0u));
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x == 0) sched[1] = 0xFFFFFFFFu;
    for (bool first_choice = true;; first_choice = false) {
        if (threadIdx.x < 64u) {
            uint64_t best = 0;
            if (first_choice) {
                // all workgroups choose at once and cannot see each other yet: workgroup g of G begins with the bin that holds chunk (g + 1/2) / G of all chunks,
                // so that the bins start with workgroups in proportion to their sizes
                const uint64_t total = bins.chunk_ptr[bins.n_bins], target = ((2u * (uint64_t)blockIdx.x + 1u) * total) / (2u * gridDim.x);
                uint32_t lo = 0, hi = bins.n_bins;                     // the last bin with chunk_ptr[bin] <= target
                while (hi - lo > 1u) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (bins.chunk_ptr[mid] <= target) lo = mid;
                    else hi = mid;
                }
                best = total ? ((uint64_t)1 << 32) | (0xFFFFFFFFu - lo) : 0u;
            } else {
                // score = chunks left * 4096 / (workgroups on the bin + 1); ties go to the lower bin
                for (uint32_t b = lane; b < bins.n_bins; b += 64u) {
                    // other workgroups change these two with atomics while this one scans them: loads that go to the device-coherent level, not to this CU's vector cache
                    // (a stale "chunks left" would send the workgroup back to a used-up bin, from which it returns here)
                    const uint32_t n = (bins.bin_count[b] + 63u) / 64u, done = __hip_atomic_load(&bins.next_chunk[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                   on_it = __hip_atomic_load(&bins.workers[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), left = done < n ? n - done : 0u;
                    const uint64_t score = (uint64_t)left * 4096u / (on_it + 1u), packed = (score << 32) | (0xFFFFFFFFu - b);
                    if (left && packed > best) best = packed;
                }
            }
            for (uint32_t d = 32; d; d >>= 1) {
                const uint64_t other = ((uint64_t)(uint32_t)__shfl_xor((int)(best >> 32), (int)d, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)best, (int)d, 64);
                best = other > best ? other : best;
            }
            if (lane == 0) {
                const uint32_t bin = best ? 0xFFFFFFFFu - (uint32_t)best : 0xFFFFFFFFu;
                sched[0] = bin;
                if (best) atomicAdd(&bins.workers[bin], 1u);
            }
        }
        __syncthreads();
        // what comes out of LDS is wave-uniform, but a vector register to the compiler: readfirstlane makes it scalar again (else every index derived
        // from the segment and the image becomes per-lane arithmetic)
        const uint32_t bin = (uint32_t)__builtin_amdgcn_readfirstlane((int)sched[0]);
        if (bin == 0xFFFFFFFFu) break;
        const uint32_t seg = bin / RSQ_SIM(S, n_tiles), tile = bin - seg * RSQ_SIM(S, n_tiles), qbase = image_qbase(S, seg, tile);
        if (bin != (uint32_t)__builtin_amdgcn_readfirstlane((int)sched[1])) {      // all read before anyone writes (barriers inside the staging)
            fill_stage_image<MASK>(S, lds_image, qbase);
            if (threadIdx.x == 0) sched[1] = bin;
        }
        const uint32_t first = bins.bin_first[bin], n_items = bins.bin_count[bin];
        for (;;) {
            uint32_t c = 0;
            if (lane == 0) c = atomicAdd(&bins.next_chunk[bin], 1u);
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
            if ((uint64_t)c * 64u >= n_items) break;
            chunk(img, qbase, seg, tile, first + c * 64u, c * 64u + lane < n_items);
        }
        __syncthreads();                                               // every wave is done with the bin
        if (threadIdx.x == 0) atomicSub(&bins.workers[bin], 1u);
    }
}

// one chunk of 64 pairs, mate `seg`: lane = row `row` of the segment's raw arrays if active.  Not binned: row = pair of the batch.  Binned: row = place in
// perm, the pair is perm[row] and its fragment record sorted[row] (a wave's rows are consecutive either way: 256-byte stores; k_format_write goes
// through perm as well); the pair index itself is read where it is needed, before and after the reads, and does not live through them.
template <uint32_t MASK, bool VAR, bool BINNED>
__device__ void fill_pair_chunk(const DevSim &S, const NameTable &names, RSQ_LDS float *img, uint32_t qbase, uint32_t seg, uint32_t bin_tile, uint64_t row, bool active,
                                const Fragment *frags, uint64_t n_pairs, uint64_t adapter_only_first, const RawLayout &raw, uint32_t *sizes, const FragmentVar *fvars,
                                const uint32_t *perm) {
    const uint64_t at = active ? row : 0u, r_out = (uint64_t)seg * n_pairs + at;
    ReadOut out = raw.out_of(r_out);
    Fragment f{};
    if (active && frags) f = frags[at];
    // the read's stream and template (CreateReads :634-721 / SimulateAdapterOnlyPairs :2359-2382)
    const bool from_fragment = frags != nullptr;
    FragmentVar fv{};
    if (VAR && active && fvars) fv = fvars[at];
    const bool need_pair = !from_fragment || raw.templates != nullptr;              // wave-uniform
    const uint64_t pair0 = BINNED ? (need_pair && active ? perm[at] : 0u) : at;
    const PairStream ps = pair_stream(from_fragment ? &f : nullptr, fv.sub, adapter_only_first + pair0);
    const Stream st{S.seed, ps.c0, ps.c1, ps.c2, pair_c3(kDomPair, ps.strand, seg, f.allele)};
    const uint32_t tile = BINNED ? bin_tile : (active ? draw_tile(S, ps.c0, ps.c1, ps.c2, pair_c3(kDomPair, ps.strand, 2, f.allele)) : 0u);
    ReadMeta meta;
    if constexpr (VAR) {                                            // launched for fragments only
        VariantSrc src = variant_src(S, f, fvars ? &fv : nullptr, seg);
        if (raw.templates) src.converted = raw.templates + ((uint64_t)seg * n_pairs + pair0) * raw.template_words;
        fill_wave_reads<MASK>(S, img, qbase, seg, active, st, tile, f.len, src, out, meta);
    } else {
        FragmentSrc src = from_fragment && active ? fragment_src(S, f, seg) : FragmentSrc{S.ref_words, 0, 0, 0, false, S.sys_fwd, nullptr, nullptr};      // len 0 = empty template
        if (from_fragment && raw.templates) src.converted = raw.templates + ((uint64_t)seg * n_pairs + pair0) * raw.template_words;
        fill_wave_reads<MASK>(S, img, qbase, seg, active, st, tile, f.len, src, out, meta);
    }
    if (active) {
        raw.meta[r_out] = meta;
        const uint64_t pair = BINNED ? perm[at] : at;
        sizes[(uint64_t)seg * n_pairs + pair] = record_size(S, names, from_fragment ? &f : nullptr, adapter_only_first + pair + 1u, meta, VAR && fvars ? &fv : nullptr);       // bytes of its FASTQ record
    }
}

template <uint32_t MASK, bool VAR, bool BINNED>
__device__ __forceinline__ void fill_reads_body(const DevSim &S, const NameTable &names, const Fragment *frags, uint64_t n_pairs, uint64_t adapter_only_first, const RawLayout &raw,
                                                uint32_t *sizes, uint32_t *chunk_counters, const FragmentVar *fvars, const FillBins &bins) {
    extern __shared__ __attribute__((aligned(16))) float lds_image[];
    if constexpr (BINNED) {
        fill_binned_loop<MASK>(S, lds_image, bins, [&](RSQ_LDS float *img, uint32_t qbase, uint32_t seg, uint32_t tile, uint32_t place, bool active) {
            fill_pair_chunk<MASK, VAR, true>(S, names, img, qbase, seg, tile, place + (threadIdx.x & 63u), active, frags ? bins.frags : nullptr, n_pairs, adapter_only_first, raw,
                                             sizes, fvars ? bins.fvars : nullptr, bins.perm);
        });
    } else {
        const uint32_t seg = blockIdx.x & 1u, qbase = image_qbase(S, seg, 0u);
        RSQ_LDS float *img = fill_stage_image<MASK>(S, lds_image, qbase);
        const uint32_t lane = threadIdx.x & 63u;
        for (;;) {
            uint32_t chunk = 0;
            if (lane == 0) chunk = atomicAdd(&chunk_counters[seg], 1u);
            chunk = __shfl(chunk, 0, 64);
            const uint64_t first = (uint64_t)chunk * 64u;                   // past the end: the wave is done
            if (first >= n_pairs) break;
            fill_pair_chunk<MASK, VAR, false>(S, names, img, qbase, seg, 0u, first + lane, first + lane < n_pairs, frags, n_pairs, adapter_only_first, raw, sizes, fvars, nullptr);
        }
    }
}
// the library's own instantiations (every shape of profile); a kernel compiled for one profile wraps the same body (rsq_spec.h)
template <uint32_t MASK, bool VAR = false, bool BINNED = false>
__global__ void __launch_bounds__(fill_block(VAR)) k_fill_reads(DevSim S, NameTable names, const Fragment *frags, uint64_t n_pairs, uint64_t adapter_only_first,
                                                          RawLayout raw, uint32_t *sizes, uint32_t *chunk_counters, const FragmentVar *fvars, FillBins bins) {
    fill_reads_body<MASK, VAR, BINNED>(S, names, frags, n_pairs, adapter_only_first, raw, sizes, chunk_counters, fvars, bins);
}

// seqToIllumina (ApplyErrorsAndQualityToFastaInput, Simulator.cpp:2403-2512) through the same workgroups: the records were
// partitioned by template segment (rec_index: segment-0 records first; rec_count[2] on the device), a workgroup serves one
// segment and its waves pull chunks of 64 records; or, BINNED, by (segment, tile) like the pairs.
struct RecordJob {
    uint64_t first_index;
    uint32_t read_len;
    const uint8_t *seqs, *dom, *rate;
    const uint32_t *frag_len;
    const uint32_t *rec_index, *rec_count;
    uint64_t n_records;
    // nullptr: record i is bytes [i * read_len, (i + 1) * read_len) of the arrays; else bytes [rec_at[i], rec_at[i] + rec_len[i]) of arrays of array_bytes bytes
    const uint32_t *rec_at, *rec_len;
    uint32_t array_bytes;
    // records parsed on the device (the PACKED kernels): a half-word per base at rec_at[i] of `codes` (array_bytes half-words), seqs / dom / rate unused
    const uint16_t *codes;
};
// lane = record i if active; `row`: its row of the raw arrays (i, or binned its place in perm: the text / array kernels go through perm as well)
template <uint32_t MASK, bool BINNED, bool PACKED>
__device__ void fill_record_chunk(const DevSim &S, const RecordJob &job, RSQ_LDS float *img, uint32_t qbase, uint32_t seg, uint32_t bin_tile, uint64_t i, uint64_t row, bool active,
                                  const RawLayout &raw) {
    const uint64_t idx = job.first_index + i;
    const Stream st{S.seed, (uint32_t)idx, (uint32_t)(idx >> 32), 0u, pair_c3(kDomErrModel, 0, seg)};
    ReadOut out = raw.out_of(active ? row : 0u);
    ReadMeta meta;
    const uint32_t tile = BINNED ? bin_tile : (active ? draw_tile(S, st.c0, st.c1, st.c2, pair_c3(kDomErrModel, 0, 2)) : 0u);
    if constexpr (PACKED) {
        const PackedRecordSrc src = packed_record_src(job.codes, job.rec_at[i], job.rec_len[i], job.array_bytes);
        fill_wave_reads<MASK>(S, img, qbase, seg, active, st, tile, job.frag_len[i], src, out, meta);
    } else {
        const RecordSrc src = job.rec_at ? record_src_at(job.seqs, job.dom, job.rate, job.rec_at[i], job.rec_len[i], job.array_bytes)
                                         : record_src(job.seqs, job.dom, job.rate, job.read_len, i, job.n_records);
        fill_wave_reads<MASK>(S, img, qbase, seg, active, st, tile, job.frag_len[i], src, out, meta);
    }
    if (active) raw.meta[row] = meta;
}
template <uint32_t MASK, bool BINNED, bool PACKED>
__device__ __forceinline__ void fill_records_body(const DevSim &S, const RecordJob &job, const RawLayout &raw, uint32_t *chunk_counters, const FillBins &bins) {
    extern __shared__ __attribute__((aligned(16))) float lds_image[];
    if constexpr (BINNED) {
        fill_binned_loop<MASK>(S, lds_image, bins, [&](RSQ_LDS float *img, uint32_t qbase, uint32_t seg, uint32_t tile, uint32_t place, bool active) {
            const uint32_t row = place + (threadIdx.x & 63u);
            fill_record_chunk<MASK, true, PACKED>(S, job, img, qbase, seg, tile, active ? bins.perm[row] : 0u, row, active, raw);
        });
    } else {
        const uint32_t seg = blockIdx.x & 1u, qbase = image_qbase(S, seg, 0u);
#if defined(RSQ_TRACE_FILL)      // measurements (exp/): when a workgroup began, had its image, and ended -- device clock, three words per workgroup behind the counters
        uint64_t *trace = reinterpret_cast<uint64_t *>(chunk_counters) + 2 + 4 * blockIdx.x;
        if (threadIdx.x == 0) trace[0] = wall_clock64();
#endif
        RSQ_LDS float *img = fill_stage_image<MASK>(S, lds_image, qbase);
#if defined(RSQ_TRACE_FILL)
        if (threadIdx.x == 0) trace[1] = wall_clock64();
#endif
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t n_mine = job.rec_count[seg];
        const uint32_t *index = job.rec_index + (seg ? job.rec_count[0] : 0u);
        for (;;) {
            uint32_t chunk = 0;
            if (lane == 0) chunk = atomicAdd(&chunk_counters[seg], 1u);
            chunk = __shfl(chunk, 0, 64);
            const uint64_t first = (uint64_t)chunk * 64u;                   // past the end: the wave is done
            if (first >= n_mine) break;
            const bool active = first + lane < n_mine;
            const uint32_t i = active ? index[first + lane] : 0u;
            fill_record_chunk<MASK, false, PACKED>(S, job, img, qbase, seg, 0u, i, i, active, raw);
#if defined(RSQ_TRACE_FILL)
            if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(trace + 3), 1ull);      // chunks this workgroup ran
#endif
        }
#if defined(RSQ_TRACE_FILL)
        if (lane == 0) atomicMax(reinterpret_cast<unsigned long long *>(trace + 2), (unsigned long long)wall_clock64());      // the workgroup's last wave
#endif
    }
}
template <uint32_t MASK, bool BINNED = false, bool PACKED = false>
__global__ void __launch_bounds__(kFillBlockWalk) k_fill_records(DevSim S, RecordJob job, RawLayout raw, uint32_t *chunk_counters, FillBins bins) {
    fill_records_body<MASK, BINNED, PACKED>(S, job, raw, chunk_counters, bins);
}
#if !defined(RSQ_SPEC)
// ReadLength (Simulator.h:185-198) looks a record's fragment length up in InsertLengths() and ReadLengthsByFragmentLength(segment) with Vect::at, which ends the
// reference's run for a length outside ("Called index ... range is from ... to ..."): the first such record, so that the caller can say the same instead of
// reading beyond the tables.  lo / hi per segment: the lengths that have rows in both (all of them for a profile with one read length).
struct FragmentRange {
    uint32_t lo[2], hi[2];
};
__global__ void k_fragment_range(const uint8_t *segs, const uint32_t *frag_len, uint64_t n, FragmentRange range, uint32_t *first_bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t seg = segs[i] ? 1u : 0u, fl = frag_len[i];
    if (fl < range.lo[seg] || fl >= range.hi[seg]) atomicMin(first_bad, (uint32_t)i);
}
// the partition: flags for the scan, then the scatter once the number of segment-1 records before every record is known
__global__ void k_record_flags(const uint8_t *segs, uint64_t n, uint32_t *flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = segs[i] ? 1u : 0u;
}
__global__ void k_record_partition(const uint8_t *segs, uint64_t n, const uint64_t *ones_before, uint32_t *rec_index, uint32_t *rec_count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n1 = (uint32_t)ones_before[n], n0 = (uint32_t)n - n1;
    if (i == 0) {
        rec_count[0] = n0;
        rec_count[1] = n1;
    }
    if (i >= n) return;
    const uint32_t before = (uint32_t)ones_before[i];
    if (segs[i]) rec_index[n0 + before] = (uint32_t)i;
    else rec_index[(uint32_t)i - before] = (uint32_t)i;
}

// --methylation: one lane per read writes its converted template before the read kernel runs
__global__ void __launch_bounds__(256) k_methylation_templates(DevSim S, const Fragment *frags, uint64_t n_pairs, RawLayout raw) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= 2u * n_pairs) return;
    const uint32_t seg = r >= n_pairs ? 1u : 0u;
    convert_template(S, frags[r - seg * n_pairs], seg, raw.templates + r * raw.template_words, raw.template_words);
}

// variants of any kind: Reference::ReferenceSequence with variants (GetOrgSeq, Simulator.cpp:1909-1914) for both mates of every pair
__global__ void __launch_bounds__(256) k_variant_templates(DevSim S, const Fragment *frags, const FragmentVar *fvars, uint64_t n_pairs, RawLayout raw) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= 2u * n_pairs) return;
    const uint32_t seg = r >= n_pairs ? 1u : 0u;
    const uint64_t pair = r - seg * n_pairs;
    variant_template(S, frags[pair], fvars[pair], seg, raw.templates + r * raw.template_words, raw.template_words);
}

#endif
#endif  // __HIPCC__

#if RSQ_DEVICE_BUILD && !defined(RSQ_SPEC)      // the library's kernels: not part of a read kernel compiled for one profile (rsq_spec.h)
// FASTQ text: one wave per 16 consecutive records of one file (grid.y = template segment = output file).  The records
// occupy one contiguous byte range of the output, so the wave formats them into an LDS image of that range (laid out with
// the same alignment modulo 16 as the destination) and then copies the image out with aligned 16-byte stores.  Four lanes
// share a record: lanes 0-15 write the id line and the first half of the bases, lanes 16-31 the second half, lanes 32-47 and
// 48-63 the two halves of the qualities.  The kernel is latency-bound (dependent byte pushes, four load round trips), so
// short per-lane work and many waves per CU matter more than instruction count: the image is as large as the records need (lds_bytes, dynamic: the host sizes
// it from the longest record of the call before -- 8 KiB a wave were twenty waves per CU and 5.6 ms per 10 M pairs, 6 KiB are 26 and 4.7 ms); a wave whose records do
// not fit writes them straight to HBM.
// (records per wave: 16, four lanes each.  Eight records with eight lanes each need half the LDS and half the work per lane, but their loads of the raw
// rows cover 32 bytes instead of 64: 2.3 ms slower per 10 M pairs)
#ifndef RSQ_FORMAT_RECORDS
#define RSQ_FORMAT_RECORDS 16
#endif
constexpr uint32_t kFormatRecords = RSQ_FORMAT_RECORDS, kFormatLdsMax = 16u * 1024u, kFormatLdsMin = 1024u;
// the image for records of at most `record_bytes` (the longest record of the call before and a few bytes for a digit more in its numbers), whole 128 bytes
RSQ_HD uint32_t format_lds_bytes(uint64_t record_bytes) {
    const uint64_t want = (kFormatRecords * record_bytes + 16u + 127u) & ~(uint64_t)127u;
    return (uint32_t)(want < kFormatLdsMin ? kFormatLdsMin : want > kFormatLdsMax ? kFormatLdsMax : want);
}
// the longest of n record sizes (one atomic per wave)
__global__ void __launch_bounds__(256) k_max_size(const uint32_t *sizes, uint64_t n, uint32_t *longest) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) m = max(m, sizes[i]);
    for (uint32_t d = 32; d; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, (int)d, 64));
    if ((threadIdx.x & 63u) == 0 && m) atomicMax(longest, m);
}
// PERM (the read kernel ran binned by tile): the wave's 16 records are those whose raw rows are consecutive -- pairs perm[first .. first + 15] --, their
// texts lie anywhere in the output, so every record has a 512-byte slot of the image (same alignment modulo 16 as its destination) and its four lanes
// copy it out.
template <bool PERM>
__global__ void __launch_bounds__(64) k_format_write(DevSim S, NameTable names, const Fragment *frags, uint64_t n_pairs, uint64_t adapter_only_first, RawLayout raw,
                                                    const uint64_t *offsets0, const uint64_t *offsets1, char *dst0, char *dst1, uint64_t cap0, uint64_t cap1,
                                                    const FragmentVar *fvars, const uint32_t *perm, uint32_t lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) char s_text[];
    const uint32_t lane = threadIdx.x, seg = blockIdx.y, rec = lane & (kFormatRecords - 1u), part = lane / kFormatRecords;
    constexpr uint32_t kLineParts = 32u / kFormatRecords;                          // lanes that share a line of a record
    const bool is_qual = part >= kLineParts;
    const uint32_t sub = part % kLineParts;
    const uint64_t first = (uint64_t)blockIdx.x * kFormatRecords;
    if (first >= n_pairs) return;
    const uint64_t *offsets = seg ? offsets1 : offsets0;
    char *dst = seg ? dst1 : dst0;
    if (offsets[n_pairs] > (seg ? cap1 : cap0)) return;                            // the caller's buffer is too small: write nothing (RSQ_ENOSPC)
    const uint64_t last = first + kFormatRecords < n_pairs ? first + kFormatRecords : n_pairs;
    const uint64_t row = first + rec;                                              // of the raw arrays, within the segment
    const bool active = row < last;
    const uint64_t pair = PERM ? (active ? perm[row] : 0u) : row;
    // the byte range of the wave's text (PERM: of the lane's record) and where it starts modulo 16
    const uint64_t g_begin = PERM ? (active ? offsets[pair] : 0u) : offsets[first], g_end = PERM ? (active ? offsets[pair + 1u] : 0u) : offsets[last];
    const uint32_t skew = (uint32_t)((uint64_t)(uintptr_t)(dst + g_begin) & 15u), bytes = (uint32_t)(g_end - g_begin);
    const uint32_t kSlot = (lds_bytes / kFormatRecords) & ~15u;
    const bool through_lds = PERM ? __all(skew + bytes <= kSlot) != 0 : skew + bytes <= lds_bytes;      // wave-uniform
    ReadMeta m;
    Fragment f;
    FragmentVar fv;
    uint64_t r = 0;
    if (active) {
        r = (uint64_t)seg * n_pairs + row;
        m = raw.meta[r];
        if (frags) f = frags[pair];
        if (frags && fvars) fv = fvars[pair];
    }
    const WordColumn seq = raw.seq_of(r), qual = raw.qual_of(r), ops = raw.ops_of(r);
    const uint64_t ao_number = adapter_only_first + pair + 1u;
    if (!through_lds) {                                                            // oversized ids: write straight to HBM
        if (active && part == 0u) format_record(S, names, frags != nullptr, f, ao_number, m, seq, qual, ops, dst + offsets[pair], frags && fvars, fv);
        return;
    }
    const uint32_t slot_at = PERM ? rec * kSlot : 0u;
    if (active) {
        RSQ_LDS char *rec_text = (RSQ_LDS char *)s_text + slot_at + skew + (PERM ? 0u : (uint32_t)(offsets[pair] - g_begin));
        const uint32_t header = (uint32_t)(offsets[pair + 1u] - offsets[pair]) - 2u * m.read_len - 4u;
        const uint32_t all_words = (m.read_len + 3u) >> 2, per = (all_words + kLineParts - 1u) / kLineParts;      // the parts end on word boundaries
        const uint32_t first_word = sub * per, line_at = header + (is_qual ? m.read_len + 3u : 0u);
        const uint32_t part_at = part == 0u ? 0u : line_at + (4u * first_word < m.read_len ? 4u * first_word : m.read_len);
        WordSinkT<RSQ_LDS char *> t(rec_text + part_at);
        if (part == 0u) format_header(S, names, frags != nullptr, f, ao_number, m, ops, t, frags && fvars, fv);
        format_line_part(is_qual ? qual : seq, m.read_len, is_qual, first_word, per, sub == kLineParts - 1u, t);
        t.finish();
    }
    __syncthreads();
    const uint32_t lo = skew, hi = skew + bytes;                                   // LDS byte range (within the slot) holding text
    char *g_chunk0 = dst + g_begin - skew;                                         // 16-byte aligned
    const char *s_from = s_text + slot_at;
    // the image goes out in aligned 16-byte stores: all lanes over the wave's range, or (PERM) a record's four lanes over its slot
    for (uint32_t c = (PERM ? part : lane) * 16u; c < hi; c += (PERM ? 64u / kFormatRecords : 64u) * 16u) {
        if (c >= lo && c + 16u <= hi) {
            *reinterpret_cast<uint4 *>(g_chunk0 + c) = *reinterpret_cast<const uint4 *>(s_from + c);
        } else {
            for (uint32_t b = c < lo ? lo : c; b < c + 16u && b < hi; ++b) g_chunk0[b] = s_from[b];
        }
    }
}

#endif

}  // namespace rsq
