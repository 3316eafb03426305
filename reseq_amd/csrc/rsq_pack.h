// rsq_pack.h -- host-side logic of the simulator that does not depend on where the packed arrays live:
// packing the profile tables and the 2-bit reference, the number of pairs / block numbering, the chain list of
// the systematic-error pre-pass and the spline / threshold arithmetic of the bias normalisation.
// The shipped library uploads through hipMalloc (rsq_sim.hip); tests/hostemu keeps the arrays in host memory so
// that the per-lane functions of rsq_core.h / rsq_kernels.h can be checked against the oracle without a GPU.
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include <atomic>
#include <exception>
#include <mutex>
#include <thread>

#include "rsq_host.h"
#include "rsq_kernels.h"

namespace rsq {

// Arrays are owned by the uploader in scopes: what is put at create time lives as long as the simulator; what a pre-pass puts
// (block numbering, thresholds) is released when that pre-pass runs again, so that repeated rsq_sim_prepare / rsq_sim_set_normalization
// calls on one simulator do not pile arrays up.
enum UploadScope : int { kScopeCreate = 0, kScopePlan = 1, kScopeNormalization = 2, kUploadScopes = 3 };
struct Uploader {                                   // copies a host array to wherever the kernels will read it
    int current_scope = kScopeCreate;
    virtual void release_scope(int scope) { (void)scope; }                        // frees what was put in the scope
    virtual void *put_bytes(const void *data, size_t bytes) = 0;
    virtual void *put_zeros(size_t bytes) = 0;                                     // an array of zero bytes (no host copy of it)
    virtual void write_bytes(void *dst, const void *src, size_t bytes) = 0;      // overwrite part of an array put earlier
    virtual void read_bytes(void *dst_host, const void *src, size_t bytes) = 0;   // read back what the pre-pass kernels wrote
    virtual void bind_thread() {}                                                 // called once by every helper thread that will call read_bytes
    virtual ~Uploader() {}
    template <class T>
    T *put(const std::vector<T> &v) {
        static const T kZero{};
        return static_cast<T *>(put_bytes(v.empty() ? &kZero : v.data(), (v.empty() ? 1 : v.size()) * sizeof(T)));
    }
};

struct ScopedUpload {                               // puts inside the block belong to `scope`; `fresh`: what the scope held before is released first
    Uploader &up;
    int before;
    ScopedUpload(Uploader &u, int scope, bool fresh) : up(u), before(u.current_scope) {
        if (fresh) up.release_scope(scope);
        up.current_scope = scope;
    }
    ~ScopedUpload() { up.current_scope = before; }
};

struct SimState {
    Options opt = options();                         // the switches as they stood when the simulator was created: later rsq_set_option calls shape later simulators
    Profile prof;
    bool has_ref = false;
    std::vector<std::string> ref_first_names, ref_ids;   // ReferenceIdFirstPart / ReferenceId
    std::vector<uint32_t> seq_len;
    std::vector<uint64_t> seq_word_off, seq_base_off;
    uint64_t total_ref_size = 0;
    uint32_t chain_chunk = 256;                      // positions per chunk of the systematic-error chains: chain_chunk_len(total_ref_size), set when a run of the chains begins
    std::vector<uint64_t> ref_words_host;            // host copy of the packed reference (32 bases per word, seq_word_off): DominantBase carry-over between chains
    uint32_t ref_code(uint32_t seq, uint32_t pos) const { return (uint32_t)(ref_words_host[seq_word_off[seq] + (pos >> 5)] >> ((pos & 31u) * 2u)) & 3u; }
    // variants (-V): host copies of what the device holds, and of the two table families their systematic errors are drawn from
    bool has_variants = false;
    uint32_t num_alleles = 1;
    int variants_mode = 0;                           // DevSim::variants_loaded
    std::vector<DevVariant> variants;
    std::vector<uint32_t> var_ptr;
    std::vector<uint8_t> var_bases;
    std::vector<uint16_t> var_err_fwd, var_err_rev;  // filled by build_variant_sys_errors
    uint16_t *dev_var_err_fwd = nullptr, *dev_var_err_rev = nullptr;
    double expected_passing = 0.0;                   // cells of one start position expected to pass the zero threshold (largest coverage group)
    std::vector<AlleleVar> allele_map;               // coordinate maps of the alleles (rsq_variants.h)
    std::vector<uint32_t> allele_map_ptr;            // [n_seqs * num_alleles + 1]
    std::vector<ExtraStart> extra;                   // starts inside inserted bases, per sequence in loop order
    std::vector<uint32_t> extra_seq_ptr;             // [n_seqs + 1]
    std::vector<double> host_pool;
    std::vector<uint8_t> host_par0;
    std::vector<DevTable> host_dom_error, host_error_rate;
    DevSim dev{};
    NameTable names{};
    uint16_t *sys_fwd = nullptr, *sys_rev = nullptr, *adapter_sys[2] = {nullptr, nullptr};   // written by the chain pre-pass
    uint32_t rmax = 0, read_stride = 0, ops_stride = 0, max_adapter = 0, template_words = 0;
    uint64_t insert_lengths_from = 0;      // InsertLengths().from(): what Vect::at names when a seqToIllumina record's fragment length lies outside
    // prepare() results
    bool prepared = false;
    bool normalized = false;                         // the sharded pre-pass: rsq_sim_prepare_normalization has run since the plan
    uint32_t prepared_lo = 0, prepared_hi = 0;       // blocks whose systematic-error tracks are finished: all of them after rsq_sim_prepare, the rank's range after the sharded pre-pass
    uint64_t seed = 0, total_pairs = 0, adapter_only_pairs = 0;
    uint32_t n_groups = 0, passes = 0, total_blocks = 0;
    std::vector<uint32_t> coverage_groups, first_block, n_blocks, block_seq;
    std::vector<double> thresholds, norm_by_len, ref_seq_bias;
    std::string ref_bias_file;                       // --refBiasFile, consumed by plan_simulation when ref_bias_mode is kRefBiasFile
    double bias_normalization = 0;
    std::string plan_note;                           // what pack_tables has to say about the read kernels' route (rsq_last_warning after rsq_sim_create)
    // kept for export_reference: the variants' mode as pack_reference chose it (pack_methylation may raise it), the methylation regions as uploaded
    int variants_mode_packed = 0;
    bool has_methylation = false;
    std::vector<uint32_t> meth_ptr_host, meth_first_host, meth_second_host;
    std::vector<double> meth_rate_host;
};

static inline uint32_t cdiv(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// --------------------------------------------------------------------------------------- packing (create)
// LDS budget of one k_fill_reads workgroup (MI355X: 160 KiB per CU, one workgroup per CU)
constexpr uint32_t kLdsBudgetBytes = 160u * 1024u;
constexpr uint32_t kLdsRateRowsFirst = 64;    // rows of the error-rate margins staged in LDS before anything optional (97 % of all positions have rate 0)

inline void pack_tables(SimState &s, Uploader &up) {
    const Profile &p = s.prof;
    std::vector<double> pool;
    std::vector<uint8_t> par0;
    auto pack = [&](const std::vector<HostTable> &tabs) {
        std::vector<DevTable> out(tabs.size());
        for (size_t i = 0; i < tabs.size(); ++i) {
            const HostTable &t = tabs[i];
            DevTable d{};
            d.k = (uint32_t)t.par0.size();
            for (uint32_t v : t.par0) d.max_value = std::max(d.max_value, v);          // par0_off: assigned below, in the order the LDS images copy the values
            const uint32_t kp = row_stride(d.k);                       // even stride: zero pad column when K is odd
            for (uint32_t n = 0; n < t.nm; ++n) {
                d.from[n] = t.from[n];
                d.rows[n] = t.to[n] - t.from[n];
                const size_t rows = d.k ? d.rows[n] : 0;
                if (pool.size() + rows * kp > 0xFFFFFFF0ull) throw Error("probability tables exceed 2^32 entries");
                d.off[n] = (uint32_t)pool.size();
                for (size_t r = 0; r < rows; ++r) {
                    pool.insert(pool.end(), t.dim2[n].begin() + r * d.k, t.dim2[n].begin() + (r + 1) * d.k);
                    pool.insert(pool.end(), kp - d.k, 0.0);
                }
            }
            out[i] = d;
        }
        return out;
    };
    std::vector<DevTable> quality = pack(p.quality), seq_quality = pack(p.seq_quality), base_call = pack(p.base_call), dom_error = pack(p.dom_error),
                          error_rate = pack(p.error_rate), indels = pack(p.indels);

    // Single-precision copies of the families the read kernel draws from per base, for its screened draws (rsq_core.h): every margin,
    // rows of one slot per family, pad columns zero.
    const uint32_t T = p.n_tiles();
    LdsPlan plan{};
    // The outcome values (par0) in the order the LDS images want them: the indel tables', then per (segment, tile) those of its quality, sequence-quality
    // and base-call tables -- an image copies the indel range and the range of its tiles --, then the two families of the systematic-error chains.  Every
    // range starts on a word.
    auto put_par0 = [&](DevTable &d, const HostTable &t) {
        d.par0_off = (uint32_t)par0.size();
        for (uint32_t v : t.par0) par0.push_back((uint8_t)v);
    };
    auto pad_par0 = [&] { par0.resize((par0.size() + 3u) & ~(size_t)3u, 0); };
    for (size_t i = 0; i < indels.size(); ++i) put_par0(indels[i], p.indels[i]);
    pad_par0();
    const uint32_t par0_indel_bytes = (uint32_t)par0.size();
    std::vector<uint32_t> par0_tile_first(2 * T + 1, 0);                                            // byte range of the tables of (segment, tile)
    for (uint32_t g = 0; g < 2 * T; ++g) {
        par0_tile_first[g] = (uint32_t)par0.size();
        for (uint32_t i = 0; i < 4; ++i) put_par0(quality[g * 4 + i], p.quality[g * 4 + i]);
        put_par0(seq_quality[g], p.seq_quality[g]);
        for (uint32_t i = 0; i < 20; ++i) put_par0(base_call[g * 20 + i], p.base_call[g * 20 + i]);
        pad_par0();
    }
    par0_tile_first[2 * T] = (uint32_t)par0.size();
    for (size_t i = 0; i < dom_error.size(); ++i) put_par0(dom_error[i], p.dom_error[i]);
    for (size_t i = 0; i < error_rate.size(); ++i) put_par0(error_rate[i], p.error_rate[i]);

    std::vector<float> pool32;
    auto kmax_of = [](const std::vector<DevTable> &tabs) {
        uint32_t kmax = 0;
        for (const DevTable &d : tabs) kmax = std::max(kmax, d.k);
        return kmax;
    };
    auto copy32 = [&](std::vector<DevTable> &tabs, uint32_t slot) {
        for (DevTable &d : tabs) {
            d.off32 = (uint32_t)pool32.size();
            d.f32_ok = 1;
            if (!d.k) continue;
            const uint32_t kp = row_stride(d.k);
            for (uint32_t n = 0; n < 4; ++n)
                for (uint32_t r = 0; r < d.rows[n]; ++r) {
                    const double *row = pool.data() + d.off[n] + (size_t)r * kp;
                    for (uint32_t c = 0; c < slot; ++c) {
                        const double v = c < d.k ? row[c] : 0.0;
                        if (!(v == 0.0 || (v >= 0x1p-60 && v <= 0x1p29))) d.f32_ok = 0;          // also NaN, negative
                        pool32.push_back((float)v);
                    }
                }
            if (pool32.size() * sizeof(float) > 0xFFFFFFF0ull) throw Error("single-precision probability tables exceed 4 GB (the kernels address them by 32-bit byte offsets)");
        }
    };
    // Draws decided by the random word alone: the indel draw almost always returns "no indel".
    // LogArrayResult::Draw (rsq_core.h draw_rows) sums the products p_c from the top column down and returns the highest column z >= 1 whose sum T(z)
    // exceeds u * S, else 0: column z iff T(z+1) <= u * S < T(z).  For every choice of rows p_c <= f_c * p_z with f_c = the product over the margins of
    // the largest ratio row[c] / row[z] over the rows that may be combined, so with B_lo = sum of the f_c below z and B_hi = the sum above it:
    // T(z+1) / S <= B_hi / (1 + B_hi) and T(z) / S >= 1 / (1 + B_lo); for u between the two (the reference's own roundings are 1e-14 of them; 1e-9 is
    // allowed for) column z is certain without reading a row.  Margin 0 enters with ONE row (the lane knows it).  A row with row[z] = 0 < row[c]
    // leaves no bound.  The bounds are kept as 16-bit fractions lo16 <= hi16:
    // the draw is certain when lo16 <= (word >> 16) < hi16.
    struct Certain {
        uint32_t lo16 = 0, hi16 = 0;
    };
    // worst[z * 8 + c]: margins 1 .. 3 over all their rows, < 0: no bound
    auto worst_ratios = [&](const DevTable &d, double (&worst)[64]) {
        const uint32_t kp = row_stride(d.k);
        for (uint32_t z = 0; z < d.k; ++z)
            for (uint32_t c = 0; c < d.k; ++c) {
                double f = 1.0;
                for (uint32_t n = 1; n < 4 && f >= 0.0; ++n) {
                    if (!d.rows[n]) continue;
                    double w = 0.0;
                    for (uint32_t r = 0; r < d.rows[n]; ++r) {
                        const double *row = pool.data() + d.off[n] + (size_t)r * kp;
                        if (row[z] > 0.0) w = std::max(w, row[c] / row[z]);
                        else if (row[c] > 0.0) w = -1.0;
                        if (w < 0.0) break;
                    }
                    f = w < 0.0 ? -1.0 : f * w;
                }
                worst[z * 8u + c] = f;
            }
    };
    auto certain_column = [&](const DevTable &d, const double (&worst)[64], uint32_t row0, uint32_t z) {
        const double *m0 = pool.data() + d.off[0] + (size_t)row0 * row_stride(d.k);
        Certain none, r;
        double below = 0.0, above = 0.0;
        for (uint32_t c = 0; c < d.k; ++c) {
            if (c == z) continue;
            double f;
            if (m0[z] > 0.0) f = m0[c] / m0[z];
            else if (m0[c] > 0.0) return none;
            else f = 0.0;
            if (worst[z * 8u + c] < 0.0) return none;
            (c < z ? below : above) += f * worst[z * 8u + c];
        }
        const double lo = above / (1.0 + above) * (1.0 + 1e-9), hi = 1.0 / (1.0 + below) * (1.0 - 1e-9);
        r.lo16 = (uint32_t)std::ceil(lo * 65536.0);
        r.hi16 = (uint32_t)std::floor(hi * 65536.0);
        return r.lo16 < r.hi16 ? r : none;
    };
    // indel tables: one bound per table, for margin 0 (the indel position) at its row 0 and the column of value 0 = "no indel" (prob_sum 0 means the
    // same, Simulator.cpp:349, so all-zero rows do no harm)
    auto certain_no_indel = [&](DevTable &d) {
        d.sure_range = 0;
        if (!d.k || d.k > 8u || !d.f32_ok || !d.rows[0]) return;
        double worst[64];
        worst_ratios(d, worst);
        for (uint32_t z = 0; z < d.k; ++z)
            if (0 == par0[d.par0_off + z]) {
                const Certain c = certain_column(d, worst, 0, z);
                d.sure_range = c.lo16 | (c.hi16 << 16);
            }
    };
    // The read kernel's three families over their common ranges (FamilyGeo, rsq_types.h): table i of the family at [i][table_rows][slot], row r of margin n = the
    // table's own row of value from[n] + r (its edge rows outside its own range: AdjustIndeces); empty tables and tables outside the screen's preconditions are zeros.
    // Then the outcome value of every column, [i][slot] bytes behind the pool of outcome values.
    auto family = [&](std::vector<DevTable> &tabs, uint32_t nm, uint32_t slot) {
        FamilyGeo g{};
        uint32_t to[4] = {0, 0, 0, 0};
        bool any = false;
        for (const DevTable &d : tabs) {
            if (!d.k) continue;
            for (uint32_t n = 0; n < nm; ++n) {
                g.from[n] = any ? std::min(g.from[n], d.from[n]) : d.from[n];
                to[n] = any ? std::max(to[n], d.from[n] + d.rows[n]) : d.from[n] + d.rows[n];
            }
            any = true;
        }
        for (uint32_t n = 0; n < 4; ++n) {
            const uint32_t rows = n < nm ? std::max(1u, to[n] - g.from[n]) : 0u;
            g.before[n] = g.table_rows;
            g.last[n] = rows ? rows - 1u : 0u;
            g.table_rows += rows;
        }
        g.off32 = (uint32_t)pool32.size();
        g.lds = g.lds2 = kNoLds;
        if ((pool32.size() + (uint64_t)tabs.size() * g.table_rows * slot) * sizeof(float) > 0xFFFFFFF0ull)      // PoolRow32 / lds_ring_item: 32-bit BYTE offsets into the pool
            throw Error("single-precision probability tables exceed 4 GB (the kernels address them by 32-bit byte offsets)");
        for (DevTable &d : tabs) {
            d.off32 = 0;
            d.f32_ok = 1;
            const uint32_t kp = row_stride(d.k);
            for (uint32_t n = 0; n < nm && d.k; ++n)
                for (uint32_t r = 0; r < d.rows[n]; ++r)
                    for (uint32_t c = 0; c < d.k; ++c) {
                        const double v = pool[d.off[n] + (size_t)r * kp + c];
                        if (!(v == 0.0 || (v >= 0x1p-60 && v <= 0x1p29))) d.f32_ok = 0;              // also NaN, negative
                    }
            const bool zeros = !d.k || !d.f32_ok;
            for (uint32_t n = 0; n < nm; ++n)
                for (uint32_t r = 0; r <= g.last[n]; ++r) {
                    const int64_t own = (int64_t)g.from[n] + r - (int64_t)d.from[n];
                    const double *row = zeros ? nullptr : pool.data() + d.off[n] + (size_t)std::min<int64_t>(std::max<int64_t>(own, 0), (int64_t)d.rows[n] - 1) * kp;
                    for (uint32_t c = 0; c < slot; ++c) pool32.push_back(row && c < d.k ? (float)row[c] : 0.f);
                }
        }
        return g;
    };
    auto family_values = [&](FamilyGeo &g, const std::vector<DevTable> &tabs, uint32_t slot) {
        g.values_src = (uint32_t)par0.size();
        for (const DevTable &d : tabs)
            for (uint32_t c = 0; c < slot; ++c) par0.push_back(c < d.k ? par0[d.par0_off + c] : (uint8_t)0);
    };
    const Options &opt = s.opt;
    const uint32_t min_quads = opt.min_quality_quads > 0 ? (uint32_t)opt.min_quality_quads : 0u;     // measurements: a wider instantiation than the profile needs
    for (uint32_t q : kQualityQuads)
        if (!plan.quads_q && q >= min_quads && quads_of(kmax_of(quality)) <= q) plan.quads_q = q;
    const bool screenable = plan.quads_q && quads_of(kmax_of(base_call)) <= kQuadsSmall && quads_of(kmax_of(indels)) <= kQuadsSmall;
    plan.slot_q = row_slot32(plan.quads_q);
    plan.slot_b = plan.slot_i = kSlotSmall;
    if (screenable) {
        plan.q = family(quality, 4, plan.slot_q);
        plan.b = family(base_call, 4, plan.slot_b);
        plan.i = family(indels, 3, plan.slot_i);
        if (!opt.no_indel_skip)
            for (DevTable &d : indels) certain_no_indel(d);
    }
    // the two families of the systematic-error chains: rows of whole quads, read from HBM
    s.dev.chain_quads = 0;
    s.dev.force_exact = opt.force_exact ? 1u : 0u;
    for (uint32_t q : kChainQuads)
        if (!s.dev.chain_quads && quads_of(kmax_of(error_rate)) <= q && quads_of(kmax_of(dom_error)) <= kQuadsSmall) s.dev.chain_quads = q;
    std::vector<uint32_t> chain_sure(1, 0u);
    if (s.dev.chain_quads) {
        copy32(dom_error, 4u * kQuadsSmall);
        copy32(error_rate, 4u * s.dev.chain_quads);
        // Error-rate draws decided by the random word alone: at most positions the table is one of "rate 0 almost surely" (no dominant error there), and a lane
        // that knows its rows of margin 0 (distance) and margin 2 (start rate) can tell from the bound above -- the worst ratios taken over the rows of margin 1
        // (G/C percent) only -- whether its word gives value 0 whatever the G/C row is.  One range per (table, row of margin 0, row of margin 2), read instead of
        // three rows of up to 104 values: the chains are bound by that traffic (the rows come from L2; without this draw a pass takes 40 % of its time).
        if (!opt.no_indel_skip)
            for (DevTable &d : error_rate) {
                d.sure_range = 0;
                if (!d.k || !d.f32_ok || !d.rows[0] || !d.rows[1] || !d.rows[2]) continue;
                uint32_t z = d.k;
                for (uint32_t c = 0; c < d.k; ++c)
                    if (0 == par0[d.par0_off + c]) z = c;
                if (z == d.k) continue;
                const uint32_t kp = row_stride(d.k);
                std::vector<double> worst1(d.k, 0.0);                // over the rows of margin 1: the largest row[c] / row[z]; < 0: no bound
                for (uint32_t c = 0; c < d.k; ++c)
                    for (uint32_t r = 0; r < d.rows[1] && worst1[c] >= 0.0; ++r) {
                        const double *row = pool.data() + d.off[1] + (size_t)r * kp;
                        if (row[z] > 0.0) worst1[c] = std::max(worst1[c], row[c] / row[z]);
                        else if (row[c] > 0.0) worst1[c] = -1.0;
                    }
                d.sure_range = (uint32_t)chain_sure.size();
                for (uint32_t r0 = 0; r0 < d.rows[0]; ++r0)
                    for (uint32_t r2 = 0; r2 < d.rows[2]; ++r2) {
                        const double *m0 = pool.data() + d.off[0] + (size_t)r0 * kp, *m2 = pool.data() + d.off[2] + (size_t)r2 * kp;
                        double below = 0.0, above = 0.0;
                        bool bound = m0[z] > 0.0 && m2[z] > 0.0;
                        for (uint32_t c = 0; c < d.k && bound; ++c) {
                            if (c == z) continue;
                            if (worst1[c] < 0.0) bound = false;
                            else (c < z ? below : above) += (m0[c] / m0[z]) * worst1[c] * (m2[c] / m2[z]);
                        }
                        uint32_t range = 0;
                        if (bound) {
                            const double lo = above / (1.0 + above) * (1.0 + 1e-9), hi = 1.0 / (1.0 + below) * (1.0 - 1e-9);
                            const uint32_t lo16 = (uint32_t)std::ceil(lo * 65536.0), hi16 = (uint32_t)std::floor(hi * 65536.0);
                            if (lo16 < hi16) range = lo16 | (hi16 << 16);
                        }
                        chain_sure.push_back(range);
                    }
            }
    }
    s.dev.chain_sure = up.put(chain_sure);
    if (screenable) {
        pad_par0();
        family_values(plan.q, quality, plan.slot_q);
        family_values(plan.b, base_call, plan.slot_b);
        family_values(plan.i, indels, plan.slot_i);
    }

    // LDS plan of the read kernels (rsq_kernels.h "LDS staging"): one image per template segment with the tables of ALL tiles when they fit the 160 KiB,
    // else one image per (segment, tile) -- the read kernel then serves one tile per workgroup (k_fill_reads<MASK, VAR, true>: reads binned by tile).  The
    // most valuable rows first: descriptors and outcome values, quality margins 0 + 1, base-call margin 0, the waves' rings and one error-rate row are
    // required; more error-rate rows, base-call margin 2 and indel margin 0 take what is left.
    uint32_t rate_rows_q = ~0u, rate_rows_b = ~0u;
    if (opt.rate_rows > 0) rate_rows_q = rate_rows_b = (uint32_t)std::min<int64_t>(opt.rate_rows, 1 << 20);       // at most so many error-rate rows
    rate_rows_q = std::min(rate_rows_q, plan.q.last[3] + 1u);         // rows of the common range (FamilyGeo)
    rate_rows_b = std::min(rate_rows_b, plan.b.last[3] + 1u);
    const uint64_t budget = kLdsBudgetBytes / 4u - kSchedWords;
    // tries an image of Ti tiles; fills `plan` and the tables' offsets when the required parts fit
    auto plan_image = [&](uint32_t Ti) {
        const uint32_t n_img = 2 * T / Ti;                                                         // images: (segment, group of Ti tiles)
        uint32_t par0_bytes = 0;
        for (uint32_t g = 0; g < n_img; ++g) par0_bytes = std::max(par0_bytes, par0_indel_bytes + par0_tile_first[(g + 1) * Ti] - par0_tile_first[g * Ti]);
        const uint32_t par0_words = (par0_bytes + 15u) / 16u * 4u;                                 // whole 16 bytes: rows stay aligned
        const uint32_t desc_words = (lds_desc_count(Ti) * kDescWords + 3u) / 4u * 4u + par0_words;
        const uint32_t values_words = (4 * Ti * plan.slot_q + 20 * Ti * plan.slot_b + 12 * plan.slot_i + 15u) / 16u * 4u;      // a byte per column
        const uint32_t rows_q = plan.q.last[0] + 1 + plan.q.last[1] + 1, rows_b = plan.b.last[0] + 1, rows_b2 = plan.b.last[2] + 1, rows_i = plan.i.last[0] + 1;
        // floats from one table's block to the next: an odd number of 16-byte groups (LdsPlan: the tables' copies of a row in different bank groups)
        auto odd_stride = [](uint64_t floats) { return (uint32_t)(((floats / 4u) | 1u) * 4u); };      // floats is a multiple of 4: an even number of groups gets one more
        const uint32_t stride_q = odd_stride((uint64_t)rows_q * plan.slot_q), stride_b = odd_stride((uint64_t)rows_b * plan.slot_b), stride_b2 = odd_stride((uint64_t)rows_b2 * plan.slot_b),
                       stride_i = odd_stride((uint64_t)rows_i * plan.slot_i);
        const uint64_t need_q = (uint64_t)4 * Ti * stride_q, need_b = (uint64_t)20 * Ti * stride_b, need_b2 = (uint64_t)20 * Ti * stride_b2, need_i0 = (uint64_t)12 * stride_i;
        uint64_t need = (uint64_t)desc_words + values_words + need_q + need_b;
        auto need_rate = [&](uint32_t rows_q, uint32_t rows_b) {
            return (uint64_t)4 * Ti * odd_stride((uint64_t)rows_q * plan.slot_q) + (uint64_t)20 * Ti * odd_stride((uint64_t)rows_b * plan.slot_b);
        };
        const uint32_t ring_stride = 4 * Ti * plan.slot_q;
        need += (uint64_t)kFillWavesMax * kRingRows * ring_stride;                              // the waves' rings
        if (need + need_rate(1, 1) > budget) return false;
        // what is left goes to: error-rate rows of the quality tables up to kLdsRateRowsFirst (a lane whose rate has no staged row
        // repeats its draw in double precision), the base-call margin over the number of errors, the indel margin over the indel
        // position, then more error-rate rows of both families
        uint32_t rq = 1, rb = 1;
        while (rq < std::min(rate_rows_q, kLdsRateRowsFirst) && need + need_rate(rq + 1, 1) <= budget) ++rq;
        const bool stage_b2 = need + need_b2 + need_rate(rq, 1) <= budget;
        if (stage_b2) need += need_b2;
        const bool stage_i0 = need + need_i0 + need_rate(rq, 1) <= budget;
        if (stage_i0) need += need_i0;
        while (rb < std::min(rate_rows_b, kLdsRateRowsFirst) && need + need_rate(rq, rb + 1) <= budget) ++rb;
        while (rq < rate_rows_q && need + need_rate(rq + 1, rb) <= budget) ++rq;
        while (rb < rate_rows_b && need + need_rate(rq, rb + 1) <= budget) ++rb;
        plan.img_tiles = Ti;
        plan.binned = (Ti < T || opt.image_tiles == 1) ? 1u : 0u;
        plan.par0_words = par0_words;
        plan.par0_indel_bytes = par0_indel_bytes;
        plan.desc_words = desc_words;
        plan.ring_stride = ring_stride;
        plan.rate_rows_q = rq;
        plan.rate_rows_b = rb;
        plan.q.values = desc_words * 4u;
        plan.b.values = plan.q.values + 4 * Ti * plan.slot_q;
        plan.i.values = plan.b.values + 20 * Ti * plan.slot_b;
        uint32_t at = desc_words + values_words;
        plan.q.lds = at, plan.q.lds_rows = rows_q, plan.q.lds_stride = stride_q, at += (uint32_t)need_q;
        plan.b.lds = at, plan.b.lds_rows = rows_b, plan.b.lds_stride = stride_b, at += (uint32_t)need_b;
        plan.b.lds2 = stage_b2 ? at : kNoLds;
        plan.b.lds2_stride = stride_b2;
        if (stage_b2) at += (uint32_t)need_b2;
        plan.i.lds = stage_i0 ? at : kNoLds;
        plan.i.lds_rows = rows_i;
        plan.i.lds_stride = stride_i;
        if (stage_i0) at += (uint32_t)need_i0;
        plan.ring_off = at;
        plan.q3_off = plan.ring_off + kFillWavesMax * kRingRows * plan.ring_stride;
        plan.q3_stride = odd_stride((uint64_t)plan.rate_rows_q * plan.slot_q);
        plan.b3_stride = odd_stride((uint64_t)plan.rate_rows_b * plan.slot_b);
        plan.b3_off = plan.q3_off + 4 * Ti * plan.q3_stride;
        plan.total_words = plan.b3_off + 20 * Ti * plan.b3_stride;
        plan.mask = plan.quads_q;
        return true;
    };
    plan.img_tiles = T;
    if (screenable && opt.fill_mode != 0) {
        const bool per_tile_first = opt.image_tiles == 1 && T > 1;
        if (!(per_tile_first && plan_image(1)) && !plan_image(T) && !(T > 1 && plan_image(1)))
            s.plan_note = "the read kernels' table image does not fit the " + std::to_string(kLdsBudgetBytes / 1024u) +
                          " KiB of local memory even for one tile: every per-base draw runs in double precision from device memory (several times slower)";
    } else if (!screenable && opt.fill_mode != 0)
        s.plan_note = "the profile's tables are outside what the screened single-precision draws are built for (more than " + std::to_string(4u * kQualityQuads[4]) +
                      " quality values, or more than 8 base-call / indel outcomes): every per-base draw runs in double precision from device memory (several times slower)";
    if ((plan.desc_words | plan.q3_off | plan.b3_off | plan.ring_off | plan.ring_stride | plan.slot_q | plan.slot_b | plan.slot_i | plan.q.off32 | plan.b.off32 | plan.i.off32 |
         (plan.mask ? plan.q.lds | plan.b.lds | (plan.b.lds2 != kNoLds ? plan.b.lds2 : 0u) | (plan.i.lds != kNoLds ? plan.i.lds : 0u) : 0u)) & 3u)
        throw Error("internal: LDS rows must start on 16-byte boundaries");      // a misaligned ds_read_b128 is 2.4x slower
    if (opt.trace_plan)
        fprintf(stderr, "[rsq] LDS image: mask %u, %u of %u tiles per image, %u words (%u KiB), desc %u, quality slot %u, rate rows %u / %u, q3 %u b3 %u, ring %u x %u, b2 %d i0 %d\n", plan.mask,
                plan.img_tiles, T, plan.total_words, plan.total_words / 256, plan.desc_words, plan.slot_q, plan.rate_rows_q, plan.rate_rows_b, plan.q3_off, plan.b3_off, plan.ring_off,
                plan.ring_stride, (int)(plan.b.lds2 != kNoLds), (int)(plan.i.lds != kNoLds));
    s.dev.lds = plan;
    s.dev.quality = up.put(quality);
    s.dev.seq_quality = up.put(seq_quality);
    s.dev.base_call = up.put(base_call);
    s.dev.dom_error = up.put(dom_error);
    s.dev.error_rate = up.put(error_rate);
    s.dev.indels = up.put(indels);
    par0.resize(((par0.size() + 3u) & ~(size_t)3u) + (size_t)plan.par0_words * 4u + 16u, 0);      // an image copies whole words, up to par0_words of them from a range's start
    pool.push_back(0.0);
    pool.push_back(0.0);
    pool32.resize(pool32.size() + 8, 0.f);
    s.dev.pool = up.put(pool);
    s.dev.pool32 = up.put(pool32);
    s.dev.par0 = up.put(par0);
    s.host_pool = pool;                                              // for the systematic errors of variants (drawn on the host)
    s.host_par0 = par0;
    s.host_dom_error = dom_error;
    s.host_error_rate = error_rate;
}

// the read kernel is instantiated for: 0 = every draw in double precision from HBM, kQualityQuads = screened draws with the LDS image;
// a forced mode (RSQ_FILL_MODE, tests) can only take the image away
inline uint32_t effective_fill_mask(uint32_t plan_mask, int forced) { return forced == 0 ? 0u : plan_mask; }

inline void pack_profile(SimState &s, Uploader &up) {
    const Profile &p = s.prof;
    DevSim &d = s.dev;
    d.n_tiles = p.n_tiles();
    d.phred_offset = p.phred_offset;
    d.max_len_deletion = p.max_len_deletion;
    d.reset_distance = p.reset_distance;
    d.tile_cp = up.put(discrete_cp(p.tile_abundance.data(), p.tile_abundance.size()));
    d.tiles = up.put(p.tiles);
    s.rmax = 0;
    uint32_t max_adapter = 0;
    for (int seg = 0; seg < 2; ++seg) {
        const HostAdapters &a = p.adapters[seg];
        DevAdapters &da = d.adapters[seg];
        da.n = a.n();
        da.seqs = up.put(a.seqs);
        da.seq_ptr = up.put(a.seq_ptr);
        da.adapter_cp = up.put(discrete_cp(a.significant.data(), a.significant.size()));
        std::vector<double> cut_cp;
        for (uint32_t i = 0; i < a.n(); ++i) {
            std::vector<double> cp = discrete_cp(a.cut.data() + a.cut_ptr[i], a.cut_ptr[i + 1] - a.cut_ptr[i]);
            cp.resize(a.cut_ptr[i + 1] - a.cut_ptr[i]);
            cut_cp.insert(cut_cp.end(), cp.begin(), cp.end());
            max_adapter = std::max(max_adapter, a.seq_ptr[i + 1] - a.seq_ptr[i]);
        }
        cut_cp.push_back(1.0);
        da.cut_cp = up.put(cut_cp);
        da.cut_ptr = up.put(a.cut_ptr);
        da.cut_from = up.put(a.cut_from);
        s.adapter_sys[seg] = up.put(std::vector<uint16_t>(a.seqs.size() + 1, 0));
        da.sys = s.adapter_sys[seg];

        const Vect<uint64_t> &rl = p.read_lengths[seg];
        DevReadLengths &dr = d.read_lengths[seg];
        if (rl.v.empty()) throw Error("profile has no read lengths");
        dr.fixed = rl.v.size() == 1 ? (uint32_t)rl.from : 0u;
        dr.to = (uint32_t)rl.to();
        dr.row_first = (uint32_t)p.rl_by_fl[seg].from;
        dr.rows = (uint32_t)p.rl_by_fl[seg].row_ptr.size() - 1;
        dr.row_ptr = up.put(p.rl_by_fl[seg].row_ptr);
        dr.row_from = up.put(p.rl_by_fl[seg].row_from);
        dr.values = up.put(p.rl_by_fl[seg].values);
        s.rmax = std::max(s.rmax, dr.to - 1);
    }
    d.polya_cp = up.put(discrete_cp(p.polya.v.data(), p.polya.v.size()));
    d.polya_n = (uint32_t)p.polya.v.size();
    d.polya_from = (uint32_t)p.polya.from;
    std::vector<double> ocp = discrete_cp(p.overrun_bases, 4);           // Simulator.h:168: the N is dropped
    for (int i = 0; i < 4; ++i) d.overrun_cp[i] = ocp[i];
    d.insert_from = (uint32_t)std::max<uint64_t>(1, p.insert_lengths.from);       // Simulator.cpp:2300
    s.insert_lengths_from = p.insert_lengths.from;
    if (p.insert_lengths.to() > 65536) throw Error("insert lengths above 65535 are not supported (fragment lengths are 16-bit fields of the sieve's records)");
    d.insert_to = (uint32_t)p.insert_lengths.to();
    std::vector<uint64_t> il(d.insert_to + 1, 0);
    std::vector<double> ilb(d.insert_to + 1, 0.0), gcb(101, 0.0);
    for (uint32_t i = 0; i < d.insert_to; ++i) {
        il[i] = p.insert_lengths[i];
        ilb[i] = p.insert_lengths_bias[i];
    }
    for (uint32_t i = 0; i < 101; ++i) gcb[i] = p.gc_bias[i];
    d.insert_lengths = up.put(il);
    d.insert_lengths_bias = up.put(ilb);
    d.gc_bias = up.put(gcb);
    d.sur_bias = up.put(p.sur_bias);
    d.dispersion[0] = p.dispersion[0];
    d.dispersion[1] = p.dispersion[1];

    s.max_adapter = max_adapter;
    s.read_stride = (s.rmax + 15u) & ~15u;        // 16-byte rows: k_format_write reads them with 16-byte loads
    const uint32_t max_iter = 2u * s.rmax + p.max_len_deletion + max_adapter + 4u;
    s.ops_stride = (max_iter + 15u) / 16u;
}

// How the kernels simulate a variant set: 1 = substitutions only: every allele gets its own copy of the packed reference;
// 2 = anything else (and more than eight alleles): alleles as coordinate maps (rsq_variants.h)
inline int variants_mode_for(const Variants &v) {
    if (v.num_alleles > kMaxDevAlleles) throw Error("variants: more than " + std::to_string(kMaxDevAlleles) + " alleles");
    if (v.num_alleles > 8) return 2;                               // a copy of the reference per allele only for a few alleles
    for (const std::vector<Variant> &seq : v.by_seq)
        for (const Variant &x : seq)
            if (x.var_seq.size() != 1) return 2;
    return 1;
}

// The coordinate maps of a sequence's alleles (rsq_variants.h): per allele the variants it has, in position order, with the running
// difference between allele and reference coordinates and the running G/C difference, and a sentinel behind them.  An allele with
// two different variants at one position cannot come out of ReadVariants (overlapping records are refused, Reference.cpp:166-168, and one
// record gives an allele one alternative); the map relies on it, so it is checked.
inline void allele_maps_of_sequence(const std::vector<Variant> &vars, const std::vector<uint8_t> &codes, uint32_t num_alleles, const std::string &name,
                                    std::vector<AlleleVar> &out, std::vector<uint32_t> &ptr) {
    auto is_gc_code = [](uint8_t c) { return c == 1 || c == 2; };
    for (uint32_t a = 0; a < num_alleles; ++a) {
        int32_t shift = 0, gc = 0;
        int64_t last_pos = -1;
        for (size_t i = 0; i < vars.size(); ++i) {
            const Variant &v = vars[i];
            if (!v.in_allele(a)) continue;
            if ((int64_t)v.position == last_pos)
                throw Error("variants: allele " + std::to_string(a) + " has two different variants at position " + std::to_string(v.position + 1) + " of " + name +
                            " (overlapping records on one haplotype)");
            last_pos = v.position;
            out.push_back(AlleleVar{v.position, (uint32_t)i, shift, gc});
            shift += (int32_t)v.var_seq.size() - 1;
            for (uint8_t b : v.var_seq) gc += is_gc_code(b) ? 1 : 0;
            gc -= is_gc_code(codes[v.position]) ? 1 : 0;
        }
        out.push_back(AlleleVar{(uint32_t)codes.size(), (uint32_t)vars.size(), shift, gc});
        ptr.push_back((uint32_t)out.size());
    }
}

// The extra passes of SimulateFromGivenBlock's do-while loop (Simulator.cpp:2299-2352) at the start positions of one sequence:
// CheckForInsertedBasesToStartFrom (:1870-1896) run over the variants, in loop order
inline void extra_starts_of_sequence(const std::vector<Variant> &vars, std::vector<ExtraStart> &out) {
    int32_t first_variant_id = 0;
    const size_t n = vars.size();
    while ((size_t)first_variant_id < n) {
        const uint32_t cur_start_position = vars[(size_t)first_variant_id].position;       // plain positions leave first_variant_id_ alone
        uint32_t start_variant_pos = 0, sub = 0;
        do {
            if (sub) out.push_back(ExtraStart{cur_start_position, sub, first_variant_id, start_variant_pos});
            if ((size_t)first_variant_id < n && vars[(size_t)first_variant_id].position == cur_start_position) {
                if (start_variant_pos) {
                    if (++start_variant_pos >= vars[(size_t)first_variant_id].var_seq.size()) {
                        start_variant_pos = 0;
                        ++first_variant_id;
                    }
                } else {
                    while ((size_t)first_variant_id < n && vars[(size_t)first_variant_id].position == cur_start_position && 2 > vars[(size_t)first_variant_id].var_seq.size())
                        ++first_variant_id;
                }
                if (0 == start_variant_pos && (size_t)first_variant_id < n && vars[(size_t)first_variant_id].position == cur_start_position) start_variant_pos = 1;
            }
            ++sub;
        } while (start_variant_pos);
    }
}

// f(item) for every item on a few host threads (items handed out in order)
template <class F>
inline void pack_on_threads(size_t n_items, F f) {
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t n_threads = std::min<size_t>(std::max<size_t>(1, n_items), std::max(1u, std::min(hw ? hw : 1u, 16u)));
    std::atomic<size_t> next{0};
    std::exception_ptr failed;
    std::mutex m;
    auto work = [&]() {
        try {
            for (size_t t; (t = next.fetch_add(1)) < n_items;) f(t);
        } catch (...) {
            std::lock_guard<std::mutex> g(m);
            failed = std::current_exception();
        }
    };
    std::vector<std::thread> helpers;
    for (size_t t = 1; t < n_threads; ++t) helpers.emplace_back(work);
    work();
    for (std::thread &t : helpers) t.join();
    if (failed) std::rethrow_exception(failed);
}

// The second half of pack_reference: the simulator's state is complete on the host (sequence tables, variants, allele maps, extra starts in SimState; `packed` =
// the 2-bit words of the reference and, with a substitution-only variant set, of every allele's copy; `gc_prefix` likewise), now everything goes to where the
// kernels read it.  Also the route of a reference another process packed (import_reference below): both end in the same arrays.
inline void upload_reference(SimState &s, Uploader &up, std::vector<uint64_t> packed, const std::vector<uint32_t> &gc_prefix) {
    DevSim &d = s.dev;
    uint64_t words = 0, bases = 0;
    for (uint32_t len : s.seq_len) {
        words += (len + 31) / 32 + 1;
        bases += len;
    }
    s.var_err_fwd.assign(s.var_bases.size() + 1, 0);
    s.var_err_rev.assign(s.var_bases.size() + 1, 0);
    d.var_bases = up.put(s.var_bases);
    {
        std::vector<uint32_t> bases_gc(s.var_bases.size() + 1, 0);
        for (size_t k = 0; k < s.var_bases.size(); ++k) bases_gc[k + 1] = bases_gc[k] + ((s.var_bases[k] == 1 || s.var_bases[k] == 2) ? 1u : 0u);
        d.var_bases_gc = up.put(bases_gc);
    }
    d.allele_map = up.put(s.allele_map);
    d.allele_map_ptr = up.put(s.allele_map_ptr);
    s.dev_var_err_fwd = up.put(s.var_err_fwd);
    s.dev_var_err_rev = up.put(s.var_err_rev);
    d.var_err_fwd = s.dev_var_err_fwd;
    d.var_err_rev = s.dev_var_err_rev;
    d.extra = up.put(s.extra);
    d.walk_error = up.put(std::vector<uint32_t>(2, 0));
    d.block_extra_ptr = nullptr;                                  // set by plan_simulation once the blocks are numbered
    if (2 == s.variants_mode) {                                   // templates are written out per mate (k_variant_templates)
        s.template_words = (s.rmax + s.prof.max_len_deletion + 31u) / 32u + 1u;
        if (s.template_words > kTemplateWordsMax) throw Error("templates longer than 2048 bases are not supported with insertion / deletion variants");
    }
    d.variants = up.put(s.variants);
    d.var_ptr = up.put(s.var_ptr);
    d.ref_words = up.put(packed);
    packed.resize(words + 1);                                     // the allele copies are the device's
    packed.shrink_to_fit();
    s.ref_words_host = std::move(packed);
    d.gc_prefix = up.put(gc_prefix);
    d.seq_word_off = up.put(s.seq_word_off);
    d.seq_len = up.put(s.seq_len);
    d.seq_base_off = up.put(s.seq_base_off);
    s.sys_fwd = static_cast<uint16_t *>(up.put_zeros((bases + 8) * sizeof(uint16_t)));
    s.sys_rev = static_cast<uint16_t *>(up.put_zeros((bases + 8) * sizeof(uint16_t)));
    d.sys_fwd = s.sys_fwd;
    d.sys_rev = s.sys_rev;
    std::string names;
    std::vector<uint32_t> ptr{0};
    for (const std::string &n : s.ref_first_names) {
        names += n;
        ptr.push_back((uint32_t)names.size());
    }
    names.push_back(' ');
    s.names.names = up.put(std::vector<char>(names.begin(), names.end()));
    s.names.name_ptr = up.put(ptr);
}

inline void pack_reference(SimState &s, Uploader &up, const Reference &r, const Variants *variants = nullptr) {
    DevSim &d = s.dev;
    d.n_seqs = (uint32_t)r.codes.size();
    s.has_ref = true;
    uint64_t words = 0, bases = 0;
    s.seq_len.clear();
    s.seq_word_off.clear();
    s.seq_base_off.clear();
    s.ref_first_names.clear();
    s.ref_ids.clear();
    for (size_t i = 0; i < r.codes.size(); ++i) {
        s.seq_len.push_back((uint32_t)r.codes[i].size());
        s.seq_word_off.push_back(words);
        s.seq_base_off.push_back(bases);
        words += (r.codes[i].size() + 31) / 32 + 1;        // one spare word per sequence
        bases += r.codes[i].size();
        s.ref_first_names.push_back(r.first_part(i));
        s.ref_ids.push_back(r.names[i]);
    }
    s.total_ref_size = bases;
    std::vector<uint64_t> packed(words + 1, 0);
    {   // 32 bases per word; stretches of 4 M bases are independent (whole words each): a few host threads share them
        struct Stretch {
            size_t seq, lo, hi;
        };
        std::vector<Stretch> stretches;
        constexpr size_t kStretch = (size_t)1 << 22;
        for (size_t i = 0; i < r.codes.size(); ++i)
            for (size_t lo = 0; lo < r.codes[i].size(); lo += kStretch) stretches.push_back(Stretch{i, lo, std::min(r.codes[i].size(), lo + kStretch)});
        std::atomic<bool> has_n{false};
        pack_on_threads(stretches.size(), [&](size_t t) {
            {                                                       // whole words: 32 bases each
                const Stretch &st = stretches[t];
                const uint8_t *c = r.codes[st.seq].data();
                uint64_t *w = &packed[s.seq_word_off[st.seq]];
                uint64_t seen = 0;
                for (size_t pos = st.lo; pos < st.hi; pos += 32) {
                    const size_t n = std::min<size_t>(32, st.hi - pos);
                    uint64_t x = 0;
                    if (32 == n) {                                  // eight codes per load: their 2-bit fields, a byte apart, are pushed together in three steps
                        for (uint32_t q = 0; q < 4; ++q) {
                            uint64_t y;
                            memcpy(&y, c + pos + 8 * q, 8);
                            seen |= y;
                            y &= 0x0303030303030303ull;
                            y = (y | y >> 6) & 0x000F000F000F000Full;
                            y = (y | y >> 12) & 0x000000FF000000FFull;
                            y = (y | y >> 24) & 0xFFFFull;
                            x |= y << (16 * q);
                        }
                    } else {
                        for (size_t k = 0; k < n; ++k) {
                            x |= (uint64_t)(c[pos + k] & 3u) << (2 * k);
                            seen |= c[pos + k];
                        }
                    }
                    w[pos >> 5] = x;
                }
                if (seen & 0xFCFCFCFCFCFCFCFCull) has_n = true;
            }
        });
        if (has_n) throw Error("reference still contains N: call rsq_ref_replace_n first");
    }
    std::vector<uint32_t> gc_prefix(words + 1, 0);                 // running G/C totals per word, restarting with every sequence
    auto gc_totals = [&](const uint64_t *copy, uint32_t *out) {     // of one copy of the reference, a sequence per thread
        pack_on_threads(r.codes.size(), [&](size_t i) {
            const size_t n_words = (r.codes[i].size() + 31) / 32;
            uint32_t total = 0;
            for (size_t w = 0; w <= n_words; ++w) {                // the spare word holds the sequence total
                out[s.seq_word_off[i] + w] = total;
                if (w < n_words) {
                    const uint64_t x = copy[s.seq_word_off[i] + w];
                    total += (uint32_t)__builtin_popcountll((x ^ (x >> 1)) & 0x5555555555555555ull);
                }
            }
        });
    };
    gc_totals(packed.data(), gc_prefix.data());
    s.has_variants = variants != nullptr;
    s.num_alleles = variants ? variants->num_alleles : 1u;
    d.num_alleles = s.num_alleles;
    d.hap_stride = 0;
    s.variants.clear();
    s.var_ptr.assign(1, 0);
    s.variants_mode = variants ? variants_mode_for(*variants) : 0;
    s.variants_mode_packed = s.variants_mode;
    d.variants_loaded = (uint32_t)s.variants_mode;
    s.var_bases.clear();
    s.extra.clear();
    s.extra_seq_ptr.assign(1, 0);
    s.allele_map.clear();
    s.allele_map_ptr.assign(1, 0);
    if (1 == s.variants_mode) {
        // copy 0 = the reference, copy 1 + a = allele a with its substitutions; G/C prefix sums per copy
        const size_t stride = words + 1;
        d.hap_stride = stride;
        packed.resize(stride * (1u + s.num_alleles));
        gc_prefix.resize(stride * (1u + s.num_alleles));
        for (uint32_t a = 0; a < s.num_alleles; ++a) {
            uint64_t *hp = &packed[stride * (1u + a)];
            std::copy(packed.begin(), packed.begin() + (ptrdiff_t)stride, hp);
            for (size_t i = 0; i < r.codes.size(); ++i)
                for (const Variant &v : variants->by_seq[i])
                    if (v.in_allele(a)) {
                        uint64_t &w = hp[s.seq_word_off[i] + (v.position >> 5)];
                        const uint32_t sh = (v.position & 31u) * 2u;
                        w = (w & ~((uint64_t)3u << sh)) | ((uint64_t)v.var_seq[0] << sh);
                    }
            gc_totals(hp, &gc_prefix[stride * (1u + a)]);
        }
    }
    if (variants) {
        // a sequence's variants, allele maps and extra starts depend on nothing outside it: a sequence per thread, then one after the other into the arrays
        struct PerSequence {
            std::vector<DevVariant> variants;                      // off counts from the sequence's first base
            std::vector<uint8_t> bases;
            std::vector<AlleleVar> allele_map;
            std::vector<uint32_t> allele_map_ptr;                  // ends, counted from the sequence's first entry
            std::vector<ExtraStart> extra;
        };
        std::vector<PerSequence> per(r.codes.size());
        pack_on_threads(r.codes.size(), [&](size_t i) {
            PerSequence &q = per[i];
            q.variants.reserve(variants->by_seq[i].size());
            for (const Variant &v : variants->by_seq[i]) {
                DevVariant dv{};
                dv.pos = v.position;
                dv.len = (uint32_t)v.var_seq.size();
                dv.off = (uint32_t)q.bases.size();
                dv.allele[0] = v.allele[0];
                dv.allele[1] = v.allele[1];
                q.bases.insert(q.bases.end(), v.var_seq.begin(), v.var_seq.end());
                q.variants.push_back(dv);
            }
            allele_maps_of_sequence(variants->by_seq[i], r.codes[i], s.num_alleles, r.first_part(i), q.allele_map, q.allele_map_ptr);
            if (2 == s.variants_mode && r.codes[i].size() >= d.insert_to) extra_starts_of_sequence(variants->by_seq[i], q.extra);   // sequences that get blocks (:1159)
        });
        size_t n_variants = 0, n_bases = 0, n_map = 0, n_extra = 0;
        for (const PerSequence &q : per) {
            n_variants += q.variants.size();
            n_bases += q.bases.size();
            n_map += q.allele_map.size();
            n_extra += q.extra.size();
        }
        if (n_bases > 0xFFFFFFF0ull) throw Error("variants: more than 2^32 variant bases");
        s.variants.reserve(n_variants);
        s.var_bases.reserve(n_bases);
        s.allele_map.reserve(n_map);
        s.extra.reserve(n_extra);
        for (PerSequence &q : per) {
            const uint32_t base0 = (uint32_t)s.var_bases.size(), map0 = (uint32_t)s.allele_map.size();
            for (DevVariant &dv : q.variants) dv.off += base0;
            s.variants.insert(s.variants.end(), q.variants.begin(), q.variants.end());
            s.var_bases.insert(s.var_bases.end(), q.bases.begin(), q.bases.end());
            s.allele_map.insert(s.allele_map.end(), q.allele_map.begin(), q.allele_map.end());
            for (uint32_t end : q.allele_map_ptr) s.allele_map_ptr.push_back(map0 + end);
            s.extra.insert(s.extra.end(), q.extra.begin(), q.extra.end());
            s.var_ptr.push_back((uint32_t)s.variants.size());
            s.extra_seq_ptr.push_back((uint32_t)s.extra.size());
            q = PerSequence();
        }
    } else {
        s.var_ptr.assign(r.codes.size() + 1, 0);
        s.extra_seq_ptr.assign(r.codes.size() + 1, 0);
    }
    upload_reference(s, up, std::move(packed), gc_prefix);
}

// utilities.hpp:238-262 on the host, only to carry DominantBase::dom_base_ from the end of one chain to the start of
// the next (Clear() does not reset it: Simulator.cpp:723-732, utilities.hpp:279-281).
template <class Acc>
inline uint32_t dom_base_after_chain(const Acc &acc, uint32_t len, uint32_t previous) {
    if (!len) return previous;
    uint32_t cnt[4] = {0, 0, 0, 0};
    for (uint32_t p = len > 5 ? len - 5 : 0; p < len; ++p) ++cnt[acc(p)];
    return find_dominant(acc, cnt, len);
}

// ------------------------------------------------------------------------------- bias normalisation (a14)
// FragmentDistributionStats.cpp:1656-1705 GetSamplePositions
inline std::vector<uint32_t> sample_positions(const Vect<uint64_t> &il) {
    const uint32_t kDist = 20;                                    // FragmentDistributionStats.h:235
    const uint32_t to = (uint32_t)il.to();
    uint32_t first_sample = (uint32_t)std::max<uint64_t>(1, il.from);
    while (first_sample < to && 0 == il[first_sample]) ++first_sample;
    uint32_t num_samples = 0, hit_zero = 0;
    for (uint32_t len = first_sample; len < to; len += kDist) {
        if (hit_zero) {
            if (il[len] >= 10) {
                num_samples += (len - hit_zero) / kDist + 1;
                hit_zero = 0;
            }
        } else if (il[len] > 0) ++num_samples;
        else hit_zero = len;
    }
    if (2 > num_samples) throw Error("Sampling insert lengths did not find at least two usable lengths.");
    std::vector<uint32_t> sp(num_samples);
    sp[0] = first_sample;
    uint32_t found_zeros = 0;
    for (uint32_t k = 1; k < num_samples - found_zeros; ++k) {
        sp[k] = sp[k - 1] + kDist;
        while (0 == il[sp[k]]) {
            ++found_zeros;
            sp[k] += kDist;
        }
    }
    sp.resize(num_samples - found_zeros);
    return sp;
}

// Natural cubic spline through log(norm/len_bias) at the sampled lengths, evaluated for every length
// (FragmentDistributionStats.cpp:418-467 PrepareSplines, :485-496, :1539-1567 FillInWithFittedRatios, :1729-1740).
inline void interpolate_normalization(const Vect<double> &len_bias, const std::vector<uint32_t> &x, std::vector<double> &norm) {
    const size_t n = x.size(), nh = n - 1;
    std::vector<double> pars(n + 1, 1.0);
    for (size_t k = 0; k < n; ++k) {
        const double v = norm[x[k]] / len_bias[x[k]];
        pars[k + 1] = v > 0.0 ? log(v) : log(1e-10);
    }
    std::vector<double> h(nh), mu(nh, 0.0), l(n, 0.0);
    std::vector<std::vector<double>> beta(nh, std::vector<double>(n, 0.0)), z(n, std::vector<double>(n, 0.0)), c(n, std::vector<double>(n, 0.0)),
        b(nh, std::vector<double>(n, 0.0)), d(nh, std::vector<double>(n, 0.0));
    for (size_t i = 0; i < nh; ++i) h[i] = (double)(x[i + 1] - x[i]);
    for (size_t k = 1; k < nh; ++k) {
        beta[k][k + 1] = 3 / h[k];
        beta[k][k] = -3 / h[k] - 3 / h[k - 1];
        beta[k][k - 1] = 3 / h[k - 1];
    }
    for (size_t k = 1; k < nh; ++k) {
        l[k] = 2 * (double)(x[k + 1] - x[k - 1]) - h[k - 1] * mu[k - 1];
        mu[k] = h[k] / l[k];
        for (size_t ai = 0; ai < n; ++ai) z[k][ai] = (beta[k][ai] - h[k - 1] * z[k - 1][ai]) / l[k];
    }
    l[n - 1] = 1.0;
    for (size_t i = nh; i--;) {
        for (size_t ai = 0; ai < n; ++ai) {
            c[i][ai] = z[i][ai] - mu[i] * c[i + 1][ai];
            b[i][ai] = -h[i] * (c[i + 1][ai] + 2 * c[i][ai]) / 3;
            d[i][ai] = (c[i + 1][ai] - c[i][ai]) / 3 / h[i];
        }
        b[i][i + 1] += 1 / h[i];
        b[i][i] -= 1 / h[i];
    }
    for (uint32_t len = 1; len < x[0]; ++len) norm[len] = 0.0;
    double ca = 0, cb = 0, cc = 0, cd = 0;
    size_t k = 0;
    for (; k < n - 1; ++k) {
        ca = pars[k + 1];
        cb = cc = cd = 0.0;
        for (size_t ai = 1; ai < n + 1; ++ai) {
            cb += pars[ai] * b[k][ai - 1];
            cc += pars[ai] * c[k][ai - 1];
            cd += pars[ai] * d[k][ai - 1];
        }
        norm[x[k]] = len_bias[x[k]] * exp(ca);
        for (uint32_t len = x[k] + 1; len < x[k + 1]; ++len) {
            const uint32_t cur = len - x[k];
            norm[len] = len_bias[len] * exp(ca + cb * cur + cc * cur * cur + cd * cur * cur * cur);
        }
    }
    norm[x[k]] = len_bias[x[k]] * exp(pars[k + 1]);
    const uint32_t cur = x[k] - x[k - 1];
    const double slope = cb + cc * cur;
    for (uint32_t len = x[k] + 1; len < norm.size(); ++len) norm[len] = len_bias[len] * exp(pars[k + 1] + (len - x[k]) * slope);
}

inline double threshold0(const double disp[2], double norm, double max_bias, uint32_t num_alleles) {      // FragmentDistributionStats.cpp:2969-2976
    double max_mean = norm * max_bias;
    double max_dispersion = get_dispersion(max_mean, disp[0], disp[1]) / num_alleles;
    max_mean /= num_alleles;
    return pow(max_dispersion / (max_dispersion + max_mean), max_dispersion);
}

// Simulator.cpp:61-78
inline double coverage_prop_lost_from_adapters(const Profile &p) {
    uint64_t adapter_bases = 0, total_bases = 0;
    for (int seg = 2; seg--;) {
        const HostRlByFl &r = p.rl_by_fl[seg];
        for (size_t row = 0; row + 1 < r.row_ptr.size(); ++row) {
            const uint64_t frag_len = r.from + row;
            for (uint32_t j = r.row_ptr[row]; j < r.row_ptr[row + 1]; ++j) {
                const uint64_t read_len = r.row_from[row] + (j - r.row_ptr[row]);
                total_bases += r.values[j] * read_len;
                if (frag_len < read_len) {
                    adapter_bases += (r.values[j] - r.non_mapped[j]) * (read_len - frag_len);
                    adapter_bases += r.non_mapped[j] * read_len;
                }
            }
        }
    }
    return (double)adapter_bases / total_bases;
}


// --------------------------------------------------------------------------------- prepare: pairs and blocks
// Simulator.cpp:2705-2743 (number of pairs, adapter-only share), :2782 (sys_gc_range_), UpdateRefSeqBias kKeep/kNo
// (FragmentDistributionStats.cpp:3352-3364) and the block numbering of CreateUnit/CreateBlock (:911-924,1149-1225).
enum : int { kRefBiasKeep = 0, kRefBiasNo = 1, kRefBiasDraw = 2, kRefBiasFile = 3 };      // RefSeqBiasSimulation (FragmentDistributionStats.h)

// sys_gc_range_ = Divide(sum_read_length, reads) / 2 (Simulator.cpp:2782,2962); returns the average read length
inline double set_sys_gc_range(SimState &s) {
    const Profile &p = s.prof;
    uint64_t reads = 0, sum_read_length = 0;                                         // :2713-2721
    for (int seg = 2; seg--;)
        for (uint64_t len = p.read_lengths[seg].from; len < p.read_lengths[seg].to(); ++len) {
            reads += p.read_lengths[seg][len];
            sum_read_length += p.read_lengths[seg][len] * len;
        }
    if (!reads) throw Error("profile has no reads");
    s.dev.sys_gc_range = (uint16_t)(((sum_read_length + reads / 2) / reads) / 2);
    return (double)sum_read_length / reads;
}

inline void plan_simulation(SimState &s, Uploader &up, uint64_t seed, uint64_t num_read_pairs, double coverage, int ref_bias_mode, const char *base_identifier) {
    const ScopedUpload scope(up, kScopePlan, true);
    const Profile &p = s.prof;
    s.seed = seed;
    s.dev.seed = seed;
    std::string base = (base_identifier && base_identifier[0]) ? base_identifier : "ReseqRead";       // Simulator.cpp:2705-2710
    if (base.size() > sizeof(s.names.base_identifier)) throw Error("record base identifier longer than 64 characters");
    memcpy(s.names.base_identifier, base.data(), base.size());
    s.names.base_len = (uint32_t)base.size();

    const double average_read_length = set_sys_gc_range(s);
    if (!s.has_ref) return;

    if (num_read_pairs) s.total_pairs = num_read_pairs;                              // :2726-2736
    else {
        const double adapter_part = coverage_prop_lost_from_adapters(p);
        if (0.0 == coverage) coverage = p.corrected_coverage;
        s.total_pairs = (uint64_t)round(coverage * s.total_ref_size / average_read_length / 2 / (1 - adapter_part));
    }
    s.adapter_only_pairs = (uint64_t)round((double)s.total_pairs * p.insert_lengths[0] / (p.total_number_reads / 2));    // :2739
    s.total_pairs -= s.adapter_only_pairs;

    const uint32_t n_seqs = s.dev.n_seqs;
    s.ref_seq_bias.assign(n_seqs, 1.0);                                                // UpdateRefSeqBias (FragmentDistributionStats.cpp:3352-3500)
    switch (ref_bias_mode) {
        case kRefBiasKeep:                                                             // falls back to kNo when the counts differ
            if (p.ref_seq_bias.size() == n_seqs) s.ref_seq_bias = p.ref_seq_bias;
            break;
        case kRefBiasNo: break;
        case kRefBiasDraw:                                                             // with replacement from the stored biases
            if (p.ref_seq_bias.empty()) throw Error("refBias draw: the profile stores no reference sequence biases");
            for (uint32_t i = 0; i < n_seqs; ++i) {
                const uint32_t k = (uint32_t)(u32_to_unit(philox(seed, i, 0u, 0u, kDomRefBias << 28).w0) * (double)p.ref_seq_bias.size());
                s.ref_seq_bias[i] = p.ref_seq_bias[k < p.ref_seq_bias.size() ? k : p.ref_seq_bias.size() - 1];
            }
            break;
        case kRefBiasFile: s.ref_seq_bias = read_ref_bias_file(s.ref_bias_file, s.ref_first_names); break;
        default: throw Error("Unknown option chosen for reference sequence bias");
    }

    s.first_block.assign(n_seqs, 0);
    s.n_blocks.assign(n_seqs, 0);
    s.block_seq.assign(1, 0);
    uint32_t next_block = 1;
    for (uint32_t i = 0; i < n_seqs; ++i) {
        if (s.seq_len[i] < s.dev.insert_to) continue;                                // :1159,1186
        s.first_block[i] = next_block;
        s.n_blocks[i] = (s.seq_len[i] + kBlockSize - 1) / kBlockSize;
        for (uint32_t b = 0; b < s.n_blocks[i]; ++b) s.block_seq.push_back(i);
        next_block += s.n_blocks[i];
    }
    s.total_blocks = next_block - 1;
    if (!s.total_blocks) throw Error("All reference sequences are too short for simulating. They should have at least " + std::to_string(s.dev.insert_to) + " bases");
    s.block_seq.push_back(0);
    s.dev.block_seq = up.put(s.block_seq);
    s.dev.first_block = up.put(s.first_block);
    s.dev.total_blocks = s.total_blocks;
    // variants of any kind: where the extra starts of every block begin (index b = block id; [total_blocks + 1] = all of them)
    std::vector<uint32_t> block_extra_ptr(s.total_blocks + 2, 0);
    if (2 == s.variants_mode) {
        for (uint32_t i = 0; i < n_seqs; ++i) {
            uint32_t k = s.extra_seq_ptr[i];                         // the extras of a sequence are sorted by position
            for (uint32_t b = 0; b < s.n_blocks[i]; ++b) {
                while (k < s.extra_seq_ptr[i + 1] && s.extra[k].pos < b * kBlockSize) ++k;
                block_extra_ptr[s.first_block[i] + b] = k;
            }
        }
        block_extra_ptr[s.total_blocks + 1] = (uint32_t)s.extra.size();
        block_extra_ptr[0] = 0;
    }
    s.dev.block_extra_ptr = up.put(block_extra_ptr);
}

// ------------------------------------------------------- systematic errors of the variants' bases (a13 with variants)
// SetSystematicErrorVariantsForward / Reverse (Simulator.cpp:771-909,1011-1147) after the chains: one sequential pass per strand on
// the host.  The reverse function is the mirror image of the forward one, so both are one routine in strand coordinates: position
// sp on the strand (forward position L-1-sp on the reverse strand), bases complemented, variants visited in the strand's order.
// Per variant: last base (the previous variant's last base when that variant sits directly before, else the strand's base before),
// the dominant base from a DominantBaseWithMemory per allele (utilities.hpp:302-351) that lives through variants at most
// kLastX bases apart, the G/C percent of the sys_gc_range_ strand bases before the variant (reference bases only; the reference
// updates or recounts, both give the window count), and the error-region state (distance, start rate) of the strand's chain at
// the variant's position -- folded from the start of the strand over the chain's rates, variants never touch it.  The uniforms:
// words 0, 1 of Philox block (variant index, sequence, 4 + strand, 3<<28 | k) for the k-th base drawn.
struct HostDomMemory {                                                // DominantBaseWithMemory
    uint32_t dom = 0, cnt[5] = {0, 0, 0, 0, 0};
    uint8_t mem[8];
    uint32_t n = 0;
    void find(uint32_t cur_pos) {                                     // DominantBase::FindDominant on memory_
        uint32_t mx = 0;
        for (int b = 4; b--;) mx = std::max(mx, cnt[b]);
        if (0 == mx) dom = (n <= cur_pos || 4 == mem[cur_pos]) ? 0u : mem[cur_pos];
        else {
            uint32_t pos = cur_pos;
            while (mx != cnt[mem[--pos]]) {}
            dom = mem[pos];
        }
    }
    void clear() {
        for (uint32_t &c : cnt) c = 0;
        n = 0;
    }
    template <class At>
    void set(const At &at, uint32_t cur_pos) {
        n = std::min(5u, cur_pos) + 1u;
        for (uint32_t k = n; k--;) mem[k] = (uint8_t)at(cur_pos + k + 1u - n);
        for (uint32_t pos = 0; pos + 1u < n; ++pos) ++cnt[mem[pos]];  // DominantBase::Set(memory_, n - 1): at most kLastX bases before
        find(n - 1u);
    }
    void update(uint32_t base) {
        if (n > 5u + 1u) {
            for (uint32_t k = 1; k < n; ++k) mem[k - 1] = mem[k];
            --n;
        }
        mem[n++] = (uint8_t)base;
        if (1 < n) {                                                  // DominantBase::Update(memory_[n-2], memory_, n-2)
            const uint32_t last_pos = n - 2u;
            ++cnt[mem[last_pos]];
            if (5u <= last_pos) --cnt[mem[last_pos - 5u]];
            find(last_pos + 1u);
        } else find(0);                                               // DominantBase::Set(memory_, 0)
    }
};

// The strand positions [lo, hi) a chain was run over and the chain state it was entered with (a rank of a sharded job runs a part of a
// strand; the whole strand: {0, L, 0}).
struct StrandWindow {
    uint32_t lo, hi, in_state;
};
// track: the strand's chain output (dom | rate << 8 per strand position) from position w.lo on.  Fills var_err_fwd / var_err_rev of the
// sequence's variants inside the window.  What a variant inherits from earlier variants (last base, dominant-base memories) reaches at
// most five positions back, so the pass may begin at any variant that lies more than five positions behind its predecessor: the variants
// between that one and the window only feed the memories.
// `states` (instead of the track): the chain state in front of every variant of the sequence on this strand, in the variants' order (dist | start_rate << 24;
// k_variant_chain_states folds the track on the device from the chunk's entering state, so the track need not come back).
inline void variant_sys_errors_strand(SimState &s, uint32_t seq, bool reverse, const uint16_t *track, StrandWindow w, const uint32_t *states = nullptr) {
    const uint32_t L = s.seq_len[seq], A = s.num_alleles, range = s.dev.sys_gc_range;
    const DevVariant *vars = s.variants.data() + s.var_ptr[seq];
    const uint32_t n = s.var_ptr[seq + 1] - s.var_ptr[seq];
    if (!n) return;
    auto at = [&](uint32_t sp) -> uint32_t { return reverse ? 3u - s.ref_code(seq, L - 1u - sp) : s.ref_code(seq, sp); };
    std::vector<uint16_t> &err = reverse ? s.var_err_rev : s.var_err_fwd;
    std::vector<uint32_t> last_sp(A, 0);
    std::vector<uint8_t> seen(A, 0), last_base_of(A, 4);
    std::vector<HostDomMemory> dom(A);
    uint32_t dist = w.in_state & 0xFFFFFFu, start_rate = w.in_state >> 24, folded = w.lo;      // chain state before strand position `folded`
    auto sp_of = [&](uint32_t k) { return reverse ? L - 1u - vars[n - 1u - k].pos : vars[k].pos; };
    uint32_t k0 = 0;
    {                                                               // the first variant inside the window, then back to a gap of more than five positions
        uint32_t a = 0, b = n;
        while (a < b) {
            const uint32_t mid = (a + b) >> 1;
            if (sp_of(mid) < w.lo) a = mid + 1u;
            else b = mid;
        }
        k0 = a;
        while (k0 > 0 && k0 < n && sp_of(k0 - 1u) + 5u >= sp_of(k0)) --k0;
        if (k0 == n) return;
    }
    for (uint32_t k = k0; k < n; ++k) {
        const uint32_t var_id = reverse ? n - 1u - k : k;
        const DevVariant &v = vars[var_id];
        const uint32_t sp = reverse ? L - 1u - v.pos : v.pos, len = v.len;
        if (sp >= w.hi) break;
        const bool inside = sp >= w.lo;                            // in front of the window: the memories only
        auto var_base = [&](uint32_t j) -> uint32_t { return reverse ? 3u - s.var_bases[v.off + len - 1u - j] : s.var_bases[v.off + j]; };   // j-th base in the strand's order
        uint32_t chosen = 0;                                        // Variant::FirstAllele
        while (chosen < A && !((v.allele[chosen >> 6] >> (chosen & 63u)) & 1u)) ++chosen;
        if (chosen == A) throw Error("variant without an allele");
        uint32_t last_base;
        if (seen[chosen] && last_sp[chosen] + 1u == sp) last_base = last_base_of[chosen];
        else last_base = sp ? at(sp - 1u) : 4u;
        if (seen[chosen] && last_sp[chosen] + 5u >= sp) {
            for (uint32_t q = last_sp[chosen] + 1u; q < sp; ++q) dom[chosen].update(at(q));
        } else {
            dom[chosen].clear();
            if (sp) dom[chosen].set(at, sp - 1u);
        }
        if (inside && states) {
            dist = states[var_id] & 0xFFFFFFu;
            start_rate = states[var_id] >> 24;
        } else
            for (; inside && folded < sp; ++folded) update_distances(s.dev.reset_distance, dist, start_rate, track[folded - w.lo] >> 8);
        const uint32_t gc_bases = std::min(sp, range);
        uint32_t gc = 0;
        for (uint32_t q = sp - gc_bases; inside && q < sp; ++q) gc += is_gc(at(q));
        const uint32_t idx[3] = {transform_distance(dist), safe_percent_u16(gc, gc_bases), start_rate};      // variants cannot start an error region: the same for all their bases
        for (uint32_t j = 0; j < len; ++j) {
            const uint32_t base = var_base(j);
            dom[chosen].update(base);
            if (inside) {
                const Words rw = philox(s.seed, var_id, seq, 4u + (reverse ? 1u : 0u), (kDomSysErr << 28) | j);
                double ps;
                uint32_t dom_error = draw<3>(s.host_dom_error[(base * 5u + last_base) * 5u + dom[chosen].dom], s.host_pool.data(), s.host_par0.data(), idx, u32_to_unit(rw.w0), ps);
                if (0.0 == ps) dom_error = 4;
                uint32_t rate = draw<3>(s.host_error_rate[base * 5u + dom_error], s.host_pool.data(), s.host_par0.data(), idx, u32_to_unit(rw.w1), ps);
                if (0.0 == ps) rate = 0;
                err[v.off + j] = (uint16_t)(dom_error | (rate << 8));
            }
            last_base = base;
        }
        // the other alleles of the variant: their dominant-base memories (Simulator.cpp:1088-1126 / 849-887)
        uint32_t ref_allele = A;
        if (!seen[chosen] || last_sp[chosen] + 5u < sp + len) ref_allele = chosen;
        for (uint32_t allele = 0; allele < A; ++allele) {
            if (!((v.allele[allele >> 6] >> (allele & 63u)) & 1u)) continue;
            if (allele != chosen) {
                if (seen[allele] && last_sp[allele] + 5u >= sp + len) {
                    for (uint32_t q = last_sp[allele] + 1u; q < sp; ++q) dom[allele].update(at(q));
                    for (uint32_t j = 0; j < len; ++j) dom[allele].update(var_base(j));
                } else if (ref_allele < A) dom[allele] = dom[ref_allele];
                else {
                    ref_allele = allele;
                    dom[allele].clear();
                    if (sp) dom[allele].set(at, sp - 1u);
                    for (uint32_t j = 0; j < len; ++j) dom[allele].update(var_base(j));
                }
            }
            seen[allele] = 1;
            last_sp[allele] = sp;
            last_base_of[allele] = (uint8_t)last_base;
        }
    }
}

// both strands of every simulated sequence (CreateUnit: the reverse strand first); the tracks come back from the device.
// `windows` (a sharded job): the parts of the strands the rank's chains were run over, instead of whole strands.
struct StrandTask {
    uint32_t seq;
    int strand;
    StrandWindow w;
};
// states_fwd / states_rev: per variant (s.variants' order) the chain state in front of it on either strand, when the caller has them; otherwise the tracks are read
inline void build_variant_sys_errors(SimState &s, Uploader &up, const std::vector<StrandTask> *windows = nullptr, const uint32_t *states_fwd = nullptr,
                                     const uint32_t *states_rev = nullptr) {
    if (!s.has_variants || s.variants.empty()) return;
    // (sequence, strand) tasks are independent (own variants, own error arrays): a few host threads share them, longest first
    std::vector<StrandTask> tasks;
    if (windows) {
        for (const StrandTask &t : *windows)
            if (s.var_ptr[t.seq] != s.var_ptr[t.seq + 1]) tasks.push_back(t);
    } else
        for (uint32_t seq = 0; seq < s.dev.n_seqs; ++seq)
            if (s.n_blocks[seq] && s.var_ptr[seq] != s.var_ptr[seq + 1])
                for (int strand = 2; strand--;) tasks.push_back(StrandTask{seq, strand, StrandWindow{0u, s.seq_len[seq], 0u}});
    std::stable_sort(tasks.begin(), tasks.end(), [&](const StrandTask &a, const StrandTask &b) { return a.w.hi - a.w.lo > b.w.hi - b.w.lo; });
    std::atomic<size_t> next{0};
    std::mutex err_mutex;
    std::string error;
    auto work = [&](bool helper) {
        try {
            if (helper) up.bind_thread();
            std::vector<uint16_t> track;
            for (size_t t; (t = next.fetch_add(1)) < tasks.size();) {
                const uint32_t seq = tasks[t].seq;
                const int strand = tasks[t].strand;
                const StrandWindow w = tasks[t].w;
                const uint32_t *states = strand ? states_rev : states_fwd;
                if (!states) {
                    track.resize(w.hi - w.lo);
                    up.read_bytes(track.data(), (strand ? s.sys_rev : s.sys_fwd) + s.seq_base_off[seq] + w.lo, track.size() * sizeof(uint16_t));
                }
                variant_sys_errors_strand(s, seq, strand != 0, track.data(), w, states ? states + s.var_ptr[seq] : nullptr);
            }
        } catch (const std::exception &e) {
            std::lock_guard<std::mutex> lock(err_mutex);
            error = e.what();
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t n_threads = std::min<size_t>(tasks.size(), std::max(1u, std::min(hw ? hw : 1u, 64u)));
    std::vector<std::thread> helpers;
    for (size_t t = 1; t < n_threads; ++t) helpers.emplace_back(work, true);
    work(false);
    for (std::thread &t : helpers) t.join();
    if (!error.empty()) throw Error(error);
    up.write_bytes(s.dev_var_err_fwd, s.var_err_fwd.data(), s.var_err_fwd.size() * sizeof(uint16_t));
    up.write_bytes(s.dev_var_err_rev, s.var_err_rev.data(), s.var_err_rev.size() * sizeof(uint16_t));
}

constexpr const char *kWalkErrorMessage =
    "systematic-error walk left the sequence: with variants this close together at a sequence end the reference follows a NULL block "
    "(GetSysErrorFromBlock, Simulator.cpp:232-292); such a variant set cannot be simulated";

// ------------------------------------------------------------------------------- chains of the a13 pre-pass
// Positions per chunk of a chain (one lane each) and the run-up in front of a chunk in pass 0 (rsq_kernels.h, "Speculative chunking").  The chunks of a pass are
// independent, so they only have to be many enough to fill the device (2 M: eight waves on each of its 1024 SIMDs twice over); beyond that longer chunks make
// the run-up -- work that is thrown away -- a smaller share: 256 positions up to 0.5 Gb of strands, 4096 for a human-sized reference (6.2 Gb, 1.5 M chunks).  The
// run-up is half a chunk, at most 384 positions (three times the distance within which two runs were measured to meet; on the human-sized reference, chunk
// length : run-up -> seconds of the chains: 1024:384 1.38, 2048:384 1.19, 2048:768 1.34, 4096:384 1.10, 4096:768 1.17, 4096:1536 1.33, 8192:768 1.14, 16384:1024 1.16; 256:0 was
// 1.86).  Ranks of a sharded job derive the same values from the same reference.  Options chain_chunk / chain_warmup override them (tests, measurements).
inline uint32_t chain_chunk_len(uint64_t total_ref_size, const Options &opt) {
    if (opt.chain_chunk > 0) return (uint32_t)opt.chain_chunk;
    uint32_t len = 256;
    while (len < 4096 && 2 * total_ref_size / len > (2u << 20)) len *= 2;
    return len;
}
inline uint32_t chain_warmup_len(uint32_t chunk_len, const Options &opt) {
    if (opt.chain_warmup >= 0) return (uint32_t)opt.chain_warmup;
    return std::min(chunk_len / 2, 384u);
}
// kChainsAdapters: SimulateErrorModelOnly (Simulator.cpp:2951-2977); kChainsSimulation: Simulate (adapters, then every sequence that
// gets a unit); kChainsProfile: CreateSystematicErrorProfile (:2597-2653), every sequence and no adapters, from a fresh Simulator.
enum ChainSet : int { kChainsAdapters = 0, kChainsSimulation = 1, kChainsProfile = 2 };

// ---- a sharded job: what the rank that simulates blocks [block_lo, block_hi) has to compute of the pre-passes.
// Per sequence the rank's start positions [p_lo, p_hi) and the positions its reads can touch [p_lo, t_hi): a fragment ends before
// start + insert_to (Simulator.cpp:2303,2316).  [g_lo, g_hi): the rank's share of the concatenated sequences, for the bias sums.
// How far behind its start position a fragment's reads can touch the sequence: a fragment ends before start + insert_to
// (Simulator.cpp:2303,2316); with variants its span on the reference grows by the bases its allele deletes, and the systematic-error walk
// of a read may run a few positions past its template (GetSysErrorFromBlock with variants), hence the read length on top.
inline uint32_t shard_halo(const SimState &s) {
    uint32_t halo = s.dev.insert_to + 16u;
    if (!s.has_variants) return halo;
    // the most bases one allele deletes inside any stretch of `span` reference positions, with span = insert_to + that number (iterated)
    uint32_t deleted = 0;
    for (int round = 0; round < 4; ++round) {
        const uint64_t span = (uint64_t)s.dev.insert_to + deleted;
        uint32_t most = 0;
        for (size_t m = 0; m + 1 < s.allele_map_ptr.size(); ++m) {
            const uint32_t seq = (uint32_t)(m / s.num_alleles);
            const DevVariant *vars = s.variants.data() + s.var_ptr[seq];
            std::vector<uint32_t> dels;
            for (uint32_t i = s.allele_map_ptr[m]; i + 1 < s.allele_map_ptr[m + 1]; ++i)
                if (0 == vars[s.allele_map[i].vid].len) dels.push_back(s.allele_map[i].pos);
            for (size_t a = 0, b = 0; b < dels.size(); ++b) {
                while (dels[b] - dels[a] >= span) ++a;
                most = std::max(most, (uint32_t)(b - a + 1));
            }
        }
        if (most <= deleted) break;
        deleted = most;
    }
    return halo + deleted + s.rmax + 64u;
}
struct ShardRange {
    uint32_t halo = 0;
    std::vector<uint32_t> p_lo, p_hi, t_hi;
    uint64_t g_lo = 0, g_hi = UINT64_MAX;
    int first_seq = -1, last_seq = -1;      // the sequences of the rank's first and last start position (-1: no blocks)
};
inline ShardRange shard_range(const SimState &s, uint32_t block_lo, uint32_t block_hi) {
    if (block_lo < 1 || block_hi > s.total_blocks + 1 || block_lo > block_hi) throw Error("block range outside [1, total_blocks]");
    const uint32_t n_seqs = s.dev.n_seqs, halo = shard_halo(s);
    ShardRange r;
    r.halo = halo;
    r.p_lo.assign(n_seqs, 0);
    r.p_hi.assign(n_seqs, 0);
    r.t_hi.assign(n_seqs, 0);
    auto global_of_block = [&](uint32_t b) { return s.seq_base_off[s.block_seq[b]] + (uint64_t)(b - s.first_block[s.block_seq[b]]) * kBlockSize; };
    r.g_lo = block_lo <= 1 ? 0 : global_of_block(block_lo);
    r.g_hi = block_hi > s.total_blocks ? UINT64_MAX : global_of_block(block_hi);
    for (uint32_t i = 0; i < n_seqs; ++i) {
        if (!s.n_blocks[i]) continue;
        const uint32_t fb = s.first_block[i], lo = std::max(fb, block_lo), hi = std::min(fb + s.n_blocks[i], block_hi);
        if (lo >= hi) continue;
        const uint32_t L = s.seq_len[i];
        r.p_lo[i] = (lo - fb) * kBlockSize;
        r.p_hi[i] = (uint32_t)std::min<uint64_t>(L, (uint64_t)(hi - fb) * kBlockSize);
        r.t_hi[i] = (uint32_t)std::min<uint64_t>(L, (uint64_t)r.p_hi[i] + halo);
        if (r.first_seq < 0) r.first_seq = (int)i;
        r.last_seq = (int)i;
    }
    return r;
}
// where a rank's chains meet its neighbours': the chains that are entered with a state of the neighbouring rank, and the chunks
// (flat indices) whose outgoing state the neighbours need.  -1: none
struct ShardEdges {
    int fwd_in_chain = -1, rev_in_chain = -1;       // forward chain of the rank's first sequence (state from the left neighbour), reverse chain of its last (from the right)
    int64_t fwd_out_chunk = -1, rev_out_chunk = -1; // for the right neighbour's forward chain / the left neighbour's reverse chain
};

// `range`: only the chunks of the rank's share (every chunk when null); `edges` is filled with them
inline void build_chains(const SimState &s, ChainSet set, std::vector<Chain> &chains, std::vector<uint32_t> &chunk_chain, const ShardRange *range = nullptr,
                         ShardEdges *edges = nullptr) {
    uint32_t dom_state = 0;                                       // DominantBase(): dom_base_(0)
    auto add = [&](Chain c, uint32_t pos_lo, uint32_t pos_hi) {  // chain positions [pos_lo, pos_hi) are needed
        c.first_chunk = (uint32_t)chunk_chain.size();
        c.initial_dom = dom_state;
        c.chunk_lo = pos_lo / s.chain_chunk;
        c.in_state = 0;
        for (uint32_t k = c.chunk_lo; k < cdiv(pos_hi, s.chain_chunk); ++k) chunk_chain.push_back((uint32_t)chains.size());
        chains.push_back(c);
    };
    const Profile &p = s.prof;
    for (int seg = 2; set != kChainsProfile && seg--;) {           // Simulator.cpp:2784-2797 adapters, segment 1 first, ids descending
        const HostAdapters &a = p.adapters[seg];
        for (uint32_t i = a.n(); i--;) {
            if (!a.counts[i]) continue;
            const uint32_t len = a.seq_ptr[i + 1] - a.seq_ptr[i];
            add(Chain{2u, i, (uint32_t)seg, len, i, 2u + (uint32_t)seg, 0, 0, s.adapter_sys[seg] + a.seq_ptr[i]}, 0, len);
            const uint8_t *codes = a.seqs.data() + a.seq_ptr[i];
            dom_state = dom_base_after_chain([&](uint32_t pos) { return (uint32_t)codes[pos]; }, len, dom_state);
        }
    }
    if (set != kChainsAdapters)
        for (uint32_t i = 0; i < s.dev.n_seqs; ++i) {
            if (set == kChainsSimulation && !s.n_blocks[i]) continue;      // no unit for sequences shorter than the longest insert
            const uint32_t L = s.seq_len[i];
            const bool mine = !range || range->p_lo[i] < range->p_hi[i];
            const uint32_t f_lo = range ? range->p_lo[i] : 0u, f_hi = range ? range->t_hi[i] : L;       // forward positions the rank needs
            for (uint32_t strand = 2; strand--;) {                  // CreateUnit: whole reverse strand first, then the forward blocks
                if (mine) {
                    const int index = (int)chains.size();
                    if (strand) add(Chain{strand, i, 0u, L, i, strand, 0, 0, s.sys_rev + s.seq_base_off[i]}, L - f_hi, L - f_lo);       // chain position = L-1-forward position
                    else add(Chain{strand, i, 0u, L, i, strand, 0, 0, s.sys_fwd + s.seq_base_off[i]}, f_lo, f_hi);
                    if (range && edges) {
                        const Chain &c = chains.back();
                        const uint32_t halo = range->halo;
                        if (!strand && (int)i == range->first_seq && c.chunk_lo) edges->fwd_in_chain = index;
                        if (strand && (int)i == range->last_seq && c.chunk_lo) edges->rev_in_chain = index;
                        if (!strand && (int)i == range->last_seq && range->p_hi[i] < L && range->p_hi[i] / s.chain_chunk)      // the right neighbour starts at p_hi
                            edges->fwd_out_chunk = (int64_t)c.first_chunk + (range->p_hi[i] / s.chain_chunk - 1u - c.chunk_lo);
                        if (strand && (int)i == range->first_seq && range->p_lo[i]) {                                           // the left neighbour ends at p_lo
                            const uint32_t t_hi = (uint32_t)std::min<uint64_t>(L, (uint64_t)range->p_lo[i] + halo), lo_chunk = (L - t_hi) / s.chain_chunk;
                            if (lo_chunk) edges->rev_out_chunk = (int64_t)c.first_chunk + (lo_chunk - 1u - c.chunk_lo);
                        }
                    }
                }
                if (strand) dom_state = dom_base_after_chain([&](uint32_t pos) { return 3u - s.ref_code(i, L - 1 - pos); }, L, dom_state);
                else dom_state = dom_base_after_chain([&](uint32_t pos) { return s.ref_code(i, pos); }, L, dom_state);
            }
        }
}

// the strand windows of a finished chain run (the reference chains among `chains`; n_chunks = all chunks of the run), each with the
// state its chain was entered with -- what build_variant_sys_errors needs of a rank's share.  Long strands are cut into windows of
// kWindowChunks chunks, each entered with the state the fixed point left in front of its first chunk (entering_state(flat chunk index)),
// so that the host pass over the variants has many independent tasks.
constexpr uint32_t kWindowChunks = 32768;                          // 8.4 M positions
inline uint32_t window_chunks(const Options &opt) {                // option window_chunks: smaller windows, so that tests on short sequences cut strands too
    const int64_t v = opt.window_chunks;
    return v > 0 ? (uint32_t)v : kWindowChunks;
}
template <class EnteringState>
inline std::vector<StrandTask> strand_tasks(const Options &opt, const std::vector<Chain> &chains, uint32_t n_chunks, uint32_t chunk_len, EnteringState &&entering_state) {
    std::vector<StrandTask> out;
    for (size_t c = 0; c < chains.size(); ++c) {
        const Chain &ch = chains[c];
        if (ch.kind > 1u) continue;
        const uint32_t chunks = (c + 1 < chains.size() ? chains[c + 1].first_chunk : n_chunks) - ch.first_chunk;
        const uint32_t per_window = window_chunks(opt);
        for (uint32_t first = 0; first < chunks; first += per_window) {
            const uint32_t count = std::min(per_window, chunks - first);
            const uint32_t lo = (ch.chunk_lo + first) * chunk_len, hi = (uint32_t)std::min<uint64_t>(ch.len, (uint64_t)(ch.chunk_lo + first + count) * chunk_len);
            out.push_back(StrandTask{ch.id, (int)ch.kind, StrandWindow{lo, hi, first ? entering_state(ch.first_chunk + first) : ch.in_state}});
        }
    }
    return out;
}

// --methylation (Reference::PrepareMethylationFile / ReadMethylation, Simulator.cpp:2770-2780): regions as CSR on the device
// the regions as the kernels read them (CSR over the sequences; num_alleles rates per region) -> device; also the route of regions another process parsed
inline void install_methylation(SimState &s, Uploader &up, std::vector<uint32_t> ptr, std::vector<uint32_t> first, std::vector<uint32_t> second, std::vector<double> rate) {
    if (s.has_variants && 2 != s.variants_mode) {
        // with methylation the templates are written out and converted per mate (k_variant_templates): the path for variants of any
        // kind.  A substitution-only set has no starts inside inserted bases, so nothing else changes.
        s.variants_mode = 2;
        s.dev.variants_loaded = 2;
    }
    s.dev.meth_ptr = up.put(ptr);
    s.dev.meth_first = up.put(first);
    s.dev.meth_second = up.put(second);
    s.dev.meth_rate = up.put(rate);
    s.template_words = (s.rmax + s.prof.max_len_deletion + 31u) / 32u + 1u;          // GetOrgSeq: at most Rmax + MaxLenDeletion bases
    if (s.template_words > kTemplateWordsMax) throw Error("templates longer than 2048 bases are not supported with --methylation");
    s.has_methylation = true;
    s.meth_ptr_host = std::move(ptr);
    s.meth_first_host = std::move(first);
    s.meth_second_host = std::move(second);
    s.meth_rate_host = std::move(rate);
}
inline void pack_methylation(SimState &s, Uploader &up, const Methylation &m) {
    std::vector<uint32_t> ptr{0}, first, second;
    std::vector<double> rate;
    for (size_t i = 0; i < m.first.size(); ++i) {
        first.insert(first.end(), m.first[i].begin(), m.first[i].end());
        second.insert(second.end(), m.second[i].begin(), m.second[i].end());
        for (size_t k = 0; k < m.first[i].size(); ++k)                                                   // num_alleles values per region: Reference::Unmethylation
            for (uint32_t a = 0; a < s.num_alleles; ++a) rate.push_back(m.rate[i][1 < m.rate[i].size() ? a : 0][k]);
        ptr.push_back((uint32_t)first.size());
    }
    install_methylation(s, up, std::move(ptr), std::move(first), std::move(second), std::move(rate));
}

// ------------------------------------------------------------------------------------------ one load per host
// Eight ranks on one host used to read and pack the same FASTA, VCF and BED eight times, sharing the host's cores (the job's longest stage at human scale).  What a
// simulator keeps of those files is the result of pack_reference / pack_methylation; export_reference writes exactly that -- sequence tables, the 2-bit words and
// G/C prefix sums (of every allele's copy too), variants, allele maps, extra starts, methylation regions -- into one file (the launcher puts it into /dev/shm), and
// import_reference gives another process's simulator the same state through the same upload_reference / install_methylation: ONE rank of a host parses and packs,
// the others map its result.  The file names what the packing depended on (longest insert, longest read, longest deletion of the profile): an importer with another
// profile is refused.  Layout: "RSQREF1\0", then records {u32 tag, u32 element size, u64 count, bytes padded to 8}.
namespace refio {
enum Tag : uint32_t {
    kScalars = 1, kNames, kNamePtr, kIds, kIdPtr, kSeqLen, kWords, kGcPrefix, kVariants, kVarPtr, kVarBases, kAlleleMap, kAlleleMapPtr, kExtra, kExtraSeqPtr, kMethPtr, kMethFirst,
    kMethSecond, kMethRate, kEnd
};
struct Scalars {
    uint64_t hap_stride, total_ref_size;
    uint32_t n_seqs, has_variants, num_alleles, variants_mode_packed, has_methylation, insert_to, rmax, max_len_deletion;
};
struct Writer {
    FILE *f;
    template <class T>
    void put(uint32_t tag, const T *data, uint64_t count) {
        const uint32_t head[2] = {tag, (uint32_t)sizeof(T)};
        static const char zeros[8] = {0};
        const size_t bytes = (size_t)count * sizeof(T);
        if (fwrite(head, 1, 8, f) != 8 || fwrite(&count, 1, 8, f) != 8 || (bytes && fwrite(data, 1, bytes, f) != bytes) || fwrite(zeros, 1, (8 - bytes % 8) % 8, f) != (8 - bytes % 8) % 8)
            throw Error("writing the packed reference failed");
    }
    template <class T>
    void put(uint32_t tag, const std::vector<T> &v) { put(tag, v.data(), v.size()); }
};
struct Record {
    const char *data = nullptr;
    uint64_t count = 0;
    uint32_t size = 0;
};
}  // namespace refio

inline void export_reference(const SimState &s, Uploader &up, const std::string &path) {
    if (!s.has_ref) throw Error("the simulator has no reference to export");
    using namespace refio;
    const std::string tmp = path + ".writing";
    // the launcher puts the file into /dev/shm, which anyone may write: never through a link someone else left there, never over an existing file, and readable by the owner only
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    FILE *f = fd < 0 ? nullptr : fdopen(fd, "wb");
    if (!f) {
        if (fd >= 0) close(fd);
        throw Error("cannot write " + tmp + ": " + strerror(errno));
    }
    try {
        if (fwrite("RSQREF1", 1, 8, f) != 8) throw Error("writing the packed reference failed");
        Writer w{f};
        const Scalars sc{s.dev.hap_stride, s.total_ref_size, s.dev.n_seqs, s.has_variants ? 1u : 0u, s.num_alleles, (uint32_t)s.variants_mode_packed, s.has_methylation ? 1u : 0u,
                         s.dev.insert_to, s.rmax, (uint32_t)s.prof.max_len_deletion};
        w.put(kScalars, &sc, 1);
        std::string names, ids;
        std::vector<uint64_t> name_ptr{0}, id_ptr{0};
        for (size_t i = 0; i < s.ref_first_names.size(); ++i) {
            names += s.ref_first_names[i];
            ids += s.ref_ids[i];
            name_ptr.push_back(names.size());
            id_ptr.push_back(ids.size());
        }
        w.put(kNames, names.data(), names.size());
        w.put(kNamePtr, name_ptr);
        w.put(kIds, ids.data(), ids.size());
        w.put(kIdPtr, id_ptr);
        w.put(kSeqLen, s.seq_len);
        uint64_t words = 0;
        for (uint32_t len : s.seq_len) words += (len + 31) / 32 + 1;
        const uint64_t copies = s.dev.hap_stride ? 1u + s.num_alleles : 1u, n_words = s.dev.hap_stride ? s.dev.hap_stride * copies : words + 1;
        {   // the packed words and the G/C prefix sums live where the kernels read them (the host keeps the reference's own words only)
            std::vector<uint64_t> packed(n_words);
            up.read_bytes(packed.data(), s.dev.ref_words, n_words * 8);
            w.put(kWords, packed);
        }
        {
            std::vector<uint32_t> gc(n_words);
            up.read_bytes(gc.data(), s.dev.gc_prefix, n_words * 4);
            w.put(kGcPrefix, gc);
        }
        w.put(kVariants, s.variants);
        w.put(kVarPtr, s.var_ptr);
        w.put(kVarBases, s.var_bases);
        w.put(kAlleleMap, s.allele_map);
        w.put(kAlleleMapPtr, s.allele_map_ptr);
        w.put(kExtra, s.extra);
        w.put(kExtraSeqPtr, s.extra_seq_ptr);
        if (s.has_methylation) {
            w.put(kMethPtr, s.meth_ptr_host);
            w.put(kMethFirst, s.meth_first_host);
            w.put(kMethSecond, s.meth_second_host);
            w.put(kMethRate, s.meth_rate_host);
        }
        w.put(kEnd, (const char *)nullptr, 0);
    } catch (...) {
        fclose(f);
        remove(tmp.c_str());
        throw;
    }
    if (fclose(f) != 0 || rename(tmp.c_str(), path.c_str()) != 0) {      // complete, then visible under its name
        remove(tmp.c_str());
        throw Error("writing " + path + " failed");
    }
}

// `data`, `size`: the file (mapped by the caller, alive during the call).  The simulator must have been created without a reference, for the same profile.
inline void import_reference(SimState &s, Uploader &up, const char *data, size_t size, const std::string &what) {
    using namespace refio;
    if (s.has_ref) throw Error("the simulator has a reference already");
    if (size < 8 || memcmp(data, "RSQREF1", 8)) throw Error(what + " is not a packed reference of this library");
    Record rec[kEnd + 1];
    bool complete = false;
    for (size_t at = 8; at + 16 <= size;) {
        uint32_t head[2];
        uint64_t count;
        memcpy(head, data + at, 8);
        memcpy(&count, data + at + 8, 8);
        at += 16;
        if (head[0] == kEnd) {
            complete = true;
            break;
        }
        if (head[0] == 0 || head[0] > kEnd || head[1] == 0 || count > (size - at) / head[1]) throw Error(what + ": damaged record");      // no product that could wrap
        const uint64_t bytes = count * head[1];
        rec[head[0]] = Record{data + at, count, head[1]};
        at += (bytes + 7) / 8 * 8;
    }
    if (!complete) throw Error(what + " is incomplete");
    auto take = [&](uint32_t tag, auto &vec) {
        using T = typename std::remove_reference_t<decltype(vec)>::value_type;
        if (rec[tag].size != sizeof(T) && rec[tag].count) throw Error(what + ": record " + std::to_string(tag) + " has another element size");
        vec.resize(rec[tag].count);
        if (rec[tag].count) memcpy(vec.data(), rec[tag].data, rec[tag].count * sizeof(T));
    };
    if (rec[kScalars].count != 1 || rec[kScalars].size != sizeof(Scalars)) throw Error(what + ": no header record");
    Scalars sc;
    memcpy(&sc, rec[kScalars].data, sizeof sc);
    if (sc.insert_to != s.dev.insert_to || sc.rmax != s.rmax || sc.max_len_deletion != (uint32_t)s.prof.max_len_deletion)
        throw Error(what + " was packed for another profile (longest insert / read / deletion " + std::to_string(sc.insert_to) + " / " + std::to_string(sc.rmax) + " / " +
                    std::to_string(sc.max_len_deletion) + ", this simulator's " + std::to_string(s.dev.insert_to) + " / " + std::to_string(s.rmax) + " / " + std::to_string(s.prof.max_len_deletion) + ")");
    DevSim &d = s.dev;
    std::vector<uint64_t> name_ptr, id_ptr;
    take(kNamePtr, name_ptr);
    take(kIdPtr, id_ptr);
    take(kSeqLen, s.seq_len);
    // an offset array of n + 1 entries over `total` items: from 0, non-decreasing, ending at total
    auto offsets_fit = [](const auto &ptr, size_t n, uint64_t total) {
        if (ptr.size() != n + 1 || ptr.front() != 0 || ptr.back() != total) return false;
        for (size_t i = 0; i < n; ++i)
            if (ptr[i] > ptr[i + 1]) return false;
        return true;
    };
    if (s.seq_len.size() != sc.n_seqs || rec[kNames].size > 1 || rec[kIds].size > 1 || !offsets_fit(name_ptr, sc.n_seqs, rec[kNames].count) || !offsets_fit(id_ptr, sc.n_seqs, rec[kIds].count))
        throw Error(what + ": sequence tables do not fit together");
    s.ref_first_names.clear();
    s.ref_ids.clear();
    s.seq_word_off.clear();
    s.seq_base_off.clear();
    uint64_t words = 0, bases = 0;
    for (uint32_t i = 0; i < sc.n_seqs; ++i) {
        s.ref_first_names.emplace_back(rec[kNames].data + name_ptr[i], name_ptr[i + 1] - name_ptr[i]);
        s.ref_ids.emplace_back(rec[kIds].data + id_ptr[i], id_ptr[i + 1] - id_ptr[i]);
        s.seq_word_off.push_back(words);
        s.seq_base_off.push_back(bases);
        words += (s.seq_len[i] + 31) / 32 + 1;
        bases += s.seq_len[i];
    }
    if (bases != sc.total_ref_size) throw Error(what + ": sequence lengths do not add up");
    d.n_seqs = sc.n_seqs;
    s.has_ref = true;
    s.total_ref_size = bases;
    s.has_variants = sc.has_variants != 0;
    s.num_alleles = sc.num_alleles;
    d.num_alleles = sc.num_alleles;
    d.hap_stride = sc.hap_stride;
    s.variants_mode = s.variants_mode_packed = (int)sc.variants_mode_packed;
    d.variants_loaded = sc.variants_mode_packed;
    take(kVariants, s.variants);
    take(kVarPtr, s.var_ptr);
    take(kVarBases, s.var_bases);
    take(kAlleleMap, s.allele_map);
    take(kAlleleMapPtr, s.allele_map_ptr);
    take(kExtra, s.extra);
    take(kExtraSeqPtr, s.extra_seq_ptr);
    std::vector<uint64_t> packed;
    std::vector<uint32_t> gc_prefix;
    take(kWords, packed);
    take(kGcPrefix, gc_prefix);
    const uint64_t n_words = sc.hap_stride ? sc.hap_stride * (1u + sc.num_alleles) : words + 1;
    if (packed.size() != n_words || gc_prefix.size() != n_words || (sc.hap_stride && sc.hap_stride < words + 1)) throw Error(what + ": arrays of unexpected sizes");
    // what the kernels index with: every offset array over its payload, every variant inside its sequence and the pool of its bases, every map entry a variant of its sequence
    if (sc.num_alleles == 0 || sc.num_alleles > 128u || sc.variants_mode_packed > 2u || !offsets_fit(s.var_ptr, sc.n_seqs, s.variants.size()) ||
        !offsets_fit(s.extra_seq_ptr, sc.n_seqs, s.extra.size()) || !offsets_fit(s.allele_map_ptr, sc.has_variants ? (size_t)sc.n_seqs * sc.num_alleles : 0, s.allele_map.size()))
        throw Error(what + ": variant tables do not fit together");
    for (uint32_t i = 0; i < sc.n_seqs; ++i) {
        const uint32_t n_var = s.var_ptr[i + 1] - s.var_ptr[i];
        for (uint64_t v = s.var_ptr[i]; v < s.var_ptr[i + 1]; ++v) {
            const DevVariant &var = s.variants[v];
            if (var.pos >= s.seq_len[i] || var.off > s.var_bases.size() || var.len > s.var_bases.size() - var.off || (v > s.var_ptr[i] && s.variants[v - 1].pos > var.pos))
                throw Error(what + ": a variant outside its sequence or the pool of variant bases");
        }
        for (uint64_t e = s.extra_seq_ptr[i]; e < s.extra_seq_ptr[i + 1]; ++e)
            if (s.extra[e].pos >= s.seq_len[i] || (s.extra[e].first_variant_id >= 0 && (uint32_t)s.extra[e].first_variant_id >= n_var)) throw Error(what + ": an extra start outside its sequence");
        if (sc.has_variants)
            for (uint64_t m = s.allele_map_ptr[(size_t)i * sc.num_alleles]; m < s.allele_map_ptr[(size_t)(i + 1) * sc.num_alleles]; ++m)
                if (s.allele_map[m].pos > s.seq_len[i] || s.allele_map[m].vid > n_var) throw Error(what + ": an allele map entry outside its sequence");
    }
    for (uint8_t b : s.var_bases)
        if (b > 4u) throw Error(what + ": a variant base that is no base");
    std::vector<uint32_t> meth_ptr, meth_first, meth_second;
    std::vector<double> meth_rate;
    if (sc.has_methylation) {
        take(kMethPtr, meth_ptr);
        take(kMethFirst, meth_first);
        take(kMethSecond, meth_second);
        take(kMethRate, meth_rate);
        if (!offsets_fit(meth_ptr, sc.n_seqs, meth_first.size()) || meth_second.size() != meth_first.size() || meth_rate.size() != meth_first.size() * (size_t)sc.num_alleles)
            throw Error(what + ": methylation tables do not fit together");
        for (uint32_t i = 0; i < sc.n_seqs; ++i)
            for (uint32_t k = meth_ptr[i]; k < meth_ptr[i + 1]; ++k)
                if (meth_first[k] > meth_second[k] || meth_second[k] > s.seq_len[i]) throw Error(what + ": a methylation region outside its sequence");
    }
    upload_reference(s, up, std::move(packed), gc_prefix);
    if (sc.has_methylation) install_methylation(s, up, std::move(meth_ptr), std::move(meth_first), std::move(meth_second), std::move(meth_rate));
}

// --readSysError: LoadSysErrorRecord (Simulator.cpp:750-769) + ReadSystematicErrors (Simulator.h:326-335).  Units consume the
// records in file order, two per unit (reverse strand first); sequences without a unit consume none, exactly like the reference
// (so a file written for a reference with too-short sequences is rejected with the length message).
inline void apply_sys_error_records(SimState &s, Uploader &up, const std::vector<SysErrorRecord> &recs) {
    size_t next = 0;
    for (uint32_t i = 0; i < s.dev.n_seqs; ++i) {
        if (!s.n_blocks[i]) continue;
        for (uint32_t strand = 2; strand--;) {
            if (next >= recs.size()) throw Error("Could not read systematic error profile for reference sequence '" + s.ref_first_names[i] + "': end of file");
            const SysErrorRecord &r = recs[next++];
            if (r.dom.size() != s.seq_len[i])
                throw Error("Systematic error profile '" + r.id + "' (length " + std::to_string(r.dom.size()) + ") does not match reference sequence '" +
                            s.ref_first_names[i] + "' (length " + std::to_string(s.seq_len[i]) + "). Wrong file or order incorrect?");
            std::vector<uint16_t> track(r.dom.size());
            for (size_t k = 0; k < track.size(); ++k) track[k] = (uint16_t)(r.dom[k] | ((uint16_t)r.rate[k] << 8));
            up.write_bytes((strand ? s.sys_rev : s.sys_fwd) + s.seq_base_off[i], track.data(), track.size() * sizeof(uint16_t));
        }
    }
}

// ------------------------------------------------- bias normalisation: parameter list and the arithmetic after SumBias
// FragmentDistributionStats.cpp:3504-3582 CalculateBiasNormalization, split around the SumBias scan.
struct BiasPlan {
    std::vector<uint32_t> sample;           // insert_length_spline.sample_positions_
    std::vector<BiasParam> params;          // FillParamsSimulation order (:2185-2200)
    uint32_t max_starts = 0;
    // the bias sums run in chunks of kBiasBlock * kBiasRun start positions: chunks [chunk_ptr[i], chunk_ptr[i + 1]) belong to params[i]
    std::vector<uint32_t> chunk_ptr, chunk_param;
};

inline BiasPlan plan_bias_normalization(SimState &s, Uploader &up) {
    const ScopedUpload scope(up, kScopePlan, false);
    const Profile &p = s.prof;
    const uint32_t n_seqs = s.dev.n_seqs;
    BiasPlan plan;
    plan.sample = sample_positions(p.insert_lengths);
    std::vector<std::pair<double, uint32_t>> sorted;                                 // :2909-2932 SplitCoverageGroups
    for (uint32_t i = n_seqs; i--;) sorted.emplace_back(s.ref_seq_bias[i], i);
    std::sort(sorted.begin(), sorted.end());
    s.coverage_groups.assign(n_seqs, 0);
    double group_start = sorted.front().first;
    uint32_t group = 0;
    for (const auto &b : sorted) {
        if (b.first > 2 * group_start) {
            group_start = b.first;
            ++group;
        }
        s.coverage_groups[b.second] = group;
    }
    s.n_groups = group + 1;
    s.dev.coverage_group = up.put(s.coverage_groups);
    s.dev.ref_seq_bias = up.put(s.ref_seq_bias);
    for (uint32_t ref_id = n_seqs; ref_id--;) {
        if (0.0 == s.ref_seq_bias[ref_id]) continue;
        for (uint32_t fl : plan.sample)
            if (fl <= s.seq_len[ref_id]) {
                plan.params.push_back(BiasParam{ref_id, fl, s.ref_seq_bias[ref_id] * p.insert_lengths_bias[fl]});
                plan.max_starts = std::max(plan.max_starts, s.seq_len[ref_id] - fl + 1);
            }
    }
    plan.chunk_ptr.assign(1, 0);
    for (size_t i = 0; i < plan.params.size(); ++i) {
        const uint32_t chunks = cdiv(s.seq_len[plan.params[i].seq] - plan.params[i].len + 1, kBiasBlock * kBiasRun);
        if ((uint64_t)plan.chunk_ptr.back() + chunks > 0xFFFFFFF0ull) throw Error("too many chunks of start positions for the bias normalisation");
        plan.chunk_param.insert(plan.chunk_param.end(), chunks, (uint32_t)i);
        plan.chunk_ptr.push_back(plan.chunk_ptr.back() + chunks);
    }
    return plan;
}

// sums[i], maxes[i]: SumBias total and maximum of plan.params[i]
inline void finish_bias_normalization(SimState &s, const BiasPlan &plan, const std::vector<double> &sums, const std::vector<double> &maxes) {
    const Profile &p = s.prof;
    const uint32_t to = s.dev.insert_to;
    const std::vector<uint32_t> &sp = plan.sample;
    std::vector<double> norm(to, 0.0), max_bias((size_t)s.n_groups * to, 0.0);
    for (size_t i = 0; i < plan.params.size(); ++i) {
        norm[plan.params[i].len] += sums[i];
        double &m = max_bias[(size_t)s.coverage_groups[plan.params[i].seq] * to + plan.params[i].len];
        m = std::max(m, maxes[i]);
    }
    interpolate_normalization(p.insert_lengths_bias, sp, norm);
    for (uint32_t g = 0; g < s.n_groups; ++g) {                                       // :3540-3559
        double *grp = &max_bias[(size_t)g * to];
        double max_ratio = 0.0;
        for (uint32_t fl : sp) max_ratio = std::max(max_ratio, grp[fl] / p.insert_lengths_bias[fl]);
        for (size_t k = 1; k < sp.size(); ++k)
            for (uint32_t fl = sp[k - 1] + 1; fl < sp[k]; ++fl) grp[fl] = max_ratio * p.insert_lengths_bias[fl];
        for (uint32_t fl = sp.back() + 1; fl < to; ++fl) grp[fl] = max_ratio * p.insert_lengths_bias[fl];
    }
    double normalization = 0.0;
    for (double v : norm) normalization += v;
    s.bias_normalization = s.total_pairs / (normalization * 2);                       // :3566
    s.thresholds.assign((size_t)s.n_groups * to * 2, 1.0);
    for (size_t i = 0; i < (size_t)s.n_groups * to; ++i)
        if (0.0 != max_bias[i]) {
            s.thresholds[2 * i] = threshold0(p.dispersion, s.bias_normalization, max_bias[i], s.num_alleles);
            s.thresholds[2 * i + 1] = pow(s.thresholds[2 * i], 2 * s.num_alleles);               // FragmentDistributionStats.cpp:3575-3576
        }
    s.norm_by_len = norm;
    if (0.0 == s.bias_normalization) throw Error("bias normalisation is zero");
}

inline void upload_normalization(SimState &s, Uploader &up) {
    const ScopedUpload scope(up, kScopeNormalization, true);
    s.dev.thresholds = up.put(s.thresholds);
    s.dev.bias_normalization = s.bias_normalization;
    // The sieve draws the gaps between the lengths whose cell passes the zero threshold instead of one uniform per cell (k_sieve_gaps,
    // rsq_kernels.h): q[len] = product of thr1 over the lengths of len's segment up to len = the probability that none of them passes.
    // A segment ends where the product falls below 2^-500 (a threshold of exactly zero ends its segment at once), so no product
    // underflows; the next segment starts from 1 with a fresh draw.
    const uint32_t to = s.dev.insert_to, from = s.dev.insert_from;
    std::vector<double> q((size_t)s.n_groups * to, 1.0);
    std::vector<uint32_t> seg_end((size_t)s.n_groups * to, from);
    s.expected_passing = 0.0;
    for (uint32_t g = 0; g < s.n_groups; ++g) {
        uint32_t seg_begin = from;
        double run = 1.0, expected = 0.0;
        for (uint32_t len = from; len < to; ++len) {
            const double thr1 = s.thresholds[2 * ((size_t)g * to + len) + 1];
            expected += 1 - thr1;
            run *= thr1;
            q[(size_t)g * to + len] = run;
            if (run < 0x1p-500 || len + 1 == to) {
                for (uint32_t l = seg_begin; l <= len; ++l) seg_end[(size_t)g * to + l] = len + 1;
                seg_begin = len + 1;
                run = 1.0;
            }
        }
        s.expected_passing = std::max(s.expected_passing, expected);
    }
    s.dev.gap_q = up.put(q);
    s.dev.gap_seg_end = up.put(seg_end);
}

// ---- the bias sums in chunks (k_sum_bias): kBiasBlock * kBiasRun start positions per chunk, the chunks of a parameter next to each
// other (BiasPlan::chunk_ptr).  The chunks' partial sums are combined in chunk order -- the same additions whoever computed the chunks,
// one GPU or the ranks of a sharded job -- and the arithmetic after SumBias follows.
inline uint32_t bias_chunks(const BiasPlan &plan) { return plan.chunk_ptr.empty() ? 0u : plan.chunk_ptr.back(); }
inline void normalization_from_partials(SimState &s, Uploader &up, const BiasPlan &plan, const double *h_sum, const double *h_max) {
    std::vector<double> sums(plan.params.size(), 0.0), maxes(plan.params.size(), 0.0);
    for (size_t i = 0; i < plan.params.size(); ++i)
        for (uint32_t b = plan.chunk_ptr[i]; b < plan.chunk_ptr[i + 1]; ++b) {
            sums[i] += h_sum[b];
            maxes[i] = std::max(maxes[i], h_max[b]);
        }
    finish_bias_normalization(s, plan, sums, maxes);
    upload_normalization(s, up);
}

}  // namespace rsq
