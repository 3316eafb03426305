// hostemu.cpp -- TEST-ONLY CPU emulation of the kernels' per-lane functions.
//
// The shipped library (reseq_amd/libreseq_amd.so) has no CPU path.  This file builds a separate test artefact
// (tests/hostemu/libhostemu.so, g++) that instantiates the very same __host__ __device__ functions the HIP
// kernels call (rsq_core.h, rsq_kernels.h) and the same packing code (rsq_pack.h) with the arrays kept in host
// memory, and walks them with plain loops in place of the grid.  `pytest -m "not gpu"` compares it with the
// oracle so that state-machine, packing and counter-layout mistakes are caught in a container without a GPU.
// Nothing in reseq_amd/ links to or loads this file.
#include <stdint.h>
static uint64_t g_screen_stats[4][2];          // quality, base call, indel: draws, draws the screen left to double precision; indel draws, those the random word alone does not decide
#define RSQ_SCREEN_STATS g_screen_stats
static uint32_t g_ring_lag = 0;                    // emu_set_ring_lag
#include <stdlib.h>

#include <map>
#include <memory>
#include <utility>

#include "../../reseq_amd/csrc/rsq_deflate.h"
#include "../../reseq_amd/csrc/rsq_fasta.h"
#include "../../reseq_amd/csrc/rsq_pack.h"

using namespace rsq;

namespace {

struct HostUploader : Uploader {
    std::vector<void *> owned[kUploadScopes];
    void release_scope(int scope) override {
        for (void *p : owned[scope]) free(p);
        owned[scope].clear();
    }
    void *put_bytes(const void *data, size_t bytes) override {
        void *p = malloc(bytes + 8);
        memcpy(p, data, bytes);
        owned[current_scope].push_back(p);
        return p;
    }
    void *put_zeros(size_t bytes) override {
        void *p = calloc(bytes + 8, 1);
        owned[current_scope].push_back(p);
        return p;
    }
    void write_bytes(void *dst, const void *src, size_t bytes) override { memcpy(dst, src, bytes); }
    void read_bytes(void *dst_host, const void *src, size_t bytes) override { memcpy(dst_host, src, bytes); }
    ~HostUploader() override {
        for (int scope = 0; scope < kUploadScopes; ++scope) release_scope(scope);
    }
};

// the chunks of a chain run and what lives between the calls of the sharded pre-pass (rsq_sim::ChainRun)
struct ChainRun {
    std::vector<Chain> chains;
    std::vector<uint32_t> chunk_chain, used, out[2];
    ShardEdges edges;
    uint32_t passes = 0, block_lo = 0, block_hi = 0;
    bool valid = false, pass_through = false;
};
struct Emu : SimState {
    HostUploader up;
    BiasPlan bias_plan;                         // the sharded pre-pass: what lives between its calls
    ChainRun chain_run;
    std::vector<FragmentVar> fvars;             // of the last emu_sieve call (variants of any kind), parallel to its fragments
    int fill_mode = -1;                         // -1: screened draws on the LDS image when the plan has one (as the product does); 0: double precision only
    std::map<uint32_t, std::vector<float>> lds; // host stand-in for the LDS images, by image_qbase: one per template segment, or per (segment, tile)
    uint32_t mask() const { return effective_fill_mask(dev.lds.mask, fill_mode); }
    void build_lds() { lds.clear(); }           // images are built when a read first asks for them
    float *image(uint32_t seg, uint32_t tile) { // what a workgroup of k_fill_reads does before it serves reads of (seg, tile)
        const uint32_t qbase = image_qbase(dev, seg, dev.lds.binned ? tile : 0u);
        std::vector<float> &img = lds[qbase];
        if (img.empty()) {
            img.assign(dev.lds.total_words + 16, 0.f);
            lds_stage_descriptors(dev, img.data(), qbase, 0, 1);
            lds_stage_rows(dev, img.data(), qbase, 0, 1);
        }
        return img.data();
    }
};

// one read through the state machine the way a lane of k_fill_reads<MASK> runs it
template <class Src>
void run_read(Emu &s, uint32_t seg, const Stream &st, uint32_t tile, uint32_t frag_len, const Src &src, ReadOut &out, ReadMeta &meta) {
    if (!s.mask()) fill_read(s.dev, GlobalTables{s.dev}, st, seg, tile, frag_len, src, out, meta);
    else {
        bool done = false;
        auto run = [&](auto tag) {                            // what a lane of k_fill_reads<M> does; the position ring holds the step's rows
            constexpr uint32_t M = decltype(tag)::value;
            if (s.mask() != M) return;
            float *img = s.image(seg, tile), *ring = img + s.dev.lds.ring_off;      // the ring of wave 0
            const uint32_t qbase = image_qbase(s.dev, seg, s.dev.lds.binned ? tile : 0u);
            ScreenTables<M> tab{s.dev, img, qbase, ring, 0u};
            ReadMachine m;
            m.init(s.dev, tab, st, seg, tile, frag_len, src);
            for (;;) {
                // g_ring_lag = L: the read as a lane of a wave that is L steps ahead of it (the other lanes lost fewer steps to deletions): the ring holds the rows over the
                // wave's last positions, and beyond kRingLag the slot behind the ring holds the rows over this read's position (fill_wave_reads)
                const uint32_t t = m.par.read_pos + g_ring_lag;
                for (uint32_t item = 0; item < lds_ring_items(s.dev); ++item) {
                    const RingItem it = lds_ring_item(s.dev, qbase, item);
                    for (uint32_t back = 0; back <= kRingLag && back <= t; ++back) lds_ring_stage(s.dev, it, ring, t - back);
                    if (g_ring_lag > kRingLag) lds_ring_stage_demand(s.dev, it, ring, m.par.read_pos);
                }
                tab.t = t;
                tab.demand = g_ring_lag > kRingLag ? m.par.read_pos : 0xFFFFFFFFu;
                // as the wave's two loops run it: an iteration of the template part through the instantiation compiled for it, anything else through the general step
                if (m.in_template()) m.template iterate<true>(s.dev, tab, st, src, out);
                else if (!m.step(s.dev, tab, st, src, out)) break;
            }
            m.finalize(meta);
            done = true;
        };
        run(std::integral_constant<uint32_t, kQualityQuads[0]>{});
        run(std::integral_constant<uint32_t, kQualityQuads[1]>{});
        run(std::integral_constant<uint32_t, kQualityQuads[2]>{});
        run(std::integral_constant<uint32_t, kQualityQuads[3]>{});
        run(std::integral_constant<uint32_t, kQualityQuads[4]>{});
        if (!done) throw Error("no screened instantiation for mask " + std::to_string(s.mask()));
    }
}

thread_local std::string g_err;

template <class F>
int guard(F &&f) {
    try {
        f();
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// the reduction order of k_sum_bias: lanes own runs of kBiasRun starts, a block tree-reduces kBiasBlock lanes,
// the host adds the block partials in order
// k_sum_bias: the chunks of one parameter whose first start position lies in the share [g_lo, g_hi); sums / maxes: the parameter's gx chunks
void sum_bias_like_kernel(const Emu &s, const BiasParam &p, uint32_t gx, uint64_t g_lo, uint64_t g_hi, double *sums, double *maxes) {
    const uint32_t L = s.seq_len[p.seq];
    const uint64_t wo = s.seq_word_off[p.seq], bo = s.seq_base_off[p.seq];
    const uint32_t n_starts = L - p.len + 1;
    for (uint32_t b = 0; b < gx; ++b) {
        sums[b] = maxes[b] = 0.0;
        const uint64_t chunk_at = bo + (uint64_t)b * kBiasBlock * kBiasRun;
        if (chunk_at < g_lo || chunk_at >= g_hi) continue;
        double ls[kBiasBlock], lm[kBiasBlock];
        for (uint32_t t = 0; t < kBiasBlock; ++t) {               // lane t: start positions t, t + kBiasBlock, ... of the chunk
            double sum = 0.0, mx = 0.0;
            for (uint32_t j = 0; j < kBiasRun; ++j) {
                const uint32_t start = b * kBiasBlock * kBiasRun + j * kBiasBlock + t;
                if (start >= n_starts) break;
                const uint32_t gc = ref_gc_count(s.dev.ref_words, wo, start, start + p.len);
                const double bias = site_bias(s.dev, wo, L, start, p.len, gc, p.general_bias);
                sum += bias;
                mx = bias > mx ? bias : mx;
            }
            ls[t] = sum;
            lm[t] = mx;
        }
        for (uint32_t d = kBiasBlock / 2; d > 0; d >>= 1)
            for (uint32_t t = 0; t < d; ++t) {
                ls[t] += ls[t + d];
                lm[t] = lm[t + d] > lm[t] ? lm[t + d] : lm[t];
            }
        sums[b] = ls[0];
        maxes[b] = lm[0];
    }
}
void bias_partials(const Emu &s, const BiasPlan &plan, uint64_t g_lo, uint64_t g_hi, std::vector<double> &h_sum, std::vector<double> &h_max) {
    h_sum.assign(bias_chunks(plan), 0.0);
    h_max.assign(h_sum.size(), 0.0);
    for (size_t i = 0; i < plan.params.size(); ++i)
        if (plan.chunk_ptr[i + 1] > plan.chunk_ptr[i])
            sum_bias_like_kernel(s, plan.params[i], plan.chunk_ptr[i + 1] - plan.chunk_ptr[i], g_lo, g_hi, &h_sum[plan.chunk_ptr[i]], &h_max[plan.chunk_ptr[i]]);
}

void iterate_chains(Emu &s, ChainRun &run, uint32_t first_pass) {
    const uint32_t n = (uint32_t)run.chunk_chain.size();
    uint32_t pass = first_pass;
    for (;; ++pass) {                                      // same pass structure as k_sys_chain + iterate_sys_chains
        bool changed = false;
        const std::vector<uint32_t> &prev = run.out[(pass + 1) & 1];
        std::vector<uint32_t> &cur = run.out[pass & 1];
        for (uint32_t c = 0; c < n; ++c) {
            const Chain &ch = run.chains[run.chunk_chain[c]];
            const uint32_t local = c - ch.first_chunk;
            uint32_t want = local == 0 ? ch.in_state : 0u;
            if (pass > 0) {
                if (local) want = prev[c - 1];
                if (want == run.used[c]) {
                    cur[c] = prev[c];
                    continue;
                }
                changed = true;
            }
            run.used[c] = want;
            ChainAcc acc{s.dev.ref_words, ch.kind, ch.len, ch.kind < 2 ? s.seq_word_off[ch.id] : 0,
                         ch.kind == 2 ? s.dev.adapters[ch.seg].seqs + s.dev.adapters[ch.seg].seq_ptr[ch.id] : nullptr};
            uint32_t dist = want & 0xFFFFFFu, start_rate = want >> 24;
            const uint32_t lo = (ch.chunk_lo + local) * s.chain_chunk, hi = std::min(lo + s.chain_chunk, ch.len), warmup = chain_warmup_len(s.chain_chunk, s.opt);
            const uint32_t from = pass == 0 && local ? lo - std::min(warmup, lo) : lo;      // k_sys_chain: the guess of pass 0 is the end of a run-up
            sys_chain_chunk(s.dev, acc, ch.c1, ch.c2, from, hi, ch.initial_dom, dist, start_rate, ch.out, lo, &run.used[c]);
            cur[c] = dist | (start_rate << 24);
        }
        if (pass > 0 && !changed) break;
    }
    run.passes = pass + 1;
}
uint32_t run_chains(Emu &s, ChainSet set, ChainRun &run, const ShardRange *range = nullptr) {
    run = ChainRun{};
    s.chain_chunk = chain_chunk_len(s.total_ref_size, s.opt);
    build_chains(s, set, run.chains, run.chunk_chain, range, &run.edges);
    const uint32_t n = (uint32_t)run.chunk_chain.size();
    if (!n) return 0;
    run.used.assign(n, 0);
    run.out[0].assign(n, 0);
    run.out[1].assign(n, 0);
    iterate_chains(s, run, 0);
    return run.passes;
}
uint32_t run_chains(Emu &s, ChainSet set) {
    ChainRun run;
    return run_chains(s, set, run);
}

struct Raw {                              // one read alone: word columns of pitch 1
    std::vector<uint32_t> seq, qual, ops;
    WordColumn seq_col() { return WordColumn{seq.data(), 1}; }
    WordColumn qual_col() { return WordColumn{qual.data(), 1}; }
    WordColumn ops_col() { return WordColumn{ops.data(), 1}; }
    ReadOut out(const Emu &s) {
        seq.assign(s.read_stride / 4 + 2, 0);
        qual.assign(s.read_stride / 4 + 2, 0);
        ops.assign(s.ops_stride + 64, 0);
        return make_read_out(seq_col(), qual_col(), ops_col());
    }
};

}  // namespace

extern "C" {

const char *emu_last_error() { return g_err.c_str(); }

int emu_create(const char *profile_path, const char *fasta_path, uint64_t replace_n_seed, const char *vcf_path, void **out) {
    return guard([&] {
        std::unique_ptr<Emu> s(new Emu());
        s->fill_mode = (int)s->opt.fill_mode;                    // as rsq_sim_create
        s->prof = Profile::load(profile_path);
        pack_tables(*s, s->up);
        pack_profile(*s, s->up);
        if (fasta_path && fasta_path[0]) {
            Reference r = Reference::read_fasta(fasta_path);
            r.replace_n(replace_n_seed);
            if (vcf_path && vcf_path[0]) {                          // what rsq_ref_read_variants + rsq_sim_create do
                std::vector<std::string> first;
                for (size_t i = 0; i < r.names.size(); ++i) first.push_back(r.first_part(i));
                const Variants v = read_variants(vcf_path, first, r.codes);
                pack_reference(*s, s->up, r, &v);
            } else pack_reference(*s, s->up, r);
        }
        *out = s.release();
    });
}
void emu_free(void *h) { delete static_cast<Emu *>(h); }
// rsq_sim_export_reference / rsq_sim_import_reference: the same two functions of rsq_pack.h behind them
int emu_export_reference(void *h, const char *path) {
    return guard([&] {
        Emu &s = *static_cast<Emu *>(h);
        export_reference(s, s.up, path);
    });
}
int emu_import_reference(void *h, const char *path) {
    return guard([&] {
        Emu &s = *static_cast<Emu *>(h);
        const std::string text = read_text_file(path);               // plain bytes
        import_reference(s, s.up, text.data(), text.size(), path);
    });
}

int emu_edit_profile(void *h, double error_multiplier, int no_substitutions, int no_indels) {
    Emu &s = *static_cast<Emu *>(h);
    return guard([&] {
        if (error_multiplier != 1.0) s.prof.change_error_rate(error_multiplier);
        if (no_substitutions) s.prof.remove_substitution_errors();
        if (no_indels) s.prof.remove_indel_errors();
        pack_tables(s, s.up);                               // repack the edited tables
    });
}

// out[8]: the counters above; reset afterwards
void emu_screen_stats(uint64_t *out) {
    memcpy(out, g_screen_stats, sizeof g_screen_stats);
    memset(g_screen_stats, 0, sizeof g_screen_stats);
}
void emu_set_ring_lag(uint32_t steps) { g_ring_lag = steps; }
void emu_set_fill_mode(void *h, int mode) { static_cast<Emu *>(h)->fill_mode = mode; }
int emu_set_option(const char *name, long long value) { return set_option(name, value) ? 0 : -1; }       // rsq_set_option
long long emu_get_option(const char *name) {                                                             // rsq_get_option; -1 for an unknown name
    int64_t v = -1;
    return get_option(name, &v) ? (long long)v : -1;
}
int emu_image_tiles(void *h) { return (int)static_cast<Emu *>(h)->dev.lds.img_tiles; }
int emu_plan_mask(void *h) { return (int)static_cast<Emu *>(h)->dev.lds.mask; }

int emu_prepare(void *h, uint64_t seed, uint64_t num_pairs, double coverage, int ref_bias_mode, const char *base_identifier) {
    Emu &s = *static_cast<Emu *>(h);
    return guard([&] {
        plan_simulation(s, s.up, seed, num_pairs, coverage, ref_bias_mode, base_identifier);
        if (s.has_ref) {
            const BiasPlan plan = plan_bias_normalization(s, s.up);
            std::vector<double> h_sum, h_max;
            bias_partials(s, plan, 0, UINT64_MAX, h_sum, h_max);
            normalization_from_partials(s, s.up, plan, h_sum.data(), h_max.data());
        }
        s.passes = run_chains(s, s.has_ref ? kChainsSimulation : kChainsAdapters, s.chain_run);
        if (s.has_variants) {                                  // as rsq_sim.hip's prepare: the variants' bases in windows of the strands
            const std::vector<StrandTask> windows = strand_tasks(s.opt, s.chain_run.chains, (uint32_t)s.chain_run.chunk_chain.size(), s.chain_chunk, [&](uint32_t c) { return s.chain_run.used[c]; });
            build_variant_sys_errors(s, s.up, &windows);
        }
        s.chain_run.valid = false;
        s.build_lds();
        s.prepared = true;
    });
}

// the sharded pre-pass: host mirror of rsq_sim_prepare_plan / rsq_sim_bias_partials / rsq_sim_prepare_normalization / rsq_sim_prepare_sys_errors /
// rsq_sim_prepare_finish (rsq_sim.hip), same shared host code (shard_range, build_chains with a range, normalization_from_partials)
int emu_prepare_plan(void *h, uint64_t seed, uint64_t num_pairs, double coverage, int ref_bias_mode, const char *base_identifier) {
    Emu &s = *static_cast<Emu *>(h);
    return guard([&] {
        if (!s.has_ref) throw Error("the sharded pre-pass needs a reference");
        s.prepared = false;
        plan_simulation(s, s.up, seed, num_pairs, coverage, ref_bias_mode, base_identifier);
        s.bias_plan = plan_bias_normalization(s, s.up);
        s.chain_run = ChainRun{};
    });
}
uint64_t emu_bias_partials_size(void *h) {
    Emu &s = *static_cast<Emu *>(h);
    return (uint64_t)bias_chunks(s.bias_plan);
}
int emu_bias_partials(void *h, uint32_t block_lo, uint32_t block_hi, double *sums, double *maxes) {
    Emu &s = *static_cast<Emu *>(h);
    return guard([&] {
        const ShardRange r = shard_range(s, block_lo, block_hi);
        std::vector<double> h_sum, h_max;
        bias_partials(s, s.bias_plan, r.g_lo, r.g_lo < r.g_hi ? r.g_hi : r.g_lo, h_sum, h_max);
        memcpy(sums, h_sum.data(), h_sum.size() * 8);
        memcpy(maxes, h_max.data(), h_max.size() * 8);
    });
}
int emu_prepare_normalization(void *h, const double *sums, const double *maxes) {
    Emu &s = *static_cast<Emu *>(h);
    return guard([&] { normalization_from_partials(s, s.up, s.bias_plan, sums, maxes); });
}
int emu_prepare_sys_errors(void *h, uint32_t block_lo, uint32_t block_hi, const uint32_t *in_state, uint32_t *out_state) {
    Emu &s = *static_cast<Emu *>(h);
    return guard([&] {
        out_state[0] = in_state[0];
        out_state[1] = in_state[1];
        ChainRun &run = s.chain_run;
        if (!(run.valid && run.block_lo == block_lo && run.block_hi == block_hi)) {
            const ShardRange r = shard_range(s, block_lo, block_hi);
            s.passes = run_chains(s, kChainsSimulation, run, &r);
            run.block_lo = block_lo;
            run.block_hi = block_hi;
            run.pass_through = r.first_seq < 0;
            run.valid = true;
        }
        if (run.pass_through || run.chunk_chain.empty()) return;
        bool replaced = false;
        const int in_chain[2] = {run.edges.fwd_in_chain, run.edges.rev_in_chain};
        for (int k = 0; k < 2; ++k)
            if (in_chain[k] >= 0 && run.chains[(size_t)in_chain[k]].in_state != in_state[k]) {
                run.chains[(size_t)in_chain[k]].in_state = in_state[k];
                replaced = true;
            }
        if (replaced) {
            iterate_chains(s, run, run.passes);
            s.passes = run.passes;
        }
        const int64_t out_chunk[2] = {run.edges.fwd_out_chunk, run.edges.rev_out_chunk};
        for (int k = 0; k < 2; ++k) out_state[k] = out_chunk[k] >= 0 ? run.out[(run.passes - 1) & 1][(size_t)out_chunk[k]] : 0u;
    });
}
int emu_prepare_finish(void *h) {
    Emu &s = *static_cast<Emu *>(h);
    return guard([&] {
        if (!s.chain_run.valid) throw Error("the sharded pre-pass has not run");
        if (s.has_variants) {
            const std::vector<StrandTask> windows = strand_tasks(s.opt, s.chain_run.chains, (uint32_t)s.chain_run.chunk_chain.size(), s.chain_chunk, [&](uint32_t c) { return s.chain_run.used[c]; });
            build_variant_sys_errors(s, s.up, &windows);
        }
        s.build_lds();
        s.prepared = true;
    });
}

struct emu_info {
    uint64_t total_pairs, adapter_only_pairs;
    uint32_t total_blocks, n_groups, insert_to, passes, n_seqs, rmax;
    double bias_normalization;
};
void emu_get_info(void *h, emu_info *o) {
    Emu &s = *static_cast<Emu *>(h);
    *o = emu_info{s.total_pairs, s.adapter_only_pairs, s.total_blocks, s.n_groups, s.dev.insert_to, s.passes, s.dev.n_seqs, s.rmax, s.bias_normalization};
}
// rsq_sim_get_sequence_lengths: n_seqs values (emu_get_info tells how many)
void emu_get_sequence_lengths(void *h, uint32_t *out) {
    const Emu &s = *static_cast<Emu *>(h);
    memcpy(out, s.seq_len.data(), s.seq_len.size() * sizeof(uint32_t));
}
void emu_get_thresholds(void *h, double *out) {
    Emu &s = *static_cast<Emu *>(h);
    memcpy(out, s.thresholds.data(), s.thresholds.size() * 8);
}
void emu_get_norm_by_len(void *h, double *out) {
    Emu &s = *static_cast<Emu *>(h);
    memcpy(out, s.norm_by_len.data(), s.norm_by_len.size() * 8);
}
int emu_set_normalization(void *h, double bias_normalization, const double *thr, size_t n) {
    Emu &s = *static_cast<Emu *>(h);
    return guard([&] {
        if (n != s.thresholds.size()) throw Error("bad threshold count");
        s.bias_normalization = bias_normalization;
        s.thresholds.assign(thr, thr + n);
        upload_normalization(s, s.up);
    });
}
// the host mirror of rsq_sim_create_sys_error_profile / rsq_sim_read_sys_errors / rsq_sim_set_ref_bias_file (rsq_sim.hip)
int emu_create_sys_error_profile(void *h, uint64_t seed, const char *path) {
    return guard([&] {
        Emu &s = *static_cast<Emu *>(h);
        s.dev.seed = seed;
        set_sys_gc_range(s);
        run_chains(s, kChainsProfile);
        s.prepared = false;
        std::string text;
        std::vector<uint8_t> dom, rate;
        for (uint32_t i = 0; i < s.dev.n_seqs; ++i)
            for (uint32_t strand = 2; strand--;) {
                const uint32_t L = s.seq_len[i];
                const uint16_t *track = (strand ? s.sys_rev : s.sys_fwd) + s.seq_base_off[i];
                dom.resize(L);
                rate.resize(L);
                for (uint32_t k = 0; k < L; ++k) {
                    dom[k] = (uint8_t)(track[k] & 0xFF);
                    rate[k] = (uint8_t)(track[k] >> 8);
                }
                text += sys_error_fastq_record(s.ref_ids[i] + (strand ? " reverse" : " forward"), dom.data(), rate.data(), L);
            }
        write_text_file(path, text);
        return 0;
    });
}
int emu_read_sys_errors(void *h, const char *path) {
    return guard([&] {
        Emu &s = *static_cast<Emu *>(h);
        if (!s.prepared || !s.has_ref) throw Error("prepare first");
        apply_sys_error_records(s, s.up, parse_sys_error_fastq(read_text_file(path)));
        build_variant_sys_errors(s, s.up);
        return 0;
    });
}
int emu_read_methylation(void *h, const char *path) {
    return guard([&] {
        Emu &s = *static_cast<Emu *>(h);
        pack_methylation(s, s.up, read_methylation_file(path, s.ref_first_names, s.seq_len, s.num_alleles));
        return 0;
    });
}
// the product's BED parser alone (rsq_host.cpp read_methylation_file), for pinning against the reference's own loading test:
// rate_out is [sum of regions][num_alleles]; a sequence with one column repeats it for every allele (Reference.h:390-397)
int emu_parse_methylation_columns(const char *path, const char *names_nl, const uint32_t *lens, uint32_t n_seqs, uint32_t num_alleles, uint32_t *n_regions, uint32_t *first_out,
                                  uint32_t *second_out, double *rate_out, uint32_t cap, uint32_t *columns_out /* rate columns per sequence, may be null */) {
    return guard([&] {
        std::vector<std::string> names;
        std::string cur;
        for (const char *c = names_nl; *c; ++c) {
            if (*c == '\n') {
                names.push_back(cur);
                cur.clear();
            } else cur += *c;
        }
        if (names.size() != n_seqs) throw Error("one name per sequence");
        const Methylation m = read_methylation_file(path, names, std::vector<uint32_t>(lens, lens + n_seqs), num_alleles);
        uint32_t at = 0;
        for (uint32_t i = 0; i < n_seqs; ++i) {
            n_regions[i] = (uint32_t)m.first[i].size();
            if (columns_out) columns_out[i] = (uint32_t)m.rate[i].size();
            for (size_t k = 0; k < m.first[i].size() && at < cap; ++k, ++at) {
                first_out[at] = m.first[i][k];
                second_out[at] = m.second[i][k];
                for (uint32_t a = 0; a < num_alleles; ++a) rate_out[(size_t)at * num_alleles + a] = m.rate[i][1 < m.rate[i].size() ? a : 0][k];
            }
        }
        return 0;
    });
}
int emu_parse_methylation(const char *path, const char *names_nl, const uint32_t *lens, uint32_t n_seqs, uint32_t num_alleles, uint32_t *n_regions, uint32_t *first_out,
                          uint32_t *second_out, double *rate_out, uint32_t cap) {
    return emu_parse_methylation_columns(path, names_nl, lens, n_seqs, num_alleles, n_regions, first_out, second_out, rate_out, cap, nullptr);
}
// the product's InsertVariant alone (rsq_host.cpp), for ReferenceTest::TestInsertVariant: calls in, list out (positions, lengths, letters, bits)
int emu_insert_variants_test(uint32_t n_calls, const uint32_t *positions, const char *const *var_seqs, const uint64_t *allele0, uint32_t *n_out, uint32_t *pos_out,
                             char *seq_out /* n x 16 */, uint64_t *allele0_out) {
    return guard([&] {
        std::vector<Variant> list;
        for (uint32_t i = 0; i < n_calls; ++i) {
            std::vector<uint8_t> codes;
            for (const char *c = var_seqs[i]; *c; ++c) codes.push_back((uint8_t)(strchr("ACGT", *c) - "ACGT"));
            const uint64_t bits[2] = {allele0[i], 0};
            insert_variant(list, positions[i], codes, bits);
        }
        *n_out = (uint32_t)list.size();
        for (size_t i = 0; i < list.size(); ++i) {
            pos_out[i] = list[i].position;
            allele0_out[i] = list[i].allele[0];
            for (size_t k = 0; k < list[i].var_seq.size(); ++k) seq_out[i * 16 + k] = "ACGT"[list[i].var_seq[k]];
            seq_out[i * 16 + list[i].var_seq.size()] = 0;
        }
        return 0;
    });
}
int emu_set_ref_bias_file(void *h, const char *path) {
    static_cast<Emu *>(h)->ref_bias_file = path;
    return 0;
}
void emu_get_ref_seq_bias(void *h, double *out) {
    const Emu &s = *static_cast<Emu *>(h);
    memcpy(out, s.ref_seq_bias.data(), s.ref_seq_bias.size() * sizeof(double));
}

void emu_get_sys(void *h, int reverse, uint32_t seq, uint8_t *dom, uint8_t *rate) {
    Emu &s = *static_cast<Emu *>(h);
    const uint16_t *src = (reverse ? s.sys_rev : s.sys_fwd) + s.seq_base_off[seq];
    for (uint32_t i = 0; i < s.seq_len[seq]; ++i) {
        dom[i] = (uint8_t)(src[i] & 0xFF);
        rate[i] = (uint8_t)(src[i] >> 8);
    }
}
// var_errors_ of one variant on one strand after the pre-pass (dom | rate << 8 per base in the strand's drawing order); returns its length
uint32_t emu_get_variant_sys(void *h, uint32_t seq, uint32_t var_id, int reverse, uint16_t *out, uint32_t cap) {
    Emu &s = *static_cast<Emu *>(h);
    const DevVariant &v = s.variants[s.var_ptr[seq] + var_id];
    const uint16_t *err = reverse ? s.dev.var_err_rev : s.dev.var_err_fwd;
    for (uint32_t k = 0; k < v.len && k < cap; ++k) out[k] = err[v.off + k];
    return v.len;
}
void emu_get_adapter_sys(void *h, int seg, uint32_t id, uint8_t *dom, uint8_t *rate) {
    Emu &s = *static_cast<Emu *>(h);
    const HostAdapters &a = s.prof.adapters[seg];
    for (uint32_t i = 0; i < a.seq_ptr[id + 1] - a.seq_ptr[id]; ++i) {
        dom[i] = (uint8_t)(s.adapter_sys[seg][a.seq_ptr[id] + i] & 0xFF);
        rate[i] = (uint8_t)(s.adapter_sys[seg][a.seq_ptr[id] + i] >> 8);
    }
}
void emu_get_codes(void *h, uint32_t seq, uint8_t *out) {            // unpacks the 2-bit reference again
    Emu &s = *static_cast<Emu *>(h);
    for (uint32_t i = 0; i < s.seq_len[seq]; ++i) out[i] = (uint8_t)ref_base(s.dev.ref_words, s.seq_word_off[seq], i);
}

// variants of any kind: k_sieve_screen<2> / k_sieve_finish<2> / k_sieve_emit<2> as a loop over the batch's slots
static int64_t emu_sieve_general(Emu &s, uint32_t block_lo, uint32_t block_hi, Fragment *out, uint64_t cap) {
    const DevSim &S = s.dev;
    const uint32_t n_slots = (block_hi - block_lo) * kBlockSize + (S.block_extra_ptr[block_hi] - S.block_extra_ptr[block_lo]);
    uint64_t n = 0;
    uint32_t number = 0, last_block = 0;
    s.fvars.clear();
    for (uint32_t slot = 0; slot < n_slots; ++slot) {
        SieveSite site;
        const uint32_t block_id = init_site_slot<2>(S, block_lo, block_hi, slot, site);
        if (block_id != last_block) number = 0;
        last_block = block_id;
        if (site.start >= site.L) continue;
        sieve_gaps(S, site, [&](uint32_t len, double probability_chosen) {
            VarCell cell;
            if (!sieve_cell_general(S, site, len, probability_chosen, cell)) return;
            for (uint32_t e = 0; e < cell.n; ++e) {
                const uint32_t allele = cell.id[e] >> 1;
                const AlleleCell ac = allele_cell(allele_view(S, site.seq, allele), site.st, site.start, len);      // what k_sieve_emit<2> derives again
                const FragmentVar fv{ac.end, site.sub, site.st.first_variant_id, site.st.start_variant_pos, ac.end_var.first_variant_id, ac.end_var.start_variant_pos};
                for (uint32_t dup = 0; dup < cell.cnt[e]; ++dup) {
                    if (n < cap) {
                        out[n] = make_fragment(site, len, dup, cell.id[e] & 1u, block_id, number + 1, allele);
                        s.fvars.push_back(fv);
                    }
                    ++number;
                    ++n;
                }
            }
        });
    }
    return (int64_t)n;
}

// k_sieve_gaps + k_sieve_finish + k_sieve_emit with a plain loop over slots and their passing lengths
int64_t emu_sieve(void *h, uint32_t block_lo, uint32_t block_hi, Fragment *out, uint64_t cap) {
    Emu &s = *static_cast<Emu *>(h);
    const DevSim &S = s.dev;
    if (2 == s.variants_mode) return emu_sieve_general(s, block_lo, block_hi, out, cap);
    uint64_t n = 0;
    for (uint32_t block_id = block_lo; block_id < block_hi; ++block_id) {
        uint32_t number = 0;
        for (uint32_t off = 0; off < kBlockSize; ++off) {
            SieveSite site;
            init_site(S, block_id, off, site);
            if (site.start >= site.L) break;
            sieve_gaps(S, site, [&](uint32_t len, double probability_chosen) {
                uint32_t cnt[2], strand_of[2];
                if (s.has_variants) {                               // k_sieve_finish<1> + k_sieve_emit
                    VarCell cell;
                    if (!sieve_cell_var(S, site, len, probability_chosen, cell)) return;
                    for (uint32_t e = 0; e < cell.n; ++e)
                        for (uint32_t dup = 0; dup < cell.cnt[e]; ++dup) {
                            if (n < cap) out[n] = make_fragment(site, len, dup, cell.id[e] & 1u, block_id, number + 1, cell.id[e] >> 1);
                            ++number;
                            ++n;
                        }
                    return;
                }
                if (!sieve_cell(S, site, len, probability_chosen, cnt, strand_of)) return;
                for (uint32_t j = 0; j < 2; ++j)
                    for (uint32_t dup = 0; dup < cnt[j]; ++dup) {
                        if (n < cap) out[n] = make_fragment(site, len, dup, strand_of[j], block_id, number + 1);
                        ++number;
                        ++n;
                    }
            });
        }
    }
    return (int64_t)n;
}

// k_fill_reads + k_format for fragments (frags != NULL) or adapter-only pairs; returns bytes written per file
int emu_pairs_text(void *h, const Fragment *frags, uint64_t n_pairs, uint64_t adapter_first, char *r1, size_t cap1, size_t *len1, char *r2, size_t cap2, size_t *len2) {
    Emu &s = *static_cast<Emu *>(h);
    return guard([&] {
        size_t pos[2] = {0, 0};
        char *dst[2] = {r1, r2};
        size_t cap[2] = {cap1, cap2};
        Raw raw;
        for (uint64_t pair = 0; pair < n_pairs; ++pair)
            for (uint32_t seg = 0; seg < 2; ++seg) {
                ReadOut out = raw.out(s);
                ReadMeta meta;
                if (frags) {
                    const Fragment &f = frags[pair];
                    const uint32_t c2 = f.len | ((uint32_t)f.dup << 16);
                    const FragmentVar *fv = 2 == s.variants_mode ? &s.fvars.at(pair) : nullptr;
                    const uint32_t c1 = f.seq | ((fv ? fv->sub : 0u) << 22);
                    const Stream st{s.dev.seed, f.start, c1, c2, pair_c3(kDomPair, f.strand, seg, f.allele)};
                    const uint32_t tile = draw_tile(s.dev, f.start, c1, c2, pair_c3(kDomPair, f.strand, 2, f.allele));
                    if (s.has_variants) {                       // k_fill_reads<MASK, true>, after k_variant_templates when variants of any kind are loaded
                        VariantSrc src = variant_src(s.dev, f, fv, seg);
                        uint64_t tmpl[kTemplateWordsMax];
                        if (fv) {
                            variant_template(s.dev, f, *fv, seg, tmpl, s.template_words);
                            src.converted = tmpl;
                        }
                        run_read(s, seg, st, tile, f.len, src, out, meta);
                    } else {
                        FragmentSrc src = fragment_src(s.dev, f, seg);
                        uint64_t tmpl[kTemplateWordsMax];
                        if (s.dev.meth_ptr) {                   // what k_methylation_templates does for this read
                            convert_template(s.dev, f, seg, tmpl, s.template_words);
                            src.converted = tmpl;
                        }
                        run_read(s, seg, st, tile, f.len, src, out, meta);
                    }
                } else {
                    const uint64_t i = adapter_first + pair;
                    run_read(s, seg, Stream{s.dev.seed, (uint32_t)i, 0xFFFFFFFFu, (uint32_t)(i >> 32), pair_c3(kDomPair, 0, seg)},
                             draw_tile(s.dev, (uint32_t)i, 0xFFFFFFFFu, (uint32_t)(i >> 32), pair_c3(kDomPair, 0, 2)), 0u, EmptySrc{}, out, meta);
                }
                out.finish();
                const Fragment *fp = frags ? &frags[pair] : nullptr;
                const FragmentVar *fvp = frags && 2 == s.variants_mode ? &s.fvars.at(pair) : nullptr;
                const uint32_t need = record_size(s.dev, s.names, fp, adapter_first + pair + 1, meta, fvp);      // what k_fill_reads stores in sizes[]
                if (pos[seg] + need > cap[seg]) throw Error("text buffer too small");
                const uint32_t wrote = format_record(s.dev, s.names, fp, adapter_first + pair + 1, meta, raw.seq_col(), raw.qual_col(), raw.ops_col(), dst[seg] + pos[seg], fvp);
                if (wrote != need) throw Error("record_size disagrees with format_record");
                pos[seg] += need;
            }
        if (frags && s.has_variants && *s.dev.walk_error) {
            *s.dev.walk_error = 0;
            throw Error(kWalkErrorMessage);
        }
        *len1 = pos[0];
        *len2 = pos[1];
    });
}

// k_fill_records + k_error_model_out
int emu_error_model(void *h, uint64_t first_index, uint64_t n, uint32_t read_len, const uint8_t *seqs, const uint8_t *segs, const uint32_t *frag_len, const uint8_t *dom,
                    const uint8_t *rate, uint8_t *seq_out, uint8_t *qual_out, uint32_t out_stride, uint16_t *read_len_out, uint16_t *nerr_out, uint16_t *tile_out,
                    char *cigar_out, uint32_t cigar_stride) {
    Emu &s = *static_cast<Emu *>(h);
    return guard([&] {
        s.ops_stride = std::max(s.ops_stride, (s.rmax + read_len + s.max_adapter + 4u + 15u) / 16u);
        Raw raw;
        for (uint64_t i = 0; i < n; ++i) {
            ReadOut out = raw.out(s);
            ReadMeta m;
            RecordSrc src = record_src(seqs, dom, rate, read_len, i, n);
            const uint64_t idx = first_index + i;              // as a lane of k_fill_records<MASK> runs it
            const uint32_t seg = segs[i];
            run_read(s, seg, Stream{s.dev.seed, (uint32_t)idx, (uint32_t)(idx >> 32), 0u, pair_c3(kDomErrModel, 0, seg)},
                     draw_tile(s.dev, (uint32_t)idx, (uint32_t)(idx >> 32), 0u, pair_c3(kDomErrModel, 0, 2)), frag_len[i], src, out, m);
            out.finish();
            read_len_out[i] = m.read_len;
            nerr_out[i] = m.num_errors;
            tile_out[i] = m.tile_id;
            if (m.read_len > out_stride || m.cigar_chars + 1u > cigar_stride) throw Error("output stride too small");
            memcpy(seq_out + i * out_stride, raw.seq.data(), m.read_len);
            memcpy(qual_out + i * out_stride, raw.qual.data(), m.read_len);
            TextSink t{cigar_out + i * cigar_stride, 0};
            cigar_replay(raw.ops_col(), m, t);
            if (t.n != m.cigar_chars) throw Error("cigar_chars disagrees with the replayed CIGAR");
            t.ch(0);
        }
    });
}

// rsq_sim_error_model_fasta's parsing stage (rsq_fasta.h: record_start as k_fasta_starts applies it, parse_record as a lane of k_fasta_records runs it).
// at[cap + 1]; len / id_len / frag_len / seg[cap]; seqs / dom / rate[text_len + 8]; packed[text_len + 8] (or null): the device's half-word per base.  *bad = the first malformed record (its kind in *bad_kind) or 0xFFFFFFFF;
// *lead = 1: text other than line ends in front of the first record.  Returns -1 if cap is too small.
int emu_parse_fasta(const uint8_t *text, uint64_t text_len, int final, uint32_t cap, uint32_t *at, uint32_t *len, uint32_t *id_len, uint32_t *frag_len, uint8_t *seg, uint8_t *seqs,
                    uint8_t *dom, uint8_t *rate, uint32_t *n_records, uint64_t *consumed, uint32_t *bad, uint32_t *bad_kind, uint32_t *lead, uint16_t *packed) {
    uint32_t starts = 0;
    for (uint64_t p = 0; p < text_len; ++p)
        if (fasta::record_start(text, p)) {
            if (starts >= cap) return -1;
            at[starts++] = (uint32_t)p;
        }
    at[starts] = (uint32_t)text_len;
    *lead = 0;
    for (uint64_t p = 0; p < at[0]; ++p)
        if (text[p] != '\n' && text[p] != '\r') *lead = 1;
    const uint32_t n = final || !starts ? starts : starts - 1;
    *bad = 0xFFFFFFFFu;
    *bad_kind = 0;
    for (uint32_t i = 0; i < n; ++i) {
        fasta::RecordFields f{0, 0, 0, 0};
        const fasta::RecordError e = fasta::parse_record(text + at[i], (uint64_t)at[i + 1] - at[i], fasta::ByteArrays{seqs + at[i], dom + at[i], rate + at[i]}, f);
        if (packed) {                                                      // the device's layout of the same record: a half-word per base
            fasta::RecordFields g{0, 0, 0, 0};
            if (fasta::parse_record(text + at[i], (uint64_t)at[i + 1] - at[i], fasta::Packed{packed + at[i]}, g) != e) return -2;
        }
        len[i] = f.len;
        id_len[i] = f.id_len;
        frag_len[i] = f.frag_len;
        seg[i] = (uint8_t)f.seg;
        if (e != fasta::kRecordOk && i < *bad) {
            *bad = i;
            *bad_kind = (uint32_t)e;
        }
    }
    *n_records = n;
    *consumed = final || !starts ? text_len : at[starts - 1];
    return 0;
}

// rsq_fasta.h converts eight bytes at a time: every byte value in every place of a word against the conversions of one byte; returns the number of disagreements
int emu_fasta_words_check() {
    int bad = 0;
    for (uint32_t place = 0; place < 8; ++place)
        for (uint32_t c = 0; c < 256; ++c)
            for (uint32_t other : {0u, 0xFFu, 0x41u, 0x7Fu, 0x80u, 0x0Au}) {
                uint64_t w = 0;
                for (uint32_t j = 0; j < 8; ++j) w |= (uint64_t)(j == place ? c : other) << (8 * j);
                const uint64_t codes = fasta::base_codes(w), rates = fasta::rate_percents(w), zeros = fasta::zero_bytes(w);
                for (uint32_t j = 0; j < 8; ++j) {
                    const uint32_t b = (uint32_t)(w >> (8 * j)) & 0xFFu;
                    bad += ((codes >> (8 * j)) & 0xFFu) != fasta::base_code(b);
                    bad += ((rates >> (8 * j)) & 0xFFu) != fasta::rate_percent(b);
                    bad += ((zeros >> (8 * j)) & 0xFFu) != (b ? 0u : 0x80u);
                }
            }
    return bad;
}

// the library's text writer (gzip / bzip2 by the name's ending), for the tests of its compressed output
int emu_write_text_file(const char *path, const char *data, size_t n) {
    return guard([&] { write_text_file(path, std::string(data, n)); });
}

// rsq_deflate.h's per-thread walk of a piece with the workgroup's threads taken in turn: text -> gzip members (what k_gzip_pieces / k_gzip_stored / k_gzip_compact write).
// *len = bytes needed; written only if cap is enough.  force_stored: every piece through the stored route (the fallback for text the sample's code does not suit).
int emu_gzip(const uint8_t *text, uint64_t n, int force_stored, uint8_t *out, uint64_t cap, uint64_t *len) {
    return guard([&] {
        std::vector<uint8_t> members;
        if (force_stored) {
            std::vector<uint8_t> slot(rsq::gz::kSlot);
            for (uint64_t at = 0; at < n; at += rsq::gz::kPiece) {
                const uint32_t m = rsq::gz::stored_piece_on_the_host(text + at, (uint32_t)std::min<uint64_t>(rsq::gz::kPiece, n - at), slot.data());
                members.insert(members.end(), slot.begin() + rsq::gz::kSlotPad, slot.begin() + rsq::gz::kSlotPad + m);
            }
        } else rsq::gz::gzip_on_the_host(text, n, members);
        *len = members.size();
        if (members.size() <= cap && !members.empty()) memcpy(out, members.data(), members.size());
    });
}
// the code a sample's counts lead to: lengths of the 286 + 30 symbols and the header's size, for the tests of the code builder
int emu_gzip_code(const uint32_t *sample, uint8_t *lengths, uint32_t *header_bits) {
    return guard([&] {
        const rsq::gz::Codes c = rsq::gz::build_codes(sample);
        for (uint32_t i = 0; i < 286; ++i) lengths[i] = (uint8_t)(c.litlen[i] & 15u);
        for (uint32_t i = 0; i < 30; ++i) lengths[286 + i] = (uint8_t)(c.dist[i] & 15u);
        *header_bits = c.header_bits;
    });
}

// single draws, for direct comparison with the oracle's Draw
uint32_t emu_draw(void *h, int family, uint32_t index, const uint32_t *idx, double u, double *prob_sum) {
    Emu &s = *static_cast<Emu *>(h);
    const DevSim &S = s.dev;
    if (family == 0 || family == 2) {
        const uint32_t i4[4] = {idx[0], idx[1], idx[2], idx[3]};
        return draw<4>((family == 0 ? S.quality : S.base_call)[index], S.pool, S.par0, i4, u, *prob_sum);
    }
    const uint32_t i3[3] = {idx[0], idx[1], idx[2]};
    const DevTable *t = family == 1 ? S.seq_quality : family == 3 ? S.dom_error : family == 4 ? S.error_rate : S.indels;
    return draw<3>(t[index], S.pool, S.par0, i3, u, *prob_sum);
}

void emu_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t *out) {
    const Words w = philox(seed, c0, c1, c2, c3);
    out[0] = w.w0;
    out[1] = w.w1;
    out[2] = w.w2;
    out[3] = w.w3;
}

}  // extern "C"
