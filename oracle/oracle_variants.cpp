// oracle_variants.cpp -- TEST INFRASTRUCTURE (part of the CPU oracle, see oracle.h): the simulation with variants.
//   * SimulateFromGivenBlock with VariantsLoaded() (Simulator.cpp:2249-2357) on top of oracle_variants.hpp
//   * SetSystematicErrorVariantsForward / Reverse (Simulator.cpp:771-909,1011-1147) with DominantBaseWithMemory (utilities.hpp:302-351)
//   * GetSysErrorFromBlock / IncrementBlockPos (Simulator.cpp:232-292) as the cursor of orc_fill_read_cursor
//   * CreateReads / CreateReadId with alleles (Simulator.cpp:596-721), GetOrgSeq (:1898-1923)
// Random numbers: the Philox streams of DESIGN.md "Random streams" (rows "with variants").
#include "oracle_variants.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <string>

using namespace orcv;

namespace {
constexpr uint32_t kBlock = 1000;                                            // Simulator.h:254

thread_local std::string g_error;
template <class F>
int guard(F f) {
    try {
        return f();
    } catch (const std::exception &e) {
        g_error = e.what();
        return -1;
    }
}

std::vector<Variant> convert_variants(const orc_variants *vs, uint32_t seq) {
    std::vector<Variant> out;
    if (!vs) return out;
    out.resize(vs->n[seq]);
    for (uint32_t i = 0; i < vs->n[seq]; ++i) {
        const orc_variant &v = vs->v[seq][i];
        out[i].position = v.position;
        out[i].var_seq.assign(v.var_seq, v.var_seq + v.len);
        out[i].allele[0] = v.allele[0];
        out[i].allele[1] = v.allele[1];
    }
    return out;
}

// utilities.hpp:302-351
struct DomMem {
    orc_dominant_base d{};
    std::vector<uint8_t> mem;
    uint8_t get() const { return d.dom_base; }
    void clear() {
        orc_dombase_clear(&d);
        mem.clear();
    }
    template <class At>
    void set(At seq_at, uint32_t cur_pos) {
        mem.resize(std::min<uint32_t>(5u, cur_pos) + 1u);
        for (size_t mp = mem.size(); mp--;) mem[mp] = seq_at((uint32_t)(cur_pos + mp + 1 - mem.size()));
        orc_dombase_set(&d, mem.data(), (uint32_t)mem.size(), (uint32_t)mem.size() - 1u);
    }
    void update(uint8_t base) {
        if (mem.size() > 5u + 1u) mem.erase(mem.begin());
        mem.push_back(base);
        if (1 < mem.size()) orc_dombase_update(&d, mem[mem.size() - 2], mem.data(), (uint32_t)mem.size(), (uint32_t)mem.size() - 2u);
        else orc_dombase_set(&d, mem.data(), (uint32_t)mem.size(), 0);
    }
};

struct SeqVars {
    std::vector<uint8_t> codes;
    std::vector<Variant> variants;
    std::vector<std::vector<std::pair<uint8_t, uint8_t>>> err[2];           // [strand][variant] = var_errors_ of the strand's err_variants_ entry
    uint32_t lb(uint32_t pos) const {                                        // first variant with position >= pos
        uint32_t lo = 0, hi = (uint32_t)variants.size();
        while (lo < hi) {
            const uint32_t mid = (lo + hi) / 2;
            if (variants[mid].position < pos) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    }
};
struct VarState {
    uint32_t num_alleles = 1;
    std::vector<SeqVars> seqs;
};
const VarState &state_of(const orc_sim *s) { return *static_cast<const VarState *>(s->var_state); }

struct SysDraw {
    const orc_sim *s;
    uint32_t seq, strand, var_id, k = 0;
    // Simulator.h:337-352 DrawSystematicError; the uniforms: words 0,1 of Philox block (variant, sequence, 4 + strand, 3<<28 | index of the base in drawing order)
    void draw(std::vector<std::pair<uint8_t, uint8_t>> &out, uint8_t ref_base, uint8_t last_base, uint8_t dom_base, uint8_t gc_percent, uint32_t start_dist_error_region,
              uint8_t start_rate) {
        const orc_profile *p = s->p;
        const orc_philox_out w = orc_philox4x32_10(s->seed, var_id, seq, 4u + strand, ((uint32_t)ORC_DOM_SYSERR << 28) | k++);
        const uint32_t index[3] = {start_dist_error_region, gc_percent, start_rate};
        double prob_sum;
        uint8_t dom_error = (uint8_t)orc_draw(&p->dom_error[ref_base][last_base][dom_base], index, orc_u32(w.w[0]), &prob_sum);
        if (0.0 == prob_sum) dom_error = 4;
        uint8_t error_rate = (uint8_t)orc_draw(&p->error_rate[ref_base][dom_error], index, orc_u32(w.w[1]), &prob_sum);
        if (0.0 == prob_sum) error_rate = 0;
        out.emplace_back(dom_error, error_rate);
    }
};

// Simulator.h:354-366 on a sequence accessor
template <class At>
void update_gc(uint16_t &gc, uint16_t &bases, uint32_t pos, At seq_at, uint16_t sys_gc_range) {
    if (is_gc(seq_at(pos))) ++gc;
    if (bases < sys_gc_range) ++bases;
    else if (is_gc(seq_at(pos - bases))) --gc;
}

// SetSystematicErrorVariantsForward (Simulator.cpp:1011-1147) for every block of one sequence, in block order; the per-allele state
// (sys_last_var_pos_per_allele_, sys_last_base_per_allele_, sys_dom_base_per_allele_) runs through the blocks as in CreateBlock
void sys_error_variants_forward(const orc_sim *s, uint32_t seq, SeqVars &sv, uint32_t A) {
    const uint32_t L = (uint32_t)sv.codes.size(), n = (uint32_t)sv.variants.size();
    const uint8_t *rate = s->sys_rate[0][seq];
    auto ref_at = [&](uint32_t pos) { return sv.codes[pos]; };
    std::vector<uint32_t> last_var_pos(A, UINT32_MAX);                      // ResetSystematicErrorCounters(ref) :734-748
    std::vector<uint8_t> last_base_of(A, 4);
    std::vector<DomMem> dom(A);
    uint32_t run_dist = 0;                                                   // distance_to_start_of_error_region_ / start_error_rate_ at the block's start
    uint8_t run_rate = 0;
    sv.err[0].assign(n, {});
    for (uint32_t bs = 0; bs < L; bs += kBlock) {
        const uint32_t end_pos = std::min(bs + kBlock, L), first = sv.lb(bs);
        uint32_t start_dist = run_dist;                                      // the tmp_ copies of CreateBlock :1236-1239 (with --readSysError the members themselves:
        uint8_t start_rate = run_rate;                                       //  the same values, the block's rates are folded in either way)
        uint16_t gc = 0, gc_bases = 0;
        for (uint32_t var_id = first; var_id < n && sv.variants[var_id].position < end_pos; ++var_id) {
            const Variant &var = sv.variants[var_id];
            const uint32_t chosen = var.first_allele();
            std::vector<std::pair<uint8_t, uint8_t>> tmp;
            uint8_t last_base;
            if (last_var_pos[chosen] < L && last_var_pos[chosen] + 1u == var.position) last_base = last_base_of[chosen];
            else last_base = var.position ? sv.codes[var.position - 1u] : 4;
            if (last_var_pos[chosen] < L && last_var_pos[chosen] + 5u >= var.position) {
                for (uint32_t pos = last_var_pos[chosen] + 1u; pos < var.position; ++pos) dom[chosen].update(sv.codes[pos]);
            } else {
                dom[chosen].clear();
                if (var.position) dom[chosen].set(ref_at, var.position - 1u);
            }
            if (first == var_id) {
                for (uint32_t pos = 0; pos < var.position - bs; ++pos) orc_update_distances(s->p->reset_distance, &start_dist, &start_rate, rate[bs + pos]);
            } else {
                for (uint32_t pos = sv.variants[var_id - 1].position - bs; pos < var.position - bs; ++pos)
                    orc_update_distances(s->p->reset_distance, &start_dist, &start_rate, rate[bs + pos]);
            }
            if (var_id && var_id != first && sv.variants[var_id - 1].position + s->sys_gc_range / 3u > var.position) {
                for (uint32_t pos = sv.variants[var_id - 1].position; pos < var.position; ++pos) update_gc(gc, gc_bases, pos, ref_at, s->sys_gc_range);
            } else {
                gc_bases = (uint16_t)std::min<uint32_t>(var.position, s->sys_gc_range);
                gc = 0;
                for (uint32_t pos = var.position - gc_bases; pos < var.position; ++pos) gc += is_gc(sv.codes[pos]) ? 1 : 0;
            }
            if (!var.var_seq.empty()) {
                SysDraw dr{s, seq, 0, var_id};
                for (uint32_t vpos = 0; vpos < var.var_seq.size(); ++vpos) {
                    const uint8_t base = var.var_seq[vpos];
                    dom[chosen].update(base);
                    dr.draw(tmp, base, last_base, dom[chosen].get(), orc_safe_percent_u16(gc, gc_bases), orc_transform_distance(start_dist), start_rate);
                    last_base = base;
                }
            }
            sv.err[0][var_id] = tmp;
            uint32_t ref_allele = A;
            if (last_var_pos[chosen] >= L || last_var_pos[chosen] + 5u < var.position + (uint32_t)var.var_seq.size()) ref_allele = chosen;
            for (uint32_t allele = 0; allele < A; ++allele) {
                if (!var.in_allele(allele)) continue;
                if (allele != chosen) {
                    if (last_var_pos[allele] < L && last_var_pos[allele] + 5u >= var.position + (uint32_t)var.var_seq.size()) {
                        for (uint32_t pos = last_var_pos[allele] + 1u; pos < var.position; ++pos) dom[allele].update(sv.codes[pos]);
                        for (uint32_t vpos = 0; vpos < var.var_seq.size(); ++vpos) dom[allele].update(var.var_seq[vpos]);
                    } else if (ref_allele < A) {
                        dom[allele] = dom[ref_allele];
                    } else {
                        ref_allele = allele;
                        dom[allele].clear();
                        if (var.position) dom[allele].set(ref_at, var.position - 1u);
                        for (uint32_t vpos = 0; vpos < var.var_seq.size(); ++vpos) dom[allele].update(var.var_seq[vpos]);
                    }
                }
                last_var_pos[allele] = var.position;
                last_base_of[allele] = last_base;
            }
        }
        for (uint32_t pos = bs; pos < end_pos; ++pos) orc_update_distances(s->p->reset_distance, &run_dist, &run_rate, rate[pos]);
    }
}

// SetSystematicErrorVariantsReverse (Simulator.cpp:771-909) for the reverse blocks of one sequence, last block first (CreateUnit :977-993)
void sys_error_variants_reverse(const orc_sim *s, uint32_t seq, SeqVars &sv, uint32_t A) {
    const uint32_t L = (uint32_t)sv.codes.size(), n = (uint32_t)sv.variants.size();
    const uint8_t *rate = s->sys_rate[1][seq];                               // index = position on the reverse complement
    auto rc_at = [&](uint32_t pos) { return (uint8_t)(3u - sv.codes[L - 1u - pos]); };      // ConstDna5StringReverseComplement
    std::vector<uint32_t> last_var_pos(A, UINT32_MAX);
    std::vector<uint8_t> last_base_of(A, 4);
    std::vector<DomMem> dom(A);
    uint32_t run_dist = 0;
    uint8_t run_rate = 0;
    sv.err[1].assign(n, {});
    const uint32_t n_blocks = (L + kBlock - 1u) / kBlock;
    for (uint32_t bi = n_blocks; bi--;) {
        const uint32_t bs = bi * kBlock, size = std::min(kBlock, L - bs);    // block.start_pos_, block.sys_errors_.size()
        const uint32_t rc_start = L - bs - size;                             // the block's first position on the reverse complement
        const int32_t first = (int32_t)sv.lb(bs + kBlock) - 1;               // block.first_variant_id_ (CreateUnit :947-957)
        uint32_t start_dist = run_dist;
        uint8_t start_rate = run_rate;
        uint16_t gc = 0, gc_bases = 0;
        for (int32_t var_id = first; 0 <= var_id && sv.variants[(size_t)var_id].position >= bs; --var_id) {
            const Variant &var = sv.variants[(size_t)var_id];
            const uint32_t chosen = var.first_allele();
            std::vector<std::pair<uint8_t, uint8_t>> tmp;
            const uint32_t rev_pos = L - var.position - 1u, block_pos = bs + size - var.position - 1u;
            uint8_t last_base;
            if (last_var_pos[chosen] < L && last_var_pos[chosen] == var.position + 1u) last_base = last_base_of[chosen];
            else last_base = var.position + 1u < L ? (uint8_t)(3u - sv.codes[var.position + 1u]) : 4;
            if (last_var_pos[chosen] < L && last_var_pos[chosen] <= var.position + 5u) {
                for (uint32_t pos = last_var_pos[chosen] - 1u; pos > var.position; --pos) dom[chosen].update((uint8_t)(3u - sv.codes[pos]));
            } else {
                dom[chosen].clear();
                if (rev_pos) dom[chosen].set(rc_at, rev_pos - 1u);
            }
            if (first == var_id) {
                for (uint32_t pos = 0; pos < block_pos; ++pos) orc_update_distances(s->p->reset_distance, &start_dist, &start_rate, rate[rc_start + pos]);
            } else {
                for (uint32_t pos = bs + size - sv.variants[(size_t)var_id + 1].position - 1u; pos < block_pos; ++pos)
                    orc_update_distances(s->p->reset_distance, &start_dist, &start_rate, rate[rc_start + pos]);
            }
            if (n > (uint32_t)var_id + 1u && var_id != first && sv.variants[(size_t)var_id + 1].position < var.position + s->sys_gc_range / 3u) {
                for (uint32_t pos = L - sv.variants[(size_t)var_id + 1].position - 1u; pos < rev_pos; ++pos) update_gc(gc, gc_bases, pos, rc_at, s->sys_gc_range);
            } else {
                gc_bases = (uint16_t)std::min<uint32_t>(rev_pos, s->sys_gc_range);
                gc = 0;
                for (uint32_t pos = var.position + 1u; pos < var.position + gc_bases + 1u; ++pos) gc += is_gc(sv.codes[pos]) ? 1 : 0;
            }
            if (!var.var_seq.empty()) {
                SysDraw dr{s, seq, 1, (uint32_t)var_id};
                for (uint32_t vpos = (uint32_t)var.var_seq.size(); vpos--;) {
                    const uint8_t base = (uint8_t)(3u - var.var_seq[vpos]);
                    dom[chosen].update(base);
                    dr.draw(tmp, base, last_base, dom[chosen].get(), orc_safe_percent_u16(gc, gc_bases), orc_transform_distance(start_dist), start_rate);
                    last_base = base;
                }
            }
            sv.err[1][(size_t)var_id] = tmp;
            uint32_t ref_allele = A;
            if (last_var_pos[chosen] >= L || last_var_pos[chosen] + (uint32_t)var.var_seq.size() > var.position + 5u) ref_allele = chosen;
            for (uint32_t allele = 0; allele < A; ++allele) {
                if (!var.in_allele(allele)) continue;
                if (allele != chosen) {
                    if (last_var_pos[allele] < L && last_var_pos[allele] + (uint32_t)var.var_seq.size() <= var.position + 5u) {
                        for (uint32_t pos = last_var_pos[allele] - 1u; pos > var.position; --pos) dom[allele].update((uint8_t)(3u - sv.codes[pos]));
                        for (uint32_t vpos = (uint32_t)var.var_seq.size(); vpos--;) dom[allele].update((uint8_t)(3u - var.var_seq[vpos]));
                    } else if (ref_allele < A) {
                        dom[allele] = dom[ref_allele];
                    } else {
                        ref_allele = allele;
                        dom[allele].clear();
                        if (rev_pos) dom[allele].set(rc_at, rev_pos - 1u);
                        for (uint32_t vpos = (uint32_t)var.var_seq.size(); vpos--;) dom[allele].update((uint8_t)(3u - var.var_seq[vpos]));
                    }
                }
                last_var_pos[allele] = var.position;
                last_base_of[allele] = last_base;
            }
        }
        for (uint32_t pos = 0; pos < size; ++pos) orc_update_distances(s->p->reset_distance, &run_dist, &run_rate, rate[rc_start + pos]);
    }
}

// the systematic-error cursor of one read: GetSysErrorFromBlock / IncrementBlockPos (Simulator.cpp:232-292) and the deletion branch of
// FillReadPart (:380-392) over the blocks of one strand.  A block is named by its index; its err_variants_ list is a run of variant ids
// (ascending from lb(start) on the forward strand, descending from lb(start + 1000) - 1 on the reverse strand).
struct Walk {
    const orc_sim *s;
    const SeqVars *sv;
    uint32_t seq, strand, allele;
    // start state (CreateReads :672-688) and running state
    int64_t block0, block;                                                   // block index; -1 / n_blocks = the NULL next_block_
    uint32_t block_pos0, block_pos;
    int32_t cur_var0, cur_var;
    uint32_t var_pos0, var_pos;
    uint32_t L() const { return (uint32_t)sv->codes.size(); }
    uint32_t n_blocks() const { return (L() + kBlock - 1u) / kBlock; }
    void check_block() const {
        if (block < 0 || block >= (int64_t)n_blocks()) throw Error("systematic-error walk left the sequence (the reference would dereference a NULL next_block_)");
    }
    uint32_t bs() const { return (uint32_t)block * kBlock; }
    uint32_t size() const { return std::min(kBlock, L() - bs()); }
    uint32_t list_size() const {                                             // err_variants_.size()
        const uint32_t lo = sv->lb(bs()), hi = sv->lb(bs() + kBlock);
        return hi - lo;
    }
    uint32_t var_id(int32_t j) const { return strand ? sv->lb(bs() + kBlock) - 1u - (uint32_t)j : sv->lb(bs()) + (uint32_t)j; }
    uint32_t position_(int32_t j) const {                                    // err_variants_.at(j).position_
        const uint32_t p = sv->variants[var_id(j)].position;
        return strand ? bs() + size() - p - 1u : p - bs();
    }
    const std::vector<std::pair<uint8_t, uint8_t>> &var_errors(int32_t j) const { return sv->err[strand][var_id(j)]; }
    bool in_allele(int32_t j) const { return sv->variants[var_id(j)].in_allele(allele); }
    void sys_at(uint32_t bp, uint8_t *dom, uint8_t *rate) const {            // block->sys_errors_.at(bp)
        if (bp >= size()) throw Error("systematic-error walk: sys_errors_.at() out of range");
        const uint32_t idx = strand ? (L() - bs() - size()) + bp : bs() + bp;
        if (dom) *dom = s->sys_dom[strand][seq][idx];
        *rate = s->sys_rate[strand][seq][idx];
    }
    void increment_block_pos(int32_t &cv) {                                  // :232-238
        check_block();
        if (size() <= ++block_pos) {
            block += strand ? -1 : 1;
            block_pos = 0;
            cv = 0;
        }
    }
    void reset() {
        block = block0;
        block_pos = block_pos0;
        cur_var = cur_var0;
        var_pos = var_pos0;
    }
    void next(uint8_t *dom_error, uint8_t *error_rate) {                     // :240-292
        check_block();
        bool no_variant = true;
        if (var_pos) {
            no_variant = false;
            sys_at(block_pos, dom_error, error_rate);
            if ((uint32_t)cur_var >= list_size()) throw Error("systematic-error walk: err_variants_.at() out of range");
            if (++var_pos >= var_errors(cur_var).size()) {
                var_pos = 0;
                ++cur_var;
                increment_block_pos(cur_var);
            }
        } else {
            while (cur_var >= 0 && (uint32_t)cur_var < list_size() && position_(cur_var) <= block_pos) {
                if (in_allele(cur_var)) {
                    if (var_errors(cur_var).empty()) {                       // deletion
                        ++cur_var;
                        increment_block_pos(cur_var);
                        check_block();
                    } else {
                        no_variant = false;
                        *dom_error = var_errors(cur_var)[0].first;
                        *error_rate = var_errors(cur_var)[0].second;
                        if (1 == var_errors(cur_var).size()) {               // substitution
                            ++cur_var;
                            increment_block_pos(cur_var);
                            ++cur_var;
                        } else var_pos = 1;                                  // insertion
                        break;
                    }
                } else ++cur_var;
            }
        }
        if (no_variant) {
            check_block();
            sys_at(block_pos, dom_error, error_rate);
            increment_block_pos(cur_var);
        }
    }
    void deleted(uint8_t *error_rate) {                                      // :380-392
        check_block();
        sys_at(block_pos, nullptr, error_rate);
        if (var_pos) {
            if ((uint32_t)cur_var >= list_size()) throw Error("systematic-error walk: err_variants_.at() out of range");
            if (++var_pos >= var_errors(cur_var).size()) var_pos = 0;
        }
        if (0 == var_pos && size() <= ++block_pos) {
            block += strand ? -1 : 1;
            block_pos = 0;
            cur_var = 0;
        }
    }
};
void walk_reset(void *ctx) { static_cast<Walk *>(ctx)->reset(); }
void walk_next(void *ctx, uint8_t *dom, uint8_t *rate) { static_cast<Walk *>(ctx)->next(dom, rate); }
void walk_deleted(void *ctx, uint8_t *rate) { static_cast<Walk *>(ctx)->deleted(rate); }

uint32_t pair_c3(uint32_t dom, uint32_t strand, uint32_t segsel, uint32_t allele) { return (dom << 28) | (strand << 27) | (segsel << 25) | (allele << 17); }
uint32_t discrete_draw(const double *cp, size_t n, double u) {               // as oracle_sim.c
    size_t i = 0;
    while (i + 1 < n && !(u < cp[i])) ++i;
    return (uint32_t)i;
}

// Simulator::CTConversion with variants (Simulator.cpp:2004-2217), variable widths as there (read_pos is a uintReadLen, cur_meth an
// intVariantId).  `deletion` is read before it is ever written in the reference; it starts as false here.  The uniform of template
// position read_pos: word read_pos&3 of Philox block (start, sequence | sub<<22, length, 7<<28 | reversed<<27 | allele<<17 | read_pos>>2).
struct MethDraw {
    const orc_sim *s;
    uint32_t c0, c1, c2, c3base;
    double operator()(uint16_t read_pos) const { return orc_u32(orc_philox4x32_10(s->seed, c0, c1, c2, c3base | (read_pos >> 2)).w[read_pos & 3]); }
};
void ct_conversion_variants(const orc_sim *s, std::vector<uint8_t> &read, uint32_t seq_id, uint32_t start_pos, uint32_t allele, int32_t cur_methylation_start, bool reversed,
                            const std::vector<Variant> &variants, std::pair<int32_t, uint32_t> first_variant, const MethDraw &draw) {
    int32_t cur_meth = cur_methylation_start;
    uint16_t read_pos = 0;
    uint32_t ref_pos = start_pos;
    const double *conversion_rate = s->meth_rate_cols[seq_id][1 < s->meth_n_cols[seq_id] ? allele : 0];      // Reference::Unmethylation (Reference.h:390-397)
    const uint32_t *first = s->meth_first[seq_id], *second = s->meth_second[seq_id];
    const int64_t n_regions = s->meth_n[seq_id], n_var = (int64_t)variants.size();
    int32_t cur_var = first_variant.first;
    uint32_t var_bases_left = 0;
    bool deletion = false;
    auto convert = [&]() {
        if (1 == read.at(read_pos))
            if (draw(read_pos) < conversion_rate[cur_meth]) read[read_pos] = 3;
    };
    if (reversed) {
        if (0 <= cur_var && variants.at((size_t)cur_var).position == ref_pos && 1 < variants[(size_t)cur_var].var_seq.size() && variants[(size_t)cur_var].in_allele(allele))
            var_bases_left = (uint32_t)variants[(size_t)cur_var].var_seq.size() - first_variant.second;
        while (cur_meth < n_regions && first[cur_meth] <= ref_pos) ++cur_meth;
        --cur_meth;
        if (0 <= cur_meth && second[cur_meth] <= ref_pos) {
            if (var_bases_left) {
                read_pos = (uint16_t)(read_pos + var_bases_left);
                var_bases_left = 0;
                --ref_pos;
                --cur_var;
            }
            while (0 <= cur_var && second[cur_meth] <= variants.at((size_t)cur_var).position && read_pos < read.size()) {
                if (variants[(size_t)cur_var].in_allele(allele)) {
                    read_pos = (uint16_t)(read_pos + (ref_pos - variants[(size_t)cur_var].position));
                    read_pos = (uint16_t)(read_pos + variants[(size_t)cur_var].var_seq.size());
                    ref_pos = variants[(size_t)cur_var].position - 1u;
                }
                --cur_var;
            }
            read_pos = (uint16_t)(read_pos + (ref_pos - (second[cur_meth] - 1u)));
            ref_pos = second[cur_meth] - 1u;
        }
        while (0 <= cur_meth && read_pos < read.size()) {
            while (ref_pos >= first[cur_meth] && read_pos < read.size()) {
                if (0 == var_bases_left) {
                    while (0 <= cur_var && variants.at((size_t)cur_var).position == ref_pos && !variants[(size_t)cur_var].in_allele(allele)) --cur_var;
                    if (0 <= cur_var && variants.at((size_t)cur_var).position == ref_pos) {
                        if (variants[(size_t)cur_var].var_seq.empty()) {
                            deletion = true;
                            --cur_var;
                        } else var_bases_left = (uint32_t)variants[(size_t)cur_var].var_seq.size();
                    }
                }
                if (deletion) deletion = false;
                else convert();
                if (var_bases_left)
                    if (0 == --var_bases_left) --cur_var;
                if (0 == var_bases_left) --ref_pos;
                ++read_pos;
            }
            if (0 <= --cur_meth && second[cur_meth] <= ref_pos) {
                while (0 <= cur_var && second[cur_meth] <= variants.at((size_t)cur_var).position && read_pos < read.size()) {
                    if (variants[(size_t)cur_var].in_allele(allele)) {
                        read_pos = (uint16_t)(read_pos + (ref_pos - variants[(size_t)cur_var].position));
                        read_pos = (uint16_t)(read_pos + variants[(size_t)cur_var].var_seq.size());
                        ref_pos = variants[(size_t)cur_var].position - 1u;
                    }
                    --cur_var;
                }
                read_pos = (uint16_t)(read_pos + (ref_pos - (second[cur_meth] - 1u)));
                ref_pos = second[cur_meth] - 1u;
            }
        }
    } else {
        if (cur_var < n_var && variants.at((size_t)cur_var).position == ref_pos && 1 < variants[(size_t)cur_var].var_seq.size() && variants[(size_t)cur_var].in_allele(allele))
            var_bases_left = (uint32_t)variants[(size_t)cur_var].var_seq.size() - first_variant.second;
        if (cur_meth < n_regions && first[cur_meth] > ref_pos) {
            if (var_bases_left) {
                read_pos = (uint16_t)(read_pos + var_bases_left);
                var_bases_left = 0;
                ++ref_pos;
                ++cur_var;
            }
            while (cur_var < n_var && first[cur_meth] > variants.at((size_t)cur_var).position && read_pos < read.size()) {
                if (variants[(size_t)cur_var].in_allele(allele)) {
                    read_pos = (uint16_t)(read_pos + (variants[(size_t)cur_var].position - ref_pos));
                    read_pos = (uint16_t)(read_pos + variants[(size_t)cur_var].var_seq.size());
                    ref_pos = variants[(size_t)cur_var].position + 1u;
                }
                ++cur_var;
            }
            read_pos = (uint16_t)(read_pos + (first[cur_meth] - ref_pos));
            ref_pos = first[cur_meth];
        }
        while (cur_meth < n_regions && read_pos < read.size()) {
            while (ref_pos < second[cur_meth] && read_pos < read.size()) {
                if (0 == var_bases_left) {
                    while (cur_var < n_var && variants.at((size_t)cur_var).position == ref_pos && !variants[(size_t)cur_var].in_allele(allele)) ++cur_var;
                    if (cur_var < n_var && variants.at((size_t)cur_var).position == ref_pos) {
                        if (variants[(size_t)cur_var].var_seq.empty()) {
                            deletion = true;
                            ++cur_var;
                        } else var_bases_left = (uint32_t)variants[(size_t)cur_var].var_seq.size();
                    }
                }
                if (deletion) deletion = false;
                else convert();
                if (var_bases_left)
                    if (0 == --var_bases_left) ++cur_var;
                if (0 == var_bases_left) ++ref_pos;
                ++read_pos;
            }
            if (++cur_meth < n_regions && first[cur_meth] > ref_pos) {
                while (cur_var < n_var && first[cur_meth] > variants.at((size_t)cur_var).position && read_pos < read.size()) {
                    if (variants[(size_t)cur_var].in_allele(allele)) {
                        read_pos = (uint16_t)(read_pos + (variants[(size_t)cur_var].position - ref_pos));
                        read_pos = (uint16_t)(read_pos + variants[(size_t)cur_var].var_seq.size());
                        ref_pos = variants[(size_t)cur_var].position + 1u;
                    }
                    ++cur_var;
                }
                read_pos = (uint16_t)(read_pos + (first[cur_meth] - ref_pos));
                ref_pos = first[cur_meth];
            }
        }
    }
}

// ---- the scenario driver of SimulatorTest::TestVariationInSimulateFromGivenBlock (SimulatorTest.cpp:116-364)
struct VarScenario {
    std::vector<uint8_t> codes, comp_codes;            // the reference and the sequence with allele 1's variants applied
    std::vector<Variant> variants, none;
    VariantBiasMod bm{0, 2};
    VarRef ref() const {
        VarRef r;
        r.codes = &codes;
        r.variants = &variants;
        r.num_alleles = 2;
        return r;
    }
    VarRef comp() const {
        VarRef r;
        r.codes = &comp_codes;
        r.variants = &none;
        return r;
    }
};
}  // namespace

extern "C" {

const char *orc_var_last_error(void) { return g_error.c_str(); }

void *orc_var_new(const uint8_t *codes, uint32_t n, const uint8_t *comp_codes, uint32_t n_comp, uint32_t n_var, const uint32_t *positions, const char *const *var_seqs,
                  const uint64_t *allele0) {
    VarScenario *v = new VarScenario();
    v->codes.assign(codes, codes + n);
    v->comp_codes.assign(comp_codes, comp_codes + n_comp);
    for (uint32_t i = 0; i < n_var; ++i) {
        Variant x;
        x.position = positions[i];
        for (const char *c = var_seqs[i]; *c; ++c) x.var_seq.push_back((uint8_t)(strchr("ACGT", *c) - "ACGT"));
        x.allele[0] = allele0[i];
        v->variants.push_back(x);
    }
    v->bm = VariantBiasMod(1, 2);                       // Simulator::VariantBiasVarModifiers bias_mod(1, 2)
    return v;
}
void orc_var_free(void *h) { delete static_cast<VarScenario *>(h); }
// Reference::ReferenceSequence(insert_string, seq_id, start_pos, frag_length, reversed, variants, first_variant, allele) (Reference.cpp:498-567) on the scenario's
// sequence and variants: base codes into out[frag_length]; the number of bases, -1 for an error.  Pinned to ReferenceTest.cpp:283-326.
int orc_var_reference_sequence(void *h, uint32_t start_pos, uint32_t frag_length, int reversed, int32_t first_variant_id, uint32_t first_variant_pos, uint32_t allele, uint8_t *out) {
    const VarScenario &v = *static_cast<VarScenario *>(h);
    try {
        const std::vector<uint8_t> seq = reference_sequence_with_variants(v.ref(), start_pos, frag_length, reversed != 0, {first_variant_id, first_variant_pos}, allele);
        memcpy(out, seq.data(), seq.size());
        return (int)seq.size();
    } catch (const std::exception &e) {
        g_error = e.what();
        return -1;
    }
}
void orc_var_set_first_variant(void *h, int32_t id) { static_cast<VarScenario *>(h)->bm.first_variant_id = id; }
void orc_var_get_start(void *h, int32_t *first_variant_id, uint32_t *start_variant_pos) {
    const VarScenario &v = *static_cast<VarScenario *>(h);
    *first_variant_id = v.bm.first_variant_id;
    *start_variant_pos = v.bm.start_variant_pos;
}
// PrepareBiasModForCurrentStartPos with the reference surrounding of cur_start; sur_out: [2 alleles][3]; ref_sur_out / comp_sur_out[3]
int orc_var_prepare_start(void *h, uint32_t cur_start, uint32_t first_fragment_length, uint32_t comp_pos, uint32_t *sur_out, uint32_t *ref_sur_out, uint32_t *comp_sur_out) {
    return guard([&] {
        VarScenario &v = *static_cast<VarScenario *>(h);
        const Sur3 start = v.ref().forward_surrounding(cur_start);
        prepare_bias_mod_for_current_start_pos(v.bm, v.ref(), cur_start, first_fragment_length, start);
        const Sur3 comp = v.comp().forward_surrounding(comp_pos);
        for (int k = 0; k < 3; ++k) {
            sur_out[k] = (uint32_t)v.bm.surrounding_start[0].b[k];
            sur_out[3 + k] = (uint32_t)v.bm.surrounding_start[1].b[k];
            ref_sur_out[k] = (uint32_t)start.b[k];
            comp_sur_out[k] = (uint32_t)comp.b[k];
        }
        return 0;
    });
}
// TestVariationInInnerLoopOfSimulateFromGivenBlock: log[allele][fragment length - from] = {unhandled_variant_id, unhandled_bases, gc_mod, end_pos_shift};
// returns the number of (allele, strand, length) cases whose end position lies inside the sequence, or -1 - (number of property mismatches)
int orc_var_inner_loop(void *h, uint32_t cur_start, uint32_t from, uint32_t to, const uint32_t *modified_start_pos /*[2]*/, const int32_t *use_comp /*[2]: 0 ref, 1 comp, -1 none*/,
                       int32_t *log /*[2][to-from][4]*/, uint32_t *n_possible) {
    int32_t result = 0;
    const int rc = guard([&] {
        VarScenario &v = *static_cast<VarScenario *>(h);
        const VarRef ref = v.ref();
        const std::vector<uint32_t> possible = possible_alleles(ref, v.bm, cur_start);
        *n_possible = (uint32_t)possible.size();
        int mismatches = 0, tests = 0;
        for (uint32_t fl = from; fl < to; ++fl)
            for (uint32_t chosen = 0; chosen < 2 * possible.size(); ++chosen) {          // every strand of every possible allele, in id order
                const uint32_t allele = possible[chosen / 2];
                prepare_bias_mod_for_current_fragment_length(v.bm, ref, cur_start, fl, allele);
                const VarRef comp = use_comp[allele] == 1 ? v.comp() : ref;
                const VarRef comp_plain = [&] {
                    VarRef r = comp;
                    r.variants = &v.none;
                    return r;
                }();
                const Sur3 want = comp_plain.reverse_surrounding(modified_start_pos[allele] + fl - 1u);
                for (int k = 0; k < 3; ++k) mismatches += want.b[k] != v.bm.surrounding_end[allele].b[k];
                int32_t *row = log + ((size_t)allele * (to - from) + (fl - from)) * 4;
                row[0] = v.bm.unhandled_variant_id[allele];
                row[1] = (int32_t)v.bm.unhandled_bases_in_variant[allele];
                row[2] = v.bm.gc_mod[allele];
                row[3] = v.bm.end_pos_shift[allele];
                const uint32_t cur_end = cur_start + fl + (uint32_t)v.bm.end_pos_shift[allele];
                if (cur_end < ref.length()) {
                    const uint32_t gc_perc = gc_percent_with_variants(v.bm, ref, cur_end, fl, allele);
                    mismatches += gc_perc != percent_u32(comp_plain.gc_content_absolut(modified_start_pos[allele], modified_start_pos[allele] + fl), fl);
                    const uint32_t tlen = std::min(fl, 100u + 0u);                      // ReadLengths().to() + MaxLenDeletion of the reference test's DataStats
                    const std::vector<uint8_t> fwd = reference_sequence_with_variants(ref, cur_start, std::min(fl, tlen), false, v.bm.start_variant(), allele);
                    const std::vector<uint8_t> rev = reference_sequence_with_variants(ref, cur_end, std::min(fl, tlen), true, v.bm.end_variant(v.variants, cur_end, allele), allele);
                    const std::vector<uint8_t> &cc = *comp_plain.codes;
                    for (uint32_t k = 0; k < fl; ++k) {
                        mismatches += k >= fwd.size() || fwd[k] != cc[modified_start_pos[allele] + k];
                        mismatches += k >= rev.size() || rev[k] != 3u - cc[modified_start_pos[allele] + fl - 1u - k];
                    }
                    ++tests;
                }
            }
        result = mismatches ? -1 - mismatches : tests;
        return 0;
    });
    return rc ? -1000000 : result;
}
int orc_var_check_inserted(void *h, uint32_t cur_start) {
    return guard([&] {
        VarScenario &v = *static_cast<VarScenario *>(h);
        check_for_inserted_bases_to_start_from(v.bm, v.ref(), cur_start);
        return 0;
    });
}

// utilitiesTest.cpp:61-137: DominantBaseWithMemory driven by a script: op 0 Clear, 1 Set(seq, arg), 2 Update(seq[arg]), 3 copy from the other object;
// out[i] = Get() of object which[i] after op i
void orc_dombase_memory_script(const uint8_t *seq, uint32_t len, uint32_t n_ops, const uint8_t *which, const uint8_t *op, const uint32_t *arg, uint8_t *out) {
    DomMem obj[2];
    for (uint32_t i = 0; i < n_ops; ++i) {
        DomMem &d = obj[which[i]];
        if (0 == op[i]) d.clear();
        else if (1 == op[i]) d.set([&](uint32_t p) { return seq[p]; }, arg[i]);
        else if (2 == op[i]) d.update(seq[arg[i]]);
        else d = obj[1 - which[i]];
        out[i] = d.get();
        (void)len;
    }
}

// ---- simulation with variants
int orc_var_attach(orc_sim *s, const orc_variants *vs) {
    return guard([&] {
        std::unique_ptr<VarState> st(new VarState());
        st->num_alleles = vs->num_alleles;
        st->seqs.resize(s->r->n_seqs);
        for (uint32_t i = 0; i < s->r->n_seqs; ++i) {
            SeqVars &sv = st->seqs[i];
            sv.codes.assign(s->r->codes[i], s->r->codes[i] + s->r->len[i]);
            sv.variants = convert_variants(vs, i);
            if (!s->n_blocks[i]) continue;
            sys_error_variants_reverse(s, i, sv, st->num_alleles);          // CreateUnit: all reverse blocks first
            sys_error_variants_forward(s, i, sv, st->num_alleles);
        }
        s->var_state = st.release();
        s->variants_source = vs;
        return 0;
    });
}
void orc_var_detach(orc_sim *s) {
    delete static_cast<VarState *>(s->var_state);
    s->var_state = nullptr;
}
uint32_t orc_var_sys_errors(const orc_sim *s, int strand, uint32_t seq, uint32_t var_id, uint8_t *dom, uint8_t *rate, uint32_t cap) {
    const auto &e = state_of(s).seqs[seq].err[strand][var_id];
    for (uint32_t i = 0; i < e.size() && i < cap; ++i) {
        dom[i] = e[i].first;
        rate[i] = e[i].second;
    }
    return (uint32_t)e.size();
}

// Simulator.cpp:2249-2357 with VariantsLoaded().  Streams: c1 = sequence | sub << 22 where sub counts the extra passes of the
// do-while loop at one start position (starts inside inserted bases); the cell uniform as without variants; SelectAllele's j-th
// random value: word j&3 of block (start, c1, length, 1<<28 | 2 + (j>>2)); the count uniform of the j-th chosen strand: u53 of words
// 2(j&1), 2(j&1)+1 of block (start, c1, length, 1<<28 | 128 + (j>>1)).
// property check in the spirit of SimulatorTest::TestVariationInInnerLoopOfSimulateFromGivenBlock (SimulatorTest.cpp:116-195), for every
// cell the sieve evaluates: GC percent, start / end surrounding and both templates of the allele equal what the sequence WITH the
// allele's variants applied gives at the corresponding position.  Cells whose 30-base windows would wrap around a sequence end are
// left out (the variant edits stop at the ends).  g_hap[0] = cells compared, [1..5] = mismatches of gc, start surrounding, end
// surrounding, forward template, reverse template.
static uint64_t g_hap[6] = {0, 0, 0, 0, 0, 0};
static bool g_hap_check = false;
struct Haplotype {
    std::vector<uint8_t> seq;
    std::vector<uint32_t> at;              // at[p]: index in seq of reference position p (of the variant's first base; of the next kept base if p is deleted)
};
static Haplotype make_haplotype(const SeqVars &sv, uint32_t allele) {
    Haplotype h;
    const uint32_t L = (uint32_t)sv.codes.size();
    h.at.resize(L + 1);
    size_t vi = 0;
    for (uint32_t p = 0; p < L; ++p) {
        h.at[p] = (uint32_t)h.seq.size();
        const Variant *use = nullptr;
        for (; vi < sv.variants.size() && sv.variants[vi].position == p; ++vi)
            if (sv.variants[vi].in_allele(allele) && !use) use = &sv.variants[vi];
        if (!use) h.seq.push_back(sv.codes[p]);
        else h.seq.insert(h.seq.end(), use->var_seq.begin(), use->var_seq.end());
    }
    h.at[L] = (uint32_t)h.seq.size();
    return h;
}
void orc_var_haplotype_check(int enable) { g_hap_check = enable != 0; }
void orc_var_haplotype_counters(uint64_t *out /*[6]*/) {
    for (int k = 0; k < 6; ++k) {
        out[k] = g_hap[k];
        g_hap[k] = 0;
    }
}

static uint64_t g_scratch_checks = 0, g_scratch_mismatches = 0;
// property check behind the device's design: the bookkeeping the reference updates incrementally over fragment lengths is a pure
// function of (start, pass, length, allele) -- a fresh VariantBiasVarModifiers taken straight to the length gives the same values
void orc_var_scratch_counters(uint64_t *checks, uint64_t *mismatches) {
    *checks = g_scratch_checks;
    *mismatches = g_scratch_mismatches;
    g_scratch_checks = g_scratch_mismatches = 0;
}

uint64_t orc_sieve_blocks_var(const orc_sim *s, uint32_t block_lo, uint32_t block_hi, orc_fragment_var **out) {
    std::vector<orc_fragment_var> frags;
    const int rc = guard([&] {
        const orc_profile *p = s->p;
        const orc_reference *r = s->r;
        const VarState &st = state_of(s);
        const uint32_t A = st.num_alleles, to = s->insert_to;
        const uint32_t frag_len_start = (uint32_t)(p->insert_lengths.from > 1 ? p->insert_lengths.from : 1);
        for (uint32_t seq = 0; seq < r->n_seqs; ++seq) {
            if (!s->n_blocks[seq]) continue;
            if (seq >= (1u << 22)) throw Error("too many sequences for the variant streams");
            const SeqVars &sv = st.seqs[seq];
            VarRef ref;
            ref.codes = &sv.codes;
            ref.variants = &sv.variants;
            ref.num_alleles = A;
            const uint32_t L = r->len[seq];
            const double *thr = &s->thresholds[(size_t)s->coverage_groups[seq] * to * 2];
            std::vector<double> gap_q(to);
            std::vector<uint32_t> gap_seg_end(to);
            std::vector<orc_gap_hit> passing(to);
            orc_gap_table(s, s->coverage_groups[seq], gap_q.data(), gap_seg_end.data());
            std::vector<Haplotype> haps;
            if (g_hap_check)
                for (uint32_t a = 0; a < A; ++a) haps.push_back(make_haplotype(sv, a));
            for (uint32_t b = 0; b < s->n_blocks[seq]; ++b) {
                const uint32_t block_id = s->first_block[seq] + b;
                if (block_id < block_lo || block_id >= block_hi) continue;
                const uint32_t block_start = b * kBlock;
                uint32_t read_number = 0;
                VariantBiasMod bm((int32_t)sv.lb(block_start), A);          // block.first_variant_id_ (CreateBlock :1206-1209)
                Sur3 sur_start = ref.forward_surrounding(0 < block_start ? block_start - 1u : L - 1u);
                for (uint32_t start = block_start; start < block_start + kBlock && start < L; ++start) {
                    orc_surrounding_update_forward(sv.codes.data(), L, start, sur_start.b);
                    uint32_t sub = 0;
                    do {
                        if (sub >= 1024u) throw Error("too many starts inside inserted bases at one position for the variant streams");
                        const uint32_t c1 = seq | (sub << 22);
                        prepare_bias_mod_for_current_start_pos(bm, ref, start, frag_len_start, sur_start);
                        const std::vector<uint32_t> possible = possible_alleles(ref, bm, start);
                        const uint32_t n_passing = orc_gap_hits(s, gap_q.data(), gap_seg_end.data(), thr, start, c1, passing.data());   // the cells that pass :2304-2306
                        for (uint32_t hit = 0; hit < n_passing; ++hit) {
                            const uint32_t len = passing[hit].len;
                            const double probability_chosen = passing[hit].probability_chosen;
                            const uint16_t non_zero_strands = orc_binomial((uint16_t)(2 * possible.size()), 1 - thr[2 * len], probability_chosen);
                            if (!non_zero_strands) continue;
                            // ChooseAlleles (:1387-1397)
                            const uint16_t possible_strands = (uint16_t)(2 * possible.size());
                            std::vector<uint16_t> chosen(possible_strands + 1u);
                            std::vector<uint8_t> reverse_selection(possible_strands, 1);
                            uint32_t n_chosen = 0, n_draws = 0;
                            const uint16_t to_draw = non_zero_strands <= possible_strands / 2 ? non_zero_strands : (uint16_t)(possible_strands - non_zero_strands);
                            while (n_chosen < to_draw) {
                                const orc_philox_out ws = orc_philox4x32_10(s->seed, start, c1, len, ((uint32_t)ORC_DOM_SIEVE << 28) | (2u + (n_draws >> 2)));
                                orc_select_allele(chosen.data(), &n_chosen, reverse_selection.data(), possible_strands, orc_u32(ws.w[n_draws & 3u]));
                                ++n_draws;
                            }
                            if (!(non_zero_strands <= possible_strands / 2)) {                       // ReverseSelection :1373-1385
                                std::fill(reverse_selection.begin(), reverse_selection.end(), 1);
                                for (uint32_t j = 0; j < n_chosen; ++j) reverse_selection[chosen[j]] = 0;
                                n_chosen = 0;
                                for (uint16_t id = 0; id < possible_strands; ++id)
                                    if (reverse_selection[id]) chosen[n_chosen++] = id;
                            }
                            for (uint32_t j = 0; j < n_chosen; ++j) {
                                const uint32_t allele = possible.at(chosen[j] / 2u);
                                const uint8_t strand = (uint8_t)(chosen[j] % 2u);
                                prepare_bias_mod_for_current_fragment_length(bm, ref, start, len, allele);
                                const uint32_t cur_end = start + len + (uint32_t)bm.end_pos_shift.at(allele);
                                if (!(cur_end < L)) continue;
                                const uint8_t gc_perc = (uint8_t)gc_percent_with_variants(bm, ref, cur_end, len, allele);
                                {
                                    VariantBiasMod fresh(bm.first_variant_id, A);
                                    fresh.start_variant_pos = bm.start_variant_pos;
                                    prepare_bias_mod_for_current_start_pos(fresh, ref, start, frag_len_start, sur_start);
                                    prepare_bias_mod_for_current_fragment_length(fresh, ref, start, len, allele);
                                    bool same = fresh.end_pos_shift[allele] == bm.end_pos_shift[allele] && fresh.gc_mod[allele] == bm.gc_mod[allele] &&
                                                fresh.unhandled_variant_id[allele] == bm.unhandled_variant_id[allele] &&
                                                fresh.unhandled_bases_in_variant[allele] == bm.unhandled_bases_in_variant[allele];
                                    for (int k = 0; k < 3; ++k)
                                        same = same && fresh.surrounding_start[allele].b[k] == bm.surrounding_start[allele].b[k] &&
                                               fresh.surrounding_end[allele].b[k] == bm.surrounding_end[allele].b[k];
                                    same = same && gc_percent_with_variants(fresh, ref, cur_end, len, allele) == gc_perc && fresh.end_variant(sv.variants, cur_end, allele) == bm.end_variant(sv.variants, cur_end, allele);
                                    ++g_scratch_checks;
                                    g_scratch_mismatches += same ? 0 : 1;
                                }
                                if (g_hap_check) {
                                    const Haplotype &h = haps[allele];
                                    const uint32_t hs = h.at[start] + bm.start_variant_pos, Lh = (uint32_t)h.seq.size();
                                    if (hs >= kSurStart && hs + len + kSurLength < Lh && hs + len >= kSurLength) {
                                        ++g_hap[0];
                                        uint32_t gc = 0;
                                        for (uint32_t k = 0; k < len; ++k) gc += is_gc(h.seq[hs + k]) ? 1 : 0;
                                        g_hap[1] += orc_percent_u32(gc, len) != gc_perc;
                                        int32_t ss[3], se[3];
                                        orc_surrounding_forward(h.seq.data(), Lh, hs, ss);
                                        orc_surrounding_reverse(h.seq.data(), Lh, hs + len - 1u, se);
                                        g_hap[2] += memcmp(ss, bm.surrounding_start.at(allele).b, sizeof ss) != 0;
                                        g_hap[3] += memcmp(se, bm.surrounding_end.at(allele).b, sizeof se) != 0;
                                        const uint32_t tl = std::min<uint32_t>(len, 40u);
                                        const std::vector<uint8_t> fwd = reference_sequence_with_variants(ref, start, tl, false, bm.start_variant(), allele);
                                        const std::vector<uint8_t> rev = reference_sequence_with_variants(ref, cur_end, tl, true, bm.end_variant(sv.variants, cur_end, allele), allele);
                                        bool f_ok = fwd.size() == tl, r_ok = rev.size() == tl;
                                        for (uint32_t k = 0; k < tl && f_ok; ++k) f_ok = fwd[k] == h.seq[hs + k];
                                        for (uint32_t k = 0; k < tl && r_ok; ++k) r_ok = rev[k] == 3u - h.seq[hs + len - 1u - k];
                                        g_hap[4] += f_ok ? 0 : 1;
                                        g_hap[5] += r_ok ? 0 : 1;
                                    }
                                }
                                const orc_philox_out wc = orc_philox4x32_10(s->seed, start, c1, len, ((uint32_t)ORC_DOM_SIEVE << 28) | (128u + (j >> 1)));
                                const double adjusted_random = thr[2 * len] + orc_u53(wc.w[2 * (j & 1u)], wc.w[2 * (j & 1u) + 1]) * (1 - thr[2 * len]);
                                const uint16_t counts = orc_get_fragment_counts(p, s->bias_normalization, s->ref_seq_bias[seq], len, gc_perc, bm.surrounding_start.at(allele).b,
                                                                                bm.surrounding_end.at(allele).b, adjusted_random, (uint16_t)A);
                                if (!counts) continue;
                                const std::pair<int32_t, uint32_t> sv_start = bm.start_variant(), sv_end = bm.end_variant(sv.variants, cur_end, allele);
                                for (uint16_t dup = 0; dup < counts; ++dup) {
                                    orc_fragment_var f;
                                    memset(&f, 0, sizeof f);
                                    f.seq = seq;
                                    f.start = start;
                                    f.len = len;
                                    f.dup = dup;
                                    f.strand = strand;
                                    f.allele = (uint8_t)allele;
                                    f.block = block_id;
                                    f.number = ++read_number;
                                    f.end = cur_end;
                                    f.sub = sub;
                                    f.start_var = sv_start.first;
                                    f.start_var_pos = sv_start.second;
                                    f.end_var = sv_end.first;
                                    f.end_var_pos = sv_end.second;
                                    frags.push_back(f);
                                }
                            }
                        }
                        check_for_inserted_bases_to_start_from(bm, ref, start);
                        ++sub;
                    } while (bm.start_variant_pos);
                }
            }
        }
        return 0;
    });
    if (rc) {
        *out = nullptr;
        return UINT64_MAX;
    }
    *out = (orc_fragment_var *)malloc((frags.size() ? frags.size() : 1) * sizeof(orc_fragment_var));
    memcpy(*out, frags.data(), frags.size() * sizeof(orc_fragment_var));
    return frags.size();
}

// Simulator.cpp:634-721 CreateReads with start / end variants, :596-632 CreateReadId, :1898-1923 GetOrgSeq.  Pair streams as without
// variants plus c1 = sequence | sub << 22 and the allele in bits 17..24 of c3.
int orc_create_reads_var(const orc_sim *s, const orc_fragment_var *frags, uint64_t n, orc_text *r1, orc_text *r2) {
    return guard([&] {
        const orc_profile *p = s->p;
        const orc_reference *r = s->r;
        const VarState &st = state_of(s);
        std::unique_ptr<orc_read[]> rd(new orc_read[2]);
        orc_text *dst[2] = {r1, r2};
        for (uint64_t i = 0; i < n; ++i) {
            const orc_fragment_var &f = frags[i];
            const SeqVars &sv = st.seqs[f.seq];
            VarRef ref;
            ref.codes = &sv.codes;
            ref.variants = &sv.variants;
            ref.num_alleles = st.num_alleles;
            const uint32_t strand = f.strand, L = r->len[f.seq];
            std::vector<uint8_t> tmpl[2];
            for (uint32_t which = 0; which < 2; ++which) {
                const uint32_t sg = which ? !strand : strand;
                const uint64_t rl_to = p->read_lengths[sg].from + p->read_lengths[sg].size;
                const uint32_t tl = (uint32_t)(f.len < rl_to + p->max_len_deletion ? f.len : rl_to + p->max_len_deletion);
                tmpl[sg] = which ? reference_sequence_with_variants(ref, f.end, tl, true, {f.end_var, f.end_var_pos}, f.allele)
                                 : reference_sequence_with_variants(ref, f.start, tl, false, {f.start_var, f.start_var_pos}, f.allele);
            }
            const uint32_t c1 = f.seq | (f.sub << 22), c2 = f.len | ((uint32_t)f.dup << 16);
            if (s->meth_n) {                                             // CTConversion's dispatcher :2219-2247: the forward template, then the reverse one
                const int32_t cur_methylation_start = (int32_t)orc_methylation_start(s, f.seq, f.start);
                for (uint32_t reversed = 0; reversed < 2; ++reversed) {
                    const MethDraw draw{s, f.start, c1, f.len, (7u << 28) | (reversed << 27) | ((uint32_t)f.allele << 17)};
                    if (!reversed) ct_conversion_variants(s, tmpl[strand], f.seq, f.start, f.allele, cur_methylation_start, false, sv.variants, {f.start_var, f.start_var_pos}, draw);
                    else ct_conversion_variants(s, tmpl[!strand], f.seq, f.end, f.allele, cur_methylation_start, true, sv.variants, {f.end_var, f.end_var_pos}, draw);
                }
            }
            uint16_t tile_id = 0;
            if (1 < p->n_tiles)
                tile_id = (uint16_t)discrete_draw(p->tile_cp, p->n_tiles, orc_u32(orc_philox4x32_10(s->seed, f.start, c1, c2, pair_c3(ORC_DOM_PAIR, strand, 2, f.allele)).w[0]));
            // the blocks and the positions in them (:649-688)
            const uint32_t start_block = f.start / kBlock;
            uint32_t end_block = start_block;
            while (end_block * kBlock + std::min(kBlock, L - end_block * kBlock) < f.end) ++end_block;
            Walk walk[2];
            for (uint32_t seg = 0; seg < 2; ++seg) {
                Walk &w = walk[seg];
                w.s = s;
                w.sv = &sv;
                w.seq = f.seq;
                w.allele = f.allele;
                if (seg == strand) {                                     // block.at(strand) = start_block
                    w.strand = 0;
                    w.block0 = start_block;
                    w.block_pos0 = f.start - start_block * kBlock;
                    w.cur_var0 = f.start_var - (int32_t)sv.lb(start_block * kBlock);
                    w.var_pos0 = f.start_var_pos;
                } else {
                    w.strand = 1;
                    w.block0 = end_block;
                    const uint32_t bs = end_block * kBlock, size = std::min(kBlock, L - bs);
                    w.block_pos0 = bs + size - f.end;
                    w.cur_var0 = ((int32_t)sv.lb(bs + kBlock) - 1) - f.end_var;
                    w.var_pos0 = f.end_var_pos ? (uint32_t)sv.variants.at((size_t)f.end_var).var_seq.size() - f.end_var_pos : 0u;
                }
                w.reset();
            }
            for (uint32_t seg = 2; seg--;) {
                orc_sys_cursor cur = {&walk[seg], walk_reset, walk_next, walk_deleted};
                orc_stream stream = {s->seed, f.start, c1, c2, pair_c3(ORC_DOM_PAIR, strand, seg, f.allele)};
                orc_fill_read_cursor(s, &rd[seg], (uint8_t)seg, tile_id, f.len, tmpl[seg].data(), (uint32_t)tmpl[seg].size(), &cur, &stream);
            }
            const uint32_t print_start = strand ? f.end : f.start + 1u, print_end = strand ? f.start + 1u : f.end;
            for (uint32_t seg = 0; seg < 2; ++seg) {
                char id[8192], allele_part[32] = "";
                if (print_start && 1 < st.num_alleles) snprintf(allele_part, sizeof allele_part, "_allele%u", (unsigned)f.allele);     // :612-614
                snprintf(id, sizeof id, "%s%u_%u%s:%u:%s:%u:%u:1337:1337 %s E%u", s->base_identifier, f.block, f.number, allele_part, print_start, r->first_name[f.seq],
                         print_end, (unsigned)p->tiles[tile_id], rd[seg].cigar, (unsigned)rd[seg].num_errors);
                orc_text_append_record(dst[seg], id, &rd[seg]);
            }
        }
        return 0;
    });
}

}  // extern "C"
