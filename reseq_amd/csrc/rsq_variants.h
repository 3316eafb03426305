// rsq_variants.h -- variants of any kind (substitutions, insertions, deletions) on the device: the per-allele modifiers of one sieve
// cell, derived from scratch.
//
// The reference keeps VariantBiasVarModifiers (Simulator.h:29-89) up to date while SimulateFromGivenBlock walks start positions and
// fragment lengths (Simulator.cpp:1399-1896).  What it holds for (start position, pass at that position, fragment length, allele) does
// not depend on the lengths visited before (the oracle checks this for every cell it evaluates: orc_var_scratch_counters), so a lane
// that owns one cell derives it directly: PrepareBiasModForCurrentStartPos for its allele, then one
// PrepareBiasModForCurrentFragmentLength from the first fragment length to its own.  Same statements, same variable widths as the
// reference; per-allele vectors become the scalars of one allele.  Also here: Reference::ReferenceSequence with variants
// (Reference.cpp:498-567) writing a 2-bit template, and the extra passes at a start position (starts inside inserted bases,
// CheckForInsertedBasesToStartFrom :1870-1896) as slots of the sieve.
#pragma once
#include "rsq_core.h"

namespace rsq {

// one sequence with its variants
struct VarView {
    const uint64_t *words;              // the reference, 2 bits per base
    const uint32_t *gc_prefix;
    uint64_t word_off;
    uint32_t L;
    const DevVariant *v;                // sorted by position
    uint32_t n;
    const uint8_t *bases;               // var_seq_ of all variants (DevVariant::off)
    RSQ_HD uint32_t at(uint32_t pos) const { return ref_base(words, word_off, pos); }
    RSQ_HD bool gc(uint32_t pos) const { return is_gc(at(pos)); }
    RSQ_HD uint32_t base(const DevVariant &var, uint32_t k) const { return bases[var.off + k]; }
    RSQ_HD bool in_allele(const DevVariant &var, uint32_t allele) const { return (var.allele[allele >> 6] >> (allele & 63u)) & 1u; }
    RSQ_HD uint32_t lower_bound(uint32_t pos) const {
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (v[mid].pos < pos) lo = mid + 1u;
            else hi = mid;
        }
        return lo;
    }
};
RSQ_HD VarView var_view(const DevSim &S, uint32_t seq) {
    return VarView{S.ref_words, S.gc_prefix, S.seq_word_off[seq], S.seq_len[seq], S.variants + S.var_ptr[seq], S.var_ptr[seq + 1] - S.var_ptr[seq], S.var_bases};
}

// views of a variant's bases for the surrounding edits
struct VarBases {                       // var_seq_[from ...)
    const uint8_t *p;
    RSQ_HD uint32_t operator[](uint32_t i) const { return p[i]; }
};
struct VarBasesRC {                     // ReverseComplementorDna of var_seq_[from, from + n)
    const uint8_t *p;
    uint32_t n;
    RSQ_HD uint32_t operator[](uint32_t i) const { return 3u - p[n - 1u - i]; }
};

// the start of a pass at a start position: bias_mod.first_variant_id_, start_variant_pos_
struct VarStart {
    int32_t first_variant_id;
    uint32_t start_variant_pos;
};
// one allele's share of VariantBiasVarModifiers
struct AlleleMod {
    int32_t unhandled_variant_id;
    uint32_t unhandled_bases_in_variant;
    int32_t gc_mod, end_pos_shift;
    uint32_t last_end_position;
    uint32_t surrounding_start[3], surrounding_end[3];
};

// Simulator.h:401-412
RSQ_HD bool allele_skipped(const VarView &r, const VarStart &st, uint32_t allele, uint32_t cur_start_position) {
    if ((uint32_t)st.first_variant_id < r.n && r.v[st.first_variant_id].pos == cur_start_position) {
        const DevVariant &var = r.v[st.first_variant_id];
        if (0u == var.len) return r.in_allele(var, allele);
        if (st.start_variant_pos) return !r.in_allele(var, allele);
    }
    return false;
}
// :1399-1402
RSQ_HD bool variant_inside_current_fragment(const VarView &r, int32_t cur_var_id, uint32_t cur_end_position, int32_t end_pos_shift) {
    return r.n > (uint32_t)cur_var_id && r.v[cur_var_id].pos < cur_end_position + (uint32_t)end_pos_shift;
}
// :1404-1455
RSQ_HD void handle_gc_mod_and_end_pos_shift_for_new_variants(AlleleMod &m, uint32_t allele, const VarView &r, uint32_t cur_end_position) {
    while (variant_inside_current_fragment(r, m.unhandled_variant_id, cur_end_position, m.end_pos_shift) && 0u == m.unhandled_bases_in_variant) {
        const DevVariant &var = r.v[m.unhandled_variant_id];
        if (!r.in_allele(var, allele)) {
            ++m.unhandled_variant_id;
            continue;
        }
        for (uint32_t pos = 0; pos < var.len && var.pos + pos < cur_end_position + (uint32_t)m.end_pos_shift; ++pos)
            if (is_gc(r.base(var, pos))) ++m.gc_mod;
        if (r.gc(var.pos)) --m.gc_mod;
        if (0u == var.len) {
            ++m.end_pos_shift;
            ++m.unhandled_variant_id;
        } else if (1u == var.len) {
            ++m.unhandled_variant_id;
        } else if (var.pos + var.len <= cur_end_position + (uint32_t)m.end_pos_shift) {
            m.end_pos_shift -= (int32_t)(var.len - 1u);
            ++m.unhandled_variant_id;
        } else {
            m.unhandled_bases_in_variant = var.pos + var.len - (cur_end_position + (uint32_t)m.end_pos_shift);
            m.end_pos_shift -= (int32_t)(var.len - m.unhandled_bases_in_variant - 1u);
        }
    }
}

RSQ_HD void sur_change(uint32_t (&s)[3], int32_t pos, uint32_t base) { sur_change_base(s, (uint32_t)(uint16_t)pos, base); }
// :1459-1527
RSQ_HD void handle_surrounding_variants_before_center(uint32_t (&sur)[3], uint32_t center_position, int32_t initial_pos_shift, int32_t center_var, uint32_t allele,
                                                      const VarView &r, bool reverse) {
    int32_t cur_var = center_var, pos_shift = initial_pos_shift;
    while (0 <= --cur_var) {
        const DevVariant &var = r.v[cur_var];
        const int32_t sur_pos = reverse ? (int32_t)(center_position - var.pos + (uint32_t)pos_shift) : (int32_t)(var.pos - center_position + (uint32_t)pos_shift);
        if (0 > sur_pos || (int32_t)kSurLength <= sur_pos) break;
        if (!r.in_allele(var, allele)) continue;
        if (0u == var.len) {
            if (reverse) {
                const int32_t new_base_pos = (int32_t)center_position + pos_shift - (int32_t)kSurLength;
                sur_delete_shift_right(sur, (uint32_t)(uint16_t)sur_pos, 3u - r.at(0 > new_base_pos ? (uint32_t)((int32_t)r.L + new_base_pos) : (uint32_t)new_base_pos));
                --pos_shift;
            } else {
                if ((uint32_t)++pos_shift > center_position) sur_delete_shift_left(sur, (uint32_t)(uint16_t)sur_pos, r.at(r.L + center_position - (uint32_t)pos_shift));
                else sur_delete_shift_left(sur, (uint32_t)(uint16_t)sur_pos, r.at(center_position - (uint32_t)pos_shift));
            }
        } else {
            sur_change(sur, sur_pos, reverse ? 3u - r.base(var, 0) : r.base(var, 0));
            if (1u < var.len) {
                if (reverse) {
                    sur_insert_shift_right(sur, (uint32_t)(uint16_t)sur_pos, VarBasesRC{r.bases + var.off + 1u, var.len - 1u}, var.len - 1u);
                    pos_shift += (int32_t)var.len - 1;
                } else {
                    sur_insert_shift_left(sur, (uint32_t)(uint16_t)sur_pos, VarBases{r.bases + var.off + 1u}, var.len - 1u);
                    pos_shift -= (int32_t)var.len - 1;
                }
            }
        }
    }
}
// :1529-1589
RSQ_HD void handle_surrounding_variants_after_center(uint32_t (&sur)[3], uint32_t center_position, int32_t initial_pos_shift, int32_t center_var, uint32_t allele,
                                                     const VarView &r, bool reverse) {
    int32_t pos_shift = initial_pos_shift;
    for (int32_t cur_var = center_var; (uint32_t)cur_var < r.n; ++cur_var) {
        const DevVariant &var = r.v[cur_var];
        const int32_t sur_pos = reverse ? (int32_t)(center_position - var.pos + (uint32_t)pos_shift) : (int32_t)(var.pos - center_position + (uint32_t)pos_shift);
        if (0 > sur_pos || (int32_t)kSurLength <= sur_pos) break;
        if (!r.in_allele(var, allele)) continue;
        if (0u == var.len) {
            if (reverse) {
                ++pos_shift;
                sur_delete_shift_left(sur, (uint32_t)(uint16_t)sur_pos, 3u - r.at((center_position + (uint32_t)pos_shift) % r.L));
            } else {
                sur_delete_shift_right(sur, (uint32_t)(uint16_t)sur_pos, r.at((center_position - (uint32_t)pos_shift + kSurLength) % r.L));
                --pos_shift;
            }
        } else {
            sur_change(sur, sur_pos, reverse ? 3u - r.base(var, 0) : r.base(var, 0));
            if (1u < var.len) {
                if (reverse) {
                    if (sur_pos) {
                        sur_insert_shift_left(sur, (uint32_t)(uint16_t)(sur_pos - 1), VarBasesRC{r.bases + var.off + 1u, var.len - 1u}, var.len - 1u);
                        pos_shift -= (int32_t)var.len - 1;
                    }
                } else if (sur_pos + 1 < (int32_t)kSurLength) {
                    sur_insert_shift_right(sur, (uint32_t)(uint16_t)(sur_pos + 1), VarBases{r.bases + var.off + 1u}, var.len - 1u);
                    pos_shift += (int32_t)var.len - 1;
                }
            }
        }
    }
}
// :1591-1636; sur starts as the reference's forward surrounding of the position
RSQ_HD void variant_mod_start_surrounding(uint32_t (&sur)[3], const VarStart &st, uint32_t allele, const VarView &r, uint32_t cur_start_position) {
    if (!r.n) return;
    int32_t pos_shift = (int32_t)kSurStart;
    int32_t cur_var = st.first_variant_id;
    if (st.start_variant_pos) {
        const DevVariant &var = r.v[cur_var];
        sur_change(sur, pos_shift, r.base(var, 0));
        const uint32_t upto = st.start_variant_pos + 1u < var.len ? st.start_variant_pos + 1u : var.len;      // infix(var_seq_, 1, start_variant_pos_ + 1)
        sur_insert_shift_left(sur, (uint32_t)pos_shift, VarBases{r.bases + var.off + 1u}, upto - 1u);
        pos_shift -= (int32_t)st.start_variant_pos;
    }
    handle_surrounding_variants_before_center(sur, cur_start_position, pos_shift, cur_var, allele, r, false);
    pos_shift = (int32_t)kSurStart;
    cur_var = st.first_variant_id;
    if (st.start_variant_pos) {
        const DevVariant &var = r.v[cur_var];
        if (var.len > st.start_variant_pos + 1u) {
            if (pos_shift + 1 < (int32_t)kSurLength) {
                sur_insert_shift_right(sur, (uint32_t)(pos_shift + 1), VarBases{r.bases + var.off + st.start_variant_pos + 1u}, var.len - st.start_variant_pos - 1u);
                pos_shift += (int32_t)var.len - (int32_t)st.start_variant_pos - 1;
            }
        }
        ++cur_var;
    }
    handle_surrounding_variants_after_center(sur, cur_start_position, pos_shift, cur_var, allele, r, false);
}
// :1638-1698 for one allele (the skipped-allele test is the caller's); surrounding_start: the reference's forward surrounding of the position
RSQ_HD void prepare_bias_mod_for_current_start_pos(AlleleMod &m, const VarStart &st, uint32_t allele, const VarView &r, uint32_t cur_start_position,
                                                   uint32_t first_fragment_length, const uint32_t (&surrounding_start)[3]) {
    const uint32_t cur_end_position = cur_start_position + first_fragment_length - 1u;
    m.unhandled_variant_id = st.first_variant_id;
    m.unhandled_bases_in_variant = 0;
    m.gc_mod = 0;
    m.end_pos_shift = 0;
    if (st.start_variant_pos) {
        const DevVariant &var = r.v[st.first_variant_id];
        for (uint32_t pos = st.start_variant_pos; pos < var.len && var.pos + pos - st.start_variant_pos < cur_end_position; ++pos)
            if (is_gc(r.base(var, pos))) ++m.gc_mod;
        if (r.gc(cur_start_position)) --m.gc_mod;
        const uint32_t a = cur_end_position - cur_start_position, b = var.len - st.start_variant_pos;
        m.end_pos_shift = 1 - (int32_t)(a < b ? a : b);
        ++m.unhandled_variant_id;
    }
    for (int k = 0; k < 3; ++k) m.surrounding_start[k] = surrounding_start[k];
    variant_mod_start_surrounding(m.surrounding_start, st, allele, r, cur_start_position);
    handle_gc_mod_and_end_pos_shift_for_new_variants(m, allele, r, cur_end_position);
    m.last_end_position = cur_end_position;
}
// :1700-1752; sur starts as the reference's reverse surrounding of last_position
RSQ_HD void variant_mod_end_surrounding(uint32_t (&sur)[3], const AlleleMod &m, const VarStart &st, uint32_t allele, const VarView &r, uint32_t last_position) {
    if (!r.n) return;
    int32_t pos_shift = (int32_t)kSurStart;
    int32_t cur_var = m.unhandled_variant_id;
    if (m.unhandled_bases_in_variant) {
        const DevVariant &var = r.v[cur_var];
        sur_insert_shift_left(sur, (uint32_t)pos_shift, VarBasesRC{r.bases + var.off + (var.len - m.unhandled_bases_in_variant), m.unhandled_bases_in_variant},
                              m.unhandled_bases_in_variant);
        pos_shift -= (int32_t)m.unhandled_bases_in_variant;
        ++cur_var;
    } else if (st.start_variant_pos && r.v[st.first_variant_id].pos == last_position &&
               r.v[st.first_variant_id].len > st.start_variant_pos - (uint32_t)m.end_pos_shift + 1u) {
        if (pos_shift) {
            const DevVariant &var = r.v[st.first_variant_id];
            const uint32_t from = st.start_variant_pos - (uint32_t)m.end_pos_shift + 1u;
            sur_insert_shift_left(sur, (uint32_t)(pos_shift - 1), VarBasesRC{r.bases + var.off + from, var.len - from}, var.len - from);
            pos_shift -= (int32_t)(var.len - from);
        }
    }
    handle_surrounding_variants_after_center(sur, last_position, pos_shift, cur_var, allele, r, true);
    pos_shift = (int32_t)kSurStart;
    cur_var = m.unhandled_variant_id;
    if (m.unhandled_bases_in_variant) {
        if (pos_shift + 1 < (int32_t)kSurLength) {
            const DevVariant &var = r.v[cur_var];
            sur_change(sur, pos_shift + 1, 3u - r.base(var, 0));
            const uint32_t n_part = var.len - m.unhandled_bases_in_variant - 1u;              // infix(var_seq_, 1, length - unhandled)
            sur_insert_shift_right(sur, (uint32_t)(pos_shift + 1), VarBasesRC{r.bases + var.off + 1u, n_part}, n_part);
            pos_shift += (int32_t)var.len - (int32_t)m.unhandled_bases_in_variant - 1;
        }
    } else if (st.start_variant_pos && r.v[st.first_variant_id].pos == last_position) {
        cur_var = st.first_variant_id;
        const DevVariant &var = r.v[cur_var];
        sur_change(sur, pos_shift, 3u - r.base(var, 0));
        uint32_t to = st.start_variant_pos - (uint32_t)m.end_pos_shift + 1u;                   // infix(var_seq_, 1, start_variant_pos_ - end_pos_shift_ + 1)
        if (to > var.len) to = var.len;
        sur_insert_shift_right(sur, (uint32_t)pos_shift, VarBasesRC{r.bases + var.off + 1u, to - 1u}, to - 1u);
        pos_shift += (int32_t)st.start_variant_pos - m.end_pos_shift;
    }
    handle_surrounding_variants_before_center(sur, last_position, pos_shift, cur_var, allele, r, true);
}
// :1754-1812
RSQ_HD void update_bias_mod_for_current_fragment_length(AlleleMod &m, const VarStart &st, uint32_t allele, const VarView &r, uint32_t cur_start_position,
                                                        uint32_t cur_end_position, uint32_t last_end_position) {
    if (!(cur_end_position > m.last_end_position)) return;
    bool need_new_variants = false;
    if (st.start_variant_pos && last_end_position + 1u - cur_start_position <= r.v[st.first_variant_id].len - st.start_variant_pos) {
        const DevVariant &var = r.v[st.first_variant_id];
        uint32_t stop_pos = st.start_variant_pos + cur_end_position - cur_start_position;
        if (stop_pos > var.len) {
            stop_pos = var.len;
            need_new_variants = true;
        }
        const uint32_t start_pos = st.start_variant_pos + last_end_position + 1u - cur_start_position - 1u;
        m.end_pos_shift -= (int32_t)(stop_pos - start_pos);
        for (uint32_t pos = start_pos; pos < stop_pos; ++pos)
            if (is_gc(r.base(var, pos))) ++m.gc_mod;
    } else if (m.unhandled_bases_in_variant) {
        const DevVariant &var = r.v[m.unhandled_variant_id];
        const uint32_t start_pos = var.len - m.unhandled_bases_in_variant;
        uint32_t stop_pos = start_pos + cur_end_position - last_end_position;
        if (stop_pos > var.len) {
            stop_pos = var.len;
            need_new_variants = true;
        }
        m.end_pos_shift -= (int32_t)(stop_pos - start_pos);
        m.unhandled_bases_in_variant -= stop_pos - start_pos;
        for (uint32_t pos = start_pos; pos < stop_pos; ++pos)
            if (is_gc(r.base(var, pos))) ++m.gc_mod;
        if (0u == m.unhandled_bases_in_variant) ++m.unhandled_variant_id;
    } else need_new_variants = true;
    if (need_new_variants) handle_gc_mod_and_end_pos_shift_for_new_variants(m, allele, r, cur_end_position);
}
// :1814-1827
RSQ_HD void prepare_end_surroundings_for_current_fragment_length(AlleleMod &m, const VarStart &st, uint32_t allele, const VarView &r, uint32_t cur_end_position) {
    const uint32_t corrected_pos = cur_end_position + (uint32_t)m.end_pos_shift;
    if (corrected_pos < r.L) {
        surrounding_reverse(r.words, r.word_off, r.L, corrected_pos, m.surrounding_end);
        variant_mod_end_surrounding(m.surrounding_end, m, st, allele, r, corrected_pos);
    }
}
// :1829-1851
RSQ_HD void prepare_bias_mod_for_current_fragment_length(AlleleMod &m, const VarStart &st, uint32_t allele, const VarView &r, uint32_t cur_start_position,
                                                         uint32_t fragment_length) {
    const uint32_t cur_end_position = cur_start_position + fragment_length - 1u;
    if (!(m.last_end_position <= cur_end_position)) return;
    update_bias_mod_for_current_fragment_length(m, st, allele, r, cur_start_position, cur_end_position, m.last_end_position);
    if (st.start_variant_pos && fragment_length <= r.v[st.first_variant_id].len - st.start_variant_pos) {
        update_bias_mod_for_current_fragment_length(m, st, allele, r, cur_start_position, cur_end_position + 1u, cur_end_position);
        prepare_end_surroundings_for_current_fragment_length(m, st, allele, r, cur_end_position);
    } else {
        prepare_end_surroundings_for_current_fragment_length(m, st, allele, r, cur_end_position);
        update_bias_mod_for_current_fragment_length(m, st, allele, r, cur_start_position, cur_end_position + 1u, cur_end_position);
    }
    m.last_end_position = cur_end_position + 1u;
}
// Simulator.h:73-89 EndVariant
RSQ_HD VarStart end_variant(const AlleleMod &m, const VarStart &st, const VarView &r, uint32_t cur_end_position) {
    if (m.unhandled_bases_in_variant) return VarStart{m.unhandled_variant_id, r.v[m.unhandled_variant_id].len - m.unhandled_bases_in_variant};
    if (st.start_variant_pos && r.v[st.first_variant_id].pos == cur_end_position - (uint32_t)m.end_pos_shift - 1u)
        return VarStart{st.first_variant_id, st.start_variant_pos - (uint32_t)m.end_pos_shift + 1u};
    int32_t first_rev = m.unhandled_variant_id;
    if ((uint32_t)first_rev == r.n) --first_rev;
    while (0 <= first_rev && r.v[first_rev].pos >= cur_end_position) --first_rev;
    return VarStart{first_rev, 0u};
}

// everything SimulateFromGivenBlock needs of one (start, pass, fragment length, allele): :2311-2330
struct VarCellSite {
    uint32_t cur_end_position;
    uint32_t gc_percent;
    VarStart end_var;
};
RSQ_HD void evaluate_allele(const VarView &r, const VarStart &st, uint32_t allele, uint32_t cur_start_position, uint32_t first_fragment_length, uint32_t fragment_length,
                            AlleleMod &m, VarCellSite &site) {
    uint32_t sur_start[3];
    surrounding_forward(r.words, r.word_off, r.L, cur_start_position, sur_start);
    prepare_bias_mod_for_current_start_pos(m, st, allele, r, cur_start_position, first_fragment_length, sur_start);
    prepare_bias_mod_for_current_fragment_length(m, st, allele, r, cur_start_position, fragment_length);
    site.cur_end_position = cur_start_position + fragment_length + (uint32_t)m.end_pos_shift;
    site.gc_percent = 0;
    site.end_var = VarStart{0, 0};
    if (site.cur_end_position < r.L) {
        // GetGCPercent :1853-1868: the reference's count over [start, end) plus the allele's modification
        const uint32_t gc_ref = cur_start_position < site.cur_end_position ? ref_gc_count_prefix(r.words, r.gc_prefix, r.word_off, cur_start_position, site.cur_end_position)
                                                                           : 0u - ref_gc_count_prefix(r.words, r.gc_prefix, r.word_off, site.cur_end_position, cur_start_position);
        site.gc_percent = percent_u32((uint32_t)((int32_t)gc_ref + m.gc_mod), fragment_length);
        site.end_var = end_variant(m, st, r, site.cur_end_position);
    }
}

// Reference::ReferenceSequence with variants (Reference.cpp:498-567): the template of one mate, 2 bits per base in read orientation
struct TemplateWriter {                 // 32 bases are collected in a register and stored as one word
    uint64_t *words;
    uint32_t n, cap;
    uint64_t acc;
    RSQ_HD void put(uint32_t base) {
        if (n < cap) {
            acc |= (uint64_t)base << ((n & 31u) * 2u);
            if ((n & 31u) == 31u) {
                words[n >> 5] = acc;
                acc = 0;
            }
        }
        ++n;
    }
    RSQ_HD void finish(uint32_t template_words) {                               // the last partial word, zeros behind it
        const uint32_t have = n < cap ? n : cap;
        uint32_t w = have >> 5;
        if (have & 31u) words[w++] = acc;
        for (; w < template_words; ++w) words[w] = 0;
    }
};
struct RefReader {                      // consecutive reference bases: one load per 32 of them
    const VarView &r;
    uint32_t index = 0xFFFFFFFFu;
    uint64_t word = 0;
    RSQ_HD uint32_t at(uint32_t pos) {
        if ((pos >> 5) != index) {
            index = pos >> 5;
            word = r.words[r.word_off + index];
        }
        return (uint32_t)(word >> ((pos & 31u) * 2u)) & 3u;
    }
};
RSQ_HD uint32_t reference_sequence_with_variants(const VarView &view, uint32_t start_pos, uint32_t frag_length, bool reversed, VarStart first_variant, uint32_t allele,
                                                 uint64_t *tmpl, uint32_t template_words) {
    RefReader r{view};
    TemplateWriter out{tmpl, 0, frag_length, 0};                                // resize(out, frag_length) at the end
    uint32_t cur_start = start_pos;
    int32_t cur_var = first_variant.first_variant_id;
    if (reversed) {
        if (first_variant.start_variant_pos) {
            const DevVariant &var = view.v[cur_var];
            for (uint32_t k = first_variant.start_variant_pos; k--;) out.put(3u - view.base(var, k));       // prefix(var_seq_, pos), reverse complemented
            --cur_var;
            --cur_start;
        }
        for (; cur_var >= 0 && out.n < frag_length; --cur_var) {
            const DevVariant &var = view.v[cur_var];
            if (!view.in_allele(var, allele)) continue;
            if (cur_start - var.pos > frag_length - out.n) {
                const uint32_t from = cur_start + out.n - frag_length;
                for (uint32_t p = cur_start; p-- > from;) out.put(3u - r.at(p));
            } else {
                for (uint32_t p = cur_start; p-- > var.pos + 1u;) out.put(3u - r.at(p));
                for (uint32_t k = var.len; k--;) out.put(3u - view.base(var, k));
                cur_start = var.pos;
            }
        }
        if (cur_var == -1 && out.n < frag_length) {
            const uint32_t from = cur_start + out.n - frag_length;
            for (uint32_t p = cur_start; p-- > from;) out.put(3u - r.at(p));
        }
    } else {
        if (first_variant.start_variant_pos) {
            const DevVariant &var = view.v[cur_var];
            for (uint32_t k = first_variant.start_variant_pos; k < var.len; ++k) out.put(view.base(var, k));
            ++cur_var;
            ++cur_start;
        }
        for (; (uint32_t)cur_var < view.n && out.n < frag_length; ++cur_var) {
            const DevVariant &var = view.v[cur_var];
            if (!view.in_allele(var, allele)) continue;
            if (var.pos - cur_start >= frag_length - out.n) {
                const uint32_t to = cur_start + frag_length - out.n;
                for (uint32_t p = cur_start; p < to; ++p) out.put(r.at(p));
            } else {
                for (uint32_t p = cur_start; p < var.pos; ++p) out.put(r.at(p));
                for (uint32_t k = 0; k < var.len; ++k) out.put(view.base(var, k));
                cur_start = var.pos + 1u;
            }
        }
        if ((uint32_t)cur_var == view.n && out.n < frag_length) {
            const uint32_t to = cur_start + frag_length - out.n;
            for (uint32_t p = cur_start; p < to; ++p) out.put(r.at(p));
        }
    }
    out.finish(template_words);
    return out.n < frag_length ? out.n : frag_length;
}

}  // namespace rsq
