// rsq_variants.h -- variants of any kind (substitutions, insertions, deletions) on the device: alleles as coordinate maps.
//
// A variant replaces ONE reference base by var_seq_ (Reference.h:24-62: empty = the base is deleted, one base = substitution, more =
// the base followed by inserted bases).  An allele of a sequence is therefore the reference with a position-sorted list of such
// replacements, and everything SimulateFromGivenBlock asks of an allele for one cell (Simulator.cpp:2311-2337: end position, G/C percent,
// start and end surrounding, EndVariant) is a property of the stretch [hs, hs + fragment length) of the ALLELE's own sequence -- the
// statement the reference's test makes of its bookkeeping (SimulatorTest.cpp:163-192) and the oracle checks for every cell it evaluates
// (orc_var_haplotype_check).  The reference reaches these values incrementally while it walks start positions and fragment lengths
// (VariantBiasVarModifiers, Simulator.cpp:1399-1851, kept as the checker in oracle/oracle_variants.hpp).  Here the host builds once per
// (sequence, allele) the map between reference and allele coordinates (AlleleVar: per replacement the running length difference and the
// running G/C difference), and a lane that owns a cell gets
//   - the allele coordinate of its start:           one search by reference position,
//   - the end position and EndVariant:              one search by allele coordinate,
//   - the G/C count of the fragment:                the difference of two prefix values (reference prefix sums + the map's running G/C),
//   - both surroundings:                            30 allele bases gathered piecewise (runs of the 2-bit reference, bases of replacements)
// -- O(log variants) per cell instead of a replay over the fragment's length.
// Also here: Reference::ReferenceSequence with variants (Reference.cpp:498-567) writing a 2-bit template.
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

// the start of a pass at a start position: bias_mod.first_variant_id_, start_variant_pos_ (a pass with start_variant_pos > 0 starts
// inside the inserted bases of variant first_variant_id, CheckForInsertedBasesToStartFrom :1870-1896)
struct VarStart {
    int32_t first_variant_id;
    uint32_t start_variant_pos;
};

// GetPossibleAlleles (:1330-1340, Simulator.h:401-412): at a start position whose first variant deletes the base the alleles with
// that deletion have nothing to start from; a pass inside inserted bases exists only for the alleles that have the insertion
RSQ_HD bool allele_starts_here(const VarView &r, const VarStart &st, uint32_t allele, uint32_t start) {
    if ((uint32_t)st.first_variant_id >= r.n) return true;
    const DevVariant &var = r.v[st.first_variant_id];
    if (var.pos != start) return true;
    const bool has = r.in_allele(var, allele);
    if (0u == var.len) return !has;
    return 0u == st.start_variant_pos || has;
}

// ---- one allele of one sequence
struct AlleleView {
    const AlleleVar *e;                 // its replacements in position order, e[n] = sentinel {L, number of variants, total shift, total G/C}
    uint32_t n;
    VarView r;
    const uint32_t *bases_gc;           // bases_gc[i] = G/C among var_bases[0, i)
    RSQ_HD const DevVariant &var(uint32_t i) const { return r.v[e[i].vid]; }
    RSQ_HD int64_t begin_of(uint32_t i) const { return (int64_t)e[i].pos + e[i].shift; }       // allele coordinate of entry i's first base
    RSQ_HD int64_t length() const { return (int64_t)r.L + e[n].shift; }
    RSQ_HD uint32_t entries_before(uint32_t pos) const {                  // entries in front of reference position pos
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (e[mid].pos < pos) lo = mid + 1u;
            else hi = mid;
        }
        return lo;
    }
    RSQ_HD uint32_t entries_upto(int64_t h) const {                       // entries that begin at or before allele coordinate h
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (begin_of(mid) <= h) lo = mid + 1u;
            else hi = mid;
        }
        return lo;
    }
    // allele coordinate of reference position pos (of the first base that replaces it; of the next kept base if pos is deleted)
    RSQ_HD int64_t to_allele(uint32_t pos) const { return (int64_t)pos + e[entries_before(pos)].shift; }
    RSQ_HD uint32_t ref_gc_before(uint32_t pos) const {                   // G/C among reference bases [0, pos), pos <= L
        return r.gc_prefix[r.word_off + (pos >> 5)] - r.gc_prefix[r.word_off] + gc_low(r.words[r.word_off + (pos >> 5)], pos & 31u);
    }
};
RSQ_HD AlleleView allele_view(const DevSim &S, uint32_t seq, uint32_t allele) {
    const uint32_t m = seq * S.num_alleles + allele, lo = S.allele_map_ptr[m], hi = S.allele_map_ptr[m + 1];
    return AlleleView{S.allele_map + lo, hi - lo - 1u, var_view(S, seq), S.var_bases_gc};
}

// what sits at allele coordinate h (0 <= h <= allele length)
struct AllelePoint {
    uint32_t j;                         // entries that begin at or before h
    uint32_t k;                         // inside: h is base k of entry j-1's bases
    uint32_t ref;                       // the reference position behind it: the entry's position, or the plain base's own
    bool inside;
};
RSQ_HD AllelePoint allele_point(const AlleleView &a, int64_t h) {
    const uint32_t j = a.entries_upto(h);
    if (j) {
        const int64_t k = h - a.begin_of(j - 1u);
        if (k < (int64_t)a.var(j - 1u).len) return AllelePoint{j, (uint32_t)k, a.e[j - 1u].pos, true};
    }
    return AllelePoint{j, 0u, (uint32_t)(h - a.e[j].shift), false};
}
// G/C among the allele's bases [0, h)
RSQ_HD uint32_t allele_gc_before(const AlleleView &a, const AllelePoint &p) {
    if (p.inside) {
        const AlleleVar &en = a.e[p.j - 1u];
        const uint32_t off = a.r.v[en.vid].off;
        return (uint32_t)((int32_t)a.ref_gc_before(en.pos) + en.gc) + a.bases_gc[off + p.k] - a.bases_gc[off];
    }
    return (uint32_t)((int32_t)a.ref_gc_before(p.ref) + a.e[p.j].gc);
}

// 30 consecutive allele bases from coordinate h0, first base in the lowest bits.  Coordinates in front of the allele and behind it
// continue in the REFERENCE around the sequence's ends, without variants: the reference's surroundings wrap around
// (SurroundingBase.hpp:64-81) and its variant edits stop at the ends (HandleSurroundingVariantsBeforeCenter / AfterCenter, :1459-1589).
RSQ_HD uint64_t allele_window(const AlleleView &a, int64_t h0) {
    constexpr uint32_t kN = kSurBlocks * kSurRange;
    uint64_t x = 0;
    uint32_t got = 0;
    int64_t h = h0;
    const uint32_t L = a.r.L;
    for (; got < kN && h < 0; ++got, ++h) x |= (uint64_t)a.r.at((uint32_t)((int64_t)L + h)) << (2u * got);
    if (got < kN && h < a.length()) {
        const AllelePoint p = allele_point(a, h);
        uint32_t j = p.j, q = p.ref;                                      // next entry, next reference position
        if (p.inside) {
            const DevVariant &var = a.var(j - 1u);
            for (uint32_t k = p.k; k < var.len && got < kN; ++k, ++got) x |= (uint64_t)a.r.base(var, k) << (2u * got);
            ++q;
        }
        while (got < kN && q < L) {
            const uint32_t stop = a.e[j].pos;                             // the sentinel stops at L
            uint32_t run = stop - q;
            if (run > kN - got) run = kN - got;
            if (run) {
                x |= (ref_bits60(a.r.words, a.r.word_off, q) & ((1ull << (2u * run)) - 1ull)) << (2u * got);
                got += run;
                q += run;
            }
            if (got == kN || q == L) break;
            const DevVariant &var = a.var(j);                             // the entry at q
            for (uint32_t k = 0; k < var.len && got < kN; ++k, ++got) x |= (uint64_t)a.r.base(var, k) << (2u * got);
            ++j;
            ++q;
        }
    }
    for (uint32_t q = 0; got < kN; ++got) {
        x |= (uint64_t)a.r.at(q) << (2u * got);
        if (++q == L) q = 0;
    }
    return x;
}
// the allele's forward surrounding of coordinate h (as surrounding_forward does for the reference) and its reverse surrounding
RSQ_HD void allele_surrounding_forward(const AlleleView &a, int64_t h, uint32_t (&sur)[3]) {
    const uint64_t x = allele_window(a, h - (int64_t)kSurStart);
    for (uint32_t b = 0; b < kSurBlocks; ++b) sur[b] = reverse_ten_bases((uint32_t)(x >> (20u * b)) & 0xFFFFFu);
}
RSQ_HD void allele_surrounding_reverse(const AlleleView &a, int64_t h, uint32_t (&sur)[3]) {
    const uint64_t x = allele_window(a, h + (int64_t)kSurStart + 1 - (int64_t)(kSurBlocks * kSurRange));
    for (uint32_t b = 0; b < kSurBlocks; ++b) sur[b] = ~(uint32_t)(x >> (20u * (kSurBlocks - 1u - b))) & 0xFFFFFu;
}

// one cell (start position, pass, fragment length) of one allele
struct AlleleCell {
    int64_t hs, he;                     // the fragment in allele coordinates
    uint32_t end;                       // cur_end_position = start + length + end_pos_shift_[allele] (:2316)
    uint32_t gc_percent;                // GetGCPercent (:1853-1868)
    VarStart end_var;                   // EndVariant (Simulator.h:73-89)
    bool inside;                        // the end position lies inside the sequence (:2318)
};
RSQ_HD AlleleCell allele_cell(const AlleleView &a, const VarStart &st, uint32_t start, uint32_t length) {
    AlleleCell c;
    c.hs = a.to_allele(start) + st.start_variant_pos;
    c.he = c.hs + length;
    c.end = 0;
    c.gc_percent = 0;
    c.end_var = VarStart{0, 0u};
    c.inside = false;
    if (c.he > a.length()) return c;                                     // the allele ends before the fragment does
    const AllelePoint last = allele_point(a, c.he - 1);
    c.end = last.ref + 1u;                                               // the reference position behind the fragment's last base
    if (!(c.end < a.r.L)) return c;
    c.inside = true;
    const AllelePoint first = allele_point(a, c.hs), behind = allele_point(a, c.he);
    c.gc_percent = percent_u32(allele_gc_before(a, behind) - allele_gc_before(a, first), length);
    // EndVariant: a fragment that ends inside inserted bases hands the variant and the number of its bases it uses to the reverse
    // read; otherwise the last variant in front of the end position.  The reference has a third case for a fragment that ends inside
    // the inserted bases it started in (Simulator.h:78-80) whose test can only hold for a fragment of one base, so such a fragment
    // gets the second answer there -- and here.
    const bool started_in_it = st.start_variant_pos && last.inside && a.e[last.j - 1u].vid == (uint32_t)st.first_variant_id;
    if (last.inside && last.k + 1u < a.var(last.j - 1u).len && !started_in_it) c.end_var = VarStart{(int32_t)a.e[last.j - 1u].vid, last.k + 1u};
    else c.end_var = VarStart{(int32_t)a.r.lower_bound(c.end) - 1, 0u};
    return c;
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
