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
// Also here: the templates of a fragment's mates (Reference::ReferenceSequence with variants, Reference.cpp:498-567) as stretches of the allele.
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
    // the same index, found by a short walk from `hint` (an index near the answer) where that suffices: what one cell of the sieve asks lies within a fragment's
    // length of what it asked before, a handful of variants at most, while a bisection of a chromosome's variants is twenty dependent loads
    RSQ_HD uint32_t lower_bound(uint32_t pos, uint32_t hint) const {
        uint32_t k = hint < n ? hint : n;
        for (uint32_t step = 0; step < kHintWalk; ++step) {
            if (k < n && v[k].pos < pos) ++k;
            else if (k > 0u && !(v[k - 1u].pos < pos)) --k;
            else return k;
        }
        return lower_bound(pos);
    }
    static constexpr uint32_t kHintWalk = 8;
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
    // the same count by a short walk from `hint` (VarView::lower_bound with a hint); kNoHint: the bisection
    static constexpr uint32_t kNoHint = 0xFFFFFFFFu;
    RSQ_HD uint32_t entries_upto(int64_t h, uint32_t hint) const {
        if (hint == kNoHint) return entries_upto(h);
        uint32_t j = hint < n ? hint : n;
        for (uint32_t step = 0; step < VarView::kHintWalk; ++step) {
            if (j < n && begin_of(j) <= h) ++j;
            else if (j > 0u && begin_of(j - 1u) > h) --j;
            else return j;
        }
        return entries_upto(h);
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
RSQ_HD AllelePoint allele_point(const AlleleView &a, int64_t h, uint32_t hint = AlleleView::kNoHint) {
    const uint32_t j = a.entries_upto(h, hint);
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

// up to 32 reference bases from position p of the sequence, first base in the lowest bits (the sequence's spare word makes w[1] readable)
RSQ_HD uint64_t ref_bits64(const uint64_t *__restrict__ words, uint64_t word_off, uint32_t p) {
    const uint64_t *w = words + word_off + (p >> 5);
    const uint32_t off = (p & 31u) * 2u;
    return off ? (w[0] >> off) | (w[1] << (64u - off)) : w[0];
}
// n <= 32 consecutive allele bases from coordinate h0, first base in the lowest bits, zeros above them.  Coordinates in front of the allele
// and behind it continue in the REFERENCE around the sequence's ends, without variants: the reference's surroundings wrap around
// (SurroundingBase.hpp:64-81) and its variant edits stop at the ends (HandleSurroundingVariantsBeforeCenter / AfterCenter, :1459-1589).
RSQ_HD uint64_t allele_bits(const AlleleView &a, int64_t h0, uint32_t n, uint32_t hint = AlleleView::kNoHint) {
    uint64_t x = 0;
    uint32_t got = 0;
    int64_t h = h0;
    const uint32_t L = a.r.L;
    for (; got < n && h < 0; ++got, ++h) x |= (uint64_t)a.r.at((uint32_t)((int64_t)L + h)) << (2u * got);
    if (got < n && h < a.length()) {
        const AllelePoint p = allele_point(a, h, hint);
        uint32_t j = p.j, q = p.ref;                                      // next entry, next reference position
        if (p.inside) {
            const DevVariant &var = a.var(j - 1u);
            for (uint32_t k = p.k; k < var.len && got < n; ++k, ++got) x |= (uint64_t)a.r.base(var, k) << (2u * got);
            ++q;
        }
        while (got < n && q < L) {
            const uint32_t stop = a.e[j].pos;                             // the sentinel stops at L
            uint32_t run = stop - q;
            if (run > n - got) run = n - got;
            if (run) {
                const uint64_t bits = ref_bits64(a.r.words, a.r.word_off, q);
                x |= (run < 32u ? bits & ((1ull << (2u * run)) - 1ull) : bits) << (2u * got);
                got += run;
                q += run;
            }
            if (got == n || q == L) break;
            const DevVariant &var = a.var(j);                             // the entry at q
            for (uint32_t k = 0; k < var.len && got < n; ++k, ++got) x |= (uint64_t)a.r.base(var, k) << (2u * got);
            ++j;
            ++q;
        }
    }
    for (uint32_t q = 0; got < n; ++got) {
        x |= (uint64_t)a.r.at(q) << (2u * got);
        if (++q == L) q = 0;
    }
    return x;
}
RSQ_HD uint64_t allele_window(const AlleleView &a, int64_t h0, uint32_t hint = AlleleView::kNoHint) { return allele_bits(a, h0, kSurBlocks * kSurRange, hint); }
// the allele's forward surrounding of coordinate h (as surrounding_forward does for the reference) and its reverse surrounding
RSQ_HD void allele_surrounding_forward(const AlleleView &a, int64_t h, uint32_t (&sur)[3], uint32_t hint = AlleleView::kNoHint) {
    const uint64_t x = allele_window(a, h - (int64_t)kSurStart, hint);
    for (uint32_t b = 0; b < kSurBlocks; ++b) sur[b] = reverse_ten_bases((uint32_t)(x >> (20u * b)) & 0xFFFFFu);
}
RSQ_HD void allele_surrounding_reverse(const AlleleView &a, int64_t h, uint32_t (&sur)[3], uint32_t hint = AlleleView::kNoHint) {
    const uint64_t x = allele_window(a, h + (int64_t)kSurStart + 1 - (int64_t)(kSurBlocks * kSurRange), hint);
    for (uint32_t b = 0; b < kSurBlocks; ++b) sur[b] = ~(uint32_t)(x >> (20u * (kSurBlocks - 1u - b))) & 0xFFFFFu;
}

// one cell (start position, pass, fragment length) of one allele
struct AlleleCell {
    int64_t hs, he;                     // the fragment in allele coordinates
    uint32_t end;                       // cur_end_position = start + length + end_pos_shift_[allele] (:2316)
    uint32_t gc_percent;                // GetGCPercent (:1853-1868)
    VarStart end_var;                   // EndVariant (Simulator.h:73-89)
    bool inside;                        // the end position lies inside the sequence (:2318)
    uint32_t first_entries, last_entries;      // entries of the allele's map that begin at or before hs / he - 1: hints for what is asked next (the surroundings)
};
RSQ_HD AlleleCell allele_cell(const AlleleView &a, const VarStart &st, uint32_t start, uint32_t length) {
    AlleleCell c;
    const uint32_t before = a.entries_before(start);                     // the cell's one bisection; everything else it asks lies a fragment's length from here
    c.hs = (int64_t)start + a.e[before].shift + st.start_variant_pos;      // to_allele(start) + ...
    c.he = c.hs + length;
    c.end = 0;
    c.gc_percent = 0;
    c.end_var = VarStart{0, 0u};
    c.inside = false;
    c.first_entries = c.last_entries = before;
    if (c.he > a.length()) return c;                                     // the allele ends before the fragment does
    const AllelePoint last = allele_point(a, c.he - 1, before);
    c.last_entries = last.j;
    c.end = last.ref + 1u;                                               // the reference position behind the fragment's last base
    if (!(c.end < a.r.L)) return c;
    c.inside = true;
    const AllelePoint first = allele_point(a, c.hs, before), behind = allele_point(a, c.he, last.j);
    c.first_entries = first.j;
    c.gc_percent = percent_u32(allele_gc_before(a, behind) - allele_gc_before(a, first), length);
    // EndVariant: a fragment that ends inside inserted bases hands the variant and the number of its bases it uses to the reverse
    // read; otherwise the last variant in front of the end position.  The reference has a third case for a fragment that ends inside
    // the inserted bases it started in (Simulator.h:78-80) whose test can only hold for a fragment of one base, so such a fragment
    // gets the second answer there -- and here.
    const bool started_in_it = st.start_variant_pos && last.inside && a.e[last.j - 1u].vid == (uint32_t)st.first_variant_id;
    if (last.inside && last.k + 1u < a.var(last.j - 1u).len && !started_in_it) c.end_var = VarStart{(int32_t)a.e[last.j - 1u].vid, last.k + 1u};
    else c.end_var = VarStart{(int32_t)a.r.lower_bound(c.end, a.e[last.j].vid) - 1, 0u};      // near the allele's next entry (the sentinel: the number of variants)
    return c;
}

// Reference::ReferenceSequence with variants (Reference.cpp:498-567): the template of one mate, 2 bits per base in read orientation.
// A template is a stretch of the allele's own sequence: the forward mate's begins at the allele coordinate of (start position, start
// variant), the reverse mate's is the reverse complement of the stretch that ends where (end position, EndVariant) points -- inside inserted
// bases after `end_var_pos` of them, else in front of everything that replaces the end position.  32 bases per word, gathered piecewise.
RSQ_HD uint64_t reverse_complement_bits(uint64_t x, uint32_t n) {        // of the lowest n <= 32 bases
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t r = __brevll(~x);
#else
    uint64_t r = 0, c = ~x;
    for (uint32_t i = 0; i < 64u; ++i) r |= ((c >> i) & 1ull) << (63u - i);
#endif
    r = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);      // the two bits of every base back in order
    return n < 32u ? r >> (64u - 2u * n) : r;
}
RSQ_HD void allele_template(const AlleleView &a, uint32_t pos, VarStart from, uint32_t n, bool reversed, uint64_t *tmpl, uint32_t template_words) {
    int64_t h;
    uint32_t near;                                                       // entries of the map around the template's first base: the one bisection, a hint for every word
    if (from.start_variant_pos) {
        near = a.entries_before(a.r.v[from.first_variant_id].pos);
        h = a.begin_of(near) + from.start_variant_pos;
    } else {
        near = a.entries_before(pos);
        h = (int64_t)pos + a.e[near].shift;                             // to_allele(pos)
    }
    uint32_t w = 0;
    for (uint32_t done = 0; done < n; done += 32u, ++w) {
        const uint32_t c = n - done < 32u ? n - done : 32u;
        tmpl[w] = reversed ? reverse_complement_bits(allele_bits(a, h - done - c, c, near), c) : allele_bits(a, h + done, c, near);
    }
    for (; w < template_words; ++w) tmpl[w] = 0;
}

}  // namespace rsq
