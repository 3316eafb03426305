// oracle_variants.hpp -- TEST INFRASTRUCTURE (part of the CPU oracle, see oracle.h): the per-allele bookkeeping of the coverage sieve when
// variants are loaded: Simulator::VariantBiasVarModifiers and the functions that keep it up to date while
// SimulateFromGivenBlock walks start positions and fragment lengths (Simulator.h:29-89,401-412; Simulator.cpp:1330-1340,1399-1896), plus
// Reference::ReferenceSequence with variants (Reference.cpp:498-567).  Restated with the reference's variable widths and statement order
// (incremental over fragment lengths, as there); pinned to SimulatorTest::TestVariationInSimulateFromGivenBlock (orc_var_* driver in
// oracle_variants.cpp).  C++ because the bookkeeping is vectors of vectors; everything it computes with comes from oracle.h.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <stdexcept>
#include <utility>
#include <vector>

#include "oracle.h"

namespace orcv {

constexpr uint32_t kSurStart = 10, kSurRange = 10, kSurBlocks = 3, kSurLength = 30;     // Surrounding.h:17
struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
inline bool is_gc(uint8_t b) { return b == 1 || b == 2; }
inline uint32_t percent_u32(uint32_t nom, uint32_t den) { return orc_percent_u32(nom, den); }

struct Variant {                                                             // Reference.h:24-62
    uint32_t position = 0;
    std::vector<uint8_t> var_seq;
    uint64_t allele[2] = {0, 0};
    bool in_allele(uint32_t a) const { return (allele[a / 64] >> (a % 64)) & 1u; }
    uint32_t first_allele() const {                                          // Reference.h:42-58
        orc_variant v;
        v.allele[0] = allele[0];
        v.allele[1] = allele[1];
        return orc_variant_first_allele(&v);
    }
};

struct Sur3 {
    int32_t b[3] = {0, 0, 0};
};

// one reference sequence with its variants (what the functions below read of reseq::Reference)
struct VarRef {
    const std::vector<uint8_t> *codes = nullptr;       // base codes 0..3
    const std::vector<Variant> *variants = nullptr;
    uint32_t num_alleles = 1;
    uint32_t length() const { return (uint32_t)codes->size(); }
    uint8_t at(uint64_t pos) const {
        if (pos >= codes->size()) throw Error("position outside the reference sequence (the reference's utilities::at would throw here)");
        return (*codes)[pos];
    }
    bool gc(uint32_t pos) const { return is_gc(at(pos)); }                                          // Reference.h:201-203
    uint32_t gc_content_absolut(uint32_t start, uint32_t end) const {                                // Reference.h:230-239
        uint32_t n = 0;
        for (uint32_t i = start; i < end; ++i) n += gc(i) ? 1u : 0u;
        return n;
    }
    // ForwardSurrounding / ReverseSurrounding (SurroundingBase.hpp:64-81,196-202) on the base codes, with wrap-around
    Sur3 forward_surrounding(uint32_t pos) const {
        Sur3 s;
        orc_surrounding_forward(codes->data(), length(), pos, s.b);
        return s;
    }
    Sur3 reverse_surrounding(uint32_t pos) const {
        Sur3 s;
        orc_surrounding_reverse(codes->data(), length(), pos, s.b);
        return s;
    }
};

struct VariantBiasMod {                                                      // Simulator.h:29-89
    std::vector<uint32_t> last_end_position;
    uint32_t last_gc = 0, last_gc_end = 0;
    int32_t first_variant_id;
    uint32_t start_variant_pos = 0;
    std::vector<int32_t> unhandled_variant_id;
    std::vector<uint32_t> unhandled_bases_in_variant;
    std::vector<int32_t> gc_mod, end_pos_shift;
    std::vector<Sur3> surrounding_start, surrounding_end;
    VariantBiasMod(int32_t first_variant, uint32_t num_alleles)
        : last_end_position(num_alleles, 0), first_variant_id(first_variant), unhandled_variant_id(num_alleles, first_variant), unhandled_bases_in_variant(num_alleles, 0),
          gc_mod(num_alleles, 0), end_pos_shift(num_alleles, 0), surrounding_start(num_alleles), surrounding_end(num_alleles) {}
    std::pair<int32_t, uint32_t> start_variant() const { return {first_variant_id, start_variant_pos}; }
    std::pair<int32_t, uint32_t> end_variant(const std::vector<Variant> &variants, uint32_t cur_end_position, uint32_t allele) const {      // :73-89
        if (unhandled_bases_in_variant[allele])
            return {unhandled_variant_id[allele], (uint32_t)variants.at((size_t)unhandled_variant_id[allele]).var_seq.size() - unhandled_bases_in_variant[allele]};
        if (start_variant_pos && variants.at((size_t)first_variant_id).position == cur_end_position - (uint32_t)end_pos_shift[allele] - 1u)
            return {first_variant_id, start_variant_pos - (uint32_t)end_pos_shift[allele] + 1u};
        int32_t first_rev = unhandled_variant_id[allele];
        if ((size_t)first_rev == variants.size()) --first_rev;
        while (0 <= first_rev && variants.at((size_t)first_rev).position >= cur_end_position) --first_rev;
        return {first_rev, 0u};
    }
};

namespace variants_detail {
inline std::vector<uint8_t> complemented_reverse(const uint8_t *b, size_t n) {                       // ReverseComplementorDna
    std::vector<uint8_t> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = (uint8_t)(3u - b[n - 1 - i]);
    return out;
}
inline void change(Sur3 &s, int32_t pos, uint32_t base) { orc_sur_change_base(s.b, (uint32_t)(uint16_t)pos, (uint8_t)base); }
inline void del_right(Sur3 &s, int32_t pos, uint32_t base) { orc_sur_delete_shift_right(s.b, (uint32_t)(uint16_t)pos, (uint8_t)base); }
inline void del_left(Sur3 &s, int32_t pos, uint32_t base) { orc_sur_delete_shift_left(s.b, (uint32_t)(uint16_t)pos, (uint8_t)base); }
inline void ins_right(Sur3 &s, int32_t pos, const std::vector<uint8_t> &bases) { orc_sur_insert_shift_right(s.b, (uint32_t)(uint16_t)pos, bases.data(), (uint32_t)bases.size()); }
inline void ins_left(Sur3 &s, int32_t pos, const std::vector<uint8_t> &bases) { orc_sur_insert_shift_left(s.b, (uint32_t)(uint16_t)pos, bases.data(), (uint32_t)bases.size()); }
inline std::vector<uint8_t> sub(const std::vector<uint8_t> &v, size_t from, size_t to) { return std::vector<uint8_t>(v.begin() + (ptrdiff_t)from, v.begin() + (ptrdiff_t)std::min(to, v.size())); }
}  // namespace variants_detail

// Simulator.h:401-412
inline bool allele_skipped(const VariantBiasMod &bm, uint32_t allele, const std::vector<Variant> &variants, uint32_t cur_start_position) {
    if ((size_t)bm.first_variant_id < variants.size() && variants[(size_t)bm.first_variant_id].position == cur_start_position) {
        const Variant &v = variants[(size_t)bm.first_variant_id];
        if (v.var_seq.empty()) return v.in_allele(allele);
        if (bm.start_variant_pos) return !v.in_allele(allele);
    }
    return false;
}
// Simulator.cpp:1330-1340
inline std::vector<uint32_t> possible_alleles(const VarRef &ref, const VariantBiasMod &bm, uint32_t cur_start_position) {
    std::vector<uint32_t> out;
    for (uint32_t allele = 0; allele < ref.num_alleles; ++allele)
        if (!allele_skipped(bm, allele, *ref.variants, cur_start_position)) out.push_back(allele);
    return out;
}
// :1399-1402
inline bool variant_inside_current_fragment(const VarRef &ref, int32_t cur_var_id, uint32_t cur_end_position, int32_t end_pos_shift) {
    return ref.variants->size() > (size_t)cur_var_id && (*ref.variants)[(size_t)cur_var_id].position < cur_end_position + (uint32_t)end_pos_shift;
}

// :1404-1455
inline void handle_gc_mod_and_end_pos_shift_for_new_variants(VariantBiasMod &bm, uint32_t allele, const VarRef &ref, uint32_t cur_end_position) {
    while (variant_inside_current_fragment(ref, bm.unhandled_variant_id[allele], cur_end_position, bm.end_pos_shift[allele]) && 0 == bm.unhandled_bases_in_variant[allele]) {
        const Variant &var = (*ref.variants)[(size_t)bm.unhandled_variant_id[allele]];
        if (!var.in_allele(allele)) {
            ++bm.unhandled_variant_id[allele];
            continue;
        }
        const uint32_t len = (uint32_t)var.var_seq.size();
        for (uint32_t pos = 0; pos < len && var.position + pos < cur_end_position + (uint32_t)bm.end_pos_shift[allele]; ++pos)      // GC from the variant
            if (is_gc(var.var_seq[pos])) ++bm.gc_mod[allele];
        if (ref.gc(var.position)) --bm.gc_mod[allele];                                                                              // GC of the replaced base
        if (0 == len) {                                                      // deletion
            ++bm.end_pos_shift[allele];
            ++bm.unhandled_variant_id[allele];
        } else if (1 == len) {
            ++bm.unhandled_variant_id[allele];
        } else if (var.position + len <= cur_end_position + (uint32_t)bm.end_pos_shift[allele]) {      // insertion completely inside
            bm.end_pos_shift[allele] -= (int32_t)(len - 1u);
            ++bm.unhandled_variant_id[allele];
        } else {                                                             // insertion reaching out of the fragment
            bm.unhandled_bases_in_variant[allele] = var.position + len - (cur_end_position + (uint32_t)bm.end_pos_shift[allele]);
            bm.end_pos_shift[allele] -= (int32_t)(len - bm.unhandled_bases_in_variant[allele] - 1u);
        }
    }
}

// :1459-1527
inline void handle_surrounding_variants_before_center(Sur3 &sur, uint32_t center_position, int32_t initial_pos_shift, int32_t center_var, uint32_t allele, const VarRef &ref,
                                                      bool reverse) {
    using namespace variants_detail;
    int32_t cur_var = center_var, pos_shift = initial_pos_shift;
    const uint32_t L = ref.length();
    while (0 <= --cur_var) {
        const Variant &var = (*ref.variants)[(size_t)cur_var];
        const int32_t sur_pos = reverse ? (int32_t)((uint32_t)(int32_t)center_position - var.position + (uint32_t)pos_shift)
                                        : (int32_t)((uint32_t)(int32_t)var.position - center_position + (uint32_t)pos_shift);
        if (0 > sur_pos || (int32_t)kSurLength <= sur_pos) break;
        if (!var.in_allele(allele)) continue;
        if (var.var_seq.empty()) {                                           // deletion
            if (reverse) {
                const int32_t new_base_pos = (int32_t)center_position + pos_shift - (int32_t)kSurLength;
                del_right(sur, sur_pos, 3u - ref.at(0 > new_base_pos ? (uint64_t)((int64_t)L + new_base_pos) : (uint64_t)new_base_pos));
                --pos_shift;
            } else {
                if ((uint32_t)++pos_shift > center_position) del_left(sur, sur_pos, ref.at((uint64_t)(uint32_t)(L + center_position - (uint32_t)pos_shift)));
                else del_left(sur, sur_pos, ref.at((uint64_t)(center_position - (uint32_t)pos_shift)));
            }
        } else {                                                             // base modification, then the inserted bases
            change(sur, sur_pos, reverse ? 3u - var.var_seq[0] : var.var_seq[0]);
            if (1 < var.var_seq.size()) {
                if (reverse) {
                    ins_right(sur, sur_pos, complemented_reverse(var.var_seq.data() + 1, var.var_seq.size() - 1));
                    pos_shift += (int32_t)var.var_seq.size() - 1;
                } else {
                    ins_left(sur, sur_pos, sub(var.var_seq, 1, var.var_seq.size()));
                    pos_shift -= (int32_t)var.var_seq.size() - 1;
                }
            }
        }
    }
}

// :1529-1589
inline void handle_surrounding_variants_after_center(Sur3 &sur, uint32_t center_position, int32_t initial_pos_shift, int32_t center_var, uint32_t allele, const VarRef &ref,
                                                     bool reverse) {
    using namespace variants_detail;
    int32_t pos_shift = initial_pos_shift;
    const uint32_t L = ref.length();
    for (int32_t cur_var = center_var; (size_t)cur_var < ref.variants->size(); ++cur_var) {
        const Variant &var = (*ref.variants)[(size_t)cur_var];
        const int32_t sur_pos = reverse ? (int32_t)((uint32_t)(int32_t)center_position - var.position + (uint32_t)pos_shift)
                                        : (int32_t)((uint32_t)(int32_t)var.position - center_position + (uint32_t)pos_shift);
        if (0 > sur_pos || (int32_t)kSurLength <= sur_pos) break;
        if (!var.in_allele(allele)) continue;
        if (var.var_seq.empty()) {
            if (reverse) {
                ++pos_shift;
                del_left(sur, sur_pos, 3u - ref.at((uint64_t)((center_position + (uint32_t)pos_shift) % L)));
            } else {
                del_right(sur, sur_pos, ref.at((uint64_t)((center_position - (uint32_t)pos_shift + kSurLength) % L)));
                --pos_shift;
            }
        } else {
            change(sur, sur_pos, reverse ? 3u - var.var_seq[0] : var.var_seq[0]);
            if (1 < var.var_seq.size()) {
                if (reverse) {
                    if (sur_pos) {
                        ins_left(sur, sur_pos - 1, complemented_reverse(var.var_seq.data() + 1, var.var_seq.size() - 1));
                        pos_shift -= (int32_t)var.var_seq.size() - 1;
                    }
                } else if (sur_pos + 1 < (int32_t)kSurLength) {
                    ins_right(sur, sur_pos + 1, sub(var.var_seq, 1, var.var_seq.size()));
                    pos_shift += (int32_t)var.var_seq.size() - 1;
                }
            }
        }
    }
}

// :1591-1636
inline void variant_mod_start_surrounding(VariantBiasMod &bm, uint32_t allele, const VarRef &ref, uint32_t cur_start_position, const Sur3 &surrounding_start) {
    using namespace variants_detail;
    bm.surrounding_start[allele] = surrounding_start;
    if (ref.variants->empty()) return;
    Sur3 &sur = bm.surrounding_start[allele];
    int32_t pos_shift = (int32_t)kSurStart;
    int32_t cur_var = bm.first_variant_id;
    if (bm.start_variant_pos) {
        const Variant &var = ref.variants->at((size_t)cur_var);
        change(sur, pos_shift, var.var_seq.at(0));                           // the substitution an insertion may include
        ins_left(sur, pos_shift, sub(var.var_seq, 1, bm.start_variant_pos + 1u));      // the part of the insertion before the start
        pos_shift -= (int32_t)bm.start_variant_pos;
    }
    handle_surrounding_variants_before_center(sur, cur_start_position, pos_shift, cur_var, allele, ref, false);
    pos_shift = (int32_t)kSurStart;
    cur_var = bm.first_variant_id;
    if (bm.start_variant_pos) {
        const Variant &var = ref.variants->at((size_t)bm.first_variant_id);
        if (var.var_seq.size() > bm.start_variant_pos + 1u) {
            if (pos_shift + 1 < (int32_t)kSurLength) {
                ins_right(sur, pos_shift + 1, sub(var.var_seq, bm.start_variant_pos + 1u, var.var_seq.size()));
                pos_shift += (int32_t)var.var_seq.size() - (int32_t)bm.start_variant_pos - 1;
            }
        }
        ++cur_var;                                                           // this insertion is done
    }
    handle_surrounding_variants_after_center(sur, cur_start_position, pos_shift, cur_var, allele, ref, false);
}

// :1638-1698
inline void prepare_bias_mod_for_current_start_pos(VariantBiasMod &bm, const VarRef &ref, uint32_t cur_start_position, uint32_t first_fragment_length, const Sur3 &surrounding_start) {
    const uint32_t cur_end_position = cur_start_position + first_fragment_length - 1u;
    const uint32_t A = ref.num_alleles;
    bm.unhandled_variant_id.assign(A, bm.first_variant_id);
    bm.unhandled_bases_in_variant.assign(A, 0);
    bm.gc_mod.assign(A, 0);
    bm.end_pos_shift.assign(A, 0);
    if (bm.start_variant_pos) {                                              // starting inside an insertion
        const Variant &var = ref.variants->at((size_t)bm.first_variant_id);
        for (uint32_t pos = bm.start_variant_pos; pos < var.var_seq.size() && var.position + pos - bm.start_variant_pos < cur_end_position; ++pos)
            if (is_gc(var.var_seq[pos])) ++bm.gc_mod[0];
        if (ref.gc(cur_start_position)) --bm.gc_mod[0];
        bm.end_pos_shift[0] = 1 - (int32_t)std::min(cur_end_position - cur_start_position, (uint32_t)var.var_seq.size() - bm.start_variant_pos);
        ++bm.unhandled_variant_id[0];
        for (uint32_t allele = 1; allele < A; ++allele) {
            bm.gc_mod[allele] = bm.gc_mod[0];
            bm.end_pos_shift[allele] = bm.end_pos_shift[0];
            ++bm.unhandled_variant_id[allele];
        }
    }
    for (uint32_t allele = 0; allele < A; ++allele)
        if (!allele_skipped(bm, allele, *ref.variants, cur_start_position)) {
            variant_mod_start_surrounding(bm, allele, ref, cur_start_position, surrounding_start);       // before the GC handling: it reads unhandled_variant_id untouched
            handle_gc_mod_and_end_pos_shift_for_new_variants(bm, allele, ref, cur_end_position);
        }
    bm.last_end_position.assign(A, cur_end_position);
    bm.last_gc = 0;
    bm.last_gc_end = cur_start_position;
}

// :1700-1752
inline void variant_mod_end_surrounding(VariantBiasMod &bm, uint32_t allele, const VarRef &ref, uint32_t last_position) {
    using namespace variants_detail;
    if (ref.variants->empty()) return;
    Sur3 &sur = bm.surrounding_end[allele];
    const std::vector<Variant> &vars = *ref.variants;
    int32_t pos_shift = (int32_t)kSurStart;
    int32_t cur_var = bm.unhandled_variant_id[allele];
    if (bm.unhandled_bases_in_variant[allele]) {                             // the handled part of a partial insertion (the rest lies to the right)
        const Variant &var = vars.at((size_t)cur_var);
        ins_left(sur, pos_shift, complemented_reverse(var.var_seq.data() + (var.var_seq.size() - bm.unhandled_bases_in_variant[allele]), bm.unhandled_bases_in_variant[allele]));
        pos_shift -= (int32_t)bm.unhandled_bases_in_variant[allele];
        ++cur_var;
    } else if (bm.start_variant_pos && vars.at((size_t)bm.first_variant_id).position == last_position &&
               vars.at((size_t)bm.first_variant_id).var_seq.size() > (size_t)(bm.start_variant_pos - (uint32_t)bm.end_pos_shift[allele] + 1u)) {
        if (pos_shift) {
            const Variant &var = vars.at((size_t)bm.first_variant_id);
            const size_t from = bm.start_variant_pos - (uint32_t)bm.end_pos_shift[allele] + 1u;
            ins_left(sur, pos_shift - 1, complemented_reverse(var.var_seq.data() + from, var.var_seq.size() - from));
            pos_shift -= (int32_t)(var.var_seq.size() - from);
        }
    }
    handle_surrounding_variants_after_center(sur, last_position, pos_shift, cur_var, allele, ref, true);
    pos_shift = (int32_t)kSurStart;
    cur_var = bm.unhandled_variant_id[allele];
    if (bm.unhandled_bases_in_variant[allele]) {
        if (pos_shift + 1 < (int32_t)kSurLength) {
            const Variant &var = vars.at((size_t)cur_var);
            change(sur, pos_shift + 1, 3u - var.var_seq.at(0));
            const std::vector<uint8_t> part = sub(var.var_seq, 1, var.var_seq.size() - bm.unhandled_bases_in_variant[allele]);
            ins_right(sur, pos_shift + 1, complemented_reverse(part.data(), part.size()));
            pos_shift += (int32_t)var.var_seq.size() - (int32_t)bm.unhandled_bases_in_variant[allele] - 1;
        }
    } else if (bm.start_variant_pos && vars.at((size_t)bm.first_variant_id).position == last_position) {
        cur_var = bm.first_variant_id;
        const Variant &var = vars.at((size_t)cur_var);
        change(sur, pos_shift, 3u - var.var_seq.at(0));
        const std::vector<uint8_t> part = sub(var.var_seq, 1, bm.start_variant_pos - (uint32_t)bm.end_pos_shift[allele] + 1u);
        ins_right(sur, pos_shift, complemented_reverse(part.data(), part.size()));
        pos_shift += (int32_t)bm.start_variant_pos - bm.end_pos_shift[allele];
    }
    handle_surrounding_variants_before_center(sur, last_position, pos_shift, cur_var, allele, ref, true);
}

// :1754-1812
inline void update_bias_mod_for_current_fragment_length(VariantBiasMod &bm, const VarRef &ref, uint32_t cur_start_position, uint32_t cur_end_position, uint32_t last_end_position,
                                                        uint32_t allele) {
    if (!(cur_end_position > bm.last_end_position[allele])) return;
    const std::vector<Variant> &vars = *ref.variants;
    bool need_new_variants = false;
    if (bm.start_variant_pos && last_end_position + 1u - cur_start_position <= (uint32_t)vars.at((size_t)bm.first_variant_id).var_seq.size() - bm.start_variant_pos) {
        const Variant &var = vars.at((size_t)bm.first_variant_id);           // still inside the insertion the fragment starts in
        uint32_t stop_pos = bm.start_variant_pos + cur_end_position - cur_start_position;
        if (stop_pos > var.var_seq.size()) {
            stop_pos = (uint32_t)var.var_seq.size();
            need_new_variants = true;
        }
        const uint32_t start_pos = bm.start_variant_pos + last_end_position + 1u - cur_start_position - 1u;
        bm.end_pos_shift[allele] -= (int32_t)(stop_pos - start_pos);
        for (uint32_t pos = start_pos; pos < stop_pos; ++pos)
            if (is_gc(var.var_seq.at(pos))) ++bm.gc_mod[allele];
    } else if (bm.unhandled_bases_in_variant[allele]) {                      // the rest of an insertion that reached out of the last fragment
        const Variant &var = vars.at((size_t)bm.unhandled_variant_id[allele]);
        const uint32_t start_pos = (uint32_t)var.var_seq.size() - bm.unhandled_bases_in_variant[allele];
        uint32_t stop_pos = start_pos + cur_end_position - last_end_position;
        if (stop_pos > var.var_seq.size()) {
            stop_pos = (uint32_t)var.var_seq.size();
            need_new_variants = true;
        }
        bm.end_pos_shift[allele] -= (int32_t)(stop_pos - start_pos);
        bm.unhandled_bases_in_variant[allele] -= stop_pos - start_pos;
        for (uint32_t pos = start_pos; pos < stop_pos; ++pos)
            if (is_gc(var.var_seq.at(pos))) ++bm.gc_mod[allele];
        if (0 == bm.unhandled_bases_in_variant[allele]) ++bm.unhandled_variant_id[allele];
    } else need_new_variants = true;
    if (need_new_variants) handle_gc_mod_and_end_pos_shift_for_new_variants(bm, allele, ref, cur_end_position);
}

// :1814-1827
inline void prepare_end_surroundings_for_current_fragment_length(VariantBiasMod &bm, const VarRef &ref, uint32_t cur_end_position, uint32_t allele) {
    const uint32_t corrected_pos = cur_end_position + (uint32_t)bm.end_pos_shift[allele];
    if (corrected_pos < ref.length()) {
        bm.surrounding_end[allele] = ref.reverse_surrounding(corrected_pos);
        variant_mod_end_surrounding(bm, allele, ref, corrected_pos);
    }
}

// :1829-1851
inline void prepare_bias_mod_for_current_fragment_length(VariantBiasMod &bm, const VarRef &ref, uint32_t cur_start_position, uint32_t fragment_length, uint32_t allele) {
    const uint32_t cur_end_position = cur_start_position + fragment_length - 1u;
    if (!(bm.last_end_position[allele] <= cur_end_position)) return;         // both strands of an allele come here: once is enough
    update_bias_mod_for_current_fragment_length(bm, ref, cur_start_position, cur_end_position, bm.last_end_position[allele], allele);
    if (bm.start_variant_pos && fragment_length <= (uint32_t)ref.variants->at((size_t)bm.first_variant_id).var_seq.size() - bm.start_variant_pos) {
        update_bias_mod_for_current_fragment_length(bm, ref, cur_start_position, cur_end_position + 1u, cur_end_position, allele);
        prepare_end_surroundings_for_current_fragment_length(bm, ref, cur_end_position, allele);
    } else {
        prepare_end_surroundings_for_current_fragment_length(bm, ref, cur_end_position, allele);
        update_bias_mod_for_current_fragment_length(bm, ref, cur_start_position, cur_end_position + 1u, cur_end_position, allele);
    }
    bm.last_end_position[allele] = cur_end_position + 1u;
}

// :1853-1868
inline uint32_t gc_percent_with_variants(VariantBiasMod &bm, const VarRef &ref, uint32_t cur_end_position, uint32_t fragment_length, uint32_t allele) {
    if (cur_end_position < bm.last_gc_end)
        return percent_u32((uint32_t)((int32_t)(bm.last_gc - ref.gc_content_absolut(cur_end_position, bm.last_gc_end)) + bm.gc_mod[allele]), fragment_length);
    bm.last_gc += ref.gc_content_absolut(bm.last_gc_end, cur_end_position);
    bm.last_gc_end = cur_end_position;
    return percent_u32((uint32_t)((int32_t)bm.last_gc + bm.gc_mod[allele]), fragment_length);
}

// :1870-1896
inline void check_for_inserted_bases_to_start_from(VariantBiasMod &bm, const VarRef &ref, uint32_t cur_start_position) {
    const std::vector<Variant> &vars = *ref.variants;
    if (!((size_t)bm.first_variant_id < vars.size() && vars[(size_t)bm.first_variant_id].position == cur_start_position)) return;
    if (bm.start_variant_pos) {
        if (++bm.start_variant_pos >= vars[(size_t)bm.first_variant_id].var_seq.size()) {
            bm.start_variant_pos = 0;
            ++bm.first_variant_id;
        }
    } else {
        while ((size_t)bm.first_variant_id < vars.size() && vars[(size_t)bm.first_variant_id].position == cur_start_position && 2 > vars[(size_t)bm.first_variant_id].var_seq.size())
            ++bm.first_variant_id;                                           // deletions and substitutions go with the first variant of the position
    }
    if (0 == bm.start_variant_pos && (size_t)bm.first_variant_id < vars.size() && vars[(size_t)bm.first_variant_id].position == cur_start_position) bm.start_variant_pos = 1;
}

// Reference::ReferenceSequence with variants (Reference.cpp:498-567): the template of one mate on one allele
inline std::vector<uint8_t> reference_sequence_with_variants(const VarRef &ref, uint32_t start_pos, uint32_t frag_length, bool reversed, std::pair<int32_t, uint32_t> first_variant,
                                                             uint32_t allele) {
    using namespace variants_detail;
    const std::vector<Variant> &vars = *ref.variants;
    const std::vector<uint8_t> &c = *ref.codes;
    std::vector<uint8_t> out;
    auto append = [&](const std::vector<uint8_t> &v) { out.insert(out.end(), v.begin(), v.end()); };
    auto infix = [&](uint32_t from, uint32_t to) {
        if (from > to || to > c.size()) throw Error("infix outside the reference sequence");
        return std::vector<uint8_t>(c.begin() + from, c.begin() + to);
    };
    uint32_t cur_start = start_pos;
    int32_t cur_var = first_variant.first;
    if (reversed) {
        if (first_variant.second) {
            append(complemented_reverse(vars.at((size_t)cur_var).var_seq.data(), first_variant.second));
            --cur_var;
            --cur_start;
        }
        for (; cur_var >= 0 && out.size() < frag_length; --cur_var) {
            const Variant &var = vars[(size_t)cur_var];
            if (!var.in_allele(allele)) continue;
            if (cur_start - var.position > frag_length - (uint32_t)out.size()) {
                const std::vector<uint8_t> part = infix(cur_start + (uint32_t)out.size() - frag_length, cur_start);      // the variant lies beyond the template
                append(complemented_reverse(part.data(), part.size()));
            } else {
                const std::vector<uint8_t> part = infix(var.position + 1u, cur_start);
                append(complemented_reverse(part.data(), part.size()));
                append(complemented_reverse(var.var_seq.data(), var.var_seq.size()));
                cur_start = var.position;
            }
        }
        if (cur_var == -1 && out.size() < frag_length) {
            const std::vector<uint8_t> part = infix(cur_start + (uint32_t)out.size() - frag_length, cur_start);
            append(complemented_reverse(part.data(), part.size()));
        }
    } else {
        if (first_variant.second) {
            append(sub(vars.at((size_t)cur_var).var_seq, first_variant.second, vars.at((size_t)cur_var).var_seq.size()));
            ++cur_var;
            ++cur_start;
        }
        for (; (size_t)cur_var < vars.size() && out.size() < frag_length; ++cur_var) {
            const Variant &var = vars[(size_t)cur_var];
            if (!var.in_allele(allele)) continue;
            if (var.position - cur_start >= frag_length - (uint32_t)out.size()) {
                append(infix(cur_start, cur_start + frag_length - (uint32_t)out.size()));
            } else {
                append(infix(cur_start, var.position));
                append(var.var_seq);
                cur_start = var.position + 1u;
            }
        }
        if ((size_t)cur_var == vars.size() && out.size() < frag_length) append(infix(cur_start, cur_start + frag_length - (uint32_t)out.size()));
    }
    if (out.size() > frag_length) out.resize(frag_length);
    return out;
}

}  // namespace orcv
