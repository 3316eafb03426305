// rsq_profile_archive.cpp -- a ReSeq profile as ReSeq itself stores it: `<name>.reseq` (DataStats) and `<name>.reseq.ipf`
// (ProbabilityEstimates), two Boost text archives.  Stands in for
//   DataStats::Load + PrepareProcessing          reseq/DataStats.cpp:1280-1300, 1322-1328, 698-703
//   AdapterStats::SumCounts / PrepareSimulation  reseq/AdapterStats.cpp:840-883, 892-908
//   ErrorStats::PrepareSimulation                reseq/ErrorStats.cpp:202-209
//   ProbabilityEstimates::Load + PrepareResult   reseq/ProbabilityEstimates.cpp:1022-1045, 961-1020
//     (LogIPF::FullExpansion ProbabilityEstimates.h:1004-1036, LogArrayCalc::Expand :253-290,
//      LogArrayResult::GetResults :386-453, ImputeMissingValues :455-479)
// The member lists below are the `serialize` functions of the reference, in their order, with the members' C++ types: they
// decide the token stream (rsq_archive.h).  Only what the simulation reads is kept; the rest is walked over.
// The fitting side (ProbabilityEstimates::Estimate with iterations) is out of scope: tables are taken as stored.
#include <math.h>
#include <stdio.h>
#include <unistd.h>

#include <algorithm>
#include <fstream>

#include "rsq_archive.h"
#include "rsq_host.h"
#include "rsq_textio.h"
#include "rsq_types.h"

namespace rsq {
namespace {
using archive::Member;
using archive::Node;
using archive::Schema;
using archive::TypeP;

struct ReseqTypes {
    Schema s;
    TypeP data_stats = nullptr, probability_estimates = nullptr;

    TypeP vect(TypeP t) {   // Vect<T>: Vect.hpp:40-42
        return s.cls("reseq::Vect<" + t->name + ">", {{"vec_", s.pair(s.u64(), s.vec(t)), true}});
    }
    TypeP seq_quality_stats(TypeP t) {   // SeqQualityStats<T>: SeqQualityStats.hpp:19-21
        return s.cls("reseq::SeqQualityStats<" + t->name + ">", {{"qualities_", vect(t), false}});
    }
    TypeP nest(TypeP t, std::initializer_list<size_t> dims) {   // std::array<std::array<T,d0>,d1>... : innermost extent first
        for (size_t d : dims) t = s.arr(t, d);
        return t;
    }

    ReseqTypes() {
        TypeP u8 = s.u8(), u16 = s.u16(), u32 = s.u32(), u64 = s.u64(), f64 = s.f64();
        TypeP v1 = vect(u64), v2 = vect(v1), v3 = vect(v2);
        TypeP sq = seq_quality_stats(u64), vsq = vect(sq), vvsq = vect(vsq);

        // AdapterStats.h:59-92
        TypeP adapter_stats = s.cls("reseq::AdapterStats", {
            {"names_", s.arr(s.vec(s.str()), 2), false},
            {"combinations_", s.vec(s.vec(s.boolean())), false},
            {"counts_", s.vec(s.vec(v2)), true},
            {"start_cut_", s.arr(s.vec(v1), 2), true},
            {"polya_tail_length_", v1, true},
            {"overrun_bases_", s.arr(u64, 5), true},
            {"seqs_archive", s.arr(s.vec(s.str()), 2), true},
        });
        // CoverageStats.h:281-315
        TypeP de = nest(v2, {4, 5, 4}), er = nest(v2, {5, 4});
        TypeP coverage_stats = s.cls("reseq::CoverageStats", {
            {"coverage_threshold_", u32, false},
            {"reset_distance_", u32, true},
            {"dominant_errors_by_distance_", de, false},
            {"dominant_errors_by_gc_", de, false},
            {"gc_by_distance_de_", de, false},
            {"dominant_errors_by_start_rates_", de, false},
            {"start_rates_by_distance_de_", de, false},
            {"start_rates_by_gc_de_", de, false},
            {"error_rates_by_distance_", er, false},
            {"error_rates_by_gc_", er, false},
            {"gc_by_distance_er_", er, false},
            {"error_rates_by_start_rates_", er, false},
            {"start_rates_by_distance_er_", er, false},
            {"start_rates_by_gc_er_", er, false},
            {"block_error_rate_", vect(u16), false},
            {"block_percent_systematic_", vect(u16), false},
            {"systematic_error_p_values_", v1, false},
            {"coverage_", v1, false},
            {"coverage_stranded_", s.arr(v1, 2), false},
            {"coverage_stranded_percent_", s.arr(v1, 2), false},
            {"coverage_stranded_percent_min_cov_10_", s.arr(v1, 2), false},
            {"coverage_stranded_percent_min_cov_20_", s.arr(v1, 2), false},
            {"error_coverage_", v1, false},
            {"error_coverage_percent_", v1, false},
            {"error_coverage_percent_min_cov_10_", v1, false},
            {"error_coverage_percent_min_cov_20_", v1, false},
            {"error_coverage_percent_stranded_", v2, false},
            {"error_coverage_percent_stranded_min_strand_cov_10_", v2, false},
            {"error_coverage_percent_stranded_min_strand_cov_20_", v2, false},
        });
        // ErrorStats.h:78-96
        TypeP per_tile = nest(v3, {5, 4, 2}), indel = nest(v2, {6, 2});
        TypeP error_stats = s.cls("reseq::ErrorStats", {
            {"called_bases_by_base_quality_per_tile_", per_tile, false},
            {"called_bases_by_position_per_tile_", per_tile, false},
            {"called_bases_by_error_num_per_tile_", per_tile, false},
            {"called_bases_by_error_rate_per_tile_", per_tile, false},
            {"error_num_by_quality_per_tile_", per_tile, false},
            {"error_num_by_position_per_tile_", per_tile, false},
            {"error_num_by_error_rate_per_tile_", per_tile, false},
            {"indel_by_indel_pos_", indel, true},
            {"indel_by_position_", indel, false},
            {"indel_by_gc_", indel, false},
            {"indel_pos_by_position_", indel, false},
            {"indel_pos_by_gc_", indel, false},
            {"gc_by_position_", indel, false},
            {"errors_per_read_", s.arr(v1, 2), false},
            {"called_bases_by_base_quality_per_previous_called_base_", nest(v1, {6, 5, 4, 2}), false},
        });
        // FragmentDuplicationStats.h:33-35
        TypeP duplication_stats = s.cls("reseq::FragmentDuplicationStats", {{"duplication_number_", v1, false}});
        // Surrounding.h:63-65, 89-91; FragmentDistributionStats.h:440-456
        TypeP surrounding_count = s.cls("reseq::SurroundingCount", {{"counts_", s.arr(s.vec(u64), 3), false}});
        TypeP surrounding_bias = s.cls("reseq::SurroundingBias", {{"bias_", s.arr(s.vec(f64), 3), true}});
        TypeP fragment_stats = s.cls("reseq::FragmentDistributionStats", {
            {"abundance_", s.vec(u64), false},
            {"insert_lengths_", v1, true},
            {"gc_fragment_content_", v1, false},
            {"fragment_surroundings_", surrounding_count, false},
            {"site_count_", v2, false},
            {"outskirt_content_", nest(v1, {4, 2}), false},
            {"ref_seq_bias_", s.vec(f64), true},
            {"insert_lengths_bias_", vect(f64), true},
            {"gc_fragment_content_bias_", vect(f64), true},
            {"fragment_surroundings_bias_", surrounding_bias, true},
            {"dispersion_parameters_", s.arr(f64, 2), true},
        });
        // QualityStats.h:156-197
        TypeP q542 = nest(v3, {5, 4, 2}), q42 = nest(v3, {4, 2}), q52 = nest(v3, {5, 2}), q2 = s.arr(v3, 2);
        TypeP quality_stats = s.cls("reseq::QualityStats", {
            {"base_quality_stats_per_tile_per_error_reference_", nest(vvsq, {5, 4, 2}), false},
            {"error_rate_for_position_per_tile_per_error_reference_", q542, false},
            {"base_quality_for_error_rate_per_tile_per_error_reference_", q542, false},
            {"base_quality_for_preceding_quality_per_tile_reference_", q42, false},
            {"preceding_quality_for_error_rate_per_tile_reference_", q42, false},
            {"preceding_quality_for_position_per_tile_reference_", q42, false},
            {"base_quality_for_sequence_quality_per_tile_reference_", q42, false},
            {"preceding_quality_for_sequence_quality_per_tile_reference_", q42, false},
            {"sequence_quality_for_error_rate_per_tile_reference_", q42, false},
            {"sequence_quality_for_position_per_tile_reference_", q42, false},
            {"sequence_quality_mean_for_gc_per_tile_reference_", s.arr(vvsq, 2), false},
            {"sequence_quality_mean_for_mean_error_rate_per_tile_reference_", q2, false},
            {"sequence_quality_mean_for_fragment_length_per_tile_reference_", q2, false},
            {"mean_error_rate_for_gc_per_tile_reference_", q2, false},
            {"mean_error_rate_for_fragment_length_per_tile_reference_", q2, false},
            {"gc_for_fragment_length_per_tile_reference_", q2, false},
            {"base_quality_for_sequence_per_tile_", q52, false},
            {"base_quality_for_preceding_quality_per_tile_", q52, false},
            {"base_quality_stats_per_tile_", nest(vvsq, {5, 2}), false},
            {"preceding_quality_for_sequence_per_tile_", q52, false},
            {"preceding_quality_for_position_per_tile_", q52, false},
            {"sequence_quality_for_position_per_tile_", q52, false},
            {"base_quality_stats_per_strand_", s.arr(vsq, 2), false},
            {"sequence_quality_for_base_per_tile_", nest(vvsq, {5, 2}), false},
            {"sequence_quality_mean_paired_per_tile_", v3, false},
            {"sequence_quality_mean_for_gc_per_tile_", s.arr(vvsq, 2), false},
            {"sequence_quality_probability_mean_", s.arr(v1, 2), false},
            {"sequence_quality_minimum_", s.arr(v1, 2), false},
            {"sequence_quality_first_quartile_", s.arr(v1, 2), false},
            {"sequence_quality_median_", s.arr(v1, 2), false},
            {"sequence_quality_third_quartile_", s.arr(v1, 2), false},
            {"sequence_quality_maximum_", s.arr(v1, 2), false},
            {"sequence_quality_content_", s.arr(v2, 2), false},
            {"homoquality_distribution_", v2, false},
            {"nucleotide_quality_", nest(sq, {5, 2}), false},
        });
        // TileStats.h:42-52
        TypeP tile_stats = s.cls("reseq::TileStats", {{"tiles_", s.vec(u16), true}, {"abundance_", s.vec(u64), true}});
        // DataStats.h:180-212
        data_stats = s.cls("reseq::DataStats", {
            {"adapters_", adapter_stats, true},
            {"coverage_", coverage_stats, true},
            {"errors_", error_stats, true},
            {"duplicates_", duplication_stats, false},
            {"fragment_distribution_", fragment_stats, true},
            {"qualities_", quality_stats, false},
            {"tiles_", tile_stats, true},
            {"creation_time_", u64, true},
            {"read_lengths_", s.arr(v1, 2), true},
            {"read_lengths_by_fragment_length_", s.arr(v2, 2), true},
            {"non_mapped_read_lengths_by_fragment_length_", s.arr(v2, 2), true},
            {"phred_quality_offset_", u8, true},
            {"minimum_quality_", u8, false},
            {"maximum_quality_", u8, false},
            {"minimum_read_length_on_reference_", u16, false},
            {"maximum_read_length_on_reference_", u16, false},
            {"corrected_coverage_", f64, true},
            {"proper_pair_mapping_quality_", v1, false},
            {"improper_pair_mapping_quality_", v1, false},
            {"single_read_mapping_quality_", v1, false},
            {"gc_read_content_", s.arr(v1, 2), false},
            {"gc_read_content_reference_", s.arr(v1, 2), false},
            {"gc_read_content_mapped_", s.arr(v1, 2), false},
            {"n_content_", s.arr(v1, 2), false},
            {"sequence_content_", nest(v1, {5, 2}), false},
            {"sequence_content_reference_", nest(v1, {4, 2, 2}), false},
            {"homopolymer_distribution_", s.arr(v1, 5), false},
        });

        // ProbabilityEstimates.h:111-114 (LogArrayCalc), :955-967 (LogIPF), :1475-1483 (ProbabilityEstimates)
        TypeP ipf[6] = {};
        for (size_t n : {4, 5}) {
            const size_t margins = n * (n - 1) / 2;
            const std::string ns = std::to_string(n) + ">";
            TypeP calc = s.cls("reseq::ProbabilityEstimatesSubClasses::LogArrayCalc<" + ns, {
                {"dim2_", s.arr(s.vec(f64), margins), true},
                {"dim_size_", s.arr(u32, n), true},
            });
            TypeP index_maps = s.arr(s.vec(u32), n);
            ipf[n] = s.cls("reseq::ProbabilityEstimatesSubClasses::LogIPF<" + ns, {
                {"steps_", u32, true},
                {"needed_updates_", u32, false},
                {"precision_", f64, true},
                {"margin_precision_", s.arr(f64, margins), false},
                {"last_margin_", u16, false},
                {"last_update_", s.arr(u32, margins), false},
                {"update_dist_", s.arr(u16, margins), false},
                {"estimates_", calc, true},
                {"dim_indices_", index_maps, true},
                {"initial_dim_indices_reduced_", index_maps, true},
                {"dim_indices_reduced_", index_maps, true},
            });
        }
        probability_estimates = s.cls("reseq::ProbabilityEstimates", {
            {"stats_creation_time_", u64, true},
            {"quality_", s.arr(s.vec(s.arr(ipf[5], 4)), 2), true},
            {"sequence_quality_", s.arr(s.vec(ipf[4]), 2), true},
            {"base_call_", s.arr(s.vec(nest(ipf[5], {5, 4})), 2), true},
            {"dom_error_", nest(ipf[4], {5, 5, 4}), true},
            {"error_rate_", nest(ipf[4], {5, 4}), true},
            {"indels_", nest(ipf[4], {6, 2}), true},
        });
    }
};

std::vector<char> slurp(const std::string &path) {
    textio::Reader f;
    if (!f.open(path)) throw Error("File '" + path + "' does not exists or no read permission given.");   // DataStats.cpp:1281-1284
    std::vector<char> buf;
    size_t have = 0;
    for (;;) {
        if (buf.size() - have < (1u << 22)) buf.resize(std::max<size_t>(buf.size() * 2, 1u << 24));
        const int n = f.read(buf.data() + have, (unsigned)std::min<size_t>(buf.size() - have, 1u << 30));
        if (n < 0) throw Error("read error in " + path);
        if (n == 0) break;
        have += (size_t)n;
    }
    buf.resize(have);
    return buf;
}

template <class T>
Vect<T> to_vect(const Node &v) {   // a kept Vect<T> node
    const Node &p = v["vec_"];
    Vect<T> out;
    out.from = p.first().uint();
    const Node &vals = p.second();
    if constexpr (std::is_same<T, double>::value) out.v = vals.f;
    else out.v.assign(vals.u.begin(), vals.u.end());
    return out;
}
uint64_t vect_from(const Node &v) { return v["vec_"].first().uint(); }
const Node &vect_items(const Node &v) { return v["vec_"].second(); }
uint64_t vect_to(const Node &v) { return vect_from(v) + vect_items(v).size(); }

// value of a Vect<Vect<u64>> at [i][j], 0 outside (Vect::operator[] const, Vect.hpp:184-191)
uint64_t at2(const Node &vv, uint64_t i, uint64_t j) {
    if (i < vect_from(vv) || i >= vect_to(vv)) return 0;
    const Node &row = vect_items(vv)[i - vect_from(vv)];
    if (j < vect_from(row) || j >= vect_to(row)) return 0;
    return vect_items(row).u[j - vect_from(row)];
}

// ------------------------------------------------------------------------------------------------ adapters
uint8_t dna_code(char c) {   // seqan::Dna from a character: everything that is not C, G, T/U is A
    switch (c) {
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        default: return 0;
    }
}

// AdapterStats::SumCounts (AdapterStats.cpp:840-883): an adapter pair's detections count from the first length on at which both
// adapters differ from their neighbours in the (content-sorted) list; PrepareSimulation (:892-908): adapters below a tenth of
// the most frequent one are not simulated.
void sum_adapter_counts(const Node &counts, const std::vector<std::string> (&seqs)[2], std::vector<uint64_t> (&sum)[2]) {
    const size_t n1 = counts.size(), n2 = n1 ? counts[0].size() : 0;
    sum[0].assign(seqs[0].size(), 0);
    sum[1].assign(seqs[1].size(), 0);
    auto common_prefix = [](const std::string &a, const std::string &b) {   // compared as seqan::Dna values
        uint16_t k = 0;
        while (k < std::min(a.size(), b.size()) && dna_code(a[k]) == dna_code(b[k])) ++k;
        return k;
    };
    uint16_t before1 = 0;
    for (size_t a1 = n1; a1--;) {
        const uint16_t after1 = a1 ? common_prefix(seqs[0].at(a1), seqs[0].at(a1 - 1)) : 0;
        uint16_t before2 = 0;
        for (size_t a2 = n2; a2--;) {
            const uint16_t after2 = a2 ? common_prefix(seqs[1].at(a2), seqs[1].at(a2 - 1)) : 0;
            const Node &by_len1 = counts[a1][a2];
            uint64_t total = 0;
            for (uint64_t pos1 = std::max<uint16_t>(std::max(before1, after1), (uint16_t)vect_from(by_len1)); pos1 < vect_to(by_len1); ++pos1) {
                const Node &by_len2 = vect_items(by_len1)[pos1 - vect_from(by_len1)];
                for (uint64_t pos2 = std::max<uint16_t>(std::max(before2, after2), (uint16_t)vect_from(by_len2)); pos2 < vect_to(by_len2); ++pos2)
                    total += vect_items(by_len2).u[pos2 - vect_from(by_len2)];
            }
            sum[0].at(a1) += total;
            sum[1].at(a2) += total;
            before2 = after2;
        }
        before1 = after1;
    }
}

void fill_from_stats(Profile &p, const Node &st) {
    p.phred_offset = (uint8_t)st["phred_quality_offset_"].uint();
    p.corrected_coverage = st["corrected_coverage_"].real();
    p.reset_distance = (uint32_t)st["coverage_"]["reset_distance_"].uint();
    // ErrorStats::PrepareSimulation: the longest deletion seen
    p.max_len_deletion = 0;
    for (const Node &v : st["errors_"]["indel_by_indel_pos_"][1].kids) p.max_len_deletion = std::max<uint16_t>(p.max_len_deletion, (uint16_t)vect_to(v));

    p.total_number_reads = 0;
    for (int seg = 0; seg < 2; ++seg) {
        p.read_lengths[seg] = to_vect<uint64_t>(st["read_lengths_"][seg]);
        for (uint64_t x : p.read_lengths[seg].v) p.total_number_reads += x;   // DataStats::PrepareGeneral
        const Node &by_fl = st["read_lengths_by_fragment_length_"][seg], &non_mapped = st["non_mapped_read_lengths_by_fragment_length_"][seg];
        HostRlByFl &r = p.rl_by_fl[seg];
        r.from = vect_from(by_fl);
        r.row_ptr.assign(1, 0);
        for (uint64_t fl = r.from; fl < vect_to(by_fl); ++fl) {
            const Node &row = vect_items(by_fl)[fl - r.from];
            r.row_from.push_back((uint32_t)vect_from(row));
            for (uint64_t rl = vect_from(row); rl < vect_to(row); ++rl) {
                r.values.push_back(vect_items(row).u[rl - vect_from(row)]);
                r.non_mapped.push_back(at2(non_mapped, fl, rl));
            }
            r.row_ptr.push_back((uint32_t)r.values.size());
        }
    }

    const Node &tiles = st["tiles_"];
    p.tiles.assign(tiles["tiles_"].u.begin(), tiles["tiles_"].u.end());
    p.tile_abundance = tiles["abundance_"].u;
    if (p.tiles.empty() || p.tiles.size() != p.tile_abundance.size()) throw Error("profile without tiles (TileStats::tiles_ / abundance_)");

    const Node &ad = st["adapters_"];
    std::vector<std::string> seqs[2];
    for (int seg = 0; seg < 2; ++seg)
        for (const Node &s : ad["seqs_archive"][seg].kids) seqs[seg].push_back(s.s);
    std::vector<uint64_t> sums[2];
    sum_adapter_counts(ad["counts_"], seqs, sums);
    for (int seg = 0; seg < 2; ++seg) {
        HostAdapters &a = p.adapters[seg];
        a.seq_ptr.assign(1, 0);
        for (const std::string &s : seqs[seg]) {
            for (char c : s) a.seqs.push_back(dna_code(c));
            a.seq_ptr.push_back((uint32_t)a.seqs.size());
        }
        a.counts = sums[seg];
        a.significant = sums[seg];
        if (!a.counts.empty()) {
            const uint64_t threshold = (uint64_t)ceil((double)*std::max_element(a.counts.begin(), a.counts.end()) * 0.1);   // kMinFractionOfMaximumForSimulation
            for (uint64_t &c : a.significant)
                if (c < threshold) c = 0;
        }
        const Node &cuts = ad["start_cut_"][seg];
        if (cuts.size() != seqs[seg].size()) throw Error("adapter start cuts and adapter sequences differ in number");
        a.cut_ptr.assign(1, 0);
        for (const Node &c : cuts.kids) {
            a.cut_from.push_back((uint32_t)vect_from(c));
            a.cut.insert(a.cut.end(), vect_items(c).u.begin(), vect_items(c).u.end());
            a.cut_ptr.push_back((uint32_t)a.cut.size());
        }
    }
    p.polya = to_vect<uint64_t>(ad["polya_tail_length_"]);
    for (int i = 0; i < 5; ++i) p.overrun_bases[i] = ad["overrun_bases_"].u.at(i);

    const Node &fd = st["fragment_distribution_"];
    p.insert_lengths = to_vect<uint64_t>(fd["insert_lengths_"]);
    p.insert_lengths_bias = to_vect<double>(fd["insert_lengths_bias_"]);
    p.gc_bias = to_vect<double>(fd["gc_fragment_content_bias_"]);
    p.ref_seq_bias = fd["ref_seq_bias_"].f;
    const Node &sur = fd["fragment_surroundings_bias_"]["bias_"];
    p.sur_bias.clear();
    for (uint32_t b = 0; b < kSurBlocks; ++b) {
        if (sur[b].f.size() != kSurSize) throw Error("surrounding bias block does not hold 4^10 values");
        p.sur_bias.insert(p.sur_bias.end(), sur[b].f.begin(), sur[b].f.end());
    }
    p.dispersion[0] = fd["dispersion_parameters_"].f.at(0);
    p.dispersion[1] = fd["dispersion_parameters_"].f.at(1);
}

// ---------------------------------------------------------------------------------------- PrepareResult
// One fitted table: the margins (outcome x condition n) of the stored LogArrayCalc, un-binned, become a LogArrayResult.
//
// Stored state: `estimates_.dim2_[m]` for margin m = (a, 0) (a = 1..N-1 are the first N-1 margins, MapDim2To1 :27-29) is
// row-major [bin of dimension a][bin of dimension 0] over the REDUCED bins; `initial_dim_indices_reduced_` maps a full index
// to its initially reduced bin, `dim_indices_reduced_` that bin to its final bin, `dim_indices_` a full index to the value in
// the statistics (quality, position, ...).
//
// FullExpansion + Expand: every full index gets the value of its bin, times (1/bins sharing it)^(1/(N-1)) for each of the
// two dimensions.  Only margins containing dimension 0 are ever read by LogArrayResult, so only those are expanded here.
// GetResults: columns (outcomes) sorted by ascending mean likelihood, rows placed at value - smallest value.
// ImputeMissingValues: rows that are entirely zero between two filled rows are interpolated -- with the weights as the
// reference has them (the nearer row gets the SMALLER weight, ProbabilityEstimates.h:472), which is reproduced on purpose.
HostTable prepare_result(const Node &ipf, uint32_t n_dims) {
    HostTable t;
    t.nm = n_dims - 1;
    std::vector<std::vector<uint32_t>> values(n_dims), bin(n_dims);
    std::vector<std::vector<double>> weight(n_dims);
    std::vector<size_t> n_bins(n_dims, 0);
    for (uint32_t d = 0; d < n_dims; ++d) {
        const std::vector<uint64_t> &full = ipf["dim_indices_"][d].u, &initial = ipf["initial_dim_indices_reduced_"][d].u, &reduced = ipf["dim_indices_reduced_"][d].u;
        if (initial.size() != full.size()) throw Error("IPF table: index maps of different lengths");
        values[d].assign(full.begin(), full.end());
        bin[d].resize(full.size());
        for (size_t i = 0; i < full.size(); ++i) {
            if (initial[i] >= reduced.size()) throw Error("IPF table: bin index outside the reduction map");
            bin[d][i] = (uint32_t)reduced[initial[i]];
        }
        n_bins[d] = bin[d].empty() ? 0 : (size_t)*std::max_element(bin[d].begin(), bin[d].end()) + 1;
    }
    const size_t k = values[0].size();
    if (!k) return t;   // no data: limits stay {0,0}, every draw falls back (ProbabilityEstimates.h:446-451)

    // is anything binned at all?  (FullExpansion's two identity tests)
    bool binned = false;
    for (uint32_t d = 0; d < n_dims && !binned; ++d) {
        const std::vector<uint64_t> &initial = ipf["initial_dim_indices_reduced_"][d].u, &reduced = ipf["dim_indices_reduced_"][d].u;
        for (size_t i = 0; i < reduced.size() && !binned; ++i) binned = reduced[i] != i;
        for (size_t i = 0; i < initial.size() && !binned; ++i) binned = initial[i] != i;
    }
    for (uint32_t d = 0; d < n_dims; ++d) {
        weight[d].assign(bin[d].size(), 1.0);
        if (!binned) continue;
        std::vector<uint32_t> sharing(n_bins[d], 0);
        for (uint32_t b : bin[d]) ++sharing[b];
        for (size_t i = 0; i < bin[d].size(); ++i) weight[d][i] = pow(1.0 / sharing[bin[d][i]], 1.0 / (n_dims - 1));
    }

    const Node &stored = ipf["estimates_"]["dim2_"];
    std::vector<std::vector<double>> margin(t.nm);   // [row of dimension a][column of dimension 0], full size
    for (uint32_t m = 0; m < t.nm; ++m) {
        const uint32_t a = m + 1;
        const std::vector<double> &src = stored[m].f;
        const size_t rows = values[a].size(), src_cols = binned ? n_bins[0] : k, src_rows = binned ? n_bins[a] : rows;
        if (src.size() != src_rows * src_cols) throw Error("IPF table: a stored margin does not match its index maps");
        margin[m].resize(rows * k);
        for (size_t i = 0; i < rows; ++i)
            for (size_t j = 0; j < k; ++j)
                margin[m][i * k + j] = binned ? src[(size_t)bin[a][i] * src_cols + bin[0][j]] * weight[a][i] * weight[0][j] : src[i * k + j];
    }

    // column order: mean likelihood over the rows, summed over the margins (last margin first, last row first)
    std::vector<std::pair<double, uint32_t>> order(k);
    for (size_t j = 0; j < k; ++j) order[j] = {0.0, (uint32_t)j};
    for (uint32_t m = t.nm; m--;) {
        const size_t rows = values[m + 1].size();
        for (size_t j = k; j--;) {
            double sum = 0.0;
            for (size_t i = rows; i--;) sum += margin[m][i * k + j];
            order[j].first += sum / rows;
        }
    }
    std::sort(order.begin(), order.end());
    std::vector<uint32_t> column(k);
    t.par0.resize(k);
    for (size_t c = 0; c < k; ++c) {
        column[order[c].second] = (uint32_t)c;
        t.par0[c] = values[0][order[c].second];
    }

    for (uint32_t m = 0; m < t.nm; ++m) {
        const std::vector<uint32_t> &val = values[m + 1];
        t.from[m] = *std::min_element(val.begin(), val.end());
        t.to[m] = *std::max_element(val.begin(), val.end()) + 1;
        std::vector<double> &dst = t.dim2[m];
        dst.assign((size_t)(t.to[m] - t.from[m]) * k, 0.0);
        for (size_t i = val.size(); i--;)
            for (size_t j = k; j--;) dst[(size_t)(val[i] - t.from[m]) * k + column[j]] = margin[m][i * k + j];

        // rows without data between rows with data
        uint32_t last_filled = 0;
        for (uint32_t i = 1; i < t.to[m] - t.from[m]; ++i) {
            bool filled = false;
            for (size_t j = 0; j < k && !filled; ++j) filled = dst[(size_t)i * k + j] != 0.0;
            if (!filled) continue;
            for (uint32_t gap = last_filled + 1; gap < i; ++gap)
                for (size_t j = 0; j < k; ++j)
                    dst[(size_t)gap * k + j] = dst[(size_t)last_filled * k + j] * (gap - last_filled) / (i - last_filled) + dst[(size_t)i * k + j] * (i - gap) / (i - last_filled);
            last_filled = i;
        }
    }
    for (uint32_t v : t.par0)
        if (v > 255) throw Error("IPF table: outcome value above 255");
    return t;
}

void fill_from_estimates(Profile &p, const Node &pe, double precision_aim, std::string &warnings) {
    const uint32_t nt = p.n_tiles();
    size_t unconverged = 0;
    auto table = [&](const Node &ipf, uint32_t n_dims) {
        if (ipf["steps_"].uint() && ipf["precision_"].real() > precision_aim) ++unconverged;
        return prepare_result(ipf, n_dims);
    };
    for (uint32_t seg = 0; seg < 2; ++seg) {
        const Node &q = pe["quality_"][seg], &sq = pe["sequence_quality_"][seg], &bc = pe["base_call_"][seg];
        if (q.size() != nt || sq.size() != nt || bc.size() != nt) throw Error("probability estimates and statistics disagree on the number of tiles");
        for (uint32_t tile = 0; tile < nt; ++tile) {
            p.seq_quality.push_back(table(sq[tile], 4));
            for (uint32_t base = 0; base < 4; ++base) {
                p.quality.push_back(table(q[tile][base], 5));
                for (uint32_t dom = 0; dom < 5; ++dom) p.base_call.push_back(table(bc[tile][base][dom], 5));
            }
        }
    }
    for (uint32_t base = 0; base < 4; ++base)
        for (uint32_t prev = 0; prev < 5; ++prev)
            for (uint32_t dom5 = 0; dom5 < 5; ++dom5) p.dom_error.push_back(table(pe["dom_error_"][base][prev][dom5], 4));
    for (uint32_t base = 0; base < 4; ++base)
        for (uint32_t dom = 0; dom < 5; ++dom) p.error_rate.push_back(table(pe["error_rate_"][base][dom], 4));
    for (uint32_t type = 0; type < 2; ++type)
        for (uint32_t call = 0; call < 6; ++call) p.indels.push_back(table(pe["indels_"][type][call], 4));
    if (unconverged)
        warnings += std::to_string(unconverged) + " fitted tables are stored with a precision above the aim: ReSeq would continue their fit (ProbabilityEstimates.h:1049-1075), "
                    "this build uses them as stored. ";
}

}  // namespace

bool Profile::is_archive(const std::string &path) {
    textio::Reader f;
    if (!f.open(path)) return false;
    char head[32];
    const int n = f.read(head, sizeof head);
    return n > 0 && archive::Reader::looks_like_archive(head, (size_t)n);
}

// ------------------------------------------------------------------------------------------------ writing
// The inverse of load_archives for a loaded profile: a DataStats and a ProbabilityEstimates value whose PrepareProcessing / PrepareResult give this profile again,
// written under the recalled token rules (rsq_archive.h) -- so that a ReSeq user can hand a profile of this library to the original binary (and tell us whether
// the binary reads it: INTEGRATION.md "Profile files").  What the simulation never reads is written default-constructed, except where the loaders insist on a shape
// (the 3 x 4^10 surrounding counts).  A result table is already laid out as GetResults leaves it, so it is stored as a converged, un-binned fit whose dimension 0
// holds the table's columns in their order; the margins between two conditions, which the simulation never reads, are ones.
namespace {

Node default_node(TypeP t) {
    Node n;
    n.type = t;
    switch (t->kind) {
        case archive::Type::UINT:
        case archive::Type::INT:
        case archive::Type::BOOL: n.u.assign(1, 0); break;
        case archive::Type::F64: n.f.assign(1, 0.0); break;
        case archive::Type::STR:
        case archive::Type::VEC: break;
        case archive::Type::ARR:
            if (t->elem->numeric()) (t->elem->kind == archive::Type::F64 ? (void)n.f.assign(t->n, 0.0) : (void)n.u.assign(t->n, 0));
            else n.kids.assign(t->n, default_node(t->elem));
            break;
        case archive::Type::PAIR: n.kids = {default_node(t->first), default_node(t->second)}; break;
        case archive::Type::CLS:
            for (const Member &m : t->members) n.kids.push_back(default_node(m.type));
            break;
    }
    return n;
}
Node &member(Node &cls, const char *name) {
    for (size_t i = 0; i < cls.type->members.size(); ++i)
        if (cls.type->members[i].name == name) return cls.kids[i];
    throw Error(std::string("archive: no member ") + name + " in " + cls.type->name);
}
void set_uint(Node &n, uint64_t v) { n.u.assign(1, v); }
void set_real(Node &n, double v) { n.f.assign(1, v); }
// a Vect<T> node (class {vec_: pair(offset, vector)}) from an offset and numbers
template <class T>
void set_vect(Node &n, uint64_t from, const T *values, size_t count) {
    Node &pair = member(n, "vec_");
    set_uint(pair.kids[0], from);
    Node &vec = pair.kids[1];
    if (vec.type->elem->kind == archive::Type::F64) vec.f.assign(values, values + count);
    else vec.u.assign(values, values + count);
}
template <class T>
void set_vect(Node &n, const Vect<T> &v) { set_vect(n, v.from, v.v.data(), v.v.size()); }
// a Vect<Vect<u64>> node: offset and rows (each an offset and numbers)
struct Row {
    uint64_t from = 0;
    std::vector<uint64_t> values;
};
void set_vect2(Node &n, uint64_t from, const std::vector<Row> &rows) {
    Node &pair = member(n, "vec_");
    set_uint(pair.kids[0], from);
    Node &vec = pair.kids[1];
    vec.kids.clear();
    for (const Row &r : rows) {
        Node row = default_node(vec.type->elem);
        set_vect(row, r.from, r.values.data(), r.values.size());
        vec.kids.push_back(std::move(row));
    }
}

class ArchiveWriter {
   public:
    ArchiveWriter(size_t n_types, uint32_t library_version) : seen_(n_types, 0), version_(library_version) { out_ = "22 serialization::archive " + std::to_string(library_version); }
    void put(TypeP t, const Node &v) {
        const bool class_info = t->kind == archive::Type::CLS || t->kind == archive::Type::ARR || t->kind == archive::Type::PAIR || (t->kind == archive::Type::VEC && !t->elem->numeric());
        if (class_info && !seen_[t->id]) {
            seen_[t->id] = 1;
            out_ += " 0 0";
        }
        switch (t->kind) {
            case archive::Type::UINT:
            case archive::Type::BOOL: uint(v.u.at(0)); break;
            case archive::Type::INT: out_ += ' ' + std::to_string((int64_t)v.u.at(0)); break;
            case archive::Type::F64: real(v.f.at(0)); break;
            case archive::Type::STR: out_ += ' ' + std::to_string(v.s.size()) + ' ' + v.s; break;
            case archive::Type::VEC: {
                const size_t count = t->elem->numeric() ? (t->elem->kind == archive::Type::F64 ? v.f.size() : v.u.size()) : v.kids.size();
                uint(count);
                if (t->elem->kind != archive::Type::BOOL && version_ > 3) out_ += " 0";      // item_version (the recalled rule: behind every count but vector<bool>'s)
                items(t->elem, count, v);
                break;
            }
            case archive::Type::ARR: {
                const size_t count = t->elem->numeric() ? (t->elem->kind == archive::Type::F64 ? v.f.size() : v.u.size()) : v.kids.size();
                if (count != t->n) throw Error("archive: " + t->name + " given " + std::to_string(count) + " items");
                uint(t->n);
                items(t->elem, count, v);
                break;
            }
            case archive::Type::PAIR:
                put(t->first, v.kids.at(0));
                put(t->second, v.kids.at(1));
                break;
            case archive::Type::CLS:
                for (size_t i = 0; i < t->members.size(); ++i) put(t->members[i].type, v.kids.at(i));
                break;
        }
    }
    void write(const std::string &path) {
        out_ += '\n';
        // under another name in the same directory first (with the same suffix: write_text_file picks the compression by it), then renamed into place -- nobody finds half a file
        const size_t slash = path.rfind('/');
        const std::string dir = slash == std::string::npos ? "" : path.substr(0, slash + 1), base = slash == std::string::npos ? path : path.substr(slash + 1);
        const std::string tmp = dir + ".writing." + std::to_string((long)getpid()) + "." + base;
        try {
            write_text_file(tmp, out_);
        } catch (...) {
            unlink(tmp.c_str());
            throw;
        }
        if (rename(tmp.c_str(), path.c_str()) != 0) {
            unlink(tmp.c_str());
            throw Error("Could not write '" + path + "'.");
        }
    }

   private:
    void uint(uint64_t v) { out_ += ' ' + std::to_string(v); }
    void real(double v) {
        char buf[40];
        snprintf(buf, sizeof buf, " %.17e", v);
        out_ += buf;
    }
    void items(TypeP e, size_t count, const Node &v) {
        if (e->kind == archive::Type::F64)
            for (size_t i = 0; i < count; ++i) real(v.f[i]);
        else if (e->kind == archive::Type::INT)
            for (size_t i = 0; i < count; ++i) out_ += ' ' + std::to_string((int64_t)v.u[i]);
        else if (e->numeric())
            for (size_t i = 0; i < count; ++i) uint(v.u[i]);
        else
            for (size_t i = 0; i < count; ++i) put(e, v.kids[i]);
    }
    std::string out_;
    std::vector<char> seen_;
    uint32_t version_;
};

// a result table as a converged, un-binned LogIPF<N> (N = margins + 1) whose PrepareResult is the table (tests/archive_fixtures.py ipf_from_table states the same)
void set_ipf(Node &ipf, const HostTable &t) {
    const uint32_t n_dims = t.nm + 1u, n_margins = n_dims * (n_dims - 1u) / 2u;
    const size_t k = t.par0.size();
    Node &estimates = member(ipf, "estimates_");
    const double kMax = 1.7976931348623157e308;
    if (!k) {                                                          // a table without data: a default-constructed fit
        set_uint(member(ipf, "steps_"), 0);
        set_real(member(ipf, "precision_"), kMax);
        member(ipf, "margin_precision_").f.assign(n_margins, kMax);
        member(ipf, "update_dist_").u.assign(n_margins, 2);
        return;
    }
    const uint32_t steps = 37;
    const double precision = 0.01;
    set_uint(member(ipf, "steps_"), steps);
    set_uint(member(ipf, "needed_updates_"), steps);
    set_real(member(ipf, "precision_"), precision);
    member(ipf, "margin_precision_").f.assign(n_margins, precision);
    set_uint(member(ipf, "last_margin_"), 1);
    member(ipf, "last_update_").u.assign(n_margins, steps);
    member(ipf, "update_dist_").u.assign(n_margins, 2);
    std::vector<uint32_t> size(n_dims);
    size[0] = (uint32_t)k;
    Node &dims = member(ipf, "dim_indices_"), &initial = member(ipf, "initial_dim_indices_reduced_"), &reduced = member(ipf, "dim_indices_reduced_");
    dims.kids[0].u.assign(t.par0.begin(), t.par0.end());
    for (uint32_t n = 1; n < n_dims; ++n) {
        size[n] = t.to[n - 1] - t.from[n - 1];
        for (uint32_t v = t.from[n - 1]; v < t.to[n - 1]; ++v) dims.kids[n].u.push_back(v);
    }
    for (uint32_t n = 0; n < n_dims; ++n) {
        for (uint32_t i = 0; i < size[n]; ++i) initial.kids[n].u.push_back(i);
        reduced.kids[n].u = initial.kids[n].u;
    }
    member(estimates, "dim_size_").u.assign(size.begin(), size.end());
    Node &dim2 = member(estimates, "dim2_");
    for (uint32_t n = 0; n < t.nm; ++n) dim2.kids[n].f = t.dim2[n];   // margin n: dimension n + 1 against the columns
    // the margins between two conditions, in the (dim_a, dim_b) walk of LogArrayCalc::SetUp
    uint32_t a = n_dims, b = n_dims - 1u;
    std::vector<std::pair<uint32_t, uint32_t>> pair_of(n_margins);
    for (uint32_t n = n_margins; n--;) {
        if (--a == b) {
            --b;
            a = n_dims - 1u;
        }
        pair_of[n] = {a, b};
    }
    for (uint32_t n = n_dims - 1u; n < n_margins; ++n) dim2.kids[n].f.assign((size_t)size[pair_of[n].first] * size[pair_of[n].second], 1.0);
}

}  // namespace

void Profile::save_archives(const std::string &stats_path, const std::string &ipf_path_in, uint64_t creation_time) const {
    static ReseqTypes types;
    const std::string ipf_path = ipf_path_in.empty() ? stats_path + ".ipf" : ipf_path_in;
    const Profile &p = *this;
    const uint32_t nt = p.n_tiles();
    {
        Node st = default_node(types.data_stats);
        set_uint(member(st, "creation_time_"), creation_time);
        set_uint(member(st, "phred_quality_offset_"), p.phred_offset);
        set_real(member(st, "corrected_coverage_"), p.corrected_coverage);
        Node &coverage = member(st, "coverage_");
        set_uint(member(coverage, "reset_distance_"), p.reset_distance);
        set_uint(member(coverage, "coverage_threshold_"), 10);
        // the longest deletion is all the simulation takes from the indel statistics (ErrorStats::PrepareSimulation): deletions of up to max_len_deletion bases
        // behind an 'A' call, shorter ones elsewhere; insertions do not count
        {
            Node &indel = member(member(st, "errors_"), "indel_by_indel_pos_");
            const uint32_t d = p.max_len_deletion;
            set_vect2(indel.kids[1].kids[0], 0, std::vector<Row>(d, Row{0, {5, 1}}));
            set_vect2(indel.kids[1].kids[2], 0, std::vector<Row>(d ? d - 1u : 0u, Row{0, {9}}));
            set_vect2(indel.kids[0].kids[1], 0, std::vector<Row>(d + 3u, Row{0, {4, 0, 1}}));
        }
        Node &fd = member(st, "fragment_distribution_");
        set_vect(member(fd, "insert_lengths_"), p.insert_lengths);
        set_vect(member(fd, "insert_lengths_bias_"), p.insert_lengths_bias);
        set_vect(member(fd, "gc_fragment_content_bias_"), p.gc_bias);
        if (p.sur_bias.size() != (size_t)kSurBlocks * kSurSize) throw Error("profile without the 3 x 4^10 surrounding biases");
        for (uint32_t b = 0; b < kSurBlocks; ++b) {
            member(member(fd, "fragment_surroundings_bias_"), "bias_").kids[b].f.assign(p.sur_bias.begin() + (size_t)b * kSurSize, p.sur_bias.begin() + (size_t)(b + 1) * kSurSize);
            member(member(fd, "fragment_surroundings_"), "counts_").kids[b].u.assign(kSurSize, 0);
        }
        member(fd, "dispersion_parameters_").f.assign(p.dispersion, p.dispersion + 2);
        member(fd, "ref_seq_bias_").f = p.ref_seq_bias;
        member(fd, "abundance_").u.assign(p.ref_seq_bias.size(), 7);
        for (int seg = 0; seg < 2; ++seg) {
            set_vect(member(st, "read_lengths_").kids[seg], p.read_lengths[seg]);
            const HostRlByFl &r = p.rl_by_fl[seg];
            std::vector<Row> rows, nm_rows;
            int64_t nm_lo = -1, nm_hi = -1;
            for (size_t i = 0; i < r.row_from.size(); ++i) {
                rows.push_back(Row{r.row_from[i], std::vector<uint64_t>(r.values.begin() + r.row_ptr[i], r.values.begin() + r.row_ptr[i + 1])});
                // the non-mapped counts are stored with another shape than the mapped ones: only the rows that hold something, trimmed to what they hold
                const uint64_t *nm = r.non_mapped.data() + r.row_ptr[i];
                const size_t len = r.row_ptr[i + 1] - r.row_ptr[i];
                size_t lo = 0, hi = len;
                while (lo < len && !nm[lo]) ++lo;
                while (hi > lo && !nm[hi - 1]) --hi;
                if (lo < hi) {
                    if (nm_lo < 0) nm_lo = (int64_t)i;
                    nm_hi = (int64_t)i + 1;
                }
            }
            set_vect2(member(st, "read_lengths_by_fragment_length_").kids[seg], r.from, rows);
            if (nm_lo >= 0) {
                for (int64_t i = nm_lo; i < nm_hi; ++i) {
                    const uint64_t *nm = r.non_mapped.data() + r.row_ptr[i];
                    const size_t len = r.row_ptr[i + 1] - r.row_ptr[i];
                    size_t lo = 0, hi = len;
                    while (lo < len && !nm[lo]) ++lo;
                    while (hi > lo && !nm[hi - 1]) --hi;
                    nm_rows.push_back(lo < hi ? Row{r.row_from[i] + lo, std::vector<uint64_t>(nm + lo, nm + hi)} : Row{});
                }
                set_vect2(member(st, "non_mapped_read_lengths_by_fragment_length_").kids[seg], r.from + (uint64_t)nm_lo, nm_rows);
            }
        }
        Node &tiles = member(st, "tiles_");
        member(tiles, "tiles_").u.assign(p.tiles.begin(), p.tiles.end());
        member(tiles, "abundance_").u = p.tile_abundance;

        Node &ad = member(st, "adapters_");
        std::vector<std::string> seqs[2];
        for (int seg = 0; seg < 2; ++seg) {
            const HostAdapters &a = p.adapters[seg];
            Node &names = member(ad, "names_").kids[seg], &archive_seqs = member(ad, "seqs_archive").kids[seg], &cuts = member(ad, "start_cut_").kids[seg];
            for (uint32_t i = 0; i < a.n(); ++i) {
                std::string sq;
                for (uint32_t k = a.seq_ptr[i]; k < a.seq_ptr[i + 1]; ++k) sq += "ACGT"[a.seqs[k] & 3u];
                seqs[seg].push_back(sq);
                Node name = default_node(names.type->elem), text = default_node(archive_seqs.type->elem), cut = default_node(cuts.type->elem);
                name.s = "adapter " + std::to_string(seg) + "/" + std::to_string(i);
                text.s = sq;
                set_vect(cut, a.cut_from[i], a.cut.data() + a.cut_ptr[i], a.cut_ptr[i + 1] - a.cut_ptr[i]);
                names.kids.push_back(std::move(name));
                archive_seqs.kids.push_back(std::move(text));
                cuts.kids.push_back(std::move(cut));
            }
        }
        // the detections of adapter pairs, at the adapters' full lengths (beyond every prefix shared with a neighbour: SumCounts counts them), spread so that the rows'
        // sums are the first adapters' counts and the columns' sums the second adapters': from the top left corner on
        {
            const uint32_t n1 = p.adapters[0].n(), n2 = p.adapters[1].n();
            std::vector<uint64_t> left1 = p.adapters[0].counts, left2 = p.adapters[1].counts;
            uint64_t s1 = 0, s2 = 0;
            for (uint64_t c : left1) s1 += c;
            for (uint64_t c : left2) s2 += c;
            if (s1 != s2)                                         // ReSeq stores detections of adapter PAIRS: row sums and column sums of one matrix have one total
                throw Error("this profile cannot be written as ReSeq's archives: the adapter counts of its two read segments sum to " + std::to_string(s1) + " and " + std::to_string(s2) +
                            " detections, and ReSeq's file holds one matrix of adapter pairs whose row and column sums they have to be (an RSQP container, rsq_profile_save, holds any counts)");
            Node &counts = member(ad, "counts_"), &comb = member(ad, "combinations_");
            uint32_t a2 = 0;
            std::vector<std::vector<uint64_t>> cell(n1, std::vector<uint64_t>(n2, 0));
            for (uint32_t a1 = 0; a1 < n1; ++a1)
                while (left1[a1] && a2 < n2) {
                    const uint64_t take = std::min(left1[a1], left2[a2]);
                    cell[a1][a2] += take;
                    left1[a1] -= take;
                    left2[a2] -= take;
                    if (!left2[a2]) ++a2;
                }
            for (uint32_t a1 = 0; a1 < n1; ++a1) {
                Node row = default_node(counts.type->elem), flags = default_node(comb.type->elem);
                for (uint32_t b2 = 0; b2 < n2; ++b2) {
                    Node c = default_node(row.type->elem);
                    if (cell[a1][b2]) set_vect2(c, seqs[0][a1].size(), {Row{seqs[1][b2].size(), {cell[a1][b2]}}});
                    row.kids.push_back(std::move(c));
                    flags.u.push_back(a1 == b2 ? 1 : 0);
                }
                counts.kids.push_back(std::move(row));
                comb.kids.push_back(std::move(flags));
            }
        }
        set_vect(member(ad, "polya_tail_length_"), p.polya);
        member(ad, "overrun_bases_").u.assign(p.overrun_bases, p.overrun_bases + 5);
        ArchiveWriter w(types.s.size(), 17);
        w.put(types.data_stats, st);
        w.write(stats_path);
    }
    {
        Node pe = default_node(types.probability_estimates);
        set_uint(member(pe, "stats_creation_time_"), creation_time);
        auto need = [&](const std::vector<HostTable> &v, size_t n, const char *what) {
            if (v.size() != n) throw Error(std::string("profile with ") + std::to_string(v.size()) + " " + what + " tables, " + std::to_string(n) + " expected");
        };
        need(p.quality, 8u * nt, "quality");
        need(p.seq_quality, 2u * nt, "sequence quality");
        need(p.base_call, 40u * nt, "base call");
        need(p.dom_error, 100, "dominant error");
        need(p.error_rate, 20, "error rate");
        need(p.indels, 12, "indel");
        for (uint32_t seg = 0; seg < 2; ++seg) {
            Node &q = member(pe, "quality_").kids[seg], &sq = member(pe, "sequence_quality_").kids[seg], &bc = member(pe, "base_call_").kids[seg];
            for (uint32_t tile = 0; tile < nt; ++tile) {
                Node qt = default_node(q.type->elem), st = default_node(sq.type->elem), bt = default_node(bc.type->elem);
                set_ipf(st, p.seq_quality[seg * nt + tile]);
                for (uint32_t base = 0; base < 4; ++base) {
                    set_ipf(qt.kids[base], p.quality[(seg * nt + tile) * 4u + base]);
                    for (uint32_t dom = 0; dom < 5; ++dom) set_ipf(bt.kids[base].kids[dom], p.base_call[((seg * nt + tile) * 4u + base) * 5u + dom]);
                }
                q.kids.push_back(std::move(qt));
                sq.kids.push_back(std::move(st));
                bc.kids.push_back(std::move(bt));
            }
        }
        for (uint32_t base = 0; base < 4; ++base) {
            for (uint32_t prev = 0; prev < 5; ++prev)
                for (uint32_t dom = 0; dom < 5; ++dom) set_ipf(member(pe, "dom_error_").kids[base].kids[prev].kids[dom], p.dom_error[(base * 5u + prev) * 5u + dom]);
            for (uint32_t dom = 0; dom < 5; ++dom) set_ipf(member(pe, "error_rate_").kids[base].kids[dom], p.error_rate[base * 5u + dom]);
        }
        for (uint32_t type = 0; type < 2; ++type)
            for (uint32_t call = 0; call < 6; ++call) set_ipf(member(pe, "indels_").kids[type].kids[call], p.indels[type * 6u + call]);
        ArchiveWriter w(types.s.size(), 17);
        w.put(types.probability_estimates, pe);
        w.write(ipf_path);
    }
}

// `ipf_path` empty: "<stats_path>.ipf" (main.cpp:837).  precision_aim as a fraction (--ipfPrecision is in percent, main.cpp:733).
Profile Profile::load_archives(const std::string &stats_path, const std::string &ipf_path_in, double precision_aim, std::string *warnings) {
    static ReseqTypes types;   // immutable after construction
    const std::string ipf_path = ipf_path_in.empty() ? stats_path + ".ipf" : ipf_path_in;
    Profile p;
    uint64_t creation_time = 0;
    uint32_t versions[2] = {0, 0};
    std::string grammar_note;
    // one file under the recalled token rules, then under the alternatives (archive::Grammar): the first that reads the file to its end with every fixed size in
    // place is taken; if none does, the recalled rules' message (member path, type, class-info sites) is the error
    auto parse = [&](const std::string &path, const char *root, archive::TypeP type, uint32_t &version, Node &out) {
        const std::vector<char> buf = slurp(path);
        std::string first_error;
        for (const archive::Grammar &g : archive::Grammar::alternatives()) {
            try {
                archive::Reader r(buf.data(), buf.data() + buf.size(), types.s.size(), path, g);
                r.set_root(root);
                version = r.library_version();
                Node n;
                r.read(type, &n);
                r.expect_end();
                out = std::move(n);
                if (!g.is_default()) grammar_note += path + " does not follow the recalled token rules but parses as: " + g.name() + ". ";
                return;
            } catch (const std::exception &e) {
                if (first_error.empty()) first_error = e.what();
            }
        }
        throw Error(first_error + "; no alternative of the doubtful token rules (" + std::to_string(archive::Grammar::alternatives().size()) + " combinations, rsq_archive.h Grammar) reads the file either");
    };
    {
        Node st;
        parse(stats_path, "DataStats", types.data_stats, versions[0], st);
        creation_time = st["creation_time_"].uint();
        fill_from_stats(p, st);
    }
    if (p.total_number_reads == 0) throw Error("the statistics in " + stats_path + " hold no reads");   // main.cpp:830-832
    std::string warn;
    {
        Node pe;
        parse(ipf_path, "ProbabilityEstimates", types.probability_estimates, versions[1], pe);
        if (pe["stats_creation_time_"].uint() != creation_time)   // ProbabilityEstimates.cpp:1079-1083: ReSeq would refit from scratch
            throw Error(ipf_path + " was fitted to another statistics file than " + stats_path + " (creation times differ); fitting is not part of this build");
        fill_from_estimates(p, pe, precision_aim, warn);
    }
    warn += grammar_note;
    // said every time: the token rules of the reader could not be checked against a file written by Boost itself (INTEGRATION.md "Profile files")
    warn += "read as Boost text archives of library version " + std::to_string(versions[0]) + " (" + stats_path + ") and " + std::to_string(versions[1]) + " (" + ipf_path +
            "); if tables look wrong, send the output of `reseq queryProfile --dumpArchiveLayout -s " + stats_path + "`. ";
    if (warnings) *warnings = warn;
    return p;
}

// Where every class type's information sits in the two files (archive::ClassInfoSite), and if a file does not parse, the reader's message with the member
// path: what a maintainer needs to tell which token rule of rsq_archive.h a real ReSeq profile contradicts.
std::string Profile::archive_layout(const std::string &stats_path, const std::string &ipf_path_in) {
    static ReseqTypes types;
    const std::string ipf_path = ipf_path_in.empty() ? stats_path + ".ipf" : ipf_path_in;
    std::string out;
    auto one = [&](const std::string &path, const char *root, archive::TypeP type) {
        out += "# " + path + " (" + root + ")\n";
        std::vector<char> buf;
        try {
            buf = slurp(path);
        } catch (const std::exception &e) {
            out += std::string("error\t") + e.what() + "\n";
            return;
        }
        out += "bytes\t" + std::to_string(buf.size()) + "\n";
        std::unique_ptr<archive::Reader> r, first;
        std::string error;
        for (const archive::Grammar &g : archive::Grammar::alternatives()) {          // the recalled token rules, then the alternatives; the first that fits is shown
            std::string e_this;
            try {
                r.reset(new archive::Reader(buf.data(), buf.data() + buf.size(), types.s.size(), path, g));
                r->set_root(root);
                r->read(type, nullptr);
                r->expect_end();
            } catch (const std::exception &e) {
                e_this = e.what();
            }
            if (e_this.empty()) {
                error.clear();
                out += "library_version\t" + std::to_string(r->library_version()) + "\n";
                out += std::string("token_rules\t") + (g.is_default() ? "as recalled: " : "NOT as recalled: ") + g.name() + "\n";
                break;
            }
            if (!first) {                                                              // no grammar fits: the recalled rules' sites and message are what is shown
                first = std::move(r);
                error = e_this;
            }
            r.reset();
        }
        if (!r) {
            r = std::move(first);
            if (r) out += "library_version\t" + std::to_string(r->library_version()) + "\n";
        }
        out += "byte\ttracking\tversion\ttype\tfirst object\n";
        if (r)
            for (const archive::ClassInfoSite &c : r->class_info_sites())
                out += std::to_string(c.byte) + "\t" + std::to_string(c.tracking) + "\t" + std::to_string(c.version) + "\t" + c.type + "\t" + c.path + "\n";
        out += error.empty() ? "parsed\tto the end\n" : "error\t" + error + "\n";
    };
    one(stats_path, "DataStats", types.data_stats);
    one(ipf_path, "ProbabilityEstimates", types.probability_estimates);
    return out;
}

// ---------------------------------------------------------------------------------------------- RSQP writer
namespace {
struct ContainerWriter {
    std::vector<uint8_t> body;
    uint32_t n = 0;
    void pad8() {
        while (body.size() % 8) body.push_back(0);
    }
    void raw(const void *p, size_t bytes) {
        const uint8_t *b = (const uint8_t *)p;
        body.insert(body.end(), b, b + bytes);
    }
    template <class T>
    void put(const std::string &name, int dtype, const T *data, size_t count, const std::vector<uint64_t> &dims) {
        const uint16_t len = (uint16_t)name.size();
        raw(&len, 2);
        raw(name.data(), len);
        const uint8_t head[2] = {(uint8_t)dtype, (uint8_t)dims.size()};
        raw(head, 2);
        pad8();
        raw(dims.data(), 8 * dims.size());
        raw(data, sizeof(T) * count);
        pad8();
        ++n;
    }
    template <class T>
    void vec(const std::string &name, int dtype, const std::vector<T> &v) {
        put(name, dtype, v.data(), v.size(), {(uint64_t)v.size()});
    }
    template <class T>
    void scalar(const std::string &name, int dtype, T v) {
        put(name, dtype, &v, 1, {1});
    }
    template <class T>
    void vect(const std::string &name, int dtype, const Vect<T> &v) {
        vec(name, dtype, v.v);
        scalar<uint64_t>(name + ".from", 3, v.from);
    }
    void table(const std::string &prefix, const HostTable &t) {
        vec("tab." + prefix + ".par0", 2, t.par0);
        std::vector<uint32_t> lim;
        std::vector<double> d;
        for (uint32_t m = 0; m < t.nm; ++m) {
            lim.push_back(t.from[m]);
            lim.push_back(t.to[m]);
            d.insert(d.end(), t.dim2[m].begin(), t.dim2[m].end());
        }
        put("tab." + prefix + ".limits", 2, lim.data(), lim.size(), {(uint64_t)t.nm, 2});
        vec("tab." + prefix + ".dim2", 6, d);
    }
};
}  // namespace

void Profile::save(const std::string &path) const {
    ContainerWriter w;
    w.scalar<uint8_t>("phred_quality_offset", 0, phred_offset);
    w.scalar<double>("corrected_coverage", 6, corrected_coverage);
    w.scalar<uint16_t>("errors.max_len_deletion", 1, max_len_deletion);
    w.scalar<uint32_t>("coverage.reset_distance", 2, reset_distance);
    w.vect("frag.insert_lengths", 3, insert_lengths);
    w.vect("frag.insert_lengths_bias", 6, insert_lengths_bias);
    w.vect("frag.gc_bias", 6, gc_bias);
    w.vec("frag.sur_bias", 6, sur_bias);
    w.put("frag.dispersion_parameters", 6, dispersion, 2, {2});
    w.vec("frag.ref_seq_bias", 6, ref_seq_bias);
    for (int seg = 0; seg < 2; ++seg) {
        const std::string s = std::to_string(seg);
        w.vect("read_lengths." + s, 3, read_lengths[seg]);
        w.scalar<uint64_t>("rl_by_fl." + s + ".from", 3, rl_by_fl[seg].from);
        w.vec("rl_by_fl." + s + ".row_ptr", 2, rl_by_fl[seg].row_ptr);
        w.vec("rl_by_fl." + s + ".row_from", 2, rl_by_fl[seg].row_from);
        w.vec("rl_by_fl." + s + ".values", 3, rl_by_fl[seg].values);
        w.vec("rl_by_fl_nonmapped." + s + ".values", 3, rl_by_fl[seg].non_mapped);
    }
    w.vec("tiles.tiles", 1, tiles);
    w.vec("tiles.abundance", 3, tile_abundance);
    for (int seg = 0; seg < 2; ++seg) {
        const std::string s = std::to_string(seg);
        const HostAdapters &a = adapters[seg];
        w.vec("adapters." + s + ".seqs", 0, a.seqs);
        w.vec("adapters." + s + ".seq_ptr", 2, a.seq_ptr);
        w.vec("adapters." + s + ".counts", 3, a.counts);
        w.vec("adapters." + s + ".significant_counts", 3, a.significant);
        w.vec("adapters." + s + ".start_cut_ptr", 2, a.cut_ptr);
        w.vec("adapters." + s + ".start_cut_from", 2, a.cut_from);
        w.vec("adapters." + s + ".start_cut", 3, a.cut);
    }
    w.vect("adapters.polya_tail_length", 3, polya);
    w.put("adapters.overrun_bases", 3, overrun_bases, 5, {5});
    const uint32_t nt = n_tiles();
    size_t iq = 0, isq = 0, ibc = 0, ide = 0, ier = 0, iin = 0;
    for (uint32_t seg = 0; seg < 2; ++seg)
        for (uint32_t tile = 0; tile < nt; ++tile) {
            const std::string st = std::to_string(seg) + "." + std::to_string(tile);
            w.table("seq_quality." + st, seq_quality.at(isq++));
            for (uint32_t base = 0; base < 4; ++base) {
                w.table("quality." + st + "." + std::to_string(base), quality.at(iq++));
                for (uint32_t dom = 0; dom < 5; ++dom) w.table("base_call." + st + "." + std::to_string(base) + "." + std::to_string(dom), base_call.at(ibc++));
            }
        }
    for (uint32_t base = 0; base < 4; ++base)
        for (uint32_t x = 0; x < 5; ++x)
            for (uint32_t y = 0; y < 5; ++y) w.table("dom_error." + std::to_string(base) + "." + std::to_string(x) + "." + std::to_string(y), dom_error.at(ide++));
    for (uint32_t base = 0; base < 4; ++base)
        for (uint32_t x = 0; x < 5; ++x) w.table("error_rate." + std::to_string(base) + "." + std::to_string(x), error_rate.at(ier++));
    for (uint32_t type = 0; type < 2; ++type)
        for (uint32_t call = 0; call < 6; ++call) w.table("indels." + std::to_string(type) + "." + std::to_string(call), indels.at(iin++));

    std::ofstream f(path, std::ios::binary);
    if (!f) throw Error("Could not open " + path + " for writing.");
    const uint32_t head[2] = {1, w.n};
    f.write("RSQPROF1", 8);
    f.write((const char *)head, 8);
    f.write((const char *)w.body.data(), (std::streamsize)w.body.size());
    if (!f) throw Error("Could not write " + path);
}

}  // namespace rsq
