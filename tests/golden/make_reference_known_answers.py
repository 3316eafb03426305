#!/usr/bin/env python3
"""Known answers held by the reference's own gtest suites for the simulation path.

Each entry is DATA (inputs + expected outputs) transcribed from the cited test of
/root/reference (schmeing/ReSeq v1.1); `reference-test.fa` next to this file is the
data file those tests read (reference/test/reference-test.fa).  Running this script
rewrites reference_known_answers.json; tests/test_oracle_pinning.py checks the
oracle against it.
"""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))

ka = {}

# reseq/SurroundingTest.cpp:42-120  (Forward/Reverse surroundings of test/reference-test.fa)
ka["surrounding_forward"] = [     # [seq, pos, [block0, block1, block2]]
    [0, 0, [83, 163795, 909796]],
    [0, 1, [332, 655183, 493456]],
    [0, 2, [1330, 523581, 925249]],
    [0, 15, [1003384, 495722, 275383]],
    [0, 497, [1015809, 313855, 325511]],
    [0, 498, [917509, 206845, 253470]],
    [0, 499, [524308, 827380, 1013881]],
    [1, 17, [500825, 144533, 54422]],
]
# SurroundingTest.cpp:69-97: Forward(pos0) followed by UpdateForward(new positions) -> expected after each update
ka["surrounding_update_forward"] = [   # [seq, set_pos, [update positions], [expected per update]]
    [0, 0, [1, 2], [[332, 655183, 493456], [1330, 523581, 925249]]],
    [0, 14, [15], [[1003384, 495722, 275383]]],
    [0, 496, [497, 498, 499], [[1015809, 313855, 325511], [917509, 206845, 253470], [524308, 827380, 1013881]]],
    [1, 16, [17], [[500825, 144533, 54422]]],
]
# SurroundingTest.cpp:99-148
ka["surrounding_reverse"] = [
    [0, 0, [57353, 846847, 855350]],
    [0, 1, [538626, 473855, 1000269]],
    [0, 2, [134656, 642751, 1036499]],
    [0, 15, [613348, 739384, 10042]],
    [0, 497, [524915, 720884, 216468]],
    [0, 498, [917660, 966653, 54117]],
    [0, 499, [229415, 241663, 275673]],
    [1, 17, [960397, 945044, 626906]],
]
# SurroundingTest.cpp:150-198 (UpdateReverse parts only; RollBackReverse is not on the simulation path)
ka["surrounding_update_reverse"] = [
    [0, 0, [1, 2], [[538626, 473855, 1000269], [134656, 642751, 1036499]]],
    [0, 14, [15], [[613348, 739384, 10042]]],
    [0, 496, [497, 498, 499], [[524915, 720884, 216468], [917660, 966653, 54117], [229415, 241663, 275673]]],
]

# SurroundingTest.cpp:541-568 TestCombiningBias
sep = [0.0] * 120
for block in (0, 40, 80):
    sep[block + 0] = 0.9
    for i in (5, 10, 15, 19, 22, 25, 28, 35, 36):
        sep[block + i] = 1.0
    sep[block + 21] = 0.8
ka["combine_positions"] = {"separated": sep, "expect": [[114252, 9.9], [113996, 9.7]]}   # same for all three blocks

# SurroundingTest.cpp:570-640 TestSeparatingBias
comb = [
    [0, 954112, 1500], [0, 771557, 192], [0, 756483, 3200], [0, 594450, 2000], [0, 258785, 192],
    [1, 8941, 1500], [1, 756483, 192], [1, 787452, 6400], [1, 196668, 2000], [1, 461635, 192],
    [2, 930377, 1500], [2, 787452, 192], [2, 1017802, 3200], [2, 297745, 2000], [2, 4080, 192],
    [0, 649012, 1500], [0, 787452, 192], [0, 377552, 3200], [0, 766430, 2000], [0, 983295, 192],
    [1, 542591, 1500], [1, 258513, 192], [1, 802803, 2000], [1, 254450, 192],
    [2, 1044692, 1500], [2, 674497, 192], [2, 258513, 3200], [2, 505785, 2000], [2, 739075, 192],
]
norm = (1 << 20) // 4
exp = {}
mean = (192 + 3200 + 192 + 3200 + 2000 + 1500 + 2000 + 1500 + 192 + 192) / 4.0
exp[0] = (192 - mean) / norm
exp[1] = (3200 - mean) / norm
exp[2] = (192 + 3200 + 2000 + 1500 + 2000 - mean) / norm
exp[3] = (1500 + 192 + 192 - mean) / norm
mean = (1500 + 1500 + 192 + 3200 + 192 + 192 + 2000 + 2000 + 3200 + 192) / 4.0
exp[36] = (1500 + 1500 + 192 + 3200 - mean) / norm
exp[37] = (192 + 192 - mean) / norm
exp[38] = (2000 + 2000 - mean) / norm
exp[39] = (3200 + 192 - mean) / norm
mean = (1500 + 2000 + 192 + 192 + 192 + 192 + 1500 + 2 * 3200 + 2000) / 4.0
exp[40] = (1500 + 2000 + 192 + 192 - mean) / norm
exp[41] = (192 - mean) / norm
exp[42] = (192 + 1500 - mean) / norm
exp[43] = (2 * 3200 + 2000 - mean) / norm
mean = (2 * 3200 + 2000 + 1500 + 192 + 192 + 192 + 192 + 1500 + 2000) / 4.0
exp[76] = (2 * 3200 + 2000 - mean) / norm
exp[77] = (1500 + 192 - mean) / norm
exp[78] = (192 - mean) / norm
exp[79] = (192 + 192 + 1500 + 2000 - mean) / norm
mean = (192 + 3200 + 2000 + 2000 + 192 + 192 + 1500 + 192 + 3200 + 1500) / 4.0
exp[80] = (192 + 3200 - mean) / norm
exp[81] = (2000 + 2000 - mean) / norm
exp[82] = (192 + 192 - mean) / norm
exp[83] = (1500 + 192 + 3200 + 1500 - mean) / norm
mean = (192 + 192 + 1500 + 1500 + 2000 + 192 + 3200 + 2000 + 3200 + 192) / 4.0
exp[116] = (192 + 192 + 1500 - mean) / norm
exp[117] = (1500 + 2000 + 192 + 3200 + 2000 - mean) / norm
exp[118] = (3200 - mean) / norm
exp[119] = (192 - mean) / norm
ka["separate_positions"] = {"combined": comb, "expect": [[k, v] for k, v in sorted(exp.items())]}

# FragmentDistributionStatsTest.cpp:783-795  DrawNumberNonZeroStrands(num_possible_alleles, zero_probability, u)
ka["draw_number_non_zero_strands"] = [
    [1, 0.75, 0.5624, 0], [1, 0.75, 0.5626, 1], [1, 0.75, 0.9374, 1], [1, 0.75, 0.9376, 2], [1, 0.75, 1.0 - 1e-15, 2],
    [2, 0.75, 0.31640, 0], [2, 0.75, 0.31641, 1], [2, 0.75, 0.99609, 3], [2, 0.75, 0.99610, 4], [2, 0.75, 1.0 - 1e-15, 4],
]

# FragmentDistributionStatsTest.cpp:797-849 + :208-228 CheckDrawnCounts
ka["fragment_counts"] = {
    "ref_seq_bias": 0.5, "insert_length": 367, "insert_length_bias": 0.5, "gc": 43,
    "start_sur": [873425, 34, 7467], "end_sur": [364, 856687, 34562],
    "sur_fill": -1000.0,
    "sur_entries": [[0, 873425, -0.3], [1, 34, -0.7], [2, 7467, -math.log(3) + 1.0],
                    [0, 364, -math.log(3) + 1.0], [1, 856687, -0.7], [2, 34562, -0.3]],
    "bias_normalization": 0.2, "other_bias_negation": 2.0 * 2.0 * 2.0 * 2.0 * 5.0, "delta": 0.000001,
    "cases": [   # dispersion parameters, bias, CDF gates (R ppois / pnbinom)
        [[0.0, 1e-100], 1.0, [0.3678794, 0.7357589, 0.9196986, 0.9810118, 0.9963402, 0.9994058]],
        [[0.0, 1e-100], 0.5, [0.6065307, 0.9097960, 0.9856123, 0.9982484, 0.9998279, 0.9999858]],
        [[0.0, 1e-100], 0.1, [0.9048374, 0.9953212, 0.9998453, 0.9999962]],
        [[0.0, 5.0], 1.0, [0.6988271, 0.8152983, 0.8735339, 0.9091223, 0.9328479, 0.9494559]],
        [[0.0, 5.0], 0.5, [0.7783705, 0.8895663, 0.9372217, 0.9621840, 0.9764482, 0.9850067]],
        [[0.0, 5.0], 0.1, [0.9221079, 0.9835818, 0.9958765, 0.9988819, 0.9996834, 0.9999078]],
        [[5000.0, 10000.0], 1.0, [0.9993591, 0.9994258, 0.9994591, 0.9994813, 0.9994979, 0.9995113]],
        [[5000.0, 10000.0], 0.5, [0.9995396, 0.9995896, 0.9996145, 0.9996312, 0.9996437, 0.9996537]],
        [[5000.0, 10000.0], 0.1, [0.9998550, 0.9998717, 0.9998800, 0.9998856, 0.9998897, 0.9998931]],
    ],
}
# SURVEY.md section 8(c) probe values of the reference binary (NegativeBinomial, Binomial, GetDispersion)
ka["survey_probe"] = {"negative_binomial": [0.3, 2.5, 0.9, 3], "binomial": [4, 0.25, 0.8, 2],
                      "get_dispersion": [1.7, 0.1, 0.2, 3.8636363636363629]}

# SimulatorTest.cpp:63-85 TestCoverageConversion
ka["coverage_conversion"] = {
    "rl_by_fl": [[1, 200, 150, 10], [0, 150, 150, 10], [1, 100, 150, 5], [1, 50, 150, 10], [0, 0, 150, 10]],   # seg, frag_len, read_len, count
    "non_mapped": [[1, 50, 150, 5]],
    "adapter_part": 3000.0 / (45 * 150),
    "coverage": 100.0, "total_ref_size": 50000, "average_read_length": 150.0, "total_pairs": 30000,
}
# SimulatorTest.cpp:87-114 TestSelectAllele: repeated SelectAllele(..., possible_strands, 0.5)
ka["select_allele"] = [[2, 0.5, [1, 0]], [4, 0.5, [2, 1, 3, 0]]]

# utilitiesTest.cpp:30-60,96-110 DominantBase on "CAGATTTTGGAANAGTNN" and its reverse complement
ka["dominant_base"] = {
    "seq": "CAGATTTTGGAANAGTNN",
    "set_0_1_2": [1, 1, 0],
    "set_from_3": [2, 0, 0, 3, 3, 3, 3, 3, 2, 0, 0, 0, 0, 0, 3],
    "revcomp_set_0_1_2": [0, 0, 0],
    "revcomp_set_from_3": [0, 1, 3, 3, 3, 3, 3, 1, 1, 0, 0, 0, 0, 0, 3],
}
# utilitiesTest.cpp:152-160,185-192
ka["divide"] = [[0, 2, 0], [17438564308265206, 17438564308265206, 1], [73500, 7000, 11], [73400, 7000, 10]]
ka["percent"] = [[0, 2, 0], [17438564308265206, 17438564308265206, 100], [735, 7000, 11], [734, 7000, 10]]
ka["safe_percent_zero_den"] = [734, 0, 50]

# ReferenceTest.cpp:275-284 (sequence lengths, ReferenceSequence forward / reversed, no variants)
ka["reference_sequence"] = {"lengths": [500, 501],
                            "cases": [[0, 0, 10, False, "AGCTTTTCAT"], [0, 500, 10, True, "ATGGTTTTTT"]]}
# ReferenceTest.cpp:339-345
ka["gc_content"] = [[0, 0, 7, 2, 29], [1, 22, 27, 4, 80]]      # seq, start, end, absolute, percent
# ReferenceTest.cpp:367-443 TestSumBias: gc_bias by percent, surrounding bias entries (everything else -1000)
ka["sum_bias"] = {
    "seq": 0, "fragment_length": 10, "general_bias": 0.5,
    "gc_bias": [[30, 0.1], [40, 0.3], [0, 1.0], [90, 0.2]],
    "sur_fill": -1000.0,
    "sur_entries": [
        [0, 591524, 0.0], [0, 954112, 0.0], [0, 771557, 0.0], [0, 756483, 0.0], [0, 594450, -1.0], [0, 332464, 0.0], [0, 258785, 0.5],
        [1, 211831, 0.0], [1, 8941, 0.0], [1, 756483, 0.0], [1, 787452, 0.5], [1, 196668, 0.0], [1, 440745, 0.0], [1, 461635, 0.0],
        [2, 768572, 0.0], [2, 930377, 0.0], [2, 787452, 0.5], [2, 1017802, 0.0], [2, 297745, 0.0], [2, 924081, 0.0], [2, 4080, -0.5],
        [0, 800017, 0.0], [0, 649012, 0.0], [0, 787452, 0.5], [0, 377552, 0.0], [0, 766430, 0.0], [0, 727476, 0.0], [0, 983295, 0.0],
        [1, 139571, 0.0], [1, 542591, 0.0], [1, 258513, 0.0], [1, 802803, 0.0], [1, 612630, 0.0], [1, 254450, 0.0],
        [2, 939769, 0.0], [2, 1044692, 0.0], [2, 674497, -0.5], [2, 258513, 0.0], [2, 505785, 0.0], [2, 989114, 0.0], [2, 739075, 0.5],
    ],
    "twice_sum": 2.9367, "twice_sum_tol": 0.0001, "max_bias": 0.774915, "max_bias_tol": 0.00001,
}

# ReferenceTest.cpp:526-593 TestReplaceN on test/reference-test.fa: stretches of 100 N get the four-base repeat of their flanks
# (two bases after, two before), the 50-N stretch and the single N are drawn
ka["replace_n"] = {
    "set_n": [[0, 50, 150], [1, 350, 450], [1, 54, 104], [0, 253, 254], [0, 256, 257], [0, 263, 264]],      # [seq, from, to)
    "n_in_reference": 253,
    "expected": [[0, 50, 2], [0, 51, 1], [0, 52, 0], [0, 53, 0], [0, 54, 2], [0, 55, 1], [0, 56, 0], [0, 57, 0], [0, 146, 2], [0, 147, 1], [0, 148, 0], [0, 149, 0],
                 [1, 350, 1], [1, 351, 0], [1, 352, 1], [1, 353, 2], [1, 354, 1], [1, 355, 0], [1, 356, 1], [1, 357, 2], [1, 446, 1], [1, 447, 0], [1, 448, 1], [1, 449, 2]],
}
# ReferenceTest.cpp:707-768 TestMethylationLoading on test/drosophila-methylation.bed (copied next to this file), two alleles:
# the sequence at index 2 is NW_007931112.1, the one at index 9 NW_007931119.1; every other sequence has no entries
ka["methylation_loading"] = {
    "num_alleles": 2, "sequence_index": {"NW_007931112.1": 2, "NW_007931119.1": 9}, "n_sequences": 21,
    "regions": {"2": [[0, 100], [100, 101], [34520, 34521]], "9": [[5000, 35000]]},
    "unmethylation": {"2": [[0.75, 0.7, 0.6], [0.74, 0.69, 0.59]], "9": [[0.9], [0.9]]},      # [allele][region]; a single column serves every allele
    "empty": [1, 10, 20],
}

# ReferenceTest.cpp:30-84 TestVariantClass: allele bit blocks -> InAllele of alleles (0, 1, 2, 64), FirstAllele
ka["variant_class"] = [   # [allele_[0], allele_[1], {allele: InAllele}, FirstAllele]
    [7, 0, {"0": True, "1": True, "2": True}, 0], [6, 0, {"0": False, "1": True, "2": True}, 1], [5, 0, {"0": True, "1": False, "2": True}, 0],
    [4, 0, {"0": False, "1": False, "2": True}, 2], [3, 0, {"0": True, "1": True, "2": False}, 0], [2, 0, {"0": False, "1": True, "2": False}, 1],
    [1, 0, {"0": True, "1": False, "2": False}, 0], [1, 1, {"0": True, "1": False, "2": False, "64": True}, 0], [0, 1, {"0": False, "64": True}, 64],
]
# ReferenceTest.cpp:86-136 TestInsertVariant: calls (position, var_seq, allele_[0]) in order -> resulting list
ka["insert_variant"] = {
    "calls": [[0, "A", 1], [1, "A", 2], [1, "ACT", 1], [1, "C", 4], [1, "", 8], [1, "C", 16], [1, "", 32], [1, "ACT", 64], [1, "TG", 128]],
    "expected": [[0, "A", 1], [1, "", 40], [1, "A", 2], [1, "C", 20], [1, "TG", 128], [1, "ACT", 65]],
}
# ReferenceTest.cpp:138-239 TestVariationLoading on test/test-var.vcf (copied next to this file; E. coli NC_000913.3, 4 641 652 bases):
# two alleles, 13 single-position variants [position, var_seq, InAllele(0), InAllele(1)].  ref_alleles are the VCF's own REF columns
# (0-based position, bases): the reference sequence has to carry them for the consistency check of ReadVariants (Reference.cpp:188).
ka["variation_loading"] = {
    "contig": "NC_000913.3", "length": 4641652, "num_alleles": 2,
    "ref_alleles": [[0, "AGCTTTTCA"], [16, "T"], [20, "A"], [11368, "CTA"], [953165, "A"], [3192437, "TT"], [3424235, "T"], [3424236, "C"]],
    "variants": [[2, "T", False, True], [3, "A", False, True], [4, "A", False, True], [5, "G", False, True], [7, "A", False, True], [8, "TTTTTCAGCTTTTCA", False, True],
                 [11368, "T", False, True], [11370, "T", False, True], [953165, "G", False, True], [3192437, "G", False, True], [3192438, "", False, True],
                 [3424235, "C", True, True], [3424236, "A", True, True]],
}

# SimulatorTest.cpp:116-364 TestVariationInSimulateFromGivenBlock: the per-allele modifiers while SimulateFromGivenBlock walks start
# positions 1003..1011 of E. coli (bases 1001-1020: GTTGCGAGATTTGGACGGAC, quoted in the test) with six variants on allele 1.
# inner: arguments of TestVariationInInnerLoopOfSimulateFromGivenBlock (from, to, valid alleles, then per allele the expected
# unhandled_variant_id / unhandled_bases_in_variant / gc_mod / end_pos_shift per fragment length, modified start positions, which
# sequence the results are compared with: "ref", "alt" (the sequence with the variants applied) or null); after: first_variant_id_ and
# start_variant_pos_ after CheckForInsertedBasesToStartFrom.
ka["variation_in_simulate_from_given_block"] = {
    "bases_1000_1019": "GTTGCGAGATTTGGACGGAC",
    "variants": [[455, "", 2], [1002, "TAC", 2], [1004, "TGA", 2], [1008, "", 2], [1011, "C", 2], [1012, "", 2]],
    "steps": [
        {"start": 1003, "set_first_variant": 2, "forward_surrounding_start": True, "alt_start_surrounding_at": 1004,
         "inner": [1, 13, 2, [[2, 3, 3, 3, 3, 4, 4, 4, 5, 6, 6, 6], [2, 2, 2, 3, 3, 3, 3, 4, 4, 5, 6, 6]], [[0] * 12, [0, 2, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]],
                   [[0] * 12, [0, -1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0]], [[0] * 12, [0, 0, -1, -2, -2, -2, -2, -1, -1, -1, 0, 0]], [1003, 1004], ["ref", "alt"]],
         "after": [2, 0]},
        {"start": 1004, "forward_surrounding_start": True, "alt_start_surrounding_at": 1005,
         "inner": [1, 12, 2, [[3, 3, 3, 3, 4, 4, 4, 5, 6, 6, 6], [2, 2, 3, 3, 3, 3, 4, 4, 5, 6, 6]], [[0] * 11, [2, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]],
                   [[0] * 11, [-1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0]], [[0] * 11, [0, -1, -2, -2, -2, -2, -1, -1, -1, 0, 0]], [1004, 1005], ["ref", "alt"]],
         "after": [2, 1]},
        {"start": 1004, "forward_surrounding_start": False, "alt_start_surrounding_at": 1006,
         "inner": [1, 11, 1, [[], [3, 3, 3, 3, 3, 4, 4, 5, 6, 6]], [[], [0] * 10], [[], [0, 0, 0, 0, 0, 0, 0, 1, 0, 0]], [[], [0, -1, -1, -1, -1, 0, 0, 0, 1, 1]],
                   [0, 1006], [None, "alt"]],
         "after": [2, 2]},
        {"start": 1004, "forward_surrounding_start": False, "alt_start_surrounding_at": 1007,
         "inner": [1, 10, 1, [[], [3, 3, 3, 3, 4, 4, 5, 6, 6]], [[], [0] * 9], [[], [-1, -1, -1, -1, -1, -1, 0, -1, -1]], [[], [0, 0, 0, 0, 1, 1, 1, 2, 2]],
                   [0, 1007], [None, "alt"]],
         "after": [3, 0]},
        {"start": 1008, "forward_surrounding_start": True, "alt_start_surrounding_at": None,
         "inner": [1, 8, 1, [[4, 4, 4, 5, 6, 6, 6], []], [[0] * 7, []], [[0] * 7, []], [[0] * 7, []], [1008, 0], ["ref", None]],
         "after": [4, 0]},
        {"start": 1011, "forward_surrounding_start": True, "alt_start_surrounding_at": 1013,
         "inner": [1, 5, 2, [[5, 6, 6, 6], [5, 6, 6, 6]], [[0] * 4, [0] * 4], [[0] * 4, [1, 0, 0, 0]], [[0] * 4, [0, 1, 1, 1]], [1011, 1013], ["ref", "alt"]],
         "after": [5, 0]},
    ],
}

# reseq/ReferenceTest.cpp:283-326 TestLoadingAndAccess: Reference::ReferenceSequence with variants on sequence 0 of test/reference-test.fa
# variants: [position, var_seq, allele bits]; calls: [start_pos, frag_length, reversed, first_variant id, position inside it, allele, expected]
ka["reference_sequence_with_variants"] = {
    "seq": 0,
    "plain": [[0, 10, False, "AGCTTTTCAT"], [500, 10, True, "ATGGTTTTTT"]],             # ReferenceTest.cpp:277-282 (no variants)
    "variants": [[2, "", 2], [4, "TAG", 3], [9, "C", 1]],
    "calls": [
        [0, 12, False, 0, 0, 0, "AGCTTAGTTCAC"],
        [0, 11, False, 0, 0, 1, "AGTTAGTTCAT"],
        [4, 8, False, 1, 0, 1, "TAGTTCAT"],
        [4, 7, False, 1, 1, 1, "AGTTCAT"],
        [4, 6, False, 1, 2, 1, "GTTCAT"],
        [10, 12, True, 2, 0, 0, "GTGAACTAAGCT"],
        [10, 11, True, 2, 0, 1, "ATGAACTAACT"],
        [5, 7, True, 1, 0, 0, "CTAAGCT"],
        [5, 6, True, 1, 2, 0, "TAAGCT"],
        [5, 5, True, 1, 1, 0, "AAGCT"],
    ],
    "added_variant": [499, "TAG", 3],                                                       # :309
    "calls_with_added_variant": [
        [500, 12, True, 3, 0, 0, "CTATGGTTTTTT"],
        # :313-325 "fragment length is shorter than variant length"
        [4, 1, False, 1, 0, 1, "T"], [4, 1, False, 1, 1, 1, "A"], [4, 1, False, 1, 2, 1, "G"],
        [5, 1, True, 1, 0, 0, "C"], [5, 1, True, 1, 2, 0, "T"], [5, 1, True, 1, 1, 0, "A"],
    ],
    # the same four variants as a VCF on reference-test.fa (the product reads variants from files only): 1-based POS, the deletion with its anchor base
    "vcf_records": [["NC_000913.3_1-500", 2, "GC", "G", "0|1"], ["NC_000913.3_1-500", 5, "T", "TAG", "1|1"], ["NC_000913.3_1-500", 10, "T", "C", "1|0"],
                    ["NC_000913.3_1-500", 500, "T", "TAG", "1|1"]],
}

# reseq/ReferenceTest.cpp:271-272,328-329: ReferenceId / ReferenceIdFirstPart of test/reference-test.fa (the first part goes into every read id, Simulator.cpp:596-632)
ka["reference_ids"] = {"full": ["NC_000913.3_1-500 bla", "NC_000913.3_10000-10500 blub"], "first_part": ["NC_000913.3_1-500", "NC_000913.3_10000-10500"],
                       "lengths": [500, 501]}

# reseq/ReferenceTest.cpp:263-266: every base of test/reference-special-chars.fa (IUPAC codes) loads as N
ka["reference_special_chars"] = {"file": "reference-special-chars.fa", "n_bases": 11, "all": "N"}

# reseq/FragmentDistributionStatsTest.cpp:1020-1048 UpdateRefSeqBias on test/reference-test.fa (two sequences) and test/ref-bias-test.txt
ka["update_ref_seq_bias"] = {
    "file": "ref-bias-test.txt",
    "cases": [   # [mode, stored bias before the call, expected]
        ["keep", [0.5], [1.0, 1.0]],             # a stored vector of the wrong size: keep falls back to no bias
        ["keep", [0.25, 0.5], [0.25, 0.5]],
        ["no", [0.5], [1.0, 1.0]],
        ["file", [0.5], [2.0, 1.0]],
    ],
}

with open(os.path.join(HERE, "reference_known_answers.json"), "w") as f:
    json.dump(ka, f, indent=1)
print("wrote", len(ka), "groups")
