/*
 * oracle_frag.c -- Surrounding, coverage-bias and fragment-count arithmetic.
 * TEST INFRASTRUCTURE (see oracle.h).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SUR_BLOCKS 3
#define SUR_RANGE 10
#define SUR_START 10
#define SUR_LEN (SUR_BLOCKS * SUR_RANGE)
#define SUR_SIZE (1u << (2 * SUR_RANGE))

/* SurroundingBase.hpp:64-73 GetBlock on the forward sequence */
static int32_t get_block_fwd(const uint8_t *seq, uint32_t len, uint64_t start_pos) {
    int32_t sur = 0;
    for (uint64_t pos = start_pos; pos < start_pos + SUR_RANGE; ++pos) {
        sur <<= 2;
        sur += seq[pos % len];
    }
    return sur;
}
/* GetBlock on ConstDna5StringReverseComplement(sequence): element q is complement(seq[len-1-q]) */
static int32_t get_block_rev(const uint8_t *seq, uint32_t len, uint64_t start_pos) {
    int32_t sur = 0;
    for (uint64_t pos = start_pos; pos < start_pos + SUR_RANGE; ++pos) {
        sur <<= 2;
        sur += 3 - seq[len - 1 - (pos % len)];
    }
    return sur;
}

/* SurroundingBase.hpp:31-33,50-52,75-81,196-198 */
void orc_surrounding_forward(const uint8_t *seq, uint32_t len, uint32_t pos, int32_t sur[3]) {
    for (int block = 0; block < SUR_BLOCKS; ++block)
        sur[block] = get_block_fwd(seq, len, (uint64_t)pos + len - SUR_START + (uint64_t)block * SUR_RANGE);
}
/* SurroundingBase.hpp:200-202: Set(reverse complement, length-pos-1) */
void orc_surrounding_reverse(const uint8_t *seq, uint32_t len, uint32_t pos, int32_t sur[3]) {
    uint64_t rpos = (uint64_t)len - pos - 1;
    for (int block = 0; block < SUR_BLOCKS; ++block)
        sur[block] = get_block_rev(seq, len, rpos + len - SUR_START + (uint64_t)block * SUR_RANGE);
}
/* SurroundingBase.hpp:94-104,220-222 */
void orc_surrounding_update_forward(const uint8_t *seq, uint32_t len, uint32_t new_pos, int32_t sur[3]) {
    uint8_t new_base = seq[((uint64_t)new_pos + len - SUR_START + SUR_LEN - 1) % len];
    for (int block = 0; block < SUR_BLOCKS - 1; ++block) {
        sur[block] <<= 2;
        sur[block] %= (int32_t)SUR_SIZE;
        sur[block] += sur[block + 1] >> (2 * (SUR_RANGE - 1));
    }
    sur[SUR_BLOCKS - 1] <<= 2;
    sur[SUR_BLOCKS - 1] %= (int32_t)SUR_SIZE;
    sur[SUR_BLOCKS - 1] += new_base;
}
/* SurroundingBase.hpp:106-114,224-226 */
void orc_surrounding_update_reverse(const uint8_t *seq, uint32_t len, uint32_t new_pos, int32_t sur[3]) {
    uint8_t new_base = (uint8_t)(3 - seq[((uint64_t)new_pos + len + SUR_START) % len]);
    for (int block = SUR_BLOCKS; block-- > 1;) {
        sur[block] >>= 2;
        sur[block] += (sur[block - 1] % 4) << (2 * (SUR_RANGE - 1));
    }
    sur[0] >>= 2;
    sur[0] += (int32_t)new_base << (2 * (SUR_RANGE - 1));
}

/* Surrounding.cpp:194-219 */
void orc_combine_positions(const double *separated, double *bias) {
    memset(bias, 0, sizeof(double) * SUR_BLOCKS * SUR_SIZE);
    uint8_t bases[SUR_RANGE + 1];
    for (int block = 0; block < SUR_BLOCKS; ++block) {
        memset(bases, 0, sizeof bases);
        for (uint32_t sur = 0; sur < SUR_SIZE; ++sur) {
            for (int pos = 0; pos < SUR_RANGE; ++pos)
                bias[(size_t)block * SUR_SIZE + sur] += separated[bases[SUR_RANGE - 1 - pos] + pos * 4 + block * SUR_RANGE * 4];
            int pos = 0;
            while (++bases[pos] > 3) bases[pos++] = 0;
        }
    }
}
/* Surrounding.cpp:221-260 */
void orc_separate_positions(const double *bias, double *separated) {
    memset(separated, 0, sizeof(double) * 4 * SUR_LEN);
    uint8_t bases[SUR_RANGE + 1];
    for (int block = 0; block < SUR_BLOCKS; ++block) {
        memset(bases, 0, sizeof bases);
        for (uint32_t sur = 0; sur < SUR_SIZE; ++sur) {
            for (int pos = 0; pos < SUR_RANGE; ++pos)
                separated[bases[SUR_RANGE - 1 - pos] + pos * 4 + block * SUR_RANGE * 4] += bias[(size_t)block * SUR_SIZE + sur];
            int pos = 0;
            while (++bases[pos] > 3) bases[pos++] = 0;
        }
    }
    for (int sur_pos = 0; sur_pos < SUR_LEN; ++sur_pos) {
        double sur_sum = 0.0;
        for (int sur = 4 * sur_pos; sur < 4 * sur_pos + 4; ++sur) sur_sum += separated[sur];
        sur_sum /= 4;
        for (int sur = 4 * sur_pos; sur < 4 * sur_pos + 4; ++sur) {
            separated[sur] -= sur_sum;
            separated[sur] /= SUR_SIZE / 4;
        }
    }
}

/* Surrounding.h:114-120: blocks are summed from the last to the first */
double orc_surrounding_bias(const double *bias_tab, const int32_t sur[3]) {
    double bias = 0.0;
    for (int block = SUR_BLOCKS; block--;) bias += bias_tab[(size_t)block * SUR_SIZE + (uint32_t)sur[block]];
    return orc_inv_logit2(bias);
}

/* FragmentDistributionStats.cpp:900-907 */
double orc_get_dispersion(double bias, double a, double b) {
    double r = bias / (a + b * bias);
    if (r > bias * 1e10) r = bias * 1e10;
    return r;
}

/* FragmentDistributionStats.cpp:3584-3596 */
uint16_t orc_binomial(uint16_t n, double p, double probability_chosen) {
    double probability_count = pow(1 - p, n), probability_left = probability_chosen - probability_count;
    uint16_t count = 0;
    while (0.0 < probability_left && count < n) {
        ++count;
        probability_count *= (double)(n + 1 - count) / count * p / (1 - p);
        probability_left -= probability_count;
    }
    return count;
}

/* FragmentDistributionStats.cpp:3602-3613 */
uint16_t orc_negative_binomial(double p, double r, double probability_chosen) {
    double probability_count = pow(1 - p, r), probability_left = probability_chosen - probability_count;
    uint16_t count = 0;
    while (0.0 < probability_left) {
        probability_count *= p * ((r - 1) / ++count + 1);
        probability_left -= probability_count;
    }
    return count;
}

/* FragmentDistributionStats.cpp:2969-2976 */
double orc_calculate_non_zero_threshold(const double disp[2], double bias_normalization, double max_bias, uint16_t num_alleles) {
    double max_mean = bias_normalization * max_bias;
    double max_dispersion = orc_get_dispersion(max_mean, disp[0], disp[1]) / num_alleles;
    max_mean /= num_alleles;
    return pow(max_dispersion / (max_dispersion + max_mean), max_dispersion);
}

static double vect_f64_at(const orc_vect_f64 *v, uint64_t i) {     /* Vect::operator[] const: 0 outside the stored range */
    if (i < v->from || i >= v->from + v->size) return 0.0;
    return v->v[i - v->from];
}

/* FragmentDistributionStats.cpp:3615-3627, Reference.h:167-169,283-285; the bias factors passed explicitly */
uint16_t orc_fragment_counts_core(const double *sur_bias, const double disp[2], double bias_normalization, double ref_seq_bias, double insert_length_bias,
                                  double gc_bias, const int32_t sur_start[3], const int32_t sur_end[3], double probability_chosen, uint16_t num_alleles) {
    double general = ref_seq_bias * insert_length_bias;
    double bias = general * gc_bias * orc_surrounding_bias(sur_bias, sur_start) * orc_surrounding_bias(sur_bias, sur_end);
    if (0.0 < bias) {
        double mean = bias * bias_normalization;
        double dispersion = orc_get_dispersion(mean, disp[0], disp[1]) / num_alleles;
        mean /= num_alleles;
        return orc_negative_binomial(mean / (mean + dispersion), dispersion, probability_chosen);
    }
    return 0;
}

uint16_t orc_get_fragment_counts(const orc_profile *p, double bias_normalization, double ref_seq_bias, uint32_t fragment_length, uint8_t gc,
                                 const int32_t sur_start[3], const int32_t sur_end[3], double probability_chosen, uint16_t num_alleles) {
    return orc_fragment_counts_core(p->sur_bias, p->dispersion, bias_normalization, ref_seq_bias, vect_f64_at(&p->insert_lengths_bias, fragment_length),
                                    vect_f64_at(&p->gc_bias, gc), sur_start, sur_end, probability_chosen, num_alleles);
}

/* Simulator.cpp:1341-1361 */
void orc_select_allele(uint16_t *chosen, uint32_t *n_chosen, uint8_t *reverse_selection, uint16_t possible_strands, double random_value) {
    uint16_t chosen_id = (uint16_t)(random_value * (possible_strands - *n_chosen));
    uint16_t replacement_correction = 0;
    for (uint32_t i = 0; i < *n_chosen; ++i)
        if (chosen[i] <= chosen_id) ++replacement_correction;
    while (replacement_correction)
        if (reverse_selection[++chosen_id]) --replacement_correction;
    chosen[(*n_chosen)++] = chosen_id;
    reverse_selection[chosen_id] = 0;
}

/* ------------------------------------------------- Surrounding edits (variants) */
#define SUR_LENGTH 30
#define SUR_ISIZE (1 << 20)
void orc_sur_change_base(int32_t sur[3], uint32_t pos, uint8_t new_base) {               /* Surrounding.cpp:22-26 */
    int bit_in_block = 2 * (SUR_RANGE - (pos % SUR_RANGE) - 1);
    sur[pos / SUR_RANGE] = (sur[pos / SUR_RANGE] & ~(3 << bit_in_block)) + ((int32_t)new_base << bit_in_block);
}
void orc_sur_delete_shift_right(int32_t sur[3], uint32_t pos, uint8_t new_end_base) {    /* :28-43 */
    int32_t new_base = new_end_base;
    int del_block = (int)(pos / SUR_RANGE);
    for (int block = SUR_BLOCKS; --block > del_block;) {                                  /* add the base at the end, shift towards the deletion */
        sur[block] = (sur[block] << 2) + new_base;
        new_base = sur[block] / SUR_ISIZE;
        sur[block] = sur[block] % SUR_ISIZE;
    }
    int bit_in_block = 2 * (SUR_RANGE - (pos % SUR_RANGE) - 1);
    new_base += (sur[del_block] % (1 << bit_in_block)) << 2;
    sur[del_block] = (sur[del_block] >> (bit_in_block + 2) << (bit_in_block + 2)) + new_base;
}
void orc_sur_delete_shift_left(int32_t sur[3], uint32_t pos, uint8_t new_end_base) {     /* :45-60 */
    int32_t new_base = new_end_base;
    int del_block = (int)(pos / SUR_RANGE);
    for (int block = 0; block < del_block; ++block) {
        sur[block] += new_base * SUR_ISIZE;
        new_base = sur[block] % 4;
        sur[block] = sur[block] >> 2;
    }
    int bit_in_block = 2 * (SUR_RANGE - (pos % SUR_RANGE) - 1);
    sur[del_block] += new_base * SUR_ISIZE;
    sur[del_block] = (sur[del_block] >> (bit_in_block + 2) << bit_in_block) + sur[del_block] % (1 << bit_in_block);
}
static int imin(int a, int b) { return a < b ? a : b; }
void orc_sur_insert_shift_right(int32_t sur[3], uint32_t pos, const uint8_t *new_bases, uint32_t n) {       /* :62-125 */
    int block = (int)(pos / SUR_RANGE);
    int bases_to_insert = imin((int)n, SUR_LENGTH - (int)pos);
    int shift_blocks = bases_to_insert / SUR_RANGE, shift_bases = bases_to_insert % SUR_RANGE;
    for (int cur_block = SUR_BLOCKS - shift_blocks; cur_block-- > block + 1;) {
        sur[cur_block] >>= 2 * shift_bases;
        sur[cur_block] += sur[cur_block - 1] % (1 << 2 * shift_bases) * (1 << 2 * (SUR_RANGE - shift_bases));
    }
    int inv_pos_in_block = SUR_RANGE - (int)(pos % SUR_RANGE);
    int32_t tmp_sur = sur[block] % (1 << 2 * inv_pos_in_block);
    sur[block] >>= 2 * inv_pos_in_block;
    tmp_sur >>= 2 * shift_bases;
    if (0 < shift_blocks && SUR_BLOCKS > block + shift_blocks) {
        for (int cur_block = SUR_BLOCKS; cur_block-- > block + shift_blocks + 1;) sur[cur_block] = sur[cur_block - shift_blocks];
        sur[block + shift_blocks] = tmp_sur;
    }
    int ins_pos = 0;
    int into_this_block = imin(bases_to_insert, inv_pos_in_block);
    bases_to_insert -= into_this_block;
    for (; into_this_block--;) {
        sur[block] <<= 2;
        sur[block] += new_bases[ins_pos++];
    }
    if (0 == shift_blocks && inv_pos_in_block > shift_bases) {
        sur[block] <<= 2 * (inv_pos_in_block - shift_bases);
        sur[block] += tmp_sur;
    } else {
        while (0 < bases_to_insert) {
            into_this_block = imin(bases_to_insert, SUR_RANGE);
            bases_to_insert -= into_this_block;
            tmp_sur = sur[++block] % (1 << 2 * (SUR_RANGE - into_this_block));
            sur[block] = 0;
            for (int i = into_this_block; i--;) {
                sur[block] <<= 2;
                sur[block] += new_bases[ins_pos++];
            }
            sur[block] <<= 2 * (SUR_RANGE - into_this_block);
            sur[block] += tmp_sur;
        }
    }
}
void orc_sur_insert_shift_left(int32_t sur[3], uint32_t pos, const uint8_t *new_bases, uint32_t n) {        /* :127-192 */
    int block = (int)(pos / SUR_RANGE);
    int bases_to_insert = imin((int)n, (int)pos + 1);
    int shift_blocks = bases_to_insert / SUR_RANGE, shift_bases = bases_to_insert % SUR_RANGE;
    for (int cur_block = shift_blocks; cur_block < block; ++cur_block) {
        sur[cur_block] %= 1 << 2 * (SUR_RANGE - shift_bases);
        sur[cur_block] <<= 2 * shift_bases;
        sur[cur_block] += sur[cur_block + 1] >> 2 * (SUR_RANGE - shift_bases);
    }
    int pos_in_block = (int)(pos % SUR_RANGE) + 1;
    int32_t tmp_sur = sur[block] % (1 << 2 * (SUR_RANGE - pos_in_block));
    if (pos_in_block > shift_bases) {
        sur[block] >>= 2 * (SUR_RANGE - pos_in_block);
        sur[block] %= 1 << 2 * (pos_in_block - shift_bases);
    } else sur[block] = 0;
    if (0 < shift_blocks) {
        if (block >= shift_blocks) {
            for (int cur_block = 0; cur_block + shift_blocks < block; ++cur_block) sur[cur_block] = sur[cur_block + shift_blocks];
            sur[block - shift_blocks] = sur[block] << 2 * (SUR_RANGE - (pos_in_block - shift_bases));
        }
        sur[block] = 0;
    }
    int into_this_block = imin(bases_to_insert, pos_in_block);
    int ins_pos_to = (int)n;
    for (int ins_pos = ins_pos_to - into_this_block; ins_pos < ins_pos_to; ++ins_pos) {
        sur[block] <<= 2;
        sur[block] += new_bases[ins_pos];
    }
    bases_to_insert -= into_this_block;
    ins_pos_to -= into_this_block;
    sur[block] <<= 2 * (SUR_RANGE - pos_in_block);
    sur[block] += tmp_sur;
    while (0 < bases_to_insert) {
        into_this_block = imin(bases_to_insert, SUR_RANGE);
        sur[--block] >>= 2 * into_this_block;
        for (int ins_pos = ins_pos_to - into_this_block; ins_pos < ins_pos_to; ++ins_pos) {
            sur[block] <<= 2;
            sur[block] += new_bases[ins_pos];
        }
        bases_to_insert -= into_this_block;
        ins_pos_to -= into_this_block;
    }
}
