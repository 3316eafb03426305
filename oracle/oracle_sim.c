/*
 * oracle_sim.c -- the simulation run: pre-passes (bias normalisation, systematic
 * errors), the coverage sieve, FillRead/FillReadPart, CreateReads and the
 * error-model-only mode.  TEST INFRASTRUCTURE (see oracle.h).
 *
 * Variants, methylation and exclusion regions are not restated (SURVEY.md
 * section 8 rows a16/a17 are "next").
 */
#define _POSIX_C_SOURCE 200809L      /* getline, ssize_t */
#include <sys/types.h>
#include "oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_SIZE 1000u                  /* Simulator.h:254 kBlockSize */
#define SAMPLING_DISTANCE 20u             /* FragmentDistributionStats.h:235 kInsertLengthSamplingDistance */
#define SQ_FRAGLEN_BIN 10u                /* QualityStats.h:15 kSqFragmentLengthBinSize */

static int is_gc(uint8_t b) { return b == 1 || b == 2; }

static uint64_t vect_u64_at(const orc_vect_u64 *v, uint64_t i) {
    if (i < v->from || i >= v->from + v->size) return 0;
    return v->v[i - v->from];
}
static double vect_f64_at(const orc_vect_f64 *v, uint64_t i) {
    if (i < v->from || i >= v->from + v->size) return 0.0;
    return v->v[i - v->from];
}

/* first cumulative probability strictly above u (inverse CDF of std::discrete_distribution) */
static uint32_t discrete_draw(const double *cp, size_t n, double u) {
    if (n < 2) return 0;
    uint32_t i = 0;
    while (i + 1 < n && !(cp[i] > u)) ++i;
    return i;
}

/* ------------------------------------------------------------------ reference */
orc_reference *orc_reference_new(uint32_t n_seqs) {
    orc_reference *r = calloc(1, sizeof *r);
    r->n_seqs = n_seqs;
    r->len = calloc(n_seqs ? n_seqs : 1, sizeof(uint32_t));
    r->codes = calloc(n_seqs ? n_seqs : 1, sizeof(uint8_t *));
    r->first_name = calloc(n_seqs ? n_seqs : 1, sizeof(char *));
    r->full_name = calloc(n_seqs ? n_seqs : 1, sizeof(char *));
    return r;
}
void orc_reference_set(orc_reference *r, uint32_t i, const char *name, const uint8_t *codes, uint32_t len) {
    r->len[i] = len;
    r->codes[i] = malloc(len ? len : 1);
    memcpy(r->codes[i], codes, len);
    size_t n = 0;
    while (name[n] && name[n] != ' ') ++n;          /* Reference.cpp:476-480 ReferenceIdFirstPart */
    r->first_name[i] = malloc(n + 1);
    memcpy(r->first_name[i], name, n);
    r->first_name[i][n] = 0;
    r->full_name[i] = malloc(strlen(name) + 1);
    strcpy(r->full_name[i], name);
}
void orc_reference_free(orc_reference *r) {
    if (!r) return;
    for (uint32_t i = 0; i < r->n_seqs; ++i) {
        free(r->codes[i]);
        free(r->first_name[i]);
        free(r->full_name[i]);
    }
    free(r->codes);
    free(r->first_name);
    free(r->full_name);
    free(r->len);
    free(r);
}
/* Reference::ReplaceN (Reference.cpp:813-886): short stretches of N become uniform bases, stretches of kMinNToReplaceNWithRepeat
 * (100) or more a four-base repeat built from the flanks.  The draw for position p is Philox(seed; p, sequence, 0, 5<<28).w0 & 3
 * (the reference consumes one mt19937_64 stream in order). */
void orc_reference_replace_n(orc_reference *r, uint64_t seed) {
#define DRAW(pos) ((uint8_t)(orc_philox4x32_10(seed, (uint32_t)(pos), s, 0, (uint32_t)ORC_DOM_REPLACEN << 28).w[0] & 3u))
    for (uint32_t s = 0; s < r->n_seqs; ++s) {
        uint8_t *seq = r->codes[s];
        uint32_t len = r->len[s];
        for (uint32_t start = 0; start < len;) {
            if (seq[start] <= 3) {
                ++start;
                continue;
            }
            uint32_t end = start;                                       /* :822-823 */
            while (++end < len && seq[end] > 3) {}
            if (end - start < 100) {                                    /* :826-831 */
                for (uint32_t pos = start; pos < end; ++pos) seq[pos] = DRAW(pos);
            } else {
                uint8_t short_repeat[4];
                if (2 > start) {
                    if (end + 4 > len) {                                /* :838-842 */
                        for (uint32_t k = 0; k < 4; ++k) short_repeat[k] = DRAW(start + k);
                    } else {                                            /* :845-852 */
                        for (uint32_t k = 0; k < 4; ++k) short_repeat[k] = seq[end + k];
                        for (uint32_t pos = 4; --pos;)
                            if (short_repeat[pos] > 3) short_repeat[pos] = DRAW(end + pos);
                    }
                } else if (end + 2 > len) {
                    if (4 > start) {                                    /* :857-861 */
                        for (uint32_t k = 0; k < 4; ++k) short_repeat[k] = DRAW(start + k);
                    } else {                                            /* :864-865 */
                        for (uint32_t k = 0; k < 4; ++k) short_repeat[k] = seq[start - 4 + k];
                    }
                } else {                                                /* :869-876 */
                    short_repeat[0] = seq[end];
                    short_repeat[1] = seq[end + 1];
                    short_repeat[2] = seq[start - 2];
                    short_repeat[3] = seq[start - 1];
                    if (short_repeat[1] > 3) short_repeat[1] = DRAW(end + 1);
                }
                for (uint32_t pos = start; pos < end; ++pos) seq[pos] = short_repeat[(pos - start) % 4];   /* :880-882 */
            }
            start = end;
        }
    }
#undef DRAW
}

/* --------------------------------------------------------- systematic errors */
/* Simulator.h:337-382 SetSystematicErrors + DrawSystematicError over one chain (sequence or adapter). */
void orc_systematic_errors(const orc_profile *p, uint64_t seed, uint32_t chain_c1, uint32_t chain_c2, const uint8_t *seq_in, uint32_t len,
                           int reverse_complement, uint16_t gc_range, uint8_t *dom_base_state, uint8_t *dom_out, uint8_t *rate_out) {
    uint8_t *tmp = NULL;
    const uint8_t *seq = seq_in;
    if (reverse_complement) {
        tmp = malloc(len ? len : 1);
        for (uint32_t i = 0; i < len; ++i) tmp[i] = (uint8_t)(3 - seq_in[len - 1 - i]);
        seq = tmp;
    }
    /* ResetSystematicErrorCounters (Simulator.cpp:723-732): Clear() leaves dom_base_ untouched */
    uint8_t last_base = 4;
    orc_dominant_base dom;
    dom.dom_base = *dom_base_state;
    orc_dombase_clear(&dom);
    uint16_t gc = 0, gc_bases = 0;
    uint32_t dist = 0;
    uint8_t start_rate = 0;
    for (uint32_t pos = 0; pos < len; ++pos) {
        uint8_t ref_base = seq[pos];
        orc_philox_out w = orc_philox4x32_10(seed, pos, chain_c1, chain_c2, (uint32_t)ORC_DOM_SYSERR << 28);
        uint32_t index[3] = {orc_transform_distance(dist), orc_safe_percent_u16(gc, gc_bases), start_rate};
        double prob_sum;
        uint8_t dom_error = (uint8_t)orc_draw(&p->dom_error[ref_base][last_base][dom.dom_base], index, orc_u32(w.w[0]), &prob_sum);
        if (0.0 == prob_sum) dom_error = 4;
        uint8_t error_rate = (uint8_t)orc_draw(&p->error_rate[ref_base][dom_error], index, orc_u32(w.w[1]), &prob_sum);
        if (0.0 == prob_sum) error_rate = 0;
        dom_out[pos] = dom_error;
        rate_out[pos] = error_rate;

        last_base = ref_base;
        orc_dombase_update(&dom, ref_base, seq, len, pos);
        orc_update_distances(p->reset_distance, &dist, &start_rate, error_rate);
        if (is_gc(seq[pos])) ++gc;                                  /* Simulator.h:354-366 UpdateGC */
        if (gc_bases < gc_range) ++gc_bases;
        else if (is_gc(seq[pos - gc_bases])) --gc;
    }
    *dom_base_state = dom.dom_base;
    free(tmp);
}

/* ------------------------------------------------- coverage <-> pairs helpers */
/* Simulator.cpp:61-78.  The non-mapped table is optional in the container (absent = all zero). */
double orc_coverage_prop_lost_from_adapters(const orc_profile *p) {
    uint64_t adapter_bases = 0, total_bases = 0;
    for (int seg = 2; seg--;) {
        const orc_rl_by_fl *r = &p->rl_by_fl[seg];
        char name[64];
        snprintf(name, sizeof name, "rl_by_fl_nonmapped.%d.values", seg);
        const orc_array *nm = orc_container_get(p->c, name);
        for (uint32_t row = 0; row < r->rows; ++row) {
            uint64_t frag_len = r->from + row;
            for (uint32_t j = r->row_ptr[row]; j < r->row_ptr[row + 1]; ++j) {
                uint64_t read_len = r->row_from[row] + (j - r->row_ptr[row]);
                uint64_t cnt = r->values[j];
                uint64_t non_mapped = nm ? ((const uint64_t *)nm->data)[j] : 0;
                total_bases += cnt * read_len;
                if (frag_len < read_len) {
                    adapter_bases += (cnt - non_mapped) * (read_len - frag_len);
                    adapter_bases += non_mapped * read_len;
                }
            }
        }
    }
    return (double)adapter_bases / total_bases;
}
uint64_t orc_coverage_to_number_pairs(double coverage, uint64_t total_ref_size, double average_read_length, double adapter_part) {
    return (uint64_t)round(coverage * total_ref_size / average_read_length / 2 / (1 - adapter_part));
}
double orc_number_pairs_to_coverage(uint64_t total_pairs, uint64_t total_ref_size, double average_read_length, double adapter_part) {
    return (double)total_pairs / total_ref_size * average_read_length * 2 * (1 - adapter_part);
}

/* ------------------------------------------------------ bias normalisation a14 */
/* Reference.cpp:622-659 SumBias */
static double sum_bias_core(const orc_vect_f64 *gc_bias, const double *sur_bias, double *max_bias, const uint8_t *seq, uint32_t len, uint32_t fragment_length,
                            double general_bias) {
    double tot = 0.0;
    uint32_t gc = 0;
    for (uint32_t i = 0; i < fragment_length; ++i)
        if (is_gc(seq[i])) ++gc;
    int32_t start_sur[3], end_sur[3];
    orc_surrounding_forward(seq, len, 0, start_sur);
    orc_surrounding_reverse(seq, len, fragment_length - 1, end_sur);
    double bias = general_bias * vect_f64_at(gc_bias, orc_percent_u32(gc, fragment_length)) * orc_surrounding_bias(sur_bias, start_sur) *
                  orc_surrounding_bias(sur_bias, end_sur);
    if (bias > *max_bias) *max_bias = bias;
    tot += bias;
    for (uint32_t start_pos = 0; start_pos < len - fragment_length;) {
        if (is_gc(seq[start_pos + fragment_length])) gc += 1;          /* Reference.h:149-165 UpdateGC */
        if (is_gc(seq[start_pos])) gc -= 1;
        orc_surrounding_update_reverse(seq, len, start_pos + fragment_length, end_sur);
        orc_surrounding_update_forward(seq, len, ++start_pos, start_sur);
        bias = general_bias * vect_f64_at(gc_bias, orc_percent_u32(gc, fragment_length)) * orc_surrounding_bias(sur_bias, start_sur) *
               orc_surrounding_bias(sur_bias, end_sur);
        if (bias > *max_bias) *max_bias = bias;
        tot += bias;
    }
    return tot;
}

static double sum_bias(const orc_profile *p, double *max_bias, const uint8_t *seq, uint32_t len, uint32_t fragment_length, double general_bias) {
    return sum_bias_core(&p->gc_bias, p->sur_bias, max_bias, seq, len, fragment_length, general_bias);
}
/* exported with explicit bias tables for the reference's ReferenceTest::TestSumBias known answer */
double orc_sum_bias(const double *gc_bias, uint32_t gc_from, uint32_t gc_size, const double *sur_bias, const uint8_t *seq, uint32_t len, uint32_t fragment_length,
                    double general_bias, double *max_bias) {
    orc_vect_f64 v = {gc_from, gc_size, gc_bias};
    return sum_bias_core(&v, sur_bias, max_bias, seq, len, fragment_length, general_bias);
}

/* FragmentDistributionStats.cpp:1656-1705 GetSamplePositions */
static uint32_t get_sample_positions(const orc_vect_u64 *il, uint32_t **out) {
    uint64_t to = il->from + il->size;
    uint32_t first_sample = (uint32_t)(il->from > 1 ? il->from : 1);
    while (first_sample < to && 0 == vect_u64_at(il, first_sample)) ++first_sample;
    uint32_t num_samples = 0, hit_zero = 0;
    for (uint32_t len = first_sample; len < to; len += SAMPLING_DISTANCE) {
        if (hit_zero) {
            if (vect_u64_at(il, len) >= 10) {
                num_samples += (len - hit_zero) / SAMPLING_DISTANCE + 1;
                hit_zero = 0;
            }
        } else if (vect_u64_at(il, len) > 0) ++num_samples;
        else hit_zero = len;
    }
    if (2 > num_samples) {
        *out = NULL;
        return 0;
    }
    uint32_t *sp = calloc(num_samples, sizeof(uint32_t));
    sp[0] = first_sample;
    uint32_t found_zeros = 0;
    for (uint32_t k = 1; k < num_samples - found_zeros; ++k) {
        sp[k] = sp[k - 1] + SAMPLING_DISTANCE;
        while (0 == vect_u64_at(il, sp[k])) {
            ++found_zeros;
            sp[k] += SAMPLING_DISTANCE;
        }
    }
    *out = sp;
    return num_samples - found_zeros;
}

/* FragmentDistributionStats.cpp:418-467 PrepareSplines; b,c,d are [(n-1)][n] row-major */
static void prepare_splines(uint32_t n, const uint32_t *x, double *b, double *c_out, double *d) {
    uint32_t nh = n - 1;
    double *h = calloc(nh, sizeof(double)), *mu = calloc(nh, sizeof(double)), *l = calloc(n, sizeof(double));
    double *beta = calloc((size_t)nh * n, sizeof(double)), *z = calloc((size_t)n * n, sizeof(double)), *c = calloc((size_t)n * n, sizeof(double));
    memset(b, 0, sizeof(double) * nh * n);
    memset(d, 0, sizeof(double) * nh * n);
    for (uint32_t i = 0; i < nh; ++i) h[i] = (double)(uint32_t)(x[i + 1] - x[i]);
    for (uint32_t k = 1; k < nh; ++k) {
        beta[(size_t)k * n + k + 1] = 3 / h[k];
        beta[(size_t)k * n + k] = -3 / h[k] - 3 / h[k - 1];
        beta[(size_t)k * n + k - 1] = 3 / h[k - 1];
    }
    l[0] = 0.0;
    mu[0] = 0.0;
    for (uint32_t k = 1; k < nh; ++k) {
        l[k] = 2 * (double)(uint32_t)(x[k + 1] - x[k - 1]) - h[k - 1] * mu[k - 1];
        mu[k] = h[k] / l[k];
        for (uint32_t ai = 0; ai < n; ++ai) z[(size_t)k * n + ai] = (beta[(size_t)k * n + ai] - h[k - 1] * z[(size_t)(k - 1) * n + ai]) / l[k];
    }
    l[n - 1] = 1.0;
    for (uint32_t i = nh; i--;) {
        for (uint32_t ai = 0; ai < n; ++ai) {
            c[(size_t)i * n + ai] = z[(size_t)i * n + ai] - mu[i] * c[(size_t)(i + 1) * n + ai];
            b[(size_t)i * n + ai] = -h[i] * (c[(size_t)(i + 1) * n + ai] + 2 * c[(size_t)i * n + ai]) / 3;
            d[(size_t)i * n + ai] = (c[(size_t)(i + 1) * n + ai] - c[(size_t)i * n + ai]) / 3 / h[i];
        }
        b[(size_t)i * n + i + 1] += 1 / h[i];
        b[(size_t)i * n + i] -= 1 / h[i];
    }
    for (uint32_t i = nh; i--;)
        for (uint32_t ai = 0; ai < n; ++ai) c_out[(size_t)i * n + ai] = c[(size_t)i * n + ai];
    free(h); free(mu); free(l); free(beta); free(z); free(c);
}

/* FragmentDistributionStats.cpp:1729-1740 InterpolateNormalizationWithSpline (+ :1217-1223, :1232, :1264-1297, :1299-1326, :1539-1567) */
static void interpolate_normalization(const orc_profile *p, uint32_t n, const uint32_t *knots, double *normalization, uint32_t norm_size) {
    double *sampled = calloc(n, sizeof(double)), *pars = calloc(n + 1, sizeof(double));
    for (uint32_t s = 0; s < n; ++s) sampled[s] = normalization[knots[s]] / vect_f64_at(&p->insert_lengths_bias, knots[s]);
    double *lb = calloc((size_t)(n - 1) * n, sizeof(double)), *lc = calloc((size_t)(n - 1) * n, sizeof(double)), *ld = calloc((size_t)(n - 1) * n, sizeof(double));
    prepare_splines(n, knots, lb, lc, ld);
    pars[0] = 1.0;
    for (uint32_t k = 0; k < n; ++k) {                              /* knots sit on the sample positions */
        pars[k + 1] = sampled[k];
        pars[k + 1] = pars[k + 1] > 0.0 ? log(pars[k + 1]) : log(1e-10);
    }
    for (uint32_t len = 1; len < knots[0]; ++len) normalization[len] = 0.0;
    uint32_t k = 0;
    double a = 0, b = 0, c = 0, d = 0;
    for (; k < n - 1; ++k) {
        a = pars[k + 1];                                            /* :485-496 GetSplineCoefficients */
        b = c = d = 0.0;
        for (uint32_t ai = 1; ai < n + 1; ++ai) {
            b += pars[ai] * lb[(size_t)k * n + ai - 1];
            c += pars[ai] * lc[(size_t)k * n + ai - 1];
            d += pars[ai] * ld[(size_t)k * n + ai - 1];
        }
        normalization[knots[k]] = vect_f64_at(&p->insert_lengths_bias, knots[k]) * exp(a);
        for (uint32_t len = knots[k] + 1; len < knots[k + 1]; ++len) {
            uint32_t cur_len = len - knots[k];
            normalization[len] = vect_f64_at(&p->insert_lengths_bias, len) * exp(a + b * cur_len + c * cur_len * cur_len + d * cur_len * cur_len * cur_len);
        }
    }
    normalization[knots[k]] = vect_f64_at(&p->insert_lengths_bias, knots[k]) * exp(pars[k + 1]);
    uint32_t cur_len = knots[k] - knots[k - 1];
    double slope = (b + c * cur_len);
    for (uint32_t len = knots[k] + 1; len < norm_size; ++len)
        normalization[len] = vect_f64_at(&p->insert_lengths_bias, len) * exp(pars[k + 1] + (len - knots[k]) * slope);
    free(sampled); free(pars); free(lb); free(lc); free(ld);
}

typedef struct { double bias; uint32_t id; } bias_id;
static int cmp_bias_id(const void *a, const void *b) {
    const bias_id *x = a, *y = b;
    if (x->bias < y->bias) return -1;
    if (x->bias > y->bias) return 1;
    return (x->id > y->id) - (x->id < y->id);
}

/* FragmentDistributionStats.cpp:3504-3582 CalculateBiasNormalization (+ :2909-2932 SplitCoverageGroups) */
static double calculate_bias_normalization(orc_sim *s) {
    const orc_profile *p = s->p;
    const orc_reference *r = s->r;
    uint32_t *sp;
    uint32_t ns = get_sample_positions(&p->insert_lengths, &sp);
    if (!ns) return 0.0;
    uint32_t to = s->insert_to;

    bias_id *sorted = calloc(r->n_seqs, sizeof(bias_id));
    for (uint32_t i = 0; i < r->n_seqs; ++i) {
        sorted[i].bias = s->ref_seq_bias[r->n_seqs - 1 - i];
        sorted[i].id = r->n_seqs - 1 - i;
    }
    qsort(sorted, r->n_seqs, sizeof(bias_id), cmp_bias_id);
    double group_start = sorted[0].bias;
    uint32_t group = 0;
    for (uint32_t i = 0; i < r->n_seqs; ++i) {
        if (sorted[i].bias > 2 * group_start) {
            group_start = sorted[i].bias;
            ++group;
        }
        s->coverage_groups[sorted[i].id] = group;
    }
    free(sorted);
    s->n_groups = group + 1;
    s->thresholds = calloc((size_t)s->n_groups * to * 2, sizeof(double));
    double *norm = calloc(to, sizeof(double));

    /* FillParamsSimulation (:2185-2200): ref ids descending, sampled lengths ascending; one worker */
    for (uint32_t ref_id = r->n_seqs; ref_id--;) {
        if (0.0 == s->ref_seq_bias[ref_id]) continue;
        for (uint32_t k = 0; k < ns; ++k) {
            uint32_t fl = sp[k];
            if (fl > r->len[ref_id]) continue;
            double *mb = &s->thresholds[((size_t)s->coverage_groups[ref_id] * to + fl) * 2];
            norm[fl] += sum_bias(p, mb, r->codes[ref_id], r->len[ref_id], fl, s->ref_seq_bias[ref_id] * vect_f64_at(&p->insert_lengths_bias, fl));
        }
    }
    interpolate_normalization(p, ns, sp, norm, to);

    for (uint32_t g = 0; g < s->n_groups; ++g) {
        double *grp = &s->thresholds[(size_t)g * to * 2];
        double max_ratio = 0.0;
        for (uint32_t k = 0; k < ns; ++k) {
            double ratio = grp[2 * sp[k]] / vect_f64_at(&p->insert_lengths_bias, sp[k]);
            if (ratio > max_ratio) max_ratio = ratio;
        }
        for (uint32_t k = 1; k < ns; ++k)
            for (uint32_t fl = sp[k - 1] + 1; fl < sp[k]; ++fl) grp[2 * fl] = max_ratio * vect_f64_at(&p->insert_lengths_bias, fl);
        for (uint32_t fl = sp[ns - 1] + 1; fl < to; ++fl) grp[2 * fl] = max_ratio * vect_f64_at(&p->insert_lengths_bias, fl);
    }
    double normalization = 0.0;
    for (uint32_t i = 0; i < to; ++i) normalization += norm[i];
    double full_normalization = s->total_pairs / (normalization * 2);
    for (size_t i = 0; i < (size_t)s->n_groups * to; ++i) {
        double *t = &s->thresholds[2 * i];
        if (0.0 == t[0]) {
            t[0] = 1.0;
            t[1] = 1.0;
        } else {
            t[0] = orc_calculate_non_zero_threshold(p->dispersion, full_normalization, t[0], s->num_alleles);   /* FDS.cpp:3575-3576 */
            t[1] = pow(t[0], 2 * s->num_alleles);
        }
    }
    s->norm_by_len = norm;
    free(sp);
    return full_normalization;
}

/* ------------------------------------------------------------------- sim setup */
static void adapter_sys_errors(orc_sim *s, uint8_t *dom_state) {
    const orc_profile *p = s->p;
    for (int seg = 2; seg--;) {                                       /* Simulator.cpp:2784-2797 */
        const orc_adapters *a = &p->adapters[seg];
        s->adapter_dom[seg] = calloc(a->n ? a->n : 1, sizeof(uint8_t *));
        s->adapter_rate[seg] = calloc(a->n ? a->n : 1, sizeof(uint8_t *));
        for (uint32_t i = a->n; i--;) {
            if (!a->counts[i]) continue;
            uint32_t len = a->seq_ptr[i + 1] - a->seq_ptr[i];
            s->adapter_dom[seg][i] = malloc(len ? len : 1);
            s->adapter_rate[seg][i] = malloc(len ? len : 1);
            orc_systematic_errors(p, s->seed, i, 2u + (uint32_t)seg, a->seqs + a->seq_ptr[i], len, 0, s->sys_gc_range, dom_state, s->adapter_dom[seg][i],
                                  s->adapter_rate[seg][i]);
        }
    }
}

static int read_ref_bias_file(const char *path, const orc_reference *r, double *bias, char *err, size_t err_cap);

orc_sim *orc_sim_new(const orc_profile *p, const orc_reference *r, uint64_t seed, uint64_t num_read_pairs, double coverage,
                     const char *record_base_identifier) {
    return orc_sim_new_bias(p, r, seed, num_read_pairs, coverage, record_base_identifier, 0, NULL, NULL, 0);
}

orc_sim *orc_sim_new_bias(const orc_profile *p, const orc_reference *r, uint64_t seed, uint64_t num_read_pairs, double coverage,
                          const char *record_base_identifier, int ref_bias_mode, const char *ref_bias_file, char *err, size_t err_cap) {
    return orc_sim_new_variants(p, r, NULL, seed, num_read_pairs, coverage, record_base_identifier, ref_bias_mode, ref_bias_file, err, err_cap);
}

orc_sim *orc_sim_new_variants(const orc_profile *p, const orc_reference *r, const orc_variants *vs, uint64_t seed, uint64_t num_read_pairs, double coverage,
                              const char *record_base_identifier, int ref_bias_mode, const char *ref_bias_file, char *err, size_t err_cap) {
    orc_sim *s = calloc(1, sizeof *s);
    s->num_alleles = vs ? (uint16_t)vs->num_alleles : 1;
    s->p = p;
    s->r = r;
    s->seed = seed;
    snprintf(s->base_identifier, sizeof s->base_identifier, "%s",
             record_base_identifier && record_base_identifier[0] ? record_base_identifier : "ReseqRead");   /* Simulator.cpp:2705-2710 */

    uint64_t reads = 0, sum_read_length = 0;                        /* Simulator.cpp:2713-2721 */
    for (int seg = 2; seg--;)
        for (uint64_t len = p->read_lengths[seg].from; len < p->read_lengths[seg].from + p->read_lengths[seg].size; ++len) {
            reads += vect_u64_at(&p->read_lengths[seg], len);
            sum_read_length += vect_u64_at(&p->read_lengths[seg], len) * len;
        }
    double average_read_length = (double)sum_read_length / reads;
    s->sys_gc_range = (uint16_t)(((sum_read_length + reads / 2) / reads) / 2);      /* :2782 Divide(sum,reads)/2 */
    uint8_t dom_state = 0;                                          /* DominantBase(): dom_base_(0) */

    if (!r) {                                                       /* SimulateErrorModelOnly: Simulator.cpp:2951-2977 */
        adapter_sys_errors(s, &dom_state);
        return s;
    }

    uint64_t total_ref_size = 0;
    for (uint32_t i = 0; i < r->n_seqs; ++i) total_ref_size += r->len[i];
    if (num_read_pairs) s->total_pairs = num_read_pairs;
    else {
        double adapter_part = orc_coverage_prop_lost_from_adapters(p);
        if (0.0 == coverage) coverage = p->corrected_coverage;
        s->total_pairs = orc_coverage_to_number_pairs(coverage, total_ref_size, average_read_length, adapter_part);
    }
    s->num_adapter_only_pairs =
        (uint64_t)round((double)s->total_pairs * vect_u64_at(&p->insert_lengths, 0) / (p->total_number_reads / 2));   /* :2739 */
    s->total_pairs -= s->num_adapter_only_pairs;

    /* UpdateRefSeqBias (FragmentDistributionStats.cpp:3352-3500) */
    s->ref_seq_bias = calloc(r->n_seqs ? r->n_seqs : 1, sizeof(double));
    for (uint32_t i = 0; i < r->n_seqs; ++i) s->ref_seq_bias[i] = 1.0;                       /* kNo, and kKeep's fallback */
    if (0 == ref_bias_mode) {                                                                  /* kKeep :3354-3360 */
        if (p->n_ref_bias == r->n_seqs)
            for (uint32_t i = 0; i < r->n_seqs; ++i) s->ref_seq_bias[i] = p->ref_seq_bias[i];
    } else if (2 == ref_bias_mode) {                                                           /* kDraw :3365-3384, with replacement; Philox domain 6 */
        for (uint32_t i = 0; i < r->n_seqs; ++i) {
            uint32_t k = (uint32_t)(orc_u32(orc_philox4x32_10(seed, i, 0, 0, 6u << 28).w[0]) * (double)p->n_ref_bias);
            s->ref_seq_bias[i] = p->ref_seq_bias[k < p->n_ref_bias ? k : p->n_ref_bias - 1];
        }
    } else if (3 == ref_bias_mode) {                                                           /* kFile :3386-3495 */
        if (read_ref_bias_file(ref_bias_file, r, s->ref_seq_bias, err, err_cap)) {
            free(s->ref_seq_bias);
            free(s);
            return NULL;
        }
    }

    s->insert_to = (uint32_t)(p->insert_lengths.from + p->insert_lengths.size);
    s->coverage_groups = calloc(r->n_seqs, sizeof(uint32_t));
    s->bias_normalization = calculate_bias_normalization(s);

    adapter_sys_errors(s, &dom_state);

    /* blocks and systematic errors per unit (Simulator.cpp:911-1009,1149-1240) */
    s->first_block = calloc(r->n_seqs, sizeof(uint32_t));
    s->n_blocks = calloc(r->n_seqs, sizeof(uint32_t));
    for (int strand = 0; strand < 2; ++strand) {
        s->sys_dom[strand] = calloc(r->n_seqs, sizeof(uint8_t *));
        s->sys_rate[strand] = calloc(r->n_seqs, sizeof(uint8_t *));
    }
    uint32_t next_block = 1;
    for (uint32_t i = 0; i < r->n_seqs; ++i) {
        if (r->len[i] < s->insert_to) continue;                      /* Simulator.cpp:1159,1186 */
        s->first_block[i] = next_block;
        s->n_blocks[i] = (r->len[i] + BLOCK_SIZE - 1) / BLOCK_SIZE;
        next_block += s->n_blocks[i];
        for (int strand = 2; strand--;) {                            /* reverse strand first (CreateUnit), then forward */
            s->sys_dom[strand][i] = malloc(r->len[i]);
            s->sys_rate[strand][i] = malloc(r->len[i]);
            orc_systematic_errors(p, seed, i, (uint32_t)strand, r->codes[i], r->len[i], strand, s->sys_gc_range, &dom_state, s->sys_dom[strand][i],
                                  s->sys_rate[strand][i]);
        }
    }
    s->total_blocks = next_block - 1;
    if (vs && orc_var_attach(s, vs)) {
        if (err) snprintf(err, err_cap, "%s", orc_var_last_error());
        orc_sim_free(s);
        return NULL;
    }
    return s;
}

void orc_sim_set_normalization(orc_sim *s, double bias_normalization, const double *thresholds) {
    s->bias_normalization = bias_normalization;
    memcpy(s->thresholds, thresholds, sizeof(double) * (size_t)s->n_groups * s->insert_to * 2);
}

void orc_sim_free(orc_sim *s) {
    if (s && s->var_state) orc_var_detach(s);
    if (!s) return;
    if (s->meth_n) {
        for (uint32_t i = 0; i < s->r->n_seqs; ++i) {
            free(s->meth_first[i]);
            free(s->meth_second[i]);
            if (s->meth_rate_cols[i]) {
                for (uint32_t a = 0; a < s->meth_n_cols[i]; ++a) free(s->meth_rate_cols[i][a]);
                free(s->meth_rate_cols[i]);
            }
        }
        free(s->meth_n);
        free(s->meth_n_cols);
        free(s->meth_first);
        free(s->meth_second);
        free(s->meth_rate);
        free(s->meth_rate_cols);
    }
    for (int seg = 0; seg < 2; ++seg) {
        if (s->adapter_dom[seg])
            for (uint32_t i = 0; i < s->p->adapters[seg].n; ++i) {
                free(s->adapter_dom[seg][i]);
                free(s->adapter_rate[seg][i]);
            }
        free(s->adapter_dom[seg]);
        free(s->adapter_rate[seg]);
        if (s->r && s->sys_dom[seg])
            for (uint32_t i = 0; i < s->r->n_seqs; ++i) {
                free(s->sys_dom[seg][i]);
                free(s->sys_rate[seg][i]);
            }
        free(s->sys_dom[seg]);
        free(s->sys_rate[seg]);
    }
    free(s->coverage_groups);
    free(s->thresholds);
    free(s->norm_by_len);
    free(s->ref_seq_bias);
    free(s->first_block);
    free(s->n_blocks);
    free(s);
}

/* ----------------------------------------------------------------------- sieve */
/* Simulator.cpp:2249-2357 for one allele, no variants: the duplicates of every (start, length, strand) site. */
/* The zero-threshold test of SimulateFromGivenBlock without one uniform per cell (hard part 3).  The reference
   draws probability_chosen ~ U[0,1) for every (start, fragment length) and goes on iff it is >= NonZeroThreshold (Simulator.cpp:2304-2306,
   Simulator.h:415-420); the cells of a start position are independent, a cell passes with probability 1 - thr1[len], and given that it
   passes probability_chosen ~ U[thr1[len], 1).  The same process drawn directly: with q[len] = product of thr1 over the lengths up to
   len (the probability that none of them passes), the first passing length behind cur-1 is the first len with q[len] <= u * q[cur-1]
   for one uniform u, and its probability_chosen = thr1 + v * (1 - thr1) for a second one: 1 + passes draws per start position
   instead of one per cell.  The running product restarts (a new segment, a fresh draw) where it falls below 2^-500, so a threshold of
   exactly zero -- a length that always passes -- ends its segment.  Draw k of a start position: Philox block (start, c1, k, 1<<28),
   u = u53(w0, w1), v = u53(w2, w3) (DESIGN.md "Random streams"). */
void orc_gap_table(const orc_sim *s, uint32_t group, double *q, uint32_t *seg_end) {
    const uint32_t to = s->insert_to;
    const uint32_t from = (uint32_t)(s->p->insert_lengths.from > 1 ? s->p->insert_lengths.from : 1);
    const double *thr = &s->thresholds[(size_t)group * to * 2];
    uint32_t seg_begin = from;
    double run = 1.0;
    for (uint32_t len = 0; len < to; ++len) {
        q[len] = 1.0;
        seg_end[len] = from;                                  /* lengths in front of the first one: unused */
    }
    for (uint32_t len = from; len < to; ++len) {
        run *= thr[2 * len + 1];
        q[len] = run;
        if (run < 0x1p-500 || len + 1 == to) {
            for (uint32_t l = seg_begin; l <= len; ++l) seg_end[l] = len + 1;
            seg_begin = len + 1;
            run = 1.0;
        }
    }
}
uint32_t orc_gap_hits(const orc_sim *s, const double *q, const uint32_t *seg_end, const double *thr, uint32_t start, uint32_t c1, orc_gap_hit *out) {
    const uint32_t to = s->insert_to;
    const uint32_t from = (uint32_t)(s->p->insert_lengths.from > 1 ? s->p->insert_lengths.from : 1);
    uint32_t cur = from, k = 0, n = 0;
    while (cur < to) {
        const orc_philox_out w = orc_philox4x32_10(s->seed, start, c1, k++, (uint32_t)ORC_DOM_SIEVE << 28);
        const uint32_t e = seg_end[cur];
        const double base = (cur == from || seg_end[cur - 1] == cur) ? 1.0 : q[cur - 1];
        const double target = orc_u53(w.w[0], w.w[1]) * base;
        uint32_t len = cur;
        while (len < e && !(q[len] <= target)) ++len;         /* the first length of the segment that the product has reached the target at */
        if (len < e) {
            out[n].len = len;
            out[n].probability_chosen = thr[2 * len + 1] + orc_u53(w.w[2], w.w[3]) * (1 - thr[2 * len + 1]);
            ++n;
            cur = len + 1;
        } else cur = e;
    }
    return n;
}

/* Simulator.cpp:2302-2306 AS WRITTEN: one uniform per (start, fragment length), the cell goes on iff ProbabilityAboveThreshold (Simulator.h:418-420).  ~1000 draws per
   start position of which ~0.2 % pass; the product and orc_gap_hits above draw the same process by its gaps.  Stream: Philox block (start, c1, len, 1<<28 | 2) -- a stream
   of its own, so the literal and the gap run of one seed are two independent samples of the process (tests/test_statistics.py compares them).  From the second
   draw on everything is shared with the gap route: strands, alleles and counts are keyed by (start, sequence, length). */
uint32_t orc_literal_hits(const orc_sim *s, const double *thr, uint32_t start, uint32_t c1, orc_gap_hit *out) {
    const uint32_t to = s->insert_to;
    const uint32_t from = (uint32_t)(s->p->insert_lengths.from > 1 ? s->p->insert_lengths.from : 1);   /* :2298 */
    uint32_t n = 0;
    for (uint32_t len = from; len < to; ++len) {                                                       /* :2302 */
        const orc_philox_out w = orc_philox4x32_10(s->seed, start, c1, len, ((uint32_t)ORC_DOM_SIEVE << 28) | 2u);
        const double probability_chosen = orc_u53(w.w[0], w.w[1]);                                     /* :2303 rdist.ZeroToOne(rgen) */
        if (probability_chosen >= thr[2 * len + 1]) {                                                  /* :2304, Simulator.h:418-420 */
            out[n].len = len;
            out[n].probability_chosen = probability_chosen;
            ++n;
        }
    }
    return n;
}

static uint64_t sieve_blocks(const orc_sim *s, uint32_t block_lo, uint32_t block_hi, orc_fragment **out, int literal);
uint64_t orc_sieve_blocks(const orc_sim *s, uint32_t block_lo, uint32_t block_hi, orc_fragment **out) { return sieve_blocks(s, block_lo, block_hi, out, 0); }
/* the reference's own loop shape, cell by cell: what `cpu_baseline` times as the reference-shaped single-thread figure and what the two-sample tests hold the gap route against */
uint64_t orc_sieve_blocks_literal(const orc_sim *s, uint32_t block_lo, uint32_t block_hi, orc_fragment **out) { return sieve_blocks(s, block_lo, block_hi, out, 1); }
static uint64_t sieve_blocks(const orc_sim *s, uint32_t block_lo, uint32_t block_hi, orc_fragment **out, int literal) {
    const orc_profile *p = s->p;
    const orc_reference *r = s->r;
    size_t cap = 1024, n = 0;
    orc_fragment *f = malloc(cap * sizeof *f);
    uint32_t to = s->insert_to;
    double *gap_q = malloc(sizeof(double) * to);
    uint32_t *gap_seg_end = malloc(sizeof(uint32_t) * to);
    orc_gap_hit *passing = malloc(sizeof(orc_gap_hit) * to);
    for (uint32_t seq = 0; seq < r->n_seqs; ++seq) {
        if (!s->n_blocks[seq]) continue;
        const uint8_t *codes = r->codes[seq];
        uint32_t L = r->len[seq];
        const double *thr = &s->thresholds[(size_t)s->coverage_groups[seq] * to * 2];
        orc_gap_table(s, s->coverage_groups[seq], gap_q, gap_seg_end);
        for (uint32_t b = 0; b < s->n_blocks[seq]; ++b) {
            uint32_t block_id = s->first_block[seq] + b;
            if (block_id < block_lo || block_id >= block_hi) continue;
            uint32_t block_start = b * BLOCK_SIZE, read_number = 0;
            int32_t sur_start[3];
            orc_surrounding_forward(codes, L, 0 < block_start ? block_start - 1 : L - 1, sur_start);      /* :2288 */
            for (uint32_t start = block_start; start < block_start + BLOCK_SIZE && start < L; ++start) {
                orc_surrounding_update_forward(codes, L, start, sur_start);
                uint32_t last_gc = 0, last_gc_end = start;                                                  /* :1696-1697 */
                const uint32_t n_passing = literal ? orc_literal_hits(s, thr, start, seq, passing)             /* the lengths whose cell passes :2304-2306 */
                                                   : orc_gap_hits(s, gap_q, gap_seg_end, thr, start, seq, passing);
                for (uint32_t h = 0; h < n_passing; ++h) {
                    const uint32_t len = passing[h].len;
                    const double probability_chosen = passing[h].probability_chosen;
                    uint16_t non_zero_strands = orc_binomial(2, 1 - thr[2 * len], probability_chosen);     /* :2307, FDS.cpp:3598 */
                    if (!non_zero_strands) continue;
                    orc_philox_out w2 = orc_philox4x32_10(s->seed, start, seq, len, ((uint32_t)ORC_DOM_SIEVE << 28) | 1u);
                    uint16_t chosen[2];
                    uint32_t n_chosen = 0;
                    uint8_t reverse_selection[2] = {1, 1};
                    if (non_zero_strands <= 1) orc_select_allele(chosen, &n_chosen, reverse_selection, 2, orc_u32(w2.w[2]));  /* :1387-1391 */
                    else {                                                                                  /* :1392-1396: complement of nothing */
                        chosen[0] = 0;
                        chosen[1] = 1;
                        n_chosen = 2;
                    }
                    for (uint32_t j = 0; j < n_chosen; ++j) {
                        uint8_t strand = chosen[j] % 2;
                        uint32_t end = start + len;
                        if (!(end < L)) continue;                                                           /* :2318 */
                        if (end < last_gc_end) {                                                            /* :1858-1873 GetGCPercent */
                            /* cannot happen for increasing lengths; kept for fidelity */
                        }
                        for (uint32_t i = last_gc_end; i < end; ++i)
                            if (is_gc(codes[i])) ++last_gc;
                        last_gc_end = end;
                        uint8_t gc_perc = orc_percent_u32(last_gc, len);
                        int32_t sur_end[3];
                        orc_surrounding_reverse(codes, L, end - 1, sur_end);                               /* :1820-1832 */
                        double adjusted_random = thr[2 * len] + orc_u53(w2.w[2 * j], w2.w[2 * j + 1]) * (1 - thr[2 * len]);   /* :2322 */
                        uint16_t counts = orc_get_fragment_counts(p, s->bias_normalization, s->ref_seq_bias[seq], len, gc_perc, sur_start, sur_end,
                                                                  adjusted_random, 1);
                        for (uint16_t dup = 0; dup < counts; ++dup) {
                            if (n == cap) f = realloc(f, (cap *= 2) * sizeof *f);
                            orc_fragment fr = {seq, start, len, dup, strand, 0, block_id, ++read_number};
                            f[n++] = fr;
                        }
                    }
                }
            }
        }
    }
    free(gap_q);
    free(gap_seg_end);
    free(passing);
    *out = f;
    return n;
}

/* -------------------------------------------------------------------- FillRead */
typedef struct {
    uint16_t read_length, read_pos;
    uint8_t previous_indel_type;
    uint16_t indel_pos, base_call, gc_seq;
    uint8_t seq_qual, qual, error_rate;
    uint16_t num_errors;
} fill_par;                                                          /* Simulator.h:215-240 ReadFillParameter */

typedef struct {
    char *buf;
    size_t len;
} cigar_buf;
static void cigar_append(cigar_buf *c, char op, uint32_t count) { c->len += (size_t)sprintf(c->buf + c->len, "%u%c", count, op); }

typedef struct {
    const orc_sim *s;
    const orc_stream *st;
    uint32_t iteration;
} draw_ctx;
static orc_philox_out stream_words(const draw_ctx *d, uint32_t step) {
    return orc_philox4x32_10(d->st->seed, d->st->c0, d->st->c1, d->st->c2, d->st->c3base | step);
}

/* the systematic errors of consecutive template bases without variants: GetSysErrorFromBlock's last branch (:286-291) */
typedef struct {
    const uint8_t *dom, *rate;
    uint32_t pos;
} flat_cursor;
static void flat_reset(void *ctx) { ((flat_cursor *)ctx)->pos = 0; }
static void flat_next(void *ctx, uint8_t *dom, uint8_t *rate) {
    flat_cursor *c = ctx;
    *dom = c->dom[c->pos];
    *rate = c->rate[c->pos];
    ++c->pos;
}
static void flat_deleted(void *ctx, uint8_t *rate) {
    flat_cursor *c = ctx;
    *rate = c->rate[c->pos];
    ++c->pos;
}

/* Simulator.cpp:294-452.  sys == NULL selects the adapter's systematic errors. */
static void fill_read_part(draw_ctx *d, orc_read *rd, cigar_buf *cg, uint8_t seg, uint16_t tile_id, const uint8_t *org, uint32_t org_len, uint32_t org_pos,
                           char base_cigar_element, const orc_sys_cursor *sys, uint32_t adapter_id, fill_par *par) {
    const orc_profile *p = d->s->p;
    uint32_t nt = p->n_tiles;
    uint16_t cigar_element_length = 0;
    char cigar_element = base_cigar_element;
    uint8_t dom_error = 0;
    while (par->read_pos < par->read_length && org_pos < org_len) {
        orc_philox_out w = stream_words(d, 2u + d->iteration++);
        double prob_sum;
        uint32_t idx_indel[3] = {par->indel_pos, par->read_pos, par->gc_seq};
        uint32_t indel = orc_draw(&p->indels[par->previous_indel_type][par->base_call], idx_indel, orc_u32(w.w[0]), &prob_sum);
        if (0.0 == prob_sum) indel = 0;
        const orc_table *qt = &p->quality[((size_t)seg * nt + tile_id) * 4 + org[org_pos]];
        if (0 == indel) {
            if (sys) sys->next(sys->ctx, &dom_error, &par->error_rate);   /* GetSysErrorFromBlock */
            else {
                dom_error = d->s->adapter_dom[seg][adapter_id][org_pos];
                par->error_rate = d->s->adapter_rate[seg][adapter_id][org_pos];
            }
            uint32_t idx_q[4] = {par->seq_qual, par->qual, par->read_pos, par->error_rate};
            par->qual = (uint8_t)orc_draw(qt, idx_q, orc_u32(w.w[1]), &prob_sum);
            if (0.0 == prob_sum) {
                if (par->read_pos) par->qual = (uint8_t)(rd->qual[par->read_pos - 1] - p->phred_offset);
                else par->qual = (uint8_t)orc_max_value(qt);
            }
            rd->qual[par->read_pos] = (uint8_t)(par->qual + p->phred_offset);
            uint32_t idx_b[4] = {par->qual, par->read_pos, par->num_errors, par->error_rate};
            par->base_call = (uint16_t)orc_draw(&p->base_call[(((size_t)seg * nt + tile_id) * 4 + org[org_pos]) * 5 + dom_error], idx_b, orc_u32(w.w[2]), &prob_sum);
            if (0.0 == prob_sum) par->base_call = org[org_pos];
            rd->seq[par->read_pos] = (uint8_t)par->base_call;
            if (base_cigar_element == cigar_element) ++cigar_element_length;
            else {
                cigar_append(cg, cigar_element, cigar_element_length);
                cigar_element = base_cigar_element;
                cigar_element_length = 1;
                par->indel_pos = 0;
                par->previous_indel_type = 0;
            }
            if (par->base_call != org[org_pos]) ++par->num_errors;
            ++par->read_pos;
            ++org_pos;
        } else if (1 == indel) {                                     /* kDeletion */
            if (sys) sys->deleted(sys->ctx, &par->error_rate);
            else par->error_rate = d->s->adapter_rate[seg][adapter_id][org_pos];
            if ('D' == cigar_element) {
                ++cigar_element_length;
                ++par->indel_pos;
            } else {
                cigar_append(cg, cigar_element, cigar_element_length);
                cigar_element = 'D';
                cigar_element_length = 1;
                par->indel_pos = 1;
                par->previous_indel_type = 1;
            }
            ++par->num_errors;
            ++org_pos;
        } else {                                                     /* insertion of base indel-2 */
            uint32_t idx_q[4] = {par->seq_qual, par->qual, par->read_pos, par->error_rate};
            uint32_t q = orc_draw(qt, idx_q, orc_u32(w.w[1]), &prob_sum);
            rd->qual[par->read_pos] = (uint8_t)(p->phred_offset + (0.0 == prob_sum ? par->qual : q));
            rd->seq[par->read_pos] = (uint8_t)(indel - 2);
            if ('I' == cigar_element) {
                ++cigar_element_length;
                ++par->indel_pos;
            } else {
                cigar_append(cg, cigar_element, cigar_element_length);
                cigar_element = 'I';
                cigar_element_length = 1;
                par->indel_pos = 1;
                par->previous_indel_type = 0;
            }
            ++par->num_errors;
            ++par->read_pos;
        }
    }
    if (cigar_element_length) cigar_append(cg, cigar_element, cigar_element_length);
}

/* Simulator.h:185-198 ReadLength */
static uint16_t draw_read_length(const orc_profile *p, uint8_t seg, uint32_t fragment_length, double u) {
    if (1 == p->read_lengths[seg].size) return (uint16_t)p->read_lengths[seg].from;
    const orc_rl_by_fl *r = &p->rl_by_fl[seg];
    double random_value = u * (double)vect_u64_at(&p->insert_lengths, fragment_length);
    double counter = 0.0;
    uint32_t row = (uint32_t)(fragment_length - r->from);
    uint16_t from = (uint16_t)r->row_from[row];
    uint16_t read_len = (uint16_t)(from + (r->row_ptr[row + 1] - r->row_ptr[row]));
    while (counter <= random_value && (read_len-- > from)) counter += (double)r->values[r->row_ptr[row] + (read_len - from)];
    return read_len;
}

/* Simulator.cpp:454-594 */
int orc_fill_read(const orc_sim *s, orc_read *rd, uint8_t seg, uint16_t tile_id, uint32_t fragment_length, const uint8_t *org, uint32_t org_len,
                  const uint8_t *sys_dom, const uint8_t *sys_rate, const orc_stream *st) {
    flat_cursor fc = {sys_dom, sys_rate, 0};
    orc_sys_cursor cur = {&fc, flat_reset, flat_next, flat_deleted};
    return orc_fill_read_cursor(s, rd, seg, tile_id, fragment_length, org, org_len, sys_dom ? &cur : NULL, st);
}

int orc_fill_read_cursor(const orc_sim *s, orc_read *rd, uint8_t seg, uint16_t tile_id, uint32_t fragment_length, const uint8_t *org, uint32_t org_len,
                         const orc_sys_cursor *sys, const orc_stream *st) {
    const orc_profile *p = s->p;
    uint32_t nt = p->n_tiles;
    draw_ctx d = {s, st, 0};
    fill_par par = {0, 0, 0, 0, 5, 0, 0, 1, 0, 0};
    orc_philox_out h0 = orc_philox4x32_10(st->seed, st->c0, st->c1, st->c2, st->c3base | 0u);
    orc_philox_out h1 = orc_philox4x32_10(st->seed, st->c0, st->c1, st->c2, st->c3base | 1u);
    par.read_length = draw_read_length(p, seg, fragment_length, orc_u32(h0.w[0]));
    rd->read_len = par.read_length;
    cigar_buf cg = {rd->cigar, 0};
    rd->cigar[0] = 0;
    const orc_adapters *ad = &p->adapters[seg];
    uint32_t adapter_id = 0;

    uint16_t seq_length = (uint16_t)(par.read_length < org_len ? par.read_length : org_len);
    uint32_t mean_error_rate = 0;
    if (seq_length) {
        for (uint16_t read_pos = seq_length; read_pos--;) {
            uint8_t dom_error, error_rate;
            if (is_gc(org[read_pos])) ++par.gc_seq;
            sys->next(sys->ctx, &dom_error, &error_rate);           /* GetSysErrorFromBlock on copies of the start state (:487-502) */
            mean_error_rate += error_rate;
        }
        sys->reset(sys->ctx);
        par.gc_seq = orc_percent_u16(par.gc_seq, seq_length);
        mean_error_rate = orc_divide_u32(mean_error_rate, seq_length);
    } else {
        adapter_id = discrete_draw(ad->adapter_cp, ad->n, orc_u32(h0.w[1]));
        uint16_t alen = (uint16_t)(ad->seq_ptr[adapter_id + 1] - ad->seq_ptr[adapter_id]);
        for (uint16_t read_pos = alen; read_pos--;) {
            if (is_gc(ad->seqs[ad->seq_ptr[adapter_id] + read_pos])) ++par.gc_seq;
            mean_error_rate += s->adapter_rate[seg][adapter_id][read_pos];
        }
        par.gc_seq = orc_percent_u16(par.gc_seq, alen);
        mean_error_rate = orc_divide_u32(mean_error_rate, alen);
    }
    double prob_sum;
    uint32_t idx_sq[3] = {par.gc_seq, mean_error_rate, fragment_length / SQ_FRAGLEN_BIN};
    const orc_table *sqt = &p->seq_quality[(size_t)seg * nt + tile_id];
    par.seq_qual = (uint8_t)orc_draw(sqt, idx_sq, orc_u32(h0.w[2]), &prob_sum);
    if (0.0 == prob_sum) par.seq_qual = (uint8_t)orc_most_likely(sqt);

    fill_read_part(&d, rd, &cg, seg, tile_id, org, org_len, 0, 'M', sys, 0, &par);

    if (par.read_pos < par.read_length) {
        if (0 == adapter_id) adapter_id = discrete_draw(ad->adapter_cp, ad->n, orc_u32(h1.w[1]));
        uint32_t adapter_pos = 0;
        if (0 == par.read_pos)
            adapter_pos = discrete_draw(ad->cut_cp[adapter_id], ad->cut_ptr[adapter_id + 1] - ad->cut_ptr[adapter_id], orc_u32(h0.w[3])) + ad->cut_from[adapter_id];
        fill_read_part(&d, rd, &cg, seg, tile_id, ad->seqs + ad->seq_ptr[adapter_id], ad->seq_ptr[adapter_id + 1] - ad->seq_ptr[adapter_id], adapter_pos, 'S', NULL,
                       adapter_id, &par);
        if (par.read_pos < par.read_length) {
            cigar_append(&cg, 'H', (uint32_t)(par.read_length - par.read_pos));
            const orc_table *q0 = &p->quality[((size_t)seg * nt + tile_id) * 4 + 0];
            uint32_t tail_length = discrete_draw(p->polya_cp, p->polya.size, orc_u32(h1.w[0])) + (uint32_t)p->polya.from;
            for (uint32_t pos_tail = 0; par.read_pos < par.read_length; ++pos_tail) {
                orc_philox_out w = stream_words(&d, 2u + d.iteration++);
                uint32_t idx_q[4] = {par.seq_qual, par.qual, par.read_pos, par.error_rate};
                par.qual = (uint8_t)orc_draw(q0, idx_q, orc_u32(w.w[1]), &prob_sum);
                if (0.0 == prob_sum && par.read_pos) par.qual = (uint8_t)(rd->qual[par.read_pos - 1] - p->phred_offset);
                rd->qual[par.read_pos] = (uint8_t)(par.qual + p->phred_offset);
                if (pos_tail < tail_length) rd->seq[par.read_pos++] = 0;                                   /* poly-A tail */
                else rd->seq[par.read_pos++] = (uint8_t)discrete_draw(p->overrun_cp, 4, orc_u32(w.w[3]));  /* random overrun base */
            }
        }
    }
    rd->num_errors = par.num_errors;
    return 0;
}

/* ----------------------------------------------------------------- CreateReads */
static void text_reserve(orc_text *t, size_t extra) {
    if (t->len + extra + 1 > t->cap) {
        t->cap = (t->len + extra + 1) * 2;
        t->data = realloc(t->data, t->cap);
    }
}
void orc_text_free(orc_text *t) {
    free(t->data);
    t->data = NULL;
    t->len = t->cap = 0;
}
void orc_text_append_record(orc_text *t, const char *id, const orc_read *rd) {
    static const char kBases[] = "ACGTN";
    size_t idl = strlen(id);
    text_reserve(t, idl + 2u * rd->read_len + 8);
    t->data[t->len++] = '@';
    memcpy(t->data + t->len, id, idl);
    t->len += idl;
    t->data[t->len++] = '\n';
    for (uint16_t i = 0; i < rd->read_len; ++i) t->data[t->len++] = kBases[rd->seq[i]];
    t->data[t->len++] = '\n';
    t->data[t->len++] = '+';
    t->data[t->len++] = '\n';
    memcpy(t->data + t->len, rd->qual, rd->read_len);
    t->len += rd->read_len;
    t->data[t->len++] = '\n';
    t->data[t->len] = 0;
}

static uint16_t draw_tile(const orc_profile *p, uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3base) {
    if (1 < p->n_tiles) return (uint16_t)discrete_draw(p->tile_cp, p->n_tiles, orc_u32(orc_philox4x32_10(seed, c0, c1, c2, c3base).w[0]));   /* Simulator.h:176-181 */
    return 0;
}

static uint32_t pair_c3(uint32_t dom, uint32_t strand, uint32_t segsel) { return (dom << 28) | (strand << 27) | (segsel << 25); }

/* cur_methylation_start of SimulateFromGivenBlock for a start position (Simulator.cpp:2273,2293-2297; CreateBlock :1214-1219 and the
 * per-position increment keep it at the first region that ends after the position) */
uint32_t orc_methylation_start(const orc_sim *s, uint32_t seq, uint32_t pos) {
    uint32_t i = 0;
    while (i < s->meth_n[seq] && s->meth_second[seq][i] <= pos) ++i;
    return i;
}

/* Simulator::CTConversion without variants (Simulator.cpp:1925-2002), variable widths as there (read_pos is a uint16_t).  Where the
 * reference would evaluate regions.at(-1) (a reverse template that begins before the first region) nothing is converted.  The
 * uniform of template position k: word k&3 of Philox block (start, seq, length, 7<<28 | reversed<<27 | k>>2). */
static void ct_conversion(const orc_sim *s, uint8_t *read, uint32_t length, uint32_t seq_id, uint32_t start_pos, uint32_t cur_methylation_start, int reversed,
                          uint32_t site_start, uint32_t site_len) {
    const uint32_t *first = s->meth_first[seq_id], *second = s->meth_second[seq_id];
    const double *conversion_rate = s->meth_rate[seq_id];
    int32_t n = (int32_t)s->meth_n[seq_id], cur_meth = (int32_t)cur_methylation_start;
    uint16_t read_pos = 0;
    uint32_t ref_pos = start_pos;
#define CONVERT()                                                                                                                              \
    if (1 == read[read_pos]) {                                                                                                                 \
        orc_philox_out w = orc_philox4x32_10(s->seed, site_start, seq_id, site_len, (7u << 28) | ((uint32_t)reversed << 27) | (read_pos >> 2)); \
        if (orc_u32(w.w[read_pos & 3]) < conversion_rate[cur_meth]) read[read_pos] = 3;                                                        \
    }
    if (reversed) {
        while (cur_meth < n && first[cur_meth] <= ref_pos) ++cur_meth;
        --cur_meth;
        if (cur_meth > 0 && second[cur_meth] <= ref_pos) {
            read_pos = (uint16_t)(read_pos + (ref_pos - (second[cur_meth] - 1)));
            ref_pos = second[cur_meth] - 1;
        }
        while (cur_meth > 0 && read_pos < length) {
            while (ref_pos >= first[cur_meth] && read_pos < length) {
                CONVERT();
                --ref_pos;
                ++read_pos;
            }
            if (--cur_meth > 0 && second[cur_meth] <= ref_pos) {
                read_pos = (uint16_t)(read_pos + (ref_pos - (second[cur_meth] - 1)));
                ref_pos = second[cur_meth] - 1;
            }
        }
    } else {
        if (cur_meth < n && first[cur_meth] > ref_pos) {
            read_pos = (uint16_t)(read_pos + (first[cur_meth] - ref_pos));
            ref_pos = first[cur_meth];
        }
        while (cur_meth < n && read_pos < length) {
            while (ref_pos < second[cur_meth] && read_pos < length) {
                CONVERT();
                ++ref_pos;
                ++read_pos;
            }
            if (++cur_meth < n) {
                read_pos = (uint16_t)(read_pos + (first[cur_meth] - ref_pos));
                ref_pos = first[cur_meth];
            }
        }
    }
#undef CONVERT
}

/* Simulator.cpp:634-721 + :596-632 + Reference.cpp:483-496 (GetOrgSeq :1916-1922) */
int orc_create_reads(const orc_sim *s, const orc_fragment *frags, uint64_t n, orc_text *r1, orc_text *r2) {
    const orc_profile *p = s->p;
    const orc_reference *r = s->r;
    orc_read *rd = malloc(2 * sizeof(orc_read));
    uint8_t *tmpl[2] = {malloc(2048), malloc(2048)};
    orc_text *dst[2] = {r1, r2};
    for (uint64_t i = 0; i < n; ++i) {
        const orc_fragment *f = &frags[i];
        const uint8_t *codes = r->codes[f->seq];
        uint32_t L = r->len[f->seq], start = f->start, end = f->start + f->len, strand = f->strand;
        uint32_t tlen[2];
        for (uint32_t which = 0; which < 2; ++which) {               /* which: 0 forward template, 1 reverse template */
            uint32_t sg = which ? !strand : strand;                  /* sim_reads.at(strand) is the forward one */
            uint64_t rl_to = p->read_lengths[sg].from + p->read_lengths[sg].size;
            uint32_t tl = (uint32_t)(f->len < rl_to + p->max_len_deletion ? f->len : rl_to + p->max_len_deletion);
            tlen[sg] = tl;
            if (!which) memcpy(tmpl[sg], codes + start, tl);
            else
                for (uint32_t k = 0; k < tl; ++k) tmpl[sg][k] = (uint8_t)(3 - codes[end - 1 - k]);
        }
        if (s->meth_n) {                                            /* CTConversion (Simulator.cpp:2219-2247): forward template, then reverse */
            uint32_t cur_methylation_start = orc_methylation_start(s, f->seq, start);
            ct_conversion(s, tmpl[strand], tlen[strand], f->seq, start, cur_methylation_start, 0, start, f->len);
            ct_conversion(s, tmpl[!strand], tlen[!strand], f->seq, end, cur_methylation_start, 1, start, f->len);
        }
        uint32_t c2 = f->len | ((uint32_t)f->dup << 16);
        uint16_t tile_id = draw_tile(p, s->seed, start, f->seq, c2, pair_c3(ORC_DOM_PAIR, strand, 2));
        for (uint32_t seg = 2; seg--;) {
            const uint8_t *sd, *sr;
            if (seg == strand) {                                     /* block.at(strand) = start_block */
                sd = s->sys_dom[0][f->seq] + start;
                sr = s->sys_rate[0][f->seq] + start;
            } else {                                                 /* reverse block, position end_block end - end */
                sd = s->sys_dom[1][f->seq] + (L - end);
                sr = s->sys_rate[1][f->seq] + (L - end);
            }
            orc_stream st = {s->seed, start, f->seq, c2, pair_c3(ORC_DOM_PAIR, strand, seg)};
            orc_fill_read(s, &rd[seg], (uint8_t)seg, tile_id, f->len, tmpl[seg], tlen[seg], sd, sr, &st);
        }
        uint32_t print_start = strand ? end : start + 1, print_end = strand ? start + 1 : end;
        for (uint32_t seg = 0; seg < 2; ++seg) {
            char id[8192];
            snprintf(id, sizeof id, "%s%u_%u:%u:%s:%u:%u:1337:1337 %s E%u", s->base_identifier, f->block, f->number, print_start, r->first_name[f->seq], print_end,
                     (unsigned)p->tiles[tile_id], rd[seg].cigar, (unsigned)rd[seg].num_errors);
            orc_text_append_record(dst[seg], id, &rd[seg]);
        }
    }
    free(tmpl[0]);
    free(tmpl[1]);
    free(rd);
    return 0;
}

/* Simulator.cpp:2359-2382 */
int orc_simulate_adapter_only_pairs(const orc_sim *s, orc_text *r1, orc_text *r2) {
    const orc_profile *p = s->p;
    orc_read *rd = malloc(2 * sizeof(orc_read));
    orc_text *dst[2] = {r1, r2};
    for (uint64_t i = 0; i < s->num_adapter_only_pairs; ++i) {
        uint16_t tile_id = draw_tile(p, s->seed, (uint32_t)i, 0xFFFFFFFFu, (uint32_t)(i >> 32), pair_c3(ORC_DOM_PAIR, 0, 2));
        for (uint32_t seg = 2; seg--;) {
            orc_stream st = {s->seed, (uint32_t)i, 0xFFFFFFFFu, (uint32_t)(i >> 32), pair_c3(ORC_DOM_PAIR, 0, seg)};
            orc_fill_read(s, &rd[seg], (uint8_t)seg, tile_id, 0, NULL, 0, NULL, NULL, &st);
        }
        for (uint32_t seg = 0; seg < 2; ++seg) {
            char id[8192];
            snprintf(id, sizeof id, "%s0_%llu:0:Adapter:0:%u:1337:1337 %s E%u", s->base_identifier, (unsigned long long)(i + 1), (unsigned)p->tiles[tile_id], rd[seg].cigar,
                     (unsigned)rd[seg].num_errors);
            orc_text_append_record(dst[seg], id, &rd[seg]);
        }
    }
    free(rd);
    return 0;
}

/* Simulator.cpp:2403-2512 ApplyErrorsAndQualityToFastaInput, header fields already parsed */
int orc_error_model_only(const orc_profile *p, uint64_t seed, uint64_t first_index, uint64_t n, uint32_t read_len, const uint8_t *seqs, const uint8_t *seg,
                         const uint32_t *frag_len, const uint8_t *dom, const uint8_t *rate, orc_read *out, uint16_t *tile_out) {
    orc_sim *s = orc_sim_new(p, NULL, seed, 0, 0.0, NULL);
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t idx = first_index + i;
        uint16_t tile_id = draw_tile(p, seed, (uint32_t)idx, (uint32_t)(idx >> 32), 0, pair_c3(ORC_DOM_ERRMODEL, 0, 2));
        orc_stream st = {seed, (uint32_t)idx, (uint32_t)(idx >> 32), 0, pair_c3(ORC_DOM_ERRMODEL, 0, seg[i])};
        orc_fill_read(s, &out[i], seg[i], tile_id, frag_len[i], seqs + (size_t)i * read_len, read_len, dom + (size_t)i * read_len, rate + (size_t)i * read_len, &st);
        if (tile_out) tile_out[i] = tile_id;
    }
    orc_sim_free(s);
    return 0;
}

/* ------------------------------------------------------- accessors for ctypes */
double orc_sim_bias_normalization(const orc_sim *s) { return s->bias_normalization; }
uint32_t orc_sim_n_groups(const orc_sim *s) { return s->n_groups; }
uint32_t orc_sim_insert_to(const orc_sim *s) { return s->insert_to; }
uint64_t orc_sim_total_pairs(const orc_sim *s) { return s->total_pairs; }
uint64_t orc_sim_adapter_only_pairs(const orc_sim *s) { return s->num_adapter_only_pairs; }
uint32_t orc_sim_total_blocks(const orc_sim *s) { return s->total_blocks; }
uint16_t orc_sim_gc_range(const orc_sim *s) { return s->sys_gc_range; }
const double *orc_sim_thresholds(const orc_sim *s) { return s->thresholds; }
const double *orc_sim_norm_by_len(const orc_sim *s) { return s->norm_by_len; }
const uint32_t *orc_sim_coverage_groups(const orc_sim *s) { return s->coverage_groups; }
const uint8_t *orc_sim_sys_dom(const orc_sim *s, int strand, uint32_t seq) { return s->sys_dom[strand][seq]; }
const uint8_t *orc_sim_sys_rate(const orc_sim *s, int strand, uint32_t seq) { return s->sys_rate[strand][seq]; }
const uint8_t *orc_sim_adapter_dom(const orc_sim *s, int seg, uint32_t id) { return s->adapter_dom[seg][id]; }
const uint8_t *orc_sim_adapter_rate(const orc_sim *s, int seg, uint32_t id) { return s->adapter_rate[seg][id]; }
const orc_table *orc_profile_table(const orc_profile *p, int family, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    uint32_t nt = p->n_tiles;
    switch (family) {
    case 0: return &p->quality[((size_t)a * nt + b) * 4 + c];
    case 1: return &p->seq_quality[(size_t)a * nt + b];
    case 2: return &p->base_call[(((size_t)a * nt + b) * 4 + c) * 5 + d];
    case 3: return &p->dom_error[a][b][c];
    case 4: return &p->error_rate[a][b];
    default: return &p->indels[a][b];
    }
}

const double *orc_sim_ref_seq_bias(const orc_sim *s) { return s->ref_seq_bias; }

/* UpdateRefSeqBias kFile (FragmentDistributionStats.cpp:3386-3495): one line per sequence "[>]identifier[ description]<sep>bias", the
 * bias is what follows the last blank or tab, the identifier ends at the first blank (or at that separator); at most one empty line,
 * at the end; every sequence of the reference must be found. */
static int read_ref_bias_file(const char *path, const orc_reference *r, double *bias, char *err, size_t err_cap) {
    FILE *f = path ? fopen(path, "rb") : NULL;
    if (!f) {
        if (err) snprintf(err, err_cap, "Unable to open reference bias file %s", path ? path : "(null)");
        return -1;
    }
    uint8_t *found = calloc(r->n_seqs ? r->n_seqs : 1, 1);
    for (uint32_t i = 0; i < r->n_seqs; ++i) bias[i] = 0.0;
    char *line = NULL;
    size_t cap = 0;
    ssize_t n;
    uint32_t nline = 0, errors = 0;
    int empty_line = 0;
    while ((n = getline(&line, &cap, f)) >= 0) {
        if (n && line[n - 1] == '\n') line[--n] = 0;
        if (empty_line) {
            ++errors;
            continue;
        }
        ++nline;
        if (0 == n) {
            empty_line = 1;
            continue;
        }
        ssize_t sep = -1;
        for (ssize_t k = n; k--;)
            if (line[k] == ' ' || line[k] == '\t') {
                sep = k;
                break;
            }
        if (sep < 0) {
            ++errors;
            continue;
        }
        char *end = NULL;
        double b = strtod(line + sep + 1, &end);
        if (end == line + sep + 1) {                                  /* stod throws: counted, bias 0 */
            ++errors;
            b = 0.0;
        }
        if (0.0 > b) ++errors;
        ssize_t id_len = -1;
        for (ssize_t k = 0; k < n; ++k)
            if (line[k] == ' ') {
                id_len = k;
                break;
            }
        if (id_len < 0) id_len = sep;
        ssize_t id_start = 0;
        if ('>' == line[0]) {
            ++id_start;
            --id_len;
        }
        for (uint32_t i = 0; i < r->n_seqs; ++i)                        /* unordered_map::emplace keeps the first of equal ids */
            if (strlen(r->first_name[i]) == (size_t)id_len && 0 == strncmp(r->first_name[i], line + id_start, (size_t)id_len)) {
                bias[i] = b;
                found[i] = 1;
                break;
            }
    }
    free(line);
    fclose(f);
    for (uint32_t i = 0; i < r->n_seqs; ++i)
        if (!found[i]) ++errors;
    free(found);
    if (errors) {
        if (err) snprintf(err, err_cap, "reference bias file %s: %u errors", path, errors);
        return -1;
    }
    return 0;
}

/* ------------------------------------------------------ systematic-error profile */
uint8_t orc_compress_sys_error_rate(uint8_t q_perc) {                /* Simulator.cpp:2569-2574 */
    if (86 < q_perc) q_perc -= (q_perc - 85) / 2;
    return q_perc;
}
uint8_t orc_expand_sys_error_rate(uint8_t error_rate) {               /* Simulator.h:329-332 */
    if (86 < error_rate) error_rate += error_rate - 86;
    return error_rate;
}

int orc_create_sys_error_profile(const orc_profile *p, const orc_reference *r, uint64_t seed, orc_text *out) {
    static const char kBases[] = "ACGTN";
    uint64_t reads = 0, sum_read_length = 0;                           /* sys_gc_range_ as Simulate sets it (Simulator.cpp:2713-2721,2782) */
    for (int seg = 2; seg--;)
        for (uint64_t len = p->read_lengths[seg].from; len < p->read_lengths[seg].from + p->read_lengths[seg].size; ++len) {
            reads += vect_u64_at(&p->read_lengths[seg], len);
            sum_read_length += vect_u64_at(&p->read_lengths[seg], len) * len;
        }
    if (!reads) return -1;
    uint16_t gc_range = (uint16_t)(((sum_read_length + reads / 2) / reads) / 2);
    uint8_t dom_state = 0;                                              /* a fresh Simulator: DominantBase() */
    for (uint32_t i = 0; i < r->n_seqs; ++i) {                          /* :2628-2646 every sequence, reverse strand first */
        uint32_t L = r->len[i];
        uint8_t *dom = malloc(L ? L : 1), *rate = malloc(L ? L : 1);
        for (int strand = 2; strand--;) {
            orc_systematic_errors(p, seed, i, (uint32_t)strand, r->codes[i], L, strand, gc_range, &dom_state, dom, rate);
            const char *label = strand ? " reverse" : " forward";
            size_t idl = strlen(r->full_name[i]) + strlen(label);
            text_reserve(out, idl + 2u * (size_t)L + 8);
            out->data[out->len++] = '@';
            memcpy(out->data + out->len, r->full_name[i], strlen(r->full_name[i]));
            memcpy(out->data + out->len + strlen(r->full_name[i]), label, strlen(label));
            out->len += idl;
            out->data[out->len++] = '\n';
            for (uint32_t k = 0; k < L; ++k) out->data[out->len++] = kBases[dom[k] < 4 ? dom[k] : 4];
            out->data[out->len++] = '\n';
            out->data[out->len++] = '+';
            out->data[out->len++] = '\n';
            for (uint32_t k = 0; k < L; ++k) out->data[out->len++] = (char)(orc_compress_sys_error_rate(rate[k]) + 33);
            out->data[out->len++] = '\n';
            out->data[out->len] = 0;
        }
        free(dom);
        free(rate);
    }
    return 0;
}

int orc_sim_load_sys_errors(orc_sim *s, const char *text, size_t len, char *err, size_t err_cap) {
    const orc_reference *r = s->r;
    size_t pos = 0;
    for (uint32_t i = 0; i < r->n_seqs; ++i) {
        if (!s->n_blocks[i]) continue;                                   /* no unit, no LoadSysErrorRecord: the file's records are NOT skipped */
        for (int strand = 2; strand--;) {                                /* CreateUnit loads the reverse record, the forward one follows */
            const char *line[4];
            size_t line_len[4];
            for (int k = 0; k < 4; ++k) {
                if (pos >= len) {
                    if (err) snprintf(err, err_cap, "Could not read systematic error profile for reference sequence '%s': end of file", r->first_name[i]);
                    return -1;
                }
                const char *e = memchr(text + pos, '\n', len - pos);
                size_t l = e ? (size_t)(e - (text + pos)) : len - pos;
                line[k] = text + pos;
                line_len[k] = l;
                pos += l + 1;
            }
            if (line[0][0] != '@' || line[2][0] != '+' || line_len[1] != line_len[3]) {
                if (err) snprintf(err, err_cap, "systematic error profile: malformed FASTQ record");
                return -1;
            }
            if (line_len[1] != r->len[i]) {                              /* Simulator.cpp:762-766 */
                if (err)
                    snprintf(err, err_cap, "Systematic error profile '%.*s' (length %zu) does not match reference sequence '%s' (length %u). Wrong file or order incorrect?",
                             (int)(line_len[0] - 1), line[0] + 1, line_len[1], r->first_name[i], r->len[i]);
                return -1;
            }
            for (uint32_t k = 0; k < r->len[i]; ++k) {                   /* ReadSystematicErrors */
                char c = line[1][k];
                s->sys_dom[strand][i][k] = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
                s->sys_rate[strand][i][k] = orc_expand_sys_error_rate((uint8_t)(line[3][k] - 33));
            }
        }
    }
    if (s->var_state) {                                                  /* the variants' errors are drawn after the records are read (CreateUnit / CreateBlock) */
        const orc_variants *vs = s->variants_source;
        orc_var_detach(s);
        if (orc_var_attach(s, vs)) {
            if (err) snprintf(err, err_cap, "%s", orc_var_last_error());
            return -1;
        }
    }
    return 0;
}

/* Reference::PrepareMethylationFile + ReadMethylation (Reference.cpp:1132-1310).  first/second: [seq][region]; rate: [seq][allele
 * column][region] with n_cols[seq] columns (1 or num_alleles_ref).  0, or -1 with the reference's message. */
typedef struct {
    uint32_t *n, *n_cols;
    uint32_t **first, **second;
    double ***rate;
} orc_methylation;

static void methylation_free(orc_methylation *m, uint32_t n_seqs) {
    if (!m->n) return;
    for (uint32_t i = 0; i < n_seqs; ++i) {
        free(m->first[i]);
        free(m->second[i]);
        if (m->rate[i]) {
            for (uint32_t a = 0; a < m->n_cols[i]; ++a) free(m->rate[i][a]);
            free(m->rate[i]);
        }
    }
    free(m->n);
    free(m->n_cols);
    free(m->first);
    free(m->second);
    free(m->rate);
    memset(m, 0, sizeof *m);
}

static int parse_methylation(const char *path, const orc_reference *r, uint32_t num_alleles_ref, orc_methylation *m, char *err, size_t err_cap) {
#define FAIL(...)                                      \
    do {                                               \
        if (err) snprintf(err, err_cap, __VA_ARGS__);  \
        free(line);                                    \
        if (f) fclose(f);                              \
        methylation_free(m, r->n_seqs);                \
        return -1;                                     \
    } while (0)
    char *line = NULL;
    size_t cap = 0;
    ssize_t len;
    memset(m, 0, sizeof *m);
    FILE *f = fopen(path, "rb");
    if (!f) FAIL("Unable to open methylation file %s", path);
    int have = 0;
    while ((len = getline(&line, &cap, f)) >= 0) {                     /* :1139-1150 skip empty and track lines */
        if (len && line[len - 1] == '\n') line[--len] = 0;
        if (len && strncmp(line, "track", 5)) {
            have = 1;
            break;
        }
    }
    if (!have) FAIL("Methylation file is empty or only contains track lines: %s", path);
    m->n = calloc(r->n_seqs, sizeof(uint32_t));
    m->n_cols = calloc(r->n_seqs, sizeof(uint32_t));
    m->first = calloc(r->n_seqs, sizeof(uint32_t *));
    m->second = calloc(r->n_seqs, sizeof(uint32_t *));
    m->rate = calloc(r->n_seqs, sizeof(double **));
    char cur_seq[1024];
    size_t sl = strcspn(line, " \t");
    snprintf(cur_seq, sizeof cur_seq, "%.*s", (int)sl, line);
    int eof = 0;
    for (uint32_t i = 0; i < r->n_seqs && !eof; ++i) {
        if (strcmp(r->first_name[i], cur_seq)) continue;                /* no entries for this sequence */
        uint32_t n = 0, room = 0, num_alleles = num_alleles_ref;        /* :1184-1185 */
        m->n_cols[i] = num_alleles_ref;
        m->rate[i] = calloc(num_alleles_ref, sizeof(double *));
        for (;;) {
            char *q = line + strlen(cur_seq);
            q += strspn(q, " \t");
            char *e;
            long long v = strtoll(q, &e, 10);
            if (e == q) FAIL("Could not convert second field to int for line:\n%s", line);
            if (!n) {
                if (v < 0) FAIL("Second field is negative in line:\n%s", line);
            } else if (v < (long long)m->second[i][n - 1]) FAIL("Region is overlapping with previous region[%u - %u] in line:\n%s", m->first[i][n - 1], m->second[i][n - 1], line);
            if (v >= (long long)r->len[i]) FAIL("Second field is larger than sequence length:\n%s", line);
            uint32_t region_start = (uint32_t)v;
            q = e + strspn(e, " \t");
            v = strtoll(q, &e, 10);
            if (e == q) FAIL("Could not convert third field to int for line:\n%s", line);
            if (v <= (long long)region_start) FAIL("Third field is smaller than second field in line:\n%s", line);
            if (v > (long long)r->len[i]) FAIL("Third field is larger than sequence length:\n%s", line);
            if (n == room) {
                room = room ? 2 * room : 16;
                m->first[i] = realloc(m->first[i], room * sizeof(uint32_t));
                m->second[i] = realloc(m->second[i], room * sizeof(uint32_t));
                for (uint32_t a = 0; a < m->n_cols[i]; ++a) m->rate[i][a] = realloc(m->rate[i][a], room * sizeof(double));
            }
            m->first[i][n] = region_start;
            m->second[i][n] = (uint32_t)v;
            uint32_t allele = 0;
            q = e + strspn(e, " \t");
            while (*q) {
                if (allele >= num_alleles) {
                    if (allele >= num_alleles_ref) FAIL("More alleles specified than in variant file [%u] in line:\n%s", num_alleles_ref, line);
                    FAIL("More alleles specified than in last line [%u] in line:\n%s", num_alleles, line);
                }
                double d = strtod(q, &e);
                if (e == q) FAIL("Could not convert field %u to double for line:\n%s", 4 + allele, line);
                if (0.0 > d || d > 1.0) FAIL("Field %u is not between 0 and 1:\n%s", 4 + allele, line);
                m->rate[i][allele++][n] = 1.0 - d;                      /* the probability of a C->T conversion */
                q = e + strspn(e, " \t");
            }
            if (0 == n && allele >= 1) {                                /* first entry: one column or one per allele (:1263-1272) */
                if (1 != allele && num_alleles_ref != allele) FAIL("%u alleles specified (must be either 1 or same as in variant file[%u]) in line:\n%s", allele, num_alleles_ref, line);
                num_alleles = allele;
                for (uint32_t a = allele; a < m->n_cols[i]; ++a) free(m->rate[i][a]);
                m->n_cols[i] = allele;
            } else if (num_alleles != allele) FAIL("%u alleles specified (must be either identical in all lines of a sequence [%u]) in line:\n%s", allele, num_alleles, line);
            ++n;
            m->n[i] = n;
            int got = 0;
            while ((len = getline(&line, &cap, f)) >= 0) {              /* ignore all empty lines */
                if (len && line[len - 1] == '\n') line[--len] = 0;
                if (len) {
                    got = 1;
                    break;
                }
            }
            if (!got) {
                eof = 1;
                break;
            }
            sl = strcspn(line, " \t");
            if (sl != strlen(cur_seq) || strncmp(line, cur_seq, sl)) {      /* a new reference sequence */
                snprintf(cur_seq, sizeof cur_seq, "%.*s", (int)sl, line);
                break;
            }
        }
    }
    free(line);
    fclose(f);
    return 0;
#undef FAIL
}

int orc_sim_read_methylation(orc_sim *s, const char *path, char *err, size_t err_cap) {
    orc_methylation m;
    if (parse_methylation(path, s->r, s->num_alleles ? s->num_alleles : 1, &m, err, err_cap)) return -1;
    const uint32_t n_seqs = s->r->n_seqs;
    s->meth_n = m.n;
    s->meth_first = m.first;
    s->meth_second = m.second;
    s->meth_rate = calloc(n_seqs, sizeof(double *));
    for (uint32_t i = 0; i < n_seqs; ++i) s->meth_rate[i] = m.rate[i] ? m.rate[i][0] : NULL;
    s->meth_rate_cols = m.rate;
    s->meth_n_cols = m.n_cols;
    return 0;
}

/* parse only, for pinning against ReferenceTest::TestMethylationLoading: rate_out is [sum of regions][num_alleles] where a sequence
 * with a single column repeats it for every allele (Reference::Unmethylation, Reference.h:390-397) */
int orc_parse_methylation(const char *path, const orc_reference *r, uint32_t num_alleles, uint32_t *n_regions, uint32_t *first_out, uint32_t *second_out,
                          double *rate_out, uint32_t cap, char *err, size_t err_cap) {
    orc_methylation m;
    if (parse_methylation(path, r, num_alleles, &m, err, err_cap)) return -1;
    uint32_t at = 0;
    for (uint32_t i = 0; i < r->n_seqs; ++i) {
        n_regions[i] = m.n[i];
        for (uint32_t k = 0; k < m.n[i] && at < cap; ++k, ++at) {
            first_out[at] = m.first[i][k];
            second_out[at] = m.second[i][k];
            for (uint32_t a = 0; a < num_alleles; ++a) rate_out[(size_t)at * num_alleles + a] = m.rate[i][1 < m.n_cols[i] ? a : 0][k];
        }
    }
    methylation_free(&m, r->n_seqs);
    return 0;
}

/* ------------------------------------------------------------------ variants */
int orc_variant_in_allele(const orc_variant *v, uint32_t allele) { return (int)((v->allele[allele / 64] >> (allele % 64)) & 1); }   /* Reference.h:38-40 */
uint32_t orc_variant_first_allele(const orc_variant *v) {                /* Reference.h:42-58 */
    uint32_t first = 0;
    for (int b = 0; b < 2; ++b) {
        if (v->allele[b]) {
            while (!orc_variant_in_allele(v, first)) ++first;
            return first;
        }
        first += 64;
    }
    return 0;
}
orc_variants *orc_variants_new(uint32_t n_seqs) {
    orc_variants *vs = calloc(1, sizeof *vs);
    vs->num_alleles = 1;
    vs->n_seqs = n_seqs;
    vs->n = calloc(n_seqs ? n_seqs : 1, sizeof(uint32_t));
    vs->v = calloc(n_seqs ? n_seqs : 1, sizeof(orc_variant *));
    return vs;
}
void orc_variants_free(orc_variants *vs) {
    if (!vs) return;
    for (uint32_t s = 0; s < vs->n_seqs; ++s) {
        for (uint32_t i = 0; i < vs->n[s]; ++i) free(vs->v[s][i].var_seq);
        free(vs->v[s]);
    }
    free(vs->n);
    free(vs->v);
    free(vs);
}
/* Reference::InsertVariant (Reference.h:115-139): at one position deletion, substitution, insertions by length */
void orc_insert_variant(orc_variants *vs, uint32_t seq, uint32_t position, const uint8_t *var_seq, uint32_t len, const uint64_t allele[2]) {
    orc_variant *list = vs->v[seq];
    uint32_t n = vs->n[seq], insert_at = n, var = n;
    while (0 < var && list[--var].position == position) {
        if (list[var].len == len && 0 == memcmp(list[var].var_seq, var_seq, len)) {      /* already in: only adjust the alleles */
            list[var].allele[0] |= allele[0];
            list[var].allele[1] |= allele[1];
            return;
        } else if (list[var].len > len) --insert_at;
    }
    list = realloc(list, (n + 1) * sizeof *list);
    memmove(list + insert_at + 1, list + insert_at, (n - insert_at) * sizeof *list);
    list[insert_at].position = position;
    list[insert_at].len = len;
    list[insert_at].var_seq = malloc(len ? len : 1);
    memcpy(list[insert_at].var_seq, var_seq, len);
    list[insert_at].allele[0] = allele[0];
    list[insert_at].allele[1] = allele[1];
    vs->v[seq] = list;
    vs->n[seq] = n + 1;
}

static uint8_t dna5_code(char c) { return c == 'A' || c == 'a' ? 0 : c == 'C' || c == 'c' ? 1 : c == 'G' || c == 'g' ? 2 : c == 'T' || c == 't' ? 3 : 4; }
static uint32_t split_tabs(char *line, char **field, uint32_t max_fields) {
    uint32_t n = 0;
    field[n++] = line;
    for (char *c = line; *c && n < max_fields; ++c)
        if (*c == '\t') {
            *c = 0;
            field[n++] = c + 1;
        }
    return n;
}

/* PrepareVariantFile (CheckVcf), ReadFirstVariants, ReadVariants over the whole file; NULL + first messages on any error */
orc_variants *orc_read_variants(const char *path, const orc_reference *r, char *err, size_t err_cap) {
    FILE *f = fopen(path, "rb");
    if (!f) {
        if (err) snprintf(err, err_cap, "Could not open vcf file '%s'.", path);
        return NULL;
    }
    char *line = NULL;
    size_t cap = 0;
    ssize_t len;
    char **contigs = NULL;
    uint32_t n_contigs = 0, errors = 0;
    int have_record = 0;
    size_t err_at = 0;
    if (err && err_cap) err[0] = 0;
#define ERROR(...)                                                                              \
    do {                                                                                        \
        if (errors++ < 20 && err && err_at < err_cap) err_at += (size_t)snprintf(err + err_at, err_cap - err_at, __VA_ARGS__); \
    } while (0)
    while ((len = getline(&line, &cap, f)) >= 0) {
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        if (0 == strncmp(line, "##", 2)) {
            if (0 == strncmp(line, "##contig=<ID=", 13)) {
                size_t e = strcspn(line + 13, ",>");
                contigs = realloc(contigs, (n_contigs + 1) * sizeof *contigs);
                contigs[n_contigs] = malloc(e + 1);
                memcpy(contigs[n_contigs], line + 13, e);
                contigs[n_contigs++][e] = 0;
            }
            continue;
        }
        if (!len || line[0] == '#') continue;
        have_record = 1;
        break;
    }
    if (n_contigs != r->n_seqs) ERROR("Number of contigs does not match between reference(%u) and variant(%u) file. ", r->n_seqs, n_contigs);      /* CheckVcf :80-97 */
    for (uint32_t c = 0; c < (n_contigs < r->n_seqs ? n_contigs : r->n_seqs); ++c)
        if (strcmp(contigs[c], r->first_name[c])) ERROR("Contigs at position %u do not match between reference(%s) and variant(%s) file. ", c, r->first_name[c], contigs[c]);
    if (!errors && !have_record) ERROR("Vcf file '%s' has no records. ", path);
    orc_variants *vs = NULL;
    if (!errors) {
        vs = orc_variants_new(r->n_seqs);
        char *rec[4096];
        uint32_t nf = split_tabs(line, rec, 4096);
        if (nf < 10) ERROR("Could not read first vcf record. ");
        else {
            uint32_t A = 0;                                            /* ReadFirstVariants :1046-1058 */
            for (uint32_t g = 9; g < nf; ++g) {
                for (const char *c = rec[g]; *c && *c != ':'; ++c)
                    if (*c == '|' || *c == '/') ++A;
                ++A;
            }
            vs->num_alleles = A;
            if (A > 128) ERROR("Currently only 128 alleles are supported, but file has %u. ", A);
            uint32_t *allele = calloc(A ? A : 1, sizeof *allele);
            uint32_t old_ref_id = 0xFFFFFFFFu, start_pos = 0, end_pos = 0, read_for = 0;
            while (!errors || errors < 20) {
                uint32_t rid = 0;
                while (rid < n_contigs && strcmp(contigs[rid], rec[0])) ++rid;
                uint32_t begin_pos = (uint32_t)(atoll(rec[1]) - 1);
                int skip = 0;
                if (rid < read_for) {                                  /* :393-414 */
                    ERROR("Variant file is not properly position sorted. Found sequence id %u after id %u ", rid, read_for);
                    skip = 1;
                } else if (rid == read_for && old_ref_id != 0xFFFFFFFFu && begin_pos < start_pos) {
                    ERROR("Variant file is not properly position sorted. Found in sequence id %u position %u after position %u ", rid, begin_pos, start_pos);
                    skip = 1;
                } else read_for = rid;
                if (!skip) {
                    if (rid >= r->n_seqs) ERROR("Variant starting in reference sequence %u does not belong to an existing reference sequence. ", rid);       /* :149 */
                    else if (begin_pos >= r->len[rid]) ERROR("Variant starting in reference sequence %u at position %u starts after the end of the reference sequence. ", rid, begin_pos);
                    else {
                        start_pos = begin_pos;
                        if (old_ref_id == rid) {
                            if (start_pos < end_pos) ERROR("Variant starting in reference sequence %u at position %u overlaps with a previous variant. ", rid, start_pos);
                        } else old_ref_id = rid;
                        const char *ref = rec[3], *alt = rec[4];
                        uint32_t ref_len = (uint32_t)strlen(ref), alt_total = (uint32_t)strlen(alt);
                        end_pos = start_pos + ref_len;
                        int ref_n = 0, differs = end_pos > r->len[rid];
                        for (uint32_t k = 0; k < ref_len; ++k) {
                            if (dna5_code(ref[k]) > 3) ref_n = 1;
                            else if (!differs && dna5_code(ref[k]) != r->codes[rid][start_pos + k]) differs = 1;
                        }
                        if (ref_n) ERROR("Variant starting in reference sequence %u at position %u has an reference column containing ambiguous bases (e.g. N). ", rid, start_pos);
                        else if (differs) ERROR("The specified reference in vcf file '%s' is not identical with the specified reference sequence %u at position %u. ", ref, rid, start_pos);
                        int ok = 1;                                    /* genotypes :196-262 */
                        uint32_t cur_allele = 0;
                        for (uint32_t g = 9; g < nf && ok; ++g) {
                            if (cur_allele >= A) {
                                ERROR("Found to many alleles in genotype definition ");
                                ok = 0;
                                break;
                            }
                            uint32_t chosen_var = 0;
                            int column_ok = 1;
                            for (const char *c = rec[g]; *c && *c != ':'; ++c) {
                                if (*c == '|' || *c == '/') {
                                    if (cur_allele < A) allele[cur_allele] = chosen_var;
                                    ++cur_allele;
                                    chosen_var = 0;
                                } else if ('0' <= *c && *c <= '9') chosen_var = chosen_var * 10 + (uint32_t)(*c - 48);
                                else {
                                    ERROR("Unallowed character '%c' in genotype definition '%s' ", *c, rec[g]);
                                    column_ok = 0;
                                }
                            }
                            if (cur_allele >= A) {
                                ERROR("Found to many alleles in genotype definition ");
                                ok = 0;
                                break;
                            }
                            allele[cur_allele++] = column_ok ? chosen_var : 0;
                            ok = ok && column_ok;
                        }
                        if (ok && cur_allele < A) {
                            ERROR("Could not find enough alleles in genotype definition ");
                            ok = 0;
                        }
                        if (ok) {                                      /* :270-370 */
                            uint32_t alt_start_pos[4096], n_alt_total = 0, chosen_var = 1;
                            uint64_t gt_has_var[4096][2];
                            alt_start_pos[0] = 0;
                            for (uint32_t pos = 0; pos <= alt_total && n_alt_total < 4095; ++pos) {
                                int last = pos == alt_total;
                                if (!last && alt[pos] != ',') continue;
                                alt_start_pos[n_alt_total + 1] = pos + 1;
                                uint64_t bits[2] = {0, 0};
                                for (uint32_t a = A; a--;) {           /* backwards: allele 0 is the rightmost bit */
                                    bits[a / 64] <<= 1;
                                    if (allele[a] == chosen_var) ++bits[a / 64];
                                    else if (last && allele[a] > chosen_var) ERROR("Variant number %u does not exist for sequence id %u and position %u ", allele[a], rid, begin_pos);
                                }
                                gt_has_var[n_alt_total][0] = bits[0];
                                gt_has_var[n_alt_total][1] = bits[1];
                                ++n_alt_total;
                                ++chosen_var;
                            }
                            for (uint32_t pos = 0; pos < ref_len; ++pos)
                                for (uint32_t n_alt = 0; n_alt < n_alt_total; ++n_alt) {
                                    if (!(gt_has_var[n_alt][0] | gt_has_var[n_alt][1])) continue;
                                    uint32_t alt_len = alt_start_pos[n_alt + 1] - 1 - alt_start_pos[n_alt];
                                    uint8_t inserted[65536];
                                    uint32_t ilen = 0;
                                    if (pos + 1 == ref_len && pos + 1 < alt_len) {                     /* insertion */
                                        for (uint32_t k = alt_start_pos[n_alt] + pos; k < alt_start_pos[n_alt + 1] - 1 && ilen < sizeof inserted; ++k) inserted[ilen++] = dna5_code(alt[k]);
                                    } else if (pos < alt_len) {                                        /* base mutation */
                                        uint8_t b = dna5_code(alt[alt_start_pos[n_alt] + pos]);
                                        if (dna5_code(ref[pos]) == b) continue;
                                        inserted[ilen++] = b;
                                    }                                                                  /* else deletion */
                                    int has_n = 0;
                                    for (uint32_t k = 0; k < ilen; ++k) has_n |= inserted[k] > 3;
                                    if (has_n) ERROR("Variant starting in reference sequence %u at position %u has an alternative column containing ambiguous bases (e.g. N). ", rid, start_pos);
                                    else orc_insert_variant(vs, rid, start_pos + pos, inserted, ilen, gt_has_var[n_alt]);
                                }
                        }
                    }
                }
                int got = 0;
                while ((len = getline(&line, &cap, f)) >= 0) {
                    while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
                    if (len) {
                        got = 1;
                        break;
                    }
                }
                if (!got) break;
                nf = split_tabs(line, rec, 4096);
                if (nf < 10) {
                    ERROR("Could not read vcf record. ");
                    break;
                }
            }
            free(allele);
        }
    }
#undef ERROR
    free(line);
    fclose(f);
    for (uint32_t c = 0; c < n_contigs; ++c) free(contigs[c]);
    free(contigs);
    if (errors) {
        orc_variants_free(vs);
        return NULL;
    }
    return vs;
}
