/*
 * oracle.h -- CPU restatement of ReSeq's read-simulation hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in reseq_amd/ (the product) includes,
 * links or calls this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py load liboracle.so, and only as the checker.
 *
 * Every function cites the file:line of /root/reference (schmeing/ReSeq v1.1)
 * it restates.  Arithmetic follows the reference (double precision, same
 * operation order, no FMA contraction: build with -ffp-contract=off).  The one
 * deliberate difference is the random source: the reference draws from
 * std::mt19937_64 streams seeded per 1000-bp block (Simulator.cpp:2258); the
 * oracle and the product both draw from Philox4x32-10 keyed by the seed with
 * the counter layout of DESIGN.md section "Random streams", so that the two can
 * be compared bit for bit.
 *
 * PARITY PINNING.  The reference cannot be built in this image (every header on
 * the path needs Boost and a generated CMakeConfig.h, both absent), so the
 * oracle is pinned against the known answers the reference's own gtest suites
 * hold for this path (tests/golden/reference_known_answers.json: SurroundingTest,
 * FragmentDistributionStatsTest::TestDrawCounts, SimulatorTest, utilitiesTest).
 * The reference has NO test of LogArrayResult::Draw values, FillRead,
 * FillReadPart, CreateReads, SimulateFromGivenBlock output or FASTQ text
 * (SURVEY.md section 4): for those functions this oracle is "parity unpinned".
 */
#ifndef RESEQ_ORACLE_H
#define RESEQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- container */
typedef struct {
    char *name;
    int dtype;            /* 0 u8, 1 u16, 2 u32, 3 u64, 4 i32, 5 i64, 6 f64 */
    int ndim;
    uint64_t dims[4];
    uint64_t count;
    const void *data;
} orc_array;

typedef struct {
    uint8_t *buf;
    size_t size;
    uint32_t n;
    orc_array *arr;
} orc_container;

orc_container *orc_container_open(const char *path);
void orc_container_close(orc_container *c);
const orc_array *orc_container_get(const orc_container *c, const char *name);

/* ------------------------------------------------------------------- Philox */
typedef struct { uint32_t w[4]; } orc_philox_out;
orc_philox_out orc_philox4x32_10(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3);
double orc_u32(uint32_t w);                 /* w * 2^-32              */
double orc_u53(uint32_t hi, uint32_t lo);   /* ((hi<<32|lo)>>11)*2^-53 */

/* Counter domains (DESIGN.md "Random streams") */
enum { ORC_DOM_SIEVE = 1, ORC_DOM_PAIR = 2, ORC_DOM_SYSERR = 3, ORC_DOM_ERRMODEL = 4, ORC_DOM_REPLACEN = 5 };

/* --------------------------------------------------- LogArrayResult<N> table */
#define ORC_MAX_MARGINS 4
typedef struct {
    uint32_t k;                               /* par0_indeces_.size()       */
    uint32_t nm;                              /* N-1 margins                */
    const uint32_t *par0;                     /* par0_indeces_[k]           */
    uint32_t from[ORC_MAX_MARGINS];           /* limits_[n].first           */
    uint32_t to[ORC_MAX_MARGINS];             /* limits_[n].second          */
    const double *dim2[ORC_MAX_MARGINS];      /* dim2_[n][(to-from)*k]      */
} orc_table;

/* ProbabilityEstimates.h:481-508 */
uint32_t orc_draw(const orc_table *t, const uint32_t *index, double random_number, double *prob_sum);
uint32_t orc_max_value(const orc_table *t);     /* :510-517 */
uint32_t orc_most_likely(const orc_table *t);   /* :519-526 */

/* ------------------------------------------------------------------ profile */
typedef struct {
    uint64_t from, size;
    const uint64_t *v;
} orc_vect_u64;
typedef struct {
    uint64_t from, size;
    const double *v;
} orc_vect_f64;

typedef struct {
    uint32_t n;
    const uint8_t *seqs;
    const uint32_t *seq_ptr;
    const uint64_t *counts;
    const uint64_t *significant;
    const uint32_t *cut_ptr;
    const uint32_t *cut_from;
    const uint64_t *cut;
    double **cut_cp;           /* cumulative probabilities per adapter (discrete_distribution) */
    double *adapter_cp;        /* over significant counts */
} orc_adapters;

typedef struct {
    uint64_t from;             /* first fragment length with a row */
    uint32_t rows;
    const uint32_t *row_ptr;
    const uint32_t *row_from;
    const uint64_t *values;
} orc_rl_by_fl;

typedef struct orc_profile {
    orc_container *c;
    uint8_t phred_offset;
    double corrected_coverage;
    uint16_t max_len_deletion;
    uint32_t reset_distance;
    orc_vect_u64 read_lengths[2];
    orc_rl_by_fl rl_by_fl[2];
    uint64_t total_number_reads;
    uint32_t n_tiles;
    const uint16_t *tiles;
    const uint64_t *tile_abundance;
    double *tile_cp;
    orc_adapters adapters[2];
    orc_vect_u64 polya;
    double *polya_cp;
    const uint64_t *overrun_bases;      /* [5] */
    double overrun_cp[4];
    orc_vect_u64 insert_lengths;
    orc_vect_f64 insert_lengths_bias;
    orc_vect_f64 gc_bias;
    const double *sur_bias;             /* [3][1<<20] */
    double dispersion[2];
    uint32_t n_ref_bias;
    const double *ref_seq_bias;
    /* tables */
    orc_table *quality;        /* [2][n_tiles][4]      */
    orc_table *seq_quality;    /* [2][n_tiles]         */
    orc_table *base_call;      /* [2][n_tiles][4][5]   */
    orc_table dom_error[4][5][5];
    orc_table error_rate[4][5];
    orc_table indels[2][6];
    /* post-load edits own their storage */
    void **owned;
    size_t n_owned;
} orc_profile;

orc_profile *orc_profile_load(const char *path);
void orc_profile_free(orc_profile *p);
/* ProbabilityEstimates.h:1516-1549 */
void orc_profile_change_error_rate(orc_profile *p, double multiplier);
void orc_profile_remove_substitution_errors(orc_profile *p);
void orc_profile_remove_indel_errors(orc_profile *p);

/* ------------------------------------------------- small arithmetic helpers */
uint32_t orc_divide_u32(uint32_t nom, uint32_t den);              /* utilities.hpp:450-452 */
uint8_t orc_percent_u16(uint16_t nom, uint16_t den);              /* utilities.hpp:552-554, T = uint16_t */
uint8_t orc_percent_u32(uint32_t nom, uint32_t den);              /* T = uint32_t */
uint8_t orc_percent_u64(uint64_t nom, uint64_t den);              /* T = uint64_t */
uint8_t orc_safe_percent_u16(uint16_t nom, uint16_t den);         /* utilities.hpp:566-573 */
uint32_t orc_transform_distance(uint32_t dist);                   /* utilities.hpp:593-595 */
void orc_update_distances(uint32_t reset_distance, uint32_t *dist, uint8_t *start_rate, uint8_t error_rate); /* CoverageStats.cpp:379-396 */
double orc_inv_logit2(double bias);                               /* utilities.hpp:505-507 */

/* utilities.hpp:229-300 DominantBase over a code sequence (A=0..T=3,N=4) */
typedef struct { uint8_t dom_base; uint16_t content[5]; } orc_dominant_base;
void orc_dombase_clear(orc_dominant_base *d);
void orc_dombase_set(orc_dominant_base *d, const uint8_t *seq, uint32_t len, uint32_t cur_pos);
void orc_dombase_update(orc_dominant_base *d, uint8_t base, const uint8_t *seq, uint32_t len, uint32_t last_pos);

/* --------------------------------------------------------------- Surrounding */
/* SurroundingBase.hpp:64-81,196-226 ; Surrounding.h:17 (3 blocks x 10 bases, start 10 before) */
void orc_surrounding_forward(const uint8_t *seq, uint32_t len, uint32_t pos, int32_t sur[3]);
void orc_surrounding_reverse(const uint8_t *seq, uint32_t len, uint32_t pos, int32_t sur[3]);
void orc_surrounding_update_forward(const uint8_t *seq, uint32_t len, uint32_t new_pos, int32_t sur[3]);
void orc_surrounding_update_reverse(const uint8_t *seq, uint32_t len, uint32_t new_pos, int32_t sur[3]);
/* Surrounding.cpp:22-192: edits of the three 20-bit blocks for a variant inside the window (pos 0..29, base codes 0..3) */
void orc_sur_change_base(int32_t sur[3], uint32_t pos, uint8_t new_base);
void orc_sur_delete_shift_right(int32_t sur[3], uint32_t pos, uint8_t new_end_base);
void orc_sur_delete_shift_left(int32_t sur[3], uint32_t pos, uint8_t new_end_base);
void orc_sur_insert_shift_right(int32_t sur[3], uint32_t pos, const uint8_t *new_bases, uint32_t n);
void orc_sur_insert_shift_left(int32_t sur[3], uint32_t pos, const uint8_t *new_bases, uint32_t n);
/* Surrounding.cpp:194-260 */
void orc_combine_positions(const double *separated /*120*/, double *bias /*3<<20*/);
void orc_separate_positions(const double *bias, double *separated);
double orc_surrounding_bias(const double *bias, const int32_t sur[3]);   /* Surrounding.h:114-120 */

/* --------------------------------------------------- fragment-count drawing */
double orc_get_dispersion(double bias, double a, double b);                         /* FragmentDistributionStats.cpp:900-907 */
uint16_t orc_binomial(uint16_t n, double p, double probability_chosen);            /* :3584-3596 */
uint16_t orc_negative_binomial(double p, double r, double probability_chosen);     /* :3602-3613 */
double orc_calculate_non_zero_threshold(const double disp[2], double bias_normalization, double max_bias, uint16_t num_alleles); /* :2969-2976 */
uint16_t orc_get_fragment_counts(const orc_profile *p, double bias_normalization, double ref_seq_bias, uint32_t fragment_length,
                                 uint8_t gc, const int32_t sur_start[3], const int32_t sur_end[3], double probability_chosen,
                                 uint16_t num_alleles);                                 /* :3615-3627 */
uint16_t orc_fragment_counts_core(const double *sur_bias, const double disp[2], double bias_normalization, double ref_seq_bias, double insert_length_bias,
                                  double gc_bias, const int32_t sur_start[3], const int32_t sur_end[3], double probability_chosen, uint16_t num_alleles);
/* Simulator.cpp:1341-1361 ; chosen/n_chosen and reverse_selection are in/out */
void orc_select_allele(uint16_t *chosen, uint32_t *n_chosen, uint8_t *reverse_selection, uint16_t possible_strands, double random_value);

/* ---------------------------------------------------------------- reference */
typedef struct {
    uint32_t n_seqs;
    uint32_t *len;
    uint8_t **codes;          /* A=0,C=1,G=2,T=3 (N=4 only before replace_n) */
    char **first_name;        /* ReferenceIdFirstPart (Reference.cpp:476-480) */
    char **full_name;         /* ReferenceId */
} orc_reference;

orc_reference *orc_reference_new(uint32_t n_seqs);
void orc_reference_set(orc_reference *r, uint32_t i, const char *name, const uint8_t *codes, uint32_t len);
void orc_reference_free(orc_reference *r);
void orc_reference_replace_n(orc_reference *r, uint64_t seed);     /* Reference.cpp:813 (Philox instead of mt19937_64) */

/* ----------------------------------------------------------- simulation run */
typedef struct {
    uint32_t seq;
    uint32_t start;           /* forward start position (0-based)     */
    uint32_t len;             /* fragment length                       */
    uint16_t dup;             /* duplicate index within this site      */
    uint8_t strand;
    uint8_t pad;
    uint32_t block;           /* read-id block number                  */
    uint32_t number;          /* read_number within the block (1-based)*/
} orc_fragment;

typedef struct orc_sim {
    const orc_profile *p;
    const orc_reference *r;
    uint64_t seed;
    uint64_t total_pairs;             /* after removing adapter-only pairs */
    uint64_t num_adapter_only_pairs;
    uint16_t sys_gc_range;
    /* bias normalisation (a14) */
    double bias_normalization;
    uint32_t n_groups;
    uint32_t *coverage_groups;        /* [n_seqs]  */
    double *thresholds;               /* [n_groups][insert_to][2] */
    uint32_t insert_to;
    double *norm_by_len;              /* [insert_to] after interpolation */
    double *ref_seq_bias;             /* [n_seqs]  */
    /* systematic errors (a13): per sequence, forward and reverse-complement tracks */
    uint8_t **sys_dom[2];
    uint8_t **sys_rate[2];
    uint8_t **adapter_dom[2];         /* [seg][adapter][pos] */
    uint8_t **adapter_rate[2];
    /* block numbering: first forward block id per sequence (0 = sequence skipped) */
    uint32_t *first_block;
    uint32_t *n_blocks;
    uint32_t total_blocks;
    char base_identifier[64];
    /* unmethylated regions per sequence (NULL without --methylation): [first, second) and the C->T probability */
    uint32_t *meth_n;
    uint32_t **meth_first, **meth_second;
    double **meth_rate;               /* the first (or only) allele column */
    uint32_t *meth_n_cols;            /* columns per sequence: 1 or NumAlleles() */
    double ***meth_rate_cols;         /* [seq][column][region]; Reference::Unmethylation(seq, allele) = column allele, or 0 if there is one */
    /* variants (oracle_variants.cpp): NumAlleles() and the per-sequence variants with their systematic errors */
    uint16_t num_alleles;
    void *var_state;
    const void *variants_source;      /* the orc_variants the state was built from (kept to rebuild it after --readSysError) */
} orc_sim;

/* Simulator.cpp:2655-2898 up to "Starting read generation": pairs, thresholds, sys errors */
orc_sim *orc_sim_new(const orc_profile *p, const orc_reference *r, uint64_t seed, uint64_t num_read_pairs, double coverage,
                     const char *record_base_identifier);
/* the same with RefSeqBiasSimulation: 0 keep, 1 no, 2 draw, 3 file (FragmentDistributionStats.cpp:3352-3500); NULL + message in
 * err on a bias-file error */
orc_sim *orc_sim_new_bias(const orc_profile *p, const orc_reference *r, uint64_t seed, uint64_t num_read_pairs, double coverage,
                          const char *record_base_identifier, int ref_bias_mode, const char *ref_bias_file, char *err, size_t err_cap);
const double *orc_sim_ref_seq_bias(const orc_sim *s);
void orc_sim_free(orc_sim *s);

/* Simulator::CreateSystematicErrorProfile + WriteOutSystematicErrorProfile (Simulator.cpp:2562-2653): FASTQ text, two records per
 * sequence ("<id> reverse" first).  sys_gc_range_ has the value Simulate() would set (the reference leaves it uninitialised here). */
uint8_t orc_compress_sys_error_rate(uint8_t percent);              /* Simulator.cpp:2569-2574 */
uint8_t orc_expand_sys_error_rate(uint8_t stored);                 /* Simulator.h:329-332 */
/* replace the pre-pass results by externally supplied ones (stage-wise parity tests) */
/* the sieve's zero-threshold test drawn as gaps between passing lengths (oracle_sim.c): table of one coverage group, passing lengths of one start */
typedef struct {
    uint32_t len;
    double probability_chosen;
} orc_gap_hit;
void orc_gap_table(const orc_sim *s, uint32_t group, double *q /*[insert_to]*/, uint32_t *seg_end /*[insert_to]*/);
uint32_t orc_gap_hits(const orc_sim *s, const double *q, const uint32_t *seg_end, const double *thr, uint32_t start, uint32_t c1, orc_gap_hit *out /*[insert_to]*/);
void orc_sim_set_normalization(orc_sim *s, double bias_normalization, const double *thresholds /*[n_groups][insert_to][2]*/);

/* Simulator.cpp:61-114 */
double orc_coverage_prop_lost_from_adapters(const orc_profile *p);
uint64_t orc_coverage_to_number_pairs(double coverage, uint64_t total_ref_size, double average_read_length, double adapter_part);
double orc_number_pairs_to_coverage(uint64_t total_pairs, uint64_t total_ref_size, double average_read_length, double adapter_part);

/* Reference.cpp:622-659 SumBias with explicit bias tables */
double orc_sum_bias(const double *gc_bias, uint32_t gc_from, uint32_t gc_size, const double *sur_bias, const uint8_t *seq, uint32_t len, uint32_t fragment_length,
                    double general_bias, double *max_bias);

/* Simulator.cpp:2249-2357 without variants/methylation: the fragments of one block range, in
 * (block, start, length, chosen-strand order, duplicate) order.  Returns the count; *out is malloc'ed. */
uint64_t orc_sieve_blocks(const orc_sim *s, uint32_t block_lo, uint32_t block_hi, orc_fragment **out);
/* Simulator.cpp:2302-2306 as written -- one uniform per (start, fragment length) -- on a random stream of its own (oracle_sim.c) */
uint32_t orc_literal_hits(const orc_sim *s, const double *thr, uint32_t start, uint32_t c1, orc_gap_hit *out /*[insert_to]*/);
uint64_t orc_sieve_blocks_literal(const orc_sim *s, uint32_t block_lo, uint32_t block_hi, orc_fragment **out);

/* One simulated read (Simulator.cpp:454-594 FillRead + :294-452 FillReadPart). */
typedef struct {
    uint16_t read_len;
    uint16_t num_errors;
    uint8_t seq[1024];        /* base codes 0..4              */
    uint8_t qual[1024];       /* quality + phred offset       */
    char cigar[4096];         /* "150M" ...                   */
} orc_read;

/* stream: domain tag word (c3 high bits) and counter words c0..c2 identifying the read */
typedef struct { uint64_t seed; uint32_t c0, c1, c2, c3base; } orc_stream;

int orc_fill_read(const orc_sim *s, orc_read *out, uint8_t template_segment, uint16_t tile_id, uint32_t fragment_length,
                  const uint8_t *org_seq, uint32_t org_len, const uint8_t *sys_dom, const uint8_t *sys_rate,
                  const orc_stream *st);
/* the same with the systematic errors behind a cursor: next = GetSysErrorFromBlock (Simulator.cpp:240-292), deleted = the deletion
 * branch of FillReadPart (:380-392), reset = back to the read's first template base (FillRead walks the errors twice, :487-502) */
typedef struct {
    void *ctx;
    void (*reset)(void *ctx);
    void (*next)(void *ctx, uint8_t *dom_error, uint8_t *error_rate);
    void (*deleted)(void *ctx, uint8_t *error_rate);
} orc_sys_cursor;
int orc_fill_read_cursor(const orc_sim *s, orc_read *out, uint8_t template_segment, uint16_t tile_id, uint32_t fragment_length,
                         const uint8_t *org_seq, uint32_t org_len, const orc_sys_cursor *sys, const orc_stream *st);

/* Simulator.cpp:634-721 CreateReads + :596-632 CreateReadId: FASTQ text of both mates of the fragments. */
typedef struct { char *data; size_t len, cap; } orc_text;
int orc_create_reads(const orc_sim *s, const orc_fragment *frags, uint64_t n, orc_text *r1, orc_text *r2);
/* Simulator.cpp:2359-2382 */
int orc_simulate_adapter_only_pairs(const orc_sim *s, orc_text *r1, orc_text *r2);
void orc_text_free(orc_text *t);
void orc_text_append_record(orc_text *t, const char *id, const orc_read *rd);
/* Reference::Variant, InsertVariant, ReadFirstVariants / ReadVariants (Reference.h:24-62,115-139; Reference.cpp:126-420,1046-1077).
 * Loading only. */
typedef struct {
    uint32_t position;
    uint32_t len;
    uint8_t *var_seq;         /* base codes */
    uint64_t allele[2];
} orc_variant;
typedef struct {
    uint32_t num_alleles, n_seqs;
    uint32_t *n;              /* variants per sequence */
    orc_variant **v;
} orc_variants;
int orc_variant_in_allele(const orc_variant *v, uint32_t allele);
uint32_t orc_variant_first_allele(const orc_variant *v);
void orc_insert_variant(orc_variants *vs, uint32_t seq, uint32_t position, const uint8_t *var_seq, uint32_t len, const uint64_t allele[2]);
orc_variants *orc_variants_new(uint32_t n_seqs);
orc_variants *orc_read_variants(const char *path, const orc_reference *r, char *err, size_t err_cap);
void orc_variants_free(orc_variants *vs);

/* ---- the simulation with variants (oracle_variants.cpp, C++; the bookkeeping in oracle_variants.hpp) */
typedef struct {
    uint32_t seq, start, len;     /* cur_start_position, fragment_length */
    uint16_t dup;
    uint8_t strand, allele;
    uint32_t block, number;
    uint32_t end;                 /* cur_end_position = start + len + end_pos_shift_[allele] */
    uint32_t sub;                 /* pass of the do-while loop at this start position (> 0: the fragment starts inside inserted bases) */
    int32_t start_var;            /* bias_mod.StartVariant() */
    uint32_t start_var_pos;
    int32_t end_var;              /* bias_mod.EndVariant(variants, cur_end_position, allele) */
    uint32_t end_var_pos;
} orc_fragment_var;
orc_sim *orc_sim_new_variants(const orc_profile *p, const orc_reference *r, const orc_variants *vs, uint64_t seed, uint64_t num_read_pairs, double coverage,
                              const char *record_base_identifier, int ref_bias_mode, const char *ref_bias_file, char *err, size_t err_cap);
int orc_var_attach(orc_sim *s, const orc_variants *vs);
void orc_var_detach(orc_sim *s);
const char *orc_var_last_error(void);
/* var_errors_ of the err_variants_ entry of one variant on one strand (SetSystematicErrorVariantsForward / Reverse); returns its size */
uint32_t orc_var_sys_errors(const orc_sim *s, int strand, uint32_t seq, uint32_t var_id, uint8_t *dom, uint8_t *rate, uint32_t cap);
/* Simulator.cpp:2249-2357 with variants; UINT64_MAX on an error (orc_var_last_error) */
uint64_t orc_sieve_blocks_var(const orc_sim *s, uint32_t block_lo, uint32_t block_hi, orc_fragment_var **out);
int orc_create_reads_var(const orc_sim *s, const orc_fragment_var *frags, uint64_t n, orc_text *r1, orc_text *r2);
/* every evaluated (start, pass, length, allele) of the sieve calls since the last query: how many were re-derived from a fresh
 * VariantBiasVarModifiers, and how many of those differed from the incrementally updated one */
void orc_var_scratch_counters(uint64_t *checks, uint64_t *mismatches);
/* with the check enabled, every cell the sieve evaluates is also compared with the sequence that has the allele's variants applied (the
 * comparison SimulatorTest.cpp:116-195 makes): out[0] = cells compared, out[1..5] = mismatches of GC percent, start surrounding, end
 * surrounding, forward template, reverse template */
void orc_var_haplotype_check(int enable);
void orc_var_haplotype_counters(uint64_t *out);
/* utilitiesTest.cpp:61-137 DominantBaseWithMemory script: op 0 Clear, 1 Set(seq, arg), 2 Update(seq[arg]), 3 copy from the other object */
void orc_dombase_memory_script(const uint8_t *seq, uint32_t len, uint32_t n_ops, const uint8_t *which, const uint8_t *op, const uint32_t *arg, uint8_t *out);
/* SimulatorTest::TestVariationInSimulateFromGivenBlock driver */
void *orc_var_new(const uint8_t *codes, uint32_t n, const uint8_t *comp_codes, uint32_t n_comp, uint32_t n_var, const uint32_t *positions, const char *const *var_seqs,
                  const uint64_t *allele0);
void orc_var_free(void *h);
void orc_var_set_first_variant(void *h, int32_t id);
void orc_var_get_start(void *h, int32_t *first_variant_id, uint32_t *start_variant_pos);
int orc_var_prepare_start(void *h, uint32_t cur_start, uint32_t first_fragment_length, uint32_t comp_pos, uint32_t *sur_out, uint32_t *ref_sur_out, uint32_t *comp_sur_out);
int orc_var_inner_loop(void *h, uint32_t cur_start, uint32_t from, uint32_t to, const uint32_t *modified_start_pos, const int32_t *use_comp, int32_t *log,
                       uint32_t *n_possible);
int orc_var_check_inserted(void *h, uint32_t cur_start);

/* --methylation without variants: Reference::PrepareMethylationFile/ReadMethylation (Reference.cpp:1132-1310) and
 * Simulator::CTConversion (Simulator.cpp:1925-2002,2219-2247).  0, or -1 with the reference's message. */
int orc_sim_read_methylation(orc_sim *s, const char *path, char *err, size_t err_cap);
uint32_t orc_methylation_start(const orc_sim *s, uint32_t seq, uint32_t pos);      /* cur_methylation_start of SimulateFromGivenBlock at a start position */
int orc_parse_methylation(const char *path, const orc_reference *r, uint32_t num_alleles, uint32_t *n_regions, uint32_t *first_out, uint32_t *second_out,
                          double *rate_out, uint32_t cap, char *err, size_t err_cap);
int orc_create_sys_error_profile(const orc_profile *p, const orc_reference *r, uint64_t seed, orc_text *out);
/* --readSysError: LoadSysErrorRecord (Simulator.cpp:750-769) + ReadSystematicErrors (Simulator.h:326-335); 0 or -1 with a message */
int orc_sim_load_sys_errors(orc_sim *s, const char *text, size_t len, char *err, size_t err_cap);

/* Simulator.cpp:2403-2512 with the header already parsed: records as arrays. */
int orc_error_model_only(const orc_profile *p, uint64_t seed, uint64_t first_index, uint64_t n, uint32_t read_len,
                         const uint8_t *seqs, const uint8_t *seg, const uint32_t *frag_len, const uint8_t *dom,
                         const uint8_t *rate, orc_read *out, uint16_t *tile_out);

/* raw pieces for stage-wise tests */
void orc_systematic_errors(const orc_profile *p, uint64_t seed, uint32_t chain_c1, uint32_t chain_c2, const uint8_t *seq, uint32_t len,
                           int reverse_complement, uint16_t gc_range, uint8_t *dom_base_state /*in/out*/,
                           uint8_t *dom_out, uint8_t *rate_out);                    /* Simulator.h:337-382 */

#ifdef __cplusplus
}
#endif
#endif
