/*
 * oracle_base.c -- container reader, Philox4x32-10, arithmetic helpers,
 * LogArrayResult<N>::Draw and profile loading.  TEST INFRASTRUCTURE (see oracle.h).
 */
#include "oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ container */
static const size_t kDtypeSize[7] = {1, 2, 4, 8, 4, 8, 8};

static size_t pad8(size_t n) { return (8 - (n & 7)) & 7; }

orc_container *orc_container_open(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    orc_container *c = calloc(1, sizeof(*c));
    c->buf = malloc((size_t)sz + 8);
    c->size = (size_t)sz;
    if (fread(c->buf, 1, c->size, f) != c->size || c->size < 16 || memcmp(c->buf, "RSQPROF1", 8) != 0) {
        fclose(f);
        free(c->buf);
        free(c);
        return NULL;
    }
    fclose(f);
    uint32_t version;
    memcpy(&version, c->buf + 8, 4);
    memcpy(&c->n, c->buf + 12, 4);
    if (version != 1) {
        free(c->buf);
        free(c);
        return NULL;
    }
    c->arr = calloc(c->n ? c->n : 1, sizeof(orc_array));
    size_t pos = 16;
    for (uint32_t i = 0; i < c->n; ++i) {
        uint16_t ln;
        memcpy(&ln, c->buf + pos, 2);
        orc_array *a = &c->arr[i];
        a->name = malloc((size_t)ln + 1);
        memcpy(a->name, c->buf + pos + 2, ln);
        a->name[ln] = 0;
        a->dtype = c->buf[pos + 2 + ln];
        a->ndim = c->buf[pos + 3 + ln];
        size_t head = (size_t)ln + 4;
        pos += head + pad8(head);
        a->count = 1;
        for (int d = 0; d < a->ndim; ++d) {
            memcpy(&a->dims[d], c->buf + pos, 8);
            a->count *= a->dims[d];
            pos += 8;
        }
        a->data = c->buf + pos;
        size_t nbytes = a->count * kDtypeSize[a->dtype];
        pos += nbytes + pad8(nbytes);
    }
    return c;
}

void orc_container_close(orc_container *c) {
    if (!c) return;
    for (uint32_t i = 0; i < c->n; ++i) free(c->arr[i].name);
    free(c->arr);
    free(c->buf);
    free(c);
}

const orc_array *orc_container_get(const orc_container *c, const char *name) {
    for (uint32_t i = 0; i < c->n; ++i)
        if (strcmp(c->arr[i].name, name) == 0) return &c->arr[i];
    return NULL;
}

/* --------------------------------------------------------------------- Philox */
/* Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC11): Philox4x32, 10 rounds. */
orc_philox_out orc_philox4x32_10(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    orc_philox_out o = {{c0, c1, c2, c3}};
    return o;
}

double orc_u32(uint32_t w) { return (double)w * (1.0 / 4294967296.0); }

double orc_u53(uint32_t hi, uint32_t lo) {
    uint64_t x = ((uint64_t)hi << 32) | lo;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

/* -------------------------------------------------------------------- helpers */
uint32_t orc_divide_u32(uint32_t nom, uint32_t den) { return (nom + den / 2u) / den; }

/* Percent(T nom, U den) = Divide(static_cast<T>(nom*100), den): the product is cast back to T. */
uint8_t orc_percent_u16(uint16_t nom, uint16_t den) {
    uint16_t n100 = (uint16_t)(nom * 100);
    return (uint8_t)(uint16_t)((n100 + den / 2) / den);
}
uint8_t orc_percent_u32(uint32_t nom, uint32_t den) {
    uint32_t n100 = nom * 100u;
    return (uint8_t)((n100 + den / 2u) / den);
}
uint8_t orc_percent_u64(uint64_t nom, uint64_t den) {
    uint64_t n100 = nom * 100u;
    return (uint8_t)((n100 + den / 2u) / den);
}
uint8_t orc_safe_percent_u16(uint16_t nom, uint16_t den) { return den ? orc_percent_u16(nom, den) : 50; }

uint32_t orc_transform_distance(uint32_t dist) { return (dist + 9u) / 10u; }

void orc_update_distances(uint32_t reset_distance, uint32_t *dist, uint8_t *start_rate, uint8_t error_rate) {
    if (*dist) {
        if (*start_rate < error_rate) {
            *dist = 0;
            *start_rate = error_rate;
        } else if (++(*dist) >= reset_distance) {
            *dist = 0;
            *start_rate = 0;
        }
    } else if (error_rate) {
        *dist = 1;
        *start_rate = error_rate;
    }
}

double orc_inv_logit2(double bias) { return 2 / (1 + exp(-bias)); }

/* utilities.hpp:229-300 */
static void dombase_find(orc_dominant_base *d, const uint8_t *seq, uint32_t len, uint32_t cur_pos) {
    uint32_t max_content = 0;
    for (int n = 4; n--;)
        if (d->content[n] > max_content) max_content = d->content[n];
    if (0 == max_content) {
        if (len <= cur_pos || 4 == seq[cur_pos]) d->dom_base = 0;
        else d->dom_base = seq[cur_pos];
    } else {
        uint32_t pos = cur_pos;
        while (max_content != d->content[seq[--pos]]) {}
        d->dom_base = seq[pos];
    }
}
void orc_dombase_clear(orc_dominant_base *d) { memset(d->content, 0, sizeof(d->content)); }
void orc_dombase_set(orc_dominant_base *d, const uint8_t *seq, uint32_t len, uint32_t cur_pos) {
    for (uint32_t pos = (5 < cur_pos ? cur_pos - 5 : 0); pos < cur_pos; ++pos) ++d->content[seq[pos]];
    dombase_find(d, seq, len, cur_pos);
}
void orc_dombase_update(orc_dominant_base *d, uint8_t base, const uint8_t *seq, uint32_t len, uint32_t last_pos) {
    ++d->content[base];
    if (5 <= last_pos) --d->content[seq[last_pos - 5]];
    dombase_find(d, seq, len, last_pos + 1);
}

/* ----------------------------------------------------------------------- Draw */
/* ProbabilityEstimates.h:359-380,481-508 */
uint32_t orc_draw(const orc_table *t, const uint32_t *index_in, double random_number, double *prob_sum) {
    *prob_sum = 0.0;
    if (!t->k) return 0;
    uint32_t row[ORC_MAX_MARGINS];
    for (uint32_t n = t->nm; n--;) {                         /* AdjustIndeces */
        if (index_in[n] < t->from[n]) row[n] = 0;
        else if (index_in[n] >= t->to[n]) row[n] = t->to[n] - t->from[n] - 1;
        else row[n] = index_in[n] - t->from[n];
    }
    double prob[256];
    for (uint32_t ind0 = 0; ind0 < t->k; ++ind0) {
        double prod = t->dim2[0][(size_t)row[0] * t->k + ind0];   /* Likelihood */
        for (uint32_t n = 1; n < t->nm; ++n) prod *= t->dim2[n][(size_t)row[n] * t->k + ind0];
        prob[ind0] = prod;
        *prob_sum += prod;
    }
    random_number *= *prob_sum;
    double sum = 0.0;
    uint32_t ind0 = t->k;
    while (sum <= random_number && --ind0) sum += prob[ind0];
    return t->par0[ind0];
}

uint32_t orc_max_value(const orc_table *t) {
    uint32_t m = 0;
    for (uint32_t i = 0; i < t->k; ++i)
        if (t->par0[i] > m) m = t->par0[i];
    return m;
}
uint32_t orc_most_likely(const orc_table *t) { return t->k ? t->par0[t->k - 1] : 0; }

/* -------------------------------------------------------------------- profile */
static const orc_array *need(const orc_container *c, const char *name) {
    const orc_array *a = orc_container_get(c, name);
    if (!a) {
        fprintf(stderr, "oracle: profile misses array '%s'\n", name);
        abort();
    }
    return a;
}
static orc_vect_u64 vect_u64(const orc_container *c, const char *name) {
    char b[128];
    snprintf(b, sizeof b, "%s.from", name);
    const orc_array *a = need(c, name);
    orc_vect_u64 v = {*(const uint64_t *)need(c, b)->data, a->count, (const uint64_t *)a->data};
    return v;
}
static orc_vect_f64 vect_f64(const orc_container *c, const char *name) {
    char b[128];
    snprintf(b, sizeof b, "%s.from", name);
    const orc_array *a = need(c, name);
    orc_vect_f64 v = {*(const uint64_t *)need(c, b)->data, a->count, (const double *)a->data};
    return v;
}

static void load_table(const orc_container *c, const char *prefix, uint32_t nm, orc_table *t) {
    char b[160];
    memset(t, 0, sizeof *t);
    t->nm = nm;
    snprintf(b, sizeof b, "tab.%s.par0", prefix);
    const orc_array *par0 = need(c, b);
    snprintf(b, sizeof b, "tab.%s.limits", prefix);
    const orc_array *lim = need(c, b);
    snprintf(b, sizeof b, "tab.%s.dim2", prefix);
    const orc_array *d2 = need(c, b);
    t->k = (uint32_t)par0->count;
    t->par0 = (const uint32_t *)par0->data;
    const uint32_t *l = (const uint32_t *)lim->data;
    const double *d = (const double *)d2->data;
    for (uint32_t n = 0; n < nm; ++n) {
        t->from[n] = l[2 * n];
        t->to[n] = l[2 * n + 1];
        t->dim2[n] = d;
        if (t->k) d += (size_t)(t->to[n] - t->from[n]) * t->k;
    }
}

/* libstdc++ std::discrete_distribution::param_type::_M_initialize: normalise, partial sums,
 * last forced to 1 (the reference uses discrete_distribution for tile, adapter, poly-A,
 * overrun-base and start-cut draws: Simulator.h:151-173, Simulator.cpp:546). */
static double *discrete_cp(const uint64_t *w, size_t n) {
    double *cp = calloc(n ? n : 1, sizeof(double));
    if (n < 2) {
        if (n) cp[0] = 1.0;
        return cp;
    }
    double sum = 0.0;
    for (size_t i = 0; i < n; ++i) sum += (double)w[i];
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) {
        acc += (double)w[i] / sum;
        cp[i] = acc;
    }
    cp[n - 1] = 1.0;
    return cp;
}

orc_profile *orc_profile_load(const char *path) {
    orc_container *c = orc_container_open(path);
    if (!c) return NULL;
    orc_profile *p = calloc(1, sizeof *p);
    p->c = c;
    p->phred_offset = *(const uint8_t *)need(c, "phred_quality_offset")->data;
    p->corrected_coverage = *(const double *)need(c, "corrected_coverage")->data;
    p->max_len_deletion = *(const uint16_t *)need(c, "errors.max_len_deletion")->data;
    p->reset_distance = *(const uint32_t *)need(c, "coverage.reset_distance")->data;
    char b[160];
    p->total_number_reads = 0;
    for (int seg = 0; seg < 2; ++seg) {
        snprintf(b, sizeof b, "read_lengths.%d", seg);
        p->read_lengths[seg] = vect_u64(c, b);
        for (uint64_t i = 0; i < p->read_lengths[seg].size; ++i) p->total_number_reads += p->read_lengths[seg].v[i]; /* DataStats.cpp:698-700 */
        orc_rl_by_fl *r = &p->rl_by_fl[seg];
        snprintf(b, sizeof b, "rl_by_fl.%d.from", seg);
        r->from = *(const uint64_t *)need(c, b)->data;
        snprintf(b, sizeof b, "rl_by_fl.%d.row_ptr", seg);
        const orc_array *rp = need(c, b);
        r->rows = (uint32_t)rp->count - 1;
        r->row_ptr = (const uint32_t *)rp->data;
        snprintf(b, sizeof b, "rl_by_fl.%d.row_from", seg);
        r->row_from = (const uint32_t *)need(c, b)->data;
        snprintf(b, sizeof b, "rl_by_fl.%d.values", seg);
        r->values = (const uint64_t *)need(c, b)->data;

        orc_adapters *a = &p->adapters[seg];
        snprintf(b, sizeof b, "adapters.%d.seq_ptr", seg);
        const orc_array *sp = need(c, b);
        a->n = (uint32_t)sp->count - 1;
        a->seq_ptr = (const uint32_t *)sp->data;
        snprintf(b, sizeof b, "adapters.%d.seqs", seg);
        a->seqs = (const uint8_t *)need(c, b)->data;
        snprintf(b, sizeof b, "adapters.%d.counts", seg);
        a->counts = (const uint64_t *)need(c, b)->data;
        snprintf(b, sizeof b, "adapters.%d.significant_counts", seg);
        a->significant = (const uint64_t *)need(c, b)->data;
        snprintf(b, sizeof b, "adapters.%d.start_cut_ptr", seg);
        a->cut_ptr = (const uint32_t *)need(c, b)->data;
        snprintf(b, sizeof b, "adapters.%d.start_cut_from", seg);
        a->cut_from = (const uint32_t *)need(c, b)->data;
        snprintf(b, sizeof b, "adapters.%d.start_cut", seg);
        a->cut = (const uint64_t *)need(c, b)->data;
        a->adapter_cp = discrete_cp(a->significant, a->n);
        a->cut_cp = calloc(a->n ? a->n : 1, sizeof(double *));
        for (uint32_t i = 0; i < a->n; ++i) a->cut_cp[i] = discrete_cp(a->cut + a->cut_ptr[i], a->cut_ptr[i + 1] - a->cut_ptr[i]);
    }
    const orc_array *tiles = need(c, "tiles.tiles");
    p->n_tiles = (uint32_t)tiles->count;
    p->tiles = (const uint16_t *)tiles->data;
    p->tile_abundance = (const uint64_t *)need(c, "tiles.abundance")->data;
    p->tile_cp = discrete_cp(p->tile_abundance, p->n_tiles);
    p->polya = vect_u64(c, "adapters.polya_tail_length");
    p->polya_cp = discrete_cp(p->polya.v, p->polya.size);
    p->overrun_bases = (const uint64_t *)need(c, "adapters.overrun_bases")->data;
    {   /* Simulator.h:168: the N at the end is dropped */
        double *cp = discrete_cp(p->overrun_bases, 4);
        memcpy(p->overrun_cp, cp, sizeof p->overrun_cp);
        free(cp);
    }
    p->insert_lengths = vect_u64(c, "frag.insert_lengths");
    p->insert_lengths_bias = vect_f64(c, "frag.insert_lengths_bias");
    p->gc_bias = vect_f64(c, "frag.gc_bias");
    p->sur_bias = (const double *)need(c, "frag.sur_bias")->data;
    memcpy(p->dispersion, need(c, "frag.dispersion_parameters")->data, 16);
    const orc_array *rsb = need(c, "frag.ref_seq_bias");
    p->n_ref_bias = (uint32_t)rsb->count;
    p->ref_seq_bias = (const double *)rsb->data;

    uint32_t nt = p->n_tiles;
    p->quality = calloc((size_t)2 * nt * 4, sizeof(orc_table));
    p->seq_quality = calloc((size_t)2 * nt, sizeof(orc_table));
    p->base_call = calloc((size_t)2 * nt * 4 * 5, sizeof(orc_table));
    for (uint32_t seg = 0; seg < 2; ++seg)
        for (uint32_t tile = 0; tile < nt; ++tile) {
            snprintf(b, sizeof b, "seq_quality.%u.%u", seg, tile);
            load_table(c, b, 3, &p->seq_quality[seg * nt + tile]);
            for (uint32_t base = 0; base < 4; ++base) {
                snprintf(b, sizeof b, "quality.%u.%u.%u", seg, tile, base);
                load_table(c, b, 4, &p->quality[(seg * nt + tile) * 4 + base]);
                for (uint32_t dom = 0; dom < 5; ++dom) {
                    snprintf(b, sizeof b, "base_call.%u.%u.%u.%u", seg, tile, base, dom);
                    load_table(c, b, 4, &p->base_call[((seg * nt + tile) * 4 + base) * 5 + dom]);
                }
            }
        }
    for (uint32_t base = 0; base < 4; ++base)
        for (uint32_t x = 0; x < 5; ++x) {
            for (uint32_t y = 0; y < 5; ++y) {
                snprintf(b, sizeof b, "dom_error.%u.%u.%u", base, x, y);
                load_table(c, b, 3, &p->dom_error[base][x][y]);
            }
            snprintf(b, sizeof b, "error_rate.%u.%u", base, x);
            load_table(c, b, 3, &p->error_rate[base][x]);
        }
    for (uint32_t type = 0; type < 2; ++type)
        for (uint32_t call = 0; call < 6; ++call) {
            snprintf(b, sizeof b, "indels.%u.%u", type, call);
            load_table(c, b, 3, &p->indels[type][call]);
        }
    return p;
}

void orc_profile_free(orc_profile *p) {
    if (!p) return;
    for (int seg = 0; seg < 2; ++seg) {
        for (uint32_t i = 0; i < p->adapters[seg].n; ++i) free(p->adapters[seg].cut_cp[i]);
        free(p->adapters[seg].cut_cp);
        free(p->adapters[seg].adapter_cp);
    }
    free(p->tile_cp);
    free(p->polya_cp);
    free(p->quality);
    free(p->seq_quality);
    free(p->base_call);
    for (size_t i = 0; i < p->n_owned; ++i) free(p->owned[i]);
    free(p->owned);
    orc_container_close(p->c);
    free(p);
}

static void *own(orc_profile *p, void *mem) {
    p->owned = realloc(p->owned, (p->n_owned + 1) * sizeof(void *));
    p->owned[p->n_owned++] = mem;
    return mem;
}

/* ProbabilityEstimates.h:532-545 ModifyPar0 */
static void modify_par0(orc_profile *p, orc_table *t, uint32_t par0_index, double multiplier) {
    uint32_t col = 0;
    while (col < t->k && t->par0[col] != par0_index) ++col;
    if (col >= t->k) return;
    size_t n = (size_t)(t->to[0] - t->from[0]) * t->k;
    double *copy = own(p, malloc(n * sizeof(double)));
    memcpy(copy, t->dim2[0], n * sizeof(double));
    for (size_t i = col; i < n; i += t->k) copy[i] *= multiplier;
    t->dim2[0] = copy;
}

/* ProbabilityEstimates.h:547-556 SetPar0 */
static void set_par0(orc_profile *p, orc_table *t, uint32_t par0_index) {
    uint32_t *par0 = own(p, malloc(sizeof(uint32_t)));
    par0[0] = par0_index;
    t->par0 = par0;
    t->k = 1;
    for (uint32_t n = 0; n < t->nm; ++n) {
        size_t rows = t->to[n] - t->from[n];
        double *d = own(p, malloc((rows ? rows : 1) * sizeof(double)));
        for (size_t i = 0; i < rows; ++i) d[i] = 1.0;
        t->dim2[n] = d;
    }
}

void orc_profile_change_error_rate(orc_profile *p, double multiplier) {      /* :1516-1527 */
    for (uint32_t i = 0; i < 2 * p->n_tiles; ++i)
        for (uint32_t base = 0; base < 4; ++base)
            for (uint32_t dom = 0; dom < 5; ++dom) modify_par0(p, &p->base_call[(i * 4 + base) * 5 + dom], base, 1.0 / multiplier);
}
void orc_profile_remove_substitution_errors(orc_profile *p) {                 /* :1529-1540 */
    for (uint32_t i = 0; i < 2 * p->n_tiles; ++i)
        for (uint32_t base = 0; base < 4; ++base)
            for (uint32_t dom = 0; dom < 5; ++dom) set_par0(p, &p->base_call[(i * 4 + base) * 5 + dom], base);
}
void orc_profile_remove_indel_errors(orc_profile *p) {                        /* :1542-1549 */
    for (uint32_t type = 0; type < 2; ++type)
        for (uint32_t call = 0; call < 6; ++call) set_par0(p, &p->indels[type][call], 0);
}
