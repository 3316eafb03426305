// rsq_core.h -- the per-lane simulation arithmetic: Philox4x32-10 streams, LogArrayResult draws, 2-bit
// reference access, Surrounding k-mers, fragment-count draws and the FillRead state machine.
//
// Design (DESIGN.md "Kernels"): ONE LANE PER READ.  Every draw is the reference's sequential double
// precision recipe (ProbabilityEstimates.h:481-508) executed by a single lane, so sums and products are
// formed in exactly the reference's order and results are bit-identical to the CPU oracle by construction;
// a wave advances 64 independent reads per instruction instead of spending 64 lanes on one K<=41 draw.
// Build with -ffp-contract=off: an FMA would change the rounding of `sum += a*b*c*d`.
//
// The functions are __host__ __device__ so that tests/hostemu can run the very same state machine on the
// CPU against the oracle without a GPU.  The shipped library only ever calls them from kernels.
#pragma once
#include "rsq_types.h"
#if !defined(__HIPCC_RTC__)
#include <math.h>
#endif

namespace rsq {

// ------------------------------------------------------------------------------------------- Philox
struct Words {
    uint32_t w0, w1, w2, w3;
};


// Philox4x32-10 (Salmon et al., SC11); key = seed, counter = (c0,c1,c2,c3).
RSQ_HD Words philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;      // one v_mad_u64_u32 each
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Words{c0, c1, c2, c3};
}

RSQ_HD double u32_to_unit(uint32_t w) { return (double)w * (1.0 / 4294967296.0); }
RSQ_HD double u53_to_unit(uint32_t hi, uint32_t lo) {
    uint64_t x = ((uint64_t)hi << 32) | lo;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

// Identity of one read's random stream: counter words c0..c2 and the tag/segment bits of c3.
struct Stream {
    uint64_t seed;
    uint32_t c0, c1, c2, c3base;
    RSQ_HD Words step(uint32_t s) const { return philox(seed, c0, c1, c2, c3base | s); }
};
RSQ_HD uint32_t pair_c3(uint32_t dom, uint32_t strand, uint32_t segsel, uint32_t allele = 0) { return (dom << 28) | (strand << 27) | (segsel << 25) | (allele << 17); }

// ------------------------------------------------------------------------- integer helpers (utilities.hpp)
RSQ_HD bool is_gc(uint32_t b) { return b == 1 || b == 2; }
RSQ_HD uint32_t divide_u32(uint32_t nom, uint32_t den) { return (nom + den / 2u) / den; }          // :450-452
// x / d for x < 2^24, 1 <= d and a quotient below 2^17 (the percentages: at most 65535 + d / 2 over d).  The device has no integer division: the compiler's
// sequence for a 32-bit one is about 28 instructions.  Here: both operands are exact in single precision, the product with the reciprocal (1 ulp) is off by less than
// 2^17 * 2^-22 of the true quotient, so its integer part is the quotient or one beside it, and the remainder says which.
RSQ_HD uint32_t div_small(uint32_t x, uint32_t d) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t q = (uint32_t)((float)x * __builtin_amdgcn_rcpf((float)d));
    const int32_t r = (int32_t)(x - q * d);
    if (r < 0) --q;
    else if ((uint32_t)r >= d) ++q;
    return q;
#else
    return x / d;
#endif
}
RSQ_HD uint32_t percent_u16(uint32_t nom, uint32_t den) {                                            // :552-554, T = uint16_t
    uint32_t n100 = (nom * 100u) & 0xFFFFu;
    return (div_small(n100 + den / 2u, den) & 0xFFFFu) & 0xFFu;
}
RSQ_HD uint32_t percent_u32(uint32_t nom, uint32_t den) { return div_small(nom * 100u + den / 2u, den) & 0xFFu; }      // nom <= den < 2^16 (G/C counts of fragments, of reads)
RSQ_HD uint32_t safe_percent_u16(uint32_t nom, uint32_t den) { return den ? percent_u16(nom, den) : 50u; }  // :566-573
RSQ_HD uint32_t transform_distance(uint32_t d) { return (d + 9u) / 10u; }                           // :593-595

// first cumulative probability strictly above u (inverse CDF of the reference's std::discrete_distribution draws)
RSQ_HD uint32_t discrete_draw(const double *cp, uint32_t n, double u) {
    if (n < 2) return 0;
    uint32_t i = 0;
    while (i + 1 < n && !(cp[i] > u)) ++i;
    return i;
}

// --------------------------------------------------------------------------- LogArrayResult<N>::Draw
// ProbabilityEstimates.h:359-380 (Likelihood, AdjustIndeces) and :481-508 (Draw).  Two passes over the K outcome
// columns: pass 1 forms prob_sum in ascending order, pass 2 re-forms the same products from the top until the running
// sum exceeds u*prob_sum.  Nothing is kept between the passes, so K is unbounded and the lane needs no per-outcome
// registers.  Rows have an even stride (zero pad column), so both passes read two columns per 16-byte load; the pad
// column adds +0.0 to prob_sum in pass 1 (exact) and is skipped in pass 2.
#if defined(__HIP_DEVICE_COMPILE__)
#define RSQ_LDS __attribute__((address_space(3)))
#else
#define RSQ_LDS
#endif

struct alignas(16) Pair {
    double x, y;
};

struct GlobalRow {                     // a margin row in HBM (read through L1/L2)
    const double *p;
    RSQ_HD Pair pair(uint32_t j) const { return *reinterpret_cast<const Pair *>(p + 2u * j); }
};
#if defined(__HIP_DEVICE_COMPILE__)
// the wave's ballot of a predicate is the compare's own result (s_and with exec); __any() takes an int, so the predicate went to a register, was compared again,
// and the mask took a trip through vcc: three vector instructions per use, four uses in a step
#define RSQ_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0ull)
#else
#define RSQ_ANY(x) (x)
#endif

// Both passes work on CHUNKS of U column pairs: all row loads of a chunk are issued before the first product is formed
// (the loops are latency-bound otherwise).  Rows are zero-padded to whole chunks (row_stride), so no chunk needs a
// validity mask: a pad column contributes the product +0.0, which changes neither sum.  The last chunk of pass 1 is the
// first chunk of pass 2 and stays in registers.
template <int U, class R>
RSQ_HD void load_chunk(Pair (&v)[U], const R &r, uint32_t base) {
#pragma unroll
    for (int i = 0; i < U; ++i) v[i] = r.pair(base + (uint32_t)i);
}
template <int U>
RSQ_HD void mul_chunk(Pair (&a)[U], const Pair (&b)[U]) {
#pragma unroll
    for (int i = 0; i < U; ++i) {
        a[i].x *= b[i].x;
        a[i].y *= b[i].y;
    }
}
// products ((r0*r1)*r2)*r3 -- the order of Likelihood() -- of the pairs base .. base+U-1
template <int U, class R0, class R1, class R2>
RSQ_HD void prod_chunk(Pair (&p)[U], uint32_t base, const R0 &r0, const R1 &r1, const R2 &r2) {
    Pair b[U], c[U];
    load_chunk<U>(p, r0, base);
    load_chunk<U>(b, r1, base);
    load_chunk<U>(c, r2, base);
    mul_chunk<U>(p, b);
    mul_chunk<U>(p, c);
}
template <int U, class R0, class R1, class R2, class R3>
RSQ_HD void prod_chunk(Pair (&p)[U], uint32_t base, const R0 &r0, const R1 &r1, const R2 &r2, const R3 &r3) {
    Pair b[U], c[U], d[U];
    load_chunk<U>(p, r0, base);
    load_chunk<U>(b, r1, base);
    load_chunk<U>(c, r2, base);
    load_chunk<U>(d, r3, base);
    mul_chunk<U>(p, b);
    mul_chunk<U>(p, c);
    mul_chunk<U>(p, d);
}

// LogArrayResult::Draw (ProbabilityEstimates.h:528-560).  Returns the outcome COLUMN (index into par0); prob_sum as in
// the reference.  Pass 2 is the reference's `while(sum <= r && --k) sum += prob[k]`: the products are non-negative, so the
// running sum never decreases and the columns with sum > r are the lowest ones of the scan; counting them per chunk gives
// the column the reference stops at without a branch per column (column 0 may be counted: the result is 0 either way).
template <int U, class... Rs>
RSQ_HD uint32_t draw_rows(uint32_t K, double u, double &prob_sum, const Rs &...rs) {
    const uint32_t nc = chunks_of(K, U);
    Pair p[U] = {};
    double s = 0.0;
    for (uint32_t c = 0; c < nc; ++c) {            // pass 1: prob_sum, ascending columns
        prod_chunk<U>(p, c * U, rs...);
#pragma unroll
        for (int i = 0; i < U; ++i) {
            s += p[i].x;
            s += p[i].y;
        }
    }
    prob_sum = s;
    const double r = u * s;
    double sum = 0.0;
    for (uint32_t c = nc; c--;) {                  // pass 2, descending columns; p holds the top chunk already
        if (c != nc - 1u) prod_chunk<U>(p, c * U, rs...);
        uint32_t above = 0;
#pragma unroll
        for (int i = U; i--;) {
            sum += p[i].y;
            above += !(sum <= r) ? 1u : 0u;
            sum += p[i].x;
            above += !(sum <= r) ? 1u : 0u;
        }
        if (above) return 2u * U * c + above - 1u;
    }
    return 0;
}

// the chunk size is a function of K alone (row_stride pads for it); wave-divergent only when lanes use tables of both classes
template <class... Rs>
RSQ_HD uint32_t draw_rows_k(uint32_t K, double u, double &prob_sum, const Rs &...rs) {
    return K <= 2u * kChunkSmall ? draw_rows<(int)kChunkSmall>(K, u, prob_sum, rs...) : draw_rows<(int)kChunkLarge>(K, u, prob_sum, rs...);
}

// ------------------------------------------------------------------------------------------ screened draw
// The read kernel's draws in SINGLE precision with a proof obligation: the outcome is taken only when it provably equals the
// outcome of the double-precision recipe above, otherwise the lane repeats the draw in double precision (draw<NM>, from HBM).
// Rows are float copies of the tables (the read kernel's families: FamilyGeo, rsq_types.h; the chains': DevTable::off32), four columns per 16-byte load, pad columns zero.
//
// Pass 1 forms the products ((r0*r1)*r2)*r3 of all columns from the bottom quad up and keeps, per quad, the sum of the columns below it; S = the sum of all.
// Pass 2 finds the last quad whose sum-below is less than r = (1-u)*S among those sums (no loads), reloads that quad and finds the column.  Let B(j) be the exact sum
// of the columns below j (over the double-precision values), T the exact total and T(j) = T - B(j) the sum of the columns j and above: the reference returns the
// highest j >= 1 with T(j) > u*T up to its own rounding (relative 1e-14 of T), else 0 -- the highest j with B(j) < (1-u)*T.
// Error of the single-precision quantities, w = 2^-24, all terms non-negative: a product carries (1+w)^7 (four roundings to float, three multiplications), a term
// passes at most 2 + Q additions in S (Q quads) and Q + 5 in a sum from the bottom, and r takes two more roundings -- 1-u = (2^32 - word) * 2^-32 is formed from the
// integer, so its error is relative to ITSELF: |S32 - T| <= (Q+9) w T, |bot32(j) - B(j)| <= (Q+12) w B(j), |r32 - (1-u) T| <= (Q+11) w (1-u) T.
// Sums of non-negative terms have errors relative to themselves, and near the decision all three are about bot32(j+1) =: hi.  So with
//     delta = kScreenSafety * (2Q + 24) * w * (hi + 2^-20 S32)
// r32 - bot32(j) > delta and bot32(j+1) - r32 >= delta imply B(j) < (1-u) T <= B(j+1) with room for the reference's own rounding (the 2^-20 S32 term: 1e-12 T):
// column j is the reference's answer.  Everything else is "undecided": a band of 2 delta around every column boundary, i.e. 2 (2Q+24) w kScreenSafety * sum over
// the boundaries of B(j)/T of all draws.  The columns are sorted by ascending likelihood (ProbabilityEstimates.h GetResults), so B(j)/T is tiny for all but the last
// few: about 5e-5 of the quality draws of profile P0 (K = 40), where the sums from the top down -- errors relative to T at every boundary -- left 2.2e-4 undecided.
// Preconditions, checked when the tables are packed (DevTable::f32_ok) and here: values are 0 or in [2^-60, 2^29] (no overflow;
// an intermediate product that underflows -- (r0*r1) itself below 2^-126, i.e. two factors near 2^-60 and below -- is lost entirely: at most 2^-68 absolutely after the
// largest factors 2^29 * 2^29, far below delta >= 2^-30 * 2^-20 * 2^-19; this assumes gradual or flushed underflow alike, no other denormal mode) and S32 >= 2^-30.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float Float2 __attribute__((ext_vector_type(2)));      // v_pk_mul_f32 / v_pk_add_f32
#else
struct Float2 {
    float x, y;
};
inline Float2 operator*(const Float2 &a, const Float2 &b) { return Float2{a.x * b.x, a.y * b.y}; }
inline Float2 operator+(const Float2 &a, const Float2 &b) { return Float2{a.x + b.x, a.y + b.y}; }
#endif
struct alignas(16) Quad {              // columns 4c .. 4c+3 of a row: lo = (4c, 4c+1), hi = (4c+2, 4c+3)
    Float2 lo, hi;
};
RSQ_HD Quad zero_quad() {
    Quad q;
    q.lo.x = q.lo.y = q.hi.x = q.hi.y = 0.f;
    return q;
}
struct GlobalRow32 {
    const float *p;
    RSQ_HD Quad quad(uint32_t c) const { return *reinterpret_cast<const Quad *>(p + 4u * c); }
};
// a row of the single-precision pool by its BYTE offset from the pool's (wave-uniform) address: the load then takes the pool's address from scalar registers and a
// 32-bit offset per lane (global_load ... v_off, s[base:base+1] offset:16c) -- one 32-bit multiply-add per row instead of a 64-bit address built per access
struct PoolRow32 {
    const float *pool;
    uint32_t at;
    RSQ_HD Quad quad(uint32_t c) const { return *reinterpret_cast<const Quad *>(reinterpret_cast<const char *>(pool) + (at + 16u * c)); }
};
struct LdsRow32 {
    const RSQ_LDS float *p;
    RSQ_HD Quad quad(uint32_t c) const { return *reinterpret_cast<const RSQ_LDS Quad *>(p + 4u * c); }
};
// A row that some lanes of the wave find in LDS while the others read it from HBM (used only when not every lane finds its row in
// LDS: the callers branch wave-uniformly between LdsRow32 and this)
struct MixedRow32 {
    LdsRow32 l;
    PoolRow32 g;
    bool use_lds;
    RSQ_HD Quad quad(uint32_t c) const { return use_lds ? l.quad(c) : g.quad(c); }
};

RSQ_HD Quad mul_quad(const Quad &a, const Quad &b) { return Quad{a.lo * b.lo, a.hi * b.hi}; }
template <class R0, class R1, class R2>
RSQ_HD Quad prod_quad(uint32_t c, const R0 &r0, const R1 &r1, const R2 &r2) {
    const Quad a = r0.quad(c), b = r1.quad(c), d = r2.quad(c);
    return mul_quad(mul_quad(a, b), d);
}
template <class R0, class R1, class R2, class R3>
RSQ_HD Quad prod_quad(uint32_t c, const R0 &r0, const R1 &r1, const R2 &r2, const R3 &r3) {
    const Quad a = r0.quad(c), b = r1.quad(c), d = r2.quad(c), e = r3.quad(c);
    return mul_quad(mul_quad(mul_quad(a, b), d), e);
}
#if defined(__HIP_DEVICE_COMPILE__)
#define RSQ_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define RSQ_SCHED_BARRIER() ((void)0)
#endif
#ifndef RSQ_SCREEN_BATCH
#define RSQ_SCREEN_BATCH 1
#endif
constexpr float kScreenSafety = 1.5f;
constexpr float kScreenMinSum = 9.313225746154785e-10f;      // 2^-30

// Q quads per row (compile-time: the loops unroll, the loads of a batch of quads are issued before their products are formed).
// Returns true and the outcome COLUMN if the draw is decided.
template <int Q, class... Rs>
RSQ_HD bool draw_screened(uint32_t word, uint32_t &col, const Rs &...rs) {
    constexpr int G = Q % RSQ_SCREEN_BATCH == 0 ? RSQ_SCREEN_BATCH : (Q % 2 == 0 ? 2 : 1);      // quads per batch
    constexpr bool kKeep = Q <= 2;                           // short rows: the products stay in registers, pass 2 loads nothing
    // pass 1 runs from the bottom quad up with ONE running sum, kept as a pair (the columns 0, 1 and 2, 3 of the quads so far): bot[c] = the sum of the quads below
    // c, S = bot[Q].  A term passes one addition inside its quad, at most Q in the running pair and one across the pair: the bounds above hold (Q + 2 additions
    // in S and in a sum from the bottom, four more inside the chosen quad).  [One sum instead of a sum per quad, a total and a second pass over the per-quad sums: 20
    // additions and the compiler's shuffles to pair them up less per draw of 40 columns.]
    float bot[Q + 1];
    Quad kept[kKeep ? Q : 1];
    Float2 run;
    run.x = run.y = 0.f;
    bot[0] = 0.f;
#pragma unroll
    for (int g = 0; g < Q; g += G) {
        Quad p[G];
#pragma unroll
        for (int i = 0; i < G; ++i) p[i] = prod_quad((uint32_t)(g + i), rs...);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            run = run + (p[i].lo + p[i].hi);
            bot[g + i + 1] = run.x + run.y;
            if constexpr (kKeep) kept[g + i] = p[i];
        }
        RSQ_SCHED_BARRIER();                                 // keeps the scheduler from hoisting the loads of every batch to the top (registers)
    }
    const float S = bot[Q];
    const float r = ((float)(0u - word) * 2.3283064365386963e-10f) * S;      // (1 - u) S, 1 - u = (2^32 - word) * 2^-32 rounded to single precision; word 0 gives 0: undecided
    // the sums from the bottom never decrease: the quads whose sum-below is less than r are the lowest ones; count them, keep the last such sum
    float base = 0.f;
    uint32_t below = 0;
#pragma unroll
    for (int c = 0; c < Q; ++c) {
        const bool hit = bot[c] < r;
        base = hit ? bot[c] : base;
        below += hit ? 1u : 0u;
    }
    const uint32_t fc = below ? below - 1u : 0u;             // below == 0: r is 0 (word 0, or S is 0), left undecided
    Quad p;
    if constexpr (kKeep) {
        p = kept[0];
#pragma unroll
        for (int c = 1; c < Q; ++c)
            if (fc == (uint32_t)c) p = kept[c];
    } else p = prod_quad(fc, rs...);
    const float b1 = base + p.lo.x, b2 = b1 + p.lo.y, b3 = b2 + p.hi.x, b4 = b3 + p.hi.y;
    uint32_t j;
    float hi, lo;
    if (b3 < r) j = 3u, lo = b3, hi = b4;
    else if (b2 < r) j = 2u, lo = b2, hi = b3;
    else if (b1 < r) j = 1u, lo = b1, hi = b2;
    else j = 0u, lo = base, hi = b1;
    col = 4u * fc + j;
    const float delta = (kScreenSafety * (float)(2 * Q + 24) * 5.9604644775390625e-08f) * (hi + 9.5367431640625e-07f * S);
    return S >= kScreenMinSum && below != 0u && r - lo > delta && hi - r >= delta;      // S >= 2^-30 also rejects NaN
}

// AdjustIndeces (:368-380): v < from ? 0 : (v - from >= rows ? rows - 1 : v - from), as a signed difference held between 0 and rows - 1 (values, first values
// and row counts are far below 2^31; a table that is drawn from has rows): one subtraction and a median of three on the device instead of two compares and two selects
RSQ_HD uint32_t clamp_row(const DevTable &t, int n, uint32_t v) {
    const int32_t d = (int32_t)v - (int32_t)t.from[n], last = (int32_t)t.rows[n] - 1;
#if defined(__clang__)
    return (uint32_t)__builtin_elementwise_min(__builtin_elementwise_max(d, 0), last);
#else
    return (uint32_t)(d < 0 ? 0 : (d > last ? last : d));
#endif
}

// generic draw with every margin in HBM; returns the outcome VALUE
template <int NM, class Par0>
RSQ_HD uint32_t draw(const DevTable &t, const double *__restrict__ pool, Par0 par0, const uint32_t (&idx)[NM], double u, double &prob_sum) {
    prob_sum = 0.0;
    if (!t.k) return 0;
    const uint32_t kp = row_stride(t.k);
    GlobalRow m[NM];
#pragma unroll
    for (int n = 0; n < NM; ++n) m[n].p = pool + t.off[n] + (size_t)clamp_row(t, n, idx[n]) * kp;
    uint32_t col;
    if constexpr (NM == 3) col = draw_rows_k(t.k, u, prob_sum, m[0], m[1], m[2]);
    else col = draw_rows_k(t.k, u, prob_sum, m[0], m[1], m[2], m[3]);
    return par0[t.par0_off + col];
}

// The same draw one column pair at a time: identical additions in identical order (the chunk size only groups the loads), a fraction of
// the registers.  For the rare draws the read kernel's screen leaves open, where speed does not matter and registers do.
template <int NM, class Par0>
RSQ_HD uint32_t draw_slim(const DevTable &t, const double *__restrict__ pool, Par0 par0, const uint32_t (&idx)[NM], double u, double &prob_sum) {
    prob_sum = 0.0;
    if (!t.k) return 0;
    const uint32_t kp = row_stride(t.k);
    GlobalRow m[NM];
#pragma unroll
    for (int n = 0; n < NM; ++n) m[n].p = pool + t.off[n] + (size_t)clamp_row(t, n, idx[n]) * kp;
    uint32_t col;
    if constexpr (NM == 3) col = draw_rows<1>(t.k, u, prob_sum, m[0], m[1], m[2]);
    else col = draw_rows<1>(t.k, u, prob_sum, m[0], m[1], m[2], m[3]);
    return par0[t.par0_off + col];
}

// How FillRead reaches its tables.  GlobalTables reads descriptors and rows from HBM; the read kernel uses LdsTables
// (rsq_kernels.h) which serves descriptors and the per-lane-varying margins from LDS.
struct GlobalTables {
    using Sum = double;            // what a draw reports of prob_sum; the callers only ask whether it is 0
    const DevSim &S;
    RSQ_HD const DevTable &quality(uint32_t i) const { return S.quality[i]; }
    RSQ_HD const DevTable &seq_quality(uint32_t i) const { return S.seq_quality[i]; }
    // `word`: the draw's 32 random bits, u = word * 2^-32
    RSQ_HD uint32_t draw_quality(uint32_t i, const uint32_t (&idx)[4], uint32_t word, double &ps) const { return draw<4>(S.quality[i], S.pool, S.par0, idx, u32_to_unit(word), ps); }
    RSQ_HD uint32_t draw_base_call(uint32_t i, const uint32_t (&idx)[4], uint32_t word, double &ps) const { return draw<4>(S.base_call[i], S.pool, S.par0, idx, u32_to_unit(word), ps); }
    RSQ_HD uint32_t draw_indel(uint32_t i, const uint32_t (&idx)[3], uint32_t word, double &ps) const { return draw<3>(S.indels[i], S.pool, S.par0, idx, u32_to_unit(word), ps); }
    RSQ_HD uint32_t draw_seq_quality(uint32_t i, const uint32_t (&idx)[3], uint32_t word, double &ps) const { return draw<3>(S.seq_quality[i], S.pool, S.par0, idx, u32_to_unit(word), ps); }
};

// ------------------------------------------------------------------------------- 2-bit reference access
RSQ_HD uint32_t ref_base(const uint64_t *__restrict__ words, uint64_t word_off, uint32_t pos) {
    return (uint32_t)(words[word_off + (pos >> 5)] >> ((pos & 31u) * 2u)) & 3u;
}
// number of G/C in [a,b): a base is G/C iff its two bits differ
RSQ_HD uint32_t ref_gc_count(const uint64_t *__restrict__ words, uint64_t word_off, uint32_t a, uint32_t b) {
    uint32_t gc = 0;
    const uint64_t kLow = 0x5555555555555555ull;
    for (uint32_t w = a >> 5; w <= ((b - 1) >> 5) && a < b; ++w) {
        uint64_t x = words[word_off + w];
        uint64_t g = (x ^ (x >> 1)) & kLow;
        uint32_t lo = w == (a >> 5) ? (a & 31u) : 0u;
        uint32_t hi = w == ((b - 1) >> 5) ? ((b - 1) & 31u) + 1u : 32u;
        if (lo) g &= ~0ull << (2u * lo);
        if (hi < 32u) g &= (1ull << (2u * hi)) - 1ull;
#if defined(__HIP_DEVICE_COMPILE__)
        gc += (uint32_t)__popcll(g);
#else
        gc += (uint32_t)__builtin_popcountll(g);
#endif
    }
    return gc;
}

// the same count from the per-word running totals built by pack_reference (gc_prefix[w] = G/C in words [0, w) of the sequence):
// four independent loads instead of a loop over up to 32 words
RSQ_HD uint32_t gc_low(uint64_t x, uint32_t n_bases) {            // G/C among the lowest n_bases (< 32) bases of a word
    const uint64_t g = (x ^ (x >> 1)) & 0x5555555555555555ull & ((1ull << (2u * n_bases)) - 1ull);
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll(g);
#else
    return (uint32_t)__builtin_popcountll(g);
#endif
}
RSQ_HD uint32_t ref_gc_count_prefix(const uint64_t *__restrict__ words, const uint32_t *__restrict__ gc_prefix, uint64_t word_off, uint32_t a, uint32_t b) {
    const uint32_t wa = a >> 5, wb = b >> 5;
    return gc_prefix[word_off + wb] - gc_prefix[word_off + wa] + gc_low(words[word_off + wb], b & 31u) - gc_low(words[word_off + wa], a & 31u);
}

// ----------------------------------------------------------------------------------------- Surrounding
// SurroundingBase.hpp:64-81,196-202: block b of the start surrounding is the 10-mer ref[pos-10+10b ..), wrapping
// around the sequence ends; the end surrounding is taken on the reverse complement at position L-1-pos.
// 60 bits = 30 bases starting at base p of the sequence (p + 30 <= L): at most two words
RSQ_HD uint64_t ref_bits60(const uint64_t *__restrict__ words, uint64_t word_off, uint32_t p) {
    const uint64_t *w = words + word_off + (p >> 5);
    const uint32_t off = (p & 31u) * 2u;
    uint64_t x = w[0] >> off;
    if (off > 4u) x |= w[1] << (64u - off);                       // the sequence's spare word makes w[1] readable
    return x & ((1ull << 60) - 1ull);
}
// ten bases packed first-base-lowest -> the reference's code with the first base most significant
RSQ_HD uint32_t reverse_ten_bases(uint32_t g) {
    uint32_t r = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    r = __brev(g) >> 12;
#else
    for (uint32_t i = 0; i < 20u; ++i) r |= ((g >> i) & 1u) << (19u - i);
#endif
    return ((r & 0x55555u) << 1) | ((r >> 1) & 0x55555u);         // the two bits of every base back in order
}
// `wrap_words`: with variants `words` is an allele's copy of the reference; the bases a window takes from beyond a sequence end
// (wrap-around) are the reference's own, because the variant edits of the surroundings stop at the ends
// (HandleSurroundingVariantsBeforeCenter / AfterCenter, Simulator.cpp:1459-1589)
RSQ_HD void surrounding_forward(const uint64_t *__restrict__ words, uint64_t word_off, uint32_t L, uint32_t pos, uint32_t (&sur)[3],
                                const uint64_t *__restrict__ wrap_words = nullptr) {
    uint64_t p = (uint64_t)pos + L - kSurStart;                    // < 2L: one conditional subtraction replaces the reference's % length
    if (p >= L) p -= L;
    bool wrapped = pos < kSurStart;                                // the window begins before base 0
    if (p + kSurBlocks * kSurRange <= L) {                         // no wrap-around: two loads instead of thirty
        const uint64_t x = ref_bits60(words, word_off, (uint32_t)p);
#pragma unroll
        for (uint32_t b = 0; b < kSurBlocks; ++b) sur[b] = reverse_ten_bases((uint32_t)(x >> (20u * b)) & 0xFFFFFu);
        return;
    }
    // windows over a sequence end (rare): rolled loops -- unrolled, their thirty loads cost the sieve's candidate kernel hundreds of vector registers
#pragma unroll 1
    for (uint32_t b = 0; b < kSurBlocks; ++b) {
        uint32_t v = 0;
#pragma unroll 1
        for (uint32_t i = 0; i < kSurRange; ++i) {
            v = (v << 2) + ref_base(wrapped && wrap_words ? wrap_words : words, word_off, (uint32_t)p);
            if (++p == L) {
                p = 0;
                wrapped = !wrapped;
            }
        }
        sur[b] = v;
    }
}
RSQ_HD void surrounding_reverse(const uint64_t *__restrict__ words, uint64_t word_off, uint32_t L, uint32_t pos, uint32_t (&sur)[3],
                                const uint64_t *__restrict__ wrap_words = nullptr) {
    // position q of the reverse complement is the complement of forward position L-1-q; q starts at (L-pos-1) + L - 10 (mod L)
    uint64_t q = (uint64_t)(L - pos - 1) + L - kSurStart;
    if (q >= L) q -= L;
    uint32_t f = L - 1u - (uint32_t)q;                             // forward position, walks downwards with wrap-around
    bool wrapped = pos + kSurStart >= L;                           // the window begins behind the last base
    if (f + 1u >= kSurBlocks * kSurRange) {                        // no wrap-around: block b is the complement of forward bases f-10b-9 .. f-10b
        const uint64_t x = ref_bits60(words, word_off, f + 1u - kSurBlocks * kSurRange);
#pragma unroll
        for (uint32_t b = 0; b < kSurBlocks; ++b) sur[b] = ~(uint32_t)(x >> (20u * (kSurBlocks - 1u - b))) & 0xFFFFFu;
        return;
    }
#pragma unroll 1
    for (uint32_t b = 0; b < kSurBlocks; ++b) {
        uint32_t v = 0;
#pragma unroll 1
        for (uint32_t i = 0; i < kSurRange; ++i) {
            v = (v << 2) + (3u - ref_base(wrapped && wrap_words ? wrap_words : words, word_off, f));
            if (f) --f;
            else {
                f = L - 1u;
                wrapped = !wrapped;
            }
        }
        sur[b] = v;
    }
}
// Surrounding.h:114-120 (blocks summed last to first) and utilities.hpp:505 InvLogit2
RSQ_HD double surrounding_bias(const double *__restrict__ sur_bias, const uint32_t (&sur)[3]) {
    double bias = 0.0;
    for (int b = (int)kSurBlocks; b--;) bias += sur_bias[(size_t)b * kSurSize + sur[b]];
    return 2 / (1 + exp(-bias));
}

// ---------------------------------------------------------------------------- fragment-count arithmetic
RSQ_HD double get_dispersion(double bias, double a, double b) {        // FragmentDistributionStats.cpp:900-907
    double r = bias / (a + b * bias);
    if (r > bias * 1e10) r = bias * 1e10;
    return r;
}
RSQ_HD uint32_t binomial(uint32_t n, double p, double probability_chosen) {     // :3584-3596
    double probability_count = pow(1 - p, (double)n), probability_left = probability_chosen - probability_count;
    uint32_t count = 0;
    while (0.0 < probability_left && count < n) {
        ++count;
        probability_count *= (double)(n + 1 - count) / count * p / (1 - p);
        probability_left -= probability_count;
    }
    return count;
}
RSQ_HD uint32_t negative_binomial(double p, double r, double probability_chosen) {   // :3602-3613 (uintDupCount wraps at 2^16)
    double probability_count = pow(1 - p, r), probability_left = probability_chosen - probability_count;
    uint32_t count = 0;
    while (0.0 < probability_left) {
        count = (count + 1u) & 0xFFFFu;
        probability_count *= p * ((r - 1) / count + 1);
        probability_left -= probability_count;
    }
    return count;
}
// :3615-3627 GetFragmentCounts with Reference::Bias (Reference.h:167-169,283-285)
RSQ_HD uint32_t fragment_counts(const DevSim &S, uint32_t seq, uint32_t fragment_length, uint32_t gc, const uint32_t (&sur_start)[3],
                                const uint32_t (&sur_end)[3], double probability_chosen) {
    double general = S.ref_seq_bias[seq] * S.insert_lengths_bias[fragment_length];
    double bias = general * S.gc_bias[gc] * surrounding_bias(S.sur_bias, sur_start) * surrounding_bias(S.sur_bias, sur_end);
    if (0.0 < bias) {
        double mean = bias * S.bias_normalization;
        double dispersion = get_dispersion(mean, S.dispersion[0], S.dispersion[1]) / S.num_alleles;
        mean /= S.num_alleles;
        return negative_binomial(mean / (mean + dispersion), dispersion, probability_chosen);
    }
    return 0;
}

// ---------------------------------------------------------------------------------- FillRead state machine
// Simulator.cpp:454-594 (FillRead), :294-452 (FillReadPart), :240-292 (GetSysErrorFromBlock, no variants),
// Simulator.h:185-198 (ReadLength).
//
// Src: org_len(), base(k) in 0..3, sys(k) = dom | rate<<8 of the k-th template base.
// Out: put(read_pos, base_code, qual_char), op(iteration, code) with code 0 = template base, 1 = deletion, 2 = insertion.
// Random stream steps: 0 = {read length, adapter (adapter-only read), sequence quality, adapter start cut},
// 1 = {poly-A tail length, adapter}, 2+t = t-th state-machine iteration {indel, quality, base call, overrun base}.

RSQ_HD uint32_t digits10(uint32_t v) { return v < 10 ? 1u : v < 100 ? 2u : v < 1000 ? 3u : v < 10000 ? 4u : 5u; }

struct CigarRun {                      // the RLE bookkeeping of FillReadPart (Simulator.cpp:311-312,358-368,398-408,428-438,447-449)
    char element;
    uint32_t length;
    uint32_t chars;                    // characters of the CIGAR string so far
    RSQ_HD void flush() { chars += digits10(length) + 1u; }
};

struct FillState {                     // Simulator.h:215-240 ReadFillParameter
    uint32_t read_length, read_pos;
    uint32_t previous_indel_type, indel_pos, base_call, gc_seq;
    uint32_t seq_qual, qual, error_rate, num_errors;
    uint32_t last_written_qual;        // at(sim_read.qual_, read_pos-1) - offset
    uint32_t iteration;
};

RSQ_HD uint32_t draw_read_length(const DevSim &S, uint32_t seg, uint32_t fragment_length, double u) {   // Simulator.h:185-198
    const DevReadLengths &rl = S.read_lengths[seg];
    if (rl.fixed) return rl.fixed;
    const double random_value = u * (double)S.insert_lengths[fragment_length];
    double counter = 0.0;
    const uint32_t row = fragment_length - rl.row_first;
    const uint32_t from = rl.row_from[row] & 0xFFFFu;
    uint32_t read_len = (from + (rl.row_ptr[row + 1] - rl.row_ptr[row])) & 0xFFFFu;
    while (counter <= random_value) {
        const bool more = read_len > from;
        read_len = (read_len - 1u) & 0xFFFFu;                      // uintReadLen post-decrement (wraps like the reference)
        if (!more) break;
        counter += (double)rl.values[rl.row_ptr[row] + (read_len - from)];
    }
    return read_len;
}

// the plain loop of Simulator.cpp:482-489 for any template source
template <class Src>
RSQ_HD void template_totals_loop(const Src &src, uint32_t n, uint32_t &gc, uint32_t &rate_sum) {
    for (uint32_t k = 0; k < n; ++k) {
        if (is_gc(src.base(k))) ++gc;
        rate_sum += src.sys_base(k) >> 8;
    }
}

struct AdapterSrc {                    // the adapter as template of FillReadPart(..., 'S', NULL, ...)
    const uint8_t *seq;
    const uint16_t *sys_;
    RSQ_HD uint32_t base(uint32_t k) const { return seq[k]; }
    RSQ_HD uint32_t sys_base(uint32_t k) const { return sys_[k]; }
    RSQ_HD uint32_t sys_deleted(uint32_t k) const { return sys_[k]; }
};

// FillRead as an explicit state machine: init() does everything before the first per-base iteration (Simulator.cpp:468-531),
// every step() executes exactly ONE iteration of FillReadPart's loop (template part 'M', then adapter part 'S', :294-452) or
// of the poly-A / overrun tail (:556-588) and consumes Philox step 2+t; finalize() fills the ReadMeta.  A kernel drives all
// lanes of a wave through step() in one uniform loop, which is what allows wave-cooperative work between iterations.
struct ReadMachine {
    enum : uint32_t { kTemplate = 0, kAdapter = 1, kTail = 2, kDone = 3 };
    FillState par;
    CigarRun cg;
    uint32_t phase, seg, tile_id, tbase;
    uint32_t org_pos, org_len;         // position in / length of the current part's template
    uint32_t adapter_id, adapter_a0;
    uint32_t iter_m, hard_clip, tail_length, pos_tail, n_indels;
    uint32_t start_cut_word;           // h0.w3, needed only if the read starts inside the adapter

    // a machine without a read: every step() returns false at once (the lanes of a wave beyond the end of the batch)
    RSQ_HD void idle() {
        par = FillState{};
        cg = CigarRun{'M', 0, 0};
        phase = kDone;
        seg = tile_id = tbase = org_pos = org_len = adapter_id = adapter_a0 = iter_m = hard_clip = tail_length = pos_tail = n_indels = start_cut_word = 0;
    }
    template <class Tab, class Src>
    RSQ_HD void init(const DevSim &S, const Tab &tab, const Stream &st, uint32_t seg_, uint32_t tile, uint32_t fragment_length, const Src &src) {
        seg = seg_;
        tile_id = tile;
        tbase = (seg * RSQ_SIM(S, n_tiles) + tile_id) * 4u;
        par.read_pos = 0;
        par.previous_indel_type = 0;
        par.indel_pos = 0;
        par.base_call = 5;
        par.gc_seq = 0;
        par.qual = 1;
        par.error_rate = 0;
        par.num_errors = 0;
        par.last_written_qual = 0;
        par.iteration = 0;
        const Words h0 = st.step(0);
        start_cut_word = h0.w3;
        par.read_length = draw_read_length(S, seg, fragment_length, u32_to_unit(h0.w0));
        const DevAdapters &ad = S.adapters[seg];
        org_len = src.org_len();
        org_pos = 0;
        adapter_id = 0;
        adapter_a0 = 0;
        iter_m = hard_clip = tail_length = pos_tail = n_indels = 0;
        const uint32_t seq_length = par.read_length < org_len ? par.read_length : org_len;
        uint32_t mean_error_rate = 0;
        if (seq_length) {                                              // Simulator.cpp:480-504
            src.totals(seq_length, par.gc_seq, mean_error_rate);       // G/C bases and the sum of the error rates of the first seq_length template bases
            par.gc_seq = percent_u16(par.gc_seq, seq_length);
            mean_error_rate = divide_u32(mean_error_rate, seq_length);
        } else {                                                       // adapter-only read :505-522
            adapter_id = discrete_draw(ad.adapter_cp, ad.n, u32_to_unit(h0.w1));
            const uint32_t a0 = ad.seq_ptr[adapter_id], alen = ad.seq_ptr[adapter_id + 1] - a0;
            for (uint32_t k = 0; k < alen; ++k) {
                if (is_gc(ad.seqs[a0 + k])) ++par.gc_seq;
                mean_error_rate += ad.sys[a0 + k] >> 8;
            }
            par.gc_seq = percent_u16(par.gc_seq, alen & 0xFFFFu);
            mean_error_rate = divide_u32(mean_error_rate, alen);
        }
        typename Tab::Sum prob_sum;
        const uint32_t sqi = seg * RSQ_SIM(S, n_tiles) + tile_id;
        const uint32_t idx_sq[3] = {par.gc_seq, mean_error_rate, fragment_length / kSqFragmentLengthBinSize};
        par.seq_qual = tab.draw_seq_quality(sqi, idx_sq, h0.w2, prob_sum);
        if (0 == prob_sum) {                                         // MostLikely(), ProbabilityEstimates.h:519-526
            const DevTable sqt = tab.seq_quality(sqi);
            par.seq_qual = sqt.k ? S.par0[sqt.par0_off + sqt.k - 1u] : 0u;
        }
        cg = CigarRun{'M', 0, 0};
        phase = kTemplate;
    }

    // Decides what the next iteration is: performs the transitions between the template part, the adapter part and the tail
    // (Simulator.cpp:447-449, 537-558) until one of them has an iteration to run.  Returns false once the read is complete.
    RSQ_HD bool advance(const DevSim &S, const Stream &st) {
        if (__builtin_expect(in_template(), 1)) return true;      // nearly every iteration
        return advance_parts(S, st);
    }
    RSQ_HD bool advance_parts(const DevSim &S, const Stream &st) {
        for (;;) {
            if (phase == kTemplate) {
                if (par.read_pos < par.read_length && org_pos < org_len) return true;
                if (cg.length) cg.flush();                          // Simulator.cpp:447-449
                iter_m = par.iteration;
                if (!(par.read_pos < par.read_length)) {
                    phase = kDone;
                    return false;
                }
                const DevAdapters &ad = S.adapters[seg];            // :537-553 the fragment ended before the read did
                const Words h1 = st.step(1);
                if (0 == adapter_id) adapter_id = discrete_draw(ad.adapter_cp, ad.n, u32_to_unit(h1.w1));
                uint32_t adapter_pos = 0;
                if (0 == par.read_pos)
                    adapter_pos = discrete_draw(ad.cut_cp + ad.cut_ptr[adapter_id], ad.cut_ptr[adapter_id + 1] - ad.cut_ptr[adapter_id], u32_to_unit(start_cut_word)) +
                                  ad.cut_from[adapter_id];
                adapter_a0 = ad.seq_ptr[adapter_id];
                org_len = ad.seq_ptr[adapter_id + 1] - adapter_a0;
                org_pos = adapter_pos;
                tail_length = discrete_draw(S.polya_cp, S.polya_n, u32_to_unit(h1.w0)) + S.polya_from;      // used only if a tail follows
                cg.element = 'S';
                cg.length = 0;
                phase = kAdapter;
            } else if (phase == kAdapter) {
                if (par.read_pos < par.read_length && org_pos < org_len) return true;
                if (cg.length) cg.flush();
                if (!(par.read_pos < par.read_length)) {
                    phase = kDone;
                    return false;
                }
                hard_clip = par.read_length - par.read_pos;         // :556-558
                cg.chars += digits10(hard_clip) + 1u;
                pos_tail = 0;
                phase = kTail;
            } else if (phase == kTail) {
                if (par.read_pos < par.read_length) return true;
                phase = kDone;
                return false;
            } else return false;
        }
    }

    // ONE iteration of FillReadPart's loop body (Simulator.cpp:322-444; template part 'M' or adapter part 'S') or of the poly-A /
    // overrun tail (:564-587).  The three kinds share one instance of every draw: the lanes of a wave may be in different parts
    // of their reads and still execute the same instructions.  Returns false once the read is complete.
    template <class Tab, class Src, class Out>
    RSQ_HD bool step(const DevSim &S, const Tab &tab, const Stream &st, const Src &src, Out &out) {
        if (!advance(S, st)) return false;
        iterate(S, tab, st, src, out);
        return true;
    }
    // whether the next iteration is one of the template part -- what advance() finds for nearly every iteration
    RSQ_HD bool in_template() const { return phase == kTemplate && par.read_pos < par.read_length && org_pos < org_len; }
    // the iteration itself, for a machine that advance() found an iteration for.  TEMPLATE_PART: the caller knows that the machine is in the template part
    // (in_template()): the adapter's and the tail's branches are not compiled in.
    template <bool TEMPLATE_PART = false, class Tab, class Src, class Out>
    RSQ_HD void iterate(const DevSim &S, const Tab &tab, const Stream &st, const Src &src, Out &out) {
        const bool tail = !TEMPLATE_PART && phase == kTail, from_template = TEMPLATE_PART || phase == kTemplate;
        const DevAdapters &ad = S.adapters[seg];
        const uint32_t it = par.iteration;                           // counted at the end of the step: old and new value never live side by side
        const Words w = st.step(2u + it);
        typename Tab::Sum prob_sum;
        uint32_t indel = 0, org_base = 0;
        if (!tail) {
            const uint32_t idx_i[3] = {par.indel_pos, par.read_pos, par.gc_seq};
            indel = tab.draw_indel(par.previous_indel_type * 6u + par.base_call, idx_i, w.w0, prob_sum);
            if (0 == prob_sum) indel = 0;
            org_base = from_template ? src.base(org_pos) : (uint32_t)ad.seqs[adapter_a0 + org_pos];
        }
        const uint32_t qi = tbase + org_base;                        // the tail draws from the tables of base A (:566)
        const bool regular = !tail && 0 == indel, deletion = !tail && 1 == indel;
        uint32_t dom_error = 0;
        if (regular) {
            // GetSysErrorFromBlock: without variants its block walk advances in step with org_pos (:286-291)
            const uint32_t se = from_template ? src.sys_base(org_pos) : (uint32_t)ad.sys[adapter_a0 + org_pos];
            dom_error = se & 0xFFu;
            par.error_rate = se >> 8;
        } else if (deletion) {
            par.error_rate = (from_template ? src.sys_deleted(org_pos) : (uint32_t)ad.sys[adapter_a0 + org_pos]) >> 8;      // :380-392
        }
        uint32_t q = 0;
        if (!deletion) {
            const uint32_t idx_q[4] = {par.seq_qual, par.qual, par.read_pos, par.error_rate};
            q = tab.draw_quality(qi, idx_q, w.w1, prob_sum);
            if (0 == prob_sum) {
                if (regular) q = par.read_pos ? par.last_written_qual : tab.quality(qi).max_value;      // :341-349
                else if (tail) q = par.read_pos ? par.last_written_qual : q;                             // at(qual_, read_pos-1) - offset
                else q = par.qual;                                                                       // insertion :417-420
            }
        }
        // what the iteration emits is decided in the branches and written in ONE place behind them (a base and its quality for every kind but a deletion, an edit
        // operation for every kind but the tail): one store site instead of three, and the output cursors change in one place
        uint32_t put_base = 0, op_code = 0;
        char element = 0;                                          // the CIGAR run the iteration belongs to; the tail has none
        const bool insertion = !tail && indel > 1u;
        if (regular) {
            par.qual = q;
            const uint32_t idx_b[4] = {par.qual, par.read_pos, par.num_errors, par.error_rate};
            uint32_t call = tab.draw_base_call(qi * 5u + dom_error, idx_b, w.w2, prob_sum);
            if (0 == prob_sum) call = org_base;
            par.base_call = call;
            put_base = call;
            element = from_template ? 'M' : 'S';
            if (call != org_base) ++par.num_errors;
        } else if (deletion) {                                     // ErrorStats::kDeletion
            op_code = 1u;
            element = 'D';
        } else if (tail) {                                         // :564-587 poly-A tail, then random overrun bases
            par.qual = q;
            put_base = pos_tail < tail_length ? 0u : discrete_draw(S.overrun_cp, 4, u32_to_unit(w.w3));
            ++pos_tail;
        } else {                                                   // insertion of base indel-2
            put_base = indel - 2u;
            op_code = 2u;
            element = 'I';
        }
        // the run-length bookkeeping of the three kinds with a CIGAR element, once (:358-368, 398-408, 428-438): the run goes on, or the one before is closed
        if (element) {
            const uint32_t is_indel = deletion || insertion ? 1u : 0u;
            n_indels += is_indel;
            par.num_errors += is_indel;
            if (element == cg.element) {
                ++cg.length;
                par.indel_pos += is_indel;
            } else {
                cg.flush();
                cg.element = element;
                cg.length = 1;
                par.indel_pos = is_indel;
                par.previous_indel_type = deletion ? 1u : 0u;
            }
            if (!insertion) ++org_pos;                             // a template base was used (or deleted)
        }
        if (!deletion) {
            out.put(par.read_pos, put_base, q + RSQ_SIM(S, phred_offset));
            par.last_written_qual = q;
            ++par.read_pos;
        }
        if (!tail) out.op(it, op_code);
        par.iteration = it + 1u;
    }

    RSQ_HD void finalize(ReadMeta &meta) const {
        meta.read_len = (uint16_t)par.read_length;
        meta.num_errors = (uint16_t)par.num_errors;
        meta.n_iter_m = (uint16_t)iter_m;
        meta.n_iter_s = (uint16_t)(par.iteration - iter_m - hard_clip);   // every tail iteration emits exactly one base
        meta.hard_clip = (uint16_t)hard_clip;
        meta.tile_id = (uint16_t)tile_id;
        meta.cigar_chars = (uint16_t)cg.chars;
        meta.plain = n_indels ? 0u : 1u;
    }
};

template <class Tab, class Src, class Out>
RSQ_HD void fill_read(const DevSim &S, const Tab &tab, const Stream &st, uint32_t seg, uint32_t tile_id, uint32_t fragment_length, const Src &src, Out &out,
                      ReadMeta &meta) {
    ReadMachine m;
    m.init(S, tab, st, seg, tile_id, fragment_length, src);
    while (m.step(S, tab, st, src, out)) {}
    m.finalize(meta);
}

}  // namespace rsq
