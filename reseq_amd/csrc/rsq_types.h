// rsq_types.h -- plain-data views shared by host code and HIP kernels.
//
// Everything a kernel reads after rsq_sim_create()/rsq_sim_prepare() sits in HBM and is described by
// the DevSim struct below (passed to kernels by value).  Naming follows the reference
// (schmeing/ReSeq): tables are LogArrayResult<N> (ProbabilityEstimates.h:351-557), "sys errors" are
// SimBlock::sys_errors_ (Simulator.h:115), thresholds are non_zero_thresholds_ (Simulator.h:298).
#pragma once
// compiled by hipcc (the library) or at run time by hiprtc (a read kernel for one profile, rsq_spec.h; its built-in header has the fixed-width types and the
// device's math), or by a host compiler (the test-only host emulation)
#if defined(__HIPCC_RTC__)
using __hip_internal::int32_t;
using __hip_internal::int64_t;
using __hip_internal::uint16_t;
using __hip_internal::uint32_t;
using __hip_internal::uint64_t;
using __hip_internal::uint8_t;
typedef unsigned long uintptr_t;
#define RSQ_DEVICE_BUILD 1
#define RSQ_HD __host__ __device__ __forceinline__
#elif defined(__HIPCC__)
#include <stdint.h>
#include <hip/hip_runtime.h>
#define RSQ_DEVICE_BUILD 1
#define RSQ_HD __host__ __device__ __forceinline__
#else
#include <stdint.h>
#define RSQ_DEVICE_BUILD 0
#define RSQ_HD inline
#endif

namespace rsq {

constexpr uint32_t kBlockSize = 1000;          // Simulator.h:254 kBlockSize
constexpr uint32_t kSurBlocks = 3;             // Surrounding.h:17
constexpr uint32_t kSurRange = 10;
constexpr uint32_t kSurStart = 10;
constexpr uint32_t kSurSize = 1u << 20;
constexpr uint32_t kSqFragmentLengthBinSize = 10;   // QualityStats.h:15

// Counter domains of the Philox streams (DESIGN.md "Random streams"); tag = c3 >> 28.
enum : uint32_t { kDomSieve = 1, kDomPair = 2, kDomSysErr = 3, kDomErrModel = 4, kDomReplaceN = 5, kDomRefBias = 6, kDomMethylation = 7 };

// One LogArrayResult<N>: K outcome columns, NM = N-1 conditioning margins.
// margin n: rows[n] rows of stride kp = row_stride(K) doubles at pool[off[n]] (zero pad columns up to whole
// chunks of column pairs); row r = clamp(value - from[n]).
// The margins of one table are contiguous in the pool: off[n+1] = off[n] + rows[n]*kp.
struct DevTable {
    uint32_t k;            // par0_indeces_.size(); 0 = empty table (Draw returns prob_sum 0)
    uint32_t par0_off;     // into the u8 outcome-value pool
    uint32_t from[4];      // limits_[n].first
    uint32_t rows[4];      // limits_[n].second - limits_[n].first
    uint32_t off[4];       // into the double pool (even: 16-byte aligned)
    uint32_t max_value;    // MaxValue()  (ProbabilityEstimates.h:510-517)
    // the two families of the systematic-error chains: single-precision copy for their screened draws (rsq_core.h "screened draw"), margins one after the
    // other at pool32[off32], rows of whole quads.  (The read kernel's families are laid out per family, FamilyGeo below.)
    uint32_t off32;
    uint32_t f32_ok;       // every value is 0 or in [2^-60, 2^29]: the error bound of the screened draw holds
    // indel tables: lo16 | hi16 << 16; a draw with lo16 <= (random word >> 16) < hi16 and margin 0 at its row 0 returns value 0 = no indel whatever the
    // rows of the other margins are (rsq_pack.h certain_no_indel); 0: no such bound.  Error-rate tables (the chains): index into DevSim::chain_sure of the table's
    // ranges [row of margin 0][row of margin 2] for value 0 = rate 0; 0: none
    uint32_t sure_range;
};
static_assert(sizeof(DevTable) == 72, "descriptor layout");
static_assert(__builtin_offsetof(DevTable, par0_off) == 4, "lds_stage_descriptors rewrites word 1 of a descriptor");
constexpr uint32_t kNoLds = 0xFFFFFFFFu;
// A double-precision draw walks a row in chunks of U column pairs: U = 3 for the small tables (K <= 6: base call, indel), 4 otherwise.
#ifndef RSQ_CHUNK_LARGE
#define RSQ_CHUNK_LARGE 4
#endif
constexpr uint32_t kChunkSmall = 3, kChunkLarge = RSQ_CHUNK_LARGE;
RSQ_HD uint32_t chunk_pairs(uint32_t k) { return k <= 2u * kChunkSmall ? kChunkSmall : kChunkLarge; }
RSQ_HD uint32_t chunks_of(uint32_t k, uint32_t u) { return (k + 2u * u - 1u) / (2u * u); }
// Row stride in doubles: whole chunks (zero pad columns, so that the draw needs no masks), even (16-byte rows).
RSQ_HD uint32_t row_stride(uint32_t k) {
    const uint32_t u = chunk_pairs(k), kp = chunks_of(k, u) * 2u * u;
    return (kp & 2u) ? kp : kp + 2u;
}
// Single-precision rows: whole quads of columns (16-byte loads) and a stride == 4 (mod 8) words, so that the rows different lanes
// read in one ds_read_b128 spread over all LDS bank groups.
RSQ_HD uint32_t quads_of(uint32_t k) { return (k + 3u) / 4u; }
RSQ_HD uint32_t row_slot32(uint32_t quads) {
    const uint32_t w = quads * 4u;
    return (w & 4u) ? w : w + 4u;
}
// what the screened draws are compiled for: the quality family with 3, 6 (binned qualities: K <= 12, K <= 24) or 10 to 12 quads per row (K <= 40 .. 48), the base-call and
// indel families with 2 (K <= 8, rows of 32 bytes)
constexpr uint32_t kQuadsSmall = 2, kSlotSmall = 8;
constexpr uint32_t kQualityQuads[] = {3, 6, 10, 11, 12};
// the systematic-error chains (k_sys_chain) screen their two draws as well: dominant error with kQuadsSmall quads, error rate with one of these
constexpr uint32_t kChainQuads[] = {8, 16, 26};
constexpr uint32_t kRingSlots = 2, kRingLag = 1;      // quality rows over the read position of the wave's last steps; a read may lag so many steps (deletions)
constexpr uint32_t kRingRows = kRingSlots + 1u;       // and one slot staged on demand: the rows over the position of a read that lags further (ScreenTables::demand)

// One table family of the read kernel (quality, base call, indel) in the screened layout.  LogArrayResult::Draw clamps every conditioning value to the table's own
// range (AdjustIndeces, ProbabilityEstimates.h:368-380): a value outside it IS the edge row.  So every table of a family can be written over the family's COMMON
// ranges -- from = the smallest first value, rows up to the largest last one, a table's edge rows repeated where its own range is shorter -- without changing a
// single draw, and a lane finds its four rows by arithmetic on (table number, value) alone: no descriptor is read, and all of this is a constant of the profile
// (literals in a kernel compiled for the profile, rsq_spec.h).  An empty table, or one outside the screen's preconditions, is all zeros: its draws come out
// undecided (prob_sum below 2^-30) and take the double-precision route, which reads the table's own descriptor.
struct FamilyGeo {
    uint32_t from[4];            // common first value of margin n
    uint32_t last[4];            // common last row of margin n (rows - 1)
    uint32_t before[4];          // rows of the margins in front of margin n within a table
    uint32_t table_rows;         // rows of one table (all margins)
    uint32_t off32;              // DevSim::pool32: [table of the profile][table_rows][slot]
    uint32_t lds;                // image: the staged leading margins, [table of the image][lds_rows][slot]; kNoLds: not staged (the indel family's margin 0 when it does not fit)
    uint32_t lds_rows;           // quality: rows of margins 0 and 1; base call, indel: rows of margin 0
    uint32_t lds_stride;         // floats from one table's staged rows to the next table's: lds_rows * slot, padded to an ODD number of 16-byte groups (below)
    uint32_t lds2;               // base call: margin 2 over the number of errors, [table of the image][last[2] + 1][slot]; kNoLds: read from device memory
    uint32_t lds2_stride;        // its table stride, padded likewise
    uint32_t values;             // image, in BYTES from its start: the outcome value of every column, [table of the image][slot] (0 in the pad columns)
    uint32_t values_src;         // DevSim::par0, bytes: the same for all tables of the profile (what an image copies)
};

// LDS image of k_fill_reads (rsq_kernels.h "LDS staging"), in single precision: one per template segment holding the tables of all tiles
// (img_tiles == n_tiles; built once per workgroup) or, when those do not fit, one per (segment, tile) (img_tiles == 1; a workgroup builds the
// image of the tile whose reads it is about to serve).  Contents:
// the table descriptors of the image's tiles and of the indel tables, their outcome values (both for the double-precision route), the outcome values by column
// of the three families (FamilyGeo::values), margins 0+1 of the quality tables
// (sequence quality, previous quality), margin 0 of its base-call tables (quality), the first rows of the error-rate margins
// (quality margin 3, base-call margin 3), and as far as the 160 KiB reach margin 0 of the indel tables and margin 2 of the base-call
// tables (number of errors).  Offsets and sizes in 32-bit words.
struct LdsPlan {
    uint32_t mask;               // the read kernel's template argument: quads_q when the image exists and the kernel draws screened; 0: double
                                 // precision from HBM only
    uint32_t img_tiles;          // tiles per image: n_tiles, or 1 (reads binned by tile, one tile per workgroup)
    uint32_t binned;             // the read kernels run binned by tile (img_tiles < n_tiles, or asked for by option image_tiles = 1)
    uint32_t desc_words;         // size of the descriptor area: the descriptors, then the outcome values (par0) of the image's tables
    uint32_t par0_words;         // size of the outcome-value area in the image: the indel tables' values (par0_indel_bytes, from byte 0 of the pool), then
    uint32_t par0_indel_bytes;   // those of the image's tiles (contiguous in the pool from the first quality table's par0_off on)
    uint32_t slot_q, slot_b, slot_i;        // row slot (floats) of the quality / base-call / indel family
    uint32_t quads_q;            // 16-byte groups of a quality row that hold columns: one of kQualityQuads
    uint32_t rate_rows_q, rate_rows_b;      // rows 0..n-1 of quality margin 3 / base-call margin 3 are staged
    uint32_t q3_off, b3_off;     // [4 img_tiles][rate_rows_q] quality slots, [20 img_tiles][rate_rows_b] base-call slots
    // Table strides of the staged blocks are ODD numbers of 16-byte groups.  A ds_read_b128 serves 16 lanes per LDS cycle, a lane's 16 bytes from one of 16 bank groups
    // (address / 16 mod 16), and lanes that read DIFFERENT addresses in one bank group take turns.  The lanes of a wave sit at (nearly) one read position and mostly
    // at error rate 0, so what differs between them is the TABLE (the template base) and the quality rows: with a table stride that is a multiple of 16 groups -- 80
    // quality rows of 11 groups, 64 error-rate rows of 11 groups -- the same row of the four tables fell into ONE bank group, a four-way conflict on every read of
    // the error-rate margin.  With an odd stride the tables' copies of a row lie in four different groups.
    uint32_t q3_stride, b3_stride;
    uint32_t ring_off, ring_stride;      // per wave: kRingRows x [4 img_tiles quality slots]: the quality rows over the read positions of the wave's current steps, and one slot staged on demand
    uint32_t total_words;        // size of the image
    FamilyGeo q, b, i;           // the three families' common geometry
};

// Fragment produced by the coverage sieve: one simulated read pair (Simulator.cpp:2249-2357 -> CreateReads).
struct Fragment {
    uint32_t seq;          // reference sequence id
    uint32_t start;        // forward start position
    uint32_t len;          // fragment length (0 = adapter-only pair)
    uint16_t dup;          // duplicate index at this (start,len,strand) site
    uint8_t strand;
    uint8_t allele;        // 0 without variants
    uint32_t block;        // block number printed in the read id
    uint32_t number;       // read_number within the block (1-based)
};

// Per-read result meta data written by the read kernel (sequence, qualities and CIGAR ops are separate arrays).
struct ReadMeta {
    uint16_t read_len;
    uint16_t num_errors;
    uint16_t n_iter_m;     // state-machine iterations spent in the template ('M') part
    uint16_t n_iter_s;     // iterations spent in the adapter ('S') part
    uint16_t hard_clip;    // length of the 'H' element (poly-A tail + overrun bases)
    uint16_t tile_id;
    uint16_t cigar_chars;  // length of the CIGAR string
    uint16_t plain;        // 1: no insertion or deletion anywhere, the CIGAR follows from the counts above without the stored ops
};

// One variant of the reference (Reference.h:24-62) as the kernels see it: var_seq_ is var_bases[off, off + len) (len 0: deletion,
// 1: substitution, more: the substituted base followed by inserted bases).  The systematic errors drawn for those bases
// (SysErrorVariant::var_errors_, Simulator.h:91-106) are var_err_fwd / var_err_rev[off + k] for the k-th base in the strand's
// drawing order (the reverse strand draws the complemented bases last to first), dom | rate << 8 like the tracks.
struct DevVariant {
    uint32_t pos;
    uint32_t len;
    uint32_t off;
    uint32_t pad;
    uint64_t allele[2];    // bit a: the variant is in allele a
};
// One replacement of ONE allele (a variant the allele has), an entry of the allele's coordinate map (rsq_variants.h).  The entries of a
// (sequence, allele) are sorted by position and end with a sentinel {sequence length, number of the sequence's variants, totals}.
struct AlleleVar {
    uint32_t pos;          // reference position
    uint32_t vid;          // the variant, index among the sequence's variants
    int32_t shift;         // allele coordinate of `pos` minus pos = sum of (len - 1) over the allele's earlier entries
    int32_t gc;            // G/C of the allele in front of this entry minus G/C of the reference in front of pos
};
// what CreateReads gets of SimulateFromGivenBlock beyond the Fragment when variants are loaded (Simulator.cpp:2334-2337)
struct FragmentVar {
    uint32_t end;          // cur_end_position = start + length + end_pos_shift_[allele]
    uint32_t sub;          // pass at the start position (> 0: the fragment starts inside inserted bases)
    int32_t start_var;     // bias_mod.StartVariant()
    uint32_t start_var_pos;
    int32_t end_var;       // bias_mod.EndVariant(variants, cur_end_position, allele)
    uint32_t end_var_pos;
};
// an extra pass of SimulateFromGivenBlock's do-while loop at a start position (CheckForInsertedBasesToStartFrom, :1870-1896)
struct ExtraStart {
    uint32_t pos;          // start position in the sequence
    uint32_t sub;          // 1, 2, ... in loop order at this position
    int32_t first_variant_id;
    uint32_t start_variant_pos;
};

struct DevAdapters {
    uint32_t n;
    const uint8_t *seqs;          // base codes, concatenated
    const uint32_t *seq_ptr;      // [n+1]
    const uint16_t *sys;          // dom | rate<<8 per adapter base (adapter_sys_error_, Simulator.h:294)
    const double *adapter_cp;     // cumulative probabilities over SignificantCounts
    const double *cut_cp;         // concatenated per adapter
    const uint32_t *cut_ptr;      // [n+1]
    const uint32_t *cut_from;     // [n]
};

struct DevReadLengths {
    uint32_t fixed;               // ReadLengths(seg).size() == 1 -> from(); else 0
    uint32_t to;                  // ReadLengths(seg).to()
    uint32_t row_first;           // first fragment length with a row
    uint32_t rows;
    const uint32_t *row_ptr;      // CSR over fragment lengths
    const uint32_t *row_from;
    const uint64_t *values;
};

struct DevSim {
    uint64_t seed;
    // ---- tables
    const double *pool;
    const uint32_t *chain_sure;    // error-rate tables: lo16 | hi16 << 16 per (table, row of margin 0, row of margin 2), DevTable::sure_range (rsq_pack.h)
    const float *pool32;           // single-precision copies: the read kernel's three families (FamilyGeo::off32), the chains' two (DevTable::off32)
    const uint8_t *par0;
    const DevTable *quality;       // [2][n_tiles][4]
    const DevTable *seq_quality;   // [2][n_tiles]
    const DevTable *base_call;     // [2][n_tiles][4][5]
    const DevTable *dom_error;     // [4][5][5]
    const DevTable *error_rate;    // [4][5]
    const DevTable *indels;        // [2][6]
    LdsPlan lds;
    uint32_t force_exact;            // tests (RSQ_FORCE_EXACT): the read kernel's screen decides nothing, every draw takes the double-precision route behind it
    uint32_t chain_quads;            // quads per row of the error-rate tables' single-precision copy (one of kChainQuads); 0: the chains draw in double precision only
    uint32_t n_tiles;
    uint8_t phred_offset;
    uint16_t max_len_deletion;
    uint32_t reset_distance;
    uint16_t sys_gc_range;
    // ---- static categorical draws (GeneralRandomDistributions, Simulator.h:146-213)
    const double *tile_cp;
    const uint16_t *tiles;
    DevAdapters adapters[2];
    DevReadLengths read_lengths[2];
    const double *polya_cp;
    uint32_t polya_n, polya_from;
    double overrun_cp[4];
    const uint64_t *insert_lengths;  // dense from 0 .. insert_to
    // ---- reference: 2 bit per base, 32 bases per uint64_t word, every sequence starts on a word
    uint32_t n_seqs;
    const uint64_t *ref_words;
    const uint32_t *gc_prefix;       // per reference word: G/C bases in the sequence's earlier words (indexed like ref_words)
    const uint64_t *seq_word_off;    // [n_seqs]
    const uint32_t *seq_len;         // [n_seqs]
    const uint64_t *seq_base_off;    // [n_seqs] offset of the sequence in the systematic-error tracks
    const uint16_t *sys_fwd;         // dom | rate<<8 per forward position
    const uint16_t *sys_rev;         // same for the reverse-complement strand, index = L-1-forward position
    // ---- variants (-V): copy 1 + a of the packed reference and of gc_prefix (hap_stride entries apart) is allele a with its
    // substitutions applied; copy 0 stays the reference itself (systematic-error chains, bias sums, wrapped surroundings)
    uint32_t num_alleles;            // Reference::NumAlleles(), 1 without variants
    uint32_t variants_loaded;        // Reference::VariantsLoaded(): 0 no, 1 substitutions only (allele copies), 2 any kind (allele coordinate maps, rsq_variants.h)
    uint64_t hap_stride;             // 0 unless variants_loaded == 1
    const DevVariant *variants;      // sorted by position within each sequence
    const uint32_t *var_ptr;         // [n_seqs + 1]
    const uint8_t *var_bases;
    const uint32_t *var_bases_gc;    // variants_loaded == 2: [bases + 1] G/C among var_bases[0, i)
    const AlleleVar *allele_map;     // variants_loaded == 2: the coordinate maps, (sequence, allele) after (sequence, allele)
    const uint32_t *allele_map_ptr;  // [n_seqs * num_alleles + 1]
    const uint16_t *var_err_fwd, *var_err_rev;
    uint32_t *walk_error;            // set by a read whose systematic-error walk leaves its sequence (the reference dereferences a NULL block there)
    // variants_loaded == 2: slots of the sieve = start positions plus the extra passes inside inserted bases, in loop order
    const ExtraStart *extra;         // sorted by (sequence, pos, sub)
    const uint32_t *block_extra_ptr; // [total_blocks + 2] extras before block b (index b, 1-based)
    // ---- coverage model
    uint32_t insert_from;            // max(1, InsertLengths().from())
    uint32_t insert_to;              // InsertLengths().to()
    const double *thresholds;        // [n_groups][insert_to][2]
    const double *gap_q;             // [n_groups][insert_to] running product of the zero thresholds inside a segment (the sieve's gap draws)
    const uint32_t *gap_seg_end;     // [n_groups][insert_to] one past the last length of the length's segment
    const uint32_t *coverage_group;  // [n_seqs]
    const double *ref_seq_bias;      // [n_seqs]
    const double *insert_lengths_bias; // dense [insert_to]
    const double *gc_bias;           // dense [101]
    const double *sur_bias;          // [3][1<<20]
    double dispersion[2];
    double bias_normalization;
    // ---- blocks
    uint32_t total_blocks;
    const uint32_t *block_seq;       // [total_blocks+1] sequence of block id b (index b, 1-based)
    const uint32_t *first_block;     // [n_seqs]
    // bisulfite conversion (--methylation): unmethylated regions [first, second) of every sequence as CSR, C->T probability each
    const uint32_t *meth_ptr;        // [n_seqs + 1], nullptr without a methylation file
    const uint32_t *meth_first, *meth_second;
    const double *meth_rate;
};

// What the read kernel knows of the profile's layout -- the LDS plan and the families' common geometry (LdsPlan, FamilyGeo) and a few scalars of DevSim -- is read
// through these two macros: from the kernel argument in the library's own build; as LITERALS in a kernel compiled for one profile (rsq_spec.h: RSQ_SPEC defines
// namespace rsq::spec with the same names), where row addresses fold into multiply-adds with constants and the values no longer occupy scalar registers.
#if defined(RSQ_SPEC)
#define RSQ_PLAN(S, field) (::rsq::spec::lds.field)
#define RSQ_SIM(S, field) (::rsq::spec::field)
#else
#define RSQ_PLAN(S, field) ((S).lds.field)
#define RSQ_SIM(S, field) ((S).field)
#endif

struct NameTable {                       // first parts of the reference ids + the record base identifier
    const char *names;
    const uint32_t *name_ptr;
    char base_identifier[64];
    uint32_t base_len;
};

}  // namespace rsq
