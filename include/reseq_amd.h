/*
 * reseq_amd.h -- C ABI of libreseq_amd.so: the MI355X (gfx950) implementation of ReSeq's read-simulation
 * hot path.  Plain pointers and sizes only; every entry point returns 0 on success or a negative RSQ_E* code
 * (message via rsq_last_error()).  All functions are thread-compatible: one rsq_sim per GPU per thread.
 *
 * The reference (schmeing/ReSeq v1.1) has no plugin/FFI boundary: `main` calls three C++ methods of
 * reseq::Simulator by value-typed references (reseq/Simulator.h:456-458).  Each entry point below names the
 * reference interface it stands in for; INTEGRATION.md shows the binding a ReSeq maintainer would add.
 *
 * Pointers named *_dev are DEVICE pointers on the GPU the rsq_sim was created for; everything else is host
 * memory.  `stream` is a hipStream_t passed as void* (NULL = the default stream).
 */
#ifndef RESEQ_AMD_H
#define RESEQ_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rsq_profile rsq_profile;      /* DataStats + ProbabilityEstimates, simulation subset */
typedef struct rsq_ref rsq_ref;              /* reseq::Reference */
typedef struct rsq_sim rsq_sim;              /* reseq::Simulator bound to one GPU */

enum {
    RSQ_OK = 0,
    RSQ_EINVAL = -1,      /* bad argument / inconsistent input */
    RSQ_EIO = -2,         /* file could not be read / parsed */
    RSQ_ENODEV = -3,      /* no usable HIP device: there is NO CPU fallback */
    RSQ_EHIP = -4,        /* a HIP call failed */
    RSQ_ENOSPC = -5,      /* caller's output buffer too small; required sizes are still reported */
    RSQ_ESTATE = -6       /* call order violated (e.g. simulate before prepare) */
};

const char *rsq_last_error(void);
const char *rsq_last_warning(void);          /* non-fatal remarks of the last rsq_profile_load* or rsq_sim_create call ("" if none) */
const char *rsq_version(void);
/* Testing and measurement switches (reseq_amd/csrc/rsq_host.h `Options`; README "Options").  The library takes no switch from the environment (only the default place of the
 * kernel cache, rsq_set_kernel_cache_dir): an embedding program sets a switch explicitly.  A simulator takes a copy of ALL values when it is created (rsq_sim_create)
 * and reads only that copy afterwards -- several simulators of one process (one per GPU and host thread, see rsq_sim_create) never look at a value another thread
 * is changing; set the switches before the simulators are created.  The loaders, which have no simulator (rsq_ref_load_fasta, rsq_ref_read_variants,
 * rsq_sim_read_methylation's parser: serial_fasta, fasta_stretch, serial_parse, parse_stretch, trace_load), read the process-wide values when they run;
 * mapped_parses is a counter they add to.  Reads and FASTQ bytes never depend on any of them -- they choose between equivalent routes (e.g. fill_mode 0: every
 * per-base draw in double precision from device memory, the reference's own recipe, ProbabilityEstimates.h:481-508).  RSQ_EINVAL for an unknown name. */
int rsq_set_option(const char *name, int64_t value);
int rsq_get_option(const char *name, int64_t *value);
/* number of visible HIP devices, or RSQ_ENODEV */
int rsq_device_count(void);

/* ---- profile: stands in for DataStats::Load + PrepareProcessing (reseq/DataStats.cpp:1280-1340) and
 *      ProbabilityEstimates::Load + PrepareResult (reseq/ProbabilityEstimates.cpp:961-1065).
 *      `path` is an RSQP container (reseq_amd/container.py documents the layout) or a `.reseq` archive (see rsq_profile_load_reseq). */
int rsq_profile_load(const char *path, rsq_profile **out);
/* *yes = 1 if the file begins like a Boost text archive (ReSeq's own .reseq), 0 otherwise (an RSQP container, or unreadable) */
int rsq_profile_is_reseq_archive(const char *path, int *yes);
/* ReSeq's own profile files: `stats_path` = the `.reseq` Boost text archive DataStats::Save writes (reseq/DataStats.cpp:1302-1320, member
 * list DataStats.h:180-212), `ipf_path` = the `.reseq.ipf` archive of ProbabilityEstimates::Save (reseq/ProbabilityEstimates.cpp:1047-1065,
 * member list ProbabilityEstimates.h:1475-1483; NULL = "<stats_path>.ipf", main.cpp:837).  Does what `main` does between loading and
 * Simulator::Simulate with --ipfIterations 0: DataStats::PrepareProcessing (total reads, AdapterStats::SumCounts / PrepareSimulation,
 * ErrorStats::PrepareSimulation) and ProbabilityEstimates::PrepareResult (FullExpansion, GetResults, ImputeMissingValues).  Fitting is not
 * part of this build: tables whose stored precision is above `ipf_precision_percent` (--ipfPrecision, default 5) are used as stored and
 * reported through rsq_last_warning().  rsq_profile_load recognises such a file by its first bytes and forwards here. */
int rsq_profile_load_reseq(const char *stats_path, const char *ipf_path, double ipf_precision_percent, rsq_profile **out);
/* Diagnosis of ReSeq's own profile files (DataStats::Save / ProbabilityEstimates::Save, reseq/DataStats.cpp:1302-1320, ProbabilityEstimates.cpp:1047-1065): a text
 * table of where every serialized C++ type's class information sits in the two archives (byte, tracking, version, type, member path of the first object) and,
 * if a file does not parse, the message naming the member path and type at which it stops.  The reader's token rules cannot be validated against a
 * Boost-written file in the build image (INTEGRATION.md "Profile files"); this is what to send back when a real profile fails.  `ipf_path` NULL: stats_path + ".ipf".
 * Writes at most cap bytes (NUL-terminated) and the full length to *need. */
int rsq_profile_archive_layout(const char *stats_path, const char *ipf_path, char *out, size_t cap, size_t *need);
/* writes the prepared profile (result tables, not the fit) as an RSQP container */
int rsq_profile_save(const rsq_profile *p, const char *path);
/* rsq_profile_save_reseq: the writer that ReSeq's own files lack here -- `stats_path` / "<stats_path>.ipf" (ipf_path NULL) as Boost text archives under the token
 * rules rsq_archive.h recalls (DataStats::Save, DataStats.cpp:1302-1320; ProbabilityEstimates::Save, ProbabilityEstimates.cpp:1047-1065).  The statistics and fits
 * written are those whose prepared form -- PrepareProcessing / PrepareResult -- is the loaded profile; what the simulation never reads is default-constructed.
 * rsq_profile_load_reseq reads the pair back to the same tables bit for bit.  Whether the ORIGINAL binary reads it cannot be checked in this image: a user who can
 * run `reseq queryProfile -s <path>` on a written pair is asked to tell (INTEGRATION.md "Profile files").  creation_time 0: now. */
int rsq_profile_save_reseq(const rsq_profile *p, const char *stats_path, const char *ipf_path, uint64_t creation_time);
void rsq_profile_free(rsq_profile *p);
/* ProbabilityEstimates::ChangeErrorRate / RemoveSubstitutionErrors / RemoveInDelErrors
 * (reseq/ProbabilityEstimates.h:1516-1549; CLI --errorMutliplier, --noSubstitutionErrors, --noInDelErrors) */
int rsq_profile_change_error_rate(rsq_profile *p, double error_multiplier);
int rsq_profile_remove_substitution_errors(rsq_profile *p);
int rsq_profile_remove_indel_errors(rsq_profile *p);
/* small getters the callers need to size buffers */
int rsq_profile_max_read_length(const rsq_profile *p, uint32_t *out);
int rsq_profile_num_tiles(const rsq_profile *p, uint32_t *out);
/* `reseq queryProfile` (reseq/main.cpp:481-610): ErrorStats::MaxLenDeletion, the stored reference sequence biases (n = their number;
 * out may be NULL to ask for n) */
int rsq_profile_max_len_deletion(const rsq_profile *p, uint32_t *out);
int rsq_profile_ref_seq_bias(const rsq_profile *p, double *out, size_t cap, size_t *n);

/* ---- reference: Reference::ReadFasta (reseq/Reference.cpp:758) and Reference::ReplaceN (:813) */
int rsq_ref_load_fasta(const char *path, rsq_ref **out);
int rsq_ref_replace_n(rsq_ref *r, uint64_t seed);
void rsq_ref_free(rsq_ref *r);
int rsq_ref_num_sequences(const rsq_ref *r, uint32_t *out);
int rsq_ref_sequence_length(const rsq_ref *r, uint32_t seq, uint32_t *out);
/* ReferenceIdFirstPart (reseq/Reference.cpp:476-480): the id up to the first blank, NUL-terminated */
int rsq_ref_sequence_name(const rsq_ref *r, uint32_t seq, char *out, size_t cap);
/* Reference::WriteFasta (reseq/Reference.cpp:896-916; `reseq replaceN` writes the reference after ReplaceN): FASTA, gzip when the
 * name ends in .gz */
int rsq_ref_write_fasta(const rsq_ref *r, const char *path);
/* Reference::PrepareVariantFile + ReadFirstVariants / ReadVariants (reseq/Reference.cpp:126-420,1003-1077): loads a VCF, split into
 * single-position variants per sequence (Reference::Variant, Reference.h:24-62).  A simulator created from such a reference simulates
 * per allele (Simulator::SimulateFromGivenBlock with VariantsLoaded, Simulator.cpp:2249-2357; read ids carry "_allele<a>").  What the
 * kernels simulate: substitutions, insertions and deletions, up to 128 alleles (Reference::Variant::kMaxAlleles), also together with
 * rsq_sim_read_methylation (one conversion rate per allele). */
int rsq_ref_read_variants(rsq_ref *r, const char *path);
int rsq_ref_num_alleles(const rsq_ref *r, uint32_t *out);
int rsq_ref_num_variants(const rsq_ref *r, uint32_t seq, uint32_t *out);
/* variant `index` of sequence `seq`: position, its bases as letters (NUL-terminated, "" = deletion), the 128 allele bits */
int rsq_ref_get_variant(const rsq_ref *r, uint32_t seq, uint32_t index, uint32_t *position, char *var_seq, size_t var_seq_cap, uint64_t allele_bits[2]);
/* copies the base codes (A=0,C=1,G=2,T=3,N=4) of one sequence into out[len] */
int rsq_ref_get_codes(const rsq_ref *r, uint32_t seq, uint8_t *out, uint32_t len);

/* ---- simulator.  rsq_sim_create packs the profile tables and the 2-bit reference into HBM of `device`
 *      (`ref` may be NULL for the error-model-only mode).  The profile and reference may be freed afterwards. */
int rsq_sim_create(const rsq_profile *p, const rsq_ref *ref, int device, rsq_sim **out);
void rsq_sim_free(rsq_sim *s);
/* A simulator reads the option switches (rsq_set_option) from the copy it took when it was created.  rsq_sim_take_options copies them again, by the thread that
 * drives the simulator: for a caller that changes a call-shaping switch (overlap, job_chunk_bytes, job_write_direct, host_gzip, fill_waves, the pre-pass switches)
 * between the calls of one simulator.  What was decided at creation -- the packed tables and the compiled kernel: fill_mode, image_tiles, rate_rows,
 * min_quality_quads, specialize -- stays. */
int rsq_sim_take_options(rsq_sim *s);

/* Reference::ReferenceSequence (reseq/Reference.cpp:483-496 plain, :498-567 with variants; reseq/Reference.h) as the kernels compute it on the device: the
 * `frag_length` bases of `allele` that start at `start_pos` (reversed: the reverse complement of those that END in front of `start_pos`), beginning
 * `first_variant_pos` bases inside the inserted bases of variant `first_variant_id` of the sequence when that is not 0 -- the arguments GetOrgSeq passes
 * (reseq/Simulator.cpp:1909-1914).  Base codes (A=0,C=1,G=2,T=3) into out[frag_length].  For checks against the reference's known answers
 * (reseq/ReferenceTest.cpp:277-326) and for callers that want a template without simulating it; RSQ_EINVAL when the stretch leaves the allele's sequence. */
int rsq_sim_reference_sequence(rsq_sim *s, uint32_t seq, uint32_t start_pos, uint32_t frag_length, int reversed, int32_t first_variant_id, uint32_t first_variant_pos, uint32_t allele,
                               uint8_t *out, size_t cap);

/* Everything Simulator::Simulate does before "Starting read generation" (reseq/Simulator.cpp:2687-2826):
 * number of pairs (num_read_pairs, or coverage, or the profile's corrected coverage when both are 0), adapter-only
 * share, CalculateBiasNormalization, systematic errors of adapters and of both strands of every sequence.
 * With ref == NULL only the adapter part runs (Simulator::SimulateErrorModelOnly, :2951-2977).
 * ref_bias_mode = RefSeqBiasSimulation (FragmentDistributionStats.cpp:3352-3500): 0 keep (falls back to 1 when the counts differ),
 * 1 no bias, 2 draw with replacement from the stored biases, 3 read the file given to rsq_sim_set_ref_bias_file. */
int rsq_sim_prepare(rsq_sim *s, uint64_t seed, uint64_t num_read_pairs, double coverage, int ref_bias_mode, const char *record_base_identifier, void *stream);

/* The same pre-pass for ONE rank of a sharded job (one process per GPU, the rank simulates blocks [block_lo, block_hi)): the rank computes
 * its own share of both pre-passes and exchanges only small arrays with the other ranks; the results equal rsq_sim_prepare's bit for bit.
 *   rsq_sim_prepare_plan            pair counts, block numbering, coverage groups (host work; reseq/Simulator.cpp:2687-2745)
 *   rsq_sim_bias_partials           CalculateBiasNormalization / SumBias (reseq/FragmentDistributionStats.cpp:3504-3582, reseq/Reference.cpp:622-659):
 *                                   partial sums and maxima of the rank's chunks of start positions, zero elsewhere; *n entries each
 *                                   (rsq_sim_bias_partials(s, 0, 0, NULL, NULL, 0, &n, NULL) asks for n).  The ranks add their arrays up
 *                                   (all-reduce SUM: every entry is non-zero on exactly one rank, so the sum is exact) ...
 *   rsq_sim_prepare_normalization   ... and combine them in chunk order: normalisation, spline, thresholds
 *   rsq_sim_prepare_sys_errors      SetSystematicErrors (reseq/Simulator.h:337-382) for the positions the rank's reads can touch.  The chains are
 *                                   entered with in_state[0] (forward chain, from the rank on the left) and in_state[1] (reverse chain, from the
 *                                   rank on the right); out_state[0] is what the rank on the right needs, out_state[1] what the rank on the left
 *                                   needs.  Called again with other in_state it redoes only what depends on them.  Ranks repeat: all-gather of
 *                                   out_state, take the neighbours' values, call again -- until no rank's in_state changed (at most world-1 rounds;
 *                                   states of unaffected chains are final after the first call).  A rank with an empty range passes in_state on.
 *   rsq_sim_prepare_finish          with variants: the systematic errors of the variants' bases inside the rank's share (their error-region state is
 *                                   folded from the state the rank's chains were entered with); marks the simulator prepared. */
int rsq_sim_prepare_plan(rsq_sim *s, uint64_t seed, uint64_t num_read_pairs, double coverage, int ref_bias_mode, const char *record_base_identifier);
int rsq_sim_bias_partials(rsq_sim *s, uint32_t block_lo, uint32_t block_hi, double *sums, double *maxes, size_t cap, size_t *n, void *stream);
int rsq_sim_prepare_normalization(rsq_sim *s, const double *sums, const double *maxes, size_t n);
int rsq_sim_prepare_sys_errors(rsq_sim *s, uint32_t block_lo, uint32_t block_hi, const uint32_t in_state[2], uint32_t out_state[2], void *stream);
int rsq_sim_prepare_finish(rsq_sim *s);

/* --methylation: Reference::PrepareMethylationFile + ReadMethylation (reseq/Reference.cpp:1132-1310) for a reference without variants:
 * extended BED "sequence start end methylation" in reference order.  Afterwards rsq_sim_pairs applies Simulator::CTConversion
 * (reseq/Simulator.cpp:1925-2002,2219-2247) to the templates of every fragment: inside the listed regions each C becomes T with
 * probability 1 - methylation, drawn once per (start, length, strand) site so that duplicates share the conversion. */
int rsq_sim_read_methylation(rsq_sim *s, const char *path);

/* --refBiasFile: lines "identifier bias" (UpdateRefSeqBias kFile, FragmentDistributionStats.cpp:3386-3495); call before rsq_sim_prepare */
int rsq_sim_set_ref_bias_file(rsq_sim *s, const char *path);
int rsq_sim_get_ref_seq_bias(const rsq_sim *s, double *out, size_t n);          /* [n_sequences], after prepare */

typedef struct {
    uint64_t total_pairs;           /* total_pairs_ after removing the adapter-only pairs */
    uint64_t adapter_only_pairs;    /* num_adapter_only_pairs_ */
    uint32_t total_blocks;          /* forward blocks of 1000 start positions; ids run 1..total_blocks */
    uint32_t n_coverage_groups;
    uint32_t insert_to;             /* InsertLengths().to() */
    uint32_t sys_chain_passes;      /* passes the speculative systematic-error chains needed */
    double bias_normalization;
} rsq_sim_info;
int rsq_sim_get_info(const rsq_sim *s, rsq_sim_info *out);
/* How the read kernels (FillRead, Simulator.cpp:454-594) reach the profile's tables on this simulator: quality_quads = 16-byte groups per quality row of
 * the screened single-precision draws, 0 = every per-base draw in double precision from device memory (rsq_last_warning() after rsq_sim_create says
 * why); image_tiles = tiles whose tables one workgroup's local-memory image holds: all of the profile's (reads of any tile served by any workgroup) or 1
 * (reads binned by the tile they draw, Simulator.h:176-181, one tile per workgroup at a time); image_bytes = size of that image. */
int rsq_sim_get_fill_plan(const rsq_sim *s, uint32_t *quality_quads, uint32_t *image_tiles, uint32_t *image_bytes);
/* One load per host.  Simulator::Simulate has one process and one copy of the reference (reseq/Simulator.cpp:2687-2700); a job of one process per GPU would read
 * and pack the same FASTA, variant and methylation files once per rank, on the cores the ranks of a host share.  rsq_sim_export_reference writes what THIS simulator
 * keeps of those files -- sequence names and lengths, the packed bases and G/C prefix sums (of every allele's copy), variants, allele maps, methylation regions; the
 * result of rsq_sim_create with a reference and of rsq_sim_read_methylation -- to `path` (a file in /dev/shm, say), complete before it appears under its name;
 * rsq_sim_import_reference gives a simulator created WITHOUT a reference (rsq_sim_create(profile, NULL, ...)) exactly that state: both end in the same device arrays
 * through the same code, so the reads do not depend on which of the two a rank did.  The file records the profile values the packing used; another profile's
 * simulator is refused (RSQ_EINVAL).  reseq_amd/simulate.py: the first rank of every host loads and exports, the others import. */
int rsq_sim_export_reference(rsq_sim *s, const char *path);
/* lengths of the simulator's reference sequences (what a launcher that imported the reference balances the ranks' block ranges with); out NULL: only the count */
int rsq_sim_get_sequence_lengths(const rsq_sim *s, uint32_t *out, size_t cap, uint32_t *n_sequences);
int rsq_sim_import_reference(rsq_sim *s, const char *path);
/* The read kernel COMPILED FOR THIS SIMULATOR'S PROFILE.  What a loaded profile fixes -- the local-memory plan, the value ranges and row counts of its quality /
 * base-call / indel tables, tiles, phred offset -- are loop bounds and address factors of LogArrayResult::Draw (reseq/ProbabilityEstimates.h:481-508, members of
 * the loaded tables there).  The library carries the kernels' source; with libhiprtc present it compiles them with those values as literals -- per kernel variant
 * when the variant is first launched, kept for the simulator's life and, as a code object, in the kernel cache directory.  This call does it NOW, so that no
 * compilation falls into a timed region (rsq_sim_prepare does it for `kind` 0 by itself): kind 0 = read pairs (rsq_sim_pairs ...), 1 = seqToIllumina records
 * (rsq_sim_error_model ...).  *specialized = 1: the profile's own kernel is in place; 0: the library's own instantiation (any profile) runs -- option
 * `specialize` 0, no libhiprtc, no table image, or a failed compilation; rsq_last_warning() says which, and how long compiling took.  The reads are the same bytes
 * either way. */
int rsq_sim_specialize(rsq_sim *s, int kind, int *specialized);
/* Directory of the compiled kernels' code objects (files rsq_spec_*_<hash>.hsaco; the hash covers sources, literals, variant, device architecture and compiler
 * version).  Default: $XDG_CACHE_HOME/reseq_amd, else ~/.cache/reseq_amd -- the one thing the library takes from the environment; "" or NULL: keep nothing on disk. */
int rsq_set_kernel_cache_dir(const char *path);
/* Host only, no device needed: compiles the read kernel for `p` as rsq_sim_specialize would for a device of architecture `arch` ("gfx950"), from a plan packed
 * into host memory; the code object goes to `out_path` (NULL: nowhere; the kernel cache is used and filled as usual; a name ending in ".hip": the program text itself
 * is written instead, nothing is compiled -- `hipcc -I reseq_amd/csrc` builds it outside).  kind as above; with_variants: the variant
 * of the read-pair kernel for a reference with variants; binned: the variant that serves one tile per workgroup (forced when the profile's tiles do not fit one
 * image).  *code_bytes = size of the code object, *seconds (may be NULL) = compilation time, 0 from the cache.  For build checks (hiprtc cross-compiles) and for
 * reading the generated code (tools/kernel_resources.py --code-object).  RSQ_EINVAL with the compiler's log when it fails. */
int rsq_profile_compile_read_kernel(const rsq_profile *p, int kind, int with_variants, int binned, const char *arch, const char *out_path, size_t *code_bytes, double *seconds);

/* pre-pass results, for inspection and stage-wise parity tests */
int rsq_sim_get_thresholds(const rsq_sim *s, double *out, size_t n);           /* [groups][insert_to][2] */
int rsq_sim_get_norm_by_len(const rsq_sim *s, double *out, size_t n);          /* [insert_to] */
int rsq_sim_set_normalization(rsq_sim *s, double bias_normalization, const double *thresholds, size_t n);
int rsq_sim_get_sys_errors(const rsq_sim *s, int reverse_strand, uint32_t seq, uint8_t *dom_out, uint8_t *rate_out, uint32_t len);
int rsq_sim_get_adapter_sys_errors(const rsq_sim *s, int template_segment, uint32_t adapter, uint8_t *dom_out, uint8_t *rate_out, uint32_t len);

/* Simulator::CreateSystematicErrorProfile (--writeSysError, reseq/Simulator.cpp:2597-2653): draws both strands of every sequence and
 * writes them as FASTQ (two records per sequence, "<id> reverse" first; seq = dominant error, qual = error percent, :2562-2588).
 * Needs no rsq_sim_prepare and invalidates an earlier one. */
int rsq_sim_create_sys_error_profile(rsq_sim *s, uint64_t seed, const char *path, void *stream);
/* --readSysError (LoadSysErrorRecord reseq/Simulator.cpp:750-769, ReadSystematicErrors Simulator.h:326-335): after rsq_sim_prepare,
 * replaces the drawn tracks of the reference strands by the file's; RSQ_EINVAL with the reference's message on a length mismatch. */
int rsq_sim_read_sys_errors(rsq_sim *s, const char *path);

/* One simulated fragment = one read pair (SimulateFromGivenBlock, reseq/Simulator.cpp:2249-2357). */
typedef struct {
    uint32_t seq, start, len;
    uint16_t dup;
    uint8_t strand, pad;
    uint32_t block, number;
} rsq_fragment;   /* with insertion / deletion variants the reference span of a fragment is not [start, start + len): the read ids carry the real end */

/* Which blocks a worker simulates.  Simulator::SimulationThread (reseq/Simulator.cpp:2384-2401) takes blocks one by one from a shared counter; here a worker -- a
 * host thread with its own simulator and device inside `reseq illuminaPE --gpus N`, or a process of the launcher -- owns a contiguous range, so that the workers'
 * texts in worker order are the single run's.  rsq_sim_block_weights: expected pairs per block up to a constant (the sequence's reference bias), weights[*n_blocks]
 * in block order (NULL: only the count), after rsq_sim_prepare.  rsq_partition_blocks: bounds[workers + 1], worker r gets blocks [bounds[r], bounds[r + 1]) of
 * 1 .. total_blocks, balanced by the weights; a worker may get an empty range.  (rsq_sim_block_weights also after rsq_sim_prepare_plan: the sharded pre-pass needs
 * the ranges before it runs.) */
int rsq_sim_block_weights(const rsq_sim *s, double *weights, size_t cap, uint32_t *n_blocks);
int rsq_partition_blocks(uint32_t total_blocks, uint32_t workers, const double *weights, uint32_t *bounds);

/* A rank's share of a job, generated once and kept (Simulator::Simulate's worker loop + Output/Flush, Simulator.cpp:2384-2401, 150-230, for one of N processes).
 * rsq_sim_job_generate simulates the blocks [block_lo, block_hi) in calls of `batch_blocks` blocks (0: about 12 M pairs per call) and keeps the FASTQ text of the
 * whole range in device memory -- 288 GB of HBM hold a rank's share of a 30x human-sized job (236 GB of text over 8 ranks) with room to spare -- so that the
 * ranks can exchange their sizes BEFORE anything is written and every rank then writes straight to its own offset of the two final files: no shard files, no
 * second copy, no second simulation.  rsq_sim_job_write copies the kept text through page-locked double buffers and pwrite()s it at the given offsets with
 * `threads_per_file` threads per file (each its own part of the range, stream and buffers; 0: 1 -- buffered writes into one file serialise on its inode lock); the files
 * are created if need be, never truncated.  r2_path NULL: a job with one file (the records' text kept by rsq_sim_error_model_file with keep_text).
 * rsq_sim_job_free releases the text (rsq_sim_free does too).  A single-process run has offset 0 and may as well stream (the `reseq` command line does). */
int rsq_sim_job_generate(rsq_sim *s, uint32_t block_lo, uint32_t block_hi, uint32_t batch_blocks, uint64_t *n_pairs, uint64_t *r1_bytes, uint64_t *r2_bytes, void *stream);
int rsq_sim_job_write(rsq_sim *s, const char *r1_path, uint64_t r1_offset, const char *r2_path, uint64_t r2_offset, uint32_t threads_per_file);
/* Compressed output of a job over several processes: the kept text becomes gzip members of 1 MB of text each in host memory (a pool of threads sized by the
 * processors the process may use; zlib's default level), the device arrays are released, *r1_bytes / *r2_bytes = the COMPRESSED sizes.  A file of concatenated
 * members is a gzip file (RFC 1952; SeqAn, zlib's gzread and gzip -d read it as one stream), so the ranks exchange these sizes and rsq_sim_job_write puts each
 * rank's members at its offset exactly as it does plain text.  The decompressed file is the single run's; where the members end depends on the ranks' shares. */
int rsq_sim_job_compress(rsq_sim *s, uint64_t *r1_bytes, uint64_t *r2_bytes);

/* gzip on the device (reseq_amd/csrc/rsq_deflate.h; the reference compresses in SeqAn's stream behind Simulator::Flush, reseq/Simulator.cpp:150-182, for the output
 * names of main.cpp:404,412): text_dev[0, text_len) -- device memory -- as gzip members (RFC 1952) one behind the other in out_dev, every member the deflate
 * (RFC 1951, one block with a dynamic Huffman code taken from a sample of the call's text) of at most 65280 bytes of text, framed like a BGZF block (extra field
 * "BC"); concatenated members are a gzip file for zlib's gzread, gzip -d, SeqAn and bgzip alike.  *out_len = their bytes; RSQ_ENOSPC (and the size needed) if
 * out_cap is smaller -- rsq_gzip_bound(text_len) always suffices.  Kernel time: "gzip".  rsq_sim_job_compress uses it unless option host_gzip is 1. */
size_t rsq_gzip_bound(size_t text_len);
/* the 28 bytes that end a BGZF file (an empty member; bgzip / htslib warn when it is missing, gzip / zlib / SeqAn read it as no text): returns 28 and, with room,
 * writes them to out.  The members of rsq_sim_gzip_device are framed as BGZF blocks; whoever finishes a file of them appends this one (the command line, the
 * launcher and rsq_sim_error_model_file on a whole file do). */
size_t rsq_gzip_eof_member(char *out, size_t cap);
/* keep = 1: the Huffman code of the NEXT rsq_sim_gzip_device call serves the calls after it as well (text of one kind, call after call: a file written in batches --
 * no sample, no code and no wait for them per call); keep = 0 (the default): every call its own code.  Either way every member is a complete gzip member. */
int rsq_sim_gzip_keep_code(rsq_sim *s, int keep);
int rsq_sim_gzip_device(rsq_sim *s, const char *text_dev, size_t text_len, char *out_dev, size_t out_cap, size_t *out_len, void *stream);
/* `bytes` of the kept text of file `file` (0 / 1) from byte `at` on, copied into the caller's device memory: a rank's contribution to one round of a gather of
 * the output (simulate.py --gatherOutput: fixed-size slices gathered on the first rank over RCCL, which writes them with rsq_dev_pwrite).  RSQ_ESTATE without text. */
int rsq_sim_job_read(rsq_sim *s, int file, uint64_t at, size_t bytes, char *dst_dev, void *stream);
int rsq_sim_job_free(rsq_sim *s);

/* The hot path: Simulator::SimulationThread over blocks [block_lo, block_hi) (reseq/Simulator.cpp:2384-2401):
 * coverage sieve, CreateReads, FASTQ text of both mates.  r1_dev / r2_dev receive the two FASTQ streams in
 * identical record order ((block, start, length, strand choice, duplicate) order).  On RSQ_ENOSPC the required
 * byte counts are returned in *r1_len / *r2_len and nothing is written.  frags_dev (optional, capacity
 * frags_cap records) receives the fragment list. */
int rsq_sim_pairs(rsq_sim *s, uint32_t block_lo, uint32_t block_hi, char *r1_dev, size_t r1_cap, size_t *r1_len, char *r2_dev, size_t r2_cap, size_t *r2_len,
                  uint64_t *n_pairs, rsq_fragment *frags_dev, size_t frags_cap, void *stream);
/* Simulator::SimulateAdapterOnlyPairs (reseq/Simulator.cpp:2359-2382): pairs [first, first+n) of the adapter-only share */
int rsq_sim_adapter_only_pairs(rsq_sim *s, uint64_t first, uint64_t n, char *r1_dev, size_t r1_cap, size_t *r1_len, char *r2_dev, size_t r2_cap, size_t *r2_len,
                               void *stream);

/* Simulator::ApplyErrorsAndQualityToFastaInput with the FASTA header already parsed (reseq/Simulator.cpp:2403-2512):
 * n records of `read_len` template bases each.  Inputs (device): seqs[n][read_len] base codes 0..3, seg[n] template
 * segment 0/1, frag_len[n], dom[n][read_len] dominant-error base codes 0..4, rate[n][read_len] error percent.
 * Outputs (device): seq_out/qual_out [n][out_stride] (base codes / phred+offset characters), read_len_out[n],
 * num_errors_out[n], tile_out[n], cigar_out[n][cigar_stride] NUL-terminated.  first_index is the index of the
 * first record in the input file (it selects the records' random streams). */
int rsq_sim_error_model(rsq_sim *s, uint64_t first_index, uint64_t n, uint32_t read_len, const uint8_t *seqs_dev, const uint8_t *seg_dev,
                        const uint32_t *frag_len_dev, const uint8_t *dom_dev, const uint8_t *rate_dev, uint8_t *seq_out_dev, uint8_t *qual_out_dev,
                        uint32_t out_stride, uint16_t *read_len_out_dev, uint16_t *num_errors_out_dev, uint16_t *tile_out_dev, char *cigar_out_dev,
                        uint32_t cigar_stride, void *stream);

/* The same with the FASTQ text written on the device (reseq/Simulator.cpp:2497-2504 and the ordered output of :184-213): record i is
 * "@{id_i} {CIGAR} E{errors}\n{bases}\n+\n{qualities}\n" with id_i = ids[id_off[i], id_off[i+1]) (device pointers, id_off has
 * n + 1 entries), records in input order in text_dev.  *text_len = bytes needed; RSQ_ENOSPC if text_cap is smaller (nothing written). */
int rsq_sim_error_model_fastq(rsq_sim *s, uint64_t first_index, uint64_t n, uint32_t read_len, const uint8_t *seqs_dev, const uint8_t *seg_dev,
                              const uint32_t *frag_len_dev, const uint8_t *dom_dev, const uint8_t *rate_dev, const char *ids_dev, const uint64_t *id_off_dev,
                              char *text_dev, size_t text_cap, size_t *text_len, void *stream);

/* The same from the FASTA text itself, parsed on the device (reseq/Simulator.cpp:2423-2485 are the header's checks, :2900-3014 the reader this replaces):
 * text_dev[0, text_len) is a stretch of the input file as it stands there, beginning at a record's '>' (line ends may precede it); records are
 * ">{id} {1|2};{fragment length};{dominant errors};{error rates}" and the template on one line or wrapped over several, line ends \n or \r\n.
 * final = 0: more text follows, so the block's last record may be cut off -- it is left out, *consumed = its offset, and the caller hands the bytes from there on
 * again in front of the text that follows (no record starts in the block: *consumed = 0, *n_records = 0 -- hand in more).  final = 1: the block ends the input,
 * *consumed = text_len.  The records' FASTQ text goes to out_dev in input order exactly as rsq_sim_error_model_fastq writes it (*out_len bytes; RSQ_ENOSPC and
 * nothing written if out_cap is smaller), *n_records = their number; first_index = the index in the input of the block's first record.
 * A malformed record -- the first in input order -- ends the call with RSQ_EIO and the reference's message about it in rsq_last_error().  So does, in this call and
 * in the two above, a fragment length that the profile's insert lengths / read lengths by fragment length do not hold (profiles with more than one read length look
 * it up there, Simulator.h:185-198; the reference's Vect::at prints "Called index ... range is from ... to ..." and throws): checked on the device before anything is simulated.
 * Kernel time: "parse_records" beside the names below.  text_len < 4 GB. */
int rsq_sim_error_model_fasta(rsq_sim *s, uint64_t first_index, const char *text_dev, size_t text_len, int final, char *out_dev, size_t out_cap, size_t *out_len,
                              uint64_t *n_records, size_t *consumed, void *stream);

/* Simulator::SimulateErrorModelOnly (reseq/Simulator.h:457, Simulator.cpp:2900-3014): seqToIllumina from file to file.  input_path NULL = stdin, output_path NULL =
 * stdout; gzip / bzip2 input by content, output by name (.gz, .bz2) as SeqAn does.  A pipeline around rsq_sim_error_model_fasta: a plain file is read at offsets by
 * several threads that upload their blocks themselves, the calling thread runs the device calls on the blocks that are there, two more threads download and write the
 * text; a compressed file or a pipe has one reader.  *n_records records written as *out_bytes bytes -- of FASTQ text, or of gzip members when the output's name ends in
 * .gz and the device compresses (the bytes that went into the file).  An input without any record gives *n_records = 0 (the
 * reference calls that an error: the caller's to report).  A malformed record: RSQ_EIO and the reference's words in rsq_last_error(); what has been written of the
 * output by then is the caller's to remove (Simulator.cpp:2888-2892 does).
 * options (NULL or zero fields: the defaults): read_threads (6), block_kb (48 MB blocks), batch_blocks (up to 8 blocks in one device call); from / to: bytes
 * [from, to) of a plain input file, `from` a record's first byte (to = 0: the file's end), first_record: the index in the whole input of the range's first record --
 * a rank's share of a job over several GPUs (rsq_fasta_count_records finds the ranges); progress: called with the records done so far, about every million;
 * trace / trace_cap: receives one line saying where each side of the pipeline spent its time; keep_text = 1 (with output_path NULL): nothing is written, the text
 * stays in device memory like the pairs' text of rsq_sim_job_generate -- rsq_sim_job_write(sim, path, offset, NULL, 0, threads) puts it at a rank's offset of the
 * one output file once the ranks know each other's *out_bytes, rsq_sim_job_read serves a gather, rsq_sim_job_free releases it. */
typedef struct {
    uint32_t read_threads, block_kb, batch_blocks;
    uint32_t keep_text;
    uint64_t from, to, first_record;
    void (*progress)(uint64_t records, void *user);
    void *user;
    char *trace;
    size_t trace_cap;
} rsq_error_model_file_options;
int rsq_sim_error_model_file(rsq_sim *s, const char *input_path, const char *output_path, const rsq_error_model_file_options *options, uint64_t *n_records,
                             uint64_t *out_bytes);
/* Record starts ('>' at the start of a line or of the file) in bytes [from, to) of a plain file (to = 0: its end): their number, and the first one's offset (`to` if
 * there is none).  Host code, no device.  Ranks of a sharded seqToIllumina run count their stretch of the file, exchange the two numbers, and know the index of
 * their first record and where the next rank's share begins (reseq_amd/simulate.py; SURVEY section 8(e): "seqToIllumina shards by input record ranges"). */
int rsq_fasta_count_records(const char *path, uint64_t from, uint64_t to, uint32_t threads, uint64_t *n_starts, uint64_t *first_start);

/* kernel timing of the last rsq_sim_pairs / rsq_sim_error_model call: HIP events recorded around each kernel on the stream it was launched on.  A call
 * over a large block range runs as several sub-ranges (blocks are independent, Simulator.cpp:2384-2401) whose sieve / reads / text stages are pipelined on
 * three streams: the time is the SUM over the call's launches of that kernel, rsq_sim_last_kernel_launches says how many there were.
 * names: "sieve" (screen + finish), "sieve_screen", "sieve_emit", "fill_reads", "format_write", "scan"; with variants of any kind also
 * "slot_table" and "variant_templates"; for a profile with tiles "bin_tiles". */
int rsq_sim_last_kernel_ms(const rsq_sim *s, const char *kernel, double *ms);
int rsq_sim_last_kernel_launches(const rsq_sim *s, const char *kernel, uint32_t *launches);

/* ---- device memory helpers so that callers without a HIP binding (ctypes tests, the CLI) can stage buffers */
int rsq_dev_alloc(int device, size_t bytes, void **out_dev);
int rsq_dev_free(int device, void *dev);
int rsq_dev_upload(int device, void *dst_dev, const void *src, size_t bytes);
int rsq_dev_download(int device, void *dst, const void *src_dev, size_t bytes);
/* `bytes` of device memory to byte `offset` of the file `path` (created if missing), through page-locked buffers with the copy of one slice under the write of
 * the one before: what the ONE writer of a gathered output does with the slices it received (reseq_amd/simulate.py --gatherOutput) */
int rsq_dev_pwrite(int device, const void *src_dev, size_t bytes, const char *path, uint64_t offset);
/* page-locked host memory: rsq_dev_download into it runs at the full PCIe rate (the CLI's output buffers) */
int rsq_host_alloc(size_t bytes, void **out_host);
int rsq_host_free(void *host);
/* Streams of the caller's own, for a pipeline whose sides run in threads (the seqToIllumina command: readers upload blocks of text, one thread runs
 * rsq_sim_error_model_fasta, one downloads the FASTQ text; the reference's shape is one reader, worker threads and ordered output, Simulator.cpp:2900-3014):
 * the plain copies above use the null stream, which waits for all others.  rsq_dev_copy_on returns when its copy is done; kind 0 = host to device,
 * 1 = device to host, 2 = device to device; host memory from rsq_host_alloc for the full rate. */
int rsq_stream_create(int device, void **out_stream);
int rsq_stream_destroy(int device, void *stream);
int rsq_dev_copy_on(int device, void *dst, const void *src, size_t bytes, int kind, void *stream);

#ifdef __cplusplus
}
#endif
#endif
