// rsq_host.h -- host-side objects: RSQP container, profile (DataStats + ProbabilityEstimates subset the
// simulation consumes, SURVEY.md section 8(a) row "in"), reference sequences.  No device code here.
#pragma once
#include <stdint.h>

#include <map>
#include <stdexcept>
#include <string>
#include <string.h>
#include <vector>

namespace rsq {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ------------------------------------------------------------------------------------------------ options
// Testing and measurement switches of the library (C ABI rsq_set_option / rsq_get_option): explicit calls of the embedding program, never
// read from the environment.  Results (reads, FASTQ bytes) do not depend on any of them; they choose between equivalent routes.  A simulator
// takes a copy of all values when it is created (rsq_sim_create; SimState::opt) and reads only the copy; the loaders read the process-wide values.
struct Options {
    int64_t fill_mode = -1;            // -1: screened draws on the LDS image when the plan has one; 0: every draw of the read kernels in double precision from HBM
    int64_t image_tiles = 0;           // 0: the image holds all tiles when they fit, else one tile per workgroup; 1: one tile per workgroup even when all fit
    int64_t rate_rows = 0;             // > 0: at most so many error-rate rows in the LDS image
    int64_t no_indel_skip = 0;         // 1: no draw decided by the random word alone (the reads' indel draw, the chains' error-rate draw)
    int64_t force_exact = 0;           // 1: the screen decides nothing, every draw takes the double-precision route behind it
    int64_t min_quality_quads = 0;     // a wider instantiation of the read kernels than the profile's quality values need
    int64_t trace_plan = 0;            // 1: the LDS plan on stderr
    int64_t trace_prepare = 0;         // 1: stage times of the pre-pass on stderr
    int64_t bias_window = 0;           // > 0: start positions per pass of the bias sums
    int64_t window_chunks = 0;         // > 0: window length (chunks) of the host pass over the variants' systematic errors
    int64_t trace_load = 0;            // 1: stage times of the FASTA reader and of replace_n on stderr
    int64_t serial_fasta = 0;          // 1: the line reader for every FASTA file
    int64_t fasta_stretch = 0;         // > 0: stretch length of the memory-mapped FASTA reader
    int64_t chain_chunk = 0;           // > 0: positions per chunk of the systematic-error chains (default: 256 to 4096 by the size of the reference)
    int64_t chain_warmup = -1;         // >= 0: positions of the run-up in front of a chunk in the chains' first pass (default: half a chunk, at most 384)
    int64_t serial_parse = 0;          // 1: the line readers for every methylation and variant file
    int64_t mapped_parses = 0;         // a count, not a switch: files the memory-mapped methylation / variant readers have read (the others went to the line readers)
    int64_t parse_stretch = 0;         // > 0: piece length of the memory-mapped methylation / variant readers (and no minimum file size)
    int64_t job_chunk_bytes = 0;       // > 0: size of the device arrays rsq_sim_job_generate keeps a rank's FASTQ text in (default 2 GiB; tests: small, so that the text spans several)
    int64_t job_write_direct = 0;      // 1: rsq_sim_job_write passes the whole 4 KB blocks of a rank's byte range around the page cache (O_DIRECT) where the file system allows it
    int64_t specialize = 1;            // 1: the read kernels are compiled for the loaded profile at run time (hiprtc, rsq_spec.h) where that is possible; 0: always the library's own instantiations
    int64_t fasta_no_stage = 0;        // 1: k_fasta_records parses every record where it lies in HBM (the instantiation that otherwise only serves records too long for the LDS staging)
    int64_t fill_waves = 0;            // > 0: waves per workgroup of the read kernels (default: by the size of the call, fill_shape in rsq_sim.hip; measurements)
    int64_t host_gzip = 0;             // 1: .gz output is compressed by zlib on host threads (the route before round 5; the checker of the device's gzip); 0: on the device (rsq_deflate.h)
    int64_t hiprtc_by_name = 0;        // 1: libhiprtc is searched by name (whatever copy the process finds first, e.g. PyTorch's) instead of the system ROCm's by path; read once, at first use
    int64_t overlap = 0;               // n > 1: rsq_sim_pairs cuts its block range into n sub-ranges whose sieve / reads / text stages are pipelined on three streams
};
Options &options();                                               // the process-wide values
bool set_option(const char *name, int64_t value);                 // false: no such option
bool get_option(const char *name, int64_t *value);
const char *option_names();                                       // space-separated, for error messages

// ---------------------------------------------------------------------------------------------- container
struct Array {
    int dtype = 0;   // 0 u8, 1 u16, 2 u32, 3 u64, 4 i32, 5 i64, 6 f64
    std::vector<uint64_t> dims;
    uint64_t count = 0;
    const uint8_t *data = nullptr;
};

class Container {
   public:
    explicit Container(const std::string &path);
    const Array &get(const std::string &name) const;
    bool has(const std::string &name) const { return arrays_.count(name) != 0; }
    template <class T>
    std::vector<T> vec(const std::string &name, int dtype) const {
        const Array &a = get(name);
        if (a.dtype != dtype) throw Error("profile array '" + name + "' has the wrong type");
        const T *p = reinterpret_cast<const T *>(a.data);
        return std::vector<T>(p, p + a.count);
    }
    template <class T>
    T scalar(const std::string &name, int dtype) const {
        return vec<T>(name, dtype).at(0);
    }

   private:
    std::vector<uint8_t> buf_;
    std::map<std::string, Array> arrays_;
};

// ------------------------------------------------------------------------------------------------ profile
template <class T>
struct Vect {                       // reseq::Vect<T> (Vect.hpp): values with an offset; operator[] is 0 outside
    uint64_t from = 0;
    std::vector<T> v;
    uint64_t to() const { return from + v.size(); }
    T operator[](uint64_t i) const { return (i < from || i >= to()) ? T(0) : v[i - from]; }
};

struct HostTable {                  // LogArrayResult<N> (ProbabilityEstimates.h:351-357)
    uint32_t nm = 0;
    std::vector<uint32_t> par0;
    uint32_t from[4] = {0, 0, 0, 0}, to[4] = {0, 0, 0, 0};
    std::vector<double> dim2[4];
    void modify_par0(uint32_t par0_index, double multiplier);   // ProbabilityEstimates.h:532-545
    void set_par0(uint32_t par0_index);                          // ProbabilityEstimates.h:547-556
};

struct HostAdapters {               // AdapterStats getters (AdapterStats.h:101-107)
    std::vector<uint8_t> seqs;
    std::vector<uint32_t> seq_ptr;
    std::vector<uint64_t> counts, significant;
    std::vector<uint32_t> cut_ptr, cut_from;
    std::vector<uint64_t> cut;
    uint32_t n() const { return (uint32_t)seq_ptr.size() - 1; }
};

struct HostRlByFl {                 // ReadLengthsByFragmentLength (DataStats.h:198) as CSR over fragment lengths
    uint64_t from = 0;
    std::vector<uint32_t> row_ptr, row_from;
    std::vector<uint64_t> values, non_mapped;
};

struct Profile {
    uint8_t phred_offset = 33;
    double corrected_coverage = 0;
    uint16_t max_len_deletion = 0;
    uint32_t reset_distance = 0;
    Vect<uint64_t> read_lengths[2];
    HostRlByFl rl_by_fl[2];
    uint64_t total_number_reads = 0;
    std::vector<uint16_t> tiles;
    std::vector<uint64_t> tile_abundance;
    HostAdapters adapters[2];
    Vect<uint64_t> polya;
    uint64_t overrun_bases[5] = {0, 0, 0, 0, 0};
    Vect<uint64_t> insert_lengths;
    Vect<double> insert_lengths_bias, gc_bias;
    std::vector<double> sur_bias;       // [3][1<<20]
    double dispersion[2] = {0, 0};
    std::vector<double> ref_seq_bias;
    std::vector<HostTable> quality, seq_quality, base_call, dom_error, error_rate, indels;
    uint32_t n_tiles() const { return (uint32_t)tiles.size(); }

    static Profile load(const std::string &path);                 // RSQP container
    // ReSeq's own files (rsq_profile_archive.cpp): DataStats::Load + PrepareProcessing, ProbabilityEstimates::Load + PrepareResult
    static bool is_archive(const std::string &path);
    static Profile load_archives(const std::string &stats_path, const std::string &ipf_path, double precision_aim = 0.05, std::string *warnings = nullptr);
    static std::string archive_layout(const std::string &stats_path, const std::string &ipf_path);      // the class-info sites of both files, the parse error if any
    void save(const std::string &path) const;                     // as an RSQP container
    // as ReSeq's own pair of files (rsq_profile_archive.cpp): statistics and fits whose prepared form is this profile; ipf_path empty: "<stats_path>.ipf"
    void save_archives(const std::string &stats_path, const std::string &ipf_path, uint64_t creation_time) const;
    void change_error_rate(double multiplier);          // ProbabilityEstimates.h:1516-1527
    void remove_substitution_errors();                  // :1529-1540
    void remove_indel_errors();                         // :1542-1549
};

// cumulative probabilities of a std::discrete_distribution over `w` (libstdc++ param_type::_M_initialize)
std::vector<double> discrete_cp(const uint64_t *w, size_t n);

// ---------------------------------------------------------------------------------------------- reference
struct Reference {
    std::vector<std::string> names;          // full id lines (Reference::ReferenceId)
    std::vector<std::vector<uint8_t>> codes; // A=0,C=1,G=2,T=3,N=4 (Dna5)
    static Reference read_fasta(const std::string &path);   // Reference.cpp:758 ReadFasta
    void replace_n(uint64_t seed);                          // Reference.cpp:813 ReplaceN
    bool has_n() const;
    uint64_t total_size() const;
    std::string first_part(size_t i) const;                 // Reference.cpp:476-480 ReferenceIdFirstPart
};

// ------------------------------------------------------------------- systematic-error profile (FASTQ) and ref-bias file
// Simulator::WriteOutSystematicErrorProfile (Simulator.cpp:2562-2588) / ReadSystematicErrors (Simulator.h:326-335):
// one FASTQ record per strand, seq = dominant error (ACGTN), qual = error percent + 33 where percents above 86 are halved
// into the 94 printable values (odd values join the even one before them, so the round trip is lossy there).
uint8_t compress_sys_error_rate(uint8_t percent);
uint8_t expand_sys_error_rate(uint8_t stored);
std::string sys_error_fastq_record(const std::string &id, const uint8_t *dom, const uint8_t *rate, size_t n);
struct SysErrorRecord {
    std::string id;
    std::vector<uint8_t> dom, rate;      // base codes 0..4, percents (expanded)
};
std::vector<SysErrorRecord> parse_sys_error_fastq(const std::string &text);
std::string read_text_file(const std::string &path);
void write_text_file(const std::string &path, const std::string &text);

// FragmentDistributionStats::UpdateRefSeqBias kFile (FragmentDistributionStats.cpp:3386-3495): lines "identifier bias";
// `first_names` are the reference ids up to the first space.  Throws with the reference's messages joined.
std::vector<double> read_ref_bias_file(const std::string &path, const std::vector<std::string> &first_names);

// Reference::PrepareMethylationFile + ReadMethylation (Reference.cpp:1132-1310): extended BED "sequence start end methylation
// [methylation of allele 1 ...]"; per sequence the unmethylated regions [first, second) and, per allele column, the C->T conversion
// probability 1 - methylation (one column serves every allele).  File order must follow the reference; throws with the reference's
// messages.  The simulation without variants has one allele.
struct Methylation {
    std::vector<std::vector<uint32_t>> first, second;
    std::vector<std::vector<std::vector<double>>> rate;      // [sequence][allele column][region]
};
Methylation read_methylation_file(const std::string &path, const std::vector<std::string> &first_names, const std::vector<uint32_t> &seq_len, uint32_t num_alleles = 1);

// ---------------------------------------------------------------------------------------------- variants (VCF)
// Reference::Variant (Reference.h:24-62), InsertVariant (:115-139), PrepareVariantFile / ReadFirstVariants / ReadVariants
// (Reference.cpp:126-420, 1003-1077): every VCF record is split into single-reference-position variants (substitution, deletion "",
// insertion = the base at the position followed by the inserted bases), kept per sequence sorted by position and, at one position,
// by length; one bit per allele (up to 128) says which alleles carry it.  Loading only: the simulation with variants (SURVEY.md
// section 8 row a17) is not built yet.
// var_seq_ of a variant: base codes 0..3, none for a deletion, one for a substitution, a few for an insertion -- kept inside the variant up to 16 of them (a
// call set's millions of substitutions then cost no allocation each), on the heap beyond
class BaseSeq {
    static constexpr uint32_t kInPlace = 16;
    union {
        uint8_t here_[kInPlace];
        uint8_t *far_;
    };
    uint32_t n_ = 0;
    void take(const uint8_t *from, size_t n) {
        n_ = (uint32_t)n;
        uint8_t *to = here_;
        if (n > kInPlace) to = far_ = new uint8_t[n];
        if (n) memcpy(to, from, n);
    }

  public:
    BaseSeq() {}
    BaseSeq(const std::vector<uint8_t> &v) { take(v.data(), v.size()); }
    BaseSeq(const BaseSeq &o) { take(o.data(), o.n_); }
    BaseSeq(BaseSeq &&o) noexcept {
        memcpy(static_cast<void *>(this), &o, sizeof *this);
        o.n_ = 0;
    }
    BaseSeq &operator=(BaseSeq o) noexcept {
        if (n_ > kInPlace) delete[] far_;
        memcpy(static_cast<void *>(this), &o, sizeof *this);
        o.n_ = 0;
        return *this;
    }
    ~BaseSeq() {
        if (n_ > kInPlace) delete[] far_;
    }
    const uint8_t *data() const { return n_ > kInPlace ? far_ : here_; }
    size_t size() const { return n_; }
    uint8_t operator[](size_t i) const { return data()[i]; }
    const uint8_t *begin() const { return data(); }
    const uint8_t *end() const { return data() + n_; }
    bool operator==(const std::vector<uint8_t> &v) const { return v.size() == n_ && (!n_ || 0 == memcmp(v.data(), data(), n_)); }
    bool operator==(const BaseSeq &o) const { return o.n_ == n_ && (!n_ || 0 == memcmp(o.data(), data(), n_)); }
};
struct Variant {
    static constexpr uint32_t kMaxAlleles = 128;
    uint32_t position = 0;
    BaseSeq var_seq;
    uint64_t allele[2] = {0, 0};
    bool in_allele(uint32_t a) const { return (allele[a / 64] >> (a % 64)) & 1; }
    uint32_t first_allele() const;         // Reference.h:42-58
};
struct Variants {
    uint32_t num_alleles = 1;
    std::vector<std::vector<Variant>> by_seq;
};
void insert_variant(std::vector<Variant> &variants, uint32_t position, const std::vector<uint8_t> &var_seq, const uint64_t (&allele)[2]);
Variants read_variants(const std::string &path, const std::vector<std::string> &first_names, const std::vector<std::vector<uint8_t>> &codes);

}  // namespace rsq
