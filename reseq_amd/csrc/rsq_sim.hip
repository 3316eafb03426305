// rsq_sim.hip -- the simulator object behind the C ABI (include/reseq_amd.h): packs profile + reference into
// HBM, runs the pre-passes and drives the kernels of rsq_kernels.h.  Compiled for gfx950 only.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <math.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <thread>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/reseq_amd.h"
#include "rsq_deflate.h"
#include "rsq_fasta.h"
#include "rsq_pack.h"
#include "rsq_spec.h"
#include "rsq_textio.h"

namespace rsq {

static thread_local std::string g_last_error, g_last_warning;

struct HipError : Error {
    using Error::Error;
};
#define HIP_CHECK(expr)                                                                                       \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) throw HipError(std::string(#expr) + ": " + hipGetErrorString(e_));              \
    } while (0)

// ------------------------------------------------------------------------------------------ device memory
class DevBuf {
   public:
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p_(o.p_), bytes_(o.bytes_) { o.p_ = nullptr; o.bytes_ = 0; }
    ~DevBuf() { if (p_) (void)hipFree(p_); }
    void reserve(size_t bytes) {                   // grow-only
        if (bytes <= bytes_) return;
        if (p_) {
            HIP_CHECK(hipDeviceSynchronize());     // a pipelined call may still read the old array on another stream
            HIP_CHECK(hipFree(p_));
        }
        p_ = nullptr;
        bytes_ = 0;
        HIP_CHECK(hipMalloc(&p_, bytes ? bytes : 8));
        bytes_ = bytes;
    }
    template <class T>
    void upload(const std::vector<T> &v) {
        reserve(v.size() * sizeof(T) + 8);
        if (!v.empty()) HIP_CHECK(hipMemcpy(p_, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    void release() {
        if (p_) {
            HIP_CHECK(hipDeviceSynchronize());
            HIP_CHECK(hipFree(p_));
        }
        p_ = nullptr;
        bytes_ = 0;
    }
    template <class T>
    T *as() const { return reinterpret_cast<T *>(p_); }
    size_t bytes() const { return bytes_; }

   private:
    void *p_ = nullptr;
    size_t bytes_ = 0;
};

}  // namespace rsq
#include "rsq_s2i.h"
namespace rsq {

struct Timer {                                     // HIP events on the stream the kernels are launched on; a call may time several launches (sub-ranges)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t used = 0;
    Timer() = default;
    Timer(const Timer &) = delete;
    Timer &operator=(const Timer &) = delete;
    ~Timer() {
        for (auto &e : events) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    }
    void reset() { used = 0; }
    void start(hipStream_t st) {
        if (used == events.size()) {
            hipEvent_t a, b;
            HIP_CHECK(hipEventCreate(&a));
            HIP_CHECK(hipEventCreate(&b));
            events.emplace_back(a, b);
        }
        HIP_CHECK(hipEventRecord(events[used].first, st));
    }
    void stop(hipStream_t st) {
        HIP_CHECK(hipEventRecord(events[used].second, st));
        ++used;
    }
    double ms() const {                            // sum over the launches since reset()
        double sum = 0.0;
        for (size_t i = 0; i < used; ++i) {
            float t = 0;
            if (hipEventElapsedTime(&t, events[i].first, events[i].second) == hipSuccess) sum += t;
        }
        return sum;
    }
    size_t launches() const { return used; }
};

}  // namespace rsq

using namespace rsq;

struct rsq_profile {
    Profile p;
};
struct rsq_ref {
    Reference r;
    Variants variants;
    bool has_variants = false;
};

// arrays packed by rsq_pack.h go to HBM; they live as long as the simulator
struct DeviceUploader : Uploader {
    std::vector<std::unique_ptr<DevBuf>> owned[kUploadScopes];
    int device = 0;
    void bind_thread() override { HIP_CHECK(hipSetDevice(device)); }
    void release_scope(int scope) override {
        HIP_CHECK(hipDeviceSynchronize());                         // nothing may still read what goes away
        owned[scope].clear();
    }
    void *put_bytes(const void *data, size_t bytes) override {
        std::vector<std::unique_ptr<DevBuf>> &own = owned[current_scope];
        own.emplace_back(new DevBuf());
        own.back()->reserve(bytes + 8);
        HIP_CHECK(hipMemcpy(own.back()->as<void>(), data, bytes, hipMemcpyHostToDevice));
        return own.back()->as<void>();
    }
    void *put_zeros(size_t bytes) override {
        std::vector<std::unique_ptr<DevBuf>> &own = owned[current_scope];
        own.emplace_back(new DevBuf());
        own.back()->reserve(bytes + 8);
        HIP_CHECK(hipMemset(own.back()->as<void>(), 0, bytes + 8));
        return own.back()->as<void>();
    }
    void write_bytes(void *dst, const void *src, size_t bytes) override { HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); }
    void read_bytes(void *dst_host, const void *src, size_t bytes) override { HIP_CHECK(hipMemcpy(dst_host, src, bytes, hipMemcpyDeviceToHost)); }
};

struct MappedFile {                                                   // a whole file, read-only
    const char *data = nullptr;
    size_t size = 0;
    explicit MappedFile(const char *path) {
        const int fd = open(path, O_RDONLY);
        if (fd < 0) throw Error(std::string("cannot open ") + path + ": " + strerror(errno));
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size <= 0) {
            close(fd);
            throw Error(std::string(path) + " is empty or cannot be examined");
        }
        size = (size_t)st.st_size;
        void *p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (p == MAP_FAILED) throw Error(std::string("cannot map ") + path);
        data = static_cast<const char *>(p);
    }
    ~MappedFile() {
        if (data) munmap(const_cast<char *>(data), size);
    }
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
};

// ------------------------------------------------------------------------------------------------ rsq_sim
struct rsq_sim : SimState {
    int device = 0;
    DeviceUploader up;
    // workspace of the hot path (grow-only): two sets, so that the sieve of one sub-range of a call can run beside the FASTQ text of the one before it
    struct Workspace {
        DevBuf fvars;                  // FragmentVar per fragment (variants of any kind)
        DevBuf slot_table;             // SlotInfo per slot of the batch (variants of any kind)
        DevBuf counts, offsets, tile_sums, scan_total, frags, raw_seq, raw_qual, raw_ops, raw_meta, sizes, off_r1, off_r2, fill_counters, hits, hit_count, cands, pairs_of, pair_off, templates, rec_flags, rec_index, rec_count;
        DevBuf bin_keys, bin_small, bin_perm, bin_frags, bin_fvars;      // reads binned by tile: key per item; histogram, bins, counters (one small buffer); the sorted items
        DevBuf fa_counts, fa_first, fa_at, fa_len, fa_id_len, fa_frag_len, fa_seg, fa_codes, fa_summary;      // rsq_sim_error_model_fasta (rsq_fasta.h)
        DevBuf cell_info;              // the sieve without variants: per candidate the first strand's count and the two strands (k_sieve_finish<0> -> k_sieve_emit<0>)
        hipEvent_t text_done = nullptr;      // the text stage that last read this set's arrays
    } ws[2];
    // rsq_sim_job_generate: the FASTQ text of a rank's block range, kept until rsq_sim_job_write / rsq_sim_job_free: per file a list of device arrays, filled in order
    struct JobText {
        std::vector<std::unique_ptr<DevBuf>> chunks[2];
        std::vector<size_t> used[2];
        uint64_t bytes[2] = {0, 0};
        bool complete = false;         // rsq_sim_job_generate ran to its end: the text is whole (not: never generated, freed, or left half-made by a failed call)
        std::vector<unsigned char> packed[2];      // rsq_sim_job_compress: the text as gzip members in host memory (the device arrays are released, bytes[] = these sizes)
        bool is_packed = false;
        bool device_packed = false;                // rsq_sim_job_compress on the device: chunks[] hold gzip members instead of text (used[] and bytes[] their sizes)
        void clear() {
            complete = false;
            is_packed = false;
            device_packed = false;
            for (auto &p : packed) std::vector<unsigned char>().swap(p);
            for (int f = 0; f < 2; ++f) {
                chunks[f].clear();
                used[f].clear();
                bytes[f] = 0;
            }
        }
    } job;
    DevBuf totals;                 // [sub-ranges + 1][2] bytes of FASTQ text in front of a sub-range, per file
    bool gz_dense = false;                                 // the route of the code in gz_codes: every position searched and probed (gz::dense_pays), else FASTQ by its lines
    bool gz_keep_code = false, gz_have_code = false;      // rsq_sim_gzip_keep_code: the code of the first call serves the calls after it
    DevBuf gz_slots, gz_sizes, gz_at, gz_hist, gz_codes, gz_total;      // gzip on the device (rsq_deflate.h): the members' slots, their sizes and places, the sample's counts, the call's code
    Workspace *cur = &ws[0];       // the set the stage being enqueued works on
    hipStream_t side[2] = {nullptr, nullptr};      // the sieve's and the text's stream of a pipelined call
    hipEvent_t ev_call = nullptr, ev_fill = nullptr, ev_emit = nullptr;
    std::map<std::string, Timer> timers;
    uint32_t n_cu = 256;
    uint64_t *mailbox = nullptr;   // pinned host words the hot path's few device-to-host scalars land in
    uint64_t format_record_bytes = 480;      // the longest FASTQ record of the last rsq_sim_pairs call and a few bytes (text_stage sizes the formatter's LDS image with it)
    DevBuf longest_record;                   // where a call's text stages leave it
    uint64_t record_text_bytes = 480;        // the same for the seqToIllumina records' text (error_model_text)
    int force_fill_mode = -1;      // RSQ_FILL_MODE=0: every draw in double precision from HBM (tests run both paths)
    // read kernels compiled for this simulator's profile (rsq_spec.h); `specialize`: option specialize when the simulator was created
    SpecKernels spec;
    std::string arch;              // hipDeviceProp_t::gcnArchName
    bool specialize = true;
    // the sharded pre-pass (rsq_sim_prepare_plan ... rsq_sim_prepare_finish): what lives between its calls
    BiasPlan bias_plan;
    bool planned = false;
    struct ChainRun {
        std::vector<Chain> chains;
        ShardEdges edges;
        uint32_t n_chunks = 0, passes = 0, block_lo = 0, block_hi = 0;
        bool pass_through = false;     // the rank has no blocks: its neighbours' states go straight through
        DevBuf d_chains, d_chunk_chain, d_used, d_out[2], d_changed, d_list;
        bool valid = false;
    } chain_run;
};

static bool fill_is_binned(const rsq_sim &s) { return effective_fill_mask(s.dev.lds.mask, s.force_fill_mode) != 0 && s.dev.lds.binned; }
// The read kernel compiled for this simulator's profile (rsq_spec.h), or nullptr: the library's own instantiation runs.  Compiles at the first request of a variant;
// what happened is kept for rsq_sim_specialize / rsq_last_warning.
static thread_local std::string g_spec_note;
static hipFunction_t spec_kernel(rsq_sim &s, SpecKind kind, uint32_t mask, bool var, bool binned) {
    if (!s.specialize || mask == 0) return nullptr;                    // mask 0: no image, no plan to compile in
    std::string note;
    hipFunction_t fn = s.spec.get(s.dev, SpecVariant{kind, mask, var, binned}, s.arch, note);
    if (s.opt.trace_plan && note != g_spec_note) fprintf(stderr, "[rsq] %s\n", note.c_str());
    g_spec_note = note;
    return fn;
}

namespace rsq {

// --------------------------------------------------------------------------------- systematic errors (a13)
// passes of k_sys_chain over the run's chunks until no chunk's incoming state changed; `first_pass`: 0 for a new run, the run's pass
// count to resume one whose entering states (Chain::in_state) were replaced
static void iterate_sys_chains(rsq_sim &s, rsq_sim::ChainRun &run, hipStream_t st, uint32_t first_pass) {
    run.d_list.reserve((size_t)run.n_chunks * 4 + 16);
    uint32_t pass = first_pass;
    for (;; ++pass) {
        uint32_t *out_prev = run.d_out[(pass + 1) & 1].as<uint32_t>(), *out_new = run.d_out[pass & 1].as<uint32_t>();
        uint32_t n_run = run.n_chunks;
        const uint32_t *list = nullptr;
        if (pass > 0) {                                             // which chunks were entered with a state that has changed since
            HIP_CHECK(hipMemsetAsync(run.d_changed.as<uint32_t>(), 0, 4, st));
            hipLaunchKernelGGL(k_sys_chain_select, dim3(cdiv(run.n_chunks, 256)), dim3(256), 0, st, run.d_chains.as<Chain>(), run.d_chunk_chain.as<uint32_t>(), run.n_chunks,
                               run.d_used.as<uint32_t>(), out_prev, out_new, run.d_list.as<uint32_t>(), run.d_changed.as<uint32_t>(), (int)pass);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipMemcpyAsync(&n_run, run.d_changed.as<uint32_t>(), 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            list = run.d_list.as<uint32_t>();
        }
        if (n_run) {
            hipLaunchKernelGGL(k_sys_chain, dim3(cdiv(n_run, 64)), dim3(64), 0, st, s.dev, run.d_chains.as<Chain>(), run.d_chunk_chain.as<uint32_t>(), list, n_run, s.chain_chunk,
                               chain_warmup_len(s.chain_chunk, s.opt), run.d_used.as<uint32_t>(), out_prev, out_new, (int)pass);
            HIP_CHECK(hipGetLastError());
        }
        if (pass > 0 && !n_run) break;
        if (pass > first_pass + run.n_chunks + 2) throw Error("systematic-error chains did not converge");
    }
    HIP_CHECK(hipStreamSynchronize(st));
    run.passes = pass + 1;                                          // the final states are in d_out[(run.passes - 1) & 1]
}
// -V after a finished run: the variants' own systematic errors.  The chain state in front of every variant comes from the device (k_variant_chain_states: 8 bytes per
// variant come back instead of the tracks' 4 bytes per reference position), the pass over the variants' bases runs on the host's threads.
static void variant_sys_errors_from_run(rsq_sim &s, hipStream_t st) {
    rsq_sim::ChainRun &run = s.chain_run;
    const uint32_t n_variants = (uint32_t)s.variants.size();
    if (!s.has_variants || !n_variants) return;
    std::vector<ChainSpan> span((size_t)s.dev.n_seqs * 2, ChainSpan{-1, 0});
    for (size_t c = 0; c < run.chains.size(); ++c) {
        const Chain &ch = run.chains[c];
        if (ch.kind > 1u) continue;
        span[(size_t)ch.id * 2 + ch.kind] = ChainSpan{(int32_t)c, (c + 1 < run.chains.size() ? run.chains[c + 1].first_chunk : run.n_chunks) - ch.first_chunk};
    }
    DevBuf d_span, d_states;
    d_span.upload(span);
    d_states.reserve((size_t)n_variants * 8);
    hipLaunchKernelGGL(k_variant_chain_states, dim3(cdiv(n_variants, 256), 2), dim3(256), 0, st, s.dev, run.d_chains.as<Chain>(), d_span.as<ChainSpan>(), run.d_used.as<uint32_t>(),
                       s.chain_chunk, n_variants, d_states.as<uint32_t>());
    HIP_CHECK(hipGetLastError());
    std::vector<uint32_t> states((size_t)n_variants * 2);
    HIP_CHECK(hipMemcpyAsync(states.data(), d_states.as<uint32_t>(), states.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const std::vector<StrandTask> windows = strand_tasks(s.opt, run.chains, run.n_chunks, s.chain_chunk, [](uint32_t) { return 0u; });      // the windows' entering states are not needed
    build_variant_sys_errors(s, s.up, &windows, states.data(), states.data() + n_variants);
}
static uint32_t run_sys_chains(rsq_sim &s, hipStream_t st, ChainSet set, const ShardRange *range = nullptr) {
    rsq_sim::ChainRun &run = s.chain_run;
    run.valid = false;
    run.chains.clear();
    run.edges = ShardEdges{};
    std::vector<uint32_t> chunk_chain;
    s.chain_chunk = chain_chunk_len(s.total_ref_size, s.opt);
    build_chains(s, set, run.chains, chunk_chain, range, &run.edges);
    run.n_chunks = (uint32_t)chunk_chain.size();
    run.passes = 0;
    if (!run.n_chunks) return 0;
    run.d_chains.upload(run.chains);
    run.d_chunk_chain.upload(chunk_chain);
    run.d_used.reserve(run.n_chunks * 4);
    run.d_out[0].reserve(run.n_chunks * 4);
    run.d_out[1].reserve(run.n_chunks * 4);
    run.d_changed.reserve(8);
    iterate_sys_chains(s, run, st, 0);
    run.valid = true;
    return run.passes;
}

// ------------------------------------------------------------------------------- bias normalisation (a14)
// FragmentDistributionStats.cpp:3504-3582 CalculateBiasNormalization; the SumBias scans (Reference.cpp:622-659) run on
// the GPU, one launch for all (sequence, sampled length) pairs; partial sums are combined in a fixed order.
constexpr uint64_t kBiasWindow = 512ull << 20;       // start positions per pass of the bias sums: 2 x 4 GB of tracks
// partial sums and maxima of the chunks (kBiasBlock * kBiasRun start positions each, BiasPlan::chunk_ptr) whose first start position lies in
// the share [g_lo, g_hi) of the concatenated sequences; zero elsewhere
static void bias_partials(rsq_sim &s, hipStream_t st, const BiasPlan &plan, uint64_t g_lo, uint64_t g_hi, std::vector<double> &h_sum, std::vector<double> &h_max) {
    const bool trace = s.opt.trace_prepare != 0;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        HIP_CHECK(hipStreamSynchronize(st));
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "prepare:   bias sums: %-18s %8.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    };
    const uint32_t n_chunks = bias_chunks(plan);
    h_sum.assign(n_chunks, 0.0);
    h_max.assign(n_chunks, 0.0);
    if (!n_chunks) return;
    DevBuf d_params, d_chunk_param, d_chunk_ptr, d_sum, d_max, d_start_bias, d_end_bias;
    d_params.upload(plan.params);
    d_chunk_param.upload(plan.chunk_param);
    d_chunk_ptr.upload(plan.chunk_ptr);
    d_sum.reserve((size_t)n_chunks * 8);
    d_max.reserve((size_t)n_chunks * 8);
    HIP_CHECK(hipMemsetAsync(d_sum.as<double>(), 0, (size_t)n_chunks * 8, st));
    HIP_CHECK(hipMemsetAsync(d_max.as<double>(), 0, (size_t)n_chunks * 8, st));
    lap("chunk tables");
    // The share in windows of kBiasWindow positions: the tracks of one window (16 bytes per position; allocating them for a whole human-sized
    // reference takes a second) are computed, its chunks summed, the buffers used again.  A window's chunks read start positions up to a chunk
    // behind its end and end positions a fragment length further.
    const uint64_t lo = std::min<uint64_t>(g_lo, s.total_ref_size), hi = std::min<uint64_t>(g_hi, s.total_ref_size);
    uint64_t window = kBiasWindow;
    if (s.opt.bias_window > 0) window = (uint64_t)s.opt.bias_window;
    const uint64_t reach = (uint64_t)kBiasBlock * kBiasRun + s.dev.insert_to, track_len = std::min(hi - lo, window) + reach;
    d_start_bias.reserve(track_len * 8 + 16);
    d_end_bias.reserve(track_len * 8 + 16);
    lap("track buffers");
    for (uint64_t a = lo; a < hi; a += window) {
        const uint64_t b = std::min(hi, a + window), w_hi = std::min<uint64_t>(s.total_ref_size, b + reach);
        hipLaunchKernelGGL(k_surrounding_bias_tracks, dim3((uint32_t)cdiv(w_hi - a, 256)), dim3(256), 0, st, s.dev, d_start_bias.as<double>(), d_end_bias.as<double>(), a, w_hi);
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(k_sum_bias, dim3(n_chunks), dim3(kBiasBlock), 0, st, s.dev, d_params.as<BiasParam>(), d_chunk_param.as<uint32_t>(), d_chunk_ptr.as<uint32_t>(),
                           d_start_bias.as<double>(), d_end_bias.as<double>(), a, d_sum.as<double>(), d_max.as<double>(), a, b);
        HIP_CHECK(hipGetLastError());
    }
    lap("track and sum kernels");
    HIP_CHECK(hipMemcpyAsync(h_sum.data(), d_sum.as<double>(), h_sum.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(h_max.data(), d_max.as<double>(), h_max.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    lap("download");
}
static void bias_normalization(rsq_sim &s, hipStream_t st) {
    const BiasPlan plan = plan_bias_normalization(s, s.up);
    std::vector<double> h_sum, h_max;
    bias_partials(s, st, plan, 0, UINT64_MAX, h_sum, h_max);
    normalization_from_partials(s, s.up, plan, h_sum.data(), h_max.data());
}

static void prepare(rsq_sim &s, uint64_t seed, uint64_t num_read_pairs, double coverage, int ref_bias_mode, const char *base_identifier, hipStream_t st) {
    HIP_CHECK(hipSetDevice(s.device));
    const bool trace = s.opt.trace_prepare != 0;                // stage times of the pre-pass on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto lap = [&](const char *what, std::chrono::steady_clock::time_point &t0) {
        if (trace) fprintf(stderr, "prepare: %-28s %8.3f s\n", what, std::chrono::duration<double>(now() - t0).count());
        t0 = now();
    };
    auto t0 = now();
    plan_simulation(s, s.up, seed, num_read_pairs, coverage, ref_bias_mode, base_identifier);
    lap("plan", t0);
    s.planned = false;
    if (s.has_ref) {
        bias_normalization(s, st);
        lap("bias normalisation", t0);
    }
    s.passes = run_sys_chains(s, st, s.has_ref ? kChainsSimulation : kChainsAdapters);
    HIP_CHECK(hipStreamSynchronize(st));
    lap("systematic-error chains", t0);
    if (s.has_variants && s.chain_run.valid) variant_sys_errors_from_run(s, st);      // -V: the variants' bases, from the finished chains
    lap("variants' systematic errors", t0);
    s.prepared = true;
    s.prepared_lo = 1;
    s.prepared_hi = s.total_blocks + 1;
    // the read kernel for this profile (rsq_spec.h), so that no compilation falls into the first rsq_sim_pairs
    if (s.has_ref) (void)spec_kernel(s, SpecKind::kReads, effective_fill_mask(s.dev.lds.mask, s.force_fill_mode), s.has_variants, fill_is_binned(s));
    lap("read kernel for the profile", t0);
}

// Simulator::CreateSystematicErrorProfile (Simulator.cpp:2597-2653): both strands of every sequence, reverse first, as FASTQ.
// The reference reads sys_gc_range_ uninitialised in this mode (it is only set in Simulate, :2782); here it has that value.
static void create_sys_error_profile(rsq_sim &s, uint64_t seed, const char *path, hipStream_t st) {
    if (!s.has_ref) throw Error("a reference is needed to draw a systematic error profile");
    HIP_CHECK(hipSetDevice(s.device));
    s.dev.seed = seed;
    set_sys_gc_range(s);
    run_sys_chains(s, st, kChainsProfile);
    s.prepared = false;                                             // the simulation tracks were overwritten: prepare again before simulating
    std::string text;
    std::vector<uint16_t> track;
    std::vector<uint8_t> dom, rate;
    for (uint32_t i = 0; i < s.dev.n_seqs; ++i)
        for (uint32_t strand = 2; strand--;) {
            const uint32_t L = s.seq_len[i];
            track.resize(L);
            dom.resize(L);
            rate.resize(L);
            HIP_CHECK(hipMemcpy(track.data(), (strand ? s.sys_rev : s.sys_fwd) + s.seq_base_off[i], (size_t)L * 2, hipMemcpyDeviceToHost));
            for (uint32_t k = 0; k < L; ++k) {
                dom[k] = (uint8_t)(track[k] & 0xFF);
                rate[k] = (uint8_t)(track[k] >> 8);
            }
            text += sys_error_fastq_record(s.ref_ids[i] + (strand ? " reverse" : " forward"), dom.data(), rate.data(), L);
        }
    write_text_file(path, text);
}

// --------------------------------------------------------------------------------------------- hot path
// out[i] = *init + in[0] + ... + in[i-1] for i = 0 .. n (init nullptr: 0); the total also goes to *total_out if given
static void exclusive_scan(rsq_sim &s, const uint32_t *in, uint64_t n, uint64_t *out, hipStream_t st, const uint64_t *init = nullptr, uint64_t *total_out = nullptr) {
    const uint32_t tiles = cdiv(n, kScanTile);
    s.cur->tile_sums.reserve((size_t)(tiles + 1) * 8);
    s.cur->scan_total.reserve(8);
    uint64_t *total = total_out ? total_out : s.cur->scan_total.as<uint64_t>();
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(tiles), dim3(kScanBlock), 0, st, in, n, s.cur->tile_sums.as<uint64_t>());
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(kScanTilesBlock), 0, st, s.cur->tile_sums.as<uint64_t>(), tiles, total, init);
    hipLaunchKernelGGL(k_scan_apply, dim3(tiles), dim3(kScanBlock), 0, st, in, n, s.cur->tile_sums.as<uint64_t>(), total, out);
    HIP_CHECK(hipGetLastError());
}

static RawLayout raw_layout(rsq_sim &s, uint64_t n_reads) {
    const uint64_t pitch = (n_reads + 63u) & ~(uint64_t)63u;            // word rows start on 256-byte boundaries
    s.cur->raw_seq.reserve((uint64_t)(s.read_stride / 4u) * pitch * 4 + 16);
    s.cur->raw_qual.reserve((uint64_t)(s.read_stride / 4u) * pitch * 4 + 16);
    s.cur->raw_ops.reserve((uint64_t)s.ops_stride * pitch * 4 + 16);
    s.cur->raw_meta.reserve(n_reads * sizeof(ReadMeta) + 16);
    return RawLayout{s.cur->raw_seq.as<uint32_t>(), s.cur->raw_qual.as<uint32_t>(), s.cur->raw_ops.as<uint32_t>(), s.cur->raw_meta.as<ReadMeta>(), pitch, nullptr, 0, nullptr};
}

// Reads binned by tile (the LDS plan holds one tile per image): keys, histogram, the bins' places and units, the scatter.  n_keys = n_tiles (pairs:
// both mates of a pair have the pair's tile) or 2 n_tiles (records); bins = (segment, tile).
template <class CountKernel>
static FillBins build_fill_bins(rsq_sim &s, uint64_t n_items, uint32_t n_keys, hipStream_t st, CountKernel &&count_keys, const Fragment *frags = nullptr,
                                const FragmentVar *fvars = nullptr) {
    if (n_items >= 0xFFFFFFFFull) throw Error("more than 2^32 reads in one call of a profile with tiles: use smaller block ranges");
    const uint32_t n_bins = 2 * s.dev.n_tiles;
    s.cur->bin_keys.reserve(n_items * 2 + 16);
    s.cur->bin_perm.reserve(n_items * 4 + 16);
    // [hist n_keys][cursor n_keys][bin_first n_bins][bin_count n_bins][next_chunk n_bins][workers n_bins][chunk_ptr n_bins + 1]
    const size_t words = 2 * (size_t)n_keys + 5 * (size_t)n_bins + 1;
    s.cur->bin_small.reserve(words * 4 + 16);
    uint32_t *hist = s.cur->bin_small.as<uint32_t>(), *cursor = hist + n_keys, *bin_first = cursor + n_keys, *bin_count = bin_first + n_bins, *next_chunk = bin_count + n_bins,
             *workers = next_chunk + n_bins, *chunk_ptr = workers + n_bins;
    HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)n_keys * 4, st));
    s.timers["bin_tiles"].start(st);
    count_keys(s.cur->bin_keys.as<uint16_t>(), hist);
    hipLaunchKernelGGL(k_bins_plan, dim3(1), dim3(1024), 0, st, hist, n_keys, n_bins, bin_first, bin_count, cursor, chunk_ptr, next_chunk, workers);
    if (frags) s.cur->bin_frags.reserve(n_items * sizeof(Fragment) + 16);
    if (fvars) s.cur->bin_fvars.reserve(n_items * sizeof(FragmentVar) + 16);
    hipLaunchKernelGGL(k_bin_scatter, dim3(cdiv(n_items, kBinBlock * kBinItemsPerThread)), dim3(kBinBlock), 0, st, s.cur->bin_keys.as<uint16_t>(), n_items, n_keys, cursor,
                       s.cur->bin_perm.as<uint32_t>(), frags, fvars, s.cur->bin_frags.as<Fragment>(), s.cur->bin_fvars.as<FragmentVar>());
    s.timers["bin_tiles"].stop(st);
    HIP_CHECK(hipGetLastError());
    return FillBins{s.cur->bin_perm.as<uint32_t>(), bin_first, bin_count, chunk_ptr, next_chunk, workers, n_bins, s.cur->bin_frags.as<Fragment>(), s.cur->bin_fvars.as<FragmentVar>()};
}
// the same opt-in for a kernel compiled for the profile (a module function).  The runtime of this image launches module functions with up to the device's 160 KB
// without it and knows the attribute for host-registered functions only, so an error here is not one of the launch -- the launch itself is checked.
static void spec_allow_lds(hipFunction_t fn, size_t lds_bytes) {
    if (lds_bytes > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) (void)hipGetLastError();
}
template <class Kernel>
static size_t fill_lds_bytes(const rsq_sim &s, bool screened, bool binned, Kernel kernel) {
    const size_t lds_bytes = (screened ? (size_t)s.dev.lds.total_words * 4u : 0) + (binned ? kSchedWords * 4u : 0);
    if (lds_bytes > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    return lds_bytes;
}
// Shape of a read kernel's launch: persistent workgroups, one (or two, when two images fit) per CU, not more than there are rounds of chunks -- and for a call
// that does not fill the device, FEWER WAVES PER WORKGROUP ON MORE CUs: a wave that has its SIMD to itself walks a chunk's 150 steps in 0.40 ms, four waves that
// start on one SIMD at the same instant take 1.0 ms for theirs (they run in phase: all in the Philox rounds, then all waiting for LDS; the waves of a long launch
// drift apart and fill each other's stalls -- DESIGN.md 4.5).  A seqToIllumina call on 107 000 records took 1.08 ms on 106 workgroups of 16 waves and takes half of
// that on 256 workgroups of 8 (tools/time_small_calls.py, profiles/r05_e_small_calls_*).  The kernels take any workgroup size up to their launch bound: the image
// is staged by blockDim.x threads, a wave's ring lies at its number in the workgroup.
struct FillShape {
    uint32_t blocks, threads;
};
static FillShape fill_shape(const rsq_sim &s, size_t lds_bytes, uint64_t n_items, uint32_t segments_per_item, uint32_t max_threads) {
    const uint32_t per_cu = lds_bytes * 2 <= kLdsBudgetBytes ? 2u : 1u;         // workgroups resident per CU (LDS image, 2048 threads)
    const uint64_t chunks = segments_per_item * ((n_items + 63) / 64), slots = (uint64_t)s.n_cu * per_cu;
    uint32_t waves = (uint32_t)std::min<uint64_t>(max_threads / 64u, std::max<uint64_t>(4u, cdiv(chunks, slots)));
    waves = std::min(max_threads / 64u, (waves + 3u) & ~3u);                 // whole waves per SIMD
    if (s.opt.fill_waves > 0) waves = (uint32_t)std::min<int64_t>(max_threads / 64u, s.opt.fill_waves);
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(slots, std::max<uint64_t>(2, cdiv(chunks, waves)));
    return FillShape{(blocks + 1u) & ~1u, waves * 64u};                     // segments alternate over blockIdx.x
}

// k_fill_reads: persistent waves, one workgroup per CU slot; MASK = quads per quality row (screened draws on the LDS image planned by
// pack_tables) or 0 (double precision from HBM: the reference path the tests compare with)
// they return the order of the raw arrays' rows: nullptr = row i is item i, else row i is item perm[i] (binned by tile)
template <uint32_t MASK, bool VAR, bool BINNED>
static const uint32_t *launch_fill_kernel(rsq_sim &s, const Fragment *frags, uint64_t n_pairs, uint64_t adapter_first, const RawLayout &raw, hipStream_t st, const FragmentVar *fvars) {
    FillBins bins{};
    if (BINNED)
        bins = build_fill_bins(s, n_pairs, s.dev.n_tiles, st, [&](uint16_t *keys, uint32_t *hist) {
            hipLaunchKernelGGL(k_pair_tiles, dim3(cdiv(n_pairs, kBinBlock)), dim3(kBinBlock), 0, st, s.dev, frags, fvars, n_pairs, adapter_first, keys, hist);
        }, frags, fvars);
    const size_t lds_bytes = fill_lds_bytes(s, MASK != 0, BINNED, &k_fill_reads<MASK, VAR, BINNED>);
    const FillShape shape = fill_shape(s, lds_bytes, n_pairs, 2, fill_block(VAR));
    const uint32_t blocks = shape.blocks, kBlock = shape.threads;
    s.cur->fill_counters.reserve(8);
    HIP_CHECK(hipMemsetAsync(s.cur->fill_counters.as<uint32_t>(), 0, 8, st));
    hipFunction_t spec = spec_kernel(s, SpecKind::kReads, MASK, VAR, BINNED);
    s.timers["fill_reads"].start(st);
    if (spec) {
        uint32_t *sizes = s.cur->sizes.as<uint32_t>(), *counters = s.cur->fill_counters.as<uint32_t>();
        RawLayout raw_arg = raw;
        void *args[] = {&s.dev, &s.names, &frags, &n_pairs, &adapter_first, &raw_arg, &sizes, &counters, &fvars, &bins};
        spec_allow_lds(spec, lds_bytes);
        HIP_CHECK(hipModuleLaunchKernel(spec, blocks, 1, 1, kBlock, 1, 1, (unsigned)lds_bytes, st, args, nullptr));
    } else
        hipLaunchKernelGGL((k_fill_reads<MASK, VAR, BINNED>), dim3(blocks), dim3(kBlock), lds_bytes, st, s.dev, s.names, frags, n_pairs, adapter_first, raw, s.cur->sizes.as<uint32_t>(),
                           s.cur->fill_counters.as<uint32_t>(), fvars, bins);
    s.timers["fill_reads"].stop(st);
    HIP_CHECK(hipGetLastError());
    return bins.perm;
}
template <uint32_t MASK, bool VAR = false>
static const uint32_t *launch_fill_mask(rsq_sim &s, const Fragment *frags, uint64_t n_pairs, uint64_t adapter_first, const RawLayout &raw, hipStream_t st, const FragmentVar *fvars = nullptr) {
    if constexpr (MASK != 0)
        if (fill_is_binned(s)) return launch_fill_kernel<MASK, VAR, true>(s, frags, n_pairs, adapter_first, raw, st, fvars);
    return launch_fill_kernel<MASK, VAR, false>(s, frags, n_pairs, adapter_first, raw, st, fvars);
}
template <uint32_t MASK, bool BINNED, bool PACKED>
static const uint32_t *launch_records_kernel(rsq_sim &s, const RecordJob &job, const uint8_t *seg_dev, uint64_t n, const RawLayout &raw, hipStream_t st) {
    FillBins bins{};
    if (BINNED)
        bins = build_fill_bins(s, n, 2 * s.dev.n_tiles, st, [&](uint16_t *keys, uint32_t *hist) {
            hipLaunchKernelGGL(k_record_tiles, dim3(cdiv(n, kBinBlock)), dim3(kBinBlock), 0, st, s.dev, seg_dev, job.first_index, n, keys, hist);
        });
    const size_t lds_bytes = fill_lds_bytes(s, MASK != 0, BINNED, &k_fill_records<MASK, BINNED, PACKED>);
    const FillShape shape = fill_shape(s, lds_bytes, n, 1, kFillBlockWalk);
    const uint32_t blocks = shape.blocks;
#if defined(RSQ_TRACE_FILL)
    s.cur->fill_counters.reserve(16 + 32 * (size_t)blocks);
    HIP_CHECK(hipMemsetAsync(s.cur->fill_counters.as<uint32_t>(), 0, 16 + 32 * (size_t)blocks, st));
#else
    s.cur->fill_counters.reserve(8);
    HIP_CHECK(hipMemsetAsync(s.cur->fill_counters.as<uint32_t>(), 0, 8, st));
#endif
    hipFunction_t spec = spec_kernel(s, SpecKind::kRecords, MASK, PACKED, BINNED);      // (the variant's `var` flag names the packed records here)
    s.timers["fill_reads"].start(st);
    if (spec) {
        uint32_t *counters = s.cur->fill_counters.as<uint32_t>();
        RecordJob job_arg = job;
        RawLayout raw_arg = raw;
        void *args[] = {&s.dev, &job_arg, &raw_arg, &counters, &bins};
        spec_allow_lds(spec, lds_bytes);
        HIP_CHECK(hipModuleLaunchKernel(spec, blocks, 1, 1, shape.threads, 1, 1, (unsigned)lds_bytes, st, args, nullptr));
    } else
        hipLaunchKernelGGL((k_fill_records<MASK, BINNED, PACKED>), dim3(blocks), dim3(shape.threads), lds_bytes, st, s.dev, job, raw, s.cur->fill_counters.as<uint32_t>(), bins);
    s.timers["fill_reads"].stop(st);
#if defined(RSQ_TRACE_FILL)
    {
        std::vector<uint64_t> t(2 + 4 * (size_t)blocks);
        HIP_CHECK(hipMemcpyAsync(t.data(), s.cur->fill_counters.as<uint64_t>(), t.size() * 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        uint64_t first = ~0ull, last_start = 0, last_end = 0, stage = 0, run = 0;
        for (uint32_t b = 0; b < blocks; ++b) {
            first = std::min(first, t[2 + 4 * b]);
            last_start = std::max(last_start, t[2 + 4 * b]);
            last_end = std::max(last_end, t[4 + 4 * b]);
            stage += t[3 + 4 * b] - t[2 + 4 * b];
            run += t[4 + 4 * b] - t[3 + 4 * b];
        }
        if (getenv("RSQ_TRACE_WORKGROUPS")) {                      // workgroup: start + duration until its last wave ended (us), chunks it ran
            std::string line;
            for (uint32_t b = 0; b < blocks; ++b) {
                line += " " + std::to_string(b) + ":" + std::to_string((t[2 + 4 * b] - first) / 100) + "+" + std::to_string((t[4 + 4 * b] - t[2 + 4 * b]) / 100) + "/" + std::to_string(t[5 + 4 * b]);
            }
            fprintf(stderr, "trace_fill workgroup:start+duration(us)/chunks:%s\n", line.c_str());
        }
        fprintf(stderr, "trace_fill: %u workgroups of %u threads, %llu items: last start %.1f us after the first, last end %.1f us; image staged in %.1f us, chunks %.1f us (means; 100 MHz clock)\n", blocks,
                shape.threads, (unsigned long long)n, (last_start - first) / 100.0, (last_end - first) / 100.0, stage / 100.0 / blocks, run / 100.0 / blocks);
    }
#endif
    HIP_CHECK(hipGetLastError());
    return bins.perm;
}
template <uint32_t MASK>
static const uint32_t *launch_records_mask(rsq_sim &s, const RecordJob &job, const uint8_t *seg_dev, uint64_t n, const RawLayout &raw, hipStream_t st) {
    if constexpr (MASK != 0)
        if (fill_is_binned(s)) return job.codes ? launch_records_kernel<MASK, true, true>(s, job, seg_dev, n, raw, st) : launch_records_kernel<MASK, true, false>(s, job, seg_dev, n, raw, st);
    return job.codes ? launch_records_kernel<MASK, false, true>(s, job, seg_dev, n, raw, st) : launch_records_kernel<MASK, false, false>(s, job, seg_dev, n, raw, st);
}
static const uint32_t *launch_fill_reads(rsq_sim &s, const Fragment *frags, uint64_t n_pairs, uint64_t adapter_first, const RawLayout &raw, hipStream_t st,
                              const FragmentVar *fvars = nullptr) {
    const uint32_t mask = effective_fill_mask(s.dev.lds.mask, s.force_fill_mode);
    const bool var = frags && s.has_variants;            // with variants the error walk is per lane state
#define RSQ_FILL_CASE(Q)                                                                       \
    if (mask == Q) {                                                                                  \
        if (var) return launch_fill_mask<Q, true>(s, frags, n_pairs, adapter_first, raw, st, fvars);  \
        return launch_fill_mask<Q>(s, frags, n_pairs, adapter_first, raw, st);                        \
    }
    RSQ_FILL_CASE(0u)
    RSQ_FILL_CASE(kQualityQuads[0])
    RSQ_FILL_CASE(kQualityQuads[1])
    RSQ_FILL_CASE(kQualityQuads[2])
    RSQ_FILL_CASE(kQualityQuads[3])
    RSQ_FILL_CASE(kQualityQuads[4])
    static_assert(sizeof(kQualityQuads) == 5 * sizeof(uint32_t), "one case per entry");
#undef RSQ_FILL_CASE
    throw Error("no k_fill_reads instantiation for " + std::to_string(mask) + " quads");
}
static const uint32_t *launch_fill_records(rsq_sim &s, const RecordJob &job, const uint8_t *seg_dev, uint64_t n, const RawLayout &raw, hipStream_t st) {
    const uint32_t mask = effective_fill_mask(s.dev.lds.mask, s.force_fill_mode);
#define RSQ_REC_CASE(Q) \
    if (mask == Q) return launch_records_mask<Q>(s, job, seg_dev, n, raw, st);
    RSQ_REC_CASE(0u)
    RSQ_REC_CASE(kQualityQuads[0])
    RSQ_REC_CASE(kQualityQuads[1])
    RSQ_REC_CASE(kQualityQuads[2])
    RSQ_REC_CASE(kQualityQuads[3])
    RSQ_REC_CASE(kQualityQuads[4])
#undef RSQ_REC_CASE
    throw Error("no k_fill_records instantiation for " + std::to_string(mask) + " quads");
}

// ---- the stages of one (sub-)range; they work on the simulator's current workspace (s.cur)
// reads of n_pairs pairs into the raw arrays (fragments on the device, or adapter-only pairs when frags == nullptr) and their record sizes
struct ReadsDone {
    RawLayout raw;
    const uint32_t *row_order;
};
static ReadsDone reads_stage(rsq_sim &s, const Fragment *frags, uint64_t n_pairs, uint64_t adapter_first, hipStream_t st, const FragmentVar *fvars) {
    RawLayout raw = raw_layout(s, 2 * n_pairs);
    s.cur->sizes.reserve(2 * n_pairs * 4 + 16);
    if (frags && s.dev.meth_ptr && !fvars) {                         // --methylation: CTConversion of both mates' templates first
        s.cur->templates.reserve(2 * n_pairs * s.template_words * 8 + 16);
        raw.templates = s.cur->templates.as<uint64_t>();
        raw.template_words = s.template_words;
        hipLaunchKernelGGL(k_methylation_templates, dim3(cdiv(2 * n_pairs, 256)), dim3(256), 0, st, s.dev, frags, n_pairs, raw);
        HIP_CHECK(hipGetLastError());
    }
    if (frags && fvars) {                                            // variants of any kind: both mates' templates with the allele's variants
        s.cur->templates.reserve(2 * n_pairs * s.template_words * 8 + 16);
        raw.templates = s.cur->templates.as<uint64_t>();
        raw.template_words = s.template_words;
        s.timers["variant_templates"].start(st);
        hipLaunchKernelGGL(k_variant_templates, dim3(cdiv(2 * n_pairs, 256)), dim3(256), 0, st, s.dev, frags, fvars, n_pairs, raw);
        s.timers["variant_templates"].stop(st);
        HIP_CHECK(hipGetLastError());
    }
    const uint32_t *row_order = launch_fill_reads(s, frags, n_pairs, adapter_first, raw, st, fvars);
    return ReadsDone{raw, row_order};
}
// FASTQ text of the pairs of reads_stage: their offsets continue at totals[2 * part] / [2 * part + 1] (bytes of text in front of this part, per
// file), where the part ends goes to totals[2 * (part + 1)] / [.. + 1].  The kernel refuses to write past the caller's capacity.
static void text_stage(rsq_sim &s, const Fragment *frags, uint64_t n_pairs, uint64_t adapter_first, const ReadsDone &rd, char *r1, size_t r1_cap, char *r2, size_t r2_cap,
                       uint32_t part, hipStream_t st, const FragmentVar *fvars) {
    rsq_sim::Workspace &w = *s.cur;
    w.off_r1.reserve((n_pairs + 1) * 8);
    w.off_r2.reserve((n_pairs + 1) * 8);
    uint64_t *totals = s.totals.as<uint64_t>();
    s.timers["scan"].start(st);
    exclusive_scan(s, w.sizes.as<uint32_t>(), n_pairs, w.off_r1.as<uint64_t>(), st, totals + 2 * part, totals + 2 * (part + 1));
    exclusive_scan(s, w.sizes.as<uint32_t>() + n_pairs, n_pairs, w.off_r2.as<uint64_t>(), st, totals + 2 * part + 1, totals + 2 * (part + 1) + 1);
    s.timers["scan"].stop(st);
    const dim3 grid(cdiv(n_pairs, kFormatRecords), 2), block(64);
    // the wave's LDS image: sized by the longest record of the call before (the first call takes room for records of 480 bytes); binned rows have a slot per
    // record, each with its own alignment.  This call's longest record goes to totals' last word for the next one.
    const uint32_t lds = rd.row_order ? std::min(kFormatLdsMax, format_lds_bytes(s.format_record_bytes) + 16u * kFormatRecords) : format_lds_bytes(s.format_record_bytes);
    hipLaunchKernelGGL(k_max_size, dim3(std::min<uint64_t>(1024, cdiv(2 * n_pairs, 256))), dim3(256), 0, st, w.sizes.as<uint32_t>(), 2 * n_pairs, s.longest_record.as<uint32_t>());
    s.timers["format_write"].start(st);
    if (rd.row_order)
        hipLaunchKernelGGL(k_format_write<true>, grid, block, lds, st, s.dev, s.names, frags, n_pairs, adapter_first, rd.raw, w.off_r1.as<uint64_t>(), w.off_r2.as<uint64_t>(), r1, r2,
                           (uint64_t)(r1 ? r1_cap : 0), (uint64_t)(r2 ? r2_cap : 0), fvars, rd.row_order, lds);
    else
        hipLaunchKernelGGL(k_format_write<false>, grid, block, lds, st, s.dev, s.names, frags, n_pairs, adapter_first, rd.raw, w.off_r1.as<uint64_t>(), w.off_r2.as<uint64_t>(), r1, r2,
                           (uint64_t)(r1 ? r1_cap : 0), (uint64_t)(r2 ? r2_cap : 0), fvars, (const uint32_t *)nullptr, lds);
    s.timers["format_write"].stop(st);
    HIP_CHECK(hipGetLastError());
}
static void reset_call_timers(rsq_sim &s) {
    for (auto &t : s.timers) t.second.reset();
}
// what a call ends with: the bytes of text of `parts` parts, the variant walk's error flag; the call's streams are idle afterwards
static int finish_call(rsq_sim &s, uint32_t parts, bool with_variants, char *r1, size_t r1_cap, size_t *r1_len, char *r2, size_t r2_cap, size_t *r2_len, hipStream_t text_stream) {
    s.mailbox[4] = 0;
    if (with_variants) HIP_CHECK(hipMemcpyAsync(&s.mailbox[4], s.dev.walk_error, 4, hipMemcpyDeviceToHost, text_stream));
    HIP_CHECK(hipMemcpyAsync(&s.mailbox[2], s.totals.as<uint64_t>() + 2 * parts, 16, hipMemcpyDeviceToHost, text_stream));
    HIP_CHECK(hipMemcpyAsync(&s.mailbox[5], s.longest_record.as<uint32_t>(), 4, hipMemcpyDeviceToHost, text_stream));
    HIP_CHECK(hipStreamSynchronize(text_stream));
    *r1_len = s.mailbox[2];
    *r2_len = s.mailbox[3];
    if ((uint32_t)s.mailbox[4]) {
        HIP_CHECK(hipMemsetAsync(s.dev.walk_error, 0, 4, text_stream));
        HIP_CHECK(hipStreamSynchronize(text_stream));
        g_last_error = kWalkErrorMessage;
        return RSQ_EINVAL;
    }
    if (*r1_len > r1_cap || *r2_len > r2_cap || !r1 || !r2) {
        g_last_error = "output buffers too small: need " + std::to_string(*r1_len) + " and " + std::to_string(*r2_len) + " bytes";
        return RSQ_ENOSPC;
    }
    return RSQ_OK;
}
static void begin_totals(rsq_sim &s, uint32_t parts, hipStream_t st) {
    s.totals.reserve((size_t)(parts + 1) * 16 + 16);
    s.longest_record.reserve(8);
    HIP_CHECK(hipMemsetAsync(s.totals.as<uint64_t>(), 0, 16, st));
    HIP_CHECK(hipMemsetAsync(s.longest_record.as<uint32_t>(), 0, 4, st));
}

// reads + FASTQ text of adapter-only pairs (Simulator::SimulateAdapterOnlyPairs, Simulator.cpp:2359-2382): one part on the caller's stream
static int adapter_only_pairs(rsq_sim &s, uint64_t n_pairs, uint64_t adapter_first, char *r1, size_t r1_cap, size_t *r1_len, char *r2, size_t r2_cap, size_t *r2_len, hipStream_t st) {
    *r1_len = *r2_len = 0;
    if (!n_pairs) return RSQ_OK;
    reset_call_timers(s);
    s.cur = &s.ws[0];
    begin_totals(s, 1, st);
    const ReadsDone rd = reads_stage(s, nullptr, n_pairs, adapter_first, st, nullptr);
    text_stage(s, nullptr, n_pairs, adapter_first, rd, r1, r1_cap, r2, r2_cap, 0, st, nullptr);
    return finish_call(s, 1, false, r1, r1_cap, r1_len, r2, r2_cap, r2_len, st);
}

// The sieve of one block range on the current workspace: attempt 0 is launched without waiting; collect() waits for it, and if a list overflowed
// repeats the pass with the exact sizes (capacities come from the expected number of passing cells + 6 sigma, so that is rare).
struct SieveRun {
    uint32_t block_lo = 0, block_hi = 0;
    uint64_t n_slots = 0, cand_cap = 0, hit_cap = 0, total = 0, n_cands = 0;
    uint32_t n_hits = 0;
    const SlotInfo *slot_table = nullptr;
};
static void sieve_attempt(rsq_sim &s, SieveRun &r, int attempt, uint64_t *mail, hipStream_t st) {
    rsq_sim::Workspace &w = *s.cur;
    const int vm = s.variants_mode;
    if (r.cand_cap >= (1ull << 32)) throw Error("more than 2^32 sieve candidates in one call: use smaller block ranges");
    w.cands.reserve(r.cand_cap * sizeof(SieveCand));
    w.pairs_of.reserve(r.cand_cap * 4 + 16);
    w.pair_off.reserve((r.cand_cap + 1) * 8);
    if (vm) w.hits.reserve(r.hit_cap * sizeof(SieveHit));
    else w.cell_info.reserve(r.cand_cap * 4 + 16);
    HIP_CHECK(hipMemsetAsync(w.hit_count.as<uint32_t>(), 0, 4, st));
    s.timers["sieve"].start(st);
    const uint32_t block_lo = r.block_lo, block_hi = r.block_hi;
    const uint64_t n_slots = r.n_slots, cand_cap = r.cand_cap;
    const dim3 ggrid(cdiv(n_slots, kSieveBlock)), gblock(kSieveBlock);
    if (!attempt) {
        if (2 == vm) {
            w.slot_table.reserve(n_slots * sizeof(SlotInfo) + 16);
            s.timers["slot_table"].start(st);
            hipLaunchKernelGGL(k_slot_table, dim3(block_hi - block_lo), dim3(256), 0, st, s.dev, block_lo, w.slot_table.as<SlotInfo>());
            s.timers["slot_table"].stop(st);
            r.slot_table = w.slot_table.as<SlotInfo>();
        }
        s.timers["sieve_screen"].start(st);
        if (2 == vm)
            hipLaunchKernelGGL((k_sieve_gaps<2, false>), ggrid, gblock, 0, st, s.dev, block_lo, block_hi, (uint32_t)n_slots, w.counts.as<uint32_t>(), (const uint64_t *)nullptr,
                               (SieveCand *)nullptr, cand_cap, r.slot_table);
        else
            hipLaunchKernelGGL((k_sieve_gaps<0, false>), ggrid, gblock, 0, st, s.dev, block_lo, block_hi, (uint32_t)n_slots, w.counts.as<uint32_t>(), (const uint64_t *)nullptr,
                               (SieveCand *)nullptr, cand_cap, r.slot_table);
        s.timers["sieve_screen"].stop(st);
        exclusive_scan(s, w.counts.as<uint32_t>(), n_slots, w.offsets.as<uint64_t>(), st);
    }
    const SlotInfo *slot_table = r.slot_table;
    if (2 == vm)
        hipLaunchKernelGGL((k_sieve_gaps<2, true>), ggrid, gblock, 0, st, s.dev, block_lo, block_hi, (uint32_t)n_slots, (uint32_t *)nullptr, w.offsets.as<uint64_t>(),
                           w.cands.as<SieveCand>(), cand_cap, slot_table);
    else
        hipLaunchKernelGGL((k_sieve_gaps<0, true>), ggrid, gblock, 0, st, s.dev, block_lo, block_hi, (uint32_t)n_slots, (uint32_t *)nullptr, w.offsets.as<uint64_t>(),
                           w.cands.as<SieveCand>(), cand_cap, slot_table);
    const dim3 fgrid(cdiv(cand_cap, kSieveBlock)), fblock(kSieveBlock);
#define RSQ_FINISH(VM, CAP)                                                                                                                                   \
    hipLaunchKernelGGL((k_sieve_finish<VM, CAP>), fgrid, fblock, 0, st, s.dev, block_lo, block_hi, (uint32_t)n_slots, w.offsets.as<uint64_t>(), w.cands.as<SieveCand>(),  \
                       cand_cap, w.pairs_of.as<uint32_t>(), w.hits.as<SieveHit>(), (uint32_t)std::min<uint64_t>(r.hit_cap, 0xFFFFFFFFull), w.hit_count.as<uint32_t>(), slot_table, \
                       w.cell_info.as<uint32_t>())
    if (2 == vm && s.num_alleles <= 8) RSQ_FINISH(2, 8);        // few alleles: the cell's (allele, strand) slots stay in registers
    else if (2 == vm) RSQ_FINISH(2, kMaxDevAlleles);
    else if (1 == vm) RSQ_FINISH(1, 8);                         // allele copies exist for at most eight alleles
    else RSQ_FINISH(0, 1);
#undef RSQ_FINISH
    s.timers["sieve"].stop(st);
    HIP_CHECK(hipGetLastError());
    exclusive_scan(s, w.pairs_of.as<uint32_t>(), cand_cap, w.pair_off.as<uint64_t>(), st);
    mail[1] = 0;
    HIP_CHECK(hipMemcpyAsync(&mail[0], w.pair_off.as<uint64_t>() + cand_cap, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(&mail[1], w.hit_count.as<uint32_t>(), 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(&mail[5], w.offsets.as<uint64_t>() + n_slots, 8, hipMemcpyDeviceToHost, st));
}
// false: the range has no start position slots at all
static bool sieve_launch(rsq_sim &s, SieveRun &r, uint32_t block_lo, uint32_t block_hi, uint64_t *mail, hipStream_t st) {
    rsq_sim::Workspace &w = *s.cur;
    const int vm = s.variants_mode;
    r = SieveRun{};
    r.block_lo = block_lo;
    r.block_hi = block_hi;
    r.n_slots = (uint64_t)(block_hi - block_lo) * kBlockSize;
    if (2 == vm && block_hi > block_lo) {                            // plus the starts inside inserted bases of these blocks
        uint32_t ptr[2];
        HIP_CHECK(hipMemcpy(&ptr[0], s.dev.block_extra_ptr + block_lo, 4, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(&ptr[1], s.dev.block_extra_ptr + block_hi, 4, hipMemcpyDeviceToHost));
        r.n_slots += ptr[1] - ptr[0];
    }
    if (!r.n_slots) return false;
    if (r.n_slots >= (1ull << 32) - kSieveBlock)                     // slot indices are 32 bits wide inside one call
        throw Error("block range too large for one call: at most " + std::to_string(((1ull << 32) - kSieveBlock) / kBlockSize - 1) + " blocks");
    w.counts.reserve(r.n_slots * 4 + 16);
    w.offsets.reserve((r.n_slots + 1) * 8);
    w.hit_count.reserve(8);
    // capacity of the candidate list: the cells expected to pass the zero threshold plus six standard deviations; of the hit list:
    // cells with fragments <= candidates (with variants a cell has one record per two chosen (allele, strand) slots)
    const double expected_cands = s.expected_passing * (double)r.n_slots;
    r.cand_cap = std::max<uint64_t>(w.cands.bytes() / sizeof(SieveCand), (uint64_t)(expected_cands + 6.0 * sqrt(expected_cands + 1.0)) + 65536);
    r.hit_cap = std::max<uint64_t>(w.hits.bytes() / sizeof(SieveHit), 0 == vm ? r.cand_cap : r.cand_cap + r.cand_cap / 4);
    sieve_attempt(s, r, 0, mail, st);
    return true;
}
static void sieve_collect(rsq_sim &s, SieveRun &r, uint64_t *mail, hipStream_t st) {
    for (int attempt = 0;; ++attempt) {
        HIP_CHECK(hipStreamSynchronize(st));
        r.total = mail[0];
        r.n_hits = (uint32_t)mail[1];
        r.n_cands = mail[5];
        if (r.n_cands <= r.cand_cap && r.n_hits <= r.hit_cap) return;
        if (attempt >= 2) throw Error("sieve lists overflowed three times");
        // the candidate count is exact; the hit count is exact once the candidates fit, before that it is scaled up with them
        const double grow = r.n_cands > r.cand_cap ? (double)r.n_cands / (double)r.cand_cap : 1.0;
        if (r.n_hits > r.hit_cap || grow > 1.0) r.hit_cap = std::max<uint64_t>(r.hit_cap, (uint64_t)((double)r.n_hits * grow * 1.25) + 65536);
        r.cand_cap = std::max(r.cand_cap, r.n_cands + 65536);
        sieve_attempt(s, r, attempt + 1, mail, st);
    }
}
static void sieve_emit(rsq_sim &s, const SieveRun &r, hipStream_t st) {
    rsq_sim::Workspace &w = *s.cur;
    w.frags.reserve(r.total * sizeof(Fragment) + 16);
    s.timers["sieve_emit"].start(st);
    if (2 == s.variants_mode) {
        w.fvars.reserve(r.total * sizeof(FragmentVar) + 16);
        hipLaunchKernelGGL(k_sieve_emit<2>, dim3(cdiv(r.n_hits, 256)), dim3(256), 0, st, s.dev, r.block_lo, r.block_hi, w.hits.as<SieveHit>(), r.n_hits, w.offsets.as<uint64_t>(),
                           w.pair_off.as<uint64_t>(), w.frags.as<Fragment>(), w.fvars.as<FragmentVar>(), w.slot_table.as<SlotInfo>());
    } else if (1 == s.variants_mode)                                // allele copies: the hit list (a cell may have several records), slots as without variants
        hipLaunchKernelGGL(k_sieve_emit<1>, dim3(cdiv(r.n_hits, 256)), dim3(256), 0, st, s.dev, r.block_lo, r.block_hi, w.hits.as<SieveHit>(), r.n_hits, w.offsets.as<uint64_t>(),
                           w.pair_off.as<uint64_t>(), w.frags.as<Fragment>(), (FragmentVar *)nullptr, (const SlotInfo *)nullptr);
    else
        hipLaunchKernelGGL(k_sieve_emit<0>, dim3(cdiv(r.n_cands, 256)), dim3(256), 0, st, s.dev, r.block_lo, r.block_hi, (const SieveHit *)nullptr, (uint32_t)r.n_cands,
                           w.offsets.as<uint64_t>(), w.pair_off.as<uint64_t>(), w.frags.as<Fragment>(), (FragmentVar *)nullptr, (const SlotInfo *)nullptr, w.cands.as<SieveCand>(),
                           w.pairs_of.as<uint32_t>(), w.cell_info.as<uint32_t>());
    s.timers["sieve_emit"].stop(st);
    HIP_CHECK(hipGetLastError());
}

// SimulateFromGivenBlock + CreateReads + Output for the blocks [block_lo, block_hi) (Simulator.cpp:2249-2357, 634-721, 215-230).  Blocks are independent
// (:2384-2401), so a range can be cut into `parts` sub-ranges that move through three stages -- sieve, reads, FASTQ text -- on three streams: the read
// kernel owns every CU while it runs (160 KB of LDS, all vector registers), so nothing overlaps IT; the sieve of part k + 1 and the text of part k run side
// by side behind it.  That does not pay (pairs_parts), so a call is one part unless option overlap asks for more; what stays is the stage structure.
//   reads(k) waits for: fragments of part k (sieve stream), the text of part k - 2 (same workspace);
//   sieve(k + 1) waits for: reads(k) launched and done (else it would only take CUs from it), the text of part k - 1 (same workspace);
//   text(k) waits for reads(k).
// The host meets the sieve once per part (list sizes decide the next launches), everything else is ordered by events.
static uint32_t pairs_parts(const rsq_sim &s, uint32_t n_blocks) {
    // Measured (bench workload, 10 M pairs): 1 part 81.6 ms per step, 2 parts 83.0, 4 parts 84.3, 8 parts 87.7 -- the sieve's candidate kernel and the
    // formatter both live on the memory system (gathers in the 24 MB surrounding table; 15 GB of raw arrays and text), so side by side each takes nearly as long
    // as the two in a row, and every part adds a read-kernel tail.  One part unless asked (option overlap = n, tests).
    const int64_t opt = s.opt.overlap;
    return opt > 0 ? (uint32_t)std::min<int64_t>(opt, std::max<uint32_t>(n_blocks, 1u)) : 1u;
}
// what rsq_sim_pairs and rsq_sim_job_generate ask of a block range before anything is sized from it
static int check_block_range(const rsq_sim &s, uint32_t block_lo, uint32_t block_hi) {
    if (!s.prepared || !s.has_ref) {
        g_last_error = "rsq_sim_prepare with a reference must run before rsq_sim_pairs";
        return RSQ_ESTATE;
    }
    if (block_lo < 1 || block_hi > s.total_blocks + 1 || block_lo > block_hi) {
        g_last_error = "block range outside [1, total_blocks]";
        return RSQ_EINVAL;
    }
    if (block_lo < block_hi && (block_lo < s.prepared_lo || block_hi > s.prepared_hi)) {       // a sharded pre-pass finished the tracks of the rank's own blocks only
        g_last_error = "blocks [" + std::to_string(block_lo) + ", " + std::to_string(block_hi) + ") lie outside the range the sharded pre-pass prepared on this simulator, [" +
                       std::to_string(s.prepared_lo) + ", " + std::to_string(s.prepared_hi) + ")";
        return RSQ_ESTATE;
    }
    return RSQ_OK;
}
static int sim_pairs(rsq_sim &s, uint32_t block_lo, uint32_t block_hi, char *r1, size_t r1_cap, size_t *r1_len, char *r2, size_t r2_cap, size_t *r2_len, uint64_t *n_pairs,
                     rsq_fragment *frags_out, size_t frags_cap, hipStream_t st) {
    if (const int rc = check_block_range(s, block_lo, block_hi)) return rc;
    HIP_CHECK(hipSetDevice(s.device));
    *n_pairs = 0;
    *r1_len = *r2_len = 0;
    if (block_lo == block_hi) return RSQ_OK;
    reset_call_timers(s);
    const int vm = s.variants_mode;
    const uint32_t parts = pairs_parts(s, block_hi - block_lo);
    // one part: everything on the caller's stream; more: the reads stay there, sieve and text get their own
    hipStream_t s_reads = st, s_sieve = parts > 1 ? s.side[0] : st, s_text = parts > 1 ? s.side[1] : st;
    if (parts > 1) {                                                 // the side streams begin where the caller's stream is
        HIP_CHECK(hipEventRecord(s.ev_call, st));
        HIP_CHECK(hipStreamWaitEvent(s_sieve, s.ev_call, 0));
        HIP_CHECK(hipStreamWaitEvent(s_text, s.ev_call, 0));
    }
    begin_totals(s, parts, s_text);
    struct Abort {                                                   // an error in the middle: nothing of the call may still run when it returns
        rsq_sim &s;
        bool armed = true;
        ~Abort() {
            if (armed) (void)hipDeviceSynchronize();
        }
    } abort_guard{s};
    auto part_range = [&](uint32_t k) {
        const uint64_t n = block_hi - block_lo;
        return std::pair<uint32_t, uint32_t>(block_lo + (uint32_t)(n * k / parts), block_lo + (uint32_t)(n * (k + 1) / parts));
    };
    SieveRun run[2];
    bool has_slots[2] = {false, false};
    s.cur = &s.ws[0];
    has_slots[0] = sieve_launch(s, run[0], part_range(0).first, part_range(0).second, s.mailbox + 0, s_sieve);
    uint64_t total_pairs = 0;
    static_assert(sizeof(rsq_fragment) == sizeof(Fragment), "ABI fragment layout");
    for (uint32_t k = 0; k < parts; ++k) {
        const uint32_t p = k & 1u;
        s.cur = &s.ws[p];
        SieveRun &r = run[p];
        uint64_t *mail = s.mailbox + 8 * p;
        if (has_slots[p]) sieve_collect(s, r, mail, s_sieve);
        const uint64_t n = has_slots[p] ? r.total : 0;
        const Fragment *frags = s.cur->frags.as<Fragment>();
        const FragmentVar *fvars = nullptr;
        if (n) {
            sieve_emit(s, r, s_sieve);
            frags = s.cur->frags.as<Fragment>();
            fvars = 2 == vm ? s.cur->fvars.as<FragmentVar>() : nullptr;
            if (frags_out) {
                if (frags_cap < total_pairs + n) {
                    g_last_error = "fragment buffer too small: need at least " + std::to_string(total_pairs + n) + " records";
                    return RSQ_ENOSPC;
                }
                HIP_CHECK(hipMemcpyAsync(frags_out + total_pairs, frags, n * sizeof(Fragment), hipMemcpyDeviceToDevice, s_sieve));
            }
            if (parts > 1) {
                HIP_CHECK(hipEventRecord(s.ev_emit, s_sieve));
                HIP_CHECK(hipStreamWaitEvent(s_reads, s.ev_emit, 0));
                if (k >= 2) HIP_CHECK(hipStreamWaitEvent(s_reads, s.cur->text_done, 0));       // the raw arrays of this workspace were last read by text(k - 2)
            }
            const ReadsDone rd = reads_stage(s, frags, n, 0, s_reads, fvars);
            if (parts > 1) {
                HIP_CHECK(hipEventRecord(s.ev_fill, s_reads));
                HIP_CHECK(hipStreamWaitEvent(s_text, s.ev_fill, 0));
            }
            text_stage(s, frags, n, 0, rd, r1, r1_cap, r2, r2_cap, k, s_text, fvars);
        } else {                                                     // no pairs in this part: its text ends where it begins
            HIP_CHECK(hipMemcpyAsync(s.totals.as<uint64_t>() + 2 * (k + 1), s.totals.as<uint64_t>() + 2 * k, 16, hipMemcpyDeviceToDevice, s_text));
        }
        if (parts > 1) HIP_CHECK(hipEventRecord(s.cur->text_done, s_text));
        total_pairs += n;
        if (k + 1 < parts) {                                         // the next part's sieve, on the other workspace
            s.cur = &s.ws[p ^ 1u];
            if (n) HIP_CHECK(hipStreamWaitEvent(s_sieve, s.ev_fill, 0));
            if (k >= 1) HIP_CHECK(hipStreamWaitEvent(s_sieve, s.cur->text_done, 0));
            has_slots[p ^ 1u] = sieve_launch(s, run[p ^ 1u], part_range(k + 1).first, part_range(k + 1).second, s.mailbox + 8 * (p ^ 1u), s_sieve);
        }
    }
    *n_pairs = total_pairs;
    const int rc = total_pairs ? finish_call(s, parts, s.has_variants, r1, r1_cap, r1_len, r2, r2_cap, r2_len, s_text) : (int)RSQ_OK;
    if (total_pairs && (rc == RSQ_OK || rc == RSQ_ENOSPC) && (uint32_t)s.mailbox[5]) s.format_record_bytes = (uint32_t)s.mailbox[5] + 8u;      // for the next call's text stage
    if (parts > 1) {                                                 // the caller's stream continues behind the whole call
        HIP_CHECK(hipStreamSynchronize(s_sieve));
        HIP_CHECK(hipStreamSynchronize(s_text));
        HIP_CHECK(hipStreamSynchronize(s_reads));
    } else HIP_CHECK(hipStreamSynchronize(st));
    abort_guard.armed = false;
    return rc;
}

// seqToIllumina's FASTQ text on the device (Simulator.cpp:2497-2504: "@{id} {CIGAR} E{errors}", bases, "+", qualities): sizes, then the
// records at the offsets of their exclusive scan; one lane per record, word-granular stores
RSQ_HD uint32_t error_model_record_size(const ReadMeta &m, uint32_t id_len) { return 1u + id_len + 1u + m.cigar_chars + 2u + digits_u32(m.num_errors) + 1u + 2u * m.read_len + 4u; }
// the records' ids: packed one after the other (off: n + 1 offsets) or where they stand in the FASTA text (at: offset of the record's '>', len: the id's length)
struct RecordIds {
    const char *chars;
    const uint64_t *off;
    const uint32_t *at, *len;
    RSQ_HD const char *begin(uint64_t i) const { return chars + (off ? off[i] : (uint64_t)at[i] + 1u); }
    RSQ_HD uint32_t length(uint64_t i) const { return off ? (uint32_t)(off[i + 1] - off[i]) : len[i]; }
};
__global__ void __launch_bounds__(256) k_record_text_sizes(RawLayout raw, uint64_t n, RecordIds ids, uint32_t *sizes) {
    const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const uint64_t i = raw.item_of(row);
    sizes[i] = error_model_record_size(raw.meta[row], ids.length(i));
}
// The text by waves, as k_format_write writes the pairs' (rsq_kernels.h): a wave takes 16 consecutive raw rows, four lanes per record (the header and the first
// half of the bases, the second half, the two halves of the qualities), formats them into an LDS image of their contiguous stretch of the output and copies
// the image out in aligned 16-byte stores; PERM (rows binned by tile): a slot of the image per record.  (One lane per record with word-granular stores, the
// kernel of rounds 2-4, wrote 0.3 TB/s: 8.3 ms per 8 M records.)  A wave whose records do not fit the image writes them lane by lane.
template <bool PERM>
__global__ void __launch_bounds__(64) k_record_text_waves(RawLayout raw, uint64_t n, RecordIds ids, const uint64_t *offsets, char *dst, uint64_t cap, uint32_t lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) char s_text[];
    constexpr uint32_t kLineParts = 32u / kFormatRecords;
    const uint32_t lane = threadIdx.x, rec = lane & (kFormatRecords - 1u), part = lane / kFormatRecords, sub = part % kLineParts;
    const bool is_qual = part >= kLineParts;
    const uint64_t first = (uint64_t)blockIdx.x * kFormatRecords;
    if (first >= n || offsets[n] > cap) return;                                      // (the caller's buffer is too small: write nothing, RSQ_ENOSPC)
    const uint64_t last = first + kFormatRecords < n ? first + kFormatRecords : n, row = first + rec;
    const bool active = row < last;
    const uint64_t item = PERM ? (active ? raw.order[row] : 0u) : row;
    const uint64_t g_begin = PERM ? (active ? offsets[item] : 0u) : offsets[first], g_end = PERM ? (active ? offsets[item + 1u] : 0u) : offsets[last];
    const uint32_t skew = (uint32_t)((uint64_t)(uintptr_t)(dst + g_begin) & 15u), bytes = (uint32_t)(g_end - g_begin);
    const uint32_t slot = (lds_bytes / kFormatRecords) & ~15u;
    const bool through_lds = PERM ? __all(skew + bytes <= slot) != 0 : skew + bytes <= lds_bytes;      // wave-uniform
    ReadMeta m{};
    if (active) m = raw.meta[row];
    const WordColumn seq = raw.seq_of(active ? row : 0u), qual = raw.qual_of(active ? row : 0u), ops = raw.ops_of(active ? row : 0u);
    auto header = [&](auto &t) {
        t.ch('@');
        t.str(ids.begin(item), ids.length(item));
        t.ch(' ');
        cigar_replay(ops, m, t);
        t.str(" E", 2);
        t.num((uint32_t)m.num_errors);
        t.ch('\n');
    };
    if (!through_lds) {
        if (active && part == 0u) {
            WordSinkT<char *> t(dst + offsets[item]);
            header(t);
            format_line(seq, m.read_len, false, t);
            format_line(qual, m.read_len, true, t);
            t.finish();
        }
        return;
    }
    const uint32_t slot_at = PERM ? rec * slot : 0u;
    if (active) {
        RSQ_LDS char *rec_text = (RSQ_LDS char *)s_text + slot_at + skew + (PERM ? 0u : (uint32_t)(offsets[item] - g_begin));
        const uint32_t head = (uint32_t)(offsets[item + 1u] - offsets[item]) - 2u * m.read_len - 4u;
        const uint32_t all_words = (m.read_len + 3u) >> 2, per = (all_words + kLineParts - 1u) / kLineParts, first_word = sub * per;
        const uint32_t line_at = head + (is_qual ? m.read_len + 3u : 0u), part_at = part == 0u ? 0u : line_at + (4u * first_word < m.read_len ? 4u * first_word : m.read_len);
        WordSinkT<RSQ_LDS char *> t(rec_text + part_at);
        if (part == 0u) header(t);
        format_line_part(is_qual ? qual : seq, m.read_len, is_qual, first_word, per, sub == kLineParts - 1u, t);
        t.finish();
    }
    __syncthreads();
    const uint32_t lo = skew, hi = skew + bytes;
    char *g_chunk0 = dst + g_begin - skew;                                           // 16-byte aligned
    const char *s_from = s_text + slot_at;
    for (uint32_t c = (PERM ? part : lane) * 16u; c < hi; c += (PERM ? 64u / kFormatRecords : 64u) * 16u) {
        if (c >= lo && c + 16u <= hi) *reinterpret_cast<uint4 *>(g_chunk0 + c) = *reinterpret_cast<const uint4 *>(s_from + c);
        else
            for (uint32_t b = c < lo ? lo : c; b < c + 16u && b < hi; ++b) g_chunk0[b] = s_from[b];
    }
}

// The reads' bases and qualities as rows of out_stride bytes (word-aligned): sixteen lanes copy a row, a word each per step -- 64 bytes of a row in one store
// instruction, where a lane per row wrote four bytes into each of 64 rows (10 ms per 8 M records of 150 bases).  Bytes past read_len inside the last word
// are zero in the raw arrays.
__global__ void __launch_bounds__(256) k_error_model_rows(RawLayout raw, uint64_t n, uint8_t *seq_out, uint8_t *qual_out, uint32_t out_stride) {
    const uint64_t row = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (row >= n) return;
    const uint64_t i = raw.item_of(row);
    const uint32_t read_len = raw.meta[row].read_len, nb = read_len < out_stride ? read_len : out_stride;
    const WordColumn seq = raw.seq_of(row), qual = raw.qual_of(row);
    uint32_t *so = reinterpret_cast<uint32_t *>(seq_out + i * out_stride), *qo = reinterpret_cast<uint32_t *>(qual_out + i * out_stride);
    for (uint32_t w = threadIdx.x & 15u; 4u * w < nb; w += 16u) {
        so[w] = seq.at(w);
        qo[w] = qual.at(w);
    }
}

// CIGAR strings and per-read scalars of the error-model-only mode (and the rows when they are not word-aligned)
__global__ void k_error_model_out(RawLayout raw, uint64_t n, uint8_t *seq_out, uint8_t *qual_out, uint32_t out_stride, uint16_t *read_len_out, uint16_t *num_errors_out,
                                  uint16_t *tile_out, char *cigar_out, uint32_t cigar_stride, uint32_t *overflow) {
    const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const uint64_t i = raw.item_of(row);
    const ReadMeta m = raw.meta[row];
    read_len_out[i] = m.read_len;
    num_errors_out[i] = m.num_errors;
    tile_out[i] = m.tile_id;
    const uint32_t nb = m.read_len < out_stride ? m.read_len : out_stride;
    const WordColumn seq = raw.seq_of(row), qual = raw.qual_of(row);
    if ((out_stride | (uint32_t)(uintptr_t)seq_out | (uint32_t)(uintptr_t)qual_out) & 3u)        // (word-aligned rows are copied by k_error_model_rows)
        for (uint32_t k = 0; k < nb; ++k) {
            seq_out[i * out_stride + k] = (uint8_t)(seq.at(k >> 2) >> (8u * (k & 3u)));
            qual_out[i * out_stride + k] = (uint8_t)(qual.at(k >> 2) >> (8u * (k & 3u)));
        }
    if (m.cigar_chars + 1u > cigar_stride || m.read_len > out_stride) {
        *overflow = 1;
        cigar_out[i * cigar_stride] = 0;
        return;
    }
    TextSink t{cigar_out + i * cigar_stride, 0};
    cigar_replay(raw.ops_of(row), m, t);
    t.ch(0);
}

}  // namespace rsq

// ================================================================================================= C ABI
// (a seqToIllumina record whose fragment length the profile's tables do not hold: check_fragment_lengths)
struct FragmentLengthOutside : rsq::Error {
    using rsq::Error::Error;
};
template <class F>
static int guard(F &&f) {
    try {
        return f();
    } catch (const HipError &e) {
        g_last_error = e.what();
        return RSQ_EHIP;
    } catch (const FragmentLengthOutside &e) {
        g_last_error = e.what();
        return RSQ_EIO;
    } catch (const Error &e) {
        g_last_error = e.what();
        return RSQ_EINVAL;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return RSQ_EINVAL;
    }
}
#define REQUIRE(cond, msg)          \
    do {                            \
        if (!(cond)) {              \
            g_last_error = msg;     \
            return RSQ_EINVAL;      \
        }                           \
    } while (0)

extern "C" {

const char *rsq_last_error(void) { return g_last_error.c_str(); }
const char *rsq_last_warning(void) { return g_last_warning.c_str(); }
const char *rsq_version(void) { return "reseq_amd 0.1 (gfx950)"; }
int rsq_set_option(const char *name, int64_t value) {
    REQUIRE(name, "null argument");
    if (set_option(name, value)) return RSQ_OK;
    g_last_error = std::string("unknown option '") + name + "' (options: " + option_names() + ")";
    return RSQ_EINVAL;
}
int rsq_get_option(const char *name, int64_t *value) {
    REQUIRE(name && value, "null argument");
    if (get_option(name, value)) return RSQ_OK;
    g_last_error = std::string("unknown option '") + name + "' (options: " + option_names() + ")";
    return RSQ_EINVAL;
}

int rsq_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        g_last_error = "no HIP device visible; this library has no CPU fallback";
        return RSQ_ENODEV;
    }
    return n;
}

// Which blocks a worker simulates (Simulator.cpp:2384-2401 hands blocks to threads one by one from a counter; here every worker owns a contiguous range, so that the
// workers' texts in worker order are the single run's): ranges balanced by expected pairs per block.  The one rule for the `reseq` command line's device threads and
// the N-process launcher (reseq_amd/sharding.py partition_blocks states the same arithmetic in the same order; tests/test_abi.py compares them).
int rsq_partition_blocks(uint32_t total_blocks, uint32_t workers, const double *weights, uint32_t *bounds) {
    REQUIRE(workers >= 1 && bounds && (weights || !total_blocks), "null argument or no worker");
    double total = 0.0;
    for (uint32_t b = 0; b < total_blocks; ++b) total += weights[b];
    bounds[0] = 1;
    double acc = 0.0;
    uint32_t b = 0;
    for (uint32_t r = 1; r < workers; ++r) {
        const double target = total * (double)r / (double)workers;
        while (b < total_blocks && acc + weights[b] / 2 <= target) acc += weights[b++];
        bounds[r] = b + 1u;
    }
    bounds[workers] = total_blocks + 1u;
    return RSQ_OK;
}
int rsq_sim_block_weights(const rsq_sim *s, double *weights, size_t cap, uint32_t *n_blocks) {
    REQUIRE(s && n_blocks && (s->prepared || s->planned) && s->has_ref, "a simulator with a reference after rsq_sim_prepare or rsq_sim_prepare_plan");
    *n_blocks = s->total_blocks;
    if (!weights) return RSQ_OK;
    if (cap < s->total_blocks) {
        g_last_error = "room for " + std::to_string(cap) + " weights, the job has " + std::to_string(s->total_blocks) + " blocks";
        return RSQ_ENOSPC;
    }
    size_t at = 0;
    for (uint32_t i = 0; i < s->dev.n_seqs; ++i)                   // a sequence's blocks share its reference bias; sequences shorter than the longest insert have none (:1159)
        for (uint32_t k = 0; k < s->n_blocks[i]; ++k) weights[at++] = s->ref_seq_bias[i];
    REQUIRE(at == s->total_blocks, "block counts do not add up");
    return RSQ_OK;
}

int rsq_profile_load_reseq(const char *stats_path, const char *ipf_path, double ipf_precision_percent, rsq_profile **out) {
    REQUIRE(stats_path && out, "null argument");
    REQUIRE(ipf_precision_percent > 0.0, "ipfPrecision must be positive.");      // main.cpp:776-779
    try {
        g_last_warning.clear();
        *out = new rsq_profile{Profile::load_archives(stats_path, ipf_path ? ipf_path : "", ipf_precision_percent / 100.0, &g_last_warning)};
        return RSQ_OK;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return RSQ_EIO;
    }
}
int rsq_profile_is_reseq_archive(const char *path, int *yes) {
    REQUIRE(path && yes, "null argument");
    *yes = Profile::is_archive(path) ? 1 : 0;
    return RSQ_OK;
}
int rsq_profile_load(const char *path, rsq_profile **out) {
    REQUIRE(path && out, "null argument");
    try {
        if (Profile::is_archive(path)) return rsq_profile_load_reseq(path, nullptr, 5.0, out);
        *out = new rsq_profile{Profile::load(path)};
        return RSQ_OK;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return RSQ_EIO;
    }
}
int rsq_profile_archive_layout(const char *stats_path, const char *ipf_path, char *out, size_t cap, size_t *need) {
    REQUIRE(stats_path && need && (out || !cap), "null argument");
    return guard([&] {
        const std::string text = Profile::archive_layout(stats_path, ipf_path ? ipf_path : "");
        *need = text.size() + 1;
        if (cap) {
            const size_t n = std::min(cap - 1, text.size());
            memcpy(out, text.data(), n);
            out[n] = 0;
        }
        return (int)RSQ_OK;
    });
}
int rsq_profile_save(const rsq_profile *p, const char *path) {
    REQUIRE(p && path, "null argument");
    try {
        p->p.save(path);
        return RSQ_OK;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return RSQ_EIO;
    }
}
int rsq_profile_save_reseq(const rsq_profile *p, const char *stats_path, const char *ipf_path, uint64_t creation_time) {
    REQUIRE(p && stats_path, "null argument");
    try {
        p->p.save_archives(stats_path, ipf_path ? ipf_path : "", creation_time ? creation_time : (uint64_t)time(nullptr));
        return RSQ_OK;
    } catch (const Error &e) {
        g_last_error = e.what();
        return RSQ_EINVAL;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return RSQ_EIO;
    }
}
void rsq_profile_free(rsq_profile *p) { delete p; }
int rsq_profile_change_error_rate(rsq_profile *p, double m) {
    REQUIRE(p && m > 0.0, "bad error multiplier");
    p->p.change_error_rate(m);
    return RSQ_OK;
}
int rsq_profile_remove_substitution_errors(rsq_profile *p) {
    REQUIRE(p, "null profile");
    p->p.remove_substitution_errors();
    return RSQ_OK;
}
int rsq_profile_remove_indel_errors(rsq_profile *p) {
    REQUIRE(p, "null profile");
    p->p.remove_indel_errors();
    return RSQ_OK;
}
int rsq_profile_max_read_length(const rsq_profile *p, uint32_t *out) {
    REQUIRE(p && out, "null argument");
    *out = (uint32_t)std::max(p->p.read_lengths[0].to(), p->p.read_lengths[1].to()) - 1;
    return RSQ_OK;
}
int rsq_profile_num_tiles(const rsq_profile *p, uint32_t *out) {
    REQUIRE(p && out, "null argument");
    *out = p->p.n_tiles();
    return RSQ_OK;
}

int rsq_profile_max_len_deletion(const rsq_profile *p, uint32_t *out) {
    REQUIRE(p && out, "null argument");
    *out = p->p.max_len_deletion;
    return RSQ_OK;
}
int rsq_profile_ref_seq_bias(const rsq_profile *p, double *out, size_t cap, size_t *n) {
    REQUIRE(p && n, "null argument");
    *n = p->p.ref_seq_bias.size();
    if (!out) return RSQ_OK;
    if (cap < *n) return RSQ_ENOSPC;
    memcpy(out, p->p.ref_seq_bias.data(), *n * sizeof(double));
    return RSQ_OK;
}
int rsq_ref_sequence_name(const rsq_ref *r, uint32_t seq, char *out, size_t cap) {
    REQUIRE(r && out && seq < r->r.codes.size(), "bad sequence id");
    const std::string name = r->r.first_part(seq);
    if (name.size() + 1 > cap) return RSQ_ENOSPC;
    memcpy(out, name.c_str(), name.size() + 1);
    return RSQ_OK;
}

int rsq_ref_load_fasta(const char *path, rsq_ref **out) {
    REQUIRE(path && out, "null argument");
    try {
        *out = new rsq_ref{Reference::read_fasta(path)};
        return RSQ_OK;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return RSQ_EIO;
    }
}
int rsq_ref_replace_n(rsq_ref *r, uint64_t seed) {
    REQUIRE(r, "null reference");
    return guard([&] {
        r->r.replace_n(seed);
        return RSQ_OK;
    });
}
void rsq_ref_free(rsq_ref *r) { delete r; }
int rsq_ref_num_sequences(const rsq_ref *r, uint32_t *out) {
    REQUIRE(r && out, "null argument");
    *out = (uint32_t)r->r.codes.size();
    return RSQ_OK;
}
int rsq_ref_sequence_length(const rsq_ref *r, uint32_t seq, uint32_t *out) {
    REQUIRE(r && out && seq < r->r.codes.size(), "bad sequence id");
    *out = (uint32_t)r->r.codes[seq].size();
    return RSQ_OK;
}
int rsq_ref_read_variants(rsq_ref *r, const char *path) {
    REQUIRE(r && path, "null argument");
    try {
        std::vector<std::string> first;
        for (size_t i = 0; i < r->r.codes.size(); ++i) first.push_back(r->r.first_part(i));
        r->variants = read_variants(path, first, r->r.codes);
        r->has_variants = true;
        return RSQ_OK;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return RSQ_EIO;
    }
}
int rsq_ref_num_alleles(const rsq_ref *r, uint32_t *out) {
    REQUIRE(r && out, "null argument");
    *out = r->variants.num_alleles;
    return RSQ_OK;
}
int rsq_ref_num_variants(const rsq_ref *r, uint32_t seq, uint32_t *out) {
    REQUIRE(r && out && r->has_variants && seq < r->variants.by_seq.size(), "no variants loaded or bad sequence id");
    *out = (uint32_t)r->variants.by_seq[seq].size();
    return RSQ_OK;
}
int rsq_ref_get_variant(const rsq_ref *r, uint32_t seq, uint32_t index, uint32_t *position, char *var_seq, size_t var_seq_cap, uint64_t allele_bits[2]) {
    REQUIRE(r && position && var_seq && allele_bits && r->has_variants && seq < r->variants.by_seq.size() && index < r->variants.by_seq[seq].size(), "bad variant index");
    const Variant &v = r->variants.by_seq[seq][index];
    REQUIRE(v.var_seq.size() + 1 <= var_seq_cap, "var_seq buffer too small");
    *position = v.position;
    for (size_t k = 0; k < v.var_seq.size(); ++k) var_seq[k] = "ACGTN"[v.var_seq[k]];
    var_seq[v.var_seq.size()] = 0;
    allele_bits[0] = v.allele[0];
    allele_bits[1] = v.allele[1];
    return RSQ_OK;
}
int rsq_ref_write_fasta(const rsq_ref *r, const char *path) {
    REQUIRE(r && path, "null argument");
    try {
        std::string text;
        for (size_t i = 0; i < r->r.codes.size(); ++i) {
            text += '>';
            text += r->r.names[i];
            text += '\n';
            const std::vector<uint8_t> &c = r->r.codes[i];
            for (size_t k = 0; k < c.size(); k += 70) {               // SeqAn's default FASTA line length
                for (size_t j = k; j < std::min(c.size(), k + 70); ++j) text += "ACGTN"[c[j] < 4 ? c[j] : 4];
                text += '\n';
            }
        }
        write_text_file(path, text);
        return RSQ_OK;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return RSQ_EIO;
    }
}
int rsq_ref_get_codes(const rsq_ref *r, uint32_t seq, uint8_t *out, uint32_t len) {
    REQUIRE(r && out && seq < r->r.codes.size() && len == r->r.codes[seq].size(), "bad sequence id or length");
    memcpy(out, r->r.codes[seq].data(), len);
    return RSQ_OK;
}

int rsq_sim_create(const rsq_profile *p, const rsq_ref *ref, int device, rsq_sim **out) {
    REQUIRE(p && out, "null argument");
    int n = rsq_device_count();
    if (n < 0) return n;
    REQUIRE(device >= 0 && device < n, "device index out of range");
    std::unique_ptr<rsq_sim> s(new rsq_sim());
    int rc = guard([&] {
        HIP_CHECK(hipSetDevice(device));
        s->device = device;
        s->up.device = device;
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        s->n_cu = (uint32_t)prop.multiProcessorCount;
        s->arch = prop.gcnArchName;
        s->specialize = s->opt.specialize != 0;
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&s->mailbox), 16 * sizeof(uint64_t), hipHostMallocDefault));
        for (hipStream_t &st : s->side) HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        for (rsq_sim::Workspace &w : s->ws) HIP_CHECK(hipEventCreateWithFlags(&w.text_done, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&s->ev_call, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&s->ev_fill, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&s->ev_emit, hipEventDisableTiming));
        s->force_fill_mode = (int)s->opt.fill_mode;
        s->prof = p->p;
        pack_tables(*s, s->up);
        g_last_warning = s->plan_note;
        pack_profile(*s, s->up);
        if (ref) pack_reference(*s, s->up, ref->r, ref->has_variants ? &ref->variants : nullptr);
        return RSQ_OK;
    });
    if (rc == RSQ_OK) *out = s.release();
    return rc;
}
// the switches again, as they stand now (a simulator otherwise keeps the copy it took when it was created): for a caller that changes a call-shaping switch --
// overlap, job_chunk_bytes, job_write_direct, host_gzip, fill_waves, the pre-pass switches -- between the calls of ONE simulator.  What was decided when the
// simulator was created (fill_mode, image_tiles, rate_rows, min_quality_quads, specialize: the packed tables and the compiled kernel) stays.
int rsq_sim_take_options(rsq_sim *s) {
    REQUIRE(s, "null argument");
    s->opt = options();
    return RSQ_OK;
}
void rsq_sim_free(rsq_sim *s) {
    if (s) {
        (void)hipSetDevice(s->device);
        if (s->mailbox) (void)hipHostFree(s->mailbox);
        for (hipStream_t st : s->side)
            if (st) (void)hipStreamDestroy(st);
        for (rsq_sim::Workspace &w : s->ws)
            if (w.text_done) (void)hipEventDestroy(w.text_done);
        for (hipEvent_t e : {s->ev_call, s->ev_fill, s->ev_emit})
            if (e) (void)hipEventDestroy(e);
    }
    delete s;
}

int rsq_sim_prepare(rsq_sim *s, uint64_t seed, uint64_t num_read_pairs, double coverage, int ref_bias_mode, const char *record_base_identifier, void *stream) {
    REQUIRE(s, "null simulator");
    return guard([&] {
        prepare(*s, seed, num_read_pairs, coverage, ref_bias_mode, record_base_identifier, (hipStream_t)stream);
        return RSQ_OK;
    });
}

int rsq_sim_prepare_plan(rsq_sim *s, uint64_t seed, uint64_t num_read_pairs, double coverage, int ref_bias_mode, const char *record_base_identifier) {
    REQUIRE(s, "null argument");
    REQUIRE(s->has_ref, "the sharded pre-pass needs a reference");
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        s->prepared = false;
        s->normalized = false;
        s->chain_run.valid = false;
        plan_simulation(*s, s->up, seed, num_read_pairs, coverage, ref_bias_mode, record_base_identifier);
        s->bias_plan = plan_bias_normalization(*s, s->up);
        s->planned = true;
        return RSQ_OK;
    });
}
int rsq_sim_bias_partials(rsq_sim *s, uint32_t block_lo, uint32_t block_hi, double *sums, double *maxes, size_t cap, size_t *n, void *stream) {
    REQUIRE(s && n, "null argument");
    REQUIRE(s->planned, "rsq_sim_prepare_plan must run first");
    *n = (size_t)bias_chunks(s->bias_plan);
    if (!sums && !maxes && !cap) return RSQ_OK;                     // the size query
    REQUIRE(sums && maxes && cap >= *n, "arrays too small");
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        const ShardRange r = shard_range(*s, block_lo, block_hi);
        std::vector<double> h_sum, h_max;
        if (r.g_lo < r.g_hi) bias_partials(*s, (hipStream_t)stream, s->bias_plan, r.g_lo, r.g_hi, h_sum, h_max);
        else {
            h_sum.assign(*n, 0.0);
            h_max.assign(*n, 0.0);
        }
        memcpy(sums, h_sum.data(), *n * 8);
        memcpy(maxes, h_max.data(), *n * 8);
        return RSQ_OK;
    });
}
int rsq_sim_prepare_normalization(rsq_sim *s, const double *sums, const double *maxes, size_t n) {
    REQUIRE(s && sums && maxes, "null argument");
    REQUIRE(s->planned, "rsq_sim_prepare_plan must run first");
    REQUIRE(n == (size_t)bias_chunks(s->bias_plan), "wrong number of partial sums");
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        normalization_from_partials(*s, s->up, s->bias_plan, sums, maxes);
        s->normalized = true;
        return RSQ_OK;
    });
}
int rsq_sim_prepare_sys_errors(rsq_sim *s, uint32_t block_lo, uint32_t block_hi, const uint32_t in_state[2], uint32_t out_state[2], void *stream) {
    REQUIRE(s && in_state && out_state, "null argument");
    REQUIRE(s->planned, "rsq_sim_prepare_plan must run first");
    return guard([&] {
        hipStream_t st = (hipStream_t)stream;
        HIP_CHECK(hipSetDevice(s->device));
        out_state[0] = in_state[0];                                 // a rank without blocks passes the states on
        out_state[1] = in_state[1];
        rsq_sim::ChainRun &run = s->chain_run;
        if (!(run.valid && run.block_lo == block_lo && run.block_hi == block_hi)) {
            const ShardRange r = shard_range(*s, block_lo, block_hi);
            s->passes = run_sys_chains(*s, st, kChainsSimulation, &r);
            run.block_lo = block_lo;
            run.block_hi = block_hi;
            run.pass_through = r.first_seq < 0;
            run.valid = true;
        }
        if (run.pass_through || !run.n_chunks) return RSQ_OK;
        bool replaced = false;
        const int in_chain[2] = {run.edges.fwd_in_chain, run.edges.rev_in_chain};
        for (int k = 0; k < 2; ++k)
            if (in_chain[k] >= 0 && run.chains[(size_t)in_chain[k]].in_state != in_state[k]) {
                run.chains[(size_t)in_chain[k]].in_state = in_state[k];
                replaced = true;
            }
        if (replaced) {
            run.d_chains.upload(run.chains);
            iterate_sys_chains(*s, run, st, run.passes);            // only the chunks behind a changed state run again
            s->passes = run.passes;
        }
        const int64_t out_chunk[2] = {run.edges.fwd_out_chunk, run.edges.rev_out_chunk};
        for (int k = 0; k < 2; ++k) {
            out_state[k] = 0;
            if (out_chunk[k] >= 0)
                HIP_CHECK(hipMemcpy(&out_state[k], run.d_out[(run.passes - 1) & 1].as<uint32_t>() + out_chunk[k], 4, hipMemcpyDeviceToHost));
        }
        return RSQ_OK;
    });
}
int rsq_sim_prepare_finish(rsq_sim *s) {
    REQUIRE(s, "null argument");
    REQUIRE(s->planned && s->chain_run.valid, "the sharded pre-pass has not run");
    REQUIRE(s->normalized, "rsq_sim_prepare_normalization must run before rsq_sim_prepare_finish (the thresholds of the sieve come from it)");
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        s->prepared_lo = s->chain_run.block_lo;                     // only these blocks' tracks are finished
        s->prepared_hi = s->chain_run.block_hi;
        if (s->has_variants) variant_sys_errors_from_run(*s, nullptr);      // -V: the variants' bases inside the rank's strand windows, from the finished chains
        s->prepared = true;
        (void)spec_kernel(*s, SpecKind::kReads, effective_fill_mask(s->dev.lds.mask, s->force_fill_mode), s->has_variants, fill_is_binned(*s));
        return RSQ_OK;
    });
}

int rsq_sim_get_sequence_lengths(const rsq_sim *s, uint32_t *out, size_t cap, uint32_t *n_sequences) {
    REQUIRE(s && n_sequences && s->has_ref, "null argument, or a simulator without a reference");
    *n_sequences = (uint32_t)s->seq_len.size();
    if (out && cap >= s->seq_len.size()) memcpy(out, s->seq_len.data(), s->seq_len.size() * sizeof(uint32_t));
    else if (out) {
        g_last_error = "room for " + std::to_string(cap) + " lengths, the reference has " + std::to_string(s->seq_len.size()) + " sequences";
        return RSQ_ENOSPC;
    }
    return RSQ_OK;
}
// Reference::ReferenceSequence (Reference.cpp:483-567) for one call, the way the kernels get a template: allele_template on the allele's coordinate map (variants of
// any kind), else FragmentSrc on the allele's copy of the packed reference.  One thread; flag 1: the stretch leaves the allele's sequence.
__global__ void k_reference_sequence(DevSim S, uint32_t seq, uint32_t allele, uint32_t pos, VarStart from, uint32_t n, bool reversed, uint64_t *tmpl, uint32_t words, uint32_t *flag) {
    if (threadIdx.x || blockIdx.x) return;
    if (2u == S.variants_loaded) {
        const AlleleView a = allele_view(S, seq, allele);
        const int64_t h = from.start_variant_pos ? a.begin_of(a.entries_before(a.r.v[from.first_variant_id].pos)) + from.start_variant_pos : a.to_allele(pos);
        if (reversed ? h < (int64_t)n : h + (int64_t)n > a.length()) {
            *flag = 1u;
            return;
        }
        allele_template(a, pos, from, n, reversed, tmpl, words);
        return;
    }
    FragmentSrc src;
    src.words = hap_words(S, allele);
    src.word_off = S.seq_word_off[seq];
    src.first = pos;
    src.len = n;
    src.reverse = reversed;
    src.sys_ = nullptr;
    src.converted = nullptr;
    src.gc_prefix = nullptr;
    for (uint32_t w = 0; w < words; ++w) tmpl[w] = 0;
    for (uint32_t k = 0; k < n; ++k) tmpl[k >> 5] |= (uint64_t)src.ref(k) << ((k & 31u) * 2u);
}
int rsq_sim_reference_sequence(rsq_sim *s, uint32_t seq, uint32_t start_pos, uint32_t frag_length, int reversed, int32_t first_variant_id, uint32_t first_variant_pos, uint32_t allele,
                               uint8_t *out, size_t cap) {
    REQUIRE(s && out && s->has_ref && seq < s->dev.n_seqs, "null argument, a simulator without a reference, or no such sequence");
    REQUIRE(allele < (s->has_variants ? s->num_alleles : 1u), "no such allele");
    REQUIRE(frag_length <= (1u << 20) && cap >= frag_length, "room for fewer bases than asked for (at most 2^20 per call)");
    const uint32_t L = s->seq_len[seq];
    REQUIRE(reversed ? start_pos <= L : start_pos < L || 0 == frag_length, "start position outside the sequence");
    if (2 != s->variants_mode)                                      // alleles as long as the reference (with coordinate maps the kernel checks against the allele's length)
        REQUIRE(reversed ? frag_length <= start_pos : frag_length <= L - start_pos, "the stretch asked for leaves the sequence");
    if (first_variant_pos) {
        REQUIRE(2 == s->variants_mode, "a start inside inserted bases needs variants with insertions");
        const uint32_t n_var = s->var_ptr[seq + 1] - s->var_ptr[seq];
        REQUIRE(first_variant_id >= 0 && (uint32_t)first_variant_id < n_var && first_variant_pos <= s->variants[s->var_ptr[seq] + first_variant_id].len,
                "first_variant does not name bases of a variant of this sequence");
    }
    if (!frag_length) return RSQ_OK;
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        const uint32_t words = (frag_length + 31u) / 32u;
        uint64_t *dev = nullptr;
        HIP_CHECK(hipMalloc(&dev, (size_t)(words + 1u) * sizeof(uint64_t)));
        std::vector<uint64_t> host(words + 1u, 0);
        hipError_t e = hipMemset(dev, 0, (size_t)(words + 1u) * sizeof(uint64_t));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_reference_sequence, dim3(1), dim3(64), 0, nullptr, s->dev, seq, allele, start_pos, VarStart{first_variant_id, first_variant_pos}, frag_length, reversed != 0,
                               dev, words, reinterpret_cast<uint32_t *>(dev + words));
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpy(host.data(), dev, host.size() * sizeof(uint64_t), hipMemcpyDeviceToHost);
        (void)hipFree(dev);
        HIP_CHECK(e);
        if (host[words]) {
            g_last_error = "the stretch asked for leaves the allele's sequence";
            return RSQ_EINVAL;
        }
        for (uint32_t k = 0; k < frag_length; ++k) out[k] = (uint8_t)((host[k >> 5] >> ((k & 31u) * 2u)) & 3u);
        return RSQ_OK;
    });
}
int rsq_sim_export_reference(rsq_sim *s, const char *path) {
    REQUIRE(s && path, "null argument");
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        HIP_CHECK(hipDeviceSynchronize());
        export_reference(*s, s->up, path);
        return RSQ_OK;
    });
}
int rsq_sim_import_reference(rsq_sim *s, const char *path) {
    REQUIRE(s && path, "null argument");
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        MappedFile m(path);
        import_reference(*s, s->up, m.data, m.size, path);
        return RSQ_OK;
    });
}
int rsq_sim_specialize(rsq_sim *s, int kind, int *specialized) {
    REQUIRE(s && specialized && (kind == 0 || kind == 1), "null argument, or a kind that is neither 0 (read pairs) nor 1 (seqToIllumina records)");
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        const uint32_t mask = effective_fill_mask(s->dev.lds.mask, s->force_fill_mode);
        g_spec_note.clear();
        hipFunction_t fn = spec_kernel(*s, kind == 0 ? SpecKind::kReads : SpecKind::kRecords, mask, kind == 0 && s->has_variants, fill_is_binned(*s));
        if (kind == 1 && fn) fn = spec_kernel(*s, SpecKind::kRecords, mask, true, fill_is_binned(*s));      // and the kernel for records parsed on the device (packed codes)
        *specialized = fn != nullptr;
        if (!s->specialize) g_spec_note = "option specialize is 0: the library's own instantiation of the read kernel runs";
        else if (!mask) g_spec_note = "the read kernel draws in double precision from device memory (no table image): nothing to compile for the profile";
        g_last_warning = g_spec_note;
        return RSQ_OK;
    });
}
// host only: the plan of the profile packed into host memory (nothing is uploaded), then the compilation rsq_sim_specialize would do
int rsq_profile_compile_read_kernel(const rsq_profile *p, int kind, int with_variants, int binned, const char *arch, const char *out_path, size_t *code_bytes, double *seconds) {
    REQUIRE(p && arch && code_bytes && (kind == 0 || kind == 1), "null argument, or a kind that is neither 0 nor 1");
    struct HostArrays : Uploader {
        std::vector<std::unique_ptr<char[]>> owned;
        void *put_bytes(const void *data, size_t bytes) override {
            owned.emplace_back(new char[bytes + 8]);
            memcpy(owned.back().get(), data, bytes);
            return owned.back().get();
        }
        void *put_zeros(size_t bytes) override {
            owned.emplace_back(new char[bytes + 8]());
            return owned.back().get();
        }
        void write_bytes(void *dst, const void *src, size_t bytes) override { memcpy(dst, src, bytes); }
        void read_bytes(void *dst, const void *src, size_t bytes) override { memcpy(dst, src, bytes); }
    };
    int rc = guard([&] {
        HostArrays up;
        SimState s;
        s.prof = p->p;
        pack_tables(s, up);
        pack_profile(s, up);
        if (!s.dev.lds.mask) throw Error("the profile has no table image (" + s.plan_note + "): there is nothing to compile for it");
        const SpecVariant variant{kind == 0 ? SpecKind::kReads : SpecKind::kRecords, s.dev.lds.mask, kind == 0 && with_variants != 0, binned != 0 || s.dev.lds.binned != 0};
        const std::string path = out_path ? out_path : "";
        if (path.size() > 4 && path.compare(path.size() - 4, 4, ".hip") == 0) {        // the program itself (for `hipcc -I reseq_amd/csrc -g ...`: listings with source lines)
            const std::string program = spec_program(spec_literals(s.dev), variant);
            FILE *f = fopen(out_path, "wb");
            if (!f || fwrite(program.data(), 1, program.size(), f) != program.size()) throw Error(std::string("cannot write ") + out_path);
            fclose(f);
            *code_bytes = program.size();
            if (seconds) *seconds = 0.0;
            return RSQ_OK;
        }
        SpecCode c;
        std::string note;
        if (!spec_compile(s.dev, variant, arch, c, note)) throw Error(note);
        *code_bytes = c.code.size();
        if (seconds) *seconds = c.seconds;
        if (out_path && *out_path) {
            FILE *f = fopen(out_path, "wb");
            if (!f || fwrite(c.code.data(), 1, c.code.size(), f) != c.code.size()) throw Error(std::string("cannot write ") + out_path);
            fclose(f);
        }
        return RSQ_OK;
    });
    return rc;
}
int rsq_set_kernel_cache_dir(const char *path) {
    spec_cache_dir_override() = path ? path : "";
    spec_cache_dir_set() = true;
    return RSQ_OK;
}
int rsq_sim_get_fill_plan(const rsq_sim *s, uint32_t *quality_quads, uint32_t *image_tiles, uint32_t *image_bytes) {
    REQUIRE(s && quality_quads && image_tiles && image_bytes, "null argument");
    *quality_quads = effective_fill_mask(s->dev.lds.mask, s->force_fill_mode);
    *image_tiles = *quality_quads ? s->dev.lds.img_tiles : 0u;
    *image_bytes = *quality_quads ? s->dev.lds.total_words * 4u : 0u;
    return RSQ_OK;
}
int rsq_sim_get_info(const rsq_sim *s, rsq_sim_info *out) {
    REQUIRE(s && out && (s->prepared || s->planned), "simulator not prepared");      // the counts are known from rsq_sim_prepare_plan on
    out->total_pairs = s->total_pairs;
    out->adapter_only_pairs = s->adapter_only_pairs;
    out->total_blocks = s->total_blocks;
    out->n_coverage_groups = s->n_groups;
    out->insert_to = s->dev.insert_to;
    out->sys_chain_passes = s->passes;
    out->bias_normalization = s->bias_normalization;
    return RSQ_OK;
}
int rsq_sim_get_thresholds(const rsq_sim *s, double *out, size_t n) {
    REQUIRE(s && out && s->prepared && n == s->thresholds.size(), "bad threshold buffer size");
    memcpy(out, s->thresholds.data(), n * 8);
    return RSQ_OK;
}
int rsq_sim_get_norm_by_len(const rsq_sim *s, double *out, size_t n) {
    REQUIRE(s && out && s->prepared && n == s->norm_by_len.size(), "bad buffer size");
    memcpy(out, s->norm_by_len.data(), n * 8);
    return RSQ_OK;
}
int rsq_sim_set_normalization(rsq_sim *s, double bias_normalization, const double *thresholds, size_t n) {
    REQUIRE(s && thresholds && s->prepared && n == s->thresholds.size(), "bad threshold buffer size");
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        s->bias_normalization = bias_normalization;
        s->thresholds.assign(thresholds, thresholds + n);
        upload_normalization(*s, s->up);
        return RSQ_OK;
    });
}
static int download_sys(const rsq_sim *s, const uint16_t *src, uint8_t *dom_out, uint8_t *rate_out, uint32_t len) {
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        std::vector<uint16_t> tmp(len);
        HIP_CHECK(hipMemcpy(tmp.data(), src, (size_t)len * 2, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < len; ++i) {
            dom_out[i] = (uint8_t)(tmp[i] & 0xFF);
            rate_out[i] = (uint8_t)(tmp[i] >> 8);
        }
        return RSQ_OK;
    });
}
int rsq_sim_get_sys_errors(const rsq_sim *s, int reverse_strand, uint32_t seq, uint8_t *dom_out, uint8_t *rate_out, uint32_t len) {
    REQUIRE(s && s->prepared && s->has_ref && seq < s->dev.n_seqs && len == s->seq_len[seq] && dom_out && rate_out, "bad arguments");
    REQUIRE(s->n_blocks[seq], "sequence is shorter than the longest insert length and is not simulated");
    REQUIRE(s->first_block[seq] >= s->prepared_lo && s->first_block[seq] + s->n_blocks[seq] <= s->prepared_hi,
            "the sequence's systematic errors were not (all) drawn on this simulator: the sharded pre-pass finishes the tracks of the rank's own blocks only");
    return download_sys(s, (reverse_strand ? s->sys_rev : s->sys_fwd) + s->seq_base_off[seq], dom_out, rate_out, len);
}
int rsq_sim_create_sys_error_profile(rsq_sim *s, uint64_t seed, const char *path, void *stream) {
    REQUIRE(s && path, "null argument");
    return guard([&] {
        create_sys_error_profile(*s, seed, path, (hipStream_t)stream);
        return RSQ_OK;
    });
}
int rsq_sim_read_sys_errors(rsq_sim *s, const char *path) {
    REQUIRE(s && path && s->prepared && s->has_ref, "rsq_sim_prepare with a reference must run before rsq_sim_read_sys_errors");
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        apply_sys_error_records(*s, s->up, parse_sys_error_fastq(read_text_file(path)));
        build_variant_sys_errors(*s, s->up);                        // their error-region state follows the loaded rates
        s->prepared_lo = 1;                                         // the file holds the tracks of every sequence
        s->prepared_hi = s->total_blocks + 1;
        return RSQ_OK;
    });
}
int rsq_sim_read_methylation(rsq_sim *s, const char *path) {
    REQUIRE(s && path && s->has_ref, "a simulator with a reference is needed");
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        pack_methylation(*s, s->up, read_methylation_file(path, s->ref_first_names, s->seq_len, s->num_alleles));
        return RSQ_OK;
    });
}
int rsq_sim_set_ref_bias_file(rsq_sim *s, const char *path) {
    REQUIRE(s && path, "null argument");
    s->ref_bias_file = path;
    return RSQ_OK;
}
int rsq_sim_get_ref_seq_bias(const rsq_sim *s, double *out, size_t n) {
    REQUIRE(s && out && (s->prepared || s->planned) && n == s->ref_seq_bias.size(), "bad arguments");
    memcpy(out, s->ref_seq_bias.data(), n * sizeof(double));
    return RSQ_OK;
}
int rsq_sim_get_adapter_sys_errors(const rsq_sim *s, int seg, uint32_t adapter, uint8_t *dom_out, uint8_t *rate_out, uint32_t len) {
    REQUIRE(s && s->prepared && (seg == 0 || seg == 1) && adapter < s->prof.adapters[seg].n() && dom_out && rate_out, "bad arguments");
    const HostAdapters &a = s->prof.adapters[seg];
    REQUIRE(len == a.seq_ptr[adapter + 1] - a.seq_ptr[adapter], "bad adapter length");
    return download_sys(s, s->adapter_sys[seg] + a.seq_ptr[adapter], dom_out, rate_out, len);
}

int rsq_sim_pairs(rsq_sim *s, uint32_t block_lo, uint32_t block_hi, char *r1_dev, size_t r1_cap, size_t *r1_len, char *r2_dev, size_t r2_cap, size_t *r2_len,
                  uint64_t *n_pairs, rsq_fragment *frags_dev, size_t frags_cap, void *stream) {
    REQUIRE(s && r1_len && r2_len && n_pairs, "null argument");
    return guard([&] { return sim_pairs(*s, block_lo, block_hi, r1_dev, r1_cap, r1_len, r2_dev, r2_cap, r2_len, n_pairs, frags_dev, frags_cap, (hipStream_t)stream); });
}

// ---- a rank's share, generated once and kept (include/reseq_amd.h rsq_sim_job_*)
constexpr size_t kJobChunkBytes = (size_t)2 << 30, kJobSliceBytes = (size_t)32 << 20;
int rsq_sim_job_generate(rsq_sim *s, uint32_t block_lo, uint32_t block_hi, uint32_t batch_blocks, uint64_t *n_pairs, uint64_t *r1_bytes, uint64_t *r2_bytes, void *stream) {
    REQUIRE(s && n_pairs && r1_bytes && r2_bytes, "null argument");
    *n_pairs = *r1_bytes = *r2_bytes = 0;
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        rsq_sim::JobText &job = s->job;
        job.clear();
        if (const int rc = check_block_range(*s, block_lo, block_hi)) return rc;      // before anything is sized from the range (block_hi - block_lo is unsigned)
        const size_t chunk_bytes = s->opt.job_chunk_bytes > 0 ? (size_t)s->opt.job_chunk_bytes : kJobChunkBytes;
        // about 12 M pairs per call, at least 2000 blocks: a call is a round of launches with a tail behind each, and a longer call shares it among more pairs --
        // the Drosophila-sized job runs at 154 M pairs/s in calls of 2.4 M pairs and at 179 M in one call of 14.5 M, the human-sized one at a tenth of its size
        // at 119 and 139 M (profiles/r05_i_other_configs.json); 12 M pairs take about 20 GB of the 288 for workspace and text
        if (!batch_blocks)
            batch_blocks = (uint32_t)std::min(400000.0, std::max(2000.0, 12e6 * (double)s->total_blocks / (double)std::max<uint64_t>(1, s->total_pairs)));
        // bytes per block a call is given room for: the largest seen so far; before the first call 400 bytes per read (2 x 150 characters, an id of 60 to 90) at the
        // job's pair density
        const double prior = 400.0 * (double)s->total_pairs / (double)std::max<uint32_t>(1, s->total_blocks);
        double per_block[2] = {prior, prior};
        auto room = [&](int f) { return job.chunks[f].empty() ? (size_t)0 : job.chunks[f].back()->bytes() - job.used[f].back(); };
        // arrays: the whole range in one when memory allows (option job_chunk_bytes unset), else pieces of 2 GiB
        size_t free_bytes = 0, total_bytes = 0;
        HIP_CHECK(hipMemGetInfo(&free_bytes, &total_bytes));
        const size_t whole = (size_t)(prior * (double)(block_hi - block_lo) * 1.1) + ((size_t)64 << 20);
        const size_t first_chunk = s->opt.job_chunk_bytes > 0 ? chunk_bytes : (2 * whole + ((size_t)8 << 30) < free_bytes ? std::max(whole, chunk_bytes) : chunk_bytes);
        auto new_chunk = [&](int f, size_t at_least) {
            job.chunks[f].emplace_back(new DevBuf());
            job.chunks[f].back()->reserve(std::max(job.chunks[f].size() == 1 ? first_chunk : chunk_bytes, at_least));
            job.used[f].push_back(0);
        };
        for (uint32_t lo = block_lo; lo < block_hi; lo += batch_blocks) {
            const uint32_t hi = (uint32_t)std::min<uint64_t>(block_hi, (uint64_t)lo + batch_blocks);
            for (int f = 0; f < 2; ++f) {
                const size_t expect = (size_t)(per_block[f] * (hi - lo) * 1.15) + 65536;
                if (room(f) < expect) new_chunk(f, expect);
            }
            for (int attempt = 0;; ++attempt) {
                size_t len[2] = {0, 0};
                uint64_t n = 0;
                char *dst[2];
                for (int f = 0; f < 2; ++f) dst[f] = job.chunks[f].back()->as<char>() + job.used[f].back();
                const int rc = sim_pairs(*s, lo, hi, dst[0], room(0), &len[0], dst[1], room(1), &len[1], &n, nullptr, 0, (hipStream_t)stream);
                if (rc == RSQ_ENOSPC && attempt < 2) {              // the estimate was too small (the first call has none): arrays for what the call asked
                    for (int f = 0; f < 2; ++f)
                        if (len[f] > room(f)) new_chunk(f, len[f] + len[f] / 16 + 65536);
                    continue;
                }
                if (rc != RSQ_OK) return rc;
                for (int f = 0; f < 2; ++f) {
                    job.used[f].back() += len[f];
                    job.bytes[f] += len[f];
                    per_block[f] = std::max(per_block[f], (double)len[f] / (double)(hi - lo));
                }
                *n_pairs += n;
                break;
            }
        }
        *r1_bytes = job.bytes[0];
        *r2_bytes = job.bytes[1];
        job.complete = true;
        return (int)RSQ_OK;
    });
}
int rsq_sim_job_free(rsq_sim *s) {
    REQUIRE(s, "null argument");
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        s->job.clear();
        return (int)RSQ_OK;
    });
}
int rsq_sim_job_write(rsq_sim *s, const char *r1_path, uint64_t r1_offset, const char *r2_path, uint64_t r2_offset, uint32_t threads_per_file) {
    REQUIRE(s && r1_path, "null argument");
    return guard([&] {
        const rsq_sim::JobText &job = s->job;
        if (!job.complete) {
            g_last_error = "rsq_sim_job_write: there is no generated text (rsq_sim_job_generate has not run to its end on this simulator, or rsq_sim_job_free has released it)";
            return (int)RSQ_ESTATE;
        }
        if (!r2_path && job.bytes[1]) throw Error("rsq_sim_job_write: the job has text for a second file, but no second path was given");
        const int n_files = r2_path ? 2 : 1;                  // one file: the text of seqToIllumina records kept by rsq_sim_error_model_file
        if (job.is_packed) {                                  // compressed in host memory already (rsq_sim_job_compress): plain writes at the offsets
            const char *names[2] = {r1_path, r2_path};
            const uint64_t at[2] = {r1_offset, r2_offset};
            for (int f = 0; f < n_files; ++f) {
                const int fd = open(names[f], O_WRONLY | O_CREAT, 0644);
                if (fd < 0) throw Error(std::string("cannot open '") + names[f] + "' for writing: " + strerror(errno));
                for (size_t done = 0; done < job.packed[f].size();) {
                    const ssize_t w = pwrite(fd, job.packed[f].data() + done, std::min<size_t>(job.packed[f].size() - done, (size_t)1 << 30), (off_t)(at[f] + done));
                    if (w < 0 && errno == EINTR) continue;
                    if (w <= 0) {
                        close(fd);
                        throw Error(std::string("writing '") + names[f] + "' failed: " + (w < 0 ? strerror(errno) : "no space"));
                    }
                    done += (size_t)w;
                }
                if (close(fd) != 0) throw Error(std::string("closing '") + names[f] + "' failed: " + strerror(errno));
            }
            return (int)RSQ_OK;
        }
        const uint32_t T = threads_per_file ? std::min(threads_per_file, 64u) : 1u;
        const char *paths[2] = {r1_path, r2_path};
        const uint64_t offsets[2] = {r1_offset, r2_offset};
        int fds[2] = {-1, -1};
        // Buffered pwrite()s into ONE file take the inode's lock one after the other, so more threads per file do not help on tmpfs or ext4 (measured on /dev/shm, 23 GB:
        // 1 thread per file 12.9 GB/s, 4 threads 7.5, 8 threads 7.2); copying into a shared mapping of the pre-sized file avoids that lock but pays a page fault per
        // 4 KB (6.4 GB/s however many threads) -- profiles/r03_e_*.  One thread per file is the default; file systems with concurrent direct I/O may want more.
        for (int f = 0; f < n_files; ++f) {
            fds[f] = open(paths[f], O_WRONLY | O_CREAT, 0644);
            if (fds[f] < 0) {
                if (f) close(fds[0]);
                throw Error(std::string("cannot open '") + paths[f] + "' for writing: " + strerror(errno));
            }
        }
        std::atomic<bool> failed{false};
        std::mutex message_mutex;
        std::string message;
        auto fail = [&](const std::string &m) {
            std::lock_guard<std::mutex> lock(message_mutex);
            if (!failed.exchange(true)) message = m;
        };
        // Option job_write_direct: the file is also opened with O_DIRECT, and what lies on whole blocks of the file (kDirectAlign) bypasses the page cache: writers of
        // ONE file then do not take turns on its inode (buffered writes into one file serialise there however many ranks write, profiles/r03_g_*).  A rank's byte
        // range begins and ends anywhere, so its head and tail up to the next block boundary go through the buffered descriptor.  Falls back to buffered writes
        // when the file system refuses O_DIRECT (tmpfs does).
        constexpr uint64_t kDirectAlign = 4096;
        int direct_fds[2] = {-1, -1};
        if (s->opt.job_write_direct)
            for (int f = 0; f < n_files; ++f) direct_fds[f] = open(paths[f], O_WRONLY | O_DIRECT);
        // thread t of file f writes bytes [bytes * t / T, bytes * (t + 1) / T) of the file's text: pieces of at most 32 MB, the copy of one overlapping the write of the one before
        auto work = [&](int f, uint32_t t) {
            try {
                HIP_CHECK(hipSetDevice(s->device));
                const uint64_t begin = job.bytes[f] * t / T, end = job.bytes[f] * (t + 1) / T;
                if (begin == end) return;
                struct Staging {                                    // released however the thread leaves (a failed pwrite or HIP call throws)
                    hipStream_t st = nullptr;
                    char *host[2] = {nullptr, nullptr};
                    ~Staging() {
                        if (st) (void)hipStreamSynchronize(st);     // a copy still in flight must not land in freed memory
                        for (char *h : host)
                            if (h) (void)hipHostFree(h);
                        if (st) (void)hipStreamDestroy(st);
                    }
                } staging;
                HIP_CHECK(hipStreamCreateWithFlags(&staging.st, hipStreamNonBlocking));
                for (char *&h : staging.host) HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&h), kJobSliceBytes, hipHostMallocDefault));      // page-aligned: fit for O_DIRECT
                hipStream_t st = staging.st;
                char *const *host = staging.host;
                struct Piece {
                    uint64_t at;                                    // place in the file's text
                    size_t n;
                    bool direct;
                };
                std::vector<Piece> pieces;
                {
                    const uint64_t file_begin = offsets[f] + begin, file_end = offsets[f] + end;      // in the file
                    uint64_t a = file_begin;
                    const bool direct = direct_fds[f] >= 0;
                    if (direct && a % kDirectAlign) {               // the head up to the first block boundary
                        const uint64_t to = std::min(file_end, (a / kDirectAlign + 1) * kDirectAlign);
                        pieces.push_back(Piece{a - offsets[f], (size_t)(to - a), false});
                        a = to;
                    }
                    const uint64_t whole_end = direct ? std::max(a, file_end / kDirectAlign * kDirectAlign) : file_end;
                    for (; a < whole_end; a += std::min<uint64_t>(kJobSliceBytes, whole_end - a)) pieces.push_back(Piece{a - offsets[f], (size_t)std::min<uint64_t>(kJobSliceBytes, whole_end - a), direct});
                    if (a < file_end) pieces.push_back(Piece{a - offsets[f], (size_t)(file_end - a), false});      // the tail behind the last whole block
                }
                // the text lies in a list of device arrays: a piece may reach over the end of one
                auto copy = [&](size_t i) {
                    uint64_t chunk_at = 0, at = pieces[i].at;
                    size_t left = pieces[i].n, put = 0;
                    for (size_t c = 0; c < job.chunks[f].size() && left; ++c) {
                        const uint64_t c_end = chunk_at + job.used[f][c];
                        if (at < c_end) {
                            const size_t n = (size_t)std::min<uint64_t>(left, c_end - at);
                            HIP_CHECK(hipMemcpyAsync(host[i & 1] + put, job.chunks[f][c]->as<char>() + (at - chunk_at), n, hipMemcpyDeviceToHost, st));
                            at += n, put += n, left -= n;
                        }
                        chunk_at = c_end;
                    }
                    if (left) throw Error("internal: a piece of the job's text lies outside its arrays");
                };
                if (!pieces.empty()) copy(0);
                for (size_t i = 0; i < pieces.size() && !failed; ++i) {
                    HIP_CHECK(hipStreamSynchronize(st));
                    if (i + 1 < pieces.size()) copy(i + 1);
                    size_t done = 0;
                    const int fd = pieces[i].direct ? direct_fds[f] : fds[f];
                    while (done < pieces[i].n) {
                        const ssize_t w = pwrite(fd, host[i & 1] + done, pieces[i].n - done, (off_t)(offsets[f] + pieces[i].at + done));
                        if (w < 0 && errno == EINTR) continue;
                        if (w <= 0) throw Error(std::string("writing '") + paths[f] + "' failed: " + (w < 0 ? strerror(errno) : "no space"));
                        done += (size_t)w;
                        if (pieces[i].direct && done < pieces[i].n && done % kDirectAlign) throw Error(std::string("writing '") + paths[f] + "': a direct write stopped inside a block");
                    }
                }
                HIP_CHECK(hipStreamSynchronize(st));
            } catch (const std::exception &e) {
                fail(e.what());
            }
        };
        std::vector<std::thread> pool;
        for (int f = 0; f < n_files; ++f)
            for (uint32_t t = 0; t < T; ++t) pool.emplace_back(work, f, t);
        for (std::thread &t : pool) t.join();
        for (int f = 0; f < n_files; ++f) {
            if (direct_fds[f] >= 0) close(direct_fds[f]);
            if (close(fds[f]) != 0) fail(std::string("closing '") + paths[f] + "' failed: " + strerror(errno));
        }
        if (failed) throw Error(message);
        return (int)RSQ_OK;
    });
}

// ------------------------------------------------------------------------------------------------ gzip on the device (rsq_deflate.h)
// The code of a call: the symbol counts of a sample of its pieces (k_gzip_pieces<true>), the code built on the host (gz::build_codes), uploaded.
static void gzip_code_of(rsq_sim &s, const uint8_t *text, size_t n, hipStream_t st) {
    const uint64_t n_pieces = cdiv(n, gz::kPiece);
    const uint32_t stride = gz::sample_stride(n_pieces);
    s.gz_hist.reserve(2 * (gz::kLitLen + gz::kDist) * 4);           // the sample walked both ways: by FASTQ lines, and densely (gz::dense_pays decides)
    s.gz_codes.reserve(sizeof(gz::Codes));
    uint32_t *lines = s.gz_hist.as<uint32_t>(), *dense = lines + gz::kLitLen + gz::kDist;
    HIP_CHECK(hipMemsetAsync(lines, 0, 2 * (gz::kLitLen + gz::kDist) * 4, st));
    const dim3 grid((uint32_t)cdiv(n_pieces, stride)), block(gz::kThreads);
    hipLaunchKernelGGL((gz::k_gzip_pieces<true, gz::kProbeStep>), grid, block, 0, st, text, (uint64_t)n, stride, (const gz::Codes *)nullptr, (uint8_t *)nullptr, (uint32_t *)nullptr, lines);
    hipLaunchKernelGGL((gz::k_gzip_pieces<true, gz::kDenseStep>), grid, block, 0, st, text, (uint64_t)n, stride, (const gz::Codes *)nullptr, (uint8_t *)nullptr, (uint32_t *)nullptr, dense);
    HIP_CHECK(hipGetLastError());
    uint32_t hist[2][gz::kLitLen + gz::kDist];
    HIP_CHECK(hipMemcpyAsync(hist, lines, sizeof hist, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    s.gz_dense = gz::dense_pays(hist[0], hist[1]);
    gz::Codes codes = gz::build_codes(hist[s.gz_dense ? 1 : 0]);
    codes.dense = s.gz_dense ? 1u : 0u;
    HIP_CHECK(hipMemcpyAsync(s.gz_codes.as<char>(), &codes, sizeof codes, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));                             // `codes` leaves scope
}
// text[0, n) of device memory as gzip members behind each other in out (device memory, room for out_cap bytes): stretches of at most kGzipStretch pieces go through
// the slots (compress, store what did not fit, scan the sizes, move the members into place).  *out_len = the bytes of all members; more than out_cap: nothing
// useful is in out, RSQ_ENOSPC.  with_code: the call's code is taken from this text first (else the caller has set it: several arrays of one job share a code).
constexpr uint64_t kGzipStretch = 8192;                              // pieces per pass: 535 MB of text, as much again in slots
static int gzip_device(rsq_sim &s, const uint8_t *text, size_t n, uint8_t *out, size_t out_cap, size_t *out_len, bool with_code, hipStream_t st) {
    *out_len = 0;
    if (!n) return RSQ_OK;
    if (with_code && !(s.gz_keep_code && s.gz_have_code)) {
        gzip_code_of(s, text, n, st);
        s.gz_have_code = true;
    }
    const uint64_t n_pieces = cdiv(n, gz::kPiece);
    const uint64_t stretch = std::min<uint64_t>(n_pieces, kGzipStretch);
    s.gz_slots.reserve(stretch * gz::kSlot);
    s.gz_sizes.reserve(stretch * 4);
    s.gz_at.reserve((stretch + 1) * 8);
    s.gz_total.reserve(8);
    size_t total = 0;
    for (uint64_t first = 0; first < n_pieces; first += stretch) {
        const uint32_t pieces = (uint32_t)std::min<uint64_t>(stretch, n_pieces - first);
        const uint8_t *t = text + first * gz::kPiece;
        const uint64_t bytes = std::min<uint64_t>((uint64_t)pieces * gz::kPiece, n - first * gz::kPiece);
        s.timers["gzip"].start(st);
        if (s.gz_dense)
            hipLaunchKernelGGL((gz::k_gzip_pieces<false, gz::kDenseStep>), dim3(pieces), dim3(gz::kThreads), 0, st, t, bytes, 1u, s.gz_codes.as<gz::Codes>(), s.gz_slots.as<uint8_t>(),
                               s.gz_sizes.as<uint32_t>(), (uint32_t *)nullptr);
        else
            hipLaunchKernelGGL((gz::k_gzip_pieces<false, gz::kProbeStep>), dim3(pieces), dim3(gz::kThreads), 0, st, t, bytes, 1u, s.gz_codes.as<gz::Codes>(), s.gz_slots.as<uint8_t>(),
                               s.gz_sizes.as<uint32_t>(), (uint32_t *)nullptr);
        hipLaunchKernelGGL(gz::k_gzip_stored, dim3(pieces), dim3(gz::kThreads), 0, st, t, bytes, s.gz_slots.as<uint8_t>(), s.gz_sizes.as<uint32_t>());
        exclusive_scan(s, s.gz_sizes.as<uint32_t>(), pieces, s.gz_at.as<uint64_t>(), st, nullptr, s.gz_total.as<uint64_t>());
        uint64_t stretch_bytes = 0;
        HIP_CHECK(hipMemcpyAsync(&stretch_bytes, s.gz_total.as<uint64_t>(), 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (total + stretch_bytes <= out_cap)
            hipLaunchKernelGGL(gz::k_gzip_compact, dim3(pieces), dim3(256), 0, st, s.gz_slots.as<uint8_t>(), s.gz_sizes.as<uint32_t>(), s.gz_at.as<uint64_t>(), out + total);
        s.timers["gzip"].stop(st);
        HIP_CHECK(hipGetLastError());
        total += stretch_bytes;
    }
    HIP_CHECK(hipStreamSynchronize(st));
    *out_len = total;
    if (total > out_cap) {
        g_last_error = "output buffer too small: the members need " + std::to_string(total) + " bytes";
        return RSQ_ENOSPC;
    }
    return RSQ_OK;
}
int rsq_sim_gzip_keep_code(rsq_sim *s, int keep) {
    REQUIRE(s, "null argument");
    s->gz_keep_code = keep != 0;
    s->gz_have_code = false;
    return RSQ_OK;
}
// The member that ends a BGZF file (SAM specification 4.1.2: an empty block, 28 bytes): bgzip / htslib warn about a file without it.  It is a complete gzip member of no
// text, so gzip, zlib and SeqAn read the file as before.  Whoever finishes a file of device-made members appends it.
size_t rsq_gzip_eof_member(char *out, size_t cap) {
    static const unsigned char kEof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (out && cap >= sizeof kEof) memcpy(out, kEof, sizeof kEof);
    return sizeof kEof;
}
size_t rsq_gzip_bound(size_t text_len) { return (size_t)cdiv(text_len, gz::kPiece) * (gz::kHeaderBytes + 5u + gz::kTrailerBytes) + text_len; }
int rsq_sim_gzip_device(rsq_sim *s, const char *text_dev, size_t text_len, char *out_dev, size_t out_cap, size_t *out_len, void *stream) {
    REQUIRE(s && out_len && (text_dev || !text_len) && (out_dev || !out_cap), "null argument");
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        reset_call_timers(*s);
        return gzip_device(*s, reinterpret_cast<const uint8_t *>(text_dev), text_len, reinterpret_cast<uint8_t *>(out_dev), out_cap, out_len, true, (hipStream_t)stream);
    });
}
// the kept text of a job on the device: every array of text becomes an array of members (the first array's sample gives the code of the file); the text is released
static void job_compress_on_device(rsq_sim &s, hipStream_t st) {
    rsq_sim::JobText &job = s.job;
    reset_call_timers(s);
    const bool kept = s.gz_keep_code;                                 // a job's text is of one kind: the first array's sample gives the code of all of them
    if (!kept) (void)rsq_sim_gzip_keep_code(&s, 1);
    struct Restore {
        rsq_sim &s;
        bool kept;
        ~Restore() {
            if (!kept) (void)rsq_sim_gzip_keep_code(&s, 0);
        }
    } restore{s, kept};
    for (int f = 0; f < 2; ++f) {
        uint64_t packed_bytes = 0;
        for (size_t c = 0; c < job.chunks[f].size(); ++c) {
            const size_t n = job.used[f][c];
            if (!n) continue;
            const uint8_t *text = job.chunks[f][c]->as<uint8_t>();
            // members need a third of the text or less; an array of that size first, the bound if the text is of another kind (the call then runs again)
            std::unique_ptr<DevBuf> packed(new DevBuf());
            size_t len = 0;
            packed->reserve(n / 2 + ((size_t)1 << 20));
            int rc = gzip_device(s, text, n, packed->as<uint8_t>(), packed->bytes(), &len, true, st);
            if (rc == RSQ_ENOSPC) {
                packed.reset(new DevBuf());
                packed->reserve(rsq_gzip_bound(n));
                rc = gzip_device(s, text, n, packed->as<uint8_t>(), packed->bytes(), &len, false, st);
            }
            if (rc != RSQ_OK) throw Error("compressing the job's text on the device failed: " + g_last_error);
            std::unique_ptr<DevBuf> fitted(new DevBuf());             // an array of the members' size (the first was sized by guess)
            fitted->reserve(len);
            HIP_CHECK(hipMemcpyAsync(fitted->as<char>(), packed->as<char>(), len, hipMemcpyDeviceToDevice, st));
            HIP_CHECK(hipStreamSynchronize(st));
            packed = std::move(fitted);
            job.chunks[f][c] = std::move(packed);                     // the text's array is released
            job.used[f][c] = len;
            packed_bytes += len;
        }
        job.bytes[f] = packed_bytes;
    }
    s.gz_slots.release();                                             // the slots are as large as the text was: not kept beyond the call
}

// The kept text as gzip members (rsq_textio.h ParallelGzip: 1 MB of text each, compressed by a pool of threads) in host memory; the device arrays are released.
// A file of concatenated members is a gzip file: ranks exchange their COMPRESSED sizes and write their members at the offsets like plain text.
int rsq_sim_job_compress(rsq_sim *s, uint64_t *r1_bytes, uint64_t *r2_bytes) {
    REQUIRE(s && r1_bytes && r2_bytes, "null argument");
    return guard([&] {
        rsq_sim::JobText &job = s->job;
        if (!job.complete || job.is_packed || job.device_packed) {
            g_last_error = "rsq_sim_job_compress: there is no generated text, or it has been compressed already";
            return (int)RSQ_ESTATE;
        }
        HIP_CHECK(hipSetDevice(s->device));
        if (!s->opt.host_gzip) {                                   // on the device: the members stay in device memory, rsq_sim_job_write / _job_read serve them like text
            job_compress_on_device(*s, nullptr);
            job.device_packed = true;
            *r1_bytes = job.bytes[0];
            *r2_bytes = job.bytes[1];
            return (int)RSQ_OK;
        }
        s2i::CopyStream st;
        s2i::Pinned host[2];
        for (auto &h : host) h.ensure(kJobSliceBytes);
        for (int f = 0; f < 2; ++f) {
            textio::ParallelGzip gz;
            gz.open_memory(job.packed[f]);
            // the text lies in a list of device arrays; slices of kJobSliceBytes, the copy of one under the compression of the one before
            struct Slice {
                size_t chunk, at, n;
            };
            std::vector<Slice> slices;
            for (size_t c = 0; c < job.chunks[f].size(); ++c)
                for (size_t at = 0; at < job.used[f][c]; at += kJobSliceBytes) slices.push_back(Slice{c, at, std::min<size_t>(kJobSliceBytes, job.used[f][c] - at)});
            auto copy = [&](size_t i) { HIP_CHECK(hipMemcpyAsync(host[i & 1].p, job.chunks[f][slices[i].chunk]->as<char>() + slices[i].at, slices[i].n, hipMemcpyDeviceToHost, st.st)); };
            if (!slices.empty()) copy(0);
            for (size_t i = 0; i < slices.size(); ++i) {
                HIP_CHECK(hipStreamSynchronize(st.st));
                if (i + 1 < slices.size()) copy(i + 1);
                gz.write(host[i & 1].chars(), slices[i].n);
            }
            if (!gz.close()) throw Error("compressing the job's text failed");
            job.chunks[f].clear();
            job.used[f].clear();
            job.bytes[f] = job.packed[f].size();
        }
        job.is_packed = true;
        *r1_bytes = job.bytes[0];
        *r2_bytes = job.bytes[1];
        return (int)RSQ_OK;
    });
}

// a stretch of the kept text into device memory of the caller (what a rank contributes to one round of a gather of the output)
int rsq_sim_job_read(rsq_sim *s, int file, uint64_t at, size_t bytes, char *dst_dev, void *stream) {
    REQUIRE(s && (file == 0 || file == 1) && (dst_dev || !bytes), "null argument, or a file that is neither 0 nor 1");
    return guard([&] {
        const rsq_sim::JobText &job = s->job;
        if (job.is_packed) {
            g_last_error = "rsq_sim_job_read: the text has been compressed into host memory (rsq_sim_job_compress); a gather works on the plain text";
            return (int)RSQ_ESTATE;
        }
        if (!job.complete) {
            g_last_error = "rsq_sim_job_read: there is no generated text (rsq_sim_job_generate has not run to its end on this simulator, or rsq_sim_job_free has released it)";
            return (int)RSQ_ESTATE;
        }
        if (at > job.bytes[file] || bytes > job.bytes[file] - at) {
            g_last_error = "rsq_sim_job_read: bytes [" + std::to_string(at) + ", " + std::to_string(at + bytes) + ") lie outside the file's " + std::to_string(job.bytes[file]) + " bytes of text";
            return (int)RSQ_EINVAL;
        }
        HIP_CHECK(hipSetDevice(s->device));
        uint64_t chunk_at = 0;
        size_t left = bytes, put = 0;
        for (size_t c = 0; c < job.chunks[file].size() && left; ++c) {            // the text lies in a list of device arrays
            const uint64_t c_end = chunk_at + job.used[file][c];
            if (at < c_end) {
                const size_t n = (size_t)std::min<uint64_t>(left, c_end - at);
                HIP_CHECK(hipMemcpyAsync(dst_dev + put, job.chunks[file][c]->as<char>() + (at - chunk_at), n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
                at += n, put += n, left -= n;
            }
            chunk_at = c_end;
        }
        HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        return (int)RSQ_OK;
    });
}

int rsq_sim_adapter_only_pairs(rsq_sim *s, uint64_t first, uint64_t n, char *r1_dev, size_t r1_cap, size_t *r1_len, char *r2_dev, size_t r2_cap, size_t *r2_len,
                               void *stream) {
    REQUIRE(s && r1_len && r2_len && s->prepared, "simulator not prepared");
    return guard([&] {
        HIP_CHECK(hipSetDevice(s->device));
        return adapter_only_pairs(*s, n, first, r1_dev, r1_cap, r1_len, r2_dev, r2_cap, r2_len, (hipStream_t)stream);
    });
}

// A profile with several read lengths draws a record's read length from tables over the fragment length (draw_read_length): a length outside them ends the call
// with the words of the reference's Vect::at (Vect.hpp:196-221 prints them before it throws) -- RSQ_EIO, nothing simulated
static void check_fragment_lengths(rsq_sim *s, uint64_t n, const uint8_t *seg_dev, const uint32_t *frag_len_dev, hipStream_t st) {
    FragmentRange range{{0u, 0u}, {0xFFFFFFFFu, 0xFFFFFFFFu}};
    bool any = false;
    for (int seg = 0; seg < 2; ++seg) {
        const DevReadLengths &rl = s->dev.read_lengths[seg];
        if (rl.fixed) continue;
        any = true;
        range.lo[seg] = std::max((uint32_t)s->insert_lengths_from, rl.row_first);
        range.hi[seg] = std::min(s->dev.insert_to, rl.row_first + rl.rows);
    }
    if (!any) return;
    s->cur->rec_count.reserve(16);
    uint32_t *flag = s->cur->rec_count.as<uint32_t>() + 2;            // (words 0 and 1 are the partition's counts)
    uint32_t *mail = reinterpret_cast<uint32_t *>(&s->mailbox[6]);
    mail[0] = 0xFFFFFFFFu;
    HIP_CHECK(hipMemcpyAsync(flag, mail, 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_fragment_range, dim3(cdiv(n, 256)), dim3(256), 0, st, seg_dev, frag_len_dev, n, range, flag);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(mail, flag, 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const uint32_t bad = mail[0];
    if (bad == 0xFFFFFFFFu) return;
    uint32_t fl = 0;
    uint8_t seg = 0;
    HIP_CHECK(hipMemcpy(&fl, frag_len_dev + bad, 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(&seg, seg_dev + bad, 1, hipMemcpyDeviceToHost));
    const DevReadLengths &rl = s->dev.read_lengths[seg ? 1 : 0];
    const bool in_insert_lengths = fl >= s->insert_lengths_from && fl < s->dev.insert_to;
    const uint64_t from = in_insert_lengths ? rl.row_first : s->insert_lengths_from, to = in_insert_lengths ? (uint64_t)rl.row_first + rl.rows : s->dev.insert_to;
    throw FragmentLengthOutside("record " + std::to_string(bad) + " of the call: Called index " + std::to_string(fl) + " range is from " + std::to_string(from) + " to " + std::to_string(to) +
                                " (the fragment length has no entry in the profile's " + (in_insert_lengths ? "read lengths by fragment length" : "insert lengths") + ")");
}

// the records' reads into the raw arrays: partition by template segment, k_fill_records
// (rec_at / rec_len: records of their own lengths at their own offsets of arrays of array_bytes bytes, read_len = the longest; nullptr: n x read_len bytes)
static RawLayout error_model_fill(rsq_sim *s, uint64_t first_index, uint64_t n, uint32_t read_len, const uint8_t *seqs_dev, const uint8_t *seg_dev, const uint32_t *frag_len_dev,
                                  const uint8_t *dom_dev, const uint8_t *rate_dev, hipStream_t st, const uint32_t *rec_at = nullptr, const uint32_t *rec_len = nullptr,
                                  uint32_t array_bytes = 0, bool fresh_timers = true, const uint16_t *codes = nullptr) {
    // a template longer than the profile's reads needs a wider op buffer than the one sized at create time
    const uint32_t need_ops = (s->rmax + read_len + s->max_adapter + 4u + 15u) / 16u;
    if (need_ops > s->ops_stride) s->ops_stride = need_ops;
    if (n >= 0xFFFFFFFFull) throw Error("at most 2^32-1 records per call");
    s->cur = &s->ws[0];
    if (fresh_timers) reset_call_timers(*s);
    check_fragment_lengths(s, n, seg_dev, frag_len_dev, st);
    RawLayout raw = raw_layout(*s, n);
    // partition the records by template segment: the read kernel's workgroups hold one segment's tables in LDS (binned by tile: build_fill_bins)
    if (!fill_is_binned(*s)) {
        s->cur->rec_flags.reserve(n * 4 + 16);
        s->cur->rec_index.reserve(n * 4 + 16);
        s->cur->rec_count.reserve(8);
        s->cur->offsets.reserve((n + 1) * 8);
        const dim3 rgrid(cdiv(n, 256)), rblock(256);
        hipLaunchKernelGGL(k_record_flags, rgrid, rblock, 0, st, seg_dev, n, s->cur->rec_flags.as<uint32_t>());
        exclusive_scan(*s, s->cur->rec_flags.as<uint32_t>(), n, s->cur->offsets.as<uint64_t>(), st);
        hipLaunchKernelGGL(k_record_partition, rgrid, rblock, 0, st, seg_dev, n, s->cur->offsets.as<uint64_t>(), s->cur->rec_index.as<uint32_t>(), s->cur->rec_count.as<uint32_t>());
        HIP_CHECK(hipGetLastError());
    }
    const RecordJob job{first_index, read_len, seqs_dev, dom_dev, rate_dev, frag_len_dev, s->cur->rec_index.as<uint32_t>(), s->cur->rec_count.as<uint32_t>(), n, rec_at, rec_len, array_bytes, codes};
    raw.order = launch_fill_records(*s, job, seg_dev, n, raw, st);
    return raw;
}

int rsq_sim_error_model(rsq_sim *s, uint64_t first_index, uint64_t n, uint32_t read_len, const uint8_t *seqs_dev, const uint8_t *seg_dev, const uint32_t *frag_len_dev,
                        const uint8_t *dom_dev, const uint8_t *rate_dev, uint8_t *seq_out_dev, uint8_t *qual_out_dev, uint32_t out_stride, uint16_t *read_len_out_dev,
                        uint16_t *num_errors_out_dev, uint16_t *tile_out_dev, char *cigar_out_dev, uint32_t cigar_stride, void *stream) {
    REQUIRE(s && s->prepared, "simulator not prepared");
    REQUIRE(seqs_dev && seg_dev && frag_len_dev && dom_dev && rate_dev && seq_out_dev && qual_out_dev && read_len_out_dev && num_errors_out_dev && tile_out_dev && cigar_out_dev,
            "null device pointer");
    if (!n) return RSQ_OK;
    return guard([&] {
        hipStream_t st = (hipStream_t)stream;
        HIP_CHECK(hipSetDevice(s->device));
        const RawLayout raw = error_model_fill(s, first_index, n, read_len, seqs_dev, seg_dev, frag_len_dev, dom_dev, rate_dev, st);
        s->cur->scan_total.reserve(8);
        HIP_CHECK(hipMemsetAsync(s->cur->scan_total.as<uint32_t>(), 0, 4, st));
        if (!((out_stride | (uint32_t)(uintptr_t)seq_out_dev | (uint32_t)(uintptr_t)qual_out_dev) & 3u))
            hipLaunchKernelGGL(k_error_model_rows, dim3(cdiv(n * 16, 256)), dim3(256), 0, st, raw, n, seq_out_dev, qual_out_dev, out_stride);
        hipLaunchKernelGGL(k_error_model_out, dim3(cdiv(n, 64)), dim3(64), 0, st, raw, n, seq_out_dev, qual_out_dev, out_stride, read_len_out_dev, num_errors_out_dev,
                           tile_out_dev, cigar_out_dev, cigar_stride, s->cur->scan_total.as<uint32_t>());
        HIP_CHECK(hipGetLastError());
        uint32_t overflow = 0;
        HIP_CHECK(hipMemcpyAsync(&overflow, s->cur->scan_total.as<uint32_t>(), 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (overflow) {
            g_last_error = "out_stride or cigar_stride too small for at least one record";
            return (int)RSQ_ENOSPC;
        }
        return (int)RSQ_OK;
    });
}

// the FASTQ text of the records in the raw arrays: sizes, scan, text; *text_len = the bytes needed (RSQ_ENOSPC, nothing written, if text_cap is smaller)
static int error_model_text(rsq_sim *s, const RawLayout &raw, uint64_t n, const RecordIds &ids, char *text_dev, size_t text_cap, size_t *text_len, hipStream_t st) {
    s->cur->sizes.reserve(n * 4 + 16);
    s->cur->off_r1.reserve((n + 1) * 8);
    s->timers["format_write"].start(st);
    hipLaunchKernelGGL(k_record_text_sizes, dim3(cdiv(n, 256)), dim3(256), 0, st, raw, n, ids, s->cur->sizes.as<uint32_t>());
    exclusive_scan(*s, s->cur->sizes.as<uint32_t>(), n, s->cur->off_r1.as<uint64_t>(), st);
    // the waves' LDS image is sized by the longest record of the call before (this call's goes to longest_record for the next one)
    s->longest_record.reserve(8);
    HIP_CHECK(hipMemsetAsync(s->longest_record.as<uint32_t>(), 0, 4, st));
    hipLaunchKernelGGL(k_max_size, dim3(std::min<uint64_t>(1024, cdiv(n, 256))), dim3(256), 0, st, s->cur->sizes.as<uint32_t>(), n, s->longest_record.as<uint32_t>());
    const uint32_t lds = raw.order ? std::min(kFormatLdsMax, format_lds_bytes(s->record_text_bytes) + 16u * kFormatRecords) : format_lds_bytes(s->record_text_bytes);
    if (raw.order)
        hipLaunchKernelGGL(k_record_text_waves<true>, dim3(cdiv(n, kFormatRecords)), dim3(64), lds, st, raw, n, ids, s->cur->off_r1.as<uint64_t>(), text_dev, (uint64_t)text_cap, lds);
    else
        hipLaunchKernelGGL(k_record_text_waves<false>, dim3(cdiv(n, kFormatRecords)), dim3(64), lds, st, raw, n, ids, s->cur->off_r1.as<uint64_t>(), text_dev, (uint64_t)text_cap, lds);
    s->timers["format_write"].stop(st);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(&s->mailbox[2], s->cur->off_r1.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(&s->mailbox[5], s->longest_record.as<uint32_t>(), 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    *text_len = s->mailbox[2];
    if ((uint32_t)s->mailbox[5]) s->record_text_bytes = (uint32_t)s->mailbox[5] + 8u;
    if (*text_len > text_cap) {
        g_last_error = "text buffer too small: need " + std::to_string(*text_len) + " bytes";
        return (int)RSQ_ENOSPC;
    }
    return (int)RSQ_OK;
}

int rsq_sim_error_model_fastq(rsq_sim *s, uint64_t first_index, uint64_t n, uint32_t read_len, const uint8_t *seqs_dev, const uint8_t *seg_dev, const uint32_t *frag_len_dev,
                              const uint8_t *dom_dev, const uint8_t *rate_dev, const char *ids_dev, const uint64_t *id_off_dev, char *text_dev, size_t text_cap,
                              size_t *text_len, void *stream) {
    REQUIRE(s && s->prepared, "simulator not prepared");
    REQUIRE(seqs_dev && seg_dev && frag_len_dev && dom_dev && rate_dev && ids_dev && id_off_dev && text_len && (text_dev || !text_cap), "null pointer");
    *text_len = 0;
    if (!n) return RSQ_OK;
    return guard([&] {
        hipStream_t st = (hipStream_t)stream;
        HIP_CHECK(hipSetDevice(s->device));
        const RawLayout raw = error_model_fill(s, first_index, n, read_len, seqs_dev, seg_dev, frag_len_dev, dom_dev, rate_dev, st);
        return error_model_text(s, raw, n, RecordIds{ids_dev, id_off_dev, nullptr, nullptr}, text_dev, text_cap, text_len, st);
    });
}

// the reference's complaint about a malformed record (Simulator.cpp:2423-2485; rec = the record's text from its '>' on)
static std::string record_message(const std::vector<uint8_t> &rec) {
    std::vector<uint8_t> scratch[3];
    for (auto &v : scratch) v.resize(rec.size() + 8);
    fasta::RecordFields f{0, 0, 0, 0};
    const fasta::RecordError e = fasta::parse_record(rec.data(), rec.size(), fasta::ByteArrays{scratch[0].data(), scratch[1].data(), scratch[2].data()}, f);
    const size_t line_end = (size_t)fasta::find_byte(rec.data(), 1, rec.size(), '\n');
    size_t header_len = line_end - 1;
    if (header_len && rec[line_end - 1] == '\r') --header_len;
    const std::string header(reinterpret_cast<const char *>(rec.data()) + 1, header_len);
    size_t L = 0;                                            // the template's length: as parse_record counts it
    for (size_t p = line_end + 1; p < rec.size(); ++p)
        if (rec[p] != '\n' && !(rec[p] == '\r' && (p + 1 == rec.size() || rec[p + 1] == '\n'))) ++L;
    size_t end = header_len > 2 * L + 2 ? header_len - 2 * L - 3 : 0;
    while (end && header[end] != ' ') --end;
    switch (e) {
        case fasta::kTooShort: return "Read description is too short to contain systematic error information and a sequence id: " + header;
        case fasta::kErrorSeparators: return "The two systematic error entries are not separated by a semicolon from themselves or the rest of the ReSeq information: " + header;
        case fasta::kNoId: return "No sequence id found that is separated by a space from the ReSeq information: " + header;
        case fasta::kSegment: return std::string("Template segment is ") + header[end + 1] + " not 1 or 2: " + header;
        case fasta::kSegmentSeparator: return "The template segment and fragment length are not separated by a semicolon: " + header;
        case fasta::kFragmentLength: {
            const size_t fl_at = end + 3, fl_end = header_len - 2 * L - 2;
            return "Fragment length '" + (fl_end > fl_at ? header.substr(fl_at, fl_end - fl_at) : std::string()) + "' is not a pure integer: " + header;
        }
        case fasta::kContainsN: return "input sequences must not contain N: " + header.substr(0, end);
        default: return "malformed record: " + header;
    }
}

int rsq_sim_error_model_fasta(rsq_sim *s, uint64_t first_index, const char *text_dev, size_t text_len, int final, char *out_dev, size_t out_cap, size_t *out_len,
                              uint64_t *n_records, size_t *consumed, void *stream) {
    REQUIRE(s && s->prepared, "simulator not prepared");
    REQUIRE(out_len && n_records && consumed && (text_dev || !text_len) && (out_dev || !out_cap), "null pointer");
    REQUIRE(text_len < 0xFFFFFFF0ull, "a block of FASTA text has to be shorter than 4 GB");
    *out_len = 0;
    *n_records = 0;
    *consumed = 0;
    if (!text_len) return RSQ_OK;
    return guard([&] {
        hipStream_t st = (hipStream_t)stream;
        HIP_CHECK(hipSetDevice(s->device));
        s->cur = &s->ws[0];
        rsq_sim::Workspace &w = *s->cur;
        const uint8_t *text = reinterpret_cast<const uint8_t *>(text_dev);
        // the record starts: counts per tile, their scan, the starts in input order
        const uint32_t n_tiles = cdiv(text_len, fasta::kTileBytes);
        auto roomy = [](size_t bytes) { return bytes + bytes / 8 + 64; };          // blocks differ by a few percent: grow once
        w.fa_counts.reserve(roomy((size_t)n_tiles * 4));
        w.fa_first.reserve(roomy(((size_t)n_tiles + 1) * 8));
        w.fa_summary.reserve(64);
        const dim3 tgrid(cdiv(n_tiles, fasta::kStartsBlock / 64u)), tblock(fasta::kStartsBlock);
        hipLaunchKernelGGL(fasta::k_fasta_count, tgrid, tblock, 0, st, text, (uint64_t)text_len, n_tiles, w.fa_counts.as<uint32_t>());
        exclusive_scan(*s, w.fa_counts.as<uint32_t>(), n_tiles, w.fa_first.as<uint64_t>(), st);
        HIP_CHECK(hipMemcpyAsync(&s->mailbox[0], w.fa_first.as<uint64_t>() + n_tiles, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        const uint64_t starts = s->mailbox[0];
        if (starts >= 0xFFFFFFFFull) throw Error("more than 2^32 records in one block of text");
        w.fa_at.reserve(roomy((starts + 1) * 4));
        hipLaunchKernelGGL(fasta::k_fasta_starts, tgrid, tblock, 0, st, text, (uint64_t)text_len, n_tiles, w.fa_first.as<uint64_t>(), w.fa_at.as<uint32_t>());
        // a block that is not the input's last leaves its last record to the caller: its end is not known yet
        const uint32_t n = (uint32_t)(final || !starts ? starts : starts - 1);
        w.fa_len.reserve(roomy((size_t)n * 4));
        w.fa_id_len.reserve(roomy((size_t)n * 4));
        w.fa_frag_len.reserve(roomy((size_t)n * 4));
        w.fa_seg.reserve(roomy(n));
        w.fa_codes.reserve(roomy(2 * (text_len + 8)));                // a half-word per byte of text: a record's codes lie at its own offset
        uint32_t *summary = w.fa_summary.as<uint32_t>();
        const uint32_t init[4] = {0u, 0xFFFFFFFFu, 0u, 0u};
        uint32_t *mail = reinterpret_cast<uint32_t *>(&s->mailbox[0]);              // [0..3] summary, [4] the first start, [5] the last start
        memcpy(mail, init, sizeof init);
        HIP_CHECK(hipMemcpyAsync(summary, mail, sizeof init, hipMemcpyHostToDevice, st));
        // text in front of the first record: [0, at[0]) -- at[0] = text_len without any record
        HIP_CHECK(hipMemcpyAsync(&mail[4], w.fa_at.as<uint32_t>(), 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(&mail[5], w.fa_at.as<uint32_t>() + (starts ? starts - 1 : 0), 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        const uint32_t lead_end = mail[4], last_start = mail[5];
        reset_call_timers(*s);
        s->timers["parse_records"].start(st);
        const fasta::Records rec{w.fa_at.as<uint32_t>(), w.fa_len.as<uint32_t>(), w.fa_id_len.as<uint32_t>(), w.fa_frag_len.as<uint32_t>(), w.fa_seg.as<uint8_t>()};
        if (lead_end) hipLaunchKernelGGL(fasta::k_fasta_lead, dim3(std::min(1024u, cdiv(lead_end, 256))), dim3(256), 0, st, text, lead_end, summary);
        if (n) {
            // per launch like fill_lds_bytes: function attributes belong to the device the call runs on, and a process may hold simulators on several
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&fasta::k_fasta_records), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fasta::kStageBytes));
            hipLaunchKernelGGL(fasta::k_fasta_records, dim3(cdiv(n, fasta::kRecordsBlock)), dim3(fasta::kRecordsBlock), fasta::kStageBytes, st, text, n, rec, w.fa_codes.as<uint16_t>(), summary,
                               s->opt.fasta_no_stage ? 0u : fasta::kStageBytes);
        }
        s->timers["parse_records"].stop(st);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(mail, summary, 16, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        const uint32_t longest = mail[0], first_bad = mail[1];
        if (mail[2]) {
            g_last_error = "sequence data without a header line in the input";
            return (int)RSQ_EIO;
        }
        if (first_bad != 0xFFFFFFFFu) {
            uint32_t span[2];
            HIP_CHECK(hipMemcpy(span, w.fa_at.as<uint32_t>() + first_bad, 8, hipMemcpyDeviceToHost));
            std::vector<uint8_t> record(span[1] - span[0]);
            HIP_CHECK(hipMemcpy(record.data(), text + span[0], record.size(), hipMemcpyDeviceToHost));
            g_last_error = record_message(record);
            return (int)RSQ_EIO;
        }
        *consumed = final ? text_len : (starts ? last_start : 0);      // no record starts in a block that is not the last: nothing consumed, the caller hands in more
        *n_records = n;
        if (!n) return (int)RSQ_OK;
        const RawLayout raw = error_model_fill(s, first_index, n, longest, nullptr, w.fa_seg.as<uint8_t>(), w.fa_frag_len.as<uint32_t>(), nullptr, nullptr, st, w.fa_at.as<uint32_t>(),
                                               w.fa_len.as<uint32_t>(), (uint32_t)text_len + 8u, false, w.fa_codes.as<uint16_t>());
        return error_model_text(s, raw, n, RecordIds{text_dev, nullptr, w.fa_at.as<uint32_t>(), w.fa_id_len.as<uint32_t>()}, out_dev, out_cap, out_len, st);
    });
}

int rsq_sim_error_model_file(rsq_sim *s, const char *input_path, const char *output_path, const rsq_error_model_file_options *options, uint64_t *n_records,
                             uint64_t *out_bytes) {
    REQUIRE(s && s->prepared, "simulator not prepared");
    REQUIRE(n_records && out_bytes, "null argument");
    *n_records = 0;
    *out_bytes = 0;
    return guard([&] {
        using namespace s2i;
        const auto t_start = Clock::now();
        rsq_error_model_file_options opt{};
        if (options) opt = *options;
        HIP_CHECK(hipSetDevice(s->device));
        const unsigned hw = std::thread::hardware_concurrency();
        // reader threads of a plain file: one copies 6-9 GB/s out of the page cache
        const uint32_t n_readers = opt.read_threads ? opt.read_threads : std::max(1u, std::min(hw > 3 ? hw - 3 : 1u, 6u));
        const size_t block_bytes = (size_t)std::min<uint32_t>(1u << 20, opt.block_kb ? opt.block_kb : 48u << 10) << 10;
        uint32_t batch_blocks = opt.batch_blocks ? opt.batch_blocks : 8u;
        while (batch_blocks > 1 && batch_blocks * block_bytes > ((size_t)3 << 30)) --batch_blocks;      // a call takes less than 4 GB of text
        uint64_t plain_size = 0;
        FileDescriptor plain;
        SequentialIn seq_in;
        if (input_path) plain.fd = open_plain_file(input_path, plain_size);
        if (plain.fd < 0) {
            if (opt.from || opt.to) throw Error("a byte range needs an input file that can be read at offsets (not compressed, not a pipe)");
            if (!seq_in.open(input_path)) {
                g_last_error = std::string("Could not open '") + input_path + "' for reading.";
                return (int)RSQ_EIO;
            }
        }
        const uint64_t to = std::min(opt.to ? opt.to : plain_size, plain_size), from = std::min(opt.from, to);
        Fault fault;
        OutPipe out(s->device, fault);
        rsq_sim::JobText &job = s->job;
        // a .gz output: every call's text becomes gzip members on the device (rsq_deflate.h) and the members go down the link and into the file as they are -- a third
        // of the bytes, and no host thread compresses (option host_gzip: zlib behind the writer, as before round 5)
        const bool gz_on_device = output_path && textio::has_suffix(output_path, ".gz") && !s->opt.host_gzip;
        DevBuf call_text;
        // the first call's sample gives the code of the whole file -- of THIS file: the caller's setting (and no code of this file's) is what later calls find
        struct RestoreCode {
            rsq_sim *s;
            bool on, kept;
            ~RestoreCode() {
                if (on) (void)rsq_sim_gzip_keep_code(s, kept ? 1 : 0);
            }
        } restore_code{s, gz_on_device, s->gz_keep_code};
        if (gz_on_device) (void)rsq_sim_gzip_keep_code(s, 1);
        if (opt.keep_text) {                                  // the text stays in device memory, as rsq_sim_job_generate keeps a rank's share of the pairs' text
            if (output_path) throw Error("keep_text and an output path exclude each other");
            job.clear();
        } else if (gz_on_device && !opt.from && !opt.to) {     // a whole file of device-made members ends like a BGZF file (a share of a job: whoever ends the file does it)
            char eof[32];
            out.tail.assign(eof, rsq_gzip_eof_member(eof, sizeof eof));
        }
        if (!opt.keep_text && !out.open(output_path, gz_on_device)) {
            g_last_error = std::string("Could not open '") + output_path + "' for writing.";
            return (int)RSQ_EIO;
        }
        auto room = [&] { return job.chunks[0].empty() ? (size_t)0 : job.chunks[0].back()->bytes() - job.used[0].back(); };
        auto new_chunk = [&](size_t bytes) {
            job.chunks[0].push_back(std::make_unique<DevBuf>());
            job.chunks[0].back()->reserve(std::max<size_t>(bytes, (size_t)256 << 20));
            job.used[0].push_back(0);
        };
        InPipe in(s->device, block_bytes, plain.fd >= 0 ? n_readers : 1, fault);
        if (plain.fd >= 0) in.start_file(plain.fd, from, to, n_readers);
        else in.start_stream(seq_in);
        // the records' read kernel for this profile, while the readers fetch the first blocks (a second of compilation the first time a profile is used)
        (void)spec_kernel(*s, SpecKind::kRecords, effective_fill_mask(s->dev.lds.mask, s->force_fill_mode), true, fill_is_binned(*s));
        CopyStream st;
        DevBuf joined[2];                                     // the text of a call: what the call before left over (it lies in the other one), then the blocks
        int next_joined = 0;
        const char *rest = nullptr;                           // the start of a record whose end the next block holds
        size_t rest_len = 0;
        uint64_t records = 0, next_report = 0, calls = 0;
        StageTime t_wait, t_join, t_call;
        double at_first_block = 0;
        int rc = RSQ_OK;
        bool last = false;
        for (uint64_t b = 0; rc == RSQ_OK && !last;) {
            auto t0 = Clock::now();
            InPipe::Slot *slot = in.take(b);
            t_wait.add(t0);
            if (!slot) break;                                 // a side has failed: `fault` says why
            if (!b) at_first_block = seconds_since(t_start);
            t0 = Clock::now();
            DevBuf &j = joined[next_joined];
            next_joined ^= 1;
            j.reserve(rest_len + batch_blocks * block_bytes + 16);
            char *text = j.as<char>();
            st.copy(text, rest, rest_len, hipMemcpyDeviceToDevice);
            size_t len = rest_len;
            for (uint32_t k = 0; slot; ++k) {
                st.copy(text + len, slot->dev.as<char>(), slot->len, hipMemcpyDeviceToDevice);
                len += slot->len;
                last = slot->last;
                in.release(b++);
                slot = last || k + 1 == batch_blocks ? nullptr : in.take(b, false);
            }
            t_join.add(t0);
            t0 = Clock::now();
            DevBuf *o = opt.keep_text ? nullptr : out.begin();
            if (!o && !opt.keep_text) break;
            size_t out_len = 0, used = 0;
            uint64_t n = 0;
            for (int attempt = 0; attempt < 2; ++attempt) {
                const size_t want = std::max(out_len + out_len / 8, len + len / 8) + 4096;
                char *dst;
                size_t cap;
                if (opt.keep_text) {
                    if (room() < (attempt ? out_len : want)) new_chunk(want);
                    dst = job.chunks[0].back()->as<char>() + job.used[0].back();
                    cap = room();
                } else {
                    DevBuf &to = gz_on_device ? call_text : *o;
                    to.reserve(want);
                    dst = to.as<char>();
                    cap = to.bytes();
                }
                rc = rsq_sim_error_model_fasta(s, opt.first_record + records, text, len, last ? 1 : 0, dst, cap, &out_len, &n, &used, st.st);
                if (rc != RSQ_ENOSPC) break;
            }
            if (rc == RSQ_OK && gz_on_device && out_len) {
                size_t packed = 0;
                o->reserve(out_len / 2 + ((size_t)1 << 20));
                rc = gzip_device(*s, call_text.as<uint8_t>(), out_len, o->as<uint8_t>(), o->bytes(), &packed, true, st.st);
                if (rc == RSQ_ENOSPC) {
                    o->reserve(rsq_gzip_bound(out_len));
                    rc = gzip_device(*s, call_text.as<uint8_t>(), out_len, o->as<uint8_t>(), o->bytes(), &packed, false, st.st);
                }
                out_len = packed;
            }
            t_call.add(t0);
            if (rc != RSQ_OK) break;                          // g_last_error holds the reason (the reference's complaint about a record: RSQ_EIO)
            ++calls;
            if (opt.keep_text) {
                job.used[0].back() += out_len;
                job.bytes[0] += out_len;
            } else if (out_len) out.submit(out_len);
            records += n;                                     // first_record + records = the index of the next record in the input (it selects the records' random streams)
            rest = text + used;
            rest_len = len - used;
            if (opt.progress && records >= next_report) {
                opt.progress(records, opt.user);
                next_report = records + 1000000;
            }
        }
        const double t_loop = seconds_since(t_start);
        if (rc != RSQ_OK) fault.raise(g_last_error);
        in.join();
        out.close();
        if (opt.trace && opt.trace_cap)
            snprintf(opt.trace, opt.trace_cap,
                     "first block on the device %.3f s into the call; the simulator side took %.3f s for %llu records in %llu calls: waiting for blocks %.3f, putting blocks together "
                     "%.3f, device calls %.3f (of these waiting for a free output buffer %.3f); readers (summed over %u threads): reading %.3f, uploads %.3f, waiting for a slot %.3f; "
                     "downloads %.3f (+ %.3f waiting for a free buffer); writing %.3f",
                     at_first_block, t_loop, (unsigned long long)records, (unsigned long long)calls, t_wait.s(), t_join.s(), t_call.s(), out.t_dev.s(), (unsigned)in.readers.size(),
                     in.t_read.s(), in.t_upload.s(), in.t_slot.s(), out.t_download.s(), out.t_stage.s(), out.t_write.s());
        if (fault.set) {
            if (opt.keep_text) job.clear();
            g_last_error = fault.what;
            return rc != RSQ_OK ? rc : (int)RSQ_EIO;
        }
        *n_records = records;
        *out_bytes = opt.keep_text ? job.bytes[0] : out.bytes + out.tail.size();
        job.complete = opt.keep_text != 0;
        return (int)RSQ_OK;
    });
}

int rsq_fasta_count_records(const char *path, uint64_t from, uint64_t to, uint32_t threads, uint64_t *n_starts, uint64_t *first_start) {
    REQUIRE(path && n_starts && first_start, "null argument");
    return guard([&] {
        s2i::count_records(path, from, to, threads, n_starts, first_start);
        return (int)RSQ_OK;
    });
}

int rsq_sim_last_kernel_ms(const rsq_sim *s, const char *kernel, double *ms) {
    REQUIRE(s && kernel && ms, "null argument");
    auto it = s->timers.find(kernel);
    REQUIRE(it != s->timers.end() && it->second.launches(), "unknown kernel name or kernel not launched in the last call");
    *ms = it->second.ms();
    return RSQ_OK;
}
int rsq_sim_last_kernel_launches(const rsq_sim *s, const char *kernel, uint32_t *launches) {
    REQUIRE(s && kernel && launches, "null argument");
    auto it = s->timers.find(kernel);
    *launches = it == s->timers.end() ? 0u : (uint32_t)it->second.launches();
    return RSQ_OK;
}

int rsq_dev_alloc(int device, size_t bytes, void **out_dev) {
    REQUIRE(out_dev, "null argument");
    return guard([&] {
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipMalloc(out_dev, bytes ? bytes : 8));
        return RSQ_OK;
    });
}
int rsq_dev_free(int device, void *dev) {
    return guard([&] {
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipFree(dev));
        return RSQ_OK;
    });
}
int rsq_dev_upload(int device, void *dst_dev, const void *src, size_t bytes) {
    return guard([&] {
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipMemcpy(dst_dev, src, bytes, hipMemcpyHostToDevice));
        return RSQ_OK;
    });
}
int rsq_host_alloc(size_t bytes, void **out_host) {
    REQUIRE(out_host, "null argument");
    return guard([&] {
        HIP_CHECK(hipHostMalloc(out_host, bytes ? bytes : 8, hipHostMallocDefault));
        return RSQ_OK;
    });
}
int rsq_host_free(void *host) {
    return guard([&] {
        if (host) HIP_CHECK(hipHostFree(host));
        return RSQ_OK;
    });
}
// device memory to its place in a file: slices through two page-locked buffers, the copy of one under the write of the one before (the writer of a gathered output)
int rsq_dev_pwrite(int device, const void *src_dev, size_t bytes, const char *path, uint64_t offset) {
    REQUIRE(path && (src_dev || !bytes), "null argument");
    return guard([&] {
        HIP_CHECK(hipSetDevice(device));
        struct Staging {
            hipStream_t st = nullptr;
            char *host[2] = {nullptr, nullptr};
            int fd = -1;
            ~Staging() {
                if (st) (void)hipStreamSynchronize(st);
                for (char *h : host)
                    if (h) (void)hipHostFree(h);
                if (st) (void)hipStreamDestroy(st);
                if (fd >= 0) close(fd);
            }
        } g;
        g.fd = open(path, O_WRONLY | O_CREAT, 0644);
        if (g.fd < 0) throw Error(std::string("cannot open '") + path + "' for writing: " + strerror(errno));
        constexpr size_t kSlice = (size_t)16 << 20;
        HIP_CHECK(hipStreamCreateWithFlags(&g.st, hipStreamNonBlocking));
        for (char *&h : g.host) HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&h), std::min(kSlice, std::max<size_t>(bytes, 1)), hipHostMallocDefault));
        const char *src = static_cast<const char *>(src_dev);
        const size_t n_slices = (bytes + kSlice - 1) / kSlice;
        auto copy = [&](size_t i) { HIP_CHECK(hipMemcpyAsync(g.host[i & 1], src + i * kSlice, std::min(kSlice, bytes - i * kSlice), hipMemcpyDeviceToHost, g.st)); };
        if (n_slices) copy(0);
        for (size_t i = 0; i < n_slices; ++i) {
            HIP_CHECK(hipStreamSynchronize(g.st));
            if (i + 1 < n_slices) copy(i + 1);
            const size_t n = std::min(kSlice, bytes - i * kSlice);
            for (size_t done = 0; done < n;) {
                const ssize_t w = pwrite(g.fd, g.host[i & 1] + done, n - done, (off_t)(offset + i * kSlice + done));
                if (w < 0 && errno == EINTR) continue;
                if (w <= 0) throw Error(std::string("writing '") + path + "' failed: " + (w < 0 ? strerror(errno) : "no space"));
                done += (size_t)w;
            }
        }
        const int fd = g.fd;
        g.fd = -1;
        if (close(fd) != 0) throw Error(std::string("closing '") + path + "' failed: " + strerror(errno));
        return RSQ_OK;
    });
}
int rsq_dev_download(int device, void *dst, const void *src_dev, size_t bytes) {
    return guard([&] {
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipMemcpy(dst, src_dev, bytes, hipMemcpyDeviceToHost));
        return RSQ_OK;
    });
}
// streams of the caller's own (the CLI's reader, simulator and writer sides each work on one): copies on them return when they are done and leave
// the other streams alone -- the plain calls above go through the null stream, which waits for every other one
int rsq_stream_create(int device, void **out_stream) {
    REQUIRE(out_stream, "null argument");
    return guard([&] {
        HIP_CHECK(hipSetDevice(device));
        hipStream_t st = nullptr;
        HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        *out_stream = st;
        return RSQ_OK;
    });
}
int rsq_stream_destroy(int device, void *stream) {
    return guard([&] {
        HIP_CHECK(hipSetDevice(device));
        if (stream) HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
        return RSQ_OK;
    });
}
int rsq_dev_copy_on(int device, void *dst, const void *src, size_t bytes, int kind, void *stream) {
    REQUIRE(kind >= 0 && kind <= 2 && (dst || !bytes) && (src || !bytes), "kind is 0 (host to device), 1 (device to host) or 2 (device to device)");
    return guard([&] {
        HIP_CHECK(hipSetDevice(device));
        static const hipMemcpyKind kinds[3] = {hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice};
        if (bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, kinds[kind], (hipStream_t)stream));
        HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        return RSQ_OK;
    });
}

}  // extern "C"
