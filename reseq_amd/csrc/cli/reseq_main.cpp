// reseq_main.cpp -- `reseq` command line for the simulation stage, on top of the C ABI (include/reseq_amd.h).
//
// Keeps the reference's modes and flags for this path (reseq/main.cpp:434-438 general, :710-753 illuminaPE,
// :1009-1021 seqToIllumina): `reseq illuminaPE` simulates paired reads from a fitted profile, `reseq seqToIllumina`
// (alias `replaceQuals`) applies the error and quality model to given sequences.  Profile creation (BAM statistics,
// bias fit, IPF) is not part of this build: the flags are recognised and rejected with a clear message.
#include <stdint.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>

#include <fstream>
#include <iostream>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <random>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/reseq_amd.h"
#include "../rsq_textio.h"

namespace {

int g_verbosity = 4;
#define INFO(msg)                                                \
    do {                                                         \
        if (g_verbosity >= 3) std::cerr << "[INFO] " << msg << std::endl; \
    } while (0)
#define WARN(msg)                                                \
    do {                                                         \
        if (g_verbosity >= 2) std::cerr << "[WARN] " << msg << std::endl; \
    } while (0)
#define ERR(msg)                                                 \
    do {                                                         \
        if (g_verbosity >= 1) std::cerr << "[ERROR] " << msg << std::endl; \
    } while (0)

struct Args {
    std::map<std::string, std::string> val;
    std::set<std::string> flag;
    bool has(const std::string &k) const { return val.count(k) || flag.count(k); }
    std::string get(const std::string &k, const std::string &def = "") const {
        auto it = val.find(k);
        return it == val.end() ? def : it->second;
    }
};

const std::map<std::string, std::string> kShort = {{"-j", "threads"}, {"-h", "help"}, {"-b", "bamIn"}, {"-r", "refIn"}, {"-s", "statsIn"}, {"-S", "statsOut"},
                                                   {"-v", "vcfIn"},   {"-p", "probabilitiesIn"}, {"-P", "probabilitiesOut"}, {"-1", "firstReadsOut"},
                                                   {"-2", "secondReadsOut"}, {"-c", "coverage"}, {"-R", "refSim"}, {"-V", "vcfSim"}, {"-i", "input"}, {"-o", "output"}};
const std::set<std::string> kFlags = {"help", "version", "noBias", "noTiles", "statsOnly", "tiles", "stopAfterEstimation", "noInDelErrors", "noSubstitutionErrors", "maxLenDeletion", "maxReadLength",
                                      "dumpArchiveLayout", "traceStages"};

bool parse(int argc, char **argv, int first, Args &a) {
    for (int i = first; i < argc; ++i) {
        std::string k = argv[i];
        if (k.rfind("--", 0) == 0) k = k.substr(2);
        else if (kShort.count(k)) k = kShort.at(k);
        else {
            ERR("unrecognised argument '" << argv[i] << "'");
            return false;
        }
        std::string v;
        size_t eq = k.find('=');
        if (eq != std::string::npos) {
            v = k.substr(eq + 1);
            k = k.substr(0, eq);
        }
        if (kFlags.count(k)) {
            a.flag.insert(k);
            continue;
        }
        if (eq == std::string::npos) {
            if (i + 1 >= argc) {
                ERR("option '" << argv[i] << "' needs a value");
                return false;
            }
            v = argv[++i];
        }
        a.val[k] = v;
    }
    return true;
}

bool check(int rc, const char *what) {
    if (rc == RSQ_OK) return true;
    ERR(what << ": " << rsq_last_error());
    return false;
}

// numeric options: the whole value has to be a number (boost::program_options rejects anything else, main.cpp:1043-1060)
bool parse_u64(const Args &a, const std::string &key, uint64_t &out) {
    const std::string v = a.get(key, "");
    char *end = nullptr;
    errno = 0;
    out = strtoull(v.c_str(), &end, 10);
    if (v.empty() || *end || errno || v[0] == '-') {
        ERR("the argument ('" << v << "') for option '--" << key << "' is invalid");
        return false;
    }
    return true;
}
bool parse_double(const Args &a, const std::string &key, double &out) {
    const std::string v = a.get(key, "");
    char *end = nullptr;
    errno = 0;
    out = strtod(v.c_str(), &end);
    if (v.empty() || *end || errno) {
        ERR("the argument ('" << v << "') for option '--" << key << "' is invalid");
        return false;
    }
    return true;
}

uint64_t get_seed(const Args &a) {                      // main.cpp:340: random seed if none is given
    if (a.has("seed")) return strtoull(a.get("seed").c_str(), nullptr, 10);
    std::random_device rd;
    uint64_t s = ((uint64_t)rd() << 32) | rd();
    INFO("Using random seed " << s);
    return s;
}

bool load_profile(const Args &a, rsq_profile **p) {
    for (const char *k : {"bamIn", "adapterFile", "adapterMatrix", "statsOut", "vcfIn", "statsOnly", "noBias", "tiles", "probabilitiesOut", "stopAfterEstimation"})
        if (a.has(k) && !(std::string(k) == "stopAfterEstimation" && a.has("writeSysError"))) {       // main.cpp:845: the profile alone may be asked for
            ERR("--" << k << ": profile creation (statistics, bias fit, IPF) is not supported in this build; create the profile with ReSeq (`reseq illuminaPE -b ... --statsOnly` / `--stopAfterEstimation`) and pass its .reseq file with -s");
            return false;
        }
    if (!a.has("statsIn")) {
        ERR("statsIn option mandatory.");
        return false;
    }
    // main.cpp:729-730,776-779: the fit itself is not part of this build, a `.reseq.ipf` file is taken as stored (--ipfIterations 0)
    if (a.has("ipfIterations") && a.get("ipfIterations") != "0") INFO("--ipfIterations is ignored: fitted tables are used as stored (as with --ipfIterations 0)");
    char *end = nullptr;
    const double ipf_precision = a.has("ipfPrecision") ? strtod(a.get("ipfPrecision").c_str(), &end) : 5.0;
    if (a.has("ipfPrecision") && (end == a.get("ipfPrecision").c_str() || *end || !(ipf_precision > 0.0))) {
        ERR("ipfPrecision must be positive.");
        return false;
    }
    // a `.reseq` statistics file (with its `.reseq.ipf`, main.cpp:837: "<statsIn>.ipf" unless -p names another) or an RSQP container
    INFO("Reading profile from " << a.get("statsIn"));
    int archives = 0;                                             // by the file's content, not by the options given
    rsq_profile_is_reseq_archive(a.get("statsIn").c_str(), &archives);
    if (!archives && (a.has("probabilitiesIn") || a.has("ipfPrecision")))
        WARN("--probabilitiesIn / --ipfPrecision are ignored: " << a.get("statsIn") << " is an RSQP container, which holds the prepared tables");
    if (!check(archives ? rsq_profile_load_reseq(a.get("statsIn").c_str(), a.has("probabilitiesIn") ? a.get("probabilitiesIn").c_str() : nullptr, ipf_precision, p)
                        : rsq_profile_load(a.get("statsIn").c_str(), p),
               "Could not load profile"))
        return false;
    if (*rsq_last_warning()) WARN(rsq_last_warning());
    const double mult = a.has("errorMutliplier") ? atof(a.get("errorMutliplier").c_str()) : 1.0;
    if (a.has("noInDelErrors") && !check(rsq_profile_remove_indel_errors(*p), "noInDelErrors")) return false;          // main.cpp:964-982
    if (a.has("noSubstitutionErrors")) {
        if (mult != 1.0) {
            ERR("noSubstitutionErrors and errorMutliplier cannot be combined.");
            return false;
        }
        if (!check(rsq_profile_remove_substitution_errors(*p), "noSubstitutionErrors")) return false;
    } else if (mult != 1.0 && !check(rsq_profile_change_error_rate(*p, mult), "errorMutliplier")) return false;
    return true;
}

// FASTQ / FASTA text files: gzip / bzip2 when the name ends in .gz / .bz2, inputs by content (SeqAn's SeqFileOut / SeqFileIn pick the
// format the same way); no file = stdout / stdin
struct TextOut {
    rsq::textio::Writer w;
    bool failed = false;
    bool open(const std::string &path) {
        try {
            return w.open(path);
        } catch (const std::exception &e) {
            ERR(e.what());
            return false;
        }
    }
    void write(const char *data, size_t n) {
        if (w.is_open()) w.write(data, n);
        else failed = failed || fwrite(data, 1, n, stdout) != n;
    }
    bool good() const { return !failed && !w.failed; }
    void close() { failed = !w.close() || failed; }
};
struct TextIn {                       // lines of a plain, gzip or bzip2 file, or of stdin
    rsq::textio::Reader r;
    bool is_file = false, failed = false;
    std::vector<char> buf = std::vector<char>(1 << 16);
    size_t at = 0, have = 0;
    bool open(const std::string &path) {
        try {
            return is_file = r.open(path);
        } catch (const std::exception &e) {
            ERR(e.what());
            return false;
        }
    }
    bool getline(std::string &line) {
        if (!is_file) return (bool)std::getline(std::cin, line);
        line.clear();
        for (;;) {
            if (at == have) {
                int n = 0;
                try {
                    n = r.read(buf.data(), (unsigned)buf.size());
                } catch (const std::exception &e) {
                    ERR(e.what());
                    failed = true;
                    return false;
                }
                if (n <= 0) return !line.empty();
                at = 0;
                have = (size_t)n;
            }
            const char *p = buf.data() + at, *e = (const char *)memchr(p, '\n', have - at);
            if (e) {
                line.append(p, (size_t)(e - p));
                at += (size_t)(e - p) + 1;
                return true;
            }
            line.append(p, have - at);
            at = have;
        }
    }
    // raw bytes (the block reader of seqToIllumina): bytes read, 0 at the end, < 0 on an error
    int read_raw(void *dst, unsigned n) {
        if (!is_file) {
            const size_t got = fread(dst, 1, n, stdin);
            return got ? (int)got : (ferror(stdin) ? -1 : 0);
        }
        try {
            return r.read(dst, n);
        } catch (const std::exception &e) {
            ERR(e.what());
            failed = true;
            return -1;
        }
    }
    void close() { r.close(); }
};

struct DevBuffer {
    void *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t n) {
        if (n <= cap) return true;
        if (p) rsq_dev_free(0, p);
        p = nullptr;
        cap = 0;
        if (!check(rsq_dev_alloc(0, n, &p), "device allocation")) return false;
        cap = n;
        return true;
    }
    ~DevBuffer() {
        if (p) rsq_dev_free(0, p);
    }
};

// One output file with its own writer thread and two page-locked staging buffers: while the thread writes (and compresses) batch i,
// the simulator produces batch i+1 and its text is copied into the other buffer.
struct AsyncOut {
    TextOut out;
    void *buf[2] = {nullptr, nullptr};
    size_t cap[2] = {0, 0}, len[2] = {0, 0};
    bool full[2] = {false, false};
    int next_fill = 0, next_write = 0;
    bool stop = false, started = false;
    std::mutex m;
    std::condition_variable cv;
    std::thread worker;
    bool open(const std::string &path) {                  // an empty path: stdout
        if (!path.empty() && !out.open(path)) return false;
        worker = std::thread([this] {
            std::unique_lock<std::mutex> lock(m);
            for (;;) {
                cv.wait(lock, [this] { return full[next_write] || stop; });
                if (!full[next_write]) return;
                const int k = next_write;
                lock.unlock();
                out.write(static_cast<const char *>(buf[k]), len[k]);
                lock.lock();
                full[k] = false;
                next_write ^= 1;
                cv.notify_all();
            }
        });
        started = true;
        return true;
    }
    // copies `bytes` of device text through the staging buffers, a chunk at a time, and queues the chunks
    static constexpr size_t kChunk = 64u << 20;
    bool push(const DevBuffer &d, size_t bytes) {
        for (size_t done = 0; done < bytes; done += kChunk) {
            const size_t n = std::min(kChunk, bytes - done);
            int k;
            {
                std::unique_lock<std::mutex> lock(m);
                k = next_fill;
                cv.wait(lock, [&] { return !full[k]; });
            }
            if (!buf[k]) {
                if (!check(rsq_host_alloc(kChunk, &buf[k]), "host buffer")) return false;
                cap[k] = kChunk;
            }
            if (!check(rsq_dev_download(0, buf[k], static_cast<const char *>(d.p) + done, n), "download")) return false;
            {
                std::lock_guard<std::mutex> lock(m);
                len[k] = n;
                full[k] = true;
                next_fill ^= 1;
            }
            cv.notify_all();
        }
        return out.good();
    }
    void close() {
        if (started) {
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [this] { return !full[0] && !full[1]; });
                stop = true;
            }
            cv.notify_all();
            worker.join();
            started = false;
        }
        out.close();
        for (int k = 0; k < 2; ++k)
            if (buf[k]) rsq_host_free(buf[k]);
        buf[0] = buf[1] = nullptr;
    }
    bool good() const { return out.good(); }
};

bool flush_pair(DevBuffer &d1, size_t l1, DevBuffer &d2, size_t l2, AsyncOut &f1, AsyncOut &f2) { return f1.push(d1, l1) && f2.push(d2, l2); }

int illumina_pe(const Args &a) {
    const std::string vcf_path = a.get("vcfSim", "");               // -V: per-allele simulation (substitutions; the library refuses what it cannot simulate yet)
    // main.cpp:862-908: --refBias keep|no|draw|file, --refBiasFile implies file; keep is the default
    int ref_bias_mode = 0;
    const std::string ref_bias_file = a.get("refBiasFile", "");
    if (a.has("refBias")) {
        const std::string m = a.get("refBias");
        if (m == "keep") ref_bias_mode = 0;
        else if (m == "no") ref_bias_mode = 1;
        else if (m == "draw") ref_bias_mode = 2;
        else if (m == "file") ref_bias_mode = 3;
        else {
            ERR("Unknown option for refSeqBias: " << m);
            return 1;
        }
        if ((ref_bias_mode == 3) != !ref_bias_file.empty()) {
            ERR((ref_bias_mode == 3 ? "refBiasFile option mandatory if refBias is set to 'file'." : "refBiasFile option only allowed if refBias is set to 'file'."));
            return 1;
        }
    } else if (!ref_bias_file.empty()) {
        INFO("Reading reference sequence biases from file.");
        ref_bias_mode = 3;
    }
    // main.cpp:351-397 WriteSysError: the two options exclude each other; a written profile is the one the simulation then reads
    const std::string sys_write = a.get("writeSysError", "");
    std::string sys_read = a.get("readSysError", "");
    if (!sys_write.empty() && !sys_read.empty()) {
        ERR("writeSysError and readSysError option are mutually exclusive. Specify the one or the other.");
        return 1;
    }
    const std::string ref_path = a.has("refSim") ? a.get("refSim") : a.get("refIn");
    if (ref_path.empty()) {
        ERR("refIn or refSim option mandatory.");
        return 1;
    }
    const std::string out1 = a.get("firstReadsOut", "reseq-R1.fq"), out2 = a.get("secondReadsOut", "reseq-R2.fq");      // main.cpp:404,412
    rsq_profile *prof = nullptr;
    rsq_ref *ref = nullptr;
    rsq_sim *sim = nullptr;
    bool ok = load_profile(a, &prof);
    const uint64_t seed = ok ? get_seed(a) : 0;
    if (ok) {
        INFO("Reading reference from " << ref_path);
        ok = check(rsq_ref_load_fasta(ref_path.c_str(), &ref), "Could not load reference") && check(rsq_ref_replace_n(ref, seed), "ReplaceN");
    }
    if (ok && !vcf_path.empty()) {
        INFO("Reading variants from " << vcf_path);
        ok = check(rsq_ref_read_variants(ref, vcf_path.c_str()), "Could not read the variant file");
    }
    ok = ok && check(rsq_sim_create(prof, ref, 0, &sim), "Could not set up the simulator");
    if (ok && !sys_write.empty()) {
        INFO("Writing systematic error profile to " << sys_write);
        ok = check(rsq_sim_create_sys_error_profile(sim, seed, sys_write.c_str(), nullptr), "Could not write systematic error profile");
        if (!ok) remove(sys_write.c_str());                  // main.cpp:392
        sys_read = sys_write;
    }
    if (ok && a.has("stopAfterEstimation")) {                 // main.cpp:845: with writeSysError the profile is all that was asked for
        rsq_sim_free(sim);
        rsq_ref_free(ref);
        rsq_profile_free(prof);
        return 0;
    }
    if (ok && ref_bias_mode == 3) ok = check(rsq_sim_set_ref_bias_file(sim, ref_bias_file.c_str()), "refBiasFile");
    if (ok && !a.get("methylation", "").empty()) {            // Simulator.cpp:2770-2780 PrepareMethylationFile
        INFO("Reading methylation from file: " << a.get("methylation"));
        ok = check(rsq_sim_read_methylation(sim, a.get("methylation").c_str()), "Could not read methylation file");
    }
    uint64_t num_reads = 0;
    double coverage = 0.0;
    if (ok && a.has("numReads") && a.has("coverage")) {          // main.cpp:783-786
        ERR("numReads and coverage option are mutually exclusive. Specify the one or the other.");
        ok = false;
    }
    if (ok && a.has("numReads")) ok = parse_u64(a, "numReads", num_reads);
    if (ok && a.has("coverage")) ok = parse_double(a, "coverage", coverage);
    if (ok) {
        INFO("Preparing for simulation");
        ok = check(rsq_sim_prepare(sim, seed, num_reads, coverage, ref_bias_mode, a.get("recordBaseIdentifier", "ReseqRead").c_str(), nullptr), "Preparation failed");
    }
    if (ok && !sys_read.empty()) ok = check(rsq_sim_read_sys_errors(sim, sys_read.c_str()), "Could not read systematic error profile");
    AsyncOut f1, f2;
    if (ok) {
        const bool o1 = f1.open(out1), o2 = f2.open(out2);
        if (!o1 || !o2) {
            ERR("Could not open '" << (o1 ? out2 : out1) << "' for writing.");
            ok = false;
        }
    }
    if (ok) {
        rsq_sim_info info;
        rsq_sim_get_info(sim, &info);
        INFO("Aiming for " << info.total_pairs + info.adapter_only_pairs << " read pairs");
        INFO("Starting read generation");
        DevBuffer d1, d2;
        uint64_t written = 0;
        // about 4 M pairs per call: large launches keep the persistent read kernel's tail short, and sparse coverage needs long block ranges
        const double pairs_per_block = (double)info.total_pairs / std::max<uint32_t>(1u, info.total_blocks);
        const uint32_t step = (uint32_t)std::min(100000.0, std::max(2000.0, 4e6 / std::max(1e-9, pairs_per_block)));
        for (uint32_t lo = 1; ok && lo <= info.total_blocks; lo += step) {
            const uint32_t hi = std::min(info.total_blocks + 1, lo + step);
            size_t l1 = 0, l2 = 0;
            uint64_t n = 0;
            int rc = rsq_sim_pairs(sim, lo, hi, (char *)d1.p, d1.cap, &l1, (char *)d2.p, d2.cap, &l2, &n, nullptr, 0, nullptr);
            if (rc == RSQ_ENOSPC) {
                ok = d1.ensure(l1 + l1 / 8 + 4096) && d2.ensure(l2 + l2 / 8 + 4096);
                if (ok) rc = rsq_sim_pairs(sim, lo, hi, (char *)d1.p, d1.cap, &l1, (char *)d2.p, d2.cap, &l2, &n, nullptr, 0, nullptr);
            }
            ok = ok && check(rc, "Simulation failed") && (n == 0 || flush_pair(d1, l1, d2, l2, f1, f2));
            written += n;
            if (ok && n) INFO("Generated " << written << " read pairs (" << (info.total_pairs ? (written * 100 + info.total_pairs / 2) / info.total_pairs : 0) << "%).");
        }
        for (uint64_t first = 0; ok && first < info.adapter_only_pairs; first += 100000) {       // Simulator.cpp:2359-2382
            const uint64_t n = std::min<uint64_t>(100000, info.adapter_only_pairs - first);
            size_t l1 = 0, l2 = 0;
            int rc = rsq_sim_adapter_only_pairs(sim, first, n, (char *)d1.p, d1.cap, &l1, (char *)d2.p, d2.cap, &l2, nullptr);
            if (rc == RSQ_ENOSPC) {
                ok = d1.ensure(l1 + 4096) && d2.ensure(l2 + 4096);
                if (ok) rc = rsq_sim_adapter_only_pairs(sim, first, n, (char *)d1.p, d1.cap, &l1, (char *)d2.p, d2.cap, &l2, nullptr);
            }
            ok = ok && check(rc, "Simulation of adapter-only pairs failed") && flush_pair(d1, l1, d2, l2, f1, f2);
        }
    }
    f1.close();
    f2.close();
    ok = ok && f1.good() && f2.good();
    rsq_sim_free(sim);
    rsq_ref_free(ref);
    rsq_profile_free(prof);
    if (!ok) {                                               // Simulator.cpp:2888-2892: do not leave partial output behind
        ERR("An error occurred in the process: Terminating simulation");
        remove(out1.c_str());
        remove(out2.c_str());
        return 1;
    }
    INFO("Simulation finished succesfully");
    return 0;
}

struct HostArray {                    // page-locked, grow-only
    void *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t n) {
        if (n <= cap) return true;
        if (p) rsq_host_free(p);
        p = nullptr;
        cap = 0;
        if (!check(rsq_host_alloc(n, &p), "host buffer")) return false;
        cap = n;
        return true;
    }
    template <class T>
    T *as() { return static_cast<T *>(p); }
    ~HostArray() {
        if (p) rsq_host_free(p);
    }
};

using Clock = std::chrono::steady_clock;
const Clock::time_point g_process_start = Clock::now();
double seconds_since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }
struct StageTime {                    // seconds a side of the pipeline spent in one of its stages, summed over its threads
    std::atomic<uint64_t> ns{0};
    void add(Clock::time_point t0) { ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - t0).count(); }
    double s() const { return (double)ns.load() * 1e-9; }
};

// ---- seqToIllumina as a pipeline (Simulator::SimulateErrorModelOnly, Simulator.cpp:2900-3014: a reader, ErrorModelOnlyThread :2514-2560, ordered output
// :184-213).  Here the FASTA text goes to the device as it stands in the file and is parsed there (rsq_sim_error_model_fasta), so the host only moves bytes:
//   input side    blocks of the file in page-locked slots, uploaded by the thread that read them -- a plain file is read by several threads at fixed offsets
//                 (records that cross a block's end are the simulator side's business), a compressed file or stdin by one;
//   simulator     the main thread takes the blocks in input order -- all that are there, up to eight -- and puts them behind what the call before left over
//                 (device to device: a call on 100 000 records takes 1.1 ms, on 800 000 four: the read kernel of a small call is a chain of 150 steps on waves
//                 that have a SIMD to themselves), runs the device call;
//   output side   a thread downloads the FASTQ text into page-locked buffers, another writes (and compresses) them: OutPipe.
// Every side has its own stream and its own buffers: they overlap.
struct InPipe {
    struct Slot {
        HostArray host;
        DevBuffer dev;
        size_t len = 0;
        bool last = false, ready = false;
        uint64_t turn = 0;                           // the block this slot serves next
    };
    const size_t block_bytes;
    std::vector<Slot> slots;
    TextIn *stream_in = nullptr;                     // sequential input (stdin, gzip, bzip2) ...
    int fd = -1;                                     // ... or a plain file read at offsets
    uint64_t file_size = 0, n_blocks = 0;
    std::atomic<uint64_t> next{0};
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::thread> readers;
    bool failed = false, abort = false;
    StageTime t_read, t_upload, t_slot;
    InPipe(size_t block, size_t n_readers) : block_bytes(block), slots(n_readers + 2) {
        for (size_t k = 0; k < slots.size(); ++k) slots[k].turn = k;
    }
    void start_file(int file, uint64_t size, size_t n_readers) {
        fd = file;
        file_size = size;
        n_blocks = std::max<uint64_t>(1, (size + block_bytes - 1) / block_bytes);
        for (size_t k = 0; k < n_readers; ++k) readers.emplace_back([this] { read_at_offsets(); });
    }
    void start_stream(TextIn &in) {
        stream_in = &in;
        readers.emplace_back([this] { read_in_sequence(); });
    }
    void fail() {
        std::lock_guard<std::mutex> lock(m);
        failed = true;
        cv.notify_all();
    }
    Slot *wait_for_slot(uint64_t b) {                // the slot of block b once the block that used it before is done with; nullptr: the run ends
        const auto t0 = Clock::now();
        Slot *s = &slots[b % slots.size()];
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&] { return (s->turn == b && !s->ready) || abort || failed; });
        t_slot.add(t0);
        return abort || failed ? nullptr : s;
    }
    bool upload_and_publish(Slot *s, size_t len, bool last, void *stream) {
        const auto t0 = Clock::now();
        if (!check(rsq_dev_copy_on(0, s->dev.p, s->host.p, len, 0, stream), "upload")) return false;
        t_upload.add(t0);
        std::lock_guard<std::mutex> lock(m);
        s->len = len;
        s->last = last;
        s->ready = true;
        cv.notify_all();
        return true;
    }
    void read_at_offsets() {
        void *stream = nullptr;
        if (!check(rsq_stream_create(0, &stream), "stream")) return fail();
        for (;;) {
            const uint64_t b = next++;
            if (b >= n_blocks) break;
            Slot *s = wait_for_slot(b);
            if (!s) break;
            const uint64_t off = b * block_bytes;
            const size_t len = (size_t)std::min<uint64_t>(block_bytes, file_size - off);
            if (!s->host.ensure(block_bytes) || !s->dev.ensure(block_bytes + 16)) return fail();
            const auto t0 = Clock::now();
            for (size_t have = 0; have < len;) {
                const ssize_t got = pread(fd, s->host.as<char>() + have, len - have, (off_t)(off + have));
                if (got <= 0) {
                    if (got < 0 && errno == EINTR) continue;
                    ERR("reading the input failed" << (got ? std::string(": ") + strerror(errno) : std::string(": the file has become shorter")));
                    return fail();
                }
                have += (size_t)got;
            }
            t_read.add(t0);
            if (!upload_and_publish(s, len, b + 1 == n_blocks, stream)) return fail();
        }
        rsq_stream_destroy(0, stream);
    }
    void read_in_sequence() {
        void *stream = nullptr;
        if (!check(rsq_stream_create(0, &stream), "stream")) return fail();
        for (uint64_t b = 0;; ++b) {
            Slot *s = wait_for_slot(b);
            if (!s) break;
            if (!s->host.ensure(block_bytes) || !s->dev.ensure(block_bytes + 16)) return fail();
            const auto t0 = Clock::now();
            size_t have = 0;
            bool end_of_input = false;
            while (have < block_bytes) {
                const int got = stream_in->read_raw(s->host.as<char>() + have, (unsigned)std::min<size_t>(block_bytes - have, 1u << 30));
                if (got < 0) return fail();
                if (!got) {
                    end_of_input = true;
                    break;
                }
                have += (size_t)got;
            }
            t_read.add(t0);
            if (!upload_and_publish(s, have, end_of_input, stream)) return fail();
            if (end_of_input) break;
        }
        rsq_stream_destroy(0, stream);
    }
    // block b, in input order; nullptr after an error -- or, if the caller does not want to wait, while the block is not there yet
    Slot *take(uint64_t b, bool wait = true) {
        Slot *s = &slots[b % slots.size()];
        std::unique_lock<std::mutex> lock(m);
        if (wait) cv.wait(lock, [&] { return (s->turn == b && s->ready) || failed; });
        return failed || !(s->turn == b && s->ready) ? nullptr : s;
    }
    void release(uint64_t b) {
        Slot &s = slots[b % slots.size()];
        std::lock_guard<std::mutex> lock(m);
        s.ready = false;
        s.turn += slots.size();
        cv.notify_all();
    }
    void join() {
        {
            std::lock_guard<std::mutex> lock(m);
            abort = true;
            cv.notify_all();
        }
        for (std::thread &t : readers)
            if (t.joinable()) t.join();
        if (fd >= 0) ::close(fd);
        fd = -1;
    }
};

// The output side: device buffers the simulator fills in turn, a thread that copies them into page-locked buffers, a thread that writes those.
struct OutPipe {
    static constexpr uint64_t kDev = 3, kStage = 3;
    static constexpr size_t kChunk = 64u << 20;
    TextOut out;
    DevBuffer dev[kDev];
    size_t dev_len[kDev] = {0, 0, 0};
    HostArray stage[kStage];
    size_t stage_len[kStage] = {0, 0, 0};
    uint64_t filled = 0, drained = 0, staged = 0, written = 0;       // texts handed in / downloaded; chunks downloaded / written
    bool closing = false, downloader_done = false, failed = false, started = false;
    std::mutex m;
    std::condition_variable cv;
    std::thread downloader, writer;
    StageTime t_download, t_stage, t_write, t_dev;
    bool open(const std::string &path) {                  // an empty path: stdout
        if (!path.empty() && !out.open(path)) return false;
        downloader = std::thread([this] { download(); });
        writer = std::thread([this] { write(); });
        started = true;
        return true;
    }
    void fail() {
        std::lock_guard<std::mutex> lock(m);
        failed = true;
        cv.notify_all();
    }
    // the device buffer the next text goes to (the caller may enlarge it), once its last text has been downloaded; nullptr after an error
    DevBuffer *begin() {
        const auto t0 = Clock::now();
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&] { return filled - drained < kDev || failed; });
        t_dev.add(t0);
        return failed ? nullptr : &dev[filled % kDev];
    }
    void submit(size_t bytes) {
        std::lock_guard<std::mutex> lock(m);
        dev_len[filled % kDev] = bytes;
        ++filled;
        cv.notify_all();
    }
    void download() {
        void *stream = nullptr;
        if (!check(rsq_stream_create(0, &stream), "stream")) return fail();
        for (;;) {
            uint64_t k;
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [&] { return drained < filled || closing || failed; });
                if (failed || drained == filled) break;
                k = drained % kDev;
            }
            for (size_t done = 0; done < dev_len[k]; done += kChunk) {
                const size_t n = std::min(kChunk, dev_len[k] - done);
                uint64_t j;
                {
                    const auto t0 = Clock::now();
                    std::unique_lock<std::mutex> lock(m);
                    cv.wait(lock, [&] { return staged - written < kStage || failed; });
                    t_stage.add(t0);
                    if (failed) break;
                    j = staged % kStage;
                }
                const auto t0 = Clock::now();
                if (!stage[j].ensure(kChunk) || !check(rsq_dev_copy_on(0, stage[j].p, static_cast<const char *>(dev[k].p) + done, n, 1, stream), "download")) return finish_download(true);
                t_download.add(t0);
                std::lock_guard<std::mutex> lock(m);
                stage_len[j] = n;
                ++staged;
                cv.notify_all();
            }
            std::lock_guard<std::mutex> lock(m);
            ++drained;
            cv.notify_all();
        }
        rsq_stream_destroy(0, stream);
        finish_download(false);
    }
    void finish_download(bool error) {
        std::lock_guard<std::mutex> lock(m);
        failed = failed || error;
        downloader_done = true;
        cv.notify_all();
    }
    void write() {
        for (;;) {
            uint64_t j;
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [&] { return written < staged || downloader_done || failed; });
                if (failed || written == staged) break;
                j = written % kStage;
            }
            const auto t0 = Clock::now();
            out.write(stage[j].as<char>(), stage_len[j]);
            t_write.add(t0);
            std::lock_guard<std::mutex> lock(m);
            ++written;
            if (!out.good()) failed = true;
            cv.notify_all();
        }
    }
    void close() {
        if (started) {
            {
                std::lock_guard<std::mutex> lock(m);
                closing = true;
                cv.notify_all();
            }
            downloader.join();
            writer.join();
            started = false;
        }
        out.close();
    }
    bool good() const { return !failed && out.good(); }
};

// a file that can be read at offsets by several threads: regular, and neither gzip nor bzip2 by its first bytes
int open_plain_file(const std::string &path, uint64_t &size) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return -1;
    struct stat st;
    unsigned char magic[3] = {0, 0, 0};
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || pread(fd, magic, 3, 0) < 0 || (magic[0] == 0x1f && magic[1] == 0x8b) || (magic[0] == 'B' && magic[1] == 'Z' && magic[2] == 'h')) {
        ::close(fd);
        return -1;
    }
    size = (uint64_t)st.st_size;
    return fd;
}

int seq_to_illumina(const Args &a) {
    rsq_profile *prof = nullptr;
    rsq_sim *sim = nullptr;
    const bool trace = a.has("traceStages");
    bool ok = load_profile(a, &prof);
    const double at_profile = seconds_since(g_process_start);
    const uint64_t seed = ok ? get_seed(a) : 0;
    ok = ok && check(rsq_sim_create(prof, nullptr, 0, &sim), "Could not set up the simulator") &&
         check(rsq_sim_prepare(sim, seed, 0, 0.0, 0, "", nullptr), "Preparation failed");
    const double at_prepared = seconds_since(g_process_start);
    TextIn fin;                                              // stdin / stdout without -i / -o (main.cpp:1009-1021)
    OutPipe fout;
    uint64_t plain_size = 0;
    int plain_fd = -1;
    if (ok && a.has("input")) {
        plain_fd = open_plain_file(a.get("input"), plain_size);
        if (plain_fd < 0 && !fin.open(a.get("input"))) {
            ERR("Could not open '" << a.get("input") << "' for reading.");
            ok = false;
        }
    }
    if (ok && !fout.open(a.get("output", ""))) {
        ERR("Could not open '" << a.get("output") << "' for writing.");
        ok = false;
    }
    if (ok) {
        INFO("Starting read generation");
        const unsigned hw = std::thread::hardware_concurrency();
        // reader threads of a plain file: one copies about 6 GB/s out of the page cache (--readThreads overrides; a stream has one reader whatever it says)
        uint32_t n_readers = std::max(1u, std::min(hw > 3 ? hw - 3 : 1u, 6u));
        for (const char *name : {"readThreads", "parseThreads"})
            if (a.has(name)) n_readers = (uint32_t)std::max(1, atoi(a.get(name).c_str()));
        // blocks of 48 MB, up to eight of them in a call (--blockKB / --batchBlocks: the tests make them small)
        const size_t block_bytes = (size_t)std::min(1 << 20, std::max(1, a.has("blockKB") ? atoi(a.get("blockKB").c_str()) : 48 << 10)) << 10;
        uint32_t batch_blocks = (uint32_t)std::max(1, a.has("batchBlocks") ? atoi(a.get("batchBlocks").c_str()) : 8);
        while (batch_blocks > 1 && batch_blocks * block_bytes > ((size_t)3 << 30)) --batch_blocks;      // a call takes less than 4 GB of text
        InPipe in(block_bytes, plain_fd >= 0 ? n_readers : 1);
        if (plain_fd >= 0) in.start_file(plain_fd, plain_size, n_readers);
        else in.start_stream(fin);
        void *stream = nullptr;
        ok = check(rsq_stream_create(0, &stream), "stream");
        DevBuffer joined[2];                                  // the text of a call: what the call before left over (it lies in the other one), then the blocks
        int next_joined = 0;
        const char *rest = nullptr;                           // the start of a record whose end the next block holds
        size_t rest_len = 0;
        uint64_t records = 0, next_report = 0, calls = 0;
        StageTime t_wait, t_join, t_call;
        double at_first_block = 0;
        const auto t_all = Clock::now();
        bool last = false;
        for (uint64_t b = 0; ok && !last;) {
            auto t0 = Clock::now();
            InPipe::Slot *slot = in.take(b);
            t_wait.add(t0);
            if (!slot) {
                ok = false;
                break;
            }
            if (!b) at_first_block = seconds_since(g_process_start);
            t0 = Clock::now();
            DevBuffer &j = joined[next_joined];
            next_joined ^= 1;
            ok = j.ensure(rest_len + batch_blocks * block_bytes + 16) && (!rest_len || check(rsq_dev_copy_on(0, j.p, rest, rest_len, 2, stream), "copy"));
            char *text = static_cast<char *>(j.p);
            size_t len = rest_len;
            for (uint32_t k = 0; ok && slot; ++k) {
                ok = check(rsq_dev_copy_on(0, text + len, slot->dev.p, slot->len, 2, stream), "copy");
                len += slot->len;
                last = slot->last;
                in.release(b++);
                slot = last || k + 1 == batch_blocks ? nullptr : in.take(b, false);
            }
            t_join.add(t0);
            t0 = Clock::now();
            DevBuffer *o = ok ? fout.begin() : nullptr;
            ok = ok && o;
            size_t out_len = 0, used = 0;
            uint64_t n = 0;
            for (int attempt = 0; ok && attempt < 2; ++attempt) {
                ok = o->ensure(std::max(out_len + out_len / 8, len + len / 8) + 4096);
                if (!ok) break;
                const int rc = rsq_sim_error_model_fasta(sim, records, text, len, last ? 1 : 0, static_cast<char *>(o->p), o->cap, &out_len, &n, &used, stream);
                if (rc == RSQ_ENOSPC && !attempt) continue;
                if (rc == RSQ_EIO) {                              // the reference's complaint about a record (Simulator.cpp:2423-2485)
                    ERR(rsq_last_error());
                    ok = false;
                } else ok = check(rc, "Simulation failed");
                break;
            }
            t_call.add(t0);
            if (!ok) break;
            ++calls;
            if (out_len) fout.submit(out_len);
            records += n;                                         // = the index of the next record in the input (it selects the records' random streams)
            rest = text + used;
            rest_len = len - used;
            if (records >= next_report) {
                INFO("Generated " << records << " reads.");
                next_report = records + 1000000;
            }
        }
        if (trace)
            fprintf(stderr,
                    "stages: profile loaded at %.3f s of the process, simulator prepared at %.3f, first block on the device at %.3f; the simulator side took %.3f s for %llu records in %llu calls: "
                    "waiting for blocks %.3f, putting blocks together %.3f, device calls %.3f (of these waiting for a free output buffer %.3f); readers (summed over %u threads): reading %.3f, "
                    "uploads %.3f, waiting for a slot %.3f; downloads %.3f (+ %.3f waiting for a free buffer); writing %.3f\n",
                    at_profile, at_prepared, at_first_block, seconds_since(t_all), (unsigned long long)records, (unsigned long long)calls, t_wait.s(), t_join.s(), t_call.s(), fout.t_dev.s(),
                    (unsigned)in.readers.size(), in.t_read.s(), in.t_upload.s(), in.t_slot.s(), fout.t_download.s(), fout.t_stage.s(), fout.t_write.s());
        in.join();
        ok = ok && !in.failed && !fin.failed;
        if (ok && !records) {
            ERR(a.get("input", "stdin") << " does not contain any sequences.");
            ok = false;
        }
        if (!ok) fout.fail();
        rsq_stream_destroy(0, stream);
    }
    fin.close();
    fout.close();
    ok = ok && fout.good();
    rsq_sim_free(sim);
    rsq_profile_free(prof);
    if (!ok) {
        ERR("An error occurred in the process: Terminating simulation");
        if (a.has("output")) remove(a.get("output").c_str());
        return 1;
    }
    INFO("Simulation finished succesfully");
    return 0;
}

const char *kUsage =
    "\nProgram: reseq (REal SEQuence replicator) -- MI355X simulation stage\n"
    "Usage:  reseq <command> [options]\n"
    "Commands:\n"
    "  illuminaPE\t\tsimulates illumina paired-end data from a fitted profile (-s) and a reference (-R)\n"
    "  seqToIllumina\t\tapplies illumina quality and error model to input sequences (alias: replaceQuals)\n"
    "General: -j/--threads N (ignored: the GPU does the work), --verbosity 0-4, --version, -h,\n"
    "         --rsqOption name:value[,...] (measurement switches of libreseq_amd, include/reseq_amd.h rsq_set_option; results never depend on them)\n";

}  // namespace

// reseq queryProfile -s <profile> [-r <ref.fa>] [--maxLenDeletion] [--maxReadLength] [--refSeqBias <file|->] (main.cpp:481-610)
int query_profile(const Args &a) {
    // -r / -s carry the long names of the illuminaPE mode in this parser; queryProfile calls them ref and stats
    const std::string stats = a.get("stats", a.get("statsIn")), ref_path = a.get("ref", a.get("refIn"));
    if (stats.empty()) {
        ERR("stats option is mandatory.");
        return 1;
    }
    if (a.has("refSeqBias") && ref_path.empty()) {
        ERR("ref option is mandatory if refSeqBias is specified.");
        return 1;
    }
    if (a.has("dumpArchiveLayout")) {                             // not in the reference: diagnosis of a .reseq / .reseq.ipf pair this build cannot read
        size_t need = 0;
        const std::string ipf = a.get("probabilitiesIn");
        if (!check(rsq_profile_archive_layout(stats.c_str(), ipf.empty() ? nullptr : ipf.c_str(), nullptr, 0, &need), "Could not lay out the profile archives")) return 1;
        std::vector<char> text(need);
        rsq_profile_archive_layout(stats.c_str(), ipf.empty() ? nullptr : ipf.c_str(), text.data(), text.size(), &need);
        std::cout << text.data();
        return 0;
    }
    rsq_profile *prof = nullptr;
    rsq_ref *ref = nullptr;
    INFO("Reading reference sequence biases from " << stats);
    bool ok = check(rsq_profile_load(stats.c_str(), &prof), "Could not load profile");
    if (ok && !ref_path.empty()) {
        INFO("Reading reference from " << ref_path);
        ok = check(rsq_ref_load_fasta(ref_path.c_str(), &ref), "Could not load reference");
    }
    bool no_output = true;
    if (ok && a.has("maxLenDeletion")) {
        uint32_t v = 0;
        rsq_profile_max_len_deletion(prof, &v);
        std::cout << "maxLenDeletion: " << v << std::endl;
        no_output = false;
    }
    if (ok && a.has("maxReadLength")) {
        uint32_t v = 0;
        rsq_profile_max_read_length(prof, &v);
        std::cout << "maxReadLength: " << v << std::endl;
        no_output = false;
    }
    if (ok && a.has("refSeqBias")) {                              // FragmentDistributionStats::WriteRefSeqBias (FragmentDistributionStats.cpp:3643-3670)
        size_t n = 0;
        uint32_t n_seqs = 0;
        rsq_profile_ref_seq_bias(prof, nullptr, 0, &n);
        rsq_ref_num_sequences(ref, &n_seqs);
        if (n != n_seqs) {
            ERR("Reference and reseq file do not match. The reference has " << n_seqs << " sequences and the reseq file has biases for " << n << " sequences");
            ok = false;
        } else {
            std::vector<double> bias(n ? n : 1);
            rsq_profile_ref_seq_bias(prof, bias.data(), bias.size(), &n);
            std::ostringstream text;
            for (uint32_t i = 0; i < n_seqs; ++i) {
                char name[4096];
                rsq_ref_sequence_name(ref, i, name, sizeof name);
                text << name << '\t' << bias[i] << '\n';
            }
            const std::string file = a.get("refSeqBias") == "-" ? "" : a.get("refSeqBias");
            if (file.empty()) {
                INFO("Writing reference sequence biases to stdout");
                std::cout << text.str();
            } else {
                INFO("Writing reference sequence biases to " << file);
                std::ofstream f(file);
                if (!f) {
                    ERR("Unable to open reference bias file " << file);
                    ok = false;
                } else f << text.str();
            }
        }
        no_output = false;
    }
    rsq_ref_free(ref);
    rsq_profile_free(prof);
    if (ok && no_output) {
        ERR("No output option was selected.");
        return 1;
    }
    return ok ? 0 : 1;
}

// reseq replaceN -r <refIn.fa> -R <refSim.fa> [--seed] (main.cpp:611-692)
int replace_n(const Args &a) {
    if (!a.has("refIn")) {
        ERR("refIn option is mandatory.");
        return 1;
    }
    if (!a.has("refSim")) {
        ERR("refSim option is mandatory.");
        return 1;
    }
    INFO("Reading reference from " << a.get("refIn"));
    INFO("Writing reference without N to " << a.get("refSim"));
    rsq_ref *ref = nullptr;
    bool ok = check(rsq_ref_load_fasta(a.get("refIn").c_str(), &ref), "Could not load reference");
    ok = ok && check(rsq_ref_replace_n(ref, get_seed(a)), "ReplaceN") && check(rsq_ref_write_fasta(ref, a.get("refSim").c_str()), "Could not write reference");
    rsq_ref_free(ref);
    if (ok) INFO("Finished replacing N's.");
    return ok ? 0 : 1;
}

int main(int argc, char **argv) {
    std::string command;
    int cmd_at = 0;
    for (int i = 1; i < argc; ++i)
        if (argv[i][0] != '-') {
            command = argv[i];
            cmd_at = i;
            break;
        } else if (!strcmp(argv[i], "-j") || !strcmp(argv[i], "--threads") || !strcmp(argv[i], "--verbosity")) ++i;
    // general options may stand before or after the command
    std::vector<char *> rest{argv[0]};
    for (int i = 1; i < argc; ++i)
        if (i != cmd_at) rest.push_back(argv[i]);
    Args a;
    if (!parse((int)rest.size(), rest.data(), 1, a)) return 1;
    if (a.has("verbosity")) g_verbosity = atoi(a.get("verbosity").c_str());
    if (a.has("rsqOption")) {                                 // measurement switches of the library (rsq_set_option): name:value[,name:value...]
        std::stringstream list(a.get("rsqOption"));
        std::string item;
        while (std::getline(list, item, ',')) {
            const size_t colon = item.find(':');
            char *end = nullptr;
            const long long v = colon == std::string::npos ? 1 : strtoll(item.c_str() + colon + 1, &end, 10);
            if ((end && *end) || !check(rsq_set_option(item.substr(0, colon).c_str(), v), "--rsqOption")) return 1;
        }
    }
    if (a.has("version")) {
        std::cerr << rsq_version() << " (stands in for ReSeq version 1.1 simulation stage)" << std::endl;
        return 0;
    }
    if (command.empty() || a.has("help")) {
        std::cerr << kUsage << std::endl;
        return command.empty() && !a.has("help") ? 1 : 0;
    }
    {                                                         // numeric options that several commands share
        uint64_t u = 0;
        double d = 0;
        if ((a.has("seed") && !parse_u64(a, "seed", u)) || (a.has("errorMutliplier") && !parse_double(a, "errorMutliplier", d)) ||
            (a.has("ipfPrecision") && !parse_double(a, "ipfPrecision", d)))
            return 1;
    }
    if (command == "illuminaPE") return illumina_pe(a);
    if (command == "seqToIllumina" || command == "replaceQuals") return seq_to_illumina(a);
    if (command == "replaceN") return replace_n(a);
    if (command == "queryProfile") return query_profile(a);
    if (command == "test") {
        ERR("command '" << command << "' is not part of this build (simulation stage only)");
        return 1;
    }
    ERR("unknown command '" << command << "'");
    std::cerr << kUsage << std::endl;
    return 1;
}
