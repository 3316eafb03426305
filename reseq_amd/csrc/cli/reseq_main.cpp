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
#include <zlib.h>

#include <algorithm>

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

// One FASTA record of seqToIllumina's input: "{id} {1|2};{fragment length};{dominant errors};{error rates}" (Simulator.cpp:2423-2485)
struct RecordFields {
    size_t id_len = 0, dom_at = 0, rate_at = 0;
    uint8_t seg = 0;
    uint32_t frag_len = 0;
};
bool parse_record(const char *header, size_t header_len, size_t L, RecordFields &r) {
    auto text = [&]() { return std::string(header, header_len); };
    if (header_len <= 2 * L + 2) {
        ERR("Read description is too short to contain systematic error information and a sequence id: " << text());
        return false;
    }
    size_t end = header_len - 2 * L - 3;
    if (header[end + 1] != ';' || header[end + 2 + L] != ';') {
        ERR("The two systematic error entries are not separated by a semicolon from themselves or the rest of the ReSeq information: " << text());
        return false;
    }
    r.dom_at = end + 2;
    r.rate_at = header_len - L;
    while (end && header[end] != ' ') --end;
    if (!end) {
        ERR("No sequence id found that is separated by a space from the ReSeq information: " << text());
        return false;
    }
    r.id_len = end;
    if (header[end + 1] == '1') r.seg = 0;
    else if (header[end + 1] == '2') r.seg = 1;
    else {
        ERR("Template segment is " << header[end + 1] << " not 1 or 2: " << text());
        return false;
    }
    if (header[end + 2] != ';') {
        ERR("The template segment and fragment length are not separated by a semicolon: " << text());
        return false;
    }
    const size_t fl_at = end + 3, fl_end = header_len - 2 * L - 2;
    uint64_t v = 0;
    bool digits = fl_end > fl_at;
    for (size_t k = fl_at; k < fl_end && digits; ++k) {
        digits = header[k] >= '0' && header[k] <= '9';
        v = v * 10 + (uint64_t)(header[k] - '0');
    }
    if (!digits) {
        ERR("Fragment length '" << std::string(header + fl_at, fl_end > fl_at ? fl_end - fl_at : 0) << "' is not a pure integer: " << text());
        return false;
    }
    r.frag_len = (uint32_t)v;
    return true;
}

struct CodeTable {
    uint8_t code[256];
    CodeTable() {
        memset(code, 4, sizeof code);
        code[(uint8_t)'A'] = code[(uint8_t)'a'] = 0;
        code[(uint8_t)'C'] = code[(uint8_t)'c'] = 1;
        code[(uint8_t)'G'] = code[(uint8_t)'g'] = 2;
        code[(uint8_t)'T'] = code[(uint8_t)'t'] = 3;
    }
};

struct HostArray {                    // page-locked, grow-only
    void *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t n, size_t keep = 0) {
        if (n <= cap) return true;
        void *q = nullptr;
        const size_t want = std::max(n, cap + cap / 2);
        if (!check(rsq_host_alloc(want, &q), "host buffer")) return false;
        if (p && keep) memcpy(q, p, std::min(keep, cap));
        if (p) rsq_host_free(p);
        p = q;
        cap = want;
        return true;
    }
    template <class T>
    T *as() { return static_cast<T *>(p); }
    ~HostArray() {
        if (p) rsq_host_free(p);
    }
};

// One block of the input text and what a parser thread makes of it: the records packed for rsq_sim_error_model_fastq, in runs of one
// template length each (a launch serves one length).  Slots go round: FREE -> TEXT (reader) -> PARSING -> PARSED (a parser) -> FREE (consumer).
struct ParseSlot {
    enum State { FREE, TEXT, PARSING, PARSED } state = FREE;
    std::vector<char> text;           // whole records: ends behind the last line of a record
    size_t text_len = 0;
    bool failed = false;
    struct Run {
        size_t first, n, L, base_at, id_first;      // records [first, first + n) of the slot, their bases from base_at on
    };
    std::vector<Run> runs;
    size_t n = 0, bases = 0, id_bytes = 0;
    HostArray seqs, dom, rate, seg, fl, ids, id_off;   // id_off: per run n + 1 offsets relative to the run's first id, stored at first + run index
    bool reserve(size_t records, size_t n_bases, size_t n_id_bytes) {
        return seqs.ensure(n_bases, bases) && dom.ensure(n_bases, bases) && rate.ensure(n_bases, bases) && seg.ensure(records, n) && fl.ensure(records * 4, n * 4) &&
               id_off.ensure((records + runs.size() + 2) * 8, (n + runs.size() + 1) * 8) && ids.ensure(n_id_bytes, id_bytes);
    }
    // parses text[0, text_len) -- complete records -- into the arrays
    bool parse() {
        static const CodeTable t;
        runs.clear();
        n = bases = id_bytes = 0;
        const char *p = text.data(), *end = p + text_len;
        std::string joined;                                   // a sequence that is wrapped over several lines
        while (p < end) {
            while (p < end && (*p == '\n' || *p == '\r')) ++p;
            if (p == end) break;
            if (*p != '>') {                                   // text in front of the first header: SeqAn skips nothing here, but such a file is not FASTA
                ERR("sequence data without a header line in the input");
                return false;
            }
            const char *h0 = p + 1, *h1 = (const char *)memchr(h0, '\n', (size_t)(end - h0));
            if (!h1) h1 = end;
            const char *line = h1 < end ? h1 + 1 : end;
            size_t header_len = (size_t)(h1 - h0);
            if (header_len && h0[header_len - 1] == '\r') --header_len;
            const char *seq = line;
            size_t L = 0;
            joined.clear();
            bool single = true;
            while (line < end && *line != '>') {
                const char *le = (const char *)memchr(line, '\n', (size_t)(end - line));
                if (!le) le = end;
                size_t len = (size_t)(le - line);
                if (len && line[len - 1] == '\r') --len;
                if (L == 0 && joined.empty()) {
                    seq = line;
                    L = len;
                } else if (len) {
                    if (single) {
                        joined.assign(seq, L);
                        single = false;
                    }
                    joined.append(line, len);
                }
                line = le < end ? le + 1 : end;
            }
            if (!single) {
                seq = joined.data();
                L = joined.size();
            }
            RecordFields r;
            if (!parse_record(h0, header_len, L, r)) return false;
            if (runs.empty() || runs.back().L != L) runs.push_back(Run{n, 0, L, bases, id_bytes});
            if (!reserve(n + 1, bases + L, id_bytes + r.id_len + 8)) return false;
            uint8_t *sq = seqs.as<uint8_t>() + bases, *dm = dom.as<uint8_t>() + bases, *rt = rate.as<uint8_t>() + bases;
            uint8_t any = 0;
            for (size_t k = 0; k < L; ++k) {
                sq[k] = t.code[(uint8_t)seq[k]];
                any |= sq[k];
                dm[k] = t.code[(uint8_t)h0[r.dom_at + k]];
                int v = (uint8_t)h0[r.rate_at + k] - 33;                 // Simulator.cpp:2439-2442
                if (v > 86) v += v - 86;
                rt[k] = (uint8_t)v;
            }
            if (any & 4u) {
                ERR("input sequences must not contain N: " << std::string(h0, r.id_len));
                return false;
            }
            seg.as<uint8_t>()[n] = r.seg;
            fl.as<uint32_t>()[n] = r.frag_len;
            memcpy(ids.as<char>() + id_bytes, h0, r.id_len);
            Run &run = runs.back();
            uint64_t *off = id_off.as<uint64_t>() + run.first + (runs.size() - 1);      // n + 1 offsets per run
            if (!run.n) off[0] = 0;
            id_bytes += r.id_len;
            off[run.n + 1] = id_bytes - run.id_first;
            ++run.n;
            ++n;
            bases += L;
            p = line;
        }
        return true;
    }
};

// reader thread (cuts the text into blocks of whole records), parser threads, and the consumer's view of the slots in input order
struct ParsePipeline {
    static constexpr size_t kBlockBytes = 24u << 20;
    TextIn &in;
    std::vector<ParseSlot> slots;
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::thread> workers;
    uint64_t blocks_read = 0, next_take = 0;
    bool eof = false, failed = false, abort = false, any = false;
    ParsePipeline(TextIn &input, size_t parsers) : in(input), slots(parsers + 2) {
        workers.emplace_back([this] { read_blocks(); });
        for (size_t k = 0; k < parsers; ++k) workers.emplace_back([this] { parse_blocks(); });
    }
    void fail() {
        std::lock_guard<std::mutex> lock(m);
        failed = true;
        cv.notify_all();
    }
    void read_blocks() {
        std::vector<char> carry;
        for (uint64_t b = 0;; ++b) {
            ParseSlot *slot = &slots[b % slots.size()];
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [&] { return slot->state == ParseSlot::FREE || abort || failed; });
                if (abort || failed) break;
            }
            std::vector<char> &text = slot->text;
            text.resize(std::max(text.size(), carry.size() + kBlockBytes + 1));
            memcpy(text.data(), carry.data(), carry.size());
            size_t have = carry.size();
            carry.clear();
            bool end_of_input = false;
            for (;;) {                                        // until the block holds the start of another record behind its first one, or the input ends
                if (have + (1u << 20) > text.size()) text.resize(text.size() * 2);
                const int got = in.read_raw(text.data() + have, (unsigned)std::min<size_t>(text.size() - have, 1u << 30));
                if (got < 0) return fail();
                if (!got) {
                    end_of_input = true;
                    break;
                }
                have += (size_t)got;
                if (have >= kBlockBytes) {
                    size_t cut = have;                        // the last "\n>" of the block
                    while (cut > 1 && !(text[cut - 1] == '>' && text[cut - 2] == '\n')) --cut;
                    if (cut > 1) {
                        carry.assign(text.begin() + (ptrdiff_t)(cut - 1), text.begin() + (ptrdiff_t)have);
                        have = cut - 1;
                        break;
                    }
                }
            }
            std::lock_guard<std::mutex> lock(m);
            slot->text_len = have;
            slot->state = ParseSlot::TEXT;
            ++blocks_read;
            if (end_of_input) eof = true;
            cv.notify_all();
            if (end_of_input) break;
        }
        std::lock_guard<std::mutex> lock(m);
        eof = true;
        cv.notify_all();
    }
    void parse_blocks() {
        for (;;) {
            ParseSlot *slot = nullptr;
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [&] {
                    for (ParseSlot &s : slots)
                        if (s.state == ParseSlot::TEXT) {
                            slot = &s;
                            return true;
                        }
                    return abort || failed || (eof && true);
                });
                if (!slot) {
                    if (abort || failed) return;
                    bool pending = false;                     // the reader is done: anything left to parse?
                    for (ParseSlot &s : slots) pending = pending || s.state == ParseSlot::TEXT;
                    if (!pending) return;
                    continue;
                }
                slot->state = ParseSlot::PARSING;
            }
            const bool ok = slot->parse();
            std::lock_guard<std::mutex> lock(m);
            slot->failed = !ok;
            slot->state = ParseSlot::PARSED;
            if (!ok) failed = true;
            cv.notify_all();
        }
    }
    // the next block in input order; nullptr at the end of the input or after an error (`failed`)
    ParseSlot *take() {
        std::unique_lock<std::mutex> lock(m);
        ParseSlot *slot = &slots[next_take % slots.size()];
        cv.wait(lock, [&] { return failed || (slot->state == ParseSlot::PARSED && true) || (eof && next_take >= blocks_read); });
        if (failed || slot->state != ParseSlot::PARSED) return nullptr;
        if (slot->n) any = true;
        return slot;
    }
    void release() {
        std::lock_guard<std::mutex> lock(m);
        slots[next_take % slots.size()].state = ParseSlot::FREE;
        ++next_take;
        cv.notify_all();
    }
    void join() {
        {
            std::lock_guard<std::mutex> lock(m);
            abort = true;
            cv.notify_all();
        }
        for (std::thread &t : workers)
            if (t.joinable()) t.join();
    }
};

// ---- seqToIllumina as a pipeline (Simulator::SimulateErrorModelOnly, Simulator.cpp:2900-3014: reader, ErrorModelOnlyThread :2514-2560,
// ordered output :184-213): a reader thread cuts the FASTA text into blocks of whole records, parser threads pack the blocks into page-locked
// arrays, the main thread takes the blocks in input order, uploads their records, runs the error model and the FASTQ formatter on the
// device (rsq_sim_error_model_fastq) and hands the text to the writer thread of AsyncOut.  Device and host buffers are reused.
int seq_to_illumina(const Args &a) {
    rsq_profile *prof = nullptr;
    rsq_sim *sim = nullptr;
    bool ok = load_profile(a, &prof);
    const uint64_t seed = ok ? get_seed(a) : 0;
    ok = ok && check(rsq_sim_create(prof, nullptr, 0, &sim), "Could not set up the simulator") &&
         check(rsq_sim_prepare(sim, seed, 0, 0.0, 0, "", nullptr), "Preparation failed");
    TextIn fin;                                              // stdin / stdout without -i / -o (main.cpp:1009-1021)
    AsyncOut fout;
    if (ok && a.has("input") && !fin.open(a.get("input"))) {
        ERR("Could not open '" << a.get("input") << "' for reading.");
        ok = false;
    }
    if (ok && !fout.open(a.get("output", ""))) {
        ERR("Could not open '" << a.get("output") << "' for writing.");
        ok = false;
    }
    if (ok) {
        INFO("Starting read generation");
        const unsigned hw = std::thread::hardware_concurrency();
        // parser threads: six keep the single writer of the output file busy (buffered writes into one file serialise on its inode: 6 GB/s); more were
        // measured slower -- every parser slot owns page-locked arrays, whose allocation is what a run of 20 M records waits for (6 threads 2.5 s, 16: 3.2 s,
        // 32: 5.0 s; the marginal rate beyond the start-up is 25 M reads/s either way).  --parseThreads overrides.
        uint32_t parsers = std::max(1u, std::min(hw > 2 ? hw - 2 : 1u, 6u));
        if (a.has("parseThreads")) parsers = (uint32_t)std::max(1, atoi(a.get("parseThreads").c_str()));
        ParsePipeline pipe(fin, parsers);
        DevBuffer d_seqs, d_dom, d_rate, d_seg, d_fl, d_ids, d_off, d_text;
        uint64_t written = 0, next_report = 0;
        // --traceStages: where the consuming thread's time goes (waiting for a parsed block, uploads, the device call, handing the text to the writer)
        const bool trace = a.has("traceStages");
        double t_wait = 0, t_up = 0, t_dev = 0, t_push = 0;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(now() - t0).count(); };
        const auto t_all = now();
        while (ok) {
            auto t0 = now();
            ParseSlot *b = pipe.take();
            t_wait += since(t0);
            if (!b) break;
            for (size_t r = 0; ok && r < b->runs.size(); ++r) {
                const ParseSlot::Run &run = b->runs[r];
                const size_t n = run.n, L = run.L;
                const size_t run_id_bytes = (size_t)b->id_off.as<uint64_t>()[run.first + r + n];
                t0 = now();
                ok = d_seqs.ensure(n * L + 8) && d_dom.ensure(n * L + 8) && d_rate.ensure(n * L + 8) && d_seg.ensure(n) && d_fl.ensure(n * 4) && d_ids.ensure(run_id_bytes + 8) &&
                     d_off.ensure((n + 1) * 8) && check(rsq_dev_upload(0, d_seqs.p, b->seqs.as<uint8_t>() + run.base_at, n * L), "upload") &&
                     check(rsq_dev_upload(0, d_dom.p, b->dom.as<uint8_t>() + run.base_at, n * L), "upload") &&
                     check(rsq_dev_upload(0, d_rate.p, b->rate.as<uint8_t>() + run.base_at, n * L), "upload") &&
                     check(rsq_dev_upload(0, d_seg.p, b->seg.as<uint8_t>() + run.first, n), "upload") &&
                     check(rsq_dev_upload(0, d_fl.p, b->fl.as<uint32_t>() + run.first, n * 4), "upload") &&
                     check(rsq_dev_upload(0, d_ids.p, b->ids.as<char>() + run.id_first, run_id_bytes + 1), "upload") &&
                     check(rsq_dev_upload(0, d_off.p, b->id_off.as<uint64_t>() + run.first + r, (n + 1) * 8), "upload");
                t_up += since(t0);
                t0 = now();
                size_t len = 0;
                for (int attempt = 0; ok && attempt < 2; ++attempt) {
                    ok = d_text.ensure(std::max(len + len / 8, n * (2 * L + 96) + run_id_bytes) + 64);
                    if (!ok) break;
                    const int rc = rsq_sim_error_model_fastq(sim, written, n, (uint32_t)L, (const uint8_t *)d_seqs.p, (const uint8_t *)d_seg.p, (const uint32_t *)d_fl.p,
                                                             (const uint8_t *)d_dom.p, (const uint8_t *)d_rate.p, (const char *)d_ids.p, (const uint64_t *)d_off.p,
                                                             (char *)d_text.p, d_text.cap, &len, nullptr);
                    if (rc == RSQ_ENOSPC && !attempt) continue;
                    ok = check(rc, "Simulation failed");
                    break;
                }
                t_dev += since(t0);
                t0 = now();
                ok = ok && fout.push(d_text, len);
                t_push += since(t0);
                written += n;                                     // = the index of the next record in the input (it selects the records' random streams)
            }
            pipe.release();                                       // the records are on the device: the slot may take the next block
            if (ok && written >= next_report) {
                INFO("Generated " << written << " reads.");
                next_report = written + 1000000;
            }
        }
        if (trace)
            fprintf(stderr, "stages of the consuming thread: %.3f s in all for %llu records; waiting for parsed blocks %.3f, uploads %.3f, device (error model + text) %.3f, text to the writer %.3f\n",
                    since(t_all), (unsigned long long)written, t_wait, t_up, t_dev, t_push);
        pipe.join();
        ok = ok && !pipe.failed && !fin.failed;
        if (ok && !pipe.any) {
            ERR(a.get("input", "stdin") << " does not contain any sequences.");
            ok = false;
        }
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
