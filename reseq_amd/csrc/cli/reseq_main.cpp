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
#include <array>
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
                                      "dumpArchiveLayout", "traceStages", "hostGzip"};

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

// FASTQ text files: gzip / bzip2 when the name ends in .gz / .bz2 (SeqAn's SeqFileOut picks the format the same way); no file = stdout
struct TextOut {
    rsq::textio::Writer w;
    bool failed = false;
    bool open(const std::string &path, bool as_it_comes = false) {      // as_it_comes: the bytes are .gz members already (made on the device), whatever the name says
        try {
            return as_it_comes ? w.open_plain(path) : w.open(path);
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
using Clock = std::chrono::steady_clock;
const Clock::time_point g_process_start = Clock::now();
double seconds_since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }
// --traceStages: seconds of the process at which a stage was reached
struct StageTrace {
    bool on = false;
    std::string line;
    void at(const char *what) {
        if (!on) return;
        char b[96];
        snprintf(b, sizeof b, "%s%s %.3f", line.empty() ? "" : ", ", what, seconds_since(g_process_start));
        line += b;
    }
    ~StageTrace() {
        if (on) fprintf(stderr, "stages (seconds of the process): %s\n", line.c_str());
    }
};

struct DevBuffer {
    void *p = nullptr;
    size_t cap = 0;
    int device = 0;
    bool ensure(size_t n) {
        if (n <= cap) return true;
        if (p) rsq_dev_free(device, p);
        p = nullptr;
        cap = 0;
        if (!check(rsq_dev_alloc(device, n, &p), "device allocation")) return false;
        cap = n;
        return true;
    }
    ~DevBuffer() {
        if (p) rsq_dev_free(device, p);
    }
};

// One output file with its own writer thread and two page-locked staging buffers: while the thread writes (and compresses) batch i,
// the simulator produces batch i+1 and its text is copied into the other buffer.
struct AsyncOut {
    TextOut out;
    std::string tail;                                   // written behind the last text when the file is closed (the BGZF end-of-file member)
    void *buf[2] = {nullptr, nullptr};
    size_t cap[2] = {0, 0}, len[2] = {0, 0};
    bool full[2] = {false, false};
    int next_fill = 0, next_write = 0;
    bool stop = false, started = false;
    std::mutex m;
    std::condition_variable cv;
    std::thread worker;
    bool open(const std::string &path, bool as_it_comes = false) {      // an empty path: stdout
        if (!path.empty() && !out.open(path, as_it_comes)) return false;
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
        for (size_t done = 0; done < bytes; done += kChunk)
            if (!push_chunk(d, done, std::min(kChunk, bytes - done))) return false;
        return out.good();
    }
    bool push_chunk(const DevBuffer &d, size_t done, size_t n) {
        {
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
            if (!check(rsq_dev_download(d.device, buf[k], static_cast<const char *>(d.p) + done, n), "download")) return false;
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
        if (!tail.empty() && out.good()) out.write(tail.data(), tail.size());
        out.close();
        for (int k = 0; k < 2; ++k)
            if (buf[k]) rsq_host_free(buf[k]);
        buf[0] = buf[1] = nullptr;
    }
    bool good() const { return out.good(); }
};

// the two files chunk by chunk in turn: while this thread waits for a free buffer of one file, the other file's writer has work (a file in /dev/shm takes 6 GB/s,
// whoever writes it: one file after the other would halve the rate)
bool flush_pair(DevBuffer &d1, size_t l1, DevBuffer &d2, size_t l2, AsyncOut &f1, AsyncOut &f2) {
    for (size_t done = 0; done < std::max(l1, l2); done += AsyncOut::kChunk) {
        if (done < l1 && !f1.push_chunk(d1, done, std::min(AsyncOut::kChunk, l1 - done))) return false;
        if (done < l2 && !f2.push_chunk(d2, done, std::min(AsyncOut::kChunk, l2 - done))) return false;
    }
    return f1.good() && f2.good();
}

// illuminaPE on several devices inside one process (Simulator::Simulate starts its own worker threads, Simulator.cpp:2830-2836, and hands them blocks, :2384-2401).
// N host threads, one simulator each on device (worker % devices): every worker prepares, simulates its contiguous share of the blocks with the text kept in device
// memory (rsq_sim_job_generate), the exclusive scan of the workers' text sizes gives every worker its place in the two files, and all workers write at once
// (rsq_sim_job_write).  No process group, no collective, no Python: the workers share nothing but the loaded profile and reference.  The files are byte for byte
// the single-device run's.
struct PeJob {
    const Args &a;
    rsq_profile *prof;
    rsq_ref *ref;
    uint64_t seed, num_reads;
    double coverage;
    int ref_bias_mode;
    std::string ref_bias_file, sys_read, sys_write, methylation, out1, out2, base_identifier;
};
struct Worker {
    int device = 0;
    rsq_sim *sim = nullptr;
    uint32_t lo = 1, hi = 1;
    uint64_t pairs = 0, bytes[2] = {0, 0};
    std::string error;                                     // rsq_last_error() is the calling thread's own: taken where the call failed
    bool fail(int rc, const char *what) {
        if (rc == RSQ_OK) return false;
        error = std::string(what) + ": " + rsq_last_error();
        return true;
    }
};
// runs f(worker index) on one thread per worker and waits for all; false if any worker reported an error (the first one is printed)
template <class F>
bool on_all_workers(std::vector<Worker> &workers, F &&f) {
    std::vector<std::thread> threads;
    for (size_t r = 1; r < workers.size(); ++r) threads.emplace_back([&, r] { f(r); });
    f(0);
    for (std::thread &t : threads) t.join();
    for (size_t r = 0; r < workers.size(); ++r)
        if (!workers[r].error.empty()) {
            ERR("worker " << r << " (device " << workers[r].device << "): " << workers[r].error);
            return false;
        }
    return true;
}
int illumina_pe_on_workers(const PeJob &job, int n_workers, int n_devices) {
    std::vector<Worker> workers((size_t)n_workers);
    for (int r = 0; r < n_workers; ++r) workers[(size_t)r].device = r % n_devices;
    INFO("Simulating with " << n_workers << " workers on " << std::min(n_workers, n_devices) << " device(s)");
    const bool gz = rsq::textio::has_suffix(job.out1, ".gz");
    auto release = [&] {
        for (Worker &w : workers) rsq_sim_free(w.sim);
    };
    // every worker: its simulator on its device
    bool ok = on_all_workers(workers, [&](size_t r) {
        Worker &w = workers[r];
        if (w.fail(rsq_sim_create(job.prof, job.ref, w.device, &w.sim), "Could not set up the simulator")) return;
        if (!job.ref_bias_file.empty() && w.fail(rsq_sim_set_ref_bias_file(w.sim, job.ref_bias_file.c_str()), "refBiasFile")) return;
        if (!job.methylation.empty() && w.fail(rsq_sim_read_methylation(w.sim, job.methylation.c_str()), "Could not read methylation file")) return;
    });
    std::string sys_read = job.sys_read;
    if (ok && !job.sys_write.empty()) {                       // main.cpp:351-397: one worker draws and writes the profile, all read it
        INFO("Writing systematic error profile to " << job.sys_write);
        ok = check(rsq_sim_create_sys_error_profile(workers[0].sim, job.seed, job.sys_write.c_str(), nullptr), "Could not write systematic error profile");
        if (!ok) remove(job.sys_write.c_str());
        sys_read = job.sys_write;
    }
    if (ok && job.a.has("stopAfterEstimation")) {
        release();
        return 0;
    }
    // The workers' block ranges: the rule of the N-process launcher (rsq_partition_blocks).  With a systematic-error profile from a file every worker runs the whole
    // pre-pass (the tracks come from the file); else the pre-pass is SHARDED like the launcher's (reseq_amd/sharding.py sharded_prepare, here in lock step between
    // joins): every worker plans, sums the coverage bias of its own chunks, all take the sum of the partial arrays (every entry is non-zero in one worker: exact),
    // every worker runs the systematic-error chains of its own positions, and the chain states at the shard borders travel worker to worker until none changes.
    auto ranges_from = [&](rsq_sim *sim) {
        uint32_t n_blocks = 0;
        if (!check(rsq_sim_block_weights(sim, nullptr, 0, &n_blocks), "block weights")) return false;
        std::vector<double> weights(n_blocks);
        std::vector<uint32_t> bounds((size_t)n_workers + 1);
        if (!check(rsq_sim_block_weights(sim, weights.data(), weights.size(), &n_blocks), "block weights") ||
            !check(rsq_partition_blocks(n_blocks, (uint32_t)n_workers, weights.data(), bounds.data()), "partition"))
            return false;
        for (int r = 0; r < n_workers; ++r) {
            workers[(size_t)r].lo = bounds[(size_t)r];
            workers[(size_t)r].hi = bounds[(size_t)r + 1];
        }
        return true;
    };
    if (ok) INFO("Preparing for simulation");
    if (ok && !sys_read.empty()) {
        ok = on_all_workers(workers, [&](size_t r) {
            Worker &w = workers[r];
            if (w.fail(rsq_sim_prepare(w.sim, job.seed, job.num_reads, job.coverage, job.ref_bias_mode, job.base_identifier.c_str(), nullptr), "Preparation failed")) return;
            w.fail(rsq_sim_read_sys_errors(w.sim, sys_read.c_str()), "Could not read systematic error profile");
        });
        ok = ok && ranges_from(workers[0].sim);
    } else if (ok) {
        ok = on_all_workers(workers, [&](size_t r) {
            workers[r].fail(rsq_sim_prepare_plan(workers[r].sim, job.seed, job.num_reads, job.coverage, job.ref_bias_mode, job.base_identifier.c_str()), "Preparation failed");
        });
        ok = ok && ranges_from(workers[0].sim);
        size_t n_partials = 0;
        ok = ok && check(rsq_sim_bias_partials(workers[0].sim, 0, 0, nullptr, nullptr, 0, &n_partials, nullptr), "Preparation failed");
        std::vector<std::vector<double>> sums((size_t)n_workers, std::vector<double>(n_partials)), maxes((size_t)n_workers, std::vector<double>(n_partials));
        ok = ok && on_all_workers(workers, [&](size_t r) {
            size_t n = 0;
            workers[r].fail(rsq_sim_bias_partials(workers[r].sim, workers[r].lo, workers[r].hi, sums[r].data(), maxes[r].data(), n_partials, &n, nullptr), "Preparation failed");
        });
        for (int r = 1; ok && r < n_workers; ++r)
            for (size_t i = 0; i < n_partials; ++i) {
                sums[0][i] += sums[(size_t)r][i];
                maxes[0][i] += maxes[(size_t)r][i];
            }
        ok = ok && on_all_workers(workers, [&](size_t r) {
            workers[r].fail(rsq_sim_prepare_normalization(workers[r].sim, sums[0].data(), maxes[0].data(), n_partials), "Preparation failed");
        });
        std::vector<std::array<uint32_t, 2>> in_state((size_t)n_workers, std::array<uint32_t, 2>{0u, 0u}), out_state = in_state;
        for (int round = 0; ok; ++round) {
            ok = on_all_workers(workers, [&](size_t r) {
                workers[r].fail(rsq_sim_prepare_sys_errors(workers[r].sim, workers[r].lo, workers[r].hi, in_state[r].data(), out_state[r].data(), nullptr), "Preparation failed");
            });
            bool changed = false;
            for (int r = 0; ok && r < n_workers; ++r) {             // forward chains hand their state to the right, reverse chains to the left
                const std::array<uint32_t, 2> now{r > 0 ? out_state[(size_t)r - 1][0] : 0u, r + 1 < n_workers ? out_state[(size_t)r + 1][1] : 0u};
                changed = changed || now != in_state[(size_t)r];
                in_state[(size_t)r] = now;
            }
            if (!changed) break;
            if (round > n_workers + 2) {
                ERR("the chain states at the shard borders did not settle");
                ok = false;
            }
        }
        ok = ok && on_all_workers(workers, [&](size_t r) { workers[r].fail(rsq_sim_prepare_finish(workers[r].sim), "Preparation failed"); });
    }
    rsq_sim_info info;
    if (ok) {
        rsq_sim_get_info(workers[0].sim, &info);
        INFO("Aiming for " << info.total_pairs + info.adapter_only_pairs << " read pairs");
    }
    if (ok) {
        INFO("Starting read generation");
        ok = on_all_workers(workers, [&](size_t r) {
            Worker &w = workers[r];
            if (w.fail(rsq_sim_job_generate(w.sim, w.lo, w.hi, 0, &w.pairs, &w.bytes[0], &w.bytes[1], nullptr), "Simulation failed")) return;
            if (gz && w.bytes[0] + w.bytes[1]) w.fail(rsq_sim_job_compress(w.sim, &w.bytes[0], &w.bytes[1]), "Compressing the output failed");
        });
    }
    uint64_t end[2] = {0, 0}, pairs = 0;
    if (ok) {                                                 // the files at the size of the workers' text, then all workers write their byte ranges at once
        for (const Worker &w : workers) {
            end[0] += w.bytes[0];
            end[1] += w.bytes[1];
            pairs += w.pairs;
        }
        INFO("Generated " << pairs << " read pairs (" << (info.total_pairs ? (pairs * 100 + info.total_pairs / 2) / info.total_pairs : 0) << "%).");
        for (int f = 0; ok && f < 2; ++f) {
            const std::string &path = f ? job.out2 : job.out1;
            const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (fd < 0 || ftruncate(fd, (off_t)end[f]) != 0) {
                ERR("Could not open '" << path << "' for writing.");
                ok = false;
            }
            if (fd >= 0) ::close(fd);
        }
    }
    if (ok) {
        std::vector<uint64_t> at0(workers.size()), at1(workers.size());
        uint64_t o0 = 0, o1 = 0;
        for (size_t r = 0; r < workers.size(); ++r) {
            at0[r] = o0;
            at1[r] = o1;
            o0 += workers[r].bytes[0];
            o1 += workers[r].bytes[1];
        }
        ok = on_all_workers(workers, [&](size_t r) {
            Worker &w = workers[r];
            if (w.bytes[0] + w.bytes[1] && w.fail(rsq_sim_job_write(w.sim, job.out1.c_str(), at0[r], job.out2.c_str(), at1[r], 0), "Writing the output failed")) return;
            rsq_sim_job_free(w.sim);
        });
    }
    if (ok && info.adapter_only_pairs) {                      // Simulator.cpp:2359-2382, behind everything else; the first worker's simulator
        Worker &w = workers[0];
        DevBuffer d1, d2, g1, g2;
        d1.device = d2.device = g1.device = g2.device = w.device;
        for (uint64_t first = 0; ok && first < info.adapter_only_pairs; first += 100000) {
            const uint64_t n = std::min<uint64_t>(100000, info.adapter_only_pairs - first);
            size_t len[2] = {0, 0};
            int rc = rsq_sim_adapter_only_pairs(w.sim, first, n, (char *)d1.p, d1.cap, &len[0], (char *)d2.p, d2.cap, &len[1], nullptr);
            if (rc == RSQ_ENOSPC) {
                ok = d1.ensure(len[0] + 4096) && d2.ensure(len[1] + 4096);
                if (ok) rc = rsq_sim_adapter_only_pairs(w.sim, first, n, (char *)d1.p, d1.cap, &len[0], (char *)d2.p, d2.cap, &len[1], nullptr);
            }
            ok = ok && check(rc, "Simulation of adapter-only pairs failed");
            for (int f = 0; ok && f < 2; ++f) {
                DevBuffer &text = f ? d2 : d1, &packed = f ? g2 : g1;
                const void *src = text.p;
                size_t bytes = len[f];
                if (gz && bytes) {
                    ok = packed.ensure(rsq_gzip_bound(bytes)) && check(rsq_sim_gzip_device(w.sim, (const char *)text.p, bytes, (char *)packed.p, packed.cap, &bytes, nullptr), "Compressing the output failed");
                    src = packed.p;
                }
                if (ok && bytes) ok = check(rsq_dev_pwrite(w.device, src, bytes, (f ? job.out2 : job.out1).c_str(), end[f]), "Writing the output failed");
                end[f] += bytes;
            }
        }
    }
    int64_t host_gzip = 0;
    rsq_get_option("host_gzip", &host_gzip);
    if (ok && gz && !host_gzip) {                             // files of device-made members (BGZF blocks) end with BGZF's end-of-file member
        char eof[32];
        const size_t n = rsq_gzip_eof_member(eof, sizeof eof);
        for (int f = 0; ok && f < 2; ++f) {
            const int fd = ::open((f ? job.out2 : job.out1).c_str(), O_WRONLY);
            ok = fd >= 0 && pwrite(fd, eof, n, (off_t)end[f]) == (ssize_t)n;
            if (fd >= 0) ::close(fd);
            if (!ok) ERR("Writing the output failed");
        }
    }
    release();
    if (!ok) {                                               // Simulator.cpp:2888-2892: do not leave partial output behind
        ERR("An error occurred in the process: Terminating simulation");
        remove(job.out1.c_str());
        remove(job.out2.c_str());
        return 1;
    }
    INFO("Simulation finished succesfully");
    return 0;
}

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
    StageTrace trace;
    trace.on = a.has("traceStages");
    bool ok = load_profile(a, &prof);
    trace.at("profile loaded");
    const uint64_t seed = ok ? get_seed(a) : 0;
    if (ok) {
        INFO("Reading reference from " << ref_path);
        ok = check(rsq_ref_load_fasta(ref_path.c_str(), &ref), "Could not load reference") && check(rsq_ref_replace_n(ref, seed), "ReplaceN");
        trace.at("reference read");
    }
    if (ok && !vcf_path.empty()) {
        INFO("Reading variants from " << vcf_path);
        ok = check(rsq_ref_read_variants(ref, vcf_path.c_str()), "Could not read the variant file");
    }
    // --gpus N: N workers inside this process, worker r on device r % devices (more workers than devices share them).  The reference's -j / --threads asks for
    // worker threads of the simulation (main.cpp:436, Simulator.cpp:2830-2836): without --gpus it asks for as many workers as there are devices to give each its own.
    int n_workers = 1, n_devices = 1;
    if (ok && (a.has("gpus") || a.has("threads"))) {
        uint64_t asked = 0;
        ok = parse_u64(a, a.has("gpus") ? "gpus" : "threads", asked);
        n_devices = ok ? rsq_device_count() : 1;
        if (ok && n_devices < 1) ok = check(n_devices, "No device");
        if (ok && a.has("gpus") && (asked < 1 || asked > 1024)) {
            ERR("gpus must be between 1 and 1024.");
            ok = false;
        }
        if (ok) n_workers = a.has("gpus") ? (int)asked : (int)std::min<uint64_t>(std::max<uint64_t>(asked, 1), (uint64_t)n_devices);
    }
    if (ok && n_workers > 1) {
        const bool gz = rsq::textio::has_suffix(out1, ".gz");
        uint64_t num_reads = 0;
        double coverage = 0.0;
        if (a.has("numReads") && a.has("coverage")) {               // main.cpp:783-786
            ERR("numReads and coverage option are mutually exclusive. Specify the one or the other.");
            ok = false;
        }
        if (ok && gz != rsq::textio::has_suffix(out2, ".gz")) {
            ERR("with more than one worker the two output files are either both plain or both .gz");
            ok = false;
        }
        if (ok && (rsq::textio::has_suffix(out1, ".bz2") || rsq::textio::has_suffix(out2, ".bz2"))) {
            ERR("bzip2 output is written by one worker only (write .gz or plain FASTQ with --gpus)");
            ok = false;
        }
        if (ok && a.has("numReads")) ok = parse_u64(a, "numReads", num_reads);
        if (ok && a.has("coverage")) ok = parse_double(a, "coverage", coverage);
        int rc = 1;
        if (ok) {
            const PeJob job{a, prof, ref, seed, num_reads, coverage, ref_bias_mode, ref_bias_file, sys_read, sys_write, a.get("methylation", ""), out1, out2, a.get("recordBaseIdentifier", "ReseqRead")};
            rc = illumina_pe_on_workers(job, n_workers, n_devices);
        }
        rsq_ref_free(ref);
        rsq_profile_free(prof);
        return rc;
    }
    uint64_t device = 0;                                         // --device D: the one worker's device (with --gpus N the workers take devices 0 .. N - 1)
    if (ok && a.has("device")) ok = parse_u64(a, "device", device);
    ok = ok && check(rsq_sim_create(prof, ref, (int)device, &sim), "Could not set up the simulator");
    trace.at("simulator created");
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
    trace.at("prepared");
    AsyncOut f1, f2;
    // .gz outputs: the text of every call becomes gzip members on the device (rsq_sim_gzip_device) -- a third of the bytes cross the link and the writer threads
    // only write (--rsqOption host_gzip:1: zlib on host threads behind the writers, as before)
    int64_t host_gzip = 0;
    rsq_get_option("host_gzip", &host_gzip);
    const bool gz1 = !host_gzip && rsq::textio::has_suffix(out1, ".gz"), gz2 = !host_gzip && rsq::textio::has_suffix(out2, ".gz");
    if (ok && (gz1 || gz2)) rsq_sim_gzip_keep_code(sim, 1);      // one Huffman code for the run: the first batch's sample
    {                                                            // files of device-made members (BGZF blocks) end with BGZF's end-of-file member
        char eof[32];
        const size_t n = rsq_gzip_eof_member(eof, sizeof eof);
        if (gz1) f1.tail.assign(eof, n);
        if (gz2) f2.tail.assign(eof, n);
    }
    if (ok) {
        const bool o1 = f1.open(out1, gz1), o2 = f2.open(out2, gz2);
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
        DevBuffer d1, d2, g1, g2;
        d1.device = d2.device = g1.device = g2.device = (int)device;
        uint64_t written = 0;
        // a call's text of one file as members in `g`: true and the members' size, or false
        auto members = [&](bool gz, DevBuffer &d, size_t &len, DevBuffer &g) {
            if (!gz || !len) return true;
            size_t packed = 0;
            if (!g.ensure(len / 2 + (1u << 20))) return false;
            int rc = rsq_sim_gzip_device(sim, (const char *)d.p, len, (char *)g.p, g.cap, &packed, nullptr);
            if (rc == RSQ_ENOSPC && g.ensure(rsq_gzip_bound(len))) rc = rsq_sim_gzip_device(sim, (const char *)d.p, len, (char *)g.p, g.cap, &packed, nullptr);
            len = packed;
            return check(rc, "Compressing the output failed");
        };
        // about 12 M pairs per call: large launches keep the persistent read kernel's tail short (one call of 14.5 M pairs runs at 179 M pairs/s, calls of 2.4 M at
        // 154 M), and sparse coverage needs long block ranges
        const double pairs_per_block = (double)info.total_pairs / std::max<uint32_t>(1u, info.total_blocks);
        const uint32_t step = (uint32_t)std::min(400000.0, std::max(2000.0, 12e6 / std::max(1e-9, pairs_per_block)));
        for (uint32_t lo = 1; ok && lo <= info.total_blocks; lo += step) {
            const uint32_t hi = std::min(info.total_blocks + 1, lo + step);
            size_t l1 = 0, l2 = 0;
            uint64_t n = 0;
            int rc = rsq_sim_pairs(sim, lo, hi, (char *)d1.p, d1.cap, &l1, (char *)d2.p, d2.cap, &l2, &n, nullptr, 0, nullptr);
            if (rc == RSQ_ENOSPC) {
                ok = d1.ensure(l1 + l1 / 8 + 4096) && d2.ensure(l2 + l2 / 8 + 4096);
                if (ok) rc = rsq_sim_pairs(sim, lo, hi, (char *)d1.p, d1.cap, &l1, (char *)d2.p, d2.cap, &l2, &n, nullptr, 0, nullptr);
            }
            ok = ok && check(rc, "Simulation failed") && (n == 0 || (members(gz1, d1, l1, g1) && members(gz2, d2, l2, g2) && flush_pair(gz1 ? g1 : d1, l1, gz2 ? g2 : d2, l2, f1, f2)));
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
            ok = ok && check(rc, "Simulation of adapter-only pairs failed") && members(gz1, d1, l1, g1) && members(gz2, d2, l2, g2) && flush_pair(gz1 ? g1 : d1, l1, gz2 ? g2 : d2, l2, f1, f2);
        }
    }
    trace.at("last text handed to the writers");
    f1.close();
    f2.close();
    trace.at("files closed");
    ok = ok && f1.good() && f2.good();
    rsq_sim_free(sim);
    rsq_ref_free(ref);
    rsq_profile_free(prof);
    trace.at("simulator released");
    if (!ok) {                                               // Simulator.cpp:2888-2892: do not leave partial output behind
        ERR("An error occurred in the process: Terminating simulation");
        remove(out1.c_str());
        remove(out2.c_str());
        return 1;
    }
    INFO("Simulation finished succesfully");
    return 0;
}

// seqToIllumina on several devices inside one process (the reference's worker threads, Simulator.cpp:2900-3014 with -j): the input -- a plain file -- is cut into the
// workers' stretches of bytes, a record belongs to the worker in whose stretch it begins (rsq_fasta_count_records; the rule of reseq_amd/sharding.py record_share), its
// index in the whole input selects its random stream, and every worker keeps its text in device memory until the sizes are known and then writes its byte range of the
// one output file.  Byte for byte the single worker's file.
int seq_to_illumina_on_workers(const Args &a, rsq_profile *prof, uint64_t seed, int n_workers, int n_devices, const std::string &input, const std::string &output,
                               rsq_error_model_file_options base) {
    struct Share {
        int device = 0;
        rsq_sim *sim = nullptr;
        uint64_t lo = 0, hi = 0, n_starts = 0, first_start = 0, begin = 0, end = 0, first_record = 0, records = 0, bytes = 0;
        std::string error;
    };
    std::vector<Share> w((size_t)n_workers);
    struct stat st;
    if (stat(input.c_str(), &st) != 0) {
        ERR("Could not open '" << input << "' for reading.");
        return 1;
    }
    const uint64_t size = (uint64_t)st.st_size;
    INFO("Simulating with " << n_workers << " workers on " << std::min(n_workers, n_devices) << " device(s)");
    auto on_all = [&](auto &&f) {
        std::vector<std::thread> threads;
        for (size_t r = 1; r < w.size(); ++r) threads.emplace_back([&, r] { f(r); });
        f(0);
        for (std::thread &t : threads) t.join();
        for (size_t r = 0; r < w.size(); ++r)
            if (!w[r].error.empty()) {
                ERR(w[r].error);                                  // the reference's complaint about a record (Simulator.cpp:2423-2485), or what went wrong
                return false;
            }
        return true;
    };
    auto fail = [&](size_t r, int rc, const char *what) {
        if (rc == RSQ_OK) return false;
        w[r].error = *what ? std::string(what) + ": " + rsq_last_error() : std::string(rsq_last_error());
        return true;
    };
    bool ok = on_all([&](size_t r) {                              // the record starts of every worker's stretch (host code), its simulator meanwhile
        w[r].device = (int)r % n_devices;
        w[r].lo = size * r / w.size();
        w[r].hi = size * (r + 1) / w.size();
        w[r].first_start = w[r].hi;
        if (w[r].hi > w[r].lo && fail(r, rsq_fasta_count_records(input.c_str(), w[r].lo, w[r].hi, 0, &w[r].n_starts, &w[r].first_start), "Counting the records")) return;
        if (fail(r, rsq_sim_create(prof, nullptr, w[r].device, &w[r].sim), "Could not set up the simulator")) return;
        fail(r, rsq_sim_prepare(w[r].sim, seed, 0, 0.0, 0, "", nullptr), "Preparation failed");
    });
    if (ok) {                                                     // the shares: from a worker's first record to the first record of the next worker that has one
        uint64_t before = 0;
        for (size_t r = 0; r < w.size(); ++r) {
            w[r].first_record = before;
            before += w[r].n_starts;
            uint64_t end = size;
            for (size_t k = r + 1; k < w.size(); ++k)
                if (w[k].n_starts) {
                    end = w[k].first_start;
                    break;
                }
            w[r].end = end;
            w[r].begin = r == 0 ? 0 : (w[r].n_starts ? w[r].first_start : end);      // what stands in front of the first record is the first worker's to complain about
        }
        INFO("Starting read generation");
        ok = on_all([&](size_t r) {
            if (w[r].end <= w[r].begin && r) return;                 // no record begins in this worker's stretch
            rsq_error_model_file_options opt = base;
            opt.keep_text = 1;
            opt.from = w[r].begin;
            opt.to = w[r].end;
            opt.first_record = w[r].first_record;
            opt.progress = nullptr;
            opt.trace = nullptr;
            opt.trace_cap = 0;
            if (w[r].end <= w[r].begin) return;                      // (an empty file: nothing for the first worker either)
            fail(r, rsq_sim_error_model_file(w[r].sim, input.c_str(), nullptr, &opt, &w[r].records, &w[r].bytes), "");
        });
    }
    const bool gz = rsq::textio::has_suffix(output, ".gz");
    if (ok && gz)
        ok = on_all([&](size_t r) {
            uint64_t none = 0;
            if (w[r].bytes) fail(r, rsq_sim_job_compress(w[r].sim, &w[r].bytes, &none), "Compressing the output failed");
        });
    uint64_t records = 0, total = 0;
    for (const Share &x : w) records += x.records, total += x.bytes;
    if (ok && !records) {
        ERR(input << " does not contain any sequences.");
        ok = false;
    }
    if (ok) {
        INFO("Generated " << records << " reads.");
        const int fd = ::open(output.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0 || ftruncate(fd, (off_t)total) != 0) {
            ERR("Could not open '" << output << "' for writing.");
            ok = false;
        }
        if (fd >= 0) ::close(fd);
    }
    if (ok) {
        std::vector<uint64_t> at(w.size());
        uint64_t o = 0;
        for (size_t r = 0; r < w.size(); ++r) at[r] = o, o += w[r].bytes;
        ok = on_all([&](size_t r) {
            if (w[r].bytes && fail(r, rsq_sim_job_write(w[r].sim, output.c_str(), at[r], nullptr, 0, 0), "Writing the output failed")) return;
            if (w[r].bytes) rsq_sim_job_free(w[r].sim);
        });
    }
    int64_t host_gzip = 0;
    rsq_get_option("host_gzip", &host_gzip);
    if (ok && gz && !host_gzip) {                                 // a file of device-made members ends with BGZF's end-of-file member
        char eof[32];
        const size_t n = rsq_gzip_eof_member(eof, sizeof eof);
        const int fd = ::open(output.c_str(), O_WRONLY);
        ok = fd >= 0 && pwrite(fd, eof, n, (off_t)total) == (ssize_t)n;
        if (fd >= 0) ::close(fd);
        if (!ok) ERR("Writing the output failed");
    }
    for (Share &x : w) rsq_sim_free(x.sim);
    if (!ok) {
        ERR("An error occurred in the process: Terminating simulation");
        remove(output.c_str());
        return 1;
    }
    INFO("Simulation finished succesfully");
    return 0;
}

// ---- seqToIllumina (main.cpp:1009-1021, 1131; Simulator::SimulateErrorModelOnly, Simulator.cpp:2900-3014): the library's file-to-file pipeline
// (rsq_sim_error_model_file: readers at file offsets, the FASTA text parsed on the device, ordered output) with the reference's options and messages
int seq_to_illumina(const Args &a) {
    rsq_profile *prof = nullptr;
    rsq_sim *sim = nullptr;
    bool ok = load_profile(a, &prof);
    const double at_profile = seconds_since(g_process_start);
    const uint64_t seed = ok ? get_seed(a) : 0;
    uint64_t device = 0;                                         // --device D
    if (ok && a.has("device")) ok = parse_u64(a, "device", device);
    // --gpus N / -j N: N workers as for illuminaPE -- for a plain input file and an output file (a stream or a compressed input has one reader: one worker then)
    if (ok && (a.has("gpus") || a.has("threads"))) {
        uint64_t asked = 0;
        ok = parse_u64(a, a.has("gpus") ? "gpus" : "threads", asked);
        const int n_devices = ok ? rsq_device_count() : 1;
        if (ok && n_devices < 1) ok = check(n_devices, "No device");
        if (ok && a.has("gpus") && (asked < 1 || asked > 1024)) {
            ERR("gpus must be between 1 and 1024.");
            ok = false;
        }
        const int n_workers = !ok ? 1 : a.has("gpus") ? (int)asked : (int)std::min<uint64_t>(std::max<uint64_t>(asked, 1), (uint64_t)n_devices);
        const std::string input = a.get("input", ""), output = a.get("output", "");
        const bool plain_files = !input.empty() && !output.empty() && !rsq::textio::has_suffix(input, ".gz") && !rsq::textio::has_suffix(input, ".bz2") &&
                                 !rsq::textio::has_suffix(output, ".bz2") && !a.has("inputFrom") && !a.has("inputTo");
        if (ok && n_workers > 1 && !plain_files) WARN("--gpus / -j: several workers need a plain input file and an output file (not .bz2); one worker runs");
        if (ok && n_workers > 1 && plain_files) {
            rsq_error_model_file_options base;
            memset(&base, 0, sizeof base);
            for (const char *name : {"readThreads", "parseThreads"})
                if (a.has(name)) base.read_threads = (uint32_t)std::max(1, atoi(a.get(name).c_str()));
            if (a.has("blockKB")) base.block_kb = (uint32_t)std::max(1, atoi(a.get("blockKB").c_str()));
            if (a.has("batchBlocks")) base.batch_blocks = (uint32_t)std::max(1, atoi(a.get("batchBlocks").c_str()));
            const int rc = seq_to_illumina_on_workers(a, prof, seed, n_workers, n_devices, input, output, base);
            rsq_profile_free(prof);
            return rc;
        }
    }
    ok = ok && check(rsq_sim_create(prof, nullptr, (int)device, &sim), "Could not set up the simulator") &&
         check(rsq_sim_prepare(sim, seed, 0, 0.0, 0, "", nullptr), "Preparation failed");
    const double at_prepared = seconds_since(g_process_start);
    if (ok) {
        INFO("Starting read generation");
        rsq_error_model_file_options opt;
        memset(&opt, 0, sizeof opt);
        // --readThreads, --blockKB, --batchBlocks: the pipeline's sizes (the tests make them small); --inputFrom / --inputTo / --firstRecord: a rank's share of
        // the input in a job over several GPUs (reseq_amd/simulate.py works them out)
        for (const char *name : {"readThreads", "parseThreads"})
            if (a.has(name)) opt.read_threads = (uint32_t)std::max(1, atoi(a.get(name).c_str()));
        if (a.has("blockKB")) opt.block_kb = (uint32_t)std::max(1, atoi(a.get("blockKB").c_str()));
        if (a.has("batchBlocks")) opt.batch_blocks = (uint32_t)std::max(1, atoi(a.get("batchBlocks").c_str()));
        ok = (!a.has("inputFrom") || parse_u64(a, "inputFrom", opt.from)) && (!a.has("inputTo") || parse_u64(a, "inputTo", opt.to)) &&
             (!a.has("firstRecord") || parse_u64(a, "firstRecord", opt.first_record));
        opt.progress = [](uint64_t records, void *) { INFO("Generated " << records << " reads."); };
        char trace[1024] = "";
        if (a.has("traceStages")) {
            opt.trace = trace;
            opt.trace_cap = sizeof trace;
        }
        uint64_t records = 0, bytes = 0;
        if (ok) {
            const int rc = rsq_sim_error_model_file(sim, a.has("input") ? a.get("input").c_str() : nullptr, a.has("output") ? a.get("output").c_str() : nullptr, &opt, &records, &bytes);
            if (rc != RSQ_OK) {
                ERR(rsq_last_error());                           // the reference's complaint about a record (Simulator.cpp:2423-2485), or what went wrong with a file
                ok = false;
            }
        }
        if (*trace) fprintf(stderr, "stages: profile loaded at %.3f s of the process, simulator prepared at %.3f; %s\n", at_profile, at_prepared, trace);
        if (ok && !records && !a.has("inputFrom")) {
            ERR(a.get("input", "stdin") << " does not contain any sequences.");
            ok = false;
        }
    }
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
    "                 \t-i in.fa[.gz|.bz2] (stdin) -o out.fq[.gz|.bz2] (stdout) -s profile; --readThreads N, --traceStages;\n"
    "                 \t--inputFrom / --inputTo BYTE, --firstRecord K: a share of a plain input (python -m reseq_amd.simulate seqToIllumina works them out)\n"
    "                 \t--gpus N: N workers in this process, worker r on device r % devices, the files byte for byte the single-device run's; --device D: the one worker's device\n"
    "Outputs named *.gz are compressed on the GPU (gzip members framed as BGZF blocks; about 15 % larger than zlib level 1, 33 % larger than level 6);\n"
    "         --hostGzip compresses them with zlib on host threads instead (smaller files, a fraction of the speed).\n"
    "General: -j/--threads N (illuminaPE, seqToIllumina: as many workers as asked for, at most one per device; --gpus N: exactly N),\n"
    "         --verbosity 0-4, --version, -h, --traceStages,\n"
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
    if (ok && a.has("statsOut")) {                                // not in the reference: the loaded profile as ReSeq's own pair of files (-S <name>.reseq [-P <fit file>]), INTEGRATION.md
        const std::string out = a.get("statsOut"), fit = a.get("probabilitiesOut");
        ok = check(rsq_profile_save_reseq(prof, out.c_str(), fit.empty() ? nullptr : fit.c_str(), 0), "Could not write the profile archives");
        if (ok) INFO("Wrote " << out << " and " << (fit.empty() ? out + ".ipf" : fit));
        if (ok)
            WARN("EXPERIMENTAL: these archives hold the prepared tables and made-up raw statistics around them (what the simulation never reads), under Boost token rules that no "
                 "Boost-written file could be checked against here; this library reads them back bit for bit, whether the original reseq does is untested (INTEGRATION.md)");
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
    if (a.has("hostGzip") && !check(rsq_set_option("host_gzip", 1), "--hostGzip")) return 1;      // .gz outputs by zlib on host threads: smaller files, slower
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
