// rsq_textio.h -- text files the way SeqAn's SeqFileIn / SeqFileOut open them for the reference: plain, gzip (zlib) or bzip2.
// Input format by content (gzip's and bzip2's magic bytes), output format by file name (.gz, .bz2).  libbz2 has no header in the
// build image, so its four stdio-style entry points (stable since bzip2 1.0) are bound at run time from libbz2.so.1; a bzip2 file
// on a machine without that library is an error, never a silent fallback.
#pragma once
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace rsq {
namespace textio {

struct Bz2 {
    void *(*open)(const char *, const char *) = nullptr;
    int (*read)(void *, void *, int) = nullptr;
    int (*write)(void *, void *, int) = nullptr;
    void (*close)(void *) = nullptr;
    static const Bz2 &get() {
        static const Bz2 api = [] {
            Bz2 a;
            void *h = dlopen("libbz2.so.1", RTLD_NOW);
            if (!h) h = dlopen("libbz2.so.1.0", RTLD_NOW);
            if (!h) throw std::runtime_error("bzip2 file, but libbz2.so.1 cannot be loaded");
            a.open = reinterpret_cast<void *(*)(const char *, const char *)>(dlsym(h, "BZ2_bzopen"));
            a.read = reinterpret_cast<int (*)(void *, void *, int)>(dlsym(h, "BZ2_bzread"));
            a.write = reinterpret_cast<int (*)(void *, void *, int)>(dlsym(h, "BZ2_bzwrite"));
            a.close = reinterpret_cast<void (*)(void *)>(dlsym(h, "BZ2_bzclose"));
            if (!a.open || !a.read || !a.write || !a.close) throw std::runtime_error("libbz2.so.1 lacks BZ2_bzopen / BZ2_bzread / BZ2_bzwrite / BZ2_bzclose");
            return a;
        }();
        return api;
    }
};
inline bool has_suffix(const std::string &path, const char *suffix) {
    const size_t n = strlen(suffix);
    return path.size() > n && path.compare(path.size() - n, n, suffix) == 0;
}
inline bool starts_with_bzip2_magic(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char m[3] = {0, 0, 0};
    const size_t n = fread(m, 1, 3, f);
    fclose(f);
    return n == 3 && m[0] == 'B' && m[1] == 'Z' && m[2] == 'h';
}

struct Reader {                          // plain and gzip through zlib (it passes plain data through), bzip2 through libbz2
    gzFile gz = nullptr;
    void *bz = nullptr;
    bool open(const std::string &path) {
        if (starts_with_bzip2_magic(path)) bz = Bz2::get().open(path.c_str(), "rb");
        else {
            gz = gzopen(path.c_str(), "rb");
            if (gz) gzbuffer(gz, 1 << 20);
        }
        return gz || bz;
    }
    int read(void *buf, unsigned n) {    // bytes read, 0 at the end; a read error or a compressed stream that ends early throws
        const int got = bz ? Bz2::get().read(bz, buf, (int)n) : gzread(gz, buf, n);
        if (got < 0) throw std::runtime_error("read error in a compressed or plain text input");
        if (!got && gz) {
            int err = Z_OK;
            gzerror(gz, &err);
            if (err != Z_OK && err != Z_STREAM_END) throw std::runtime_error("gzip input ends in the middle of the stream (truncated file)");
        }
        return got;
    }
    void close() {
        if (gz) gzclose(gz);
        if (bz) Bz2::get().close(bz);
        gz = nullptr;
        bz = nullptr;
    }
    ~Reader() { close(); }
};

// the processors this process may use: the hardware's threads, or fewer under a cgroup quota (cpu.max: "<quota> <period>" -- a container with 16 of a host's
// 256 hardware threads runs 64 compressing threads no faster than 16)
inline unsigned usable_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    long long quota = 0, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                  // cgroup v2 ("max <period>" without a quota: fscanf reads nothing)
        if (fscanf(f, "%lld %lld", &quota, &period) != 2) quota = period = 0;
        fclose(f);
    } else {                                                               // cgroup v1
        FILE *q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *p = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (!q || !p || fscanf(q, "%lld", &quota) != 1 || fscanf(p, "%lld", &period) != 1) quota = period = 0;
        if (q) fclose(q);
        if (p) fclose(p);
    }
    if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    return n;
}

// gzip output by several threads: the text is cut into pieces of kPiece bytes, every piece becomes a gzip member of its own (RFC 1952 allows any number of
// members in a file; zlib's gzread, gzip -d and SeqAn's readers take them as one stream), compressed at zlib's default level like gzopen("wb") does.  The bytes
// written do not depend on the number of threads.  One thread compresses about 20 MB/s of FASTQ text: a single gzip stream behind a device that makes 50 GB/s of
// text is where a run with .gz output spends its time (20 M seqToIllumina records: 330 s; with the 16 processors of the measuring box's container 25 s).
struct ParallelGzip {
    static constexpr size_t kPiece = 1u << 20;
    FILE *f = nullptr;
    bool any = false, failed = false;
    std::vector<unsigned char> carry;                     // text that does not fill a piece yet
    // the workers live as long as the file is open and keep their deflate state and the pieces' output arrays
    std::vector<std::thread> pool;
    std::vector<std::vector<unsigned char>> out;
    std::mutex m;
    std::condition_variable work_cv, done_cv;
    const unsigned char *job = nullptr;
    size_t job_pieces = 0, next_piece = 0, pieces_done = 0;
    uint64_t generation = 0;
    bool stop = false, bad = false;
    std::vector<unsigned char> *memory = nullptr;         // members are appended here instead of written to a file (a rank's compressed share of a job's output)
    bool open(const std::string &path) {
        f = fopen(path.c_str(), "wb");
        if (!f) return false;
        start();
        return true;
    }
    void open_memory(std::vector<unsigned char> &into) {
        memory = &into;
        start();
    }
    bool put(const std::vector<unsigned char> &bytes) {
        if (memory) {
            memory->insert(memory->end(), bytes.begin(), bytes.end());
            return true;
        }
        return fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    }
    void start() {
        // two threads per usable processor: a chunk of 64 pieces then divides evenly enough, and under a quota the scheduler's throttling falls on many short
        // runs instead of stalling few long ones (a container with 16 processors: 14 threads 197 MB/s, 32 threads 223-256, 64 threads 262; one thread 19.4)
        const unsigned threads = std::max(2u, std::min(2u * usable_cpus(), 64u));
        for (unsigned t = 0; t < threads; ++t) pool.emplace_back([this] { worker(); });
    }
    static bool begin_member(z_stream &z) {
        memset(&z, 0, sizeof z);
        return deflateInit2(&z, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) == Z_OK;
    }
    static bool member(z_stream &z, const unsigned char *data, size_t n, std::vector<unsigned char> &out) {
        if (deflateReset(&z) != Z_OK) return false;
        out.resize(deflateBound(&z, (uLong)n) + 32);
        z.next_in = const_cast<unsigned char *>(data);
        z.avail_in = (uInt)n;
        z.next_out = out.data();
        z.avail_out = (uInt)out.size();
        const int rc = deflate(&z, Z_FINISH);
        out.resize(out.size() - z.avail_out);
        return rc == Z_STREAM_END;
    }
    void worker() {
        z_stream z;
        const bool have = begin_member(z);
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lock(m);
        for (;;) {
            work_cv.wait(lock, [&] { return stop || (generation != seen && next_piece < job_pieces); });
            if (stop) break;
            const uint64_t g = generation;
            while (generation == g && next_piece < job_pieces) {
                const size_t p = next_piece++;
                lock.unlock();
                const bool ok = have && member(z, job + p * kPiece, kPiece, out[p]);
                lock.lock();
                bad = bad || !ok;
                if (++pieces_done == job_pieces) done_cv.notify_all();
            }
            seen = g;
        }
        lock.unlock();
        if (have) deflateEnd(&z);
    }
    // whole pieces of data[0, n) as members, in order
    void pieces(const unsigned char *data, size_t n) {
        const size_t count = n / kPiece;
        if (!count || failed) return;
        {
            std::unique_lock<std::mutex> lock(m);
            if (out.size() < count) out.resize(count);
            job = data;
            job_pieces = count;
            next_piece = pieces_done = 0;
            ++generation;
            work_cv.notify_all();
            done_cv.wait(lock, [&] { return pieces_done == job_pieces; });
            job_pieces = 0;
            failed = failed || bad;
        }
        for (size_t p = 0; p < count && !failed; ++p) failed = !put(out[p]);
        any = true;
    }
    void write(const char *text, size_t n) {
        const unsigned char *data = reinterpret_cast<const unsigned char *>(text);
        if (!carry.empty()) {                              // fill the piece that was begun
            const size_t take = std::min(n, kPiece - carry.size());
            carry.insert(carry.end(), data, data + take);
            data += take;
            n -= take;
            if (carry.size() < kPiece) return;
            pieces(carry.data(), carry.size());
            carry.clear();
        }
        pieces(data, n);
        carry.assign(data + n / kPiece * kPiece, data + n);
    }
    // (a file without text is one empty member, as gzclose leaves it; a rank's share in memory may be empty)
    bool close() {
        if (!f && !memory) return !failed;
        {
            std::lock_guard<std::mutex> lock(m);
            stop = true;
            work_cv.notify_all();
        }
        for (std::thread &t : pool) t.join();
        pool.clear();
        if (!carry.empty() || (!any && f)) {               // the rest
            z_stream z;
            std::vector<unsigned char> last;
            const bool ok = begin_member(z) && member(z, carry.data(), carry.size(), last);
            if (ok) deflateEnd(&z);
            failed = !ok || !put(last) || failed;
            carry.clear();
        }
        if (f) failed = (fclose(f) != 0) || failed;
        f = nullptr;
        memory = nullptr;
        return !failed;
    }
    ~ParallelGzip() { close(); }
};

struct Writer {
    FILE *plain = nullptr;
    ParallelGzip gz;
    void *bz = nullptr;
    bool failed = false;
    bool open(const std::string &path) {
        if (has_suffix(path, ".gz")) return gz.open(path);
        if (has_suffix(path, ".bz2")) bz = Bz2::get().open(path.c_str(), "wb");
        else plain = fopen(path.c_str(), "wb");
        return plain || bz;
    }
    bool open_plain(const std::string &path) {       // the bytes as they come, whatever the name says: .gz members made on the device (rsq_deflate.h)
        plain = fopen(path.c_str(), "wb");
        return plain != nullptr;
    }
    bool is_open() const { return plain || gz.f || bz; }
    void write(const char *data, size_t n) {
        if (gz.f) {
            gz.write(data, n);
            failed = failed || gz.failed;
            return;
        }
        for (size_t done = 0; done < n && !failed;) {
            const unsigned chunk = (unsigned)std::min<size_t>(n - done, 1u << 30);
            if (bz) failed = Bz2::get().write(bz, const_cast<char *>(data + done), (int)chunk) != (int)chunk;
            else failed = fwrite(data + done, 1, chunk, plain) != chunk;
            done += chunk;
        }
    }
    bool close() {                       // true if everything was written
        if (gz.f) failed = !gz.close() || failed;
        if (bz) Bz2::get().close(bz);
        if (plain) failed = (fclose(plain) != 0) || failed;
        bz = nullptr;
        plain = nullptr;
        return !failed;
    }
};

}  // namespace textio
}  // namespace rsq
