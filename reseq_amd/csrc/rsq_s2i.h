// rsq_s2i.h -- seqToIllumina from file to file (Simulator::SimulateErrorModelOnly, reseq/Simulator.cpp:2900-3014: a reader, ErrorModelOnlyThread
// :2514-2560 on the worker threads, ordered output :184-213) as a pipeline around rsq_sim_error_model_fasta.  The FASTA text goes to the device as it
// stands in the file and is parsed there (rsq_fasta.h), so the host only moves bytes:
//   input side    blocks of the file in page-locked slots, uploaded by the thread that read them -- a plain file is read by several threads at fixed
//                 offsets (records that cross a block's end are the simulator side's business), a compressed file or stdin by one, in sequence;
//   simulator     the calling thread takes the blocks in input order -- all that are there, up to a batch -- and puts them behind what the call before left
//                 over (device to device: a call on 100 000 records takes 1.1 ms, on 800 000 four: the read kernel of a small call is a chain of 150 steps
//                 on waves that have a SIMD to themselves), runs the device call;
//   output side   a thread downloads the FASTQ text into page-locked buffers, another writes (and compresses) them.
// Every side has its own stream and its own buffers: they overlap.  Included by rsq_sim.hip only (it needs the simulator and HIP_CHECK).
#pragma once
#include <condition_variable>
#include <mutex>

namespace rsq {
namespace s2i {

using Clock = std::chrono::steady_clock;
struct StageTime {                    // seconds a side of the pipeline spent in one of its stages, summed over its threads
    std::atomic<uint64_t> ns{0};
    void add(Clock::time_point t0) { ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - t0).count(); }
    double s() const { return (double)ns.load() * 1e-9; }
};
inline double seconds_since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

struct Pinned {                       // page-locked, grow-only
    void *p = nullptr;
    size_t cap = 0;
    void ensure(size_t n) {
        if (n <= cap) return;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        HIP_CHECK(hipHostMalloc(&p, n, hipHostMallocDefault));
        cap = n;
    }
    char *chars() { return static_cast<char *>(p); }
    ~Pinned() {
        if (p) (void)hipHostFree(p);
    }
};
struct CopyStream {
    hipStream_t st = nullptr;
    CopyStream() { HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); }
    ~CopyStream() {
        if (st) (void)hipStreamDestroy(st);
    }
    void copy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
        if (bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, kind, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
};

// what a side's threads share: the first error ends the run for all of them
struct Fault {
    std::mutex m;
    std::string what;
    std::atomic<bool> set{false};
    void raise(const std::string &message) {
        std::lock_guard<std::mutex> lock(m);
        if (!set) what = message;
        set = true;
    }
};

// sequential input: a plain, gzip or bzip2 file by its content (SeqAn's SeqFileIn does the same), or stdin
struct SequentialIn {
    textio::Reader r;
    bool is_file = false;
    bool open(const char *path) {
        if (!path) return true;
        is_file = r.open(path);
        return is_file;
    }
    size_t read(void *dst, size_t n) {           // bytes read, 0 at the end; a read error throws
        if (is_file) return (size_t)r.read(dst, (unsigned)std::min<size_t>(n, 1u << 30));
        const size_t got = fread(dst, 1, n, stdin);
        if (!got && ferror(stdin)) throw Error("reading the standard input failed");
        return got;
    }
};
struct SequentialOut {
    textio::Writer w;
    bool failed = false;
    bool open(const char *path, bool as_it_comes = false) { return !path || (as_it_comes ? w.open_plain(path) : w.open(path)); }
    void write(const char *data, size_t n) {
        if (w.is_open()) w.write(data, n);
        else failed = failed || fwrite(data, 1, n, stdout) != n;
    }
    bool good() const { return !failed && !w.failed; }
    void close() {
        failed = !w.close() || failed;
        if (!w.is_open()) fflush(stdout);
    }
};

struct InPipe {
    struct Slot {
        Pinned host;
        DevBuf dev;
        size_t len = 0;
        bool last = false, ready = false;
        uint64_t turn = 0;                           // the block this slot serves next
    };
    const int device;
    const size_t block_bytes;
    std::vector<Slot> slots;
    SequentialIn *stream_in = nullptr;               // sequential input ...
    int fd = -1;                                     // ... or a plain file read at offsets: bytes [from, to)
    uint64_t from = 0, to = 0, n_blocks = 0;
    std::atomic<uint64_t> next{0};
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::thread> readers;
    bool abort = false;
    Fault &fault;
    StageTime t_read, t_upload, t_slot;
    InPipe(int dev, size_t block, size_t n_readers, Fault &f) : device(dev), block_bytes(block), slots(n_readers + 2), fault(f) {
        for (size_t k = 0; k < slots.size(); ++k) slots[k].turn = k;
    }
    void start_file(int file, uint64_t first_byte, uint64_t end_byte, size_t n_readers) {
        fd = file;
        from = first_byte;
        to = end_byte;
        n_blocks = std::max<uint64_t>(1, (to - from + block_bytes - 1) / block_bytes);
        for (size_t k = 0; k < n_readers; ++k) readers.emplace_back([this] { run([this](CopyStream &st) { read_at_offsets(st); }); });
    }
    void start_stream(SequentialIn &in) {
        stream_in = &in;
        readers.emplace_back([this] { run([this](CopyStream &st) { read_in_sequence(st); }); });
    }
    template <class F>
    void run(F &&body) {
        try {
            HIP_CHECK(hipSetDevice(device));
            CopyStream st;
            body(st);
        } catch (const std::exception &e) {
            fault.raise(e.what());
            std::lock_guard<std::mutex> lock(m);
            cv.notify_all();
        }
    }
    Slot *wait_for_slot(uint64_t b) {                // the slot of block b once the block that used it before is done with; nullptr: the run ends
        const auto t0 = Clock::now();
        Slot *s = &slots[b % slots.size()];
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&] { return (s->turn == b && !s->ready) || abort || fault.set; });
        t_slot.add(t0);
        return abort || fault.set ? nullptr : s;
    }
    void upload_and_publish(Slot *s, size_t len, bool last, CopyStream &st) {
        const auto t0 = Clock::now();
        st.copy(s->dev.as<char>(), s->host.p, len, hipMemcpyHostToDevice);
        t_upload.add(t0);
        std::lock_guard<std::mutex> lock(m);
        s->len = len;
        s->last = last;
        s->ready = true;
        cv.notify_all();
    }
    void read_at_offsets(CopyStream &st) {
        for (;;) {
            const uint64_t b = next++;
            if (b >= n_blocks) break;
            Slot *s = wait_for_slot(b);
            if (!s) break;
            const uint64_t off = from + b * block_bytes;
            const size_t len = (size_t)std::min<uint64_t>(block_bytes, to - off);
            s->host.ensure(block_bytes);
            s->dev.reserve(block_bytes + 16);
            const auto t0 = Clock::now();
            for (size_t have = 0; have < len;) {
                const ssize_t got = pread(fd, s->host.chars() + have, len - have, (off_t)(off + have));
                if (got < 0 && errno == EINTR) continue;
                if (got <= 0) throw Error(got ? std::string("reading the input failed: ") + strerror(errno) : std::string("reading the input failed: the file has become shorter"));
                have += (size_t)got;
            }
            t_read.add(t0);
            upload_and_publish(s, len, b + 1 == n_blocks, st);
        }
    }
    void read_in_sequence(CopyStream &st) {
        for (uint64_t b = 0;; ++b) {
            Slot *s = wait_for_slot(b);
            if (!s) break;
            s->host.ensure(block_bytes);
            s->dev.reserve(block_bytes + 16);
            const auto t0 = Clock::now();
            size_t have = 0;
            bool end_of_input = false;
            while (have < block_bytes) {
                const size_t got = stream_in->read(s->host.chars() + have, block_bytes - have);
                if (!got) {
                    end_of_input = true;
                    break;
                }
                have += got;
            }
            t_read.add(t0);
            upload_and_publish(s, have, end_of_input, st);
            if (end_of_input) break;
        }
    }
    // block b, in input order; nullptr after an error -- or, if the caller does not want to wait, while the block is not there yet
    Slot *take(uint64_t b, bool wait = true) {
        Slot *s = &slots[b % slots.size()];
        std::unique_lock<std::mutex> lock(m);
        if (wait) cv.wait(lock, [&] { return (s->turn == b && s->ready) || fault.set; });
        return fault.set || !(s->turn == b && s->ready) ? nullptr : s;
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
    }
    ~InPipe() { join(); }
};

// The output side: device buffers the simulator fills in turn, a thread that copies them into page-locked buffers, a thread that writes those.
struct OutPipe {
    static constexpr uint64_t kDev = 3, kStage = 3;
    static constexpr size_t kChunk = 64u << 20;
    const int device;
    SequentialOut out;
    DevBuf dev[kDev];
    size_t dev_len[kDev] = {0, 0, 0};
    Pinned stage[kStage];
    size_t stage_len[kStage] = {0, 0, 0};
    uint64_t filled = 0, drained = 0, staged = 0, written = 0, bytes = 0;       // texts handed in / downloaded; chunks downloaded / written
    bool closing = false, downloader_done = false, started = false;
    std::mutex m;
    std::condition_variable cv;
    std::thread downloader, writer;
    Fault &fault;
    std::string tail;
    StageTime t_download, t_stage, t_write, t_dev;
    OutPipe(int dev_id, Fault &f) : device(dev_id), fault(f) {}
    bool open(const char *path, bool as_it_comes = false) {      // nullptr: stdout; as_it_comes: no compression by the name (the bytes are .gz members already)
        if (!out.open(path, as_it_comes)) return false;
        downloader = std::thread([this] { guarded([this] { download(); }); finish_download(); });
        writer = std::thread([this] { guarded([this] { write(); }); });
        started = true;
        return true;
    }
    template <class F>
    void guarded(F &&body) {
        try {
            body();
        } catch (const std::exception &e) {
            raise(e.what());
        }
    }
    void raise(const std::string &what) {
        fault.raise(what);
        std::lock_guard<std::mutex> lock(m);
        cv.notify_all();
    }
    // the device buffer the next text goes to (the caller may enlarge it), once its last text has been downloaded; nullptr after an error
    DevBuf *begin() {
        const auto t0 = Clock::now();
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&] { return filled - drained < kDev || fault.set; });
        t_dev.add(t0);
        return fault.set ? nullptr : &dev[filled % kDev];
    }
    void submit(size_t n) {
        std::lock_guard<std::mutex> lock(m);
        dev_len[filled % kDev] = n;
        bytes += n;
        ++filled;
        cv.notify_all();
    }
    void download() {
        HIP_CHECK(hipSetDevice(device));
        CopyStream st;
        for (;;) {
            uint64_t k;
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [&] { return drained < filled || closing || fault.set; });
                if (fault.set || drained == filled) return;
                k = drained % kDev;
            }
            for (size_t done = 0; done < dev_len[k]; done += kChunk) {
                const size_t n = std::min(kChunk, dev_len[k] - done);
                uint64_t j;
                {
                    const auto t0 = Clock::now();
                    std::unique_lock<std::mutex> lock(m);
                    cv.wait(lock, [&] { return staged - written < kStage || fault.set; });
                    t_stage.add(t0);
                    if (fault.set) return;
                    j = staged % kStage;
                }
                const auto t0 = Clock::now();
                stage[j].ensure(kChunk);
                st.copy(stage[j].p, dev[k].as<char>() + done, n, hipMemcpyDeviceToHost);
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
    }
    void finish_download() {
        std::lock_guard<std::mutex> lock(m);
        downloader_done = true;
        cv.notify_all();
    }
    void write() {
        for (;;) {
            uint64_t j;
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [&] { return written < staged || downloader_done || fault.set; });
                if (fault.set || written == staged) return;
                j = written % kStage;
            }
            const auto t0 = Clock::now();
            out.write(stage[j].chars(), stage_len[j]);
            t_write.add(t0);
            if (!out.good()) throw Error("writing the output failed");
            std::lock_guard<std::mutex> lock(m);
            ++written;
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
            if (!tail.empty() && !fault.set) out.write(tail.data(), tail.size());      // what ends the file behind the last text (the BGZF end-of-file member)
            out.close();
            if (!out.good()) fault.raise("writing the output failed");
        }
    }
    ~OutPipe() {
        if (started) {
            fault.raise("the run was given up");
            {
                std::lock_guard<std::mutex> lock(m);
                cv.notify_all();
            }
            close();
        }
    }
};

// a file that can be read at offsets by several threads: regular, and neither gzip nor bzip2 by its first bytes
inline int open_plain_file(const char *path, uint64_t &size) {
    const int fd = ::open(path, O_RDONLY);
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
struct FileDescriptor {
    int fd = -1;
    ~FileDescriptor() {
        if (fd >= 0) ::close(fd);
    }
};

// record starts ('>' at the start of a line, or of the file) in bytes [from, to) of a plain file: their number and the first one's offset (`to` if none)
inline void count_records(const char *path, uint64_t from, uint64_t to, uint32_t threads, uint64_t *n_starts, uint64_t *first_start) {
    uint64_t size = 0;
    FileDescriptor f;
    f.fd = open_plain_file(path, size);
    if (f.fd < 0) throw Error(std::string("'") + path + "' is not a plain file that can be read at offsets (compressed input and pipes cannot be shared among ranks)");
    to = std::min(to ? to : size, size);
    from = std::min(from, to);
    const uint64_t piece = 8u << 20, pieces = (to - from + piece - 1) / piece;
    threads = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(threads ? threads : 4, pieces));
    std::vector<uint64_t> count(pieces, 0), first(pieces, ~0ull);
    std::atomic<uint64_t> next{0};
    Fault fault;
    auto work = [&] {
        std::vector<char> buf(piece + 1);
        for (uint64_t p = next++; p < pieces && !fault.set; p = next++) {
            const uint64_t a = from + p * piece, b = std::min(to, a + piece), lead = a ? 1 : 0;        // one byte in front: is the piece's first byte a line's first?
            size_t have = 0;
            const size_t want = (size_t)(b - a + lead);
            while (have < want) {
                const ssize_t got = pread(f.fd, buf.data() + have, want - have, (off_t)(a - lead + have));
                if (got < 0 && errno == EINTR) continue;
                if (got <= 0) return fault.raise("reading the input failed");
                have += (size_t)got;
            }
            const char *t = buf.data() + lead, *end = buf.data() + want;
            for (const char *q = t; q < end;) {
                q = static_cast<const char *>(memchr(q, '>', (size_t)(end - q)));
                if (!q) break;
                if (q == buf.data() ? a == 0 : q[-1] == '\n') {
                    if (!count[p]++) first[p] = a + (uint64_t)(q - t);
                }
                ++q;
            }
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t k = 1; k < threads; ++k) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
    if (fault.set) throw Error(fault.what);
    *n_starts = 0;
    *first_start = to;
    for (uint64_t p = pieces; p--;) {
        *n_starts += count[p];
        if (count[p]) *first_start = first[p];
    }
}

}  // namespace s2i
}  // namespace rsq
