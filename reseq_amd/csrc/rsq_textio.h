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
#include <stdexcept>
#include <string>

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

struct Writer {
    FILE *plain = nullptr;
    gzFile gz = nullptr;
    void *bz = nullptr;
    bool failed = false;
    bool open(const std::string &path) {
        if (has_suffix(path, ".gz")) gz = gzopen(path.c_str(), "wb");
        else if (has_suffix(path, ".bz2")) bz = Bz2::get().open(path.c_str(), "wb");
        else plain = fopen(path.c_str(), "wb");
        return plain || gz || bz;
    }
    bool is_open() const { return plain || gz || bz; }
    void write(const char *data, size_t n) {
        for (size_t done = 0; done < n && !failed;) {
            const unsigned chunk = (unsigned)std::min<size_t>(n - done, 1u << 30);
            if (gz) failed = gzwrite(gz, data + done, chunk) != (int)chunk;
            else if (bz) failed = Bz2::get().write(bz, const_cast<char *>(data + done), (int)chunk) != (int)chunk;
            else failed = fwrite(data + done, 1, chunk, plain) != chunk;
            done += chunk;
        }
    }
    bool close() {                       // true if everything was written
        if (gz) failed = (gzclose(gz) != Z_OK) || failed;
        if (bz) Bz2::get().close(bz);
        if (plain) failed = (fclose(plain) != 0) || failed;
        gz = nullptr;
        bz = nullptr;
        plain = nullptr;
        return !failed;
    }
};

}  // namespace textio
}  // namespace rsq
