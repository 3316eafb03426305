// rsq_host.cpp -- RSQP container reader, profile loading and post-load edits, FASTA reader.
#include <stdio.h>
#include "rsq_host.h"
#include "rsq_textio.h"

#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <exception>
#include <memory>
#include <mutex>

#include <fstream>
#include <thread>

#include "rsq_core.h"

namespace rsq {

// ------------------------------------------------------------------------------------------------ options
Options &options() {
    static Options o;
    return o;
}
namespace {
struct OptionEntry {
    const char *name;
    int64_t Options::*field;
};
const OptionEntry kOptionTable[] = {
    {"fill_mode", &Options::fill_mode},         {"image_tiles", &Options::image_tiles},     {"rate_rows", &Options::rate_rows},
    {"no_indel_skip", &Options::no_indel_skip}, {"force_exact", &Options::force_exact},     {"min_quality_quads", &Options::min_quality_quads},
    {"trace_plan", &Options::trace_plan},       {"trace_prepare", &Options::trace_prepare},
    {"bias_window", &Options::bias_window},     {"window_chunks", &Options::window_chunks}, {"serial_fasta", &Options::serial_fasta},   {"trace_load", &Options::trace_load},
    {"chain_chunk", &Options::chain_chunk},     {"chain_warmup", &Options::chain_warmup},
    {"serial_parse", &Options::serial_parse},   {"parse_stretch", &Options::parse_stretch}, {"mapped_parses", &Options::mapped_parses},
    {"fasta_stretch", &Options::fasta_stretch}, {"overlap", &Options::overlap}, {"specialize", &Options::specialize}, {"job_write_direct", &Options::job_write_direct},             {"job_chunk_bytes", &Options::job_chunk_bytes}, {"host_gzip", &Options::host_gzip}, {"fill_waves", &Options::fill_waves}, {"fasta_no_stage", &Options::fasta_no_stage},
    {"hiprtc_by_name", &Options::hiprtc_by_name},
};
}  // namespace
bool set_option(const char *name, int64_t value) {
    for (const OptionEntry &e : kOptionTable)
        if (0 == strcmp(name, e.name)) {
            options().*(e.field) = value;
            return true;
        }
    return false;
}
bool get_option(const char *name, int64_t *value) {
    for (const OptionEntry &e : kOptionTable)
        if (0 == strcmp(name, e.name)) {
            *value = options().*(e.field);
            return true;
        }
    return false;
}
const char *option_names() {
    static const std::string names = [] {
        std::string s;
        for (const OptionEntry &e : kOptionTable) s += (s.empty() ? "" : " ") + std::string(e.name);
        return s;
    }();
    return names.c_str();
}

// ---------------------------------------------------------------------------------------------- container
static size_t pad8(size_t n) { return (8 - (n & 7)) & 7; }

Container::Container(const std::string &path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw Error("cannot open profile '" + path + "'");
    std::streamsize sz = f.tellg();
    f.seekg(0);
    buf_.resize((size_t)sz);
    if (!f.read(reinterpret_cast<char *>(buf_.data()), sz)) throw Error("cannot read profile '" + path + "'");
    if (buf_.size() < 16 || memcmp(buf_.data(), "RSQPROF1", 8) != 0) throw Error("'" + path + "' is not an RSQP profile container");
    uint32_t version, n;
    memcpy(&version, &buf_[8], 4);
    memcpy(&n, &buf_[12], 4);
    if (version != 1) throw Error("unsupported RSQP version");
    static const size_t kSize[7] = {1, 2, 4, 8, 4, 8, 8};
    size_t pos = 16;
    for (uint32_t i = 0; i < n; ++i) {
        if (pos + 4 > buf_.size()) throw Error("truncated RSQP container");
        uint16_t ln;
        memcpy(&ln, &buf_[pos], 2);
        std::string name(reinterpret_cast<const char *>(&buf_[pos + 2]), ln);
        Array a;
        a.dtype = buf_[pos + 2 + ln];
        int ndim = buf_[pos + 3 + ln];
        if (a.dtype > 6 || ndim > 4) throw Error("corrupt RSQP container");
        size_t head = (size_t)ln + 4;
        pos += head + pad8(head);
        a.count = 1;
        for (int d = 0; d < ndim; ++d) {
            uint64_t dim;
            memcpy(&dim, &buf_[pos], 8);
            a.dims.push_back(dim);
            a.count *= dim;
            pos += 8;
        }
        size_t nbytes = a.count * kSize[a.dtype];
        if (pos + nbytes > buf_.size()) throw Error("truncated RSQP container");
        a.data = &buf_[pos];
        pos += nbytes + pad8(nbytes);
        arrays_[name] = a;
    }
}

const Array &Container::get(const std::string &name) const {
    auto it = arrays_.find(name);
    if (it == arrays_.end()) throw Error("profile misses array '" + name + "'");
    return it->second;
}

// ------------------------------------------------------------------------------------------------ profile
std::vector<double> discrete_cp(const uint64_t *w, size_t n) {
    std::vector<double> cp(n ? n : 1, 1.0);
    if (n < 2) return cp;
    double sum = 0.0;
    for (size_t i = 0; i < n; ++i) sum += (double)w[i];
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) {
        acc += (double)w[i] / sum;
        cp[i] = acc;
    }
    cp[n - 1] = 1.0;
    return cp;
}

void HostTable::modify_par0(uint32_t par0_index, double multiplier) {
    size_t col = 0;
    while (col < par0.size() && par0[col] != par0_index) ++col;
    if (col >= par0.size()) return;
    for (size_t i = col; i < dim2[0].size(); i += par0.size()) dim2[0][i] *= multiplier;
}

void HostTable::set_par0(uint32_t par0_index) {
    par0.assign(1, par0_index);
    for (uint32_t n = 0; n < nm; ++n) dim2[n].assign(to[n] - from[n], 1.0);
}

template <class T>
static Vect<T> load_vect(const Container &c, const std::string &name, int dtype) {
    Vect<T> v;
    v.v = c.vec<T>(name, dtype);
    v.from = c.scalar<uint64_t>(name + ".from", 3);
    return v;
}

static HostTable load_table(const Container &c, const std::string &prefix, uint32_t nm) {
    HostTable t;
    t.nm = nm;
    t.par0 = c.vec<uint32_t>("tab." + prefix + ".par0", 2);
    std::vector<uint32_t> lim = c.vec<uint32_t>("tab." + prefix + ".limits", 2);
    std::vector<double> d = c.vec<double>("tab." + prefix + ".dim2", 6);
    if (lim.size() != 2 * (size_t)nm) throw Error("table '" + prefix + "': wrong number of limits");
    size_t pos = 0;
    for (uint32_t n = 0; n < nm; ++n) {
        t.from[n] = lim[2 * n];
        t.to[n] = lim[2 * n + 1];
        if (t.to[n] < t.from[n]) throw Error("table '" + prefix + "': inverted limits");
        size_t cnt = t.par0.empty() ? 0 : (size_t)(t.to[n] - t.from[n]) * t.par0.size();
        if (pos + cnt > d.size()) throw Error("table '" + prefix + "': dim2 too short");
        t.dim2[n].assign(d.begin() + pos, d.begin() + pos + cnt);
        pos += cnt;
    }
    for (uint32_t v : t.par0)
        if (v > 255) throw Error("table '" + prefix + "': outcome value above 255");
    return t;
}

Profile Profile::load(const std::string &path) {
    Container c(path);
    Profile p;
    p.phred_offset = c.scalar<uint8_t>("phred_quality_offset", 0);
    p.corrected_coverage = c.scalar<double>("corrected_coverage", 6);
    p.max_len_deletion = c.scalar<uint16_t>("errors.max_len_deletion", 1);
    p.reset_distance = c.scalar<uint32_t>("coverage.reset_distance", 2);
    for (int seg = 0; seg < 2; ++seg) {
        const std::string s = std::to_string(seg);
        p.read_lengths[seg] = load_vect<uint64_t>(c, "read_lengths." + s, 3);
        for (uint64_t x : p.read_lengths[seg].v) p.total_number_reads += x;       // DataStats.cpp:698-700
        HostRlByFl &r = p.rl_by_fl[seg];
        r.from = c.scalar<uint64_t>("rl_by_fl." + s + ".from", 3);
        r.row_ptr = c.vec<uint32_t>("rl_by_fl." + s + ".row_ptr", 2);
        r.row_from = c.vec<uint32_t>("rl_by_fl." + s + ".row_from", 2);
        r.values = c.vec<uint64_t>("rl_by_fl." + s + ".values", 3);
        if (c.has("rl_by_fl_nonmapped." + s + ".values")) r.non_mapped = c.vec<uint64_t>("rl_by_fl_nonmapped." + s + ".values", 3);
        else r.non_mapped.assign(r.values.size(), 0);
        HostAdapters &a = p.adapters[seg];
        a.seqs = c.vec<uint8_t>("adapters." + s + ".seqs", 0);
        a.seq_ptr = c.vec<uint32_t>("adapters." + s + ".seq_ptr", 2);
        a.counts = c.vec<uint64_t>("adapters." + s + ".counts", 3);
        a.significant = c.vec<uint64_t>("adapters." + s + ".significant_counts", 3);
        a.cut_ptr = c.vec<uint32_t>("adapters." + s + ".start_cut_ptr", 2);
        a.cut_from = c.vec<uint32_t>("adapters." + s + ".start_cut_from", 2);
        a.cut = c.vec<uint64_t>("adapters." + s + ".start_cut", 3);
        if (a.seq_ptr.empty() || a.counts.size() != a.n() || a.significant.size() != a.n() || a.cut_ptr.size() != a.seq_ptr.size())
            throw Error("inconsistent adapter arrays");
        for (uint8_t b : a.seqs)
            if (b > 3) throw Error("adapter sequences must not contain N");
    }
    p.tiles = c.vec<uint16_t>("tiles.tiles", 1);
    p.tile_abundance = c.vec<uint64_t>("tiles.abundance", 3);
    if (p.tiles.empty() || p.tiles.size() != p.tile_abundance.size()) throw Error("inconsistent tile arrays");
    p.polya = load_vect<uint64_t>(c, "adapters.polya_tail_length", 3);
    std::vector<uint64_t> ob = c.vec<uint64_t>("adapters.overrun_bases", 3);
    for (int i = 0; i < 5; ++i) p.overrun_bases[i] = ob.at(i);
    p.insert_lengths = load_vect<uint64_t>(c, "frag.insert_lengths", 3);
    p.insert_lengths_bias = load_vect<double>(c, "frag.insert_lengths_bias", 6);
    p.gc_bias = load_vect<double>(c, "frag.gc_bias", 6);
    p.sur_bias = c.vec<double>("frag.sur_bias", 6);
    if (p.sur_bias.size() != (size_t)kSurBlocks * kSurSize) throw Error("frag.sur_bias must hold 3 x 4^10 values");
    std::vector<double> disp = c.vec<double>("frag.dispersion_parameters", 6);
    p.dispersion[0] = disp.at(0);
    p.dispersion[1] = disp.at(1);
    p.ref_seq_bias = c.vec<double>("frag.ref_seq_bias", 6);

    const uint32_t nt = p.n_tiles();
    for (uint32_t seg = 0; seg < 2; ++seg)
        for (uint32_t tile = 0; tile < nt; ++tile) {
            const std::string st = std::to_string(seg) + "." + std::to_string(tile);
            p.seq_quality.push_back(load_table(c, "seq_quality." + st, 3));
            for (uint32_t base = 0; base < 4; ++base) {
                p.quality.push_back(load_table(c, "quality." + st + "." + std::to_string(base), 4));
                for (uint32_t dom = 0; dom < 5; ++dom) p.base_call.push_back(load_table(c, "base_call." + st + "." + std::to_string(base) + "." + std::to_string(dom), 4));
            }
        }
    for (uint32_t base = 0; base < 4; ++base)
        for (uint32_t x = 0; x < 5; ++x)
            for (uint32_t y = 0; y < 5; ++y)
                p.dom_error.push_back(load_table(c, "dom_error." + std::to_string(base) + "." + std::to_string(x) + "." + std::to_string(y), 3));
    for (uint32_t base = 0; base < 4; ++base)
        for (uint32_t x = 0; x < 5; ++x) p.error_rate.push_back(load_table(c, "error_rate." + std::to_string(base) + "." + std::to_string(x), 3));
    for (uint32_t type = 0; type < 2; ++type)
        for (uint32_t call = 0; call < 6; ++call) p.indels.push_back(load_table(c, "indels." + std::to_string(type) + "." + std::to_string(call), 3));
    return p;
}

void Profile::change_error_rate(double multiplier) {
    for (size_t i = 0; i < base_call.size(); ++i) base_call[i].modify_par0((uint32_t)((i / 5) % 4), 1.0 / multiplier);
}
void Profile::remove_substitution_errors() {
    for (size_t i = 0; i < base_call.size(); ++i) base_call[i].set_par0((uint32_t)((i / 5) % 4));
}
void Profile::remove_indel_errors() {
    for (HostTable &t : indels) t.set_par0(0);
}

// ---------------------------------------------------------------------------------------------- reference
namespace {
// plain, gzip- or bzip2-compressed text (rsq_textio.h; SeqAn picks the format the same way)
struct GzLines {
    textio::Reader f;
    std::vector<char> buf;
    size_t at = 0, have = 0;
    explicit GzLines(const std::string &path) : buf(1 << 20) {
        if (!f.open(path)) throw Error("Could not open " + path + " for reading.");
    }
    bool getline(std::string &line) {
        line.clear();
        for (;;) {
            if (at == have) {
                const int n = f.read(buf.data(), (unsigned)buf.size());
                if (n < 0) throw Error("read error in a compressed input file");
                if (n == 0) return !line.empty();
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
};
}  // namespace

namespace {
struct BaseCodes {
    uint8_t lut[256];
    BaseCodes() {
        memset(lut, 4, sizeof lut);            // every IUPAC code that is not ACGT becomes N (IupacString -> Dna5String)
        const char *acgt = "ACGT";
        for (int i = 0; i < 4; ++i) {
            lut[(uint8_t)acgt[i]] = (uint8_t)i;
            lut[(uint8_t)(acgt[i] + 32)] = (uint8_t)i;
        }
        lut[(uint8_t)'U'] = lut[(uint8_t)'u'] = 3;
    }
};

template <class F>
void on_threads(size_t n_items, unsigned n_threads, F f) {      // f(item) for every item, items handed out in order
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    std::exception_ptr failed;
    std::mutex m;
    auto work = [&] {
        try {
            for (size_t i; (i = next.fetch_add(1)) < n_items;) f(i);
        } catch (...) {
            std::lock_guard<std::mutex> g(m);
            failed = std::current_exception();
        }
    };
    for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
    if (failed) std::rethrow_exception(failed);
}

// A plain-text FASTA file in memory, read by several threads: the header lines are found first, then the text between them is cut into
// stretches that are counted and converted independently (a base's code does not depend on the line it stands in).  The same rules as the
// line reader below: a line starts a record iff its first character is '>', one '\r' before the line end is dropped, blanks and tabs are
// skipped, empty lines ignored.  Returns false (nothing read) for anything that is not a regular uncompressed file.
// option trace_load: stage times of the readers on stderr
struct LoadLap {
    const char *who;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void operator()(const char *what) {
        if (!options().trace_load) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "load: %-10s %-22s %8.3f s\n", who, what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};
bool read_fasta_mapped(const std::string &path, Reference &r) {
    LoadLap lap{"fasta"};
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    // small files: the line reader.  Option fasta_stretch (tests): the stretch length in bytes, also the size from which a file is mapped
    const int64_t stretch_opt = options().fasta_stretch;
    const size_t kStretch = stretch_opt > 0 ? (size_t)stretch_opt : (size_t)8u << 20, min_size = stretch_opt > 0 ? 4 : (size_t)1 << 20;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || (size_t)st.st_size < min_size) {
        close(fd);
        return false;
    }
    const size_t n = (size_t)st.st_size;
    void *map = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return false;
    struct Unmap {
        void *p;
        size_t n;
        ~Unmap() { munmap(p, n); }
    } unmap{map, n};
    const char *d = static_cast<const char *>(map);
    if (((uint8_t)d[0] == 0x1f && (uint8_t)d[1] == 0x8b) || !memcmp(d, "BZh", 3)) return false;      // gzip, bzip2
    madvise(map, n, MADV_SEQUENTIAL);
    const unsigned hw = std::thread::hardware_concurrency(), n_threads = std::max(1u, std::min(hw ? hw : 4u, 32u));
    const size_t n_scan = (n + kStretch - 1) / kStretch;
    std::vector<std::vector<size_t>> found(n_scan);
    on_threads(n_scan, n_threads, [&](size_t i) {
        const size_t lo = i * kStretch, hi = std::min(n, lo + kStretch);
        for (const char *p = d + lo; p < d + hi && (p = (const char *)memchr(p, '>', (size_t)(d + hi - p))); ++p)
            if (p == d || p[-1] == '\n') found[i].push_back((size_t)(p - d));
    });
    lap("map + find headers");
    std::vector<size_t> header;
    for (const auto &f : found) header.insert(header.end(), f.begin(), f.end());
    if (header.empty()) return false;                                  // the line reader words the error
    for (size_t p = 0; p < header[0]; ++p)
        if (d[p] != '\n' && d[p] != '\r') return false;

    struct Stretch {
        size_t record, lo, hi, kept;
    };
    std::vector<Stretch> stretches;
    std::vector<size_t> first_stretch(header.size() + 1);
    r.names.resize(header.size());
    r.codes.resize(header.size());
    for (size_t i = 0; i < header.size(); ++i) {
        const size_t end = i + 1 < header.size() ? header[i + 1] : n;
        const char *eol = (const char *)memchr(d + header[i], '\n', end - header[i]);
        size_t name_end = eol ? (size_t)(eol - d) : end;
        const size_t body = eol ? name_end + 1 : end;
        if (name_end > header[i] + 1 && d[name_end - 1] == '\r') --name_end;
        r.names[i].assign(d + header[i] + 1, name_end - header[i] - 1);
        first_stretch[i] = stretches.size();
        for (size_t lo = body; lo < end; lo += kStretch) stretches.push_back(Stretch{i, lo, std::min(end, lo + kStretch), 0});
    }
    first_stretch[header.size()] = stretches.size();
    auto dropped = [&](size_t p) {                                     // not a base: line ends, blanks, the '\r' of a "\r\n" (or before the end of the file)
        const char c = d[p];
        return c == '\n' || c == ' ' || c == '\t' || (c == '\r' && (p + 1 == n || d[p + 1] == '\n'));
    };
    on_threads(stretches.size(), n_threads, [&](size_t i) {
        Stretch &s = stretches[i];
        size_t kept = 0;
        for (size_t p = s.lo; p < s.hi; ++p) kept += !dropped(p);
        s.kept = kept;
    });
    lap("count bases");
    std::vector<size_t> offset(stretches.size());
    for (size_t i = 0; i < header.size(); ++i) {
        size_t total = 0;
        for (size_t k = first_stretch[i]; k < first_stretch[i + 1]; ++k) {
            offset[k] = total;
            total += stretches[k].kept;
        }
        if (total > 0xFFFFFFFFull) throw Error("reference sequence longer than 2^32-1 bases");
    }
    on_threads(header.size(), n_threads, [&](size_t i) {                // the allocation (and its zero fill) of one sequence per thread
        const size_t k = first_stretch[i + 1];
        r.codes[i].resize(k > first_stretch[i] ? offset[k - 1] + stretches[k - 1].kept : 0);
    });
    lap("allocate");
    static const BaseCodes codes;
    on_threads(stretches.size(), n_threads, [&](size_t i) {
        const Stretch &s = stretches[i];
        uint8_t *out = r.codes[s.record].data() + offset[i];
        for (size_t p = s.lo; p < s.hi; ++p)
            if (!dropped(p)) *out++ = codes.lut[(uint8_t)d[p]];
    });
    lap("convert");
    return true;
}
}  // namespace

Reference Reference::read_fasta(const std::string &path) {
    Reference r;
    if (!options().serial_fasta && read_fasta_mapped(path, r)) return r;
    r = Reference();
    GzLines f(path);
    static const BaseCodes codes;
    const uint8_t *lut = codes.lut;
    std::string line;
    while (f.getline(line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') {
            r.names.push_back(line.substr(1));
            r.codes.emplace_back();
        } else {
            if (r.codes.empty()) throw Error(path + " does not start with a FASTA header");
            std::vector<uint8_t> &c = r.codes.back();
            const size_t old = c.size();
            if (c.capacity() < old + line.size()) c.reserve(std::max(c.capacity() * 2, old + line.size()));
            c.resize(old + line.size());
            uint8_t *d = c.data() + old;
            for (char ch : line)
                if (ch != ' ' && ch != '\t') *d++ = lut[(uint8_t)ch];
            c.resize((size_t)(d - c.data()));
        }
    }
    if (r.codes.empty()) throw Error(path + " does not contain any reference sequences.");
    for (const auto &c : r.codes)
        if (c.size() > 0xFFFFFFFFull) throw Error("reference sequence longer than 2^32-1 bases");
    return r;
}

bool Reference::has_n() const {
    for (const auto &c : codes) {
        size_t i = 0;
        for (uint64_t w; i + 8 <= c.size(); i += 8) {
            memcpy(&w, c.data() + i, 8);
            if (w & 0x0404040404040404ull) return true;
        }
        for (; i < c.size(); ++i)
            if (c[i] > 3) return true;
    }
    return false;
}

uint64_t Reference::total_size() const {
    uint64_t s = 0;
    for (const auto &c : codes) s += c.size();
    return s;
}

std::string Reference::first_part(size_t i) const {
    const std::string &n = names.at(i);
    return n.substr(0, n.find(' '));
}

// Reference.cpp:813-: N's are replaced so that every simulation from a position sees the same base.  The random
// source is Philox keyed by (seed, sequence, position) instead of one mt19937_64 stream.  Stretches of at least
// kMinNToReplaceNWithRepeat (100) N, which the reference fills with a 4-base repeat, are rejected for now.
// Reference.cpp:813-886.  Stretches of fewer than kMinNToReplaceNWithRepeat (100) N become uniform bases; longer stretches are filled
// with a four-base repeat taken from the flanks (two bases after and two before the stretch; four after it at the sequence start,
// four before it at the end; drawn when the stretch is the whole sequence).  The draw of position p is Philox word w0 & 3 of
// counter (p, sequence, 0, 5<<28), also when p lies in a flank that is itself N.
void Reference::replace_n(uint64_t seed) {
    LoadLap lap{"replace_n"};
    // the draws are keyed by (seed, sequence, position): sequences are independent, one per thread
    const unsigned hw = std::thread::hardware_concurrency();
    on_threads(codes.size(), std::max(1u, std::min(hw ? hw : 4u, 32u)), [&](size_t s) {
        std::vector<uint8_t> &c = codes[s];
        auto draw = [&](size_t pos) { return (uint8_t)(philox(seed, (uint32_t)pos, (uint32_t)s, 0, kDomReplaceN << 28).w0 & 3u); };
        for (size_t start = 0; start < c.size();) {
            if (c[start] <= 3) {
                ++start;
                for (uint64_t w; start + 8 <= c.size(); start += 8) {              // eight bases at a time while none of them is N (code 4)
                    memcpy(&w, c.data() + start, 8);
                    if (w & 0x0404040404040404ull) break;
                }
                continue;
            }
            size_t end = start;
            while (++end < c.size() && c[end] > 3) {}
            if (end - start < 100) {
                for (size_t pos = start; pos < end; ++pos) c[pos] = draw(pos);
            } else {
                uint8_t rep[4];
                if (2 > start) {
                    if (end + 4 > c.size()) {
                        for (size_t k = 0; k < 4; ++k) rep[k] = draw(start + k);                 // the stretch is the complete sequence
                    } else {
                        for (size_t k = 0; k < 4; ++k) rep[k] = c[end + k];                       // N's at the start of the sequence
                        for (size_t k = 4; --k;)
                            if (rep[k] > 3) rep[k] = draw(end + k);
                    }
                } else if (end + 2 > c.size()) {
                    if (4 > start) {
                        for (size_t k = 0; k < 4; ++k) rep[k] = draw(start + k);
                    } else {
                        for (size_t k = 0; k < 4; ++k) rep[k] = c[start - 4 + k];                 // N's at the end of the sequence
                    }
                } else {                                                                        // N's in the middle
                    rep[0] = c[end];
                    rep[1] = c[end + 1];
                    rep[2] = c[start - 2];
                    rep[3] = c[start - 1];
                    if (rep[1] > 3) rep[1] = draw(end + 1);
                }
                for (size_t pos = start; pos < end; ++pos) c[pos] = rep[(pos - start) % 4];
            }
            start = end;
        }
    });
    lap("all sequences");
}

// ------------------------------------------------------------------- systematic-error profile (FASTQ) and ref-bias file
uint8_t compress_sys_error_rate(uint8_t q) {                     // Simulator.cpp:2569-2574
    if (86 < q) q = (uint8_t)(q - (q - 85) / 2);
    return q;
}
uint8_t expand_sys_error_rate(uint8_t r) {                       // Simulator.h:329-332
    if (86 < r) r = (uint8_t)(r + (r - 86));
    return r;
}
std::string sys_error_fastq_record(const std::string &id, const uint8_t *dom, const uint8_t *rate, size_t n) {
    std::string out;
    out.reserve(id.size() + 2 * n + 8);
    out += '@';
    out += id;
    out += '\n';
    for (size_t i = 0; i < n; ++i) out += "ACGTN"[dom[i] < 4 ? dom[i] : 4];
    out += "\n+\n";
    for (size_t i = 0; i < n; ++i) out += (char)(compress_sys_error_rate(rate[i]) + 33);
    out += '\n';
    return out;
}
std::vector<SysErrorRecord> parse_sys_error_fastq(const std::string &text) {
    std::vector<SysErrorRecord> recs;
    size_t pos = 0;
    auto line = [&](std::string &l) {
        if (pos >= text.size()) return false;
        size_t e = text.find('\n', pos);
        if (e == std::string::npos) e = text.size();
        l.assign(text, pos, e - pos);
        if (!l.empty() && l.back() == '\r') l.pop_back();
        pos = e + 1;
        return true;
    };
    std::string id, seq, plus, qual;
    while (line(id)) {
        if (id.empty()) continue;
        if (id[0] != '@' || !line(seq) || !line(plus) || plus.empty() || plus[0] != '+' || !line(qual))
            throw Error("systematic error profile: malformed FASTQ record " + std::to_string(recs.size()));
        if (seq.size() != qual.size()) throw Error("systematic error profile '" + id.substr(1) + "': sequence and quality lengths differ");
        SysErrorRecord r;
        r.id = id.substr(1);
        r.dom.resize(seq.size());
        r.rate.resize(seq.size());
        for (size_t i = 0; i < seq.size(); ++i) {
            switch (seq[i]) {
                case 'A': case 'a': r.dom[i] = 0; break;
                case 'C': case 'c': r.dom[i] = 1; break;
                case 'G': case 'g': r.dom[i] = 2; break;
                case 'T': case 't': r.dom[i] = 3; break;
                default: r.dom[i] = 4;
            }
            r.rate[i] = expand_sys_error_rate((uint8_t)(qual[i] - 33));
        }
        recs.push_back(std::move(r));
    }
    return recs;
}
std::string read_text_file(const std::string &path) {                       // plain, gzip or bzip2
    textio::Reader f;
    if (!f.open(path)) throw Error("Could not open '" + path + "' for reading.");
    std::string text;
    std::vector<char> buf(1 << 20);
    int n;
    while ((n = f.read(buf.data(), (unsigned)buf.size())) > 0) text.append(buf.data(), (size_t)n);
    if (n < 0) throw Error("Could not read '" + path + "'.");
    return text;
}
void write_text_file(const std::string &path, const std::string &text) {   // gzip / bzip2 when the name ends in .gz / .bz2, like SeqAn's SeqFileOut
    textio::Writer f;
    if (!f.open(path)) throw Error("Could not open '" + path + "' for writing.");
    f.write(text.data(), text.size());
    if (!f.close()) throw Error("Could not write '" + path + "'.");
}

std::vector<double> read_ref_bias_file(const std::string &path, const std::vector<std::string> &first_names) {
    const std::string text = read_text_file(path);                      // "Unable to open reference bias file" otherwise
    std::map<std::string, uint32_t> ids;
    for (uint32_t i = 0; i < first_names.size(); ++i) ids.emplace(first_names[i], i);      // emplace keeps the first of duplicates, like the reference
    std::vector<double> bias(first_names.size(), 0.0);
    std::vector<bool> found(first_names.size(), false);
    std::string errors;
    uint32_t nline = 0;
    bool empty_line = false;
    size_t pos = 0;
    while (pos < text.size()) {
        size_t e = text.find('\n', pos);
        if (e == std::string::npos) e = text.size();
        const std::string line = text.substr(pos, e - pos);
        pos = e + 1;
        if (empty_line) {
            errors += "Line " + std::to_string(nline) + " is empty. ";
            continue;
        }
        ++nline;
        if (line.empty()) {
            empty_line = true;
            continue;
        }
        const size_t sep = line.find_last_of(" \t");
        if (sep == std::string::npos) {
            errors += "Line " + std::to_string(nline) + " does not have a field separator. ";
            continue;
        }
        double b = 0.0;
        try {
            b = std::stod(line.substr(sep + 1));
        } catch (const std::exception &) {
            errors += "Could not convert bias in line " + std::to_string(nline) + " into double. ";
            b = 0.0;
        }
        if (0.0 > b) errors += "Bias in line " + std::to_string(nline) + " is negative. ";
        size_t id_len = line.find(' ');
        if (id_len == std::string::npos) id_len = sep;
        size_t id_start = 0;
        if ('>' == line[0]) {
            ++id_start;
            --id_len;
        }
        auto it = ids.find(line.substr(id_start, id_len));
        if (it != ids.end()) {
            bias[it->second] = b;
            found[it->second] = true;
        }
    }
    for (uint32_t i = 0; i < first_names.size(); ++i)
        if (!found[i]) errors += "Could not find bias for reference sequence " + first_names[i] + ". ";
    if (!errors.empty()) throw Error("reference bias file " + path + ": " + errors);
    return bias;
}

// -------------------------------------------------------------------------------------------- methylation (BED)
namespace {
// A regular uncompressed file of at least `min_size` bytes that ends with a line end, mapped for several readers.
struct MappedText {
    const char *d = nullptr;
    size_t n = 0;
    bool open(const std::string &path, size_t min_size) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || (size_t)st.st_size < std::max<size_t>(min_size, 4)) {
            close(fd);
            return false;
        }
        void *map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (map == MAP_FAILED) return false;
        d = static_cast<const char *>(map);
        n = (size_t)st.st_size;
        madvise(map, n, MADV_SEQUENTIAL);
        return '\n' == d[n - 1] && !((uint8_t)d[0] == 0x1f && (uint8_t)d[1] == 0x8b) && 0 != memcmp(d, "BZh", 3);
    }
    ~MappedText() {
        if (d) munmap(const_cast<char *>(d), n);
    }
    // [lo, hi) of piece i of about `stretch` bytes, cut at line starts
    std::vector<size_t> cut(size_t stretch) const {
        std::vector<size_t> at{0};
        for (size_t p = stretch; p < n; p += stretch) {
            const char *e = (const char *)memchr(d + p, '\n', n - p);
            const size_t start = e ? (size_t)(e - d) + 1 : n;
            if (start > at.back() && start < n) at.push_back(start);
        }
        at.push_back(n);
        return at;
    }
};
inline bool blank(char c) { return ' ' == c || '\t' == c; }

// strtod for the text at p (a digit or '.'), which some line end follows: plain decimals of at most 15 digits are an integer over a power of ten, both exact
// in binary64, and one IEEE division rounds their quotient correctly -- the value strtod returns; everything else goes to strtod.  False where std::stod would throw.
inline bool decimal_at(const char *p, const char *&end, double &v) {
    static const double pow10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
    const char *q = p;
    uint64_t mant = 0;
    unsigned digits = 0, after = 0;
    for (; *q >= '0' && *q <= '9'; ++q, ++digits) mant = mant * 10 + (uint64_t)(*q - '0');
    if ('.' == *q)
        for (++q; *q >= '0' && *q <= '9'; ++q, ++digits, ++after) mant = mant * 10 + (uint64_t)(*q - '0');
    if (digits && digits <= 15 && (blank(*q) || '\n' == *q)) {
        end = q;
        v = (double)mant / pow10[after];
        return true;
    }
    char *e;
    errno = 0;
    v = strtod(p, &e);
    end = e;
    return e != p && errno != ERANGE;
}

// The methylation file read by several threads, for the file every tool writes: lines `name start end rate [rate ...]` separated by blanks or tabs, plain
// decimal numbers, sequences in the order of the reference, regions in order.  Pieces of the file are parsed independently into (start, end, rates) and
// runs of one name; the runs are then matched to the reference's sequences and the pieces checked against the sequence lengths and copied to their place.
// Anything else -- a '\r', a number in another notation, a region out of order or out of range, an unknown or repeated name, a change in the number of
// alleles -- returns false with nothing kept, and the line reader below reads the file from its start: the messages, and what is silently ignored, are its.
bool read_methylation_mapped(const std::string &path, const std::vector<std::string> &first_names, const std::vector<uint32_t> &seq_len, uint32_t num_alleles_ref, Methylation &m) {
    const int64_t stretch_opt = options().parse_stretch;
    const size_t stretch = stretch_opt > 0 ? (size_t)stretch_opt : (size_t)4u << 20;
    MappedText text;
    if (!text.open(path, stretch_opt > 0 ? 4 : (size_t)1 << 20)) return false;
    const char *d = text.d;
    size_t begin = 0;                                                    // empty and track lines before the first record (Reference.cpp:1150)
    while (begin < text.n) {
        const size_t len = (size_t)((const char *)memchr(d + begin, '\n', text.n - begin) - (d + begin));
        if (len && !(len >= 5 && 0 == memcmp(d + begin, "track", 5))) break;
        begin += len + 1;
    }
    if (begin >= text.n) return false;
    std::vector<size_t> at = text.cut(stretch);
    at.erase(at.begin(), std::upper_bound(at.begin(), at.end(), begin));
    at.insert(at.begin(), begin);
    struct Run {
        const char *name;
        size_t name_len, first_region, n_regions;
        uint32_t alleles, seq;
    };
    struct Piece {
        std::vector<uint32_t> first, second;
        std::vector<double> rate;                                        // the rates of a region side by side
        std::vector<Run> runs;
        bool plain = true;
        size_t out = 0;                                                  // where its first run continues in the sequence's arrays
    };
    std::vector<Piece> pieces(at.size() - 1);
    const unsigned hw = std::thread::hardware_concurrency(), n_threads = std::max(1u, std::min(hw ? hw : 4u, 32u));
    on_threads(pieces.size(), n_threads, [&](size_t i) {
        Piece &pc = pieces[i];
        const size_t guess = (at[i + 1] - at[i]) / 24;
        pc.first.reserve(guess);
        pc.second.reserve(guess);
        pc.rate.reserve(guess * num_alleles_ref);
        auto number = [](const char *&p, uint64_t &v) {                  // digits followed by a blank
            const char *q = p;
            for (v = 0; *q >= '0' && *q <= '9' && q - p < 11; ++q) v = v * 10 + (uint64_t)(*q - '0');
            const bool fine = q != p && blank(*q) && v <= 0xFFFFFFFFull;
            for (p = q; blank(*p); ++p) {}
            return fine;
        };
        for (const char *p = d + at[i], *const stop = d + at[i + 1]; p < stop;) {
            const char *e = (const char *)memchr(p, '\n', (size_t)(stop - p));
            if (e == p) {                                                // empty lines are ignored
                ++p;
                continue;
            }
            const char *name = p;
            while (!blank(*p) && p < e) ++p;
            const size_t name_len = (size_t)(p - name);
            uint64_t first, second;
            if ('\r' == e[-1] || !name_len || p == e) return void(pc.plain = false);
            while (blank(*p)) ++p;
            if (!number(p, first) || !number(p, second) || second <= first) return void(pc.plain = false);
            uint32_t alleles = 0;
            while (p < e) {
                double v;
                const char *q;
                if (!((*p >= '0' && *p <= '9') || '.' == *p) || !decimal_at(p, q, v) || q > e || !(blank(*q) || q == e) || !(0.0 <= v && v <= 1.0)) return void(pc.plain = false);
                pc.rate.push_back(1.0 - v);                              // the probability of a C->T conversion
                ++alleles;
                for (p = q; blank(*p); ++p) {}
            }
            if (pc.runs.empty() || pc.runs.back().name_len != name_len || 0 != memcmp(pc.runs.back().name, name, name_len))
                pc.runs.push_back(Run{name, name_len, pc.first.size(), 0, alleles, 0});
            else if (first < pc.second.back())
                return void(pc.plain = false);
            if (alleles != pc.runs.back().alleles || !alleles) return void(pc.plain = false);
            ++pc.runs.back().n_regions;
            pc.first.push_back((uint32_t)first);
            pc.second.push_back((uint32_t)second);
            p = e + 1;
        }
    });
    // runs -> sequences, as the line reader walks them: each new name is looked for among the sequences after the last one
    const size_t n = first_names.size();
    std::vector<size_t> regions(n, 0);
    std::vector<uint32_t> alleles(n, 0);
    size_t seq = 0;
    const Run *last = nullptr;
    uint32_t last_end = 0;
    for (Piece &pc : pieces) {
        if (!pc.plain) return false;
        for (Run &r : pc.runs) {
            if (last && last->name_len == r.name_len && 0 == memcmp(last->name, r.name, r.name_len)) {      // the run goes on in the next piece
                if (r.alleles != last->alleles || pc.first[r.first_region] < last_end) return false;
                if (&r == &pc.runs[0]) pc.out = regions[seq];
            } else {
                if (last) ++seq;
                while (seq < n && (first_names[seq].size() != r.name_len || 0 != memcmp(first_names[seq].data(), r.name, r.name_len))) ++seq;
                if (seq == n || (1 != r.alleles && num_alleles_ref != r.alleles)) return false;
                alleles[seq] = r.alleles;
            }
            r.seq = (uint32_t)seq;
            regions[seq] += r.n_regions;
            last = &r;
            last_end = pc.second[r.first_region + r.n_regions - 1];
        }
    }
    if (!last) return false;
    m.first.assign(n, {});
    m.second.assign(n, {});
    m.rate.assign(n, {});
    on_threads(n, n_threads, [&](size_t i) {
        if (!alleles[i]) return;
        m.first[i].resize(regions[i]);
        m.second[i].resize(regions[i]);
        m.rate[i].assign(alleles[i], std::vector<double>(regions[i]));
    });
    std::atomic<bool> in_range{true};
    on_threads(pieces.size(), n_threads, [&](size_t i) {
        const Piece &pc = pieces[i];
        size_t out = pc.out, rates = 0;                                  // the piece's rates lie run after run, a region's alleles side by side
        for (const Run &r : pc.runs) {
            if (&r != &pc.runs[0]) out = 0;                              // only a piece's first run can continue one of the piece before
            for (size_t k = 0; k < r.n_regions; ++k) {
                const size_t src = r.first_region + k;
                if (pc.first[src] >= seq_len[r.seq] || pc.second[src] > seq_len[r.seq]) return void(in_range = false);
                m.first[r.seq][out + k] = pc.first[src];
                m.second[r.seq][out + k] = pc.second[src];
                for (uint32_t a = 0; a < r.alleles; ++a) m.rate[r.seq][a][out + k] = pc.rate[rates + k * r.alleles + a];
            }
            rates += r.n_regions * r.alleles;
        }
    });
    return in_range;
}
}      // namespace

Methylation read_methylation_file(const std::string &path, const std::vector<std::string> &first_names, const std::vector<uint32_t> &seq_len, uint32_t num_alleles_ref) {
    if (!options().serial_parse) {
        Methylation mapped;
        if (read_methylation_mapped(path, first_names, seq_len, num_alleles_ref, mapped)) {
            __atomic_add_fetch(&options().mapped_parses, 1, __ATOMIC_RELAXED);      // files may be loaded from several threads
            return mapped;
        }
    }
    std::ifstream f(path);
    if (!f.is_open()) throw Error("Unable to open methylation file " + path);
    std::string line;
    if (!std::getline(f, line)) throw Error("Methylation file is empty: " + path);
    while ((line.empty() || !line.compare(0, 5, "track")) && std::getline(f, line)) {}       // Reference.cpp:1150 ignore track lines
    if (f.fail()) throw Error("Methylation file only contains track lines: " + path);
    std::string cur_seq = line.substr(0, line.find_first_of(" \t"));
    const size_t n = first_names.size();
    Methylation m;
    m.first.resize(n);
    m.second.resize(n);
    m.rate.resize(n);
    bool file_done = false;
    // std::stoll / std::stod of the text from `at` on, without the copies: false where they would throw (nothing to convert, out of range)
    auto to_int = [](const std::string &text, size_t at, long long &v) {
        if (at >= text.size()) return false;
        char *end;
        errno = 0;
        v = strtoll(text.c_str() + at, &end, 10);
        return end != text.c_str() + at && errno != ERANGE;
    };
    auto to_double = [](const std::string &text, size_t at, double &v) {
        if (at >= text.size()) return false;
        char *end;
        errno = 0;
        v = strtod(text.c_str() + at, &end);
        return end != text.c_str() + at && errno != ERANGE;
    };
    for (size_t i = 0; i < n && !file_done; ++i) {
        if (first_names[i] != cur_seq) continue;                                              // no entries for this sequence
        m.rate[i].resize(num_alleles_ref);
        uint32_t num_alleles = num_alleles_ref;
        while (!f.fail()) {
            size_t a = line.find_first_not_of(" \t", cur_seq.size() + 1), b = line.find_first_of(" \t", a);
            long long v;
            if (!to_int(line, a, v)) throw Error("Could not convert second field to int for line:\n" + line);
            if (m.first[i].empty()) {
                if (v < 0) throw Error("Second field is negative in line:\n" + line);
            } else if (v < (long long)m.second[i].back()) {
                throw Error("Region is overlapping with previous region[" + std::to_string(m.first[i].back()) + " - " + std::to_string(m.second[i].back()) + "] in line:\n" + line);
            }
            if (v >= (long long)seq_len[i]) throw Error("Second field is larger than sequence length:\n" + line);
            const uint32_t region_start = (uint32_t)v;
            a = line.find_first_not_of(" \t", b);
            b = line.find_first_of(" \t", a);
            if (!to_int(line, a, v)) throw Error("Could not convert third field to int for line:\n" + line);
            if (v <= (long long)region_start) throw Error("Third field is smaller than second field in line:\n" + line);
            if (v > (long long)seq_len[i]) throw Error("Third field is larger than sequence length:\n" + line);
            m.first[i].push_back(region_start);
            m.second[i].push_back((uint32_t)v);
            uint32_t allele = 0;
            a = line.find_first_not_of(" \t", b);
            while (a < line.size()) {
                if (allele >= num_alleles)
                    throw Error(allele >= num_alleles_ref ? "More alleles specified than in variant file [" + std::to_string(num_alleles_ref) + "] in line:\n" + line
                                                          : "More alleles specified than in last line [" + std::to_string(num_alleles) + "] in line:\n" + line);
                b = line.find_first_of(" \t", a);
                double d;
                if (!to_double(line, a, d)) throw Error("Could not convert field " + std::to_string(4 + allele) + " to double for line:\n" + line);
                if (0.0 > d || d > 1.0) throw Error("Field " + std::to_string(4 + allele) + " is not between 0 and 1:\n" + line);
                m.rate[i][allele++].push_back(1.0 - d);                                        // the probability of a C->T conversion
                a = line.find_first_not_of(" \t", b);
            }
            if (1 == m.rate[i][0].size()) {                                                    // first entry of this sequence
                if (1 != allele && num_alleles_ref != allele)
                    throw Error(std::to_string(allele) + " alleles specified (must be either 1 or same as in variant file[" + std::to_string(num_alleles_ref) + "]) in line:\n" + line);
                num_alleles = allele;
                m.rate[i].resize(allele);
            } else if (num_alleles != allele) {
                throw Error(std::to_string(allele) + " alleles specified (must be either identical in all lines of a sequence [" + std::to_string(num_alleles) + "]) in line:\n" + line);
            }
            while (std::getline(f, line) && line.empty()) {}                                   // ignore all empty lines
            if (!f.fail()) {
                const size_t sp = line.find_first_of(" \t");
                if (line.compare(0, sp, cur_seq)) {                                            // a new reference sequence
                    cur_seq = line.substr(0, sp);
                    break;
                }
            }
        }
        if (f.fail()) {
            if (!f.eof()) throw Error("Could not read methylation file for reference sequence: " + first_names[i]);
            file_done = true;
        }
    }
    return m;
}

// ---------------------------------------------------------------------------------------------- variants (VCF)
uint32_t Variant::first_allele() const {                              // Reference.h:42-58
    uint32_t first = 0;
    for (uint64_t block : allele) {
        if (block) {
            while (!in_allele(first)) ++first;
            return first;
        }
        first += 64;
    }
    return 0;                                                         // "Variant belonging to no allele"
}

void insert_variant(std::vector<Variant> &variants, uint32_t position, const std::vector<uint8_t> &var_seq, const uint64_t (&allele)[2]) {      // Reference.h:115-139
    Variant v;
    v.position = position;
    v.var_seq = var_seq;
    v.allele[0] = allele[0];
    v.allele[1] = allele[1];
    if (variants.empty()) {
        variants.push_back(v);
        return;
    }
    size_t insert_at = variants.size();                               // at one position: deletion / substitution / insertions by length
    size_t var = variants.size();
    while (0 < var && variants[--var].position == position) {
        if (variants[var].var_seq == var_seq) {                       // already in: only adjust the alleles
            variants[var].allele[0] |= allele[0];
            variants[var].allele[1] |= allele[1];
            return;
        } else if (variants[var].var_seq.size() > var_seq.size()) --insert_at;
    }
    variants.insert(variants.begin() + (ptrdiff_t)insert_at, v);
}

namespace {
void split_tabs(const std::string &line, std::vector<std::string> &f) {      // f is reused from record to record: its strings keep their storage
    size_t at = 0, n = 0;
    for (;; ++n) {
        const size_t e = line.find('\t', at), len = (e == std::string::npos ? line.size() : e) - at;
        if (n == f.size()) f.emplace_back();
        f[n].assign(line, at, len);
        if (e == std::string::npos) break;
        at = e + 1;
    }
    f.resize(n + 1);
}
uint8_t dna5_code(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}
// The records of a variant file, or of a piece of it, one after the other (what Reference::ReadVariants does with a record, Reference.cpp:126-420; the messages
// are the reference's, in its order -- they are part of the command line's behaviour).  A record goes through four stages, each with its own small state:
//   place        sequence id and position, and the file's sort order against the record before
//   reference    the REF column against the sequence
//   genotypes    the sample columns -> for every allele the number of the alternative it carries (0: the reference)
//   alternatives the ALT column cut at its commas; per alternative the set of alleles that carry it, and per reference base of the record what the
//                alternative puts in its place (a substitution, the last base plus inserted ones, or nothing: a deletion) -> insert_variant
struct VcfRecords {
    const std::vector<std::string> &first_names, &contigs;
    const std::vector<std::vector<uint8_t>> &codes;
    const uint32_t A;
    std::vector<std::vector<Variant>> *by_seq;                        // the whole file: the lists per sequence
    std::vector<std::pair<uint32_t, std::vector<Variant>>> runs;      // a piece (by_seq null): its sequences in the order met, which a sorted file meets once each
    std::string errors;
    uint32_t n_errors = 0, records = 0;
    size_t reserve_hint = 0;
    uint32_t first_rid = 0, first_begin = 0;                          // of the first record
    // the order of the file so far: the sequence the records have reached, the sequence and span of the last record that was placed
    uint32_t reached_seq = 0, placed_seq = 0xFFFFFFFFu, placed_begin = 0, placed_end = 0, name_cache = 0xFFFFFFFFu;
    std::vector<uint32_t> carried;                                    // per allele: number of the alternative it carries in this record
    std::vector<uint8_t> ref_codes, put;                              // per record, kept for their storage
    struct Span {
        size_t at, len;
    };
    std::vector<Span> alternatives;

    VcfRecords(const std::vector<std::string> &names, const std::vector<std::string> &contig_names, const std::vector<std::vector<uint8_t>> &seqs, uint32_t num_alleles,
               std::vector<std::vector<Variant>> *lists)
        : first_names(names), contigs(contig_names), codes(seqs), A(num_alleles), by_seq(lists), carried(num_alleles) {}
    void error(const std::string &msg) {
        if (n_errors++ < 20) errors += msg + " ";                     // kMaxErrorsShownPerFile
    }
    std::vector<Variant> &list_of(uint32_t rid) {
        if (by_seq) return (*by_seq)[rid];
        if (runs.empty() || runs.back().first != rid) {
            runs.emplace_back(rid, std::vector<Variant>());
            runs.back().second.reserve(reserve_hint);
        }
        return runs.back().second;
    }
    static std::string at(uint32_t rid, uint32_t pos) { return "Variant starting in reference sequence " + std::to_string(rid) + " at position " + std::to_string(pos); }

    // the record's sequence (names the header did not list get an id behind the known ones) and whether the file is still sorted (:393-401)
    bool place(const std::string &name, uint32_t begin, uint32_t &rid) {
        if (name_cache >= contigs.size() || contigs[name_cache] != name) name_cache = (uint32_t)(std::find(contigs.begin(), contigs.end(), name) - contigs.begin());
        rid = name_cache;
        if (rid < reached_seq) {
            error("Variant file is not properly position sorted. Found sequence id " + std::to_string(rid) + " after id " + std::to_string(reached_seq));
            return false;
        }
        if (rid == reached_seq && placed_seq != 0xFFFFFFFFu && begin < placed_begin) {
            error("Variant file is not properly position sorted. Found in sequence id " + std::to_string(rid) + " position " + std::to_string(begin) + " after position " +
                  std::to_string(placed_begin));
            return false;
        }
        reached_seq = rid;
        return true;
    }
    // the REF column: inside the sequence, free of N, equal to the sequence (:176-194); the record counts as placed whatever the column says
    bool reference_column(uint32_t rid, uint32_t begin, const std::string &ref) {
        if (rid >= first_names.size()) {
            error("Variant starting in reference sequence " + std::to_string(rid) + " does not belong to an existing reference sequence.");
            return false;
        }
        if (begin >= codes[rid].size()) {
            error(at(rid, begin) + " starts after the end of the reference sequence.");
            return false;
        }
        if (placed_seq == rid && begin < placed_end) error(at(rid, begin) + " overlaps with a previous variant.");
        placed_seq = rid;
        placed_begin = begin;
        placed_end = begin + (uint32_t)ref.size();
        ref_codes.resize(ref.size());
        bool ambiguous = false;
        for (size_t k = 0; k < ref.size(); ++k) ambiguous |= (ref_codes[k] = dna5_code(ref[k])) > 3;
        if (ambiguous) error(at(rid, begin) + " has an reference column containing ambiguous bases (e.g. N).");
        else if (placed_end > codes[rid].size() || !std::equal(ref_codes.begin(), ref_codes.end(), codes[rid].begin() + begin))
            error("The specified reference in vcf file '" + ref + "' is not identical with the specified reference sequence " + std::to_string(rid) + " at position " + std::to_string(begin) + ".");
        return true;
    }
    // the sample columns "1|0:..." (:196-262): numbers separated by | or /, up to the first colon; one allele per number, A of them over all columns
    bool genotypes(const std::vector<std::string> &rec) {
        uint32_t filled = 0;
        for (size_t g = 9; g < rec.size(); ++g) {
            const std::string &column = rec[g];
            if (filled >= A) {
                error("Found to many alleles in genotype definition");
                return false;
            }
            uint32_t number = 0;
            bool clean = true;
            for (size_t k = 0; k < column.size() && ':' != column[k]; ++k) {
                const char c = column[k];
                if ('|' == c || '/' == c) {
                    if (filled < A) carried[filled] = number;
                    ++filled;
                    number = 0;
                } else if ('0' <= c && '9' >= c) number = number * 10 + (uint32_t)(c - '0');
                else {
                    error(std::string("Unallowed character '") + c + "' in genotype definition '" + column + "'");
                    clean = false;
                }
            }
            if (filled >= A) {                                       // the reference would write behind its array here
                error("Found to many alleles in genotype definition");
                return false;
            }
            carried[filled++] = clean ? number : 0;
            if (!clean) return false;
        }
        if (filled < A) {
            error("Could not find enough alleles in genotype definition");
            return false;
        }
        return true;
    }
    // the alleles that carry alternative `number` (1-based), one bit each
    std::array<uint64_t, 2> carriers(uint32_t number) const {
        std::array<uint64_t, 2> bits{0, 0};
        for (uint32_t a = 0; a < A; ++a)
            if (carried[a] == number) bits[a / 64] |= (uint64_t)1 << (a % 64);
        return bits;
    }
    // the ALT column (:270-370)
    void alternatives_column(uint32_t rid, uint32_t begin, const std::string &alt) {
        alternatives.clear();
        for (size_t from = 0;;) {
            const size_t comma = alt.find(',', from);
            alternatives.push_back(Span{from, (comma == std::string::npos ? alt.size() : comma) - from});
            if (comma == std::string::npos) break;
            from = comma + 1;
        }
        for (uint32_t a = A; a--;)                                   // numbers beyond the last alternative
            if (carried[a] > alternatives.size())
                error("Variant number " + std::to_string(carried[a]) + " does not exist for sequence id " + std::to_string(rid) + " and position " + std::to_string(begin));
        const size_t ref_len = ref_codes.size();
        for (size_t k = 0; k < ref_len; ++k)
            for (size_t n = 0; n < alternatives.size(); ++n) {
                const std::array<uint64_t, 2> who = carriers((uint32_t)n + 1u);
                if (!(who[0] | who[1])) continue;
                const Span &s = alternatives[n];
                put.clear();
                if (k + 1 == ref_len && k + 1 < s.len) {              // the record's last reference base and what the alternative has beyond it: an insertion
                    for (size_t j = k; j < s.len; ++j) put.push_back(dna5_code(alt[s.at + j]));
                } else if (k < s.len) {                               // base for base
                    const uint8_t b = dna5_code(alt[s.at + k]);
                    if (ref_codes[k] == b) continue;
                    put.push_back(b);
                }                                                     // else the alternative is shorter: the base is deleted
                if (std::any_of(put.begin(), put.end(), [](uint8_t b) { return b > 3; })) {
                    error(at(rid, begin) + " has an alternative column containing ambiguous bases (e.g. N).");
                    continue;
                }
                const uint64_t bits[2] = {who[0], who[1]};
                insert_variant(list_of(rid), begin + (uint32_t)k, put, bits);
            }
    }
    void record(const std::vector<std::string> &rec) {
        const uint32_t begin = (uint32_t)(atoll(rec[1].c_str()) - 1);
        uint32_t rid = 0;
        if (place(rec[0], begin, rid) && reference_column(rid, begin, rec[3]) && genotypes(rec)) alternatives_column(rid, begin, rec[4]);
        if (!records++) {
            first_rid = rid;
            first_begin = begin;
        }
    }
};

// The variant file read by several threads, for a sorted, plain-text file without a single complaint: the header and the first record are read as below, the
// rest is cut into pieces at line starts and every piece's records go through VcfRecords on their own -- a record's variants depend on the records before it
// only through the checks of the order (sequence ids not decreasing, positions not decreasing and not inside the record before), which are made again where
// the pieces meet.  A piece's lists are appended to the sequences' in file order: records of a sorted file share no position, so InsertVariant never reaches
// into the record before.  Any message anywhere returns false with nothing kept, and the line reader reads the file from its start and words the complaint.
bool read_variants_mapped(const std::string &path, const std::vector<std::string> &first_names, const std::vector<std::vector<uint8_t>> &codes, Variants &out) {
    const int64_t stretch_opt = options().parse_stretch;
    const size_t stretch = stretch_opt > 0 ? (size_t)stretch_opt : (size_t)4u << 20;
    MappedText text;
    if (!text.open(path, stretch_opt > 0 ? 4 : (size_t)1 << 20)) return false;
    const char *d = text.d;
    auto line_at = [&](size_t at, std::string &line) {                  // the line without its end; returns the start of the next
        const char *e = (const char *)memchr(d + at, '\n', text.n - at);
        line.assign(d + at, (size_t)(e - (d + at)));
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return (size_t)(e - d) + 1;
    };
    std::vector<std::string> contigs, rec;
    std::string line;
    size_t begin = 0;
    bool have_record = false;
    while (begin < text.n && !have_record) {
        const size_t next = line_at(begin, line);
        if (line.compare(0, 13, "##contig=<ID=") == 0) {
            const size_t e = line.find_first_of(",>", 13);
            contigs.push_back(line.substr(13, e == std::string::npos ? std::string::npos : e - 13));
        }
        if (line.empty() || '#' == line[0]) begin = next;
        else have_record = true;
    }
    if (!have_record || contigs != first_names) return false;
    split_tabs(line, rec);
    if (rec.size() < 10) return false;
    uint32_t A = 0;
    for (size_t g = 9; g < rec.size(); ++g) {
        for (size_t pos = 0; pos < rec[g].size() && ':' != rec[g][pos]; ++pos)
            if ('|' == rec[g][pos] || '/' == rec[g][pos]) ++A;
        ++A;
    }
    if (A > Variant::kMaxAlleles) return false;
    std::vector<size_t> at = text.cut(stretch);
    at.erase(at.begin(), std::upper_bound(at.begin(), at.end(), begin));
    at.insert(at.begin(), begin);
    std::vector<std::unique_ptr<VcfRecords>> pieces(at.size() - 1);
    const unsigned hw = std::thread::hardware_concurrency(), n_threads = std::max(1u, std::min(hw ? hw : 4u, 32u));
    std::atomic<bool> plain{true};
    on_threads(pieces.size(), n_threads, [&](size_t i) {
        pieces[i].reset(new VcfRecords(first_names, contigs, codes, A, nullptr));
        VcfRecords &r = *pieces[i];
        r.reserve_hint = (at[i + 1] - at[i]) / 24;
        std::string text_line;
        std::vector<std::string> fields;
        for (size_t p = at[i]; p < at[i + 1] && plain;) {
            p = line_at(p, text_line);
            if (text_line.empty()) continue;
            split_tabs(text_line, fields);
            if (fields.size() < 10) return void(plain = false);
            r.record(fields);
            if (r.n_errors) return void(plain = false);
        }
    });
    if (!plain) return false;
    const VcfRecords *last = nullptr;
    std::vector<size_t> total(first_names.size(), 0);
    for (const auto &pc : pieces) {
        if (!pc->records) continue;
        if (last && (pc->first_rid < last->reached_seq || (pc->first_rid == last->reached_seq && (pc->first_begin < last->placed_begin || pc->first_begin < last->placed_end)))) return false;
        last = pc.get();
        for (const auto &run : pc->runs) total[run.first] += run.second.size();
    }
    out.num_alleles = A;
    out.by_seq.assign(first_names.size(), {});
    for (size_t i = 0; i < total.size(); ++i) out.by_seq[i].reserve(total[i]);
    for (auto &pc : pieces)
        for (auto &run : pc->runs) {
            std::vector<Variant> &to = out.by_seq[run.first];
            to.insert(to.end(), std::make_move_iterator(run.second.begin()), std::make_move_iterator(run.second.end()));
        }
    return true;
}
}  // namespace

Variants read_variants(const std::string &path, const std::vector<std::string> &first_names, const std::vector<std::vector<uint8_t>> &codes) {
    if (!options().serial_parse) {
        Variants mapped;
        if (read_variants_mapped(path, first_names, codes, mapped)) {
            __atomic_add_fetch(&options().mapped_parses, 1, __ATOMIC_RELAXED);      // files may be loaded from several threads
            return mapped;
        }
    }
    GzLines f(path);
    std::string line;
    std::vector<std::string> contigs;
    bool have_record = false;
    while (f.getline(line)) {                                         // readHeader: ## lines, then the #CHROM line
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.compare(0, 2, "##") == 0) {
            if (line.compare(0, 13, "##contig=<ID=") == 0) {
                const size_t e = line.find_first_of(",>", 13);
                contigs.push_back(line.substr(13, e == std::string::npos ? std::string::npos : e - 13));
            }
            continue;
        }
        if (!line.empty() && line[0] == '#') continue;
        if (line.empty()) continue;
        have_record = true;
        break;
    }
    // CheckVcf (Reference.cpp:80-97)
    std::string errors;
    uint32_t n_errors = 0;
    auto error = [&](const std::string &msg) {
        if (n_errors++ < 20) errors += msg + " ";
    };
    if (contigs.size() != first_names.size())
        error("Number of contigs does not match between reference(" + std::to_string(first_names.size()) + ") and variant(" + std::to_string(contigs.size()) + ") file.");
    for (size_t c = 0; c < std::min(contigs.size(), first_names.size()); ++c)
        if (contigs[c] != first_names[c]) error("Contigs at position " + std::to_string(c) + " do not match between reference(" + first_names[c] + ") and variant(" + contigs[c] + ") file.");
    if (n_errors) throw Error(errors);
    if (!have_record) throw Error("Vcf file '" + path + "' has no records.");

    Variants out;
    out.by_seq.resize(first_names.size());
    std::vector<std::string> rec;
    split_tabs(line, rec);
    if (rec.size() < 10) throw Error("Could not read first vcf record: fewer than 10 columns");
    out.num_alleles = 0;                                              // ReadFirstVariants (:1046-1058)
    for (size_t g = 9; g < rec.size(); ++g) {
        for (size_t pos = 0; pos < rec[g].size() && ':' != rec[g][pos]; ++pos)
            if ('|' == rec[g][pos] || '/' == rec[g][pos]) ++out.num_alleles;
        ++out.num_alleles;
    }
    if (out.num_alleles > Variant::kMaxAlleles) throw Error("Currently only 128 alleles are supported, but file has " + std::to_string(out.num_alleles) + ".");

    VcfRecords r(first_names, contigs, codes, out.num_alleles, &out.by_seq);
    for (;;) {
        r.record(rec);
        if (r.n_errors >= 20) break;                                  // kMaxErrorsShownPerFile
        bool got = false;
        while (f.getline(line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (!line.empty()) {
                got = true;
                break;
            }
        }
        if (!got) break;
        split_tabs(line, rec);
        if (rec.size() < 10) {
            r.error("Could not read vcf record: fewer than 10 columns");
            break;
        }
    }
    if (r.n_errors) throw Error(r.errors);
    return out;
}

}  // namespace rsq
