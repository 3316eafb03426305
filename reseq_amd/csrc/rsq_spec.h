// rsq_spec.h -- read kernels compiled for ONE profile at run time (host side; included by rsq_sim.hip only).
//
// What a loaded profile fixes -- the LDS plan, the three table families' common geometry (FamilyGeo), tiles, phred offset, longest deletion -- are run-time values of the
// by-value kernel argument in the library's own instantiations of k_fill_reads / k_fill_records: they occupy scalar registers (the kernels sit at the cap and spill
// them into vector lanes), and every row address is integer arithmetic on them.  Here the same kernel bodies (rsq_kernels.h, embedded in the library as text) are
// compiled by hiprtc with those values as LITERALS (RSQ_SPEC: the macros RSQ_PLAN / RSQ_SIM of rsq_types.h read namespace rsq::spec instead of the argument).  One
// program per kernel variant (reads / records, with variants, binned by tile), compiled when the variant is first needed, kept per simulator; the code object is
// also kept on disk ($XDG_CACHE_HOME or ~/.cache, under reseq_amd/) under a hash of sources, literals, variant, architecture and compiler version.
// Without libhiprtc, or when a compilation fails, the library's own instantiation runs (option `specialize` = 0 asks for that); rsq_last_warning() says which.
// Results are the same bytes either way (tests run both).  The reference's counterpart of "shapes known per profile": the loop bounds of LogArrayResult::Draw,
// ProbabilityEstimates.h:481-508, are members of the loaded tables.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "rsq_host.h"
#include "rsq_kernels.h"

namespace rsq {

// the kernel sources as the build embedded them (build/rsq_embedded.inc: one raw string per header)
#include "build/rsq_embedded.inc"

struct Hiprtc {                                   // the few entry points, bound at run time: the library must load where libhiprtc is absent
    using Program = void *;
    int (*create)(Program *, const char *, const char *, int, const char *const *, const char *const *) = nullptr;
    int (*compile)(Program, int, const char *const *) = nullptr;
    int (*log_size)(Program, size_t *) = nullptr;
    int (*log)(Program, char *) = nullptr;
    int (*code_size)(Program, size_t *) = nullptr;
    int (*code)(Program, char *) = nullptr;
    int (*destroy)(Program *) = nullptr;
    int (*version)(int *, int *) = nullptr;
    bool ok = false;
    std::string path;                             // the file the entry points came from (dladdr), for rsq_sim_specialize's note
    std::string comgr;                            // ... and the code generator behind it: the libamd_comgr that libhiprtc's own lookups find
    // The system ROCm's libhiprtc by its full path first: by name a Python process that has imported PyTorch finds the wheel's copy, a process under rocprofv3 or
    // without PyTorch the system's (option hiprtc_by_name 1 restores that search).  That pins the front end only.  The code generator is libamd_comgr, which
    // libhiprtc needs by its soname -- and a process that has imported PyTorch has the wheel's copy loaded under that soname already, so the loader hands it to the
    // system's libhiprtc as well (measured: 28 spilled registers in the variant kernel from the wheel's ROCm 7.0 code generator, 43 from the system's 7.2; 8 % of the
    // variant path, nothing of the plain one).  Which copy compiled a kernel is therefore recorded (note, bench line) and part of the cache key, not forced: the
    // command line and every process without PyTorch get the system's.
    static const Hiprtc &get() {
        static Hiprtc h = [] {
            Hiprtc r;
            void *lib = nullptr;
            const bool by_name = options().hiprtc_by_name != 0;
            for (const char *name : {"/opt/rocm/lib/libhiprtc.so", "libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6"}) {
                if (by_name && name[0] == '/') continue;
                if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
            }
            if (!lib && by_name) lib = dlopen("/opt/rocm/lib/libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
            if (!lib) return r;
            auto bind = [&](auto &f, const char *name) { f = reinterpret_cast<std::remove_reference_t<decltype(f)>>(dlsym(lib, name)); return f != nullptr; };
            int bound = 0;
            bound += bind(r.create, "hiprtcCreateProgram") + bind(r.compile, "hiprtcCompileProgram") + bind(r.log_size, "hiprtcGetProgramLogSize") + bind(r.log, "hiprtcGetProgramLog");
            bound += bind(r.code_size, "hiprtcGetCodeSize") + bind(r.code, "hiprtcGetCode") + bind(r.destroy, "hiprtcDestroyProgram") + bind(r.version, "hiprtcVersion");
            r.ok = bound == 8;
            Dl_info info;
            if (r.ok && dladdr(reinterpret_cast<void *>(r.create), &info) && info.dli_fname) {
                char real[4096];
                r.path = realpath(info.dli_fname, real) ? real : info.dli_fname;
            }
            // libhiprtc loads the code generator itself, by its soname, when it first compiles: the copy the process already holds under that name (PyTorch's), else the
            // one the loader finds -- asked for here the same way, so that its file is known before the first compilation (it is part of the cache key)
            if (r.ok) {
                void *comgr = nullptr;
                for (const char *name : {"libamd_comgr.so.3", "libamd_comgr.so.2", "libamd_comgr.so"}) {
                    if ((comgr = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
                }
                if (!comgr)
                    for (const char *name : {"libamd_comgr.so.3", "libamd_comgr.so.2", "libamd_comgr.so"}) {
                        if ((comgr = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
                    }
                void *sym = comgr ? dlsym(comgr, "amd_comgr_get_version") : nullptr;
                if (sym && dladdr(sym, &info) && info.dli_fname) {
                    char real[4096];
                    r.comgr = realpath(info.dli_fname, real) ? real : info.dli_fname;
                }
            }
            return r;
        }();
        return h;
    }
};

inline uint64_t fnv1a(const void *p, size_t n, uint64_t h = 0xcbf29ce484222325ull) {
    const unsigned char *c = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; ++i) h = (h ^ c[i]) * 0x100000001b3ull;
    return h;
}
inline uint64_t fnv1a(const std::string &s, uint64_t h) { return fnv1a(s.data(), s.size(), h); }

// the literals of one profile: namespace rsq::spec as RSQ_PLAN / RSQ_SIM expect it.  The plan is written as the words of the structure (every member of LdsPlan and
// FamilyGeo is a 32-bit word; brace elision lets an aggregate take them as a flat list), so a member added there needs nothing here.
inline std::string spec_literals(const DevSim &d) {
    static_assert(sizeof(LdsPlan) % 4 == 0 && std::is_trivially_copyable<LdsPlan>::value, "LdsPlan is a list of 32-bit words");
    std::string t = "namespace rsq { namespace spec {\nconstexpr LdsPlan lds = {";
    uint32_t words[sizeof(LdsPlan) / 4];
    memcpy(words, &d.lds, sizeof(LdsPlan));
    for (size_t i = 0; i < sizeof(LdsPlan) / 4; ++i) t += (i ? ", " : "") + std::to_string(words[i]) + "u";
    t += "};\n";
    t += "constexpr uint32_t n_tiles = " + std::to_string(d.n_tiles) + "u, phred_offset = " + std::to_string((unsigned)d.phred_offset) + "u, max_len_deletion = " +
         std::to_string((unsigned)d.max_len_deletion) + "u, force_exact = " + std::to_string(d.force_exact) + "u;\n";
    t += "} }\n";
    return t;
}

enum class SpecKind : int { kReads = 0, kRecords = 1 };
struct SpecVariant {
    SpecKind kind;
    uint32_t mask;
    bool var, binned;
    int key() const { return ((int)kind << 2) | ((int)var << 1) | (int)binned; }
};
inline const char *spec_kernel_name(SpecKind k) { return k == SpecKind::kReads ? "rsq_spec_fill_reads" : "rsq_spec_fill_records"; }

inline std::string spec_program(const std::string &literals, const SpecVariant &v) {
    std::string t = "#define RSQ_SPEC 1\n#include \"rsq_types.h\"\n" + literals + "#include \"rsq_kernels.h\"\nusing namespace rsq;\n";
    const std::string m = std::to_string(v.mask) + "u, ";
    if (v.kind == SpecKind::kReads)
        t += std::string("extern \"C\" __global__ void __launch_bounds__(fill_block(") + (v.var ? "true" : "false") + ")) rsq_spec_fill_reads(DevSim S, NameTable names, const Fragment *frags, uint64_t n_pairs, uint64_t adapter_only_first, "
             "RawLayout raw, uint32_t *sizes, uint32_t *chunk_counters, const FragmentVar *fvars, FillBins bins) {\n    fill_reads_body<" + m + (v.var ? "true, " : "false, ") +
             (v.binned ? "true" : "false") + ">(S, names, frags, n_pairs, adapter_only_first, raw, sizes, chunk_counters, fvars, bins);\n}\n";
    else
        t += "extern \"C\" __global__ void __launch_bounds__(kFillBlockWalk) rsq_spec_fill_records(DevSim S, RecordJob job, RawLayout raw, uint32_t *chunk_counters, FillBins bins) {\n"
             "    fill_records_body<" + m + (v.binned ? "true, " : "false, ") + (v.var ? "true" : "false") + ">(S, job, raw, chunk_counters, bins);\n}\n";      // (var: the records are packed)
    return t;
}

// where compiled code objects are kept between processes: rsq_set_kernel_cache_dir ("" = nowhere), else $XDG_CACHE_HOME/reseq_amd or ~/.cache/reseq_amd
inline std::string &spec_cache_dir_override() {
    static std::string dir;
    return dir;
}
inline bool &spec_cache_dir_set() {
    static bool set = false;
    return set;
}
inline std::string spec_cache_dir() {
    if (spec_cache_dir_set()) {
        if (!spec_cache_dir_override().empty()) mkdir(spec_cache_dir_override().c_str(), 0755);
        return spec_cache_dir_override();
    }
    const char *x = getenv("XDG_CACHE_HOME"), *h = getenv("HOME");
    std::string base = x && *x ? x : (h && *h ? std::string(h) + "/.cache" : "");
    if (base.empty()) return "";
    mkdir(base.c_str(), 0755);
    base += "/reseq_amd";
    mkdir(base.c_str(), 0755);
    return base;
}

// Experiments on the read kernel without a rebuild of the library: RSQ_SPEC_OPTIONS="-DNAME=1 -DOTHER=2" adds compiler options (part of the cache key).
inline std::vector<std::string> spec_extra_options() {
    std::vector<std::string> out;
    const char *e = getenv("RSQ_SPEC_OPTIONS");
    if (!e) return out;
    std::string cur;
    for (const char *p = e;; ++p) {
        if (*p == ' ' || *p == '\0') {
            if (!cur.empty()) out.push_back(cur);
            cur.clear();
            if (!*p) break;
        } else cur += *p;
    }
    return out;
}

// The code object of one variant for one profile: from the disk cache, else compiled (and put there).  false: not available, `note` says why.
struct SpecCode {
    std::vector<char> code;
    std::string cache_path;        // "" when nothing is kept on disk
    double seconds = 0.0;          // compilation (0 when the code object came from the disk cache)
    bool from_cache = false;
    int rtc_major = 0, rtc_minor = 0;
};
inline bool spec_compile(const DevSim &dev, const SpecVariant &v, const std::string &arch, SpecCode &out, std::string &note) {
    const Hiprtc &rtc = Hiprtc::get();
    if (!rtc.ok) {
        note = "libhiprtc not found: the read kernels run in the library's own instantiation (not compiled for this profile)";
        return false;
    }
    const std::string literals = spec_literals(dev), program = spec_program(literals, v);
    rtc.version(&out.rtc_major, &out.rtc_minor);
    const char *headers[] = {kSrc_rsq_types_h, kSrc_rsq_core_h, kSrc_rsq_variants_h, kSrc_rsq_kernels_h};
    const char *names[] = {"rsq_types.h", "rsq_core.h", "rsq_variants.h", "rsq_kernels.h"};
    uint64_t h = fnv1a(program + " " + std::to_string(RSQ_FILL_BLOCK) + " " + std::to_string(RSQ_FILL_BLOCK_WALK) + " " + std::to_string(RSQ_SCREEN_BATCH) + " " + std::to_string(RSQ_CHUNK_LARGE)
#if defined(RSQ_TRACE_FILL)
                                + " trace"
#endif
                            , fnv1a(arch, 0xcbf29ce484222325ull));
    for (const std::string &o : spec_extra_options()) h = fnv1a(o.data(), o.size(), h);
    for (const char *src : headers) h = fnv1a(src, strlen(src), h);
    h = fnv1a(&out.rtc_major, sizeof(int), fnv1a(&out.rtc_minor, sizeof(int), h));
    h = fnv1a(rtc.path.data(), rtc.path.size(), h);                   // two copies of libhiprtc that report one version are still two compilers
    h = fnv1a(rtc.comgr.data(), rtc.comgr.size(), h);                 // ... and so are two code generators behind one libhiprtc
    char name[64];
    snprintf(name, sizeof name, "/%s_%016llx.hsaco", spec_kernel_name(v.kind), (unsigned long long)h);
    const std::string dir = spec_cache_dir();
    out.cache_path = dir.empty() ? "" : dir + name;
    out.code.clear();
    // A process started by rocprofv3 gets another code object from the same sources (measured, round 5: 768 bytes longer, 6.7 % more vector instructions per launch of the
    // read kernel, 3 % slower).  The cause seems to be WHICH compiler is loaded: a Python process that has imported PyTorch finds the libhiprtc / libamd_comgr of the
    // wheel (ROCm 7.0: 9 spilled vector registers in the kernel compiled for P0), rocprofv3 puts the system's /opt/rocm-7.2.0/lib first (12, like the image's hipcc).
    // Both report version 9.0, so the key cannot tell them apart.  A profiled process may USE what an un-profiled one left in the cache -- then the profiler sees the
    // kernel the product runs -- but what it compiles itself goes under a name of its own: it must not become every later run's kernel (profiles/collect.sh therefore
    // begins with an un-profiled run).
    const bool under_profiler = getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_LIBRARY_CTOR");
    if (under_profiler && !out.cache_path.empty() && access(out.cache_path.c_str(), R_OK) != 0) out.cache_path += ".under_profiler";
    if (!out.cache_path.empty()) {
        if (FILE *f = fopen(out.cache_path.c_str(), "rb")) {
            fseek(f, 0, SEEK_END);
            const long n = ftell(f);
            fseek(f, 0, SEEK_SET);
            out.code.resize(n > 0 ? (size_t)n : 0);
            if (n <= 0 || fread(out.code.data(), 1, out.code.size(), f) != out.code.size()) out.code.clear();
            fclose(f);
            out.from_cache = !out.code.empty();
        }
    }
    if (!out.code.empty()) return true;
    const auto t0 = std::chrono::steady_clock::now();
    Hiprtc::Program prog = nullptr;
    if (rtc.create(&prog, program.c_str(), "rsq_spec.hip", 4, headers, names) != 0) {
        note = "hiprtcCreateProgram failed: the read kernels run in the library's own instantiation";
        return false;
    }
    const std::string arch_opt = "--offload-arch=" + arch;
    // the library's own flags (Makefile): the double-precision route must round like the reference's separate multiply / add; and the build's geometry macros as
    // this library was compiled with them (workgroup size, batch of the screen, chunk of the double-precision draws): the launch and the LDS plan are the library's
    const std::string block = "-DRSQ_FILL_BLOCK=" + std::to_string(RSQ_FILL_BLOCK), walk = "-DRSQ_FILL_BLOCK_WALK=" + std::to_string(RSQ_FILL_BLOCK_WALK),
                      batch = "-DRSQ_SCREEN_BATCH=" + std::to_string(RSQ_SCREEN_BATCH), chunk = "-DRSQ_CHUNK_LARGE=" + std::to_string(RSQ_CHUNK_LARGE);
    std::vector<const char *> opts = {arch_opt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-missing-braces", block.c_str(), walk.c_str(), batch.c_str(), chunk.c_str()};
#if defined(RSQ_TRACE_FILL)
    opts.push_back("-DRSQ_TRACE_FILL=1");
#endif
    const std::vector<std::string> extra = spec_extra_options();
    for (const std::string &o : extra) opts.push_back(o.c_str());
    if (rtc.compile(prog, (int)opts.size(), opts.data()) != 0) {
        size_t n = 0;
        rtc.log_size(prog, &n);
        std::string log(n, '\0');
        if (n) rtc.log(prog, &log[0]);
        rtc.destroy(&prog);
        note = "compiling the read kernel for this profile failed (the library's own instantiation runs instead): " + log.substr(0, 3000);
        return false;
    }
    size_t n = 0;
    rtc.code_size(prog, &n);
    out.code.resize(n);
    rtc.code(prog, out.code.data());
    rtc.destroy(&prog);
    out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!out.cache_path.empty()) {                                 // written under another name first: a reader never sees half a file
        const std::string tmp = out.cache_path + "." + std::to_string((long)getpid());
        if (FILE *f = fopen(tmp.c_str(), "wb")) {
            const bool whole = fwrite(out.code.data(), 1, out.code.size(), f) == out.code.size();
            fclose(f);
            if (!whole || rename(tmp.c_str(), out.cache_path.c_str()) != 0) unlink(tmp.c_str());
        }
    }
    return true;
}

// The kernels of one simulator.  get() returns the function of a variant, compiling (or reading from the disk cache) on first use; nullptr: not available -- the
// caller launches the library's own instantiation.
class SpecKernels {
   public:
    struct Entry {
        hipModule_t module = nullptr;
        hipFunction_t fn = nullptr;
        bool tried = false;
        std::string note;              // what happened when the variant was first asked for
    };
    ~SpecKernels() {
        for (auto &e : entries_)
            if (e.second.module) (void)hipModuleUnload(e.second.module);
    }
    // `arch`: hipDeviceProp_t::gcnArchName of the simulator's device
    hipFunction_t get(const DevSim &dev, const SpecVariant &v, const std::string &arch, std::string &note) {
        Entry &e = entries_[v.key()];
        if (e.tried) {
            note = e.note;
            return e.fn;
        }
        e.tried = true;
        get_once(e, dev, v, arch);
        note = e.note;
        return e.fn;
    }

   private:
    void get_once(Entry &e, const DevSim &dev, const SpecVariant &v, const std::string &arch) {
        std::string &note = e.note;
        SpecCode c;
        if (!spec_compile(dev, v, arch, c, note)) return;
        if (hipModuleLoadData(&e.module, c.code.data()) != hipSuccess || hipModuleGetFunction(&e.fn, e.module, spec_kernel_name(v.kind)) != hipSuccess) {
            (void)hipGetLastError();
            if (e.module) (void)hipModuleUnload(e.module);
            e.module = nullptr;
            e.fn = nullptr;
            if (c.from_cache && !c.cache_path.empty()) unlink(c.cache_path.c_str());    // a stale or damaged file: the next simulator compiles afresh
            note = "loading the read kernel compiled for this profile failed: the library's own instantiation runs instead";
            return;
        }
        char buf[1024];
        snprintf(buf, sizeof buf, "%s compiled for this profile (%s, hiprtc %d.%d of %s, code generator %s, %s)", spec_kernel_name(v.kind), arch.c_str(), c.rtc_major, c.rtc_minor,
                 Hiprtc::get().path.empty() ? "an unnamed file" : Hiprtc::get().path.c_str(), Hiprtc::get().comgr.empty() ? "unknown" : Hiprtc::get().comgr.c_str(),
                 c.from_cache ? "code object from the kernel cache" : (std::to_string((int)(c.seconds * 1000)) + " ms").c_str());
        note = buf;
    }
    std::map<int, Entry> entries_;
};

}  // namespace rsq
