// rsq_archive.h -- reader for the Boost.Serialization *text* archives ReSeq keeps its profiles in
// (`.reseq`: DataStats::Save, reseq/DataStats.cpp:1302-1320; `.reseq.ipf`: ProbabilityEstimates::Save,
// reseq/ProbabilityEstimates.cpp:1047-1065).  No Boost is involved: the archive is a stream of blank-separated tokens and
// which tokens appear is decided by the static C++ type of every serialized member, so the reader is driven by a description
// of those types (built with the helpers below, see rsq_profile_archive.cpp).
//
// Token rules of boost::archive::text_oarchive that matter here (the rules are recalled from Boost 1.6x/1.7x, there is no
// Boost and no sample profile in this image: compatibility with a Boost-written file is UNVERIFIED):
//   header            "22 serialization::archive <library version>"
//   arithmetic types  printed as numbers (char-sized integers too), bool as 0/1, double with 17 significant digits
//   std::string       "<length> <bytes>"
//   class types       the FIRST time an object of a given C++ type is saved: "<tracking> <version>" (both 0 here: nothing is
//                     saved through a pointer, no BOOST_CLASS_VERSION is used); afterwards nothing.  Class types are: every
//                     user class, std::pair, std::array, and std::vector of anything that is not a built-in arithmetic type
//                     (boost/serialization/collection_traits.hpp makes vectors of arithmetic types "object_serializable")
//   std::vector<T>    "<count> <item_version>" + items (item_version only when the library version is above 3)
//   std::vector<bool> "<count>" + items
//   std::array<T,N>   its class info, then the C array inside: "<N>" + items
//   std::pair         its class info, first, second
// The rules marked doubtful in SURVEY.md appendix A are switches (Grammar below): a file is read under the recalled set first and, if that fails, under the others.
#pragma once
#include <stdint.h>
#include <string.h>

#include <charconv>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "rsq_host.h"

namespace rsq {
namespace archive {

struct Type;
using TypeP = const Type *;

// The token rules above that cannot be checked against a Boost-written file in this image, as switches.  The default-constructed Grammar is the recalled rule set;
// Grammar::alternatives() lists every combination, the recalled one first.  A file that does not parse under one grammar is tried under the next: the schema's fixed
// sizes (std::array<T, N> counts, the 3 x 4^10 surroundings, the margin counts of LogIPF<N>), "tracking must be 0" and "no token left after the last member" are
// the self-check that tells the right one -- a surplus or missing token shifts everything behind it.
struct Grammar {
    enum ItemVersion : uint8_t { kNever = 0, kAllButBool = 1, kClassItemsOnly = 2, kEveryVector = 3 };
    uint8_t item_version = kAllButBool;           // which std::vector counts are followed by "<item_version>" (library version > 3)
    bool array_class_info = true;                 // std::array<T, N> is a class type: "<tracking> <version>" the first time
    bool array_count = true;                      // the C array inside a std::array is preceded by its count
    bool pair_class_info = true;                  // std::pair is a class type
    bool arithmetic_vector_class_info = false;    // std::vector of an arithmetic type is a class type too
    bool is_default() const { return item_version == kAllButBool && array_class_info && array_count && pair_class_info && !arithmetic_vector_class_info; }
    std::string name() const {
        static const char *iv[] = {"no item_version", "item_version behind every vector count but vector<bool>'s", "item_version behind the counts of vectors of class-type items only",
                                   "item_version behind every vector count"};
        return std::string(iv[item_version]) + "; std::array " + (array_class_info ? "with" : "without") + " class information, " + (array_count ? "with" : "without") +
               " a count; std::pair " + (pair_class_info ? "with" : "without") + " class information; vectors of arithmetic types " +
               (arithmetic_vector_class_info ? "with" : "without") + " class information";
    }
    static std::vector<Grammar> alternatives() {
        std::vector<Grammar> out(1);                                    // the recalled rules first
        for (uint8_t iv : {kAllButBool, kClassItemsOnly, kEveryVector, kNever})
            for (bool aci : {true, false})
                for (bool ac : {true, false})
                    for (bool pci : {true, false})
                        for (bool avci : {false, true}) {
                            Grammar g;
                            g.item_version = iv, g.array_class_info = aci, g.array_count = ac, g.pair_class_info = pci, g.arithmetic_vector_class_info = avci;
                            if (!g.is_default()) out.push_back(g);
                        }
        return out;
    }
};

struct Member {
    std::string name;
    TypeP type;
    bool keep;        // parsed into the Node tree (true) or only walked over (false)
};

struct Type {
    enum Kind { UINT, INT, F64, BOOL, STR, VEC, ARR, PAIR, CLS };
    Kind kind = UINT;
    std::string name;             // canonical C++ spelling: the identity that decides "first time this type is saved"
    TypeP elem = nullptr;         // VEC, ARR
    size_t n = 0;                 // ARR
    TypeP first = nullptr, second = nullptr;   // PAIR
    std::vector<Member> members;  // CLS
    bool class_info = false;
    int id = 0;
    bool numeric() const { return kind == UINT || kind == INT || kind == F64 || kind == BOOL; }
};

// Interns types by name, so that `Vect<uint64_t>` reached through two different members is ONE type (one class-info record).
class Schema {
   public:
    TypeP prim(Type::Kind k, const std::string &name) {
        Type t;
        t.kind = k;
        t.name = name;
        return intern(std::move(t));
    }
    TypeP u8() { return prim(Type::UINT, "unsigned char"); }
    TypeP u16() { return prim(Type::UINT, "unsigned short"); }
    TypeP u32() { return prim(Type::UINT, "unsigned int"); }
    TypeP u64() { return prim(Type::UINT, "unsigned long"); }
    TypeP i8() { return prim(Type::INT, "signed char"); }
    TypeP f64() { return prim(Type::F64, "double"); }
    TypeP boolean() { return prim(Type::BOOL, "bool"); }
    TypeP str() { return prim(Type::STR, "std::string"); }
    TypeP vec(TypeP e) {
        Type t;
        t.kind = Type::VEC;
        t.name = "std::vector<" + e->name + ">";
        t.elem = e;
        t.class_info = !e->numeric();
        return intern(std::move(t));
    }
    TypeP arr(TypeP e, size_t n) {
        Type t;
        t.kind = Type::ARR;
        t.name = "std::array<" + e->name + "," + std::to_string(n) + ">";
        t.elem = e;
        t.n = n;
        t.class_info = true;
        return intern(std::move(t));
    }
    TypeP pair(TypeP a, TypeP b) {
        Type t;
        t.kind = Type::PAIR;
        t.name = "std::pair<" + a->name + "," + b->name + ">";
        t.first = a;
        t.second = b;
        t.class_info = true;
        return intern(std::move(t));
    }
    TypeP cls(const std::string &name, std::vector<Member> members) {
        Type t;
        t.kind = Type::CLS;
        t.name = name;
        t.members = std::move(members);
        t.class_info = true;
        return intern(std::move(t));
    }
    size_t size() const { return types_.size(); }

   private:
    TypeP intern(Type &&t) {
        auto it = by_name_.find(t.name);
        if (it != by_name_.end()) return it->second;
        t.id = (int)types_.size();
        types_.push_back(std::make_unique<Type>(std::move(t)));
        by_name_[types_.back()->name] = types_.back().get();
        return types_.back().get();
    }
    std::vector<std::unique_ptr<Type>> types_;
    std::unordered_map<std::string, TypeP> by_name_;
};

// What was read.  Numbers of a vector / array of arithmetic type lie flat in `u` (integers, two's complement for signed) or
// `f`; everything else has one kid per item / member (members that were not kept stay empty).
struct Node {
    TypeP type = nullptr;
    std::vector<Node> kids;
    std::vector<uint64_t> u;
    std::vector<double> f;
    std::string s;
    const Node &operator[](const char *member) const {
        if (!type || type->kind != Type::CLS) throw Error("archive: member access on a value that is not a class");
        for (size_t i = 0; i < type->members.size(); ++i)
            if (type->members[i].name == member) {
                if (!type->members[i].keep) throw Error(std::string("archive: member ") + member + " was not kept");
                return kids[i];
            }
        throw Error(std::string("archive: no member ") + member + " in " + type->name);
    }
    const Node &operator[](size_t i) const { return kids.at(i); }
    const Node &operator[](int i) const { return kids.at((size_t)i); }
    const Node &operator[](uint32_t i) const { return kids.at(i); }
    size_t size() const { return type && type->elem && type->elem->numeric() ? (type->elem->kind == Type::F64 ? f.size() : u.size()) : kids.size(); }
    uint64_t uint() const { return u.at(0); }
    double real() const { return f.at(0); }
    const Node &first() const { return kids.at(0); }
    const Node &second() const { return kids.at(1); }
};

// Where the class information of a type was found: the first object of the type in the stream.  The reader cannot be checked against a file written by
// Boost in this image, so when a real profile does not parse, this list (`reseq queryProfile --dumpArchiveLayout`) and the member path in the error
// message are what tells which of the token rules above is wrong.
struct ClassInfoSite {
    std::string type, path;
    size_t byte = 0;
    uint64_t tracking = 0, version = 0;
};

class Reader {
   public:
    Reader(const char *begin, const char *end, size_t n_types, const std::string &what, const Grammar &grammar = Grammar())
        : p_(begin), end_(end), seen_(n_types, 0), what_(what), g_(grammar) {
        const std::string sig = str();
        if (sig != "serialization::archive") fail("not a Boost text archive");
        library_version_ = (uint32_t)unsigned_int();
    }
    const std::vector<ClassInfoSite> &class_info_sites() const { return sites_; }
    static bool looks_like_archive(const char *begin, size_t n) {
        static const char head[] = "22 serialization::archive";
        return n >= sizeof head - 1 && !memcmp(begin, head, sizeof head - 1);
    }
    uint32_t library_version() const { return library_version_; }
    void read(TypeP t, Node *out) {
        current_ = t;
        if (has_class_info(t) && !seen_[t->id]) {
            seen_[t->id] = (uint32_t)sites_.size() + 1u;
            ClassInfoSite site;
            site.type = t->name;
            site.path = path_string();
            skip_blank();
            site.byte = (size_t)(p_ - begin());
            sites_.push_back(site);
            const uint64_t tracking = unsigned_int();
            const uint64_t version = unsigned_int();   // class version (0 everywhere in ReSeq)
            sites_.back().tracking = tracking;
            sites_.back().version = version;
            if (tracking) fail("class " + t->name + " is saved with object tracking, which ReSeq's profiles do not use");
        }
        if (out) out->type = t;
        switch (t->kind) {
            case Type::UINT:
            case Type::BOOL: {
                const uint64_t v = unsigned_int();
                if (out) out->u.assign(1, v);
                break;
            }
            case Type::INT: {
                const int64_t v = signed_int();
                if (out) out->u.assign(1, (uint64_t)v);
                break;
            }
            case Type::F64: {
                const double v = real();
                if (out) out->f.assign(1, v);
                break;
            }
            case Type::STR: {
                std::string v = str();
                if (out) out->s = std::move(v);
                break;
            }
            case Type::VEC: {
                const uint64_t count = unsigned_int();
                const bool item_version = g_.item_version == Grammar::kEveryVector || (g_.item_version == Grammar::kAllButBool && t->elem->kind != Type::BOOL) ||
                                          (g_.item_version == Grammar::kClassItemsOnly && !t->elem->numeric());
                if (item_version && library_version_ > 3) unsigned_int();
                items(t->elem, count, out);
                break;
            }
            case Type::ARR: {
                const uint64_t count = g_.array_count ? unsigned_int() : t->n;
                if (count != t->n) fail("array " + t->name + " holds " + std::to_string(count) + " items");
                items(t->elem, count, out);
                break;
            }
            case Type::PAIR: {
                static const std::string kFirst = "first", kSecond = "second";
                if (out) out->kids.resize(2);
                path_.push_back(Seg{t, &kFirst, 0});
                read(t->first, out ? &out->kids[0] : nullptr);
                path_.back().name = &kSecond;
                read(t->second, out ? &out->kids[1] : nullptr);
                path_.pop_back();
                break;
            }
            case Type::CLS:
                if (out) out->kids.resize(t->members.size());
                path_.push_back(Seg{t, nullptr, 0});
                for (size_t i = 0; i < t->members.size(); ++i) {
                    path_.back().name = &t->members[i].name;
                    read(t->members[i].type, out && t->members[i].keep ? &out->kids[i] : nullptr);
                }
                path_.pop_back();
                break;
        }
    }
    void expect_end() {
        skip_blank();
        if (p_ != end_) fail("tokens left after the last member");
    }

   private:
    bool has_class_info(TypeP t) const {
        switch (t->kind) {
            case Type::CLS: return true;
            case Type::ARR: return g_.array_class_info;
            case Type::PAIR: return g_.pair_class_info;
            case Type::VEC: return !t->elem->numeric() || g_.arithmetic_vector_class_info;
            default: return false;
        }
    }
    void items(TypeP e, uint64_t count, Node *out) {
        if (count > (uint64_t)(end_ - p_)) fail("item count " + std::to_string(count) + " larger than the file");
        const TypeP holder = current_;
        path_.push_back(Seg{holder, nullptr, 0});
        uint64_t &i = path_.back().index;
        if (e->numeric()) {
            current_ = e;
            if (e->kind == Type::F64) {
                if (out) out->f.resize(count);
                for (i = 0; i < count; ++i) {
                    const double v = real();
                    if (out) out->f[i] = v;
                }
            } else {
                if (out) out->u.resize(count);
                for (i = 0; i < count; ++i) {
                    const uint64_t v = e->kind == Type::INT ? (uint64_t)signed_int() : unsigned_int();
                    if (out) out->u[i] = v;
                }
            }
        } else {
            if (out) out->kids.resize(count);
            for (uint64_t k = 0; k < count; ++k) {                   // the path's back may move when deeper levels push
                path_.back().index = k;
                read(e, out ? &out->kids[k] : nullptr);
            }
        }
        path_.pop_back();
    }
    // "DataStats.errors_.indel_by_indel_pos_[1][3].second[17]"
    std::string path_string() const {
        std::string s = root_;
        for (const Seg &g : path_) {
            if (g.name) s += "." + *g.name;
            else s += "[" + std::to_string(g.index) + "]";
        }
        return s;
    }
    // the message names the member being read, its type, and for the class types around it where their class information was taken from the stream:
    // a missing or surplus "tracking version" pair shifts every later token, so the last sites in front of the failure are the suspects
    [[noreturn]] void fail(const std::string &msg) const {
        std::string m = what_ + ": " + msg + " at " + path_string() + (current_ ? " (" + current_->name + ")" : "") + ", near byte " + std::to_string(p_ - begin());
        if (library_version_) m += "; archive library version " + std::to_string(library_version_);
        std::string chain;
        for (size_t k = path_.size(); k-- && chain.size() < 600;) {
            const TypeP t = path_[k].holder;
            if (!t || !has_class_info(t)) continue;
            const uint32_t site = seen_[t->id];
            chain += (chain.empty() ? "" : "; ") + t->name + (site ? ": class info read at byte " + std::to_string(sites_[site - 1].byte) + " (" + sites_[site - 1].path + ")" : ": no class info read");
        }
        if (!chain.empty()) m += "; enclosing types: " + chain;
        if (!sites_.empty()) m += "; last class info read: " + sites_.back().type + " at byte " + std::to_string(sites_.back().byte);
        throw Error(m);
    }
    const char *begin() const { return end_ - size_; }
    void skip_blank() {
        while (p_ != end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\r' || *p_ == '\t')) ++p_;
    }
    uint64_t unsigned_int() {
        skip_blank();
        uint64_t v = 0;
        auto r = std::from_chars(p_, end_, v);
        if (r.ec != std::errc()) fail("expected an unsigned integer");
        p_ = r.ptr;
        return v;
    }
    int64_t signed_int() {
        skip_blank();
        int64_t v = 0;
        auto r = std::from_chars(p_, end_, v);
        if (r.ec != std::errc()) fail("expected an integer");
        p_ = r.ptr;
        return v;
    }
    double real() {
        skip_blank();
        double v = 0;
        auto r = std::from_chars(p_, end_, v);
        if (r.ec == std::errc()) {
            p_ = r.ptr;
            return v;
        }
        // "nan", "-nan", "inf" as printed by iostreams, or a value from_chars rejects (out of range): strtod decides
        char buf[64];
        size_t n = 0;
        while (p_ + n != end_ && n < sizeof buf - 1 && !(p_[n] == ' ' || p_[n] == '\n' || p_[n] == '\r' || p_[n] == '\t')) {
            buf[n] = p_[n];
            ++n;
        }
        buf[n] = 0;
        char *e = nullptr;
        v = strtod(buf, &e);
        if (e == buf || *e) fail("expected a floating point number");
        p_ += n;
        return v;
    }
    std::string str() {
        const uint64_t n = unsigned_int();
        if (p_ == end_ || n > (uint64_t)(end_ - p_ - 1)) fail("string longer than the file");
        ++p_;   // the one blank behind the length
        std::string v(p_, p_ + n);
        p_ += n;
        return v;
    }
    struct Seg {
        TypeP holder;                  // the class / pair / container the segment is a part of
        const std::string *name;       // member name, or nullptr: item `index`
        uint64_t index;
    };
    const char *p_, *end_;
    size_t size_ = (size_t)(end_ - p_);
    std::vector<uint32_t> seen_;       // per type: 0, or 1 + index of its class-info site
    std::vector<ClassInfoSite> sites_;
    std::vector<Seg> path_;
    TypeP current_ = nullptr;
    std::string what_, root_;
    uint32_t library_version_ = 0;
    Grammar g_;

   public:
    void set_root(const std::string &name) { root_ = name; }
};

}  // namespace archive
}  // namespace rsq
