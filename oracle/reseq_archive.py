"""CPU restatement (numpy / plain Python) of how ReSeq turns its two profile files into the state the simulator reads.

TEST INFRASTRUCTURE ONLY -- like everything under oracle/: imported by tests/ as the checker of
reseq_amd/csrc/rsq_profile_archive.cpp, never by the product.

What is restated (citations relative to /root/reference):
  * the Boost *text archive* token stream of `DataStats` (reseq/DataStats.h:180-212 and the `serialize` members of
    AdapterStats.h:59-92, CoverageStats.h:281-315, ErrorStats.h:78-96, FragmentDuplicationStats.h:33-35,
    FragmentDistributionStats.h:440-456, Surrounding.h:63-65,89-91, QualityStats.h:156-197, TileStats.h:42-52,
    Vect.hpp:40-42, SeqQualityStats.hpp:19-21) and of `ProbabilityEstimates` (ProbabilityEstimates.h:1475-1483,
    LogIPF :955-967, LogArrayCalc :111-114) -- reader AND writer, so that tests can produce archives;
  * DataStats::PrepareProcessing (DataStats.cpp:1322-1328,698-703): total reads, AdapterStats::SumCounts
    (AdapterStats.cpp:840-883) and PrepareSimulation (:892-908), ErrorStats::PrepareSimulation (ErrorStats.cpp:202-209);
  * ProbabilityEstimates::PrepareResult (ProbabilityEstimates.cpp:961-1020): LogIPF::FullExpansion
    (ProbabilityEstimates.h:1004-1036) with LogArrayCalc::Expand (:253-290), LogArrayResult::GetResults (:386-453) and
    ImputeMissingValues (:455-479, including its weights, which give the NEARER filled row the smaller weight).

PARITY UNPINNED against Boost itself: neither Boost nor a ReSeq-written profile exists in this image, the reference's
own save/load test (ProbabilityEstimatesTest.cpp:1017-1035) needs Boost to run.  The token rules below are Boost's as
recalled (text_oarchive of Boost 1.6x-1.7x):
    header "22 serialization::archive <library version>"; numbers blank separated, char-sized integers as numbers, bool as
    0/1, double as %.17e; std::string "<length> <bytes>"; an object of class type is preceded by "<tracking> <version>"
    (0 0 here) the FIRST time its C++ type occurs in the archive -- class types being user classes, std::pair, std::array
    and std::vector of non-arithmetic items; std::vector "<count> <item_version>" + items (std::vector<bool>: no item
    version); std::array "<N>" + items; std::pair first, second.
The well-known `std::map<int,int>{{1,2},{3,4}}` archive "22 serialization::archive 17 0 0 2 0 0 0 1 2 3 4" is the shape
these rules reproduce for a vector of pairs (tests/test_profile_archive.py).
"""
import math
import re

import numpy as np

# ------------------------------------------------------------------------------------------------------------- types
_PRIMS = {
    # spelling used below -> (kind, C++ type that decides the identity of containers built from it)
    "u8": ("u", "unsigned char"), "u16": ("u", "unsigned short"), "u32": ("u", "unsigned int"),
    "u64": ("u", "unsigned long"), "i8": ("i", "signed char"), "f64": ("f", "double"), "bool": ("b", "bool"),
    "string": ("s", "std::string"),
}

# class name -> members in `serialize` order; T / N are template parameters
_CLASSES = {
    "Vect<T>": [("vec_", "pair<u64,vector<T>>")],
    "SeqQualityStats<T>": [("qualities_", "Vect<T>")],
    "AdapterStats": [
        ("names_", "array<vector<string>,2>"),
        ("combinations_", "vector<vector<bool>>"),
        ("counts_", "vector<vector<Vect<Vect<u64>>>>"),
        ("start_cut_", "array<vector<Vect<u64>>,2>"),
        ("polya_tail_length_", "Vect<u64>"),
        ("overrun_bases_", "array<u64,5>"),
        ("seqs_archive", "array<vector<string>,2>"),
    ],
    "CoverageStats": [
        ("coverage_threshold_", "u32"), ("reset_distance_", "u32"),
    ] + [(n, "array<array<array<Vect<Vect<u64>>,4>,5>,4>") for n in (
        "dominant_errors_by_distance_", "dominant_errors_by_gc_", "gc_by_distance_de_", "dominant_errors_by_start_rates_",
        "start_rates_by_distance_de_", "start_rates_by_gc_de_")
    ] + [(n, "array<array<Vect<Vect<u64>>,5>,4>") for n in (
        "error_rates_by_distance_", "error_rates_by_gc_", "gc_by_distance_er_", "error_rates_by_start_rates_",
        "start_rates_by_distance_er_", "start_rates_by_gc_er_")
    ] + [
        ("block_error_rate_", "Vect<u16>"), ("block_percent_systematic_", "Vect<u16>"),
        ("systematic_error_p_values_", "Vect<u64>"), ("coverage_", "Vect<u64>"),
        ("coverage_stranded_", "array<Vect<u64>,2>"), ("coverage_stranded_percent_", "array<Vect<u64>,2>"),
        ("coverage_stranded_percent_min_cov_10_", "array<Vect<u64>,2>"),
        ("coverage_stranded_percent_min_cov_20_", "array<Vect<u64>,2>"),
        ("error_coverage_", "Vect<u64>"), ("error_coverage_percent_", "Vect<u64>"),
        ("error_coverage_percent_min_cov_10_", "Vect<u64>"), ("error_coverage_percent_min_cov_20_", "Vect<u64>"),
        ("error_coverage_percent_stranded_", "Vect<Vect<u64>>"),
        ("error_coverage_percent_stranded_min_strand_cov_10_", "Vect<Vect<u64>>"),
        ("error_coverage_percent_stranded_min_strand_cov_20_", "Vect<Vect<u64>>"),
    ],
    "ErrorStats": [(n, "array<array<array<Vect<Vect<Vect<u64>>>,5>,4>,2>") for n in (
        "called_bases_by_base_quality_per_tile_", "called_bases_by_position_per_tile_", "called_bases_by_error_num_per_tile_",
        "called_bases_by_error_rate_per_tile_", "error_num_by_quality_per_tile_", "error_num_by_position_per_tile_",
        "error_num_by_error_rate_per_tile_")
    ] + [(n, "array<array<Vect<Vect<u64>>,6>,2>") for n in (
        "indel_by_indel_pos_", "indel_by_position_", "indel_by_gc_", "indel_pos_by_position_", "indel_pos_by_gc_", "gc_by_position_")
    ] + [
        ("errors_per_read_", "array<Vect<u64>,2>"),
        ("called_bases_by_base_quality_per_previous_called_base_", "array<array<array<array<Vect<u64>,6>,5>,4>,2>"),
    ],
    "FragmentDuplicationStats": [("duplication_number_", "Vect<u64>")],
    "SurroundingCount": [("counts_", "array<vector<u64>,3>")],
    "SurroundingBias": [("bias_", "array<vector<f64>,3>")],
    "FragmentDistributionStats": [
        ("abundance_", "vector<u64>"), ("insert_lengths_", "Vect<u64>"), ("gc_fragment_content_", "Vect<u64>"),
        ("fragment_surroundings_", "SurroundingCount"), ("site_count_", "Vect<Vect<u64>>"),
        ("outskirt_content_", "array<array<Vect<u64>,4>,2>"), ("ref_seq_bias_", "vector<f64>"),
        ("insert_lengths_bias_", "Vect<f64>"), ("gc_fragment_content_bias_", "Vect<f64>"),
        ("fragment_surroundings_bias_", "SurroundingBias"), ("dispersion_parameters_", "array<f64,2>"),
    ],
    "QualityStats": [
        ("base_quality_stats_per_tile_per_error_reference_", "array<array<array<Vect<Vect<SeqQualityStats<u64>>>,5>,4>,2>"),
        ("error_rate_for_position_per_tile_per_error_reference_", "array<array<array<Vect<Vect<Vect<u64>>>,5>,4>,2>"),
        ("base_quality_for_error_rate_per_tile_per_error_reference_", "array<array<array<Vect<Vect<Vect<u64>>>,5>,4>,2>"),
    ] + [(n, "array<array<Vect<Vect<Vect<u64>>>,4>,2>") for n in (
        "base_quality_for_preceding_quality_per_tile_reference_", "preceding_quality_for_error_rate_per_tile_reference_",
        "preceding_quality_for_position_per_tile_reference_", "base_quality_for_sequence_quality_per_tile_reference_",
        "preceding_quality_for_sequence_quality_per_tile_reference_", "sequence_quality_for_error_rate_per_tile_reference_",
        "sequence_quality_for_position_per_tile_reference_")
    ] + [
        ("sequence_quality_mean_for_gc_per_tile_reference_", "array<Vect<Vect<SeqQualityStats<u64>>>,2>"),
    ] + [(n, "array<Vect<Vect<Vect<u64>>>,2>") for n in (
        "sequence_quality_mean_for_mean_error_rate_per_tile_reference_", "sequence_quality_mean_for_fragment_length_per_tile_reference_",
        "mean_error_rate_for_gc_per_tile_reference_", "mean_error_rate_for_fragment_length_per_tile_reference_",
        "gc_for_fragment_length_per_tile_reference_")
    ] + [
        ("base_quality_for_sequence_per_tile_", "array<array<Vect<Vect<Vect<u64>>>,5>,2>"),
        ("base_quality_for_preceding_quality_per_tile_", "array<array<Vect<Vect<Vect<u64>>>,5>,2>"),
        ("base_quality_stats_per_tile_", "array<array<Vect<Vect<SeqQualityStats<u64>>>,5>,2>"),
        ("preceding_quality_for_sequence_per_tile_", "array<array<Vect<Vect<Vect<u64>>>,5>,2>"),
        ("preceding_quality_for_position_per_tile_", "array<array<Vect<Vect<Vect<u64>>>,5>,2>"),
        ("sequence_quality_for_position_per_tile_", "array<array<Vect<Vect<Vect<u64>>>,5>,2>"),
        ("base_quality_stats_per_strand_", "array<Vect<SeqQualityStats<u64>>,2>"),
        ("sequence_quality_for_base_per_tile_", "array<array<Vect<Vect<SeqQualityStats<u64>>>,5>,2>"),
        ("sequence_quality_mean_paired_per_tile_", "Vect<Vect<Vect<u64>>>"),
        ("sequence_quality_mean_for_gc_per_tile_", "array<Vect<Vect<SeqQualityStats<u64>>>,2>"),
    ] + [(n, "array<Vect<u64>,2>") for n in (
        "sequence_quality_probability_mean_", "sequence_quality_minimum_", "sequence_quality_first_quartile_",
        "sequence_quality_median_", "sequence_quality_third_quartile_", "sequence_quality_maximum_")
    ] + [
        ("sequence_quality_content_", "array<Vect<Vect<u64>>,2>"),
        ("homoquality_distribution_", "Vect<Vect<u64>>"),
        ("nucleotide_quality_", "array<array<SeqQualityStats<u64>,5>,2>"),
    ],
    "TileStats": [("tiles_", "vector<u16>"), ("abundance_", "vector<u64>")],
    "DataStats": [
        ("adapters_", "AdapterStats"), ("coverage_", "CoverageStats"), ("errors_", "ErrorStats"),
        ("duplicates_", "FragmentDuplicationStats"), ("fragment_distribution_", "FragmentDistributionStats"),
        ("qualities_", "QualityStats"), ("tiles_", "TileStats"),
        ("creation_time_", "u64"), ("read_lengths_", "array<Vect<u64>,2>"),
        ("read_lengths_by_fragment_length_", "array<Vect<Vect<u64>>,2>"),
        ("non_mapped_read_lengths_by_fragment_length_", "array<Vect<Vect<u64>>,2>"),
        ("phred_quality_offset_", "u8"), ("minimum_quality_", "u8"), ("maximum_quality_", "u8"),
        ("minimum_read_length_on_reference_", "u16"), ("maximum_read_length_on_reference_", "u16"),
        ("corrected_coverage_", "f64"),
        ("proper_pair_mapping_quality_", "Vect<u64>"), ("improper_pair_mapping_quality_", "Vect<u64>"),
        ("single_read_mapping_quality_", "Vect<u64>"),
        ("gc_read_content_", "array<Vect<u64>,2>"), ("gc_read_content_reference_", "array<Vect<u64>,2>"),
        ("gc_read_content_mapped_", "array<Vect<u64>,2>"), ("n_content_", "array<Vect<u64>,2>"),
        ("sequence_content_", "array<array<Vect<u64>,5>,2>"),
        ("sequence_content_reference_", "array<array<array<Vect<u64>,4>,2>,2>"),
        ("homopolymer_distribution_", "array<Vect<u64>,5>"),
    ],
    # margins of an N-dimensional fit: N(N-1)/2
    "LogArrayCalc<N>": [("dim2_", "array<vector<f64>,M>"), ("dim_size_", "array<u32,N>")],
    "LogIPF<N>": [
        ("steps_", "u32"), ("needed_updates_", "u32"), ("precision_", "f64"), ("margin_precision_", "array<f64,M>"),
        ("last_margin_", "u16"), ("last_update_", "array<u32,M>"), ("update_dist_", "array<u16,M>"),
        ("estimates_", "LogArrayCalc<N>"), ("dim_indices_", "array<vector<u32>,N>"),
        ("initial_dim_indices_reduced_", "array<vector<u32>,N>"), ("dim_indices_reduced_", "array<vector<u32>,N>"),
    ],
    "ProbabilityEstimates": [
        ("stats_creation_time_", "u64"),
        ("quality_", "array<vector<array<LogIPF<5>,4>>,2>"),
        ("sequence_quality_", "array<vector<LogIPF<4>>,2>"),
        ("base_call_", "array<vector<array<array<LogIPF<5>,5>,4>>,2>"),
        ("dom_error_", "array<array<array<LogIPF<4>,5>,5>,4>"),
        ("error_rate_", "array<array<LogIPF<4>,5>,4>"),
        ("indels_", "array<array<LogIPF<4>,6>,2>"),
    ],
}


class Type:
    __slots__ = ("kind", "name", "elem", "n", "members", "class_info")

    def __init__(self, kind, name, elem=None, n=0, members=None, class_info=False):
        self.kind, self.name, self.elem, self.n, self.members, self.class_info = kind, name, elem, n, members, class_info


_TYPES = {}


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return out


def get_type(expr):
    """Type of a C++-like spelling ("array<vector<Vect<u64>>,2>"); one object per distinct type."""
    expr = expr.replace(" ", "")
    if expr in _TYPES:
        return _TYPES[expr]
    if expr in _PRIMS:
        t = Type(_PRIMS[expr][0], _PRIMS[expr][1])
    else:
        head, _, rest = expr.partition("<")
        args = _split_args(rest[:-1]) if rest else []
        if head == "vector":
            e = get_type(args[0])
            t = Type("vec", "std::vector<%s>" % e.name, elem=e, class_info=e.kind not in "uifb")
        elif head == "array":
            e = get_type(args[0])
            t = Type("arr", "std::array<%s,%s>" % (e.name, args[1]), elem=e, n=int(args[1]), class_info=True)
        elif head == "pair":
            a, b = get_type(args[0]), get_type(args[1])
            t = Type("pair", "std::pair<%s,%s>" % (a.name, b.name), members=[("first", a), ("second", b)], class_info=True)
        else:
            if args and (head + "<T>") in _CLASSES:
                inner = get_type(args[0])
                subst = {"T": args[0]}
                name = "%s<%s>" % (head, inner.name)
                decl = _CLASSES[head + "<T>"]
            elif args and (head + "<N>") in _CLASSES:
                n = int(args[0])
                subst = {"N": str(n), "M": str(n * (n - 1) // 2)}
                name = "%s<%d>" % (head, n)
                decl = _CLASSES[head + "<N>"]
            else:
                subst, name, decl = {}, head, _CLASSES[expr]
            members = []
            for mname, mexpr in decl:
                for k, v in subst.items():
                    mexpr = re.sub(r"\b%s\b" % k, v, mexpr)
                members.append((mname, get_type(mexpr)))
            t = Type("cls", "reseq::" + name, members=members, class_info=True)
    if t.name in _TYPES:            # two spellings of one C++ type
        t = _TYPES[t.name]
    _TYPES[expr] = t
    _TYPES[t.name] = t
    return t


def default_value(t):
    """What a default-constructed member looks like (used by the writer for members a test does not fill)."""
    if t.kind in "uib":
        return 0
    if t.kind == "f":
        return 0.0
    if t.kind == "s":
        return ""
    if t.kind == "vec":
        return []
    if t.kind == "arr":
        return [default_value(t.elem) for _ in range(t.n)]
    if t.kind == "pair":
        return (default_value(t.members[0][1]), default_value(t.members[1][1]))
    return {}


# ------------------------------------------------------------------------------------------------------------ writer
def _fmt_double(x):
    return "%.17e" % x


# The token rules that could not be checked against a Boost-written file, as switches (reseq_amd/csrc/rsq_archive.h Grammar): the default is the recalled rule set.
DEFAULT_GRAMMAR = dict(item_version=1, array_class_info=True, array_count=True, pair_class_info=True, arithmetic_vector_class_info=False)


def all_grammars():
    """every combination of the doubtful rules, the recalled one first"""
    out = [dict(DEFAULT_GRAMMAR)]
    for iv in (1, 2, 3, 0):
        for aci in (True, False):
            for ac in (True, False):
                for pci in (True, False):
                    for avci in (False, True):
                        g = dict(item_version=iv, array_class_info=aci, array_count=ac, pair_class_info=pci, arithmetic_vector_class_info=avci)
                        if g != DEFAULT_GRAMMAR:
                            out.append(g)
    return out


def _has_class_info(t, g):
    if t.kind == "arr":
        return g["array_class_info"]
    if t.kind == "pair":
        return g["pair_class_info"]
    if t.kind == "vec":
        return t.elem.kind not in "uifb" or g["arithmetic_vector_class_info"]
    return t.class_info


def _has_item_version(t, g):
    iv = g["item_version"]
    return iv == 3 or (iv == 1 and t.elem.kind != "b") or (iv == 2 and t.elem.kind not in "uifb")


class _Writer:
    def __init__(self, library_version, grammar=None):
        self.tok = ["22 serialization::archive", str(library_version)]
        self.seen = set()
        self.library_version = library_version
        self.g = dict(DEFAULT_GRAMMAR, **(grammar or {}))

    def put(self, t, v):
        if _has_class_info(t, self.g) and t.name not in self.seen:
            self.seen.add(t.name)
            self.tok.append("0 0")
        k = t.kind
        if k in "ui":
            self.tok.append(str(int(v)))
        elif k == "b":
            self.tok.append("1" if v else "0")
        elif k == "f":
            self.tok.append(_fmt_double(float(v)))
        elif k == "s":
            self.tok.append("%d %s" % (len(v.encode()), v))
        elif k == "vec":
            self.tok.append(str(len(v)))
            if _has_item_version(t, self.g) and self.library_version > 3:
                self.tok.append("0")
            self.items(t.elem, v)
        elif k == "arr":
            if len(v) != t.n:
                raise ValueError("%s given %d items" % (t.name, len(v)))
            if self.g["array_count"]:
                self.tok.append(str(t.n))
            self.items(t.elem, v)
        elif k == "pair":
            self.put(t.members[0][1], v[0])
            self.put(t.members[1][1], v[1])
        else:
            unknown = set(v) - {m for m, _ in t.members}
            if unknown:
                raise KeyError("%s has no member %s" % (t.name, sorted(unknown)))
            for mname, mt in t.members:
                self.put(mt, v[mname] if mname in v else default_value(mt))

    def items(self, e, v):
        if e.kind == "f" and len(v):
            a = np.asarray(v, dtype=np.float64)
            self.tok.append(" ".join(["%.17e"] * len(a)) % tuple(a.tolist()))
        elif e.kind in "ui" and len(v):
            self.tok.append(" ".join(map(str, np.asarray(v).tolist())))
        else:
            for x in v:
                self.put(e, x)


def dumps(type_expr, value, library_version=17, grammar=None):
    w = _Writer(library_version, grammar)
    w.put(get_type(type_expr), value)
    return " ".join(w.tok) + "\n"


def write_archive(path, type_expr, value, library_version=17, grammar=None):
    with open(path, "w") as f:
        f.write(dumps(type_expr, value, library_version, grammar))


# ------------------------------------------------------------------------------------------------------------ reader
class _Reader:
    def __init__(self, data):
        self.data = data
        self.pos = 0
        if self.string() != "serialization::archive":
            raise ValueError("not a Boost text archive")
        self.library_version = self.integer()
        self.seen = set()

    _tok = re.compile(rb"\s*(\S+)")

    def token(self):
        m = self._tok.match(self.data, self.pos)
        if not m:
            raise ValueError("archive ends early")
        self.pos = m.end()
        return m.group(1)

    def integer(self):
        return int(self.token())

    def string(self):
        n = self.integer()
        s = self.data[self.pos + 1:self.pos + 1 + n]
        if len(s) != n:
            raise ValueError("archive ends inside a string")
        self.pos += 1 + n
        return s.decode()

    def bulk(self, n):
        """the next n tokens as a list of bytes"""
        m = re.compile(rb"\s*(?:\S+\s+){%d}\S+" % (n - 1)).match(self.data, self.pos)
        if not m:
            raise ValueError("archive ends early")
        self.pos = m.end()
        return m.group(0).split()

    def get(self, t):
        if t.class_info and t.name not in self.seen:
            self.seen.add(t.name)
            tracking, _version = self.integer(), self.integer()
            if tracking:
                raise ValueError("object tracking in " + t.name)
        k = t.kind
        if k in "ui":
            return self.integer()
        if k == "b":
            return bool(self.integer())
        if k == "f":
            return float(self.token())
        if k == "s":
            return self.string()
        if k in ("vec", "arr"):
            n = self.integer()
            if k == "arr" and n != t.n:
                raise ValueError("%s holds %d items" % (t.name, n))
            if k == "vec" and t.elem.kind != "b" and self.library_version > 3:
                self.integer()
            e = t.elem
            if e.kind == "f":
                return np.array(list(map(float, self.bulk(n))) if n else [], dtype=np.float64)
            if e.kind in "ui":
                return list(map(int, self.bulk(n))) if n else []
            return [self.get(e) for _ in range(n)]
        if k == "pair":
            a = self.get(t.members[0][1])
            return (a, self.get(t.members[1][1]))
        return {mname: self.get(mt) for mname, mt in t.members}


def loads(type_expr, data):
    if isinstance(data, str):
        data = data.encode()
    r = _Reader(data)
    v = r.get(get_type(type_expr))
    if data[r.pos:].strip():
        raise ValueError("tokens left after the last member")
    return v


def read_archive(path, type_expr):
    with open(path, "rb") as f:
        return loads(type_expr, f.read())


# --------------------------------------------------------------------------------------- DataStats::PrepareProcessing
def vect(offset, values):
    """a reseq::Vect as the (reader's / writer's) value tree has it"""
    return {"vec_": (int(offset), values)}


def _from(v):
    return v["vec_"][0]


def _vals(v):
    return v["vec_"][1]


def _to(v):
    return _from(v) + len(_vals(v))


def _dna(c):                      # seqan::Dna from a character
    return {"C": 1, "c": 1, "G": 2, "g": 2, "T": 3, "t": 3, "U": 3, "u": 3}.get(c, 0)


def sum_adapter_counts(counts, seqs, n_adapters):
    """AdapterStats::SumCounts (AdapterStats.cpp:840-883)."""
    count_sum = [[0] * n_adapters[0], [0] * n_adapters[1]]

    def first_diff(a, b):
        k = 0
        while k < min(len(a), len(b)) and _dna(a[k]) == _dna(b[k]):
            k += 1
        return k

    before_a1 = 0
    for a1 in range(len(counts) - 1, -1, -1):
        after_a1 = first_diff(seqs[0][a1], seqs[0][a1 - 1]) if a1 else 0
        before_a2 = 0
        for a2 in range(len(counts[0]) - 1, -1, -1):
            after_a2 = first_diff(seqs[1][a2], seqs[1][a2 - 1]) if a2 else 0
            total = 0
            c = counts[a1][a2]
            for pos1 in range(max(before_a1, after_a1, _from(c) & 0xFFFF), _to(c)):
                row = _vals(c)[pos1 - _from(c)]
                for pos2 in range(max(before_a2, after_a2, _from(row) & 0xFFFF), _to(row)):
                    total += _vals(row)[pos2 - _from(row)]
            count_sum[0][a1] += total
            count_sum[1][a2] += total
            before_a2 = after_a2
        before_a1 = after_a1
    return count_sum


def prepare_stats(st):
    """The RSQP arrays (reseq_amd/container.py) that come from DataStats."""
    out = {}
    out["phred_quality_offset"] = np.asarray([st["phred_quality_offset_"]], np.uint8)
    out["corrected_coverage"] = np.asarray([st["corrected_coverage_"]], np.float64)
    max_del = 0
    for v in st["errors_"]["indel_by_indel_pos_"][1]:          # ErrorStats::PrepareSimulation
        max_del = max(max_del, _to(v))
    out["errors.max_len_deletion"] = np.asarray([max_del], np.uint16)
    out["coverage.reset_distance"] = np.asarray([st["coverage_"]["reset_distance_"]], np.uint32)

    def put_vect(name, v, dtype):
        out[name] = np.asarray(_vals(v), dtype=dtype)
        out[name + ".from"] = np.asarray([_from(v)], np.uint64)

    fd = st["fragment_distribution_"]
    put_vect("frag.insert_lengths", fd["insert_lengths_"], np.uint64)
    put_vect("frag.insert_lengths_bias", fd["insert_lengths_bias_"], np.float64)
    put_vect("frag.gc_bias", fd["gc_fragment_content_bias_"], np.float64)
    out["frag.sur_bias"] = np.concatenate([np.asarray(b, np.float64) for b in fd["fragment_surroundings_bias_"]["bias_"]])
    out["frag.dispersion_parameters"] = np.asarray(fd["dispersion_parameters_"], np.float64)
    out["frag.ref_seq_bias"] = np.asarray(fd["ref_seq_bias_"], np.float64)

    for seg in range(2):
        put_vect("read_lengths.%d" % seg, st["read_lengths_"][seg], np.uint64)
        by_fl = st["read_lengths_by_fragment_length_"][seg]
        non_mapped = st["non_mapped_read_lengths_by_fragment_length_"][seg]
        row_ptr, row_from, values, nm_values = [0], [], [], []
        for fl in range(_from(by_fl), _to(by_fl)):
            row = _vals(by_fl)[fl - _from(by_fl)]
            row_from.append(_from(row))
            for rl in range(_from(row), _to(row)):
                values.append(_vals(row)[rl - _from(row)])
                nm = 0                                           # Vect::operator[] const: 0 outside the stored range
                if _from(non_mapped) <= fl < _to(non_mapped):
                    nrow = _vals(non_mapped)[fl - _from(non_mapped)]
                    if _from(nrow) <= rl < _to(nrow):
                        nm = _vals(nrow)[rl - _from(nrow)]
                nm_values.append(nm)
            row_ptr.append(len(values))
        out["rl_by_fl.%d.from" % seg] = np.asarray([_from(by_fl)], np.uint64)
        out["rl_by_fl.%d.row_ptr" % seg] = np.asarray(row_ptr, np.uint32)
        out["rl_by_fl.%d.row_from" % seg] = np.asarray(row_from, np.uint32)
        out["rl_by_fl.%d.values" % seg] = np.asarray(values, np.uint64)
        out["rl_by_fl_nonmapped.%d.values" % seg] = np.asarray(nm_values, np.uint64)

    out["tiles.tiles"] = np.asarray(st["tiles_"]["tiles_"], np.uint16)
    out["tiles.abundance"] = np.asarray(st["tiles_"]["abundance_"], np.uint64)

    ad = st["adapters_"]
    seqs = ad["seqs_archive"]
    sums = sum_adapter_counts(ad["counts_"], seqs, [len(ad["start_cut_"][0]), len(ad["start_cut_"][1])])
    for seg in range(2):
        codes = [np.asarray([_dna(c) for c in s], np.uint8) for s in seqs[seg]]
        out["adapters.%d.seqs" % seg] = np.concatenate(codes) if codes else np.zeros(0, np.uint8)
        out["adapters.%d.seq_ptr" % seg] = np.cumsum([0] + [len(c) for c in codes]).astype(np.uint32)
        cnt = np.asarray(sums[seg], np.uint64)
        out["adapters.%d.counts" % seg] = cnt
        sig = cnt.copy()
        if len(cnt):                                             # AdapterStats::PrepareSimulation
            sig[cnt < np.uint64(math.ceil(int(cnt.max()) * 0.1))] = 0
        out["adapters.%d.significant_counts" % seg] = sig
        cuts = ad["start_cut_"][seg]
        out["adapters.%d.start_cut_ptr" % seg] = np.cumsum([0] + [len(_vals(c)) for c in cuts]).astype(np.uint32)
        out["adapters.%d.start_cut_from" % seg] = np.asarray([_from(c) for c in cuts], np.uint32)
        out["adapters.%d.start_cut" % seg] = np.asarray([x for c in cuts for x in _vals(c)], np.uint64)
    put_vect("adapters.polya_tail_length", ad["polya_tail_length_"], np.uint64)
    out["adapters.overrun_bases"] = np.asarray(ad["overrun_bases_"], np.uint64)
    return out


# ------------------------------------------------------------------------------- ProbabilityEstimates::PrepareResult
def full_expansion(ipf, n_dims):
    """LogIPF::FullExpansion + LogArrayCalc::Expand: the margins (a, 0), a = 1..N-1, at full size.

    Returns a list of arrays [rows of dimension a, bins of dimension 0] in the order of the stored data."""
    size0 = len(ipf["dim_indices_"][0])
    necessary = False
    for key in ("dim_indices_reduced_", "initial_dim_indices_reduced_"):
        for n in range(n_dims):
            m = ipf[key][n]
            if any(m[i] != i for i in range(len(m))):
                necessary = True
    stored = ipf["estimates_"]["dim2_"]
    if not necessary:
        return [np.asarray(stored[m], np.float64).reshape(len(ipf["dim_indices_"][m + 1]), size0) for m in range(n_dims - 1)]
    combined = [[ipf["dim_indices_reduced_"][n][b] for b in ipf["initial_dim_indices_reduced_"][n]] for n in range(n_dims)]
    count = []
    for n in range(n_dims):
        c = [0] * (max(combined[n]) + 1)
        for b in combined[n]:
            c[b] += 1
        count.append(c)
    mult = [np.asarray([math.pow(1.0 / count[n][b], 1.0 / (n_dims - 1)) for b in combined[n]], np.float64) for n in range(n_dims)]
    out = []
    for m in range(n_dims - 1):
        a = m + 1
        old = np.asarray(stored[m], np.float64).reshape(len(count[a]), len(count[0]))
        new = old[np.ix_(combined[a], combined[0])]
        out.append((new * mult[a][:, None]) * mult[0][None, :])
    return out


def get_results(margins, dim_indices):
    """LogArrayResult::GetResults: dict(par0, limits, dim2 as list of 2-D arrays)."""
    nm = len(margins)
    k = len(dim_indices[0])
    if k == 0:
        return dict(par0=np.zeros(0, np.uint32), limits=np.zeros((nm, 2), np.uint32), dim2=[np.zeros((0, 0)) for _ in range(nm)])
    key = np.zeros(k)
    for n in range(nm - 1, -1, -1):
        m = margins[n]
        col_sum = np.cumsum(m[::-1], axis=0)[-1]               # sequential sum, last row first
        key = key + col_sum / m.shape[0]
    order = sorted(range(k), key=lambda j: (key[j], j))        # std::sort of pair<double, index>
    column = [0] * k
    for c, j in enumerate(order):
        column[j] = c
    limits = np.zeros((nm, 2), np.uint32)
    dim2 = []
    for n in range(nm):
        val = dim_indices[n + 1]
        lo, hi = min(val), max(val) + 1
        limits[n] = (lo, hi)
        d = np.zeros((hi - lo, k))
        for i in range(len(val) - 1, -1, -1):                   # later (lower i) writes win, as in the reference
            d[val[i] - lo, column] = margins[n][i]
        dim2.append(d)
    par0 = np.asarray([dim_indices[0][j] for j in order], np.uint32)
    return dict(par0=par0, limits=limits, dim2=dim2)


def impute_missing_values(res):
    """LogArrayResult::ImputeMissingValues, weights as written there."""
    for d in res["dim2"]:
        last = 0
        for i in range(1, d.shape[0]):
            if not np.any(d[i] != 0.0):
                continue
            for g in range(last + 1, i):
                d[g] = d[last] * (g - last) / (i - last) + d[i] * (i - g) / (i - last)
            last = i
    return res


def prepare_table(ipf, n_dims):
    res = impute_missing_values(get_results(full_expansion(ipf, n_dims), ipf["dim_indices_"]))
    k = len(res["par0"])
    flat = np.concatenate([d.ravel() for d in res["dim2"]]) if k else np.zeros(0)
    return dict(par0=res["par0"], limits=res["limits"], dim2=flat)


def prepare_estimates(pe, n_tiles):
    out = {}

    def put(prefix, ipf, n_dims):
        for kk, v in prepare_table(ipf, n_dims).items():
            out["tab.%s.%s" % (prefix, kk)] = v

    for seg in range(2):
        for tile in range(n_tiles):
            put("seq_quality.%d.%d" % (seg, tile), pe["sequence_quality_"][seg][tile], 4)
            for base in range(4):
                put("quality.%d.%d.%d" % (seg, tile, base), pe["quality_"][seg][tile][base], 5)
                for dom in range(5):
                    put("base_call.%d.%d.%d.%d" % (seg, tile, base, dom), pe["base_call_"][seg][tile][base][dom], 5)
    for base in range(4):
        for prev in range(5):
            for dom5 in range(5):
                put("dom_error.%d.%d.%d" % (base, prev, dom5), pe["dom_error_"][base][prev][dom5], 4)
        for dom in range(5):
            put("error_rate.%d.%d" % (base, dom), pe["error_rate_"][base][dom], 4)
    for t in range(2):
        for call in range(6):
            put("indels.%d.%d" % (t, call), pe["indels_"][t][call], 4)
    return out


def load_profile(stats_path, ipf_path=None):
    """`.reseq` + `.reseq.ipf` -> the dict of RSQP arrays the simulation is packed from."""
    st = read_archive(stats_path, "DataStats")
    pe = read_archive(ipf_path or stats_path + ".ipf", "ProbabilityEstimates")
    if pe["stats_creation_time_"] != st["creation_time_"]:
        raise ValueError("probability estimates belong to other statistics")
    out = prepare_stats(st)
    out.update(prepare_estimates(pe, len(st["tiles_"]["tiles_"])))
    return out
