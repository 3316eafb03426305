"""ctypes access to oracle/liboracle.so -- the CPU checker (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)


class Fragment(C.Structure):
    _fields_ = [("seq", C.c_uint32), ("start", C.c_uint32), ("len", C.c_uint32), ("dup", C.c_uint16),
                ("strand", C.c_uint8), ("pad", C.c_uint8), ("block", C.c_uint32), ("number", C.c_uint32)]


FRAGMENT_DTYPE = np.dtype([("seq", "<u4"), ("start", "<u4"), ("len", "<u4"), ("dup", "<u2"), ("strand", "u1"),
                           ("pad", "u1"), ("block", "<u4"), ("number", "<u4")])


FRAGMENT_VAR_DTYPE = np.dtype([("seq", "<u4"), ("start", "<u4"), ("len", "<u4"), ("dup", "<u2"), ("strand", "u1"), ("allele", "u1"), ("block", "<u4"),
                               ("number", "<u4"), ("end", "<u4"), ("sub", "<u4"), ("start_var", "<i4"), ("start_var_pos", "<u4"), ("end_var", "<i4"),
                               ("end_var_pos", "<u4")])


class OrcVariant(C.Structure):
    _fields_ = [("position", C.c_uint32), ("len", C.c_uint32), ("var_seq", u8p), ("allele", C.c_uint64 * 2)]


class OrcVariants(C.Structure):
    _fields_ = [("num_alleles", C.c_uint32), ("n_seqs", C.c_uint32), ("n", u32p), ("v", C.POINTER(C.POINTER(OrcVariant)))]


class Read(C.Structure):
    _fields_ = [("read_len", C.c_uint16), ("num_errors", C.c_uint16), ("seq", C.c_uint8 * 1024),
                ("qual", C.c_uint8 * 1024), ("cigar", C.c_char * 4096)]


class Text(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t), ("cap", C.c_size_t)]


class PhiloxOut(C.Structure):
    _fields_ = [("w", C.c_uint32 * 4)]


class Table(C.Structure):
    _fields_ = [("k", C.c_uint32), ("nm", C.c_uint32), ("par0", u32p), ("from_", C.c_uint32 * 4),
                ("to", C.c_uint32 * 4), ("dim2", f64p * 4)]


class DomBase(C.Structure):
    _fields_ = [("dom_base", C.c_uint8), ("content", C.c_uint16 * 5)]


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def _ptr(a, t):
    return a.ctypes.data_as(t)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    sig = {
        "orc_philox4x32_10": (PhiloxOut, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
        "orc_u32": (C.c_double, [C.c_uint32]),
        "orc_u53": (C.c_double, [C.c_uint32, C.c_uint32]),
        "orc_draw": (C.c_uint32, [C.POINTER(Table), u32p, C.c_double, f64p]),
        "orc_max_value": (C.c_uint32, [C.POINTER(Table)]),
        "orc_most_likely": (C.c_uint32, [C.POINTER(Table)]),
        "orc_profile_load": (C.c_void_p, [C.c_char_p]),
        "orc_profile_free": (None, [C.c_void_p]),
        "orc_profile_change_error_rate": (None, [C.c_void_p, C.c_double]),
        "orc_profile_remove_substitution_errors": (None, [C.c_void_p]),
        "orc_profile_remove_indel_errors": (None, [C.c_void_p]),
        "orc_profile_table": (C.POINTER(Table), [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
        "orc_divide_u32": (C.c_uint32, [C.c_uint32, C.c_uint32]),
        "orc_percent_u16": (C.c_uint8, [C.c_uint16, C.c_uint16]),
        "orc_percent_u32": (C.c_uint8, [C.c_uint32, C.c_uint32]),
        "orc_percent_u64": (C.c_uint8, [C.c_uint64, C.c_uint64]),
        "orc_safe_percent_u16": (C.c_uint8, [C.c_uint16, C.c_uint16]),
        "orc_transform_distance": (C.c_uint32, [C.c_uint32]),
        "orc_inv_logit2": (C.c_double, [C.c_double]),
        "orc_dombase_clear": (None, [C.POINTER(DomBase)]),
        "orc_dombase_set": (None, [C.POINTER(DomBase), u8p, C.c_uint32, C.c_uint32]),
        "orc_dombase_update": (None, [C.POINTER(DomBase), C.c_uint8, u8p, C.c_uint32, C.c_uint32]),
        "orc_surrounding_forward": (None, [u8p, C.c_uint32, C.c_uint32, i32p]),
        "orc_surrounding_reverse": (None, [u8p, C.c_uint32, C.c_uint32, i32p]),
        "orc_surrounding_update_forward": (None, [u8p, C.c_uint32, C.c_uint32, i32p]),
        "orc_surrounding_update_reverse": (None, [u8p, C.c_uint32, C.c_uint32, i32p]),
        "orc_combine_positions": (None, [f64p, f64p]),
        "orc_separate_positions": (None, [f64p, f64p]),
        "orc_surrounding_bias": (C.c_double, [f64p, i32p]),
        "orc_get_dispersion": (C.c_double, [C.c_double, C.c_double, C.c_double]),
        "orc_binomial": (C.c_uint16, [C.c_uint16, C.c_double, C.c_double]),
        "orc_negative_binomial": (C.c_uint16, [C.c_double, C.c_double, C.c_double]),
        "orc_calculate_non_zero_threshold": (C.c_double, [f64p, C.c_double, C.c_double, C.c_uint16]),
        "orc_fragment_counts_core": (C.c_uint16, [f64p, f64p, C.c_double, C.c_double, C.c_double, C.c_double, i32p, i32p,
                                                  C.c_double, C.c_uint16]),
        "orc_sum_bias": (C.c_double, [f64p, C.c_uint32, C.c_uint32, f64p, u8p, C.c_uint32, C.c_uint32, C.c_double, f64p]),
        "orc_select_allele": (None, [u16p, u32p, u8p, C.c_uint16, C.c_double]),
        "orc_reference_new": (C.c_void_p, [C.c_uint32]),
        "orc_reference_set": (None, [C.c_void_p, C.c_uint32, C.c_char_p, u8p, C.c_uint32]),
        "orc_reference_free": (None, [C.c_void_p]),
        "orc_reference_replace_n": (None, [C.c_void_p, C.c_uint64]),
        "orc_sim_new": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_double, C.c_char_p]),
        "orc_sim_new_bias": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_double, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]),
        "orc_sim_ref_seq_bias": (f64p, [C.c_void_p]),
        "orc_compress_sys_error_rate": (C.c_uint8, [C.c_uint8]),
        "orc_expand_sys_error_rate": (C.c_uint8, [C.c_uint8]),
        "orc_create_sys_error_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Text)]),
        "orc_sim_read_methylation": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]),
        "orc_parse_methylation": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint32, u32p, u32p, u32p, f64p, C.c_uint32, C.c_char_p, C.c_size_t]),
        "orc_variant_in_allele": (C.c_int, [C.POINTER(OrcVariant), C.c_uint32]),
        "orc_variant_first_allele": (C.c_uint32, [C.POINTER(OrcVariant)]),
        "orc_insert_variant": (None, [C.POINTER(OrcVariants), C.c_uint32, C.c_uint32, u8p, C.c_uint32, C.POINTER(C.c_uint64)]),
        "orc_variants_new": (C.POINTER(OrcVariants), [C.c_uint32]),
        "orc_read_variants": (C.POINTER(OrcVariants), [C.c_char_p, C.c_void_p, C.c_char_p, C.c_size_t]),
        "orc_variants_free": (None, [C.POINTER(OrcVariants)]),
        "orc_sur_change_base": (None, [i32p, C.c_uint32, C.c_uint8]),
        "orc_sur_delete_shift_right": (None, [i32p, C.c_uint32, C.c_uint8]),
        "orc_sur_delete_shift_left": (None, [i32p, C.c_uint32, C.c_uint8]),
        "orc_sur_insert_shift_right": (None, [i32p, C.c_uint32, u8p, C.c_uint32]),
        "orc_sur_insert_shift_left": (None, [i32p, C.c_uint32, u8p, C.c_uint32]),
        "orc_sim_load_sys_errors": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
        "orc_sim_free": (None, [C.c_void_p]),
        "orc_sim_set_normalization": (None, [C.c_void_p, C.c_double, f64p]),
        "orc_coverage_prop_lost_from_adapters": (C.c_double, [C.c_void_p]),
        "orc_coverage_to_number_pairs": (C.c_uint64, [C.c_double, C.c_uint64, C.c_double, C.c_double]),
        "orc_number_pairs_to_coverage": (C.c_double, [C.c_uint64, C.c_uint64, C.c_double, C.c_double]),
        "orc_sieve_blocks": (C.c_uint64, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.POINTER(Fragment))]),
        "orc_create_reads": (C.c_int, [C.c_void_p, C.POINTER(Fragment), C.c_uint64, C.POINTER(Text), C.POINTER(Text)]),
        "orc_simulate_adapter_only_pairs": (C.c_int, [C.c_void_p, C.POINTER(Text), C.POINTER(Text)]),
        "orc_text_free": (None, [C.POINTER(Text)]),
        "orc_error_model_only": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, u8p, u8p, u32p, u8p,
                                           u8p, C.POINTER(Read), u16p]),
        "orc_systematic_errors": (None, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, u8p, C.c_uint32, C.c_int,
                                         C.c_uint16, u8p, u8p, u8p]),
        "orc_sim_bias_normalization": (C.c_double, [C.c_void_p]),
        "orc_sieve_blocks_literal": (C.c_uint64, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.POINTER(Fragment))]),
        "orc_literal_hits": (C.c_uint32, [C.c_void_p, f64p, C.c_uint32, C.c_uint32, C.c_void_p]),
        "orc_gap_table": (None, [C.c_void_p, C.c_uint32, f64p, u32p]),
        "orc_gap_hits": (C.c_uint32, [C.c_void_p, f64p, u32p, f64p, C.c_uint32, C.c_uint32, C.c_void_p]),
        "orc_sim_n_groups": (C.c_uint32, [C.c_void_p]),
        "orc_sim_insert_to": (C.c_uint32, [C.c_void_p]),
        "orc_sim_total_pairs": (C.c_uint64, [C.c_void_p]),
        "orc_sim_adapter_only_pairs": (C.c_uint64, [C.c_void_p]),
        "orc_sim_total_blocks": (C.c_uint32, [C.c_void_p]),
        "orc_sim_gc_range": (C.c_uint16, [C.c_void_p]),
        "orc_sim_thresholds": (f64p, [C.c_void_p]),
        "orc_sim_norm_by_len": (f64p, [C.c_void_p]),
        "orc_sim_coverage_groups": (u32p, [C.c_void_p]),
        "orc_sim_sys_dom": (u8p, [C.c_void_p, C.c_int, C.c_uint32]),
        "orc_sim_sys_rate": (u8p, [C.c_void_p, C.c_int, C.c_uint32]),
        "orc_sim_adapter_dom": (u8p, [C.c_void_p, C.c_int, C.c_uint32]),
        "orc_sim_adapter_rate": (u8p, [C.c_void_p, C.c_int, C.c_uint32]),
        "orc_sim_new_variants": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_double, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p,
                                              C.c_size_t]),
        "orc_var_last_error": (C.c_char_p, []),
        "orc_var_sys_errors": (C.c_uint32, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]),
        "orc_sieve_blocks_var": (C.c_uint64, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
        "orc_create_reads_var": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
        "free": (None, [C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


# ------------------------------------------------------------------ high level
class Profile:
    def __init__(self, path):
        self.h = lib().orc_profile_load(str(path).encode())
        if not self.h:
            raise RuntimeError(f"oracle could not load {path}")

    def table(self, family, a=0, b=0, c=0, d=0):
        fam = {"quality": 0, "seq_quality": 1, "base_call": 2, "dom_error": 3, "error_rate": 4, "indels": 5}[family]
        return lib().orc_profile_table(self.h, fam, a, b, c, d)

    def close(self):
        if self.h:
            lib().orc_profile_free(self.h)
            self.h = None


class Reference:
    def __init__(self, seqs):
        L = lib()
        self.seqs = [(n, np.ascontiguousarray(c, dtype=np.uint8)) for n, c in seqs]
        self.h = L.orc_reference_new(len(self.seqs))
        for i, (name, codes) in enumerate(self.seqs):
            L.orc_reference_set(self.h, i, name.encode(), _ptr(codes, u8p), len(codes))

    def close(self):
        if self.h:
            lib().orc_reference_free(self.h)
            self.h = None


def _take_text(t):
    out = C.string_at(t.data, t.len) if t.len else b""
    lib().orc_text_free(C.byref(t))
    return out


def variant_list(vs, seq):
    """[(position, letters, allele bits as int)] of one sequence of an OrcVariants"""
    out = []
    for i in range(vs.contents.n[seq]):
        v = vs.contents.v[seq][i]
        out.append((v.position, "".join("ACGT"[v.var_seq[k]] for k in range(v.len)), v.allele[0] | (v.allele[1] << 64)))
    return out


def create_sys_error_profile(profile, reference, seed):
    """FASTQ text of Simulator::CreateSystematicErrorProfile"""
    t = Text()
    if lib().orc_create_sys_error_profile(profile.h, reference.h, seed, C.byref(t)):
        raise RuntimeError("orc_create_sys_error_profile failed")
    return _take_text(t)


class Sim:
    def __init__(self, profile, reference, seed, num_pairs=0, coverage=0.0, base_identifier=b"", ref_bias_mode=0, ref_bias_file=None, variants=None):
        """variants: an OrcVariants pointer (orc_read_variants); it has to outlive the Sim"""
        self.profile, self.reference, self.variants = profile, reference, variants
        err = C.create_string_buffer(1024)
        self.h = lib().orc_sim_new_variants(profile.h, reference.h if reference else None, C.cast(variants, C.c_void_p) if variants else None, seed, num_pairs, coverage,
                                            base_identifier, ref_bias_mode, str(ref_bias_file).encode() if ref_bias_file else None, err, len(err))
        if not self.h:
            raise RuntimeError(err.value.decode())

    def ref_seq_bias(self):
        return np.ctypeslib.as_array(lib().orc_sim_ref_seq_bias(self.h), shape=(len(self.reference.seqs),)).copy()

    def read_methylation(self, path):
        err = C.create_string_buffer(2048)
        if lib().orc_sim_read_methylation(self.h, str(path).encode(), err, len(err)):
            raise RuntimeError(err.value.decode())

    def load_sys_errors(self, text):
        err = C.create_string_buffer(1024)
        if lib().orc_sim_load_sys_errors(self.h, text, len(text), err, len(err)):
            raise RuntimeError(err.value.decode())

    # pre-pass results
    def thresholds(self):
        L = lib()
        n = L.orc_sim_n_groups(self.h) * L.orc_sim_insert_to(self.h) * 2
        return np.ctypeslib.as_array(L.orc_sim_thresholds(self.h), shape=(n,)).reshape(L.orc_sim_n_groups(self.h), -1, 2).copy()

    def norm_by_len(self):
        L = lib()
        return np.ctypeslib.as_array(L.orc_sim_norm_by_len(self.h), shape=(L.orc_sim_insert_to(self.h),)).copy()

    def gap_passes(self, group, starts, c1=0):
        """the sieve's gap draws of coverage group `group` at the given start positions: (start index, length, probability_chosen) of
        every cell that passes the zero threshold, plus the table (q, seg_end)"""
        L = lib()
        to = L.orc_sim_insert_to(self.h)
        q, seg_end = np.zeros(to), np.zeros(to, np.uint32)
        L.orc_gap_table(self.h, group, _ptr(q, f64p), _ptr(seg_end, u32p))
        thr = np.ascontiguousarray(self.thresholds()[group].reshape(-1))
        hit = np.dtype([("len", np.uint32), ("pad", np.uint32), ("probability_chosen", np.float64)])
        buf = np.zeros(to, hit)
        out = []
        for i, start in enumerate(starts):
            n = L.orc_gap_hits(self.h, _ptr(q, f64p), _ptr(seg_end, u32p), _ptr(thr, f64p), int(start), c1, buf.ctypes.data)
            for k in range(n):
                out.append((i, int(buf["len"][k]), float(buf["probability_chosen"][k])))
        return out, q, seg_end

    def literal_passes(self, group, starts, c1=0):
        """the same cells by the reference's own loop, one uniform per (start, length) (orc_literal_hits, Simulator.cpp:2302-2306): (start index, length, probability_chosen)"""
        L = lib()
        to = L.orc_sim_insert_to(self.h)
        thr = np.ascontiguousarray(self.thresholds()[group].reshape(-1))
        hit = np.dtype([("len", np.uint32), ("pad", np.uint32), ("probability_chosen", np.float64)])
        buf = np.zeros(to, hit)
        out = []
        for i, start in enumerate(starts):
            n = L.orc_literal_hits(self.h, _ptr(thr, f64p), int(start), c1, buf.ctypes.data)
            out += [(i, int(buf["len"][k]), float(buf["probability_chosen"][k])) for k in range(n)]
        return out

    def bias_normalization(self):
        return lib().orc_sim_bias_normalization(self.h)

    def set_normalization(self, bias_normalization, thresholds):
        t = np.ascontiguousarray(thresholds, dtype=np.float64)
        lib().orc_sim_set_normalization(self.h, bias_normalization, _ptr(t, f64p))

    def sys_errors(self, strand, seq):
        L = lib()
        n = len(self.reference.seqs[seq][1])
        dom = np.ctypeslib.as_array(L.orc_sim_sys_dom(self.h, strand, seq), shape=(n,)).copy()
        rate = np.ctypeslib.as_array(L.orc_sim_sys_rate(self.h, strand, seq), shape=(n,)).copy()
        return dom, rate

    def adapter_sys_errors(self, seg, adapter, n):
        L = lib()
        p = L.orc_sim_adapter_dom(self.h, seg, adapter)
        if not p:
            return None
        return (np.ctypeslib.as_array(p, shape=(n,)).copy(),
                np.ctypeslib.as_array(L.orc_sim_adapter_rate(self.h, seg, adapter), shape=(n,)).copy())

    def total_pairs(self):
        return lib().orc_sim_total_pairs(self.h)

    def adapter_only_pairs(self):
        return lib().orc_sim_adapter_only_pairs(self.h)

    def total_blocks(self):
        return lib().orc_sim_total_blocks(self.h)

    def sieve(self, block_lo, block_hi):
        L = lib()
        out = C.POINTER(Fragment)()
        n = L.orc_sieve_blocks(self.h, block_lo, block_hi, C.byref(out))
        arr = np.frombuffer(C.string_at(out, n * C.sizeof(Fragment)), dtype=FRAGMENT_DTYPE).copy() if n else np.zeros(0, FRAGMENT_DTYPE)
        L.free(out)
        return arr

    def sieve_literal(self, block_lo, block_hi):
        """orc_sieve_blocks_literal: the sieve with the reference's per-cell loop (a random stream of its own)"""
        L = lib()
        out = C.POINTER(Fragment)()
        n = L.orc_sieve_blocks_literal(self.h, block_lo, block_hi, C.byref(out))
        arr = np.frombuffer(C.string_at(out, n * C.sizeof(Fragment)), dtype=FRAGMENT_DTYPE).copy() if n else np.zeros(0, FRAGMENT_DTYPE)
        L.free(out)
        return arr

    def create_reads(self, frags):
        L = lib()
        frags = np.ascontiguousarray(frags, dtype=FRAGMENT_DTYPE)
        r1, r2 = Text(), Text()
        L.orc_create_reads(self.h, frags.ctypes.data_as(C.POINTER(Fragment)), len(frags), C.byref(r1), C.byref(r2))
        return _take_text(r1), _take_text(r2)

    # with variants
    def sieve_var(self, block_lo, block_hi):
        L = lib()
        out = C.c_void_p()
        n = L.orc_sieve_blocks_var(self.h, block_lo, block_hi, C.byref(out))
        if n == 2**64 - 1:
            raise RuntimeError(L.orc_var_last_error().decode())
        arr = np.frombuffer(C.string_at(out, n * FRAGMENT_VAR_DTYPE.itemsize), dtype=FRAGMENT_VAR_DTYPE).copy() if n else np.zeros(0, FRAGMENT_VAR_DTYPE)
        L.free(out)
        return arr

    def create_reads_var(self, frags):
        L = lib()
        frags = np.ascontiguousarray(frags, dtype=FRAGMENT_VAR_DTYPE)
        r1, r2 = Text(), Text()
        if L.orc_create_reads_var(self.h, frags.ctypes.data, len(frags), C.byref(r1), C.byref(r2)):
            raise RuntimeError(L.orc_var_last_error().decode())
        return _take_text(r1), _take_text(r2)

    def var_sys_errors(self, strand, seq, var_id):
        dom, rate = np.zeros(4096, np.uint8), np.zeros(4096, np.uint8)
        n = lib().orc_var_sys_errors(self.h, strand, seq, var_id, dom.ctypes.data, rate.ctypes.data, 4096)
        return dom[:n].copy(), rate[:n].copy()

    def adapter_only(self):
        r1, r2 = Text(), Text()
        lib().orc_simulate_adapter_only_pairs(self.h, C.byref(r1), C.byref(r2))
        return _take_text(r1), _take_text(r2)

    def close(self):
        if self.h:
            lib().orc_sim_free(self.h)
            self.h = None


def error_model_only(profile, seed, rec, first_index=0):
    """rec: dict from synth.make_error_model_input.  Returns list of (seq codes, qual bytes, cigar, nerr, tile)."""
    L = lib()
    n, rl = rec["seqs"].shape
    out = (Read * n)()
    tiles = np.zeros(n, np.uint16)
    seqs = np.ascontiguousarray(rec["seqs"], np.uint8)
    seg = np.ascontiguousarray(rec["seg"], np.uint8)
    fl = np.ascontiguousarray(rec["frag_len"], np.uint32)
    dom = np.ascontiguousarray(rec["dom"], np.uint8)
    rate = np.ascontiguousarray(rec["rate"], np.uint8)
    L.orc_error_model_only(profile.h, seed, first_index, n, rl, _ptr(seqs, u8p), _ptr(seg, u8p), _ptr(fl, u32p), _ptr(dom, u8p),
                           _ptr(rate, u8p), out, _ptr(tiles, u16p))
    res = []
    for i in range(n):
        r = out[i]
        res.append((bytes(r.seq[:r.read_len]), bytes(r.qual[:r.read_len]), r.cigar.decode(), int(r.num_errors), int(tiles[i])))
    return res
