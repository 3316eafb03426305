"""ctypes binding of libreseq_amd.so (C ABI: include/reseq_amd.h).

There is no CPU fallback: importing this module without the built library, or
creating a Simulator without a visible MI355X, raises.  The class names follow
the reference's objects for this path (reseq::DataStats/ProbabilityEstimates ->
Profile, reseq::Reference -> Reference, reseq::Simulator -> Simulator).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libreseq_amd.so")

RSQ_OK, RSQ_EINVAL, RSQ_EIO, RSQ_ENODEV, RSQ_EHIP, RSQ_ENOSPC, RSQ_ESTATE = 0, -1, -2, -3, -4, -5, -6

FRAGMENT_DTYPE = np.dtype([("seq", "<u4"), ("start", "<u4"), ("len", "<u4"), ("dup", "<u2"), ("strand", "u1"),
                           ("allele", "u1"), ("block", "<u4"), ("number", "<u4")])


class SimInfo(C.Structure):
    _fields_ = [("total_pairs", C.c_uint64), ("adapter_only_pairs", C.c_uint64), ("total_blocks", C.c_uint32),
                ("n_coverage_groups", C.c_uint32), ("insert_to", C.c_uint32), ("sys_chain_passes", C.c_uint32),
                ("bias_normalization", C.c_double)]


class RsqError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rsq error {code}: {message}")
        self.code = code


# every symbol include/reseq_amd.h declares: name -> (restype, argtypes)
_vp, _u64, _u32, _sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t
_pp = C.POINTER(C.c_void_p)
_psz = C.POINTER(C.c_size_t)
SYMBOLS = {
    "rsq_last_error": (C.c_char_p, []),
    "rsq_version": (C.c_char_p, []),
    "rsq_set_option": (C.c_int, [C.c_char_p, C.c_int64]),
    "rsq_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
    "rsq_device_count": (C.c_int, []),
    "rsq_profile_load": (C.c_int, [C.c_char_p, _pp]),
    "rsq_profile_load_reseq": (C.c_int, [C.c_char_p, C.c_char_p, C.c_double, _pp]),
    "rsq_profile_is_reseq_archive": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "rsq_profile_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rsq_profile_save_reseq": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint64]),
    "rsq_profile_archive_layout": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rsq_last_warning": (C.c_char_p, []),
    "rsq_profile_free": (None, [_vp]),
    "rsq_profile_change_error_rate": (C.c_int, [_vp, C.c_double]),
    "rsq_profile_remove_substitution_errors": (C.c_int, [_vp]),
    "rsq_profile_remove_indel_errors": (C.c_int, [_vp]),
    "rsq_profile_max_read_length": (C.c_int, [_vp, C.POINTER(_u32)]),
    "rsq_profile_num_tiles": (C.c_int, [_vp, C.POINTER(_u32)]),
    "rsq_profile_max_len_deletion": (C.c_int, [_vp, C.POINTER(_u32)]),
    "rsq_profile_ref_seq_bias": (C.c_int, [_vp, _vp, _sz, _psz]),
    "rsq_ref_sequence_name": (C.c_int, [_vp, _u32, C.c_char_p, _sz]),
    "rsq_ref_load_fasta": (C.c_int, [C.c_char_p, _pp]),
    "rsq_ref_replace_n": (C.c_int, [_vp, _u64]),
    "rsq_ref_free": (None, [_vp]),
    "rsq_ref_num_sequences": (C.c_int, [_vp, C.POINTER(_u32)]),
    "rsq_ref_sequence_length": (C.c_int, [_vp, _u32, C.POINTER(_u32)]),
    "rsq_ref_write_fasta": (C.c_int, [_vp, C.c_char_p]),
    "rsq_ref_read_variants": (C.c_int, [_vp, C.c_char_p]),
    "rsq_ref_num_alleles": (C.c_int, [_vp, C.POINTER(_u32)]),
    "rsq_ref_num_variants": (C.c_int, [_vp, _u32, C.POINTER(_u32)]),
    "rsq_ref_get_variant": (C.c_int, [_vp, _u32, _u32, C.POINTER(_u32), C.c_char_p, _sz, C.POINTER(_u64)]),
    "rsq_ref_get_codes": (C.c_int, [_vp, _u32, _vp, _u32]),
    "rsq_sim_create": (C.c_int, [_vp, _vp, C.c_int, _pp]),
    "rsq_sim_free": (None, [_vp]),
    "rsq_sim_take_options": (C.c_int, [_vp]),
    "rsq_sim_prepare": (C.c_int, [_vp, _u64, _u64, C.c_double, C.c_int, C.c_char_p, _vp]),
    "rsq_sim_prepare_plan": (C.c_int, [_vp, _u64, _u64, C.c_double, C.c_int, C.c_char_p]),
    "rsq_sim_bias_partials": (C.c_int, [_vp, _u32, _u32, _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t), _vp]),
    "rsq_sim_prepare_normalization": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "rsq_sim_prepare_sys_errors": (C.c_int, [_vp, _u32, _u32, _vp, _vp, _vp]),
    "rsq_sim_prepare_finish": (C.c_int, [_vp]),
    "rsq_sim_get_info": (C.c_int, [_vp, C.POINTER(SimInfo)]),
    "rsq_sim_get_fill_plan": (C.c_int, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "rsq_sim_specialize": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    "rsq_sim_export_reference": (C.c_int, [_vp, C.c_char_p]),
    "rsq_sim_block_weights": (C.c_int, [_vp, _vp, _sz, C.POINTER(_u32)]),
    "rsq_partition_blocks": (C.c_int, [_u32, _u32, _vp, _vp]),
    "rsq_sim_reference_sequence": (C.c_int, [_vp, _u32, _u32, _u32, C.c_int, C.c_int32, _u32, _u32, _vp, _sz]),
    "rsq_sim_job_read": (C.c_int, [_vp, C.c_int, _u64, _sz, _vp, _vp]),
    "rsq_dev_pwrite": (C.c_int, [C.c_int, _vp, _sz, C.c_char_p, _u64]),
    "rsq_sim_get_sequence_lengths": (C.c_int, [_vp, _vp, _sz, C.POINTER(_u32)]),
    "rsq_sim_import_reference": (C.c_int, [_vp, C.c_char_p]),
    "rsq_set_kernel_cache_dir": (C.c_int, [C.c_char_p]),
    "rsq_profile_compile_read_kernel": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_double)]),
    "rsq_sim_get_thresholds": (C.c_int, [_vp, _vp, _sz]),
    "rsq_sim_get_norm_by_len": (C.c_int, [_vp, _vp, _sz]),
    "rsq_sim_set_normalization": (C.c_int, [_vp, C.c_double, _vp, _sz]),
    "rsq_sim_get_sys_errors": (C.c_int, [_vp, C.c_int, _u32, _vp, _vp, _u32]),
    "rsq_sim_get_adapter_sys_errors": (C.c_int, [_vp, C.c_int, _u32, _vp, _vp, _u32]),
    "rsq_sim_create_sys_error_profile": (C.c_int, [_vp, _u64, C.c_char_p, _vp]),
    "rsq_sim_read_sys_errors": (C.c_int, [_vp, C.c_char_p]),
    "rsq_sim_set_ref_bias_file": (C.c_int, [_vp, C.c_char_p]),
    "rsq_sim_read_methylation": (C.c_int, [_vp, C.c_char_p]),
    "rsq_sim_get_ref_seq_bias": (C.c_int, [_vp, _vp, _sz]),
    "rsq_sim_pairs": (C.c_int, [_vp, _u32, _u32, _vp, _sz, _psz, _vp, _sz, _psz, C.POINTER(_u64), _vp, _sz, _vp]),
    "rsq_sim_job_generate": (C.c_int, [_vp, _u32, _u32, _u32, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64), _vp]),
    "rsq_sim_job_write": (C.c_int, [_vp, C.c_char_p, _u64, C.c_char_p, _u64, _u32]),
    "rsq_sim_job_compress": (C.c_int, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "rsq_gzip_bound": (C.c_size_t, [C.c_size_t]),
    "rsq_gzip_eof_member": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "rsq_sim_gzip_keep_code": (C.c_int, [_vp, C.c_int]),
    "rsq_sim_gzip_device": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_size_t), _vp]),
    "rsq_sim_job_free": (C.c_int, [_vp]),
    "rsq_sim_adapter_only_pairs": (C.c_int, [_vp, _u64, _u64, _vp, _sz, _psz, _vp, _sz, _psz, _vp]),
    "rsq_sim_error_model": (C.c_int, [_vp, _u64, _u64, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _u32, _vp]),
    "rsq_sim_error_model_fastq": (C.c_int, [_vp, _u64, _u64, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t), _vp]),
    "rsq_sim_error_model_file": (C.c_int, [_vp, C.c_char_p, C.c_char_p, _vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "rsq_fasta_count_records": (C.c_int, [C.c_char_p, _u64, _u64, _u32, C.POINTER(_u64), C.POINTER(_u64)]),
    "rsq_sim_error_model_fasta": (C.c_int, [_vp, _u64, _vp, C.c_size_t, C.c_int, _vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(_u64), C.POINTER(C.c_size_t), _vp]),
    "rsq_sim_last_kernel_ms": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_double)]),
    "rsq_sim_last_kernel_launches": (C.c_int, [_vp, C.c_char_p, C.POINTER(_u32)]),
    "rsq_dev_alloc": (C.c_int, [C.c_int, _sz, _pp]),
    "rsq_dev_free": (C.c_int, [C.c_int, _vp]),
    "rsq_dev_upload": (C.c_int, [C.c_int, _vp, _vp, _sz]),
    "rsq_dev_download": (C.c_int, [C.c_int, _vp, _vp, _sz]),
    "rsq_host_alloc": (C.c_int, [_sz, _pp]),
    "rsq_host_free": (C.c_int, [_vp]),
    "rsq_stream_create": (C.c_int, [C.c_int, _pp]),
    "rsq_stream_destroy": (C.c_int, [C.c_int, _vp]),
    "rsq_dev_copy_on": (C.c_int, [C.c_int, _vp, _vp, _sz, C.c_int, _vp]),
}

_lib = None


def lib():
    """Load libreseq_amd.so; raise if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C reseq_amd/csrc); reseq_amd has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def use_library(path):
    """Bind another build of the same library (experiment builds under exp/); must be called before anything else."""
    global LIB_PATH, _lib
    if _lib is not None:
        raise RuntimeError("the library is already loaded")
    LIB_PATH = os.path.abspath(path)


def set_option(name, value):
    """Testing / measurement switches of the library (include/reseq_amd.h rsq_set_option); results never depend on them."""
    _check(lib().rsq_set_option(name.encode(), int(value)))


def get_option(name):
    v = C.c_int64(0)
    _check(lib().rsq_get_option(name.encode(), C.byref(v)))
    return v.value


def archive_layout(stats_path, ipf_path=None):
    """where the class information of every serialized type sits in a .reseq / .reseq.ipf pair, and the parse error if there is one"""
    need = C.c_size_t(0)
    ipf = ipf_path.encode() if ipf_path else None
    _check(lib().rsq_profile_archive_layout(str(stats_path).encode(), ipf, None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    _check(lib().rsq_profile_archive_layout(str(stats_path).encode(), ipf, buf, need.value, C.byref(need)))
    return buf.value.decode(errors="replace")


def gzip_eof_member():
    """the 28 bytes that end a BGZF file (rsq_gzip_eof_member)"""
    buf = C.create_string_buffer(32)
    n = lib().rsq_gzip_eof_member(buf, 32)
    return buf.raw[:n]


def partition_blocks(total_blocks, workers, weights=None):
    """rsq_partition_blocks: [(lo, hi)] per worker -- the library's statement of sharding.partition_blocks (what `reseq illuminaPE --gpus N` uses)"""
    w = np.ascontiguousarray(np.ones(total_blocks) if weights is None else weights, dtype=np.float64)
    if w.size != total_blocks:
        raise ValueError("one weight per block")
    bounds = np.zeros(workers + 1, np.uint32)
    _check(lib().rsq_partition_blocks(total_blocks, workers, w.ctypes.data, bounds.ctypes.data))
    return [(int(bounds[r]), int(bounds[r + 1])) for r in range(workers)]


def dev_pwrite(device, src_ptr, nbytes, path, offset):
    """device memory at `src_ptr` to byte `offset` of a file (page-locked double buffers)"""
    _check(lib().rsq_dev_pwrite(device, C.c_void_p(src_ptr), nbytes, os.fsencode(path), offset))


def set_kernel_cache_dir(path):
    """where the read kernels compiled for a profile are kept between processes (None / "": nowhere); default ~/.cache/reseq_amd"""
    _check(lib().rsq_set_kernel_cache_dir((path or "").encode()))


def last_warning():
    return lib().rsq_last_warning().decode(errors="replace")


def _check(rc):
    if rc != RSQ_OK:
        raise RsqError(rc, lib().rsq_last_error().decode(errors="replace"))


def device_count():
    n = lib().rsq_device_count()
    if n < 0:
        raise RsqError(n, lib().rsq_last_error().decode())
    return n


class Profile:
    """DataStats::Load + ProbabilityEstimates::Load/PrepareResult: an RSQP container, or ReSeq's own `.reseq` archive
    (then `ipf_path` names the `.reseq.ipf` archive, default `<path>.ipf`)."""

    def __init__(self, path, ipf_path=None, ipf_precision=5.0):
        self.h = C.c_void_p()
        if ipf_path is None and ipf_precision == 5.0:
            _check(lib().rsq_profile_load(os.fsencode(path), C.byref(self.h)))
        else:
            _check(lib().rsq_profile_load_reseq(os.fsencode(path), os.fsencode(ipf_path) if ipf_path else None, ipf_precision, C.byref(self.h)))
        self.warning = lib().rsq_last_warning().decode()

    def save(self, path):
        """the prepared profile as an RSQP container"""
        _check(lib().rsq_profile_save(self.h, os.fsencode(path)))

    def save_reseq(self, stats_path, ipf_path=None, creation_time=0):
        """as ReSeq's own pair of files (`stats_path` and `<stats_path>.ipf`): rsq_profile_save_reseq"""
        _check(lib().rsq_profile_save_reseq(self.h, os.fsencode(stats_path), os.fsencode(ipf_path) if ipf_path else None, int(creation_time)))

    def compile_read_kernel(self, kind=0, with_variants=False, binned=False, arch="gfx950", out_path=None):
        """host only: the read kernel compiled for this profile (what Simulator.specialize does on the device); returns (bytes of the code object, seconds)"""
        n, t = C.c_size_t(0), C.c_double(0.0)
        _check(lib().rsq_profile_compile_read_kernel(self.h, kind, int(with_variants), int(binned), arch.encode(), os.fsencode(out_path) if out_path else None, C.byref(n), C.byref(t)))
        return n.value, t.value

    def change_error_rate(self, multiplier):
        _check(lib().rsq_profile_change_error_rate(self.h, multiplier))

    def remove_substitution_errors(self):
        _check(lib().rsq_profile_remove_substitution_errors(self.h))

    def remove_indel_errors(self):
        _check(lib().rsq_profile_remove_indel_errors(self.h))

    def max_read_length(self):
        v = C.c_uint32()
        _check(lib().rsq_profile_max_read_length(self.h, C.byref(v)))
        return v.value

    def close(self):
        if self.h:
            lib().rsq_profile_free(self.h)
            self.h = C.c_void_p()


def load_profile(path, ipf_path=None, ipf_precision=5.0):
    """a profile by the file's content: ReSeq's `.reseq` archive (with its `.reseq.ipf`: ipf_path, default `<path>.ipf`) or an RSQP container -- what the command line does"""
    archive = C.c_int(0)
    lib().rsq_profile_is_reseq_archive(os.fsencode(path), C.byref(archive))
    if not archive.value:
        return Profile(path)
    p = Profile.__new__(Profile)
    p.h = C.c_void_p()
    _check(lib().rsq_profile_load_reseq(os.fsencode(path), os.fsencode(ipf_path) if ipf_path else None, float(ipf_precision), C.byref(p.h)))
    p.warning = lib().rsq_last_warning().decode()
    return p


class Reference:
    """Reference::ReadFasta (+ ReplaceN)."""

    def __init__(self, fasta_path, replace_n_seed=None):
        self.h = C.c_void_p()
        _check(lib().rsq_ref_load_fasta(os.fsencode(fasta_path), C.byref(self.h)))
        if replace_n_seed is not None:
            _check(lib().rsq_ref_replace_n(self.h, replace_n_seed))

    def num_sequences(self):
        v = C.c_uint32()
        _check(lib().rsq_ref_num_sequences(self.h, C.byref(v)))
        return v.value

    def sequence_length(self, i):
        v = C.c_uint32()
        _check(lib().rsq_ref_sequence_length(self.h, i, C.byref(v)))
        return v.value

    def sequence_name(self, i):
        buf = C.create_string_buffer(1 << 16)
        _check(lib().rsq_ref_sequence_name(self.h, i, buf, len(buf)))
        return buf.value.decode()

    def read_variants(self, path):
        """Reference::PrepareVariantFile + ReadFirstVariants: returns the number of alleles"""
        _check(lib().rsq_ref_read_variants(self.h, str(path).encode()))
        n = _u32()
        _check(lib().rsq_ref_num_alleles(self.h, C.byref(n)))
        return n.value

    def variants(self, seq):
        """[(position, bases, allele bits as int)] of one sequence"""
        n = _u32()
        _check(lib().rsq_ref_num_variants(self.h, seq, C.byref(n)))
        out = []
        for i in range(n.value):
            pos, buf, bits = _u32(), C.create_string_buffer(1 << 16), (_u64 * 2)()
            _check(lib().rsq_ref_get_variant(self.h, seq, i, C.byref(pos), buf, len(buf), bits))
            out.append((pos.value, buf.value.decode(), bits[0] | (bits[1] << 64)))
        return out

    def write_fasta(self, path):
        _check(lib().rsq_ref_write_fasta(self.h, str(path).encode()))

    def codes(self, i):
        out = np.zeros(self.sequence_length(i), np.uint8)
        _check(lib().rsq_ref_get_codes(self.h, i, out.ctypes.data, len(out)))
        return out

    def close(self):
        if self.h:
            lib().rsq_ref_free(self.h)
            self.h = C.c_void_p()


class ErrorModelFileOptions(C.Structure):
    """rsq_error_model_file_options (include/reseq_amd.h)"""
    _fields_ = [("read_threads", C.c_uint32), ("block_kb", C.c_uint32), ("batch_blocks", C.c_uint32), ("keep_text", C.c_uint32), ("from_", C.c_uint64), ("to", C.c_uint64), ("first_record", C.c_uint64),
                ("progress", C.c_void_p), ("user", C.c_void_p), ("trace", C.c_char_p), ("trace_cap", C.c_size_t)]


def count_fasta_records(path, from_=0, to=0, threads=0):
    """record starts in bytes [from_, to) of a plain FASTA file (to = 0: its end): (their number, the first one's offset -- `to` if none).  Host code, no device."""
    n, first = _u64(0), _u64(0)
    _check(lib().rsq_fasta_count_records(os.fsencode(path), from_, to, threads, C.byref(n), C.byref(first)))
    return n.value, first.value


class DeviceArray:
    """A device allocation made through the C ABI (tests and the bench stage their buffers with it)."""

    def __init__(self, device, nbytes):
        self.device, self.nbytes = device, int(nbytes)
        self.ptr = C.c_void_p()
        _check(lib().rsq_dev_alloc(device, self.nbytes, C.byref(self.ptr)))

    @classmethod
    def from_numpy(cls, device, a):
        a = np.ascontiguousarray(a)
        d = cls(device, max(a.nbytes, 8))
        _check(lib().rsq_dev_upload(device, d.ptr, a.ctypes.data, a.nbytes))
        return d

    def to_numpy(self, dtype, count):
        out = np.zeros(count, dtype)
        if out.nbytes:
            _check(lib().rsq_dev_download(self.device, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().rsq_dev_free(self.device, self.ptr)
            self.ptr = C.c_void_p()


class Simulator:
    """reseq::Simulator bound to one GPU (Simulator.h:453-459)."""

    def __init__(self, profile, reference=None, device=0):
        self.device = device
        self.h = C.c_void_p()
        _check(lib().rsq_sim_create(profile.h, reference.h if reference is not None else None, device, C.byref(self.h)))

    def prepare(self, seed, num_read_pairs=0, coverage=0.0, ref_bias_mode=0, record_base_identifier="", stream=None):
        _check(lib().rsq_sim_prepare(self.h, seed, num_read_pairs, coverage, ref_bias_mode, record_base_identifier.encode(), stream))
        return self.info()

    # the pre-pass of one rank of a sharded job (include/reseq_amd.h "The same pre-pass for ONE rank"); reseq_amd/sharding.py drives these
    def prepare_plan(self, seed, num_read_pairs=0, coverage=0.0, ref_bias_mode=0, record_base_identifier=""):
        _check(lib().rsq_sim_prepare_plan(self.h, seed, num_read_pairs, coverage, ref_bias_mode, record_base_identifier.encode()))
        return self.info()

    def bias_partials(self, block_lo, block_hi, stream=None):
        n = C.c_size_t(0)
        _check(lib().rsq_sim_bias_partials(self.h, 0, 0, None, None, 0, C.byref(n), None))
        sums, maxes = np.zeros(n.value), np.zeros(n.value)
        _check(lib().rsq_sim_bias_partials(self.h, block_lo, block_hi, sums.ctypes.data, maxes.ctypes.data, n.value, C.byref(n), stream))
        return sums, maxes

    def prepare_normalization(self, sums, maxes):
        sums, maxes = np.ascontiguousarray(sums, np.float64), np.ascontiguousarray(maxes, np.float64)
        _check(lib().rsq_sim_prepare_normalization(self.h, sums.ctypes.data, maxes.ctypes.data, sums.size))

    def prepare_sys_errors(self, block_lo, block_hi, in_state, stream=None):
        i, o = np.asarray(in_state, np.uint32), np.zeros(2, np.uint32)
        _check(lib().rsq_sim_prepare_sys_errors(self.h, block_lo, block_hi, i.ctypes.data, o.ctypes.data, stream))
        return [int(o[0]), int(o[1])]

    def prepare_finish(self):
        _check(lib().rsq_sim_prepare_finish(self.h))
        return self.info()

    def info(self):
        i = SimInfo()
        _check(lib().rsq_sim_get_info(self.h, C.byref(i)))
        return i

    def fill_plan(self):
        """{'mask': quads per quality row of the screened draws (0: double precision only), 'image_tiles', 'image_bytes'}"""
        q, t, b = _u32(0), _u32(0), _u32(0)
        _check(lib().rsq_sim_get_fill_plan(self.h, C.byref(q), C.byref(t), C.byref(b)))
        return {"mask": q.value, "image_tiles": t.value, "image_bytes": b.value}

    def sequence_lengths(self):
        n = _u32(0)
        _check(lib().rsq_sim_get_sequence_lengths(self.h, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.uint32)
        _check(lib().rsq_sim_get_sequence_lengths(self.h, out.ctypes.data, out.size, C.byref(n)))
        return [int(x) for x in out]

    def take_options(self):
        """the option switches as they stand now (a simulator otherwise keeps the copy it took when it was created)"""
        _check(lib().rsq_sim_take_options(self.h))

    def block_weights(self):
        n = _u32(0)
        _check(lib().rsq_sim_block_weights(self.h, None, 0, C.byref(n)))
        w = np.zeros(n.value, np.float64)
        _check(lib().rsq_sim_block_weights(self.h, w.ctypes.data, w.size, C.byref(n)))
        return w

    def reference_sequence(self, seq, start_pos, frag_length, reversed=False, first_variant=(0, 0), allele=0):
        """Reference::ReferenceSequence (with variants: of one allele, from inside inserted bases when first_variant[1] > 0) as letters"""
        out = np.zeros(max(1, frag_length), np.uint8)
        _check(lib().rsq_sim_reference_sequence(self.h, seq, start_pos, frag_length, int(bool(reversed)), int(first_variant[0]), int(first_variant[1]), allele, out.ctypes.data, out.size))
        return "".join("ACGT"[c] for c in out[:frag_length])

    def export_reference(self, path):
        """what the simulator keeps of its reference, variant and methylation files, for the other ranks of the host (import_reference)"""
        _check(lib().rsq_sim_export_reference(self.h, os.fsencode(path)))

    def import_reference(self, path):
        """the state another process exported; this simulator was created with reference None"""
        _check(lib().rsq_sim_import_reference(self.h, os.fsencode(path)))

    def specialize(self, kind=0):
        """Compiles the read kernel (kind 0: read pairs, 1: seqToIllumina records) for this simulator's profile now.  Returns (specialized, what the library says):
        False when its own instantiation runs instead (option specialize 0, no libhiprtc, no table image, a failed compilation)."""
        done = C.c_int(0)
        _check(lib().rsq_sim_specialize(self.h, kind, C.byref(done)))
        return bool(done.value), last_warning()

    def thresholds(self):
        i = self.info()
        out = np.zeros((i.n_coverage_groups, i.insert_to, 2), np.float64)
        _check(lib().rsq_sim_get_thresholds(self.h, out.ctypes.data, out.size))
        return out

    def norm_by_len(self):
        out = np.zeros(self.info().insert_to, np.float64)
        _check(lib().rsq_sim_get_norm_by_len(self.h, out.ctypes.data, out.size))
        return out

    def set_normalization(self, bias_normalization, thresholds):
        t = np.ascontiguousarray(thresholds, np.float64)
        _check(lib().rsq_sim_set_normalization(self.h, bias_normalization, t.ctypes.data, t.size))

    def sys_errors(self, reverse, seq, length):
        dom, rate = np.zeros(length, np.uint8), np.zeros(length, np.uint8)
        _check(lib().rsq_sim_get_sys_errors(self.h, int(reverse), seq, dom.ctypes.data, rate.ctypes.data, length))
        return dom, rate

    def adapter_sys_errors(self, seg, adapter, length):
        dom, rate = np.zeros(length, np.uint8), np.zeros(length, np.uint8)
        _check(lib().rsq_sim_get_adapter_sys_errors(self.h, seg, adapter, dom.ctypes.data, rate.ctypes.data, length))
        return dom, rate

    def create_sys_error_profile(self, seed, path, stream=None):
        """Simulator::CreateSystematicErrorProfile (--writeSysError)"""
        _check(lib().rsq_sim_create_sys_error_profile(self.h, seed, str(path).encode(), stream))

    def read_sys_errors(self, path):
        """--readSysError: after prepare()"""
        _check(lib().rsq_sim_read_sys_errors(self.h, str(path).encode()))

    def read_methylation(self, path):
        """--methylation: extended BED, bisulfite conversion of the templates (Simulator::CTConversion)"""
        _check(lib().rsq_sim_read_methylation(self.h, str(path).encode()))

    def set_ref_bias_file(self, path):
        _check(lib().rsq_sim_set_ref_bias_file(self.h, str(path).encode()))

    def ref_seq_bias(self, n_sequences):
        out = np.zeros(n_sequences, np.float64)
        _check(lib().rsq_sim_get_ref_seq_bias(self.h, out.ctypes.data, out.size))
        return out

    def pairs_device(self, block_lo, block_hi, r1, r2, frags=None, stream=None):
        """Run the hot path into caller-owned DeviceArrays.  Returns (n_pairs, len1, len2, rc)."""
        l1, l2, n = C.c_size_t(), C.c_size_t(), C.c_uint64()
        rc = lib().rsq_sim_pairs(self.h, block_lo, block_hi, r1.ptr if r1 else None, r1.nbytes if r1 else 0, C.byref(l1), r2.ptr if r2 else None,
                                 r2.nbytes if r2 else 0, C.byref(l2), C.byref(n), frags.ptr if frags else None,
                                 frags.nbytes // FRAGMENT_DTYPE.itemsize if frags else 0, stream)
        return n.value, l1.value, l2.value, rc

    def pairs(self, block_lo, block_hi, stream=None):
        """Convenience for tests: sizes the buffers with a first call, returns (fragments, fastq1 bytes, fastq2 bytes)."""
        n, l1, l2, rc = self.pairs_device(block_lo, block_hi, None, None, None, stream)
        if rc == RSQ_OK and n == 0:
            return np.zeros(0, FRAGMENT_DTYPE), b"", b""
        if rc != RSQ_ENOSPC:
            _check(rc)
        r1, r2 = DeviceArray(self.device, l1 + 64), DeviceArray(self.device, l2 + 64)
        fr = DeviceArray(self.device, (n + 1) * FRAGMENT_DTYPE.itemsize)
        try:
            n2, l1b, l2b, rc = self.pairs_device(block_lo, block_hi, r1, r2, fr, stream)
            _check(rc)
            assert (n2, l1b, l2b) == (n, l1, l2)
            return fr.to_numpy(FRAGMENT_DTYPE, n), r1.to_numpy(np.uint8, l1).tobytes(), r2.to_numpy(np.uint8, l2).tobytes()
        finally:
            for d in (r1, r2, fr):
                d.free()

    def job_generate(self, block_lo, block_hi, batch_blocks=0, stream=None):
        """the rank's share of a job: simulated once, its FASTQ text kept in device memory; returns (pairs, bytes of file 1, bytes of file 2)"""
        n, b1, b2 = _u64(0), _u64(0), _u64(0)
        _check(lib().rsq_sim_job_generate(self.h, block_lo, block_hi, batch_blocks, C.byref(n), C.byref(b1), C.byref(b2), stream))
        return n.value, b1.value, b2.value

    def job_write(self, r1_path, r1_offset, r2_path, r2_offset, threads_per_file=0):
        """the kept text to its place in the two final files (parallel pwrite from page-locked buffers); r2_path None: a job with one file"""
        _check(lib().rsq_sim_job_write(self.h, str(r1_path).encode(), r1_offset, str(r2_path).encode() if r2_path is not None else None, r2_offset, threads_per_file))

    def gzip_device(self, text_dev, text_len, out_dev=None, out_cap=0, stream=None):
        """text_dev[0, text_len) -- a DeviceArray or an address in device memory -- as gzip members in out_dev (rsq_sim_gzip_device); returns (bytes of the members, rc):
        rc RSQ_ENOSPC when out_cap is smaller than that"""
        ptr = lambda d: d.ptr if isinstance(d, DeviceArray) else C.c_void_p(d) if d else None
        n = C.c_size_t(0)
        rc = lib().rsq_sim_gzip_device(self.h, ptr(text_dev), text_len, ptr(out_dev), out_cap, C.byref(n), stream)
        if rc not in (RSQ_OK, RSQ_ENOSPC):
            _check(rc)
        return n.value, rc

    def gzip_keep_code(self, keep=True):
        """the next gzip_device call's Huffman code serves the calls after it (rsq_sim_gzip_keep_code)"""
        _check(lib().rsq_sim_gzip_keep_code(self.h, 1 if keep else 0))

    def gzip(self, text):
        """bytes -> the gzip members the device makes of them (a test's and a tool's convenience: upload, rsq_sim_gzip_device, download)"""
        if not text:
            return b""
        src = DeviceArray.from_numpy(self.device, np.frombuffer(text, np.uint8))
        out = DeviceArray(self.device, lib().rsq_gzip_bound(len(text)))
        try:
            n, rc = self.gzip_device(src, len(text), out, out.nbytes)
            _check(rc)
            return out.to_numpy(np.uint8, n).tobytes()
        finally:
            src.free()
            out.free()

    def gzip_end(self):
        """what ends a .gz file whose members this simulator made: BGZF's end-of-file member behind device-made members, nothing behind zlib's (option host_gzip)"""
        return b"" if get_option("host_gzip") else gzip_eof_member()

    def job_compress(self):
        """the kept text as gzip members (rsq_sim_job_compress: made on the device and kept there; in host memory with option host_gzip); returns the compressed
        sizes of the two files, which job_write then writes"""
        b1, b2 = _u64(0), _u64(0)
        _check(lib().rsq_sim_job_compress(self.h, C.byref(b1), C.byref(b2)))
        return b1.value, b2.value

    def job_read(self, file, at, nbytes, dst_ptr, stream=None):
        """bytes [at, at + nbytes) of the kept text of file 0 / 1 into device memory at `dst_ptr` (an integer address, e.g. a torch tensor's data_ptr())"""
        _check(lib().rsq_sim_job_read(self.h, file, at, nbytes, C.c_void_p(dst_ptr), stream))

    def job_free(self):
        _check(lib().rsq_sim_job_free(self.h))

    def adapter_only_pairs(self, first, n, stream=None):
        l1, l2 = C.c_size_t(), C.c_size_t()
        rc = lib().rsq_sim_adapter_only_pairs(self.h, first, n, None, 0, C.byref(l1), None, 0, C.byref(l2), stream)
        if rc == RSQ_OK:
            return b"", b""
        if rc != RSQ_ENOSPC:
            _check(rc)
        r1, r2 = DeviceArray(self.device, l1.value + 64), DeviceArray(self.device, l2.value + 64)
        try:
            _check(lib().rsq_sim_adapter_only_pairs(self.h, first, n, r1.ptr, r1.nbytes, C.byref(l1), r2.ptr, r2.nbytes, C.byref(l2), stream))
            return r1.to_numpy(np.uint8, l1.value).tobytes(), r2.to_numpy(np.uint8, l2.value).tobytes()
        finally:
            r1.free()
            r2.free()

    def error_model(self, rec, first_index=0, out_stride=None, cigar_stride=256, stream=None):
        """Simulator::SimulateErrorModelOnly on arrays (rec as made by synth.make_error_model_input).
        Returns list of (seq codes bytes, qual bytes, cigar str, num_errors, tile id)."""
        n, rl = rec["seqs"].shape
        out_stride = out_stride or 1024
        dev = self.device
        ins = [DeviceArray.from_numpy(dev, np.ascontiguousarray(rec[k], dt)) for k, dt in
               (("seqs", np.uint8), ("seg", np.uint8), ("frag_len", np.uint32), ("dom", np.uint8), ("rate", np.uint8))]
        outs = [DeviceArray(dev, n * out_stride), DeviceArray(dev, n * out_stride), DeviceArray(dev, n * 2), DeviceArray(dev, n * 2),
                DeviceArray(dev, n * 2), DeviceArray(dev, n * cigar_stride)]
        try:
            _check(lib().rsq_sim_error_model(self.h, first_index, n, rl, ins[0].ptr, ins[1].ptr, ins[2].ptr, ins[3].ptr, ins[4].ptr, outs[0].ptr, outs[1].ptr,
                                             out_stride, outs[2].ptr, outs[3].ptr, outs[4].ptr, outs[5].ptr, cigar_stride, stream))
            seq = outs[0].to_numpy(np.uint8, n * out_stride).reshape(n, out_stride)
            qual = outs[1].to_numpy(np.uint8, n * out_stride).reshape(n, out_stride)
            rlen = outs[2].to_numpy(np.uint16, n)
            nerr = outs[3].to_numpy(np.uint16, n)
            tile = outs[4].to_numpy(np.uint16, n)
            cig = outs[5].to_numpy(np.uint8, n * cigar_stride).reshape(n, cigar_stride)
            return [(seq[i, :rlen[i]].tobytes(), qual[i, :rlen[i]].tobytes(), cig[i].tobytes().split(b"\0")[0].decode(), int(nerr[i]), int(tile[i]))
                    for i in range(n)]
        finally:
            for d in ins + outs:
                d.free()

    def error_model_fastq(self, rec, ids, first_index=0, stream=None):
        """the same records as FASTQ text formatted on the device: "@{id} {CIGAR} E{errors}" (Simulator.cpp:2497-2504); ids: list of bytes"""
        n, rl = rec["seqs"].shape
        dev = self.device
        ins = [DeviceArray.from_numpy(dev, np.ascontiguousarray(rec[k], dt)) for k, dt in
               (("seqs", np.uint8), ("seg", np.uint8), ("frag_len", np.uint32), ("dom", np.uint8), ("rate", np.uint8))]
        blob = b"".join(ids)
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum([len(x) for x in ids])
        ins.append(DeviceArray.from_numpy(dev, np.frombuffer(blob + b"\0", np.uint8)))
        ins.append(DeviceArray.from_numpy(dev, off))
        need = C.c_size_t(0)
        text = None
        try:
            rc = lib().rsq_sim_error_model_fastq(self.h, first_index, n, rl, ins[0].ptr, ins[1].ptr, ins[2].ptr, ins[3].ptr, ins[4].ptr, ins[5].ptr, ins[6].ptr,
                                                 None, 0, C.byref(need), stream)
            if rc != RSQ_ENOSPC:
                _check(rc)
            text = DeviceArray(dev, need.value + 16)
            _check(lib().rsq_sim_error_model_fastq(self.h, first_index, n, rl, ins[0].ptr, ins[1].ptr, ins[2].ptr, ins[3].ptr, ins[4].ptr, ins[5].ptr, ins[6].ptr,
                                                   text.ptr, need.value + 16, C.byref(need), stream))
            return text.to_numpy(np.uint8, need.value).tobytes()
        finally:
            for d in ins + ([text] if text else []):
                d.free()

    def error_model_fasta(self, text, first_index=0, final=True, stream=None, skew=0):
        """seqToIllumina's FASTA text parsed on the device (rsq_sim_error_model_fasta; Simulator.cpp:2403-2512): text = bytes of the input file from a record's
        '>' on.  Returns (FASTQ text, records written, bytes consumed); final=False leaves the block's last record to the caller (hand text[consumed:] in
        again in front of what follows).  A malformed record raises with the reference's message."""
        dev = self.device
        src = DeviceArray.from_numpy(dev, np.frombuffer(b"\n" * skew + bytes(text) + b"\0" * 8, np.uint8))      # skew: the text begins that many bytes into the allocation
        text_ptr = C.c_void_p(src.ptr.value + skew)
        need, n, used = C.c_size_t(0), _u64(0), C.c_size_t(0)
        out = None
        try:
            rc = lib().rsq_sim_error_model_fasta(self.h, first_index, text_ptr, len(text), 1 if final else 0, None, 0, C.byref(need), C.byref(n), C.byref(used), stream)
            if rc != RSQ_ENOSPC:
                _check(rc)
                return b"", n.value, used.value
            out = DeviceArray(dev, need.value + 16)
            _check(lib().rsq_sim_error_model_fasta(self.h, first_index, text_ptr, len(text), 1 if final else 0, out.ptr, need.value + 16, C.byref(need), C.byref(n),
                                                   C.byref(used), stream))
            return out.to_numpy(np.uint8, need.value).tobytes(), n.value, used.value
        finally:
            for d in [src] + ([out] if out else []):
                d.free()

    def error_model_file(self, input_path, output_path, from_=0, to=0, first_record=0, read_threads=0, block_kb=0, batch_blocks=0, trace=False, keep_text=False):
        """Simulator::SimulateErrorModelOnly (Simulator.cpp:2900-3014) from file to file (rsq_sim_error_model_file); from_ / to / first_record: a rank's share of a
        plain input file; keep_text (output_path None): the text stays in device memory for job_write / job_read.  Returns (records, bytes of FASTQ) and, with
        trace, the line about the pipeline's stages."""
        opt = ErrorModelFileOptions(read_threads=read_threads, block_kb=block_kb, batch_blocks=batch_blocks, keep_text=1 if keep_text else 0, from_=from_, to=to,
                                    first_record=first_record)
        buf = C.create_string_buffer(2048) if trace else None
        if trace:
            opt.trace, opt.trace_cap = C.cast(buf, C.c_char_p), 2048
        n, nbytes = _u64(0), _u64(0)
        _check(lib().rsq_sim_error_model_file(self.h, os.fsencode(input_path) if input_path else None, os.fsencode(output_path) if output_path else None, C.byref(opt),
                                              C.byref(n), C.byref(nbytes)))
        return (n.value, nbytes.value, buf.value.decode()) if trace else (n.value, nbytes.value)

    def last_kernel_ms(self, name):
        """sum over the last call's launches of the kernel (a large call runs in pipelined sub-ranges)"""
        v = C.c_double()
        _check(lib().rsq_sim_last_kernel_ms(self.h, name.encode(), C.byref(v)))
        return v.value

    def last_kernel_launches(self, name):
        v = _u32(0)
        _check(lib().rsq_sim_last_kernel_launches(self.h, name.encode(), C.byref(v)))
        return v.value

    def close(self):
        if self.h:
            lib().rsq_sim_free(self.h)
            self.h = C.c_void_p()
