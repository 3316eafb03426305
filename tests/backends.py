"""Two drivers with one interface for the parity tests:

EmuBackend   tests/hostemu/libhostemu.so -- the kernels' per-lane functions looped on the CPU (TEST-ONLY artefact,
             lets `-m "not gpu"` check state machine / packing / counters against the oracle in this container);
GpuBackend   reseq_amd.api -> libreseq_amd.so through the C ABI (the product; `-m gpu`).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from reseq_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "hostemu")
FRAGMENT_DTYPE = api.FRAGMENT_DTYPE


class EmuInfo(C.Structure):
    _fields_ = [("total_pairs", C.c_uint64), ("adapter_only_pairs", C.c_uint64), ("total_blocks", C.c_uint32), ("n_groups", C.c_uint32),
                ("insert_to", C.c_uint32), ("passes", C.c_uint32), ("n_seqs", C.c_uint32), ("rmax", C.c_uint32), ("bias_normalization", C.c_double)]


_emu = None


def emu_lib():
    global _emu
    if _emu is None:
        subprocess.run(["make", "-C", EMU_DIR, "-s"], check=True)
        L = C.CDLL(os.path.join(EMU_DIR, "libhostemu.so"))
        L.emu_last_error.restype = C.c_char_p
        L.emu_create.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_char_p, C.POINTER(C.c_void_p)]
        L.emu_get_variant_sys.restype = C.c_uint32
        L.emu_get_variant_sys.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32]
        L.emu_free.argtypes = [C.c_void_p]
        L.emu_export_reference.argtypes = [C.c_void_p, C.c_char_p]
        L.emu_import_reference.argtypes = [C.c_void_p, C.c_char_p]
        L.emu_edit_profile.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int]
        L.emu_set_fill_mode.argtypes = [C.c_void_p, C.c_int]
        L.emu_set_ring_lag.argtypes = [C.c_uint32]
        L.emu_set_option.argtypes = [C.c_char_p, C.c_longlong]
        L.emu_get_option.argtypes = [C.c_char_p]
        L.emu_get_option.restype = C.c_longlong
        L.emu_image_tiles.argtypes = [C.c_void_p]
        L.emu_plan_mask.argtypes = [C.c_void_p]
        L.emu_prepare.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_double, C.c_int, C.c_char_p]
        L.emu_prepare_plan.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_double, C.c_int, C.c_char_p]
        L.emu_bias_partials_size.restype = C.c_uint64
        L.emu_bias_partials_size.argtypes = [C.c_void_p]
        L.emu_bias_partials.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.emu_prepare_normalization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.emu_prepare_sys_errors.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.emu_prepare_finish.argtypes = [C.c_void_p]
        L.emu_get_info.argtypes = [C.c_void_p, C.POINTER(EmuInfo)]
        L.emu_get_thresholds.argtypes = [C.c_void_p, C.c_void_p]
        L.emu_get_sequence_lengths.argtypes = [C.c_void_p, C.c_void_p]
        L.emu_get_norm_by_len.argtypes = [C.c_void_p, C.c_void_p]
        L.emu_set_normalization.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_size_t]
        L.emu_get_sys.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        L.emu_get_adapter_sys.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        L.emu_create_sys_error_profile.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p]
        L.emu_read_sys_errors.argtypes = [C.c_void_p, C.c_char_p]
        L.emu_set_ref_bias_file.argtypes = [C.c_void_p, C.c_char_p]
        L.emu_read_methylation.argtypes = [C.c_void_p, C.c_char_p]
        L.emu_get_ref_seq_bias.argtypes = [C.c_void_p, C.c_void_p]
        L.emu_get_codes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.emu_sieve.restype = C.c_int64
        L.emu_sieve.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]
        L.emu_pairs_text.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_size_t)]
        L.emu_error_model.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32] + [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint32]
        L.emu_parse_fasta.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint32] + [C.c_void_p] * 8 + [C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)] + [C.POINTER(C.c_uint32)] * 3 + [C.c_void_p]
        L.emu_draw.restype = C.c_uint32
        L.emu_draw.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_double, C.POINTER(C.c_double)]
        L.emu_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        _emu = L
    return _emu


def _ok(rc):
    if rc != 0:
        raise RuntimeError(emu_lib().emu_last_error().decode())


def set_option_everywhere(name, value):
    """rsq_set_option of the product library (host-side call: works without a GPU) and its twin in the host emulation"""
    api.set_option(name, value)
    if emu_lib().emu_set_option(name.encode(), int(value)) != 0:
        raise KeyError(name)


def emu_parse_fasta(text, final=True):
    """rsq_fasta.h's record_start / parse_record run on the CPU over a block of seqToIllumina input.  Returns a dict: n, consumed, bad (index of the first
    malformed record or None), bad_kind, lead (text in front of the first record), and per record at, len, id_len, frag_len, seg, seqs, dom, rate"""
    import numpy as np
    L = emu_lib()
    buf = np.frombuffer(bytes(text) + b"\0" * 8, np.uint8)
    cap = text.count(b">") + 1
    at = np.zeros(cap + 1, np.uint32)
    ln, idl, fl = (np.zeros(cap, np.uint32) for _ in range(3))
    seg = np.zeros(cap, np.uint8)
    seqs, dom, rate = (np.full(len(text) + 8, 0xEE, np.uint8) for _ in range(3))
    packed = np.full(len(text) + 8, 0xEEEE, np.uint16)
    n, used, bad, kind, lead = C.c_uint32(0), C.c_uint64(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    rc = L.emu_parse_fasta(buf.ctypes.data, len(text), 1 if final else 0, cap, at.ctypes.data, ln.ctypes.data, idl.ctypes.data, fl.ctypes.data, seg.ctypes.data,
                           seqs.ctypes.data, dom.ctypes.data, rate.ctypes.data, C.byref(n), C.byref(used), C.byref(bad), C.byref(kind), C.byref(lead), packed.ctypes.data)
    assert rc == 0
    k = n.value
    out = {"n": k, "consumed": used.value, "bad": None if bad.value == 0xFFFFFFFF else bad.value, "bad_kind": kind.value, "lead": bool(lead.value), "at": at[:k + 1].copy(),
           "len": ln[:k].copy(), "id_len": idl[:k].copy(), "frag_len": fl[:k].copy(), "seg": seg[:k].copy()}
    for name, arr in (("seqs", seqs), ("dom", dom), ("rate", rate)):
        out[name] = [arr[at[i]:at[i] + ln[i]].copy() for i in range(k)]
    out["arrays"] = (seqs, dom, rate)
    out["packed"] = packed               # the device's layout: base | dominant error << 2 | percent << 8 per base, at the record's offset
    return out


class EmuBackend:
    name = "hostemu"

    fill_mode = -1          # class attribute: tests subclass to cap the LDS staging mode

    def __init__(self, profile_path, fasta_path=None, replace_n_seed=0, edits=None, vcf_path=None):
        self.L = emu_lib()
        self.h = C.c_void_p()
        _ok(self.L.emu_create(str(profile_path).encode(), (str(fasta_path) if fasta_path else "").encode(), replace_n_seed,
                              (str(vcf_path) if vcf_path else "").encode(), C.byref(self.h)))
        if edits:
            _ok(self.L.emu_edit_profile(self.h, edits.get("error_multiplier", 1.0), int(edits.get("no_substitutions", False)), int(edits.get("no_indels", False))))
        if self.fill_mode != -1:
            self.L.emu_set_fill_mode(self.h, self.fill_mode)

    def fill_plan(self):
        return {"mask": self.L.emu_plan_mask(self.h), "image_tiles": self.L.emu_image_tiles(self.h)}

    def export_reference(self, path):
        _ok(self.L.emu_export_reference(self.h, str(path).encode()))

    def import_reference(self, path):
        _ok(self.L.emu_import_reference(self.h, str(path).encode()))

    def prepare(self, seed, num_pairs=0, coverage=0.0, ref_bias_mode=0, base_identifier=""):
        _ok(self.L.emu_prepare(self.h, seed, num_pairs, coverage, ref_bias_mode, base_identifier.encode()))
        return self.info()

    # the pre-pass of one rank of a sharded job, as api.Simulator offers it
    def prepare_plan(self, seed, num_pairs=0, coverage=0.0, ref_bias_mode=0, base_identifier=""):
        _ok(self.L.emu_prepare_plan(self.h, seed, num_pairs, coverage, ref_bias_mode, base_identifier.encode()))
        return self.info()

    def bias_partials(self, block_lo, block_hi):
        n = self.L.emu_bias_partials_size(self.h)
        sums, maxes = np.zeros(n), np.zeros(n)
        _ok(self.L.emu_bias_partials(self.h, block_lo, block_hi, sums.ctypes.data, maxes.ctypes.data))
        return sums, maxes

    def prepare_normalization(self, sums, maxes):
        sums, maxes = np.ascontiguousarray(sums, np.float64), np.ascontiguousarray(maxes, np.float64)
        _ok(self.L.emu_prepare_normalization(self.h, sums.ctypes.data, maxes.ctypes.data))

    def prepare_sys_errors(self, block_lo, block_hi, in_state):
        i, o = np.asarray(in_state, np.uint32), np.zeros(2, np.uint32)
        _ok(self.L.emu_prepare_sys_errors(self.h, block_lo, block_hi, i.ctypes.data, o.ctypes.data))
        return [int(o[0]), int(o[1])]

    def prepare_finish(self):
        _ok(self.L.emu_prepare_finish(self.h))
        return self.info()

    def info(self):
        i = EmuInfo()
        self.L.emu_get_info(self.h, C.byref(i))
        return dict(total_pairs=i.total_pairs, adapter_only_pairs=i.adapter_only_pairs, total_blocks=i.total_blocks, n_groups=i.n_groups, insert_to=i.insert_to,
                    passes=i.passes, bias_normalization=i.bias_normalization)

    def sequence_lengths(self):
        i = EmuInfo()
        self.L.emu_get_info(self.h, C.byref(i))
        out = np.zeros(i.n_seqs, np.uint32)
        self.L.emu_get_sequence_lengths(self.h, out.ctypes.data)
        return [int(x) for x in out]

    def thresholds(self):
        i = self.info()
        out = np.zeros((i["n_groups"], i["insert_to"], 2))
        self.L.emu_get_thresholds(self.h, out.ctypes.data)
        return out

    def norm_by_len(self):
        out = np.zeros(self.info()["insert_to"])
        self.L.emu_get_norm_by_len(self.h, out.ctypes.data)
        return out

    def set_normalization(self, bias_normalization, thresholds):
        t = np.ascontiguousarray(thresholds, np.float64)
        _ok(self.L.emu_set_normalization(self.h, bias_normalization, t.ctypes.data, t.size))

    def sys_errors(self, reverse, seq, length):
        dom, rate = np.zeros(length, np.uint8), np.zeros(length, np.uint8)
        self.L.emu_get_sys(self.h, int(reverse), seq, dom.ctypes.data, rate.ctypes.data)
        return dom, rate

    def adapter_sys_errors(self, seg, adapter, length):
        dom, rate = np.zeros(length, np.uint8), np.zeros(length, np.uint8)
        self.L.emu_get_adapter_sys(self.h, seg, adapter, dom.ctypes.data, rate.ctypes.data)
        return dom, rate

    def variant_sys_errors(self, seq, var_id, reverse):
        """dom | rate << 8 per base of one variant on one strand, in the strand's drawing order"""
        out = np.zeros(4096, np.uint16)
        n = self.L.emu_get_variant_sys(self.h, seq, var_id, int(reverse), out.ctypes.data, len(out))
        return out[:n].copy()

    def codes(self, seq, length):
        out = np.zeros(length, np.uint8)
        self.L.emu_get_codes(self.h, seq, out.ctypes.data)
        return out

    def create_sys_error_profile(self, seed, path):
        _ok(self.L.emu_create_sys_error_profile(self.h, seed, str(path).encode()))

    def read_sys_errors(self, path):
        _ok(self.L.emu_read_sys_errors(self.h, str(path).encode()))

    def set_ref_bias_file(self, path):
        _ok(self.L.emu_set_ref_bias_file(self.h, str(path).encode()))

    def read_methylation(self, path):
        _ok(self.L.emu_read_methylation(self.h, str(path).encode()))

    def ref_seq_bias(self, n_sequences):
        out = np.zeros(n_sequences, np.float64)
        self.L.emu_get_ref_seq_bias(self.h, out.ctypes.data)
        return out

    def _text(self, frags, n, first):
        cap = max(4096, int(n) * 4096)
        b1, b2 = C.create_string_buffer(cap), C.create_string_buffer(cap)
        l1, l2 = C.c_size_t(), C.c_size_t()
        _ok(self.L.emu_pairs_text(self.h, frags.ctypes.data if frags is not None else None, n, first, b1, cap, C.byref(l1), b2, cap, C.byref(l2)))
        return b1.raw[:l1.value], b2.raw[:l2.value]

    def pairs(self, block_lo, block_hi):
        n = self.L.emu_sieve(self.h, block_lo, block_hi, None, 0)
        frags = np.zeros(max(n, 1), FRAGMENT_DTYPE)
        self.L.emu_sieve(self.h, block_lo, block_hi, frags.ctypes.data, n)
        frags = frags[:n]
        r1, r2 = self._text(frags, n, 0) if n else (b"", b"")
        return frags, r1, r2

    def adapter_only_pairs(self, first, n):
        return self._text(None, n, first) if n else (b"", b"")

    def error_model(self, rec, first_index=0, out_stride=1024, cigar_stride=256):
        n, rl = rec["seqs"].shape
        a = [np.ascontiguousarray(rec[k], dt) for k, dt in (("seqs", np.uint8), ("seg", np.uint8), ("frag_len", np.uint32), ("dom", np.uint8), ("rate", np.uint8))]
        seq, qual = np.zeros((n, out_stride), np.uint8), np.zeros((n, out_stride), np.uint8)
        rlen, nerr, tile = np.zeros(n, np.uint16), np.zeros(n, np.uint16), np.zeros(n, np.uint16)
        cig = np.zeros((n, cigar_stride), np.uint8)
        _ok(self.L.emu_error_model(self.h, first_index, n, rl, *[x.ctypes.data for x in a], seq.ctypes.data, qual.ctypes.data, out_stride, rlen.ctypes.data,
                                   nerr.ctypes.data, tile.ctypes.data, cig.ctypes.data, cigar_stride))
        return [(seq[i, :rlen[i]].tobytes(), qual[i, :rlen[i]].tobytes(), cig[i].tobytes().split(b"\0")[0].decode(), int(nerr[i]), int(tile[i])) for i in range(n)]

    def draw(self, family, index, idx, u):
        fam = {"quality": 0, "seq_quality": 1, "base_call": 2, "dom_error": 3, "error_rate": 4, "indels": 5}[family]
        ii = np.asarray(list(idx) + [0] * (4 - len(idx)), np.uint32)
        ps = C.c_double()
        v = self.L.emu_draw(self.h, fam, index, ii.ctypes.data, u, C.byref(ps))
        return v, ps.value

    def close(self):
        if self.h:
            self.L.emu_free(self.h)
            self.h = C.c_void_p()


class GpuBackend:
    name = "gpu"

    def __init__(self, profile_path, fasta_path=None, replace_n_seed=0, edits=None, device=0, vcf_path=None):
        self.prof = api.Profile(profile_path)
        if edits:
            if edits.get("error_multiplier", 1.0) != 1.0:
                self.prof.change_error_rate(edits["error_multiplier"])
            if edits.get("no_substitutions"):
                self.prof.remove_substitution_errors()
            if edits.get("no_indels"):
                self.prof.remove_indel_errors()
        self.ref = api.Reference(fasta_path, replace_n_seed) if fasta_path else None
        if vcf_path:
            self.ref.read_variants(vcf_path)
        self.sim = api.Simulator(self.prof, self.ref, device)

    def prepare(self, seed, num_pairs=0, coverage=0.0, ref_bias_mode=0, base_identifier=""):
        self.sim.prepare(seed, num_pairs, coverage, ref_bias_mode, base_identifier)
        return self.info()

    def fill_plan(self):
        return self.sim.fill_plan()

    def export_reference(self, path):
        self.sim.export_reference(str(path))

    def import_reference(self, path):
        self.sim.import_reference(str(path))

    def info(self):
        i = self.sim.info()
        return dict(total_pairs=i.total_pairs, adapter_only_pairs=i.adapter_only_pairs, total_blocks=i.total_blocks, n_groups=i.n_coverage_groups,
                    insert_to=i.insert_to, passes=i.sys_chain_passes, bias_normalization=i.bias_normalization)

    def thresholds(self):
        return self.sim.thresholds()

    def norm_by_len(self):
        return self.sim.norm_by_len()

    def set_normalization(self, bias_normalization, thresholds):
        self.sim.set_normalization(bias_normalization, thresholds)

    def sys_errors(self, reverse, seq, length):
        return self.sim.sys_errors(reverse, seq, length)

    def adapter_sys_errors(self, seg, adapter, length):
        return self.sim.adapter_sys_errors(seg, adapter, length)

    def codes(self, seq, length):
        return self.ref.codes(seq)

    def create_sys_error_profile(self, seed, path):
        self.sim.create_sys_error_profile(seed, path)

    def read_sys_errors(self, path):
        self.sim.read_sys_errors(path)

    def set_ref_bias_file(self, path):
        self.sim.set_ref_bias_file(path)

    def read_methylation(self, path):
        self.sim.read_methylation(path)

    def ref_seq_bias(self, n_sequences):
        return self.sim.ref_seq_bias(n_sequences)

    def pairs(self, block_lo, block_hi):
        return self.sim.pairs(block_lo, block_hi)

    def adapter_only_pairs(self, first, n):
        return self.sim.adapter_only_pairs(first, n)

    def error_model(self, rec, first_index=0, out_stride=1024, cigar_stride=256):
        return self.sim.error_model(rec, first_index, out_stride, cigar_stride)

    def error_model_fastq(self, rec, ids, first_index=0):
        return self.sim.error_model_fastq(rec, ids, first_index)

    def error_model_fasta(self, text, first_index=0, final=True, skew=0):
        return self.sim.error_model_fasta(text, first_index, final, skew=skew)

    # the pre-pass of one rank of a sharded job
    def prepare_plan(self, seed, num_pairs=0, coverage=0.0, ref_bias_mode=0, base_identifier=""):
        self.sim.prepare_plan(seed, num_pairs, coverage, ref_bias_mode, base_identifier)
        return self.info()

    def bias_partials(self, lo, hi):
        return self.sim.bias_partials(lo, hi)

    def prepare_normalization(self, sums, maxes):
        self.sim.prepare_normalization(sums, maxes)

    def prepare_sys_errors(self, lo, hi, in_state):
        return self.sim.prepare_sys_errors(lo, hi, in_state)

    def prepare_finish(self):
        self.sim.prepare_finish()
        return self.info()

    def close(self):
        self.sim.close()
        if self.ref:
            self.ref.close()
        self.prof.close()
