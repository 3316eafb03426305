"""Test infrastructure: the host emulation of the kernels (tests/hostemu) behind the two interfaces reseq_amd.simulate drives -- the GPU run uses
simulate.GpuBackend and api.Simulator.  tests/simulate_under_test.py --emulate hands these classes to reseq_amd.simulate.main as its hooks, so that the launcher's
N-rank path -- one process per rank under torch.distributed.run, the exchanges, the offsets, the failure
handling -- runs in the CPU suite exactly as it will run over RCCL (tests/test_multi_gpu.py has both modes of every test)."""
import os

import numpy as np

from backends import EmuBackend, emu_parse_fasta


def _fail(switch, message, error=IOError):
    """RSQ_FAIL_<SWITCH>=<rank>: the named rank fails in the named place (the tests of simulate._agree)"""
    if os.environ.get("RSQ_FAIL_" + switch) == os.environ.get("RANK", "0"):
        raise error(message + " (the test's)")


class EmuRankBackend:
    """what simulate.run_rank / load_once_per_host / sharding.sharded_prepare drive"""

    def __init__(self, profile_path, fasta_path, replace_n_seed=0, vcf_path=None, methylation_path=None, sys_error_path=None, packed_from=None, edits=None, ref_bias_file=None):
        if packed_from:             # the reference another rank of the "host" packed (simulate.load_once_per_host)
            _fail("IMPORT", "cannot map the packed reference")
            self.b = EmuBackend(profile_path, None, 0, edits)
            self.b.import_reference(packed_from)
        else:
            _fail("LOAD", "reference file not found")
            self.b = EmuBackend(profile_path, fasta_path, replace_n_seed, edits, vcf_path=vcf_path)
            if methylation_path:
                self.b.read_methylation(methylation_path)
        if ref_bias_file:
            self.b.set_ref_bias_file(ref_bias_file)
        self.sys_error_path = sys_error_path
        self.seq_len = self.b.sequence_lengths()
        self.text = None

    @property
    def can_shard_prepare(self):
        return not self.sys_error_path

    def export_reference(self, path):
        self.b.export_reference(path)

    def create_sys_error_profile(self, seed, path):
        self.b.create_sys_error_profile(seed, path)

    def close(self):
        self.b.close()

    def prepare(self, *a):
        info = self.b.prepare(*a)
        if self.sys_error_path:
            self.b.read_sys_errors(self.sys_error_path)
        return info

    def prepare_plan(self, *a):
        return self.b.prepare_plan(*a)

    def bias_partials(self, lo, hi):
        _fail("BIAS", "no memory for the bias sums", MemoryError)
        return self.b.bias_partials(lo, hi)

    def prepare_normalization(self, sums, maxes):
        self.b.prepare_normalization(sums, maxes)

    def prepare_sys_errors(self, lo, hi, in_state):
        _fail("CHAINS", "the chains failed", RuntimeError)
        return self.b.prepare_sys_errors(lo, hi, in_state)

    def prepare_finish(self):
        return self.b.prepare_finish()

    def ref_seq_bias(self):
        return self.b.ref_seq_bias(len(self.seq_len))

    def job_generate(self, lo, hi, batch_blocks):          # what rsq_sim_job_generate / rsq_sim_job_write do, for the host emulation: the text kept, then put in place
        from reseq_amd import sharding
        self.text, n = [bytearray(), bytearray()], 0
        for a, b in sharding.batches(lo, hi, batch_blocks or 2000):
            fr, t1, t2 = self.b.pairs(a, b)
            n += len(fr)
            self.text[0] += t1
            self.text[1] += t2
        return n, len(self.text[0]), len(self.text[1])

    def job_compress(self):                                # rsq_sim_job_compress: gzip members of 1 MB of text
        import gzip
        self.text = [bytearray(b"".join(gzip.compress(bytes(t[k:k + (1 << 20)]), 6) for k in range(0, len(t), 1 << 20))) for t in self.text]
        return len(self.text[0]), len(self.text[1])

    def job_write(self, path1, offset1, path2, offset2):
        _fail("WRITE", "no space left on the device")
        for path, offset, text in ((path1, offset1, self.text[0]), (path2, offset2, self.text[1])):
            fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)
            os.pwrite(fd, bytes(text), offset)
            os.close(fd)
        self.text = None

    def adapter_only_pairs(self, first, n):
        return self.b.adapter_only_pairs(first, n)

    def job_slice(self, file, at, n, size):                # --gatherOutput: a fixed-size slice of the kept text as a tensor, a received one to its place
        import torch
        t = torch.zeros(size, dtype=torch.uint8)
        if n:
            t[:n] = torch.frombuffer(bytearray(self.text[file][at:at + n]), dtype=torch.uint8)
        return t

    def write_slice(self, tensor, n, path, offset):
        fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)
        os.pwrite(fd, tensor[:n].numpy().tobytes(), offset)
        os.close(fd)

    def job_free(self):
        self.text = None


class EmuRecordsSim:
    """what simulate.run_records_rank drives (api.Simulator on a GPU): seqToIllumina's records of a byte range of the input through rsq_fasta.h's parser and
    the read machine, both run on the CPU by tests/hostemu; the FASTQ text as rsq_sim_error_model_fasta writes it"""

    def __init__(self, profile_path, seed, edits=None):
        self.b = EmuBackend(profile_path, None, 0, edits)
        self.b.prepare(seed)
        self.text = None

    def error_model_file(self, input_path, output_path, from_=0, to=0, first_record=0, keep_text=False, **_):
        assert keep_text and output_path is None
        _fail("RECORDS", "Template segment is 3 not 1 or 2: r 3;1;N;!", RuntimeError)
        with open(input_path, "rb") as f:
            f.seek(from_)
            text = f.read((to - from_) if to else -1)
        p = emu_parse_fasta(text, final=True)
        if p["bad"] is not None:
            raise RuntimeError(f"malformed record {first_record + p['bad']} in {input_path}")
        out = []
        order = {}
        for i in range(p["n"]):                                                # the read machine takes records of one length per call
            order.setdefault(int(p["len"][i]), []).append(i)
        done = [None] * p["n"]
        for length, members in order.items():
            for i in members:                                                  # a record's random stream is selected by its index in the input
                rec = {"seqs": p["seqs"][i][None, :], "seg": p["seg"][i:i + 1], "frag_len": p["frag_len"][i:i + 1], "dom": p["dom"][i][None, :], "rate": p["rate"][i][None, :]}
                done[i] = self.b.error_model(rec, first_index=first_record + i)[0]
        for i, (seq, qual, cigar, errors, _tile) in enumerate(done):
            at = int(p["at"][i])
            rid = text[at + 1:at + 1 + int(p["id_len"][i])]
            out.append(b"@" + rid + b" " + cigar.encode() + b" E%d\n" % errors + bytes(b"ACGTN"[c] for c in seq) + b"\n+\n" + qual + b"\n")
        self.text = b"".join(out)
        return p["n"], len(self.text)

    def job_compress(self):
        import gzip
        self.text = b"".join(gzip.compress(self.text[k:k + (1 << 20)], 6) for k in range(0, len(self.text), 1 << 20))
        return len(self.text), 0

    def job_write(self, path, offset, path2, offset2):
        assert path2 is None
        fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)
        os.pwrite(fd, self.text, offset)
        os.close(fd)

    def job_free(self):
        self.text = None

    def close(self):
        self.b.close()
