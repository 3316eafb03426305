"""GPU parity: the product (libreseq_amd.so through the C ABI, reseq_amd.api) against the CPU oracle on the same
seeded inputs.  Same cases as tests/test_parity_hostemu.py (tests/parity_cases.py)."""
import pytest

import parity_cases as P
from backends import GpuBackend

pytestmark = pytest.mark.gpu


def test_device_visible():
    from reseq_amd import api
    assert api.device_count() >= 1


def test_reference_packing(workdir):
    P.case_reference_packing(GpuBackend, workdir)


def test_prepass(workdir):
    P.case_prepass(GpuBackend, workdir)


def test_sieve_and_reads_tiny(workdir):
    P.case_sieve_and_reads_tiny(GpuBackend, workdir)


def test_sieve_own_thresholds(workdir):
    P.case_sieve_own_thresholds(GpuBackend, workdir)


def test_dense_coverage(workdir):
    P.case_dense_coverage(GpuBackend, workdir)


def test_adapter_only(workdir):
    P.case_adapter_only(GpuBackend, workdir)


def test_p0_reads(workdir):
    P.case_p0_reads(GpuBackend, workdir)


def test_profile_edits(workdir):
    P.case_profile_edits(GpuBackend, workdir)


def test_error_model_tiny(workdir):
    P.case_error_model_tiny(GpuBackend, workdir)


def test_error_model_long_templates(workdir):
    P.case_error_model_long_templates(GpuBackend, workdir)


def test_error_model_p0(workdir):
    P.case_error_model_p0(GpuBackend, workdir)


@pytest.mark.parametrize("mode", [0, 1, 3, 7, 19])
def test_every_lds_staging_mode(workdir, mode, monkeypatch):
    """k_fill_reads<MASK>: tables from HBM only (0), descriptors in LDS (1), + quality margins (3), + base-call margin (7),
    quality margins + error-rate rows (19); everything the plan allows (23) is the default of the tests above"""
    monkeypatch.setenv("RSQ_FILL_MODE", str(mode))
    P.case_sieve_and_reads_tiny(GpuBackend, workdir)
    P.case_p0_reads(GpuBackend, workdir)


@pytest.mark.gpu
def test_error_rate_rows_fall_back_to_hbm(workdir, monkeypatch):
    """only row 0 of the error-rate margins staged: every position with a systematic error rate takes the HBM branch"""
    monkeypatch.setenv("RSQ_RATE_ROWS", "1")
    P.case_sieve_and_reads_tiny(GpuBackend, workdir)
    P.case_p0_reads(GpuBackend, workdir)
