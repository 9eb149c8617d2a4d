import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import sanerf_hq_amd
    from sanerf_hq_amd import _lib
    # the product path must be the native library: fail loudly if it is not there
    assert os.path.exists(_lib.LIB_PATH), f"{_lib.LIB_PATH} missing on the GPU box"
    assert _lib.lib().sn_device_count() >= 1, _lib.lib().sn_last_error()
    # SN_TEST_TUNING="band_streams=2,...": run the whole GPU suite under a non-default kernel selection (bit-neutral fields only make sense:
    # e.g. every image-mode render with proposal stages as two row bands on two HIP streams)
    if os.environ.get("SN_TEST_TUNING"):
        from sanerf_hq_amd import raymarching as rm
        for item in os.environ["SN_TEST_TUNING"].split(","):
            k, v = item.split("=")
            assert k.strip() in rm.Tuning.FIELDS, k
            setattr(rm.tuning, k.strip(), int(v))
    return torch.device("cuda:0")


@pytest.fixture
def experiments_build(gpu):
    """Tests of the measured-and-rejected kernel variants need a library built with -DSN_EXPERIMENTS (make -C sanerf-hq_amd/csrc exp,
    SN_LIB=sanerf-hq_amd/libsanerf_hip_exp.so): the product library does not carry those kernels."""
    from sanerf_hq_amd import _lib
    if not (_lib.lib().sn_build_flags() & _lib.BUILD_EXPERIMENTS):
        pytest.skip("needs the experiments build of the library (SN_LIB=.../libsanerf_hip_exp.so)")


@pytest.fixture
def per_sample_form(monkeypatch):
    """The default final stage applies the third MLP layer's geometry rows once per ray to the weight-accumulated hidden vector
    (the "linear tail", render.hip / DESIGN.md section 5) wherever no per-sample tensor leaves the kernel.  The other final-stage
    kernels (several lanes per ray, live-sample compaction, role-split waves) and every call that exports per-sample tensors keep
    the per-sample form; their BIT identity with the default kernel is a statement about that form, so tests that assert it pin
    the default kernel to it.  (test_linear_tail_form_* bounds the difference between the two forms: fp32 round-off.)"""
    from sanerf_hq_amd import raymarching as rm
    monkeypatch.setattr(rm.tuning, "per_sample_form", 1)
