import os
import subprocess
import sys

import numpy as np
import pytest

# Device-side text is chosen per region by the share of lines the host would have to rewrite (deep indel-rich fuzz data is
# above the production threshold): the tests force it so that the rewriting is what they exercise; test_cli covers the
# default threshold too.
# (until round 6 the suite forced device-side text for indel-rich data here: BRC_DEVICE_TEXT_MAX_SHARE; the product has no such threshold any more)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))       # synthgen, bamio, cramio (test / bench infrastructure)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """CPU oracle (test infrastructure only), built on demand with gcc."""
    from bam_readcount_amd import capi
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return capi.Library(os.path.join(ROOT, "oracle", "libbrc_oracle.so"))


@pytest.fixture(scope="session")
def sim_lib():
    """The product's host code + shared device functions executed on the CPU by tests/sim (test infrastructure only)."""
    from bam_readcount_amd import capi
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    return capi.Library(os.path.join(ROOT, "tests", "sim", "libbrc_sim.so"))


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; fails loudly when it was not built (no fallback)."""
    from bam_readcount_amd import capi
    return capi.load_product()


@pytest.fixture(scope="session", params=["sim", pytest.param("hip", marks=pytest.mark.gpu)])
def dev_lib(request):
    """The device algorithm behind the C-ABI, twice: [hip] = the product library on the GPU (gpu-marked: the parity tests
    proper), [sim] = the same test body against the CPU lane simulator, so that the CPU suite executes every assertion a
    GPU test makes (a stale literal cannot hide until the GPU box runs it)."""
    from bam_readcount_amd import capi
    if request.param == "hip":
        return capi.load_product()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    return capi.Library(os.path.join(ROOT, "tests", "sim", "libbrc_sim.so"))


@pytest.fixture(scope="session", params=["sim", pytest.param("hip", marks=pytest.mark.gpu)])
def knob_lib(request):
    """For the tests that steer the device paths with the TEST KNOBS (BRC_FLUSH_K, BRC_PACK_LIM, BRC_FORCE_DOM, BRC_XEV_CAP, BRC_NO_TABLE,
    BRC_IBUCKET_SHIFT, BRC_DEVICE_TEXT_LIMIT): [hip] = libbrc_hip_testknobs.so — the product's own objects (the same kernels, the same host
    code) linked with brc_knobs.cpp -DBRC_TEST_KNOBS; the product library itself does not read them (tests/test_abi.py).  [sim]: the
    simulator reads them itself."""
    from bam_readcount_amd import capi
    if request.param == "hip":
        path = os.path.join(ROOT, "bam_readcount_amd", "csrc", "libbrc_hip_testknobs.so")
        if not os.path.exists(path):
            raise RuntimeError("libbrc_hip_testknobs.so is not built (make -C bam_readcount_amd/csrc)")
        return capi.Library(path)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    return capi.Library(os.path.join(ROOT, "tests", "sim", "libbrc_sim.so"))


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    arrs = {k: z[k] for k in z.files}
    ref = np.full(int(arrs["ref_len"]), ord("N"), np.uint8)
    s = int(arrs["ref_start"])
    ref[s:s + arrs["ref_slice"].size] = arrs["ref_slice"]
    arrs["ref"] = ref
    return arrs


@pytest.fixture(scope="session")
def test_bam():
    return load_fixture("test_bam.npz")


@pytest.fixture(scope="session")
def twolib():
    return load_fixture("twolib.npz")


@pytest.fixture(scope="session")
def workdir(tmp_path_factory, test_bam):
    d = tmp_path_factory.mktemp("cli")
    ref = test_bam["ref"]
    n = ref.size; L = 60; rows = (n + L - 1) // L
    pad = np.full(rows * L, ord("\n"), np.uint8); pad[:n] = ref
    body = np.concatenate([pad.reshape(rows, L), np.full((rows, 1), 10, np.uint8)], axis=1).tobytes()
    with open(d / "ref.fa", "wb") as f:
        f.write(b">21\n"); f.write(body)
    open(d / "ref.fa.fai", "w").write("21\t%d\t4\t60\t61\n" % n)      # same index line as test-data/ref.fa.fai
    for f in ("test.bam", "test.bam.bai", "test_bad_rg.bam", "test_bad_rg.bam.bai", "site_list"):
        os.symlink(os.path.join(GOLDEN, f), d / f)
    return d
