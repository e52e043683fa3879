"""The reference's six integration tests (integration-test/bam-readcount_test.py:29-116) re-expressed for the drop-in
command line: same arguments, byte-exact stdout against the reference's own golden files.

CPU: the CLI linked against the lane simulator (tests/sim/bam-readcount-sim) exercises option parsing, BGZF/BAM/BAI/
FASTA input and the region plumbing end-to-end.  GPU (-m gpu): the product binary bam_readcount_amd/csrc/bam-readcount."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

SIM_CLI = os.path.join(ROOT, "tests", "sim", "bam-readcount-sim")
HIP_CLI = os.path.join(ROOT, "bam_readcount_amd", "csrc", "bam-readcount")

RUNS = [  # (expected file, bam, extra args, how the sites are given)
    ("expected_all_lib", "test.bam", [], "list"),
    ("expected_per_lib", "test.bam", ["-p"], "list"),
    ("expected_all_lib", "test.bam", [], "regions"),
    ("expected_all_lib", "test_bad_rg.bam", [], "list"),
    ("expected_insertion_centric_all_lib", "test.bam", ["-i"], "list"),
    ("expected_insertion_centric_per_lib", "test.bam", ["-i", "-p"], "list"),
]


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


def run_cli(exe, workdir, bam, extra, how):
    args = [exe, "-w", "1"] + extra + ["-f", "ref.fa"]
    if how == "list":
        args += ["-l", "site_list", bam]
    else:
        args += [bam, "21:10402985-10402985", "21:10405200-10405200"]
    p = subprocess.run(args, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout, p.stderr.decode()


def check(exe, workdir):
    for exp, bam, extra, how in RUNS:
        rc, out, err = run_cli(exe, workdir, bam, extra, how)
        assert rc == 0, err
        assert out == open(os.path.join(GOLDEN, exp), "rb").read(), (exp, bam, extra, how)
        assert "Minimum mapping quality is set to 0" in err
        if bam == "test.bam":
            assert "Expect library: Solexa-135852 in BAM" in err and "Expect library: Solexa-135853 in BAM" in err


def test_cli_reference_integration_tests_cpu(workdir):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    check(SIM_CLI, workdir)


def test_cli_option_spellings_and_errors_cpu(workdir):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    exp = open(os.path.join(GOLDEN, "expected_insertion_centric_per_lib"), "rb").read()
    for extra in (["-pi"], ["--per-library", "--insertion-centric"], ["--per-lib", "--insertion-c"], ["-i", "-p", "-q0", "-b", "0", "--max-count=10000000"]):
        rc, out, _ = run_cli(SIM_CLI, workdir, "test.bam", extra, "list")
        assert rc == 0 and out == exp, extra
    # -h / -v print to stdout and return 1 (bamreadcount.cpp:467-475); missing index / unknown contig messages
    p = subprocess.run([SIM_CLI, "-h"], stdout=subprocess.PIPE)
    assert p.returncode == 1 and b"Usage: bam-readcount [OPTIONS] bam_file|cram_file [region]" in p.stdout
    p = subprocess.run([SIM_CLI, "-v"], stdout=subprocess.PIPE)
    assert p.returncode == 1 and p.stdout.startswith(b"bam-readcount version: ")
    p = subprocess.run([SIM_CLI, "-f", "ref.fa", "test.bam", "nochr:1-10"], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"Invalid region nochr:1-10" in p.stderr
    open(workdir / "sl2", "w").write("junk line\nchrZ 5 6\n21 10402985 10402985\n")
    p = subprocess.run([SIM_CLI, "-w1", "-f", "ref.fa", "-l", "sl2", "test.bam"], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"chrZ not found in bam file. Region chrZ 5 6 skipped." in p.stderr
    assert p.stdout == open(os.path.join(GOLDEN, "expected_all_lib"), "rb").read().split(b"\n")[0] + b"\n"
    # a thresholded run and a multi-kilobase region, chunked vs unchunked: identical text
    a = subprocess.run([SIM_CLI, "-q", "20", "-b", "13", "-f", "ref.fa", "test.bam", "21:10402737-10405248"], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    b = subprocess.run([SIM_CLI, "-q", "20", "-b", "13", "--brc-chunk", "500", "-f", "ref.fa", "test.bam", "21:10402737-10405248"], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and a.stdout.count(b"\n") > 700


@pytest.mark.gpu
def test_cli_reference_integration_tests_gpu(workdir):
    assert os.path.exists(HIP_CLI), "build the product first (python __graft_entry__.py)"
    check(HIP_CLI, workdir)
