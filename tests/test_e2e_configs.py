"""BASELINE configs 4 and 5 through the drop-in command line on multi-contig BAM + BAI files (tools/e2e_configs.py), at
test scale: the site list spread over every contig (duplicates, overlapping multi-base lines, lines repeated out of order)
and a -p -i region of a 4-library / 8-read-group file, both against the reference's OWN main() (oracle/_ref/bam-readcount-ref,
the shim's independent whole-file BAM decoder underneath) byte for byte.

CPU: the command line linked to the lane simulator.  GPU (-m gpu): the product binary."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SIM_CLI = os.path.join(ROOT, "tests", "sim", "bam-readcount-sim")
HIP_CLI = os.path.join(ROOT, "bam_readcount_amd", "csrc", "bam-readcount")
REF_CLI = os.path.join(ROOT, "oracle", "_ref", "bam-readcount-ref")
TOOL = os.path.join(ROOT, "tools", "e2e_configs.py")

needs_ref = pytest.mark.skipif(not os.path.exists(REF_CLI), reason="oracle/_ref/bam-readcount-ref is built where the reference checkout exists")


def leg(cli, *extra):
    out = subprocess.run([sys.executable, TOOL, "--cli", cli, "--reps", "1", "--procs", "4"] + list(extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    return json.loads(out.stdout.decode().strip().splitlines()[-1])


def check_sites(cli):
    j = leg(cli, "--leg", "sites", "--contigs", "5", "--contig-mbp", "0.3", "--sites", "400", "--check-lines", "150")
    v = j["validated"]
    assert v["byte_exact_vs_reference_main"] and v["found_in_order_in_full_output"] and v["full_line_count_equals_covered_positions"]
    assert v["lines_vs_reference_main"] >= 150 and j["site_lines"] > 400 and j["printed_lines"] > j["site_lines"]      # (the 20-base lines print 20 lines)
    assert any(s.startswith("sites:") for s in j["stages"])
    assert "independent" in j["cpu_reference_main"]["what"]
    # many small planner batches: the fetch of batch k + 1 runs behind batch k (cli.cpp: fetch_site_batch); same text, same checks
    k = leg(cli, "--leg", "sites", "--contigs", "5", "--contig-mbp", "0.3", "--sites", "400", "--check-lines", "150", "--", "--brc-plan", "37")
    assert k["validated"]["byte_exact_vs_reference_main"] and k["validated"]["full_output_md5"] == v["full_output_md5"]
    # one process per GPU (--brc-ranks; here: three ranks, on a GPU box all on device 0): every rank's slice of the list, in file order,
    # is what one process prints for it — the whole output byte for byte —, and the slices are balanced by the index's offsets
    r = leg(cli, "--leg", "sites", "--contigs", "5", "--contig-mbp", "0.3", "--sites", "400", "--check-lines", "150", "--ranks", "3", "--rank-devices", "0,0,0")
    sh = r["sharded"]
    assert r["validated"]["full_output_md5"] == v["full_output_md5"] and sh["whole_output_byte_identical_to_one_process"] and sh["output_lines"] == r["printed_lines"]
    assert len(sh["per_rank"]) == 3 and all(0.2 < pr["share_of_estimated_work"] < 0.47 and pr["weights"] == "index offsets" for pr in sh["per_rank"])


def check_tumor(cli):
    j = leg(cli, "--leg", "tumor", "--contig-mbp", "0.05", "--check-mbp", "0.02")
    v = j["validated"]
    assert v["byte_exact_vs_reference_main"] and v["full_line_count_equals_covered_positions"] and v["text_bytes_checked"] > 10_000_000
    assert j["events"] > 9_000_000
    r = leg(cli, "--leg", "tumor", "--contig-mbp", "0.2" if cli == HIP_CLI else "0.14", "--check-mbp", "0.01", "--ranks", "2", "--rank-devices", "0,0")      # (the CPU twin: three 64-kb atoms)
    sh = r["sharded"]
    assert sh["whole_output_byte_identical_to_one_process"] and sh["output_lines"] == r["printed_lines"] and len(sh["per_rank"]) == 2
    assert all(0.2 < pr["share_of_estimated_work"] < 0.8 for pr in sh["per_rank"])          # (four 64-kb atoms: as even as they allow)


@needs_ref
def test_config4_site_list_over_all_contigs_through_the_cli_cpu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    check_sites(SIM_CLI)


@needs_ref
def test_config5_per_library_region_of_a_multi_contig_bam_through_the_cli_cpu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "sim")])
    check_tumor(SIM_CLI)


@pytest.mark.gpu
@needs_ref
def test_config4_site_list_over_all_contigs_through_the_cli_gpu():
    check_sites(HIP_CLI)


@pytest.mark.gpu
@needs_ref
def test_config5_per_library_region_of_a_multi_contig_bam_through_the_cli_gpu():
    check_tumor(HIP_CLI)
