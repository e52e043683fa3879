#!/bin/bash
# ThreadSanitizer build of the command line over the CPU lane simulator (test infrastructure): out/bam-readcount-tsan.
# Round 5: striped BAM and CRAM fetches (every CRAM file of tests/test_cli.py's fixture), the region pipeline with -p -i, a site list with the
# planner, two engines, a small formatter chunk — no data race (the one report per run is the background thread the fast exit does not join).
#   TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" BRC_FETCH_STRIPE_MIN=1 BRC_FETCH_THREADS=5 out/bam-readcount-tsan -w 0 -p -f syn.fa --brc-chunk 900 syn_rans.cram chrA
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"; O="${1:-/tmp/brc_tsan}"; mkdir -p "$O"
F="-O1 -g -fsanitize=thread -fno-omit-frame-pointer -std=c++17"
g++ $F -fPIC -ffp-contract=off -shared "$R/tests/sim/brc_sim.cpp" "$R/bam_readcount_amd/csrc/brc_host.cpp" -DBRC_TEST_KNOBS "$R/bam_readcount_amd/csrc/brc_knobs.cpp" -o "$O/libbrc_sim.so" -pthread
g++ $F "$R/bam_readcount_amd/csrc/io/cli.cpp" "$R/bam_readcount_amd/csrc/io/bamio.cpp" "$R/bam_readcount_amd/csrc/io/cram.cpp" -o "$O/bam-readcount-tsan" -L"$O" -lbrc_sim -lz -ldl -pthread -Wl,-rpath,"$O"
echo "$O/bam-readcount-tsan"
