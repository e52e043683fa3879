#!/bin/bash
# AddressSanitizer + UBSan build of the command line over the CPU lane simulator (test infrastructure): out/bam-readcount-asan
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"; O="${1:-/tmp/brc_asan}"; mkdir -p "$O"
F="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17"
g++ $F -fPIC -ffp-contract=off -shared "$R/tests/sim/brc_sim.cpp" "$R/bam_readcount_amd/csrc/brc_host.cpp" -DBRC_TEST_KNOBS "$R/bam_readcount_amd/csrc/brc_knobs.cpp" -o "$O/libbrc_sim.so" -pthread
g++ $F "$R/bam_readcount_amd/csrc/io/cli.cpp" "$R/bam_readcount_amd/csrc/io/bamio.cpp" "$R/bam_readcount_amd/csrc/io/cram.cpp" -o "$O/bam-readcount-asan" -L"$O" -lbrc_sim -lz -ldl -pthread -Wl,-rpath,"$O"
echo "$O/bam-readcount-asan"
