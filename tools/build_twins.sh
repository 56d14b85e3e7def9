#!/bin/bash
# Builds one twin library per experiment: libatlaspatch_hip_twin_<tag>.so = product objects + gemm256 compiled with
# -DAP_G256_ALT <flags> as impl 257 (tools/gemm_twin_ab.py A/Bs it against the product kernel in one process).
#   tools/build_twins.sh tag1="-DAP_EXP_A" tag2="-DAP_EXP_A -DAP_EXP_B" ...
set -e
cd "$(dirname "$0")/../atlaspatch_amd/csrc"
make -j8 >/dev/null
for spec in "$@"; do
    tag="${spec%%=*}"; flags="${spec#*=}"
    rm -f build/gemm256_alt.o
    make twin ALT_FLAGS="-DAP_G256_ALT $flags" TWIN_OUT="../libatlaspatch_hip_twin_${tag}.so" 2>&1 | grep -E "error|warning: v|spill" || true
    echo "built libatlaspatch_hip_twin_${tag}.so ($flags)"
done
rm -f build/gemm256_alt.o
