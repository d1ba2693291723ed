#!/bin/bash
# Builds the UNMODIFIED reference RAL demo (ral/test.cpp + ral/l1_irls.cpp of ajparra/iRotAvg) with
# its own CMake project and runs it on the cases of tools/ref_golden/inputs/. Needs what the
# reference needs: Eigen >= 3.3 and SuiteSparse (CHOLMOD, SPQR, UMFPACK) -- neither exists in the
# build image of this repository, so this script is for a maintainer's machine.
#
#   python3 tools/ref_golden/make_cases.py
#   bash tools/ref_golden/run_reference.sh /path/to/iRotAvg
#
# Outputs: tests/golden/ref_<case>.out (the reference's l1_irls_out.txt: n rotation lines `w x y z`,
# then m IRLS weights, Eigen::FullPrecision) and tests/golden/ref_<case>.log (its stdout). Commit
# them; tests/test_ref_golden.py then pins the oracle (and through it the GPU path) to the reference.
set -euo pipefail
REF=${1:?usage: run_reference.sh /path/to/iRotAvg}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
BUILD=${BUILD_DIR:-$HERE/build}
cmake -S "$REF/ral" -B "$BUILD" -DCMAKE_BUILD_TYPE=Release
cmake --build "$BUILD" -j
BIN="$BUILD/l1_irls"
while IFS=$'\t' read -r name input args; do
    echo "== $name: $args"
    # shellcheck disable=SC2086
    "$BIN" "$HERE/inputs/$input" "$ROOT/tests/golden/ref_$name.out" $args > "$ROOT/tests/golden/ref_$name.log"
done < "$HERE/inputs/cases.tsv"
ls -l "$ROOT"/tests/golden/ref_*.out
