#!/bin/sh
# AddressSanitizer + UndefinedBehaviorSanitizer over the HOST-ONLY parts of the library (no GPU involved): the OpenFOAM and
# MatrixMarket readers on the committed fixtures, and the row-block partitioner on a stencil and a skewed matrix for
# 1-8 ranks.  Run from the repo root: sh tools/sanitize/run.sh
set -e
OUT=${TMPDIR:-/tmp}/cgh_sanitize
mkdir -p "$OUT"
gcc -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude tools/sanitize/parsers_main.c \
    conjugate_gradient_amd/csrc/foam_parser.c conjugate_gradient_amd/csrc/mm_reader.c -o "$OUT/parsers" -lm
"$OUT/parsers"
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -D_GLIBCXX_ASSERTIONS -Iinclude \
    tools/sanitize/partition_main.cpp conjugate_gradient_amd/csrc/partition.cpp -o "$OUT/partition"
"$OUT/partition"
echo "host sanitizer runs clean"
