#!/bin/sh
# Host-side AddressSanitizer over the whole drop-in path on a real GPU: every .hip of the library compiled with
# -Xarch_host -fsanitize=address (device code untouched), linked with the C driver into build/asan/test_cg_asan.
# Build here (no GPU needed), then on the GPU box:
#   ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 CG_HIP_QUIET=1 \
#     ./build/asan/test_cg_asan tests/golden/cylinder_2000.system tests/golden/Phi_2000 --format ell|coo|csr
# Round 1: clean for all three formats (254 iterations, RMSE 0.003061 / max difference 0.004555 as the reference reports).
set -e
SRC=conjugate_gradient_amd/csrc
OUT=build/asan
mkdir -p $OUT
for f in $SRC/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -Xarch_host -fsanitize=address \
      -Xarch_host -fno-omit-frame-pointer -c $f -o $OUT/$(basename $f .hip).o
done
g++ -O1 -g -std=c++17 -fsanitize=address -fPIC -c $SRC/partition.cpp -o $OUT/partition.o
gcc -std=c99 -O1 -g -fsanitize=address -c $SRC/foam_parser.c -o $OUT/foam_parser.o
gcc -std=c99 -O1 -g -fsanitize=address -c $SRC/mm_reader.c -o $OUT/mm_reader.o
gcc -std=c99 -D_POSIX_C_SOURCE=199309L -O1 -g -fsanitize=address -c conjugate_gradient_amd/host/test_cg.c -o $OUT/test_cg.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -o $OUT/test_cg_asan $OUT/*.o -Wl,-rpath,/opt/rocm/lib -ldl -lm
echo "built $OUT/test_cg_asan"
