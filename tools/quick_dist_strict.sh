#!/bin/bash
# seconds for 30 iterations of the bit-identical multi-rank mode, 256^3 7-point as 8 processes on ONE GPU: composed sums vs the
# rank-by-rank chain of round 2 (tests/dist_fullsize_worker.py; the ranks share the GPU, so only the ratio means anything)
for mode in 1 0; do
  d=$(mktemp -d)
  for r in 0 1 2 3 4 5 6 7; do
    CGH_DIST_STRICT_COMPOSED=$mode timeout -k 10 200 python tests/dist_fullsize_worker.py $r 8 $d 256 256 256 7 0 30 0 > $d/out$r.txt 2>&1 &
  done
  wait
  echo "composed=$mode: $(grep strict: $d/out0.txt)"
  rm -rf $d
done
