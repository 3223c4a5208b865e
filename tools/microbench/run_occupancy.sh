#!/bin/bash
# macbench occupancy variants, run alternately (A B C D A B C D ...): G streams per wave x waves per SIMD the register allocation admits
cd tools/microbench/_bin
for rep in 1 2 3; do
  for b in macbench_g4w1 macbench_g2w4 macbench_g2w3 macbench_g2w1; do
    echo "== $b (pass $rep)"; TILED=1 timeout -k 10 120 ./$b 1 || exit 1
  done
done
