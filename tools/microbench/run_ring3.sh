#!/bin/bash
# k_mac with three register sets (two terms in flight, loads issued as one group behind a scheduling barrier) against the library's two-set loop
cd "$(dirname "$0")/_bin"
for rep in 1 2 3; do
  for b in macbench_r2 macbench_r3 macbench_r3k; do
    echo "== $b (pass $rep)"; TILED=1 timeout -k 10 120 ./$b 1 || exit 1
  done
done
