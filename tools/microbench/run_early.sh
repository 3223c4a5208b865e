#!/bin/bash
# k_mac with the unpack-first schedule (-DAPSU_MAC_EARLY) against the library's loop: binaries run alternately, many processes (the launch
# time depends on where the process's database lands, profiles/r05_mac_placement.txt, so single pairs mean nothing)
cd "$(dirname "$0")/_bin"
for rep in 1 2 3 4 5 6 7 8; do
  for b in macbench_r2 macbench_early; do
    echo "== $b (pass $rep)"; TILED=1 timeout -k 10 120 ./$b 1 | grep "ring=\|packed 56 bits         rows" || exit 1
  done
done
