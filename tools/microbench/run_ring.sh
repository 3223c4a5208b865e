#!/bin/bash
# LDS-DMA ring variants of the dyadic multiply-accumulate against k_mac (same jobs, bit-compared)
cd "$(dirname "$0")/_bin"
for b in macbench_ring_d3_t256 macbench_ring_d4_t128 macbench_ring_d2_t256 macbench_ring_d4_t256 macbench_ring_d5_t128 macbench_ring_d3_t128; do
  echo "== $b"; RING=1 timeout -k 10 120 ./$b 1 || exit 1
done
