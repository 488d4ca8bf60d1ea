#!/bin/bash
# S2D alone, same box: the matrix-core form (default) against the fp32 form (KBN_S2D_DEBUG=32), batch 8 and 32, KITTI and VOID
for preset in kitti void; do
for b in 8 32; do
for dbg in 0 32 0 32; do
  S2D_PRESET=$preset S2D_BATCH=$b KBN_S2D_DEBUG=$dbg python tools/s2d_bench.py --one | sed "s/^/$preset batch $b dbg $dbg /"
done; done; done
