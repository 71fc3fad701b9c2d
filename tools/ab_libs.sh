#!/bin/bash
# same-box A/B of two builds: bash tools/ab_libs.sh <lib A> <lib B>   (kernel ms per step of the flow and depth bands, split mode)
for lib in "$@"; do
  for s in flow depth; do
    PRISMA_BANDS_LIB=$PWD/$lib AB_PREC=1 python tools/ab_$s.py 2>&1 | tail -1 | cut -c1-1100 | sed "s|^|[$lib $s] |"
  done
done
