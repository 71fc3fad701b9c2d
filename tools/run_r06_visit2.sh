#!/bin/bash
# round 6, second visit: overlap layouts with the socket power of THIS GPU beside, XCD / CU partitions, the 384 x 128 kernel under overlap
set -x
mkdir -p gpurun_out
python -c "
import sys; sys.path.insert(0,'.')
from prisma_amd import _lib; _lib.load()
from prisma_amd import power; print(power._pci_bdf(0), power._hwmon_files(0))" > gpurun_out/r06b_hwmon.txt 2>&1
python tools/overlap_bench.py --steps 6 --layouts seq,2way,flow1,flow2,depth1,depth2,2way-xcd,2way-cu,2way > gpurun_out/r06b_overlap.txt 2>&1
tail -1 gpurun_out/r06b_overlap.txt
PB_TILE_N128=1 python tools/overlap_bench.py --steps 6 --layouts seq,2way > gpurun_out/r06b_overlap_n128.txt 2>&1
tail -1 gpurun_out/r06b_overlap_n128.txt
