#!/bin/bash
O=gpurun_out/r5_lw16; mkdir -p $O
timeout 600 python3 scripts/exp/lean_waves.py > $O/stock.txt 2>&1; echo "stock rc=$?"
LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_lw16.so timeout 600 python3 scripts/exp/lean_waves.py > $O/lw16.txt 2>&1; echo "lw16 rc=$?"
paste -d'|' $O/stock.txt $O/lw16.txt | cut -c1-200
