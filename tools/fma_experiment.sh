#!/bin/bash
# On the GPU box (round 5): what the FMA lever is worth ON THE CHIP, and what it costs in parity.  The variant library
# (python tools/build_variants.py small fma=-DFX_EXP_FMA_TWIDDLES: every twiddle product's second multiply fused into the sum, two
# VOP3P instructions instead of three) against the shipped one: kernel time A/B interleaved, VALU instructions per frame, and the
# stress run's verdict on the variant.  The variant is selected by path (FX_LIBRARY_OVERRIDE, feature-extractor_amd/capi.py): the shipped
# library is never overwritten, so a killed run leaves nothing to restore.
set -u
L=$(pwd)/feature-extractor_amd/lib
pick() { if [ "$1" = base ]; then unset FX_LIBRARY_OVERRIDE; else export FX_LIBRARY_OVERRIDE=$L/variants/$1.so; fi; }
for round in 1 2 3; do
  for v in base fma; do
    pick $v
    echo -n "$v: "; timeout -k 10 120 python3 tools/window_timing.py 1024 1024 512 2>&1 | tail -1
  done
done
for v in base fma; do
  pick $v
  echo -n "$v: "; timeout -k 10 300 python3 tools/pmc_quick.py 1024 1024 512 $v 2>&1 | tail -1
done
pick fma
echo "stress on the FMA variant (1024-point cases only):"
timeout -k 10 400 python3 tools/stress_parity.py 240 61 1024 2>&1 | grep -v "^MISMATCH" | tail -4
echo "first mismatches:"
timeout -k 10 200 python3 tools/stress_parity.py 60 62 1024 2>&1 | grep "^MISMATCH" | head -12
