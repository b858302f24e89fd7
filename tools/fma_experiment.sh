#!/bin/bash
# On the GPU box (round 5): what the FMA lever is worth ON THE CHIP, and what it costs in parity.  The variant library
# (python tools/build_variants.py small fma=-DFX_EXP_FMA_TWIDDLES: every twiddle product's second multiply fused into the sum, two
# VOP3P instructions instead of three) against the shipped one: kernel time A/B interleaved, VALU instructions per frame, and the
# stress run's verdict on the variant.  The shipped library is back in place when the script ends, whatever happens.
set -u
L=feature-extractor_amd/lib
cp $L/libfx_hip.so $L/variants/base.so
cp $L/libfx_hip.so $L/variants/_shipped.so
trap 'cp $L/variants/_shipped.so $L/libfx_hip.so' EXIT
for round in 1 2 3; do
  for v in base fma; do
    cp $L/variants/$v.so $L/libfx_hip.so
    echo -n "$v: "; timeout -k 10 120 python3 tools/window_timing.py 1024 1024 512 2>&1 | tail -1
  done
done
for v in base fma; do
  cp $L/variants/$v.so $L/libfx_hip.so
  echo -n "$v: "; timeout -k 10 300 python3 tools/pmc_quick.py 1024 1024 512 $v 2>&1 | tail -1
done
cp $L/variants/fma.so $L/libfx_hip.so
echo "stress on the FMA variant (1024-point cases only):"
timeout -k 10 400 python3 tools/stress_parity.py 240 61 1024 2>&1 | grep -v "^MISMATCH" | tail -4
echo "first mismatches:"
timeout -k 10 200 python3 tools/stress_parity.py 60 62 1024 2>&1 | grep "^MISMATCH" | head -12
