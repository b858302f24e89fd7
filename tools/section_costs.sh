#!/bin/bash
# Dynamic cost of each section of fx_frame_kernel<N>: builds that end a frame's work at stop point k (-DFX_EXP_STOP_AT=k, see
# fx_frame_kernel.hip.h), tools/pmc_quick.py per build; the differences between consecutive lines are the sections.
#   here   : python tools/build_variants.py small $(for k in $(seq 1 11); do echo stop$k=-DFX_EXP_STOP_AT=$k; done)
#   GPU box: tools/section_costs.sh [N C T]            (variants are selected by path, FX_LIBRARY_OVERRIDE: the shipped library is never touched)
N=${1:-1024}; C=${2:-1024}; T=${3:-512}
L=$(pwd)/feature-extractor_amd/lib
names=(- load+rms lpf pitch_fft ifft scan spec_fft spec_sums flux flatprod spec_pass2 harm1 harm2)
for k in 1 2 3 4 5 6 7 8 9 10 11 12; do
  if [ $k = 12 ]; then unset FX_LIBRARY_OVERRIDE; else [ -f $L/variants/stop$k.so ] || continue; export FX_LIBRARY_OVERRIDE=$L/variants/stop$k.so; fi
  timeout -k 10 120 python3 tools/pmc_quick.py $N $C $T "[through ${names[$k]}]" 2>&1 | tail -1
done
