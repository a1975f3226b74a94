#!/bin/bash
# softmax.PV for ONE latent group (BASELINE config 5's per-GPU slice) under the split-count knobs of an experiments build:
#   PALU_EXTRA_CFLAGS=-DPALU_EXPERIMENTS python -m palu_amd.build --force && bash tools/g1_pv_sweep.sh [L]
L=${1:-262145}
export PYTHONPATH=${PYTHONPATH:-.}
for w in 1 2 3 4 6 8; do
  echo "== PALU_PV_WGS_PER_CU=$w"; PALU_PV_WGS_PER_CU=$w python tools/time_g1_pieces.py $L | grep softmax
done
for w in 1 2; do
  echo "== PALU_PV_DIRECT=1 PALU_PVQ_WGS=$w"; PALU_PV_DIRECT=1 PALU_PVQ_WGS=$w python tools/time_g1_pieces.py $L | grep softmax
done
