#!/bin/bash
export MVAE_TUNING=1   # schedule knobs / MVAE_LIB are honoured only under this switch
# Same-box A/B of two builds of the library (device-to-device spread is +-2 %, larger than most single changes): alternates
#   MVAE_LIB=<a> / <b> over whole-step timings.   bash tests/tuning/ab_libs.sh libA.so libB.so "1024 128" [rounds]
set -uo pipefail
A=$1; B=$2; BATCHES=${3:-1024}; R=${4:-3}
for b in $BATCHES; do
  for r in $(seq $R); do
    for lib in $A $B; do
      echo -n "B=$b $(basename $lib): "
      MVAE_LIB=$(pwd)/$lib timeout -k 10 300 python3 tests/ab_step.py --batch $b --rounds 2 --steps 10 2>&1 | tail -2 | tr '\n' ' '; echo
    done
  done
done
