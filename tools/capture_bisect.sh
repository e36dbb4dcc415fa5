#!/bin/bash
# GPU box: variants of the capture flow of the generic-primitive convs, each in its own process; prints which ones crash.
cd "$(dirname "$0")/.."
run() { local tag="$1"; shift; env "$@" timeout -k 5 120 python tools/capture_probe_convs.py GAT_plus step $SZ > gpurun_out/bis_$tag.log 2>&1; echo "$tag rc=$? $(tail -1 gpurun_out/bis_$tag.log | cut -c1-80)"; }
SZ="64 32"
run probe_default PROBE_LAYERS=4
run eager_first PROBE_LAYERS=4 PROBE_EAGER_FIRST=2
run warm1 PROBE_LAYERS=4 PROBE_WARM=1
run root PROBE_LAYERS=4 PROBE_ROOT=1
run zero_outside PROBE_LAYERS=4 PROBE_ZERO_OUTSIDE=1
run benchlike PROBE_LAYERS=4 PROBE_EAGER_FIRST=2 PROBE_WARM=1 PROBE_ROOT=1 PROBE_ZERO_OUTSIDE=1
SZ="16 4"
run benchlike_small PROBE_LAYERS=4 PROBE_EAGER_FIRST=2 PROBE_WARM=1 PROBE_ROOT=1 PROBE_ZERO_OUTSIDE=1
