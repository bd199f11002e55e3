#!/bin/bash
# debug: scoped profile of the lane-0 procedures. usage: evprof3.sh c2,c3 [fused|step] [min ticks of a recorded step]
# (SSS_SECTIONS=rel / fast: section timers inside batch_released_events / fast_run instead). The timing build is a test build
# of the library (tests/gpu_variant.py -> tests/_build/); the product library is not touched.
cd "$(dirname "$0")/../.."
python tools/debug/evprof3.py "$@" 2>&1 | grep -v amdgpu.ids
