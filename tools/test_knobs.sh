#!/bin/bash
# The GPU parity suite once per fallback path (every knob below changes scheduling or layout only; frames must not move).
# TRT_RUNTIME_WALK=1: the kernels that choose the walk at run time instead of the ones specialised at compile time.
# Since round 4 (ABI v3) the library reads these variables ONCE, when it is loaded, into the defaults that trt_tuning_default() /
# trt_scene_options_default() return - one pytest process per knob, as here, is exactly that.  (Inside a process the knobs are struct
# fields: tests/test_gpu_at_size.py::test_at_size_schedules_agree and friends set them per render.)
#
# Every knob's FULL output is kept (gpurun_out/knobs/<knob>.log), pytest runs under -X faulthandler (a host SIGSEGV / SIGABRT
# leaves the Python stack of every thread in the log), HIP runtime errors are logged (AMD_LOG_LEVEL=1) and a GPU memory fault
# leaves the runtime's own "Memory access fault by GPU node ..." line there too: one normal pass records WHAT a failure was.
# (Round 2's version piped pytest through `tail -1` and lost exactly that when `TRT_STREAM_MINW=8` dumped core once.)
# The script stops at the first knob that does not pass: after a fault, no further GPU work in the same call.
# If build/libtinyrt_cxxloops.so exists (make -C tiny-raytracer_amd/csrc cxxloops: the C++ box-step loops instead of the hand-written ones),
# the suite also runs once on that build.
out=gpurun_out/knobs; mkdir -p $out
[ -f build/libtinyrt_cxxloops.so ] && TRT_EXTRA_KNOBS="$TRT_EXTRA_KNOBS TRT_LIB_PATH=$PWD/build/libtinyrt_cxxloops.so"
skip='not leaf_slots_are and not lockstep and not global_memory_walks and not full_size_schedules and not full_baseline and not cfg5 and not at_size and not full_sample_count and not hbm_held'
for e in "TRT_RAY_POOL=0" "TRT_FLAT_WALK=0" "TRT_LDS_LEAF_STACK=0" "TRT_COMPACT_NODES=0" "TRT_STREAM_MINW=5" "TRT_STREAM_MINW=7" "TRT_STREAM_MINW=8" "TRT_LEAF_SLOTS=1" "TRT_LEAF_SLOTS=2" \
         "TRT_RUNTIME_WALK=1" "TRT_BIG_THREADS=512" "TRT_STRAGGLERS=0" "TRT_STRAGGLERS=40" "TRT_LDS_STRAGGLERS=0" "TRT_LDS_STRAGGLERS=24" "TRT_CULL_PRUNE=0.8" \
         "TRT_DUAL_WALK=1" "TRT_DUAL_WALK=2" "TRT_DUAL_WALK=1 TRT_STREAM_MINW=6" "TRT_DUAL_WALK=1 TRT_STREAM_MINW=4 TRT_STRAGGLERS=0" "TRT_STREAM_BATCH_SPP=3" "TRT_RADIANCE_GB=1" $TRT_EXTRA_KNOBS; do
  log=$out/$(echo "$e" | tr "/ " "__").log
  env $e AMD_LOG_LEVEL=1 PYTHONFAULTHANDLER=1 timeout -k 10 300 python3 -X faulthandler -m pytest tests -x -q -m gpu -k "$skip" > "$log" 2>&1
  rc=$?
  echo "== $e: rc=$rc $(tail -1 "$log")"
  if [ $rc -ne 0 ]; then
    echo "---- last 60 lines of $log ----"; tail -60 "$log"
    exit $rc
  fi
done
