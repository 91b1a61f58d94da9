#!/bin/bash
# The GPU parity suite once per fallback path (every knob below changes scheduling or layout only; frames must not move).
# TRT_RUNTIME_WALK=1: the kernels that choose the walk at run time instead of the ones specialised at compile time.
for e in "TRT_RAY_POOL=0" "TRT_FLAT_WALK=0" "TRT_LDS_LEAF_STACK=0" "TRT_COMPACT_NODES=0" "TRT_STREAM_MINW=5" "TRT_STREAM_MINW=7" "TRT_STREAM_MINW=8" "TRT_LEAF_SLOTS=1" "TRT_LEAF_SLOTS=2" \
         "TRT_ORDERED_WALK=1" "TRT_RUNTIME_WALK=1" "TRT_BIG_THREADS=512" "TRT_STRAGGLERS=0" "TRT_STRAGGLERS=40"; do
  echo "== $e: $(env $e timeout -k 10 300 python -m pytest tests -x -q -m gpu -k 'not leaf_slots_are and not lockstep and not global_memory_walks and not full_size_schedules and not full_baseline and not cfg5' 2>&1 | tail -1)"
done
