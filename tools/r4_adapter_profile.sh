#!/bin/bash
# the drop-in adapter over worker counts, synchronous seam and four regions in flight, with the host phase profile
# (ADAPTER_BENCH_PHASES=1): the round-4 counterpart of profiles/r3_adapter_host_profile.txt.  usage: tools/r4_adapter_profile.sh > out.txt
cd "$GRAFT_REPO_ROOT"
export ADAPTER_BENCH_PHASES=1
for d in 1 4; do for th in 1 8 30 64 128; do
  echo "=== $th workers, $d region(s) in flight per worker"
  timeout -k 5 120 rovaca_amd/lib/adapter_bench $th 2000 48 9 0 $d 2>&1 | tail -16
done; done
echo "=== regions of mixed shape (ADAPTER_BENCH_MIXED=1)"
for d in 1 4; do for th in 30 64; do
  echo "--- $th workers, $d in flight"
  ADAPTER_BENCH_MIXED=1 timeout -k 5 120 rovaca_amd/lib/adapter_bench $th 1000 48 9 0 $d 2>&1 | tail -16
done; done
