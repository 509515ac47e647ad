#!/bin/bash
# the synchronous seam at 30 workers, several runs in a row (each a fresh process): how much the rate moves from run to run
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4 5 6 7 8; do
  ADAPTER_BENCH_PHASES=1 timeout -k 5 120 rovaca_amd/lib/adapter_bench 30 2000 48 9 0 1 2>&1 | grep -E "phmm_stage_wait|regions_per_s" | sed 's/.*phmm_stage_wait (device) *[0-9.]* s *\([0-9.]*\) us.*/  device_wait \1 us\/region/; s/{"threads".*"regions_per_s": \([0-9]*\), "gcups_end_to_end": \([0-9.]*\).*"regions_per_submission": \([0-9.]*\).*/  regions_per_s \1 gcups \2 regions_per_submission \3/' | tr '\n' ' '; echo
done
