#!/bin/bash
# the synchronous seam (and four regions in flight) at 30 workers over the number of dispatcher threads per device, uniform and mixed
# regions; usage: tools/r4_dispatchers.sh
cd "$GRAFT_REPO_ROOT"
one() { timeout -k 5 120 rovaca_amd/lib/adapter_bench 30 2000 48 9 0 $1 2>&1 | grep regions_per_s | sed 's/.*"regions_per_s": \([0-9]*\), "gcups_end_to_end": \([0-9.]*\).*"regions_per_submission": \([0-9.]*\).*/\1 regions\/s \2 GCUPS \3 per submission/'; }
for d in 2 3 4 5 6 8; do
  export ROVACA_AMD_PAIRHMM_DISPATCHERS=$d
  echo "dispatchers=$d sync:     $(one 1) | $(one 1)"
  echo "dispatchers=$d depth 4:  $(one 4)"
  echo "dispatchers=$d mixed sync: $(ADAPTER_BENCH_MIXED=1 one 1)"
done
