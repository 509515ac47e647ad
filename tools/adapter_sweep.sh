# adapter_sweep.sh -- the drop-in adapter's throughput over dispatcher counts, worker counts and regions in flight (run on the GPU box)
# usage: tools/adapter_sweep.sh [regions per thread]   env: SW=1 adds the realignment stage
N=${1:-2000}
for disp in 2 3 4 6; do for d in 1 4; do for th in 8 30 64; do
  echo -n "dispatchers $disp threads $th depth $d: "
  ROVACA_AMD_PAIRHMM_DISPATCHERS=$disp timeout -k 5 120 rovaca_amd/lib/adapter_bench $th $N 48 9 ${SW:-0} $d | grep -o '"regions_per_s": [0-9]*\|"gcups_end_to_end": [0-9.]*\|"regions_per_submission": [0-9.]*\|"cpu_us_per_region": [0-9.]*' | tr '\n' ' '; echo
done; done; done
