# sw_adapter_sweep.sh -- PairHMM + realignment through the adapter over Smith-Waterman dispatcher counts (run on the GPU box)
N=${1:-1000}
G='"regions_per_s": [0-9]*\|"gcups_end_to_end": [0-9.]*\|"sw_gcups_end_to_end": [0-9.]*\|"cpu_us_per_region": [0-9.]*'
for disp in 1 2 3; do for d in 1 4; do for th in 1 8 30 128; do
  echo -n "sw dispatchers $disp threads $th depth $d: "
  ROVACA_AMD_SW_DISPATCHERS=$disp timeout -k 5 120 rovaca_amd/lib/adapter_bench $th $N 48 9 1 $d | grep -o "$G" | tr '\n' ' '; echo
done; done; done
