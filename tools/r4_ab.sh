#!/bin/bash
# Round-4 A/B on one GPU box: ragged launches (default) against one launch per class (PHMM_NO_RAGGED) and against round 3's library
# (rovaca_amd/lib/variants/lib_r3.so, built from the round-3 commit).  Outputs under gpurun_out/r4_ab/.
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_ab; mkdir -p $OUT
R3=$PWD/rovaca_amd/lib/variants/lib_r3.so
B="python3 bench.py --no-cpu-baseline --no-adapter --steps 20 --warmup 5"
echo "== c2 new";  $B > $OUT/c2_new.json 2> $OUT/c2_new.err; cut -c1-400 $OUT/c2_new.json
if [ -f $R3 ]; then echo "== c2 r3"; ROVACA_AMD_HIP_LIB=$R3 $B > $OUT/c2_r3.json 2> $OUT/c2_r3.err; cut -c1-400 $OUT/c2_r3.json; fi
echo "== c2 new again"; $B > $OUT/c2_new2.json 2> $OUT/c2_new2.err; cut -c1-400 $OUT/c2_new2.json
echo "== by K new"; python3 tools/bench_by_k.py 4,6,8,9,10,11,12,14,16 > $OUT/byk_new.txt 2>&1; cat $OUT/byk_new.txt
if [ -f $R3 ]; then echo "== by K r3"; ROVACA_AMD_HIP_LIB=$R3 python3 tools/bench_by_k.py 4,6,8,9,10,11,12,14,16 > $OUT/byk_r3.txt 2>&1; cat $OUT/byk_r3.txt; fi
echo "== c3 ragged"; $B --config c3 > $OUT/c3_ragged.json 2> $OUT/c3_ragged.err; cut -c1-300 $OUT/c3_ragged.json; grep -o '"e2e": {[^}]*}' $OUT/c3_ragged.json
echo "== c3 per-class (PHMM_NO_RAGGED)"; PHMM_ENABLE_TEST_HOOKS=1 PHMM_NO_RAGGED=1 $B --config c3 > $OUT/c3_perclass.json 2> $OUT/c3_perclass.err; cut -c1-300 $OUT/c3_perclass.json; grep -o '"e2e": {[^}]*}' $OUT/c3_perclass.json
if [ -f $R3 ]; then echo "== c3 r3"; ROVACA_AMD_HIP_LIB=$R3 $B --config c3 > $OUT/c3_r3.json 2> $OUT/c3_r3.err; cut -c1-300 $OUT/c3_r3.json; grep -o '"e2e": {[^}]*}' $OUT/c3_r3.json; fi
echo "== adapter (default bench, ragged)"; python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $OUT/full_ragged.json 2> $OUT/full_ragged.err; grep -o '"adapter": {.*' $OUT/full_ragged.json | cut -c1-1500
echo "== adapter (PHMM_NO_RAGGED)"; PHMM_ENABLE_TEST_HOOKS=1 PHMM_NO_RAGGED=1 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $OUT/full_perclass.json 2> $OUT/full_perclass.err; grep -o '"mixed_shapes": {.*' $OUT/full_perclass.json | cut -c1-600
