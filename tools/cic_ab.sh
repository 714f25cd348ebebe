#!/bin/bash
# A/B of two builds of the sinc5 kernel on ONE box, alternated: the shipped libuchirp.so against a variant library
# (ultrasonic-communication_amd/libuchirp_ab.so, built by hand from csrc/uc_cic_kernel.hip with the -D under test).
# usage (GPU box): bash tools/cic_ab.sh <label of the variant>  -> gpurun_out/cic_ab.txt
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/cic_ab.txt
: > $out
for rep in 1 2 3; do
  echo "== shipped, pass $rep" >> $out
  timeout -k 10 200 python tools/run_cic.py 28 10 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-32s %7.1f GB/s  %.3f of copy  %s GHz' % (d['what'], d['GBs_algorithmic_median_of_3'], d['frac_of_copy_probe'], d['shader_ghz_in_kernel']))" >> $out || exit 1
  echo "== variant ($1), pass $rep" >> $out
  UCHIRP_LIB=$GRAFT_REPO_ROOT/ultrasonic-communication_amd/libuchirp_ab.so timeout -k 10 200 python tools/run_cic.py 28 10 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-32s %7.1f GB/s  %.3f of copy  %s GHz' % (d['what'], d['GBs_algorithmic_median_of_3'], d['frac_of_copy_probe'], d['shader_ghz_in_kernel']))" >> $out || exit 1
done
cat $out
