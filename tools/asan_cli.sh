#!/bin/bash
# Host AddressSanitizer over the whole library (tools/jpeg2ppm_gpu_asan = jpeg2ppm_gpu.cpp + every library source, built with
# hipcc ... -Xarch_host -fsanitize=address; device code is not instrumented): every fixture through jpeggpu_jpeg2ppm on the GPU,
# default steps and steps of one MCU row.  Prints the files on which the sanitizer reported something.
OUT=${1:-$(mktemp -d)}
mkdir -p $OUT
bad=0; n=0
for step in 0 1; do
  for f in tests/golden/jpeg/*.jp*g; do
    n=$((n+1))
    if [ $step = 1 ]; then export JPEGGPU_STRIPE_BYTES=1; else unset JPEGGPU_STRIPE_BYTES; fi
    ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:protect_shadow_gap=0 timeout -k 5 60 tools/jpeg2ppm_gpu_asan -o $OUT/x.pnm $f > $OUT/log.txt 2>&1
    rc=$?
    if grep -q "AddressSanitizer\|runtime error" $OUT/log.txt || [ $rc -ge 124 ]; then bad=$((bad+1)); echo "== $f (step $step) rc=$rc"; grep -m3 "AddressSanitizer\|runtime error\|SUMMARY" $OUT/log.txt; fi
    rm -f $OUT/x*.pnm $OUT/x*
  done
done
echo "asan_cli: $n runs, $bad with findings"
[ $bad = 0 ]
