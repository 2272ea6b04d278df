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
    # exitcode=99: a sanitizer report (or a failed sanitizer CHECK) cannot be mistaken for a decoder status (erreur_code_t values and 1)
    ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:protect_shadow_gap=0:exitcode=99 timeout -k 5 60 tools/jpeg2ppm_gpu_asan -o $OUT/x.pnm $f > $OUT/log.txt 2>&1
    rc=$?
    if grep -q "AddressSanitizer\|runtime error" $OUT/log.txt || [ $rc -ge 99 ]; then bad=$((bad+1)); echo "== $f (step $step) rc=$rc"; grep -m3 "AddressSanitizer\|runtime error\|SUMMARY" $OUT/log.txt; fi
    rm -f $OUT/x*.pnm $OUT/x*
  done
done
# the host-pointer entries (stripes, helper threads, staging rings), default and tiny stripes with a small device ring
if [ -x tools/host_pipe_asan ]; then
  for sb in 0 300000; do
    n=$((n+1))
    if [ $sb != 0 ]; then export JPEGGPU_STRIPE_BYTES=$sb JPEGGPU_DEVICE_STAGING_BYTES=2000000; else unset JPEGGPU_STRIPE_BYTES JPEGGPU_DEVICE_STAGING_BYTES; fi
    ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:protect_shadow_gap=0:exitcode=99 timeout -k 5 200 tools/host_pipe_asan > $OUT/log.txt 2>&1
    rc=$?
    if [ $rc != 0 ] || grep -q "AddressSanitizer\|runtime error\|MISMATCH" $OUT/log.txt; then bad=$((bad+1)); echo "== host_pipe_asan (stripe $sb) rc=$rc"; grep -m5 "AddressSanitizer\|runtime error\|SUMMARY\|MISMATCH" $OUT/log.txt; fi
  done
  unset JPEGGPU_STRIPE_BYTES JPEGGPU_DEVICE_STAGING_BYTES
fi
echo "asan_cli: $n runs, $bad with findings"
[ $bad = 0 ]
