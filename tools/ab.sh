set -e
python -m pytest tests/test_gpu_tables.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -3 gpurun_out/ab_tests.log
for lib in "" "--lib tools/liblfmcmc_alt.so"; do
  echo "=== lib: $lib"
  python tools/time_parts.py --nsrc 1000000 --rows 128 --sets "default;skip_grid=1;cells=0;cells=0,skip_grid=1" $lib
  python tools/time_parts.py --nsrc 1000000 --rows 256 --sets "default" $lib
  python tools/time_parts.py --nsrc 100000 --rows 128 --sets "default;cells=0" $lib
  python tools/time_parts.py --nsrc 1000000 --rows 128 --variant zevol --sets "default" $lib
  python tools/time_parts.py --nsrc 1000 --rows 128 --sets "default" $lib
done
