set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_raster.py -x -q -m gpu > gpurun_out/pitch_tests.log 2>&1 || { tail -30 gpurun_out/pitch_tests.log; exit 1; }
tail -3 gpurun_out/pitch_tests.log
for i in 1 2 3; do
  for p in 0 1; do
    echo "PITCH=$p f64" >> gpurun_out/pitch_kbench.log
    MOD16_PITCH=$p timeout -k 10 200 python tools/kbench.py --rows 21600 --launches 10 --rounds 3 >> gpurun_out/pitch_kbench.log 2>&1
  done
done
for i in 1 2; do
  for p in 0 1; do
    echo "PITCH=$p f32" >> gpurun_out/pitch_kbench.log
    MOD16_PITCH=$p timeout -k 10 200 python tools/kbench.py --rows 21600 --launches 10 --rounds 3 --dtype float32 >> gpurun_out/pitch_kbench.log 2>&1
  done
done
cat gpurun_out/pitch_kbench.log
