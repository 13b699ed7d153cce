# rocprofv3 kernel stats of tools/variantbench.py (the pipeline forms and the plain kernels)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/variant_stats
rm -rf $O && mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/f64 -- python3 tools/variantbench.py 10800 float64 > $O/f64.out 2> $O/f64.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/f32mixed -- python3 tools/variantbench.py 10800 float32 mixed > $O/f32mixed.out 2> $O/f32mixed.err
for t in f64 f32mixed; do cp $(find $O/$t -name "*kernel_stats.csv" | head -1) $O/${t}_kernel_stats.csv; grep -v amdgpu $O/$t.out > $O/${t}_variantbench.jsonl; done
find $O -name "*.csv" -size +1M -delete
cut -c1-140 $O/f64_kernel_stats.csv | head -16
