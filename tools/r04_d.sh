set -e
O=gpurun_out/r04_d; mkdir -p $O
for m in 30 31; do python tools/dense_bench.py 4096 $m > $O/dense_bench_$m.txt 2>&1; head -2 $O/dense_bench_$m.txt | grep -v amdgpu; done
python tools/fuzz_dense.py 200 11 64 31,31a20,30a20 > $O/fuzz_seed11.txt 2>&1; tail -3 $O/fuzz_seed11.txt
python tools/fuzz_dense.py 200 12 64 pivoted,natural,30,31,32 > $O/fuzz_seed12.txt 2>&1; tail -5 $O/fuzz_seed12.txt
python tools/fuzz_dense.py 200 13 64 pivoted,natural,30,31,32 > $O/fuzz_seed13.txt 2>&1; tail -5 $O/fuzz_seed13.txt
