set -x
mkdir -p gpurun_out/r02_base
O=gpurun_out/r02_base
python bench.py --cpu-sample 0 > $O/bench.json 2> $O/bench.err
python bench.py --cpu-sample 0 --pipeline 1 --steps 6 > $O/bench_serial.json 2>> $O/bench.err
FBSTAB_HIP_LIB=fbstab_amd/var_stamp.so python tools/stamp_report.py 8192 > $O/stamp_8192.txt 2>&1
FBSTAB_HIP_LIB=fbstab_amd/var_stamp.so python tools/stamp_report.py 4 > $O/stamp_4.txt 2>&1
FBSTAB_HIP_LIB=fbstab_amd/var_clock.so python tools/stamp_report.py 8192 > $O/clock_8192.txt 2>&1
FBSTAB_HIP_LIB=fbstab_amd/var_clock.so python tools/stamp_report.py 32768 > $O/clock_32768.txt 2>&1
python tools/dense_bench.py > $O/dense.txt 2>&1
