set -x
O=gpurun_out/r02_$1; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -n 4 $O/pytest.txt
for v in g16 g32; do
FBSTAB_HIP_LIB=$PWD/fbstab_amd/var_$v.so python tools/shape_bench.py 30 20 6 16 2048 > $O/shape_$v.txt 2>&1; tail -n 1 $O/shape_$v.txt
done
FBSTAB_HIP_LIB=$PWD/fbstab_amd/var_g32.so python tools/shape_bench.py 30 20 6 30 2048 > $O/shape_g32_nc30.txt 2>&1; tail -n 1 $O/shape_g32_nc30.txt
FBSTAB_HIP_LIB=$PWD/fbstab_amd/var_g32.so python tools/shape_bench.py 20 24 8 32 2048 > $O/shape_g32_full.txt 2>&1; tail -n 1 $O/shape_g32_full.txt
FBSTAB_HIP_LIB=$PWD/fbstab_amd/var_g32.so python tools/shape_bench.py 40 14 2 8 2048 > $O/shape_g32_small.txt 2>&1; tail -n 1 $O/shape_g32_small.txt
python tools/shape_bench.py 30 20 6 30 2048 > $O/shape_flat_nc30.txt 2>&1; tail -n 1 $O/shape_flat_nc30.txt
python tools/shape_bench.py 40 14 2 8 2048 > $O/shape_flat_small.txt 2>&1; tail -n 1 $O/shape_flat_small.txt
python tools/variant_bench.py 8192 2 2>&1 | tail -n 1
