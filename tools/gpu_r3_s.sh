# round 3, call S: mismatch census of the library as shipped, single launches and four streams at once
export TMPDIR=/tmp
O=gpurun_out/r03s; mkdir -p $O
FR_B=10000 timeout 900 python tools/fr_check.py > $O/census_10000.txt 2>&1; grep -v amdgpu.ids $O/census_10000.txt
FR_B=1500 timeout 600 python tools/fr_check.py > $O/census_1500.txt 2>&1; grep -v amdgpu.ids $O/census_1500.txt
