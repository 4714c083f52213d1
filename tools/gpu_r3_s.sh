export TMPDIR=/tmp
V=protocols/distributed_keygen_amd/build/variants
FR_B=2048 MX_LIBRARY=$PWD/$V/base.so timeout 800 python tools/fr_check.py 2>&1 | tail -14; FR_B=2048 MX_LIBRARY=$PWD/$V/fr1w.so timeout 800 python tools/fr_check.py 2>&1 | tail -14
