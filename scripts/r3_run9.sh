#!/bin/bash
set -u
echo "== one-list kernel on the two failing cases"
HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat_onelist.so python scripts/sweep_case.py 24 9 2>&1 | grep -E "^case|means3D|row 901" | head -4
HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat_onelist.so python scripts/sweep_case.py 22 156 2>&1 | grep -A3 -E "^case|means2D" | head -8
echo "== two-group kernel"
python scripts/sweep_case.py 22 156 2>&1 | grep -A3 -E "^case|means2D" | head -8
