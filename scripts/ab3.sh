for v in "" _sw_noret _sw_nowait _sw_nofine _sw_none; do
  echo -n "lib$v: "; HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$v.so timeout 300 bash scripts/kstats.sh --steps 5 > /dev/null 2>&1; python scripts/ktimeline.py | grep -E "radix_sweep" | awk '{printf "%s ", $2} END {print ""}'
done
