for v in "" _l8 _e24; do
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$v.so python scripts/ab_render.py --iters 30 2>/dev/null | tail -1
done
