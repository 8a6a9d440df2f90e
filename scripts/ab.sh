for v in "" _s8 _s32 "" _s8 _s32; do
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$v.so python scripts/ab_render.py --iters 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], d['binning_ms'], d['step_med'])"
done
