for v in "" _atomic; do
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$v.so python scripts/ab_render.py --iters 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'render_bwd', d['render_bwd_ms'], d['render_bwd_med'])"
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$v.so python scripts/ab_render.py --config c4 --iters 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'c4 render_bwd', d['render_bwd_ms'], d['render_bwd_med'])"
done
