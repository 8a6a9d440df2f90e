HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat_pg.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ldr_forward_backward or (golden and not antialias) or motion_blur or edge_cases" 2>&1 | tail -4
for v in "" _pg; do
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$v.so timeout 300 python scripts/ab_render.py --iters 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'c3 render_bwd', d['render_bwd_ms'], d['render_bwd_med'])"
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$v.so timeout 300 python scripts/ab_render.py --config c4 --iters 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'c4 render_bwd', d['render_bwd_ms'])"
done
