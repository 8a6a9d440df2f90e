HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat_persist.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "ldr_forward_backward or golden or motion_blur" 2>&1 | tail -3
for v in "" _persist "" _persist; do
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$v.so timeout 300 python scripts/ab_render.py --iters 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'render_bwd', d['render_bwd_ms'], d['render_bwd_med'], 'step', d['step_med'])"
done
HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat_persist.so timeout 300 python scripts/ab_render.py --config c4 --iters 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'c4 render_bwd', d['render_bwd_ms'])"
HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat.so timeout 300 python scripts/ab_render.py --config c4 --iters 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'c4 render_bwd', d['render_bwd_ms'])"
