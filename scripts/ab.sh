# usage: bash scripts/ab.sh variant [variant ...]   -- parity subset on the first variant, then render-stage timing of all
V1=$1
HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat_$V1.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ldr_forward_backward or (golden and not antialias) or motion_blur or edge_cases or determinism" 2>&1 | grep -E "^E  |passed|failed" | head -8
for v in "" "$@" ""; do
  s=${v:+_$v}
  HS_LIB_PATH=$PWD/casualhdrsplat_amd/libhdrsplat$s.so timeout 300 python scripts/ab_render.py --iters 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'c3 bwd', d['render_bwd_ms'], d['render_bwd_med'], 'fwd', d['render_fwd_ms'], 'step', d['step_ms'])"
done
