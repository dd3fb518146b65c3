cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp gsworld_amd/libgsr_hip.so /tmp/base.so
timeout 300 python -m pytest tests/test_forward_gpu.py -m gpu -x -q 2>&1 | tail -1
for v in base b2 b8; do
  if [ $v != base ]; then cp gsworld_amd/libgsr_$v.so gsworld_amd/libgsr_hip.so; else cp /tmp/base.so gsworld_amd/libgsr_hip.so; fi
  echo "== $v"; python tools/ab_render.py "4,6" 2>&1 | grep variant | sed 's/.*render=/render=/'
done
cp /tmp/base.so gsworld_amd/libgsr_hip.so
bash tools/sweep_bench.sh "1 6" "3 6" 2>&1 | tail -2
