#!/bin/bash
# One MI355X node, one process per GPU (BASELINE.json configs[3]: 8 independent scenes, RCCL frame gather over xGMI).
#
#   tools/launch_node.sh [N=8] [bench.py args...]           e.g.  tools/launch_node.sh 8 --steps 400 --warmup 40
#
# What torchrun would do (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), plus what it does not:
#   * every rank is pinned to the NUMA node its GPU hangs off (numactl when present, taskset otherwise): a rank replays
#     ~9 k hipGraphs per second and feeds 3 streams -- the launch thread must not migrate across sockets;
#   * GSWORLD_ISOLATE=1 gives each rank ONLY its own device (ROCR_VISIBLE_DEVICES=<rank>, LOCAL_RANK=0) instead of all
#     eight (LOCAL_RANK=<rank>); RCCL connects the ranks over xGMI either way (dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0).
# The driver's own launch line (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N) is equivalent
# up to the pinning; this script exists so that the pinning is written down and testable.
set -euo pipefail
N=${1:-8}; shift || true
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29531} WORLD_SIZE=$N HSA_ENABLE_IPC_MODE_LEGACY=0
numa_of_gpu() {  # NUMA node of GPU $1 (sysfs; -1 / missing -> 0)
  local n=-1 i=0 d
  for d in /sys/class/drm/card*/device; do
    [ -e "$d/vendor" ] && grep -qi 0x1002 "$d/vendor" || continue
    if [ "$i" -eq "$1" ]; then n=$(cat "$d/numa_node" 2>/dev/null || echo -1); break; fi
    i=$((i+1))
  done
  [ "$n" -lt 0 ] && n=0
  echo "$n"
}
pids=()
for r in $(seq 0 $((N-1))); do
  node=$(numa_of_gpu "$r")
  if command -v numactl >/dev/null 2>&1; then pin=(numactl --cpunodebind="$node" --membind="$node")
  else cpus=$(cat /sys/devices/system/node/node"$node"/cpulist 2>/dev/null || echo ""); pin=(); [ -n "$cpus" ] && pin=(taskset -c "$cpus"); fi
  if [ "${GSWORLD_ISOLATE:-0}" = "1" ]; then dev_env=(ROCR_VISIBLE_DEVICES="$r" LOCAL_RANK=0); else dev_env=(LOCAL_RANK="$r"); fi
  echo "[launch_node] rank $r -> GPU $r, NUMA node $node, ${pin[*]:-unpinned}" >&2
  env RANK="$r" "${dev_env[@]}" "${pin[@]}" python "$ROOT/bench.py" --gpus "$N" "$@" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
exit $rc
