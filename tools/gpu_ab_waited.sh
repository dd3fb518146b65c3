#!/bin/bash
# A/B: a waited-for step (policy in the loop) as one graph replay against eleven launches from the kept argument pack
cd "${GRAFT_REPO_ROOT:-/root/repo}"
line() { python -c "
import json,sys
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print(' '.join(('policy' if r['policy_in_loop'] else 'ahead')+'='+str(round(r['frames_per_s'])) for r in rows))"; }
for rep in 1 2; do
for E in ${CL_ENVS:-1 2 4}; do
  echo "E=$E graph when waited: $(CL_ONLY=1 timeout 300 python tools/ab_closed_loop.py 1468850 $E 2>/dev/null | line)"
  echo "E=$E eager when waited: $(CL_EAGER_WAITED=1 CL_ONLY=1 timeout 300 python tools/ab_closed_loop.py 1468850 $E 2>/dev/null | line)"
done
done
