#!/bin/bash
# sweep launch geometry / cache policy of the PCA X pass
for nt in 0 1 2 3; do for wg in 2 3 4 6 8; do
  r=$(VMP_PCA_XPASS_NT=$nt VMP_PCA_XPASS_WGS_PER_CU=$wg python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f it/s pass %.3f ms %.0f GB/s' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['achieved']))")
  echo "nt=$nt wgs/cu=$wg : $r"
done; done
