#!/bin/bash
cd "$(dirname "$0")/../.."
for w in 8 12 16 24; do
  python bench.py --steps 3 --warmup 1 --prewarm 0.1 --no-cpu-baseline --no-extras --workers $w --batch-items 2048 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['batch']; print('workers',b['host_threads_per_rank'],'images/s',b['value'],'ref',b['n1_reference_images_per_s'])"
done
python tools/time_batch_jpeg_native.py 2>/dev/null | tail -8
