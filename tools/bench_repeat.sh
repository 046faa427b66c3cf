#!/bin/bash
# repeat the default bench N times and print ms/step of every run (median is the figure to compare builds with)
N=${1:-5}; shift
for i in $(seq $N); do
  python bench.py --steps 400 --warmup 20 --no_cpu_baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_ms'], 'host enqueue', d.get('host_enqueue_ms_per_step'))"
done
