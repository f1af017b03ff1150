#!/bin/bash
# bench.py A/B over one extra flag, interleaved on one box:  gpu_iter6.sh TAG "--flag"
O=gpurun_out/${1:-it6}; mkdir -p $O
for i in 1 2 3; do
timeout 200 python bench.py --no-cpu-baseline 2> $O/base$i.err | grep -o '"ms_per_step": [0-9.]*' | head -n 1 | sed "s/^/base: /"
timeout 200 python bench.py --no-cpu-baseline $2 2> $O/alt$i.err | grep -o '"ms_per_step": [0-9.]*' | head -n 1 | sed "s/^/$2: /"
done | tee $O/ab.txt
