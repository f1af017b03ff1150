#!/bin/bash
# graph A / graph B times, A/B over an environment switch, interleaved on one box:  gpu_iter5.sh TAG VAR=VALUE
O=gpurun_out/${1:-it5}; mkdir -p $O
for i in 1 2 3; do
timeout 200 python scripts/graph_split.py 2> $O/split$i.err | grep "overlap=True" | sed "s/^/base: /"
env $2 timeout 200 python scripts/graph_split.py 2> $O/split_alt$i.err | grep "overlap=True" | sed "s/^/$2: /"
done | tee $O/split.txt
