#!/bin/bash
# GPU box: socket power / clocks sampled with rocm-smi (twice a second) while the headline bench runs 60 steps; idle samples dropped
cd "$(dirname "$0")/.."
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
rocm-smi --showpowercap 2>/dev/null | grep -i "cap\|power" | head -5
python bench.py --steps 60 --warmup 3 --no-cpu --no-secondary > /tmp/bench_power.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  L=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' ')
  case "$L" in *Mhz*) echo "$L";; esac
  sleep 0.4
done | awk '{ if ($NF+0 > 400) print }' | tail -25
cut -c1-200 /tmp/bench_power.json
