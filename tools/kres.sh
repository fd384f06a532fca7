#!/bin/bash
# kernel resource usage of one .hip file (VGPRs / AGPRs / scratch / LDS per kernel), compile-only: tools/kres.sh gemm_pp.hip [extra flags]
cd "$(dirname "$0")/../kddcup_2020_multimodalitiesrecall_2nd_place_amd/csrc" || exit 1
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  awk '/Function Name:/ {name=$5} / VGPRs:/ {v=$4} / AGPRs:/ {a=$4} /ScratchSize/ {s=$5} /LDS Size/ {l=$6; printf "%s vgpr %s agpr %s scratch %s lds %s\n", name, v, a, s, l}' | c++filt
