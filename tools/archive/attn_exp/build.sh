#!/bin/bash
# builds tools/attn_exp/libattn_abl<mask>.so for every mask given
set -e
cd "$(dirname "$0")"
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DAP_PIPE_ABL=$m $EXTRA -shared -o libattn_abl$m.so -x hip wrap.cpp -x hip attention_pipe_experiment.hip &
done
wait
