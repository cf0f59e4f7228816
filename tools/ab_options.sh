#!/bin/bash
# A/B of the whole training step inside ONE GPU-box call, same library, different option strings (MEDNET_OPTIONS) and / or
# environment knobs, interleaved over three rounds.  usage: AB="name1:ENV1=..,ENV2=..;name2:..." tools/ab_options.sh [bench args]
export PYTHONPATH="$PWD:$PWD/torch-mednet_amd:$PYTHONPATH"
IFS=';' read -ra ARMS <<< "${AB:-default:}"
for round in 1 2 3; do
  for arm in "${ARMS[@]}"; do
    name="${arm%%:*}"; envs="${arm#*:}"
    ( IFS=' ' ; for kv in ${envs//|/ }; do export "$kv"; done
      python bench.py --steps 10 --warmup 3 --cpu-steps 0 --fp32-steps 0 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], 'patches/s', d['ms_per_step'], 'ms')" )
  done
done
