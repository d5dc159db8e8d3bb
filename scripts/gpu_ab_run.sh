#!/bin/bash
# A/B of library builds on the benchmark frame, same box, alternating: gpu_ab_run.sh <out file> <lib> [<lib> ...]
# (a lib may be prefixed with ENV=VALUE, settings separated by commas, e.g. NOVA_HIP_SPIN_US=0:build_ab/libx.so)
out=$1; shift
: > "$out"
for rep in 1 2; do
  for spec in "$@"; do
    lib=${spec##*:}
    envs=""
    if [ "$lib" != "$spec" ]; then envs=$(echo "${spec%:*}" | tr ',' ' '); fi
    echo "== $spec" >> "$out"
    env $envs NOVA_HIP_LIB=$PWD/$lib timeout -k 10 120 python3 scripts/gpu_lib_ab.py >> "$out" 2>&1 || exit 1
  done
done
cat "$out"
