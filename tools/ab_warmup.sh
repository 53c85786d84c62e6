#!/bin/bash
# development aid: A/B the warmup kernel variants built as exmc_amd/lib/libexmc_hip_<v>.so with
# -DEXMC_XCC_PROBE; prints (xcc/cu, ms) per launch so runs on the same CU can be compared
for round in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = single ]; then
      EXMC_HIP_WARMUP_PIPE=0 EXMC_HIP_LIB=$PWD/exmc_amd/lib/libexmc_hip_spec2.so timeout -k 10 60 python tools/prof_warmup.py 6 2>&1 | grep "xcc probe" | awk -v v=$v '{print v, $7, $8}' | tr -d ',' | tail -n +2
    else
      EXMC_HIP_LIB=$PWD/exmc_amd/lib/libexmc_hip_$v.so timeout -k 10 60 python tools/prof_warmup.py 6 2>&1 | grep "xcc probe" | awk -v v=$v '{print v, $7, $8}' | tr -d ',' | tail -n +2
    fi
  done
done | sort | awk '{k=$1" "$2; s[k]+=$3; n[k]++; if(!(k in mn)||$3<mn[k])mn[k]=$3} END{for(k in s) printf "%-22s n=%2d mean %.2f min %.2f\n", k, n[k], s[k]/n[k], mn[k]}' | sort
