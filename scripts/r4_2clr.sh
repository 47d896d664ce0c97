#!/bin/bash
# 2clr: split knobs, each against the default on one box
for setting in "AGBNP_HIP_SPLIT_BIG=1" "AGBNP_HIP_SPLIT_PERMILLE=700" "AGBNP_HIP_SPLIT_BIG=2"; do
  echo "== $setting"
  bash scripts/ab_env.sh "$setting" 1 --system 2clr --steps 200 --warmup 20
done
