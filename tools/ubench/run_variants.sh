#!/bin/bash
# run every tools/ubench/bin/vv_* (GPU box), each under its own timeout
for b in tools/ubench/bin/vv_*; do
  timeout -k 5 60 $b "$@" || exit 1
done
