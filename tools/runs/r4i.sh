#!/bin/bash
REPO=$PWD
OUT=$REPO/gpurun_out/r4i
mkdir -p $OUT
timeout 400 python tools/energy_account_probe.py $OUT/energy_account.json > $OUT/energy_account.log 2>&1
tail -70 $OUT/energy_account.log
