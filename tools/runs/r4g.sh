#!/bin/bash
# GPU session r4g: the round's PMC passes (MFMA busy, FETCH, WRITE, LDS) and the profiled + plain bench lines at HEAD
REPO=$PWD
PREFIX=r4 bash tools/pmc_passes.sh r4 > gpurun_out/r4g_pmc.log 2>&1
tail -5 gpurun_out/r4g_pmc.log
bash tools/profile_bench.sh r4 > gpurun_out/r4g_profile.log 2>&1
tail -3 gpurun_out/r4g_profile.log
