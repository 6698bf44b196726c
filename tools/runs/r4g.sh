#!/bin/bash
# GPU session r4g: the round's PMC passes (MFMA busy, FETCH, WRITE, LDS) at HEAD -> gpurun_out/pmc_r4_summary/r4_*.json
PREFIX=r4 bash tools/pmc_passes.sh r4 > gpurun_out/r4g_pmc.log 2>&1
tail -5 gpurun_out/r4g_pmc.log
