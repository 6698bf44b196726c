#!/bin/bash
# GPU session r4s: the driver's N = 8 command shapes rehearsed on one GPU (profiles/r4_rehearsal_8_ranks.json)
mkdir -p gpurun_out/r4s
timeout 1500 python tools/rehearse_n8.py gpurun_out/r4s/rehearsal_8_ranks.json
