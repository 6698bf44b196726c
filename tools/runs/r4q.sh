#!/bin/bash
# GPU session r4q: the prefilter probe (profiles/r4_prefilter_probe.json)
mkdir -p gpurun_out/r4q
timeout 600 python tools/prefilter_probe.py gpurun_out/r4q/prefilter_probe.json 2>&1 | grep "^{"
