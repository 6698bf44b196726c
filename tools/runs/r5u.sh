#!/bin/bash
# two-chain regime knobs re-checked after the dispatcher changes
mkdir -p gpurun_out
timeout 400 python tools/knob_ab_probe.py g_enc_two_chain_small_tiles 64 96 gpurun_out/tcs_96.json vision:13,16,20,24,28,32 2>&1 | grep ratio
timeout 400 python tools/knob_ab_probe.py g_enc_two_chain_small_tiles 64 40 gpurun_out/tcs_40.json vision:13,16,20,24,28,32 2>&1 | grep ratio
timeout 400 python tools/knob_ab_probe.py g_enc_split_num 128 112 gpurun_out/sn_112.json vision:13,16,24,32 2>&1 | grep ratio
timeout 400 python tools/knob_ab_probe.py g_enc_side_priority 0 1 gpurun_out/prio.json vision:16,32 2>&1 | grep ratio
