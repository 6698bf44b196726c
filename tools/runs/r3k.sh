#!/bin/bash
mkdir -p gpurun_out/r3k
timeout 600 python tools/joint_probe.py > gpurun_out/r3k/joint.log 2>&1
tail -4 gpurun_out/r3k/joint.log
