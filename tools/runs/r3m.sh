#!/bin/bash
mkdir -p gpurun_out/r3m
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r3m/smoke_build_first.log 2>&1; echo "build+smoke rc=$?"; tail -2 gpurun_out/r3m/smoke_build_first.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3m/smoke_only.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r3m/smoke_only.log
python __graft_entry__.py > gpurun_out/r3m/main.log 2>&1; echo "main rc=$?"; tail -1 gpurun_out/r3m/main.log
python -m pytest tests/test_cpu_abi.py tests/test_gpu_distributed.py tests/test_gpu_select.py -q 2>&1 | tail -2
timeout 900 python tools/mfma_power_probe.py gpurun_out/r3m/mfma_power.json > gpurun_out/r3m/mfma_power.log 2>&1
grep "^{" gpurun_out/r3m/mfma_power.log
