#!/bin/bash
# final check at HEAD: build() + smoke(), the whole GPU suite, soak
mkdir -p gpurun_out/r3l
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r3l/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3l/smoke.log
python -m pytest tests -m gpu -q > gpurun_out/r3l/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3l/tests.log
