#!/bin/bash
# GPU session r6s: five-row re-score round, first range of eight frames, fast wav reader: their tests, the finish's stamps, the two real-call legs
REPO=$PWD
OUT=$REPO/gpurun_out/r6s
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_scan_prefilter.py tests/test_gpu_segments.py tests/test_gpu_retrieval.py tests/test_preprocess.py tests/test_gpu_formation.py tests/test_audio_fbank.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -1
timeout 300 python tools/prefilter_stress.py 100 > $OUT/stress.log 2>&1; echo "stress rc=$?"; tail -1 $OUT/stress.log
cd tools && timeout 300 python prefilter_final_stamps_probe.py $OUT/prefilter_final_stamps.json; cd ..
timeout 600 python - > $OUT/legs.json 2> $OUT/legs.err <<'PY'
import json, bench, torch
torch.cuda.set_device(0)
print(json.dumps({"formation_from_files": bench.formation_bench(False), "audio_from_wav": bench.audio_bench()}))
PY
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r6s/legs.json"))
f, a = d["formation_from_files"], d["audio_from_wav"]
print("formation", f["paths_32"]["ms_end_to_end"], f["paths_32"]["pipeline"]["ranges_issued"], f["paths_256"]["ms_end_to_end"], f["paths_256"]["frames_per_s"], f["paths_256"]["pipeline"]["ranges_issued"])
print("audio", a["wav_1"]["ms_end_to_end"], a["wav_1"]["stages_alone_ms"], a["wav_16"]["ms_end_to_end"], a["wav_16"]["stages_alone_ms"])
PY
