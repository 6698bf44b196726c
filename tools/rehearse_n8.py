"""The driver's N = 8 command shapes rehearsed on ONE GPU (HMM_BENCH_REHEARSAL=1: eight ranks on cuda:0, collectives over gloo):
`bench.py --gpus 8` (cfg 2, weak, with the row-sharded scan leg) and `--workload cfg5` (8 x 450 frames), wall time of each whole
command and the start-up per rank.  The numbers of the lines mean nothing (the ranks share a GPU); that the commands finish, how
long they take and what they print is the point.  usage: rehearse_n8.py [json_out]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = []
for wl in ("cfg2", "cfg5"):
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--workload", wl],
                       capture_output=True, text=True, env=dict(os.environ, HMM_BENCH_REHEARSAL="1"), cwd=ROOT)
    wall = time.perf_counter() - t0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    d = json.loads(lines[-1]) if lines else {}
    rec = {"command": f"HMM_BENCH_REHEARSAL=1 python bench.py --gpus 8 --steps 2 --warmup 1 --workload {wl}", "rc": r.returncode,
           "wall_s": round(wall, 1), "json_lines": len(lines), "n_gpus": d.get("n_gpus"), "startup_s_rank0": d.get("startup_s"),
           "ms_per_step_eight_ranks_on_one_gpu": d.get("ms_per_step"), "all_reduced_rank_count": d.get("all_reduced_rank_count"),
           "collective_backend": d.get("collective_backend"), "frames_per_rank": [x.get("frames") for x in d.get("ranks", [])],
           "all_gather_ms": d.get("config", {}).get("all_gather_ms"),
           "scan_leg": {k: v for k, v in (d.get("scan") or {}).items() if k in ("value", "weak_1M_rows_per_gpu", "error")},
           "kept_equal_cpu_oracle": d.get("config", {}).get("kept_equal_cpu_oracle_on_gathered_matrix"),
           "stderr_tail": r.stderr[-300:] if r.returncode else ""}
    out.append(rec)
    print(json.dumps(rec), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
sys.exit(0 if all(o["rc"] == 0 for o in out) else 1)
