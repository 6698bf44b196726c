#!/usr/bin/env python3
"""tools/validate_checkpoint.py plus the fp32 CPU oracle on the same weights and inputs.

    python tests/checkpoint_vs_oracle.py /path/to/imagebind_huge.pth [--towers vision audio text] [--depth N]

Reports, per tower that loaded cleanly, the cosine between the HIP tower's embeddings and oracle/imagebind_oracle.py's
(tolerance of the parity tests: >= 1 - 5e-5).  With a real imagebind_huge.pth this is the run that can retire the
encoder oracle's "parity unpinned" caveat.  Lives under tests/ because it uses the oracle."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch
    import validate_checkpoint as vc
    from oracle import imagebind_oracle as ib
    args = vc.parse_args()
    report, ok, sd, inputs, emb = vc.validate(args.checkpoint, args.towers, args.batch, args.depth)
    specs = {"vision": ib.VISION_HUGE, "audio": ib.AUDIO_HUGE, "text": ib.TEXT_HUGE}
    if args.depth > 0:
        specs = {t: ib.reduced(s, args.depth) for t, s in specs.items()}
    towers = list(emb)
    if towers:
        want = ib.forward({t: inputs[t] for t in towers}, {t: sd for t in towers}, specs)
        for t in towers:
            cos = torch.nn.functional.cosine_similarity(emb[t], want[t], dim=1)
            report[t]["cos_vs_fp32_oracle"] = [round(float(c), 7) for c in cos]
            ok &= bool((1 - cos).max() <= 5e-5)
    report["ok"] = bool(ok)
    print(json.dumps(report, indent=1))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
