"""CPU oracle for the HippoMM hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the reported CPU baseline.  The
product package (``hippomm_amd``) never imports this package and fails loudly
when its HIP extension is missing.

Contents
--------
vector_ops_oracle.py     numpy restatement of ``top_k_cosine_similarity``
                         (reference hippomm/utils/vector_ops.py:151-188).
                         PINNED: tests/golden/scan_*.json were produced by the
                         unmodified reference function imported from
                         /root/reference (tests/golden/make_golden.py).
consolidation_oracle.py  numpy restatement of ``_select_key_frames``
                         (reference hippomm/core/hippocampal_memory.py:944-967).
                         PINNED the same way (reference imported behind stub
                         modules for its absent third-party imports).
imagebind_oracle.py      fp32 PyTorch restatement of the ImageBind-huge vision
                         and audio towers.  PARITY UNPINNED: the arithmetic
                         lives in the un-vendored, un-pinned third-party
                         package ``imagebind`` (facebookresearch/ImageBind,
                         installed by ``git clone`` + ``pip install .`` with no
                         commit, reference README.md:35-48); neither its source
                         nor its checkpoint is in /root/reference or in this
                         image, and the reference holds no tests or golden
                         vectors for it.  The restatement follows the published
                         architecture and keeps upstream state-dict key names so
                         that a real ``imagebind_huge.pth`` can validate it later.

The reference is a Python program, so the oracle is numpy / torch-CPU; there is
no C restatement to compile and no ``oracle/_ref`` build.
"""
