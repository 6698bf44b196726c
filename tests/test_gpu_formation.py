"""GPU: the FORMATION path end to end through files -- the twin of tests/test_gpu_retrieval.py -- as the reference runs it:

    frames on disk (JPEG paths) -> ImageBind.extract_features({'vision': paths}, ['vision'])['vision']
                                    .detach().cpu().numpy()                         hippocampal_memory.py:1180-1186, :1328-1335
    per-segment (b,1024) blocks -> _process_vision_features: one row per frame, shape guard, stable time sort,
                                   np.stack -> _select_key_frames(features, times)    hippocampal_memory.py:815-867, :944-967
    ThetaEvent -> save_theta_event (json.dump(to_dict(), indent=2))                   hippocampal_memory.py:110-133, :331-335
    load_theta_event (json.load + np.array -> float64) -> feature_search with a text query   :369-395, :2173-2176, :3143-3153

Here: 40 JPEGs of two sizes in tmp_path; the package's ImageBind (decode on the host, Pillow-identical resize / crop / normalise
on the GPU, HIP vision tower), a LITERAL restatement of _process_vision_features' flatten / sort / stack feeding the drop-in
_select_key_frames, event_store.save_event, both readers, and top_k_cosine_similarity with a query from the HIP text tower.
Checked against: the fp32 oracle tower on the Pillow-preprocessed tensors (tests/host_vision_pipeline.py), the selection
oracle on the same embeddings, json.dumps(indent=2) of the ThetaEvent dict, and the scan oracle on what the reference's reader
yields."""
import json

import numpy as np
import pytest
import torch

from oracle import imagebind_oracle as ib
from oracle.consolidation_oracle import select_key_frames_oracle
from oracle.vector_ops_oracle import top_k_cosine_similarity_oracle

pytestmark = pytest.mark.gpu
DEPTH = 2


def _write_frames(tmp_path):
    """Five scenes x eight frames, scenes alternating between 320x240 and 288x360 (landscape / portrait: both resize branches),
    frames of a scene = the scene plus sensor noise, saved as JPEG.  Returns (paths, times) in time order."""
    from PIL import Image
    rng = np.random.default_rng(5)
    paths, times = [], []
    for s in range(5):
        h, w = (240, 320) if s % 2 == 0 else (360, 288)
        base = np.random.default_rng(100 + s).uniform(0, 255, (6, 8, 3)).astype(np.uint8)
        scene = np.asarray(Image.fromarray(base).resize((w, h), Image.BICUBIC)).astype(np.float32)
        yy, xx = np.mgrid[0:h, 0:w]
        scene = scene + (40 * np.sin(xx * (0.05 + 0.03 * s)) * np.cos(yy * (0.04 + 0.02 * s)))[..., None]
        for f in range(8):
            frame = np.clip(scene + rng.normal(0, 4, (h, w, 3)), 0, 255).astype(np.uint8)
            p = tmp_path / "frames" / f"video_00_{s * 8 + f:04d}.jpg"
            p.parent.mkdir(parents=True, exist_ok=True)
            Image.fromarray(frame).save(p, quality=92)
            paths.append(str(p))
            times.append(float(s * 8 + f))
    return paths, times


class _ShortTermMemory:
    """The three fields of hippocampal_memory.ShortTermMemory that _process_vision_features reads."""
    def __init__(self, frames, frame_times, vision):
        self.modalities = ["vision"]
        self.content = {"frames": frames, "frame_times": frame_times}
        self.features = {"vision": vision}


def _extract_frame_feature(features, idx):
    """hippocampal_memory.py:929-942, restated."""
    if features is None:
        return None
    if isinstance(features, torch.Tensor):
        features = features.detach().cpu().numpy()
    if isinstance(features, np.ndarray):
        if len(features.shape) > 1 and features.shape[0] > 1 and idx < features.shape[0]:
            return features[idx]
        return features
    return None


def _process_vision_features(memories, select_key_frames, **select_kw):
    """hippocampal_memory.py:815-867, restated step for step (the pool argument is unused there too)."""
    frames_data = []
    for memory in memories:
        if "vision" in memory.modalities and "frames" in memory.content:
            for idx, frame in enumerate(memory.content["frames"]):
                if idx < len(memory.content.get("frame_times", [])):
                    frame_time = memory.content["frame_times"][idx]
                    feature = _extract_frame_feature(memory.features.get("vision"), idx)
                    if feature is not None:
                        if len(feature.shape) > 1:
                            feature = feature.flatten()
                        if feature.shape[0] != 1024:
                            continue
                        frames_data.append((frame, feature, frame_time))
    if not frames_data:
        return {"features": {}, "content": {}}
    frames_data.sort(key=lambda x: x[2])
    features = np.stack([f[1] for f in frames_data])
    times = np.array([f[2] for f in frames_data])
    key_indices = select_key_frames(None, features, times, **select_kw)
    return {"features": {"vision": features, "vision_times": times},
            "content": {"frames": [frames_data[i][0] for i in key_indices], "frame_times": times[key_indices].tolist()}}


def test_frames_on_disk_to_a_consolidated_event_and_back_to_a_query(tmp_path):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from host_vision_pipeline import load_and_transform_vision_data
    from test_tokenizer import _learn_merges
    from hippomm_amd import event_store as es
    from hippomm_amd.consolidation import _select_key_frames
    from hippomm_amd.encoder import ImageBind
    from hippomm_amd.tokenizer import SimpleTokenizer
    from hippomm_amd.vector_ops import top_k_cosine_similarity

    paths, times = _write_frames(tmp_path)
    vspec, tspec = ib.reduced(ib.VISION_HUGE, DEPTH), ib.reduced(ib.TEXT_HUGE, DEPTH)
    st = {**ib.synthetic_state(vspec, seed=77, init="rich"), **ib.synthetic_state(tspec, seed=78, init="rich")}
    tok = SimpleTokenizer("", merges=_learn_merges())
    model = ImageBind(state_dict=st, towers=("vision", "text"), depth={"vision": DEPTH, "text": DEPTH}, tokenizer=tok)

    # ---- encoding, segment by segment, exactly the caller's expression (:1180-1186); segments arrive out of time order
    memories, order = [], [2, 0, 4, 1, 3]
    for s in order:
        seg_paths, seg_times = paths[s * 8:(s + 1) * 8], times[s * 8:(s + 1) * 8]
        vision = model.extract_features({"vision": seg_paths}, ["vision"])["vision"].detach().cpu().numpy()
        assert vision.shape == (8, 1024) and vision.dtype == np.float32
        memories.append(_ShortTermMemory(seg_paths, seg_times, vision))
    # the 32-frame buffer call (:1328-1335) gives the same rows as the segment calls (one regime: bitwise)
    buffered = model.extract_features({"vision": paths[:32]}, ["vision"])["vision"].detach().cpu().numpy()
    by_time = {t: m.features["vision"][i] for m in memories for i, t in enumerate(m.content["frame_times"])}
    assert np.array_equal(buffered, np.stack([by_time[float(t)] for t in range(32)]))
    # embeddings vs the fp32 oracle on the tensors the reference's HOST pipeline (Pillow) produces from the same files
    x_host = load_and_transform_vision_data(paths, "cpu")
    want = ib.vision_forward(x_host, st, vspec).numpy()
    got = np.stack([by_time[float(t)] for t in range(40)])
    cos = (got * want).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
    assert (1 - cos).max() <= 5e-5 and np.abs(got - want).max() <= 2e-3, (float((1 - cos).max()), float(np.abs(got - want).max()))

    # ---- consolidation: the literal caller around the drop-in selection
    result = _process_vision_features(memories, _select_key_frames)
    feats, ftimes = result["features"]["vision"], result["features"]["vision_times"]
    assert feats.shape == (40, 1024) and np.array_equal(ftimes, np.arange(40.0)) and np.array_equal(feats, got)   # ALL rows, time-sorted
    kept_default = select_key_frames_oracle(feats, ftimes)
    assert result["content"]["frame_times"] == ftimes[kept_default].tolist()
    assert result["content"]["frames"] == [paths[i] for i in kept_default]
    # random-init towers embed every image within cos 0.99 of every other, so the reference's 0.9 keeps frame 0 only; the same
    # call with a threshold between the scenes' similarities (the signature has the parameter, :944-945) must split the scenes
    gram = (feats / np.linalg.norm(feats, axis=1, keepdims=True)) @ (feats / np.linalg.norm(feats, axis=1, keepdims=True)).T
    within = min(gram[s * 8:(s + 1) * 8, s * 8:(s + 1) * 8].min() for s in range(5))
    between = max(gram[a * 8:(a + 1) * 8, b * 8:(b + 1) * 8].max() for a in range(5) for b in range(5) if a != b)
    assert within - between > 1e-4, (within, between)
    thr = float((within + between) / 2)
    split = _process_vision_features(memories, _select_key_frames, similarity_threshold=thr)
    want_split = select_key_frames_oracle(feats, ftimes, thr)
    assert split["content"]["frame_times"] == ftimes[want_split].tolist() and len(want_split) == 5
    assert all(int(t) % 8 == 0 for t in split["content"]["frame_times"])            # the first frame of a scene, never a repeat

    # ---- the event file: ThetaEvent.to_dict()'s text, byte for byte, plus sidecars
    event = {"features": result["features"], "frames": split["content"]["frames"], "frame_times": split["content"]["frame_times"],
             "frame_captions": ["" for _ in split["content"]["frames"]], "audio_times": [], "audio_transcription": [],
             "holistic_audio_transcription": [], "summary": "five scenes", "start_time": 0.0, "end_time": 39.0}
    path = es.save_event(event, tmp_path / "memory_store" / "events" / "video_00" / "video_00_0.json")
    to_dict = {"features": {"vision": feats.tolist()}, "feature_times": {"vision_times": ftimes.tolist()},
               **{k: event[k] for k in ("frames", "frame_times", "frame_captions", "audio_times", "audio_transcription",
                                        "holistic_audio_transcription", "summary", "start_time", "end_time")}}
    assert path.read_text() == json.dumps(to_dict, indent=2)                          # :110-133 + :334-335
    # ---- reload: the reference's reader rules (float64 lists) and the sidecar fast path give the same values
    data = json.loads(path.read_text())
    ref_rows = np.array(data["features"]["vision"])                                   # :387-395 -> float64
    assert ref_rows.dtype == np.float64 and np.array_equal(ref_rows.astype(np.float32), feats)
    fast = es.load_event_features(path)
    assert np.array_equal(fast["vision"], feats)

    # ---- a question against the reloaded event (:2173-2176 -> :3153)
    question = "the bright scene with the narrow stripes"
    q = model.extract_features({"text": [question]}, ["text"])["text"].cpu().numpy().flatten()
    q_want = ib.text_forward(tok([question]), st, tspec).numpy().flatten()
    assert 1 - float(q @ q_want / (np.linalg.norm(q) * np.linalg.norm(q_want))) <= 5e-5
    idx, sims = top_k_cosine_similarity(q, ref_rows, 5)
    w_idx, w_sims = top_k_cosine_similarity_oracle(q, ref_rows, 5)
    assert idx.dtype == np.int64 and sims.dtype == np.float64 and len(idx) == 5
    np.testing.assert_allclose(sims, w_sims, rtol=0, atol=2e-6)
    full = (ref_rows @ q) / (np.linalg.norm(ref_rows, axis=1) * np.linalg.norm(q))
    order_all = np.sort(full)[::-1]
    gaps_ok = np.abs(np.diff(order_all[:6])) > 4e-6                                  # ranks clear of both neighbours
    for r in range(5):
        if (r == 0 or gaps_ok[r - 1]) and gaps_ok[r]:
            assert idx[r] == w_idx[r]
    np.testing.assert_allclose(full[idx], sims, rtol=0, atol=2e-6)                    # every hit carries its own similarity


def test_extract_features_from_files_equals_forward_of_load_data_bitwise(tmp_path):
    """The overlapped files -> embeddings pipeline behind extract_features (decode | upload + resize | tower, cut into ranges by
    the timing of the decoders) gives the bits of the two-step route forward(load_data(...)) (foundation_models.py:135-151), for
    one frame, a segment, the 32-frame buffer and a call that mixes both frame sizes; a failing file leaves the modality out."""
    from hippomm_amd.encoder import ImageBind
    paths, _ = _write_frames(tmp_path)
    vspec = ib.reduced(ib.VISION_HUGE, DEPTH)
    model = ImageBind(state_dict=ib.synthetic_state(vspec, seed=77, init="rich"), towers=("vision",), depth={"vision": DEPTH})
    for sel in (paths[:1], paths[:8], paths[:32], paths[4:40:3], paths):
        fused = model.extract_features({"vision": sel}, ["vision"])["vision"]
        two_step = model.forward(model.load_data({"vision": sel}, ["vision"]))["vision"]
        assert fused.shape == (len(sel), 1024) and torch.equal(fused, two_step)
    class _Img:                                                  # PIL images are accepted through .filename (:83-86)
        def __init__(self, filename):
            self.filename = filename
    assert torch.equal(model.extract_features({"vision": [_Img(p) for p in paths[:8]]}, ["vision"])["vision"],
                       model.extract_features({"vision": paths[:8]}, ["vision"])["vision"])
    assert model.extract_features({"vision": paths[:3] + [str(tmp_path / "nope.jpg")]}, ["vision"]) == {}
    assert "vision" in model.extract_features({"vision": paths[:3]}, ["vision"])             # and the next call works


def test_sharded_consolidation_from_files_at_world_size_one(tmp_path):
    """consolidate_paths_sharded (BASELINE cfg 5 from files) with no process group = the single-process formation: every frame's
    embedding in time order and the kept indices of the selection oracle on them."""
    from hippomm_amd.encoder import ImageBind
    from hippomm_amd.sharding import consolidate_paths_sharded
    paths, _ = _write_frames(tmp_path)
    vspec = ib.reduced(ib.VISION_HUGE, DEPTH)
    model = ImageBind(state_dict=ib.synthetic_state(vspec, seed=77, init="rich"), towers=("vision",), depth={"vision": DEPTH})
    extract = lambda ps: model.extract_features({"vision": ps}, ["vision"])["vision"]
    feats, kept = consolidate_paths_sharded(paths, extract, 0.9)
    assert torch.equal(feats, extract(paths))
    assert kept.cpu().tolist() == select_key_frames_oracle(feats.cpu().numpy(), None, 0.9).tolist()
