"""CLIP-BPE tokenizer.  The upstream merge table is not available here, so the ALGORITHM is pinned
against an independent implementation of the same published scheme -- the HuggingFace `tokenizers` library configured
the way `transformers`' CLIP fast tokenizer is -- on merge tables learned from a small corpus."""
import gzip

import pytest
import torch

from hippomm_amd.tokenizer import PATTERN, SimpleTokenizer, bytes_to_unicode, read_merges

CORPUS = [
    "what is the person doing in the kitchen?", "a man opens the door and walks into the room",
    "the woman is holding a red cup, isn't she?", "who's talking at 3:45pm about the weather",
    "two dogs are running on the beach; they've found a ball", "what colour is the car parked outside",
    "she said: \"i'll be back in 10 minutes\"", "the children's toys were scattered everywhere!!!",
    "describe what happened after the phone rang", "is there any music playing in the background",
    "naïve café déjà-vu ünïcödé strings 你好 мир", "numbers 1234567890 and symbols #@$%^&*()",
] * 3


def _learn_merges(n_merges=300):
    """Train a byte-level BPE with CLIP's end-of-word convention using the `tokenizers` trainer."""
    from tokenizers import Regex, Tokenizer, models, normalizers, pre_tokenizers, trainers
    tok = Tokenizer(models.BPE(end_of_word_suffix="</w>"))
    tok.normalizer = normalizers.Sequence([normalizers.NFC(), normalizers.Replace(Regex(r"\s+"), " "), normalizers.Lowercase()])
    tok.pre_tokenizer = pre_tokenizers.Sequence([
        pre_tokenizers.Split(Regex(PATTERN), behavior="removed", invert=True),
        pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    alphabet = list(bytes_to_unicode().values())
    trainer = trainers.BpeTrainer(vocab_size=512 + n_merges, initial_alphabet=alphabet, end_of_word_suffix="</w>",
                                  special_tokens=[], show_progress=False)
    tok.train_from_iterator(CORPUS, trainer)
    import json
    state = json.loads(tok.to_str())
    merges = [tuple(m) if isinstance(m, list) else tuple(m.split(" ")) for m in state["model"]["merges"]]
    return merges


def _hf_reference(merges):
    """The same merge table inside a `tokenizers` BPE model with our id layout."""
    from tokenizers import Regex, Tokenizer, models, normalizers, pre_tokenizers
    ours = SimpleTokenizer("", merges=merges)
    vocab = {k: v for k, v in ours.encoder.items() if not k.startswith("<|")}
    tok = Tokenizer(models.BPE(vocab=vocab, merges=list(merges), end_of_word_suffix="</w>"))
    tok.normalizer = normalizers.Sequence([normalizers.NFC(), normalizers.Replace(Regex(r"\s+"), " "), normalizers.Lowercase()])
    tok.pre_tokenizer = pre_tokenizers.Sequence([
        pre_tokenizers.Split(Regex(PATTERN), behavior="removed", invert=True),
        pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    return ours, tok


def test_byte_table_is_a_bijection():
    t = bytes_to_unicode()
    assert len(t) == 256 and len(set(t.values())) == 256
    assert t[ord("a")] == "a" and t[ord(" ")] == "Ġ" and all(not c.isspace() for c in t.values())


def test_matches_independent_bpe_implementation():
    merges = _learn_merges()
    assert len(merges) > 100
    ours, ref = _hf_reference(merges)
    texts = CORPUS[:12] + ["The  QUICK\tbrown fox's 2nd jump... wasn't & isn't", "x", "!!!", "o'clock they'd've"]
    for text in texts:
        assert ours.encode(text) == ref.encode(text.strip()).ids, text
    # upstream additionally un-escapes HTML entities (twice) before tokenising
    assert ours.encode("cats &amp;amp; dogs") == ours.encode("cats & dogs")


def test_call_layout_sot_eot_padding_truncation():
    merges = _learn_merges(120)
    tok = SimpleTokenizer("", merges=merges)
    sot, eot = tok.encoder["<|startoftext|>"], tok.encoder["<|endoftext|>"]
    assert eot == len(tok.encoder) - 1 and sot == eot - 1              # EOT has the largest id: argmax finds it
    out = tok(["what is the person doing", ""])
    assert out.shape == (2, 77) and out.dtype == torch.long
    ids = tok.encode("what is the person doing")
    assert out[0, 0] == sot and out[0, 1:1 + len(ids)].tolist() == ids and out[0, 1 + len(ids)] == eot
    assert (out[0, 2 + len(ids):] == 0).all()
    assert out[1, :2].tolist() == [sot, eot] and (out[1, 2:] == 0).all()
    long = tok(" ".join(["kitchen"] * 200))
    assert long.shape == (1, 77) and long[0, 0] == sot and (long[0] != 0).all()       # cut at 77, as upstream
    assert tok.decode(tok.encode("the woman is holding a red cup")).strip() == "the woman is holding a red cup"
    assert tok("single string").shape == (1, 77)


def test_full_size_vocabulary_ids_and_file_reading(tmp_path):
    """With 48 894 merges the special tokens get CLIP's ids 49406 / 49407; header line and gzip are handled."""
    base = list(bytes_to_unicode().values())
    fake = [(base[i % 256], base[(i // 256) % 256] + "</w>") for i in range(48894)]     # distinct dummy pairs
    lines = ["#version: 0.2"] + [" ".join(m) for m in fake] + ["tail beyond the slice"] * 5
    path = tmp_path / "bpe_simple_vocab_16e6.txt.gz"
    path.write_bytes(gzip.compress("\n".join(lines).encode("utf-8")))
    assert read_merges(str(path)) == fake
    tok = SimpleTokenizer(str(path))
    assert len(tok.encoder) == 49408
    assert tok.encoder["<|startoftext|>"] == 49406 and tok.encoder["<|endoftext|>"] == 49407
    from hippomm_amd.tokenizer import find_bpe_vocab
    assert find_bpe_vocab(str(tmp_path)) == path


@pytest.mark.gpu
def test_text_strings_through_imagebind(tmp_path):
    """foundation_models.py:73-78: {'text': [question]} -> token ids -> text tower, with a merge table on disk."""
    from hippomm_amd.encoder import ImageBind, synthetic_state_dict
    merges = _learn_merges(200)
    base = list(bytes_to_unicode().values())
    pad = [(base[i % 256], base[(i // 256) % 256] + "</w>") for i in range(48894 - len(merges))]
    path = tmp_path / "bpe_simple_vocab_16e6.txt.gz"
    path.write_bytes(gzip.compress("\n".join(["#version"] + [" ".join(m) for m in list(merges) + pad]).encode()))
    model = ImageBind(str(tmp_path), state_dict=synthetic_state_dict(("text",), depth={"text": 1}), towers=("text",),
                      depth={"text": 1})
    feats = model.extract_features({"text": ["what is the person doing in the kitchen?", "who opened the door"]}, ["text"])
    assert feats["text"].shape == (2, 1024) and torch.isfinite(feats["text"]).all()
    assert not torch.allclose(feats["text"][0], feats["text"][1])
