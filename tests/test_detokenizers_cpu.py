"""The SPM and byte-level-BPE streaming detokenizers (mlx-vlm_amd/tokenizer_utils.py) against the reference's own classes
(tests/golden/make_golden_detok.py ran them; reference mlx_vlm/tokenizer_utils.py:121-285): the segment produced after EVERY
token and the final text must be identical - plain English, leading spaces, accents, CJK and emoji through byte tokens,
code with newlines, the empty string - and the class picked from tokenizer.json's decoder section (reference 413-480)."""
import json
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "detok_ref.npz"), allow_pickle=False)


def _fast(name):
    from tokenizers import Tokenizer
    from transformers import PreTrainedTokenizerFast

    return PreTrainedTokenizerFast(tokenizer_object=Tokenizer.from_str(str(G[f"{name}.json"][0])))


def _make(name, tok):
    from mlx_vlm_amd.tokenizer_utils import BPEStreamingDetokenizer, SPMStreamingDetokenizer

    if name == "bpe":
        return BPEStreamingDetokenizer(tok)
    return SPMStreamingDetokenizer(tok, trim_space=(name == "spm"))


@pytest.mark.parametrize("name", ["bpe", "spm", "spm_nostrip"])
def test_segments_and_text_equal_the_reference_token_by_token(name):
    tok = _fast(name)
    det = _make(name, tok)
    n_cases = sum(1 for k in G.files if k.startswith(name + ".") and k.endswith(".ids"))
    assert n_cases == 5
    for i in range(n_cases):
        ids = G[f"{name}.{i}.ids"].tolist()
        det.reset()
        segs = []
        for t in ids:
            det.add_token(t)
            segs.append(det.last_segment)
        det.finalize()
        segs.append(det.last_segment)
        ref = [str(x) for x in G[f"{name}.{i}.segments"]] if ids else [""]
        assert segs == ref, (name, i)
        assert det.text == str(G[f"{name}.{i}.text"][0]) == "".join(segs)
        # linear time by construction, and the result is what the tokenizer itself decodes (modulo the dummy prefix space)
        full = tok.decode(ids)
        assert det.text.strip() == full.strip(), (name, i)


def test_skip_special_ids_and_copy_reset_isolation():
    import copy

    tok = _fast("bpe")
    det = _make("bpe", tok)
    ids = G["bpe.0.ids"].tolist()
    skip = tok.convert_tokens_to_ids("<|im_end|>")
    for t in ids[:5] + [skip] + ids[5:]:
        det.add_token(t, skip_special_token_ids=[skip])
    other = copy.copy(det)
    other.reset()                              # what make_streaming_detokenizer hands to a new generation
    assert other.text == "" and det.text != ""
    det.finalize()
    assert det.text == str(G["bpe.0.text"][0])


def test_detokenizer_class_is_picked_from_tokenizer_json(tmp_path):
    from functools import partial

    from mlx_vlm_amd.tokenizer_utils import BPEStreamingDetokenizer, SPMStreamingDetokenizer, detokenizer_class_for
    from mlx_vlm_amd.utils import NaiveStreamingDetokenizer

    assert detokenizer_class_for(str(tmp_path)) is NaiveStreamingDetokenizer            # no tokenizer.json
    for name, want in (("bpe", BPEStreamingDetokenizer), ("spm", SPMStreamingDetokenizer)):
        (tmp_path / "tokenizer.json").write_text(str(G[f"{name}.json"][0]))
        assert detokenizer_class_for(str(tmp_path)) is want
    (tmp_path / "tokenizer.json").write_text(str(G["spm_nostrip.json"][0]))
    got = detokenizer_class_for(str(tmp_path))
    assert isinstance(got, partial) and got.func is SPMStreamingDetokenizer and got.keywords == {"trim_space": False}
    (tmp_path / "tokenizer.json").write_text(json.dumps({"decoder": {"type": "WordPiece"}}))
    assert detokenizer_class_for(str(tmp_path)) is NaiveStreamingDetokenizer
    # the reference's _match compares the WHOLE description (tokenizer_utils.py:413-421): an SPM Sequence with an extra key
    # anywhere, or a value of another type, is NOT the SPM decoder there and falls back to the naive detokenizer
    spm = json.loads(str(G["spm.json"][0]))
    for mutate in (lambda d: d["decoder"].update(extra=1), lambda d: d["decoder"]["decoders"][0].update(note="x"),
                   lambda d: d["decoder"]["decoders"][3].update(start=1.0), lambda d: d["decoder"]["decoders"].append({"type": "Fuse"})):
        d = json.loads(json.dumps(spm))
        mutate(d)
        (tmp_path / "tokenizer.json").write_text(json.dumps(d))
        assert detokenizer_class_for(str(tmp_path)) is NaiveStreamingDetokenizer
    (tmp_path / "tokenizer.json").write_text("{ not json")
    with pytest.raises(json.JSONDecodeError):
        detokenizer_class_for(str(tmp_path))
