"""Golden vectors for the SPM / byte-level-BPE streaming detokenizers: the reference's own classes
(/root/reference/mlx_vlm/tokenizer_utils.py, imported by file path - it needs only `transformers`) run over two small
tokenizers trained here with the `tokenizers` library.  Saves the tokenizer.json texts, the token streams, the segment each
token produced and the final text into tests/golden/detok_ref.npz.

    python tests/golden/make_golden_detok.py
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/mlx_vlm/tokenizer_utils.py"

CORPUS = [
    "The quick brown fox jumps over the lazy dog. " * 3,
    "Streaming detokenizers commit text word by word; punctuation, numbers 12345 and CamelCaseWords included.",
    "A picture of two cats sleeping on a red couch next to a window.",
    "def add(a, b):\n    return a + b  # indentation and newlines\n\nprint(add(2, 3))",
    "naive cafe resume - and some accents: naïve café résumé über straße",
]
TEXTS = [
    "The lazy dog sleeps. Two cats jump over the red window!",
    "  leading spaces, then naïve café 北京 and an emoji 🙂 at the end",
    "def f(x):\n    return x + 1\n\n# 夢 <- a byte-fallback character in the middle of a line",
    "",
    "word",
]


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_tokenizer_utils", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build_bpe():
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers

    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    tr = trainers.BpeTrainer(vocab_size=420, special_tokens=["<|endoftext|>", "<|im_end|>"],
                             initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(CORPUS, tr)
    return tok


def build_spm(strip: bool):
    from tokenizers import Tokenizer, decoders, models, normalizers, trainers

    tok = Tokenizer(models.BPE(byte_fallback=True, unk_token="<unk>", fuse_unk=True))
    tok.normalizer = normalizers.Sequence([normalizers.Prepend("▁"), normalizers.Replace(" ", "▁")])
    steps = [decoders.Replace("▁", " "), decoders.ByteFallback(), decoders.Fuse()]
    if strip:
        steps.append(decoders.Strip(" ", 1, 0))
    tok.decoder = decoders.Sequence(steps)
    tr = trainers.BpeTrainer(vocab_size=620, special_tokens=["<unk>", "<s>", "</s>"] + [f"<0x{i:02X}>" for i in range(256)],
                             show_progress=False)
    tok.train_from_iterator(CORPUS, tr)
    return tok


def main():
    from transformers import PreTrainedTokenizerFast

    ref = load_reference()
    blob = {}
    cases = [("bpe", build_bpe(), lambda t: ref.BPEStreamingDetokenizer(t)),
             ("spm", build_spm(True), lambda t: ref.SPMStreamingDetokenizer(t)),
             ("spm_nostrip", build_spm(False), lambda t: ref.SPMStreamingDetokenizer(t, trim_space=False))]
    for name, tok, make in cases:
        blob[f"{name}.json"] = np.array([tok.to_str()])
        fast = PreTrainedTokenizerFast(tokenizer_object=tok)
        det = make(fast)
        for i, text in enumerate(TEXTS):
            ids = fast.encode(text, add_special_tokens=False)
            det.reset()
            segs = []
            for t in ids:
                det.add_token(t)
                segs.append(det.last_segment)
            det.finalize()
            segs.append(det.last_segment)
            blob[f"{name}.{i}.ids"] = np.array(ids, dtype=np.int64)
            blob[f"{name}.{i}.segments"] = np.array(segs if segs else [""])
            blob[f"{name}.{i}.text"] = np.array([det.text])
            print(name, i, len(ids), repr(det.text)[:70])
    np.savez_compressed(os.path.join(HERE, "detok_ref.npz"), **blob)
    print("wrote detok_ref.npz", os.path.getsize(os.path.join(HERE, "detok_ref.npz")), "bytes")


if __name__ == "__main__":
    sys.exit(main())
