"""GPU parity, nanoLLaVA (`llava_bunny`): the HIP path (SigLIP tower over the C-ABI ops, padded-head
Qwen1.5 decoder on the decode engine) against oracle/llava_bunny.py on the same seeded bf16 weights, and against the
vectors of the reference's own files (tests/golden/llava_bunny_tiny_ref.npz; only the .npz is read).

Tolerances as in test_engine_gpu.py: activations are bf16 with ~1-ulp differences per op that compound over depth
(stated per test against the tensor's rms); greedy tokens must be identical unless the oracle's own logit margin between
the two candidates is below the stated fraction of the logit rms (tie-aware)."""
import os

import numpy as np
import pytest
import torch

from oracle import llava_bunny as ob
from tests.helpers import build_bunny_model

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "llava_bunny_tiny_ref.npz"))
ROWS = slice(None, None, 7)


@pytest.fixture(scope="module")
def bunny():
    cfg = ob.tiny_cfg()
    W = {k: v.to(BF) for k, v in ob.random_weights(cfg, seed=4321, dtype=torch.float32, **ob.TEST_WEIGHT_SCALES).items()}
    model = build_bunny_model(cfg, W, kv_pool_tokens=4096, max_seqs=8)
    return cfg, W, model


def _pixels(i):
    return torch.from_numpy(ob.preprocess([G[f"img{i}.image_hwc"]]))


def _rel_rms(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


def _tie_aware(toks, ref_toks, ref_logits, tol):
    """Tokens equal, or the first divergence is a tie by the reference's own decision rule.  Greedy decoding takes the
    argmax of bf16 LOG-PROBS (logits - logsumexp, rounded to the model dtype, ar.py:368-379; lowest index wins among
    equals): two candidates whose logits are closer than one bf16 ulp of their log-prob (0.031 at -6, where a
    1024-way vocabulary sits) are indistinguishable to the reference itself.  Allowed margin = that ulp +
    tol * rms(logits) for the engine's own logit error.  `ref_logits` rows may be logits or log-probs."""
    for n, (a, b) in enumerate(zip(toks, ref_toks)):
        if a != b:
            row = ref_logits[n].float()
            lp = float(row[b] - torch.logsumexp(row, -1))
            ulp = 2.0 ** (np.floor(np.log2(max(abs(lp), 1e-30))) - 7)
            return abs(float(row[a]) - float(row[b])) <= ulp + tol * float(row.pow(2).mean().sqrt()), n
    return True, None


@pytest.mark.parametrize("i", [0, 1])
def test_siglip_tower_and_projector_vs_oracle_and_reference(bunny, i):
    cfg, W, model = bunny
    pix = _pixels(i)
    ref_last = ob.vision_tower(W, cfg, pix.to(BF))
    last = model.vision_tower(pix)
    assert last.shape == (729, cfg.vision.hidden_size)
    assert _rel_rms(last, ref_last[0]) < 2e-2
    assert _rel_rms(last[ROWS], torch.from_numpy(G[f"case{i}.bf16.ref_vision_last"])) < 2e-2
    feats = model.encode_image(pix)
    assert _rel_rms(feats, ob.mm_projector(W, ref_last)[0]) < 2e-2
    assert _rel_rms(feats[ROWS], torch.from_numpy(G[f"case{i}.bf16.ref_image_features"])) < 2e-2
    both = model.vision_tower(torch.cat([pix, _pixels(1 - i)]))       # two images per call: same rows for the first
    assert torch.equal(both[:729], last)


@pytest.mark.parametrize("i", [0, 1])
def test_splice_and_prefill_logits_vs_oracle(bunny, i):
    cfg, W, model = bunny
    ids, pix = G[f"case{i}.input_ids"], _pixels(i)
    f = model.get_input_embeddings(ids, pix)
    ref = ob.get_input_embeddings(W, cfg, ids, pix)
    assert f.inputs_embeds.shape == ref.shape == (1, ids.shape[1] + 728, cfg.text.hidden_size)
    pos = int(np.argmax(ids[0] == cfg.image_token_index))
    text_rows = list(range(pos)) + list(range(pos + 729, ref.shape[1]))
    assert torch.equal(f.inputs_embeds[0, text_rows].cpu(), ref[0, text_rows])           # embedding rows: exact
    assert _rel_rms(f.inputs_embeds[0, pos:pos + 729], ref[0, pos:pos + 729]) < 2e-2
    assert np.array_equal(np.asarray(f.position_ids)[0, 0], np.arange(ref.shape[1]))
    from mlx_vlm_amd.models import cache as cache_mod

    lm = model.language_model
    L = ref.shape[1]
    cache = cache_mod.make_prompt_cache(lm)
    logits = lm.prefill(f.inputs_embeds.reshape(L, -1), np.asarray(f.position_ids).reshape(3, L), [cache], [L], "last")
    ref_logits = ob.language_model(W, cfg, ref)[0, -1]
    err = (logits[0].float().cpu() - ref_logits.float()).abs().max() / ref_logits.float().pow(2).mean().sqrt()
    assert float(err) < 6e-2, float(err)
    gold = torch.from_numpy(G[f"case{i}.bf16.ref_prefill_logits_last"])
    err = (logits[0].float().cpu() - gold).abs().max() / gold.pow(2).mean().sqrt()
    assert float(err) < 6e-2, float(err)
    cache[0]._seq.release()


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("case", ["image0", "image1", "text"])
def test_greedy_generate_step_vs_oracle(bunny, case, use_graph):
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = bunny
    if case == "text":
        ids, pix = G["generate_step.text.input_ids"], None
    else:
        i = int(case[-1])
        ids, pix = G[f"case{i}.input_ids"], _pixels(i)
    n_new = 12
    ref_toks, ref_logits = ob.generate_greedy(W, cfg, ids, pix, max_tokens=n_new, return_logits=True)
    got = [t for t, _ in generate_step(ids, model, pix, None, max_tokens=n_new, temperature=0.0, use_graph=use_graph)]
    assert len(got) == n_new
    ok, n = _tie_aware(got, ref_toks, ref_logits, tol=3e-2)
    assert ok, (got, ref_toks, n)


def test_greedy_vs_reference_goldens_both_numeric_paths(bunny):
    """The reference's own runs: pixels pre-cast to bf16 (`case0.bf16`, the graph this engine computes) and float32
    pixels as its pipeline ships them (`generate_step.image`: float32 activations by type promotion).  First tokens
    identical; a later divergence must be a tie by the reference's own rule (see _tie_aware)."""
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = bunny
    ids, pix = G["case0.input_ids"], _pixels(0)
    got = [t for t, _ in generate_step(ids, model, pix, None, max_tokens=6, temperature=0.0)]
    gold = G["case0.bf16.ref_greedy"].tolist()
    rows = torch.cat([torch.from_numpy(G["case0.bf16.ref_prefill_logits_last"])[None],
                      torch.from_numpy(G["case0.bf16.ref_decode_logits"])])
    ok, n = _tie_aware(got, gold, rows, tol=3e-2)
    assert ok and got[0] == gold[0], (got, gold, n)
    shipped = G["generate_step.image.tokens"].tolist()
    ok, n = _tie_aware(got, shipped, torch.from_numpy(G["generate_step.image.logprobs"]), tol=3e-2)
    assert ok and got[0] == shipped[0], (got, shipped, n)
    text = [t for t, _ in generate_step(G["generate_step.text.input_ids"], model, None, None, max_tokens=6, temperature=0.0)]
    ok, n = _tie_aware(text, G["generate_step.text.tokens"].tolist(), torch.from_numpy(G["generate_step.text.logprobs"]), tol=3e-2)
    assert ok, (text, n)


def write_bunny_checkpoint(path, cfg, W):
    """A nanoLLaVA checkpoint directory in the HF layout the reference loads (llava_bunny.py:180-222 renames it):
    `model.vision_tower...` with the torch conv layout (O, C, kH, kW), `model.mm_projector.{0,2}`, `model.layers...`,
    tied embeddings (no lm_head), config.json with the text parameters at the root, a WordLevel tokenizer."""
    import json

    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    t, v = cfg.text, cfg.vision
    conf = dict(model_type="llava_bunny", auto_map={}, hidden_size=t.hidden_size, mm_hidden_size=v.hidden_size,
                num_hidden_layers=t.num_hidden_layers, intermediate_size=t.intermediate_size,
                num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
                rms_norm_eps=t.rms_norm_eps, vocab_size=t.vocab_size, rope_theta=t.rope_theta, attention_bias=True,
                tie_word_embeddings=True, eos_token_id=[1],
                vision_config=dict(num_hidden_layers=v.num_hidden_layers, hidden_size=v.hidden_size,
                                   intermediate_size=v.intermediate_size, num_attention_heads=v.num_attention_heads,
                                   image_size=v.image_size, patch_size=v.patch_size))
    (path / "config.json").write_text(json.dumps(conf))
    hf = {}
    for k, w in W.items():
        w = w.contiguous()
        if k.startswith("vision_tower."):
            if k.endswith("patch_embedding.weight"):
                w = w.permute(0, 3, 1, 2).contiguous()                   # (O, kH, kW, C) -> torch (O, C, kH, kW)
            hf["model." + k] = w
        elif k.startswith("mm_projector.linear_1."):
            hf["model.mm_projector.0." + k.rsplit(".", 1)[1]] = w
        elif k.startswith("mm_projector.linear_2."):
            hf["model.mm_projector.2." + k.rsplit(".", 1)[1]] = w
        else:
            hf[k[len("language_model."):]] = w                           # language_model.model.X -> model.X
    hf["model.layers.0.self_attn.rotary_emb.inv_freq"] = torch.zeros(32)  # dropped by sanitize (language.py:172-174)
    save_file(hf, str(path / "model.safetensors"))
    vocab = {"<unk>": 0, "<eos>": 1, "<pad>": 2, **{f"t{i}": i for i in range(3, t.vocab_size)}}
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<eos>", pad_token="<pad>", unk_token="<unk>").save_pretrained(str(path))


def test_load_and_generate_user_api_end_to_end(tmp_path):
    """load(path) dispatches on model_type to models/llava_bunny; generate() with an "<image>" prompt goes through the
    tokenizer split, the 384 x 384 image processor, the SigLIP tower, projector, splice, prefill and graph decode."""
    from mlx_vlm_amd import generate, load

    cfg = ob.tiny_cfg()
    W = {k: v.to(BF) for k, v in ob.random_weights(cfg, seed=4321, dtype=torch.float32, **ob.TEST_WEIGHT_SCALES).items()}
    write_bunny_checkpoint(tmp_path, cfg, W)
    model, processor = load(str(tmp_path), kv_pool_tokens=4096, max_seqs=4)
    assert type(model).__module__.endswith("llava_bunny.llava_bunny")
    img = G["img0.image_hwc"]
    ids = G["case0.input_ids"][0].tolist()
    cut = ids.index(cfg.image_token_index)
    prompt = " ".join(f"t{t}" for t in ids[:cut]) + " <image> " + " ".join(f"t{t}" for t in ids[cut + 1:])
    out = generate(model, processor, prompt, image=img, max_tokens=6, temperature=0.0)
    assert out.prompt_tokens == len(ids) and out.generation_tokens >= 1
    toks = [int(w[1:]) for w in out.text.split() if w.startswith("t")]
    ref_toks, ref_logits = ob.generate_greedy(W, cfg, np.array([ids]), _pixels(0), max_tokens=6, return_logits=True)
    ok, n = _tie_aware(toks, ref_toks[:len(toks)], ref_logits, tol=3e-2)
    assert ok and len(toks) >= 1, (toks, ref_toks, n)
