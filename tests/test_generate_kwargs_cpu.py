"""`GenerateKwargs` (reference generate/types.py:20-63) and the chunked-prefill hooks (generate/common.py:39-74)
on the host side."""
import pytest

# the reference's keys, in its order (generate/types.py:20-63)
REFERENCE_KEYS = ["max_tokens", "temperature", "repetition_penalty", "repetition_context_size", "presence_penalty",
                  "presence_context_size", "frequency_penalty", "frequency_context_size", "top_p", "min_p", "top_k", "logit_bias",
                  "prompt_cache", "max_kv_size", "kv_bits", "kv_key_bits", "kv_value_bits", "kv_key_scheme", "kv_value_scheme",
                  "kv_group_size", "kv_quant_scheme", "quantized_kv_start", "sampler", "logits_processors", "prefill_step_size",
                  "input_ids", "pixel_values", "mask", "resize_shape", "eos_tokens", "stopping_criteria", "thinking_budget",
                  "thinking_end_token", "thinking_start_token", "enable_thinking", "skip_special_tokens", "vision_cache",
                  "prompt_cache_state", "apc_manager", "apc_tenant", "seed", "verbose", "video"]


def test_generate_kwargs_has_the_references_keys_all_optional():
    from mlx_vlm_amd.generate import GenerateKwargs

    assert list(GenerateKwargs.__annotations__) == REFERENCE_KEYS
    assert GenerateKwargs.__required_keys__ == frozenset() and GenerateKwargs.__optional_keys__ == frozenset(REFERENCE_KEYS)
    kw: GenerateKwargs = {"max_tokens": 4, "temperature": 0.0}           # a TypedDict is a dict
    assert dict(**kw) == {"max_tokens": 4, "temperature": 0.0}


class _Obj:
    pass


def _model(**lm_attrs):
    m = _Obj()
    m.language_model = _Obj()
    for k, v in lm_attrs.items():
        setattr(m.language_model, k, v)
    return m


def test_chunked_prefill_enabled_truth_table():
    """generate/common.py:39-74: policy callable first (model, then language model; bool or `.enabled`), then the
    no_chunked_prefill veto on either, then `draft_model is None`"""
    from mlx_vlm_amd.generate import _chunked_prefill_enabled as en

    assert en(_model()) is True
    assert en(_model(), draft_model=object()) is False
    assert en(_model(no_chunked_prefill=True)) is False
    m = _model()
    m.no_chunked_prefill = True
    assert en(m) is False
    seen = {}

    def policy(**kw):
        seen.update(kw)
        return _Obj.__new__(type("P", (), {"enabled": False}))

    assert en(_model(chunked_prefill_policy=policy), input_ids=[1], prefill_kwargs={"a": 1}) is False
    assert set(seen) == {"input_ids", "inputs_embeds", "prompt_cache", "draft_model", "draft_kind", "prefill_kwargs"} and seen["prefill_kwargs"] == {"a": 1}
    # a policy wins over the veto and over a draft model; the model's policy is asked before the language model's
    m = _model(no_chunked_prefill=True, chunked_prefill_policy=lambda **kw: True)
    assert en(m, draft_model=object()) is True
    m.chunked_prefill_policy = lambda **kw: False
    assert en(m) is False
    # the language model IS the model: asked once
    lm = _Obj()
    lm.language_model = lm
    lm.no_chunked_prefill = True
    assert en(lm) is False
