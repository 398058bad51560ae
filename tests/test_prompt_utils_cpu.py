"""`prompt_utils.apply_chat_template` of the product against the reference's own module (tests/golden/make_golden_prompts.py
ran /root/reference/mlx_vlm/prompt_utils.py here; only prompts_ref.json is read): message shapes per model family, image
allocation inside conversations, the template call, the no-template fallback, tool messages, the single-image rule."""
import json
import os

import pytest

from mlx_vlm_amd import prompt_utils as pu

HERE = os.path.dirname(os.path.abspath(__file__))
ROWS = json.load(open(os.path.join(HERE, "golden", "prompts_ref.json")))


def _stub(template="default"):
    import importlib.util

    spec = importlib.util.spec_from_file_location("mk_prompts", os.path.join(HERE, "golden", "make_golden_prompts.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.StubProcessor() if template == "default" else m.StubProcessor(template=None)


@pytest.mark.parametrize("row", ROWS, ids=[f"{r['model_type']}-{i}" for i, r in enumerate(ROWS)])
def test_apply_chat_template_vs_reference(row):
    cfg, prompt, n = {"model_type": row["model_type"]}, row["prompt"], row["num_images"]
    if "error" in row:
        with pytest.raises(Exception) as ei:
            pu.apply_chat_template(_stub(), cfg, prompt, return_messages=True, num_images=n)
        assert type(ei.value).__name__ == row["error"]
        return
    assert pu.apply_chat_template(_stub(), cfg, prompt, return_messages=True, num_images=n) == row["messages"]
    assert pu.apply_chat_template(_stub(), cfg, prompt, num_images=n) == row["with_template"]
    assert pu.apply_chat_template(_stub(), cfg, prompt, add_generation_prompt=False, num_images=n) == row["no_generation_prompt"]
    assert pu.apply_chat_template(_stub(None), cfg, prompt, num_images=n) == row["without_template"]


def test_audio_and_video_are_refused_not_dropped():
    with pytest.raises(NotImplementedError):
        pu.apply_chat_template(_stub(), {"model_type": "qwen2_vl"}, "x", num_audios=1)
    with pytest.raises(NotImplementedError):
        pu.apply_chat_template(_stub(), {"model_type": "qwen2_vl"}, "x", video="v.mp4")
