"""Golden prompt formats produced by the REFERENCE'S OWN `mlx_vlm/prompt_utils.py` (stdlib only: it runs here without `mlx`).

    python tests/golden/make_golden_prompts.py      # needs /root/reference; writes tests/golden/prompts_ref.json

Every case = (model_type, prompt form, num_images, template on / off) -> the messages (`return_messages=True`) and the final
string of `apply_chat_template`, rendered through the stub processor below (a Jinja template that prints roles, text items and
image items, so that both the message shapes and the template call are visible in the output)."""
import importlib.util
import json
import os

REF = os.environ.get("VLM_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

TEMPLATE = ("{% for m in messages %}[{{ m['role'] }}]{% if m['content'] is string %}{{ m['content'] }}{% else %}"
            "{% for it in m['content'] %}{% if it['type'] == 'image' %}<IMG>{% else %}{{ it['text'] }}{% endif %}{% endfor %}"
            "{% endif %}{% if m.get('tool_calls') %}<CALLS:{{ m['tool_calls'] | tojson }}>{% endif %}\n{% endfor %}"
            "{% if add_generation_prompt %}[assistant]{% endif %}{% if enable_thinking is defined %}<think={{ enable_thinking }}>{% endif %}")


class StubProcessor:
    def __init__(self, template=TEMPLATE):
        self.chat_template = template

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=False, **kwargs):
        from jinja2 import Template

        return Template(self.chat_template).render(messages=messages, add_generation_prompt=add_generation_prompt, **kwargs)


def cases():
    conv = [{"role": "system", "content": "be brief"},
            {"role": "user", "content": [{"type": "text", "text": "first"}, {"type": "image_url", "image_url": {"url": "x"}}]},
            {"role": "assistant", "content": "ok"},
            {"role": "user", "content": "and this one?"}]
    tool = [{"role": "user", "content": "weather?"},
            {"role": "assistant", "content": None, "tool_calls": [{"id": "1", "function": {"name": "w", "arguments": "{\"city\": \"x\"}"}}]},
            {"role": "tool", "tool_call_id": "1", "content": "sunny"}]
    out = []
    for mt in ("qwen2_vl", "idefics2", "llava-qwen2", "bunny-llama", "phi3_v", "some_text_model"):
        for n in (0, 1, 2):
            if n == 2 and mt in ("llava-qwen2", "bunny-llama"):
                continue
            out.append((mt, "describe it", n))
            out.append((mt, {"role": "user", "content": [{"type": "text", "text": "look"}, {"type": "text", "text": "closely"}]}, n))
            out.append((mt, conv, n))
            out.append((mt, ["one", "two"], n))
        out.append((mt, tool, 0))
    return out


def main():
    spec = importlib.util.spec_from_file_location("ref_prompt_utils", os.path.join(REF, "mlx_vlm", "prompt_utils.py"))
    pu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pu)
    assert pu.__file__.startswith(REF)
    rows = []
    for mt, prompt, n in cases():
        cfg = {"model_type": mt}
        row = {"model_type": mt, "prompt": prompt, "num_images": n}
        try:
            row["messages"] = pu.apply_chat_template(StubProcessor(), cfg, prompt, return_messages=True, num_images=n)
            row["with_template"] = pu.apply_chat_template(StubProcessor(), cfg, prompt, num_images=n)
            row["no_generation_prompt"] = pu.apply_chat_template(StubProcessor(), cfg, prompt, add_generation_prompt=False, num_images=n)
            row["without_template"] = pu.apply_chat_template(StubProcessor(template=None), cfg, prompt, num_images=n)
        except Exception as e:      # recorded: the product must raise the same kind of error
            row["error"] = type(e).__name__
        rows.append(row)
    # the single-image rule
    for mt in ("llava-qwen2", "bunny-llama"):
        try:
            pu.apply_chat_template(StubProcessor(), {"model_type": mt}, "x", num_images=2)
            rows.append({"model_type": mt, "prompt": "x", "num_images": 2, "messages": "no error"})
        except Exception as e:
            rows.append({"model_type": mt, "prompt": "x", "num_images": 2, "error": type(e).__name__})
    path = os.path.join(HERE, "prompts_ref.json")
    json.dump(rows, open(path, "w"), indent=0)
    print("wrote", path, len(rows), "cases", sum("error" in r for r in rows), "errors")


if __name__ == "__main__":
    main()
