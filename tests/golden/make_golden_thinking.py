"""Known answers of the reference's ThinkingBudgetCriteria (mlx_vlm/utils.py:2252-2335), produced by executing that class's
own source (it needs no mlx) on seeded token streams.

    python tests/golden/make_golden_thinking.py        # needs /root/reference; writes tests/golden/thinking_ref.json
"""
import json
import os
import random
from typing import List, Optional

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/mlx_vlm/utils.py"


class Tok:
    def encode(self, t, add_special_tokens=False):
        return {"<think>": [5, 7], "</think>": [8], "\n": [9]}.get(t, [1])


def main():
    src = open(SRC).read()
    ns = {"List": List, "Optional": Optional}
    exec(src[src.index("class ThinkingBudgetCriteria:"):src.index("def print_array_report")], ns)
    Ref = ns["ThinkingBudgetCriteria"]
    rng = random.Random(1234)
    cases = []
    for _ in range(60):
        kw = dict(thinking_budget=rng.randint(0, 6), thinking_end_token="</think>", thinking_start_token="<think>",
                  enable_thinking=rng.random() < 0.8, prompt_preopens_thinking=rng.random() < 0.5)
        c = Ref(Tok(), **kw)
        toks, pops, trace = [], [], []
        for i in range(48):
            t = rng.choice([7, 8, 9, 1, 2, 3, 3, 3])
            pop = rng.random() < 0.7
            ret = c(t)
            popped = c.pop_forced_token_id() if pop else "-"
            toks.append(t)
            pops.append(pop)
            trace.append([ret, popped, c.in_thinking, c.thinking_token_count, c.budget_exceeded, c.forced_token_id])
            if i == 30:
                c.reset_thinking_state()
        cases.append(dict(kw=kw, tokens=toks, pops=pops, trace=trace))
    with open(os.path.join(HERE, "thinking_ref.json"), "w") as f:
        json.dump(cases, f)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
